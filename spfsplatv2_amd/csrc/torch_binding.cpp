// Compiled host binding of the rasterizer's C ABI (include/spfsplat_hip.h) for PyTorch callers.
//
// The reference's rasterizer package is a compiled torch extension whose Python side only packs arguments
// (/root/reference/src/model/decoder/cuda_splatting.py:5,124-138 imports and calls it).  The ctypes binding in
// rasterizer.py does the same job in Python -- about 0.3 ms of interpreter time per call and direction (eleven
// torch.empty, ten tensor slices, three ctypes structs), which is what a per-view caller of the drop-in surface pays
// b*v times per step.  This file is that host logic once more in C++: workspace allocation through ATen's caching
// allocator, the C-ABI structs, the launch chain on torch's current stream, the 16-byte read-back of exact mode.
// It owns no arithmetic and no device code; rasterizer.py uses it when it has been built and keeps its own path
// otherwise (same library, same kernels, same results).
//
// Mirrors rasterizer.py::_forward_impl (camera=None) and ::_backward_impl statement for statement.
#include <torch/extension.h>

#include <c10/core/DeviceGuard.h>
#include <c10/hip/HIPStream.h>
#include <hip/hip_runtime_api.h>

#include <optional>
#include <stdexcept>
#include <string>
#include <vector>

#include "spfsplat_hip.h"

namespace {

using at::Tensor;
using OptTensor = std::optional<Tensor>;

template <typename T>
T* ptr(const Tensor& t) {
    return t.defined() ? static_cast<T*>(t.data_ptr()) : nullptr;
}
template <typename T>
T* ptr(const OptTensor& t) {
    return (t.has_value() && t->defined()) ? static_cast<T*>(t->data_ptr()) : nullptr;
}

// library failures travel as RuntimeError("SPF: ...") and become spfsplatv2_amd._lib.SpfError on the Python side
void check(int rc, const char* what) {
    if (rc != 0) throw std::runtime_error(std::string("SPF: ") + what + " failed (code " + std::to_string(rc) + "): " + spf_last_error());
}

SpfInputs make_inputs(const Tensor& means3D, const Tensor& scales, const Tensor& rotations, const Tensor& opacities,
                      const OptTensor& shs, const OptTensor& colors, const Tensor& viewmatrix, const Tensor& projmatrix,
                      const Tensor& tanfov, const Tensor& bg, const OptTensor& view_scale, const OptTensor& view64) {
    SpfInputs in;
    in.means3D = ptr<const float>(means3D); in.scales = ptr<const float>(scales);
    in.rotations = ptr<const float>(rotations); in.opacities = ptr<const float>(opacities);
    in.shs = ptr<const float>(shs); in.colors = ptr<const float>(colors);
    in.viewmatrix = ptr<const float>(viewmatrix); in.projmatrix = ptr<const float>(projmatrix);
    in.tanfov = ptr<const float>(tanfov); in.bg = ptr<const float>(bg);
    in.view_scale = ptr<const float>(view_scale); in.viewmatrix64 = ptr<const double>(view64);
    in.shs_high = nullptr; in.raw = nullptr; in.sh_mask = nullptr;   // (band-split / raw rows are the batched decoder's: this per-view surface takes [G,K,3])
    return in;
}

// layout of the state tensors as rasterizer.py allocates them (see _state_struct there)
SpfState make_state(const Tensor& rec, const Tensor& radii, const Tensor& rect, const Tensor& tiles, const Tensor& pairs,
                    const Tensor& pair_idx, const Tensor& final_T, const Tensor& n_contrib, int64_t RT, int64_t RG,
                    int64_t RB) {
    SpfState st;
    uint32_t* const t = ptr<uint32_t>(tiles);
    uint32_t* const pi = ptr<uint32_t>(pair_idx);
    st.rec = ptr<float>(rec); st.radii = ptr<int32_t>(radii);
    st.rect = ptr<uint32_t>(rect); st.zkey = reinterpret_cast<float*>(ptr<uint32_t>(rect) + RG);
    st.tile_count = t; st.tile_flags = t + RT; st.tile_start = t + 2 * RT; st.tile_fill = t + 3 * RT + 1;
    st.counters = t + 4 * RT + 1;
    st.pair_cursor = tiles.numel() >= 4 * RT + 13 ? t + 4 * RT + 5 : nullptr;
    st.pairs = pairs.defined() ? reinterpret_cast<uint64_t*>(pairs.data_ptr()) : nullptr;
    st.pair_off = pi; st.blk_total = pi + 2 * RG; st.blk_base = pi + 2 * RG + RB;      // pair_off: (rect, first pair) per (render, Gaussian)
    st.final_T = ptr<float>(final_T); st.n_contrib = ptr<uint32_t>(n_contrib);
    st.sh_clamp = rect.numel() > 2 * RG ? reinterpret_cast<uint8_t*>(ptr<uint32_t>(rect) + 2 * RG) : nullptr;   // SH clamp masks ride behind rect | zkey
    st.verdict_host = nullptr;
    return st;
}

SpfDims make_dims(int64_t S, int64_t V, int64_t G, int64_t K, int64_t sh_degree, int64_t H, int64_t W,
                  double scale_modifier, int64_t sh_layout, bool sh_band4, int64_t bin_cap = 0, int64_t pair_capacity = 0) {
    SpfDims d;
    d.S = (int32_t)S; d.V = (int32_t)V; d.G = (int32_t)G; d.K = (int32_t)K; d.sh_degree = (int32_t)sh_degree;
    d.H = (int32_t)H; d.W = (int32_t)W; d.scale_modifier = (float)scale_modifier; d.sh_layout = (int32_t)sh_layout;
    d.sh_band4 = sh_band4 ? 1 : 0;
    d.bin_cap = (int32_t)bin_cap; d.pair_capacity = bin_cap ? pair_capacity : 0; d.raw_stride = 0; d.adapter_eps = 0.f;
    return d;
}

// the callers (rasterizer.py) have validated shapes, dtypes and contiguity; what must never reach a launch is a host pointer
void require_device(const Tensor& t, const char* name) {
    TORCH_CHECK(t.defined() && t.is_cuda(), name, " is not on a HIP device: the rasterizer only runs on a HIP device (there is no CPU fallback)");
    TORCH_CHECK(t.is_contiguous(), name, " must be contiguous");
}

// Forward chain without the fused camera kernel.  capacity < 0: exact mode (one 16-byte read-back sizes the pair
// buffer); otherwise the caller's plan (capacity, max_tile, dense as spf_raster_forward_render takes them).
// Returns ([image, depth, alpha, radii, rec, rect, tiles, pairs, pair_idx, final_T, n_contrib], [D, max_tile, dense, R*T]);
// D = -1 in planned mode.
std::tuple<std::vector<Tensor>, std::vector<int64_t>> raster_forward(
    const Tensor& means3D, const Tensor& scales, const Tensor& rotations, const Tensor& opacities, const OptTensor& shs,
    const OptTensor& colors, const Tensor& viewmatrix, const Tensor& projmatrix, const Tensor& tanfov, const Tensor& bg,
    const OptTensor& view_scale, const OptTensor& view64, int64_t H, int64_t W, int64_t sh_degree, double scale_modifier,
    int64_t sh_layout, bool sh_band4, int64_t capacity, int64_t max_tile, int64_t dense) {
    require_device(means3D, "means3D"); require_device(scales, "scales"); require_device(rotations, "rotations");
    require_device(opacities, "opacities"); require_device(viewmatrix, "viewmatrix"); require_device(projmatrix, "projmatrix");
    require_device(tanfov, "tanfov"); require_device(bg, "bg");
    if (shs.has_value() && shs->defined()) require_device(*shs, "shs");
    if (colors.has_value() && colors->defined()) require_device(*colors, "colors");
    const c10::DeviceGuard guard(means3D.device());   // (ROCm torch reports its devices as "cuda": the generic guard)
    const int64_t S = means3D.size(0), G = means3D.size(1), V = viewmatrix.size(1), R = S * V;
    const bool have_sh = shs.has_value() && shs->defined();
    const int64_t K = have_sh ? shs->size(sh_layout ? 3 : 2) : 0;
    const SpfDims dims = make_dims(S, V, G, K, sh_degree, H, W, scale_modifier, sh_layout, sh_band4);
    const int64_t T = spf_raster_num_tiles((int32_t)H, (int32_t)W), P = H * W, RT = R * T, RG = R * G;
    const int64_t nblk = spf_raster_view_partial_blocks((int32_t)G), RB = R * nblk;
    const auto i32 = means3D.options().dtype(at::kInt), f32 = means3D.options().dtype(at::kFloat);

    Tensor rec = at::empty({RG, 12}, f32), radii = at::empty({RG}, i32), rect = at::empty({2 * RG + (RG + 3) / 4}, i32);
    Tensor pair_idx = at::empty({2 * RG + 2 * RB}, i32), tiles = at::empty({4 * RT + 16}, i32);
    Tensor final_T = at::empty({R * P}, f32), n_contrib = at::empty({R * P}, i32);
    Tensor image = at::empty({S, V, 3, H, W}, f32), depth = at::empty({S, V, 1, H, W}, f32),
           alpha = at::empty({S, V, 1, H, W}, f32);

    const SpfInputs in = make_inputs(means3D, scales, rotations, opacities, shs, colors, viewmatrix, projmatrix, tanfov,
                                     bg, view_scale, view64);
    SpfState st = make_state(rec, radii, rect, tiles, Tensor(), pair_idx, final_T, n_contrib, RT, RG, RB);
    void* const stream = c10::hip::getCurrentHIPStream(means3D.device().index()).stream();
    check(spf_raster_forward_project(&dims, &in, &st, stream), "spf_raster_forward_project");

    int64_t D = -1;
    if (capacity < 0) {
        const Tensor host = tiles.narrow(0, 4 * RT + 1, 4).to(at::kCPU);       // D, longest list, verdict, dense tiles
        const int32_t* h = host.data_ptr<int32_t>();
        D = (int64_t)(uint32_t)h[0]; max_tile = (int64_t)(uint32_t)h[1]; dense = (int64_t)(uint32_t)h[3];
        capacity = D;
    }
    Tensor pairs = at::empty({capacity > 1 ? capacity : 1}, means3D.options().dtype(at::kLong));
    st.pairs = reinterpret_cast<uint64_t*>(pairs.data_ptr());
    SpfOutputs out{ptr<float>(image), ptr<float>(depth), ptr<float>(alpha)};
    check(spf_raster_forward_render(&dims, &in, &st, &out, (uint64_t)capacity, (uint32_t)max_tile, (uint32_t)dense, stream),
          "spf_raster_forward_render");
    return {{image, depth, alpha, radii.view({S, V, G}), rec, rect, tiles, pairs, pair_idx, final_T, n_contrib},
            {D, max_tile, dense, RT}};
}

// Backward chain.  want_view: 0 = none, 1 = dL/dviewmatrix, 2 = per-block partial sums in its place.
// Returns [d_means, d_scales, d_rot, d_opac, d_shs, d_col, d_view | vpartial, d_means2D]; undefined = None.
std::vector<Tensor> raster_backward(
    const Tensor& means3D, const Tensor& scales, const Tensor& rotations, const Tensor& opacities, const OptTensor& shs,
    const OptTensor& colors, const Tensor& viewmatrix, const Tensor& projmatrix, const Tensor& tanfov, const Tensor& bg,
    const OptTensor& view_scale, const OptTensor& view64, const Tensor& rec, const Tensor& radii, const Tensor& rect,
    const Tensor& tiles, const Tensor& pairs, const Tensor& pair_idx, const Tensor& final_T, const Tensor& n_contrib,
    int64_t H, int64_t W, int64_t sh_degree, double scale_modifier, int64_t sh_layout, bool sh_band4, int64_t dense,
    int64_t bin_cap, int64_t capacity, const OptTensor& g_image, const OptTensor& g_depth, const OptTensor& g_alpha, bool want_scales_rot, bool want_shs,
    bool want_colors, int64_t want_view, bool want_means2D) {
    require_device(means3D, "means3D"); require_device(rec, "rec"); require_device(pairs, "pairs");
    const c10::DeviceGuard guard(means3D.device());   // (ROCm torch reports its devices as "cuda": the generic guard)
    const int64_t S = means3D.size(0), G = means3D.size(1), V = viewmatrix.size(1), R = S * V;
    const bool have_sh = shs.has_value() && shs->defined(), have_col = colors.has_value() && colors->defined();
    const int64_t K = have_sh ? shs->size(sh_layout ? 3 : 2) : 0;
    // `capacity`: gradient records (= the forward's pair capacity; with direct bins `pairs` holds the bins instead)
    const SpfDims dims = make_dims(S, V, G, K, sh_degree, H, W, scale_modifier, sh_layout, sh_band4, bin_cap, capacity);
    const int64_t T = spf_raster_num_tiles((int32_t)H, (int32_t)W), RT = R * T, RG = R * G;
    const int64_t nblk = spf_raster_view_partial_blocks((int32_t)G), RB = R * nblk;
    const auto f32 = means3D.options().dtype(at::kFloat);
    auto grad_in = [&](const OptTensor& g) -> Tensor {
        if (!g.has_value() || !g->defined()) return Tensor();
        return g->contiguous().to(at::kFloat);
    };
    const Tensor gi = grad_in(g_image), gd = grad_in(g_depth), ga = grad_in(g_alpha);
    Tensor gpair = at::empty({capacity, 10}, f32);
    Tensor d_means = at::empty_like(means3D), d_opac = at::empty_like(opacities);
    Tensor d_scales, d_rot, d_shs, d_col, d_view, vpartial, d_m2d;
    if (want_scales_rot) { d_scales = at::empty_like(scales); d_rot = at::empty_like(rotations); }
    if (have_sh && want_shs) d_shs = at::empty_like(*shs);
    if (have_col && want_colors) d_col = at::empty_like(*colors);
    if (want_view == 1) d_view = at::empty_like(viewmatrix);
    if (want_view != 0) vpartial = at::empty({R, nblk, 12}, f32);
    if (want_means2D) d_m2d = at::zeros({R, G, 3}, f32);

    const SpfInputs in = make_inputs(means3D, scales, rotations, opacities, shs, colors, viewmatrix, projmatrix, tanfov,
                                     bg, view_scale, view64);
    const SpfState st = make_state(rec, radii, rect, tiles, pairs, pair_idx, final_T, n_contrib, RT, RG, RB);
    SpfGrads gr;
    gr.dL_dimage = ptr<const float>(gi); gr.dL_ddepth = ptr<const float>(gd); gr.dL_dalpha = ptr<const float>(ga);
    gr.gpair = ptr<float>(gpair); gr.vpartial = ptr<float>(vpartial);
    gr.dL_dmeans3D = ptr<float>(d_means); gr.dL_dscales = ptr<float>(d_scales); gr.dL_drotations = ptr<float>(d_rot);
    gr.dL_dopacities = ptr<float>(d_opac); gr.dL_dshs = ptr<float>(d_shs); gr.dL_dcolors = ptr<float>(d_col);
    gr.dL_dviewmatrix = ptr<float>(d_view); gr.dL_dmeans2D = ptr<float>(d_m2d); gr.dL_dshs_high = nullptr; gr.dL_draw = nullptr;
    void* const stream = c10::hip::getCurrentHIPStream(means3D.device().index()).stream();
    check(spf_raster_backward(&dims, &in, &st, &gr, (uint64_t)capacity, (uint32_t)dense, stream), "spf_raster_backward");
    return {d_means, d_scales, d_rot, d_opac, d_shs, d_col, want_view == 2 ? vpartial : d_view, d_m2d};
}

// ---- a PREPARED training step (rasterizer.py::StaticStep), driven from C++ ---------------------------------------
// The decoder module's training calls whose shapes repeat run on state at fixed addresses with argument structs built
// once (rasterizer.py::StaticStep builds them; this class takes them over by value).  What a call still costs the host in
// Python -- three output and five gradient allocations, a dozen pointer fields, five ctypes calls, an event -- is ~75 us
// per step on a loaded host, where the whole step is host-bound (0.42 ms against 0.35 on the GPU); here it is two
// calls.  Same C ABI, same kernels, same results; StaticStep keeps its Python path for when this file is not built.
class PreparedStep {
 public:
    PreparedStep(uintptr_t dims, uintptr_t inp, uintptr_t st, uintptr_t cam, uintptr_t cam_b, uintptr_t gr,
                 const Tensor& tiles, int64_t capacity, int64_t max_tile, int64_t nblk, const Tensor& verdict,
                 const Tensor& view, const OptTensor& vpartial, bool want_scales_rot, bool want_shs, bool want_high,
                 bool want_view)
        : tiles_(tiles), verdict_(verdict), view_(view), capacity_((uint64_t)capacity), max_tile_((uint32_t)max_tile),
          nblk_((int32_t)nblk), want_scales_rot_(want_scales_rot), want_shs_(want_shs), want_high_(want_high),
          want_view_(want_view) {
        dims_ = *reinterpret_cast<const SpfDims*>(dims); in_ = *reinterpret_cast<const SpfInputs*>(inp);
        st_ = *reinterpret_cast<const SpfState*>(st); cam_ = *reinterpret_cast<const SpfCamera*>(cam);
        cam_b_ = *reinterpret_cast<const SpfCamera*>(cam_b); gr_ = *reinterpret_cast<const SpfGrads*>(gr);
        if (vpartial.has_value() && vpartial->defined()) vpartial_ = *vpartial;
        TORCH_CHECK(tiles.is_cuda() && view.is_cuda() && !verdict.is_cuda() && verdict.is_pinned(), "PreparedStep: state on the device, verdict word in pinned host memory");
        const c10::DeviceGuard guard(view_.device());
        TORCH_CHECK(hipEventCreateWithFlags(&event_, hipEventDisableTiming) == hipSuccess, "PreparedStep: hipEventCreate failed");
    }
    ~PreparedStep() {
        if (event_) (void)hipEventDestroy(event_);
    }
    PreparedStep(const PreparedStep&) = delete;
    PreparedStep& operator=(const PreparedStep&) = delete;

    // this call's inputs (shapes, dtype, contiguity: the caller's key); non-leaf tensors are held as detached aliases
    void bind(const Tensor& extrinsics, const Tensor& intrinsics, const Tensor& near, const Tensor& far, const Tensor& means,
              const Tensor& scales, const Tensor& rotations, const Tensor& opacities, const Tensor& shs,
              const OptTensor& shs_high) {
        auto keep = [](const Tensor& t) { return t.grad_fn() ? t.detach() : t; };
        held_ = {keep(extrinsics), keep(intrinsics), keep(near), keep(far), keep(means), keep(scales), keep(rotations),
                 keep(opacities), keep(shs), (shs_high.has_value() && shs_high->defined()) ? keep(*shs_high) : Tensor()};
        for (size_t i = 0; i < 9; ++i) require_device(held_[i], "a decoder input");
        cam_.extrinsics = ptr<const float>(held_[0]); cam_.intrinsics = ptr<const float>(held_[1]);
        cam_.near = ptr<const float>(held_[2]); cam_.far = ptr<const float>(held_[3]);
        cam_b_.near = cam_.near;
        in_.means3D = ptr<const float>(held_[4]); in_.scales = ptr<const float>(held_[5]);
        in_.rotations = ptr<const float>(held_[6]); in_.opacities = ptr<const float>(held_[7]);
        in_.shs = ptr<const float>(held_[8]); in_.shs_high = ptr<const float>(held_[9]);
        near_b_ = cam_.scale_invariant ? held_[2].view({dims_.S, dims_.V, 1, 1}) : Tensor();
    }

    // [extrinsics, intrinsics, near, far, means, scales, rotations, opacities, shs, shs_high | None] of the current binding
    std::vector<Tensor> held() const { return held_; }

    // camera + projection + bins, sort, compositing into fresh outputs; early: wait for the projection's verdict (the GPU
    // works on through the wait).  Returns (colour [S,V,3,H,W], depth [S,V,H,W] x near when scale-invariant, alpha, failed).
    std::tuple<Tensor, Tensor, Tensor, bool> forward(bool early) {
        const c10::DeviceGuard guard(view_.device());
        void* const stream = c10::hip::getCurrentHIPStream(view_.device().index()).stream();
        int32_t* const word = verdict_.data_ptr<int32_t>();
        if (early) *word = 0;                      // (host memory: the projection kernel stores here if it raises a flag)
        const uint64_t tile_bytes = 4 * (uint64_t)tiles_.numel();
        check(spf_decoder_prepare(&cam_, tiles_.data_ptr(), tile_bytes, stream), "spf_decoder_prepare");
        check(spf_raster_forward_project_prepared(&dims_, &in_, &st_, tile_bytes, stream), "spf_raster_forward_project_prepared");
        if (early) TORCH_CHECK(hipEventRecord(event_, static_cast<hipStream_t>(stream)) == hipSuccess, "hipEventRecord failed");
        const auto f32 = view_.options();
        const int64_t S = dims_.S, V = dims_.V, H = dims_.H, W = dims_.W;
        Tensor color = at::empty({S, V, 3, H, W}, f32), depth = at::empty({S, V, H, W}, f32), alpha = at::empty({S, V, 1, H, W}, f32);
        SpfOutputs out{ptr<float>(color), ptr<float>(depth), ptr<float>(alpha)};
        check(spf_raster_forward_render(&dims_, &in_, &st_, &out, capacity_, max_tile_, 0xFFFFFFFFu, stream), "spf_raster_forward_render");
        if (near_b_.defined()) depth.mul_(near_b_);
        bool failed = false;
        if (early) {
            pybind11::gil_scoped_release nogil;
            TORCH_CHECK(hipEventSynchronize(event_) == hipSuccess, "hipEventSynchronize failed");
            failed = *static_cast<volatile int32_t*>(word) != 0;
        }
        return {color, depth, alpha, failed};
    }

    // the whole backward chain into fresh gradients: [means, opacities, scales, rotations, harmonics, harmonics_band4,
    // extrinsics]; undefined = None.  g_depth is the gradient of the [S,V,H,W] depth output (x near is undone here).
    std::vector<Tensor> backward(const OptTensor& g_image, const OptTensor& g_depth, const OptTensor& g_alpha) {
        const c10::DeviceGuard guard(view_.device());
        void* const stream = c10::hip::getCurrentHIPStream(view_.device().index()).stream();
        auto grad_in = [&](const OptTensor& g) -> Tensor {
            if (!g.has_value() || !g->defined()) return Tensor();
            return g->contiguous().to(at::kFloat);
        };
        const Tensor gi = grad_in(g_image), ga = grad_in(g_alpha);
        Tensor gd = grad_in(g_depth);
        if (gd.defined() && near_b_.defined()) gd = gd * near_b_;
        Tensor d_means = at::empty_like(held_[4]), d_opac = at::empty_like(held_[7]);
        Tensor d_scales, d_rot, d_shs, d_high, d_ext;
        if (want_scales_rot_) { d_scales = at::empty_like(held_[5]); d_rot = at::empty_like(held_[6]); }
        if (want_shs_) d_shs = at::empty_like(held_[8]);
        if (want_high_) d_high = at::empty_like(held_[9]);
        if (want_view_) d_ext = at::empty_like(view_);
        SpfGrads gr = gr_;
        gr.dL_dimage = ptr<const float>(gi); gr.dL_ddepth = ptr<const float>(gd); gr.dL_dalpha = ptr<const float>(ga);
        gr.dL_dmeans3D = ptr<float>(d_means); gr.dL_dopacities = ptr<float>(d_opac); gr.dL_dscales = ptr<float>(d_scales);
        gr.dL_drotations = ptr<float>(d_rot); gr.dL_dshs = ptr<float>(d_shs); gr.dL_dshs_high = ptr<float>(d_high);
        check(spf_raster_backward(&dims_, &in_, &st_, &gr, capacity_, 0xFFFFFFFFu, stream), "spf_raster_backward");
        if (want_view_)
            check(spf_camera_backward_partials(&cam_b_, ptr<const float>(vpartial_), nblk_, ptr<float>(d_ext), stream),
                  "spf_camera_backward_partials");
        return {d_means, d_opac, d_scales, d_rot, d_shs, d_high, d_ext};
    }

 private:
    SpfDims dims_; SpfInputs in_; SpfState st_; SpfCamera cam_, cam_b_; SpfGrads gr_;
    Tensor tiles_, verdict_, view_, vpartial_, near_b_;
    std::vector<Tensor> held_;
    uint64_t capacity_; uint32_t max_tile_; int32_t nblk_;
    bool want_scales_rot_, want_shs_, want_high_, want_view_;
    hipEvent_t event_ = nullptr;
};

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
    pybind11::class_<PreparedStep>(m, "PreparedStep")
        .def(pybind11::init<uintptr_t, uintptr_t, uintptr_t, uintptr_t, uintptr_t, uintptr_t, const Tensor&, int64_t, int64_t,
                            int64_t, const Tensor&, const Tensor&, const OptTensor&, bool, bool, bool, bool>())
        .def("bind", &PreparedStep::bind)
        .def("held", &PreparedStep::held)
        .def("forward", &PreparedStep::forward)
        .def("backward", &PreparedStep::backward);
    m.doc() = "compiled host binding of libspfsplat_hip.so's rasterizer entry points (no arithmetic of its own)";
    m.def("abi_version", []() { return spf_abi_version(); });
    m.def("raster_forward", &raster_forward);
    m.def("raster_backward", &raster_backward);
}
