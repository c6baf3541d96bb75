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

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
    m.doc() = "compiled host binding of libspfsplat_hip.so's rasterizer entry points (no arithmetic of its own)";
    m.def("abi_version", []() { return spf_abi_version(); });
    m.def("raster_forward", &raster_forward);
    m.def("raster_backward", &raster_backward);
}
