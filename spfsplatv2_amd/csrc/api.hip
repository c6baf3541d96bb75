// extern "C" entry points of libspfsplat_hip.so (see include/spfsplat_hip.h).
// Validation + launch sequencing only; no device allocation, no device synchronisation.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include "spf_common.h"

namespace spf {
hipError_t launch_project_fwd(const SpfDims&, const SpfInputs&, const SpfState&, int, int, hipStream_t);
hipError_t launch_project_bwd(const SpfDims&, const SpfInputs&, const SpfState&, const SpfGrads&, int, uint64_t, hipStream_t);
hipError_t launch_tile_scan(const SpfState&, int, int, int, uint32_t, bool, hipStream_t);
uint32_t dense_threshold();
hipError_t launch_bin_pairs(const SpfDims&, const SpfState&, uint64_t, int, int, uint32_t, hipStream_t);
hipError_t launch_tile_sort(const SpfState&, const TileLists&, int, int, uint64_t, uint32_t, const uint2*, int,
                            hipStream_t);
hipError_t launch_render_fwd(const SpfDims&, const SpfInputs&, const SpfState&, const SpfOutputs&, uint64_t, int, int,
                             bool, hipStream_t);
hipError_t launch_render_bwd(const SpfDims&, const SpfInputs&, const SpfState&, const SpfGrads&, int, int, uint64_t, bool,
                             hipStream_t);
hipError_t launch_adapter_fwd(const float*, int64_t, int64_t, int, const float*, float, float*, float*, float*, float*,
                              hipStream_t);
hipError_t launch_adapter_bwd(const float*, int64_t, int64_t, int, const float*, float, const float*, const float*,
                              const float*, const float*, int, float*, hipStream_t);
int mse_partial_blocks();
hipError_t launch_mse_fwd(const float*, const float*, int64_t, float, float*, float*, float, float*, hipStream_t);
hipError_t launch_mse_scale(float*, int64_t, const float*, hipStream_t);
hipError_t launch_mse_bwd(const float*, const float*, int64_t, float, const float*, float*, hipStream_t);
hipError_t launch_camera_fwd(const SpfCamera&, hipStream_t);
hipError_t launch_camera_bwd(const SpfCamera&, const float*, float*, hipStream_t);
hipError_t launch_camera_fwd_zero(const SpfCamera&, void*, uint64_t, hipStream_t);
hipError_t launch_camera_bwd_reduce(const SpfCamera&, const float*, int, float*, hipStream_t);
hipError_t launch_rope2d(void*, void*, const int64_t*, int, int, int, int, int64_t, int64_t, int64_t, int, int, float,
                         float, hipStream_t);
}  // namespace spf

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

#define SPF_HIP(expr)                                                                         \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess) return fail(SPF_E_LAUNCH, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

// ---- stage timing --------------------------------------------------------------------------
struct StageLog {
    hipEvent_t ev[SPF_STAGE_LOG][2];
    int created = 0;
    int used = 0;
};
StageLog g_log[SPF_STAGE_COUNT];
uint32_t g_timing = 0;   // bit i: record stage i
int g_sample_every = 1;  // record every n-th launch of an enabled stage (an event pair costs ~11 us of idle GPU)
int g_calls[SPF_STAGE_COUNT] = {};

struct StageScope {
    int stage;
    hipStream_t stream;
    int slot = -1;
    StageScope(int st, hipStream_t s) : stage(st), stream(s) {
        if (!((g_timing >> stage) & 1u)) return;
        // (a stream that is being captured launches nothing now: an event recorded here would become a graph node and
        //  could never be timed -- a caller that captures while stage timing is on simply gets no sample)
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(s, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) return;
        if (g_calls[stage]++ % g_sample_every != 0) return;
        StageLog& L = g_log[stage];
        if (L.used >= SPF_STAGE_LOG) return;
        if (L.used >= L.created) {
            if (hipEventCreate(&L.ev[L.created][0]) != hipSuccess) return;
            if (hipEventCreate(&L.ev[L.created][1]) != hipSuccess) return;
            L.created++;
        }
        slot = L.used;
        (void)hipEventRecord(L.ev[slot][0], stream);
    }
    ~StageScope() {
        if (slot < 0) return;
        (void)hipEventRecord(g_log[stage].ev[slot][1], stream);
        g_log[stage].used = slot + 1;
    }
};


// ---- two lanes: one batched call as several chunks of renders on two streams (OFF by default) -----------------
// Renders are independent, so after the joint tile scan the rest of the forward (bin -> sort -> composite) and the whole
// backward (composite backward -> projection backward) can run as C chunks of whole scenes alternating between the
// caller's stream and one auxiliary stream, the second lane one kernel behind the first (fork / join by events,
// capturable in a HIP graph; every buffer is render-major, so a chunk is the same launcher on offset pointers and the
// results are bit-identical to the single chain: tests/test_gpu_configs.py).  The idea: a latency-bound kernel of one
// chunk fills the launch gaps and tails of the other chunk's compositing kernel, as two whole-step micro-batches on two
// streams do (bench.py --streams 2: +13 %).  MEASURED (C2, 8 x 4 renders, HIP-graph replay, same box, ms per step):
// 1 chain 0.442 / 0.443, 2 chunks 0.472 / 0.464, 4 chunks 0.514 / 0.508, 8 chunks 0.578 / 0.568 -- it LOSES.  A call must
// hand complete outputs to the caller's stream, so every forward and every backward ends in a join at which both lanes
// drain, half-size launches have twice the tail, and the cross-queue event edges cost as much as the dependent
// launches they were meant to hide; free-running micro-batches never join.  SPF_CHUNKS=n (n > 1) enables it for
// experiments; the default is the single chain, whose per-kernel timings are exclusive.
constexpr int kMaxChunks = 8;
struct LaneSet {
    hipStream_t s = nullptr;
    hipEvent_t stagger = nullptr, join = nullptr;
};
LaneSet* lane_set() {
    // per (device, calling thread): see render.hip::aux_stream.  One stream and two events per pair, created on first
    // use and kept for the life of the process (SPF_CHUNKS experiments only; nothing is created by default)
    static thread_local LaneSet lanes[32];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 32) return nullptr;
    LaneSet& a = lanes[dev];
    if (!a.s) {
        if (hipStreamCreateWithFlags(&a.s, hipStreamNonBlocking) != hipSuccess) { a.s = nullptr; return nullptr; }
        if (hipEventCreateWithFlags(&a.stagger, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&a.join, hipEventDisableTiming) != hipSuccess) {
            (void)hipStreamDestroy(a.s);
            a.s = nullptr;
            return nullptr;
        }
    }
    return &a;
}
// Chunk boundaries in RENDERS: bounds[0..C].  Whole scenes per chunk when there are several scenes (`by_scene`),
// else groups of views of the one scene.  One chunk unless SPF_CHUNKS asks for more (experiments only: no lower bound
// on a chunk's size is applied -- small chunks are what the bit-identity tests run).
int plan_chunks(int S, int V, int T, int* bounds, bool* by_scene) {
    const char* e = getenv("SPF_CHUNKS");
    int want = e ? atoi(e) : 1;
    if (want > kMaxChunks) want = kMaxChunks;
    const int units = S > 1 ? S : V, per_unit = S > 1 ? V : 1;
    *by_scene = S > 1;
    int C = want < 1 ? 1 : want;
    (void)T;
    if (C > units) C = units;
    for (int c = 0; c <= C; ++c) bounds[c] = (int)(((long)units * c) / C) * per_unit;
    return C;
}
struct Chunk {
    SpfDims d;
    SpfInputs in;
    SpfState st;
    SpfOutputs out;
    SpfGrads g;
};
template <typename P>
P* off(P* p, size_t n) { return p ? p + n : p; }
// renders [r0, r1) of the batch as a call of its own.  `scene0` >= 0: the chunk is scenes [scene0, scene0 + nscene)
// (per-scene inputs and gradients are offset too: the projection backward); < 0: tile-stage kernels only.
Chunk make_chunk(const SpfDims& d, const SpfInputs& in, const SpfState& st, const SpfOutputs* out, const SpfGrads* g,
                 int r0, int r1, int scene0, int nscene) {
    Chunk c;
    const size_t G = (size_t)d.G, P = (size_t)d.H * d.W;
    const size_t T = (size_t)spf_raster_num_tiles(d.H, d.W), nblk = (size_t)spf_raster_view_partial_blocks(d.G);
    const size_t r = (size_t)r0;
    c.d = d;
    if (scene0 >= 0) { c.d.S = nscene; } else { c.d.S = 1; c.d.V = r1 - r0; }
    c.in = in;
    c.in.viewmatrix = off(in.viewmatrix, 16 * r); c.in.projmatrix = off(in.projmatrix, 16 * r);
    c.in.tanfov = off(in.tanfov, 2 * r); c.in.bg = off(in.bg, 3 * r); c.in.view_scale = off(in.view_scale, r);
    c.in.viewmatrix64 = off(in.viewmatrix64, 16 * r);
    if (scene0 >= 0) {
        const size_t sg = (size_t)scene0 * G;
        c.in.means3D = off(in.means3D, 3 * sg); c.in.scales = off(in.scales, 3 * sg);
        c.in.rotations = off(in.rotations, 4 * sg); c.in.opacities = off(in.opacities, sg);
        c.in.shs = off(in.shs, 3 * sg * (size_t)(d.sh_layout == 2 ? 16 : d.K)); c.in.colors = off(in.colors, 3 * sg);
        c.in.shs_high = off(in.shs_high, 3 * sg * 9);
        c.in.raw = off(in.raw, sg * (size_t)d.raw_stride);
    }
    c.st = st;
    c.st.rec = off(st.rec, r * G * spf::kRec); c.st.radii = off(st.radii, r * G); c.st.rect = off(st.rect, r * G);
    c.st.zkey = off(st.zkey, r * G); c.st.tile_count = off(st.tile_count, r * T);
    c.st.tile_start = off(st.tile_start, r * T); c.st.tile_fill = off(st.tile_fill, r * T);
    c.st.tile_flags = off(st.tile_flags, r * T); c.st.pair_off = off(st.pair_off, 2 * r * G);
    c.st.blk_total = off(st.blk_total, r * nblk); c.st.blk_base = off(st.blk_base, r * nblk);
    c.st.final_T = off(st.final_T, r * P); c.st.n_contrib = off(st.n_contrib, r * P);
    c.st.sh_clamp = off(st.sh_clamp, r * G);
    if (out) {
        c.out.image = off(out->image, 3 * r * P); c.out.depth = off(out->depth, r * P);
        c.out.alpha = off(out->alpha, r * P);
    } else {
        c.out = SpfOutputs{nullptr, nullptr, nullptr};
    }
    if (g) {
        c.g = *g;
        c.g.dL_dimage = off(g->dL_dimage, 3 * r * P); c.g.dL_ddepth = off(g->dL_ddepth, r * P);
        c.g.dL_dalpha = off(g->dL_dalpha, r * P);
        c.g.vpartial = off(g->vpartial, r * nblk * 12); c.g.dL_dviewmatrix = off(g->dL_dviewmatrix, 16 * r);
        c.g.dL_dmeans2D = off(g->dL_dmeans2D, 3 * r * G);
        if (scene0 >= 0) {
            const size_t sg = (size_t)scene0 * G;
            c.g.dL_dmeans3D = off(g->dL_dmeans3D, 3 * sg); c.g.dL_dscales = off(g->dL_dscales, 3 * sg);
            c.g.dL_drotations = off(g->dL_drotations, 4 * sg); c.g.dL_dopacities = off(g->dL_dopacities, sg);
            c.g.dL_dshs = off(g->dL_dshs, 3 * sg * (size_t)(d.sh_layout == 2 ? 16 : d.K)); c.g.dL_dcolors = off(g->dL_dcolors, 3 * sg);
            c.g.dL_dshs_high = off(g->dL_dshs_high, 3 * sg * 9);
            c.g.dL_draw = off(g->dL_draw, sg * (size_t)(7 + 3 * d.K));
        }
    } else {
        memset(&c.g, 0, sizeof c.g);
    }
    return c;
}
const char* kStageKernel[SPF_STAGE_COUNT] = {
    "spf_project_fwd_kernel", "spf_tile_scan_kernel",  "spf_bin_pairs_kernel",   "spf_sort_tiles_wave_kernel",
    "spf_render_fwd_lists_kernel", "spf_render_bwd_lists_kernel", "spf_project_bwd_kernel", "spf_rope2d_vec_kernel"};

int check_dims(const SpfDims* d) {
    if (!d) return fail(SPF_E_INVALID, "dims is null");
    if (d->S <= 0 || d->V <= 0 || d->G <= 0 || d->H <= 0 || d->W <= 0)
        return fail(SPF_E_INVALID, "S, V, G, H, W must be positive (got %d %d %d %d %d)", d->S, d->V, d->G, d->H, d->W);
    if ((d->W + SPF_TILE - 1) / SPF_TILE > 255 || (d->H + SPF_TILE - 1) / SPF_TILE > 255)
        return fail(SPF_E_INVALID, "image larger than 4080 px per side is not supported");
    if (d->sh_degree < 0 || d->sh_degree > 4) return fail(SPF_E_INVALID, "sh_degree %d outside 0..4", d->sh_degree);
    if (d->K < 0) return fail(SPF_E_INVALID, "K must be >= 0");
    if (d->sh_band4 != 0 && d->sh_band4 != 1) return fail(SPF_E_INVALID, "sh_band4 must be 0 or 1");
    if (d->sh_layout < 0 || d->sh_layout > 3) return fail(SPF_E_INVALID, "sh_layout must be 0 .. 3");
    if (d->sh_layout == 3 && (d->K < 1 || d->raw_stride < 7 + 3 * (int64_t)d->K))
        return fail(SPF_E_INVALID, "sh_layout 3 (raw rows): K >= 1 and raw_stride >= 7 + 3 K (got K = %d, raw_stride = %lld)", d->K, (long long)d->raw_stride);
    if (d->sh_layout == 2 && d->K != 25 && d->K != 0)
        return fail(SPF_E_INVALID, "sh_layout 2 (band split) is the 16 + 9 split of K = 25 coefficients (got K = %d)", d->K);
    if (d->bin_cap < 0) return fail(SPF_E_INVALID, "bin_cap must be >= 0");
    if (d->bin_cap > 0) {
        const int64_t rt = (int64_t)d->S * d->V * spf_raster_num_tiles(d->H, d->W);
        if (rt * d->bin_cap > ((int64_t)1 << 31))
            return fail(SPF_E_INVALID, "direct bins: S*V*tiles*bin_cap = %lld exceeds 2^31", (long long)(rt * d->bin_cap));
        if (spf_raster_num_tiles(d->H, d->W) > spf::max_lds_tiles())
            return fail(SPF_E_INVALID, "direct bins need the per-render tile histogram in LDS (<= %d tiles)", spf::max_lds_tiles());
        if (d->pair_capacity <= 0) return fail(SPF_E_INVALID, "direct bins: pair_capacity must be positive");
    }
    return SPF_OK;
}

int check_inputs(const SpfDims* d, const SpfInputs* in) {
    if (!in) return fail(SPF_E_INVALID, "inputs is null");
    const bool raw = d->sh_layout == 3;
    if (!in->means3D || !in->opacities || !in->viewmatrix || !in->projmatrix || !in->tanfov || !in->bg)
        return fail(SPF_E_INVALID, "a required input pointer is null");
    if (raw) {
        if (!in->raw || !in->sh_mask) return fail(SPF_E_INVALID, "sh_layout 3: raw and sh_mask are required");
        if (in->shs || in->colors) return fail(SPF_E_INVALID, "sh_layout 3: shs / colors must be null (the harmonics are in the raw rows)");
    } else {
        if (!in->scales || !in->rotations) return fail(SPF_E_INVALID, "a required input pointer is null");
        if ((in->shs == nullptr) == (in->colors == nullptr))
            return fail(SPF_E_INVALID, "exactly one of shs / colors must be given");
    }
    if (in->shs || raw) {
        const int cap = d->sh_band4 ? 4 : 3;
        const int deg = d->sh_degree > cap ? cap : d->sh_degree;
        if (d->K < (deg + 1) * (deg + 1))
            return fail(SPF_E_INVALID, "K = %d is too small for sh_degree %d", d->K, d->sh_degree);
        if (d->sh_layout == 2 && deg == 4 && !in->shs_high)
            return fail(SPF_E_INVALID, "sh_layout 2 with sh_band4: shs_high (band 4) is null");
    }
    return SPF_OK;
}

}  // namespace

extern "C" {

int spf_abi_version(void) { return SPF_ABI_VERSION; }
const char* spf_last_error(void) { return g_err; }

int spf_raster_num_tiles(int32_t H, int32_t W) {
    return ((W + SPF_TILE - 1) / SPF_TILE) * ((H + SPF_TILE - 1) / SPF_TILE);
}
int spf_raster_view_partial_blocks(int32_t G) { return (G + spf::kBlock - 1) / spf::kBlock; }
int spf_raster_launch_slot_tile(int32_t R, int32_t T, int32_t xcd, int32_t slot) {
    if (R < 1 || T < 1 || (int64_t)R * T > (int64_t)1 << 30 || (((int64_t)R * T) & 7) != 0) return -1;
    const int RT = R * T;
    if (xcd < 0 || xcd > 7 || slot < 0 || slot >= (RT >> 3)) return -1;
    return spf::xcd_tile(spf::xcd_map(RT, T), T, xcd, slot);
}
int spf_raster_pair_shards(int32_t S, int32_t G) {
    if (S < 1 || G < 1) return 1;
    const int64_t nblocks = (int64_t)S * ((G + spf::kBlock - 1) / spf::kBlock);
    return spf::pair_shards(nblocks > 0x7fffffff ? 0x7fffffff : (int)nblocks);
}
int spf_raster_max_lds_tiles(void) { return spf::max_lds_tiles(); }
int spf_raster_chunks(int32_t S, int32_t V, int32_t H, int32_t W, int32_t backward) {
    if (S < 1 || V < 1 || H < 1 || W < 1) return 1;
    int bounds[kMaxChunks + 1];
    bool by_scene = false;
    const int C = plan_chunks(S, V, spf_raster_num_tiles(H, W), bounds, &by_scene);
    return (backward && !by_scene) ? 1 : C;
}

static int check_camera(const SpfCamera* c, bool fwd) {
    if (!c) return fail(SPF_E_INVALID, "camera is null");
    if (c->R <= 0) return fail(SPF_E_INVALID, "camera R must be positive");
    if (!c->near || !c->viewmatrix) return fail(SPF_E_INVALID, "a camera pointer is null");
    if (fwd && (!c->extrinsics || !c->intrinsics || !c->far || !c->projmatrix || !c->tanfov))
        return fail(SPF_E_INVALID, "a camera pointer is null");
    return SPF_OK;
}

int spf_camera_forward(const SpfCamera* cam, void* stream_) {
    int rc = check_camera(cam, true);
    if (rc) return rc;
    SPF_HIP(spf::launch_camera_fwd(*cam, static_cast<hipStream_t>(stream_)));
    return SPF_OK;
}

int spf_camera_backward(const SpfCamera* cam, const float* dL_dviewmatrix, float* dL_dextrinsics, void* stream_) {
    int rc = check_camera(cam, false);
    if (rc) return rc;
    if (!dL_dviewmatrix || !dL_dextrinsics) return fail(SPF_E_INVALID, "gradient pointer is null");
    SPF_HIP(spf::launch_camera_bwd(*cam, dL_dviewmatrix, dL_dextrinsics, static_cast<hipStream_t>(stream_)));
    return SPF_OK;
}

// tiles_cleared: 0 = nothing (this call clears the counts), 1 = tile_count | tile_flags, 2 = all the tile bookkeeping
static int forward_project(const SpfDims* d, const SpfInputs* in, SpfState* st, int tiles_cleared, void* stream_,
                           uint64_t cleared_words = 0) {
    int rc = check_dims(d);
    if (rc) return rc;
    rc = check_inputs(d, in);
    if (rc) return rc;
    if (!st || !st->rec || !st->radii || !st->rect || !st->zkey || !st->tile_count || !st->tile_start ||
        !st->tile_fill || !st->tile_flags || !st->counters || !st->blk_total || !st->blk_base)
        return fail(SPF_E_INVALID, "a state pointer needed by forward_project is null");
    if ((in->shs || in->raw) && !st->sh_clamp) return fail(SPF_E_INVALID, "sh_clamp is needed by forward_project when shs are given");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const int tiles_x = (d->W + SPF_TILE - 1) / SPF_TILE, tiles_y = (d->H + SPF_TILE - 1) / SPF_TILE;
    const int RT = d->S * d->V * tiles_x * tiles_y;
    const bool direct = d->bin_cap > 0;
    if (direct) {
        if (!st->pairs || !st->pair_cursor || !st->pair_off)
            return fail(SPF_E_INVALID, "direct bins: pairs, pair_cursor and pair_off are needed by forward_project");
        // counters[0..3] and the cursors start at zero: inside what spf_decoder_prepare cleared, or cleared here
        const uint32_t* const end = st->tile_count + cleared_words;
        if (!(tiles_cleared == 2 && st->counters >= st->tile_count && st->counters + 4 <= end))
            SPF_HIP(hipMemsetAsync(st->counters, 0, 4 * sizeof(uint32_t), stream));
        if (!(tiles_cleared == 2 && st->pair_cursor >= st->tile_count && st->pair_cursor + 8 <= end))
            SPF_HIP(hipMemsetAsync(st->pair_cursor, 0, 8 * sizeof(uint32_t), stream));
    }
    if (tiles_cleared) {
        // spf_decoder_prepare cleared tile_count | tile_flags | tile_start | tile_fill | counters with the camera set-up
    } else if (st->tile_flags == st->tile_count + RT) {   // adjacent (the Python binding lays them out so): one fill
        SPF_HIP(hipMemsetAsync(st->tile_count, 0, sizeof(uint32_t) * 2 * (size_t)RT, stream));
    } else {
        SPF_HIP(hipMemsetAsync(st->tile_count, 0, sizeof(uint32_t) * (size_t)RT, stream));
        SPF_HIP(hipMemsetAsync(st->tile_flags, 0, sizeof(uint32_t) * (size_t)RT, stream));
    }
    {
        StageScope t(SPF_STAGE_PROJECT, stream);
        SPF_HIP(spf::launch_project_fwd(*d, *in, *st, tiles_x, tiles_y, stream));
    }
    if (direct) return SPF_OK;              // the projection kernel binned; tile_count is the bins' fill: no scan
    {
        StageScope t(SPF_STAGE_SCAN, stream);
        SPF_HIP(spf::launch_tile_scan(*st, d->S * d->V, tiles_x * tiles_y, spf_raster_view_partial_blocks(d->G),
                                      spf::dense_threshold(), tiles_cleared == 2, stream));
    }
    return SPF_OK;
}

int spf_raster_forward_project(const SpfDims* d, const SpfInputs* in, SpfState* st, void* stream_) {
    return forward_project(d, in, st, 0, stream_);
}

int spf_raster_forward_project_prepared(const SpfDims* d, const SpfInputs* in, SpfState* st, uint64_t cleared_bytes,
                                        void* stream_) {
    // `cleared_bytes`: how much of tile_count | tile_flags | tile_start | tile_fill | counters (one buffer, in this order)
    // the caller cleared.  All of it: one-block-per-render scan.  Only the two count arrays (the older contract): the
    // self-initialising single-block scan.  Less than that: this call clears the counts itself.
    if (!d || !st) return fail(SPF_E_INVALID, "dims / state is null");
    const uint64_t RT = (uint64_t)d->S * d->V * spf_raster_num_tiles(d->H, d->W);
    const bool laid_out = st->tile_count && st->tile_flags == st->tile_count + RT && st->tile_start == st->tile_flags + RT &&
                          st->tile_fill == st->tile_start + RT + 1 && st->counters == st->tile_fill + RT;
    if (laid_out && cleared_bytes >= 4 * (4 * RT + 5)) return forward_project(d, in, st, 2, stream_, cleared_bytes / 4);
    if (laid_out && cleared_bytes >= 8 * RT) return forward_project(d, in, st, 1, stream_, cleared_bytes / 4);
    return forward_project(d, in, st, 0, stream_);
}

int spf_decoder_prepare(const SpfCamera* cam, void* zero, uint64_t zero_bytes, void* stream_) {
    int rc = check_camera(cam, true);
    if (rc) return rc;
    if (!zero || (reinterpret_cast<uintptr_t>(zero) & 15) || (zero_bytes & 15))
        return fail(SPF_E_INVALID, "decoder_prepare: the buffer to clear must be 16-byte aligned and sized");
    SPF_HIP(spf::launch_camera_fwd_zero(*cam, zero, zero_bytes, static_cast<hipStream_t>(stream_)));
    return SPF_OK;
}

int spf_camera_backward_partials(const SpfCamera* cam, const float* vpartial, int32_t nblk, float* dL_dextrinsics,
                                 void* stream_) {
    int rc = check_camera(cam, false);
    if (rc) return rc;
    if (!vpartial || !dL_dextrinsics || nblk < 1) return fail(SPF_E_INVALID, "camera_backward_partials: bad argument");
    if (reinterpret_cast<uintptr_t>(vpartial) & 15)
        return fail(SPF_E_INVALID, "camera_backward_partials: vpartial must be 16-byte aligned");
    SPF_HIP(spf::launch_camera_bwd_reduce(*cam, vpartial, nblk, dL_dextrinsics, static_cast<hipStream_t>(stream_)));
    return SPF_OK;
}

// SPF_TILE_ORDER=0: the composite lists kernels take their tiles in image order also with direct bins (experiments)
static bool tile_order_enabled() {
    const char* e = getenv("SPF_TILE_ORDER");       // (read per call: the tests flip it; a backward follows its forward's)
    return !(e && e[0] == '0');
}

// SPF_XCD_DEAL=0: every XCD keeps a contiguous range of renders whatever their number (experiments, tests; see
// spf_common.h::xcd_map)
static bool xcd_deal_enabled() {
    const char* e = getenv("SPF_XCD_DEAL");
    return !(e && e[0] == '0');
}

int spf_raster_forward_render(const SpfDims* d, const SpfInputs* in, SpfState* st, SpfOutputs* out, uint64_t capacity,
                              uint32_t max_tile_hint, uint32_t dense_tiles_hint, void* stream_) {
    (void)dense_tiles_hint;       // (ignored: one kernel composites sparse and dense tiles -- see the header)
    int rc = check_dims(d);
    if (rc) return rc;
    rc = check_inputs(d, in);
    if (rc) return rc;
    if (!st || !st->rec || !st->rect || !st->zkey || !st->tile_start || !st->tile_fill || !st->tile_flags ||
        !st->counters || !st->tile_count ||
        !st->final_T || !st->n_contrib || !st->pair_off || !st->blk_base)
        return fail(SPF_E_INVALID, "a state pointer needed by forward_render is null");
    if (capacity > 0 && !st->pairs) return fail(SPF_E_INVALID, "pairs is null but capacity > 0");
    if (!out || !out->image || !out->depth || !out->alpha) return fail(SPF_E_INVALID, "an output pointer is null");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const int tiles_x = (d->W + SPF_TILE - 1) / SPF_TILE, tiles_y = (d->H + SPF_TILE - 1) / SPF_TILE;
    const int T = tiles_x * tiles_y, RT = d->S * d->V * T;
    if (d->bin_cap > 0) {
        // direct bins: the lists are already in their bins (spf_raster_forward_project*): sort and composite.  `capacity`
        // is the number of gradient records here; the bins are memory-safe by construction.
        if (max_tile_hint != 0u && (uint32_t)d->bin_cap > max_tile_hint) max_tile_hint = (uint32_t)d->bin_cap;
        const spf::TileLists tl = spf::tile_lists(*st, *d);
        const bool ordered = tile_order_enabled();
        {
            StageScope t(SPF_STAGE_SORT, stream);
            // (with `ordered`, eight blocks of the sort's first kernel also write the composite lists kernels' launch order:
            //  long lists first, see tile_order_ptr)
            SPF_HIP(spf::launch_tile_sort(*st, tl, RT, RT, ~0ull, max_tile_hint ? max_tile_hint : (uint32_t)d->bin_cap,
                                          ordered ? spf::tile_order_ptr(*st, *d, RT) : nullptr,
                                          xcd_deal_enabled() ? T : 0, stream));
        }
        {
            StageScope t(SPF_STAGE_RENDER_FWD, stream);
            SPF_HIP(spf::launch_render_fwd(*d, *in, *st, *out, ~0ull, T, tiles_x, ordered, stream));
        }
        return SPF_OK;
    }
    int bounds[kMaxChunks + 1];
    bool by_scene = false;
    int C = plan_chunks(d->S, d->V, T, bounds, &by_scene);
    LaneSet* lanes = C > 1 ? lane_set() : nullptr;
    if (!lanes) { C = 1; bounds[0] = 0; bounds[1] = d->S * d->V; }
    // (a failure inside the loop must not leave the auxiliary lane forked: an un-joined stream invalidates an ongoing
    //  HIP-graph capture -- the chunk body reports, the join below runs either way)
    auto chunk = [&](int c) -> int {
        const hipStream_t cs = (c & 1) ? lanes->s : stream;
        const Chunk ch = make_chunk(*d, *in, *st, out, nullptr, bounds[c], bounds[c + 1], -1, 0);
        const int rt = (bounds[c + 1] - bounds[c]) * T;
        if (c == 1) SPF_HIP(hipStreamWaitEvent(cs, lanes->stagger, 0));     // second lane: one kernel behind the first
        {
            StageScope t(SPF_STAGE_BIN, cs);
            SPF_HIP(spf::launch_bin_pairs(ch.d, ch.st, capacity, T, tiles_x, max_tile_hint, cs));
        }
        if (c == 0 && C > 1) SPF_HIP(hipEventRecord(lanes->stagger, cs));
        {
            StageScope t(SPF_STAGE_SORT, cs);
            SPF_HIP(spf::launch_tile_sort(ch.st, spf::tile_lists(ch.st, ch.d), rt, RT, capacity, max_tile_hint,
                                          /*order*/ nullptr, 0, cs));
        }
        {
            StageScope t(SPF_STAGE_RENDER_FWD, cs);
            SPF_HIP(spf::launch_render_fwd(ch.d, ch.in, ch.st, ch.out, capacity, T, tiles_x, false, cs));
        }
        return SPF_OK;
    };
    for (int c = 0; c < C && rc == SPF_OK; ++c) rc = chunk(c);
    if (C > 1) {
        const hipError_t e1 = hipEventRecord(lanes->join, lanes->s), e2 = hipStreamWaitEvent(stream, lanes->join, 0);
        if (rc == SPF_OK && (e1 != hipSuccess || e2 != hipSuccess))
            return fail(SPF_E_LAUNCH, "joining the auxiliary lane: %s", hipGetErrorString(e1 != hipSuccess ? e1 : e2));
    }
    return rc;
}

int spf_raster_backward(const SpfDims* d, const SpfInputs* in, const SpfState* st, const SpfGrads* g,
                        uint64_t capacity, uint32_t dense_tiles_hint, void* stream_) {
    (void)dense_tiles_hint;
    int rc = check_dims(d);
    if (rc) return rc;
    rc = check_inputs(d, in);
    if (rc) return rc;
    if (!st || !st->rec || !st->radii || !st->rect || !st->tile_start || !st->tile_flags || !st->pairs ||
        !st->final_T || !st->n_contrib || !st->pair_off || !st->tile_count)
        return fail(SPF_E_INVALID, "a state pointer needed by backward is null");
    if (!g || !g->gpair || !g->dL_dmeans3D || !g->dL_dopacities)
        return fail(SPF_E_INVALID, "gpair, dL_dmeans3D and dL_dopacities are required");
    if (d->sh_layout == 3 && !g->dL_draw) return fail(SPF_E_INVALID, "sh_layout 3: dL_draw is required");
    if ((in->shs || in->raw) && !st->sh_clamp) return fail(SPF_E_INVALID, "sh_clamp (written by the forward) is needed by backward when shs are given");
    if (g->dL_dviewmatrix && !g->vpartial) return fail(SPF_E_INVALID, "vpartial is required with dL_dviewmatrix");
    if ((g->dL_dscales == nullptr) != (g->dL_drotations == nullptr))
        return fail(SPF_E_INVALID, "dL_dscales and dL_drotations must be given together");
    if (in->shs && d->sh_layout == 2 && d->sh_band4 && d->sh_degree == 4 && g->dL_dshs && !g->dL_dshs_high)
        return fail(SPF_E_INVALID, "sh_layout 2 with sh_band4: dL_dshs_high is needed next to dL_dshs");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const int tiles_x = (d->W + SPF_TILE - 1) / SPF_TILE, tiles_y = (d->H + SPF_TILE - 1) / SPF_TILE;
    const int T = tiles_x * tiles_y;
    // (every pair record is written exactly once by its tile: no memset of gpair)
    int bounds[kMaxChunks + 1];
    bool by_scene = false;
    int C = d->bin_cap > 0 ? 1 : plan_chunks(d->S, d->V, T, bounds, &by_scene);
    LaneSet* lanes = (C > 1 && by_scene) ? lane_set() : nullptr;     // the projection backward owns whole scenes
    if (!lanes) { C = 1; bounds[0] = 0; bounds[1] = d->S * d->V; }
    auto chunk = [&](int c) -> int {                                        // (see spf_raster_forward_render)
        const hipStream_t cs = (c & 1) ? lanes->s : stream;
        const int s0 = bounds[c] / d->V, ns = (bounds[c + 1] - bounds[c]) / d->V;
        const Chunk ch = C > 1 ? make_chunk(*d, *in, *st, nullptr, g, bounds[c], bounds[c + 1], s0, ns)
                               : Chunk{*d, *in, *st, SpfOutputs{nullptr, nullptr, nullptr}, *g};
        if (c == 1) SPF_HIP(hipStreamWaitEvent(cs, lanes->stagger, 0));     // second lane: one kernel behind the first
        {
            StageScope t(SPF_STAGE_RENDER_BWD, cs);
            SPF_HIP(spf::launch_render_bwd(ch.d, ch.in, ch.st, ch.g, T, tiles_x, capacity,
                                           d->bin_cap > 0 && C == 1 && tile_order_enabled(), cs));
        }
        if (c == 0 && C > 1) SPF_HIP(hipEventRecord(lanes->stagger, cs));
        {
            StageScope t(SPF_STAGE_PROJECT_BWD, cs);
            SPF_HIP(spf::launch_project_bwd(ch.d, ch.in, ch.st, ch.g, spf_raster_view_partial_blocks(d->G), capacity,
                                            cs));
        }
        return SPF_OK;
    };
    for (int c = 0; c < C && rc == SPF_OK; ++c) rc = chunk(c);
    if (C > 1) {
        const hipError_t e1 = hipEventRecord(lanes->join, lanes->s), e2 = hipStreamWaitEvent(stream, lanes->join, 0);
        if (rc == SPF_OK && (e1 != hipSuccess || e2 != hipSuccess))
            return fail(SPF_E_LAUNCH, "joining the auxiliary lane: %s", hipGetErrorString(e1 != hipSuccess ? e1 : e2));
    }
    return rc;
}

int spf_adapter_forward(const float* raw, int64_t raw_stride, int64_t N, int32_t K, const float* sh_mask, float eps,
                        float* scales, float* rotations, float* harmonics, float* harmonics_high, void* stream_) {
    if (!raw || !sh_mask || !scales || !rotations || !harmonics) return fail(SPF_E_INVALID, "adapter: null pointer");
    if (N < 0 || K < 1 || K > 64) return fail(SPF_E_INVALID, "adapter: N must be >= 0 and K in 1..64");
    if (raw_stride < 7 + 3 * (int64_t)K) return fail(SPF_E_INVALID, "adapter: raw_stride %lld is below the %d channels of a row", (long long)raw_stride, 7 + 3 * K);
    if (harmonics_high && K != 25) return fail(SPF_E_INVALID, "adapter: the band-split layout is the 16 + 9 split of K = 25 (got K = %d)", K);
    if (N == 0) return SPF_OK;
    SPF_HIP(spf::launch_adapter_fwd(raw, raw_stride, N, K, sh_mask, eps, scales, rotations, harmonics, harmonics_high,
                                    static_cast<hipStream_t>(stream_)));
    return SPF_OK;
}

int spf_adapter_backward(const float* raw, int64_t raw_stride, int64_t N, int32_t K, const float* sh_mask, float eps,
                         const float* dL_dscales, const float* dL_drotations, const float* dL_dharmonics,
                         const float* dL_dharmonics_high, int32_t split, float* dL_draw, void* stream_) {
    if (!raw || !sh_mask || !dL_draw) return fail(SPF_E_INVALID, "adapter: null pointer");
    if (N < 0 || K < 1 || K > 64) return fail(SPF_E_INVALID, "adapter: N must be >= 0 and K in 1..64");
    if (raw_stride < 7 + 3 * (int64_t)K) return fail(SPF_E_INVALID, "adapter: raw_stride %lld is below the %d channels of a row", (long long)raw_stride, 7 + 3 * K);
    if (split && K != 25) return fail(SPF_E_INVALID, "adapter: the band-split layout is the 16 + 9 split of K = 25 (got K = %d)", K);
    if (!split && dL_dharmonics_high) return fail(SPF_E_INVALID, "adapter: dL_dharmonics_high without split");
    if (dL_drotations && (reinterpret_cast<uintptr_t>(dL_drotations) & 15))
        return fail(SPF_E_INVALID, "adapter: dL_drotations must be 16-byte aligned");
    if (N == 0) return SPF_OK;
    SPF_HIP(spf::launch_adapter_bwd(raw, raw_stride, N, K, sh_mask, eps, dL_dscales, dL_drotations, dL_dharmonics,
                                    dL_dharmonics_high, split ? 1 : 0, dL_draw, static_cast<hipStream_t>(stream_)));
    return SPF_OK;
}

int spf_mse_partial_blocks(void) { return spf::mse_partial_blocks(); }

int spf_mse_forward(const float* prediction, const float* image, int64_t n, float weight, float* partial,
                    float* loss, void* stream_) {
    if (!prediction || !image || !partial || !loss) return fail(SPF_E_INVALID, "mse: null pointer");
    if (n <= 0) return fail(SPF_E_INVALID, "mse: n must be positive (got %lld)", (long long)n);
    if ((reinterpret_cast<uintptr_t>(prediction) | reinterpret_cast<uintptr_t>(image)) & 15)
        return fail(SPF_E_INVALID, "mse: prediction / image must be 16-byte aligned");
    SPF_HIP(spf::launch_mse_fwd(prediction, image, n, weight / (float)n, partial, loss, 0.f, nullptr,
                                static_cast<hipStream_t>(stream_)));
    return SPF_OK;
}

int spf_mse_forward_grad(const float* prediction, const float* image, int64_t n, float weight, float* partial,
                         float* loss, float* dL_dprediction_unit, void* stream_) {
    if (!prediction || !image || !partial || !loss || !dL_dprediction_unit) return fail(SPF_E_INVALID, "mse: null pointer");
    if (n <= 0) return fail(SPF_E_INVALID, "mse: n must be positive (got %lld)", (long long)n);
    if ((reinterpret_cast<uintptr_t>(prediction) | reinterpret_cast<uintptr_t>(image) |
         reinterpret_cast<uintptr_t>(dL_dprediction_unit)) & 15)
        return fail(SPF_E_INVALID, "mse: tensors must be 16-byte aligned");
    SPF_HIP(spf::launch_mse_fwd(prediction, image, n, weight / (float)n, partial, loss, 2.0f * weight / (float)n,
                                dL_dprediction_unit, static_cast<hipStream_t>(stream_)));
    return SPF_OK;
}

int spf_mse_scale_grad(float* dL_dprediction, int64_t n, const float* dL_dloss, void* stream_) {
    if (!dL_dprediction || !dL_dloss) return fail(SPF_E_INVALID, "mse: null pointer");
    if (n <= 0) return fail(SPF_E_INVALID, "mse: n must be positive (got %lld)", (long long)n);
    if (reinterpret_cast<uintptr_t>(dL_dprediction) & 15) return fail(SPF_E_INVALID, "mse: tensors must be 16-byte aligned");
    SPF_HIP(spf::launch_mse_scale(dL_dprediction, n, dL_dloss, static_cast<hipStream_t>(stream_)));
    return SPF_OK;
}

int spf_mse_backward(const float* prediction, const float* image, int64_t n, float weight, const float* dL_dloss,
                     float* dL_dprediction, void* stream_) {
    if (!prediction || !image || !dL_dloss || !dL_dprediction) return fail(SPF_E_INVALID, "mse: null pointer");
    if (n <= 0) return fail(SPF_E_INVALID, "mse: n must be positive (got %lld)", (long long)n);
    if ((reinterpret_cast<uintptr_t>(prediction) | reinterpret_cast<uintptr_t>(image) |
         reinterpret_cast<uintptr_t>(dL_dprediction)) & 15)
        return fail(SPF_E_INVALID, "mse: tensors must be 16-byte aligned");
    SPF_HIP(spf::launch_mse_bwd(prediction, image, n, 2.0f * weight / (float)n, dL_dloss, dL_dprediction,
                                static_cast<hipStream_t>(stream_)));
    return SPF_OK;
}

static int rope2d_impl(void* tokens, void* tokens2, const int64_t* positions, int32_t B, int32_t N, int32_t H, int32_t D,
                       int64_t stride_b, int64_t stride_n, int64_t stride_h, int32_t pos_div, int32_t dtype, float base,
                       float fwd, void* stream_) {
    if (!tokens || !positions) return fail(SPF_E_INVALID, "tokens / positions is null");
    if (B < 0 || N < 0 || H < 0 || D <= 0) return fail(SPF_E_INVALID, "negative size");
    if (D % 4 != 0) return fail(SPF_E_INVALID, "token dim must be multiple of 4");
    if (D > 256) return fail(SPF_E_INVALID, "token dim > 256 is not supported");
    if (dtype < 0 || dtype > 2) return fail(SPF_E_INVALID, "dtype must be 0 (f32), 1 (f16) or 2 (bf16)");
    if (pos_div < 1) return fail(SPF_E_INVALID, "pos_div must be >= 1");
    if ((size_t)B * N * H == 0) return SPF_OK;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    StageScope t(SPF_STAGE_ROPE, stream);
    SPF_HIP(spf::launch_rope2d(tokens, tokens2, positions, B, N, H, D, stride_b, stride_n, stride_h, pos_div, dtype, base,
                               fwd, stream));
    return SPF_OK;
}

int spf_rope2d(void* tokens, const int64_t* positions, int32_t B, int32_t N, int32_t H, int32_t D, int64_t stride_b,
               int64_t stride_n, int64_t stride_h, int32_t pos_div, int32_t dtype, float base, float fwd,
               void* stream_) {
    return rope2d_impl(tokens, nullptr, positions, B, N, H, D, stride_b, stride_n, stride_h, pos_div, dtype, base, fwd,
                       stream_);
}

int spf_rope2d_pair(void* tokens, void* tokens2, const int64_t* positions, int32_t B, int32_t N, int32_t H, int32_t D,
                    int64_t stride_b, int64_t stride_n, int64_t stride_h, int32_t pos_div, int32_t dtype, float base,
                    float fwd, void* stream_) {
    if (!tokens2) return fail(SPF_E_INVALID, "tokens2 is null");
    return rope2d_impl(tokens, tokens2, positions, B, N, H, D, stride_b, stride_n, stride_h, pos_div, dtype, base, fwd,
                       stream_);
}

int spf_stage_timing_enable(int32_t mask) {
    g_timing = (uint32_t)mask;
    if (g_timing)
        for (int s = 0; s < SPF_STAGE_COUNT; ++s) g_log[s].used = 0;
    return SPF_OK;
}

int spf_stage_timing_sample_every(int32_t n) {
    if (n < 1) return fail(SPF_E_INVALID, "sample stride must be >= 1 (got %d)", n);
    g_sample_every = n;
    for (int s = 0; s < SPF_STAGE_COUNT; ++s) g_calls[s] = 0;
    return SPF_OK;
}

int spf_stage_times_ms(float* total_ms, int32_t* count) {
    if (!total_ms || !count) return fail(SPF_E_INVALID, "null output");
    for (int s = 0; s < SPF_STAGE_COUNT; ++s) {
        total_ms[s] = 0.f;
        count[s] = g_log[s].used;
        for (int i = 0; i < g_log[s].used; ++i) {
            SPF_HIP(hipEventSynchronize(g_log[s].ev[i][1]));
            float ms = 0.f;
            SPF_HIP(hipEventElapsedTime(&ms, g_log[s].ev[i][0], g_log[s].ev[i][1]));
            total_ms[s] += ms;
        }
    }
    return SPF_OK;
}

const char* spf_stage_kernel_name(int32_t stage) {
    return (stage >= 0 && stage < SPF_STAGE_COUNT) ? kStageKernel[stage] : "";
}

}  // extern "C"
