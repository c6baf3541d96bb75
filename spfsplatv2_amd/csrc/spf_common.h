// Shared device helpers for the gfx950 Gaussian-splat rasterizer.
// Wave = 64 lanes everywhere in this library (CDNA4); no other target is supported.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/spfsplat_hip.h"

namespace spf {

constexpr int kWave = 64;
constexpr int kRec = 12;        // floats per screen-space record / gradient record
constexpr int kTile = SPF_TILE; // 16
constexpr int kBlock = 256;     // one 16x16 tile = 4 waves, one 8x8 sub-tile per wave
constexpr int kMaxLdsTiles = 4096;  // per-render tile histograms up to this many tiles live in LDS (1024x1024 px)
// (SPF_MAX_LDS_TILES lowers the limit: lets a test drive the global-atomics fall-back on a small image; read per launch)
inline int max_lds_tiles() {
    const char* e = getenv("SPF_MAX_LDS_TILES");
    const int v = e ? atoi(e) : kMaxLdsTiles;
    return v < kMaxLdsTiles ? v : kMaxLdsTiles;
}

// Record layout (floats): 0 x, 1 y, 2 conic A, 3 conic B | 4 conic C, 5 opacity, 6 depth,
// 7 cull radius^2 | 8 r, 9 g, 10 b, 11 flags (int bits: colour-channel clamp mask).
// Gradient record (one per (Gaussian, tile) pair): 0 dx, 1 dy (pixel space), 2 da, 3 db, 4 dc, 5 dopacity,
// 6..8 drgb [, 9 ddepth]: 9 floats, 10 when the depth output has an upstream gradient -- packed (36 / 40-byte
// stride, dword aligned), because every byte of it crosses HBM twice (render backward -> projection backward).
// (da, db, dc) = dL/d(a, b, c) of the 2-D covariance [[a, b], [b, c]] ITSELF, not of the conic: with u = dL/dpower of
// a pixel and v = conic * (pixel offset), da = sum 1/2 u vx^2, db = sum u vx vy, dc = sum 1/2 u vy^2.  The classic
// route (sum dL/dconic over the pixels, then -Q G Q once per Gaussian) cancels catastrophically in float32 when the
// centre of an anisotropic splat lies hundreds of pixels from the pixels it touches (three terms of ~1e-2 adding up
// to ~1e-5: 1 % gradient error, found by tools/fuzz_campaign.py); v is small wherever alpha is not.

typedef float f4a __attribute__((ext_vector_type(4)));
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));   // dwordx4 access at dword alignment (legal for global memory)
__host__ __device__ __forceinline__ constexpr int grec_floats(bool depth_grad) { return depth_grad ? 10 : 9; }
template <bool DEPTH_GRAD>
__device__ __forceinline__ void store_grec(float* __restrict__ gpair, uint32_t slot, float v0, float v1, float v2,
                                           float v3, float v4, float v5, float v6, float v7, float v8, float v9) {
    float* __restrict__ o = gpair + (size_t)slot * grec_floats(DEPTH_GRAD);
    *reinterpret_cast<f4u*>(o) = f4a{v0, v1, v2, v3};
    *reinterpret_cast<f4u*>(o + 4) = f4a{v4, v5, v6, v7};
    o[8] = v8;
    if (DEPTH_GRAD) o[9] = v9;
}

constexpr float kNearCull = 0.2f;
constexpr float kLowPass = 0.3f;
constexpr float kAlphaMax = 0.99f;
constexpr float kAlphaMin = 1.0f / 255.0f;
constexpr float kTMin = 1e-4f;
constexpr float kFovClamp = 1.3f;

// torch.nn.functional.softplus (beta 1, threshold 20): the Gaussian adapter's scale activation
// (/root/reference/src/model/encoder/common/gaussian_adapter.py:132-133) -- shared by adapter.hip and the projection
// kernels' raw-row mode, which must produce the same bits
__device__ __forceinline__ float softplus_torch(float x) { return x > 20.f ? x : log1pf(expf(x)); }

constexpr float SH_C0 = 0.28209479177387814f;
constexpr float SH_C1 = 0.4886025119029199f;
__device__ constexpr float SH_C2[5] = {1.0925484305920792f, -1.0925484305920792f,
                                       0.31539156525252005f, -1.0925484305920792f,
                                       0.5462742152960396f};
__device__ constexpr float SH_C3[7] = {-0.5900435899266435f, 2.890611442640554f,
                                       -0.4570457994644658f, 0.3731763325901154f,
                                       -0.4570457994644658f, 1.445305721320277f,
                                       -0.5900435899266435f};
// Band 4 (evaluated only when SpfDims.sh_band4 is set): same real-SH family, index n(n+1)+m; pinned against the
// reference's own table src/misc/sht.py::rsh_cart_4 through the oracle (tests/golden/sh_basis_goldens.pt).
__device__ constexpr float SH_C4[9] = {2.5033429417967046f, -1.7701307697799304f, 0.9461746957575601f,
                                       -0.6690465435572892f, 0.10578554691520431f, -0.6690465435572892f,
                                       0.47308734787878004f, -1.7701307697799304f, 0.6258357354491761f};

// ---- DPP wave-64 reductions (result valid in lane 63) --------------------------------------
template <int CTRL, int ROW_MASK = 0xf, int BANK_MASK = 0xf>
__device__ __forceinline__ float dpp_f(float x) {
    return __builtin_bit_cast(
        float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, ROW_MASK, BANK_MASK, true));
}

// Sum over the 64 lanes; lane 63 holds the total (other lanes hold partial sums).
__device__ __forceinline__ float wave_sum_to63(float x) {
    float t = x;
    t += dpp_f<0x111>(x);             // row_shr:1
    t += dpp_f<0x112>(x);             // row_shr:2
    t += dpp_f<0x113>(x);             // row_shr:3
    t += dpp_f<0x114, 0xf, 0xe>(t);   // row_shr:4, banks 1-3
    t += dpp_f<0x118, 0xf, 0xc>(t);   // row_shr:8, banks 2-3
    t += dpp_f<0x142, 0xa>(t);        // row_bcast:15 into rows 1,3
    t += dpp_f<0x143, 0xc>(t);        // row_bcast:31 into rows 2,3
    return t;
}

__device__ __forceinline__ float wave_sum(float x) {
    float t = wave_sum_to63(x);
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, t), 63));
}

// Twelve wave sums at once.  On return out[j] holds, in EVERY lane l, the sum over the 64 lanes of v[4*j + (l & 3)].
// Two value-pairing butterfly steps (lane bit 0, then bit 1: each halves the registers -- a lane keeps one value of a
// pair and hands the partner the other), then plain sums over the remaining lane bits: 45 instructions instead of
// 12 x 8 for one ladder per value.
__device__ __forceinline__ void wave_sum12(const float (&v)[12], float (&out)[3]) {
    const int lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    const bool b0 = lane & 1, b1 = lane & 2;
    float a[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const float keep = b0 ? v[2 * i + 1] : v[2 * i], send = b0 ? v[2 * i] : v[2 * i + 1];
        a[i] = keep + dpp_f<0xB1>(send);                     // quad_perm [1,0,3,2]: lane ^ 1
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const float keep = b1 ? a[2 * j + 1] : a[2 * j], send = b1 ? a[2 * j] : a[2 * j + 1];
        float x = keep + dpp_f<0x4E>(send);                  // quad_perm [2,3,0,1]: lane ^ 2
        x += dpp_f<0x124>(x);                                // row_ror:4   (lanes with equal low bits inside the row)
        x += dpp_f<0x128>(x);                                // row_ror:8
        x += __shfl_xor(x, 16, kWave);
        x += __shfl_xor(x, 32, kWave);
        out[j] = x;
    }
}

// Inclusive prefix sum over the 64 lanes (same DPP ladder as above, on integers).
template <int CTRL, int ROW_MASK = 0xf, int BANK_MASK = 0xf>
__device__ __forceinline__ uint32_t dpp_scan_u(uint32_t x) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, CTRL, ROW_MASK, BANK_MASK, true);
}
__device__ __forceinline__ uint32_t wave_iscan_u32(uint32_t x) {
    uint32_t t = x;
    t += dpp_scan_u<0x111>(x);
    t += dpp_scan_u<0x112>(x);
    t += dpp_scan_u<0x113>(x);
    t += dpp_scan_u<0x114, 0xf, 0xe>(t);
    t += dpp_scan_u<0x118, 0xf, 0xc>(t);
    t += dpp_scan_u<0x142, 0xa>(t);
    t += dpp_scan_u<0x143, 0xc>(t);
    return t;
}

// Ballot of a lane predicate as the compiler keeps it (an SGPR pair).  HIP's __ballot(int) widens the predicate to
// 0 / 1 in a VGPR and compares it again (v_cndmask + v_cmp per call).
__device__ __forceinline__ uint64_t lane_ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }
// value of the previous lane (lane 0 receives 0): DPP wave_shr:1
__device__ __forceinline__ int dpp_wave_shr1(int x) { return __builtin_amdgcn_update_dpp(0, x, 0x138, 0xf, 0xf, false); }

// Runs of neighbouring lanes with equal keys (a pixel-aligned scene puts whole runs of a wave's Gaussians into the same
// tile).  Per-lane LDS atomics on one address serialise -- 64 lanes, 64 passes --, so a run acts through its first lane:
// `head` = first lane of the calling lane's run, `len` (valid in head lanes) = its length.  Keys < 0 never merge with
// a neighbour's into something that matters: callers ignore runs of negative keys.
struct LaneRun { int head, len; };
__device__ __forceinline__ LaneRun lane_runs(int key, int lane) {
    // (the DPP move must run with every lane active: `lane == 0 || key != dpp(key)` short-circuits it under a partial
    //  exec mask and lane 1 then reads an inactive lane 0)
    const int prev = dpp_wave_shr1(key);
    const uint64_t heads = lane_ballot(key != prev) | 1ull;
    LaneRun r;
    r.head = 63 - __builtin_clzll(heads & (~0ull >> (63 - lane)));
    const uint64_t above = lane < 63 ? heads >> (lane + 1) : 0ull;
    r.len = (above ? lane + 1 + __builtin_ctzll(above) : kWave) - lane;
    return r;
}

__device__ __forceinline__ uint32_t wave_max_u32(uint32_t x) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        uint32_t y = (uint32_t)__shfl_xor((int)x, o, 64);
        x = x > y ? x : y;
    }
    return x;
}

__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t x) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x += (uint32_t)__shfl_xor((int)x, o, 64);
    return x;
}

__device__ __forceinline__ uint64_t readfirstlane64(uint64_t v) {
    uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v);
    uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return ((uint64_t)hi << 32) | lo;
}

// Pixel bounding box of a Gaussian's conservative cull disc (centre (gx,gy), squared radius r2), padded so that
// float rounding can only make it larger.  The exact per-pixel test is always `!(dx*dx + dy*dy > r2)`.
struct DiscBox {
    float xlo, xhi, ylo, yhi;   // integral values
    bool any;
};
__device__ __forceinline__ DiscBox disc_box(float gx, float gy, float r2) {
    DiscBox b;
    b.any = r2 >= 0.f;
    const float rb = sqrtf(fmaxf(r2, 0.f)) * 1.0001f + 1e-3f;
    b.xlo = ceilf(gx - rb); b.xhi = floorf(gx + rb);
    b.ylo = ceilf(gy - rb); b.yhi = floorf(gy + rb);
    return b;
}
// The same box from the hardware square root (v_sqrt_f32, 1 ulp; the IEEE-correct sqrtf costs ~15 instructions more)
// for the render kernels, where the box only bounds a loop whose per-pixel test is exact: the 1.0001 / 1e-3 padding
// covers the ulp.  (The projection kernel uses it too since round 3: the binning reads the rect the projection stored.)
__device__ __forceinline__ DiscBox disc_box_fast(float gx, float gy, float r2) {
    DiscBox b;
    b.any = r2 >= 0.f;
    const float rb = __builtin_amdgcn_sqrtf(fmaxf(r2, 0.f)) * 1.0001f + 1e-3f;
    b.xlo = ceilf(gx - rb); b.xhi = floorf(gx + rb);
    b.ylo = ceilf(gy - rb); b.yhi = floorf(gy + rb);
    return b;
}
// bounding-box area of the disc in pixels, capped at one tile (feeds the per-tile sparse/dense decision)
__device__ __forceinline__ uint32_t disc_area_capped(float gx, float gy, float r2) {
    const DiscBox b = disc_box(gx, gy, r2);
    if (!b.any) return 0u;
    const float w = fminf(fmaxf(b.xhi - b.xlo + 1.f, 0.f), (float)kTile), h = fminf(fmaxf(b.yhi - b.ylo + 1.f, 0.f), (float)kTile);
    return (uint32_t)(w * h);
}
__device__ __forceinline__ uint32_t disc_area_capped_fast(float gx, float gy, float r2) {
    const DiscBox b = disc_box_fast(gx, gy, r2);
    if (!b.any) return 0u;
    const float w = fminf(fmaxf(b.xhi - b.xlo + 1.f, 0.f), (float)kTile), h = fminf(fmaxf(b.yhi - b.ylo + 1.f, 0.f), (float)kTile);
    return (uint32_t)(w * h);
}
__host__ __device__ __forceinline__ bool tile_is_dense(uint32_t area_sum, uint32_t n, uint32_t thr = SPF_DENSE_AREA) {
    return area_sum > thr * n;
}

// Where tile `vid` (= render * T + tile) keeps its list: packed lists (start = exclusive scan of the counts) or, with
// direct bins (SpfDims.bin_cap), a fixed bin of `cap` entries per tile filled by the projection kernel.
struct TileLists {
    const uint32_t* __restrict__ start;   // [R*T+1] (packed lists)
    const uint32_t* __restrict__ count;   // [R*T]   (direct bins)
    uint32_t cap;                         // 0: packed lists
    const uint2* __restrict__ order;      // launch order of the composite lists kernels (see spf_tile_order_kernel), or null
};
__device__ __forceinline__ void tile_range(const TileLists& tl, size_t vid, uint32_t& beg, uint32_t& n) {
    if (tl.cap) {
        beg = (uint32_t)vid * tl.cap;
        n = min(tl.count[vid], tl.cap);
    } else {
        beg = tl.start[vid];
        n = tl.start[vid + 1] - beg;
    }
}
inline TileLists tile_lists(const SpfState& st, const SpfDims& d) {
    return TileLists{st.tile_start, st.tile_count, (uint32_t)(d.bin_cap > 0 ? d.bin_cap : 0), nullptr};
}
// Launch order (direct bins, many tiles): the composite lists kernels run one block per tile, a few rounds of blocks
// per launch, and the launch ends when the LAST block does -- with the tiles in image order a long list that starts in
// the last round finishes alone (measured, lists backward: the chip drains for 15 % of the launch on C2, 45 % on C3).
// The tile sort's order blocks (binning.hip::tile_order_block) sort the END of every XCD's contiguous range of tiles by
// list length, longest first, into order[slot] = (tile | dense << 31, list length) -- the slots before it keep the image
// order --, and block b of a lists launch takes slot xcd_remap(b): one 8-byte
// scalar load that also replaces the loads of the tile's count and footprint sum.  The array lives in tile_start |
// tile_fill, which nothing else uses with direct bins (they have to be one 8-byte aligned piece, as the decoder lays
// them out).
constexpr int kOrderMinTiles = 2048;
// Which tiles XCD x's blocks work on (slot (x, j), j < per = R*T / 8), before the end of the range is sorted.
// Normally a contiguous range: XCD x takes renders [x*R/8, (x+1)*R/8) -- with several renders per XCD their differences
// average out (C2, four views of a scene per XCD: 18.4 - 20.3 ms of block time per XCD in the lists backward).  With
// exactly ONE render per XCD (8 renders: C5's 1 scene x 8 views, C3's 2 x 4) nothing averages: per-XCD stamps
// (tools/block_stamps.py) showed 35 - 45 ms on C5, the XCDs finishing between 252 and 330 us.  Then STRIPS of 64 tiles
// (whole tile rows) are dealt out instead -- strip q of render r goes to XCD (r + q) % 8: every XCD gets a different
// part of every render.  Measured: C5 lists backward 340 -> 321 us, C3 +-0; with more renders per XCD dealing LOSES
// (REF2V, two per XCD: 133 -> 139 us; C2 +-0), so it is not done there.
constexpr int kXcdStrip = 64;
struct XcdMap { int per, strip, strips_per_render, groups; bool dealt; };
__host__ __device__ inline XcdMap xcd_map(int RT, int T) {
    XcdMap m;
    const int R = T > 0 ? RT / T : 0;
    m.per = RT >> 3;
    m.strip = T >= kXcdStrip ? kXcdStrip : (T > 0 ? T : 1);
    m.strips_per_render = T > 0 ? T / m.strip : 1;
    m.groups = R >> 3;
    m.dealt = R == 8 && T % m.strip == 0;
    return m;
}
__host__ __device__ inline int xcd_tile(const XcdMap& m, int T, int x, int j) {
    if (!m.dealt) return x * m.per + j;
    const int l = j / m.strip, i = j - l * m.strip;          // l-th strip of this XCD
    const int q = l / m.groups, g = l - q * m.groups;
    const int r = ((x - q) & 7) + 8 * g;
    return r * T + q * m.strip + i;
}
inline const uint2* tile_order_ptr(const SpfState& st, const SpfDims& d, int RT) {
    if (d.bin_cap <= 0 || d.bin_cap > 65536 || RT < kOrderMinTiles || (RT & 7) != 0) return nullptr;
    if (!st.tile_start || st.tile_fill != st.tile_start + RT + 1 || (reinterpret_cast<uintptr_t>(st.tile_start) & 7) != 0)
        return nullptr;
    return reinterpret_cast<const uint2*>(st.tile_start);
}
// shards of the direct-bins pair numbering (one cursor each): blocks spread over up to 8 cursors so that the returning
// atomics of a whole round of blocks do not queue on one address (~88 per us); few blocks -> one shard (no imbalance)
__host__ __device__ inline int pair_shards(int nblocks) { return nblocks >= 512 ? 8 : 1; }

// XCD-aware block remap: the dispatcher places block b on XCD b % 8, so give each XCD one
// contiguous range of work ids (contiguous renders -> their records stay in that XCD's L2).
// Speed only; correctness never depends on it.  grid must be a multiple of 8.
__device__ __forceinline__ int xcd_remap(int block, int grid) {
    const int per = grid >> 3;
    return (block & 7) * per + (block >> 3);
}
// launch-order records: tile id in the low 30 bits, "dense for the backward" / "dense for the forward" above
constexpr uint32_t kOrderTileMask = 0x3fffffffu, kOrderDenseBwd = 0x80000000u, kOrderDenseFwd = 0x40000000u;
// The tile this block of a composite lists launch works on (false: none), where its list is and whether the tile
// belongs to the dense (rows) kernel of this direction (`dense_thr`, `forward`): in image order, or from the launch order.
__device__ __forceinline__ bool lists_tile(const TileLists& tl, const uint32_t* __restrict__ tile_flags,
                                           uint32_t dense_thr, bool forward, int RT, int& vid, uint32_t& beg, uint32_t& n,
                                           bool& dense) {
    const int slot = xcd_remap(blockIdx.x, gridDim.x);
    if (tl.order) {
        const uint2 o = tl.order[slot];
        vid = (int)(o.x & kOrderTileMask);
        dense = (o.x & (forward ? kOrderDenseFwd : kOrderDenseBwd)) != 0u;
        n = o.y;
        beg = (uint32_t)vid * tl.cap;
        return vid < RT;
    }
    vid = slot;
    if (vid >= RT) return false;
    tile_range(tl, (size_t)vid, beg, n);
    dense = tile_is_dense(tile_flags[vid], n, dense_thr);
    return true;
}

}  // namespace spf
