// Tile binning for gfx950: scan of per-tile counts, scatter-with-keys into per-tile bins
// (an MSD counting pass on the tile id) and a per-tile depth sort done entirely in LDS.
//
// The classic formulation (duplicate-with-keys, then one global radix sort of 64-bit
// tile|depth keys: SURVEY.md section 8a R2-R5) moves every pair through HBM once per radix pass.
// Here the tile digit is resolved by a counting pass (atomics on R*T counters, which live in L2)
// and the remaining depth order is resolved inside the 160 KiB LDS of one CU, so a pair crosses
// HBM twice (bin write, sorted write) however long the key is.
//
// Result contract (Appendix B #10): every tile's list is ordered by (depth bits, Gaussian index)
// ascending.  Keys are unique, so the result does not depend on the (non-deterministic) order in
// which the atomics fill a bin.
#include <atomic>

#include "spf_common.h"

namespace spf {

uint32_t dense_threshold();     // render.hip
uint32_t dense_threshold_fwd();

// ---- scan of tile counts: single block, n = R*T is small (<= a few 10^5) --------------------
constexpr int kScanThreads = 1024;

// One 1024-thread block: exclusive scan of the per-tile counts (-> tile_start, zeroed tile_fill, D, longest list,
// number of dense tiles) and of the per-block pair totals (-> blk_base).  Every wave owns one contiguous segment of
// each array and walks it 64 elements at a time, so every load and store of a wave is one contiguous 256 bytes (a
// thread-chunked layout costs 8x the L1 transactions: 24 us instead of 6); pass 1 sums the segments, pass 2 re-reads
// them (cache hits) and writes the running offsets from a DPP prefix sum per step.
__global__ __launch_bounds__(kScanThreads) void spf_tile_scan_kernel(const uint32_t* __restrict__ count,
                                                                     uint32_t* __restrict__ start,
                                                                     uint32_t* __restrict__ fill,
                                                                     const uint32_t* __restrict__ flags,
                                                                     uint32_t dense_thr,
                                                                     uint32_t* __restrict__ counters, int n,
                                                                     const uint32_t* __restrict__ blk_total,
                                                                     uint32_t* __restrict__ blk_base, int nb) {
    constexpr int kWaves = kScanThreads / kWave;
    __shared__ uint32_t s_a[kWaves], s_b[kWaves], s_mx[kWaves], s_dn[kWaves];
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
    const int sega = ((n + kWaves - 1) / kWaves + kWave - 1) / kWave * kWave;      // multiples of 64 elements
    const int segb = ((nb + kWaves - 1) / kWaves + kWave - 1) / kWave * kWave;
    const int a0 = min(n, wave * sega), a1 = min(n, a0 + sega), b0 = min(nb, wave * segb), b1 = min(nb, b0 + segb);
    uint32_t sa = 0, sb = 0, mx = 0, dn = 0;
    for (int i = a0 + lane; i < a1; i += kWave) {
        const uint32_t c = count[i];
        sa += c;
        mx = max(mx, c);
        dn += tile_is_dense(flags[i], c, dense_thr) ? 1u : 0u;
    }
    for (int i = b0 + lane; i < b1; i += kWave) sb += blk_total[i];
    sa = wave_sum_u32(sa); sb = wave_sum_u32(sb); mx = wave_max_u32(mx); dn = wave_sum_u32(dn);
    if (lane == 0) { s_a[wave] = sa; s_b[wave] = sb; s_mx[wave] = mx; s_dn[wave] = dn; }
    __syncthreads();
    uint32_t runa = 0, runb = 0, total = 0, gmax = 0, dense = 0;
#pragma unroll
    for (int w = 0; w < kWaves; ++w) {
        if (w < wave) { runa += s_a[w]; runb += s_b[w]; }
        total += s_a[w];
        gmax = max(gmax, s_mx[w]);
        dense += s_dn[w];
    }
    for (int i0 = a0; i0 < a1; i0 += kWave) {
        const int i = i0 + lane;
        const uint32_t c = i < a1 ? count[i] : 0u;
        const uint32_t inc = wave_iscan_u32(c);
        if (i < a1) { start[i] = runa + inc - c; fill[i] = 0u; }
        runa += (uint32_t)__builtin_amdgcn_readlane((int)inc, kWave - 1);
    }
    for (int i0 = b0; i0 < b1; i0 += kWave) {
        const int i = i0 + lane;
        const uint32_t c = i < b1 ? blk_total[i] : 0u;
        const uint32_t inc = wave_iscan_u32(c);
        if (i < b1) blk_base[i] = runb + inc - c;
        runb += (uint32_t)__builtin_amdgcn_readlane((int)inc, kWave - 1);
    }
    if (threadIdx.x == 0) {
        start[n] = total;
        counters[0] = total;     // D
        counters[1] = gmax;      // longest tile list
        counters[2] = 0;         // plan verdict (set by the binning kernel)
        counters[3] = dense;     // tiles the dense "rows" render kernels take
    }
}

// The same scan with one block PER RENDER (grid = R): block r first adds up the tile counts of the renders before it
// (its base -- at most R*T coalesced reads that hit the L2), then scans only its own T tile counts and its own nb block
// totals (both add up to the render's pairs, so one base serves both).  R blocks instead of one: the single-block scan
// is pure latency (13 us for 8,192 + 8,192 counters on the bench step, 3 % of it) and it is on the critical path.
// counters[1] / counters[3] are combined with atomics, so counters[0..3] must be ZERO on entry.
__global__ __launch_bounds__(kScanThreads) void spf_tile_scan_render_kernel(const uint32_t* __restrict__ count,
                                                                            uint32_t* __restrict__ start,
                                                                            const uint32_t* __restrict__ flags,
                                                                            uint32_t dense_thr,
                                                                            uint32_t* __restrict__ counters, int T,
                                                                            const uint32_t* __restrict__ blk_total,
                                                                            uint32_t* __restrict__ blk_base, int nb) {
    constexpr int kWaves = kScanThreads / kWave;
    __shared__ uint32_t s_red[kWaves], s_a[kWaves], s_b[kWaves], s_mx[kWaves], s_dn[kWaves];
    const int r = blockIdx.x, R = gridDim.x;
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
    // base = pairs of the renders before this one
    uint32_t pre = 0;
    for (int i = threadIdx.x; i < r * T; i += kScanThreads) pre += count[i];
    pre = wave_sum_u32(pre);
    if (lane == 0) s_red[wave] = pre;
    count += (size_t)r * T; flags += (size_t)r * T; start += (size_t)r * T;
    blk_total += (size_t)r * nb; blk_base += (size_t)r * nb;
    const int sega = ((T + kWaves - 1) / kWaves + kWave - 1) / kWave * kWave;      // multiples of 64 elements
    const int segb = ((nb + kWaves - 1) / kWaves + kWave - 1) / kWave * kWave;
    const int a0 = min(T, wave * sega), a1 = min(T, a0 + sega), b0 = min(nb, wave * segb), b1 = min(nb, b0 + segb);
    uint32_t sa = 0, sb = 0, mx = 0, dn = 0;
    for (int i = a0 + lane; i < a1; i += kWave) {
        const uint32_t c = count[i];
        sa += c;
        mx = max(mx, c);
        dn += tile_is_dense(flags[i], c, dense_thr) ? 1u : 0u;
    }
    for (int i = b0 + lane; i < b1; i += kWave) sb += blk_total[i];
    sa = wave_sum_u32(sa); sb = wave_sum_u32(sb); mx = wave_max_u32(mx); dn = wave_sum_u32(dn);
    if (lane == 0) { s_a[wave] = sa; s_b[wave] = sb; s_mx[wave] = mx; s_dn[wave] = dn; }
    __syncthreads();
    uint32_t base = 0, runa = 0, runb = 0, total = 0, gmax = 0, dense = 0;
#pragma unroll
    for (int w = 0; w < kWaves; ++w) {
        base += s_red[w];
        if (w < wave) { runa += s_a[w]; runb += s_b[w]; }
        total += s_a[w];
        gmax = max(gmax, s_mx[w]);
        dense += s_dn[w];
    }
    runa += base; runb += base;
    for (int i0 = a0; i0 < a1; i0 += kWave) {
        const int i = i0 + lane;
        const uint32_t c = i < a1 ? count[i] : 0u;
        const uint32_t inc = wave_iscan_u32(c);
        if (i < a1) start[i] = runa + inc - c;
        runa += (uint32_t)__builtin_amdgcn_readlane((int)inc, kWave - 1);
    }
    for (int i0 = b0; i0 < b1; i0 += kWave) {
        const int i = i0 + lane;
        const uint32_t c = i < b1 ? blk_total[i] : 0u;
        const uint32_t inc = wave_iscan_u32(c);
        if (i < b1) blk_base[i] = runb + inc - c;
        runb += (uint32_t)__builtin_amdgcn_readlane((int)inc, kWave - 1);
    }
    if (threadIdx.x == 0) {
        if (gmax) atomicMax(&counters[1], gmax);      // longest tile list
        if (dense) atomicAdd(&counters[3], dense);    // tiles the dense "rows" render kernels take
        if (r == R - 1) {
            start[T] = base + total;                  // = tile_start[R*T]
            counters[0] = base + total;               // D
        }
    }
}

// ---- scatter-with-keys: block = 256 consecutive Gaussians of one render -----------------------------
// Three phases per block, all tile bookkeeping in LDS: (1) count the block's pairs per tile, (2) reserve one
// contiguous range per touched tile with ONE global atomic, (3) hand out slots inside the range with LDS
// atomics and write the keys.  lds = 0 (more than kMaxLdsTiles tiles): one global atomic per pair.
__global__ __launch_bounds__(kBlock) void spf_bin_pairs_kernel(const float* __restrict__ zkey,
                                                               const uint32_t* __restrict__ rect,
                                                               const uint32_t* __restrict__ tile_start,
                                                               uint32_t* __restrict__ tile_fill,
                                                               uint32_t* __restrict__ counters,
                                                               uint64_t* __restrict__ pairs, uint64_t capacity,
                                                               const uint32_t* __restrict__ blk_base,
                                                               uint32_t* __restrict__ pair_off,
                                                               int G, int T, int tiles_x, int lds,
                                                               uint32_t max_tile_hint) {
    extern __shared__ __attribute__((aligned(16))) uint32_t s_bin[];   // [T] counts, [T] bases
    __shared__ uint32_t s_wtot[4];
    uint32_t* s_cnt = s_bin;
    uint32_t* s_base = s_bin + T;
    const int r = blockIdx.y;
    const int g = blockIdx.x * kBlock + threadIdx.x;
    // The host may launch this chain from a PLAN (capacity, longest list of an earlier call) instead of reading the
    // counters back; the plan is checked here, on the device.  Flag bits: 1 = pair buffer too small (nothing is
    // rendered), 2 = a tile list is longer than planned (it was not sorted).
    if (r == 0 && g == 0) {
        uint32_t flag = counters[0] > capacity ? 1u : 0u;
        if (max_tile_hint != 0u && counters[1] > max_tile_hint) flag |= 2u;
        if (flag) counters[2] = flag;
    }
    if (counters[0] > capacity) return;
    const bool live = g < G;
    const size_t rg = (size_t)r * G + (live ? g : 0);
    const size_t tb = (size_t)r * T;
    // The block is a latency chain (65 % of its wave-cycles were spent waiting), so every global read is issued up
    // front -- the Gaussian's rect and depth key, the block's pair base, and the start of EVERY tile this thread may have
    // to reserve a range in later (tiles t = tid, tid + 256, ...: up to four per thread, i.e. T <= 1024; 256 x 256 px
    // is one) -- and the pair numbering shares its barrier with the clearing of the tile counters.
    const uint32_t rc = live ? rect[rg] : 0u;
    const float zk = live ? zkey[rg] : 0.f;
    const uint32_t bbase = blk_base[(size_t)r * gridDim.x + blockIdx.x];
    constexpr int kPre = 4;
    uint32_t ts_pre[kPre] = {0u, 0u, 0u, 0u};
    const bool pre = lds && T <= kPre * kBlock;
    if (pre) {
#pragma unroll
        for (int k = 0; k < kPre; ++k)
            if (threadIdx.x + k * kBlock < T) ts_pre[k] = tile_start[tb + threadIdx.x + k * kBlock];
    }
    const int x0 = rc & 0xff, y0 = (rc >> 8) & 0xff, x1 = (rc >> 16) & 0xff, y1 = rc >> 24;
    const bool any = x1 > x0 && y1 > y0;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // Gaussian-major pair numbering: exclusive scan of pairs-per-Gaussian inside the block + block base
    const uint32_t cnt = any ? (uint32_t)((x1 - x0) * (y1 - y0)) : 0u;
    const uint32_t inc = wave_iscan_u32(cnt);
    if (lane == kWave - 1) s_wtot[wave] = inc;
    if (lds)
        for (int t = threadIdx.x; t < T; t += kBlock) s_cnt[t] = 0;
    __syncthreads();
    {
        uint32_t off = bbase + inc - cnt;
        for (int w = 0; w < wave; ++w) off += s_wtot[w];
        if (live) reinterpret_cast<uint2*>(pair_off)[rg] = make_uint2(rc, off);      // (rect, first pair): SpfState.pair_off
    }
    const uint64_t key = any ? (((uint64_t)__float_as_uint(zk) << 32) | (uint32_t)g) : 0ull;
    if (!lds) {
        if (any)
            for (int ty = y0; ty < y1; ++ty)
                for (int tx = x0; tx < x1; ++tx) {
                    const size_t t = tb + ty * tiles_x + tx;
                    pairs[tile_start[t] + atomicAdd(&tile_fill[t], 1u)] = key;
                }
        return;
    }
    // Gaussians with exactly one tile (93 % of a pixel-aligned scene) act run-wise, through the first lane of every run
    // of neighbouring lanes in the same tile: one LDS atomic per run instead of one per lane on the same address.
    const bool single = any && cnt == 1u;
    const int stile = single ? y0 * tiles_x + x0 : -1;
    const LaneRun run = lane_runs(stile, lane);
    const bool run_head = single && run.head == lane;
    if (run_head) atomicAdd(&s_cnt[stile], (uint32_t)run.len);
    if (any && !single)
        for (int ty = y0; ty < y1; ++ty)
            for (int tx = x0; tx < x1; ++tx) atomicAdd(&s_cnt[ty * tiles_x + tx], 1u);
    __syncthreads();
    if (pre) {
#pragma unroll
        for (int k = 0; k < kPre; ++k) {
            const int t = threadIdx.x + k * kBlock;
            if (t < T) {
                const uint32_t c = s_cnt[t];
                if (c) {
                    s_base[t] = ts_pre[k] + atomicAdd(&tile_fill[tb + t], c);
                    s_cnt[t] = 0;
                }
            }
        }
    } else {
        for (int t = threadIdx.x; t < T; t += kBlock) {
            const uint32_t c = s_cnt[t];
            if (c) {
                s_base[t] = tile_start[tb + t] + atomicAdd(&tile_fill[tb + t], c);
                s_cnt[t] = 0;
            }
        }
    }
    __syncthreads();
    uint32_t first = 0u;
    if (run_head) first = atomicAdd(&s_cnt[stile], (uint32_t)run.len);       // the run's range inside the block's range
    first = (uint32_t)__shfl((int)first, run.head, kWave);
    if (single) pairs[s_base[stile] + first + (uint32_t)(lane - run.head)] = key;
    if (any && !single)
        for (int ty = y0; ty < y1; ++ty)
            for (int tx = x0; tx < x1; ++tx) {
                const int t = ty * tiles_x + tx;
                pairs[s_base[t] + atomicAdd(&s_cnt[t], 1u)] = key;
            }
}

// ---- the same scatter for VB views of a scene at once ------------------------------------------------------------
// The block above is a latency chain -- loads, barrier, LDS count, barrier, one global atomic round trip, barrier, LDS
// slots, store -- and a launch of (G / 256) x R such blocks is many rounds of them.  A scene's V views project the SAME 256
// Gaussians, so one block can walk the chain once for VB of them: VB histograms in LDS, VB rects and depth keys per
// thread, all their global atomics in one round trip.  V / VB times fewer blocks, the same number of barriers per block --
// but the LDS phases of a block grow with VB, so it only pays for renders of many blocks (see launch_bin_pairs).
template <int VB>
__global__ __launch_bounds__(kBlock) void spf_bin_pairs_views_kernel(const float* __restrict__ zkey,
                                                                     const uint32_t* __restrict__ rect,
                                                                     const uint32_t* __restrict__ tile_start,
                                                                     uint32_t* __restrict__ tile_fill,
                                                                     uint32_t* __restrict__ counters,
                                                                     uint64_t* __restrict__ pairs, uint64_t capacity,
                                                                     const uint32_t* __restrict__ blk_base,
                                                                     uint32_t* __restrict__ pair_off,
                                                                     int G, int T, int tiles_x, int V, int nvb,
                                                                     uint32_t max_tile_hint) {
    extern __shared__ __attribute__((aligned(16))) uint32_t s_bin[];   // [VB][T] counts, [VB][T] bases
    __shared__ uint32_t s_wtot[VB][4];
    uint32_t* const s_cnt = s_bin;
    uint32_t* const s_base = s_bin + VB * T;
    const int sc = blockIdx.y / nvb, vb = blockIdx.y - sc * nvb;
    const int v0 = vb * VB, nv = min(VB, V - v0);
    const int r0 = sc * V + v0;                                        // first render of this block
    const int g = blockIdx.x * kBlock + threadIdx.x;
    if (blockIdx.y == 0 && g == 0) {                                   // the plan check (see the kernel above)
        uint32_t flag = counters[0] > capacity ? 1u : 0u;
        if (max_tile_hint != 0u && counters[1] > max_tile_hint) flag |= 2u;
        if (flag) counters[2] = flag;
    }
    if (counters[0] > capacity) return;
    const bool live = g < G;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int kPre = 4;                                            // tiles t = tid + 256 j of every view: T <= 1024
    uint32_t rc[VB], bbase[VB], ts_pre[VB][kPre];
    float zk[VB];
#pragma unroll
    for (int k = 0; k < VB; ++k) {
        const bool on = k < nv;
        const size_t rg = (size_t)(r0 + (on ? k : 0)) * G + (live ? g : 0);
        rc[k] = (on && live) ? rect[rg] : 0u;
        zk[k] = (on && live) ? zkey[rg] : 0.f;
        bbase[k] = on ? blk_base[(size_t)(r0 + k) * gridDim.x + blockIdx.x] : 0u;
#pragma unroll
        for (int j = 0; j < kPre; ++j) {
            const int t = threadIdx.x + j * kBlock;
            ts_pre[k][j] = (on && t < T) ? tile_start[(size_t)(r0 + k) * T + t] : 0u;
        }
    }
    uint32_t cnt[VB], inc[VB];
#pragma unroll
    for (int k = 0; k < VB; ++k) {
        const int x0 = rc[k] & 0xff, y0 = (rc[k] >> 8) & 0xff, x1 = (rc[k] >> 16) & 0xff, y1 = rc[k] >> 24;
        cnt[k] = (x1 > x0 && y1 > y0) ? (uint32_t)((x1 - x0) * (y1 - y0)) : 0u;
        inc[k] = wave_iscan_u32(cnt[k]);
        if (lane == kWave - 1) s_wtot[k][wave] = inc[k];
    }
    for (int t = threadIdx.x; t < VB * T; t += kBlock) s_cnt[t] = 0;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < VB; ++k) {
        uint32_t off = bbase[k] + inc[k] - cnt[k];
        for (int w = 0; w < wave; ++w) off += s_wtot[k][w];
        if (live && k < nv) reinterpret_cast<uint2*>(pair_off)[(size_t)(r0 + k) * G + g] = make_uint2(rc[k], off);
    }
    // ---- count: run-wise for single-tile Gaussians (see the kernel above), one view after the other ----
    LaneRun run[VB];
    int stile[VB];
#pragma unroll
    for (int k = 0; k < VB; ++k) {
        const int x0 = rc[k] & 0xff, y0 = (rc[k] >> 8) & 0xff, x1 = (rc[k] >> 16) & 0xff, y1 = rc[k] >> 24;
        stile[k] = cnt[k] == 1u ? y0 * tiles_x + x0 : -1;
        run[k] = lane_runs(stile[k], lane);
        uint32_t* __restrict__ c = s_cnt + k * T;
        if (cnt[k] == 1u && run[k].head == lane) atomicAdd(&c[stile[k]], (uint32_t)run[k].len);
        if (cnt[k] > 1u)
            for (int ty = y0; ty < y1; ++ty)
                for (int tx = x0; tx < x1; ++tx) atomicAdd(&c[ty * tiles_x + tx], 1u);
    }
    __syncthreads();
    // ---- reserve: every touched (view, tile) of the block gets its range with ONE global atomic; all in one round trip ----
#pragma unroll
    for (int k = 0; k < VB; ++k)
#pragma unroll
        for (int j = 0; j < kPre; ++j) {
            const int t = threadIdx.x + j * kBlock;
            if (k < nv && t < T) {
                const uint32_t c = s_cnt[k * T + t];
                if (c) {
                    s_base[k * T + t] = ts_pre[k][j] + atomicAdd(&tile_fill[(size_t)(r0 + k) * T + t], c);
                    s_cnt[k * T + t] = 0;
                }
            }
        }
    __syncthreads();
    // ---- slots inside the ranges, keys out ----
#pragma unroll
    for (int k = 0; k < VB; ++k) {
        const int x0 = rc[k] & 0xff, y0 = (rc[k] >> 8) & 0xff, x1 = (rc[k] >> 16) & 0xff, y1 = rc[k] >> 24;
        const uint64_t key = ((uint64_t)__float_as_uint(zk[k]) << 32) | (uint32_t)g;
        uint32_t* __restrict__ c = s_cnt + k * T;
        const uint32_t* __restrict__ bs = s_base + k * T;
        const bool single = cnt[k] == 1u;
        uint32_t first = 0u;
        if (single && run[k].head == lane) first = atomicAdd(&c[stile[k]], (uint32_t)run[k].len);
        first = (uint32_t)__shfl((int)first, run[k].head, kWave);          // (every lane takes part: uniform flow)
        if (single) pairs[bs[stile[k]] + first + (uint32_t)(lane - run[k].head)] = key;
        if (cnt[k] > 1u)
            for (int ty = y0; ty < y1; ++ty)
                for (int tx = x0; tx < x1; ++tx) {
                    const int t = ty * tiles_x + tx;
                    pairs[bs[t] + atomicAdd(&c[t], 1u)] = key;
                }
    }
}

// ---- launch order of the composite lists kernels (direct bins; see tile_order_ptr in spf_common.h) --------------------
// One block per window of an XCD's range [x * per, (x + 1) * per), per = RT / 8 -- the range xcd_remap gives XCD x --
// sorts its tiles by list length into 64 classes, longest first (counting sort in LDS: histogram, scan, scatter; the order inside a class
// is whatever the LDS atomics make it: nothing depends on it).  These blocks ride in FRONT of the tile sort's first
// kernel (they only read the tile counters, and a launch of their own is 4 - 6 us on the critical path; forked beside
// the sort the two cross-stream edges cost 12).
constexpr int kOrderClasses = 64;
constexpr int kOrderWindow = 256;        // tiles ordered together
static_assert(kOrderClasses == kWave, "the class bases are one wave's prefix sum");
// Only the LAST window (256 tiles -- the last round or two of blocks on an XCD) of an XCD's range is ordered, the tiles
// before it keep the image order: what the order has to fix is the END of the launch, and image order keeps the blocks
// that run together on neighbouring tiles, whose lists share Gaussians.  Measured on C2 (lists backward, same box):
// image order 153.9 us / 298 MB of HBM traffic per launch; the whole range of 1,024 tiles ordered 143.8 us / 456 MB;
// every window of 256 ordered 145.9 / 339; only the last window 142.7 us / 310 MB.  (64 scenes x 4 views with whole
// ranges of 8,192 tiles ordered: -6 %.)
__host__ __device__ inline int order_windows(int RT) { return ((RT >> 3) + kOrderWindow - 1) / kOrderWindow; }
__device__ __forceinline__ void tile_order_block(int id, const uint32_t* __restrict__ count,
                                                 const uint32_t* __restrict__ flags, uint2* __restrict__ order, int RT,
                                                 int T, uint32_t cap, uint32_t dense_thr, uint32_t dense_thr_fwd) {
    auto dense_bits = [&](uint32_t f, uint32_t n) {
        return (tile_is_dense(f, n, dense_thr) ? kOrderDenseBwd : 0u) | (tile_is_dense(f, n, dense_thr_fwd) ? kOrderDenseFwd : 0u);
    };
    __shared__ uint32_t s_cnt[kOrderClasses];
    const int nwin = order_windows(RT), x = id / nwin, w = id - x * nwin;
    const XcdMap xm = xcd_map(RT, T);                               // (which tiles this XCD's slots stand for)
    const int per = RT >> 3, j0 = w * kOrderWindow, lo = x * per + j0, len = min(kOrderWindow, per - j0);
    const int nthr = (int)blockDim.x;
    if (w + 1 < nwin) {                                            // image order
        for (int i = threadIdx.x; i < len; i += nthr) {
            const int vid = xcd_tile(xm, T, x, j0 + i);
            const uint32_t n = min(count[vid], cap);
            order[lo + i] = make_uint2((uint32_t)vid | dense_bits(flags[vid], n), n);
        }
        return;
    }
    if (threadIdx.x < kOrderClasses) s_cnt[threadIdx.x] = 0u;
    __syncthreads();
    // (the key is the list length.  Measured against footprint sum + 8 per entry, scaled to the range's largest: the
    //  plain length orders better on all four bench configs -- C2 +3.8 % vs +2.5 %, C3 +5.4 / +4.3, C5 +1.5 / +0.9)
    auto cls = [&](uint32_t n) {      // 0 = the longest lists
        return (uint32_t)(kOrderClasses - 1) - min((uint32_t)(kOrderClasses - 1), n * (uint32_t)kOrderClasses / (cap + 1u));
    };
    for (int i = threadIdx.x; i < len; i += nthr) atomicAdd(&s_cnt[cls(min(count[xcd_tile(xm, T, x, j0 + i)], cap))], 1u);
    __syncthreads();
    uint32_t base = 0u;
    if (threadIdx.x < kWave) {
        const uint32_t c = s_cnt[threadIdx.x];
        base = wave_iscan_u32(c) - c;
    }
    __syncthreads();
    if (threadIdx.x < kWave) s_cnt[threadIdx.x] = base;          // (now: next free slot of the class)
    __syncthreads();
    for (int i = threadIdx.x; i < len; i += nthr) {
        const int vid = xcd_tile(xm, T, x, j0 + i);
        const uint32_t n = min(count[vid], cap);
        const uint32_t pos = atomicAdd(&s_cnt[cls(n)], 1u);
        order[lo + pos] = make_uint2((uint32_t)vid | dense_bits(flags[vid], n), n);
    }
}
// (on its own: when no tile has more than one entry, the sort launches nothing)
__global__ __launch_bounds__(kBlock) void spf_tile_order_kernel(const uint32_t* __restrict__ count,
                                                                const uint32_t* __restrict__ flags,
                                                                uint2* __restrict__ order, int RT, int T, uint32_t cap,
                                                                uint32_t dense_thr, uint32_t dense_thr_fwd) {
    tile_order_block((int)blockIdx.x, count, flags, order, RT, T, cap, dense_thr, dense_thr_fwd);
}

// ---- per-tile sort in LDS ---------------------------------------------------------------------
// Bitonic network in its "all comparators ascending" form (first step of every merge compares
// i with i ^ (k-1), the rest with i ^ j): it needs no padding, because a missing partner above n
// behaves as +infinity and an ascending comparator never moves +infinity.
template <int THREADS>
__global__ __launch_bounds__(THREADS) void spf_sort_tiles_lds_kernel(TileLists tl,
                                                                     const uint32_t* __restrict__ counters,
                                                                     uint64_t* __restrict__ pairs,
                                                                     uint64_t capacity, uint32_t lo, uint32_t hi) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    uint64_t* s = reinterpret_cast<uint64_t*>(smem_raw);
    if (counters[0] > capacity) return;
    uint32_t b, n;
    tile_range(tl, blockIdx.x, b, n);
    if (n <= lo || n > hi) return;
    uint64_t* __restrict__ p = pairs + b;
    for (uint32_t i = threadIdx.x; i < n; i += THREADS) s[i] = p[i];
    __syncthreads();
    uint32_t m = 1;
    while (m < n) m <<= 1;
    const uint32_t half = m >> 1;
    for (uint32_t k = 2; k <= m; k <<= 1) {
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            const uint32_t mask = (j == (k >> 1)) ? (k - 1) : j;
            for (uint32_t i = threadIdx.x; i < half; i += THREADS) {
                // i-th comparator of this step: insert a 0 bit at position log2(j)
                const uint32_t a = ((i & ~(j - 1)) << 1) | (i & (j - 1));
                const uint32_t c = a ^ mask;
                if (c < n) {
                    const uint64_t x = s[a], y = s[c];
                    if (x > y) { s[a] = y; s[c] = x; }
                }
            }
            __syncthreads();
        }
    }
    for (uint32_t i = threadIdx.x; i < n; i += THREADS) p[i] = s[i];
}

// ---- per-tile sort in registers: one wave per tile, E keys per lane -----------------------------------
// Lists of up to 64*E entries.  Lane l holds elements E*l .. E*l+E-1 (missing ones are +infinity), so the steps of
// the network with partner distance j < E are compare-exchanges between a lane's own registers, the others are lane
// exchanges (partner lane = lane ^ mask: DPP / ds_bpermute, no LDS storage, no barrier).  Same all-ascending
// network as above, fixed size 64*E: keys are unique, so the result is the same permutation the LDS kernel produces.
__device__ __forceinline__ void cswap_u64(uint64_t& a, uint64_t& b) {
    const bool sw = a > b;
    const uint64_t lo = sw ? b : a, hi = sw ? a : b;
    a = lo;
    b = hi;
}

// value of lane (l ^ MASK).  Most masks of the network are a DPP pattern (a VALU move inside a row of 16 lanes);
// only 16 / 31 / 32 / 63 go through ds_bpermute.
template <int CTRL>
__device__ __forceinline__ uint32_t dpp_mov_u32(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, 0xf, 0xf, true);   // every lane has a source
}
template <int MASK>
__device__ __forceinline__ uint32_t lane_xor_u32(uint32_t v) {
    if constexpr (MASK == 1) return dpp_mov_u32<0xB1>(v);                           // quad_perm [1,0,3,2]
    else if constexpr (MASK == 2) return dpp_mov_u32<0x4E>(v);                      // quad_perm [2,3,0,1]
    else if constexpr (MASK == 3) return dpp_mov_u32<0x1B>(v);                      // quad_perm [3,2,1,0]
    else if constexpr (MASK == 4) return dpp_mov_u32<0x1B>(dpp_mov_u32<0x141>(v));  // (l ^ 7) ^ 3
    else if constexpr (MASK == 7) return dpp_mov_u32<0x141>(v);                     // row_half_mirror
    else if constexpr (MASK == 8) return dpp_mov_u32<0x128>(v);                     // row_ror:8
    else if constexpr (MASK == 15) return dpp_mov_u32<0x140>(v);                    // row_mirror
    else return (uint32_t)__shfl_xor((int)v, MASK, kWave);
}
template <int MASK>
__device__ __forceinline__ uint64_t lane_xor_u64(uint64_t v) {
    return ((uint64_t)lane_xor_u32<MASK>((uint32_t)(v >> 32)) << 32) | lane_xor_u32<MASK>((uint32_t)v);
}

// compare-exchanges between a lane's own keys: partner distance J, or the mirror step of a merge of size K
template <int E, int J>
__device__ __forceinline__ void thread_steps_down(uint64_t (&r)[E]) {
    if constexpr (J >= 1) {
#pragma unroll
        for (int s = 0; s < E; ++s)
            if (s < (s ^ J)) cswap_u64(r[s], r[s ^ J]);
        thread_steps_down<E, J / 2>(r);
    }
}
template <int E, int K>
__device__ __forceinline__ void merges_inside(uint64_t (&r)[E]) {
    if constexpr (K <= E) {
#pragma unroll
        for (int s = 0; s < E; ++s)
            if (s < (s ^ (K - 1))) cswap_u64(r[s], r[s ^ (K - 1)]);
        thread_steps_down<E, K / 4>(r);
        merges_inside<E, 2 * K>(r);
    }
}
// exchange with lane (l ^ MASK); a lane keeps the smaller key of a pair iff it is the lower lane.  MIRROR: first step
// of a merge (partner index = i ^ (k-1)): the partner's keys come in reverse order.
template <int E, int MASK, bool MIRROR>
__device__ __forceinline__ void lane_step(uint64_t (&r)[E], int lane) {
    const bool lower = (lane & (MIRROR ? (MASK + 1) >> 1 : MASK)) == 0;
    uint64_t o[E];
#pragma unroll
    for (int s = 0; s < E; ++s) o[s] = lane_xor_u64<MASK>(r[MIRROR ? E - 1 - s : s]);
#pragma unroll
    for (int s = 0; s < E; ++s) r[s] = ((o[s] < r[s]) == lower) ? o[s] : r[s];
}
template <int E, int J>
__device__ __forceinline__ void lane_steps_down(uint64_t (&r)[E], int lane) {
    if constexpr (J >= E) {
        lane_step<E, J / E, false>(r, lane);
        lane_steps_down<E, J / 2>(r, lane);
    }
}
template <int E, int K>
__device__ __forceinline__ void merges_across(uint64_t (&r)[E], int lane) {
    if constexpr (K <= kWave * E) {
        lane_step<E, K / E - 1, true>(r, lane);
        lane_steps_down<E, K / 4>(r, lane);
        thread_steps_down<E, E / 2>(r);
        merges_across<E, 2 * K>(r, lane);
    }
}

template <int E>
__device__ __forceinline__ void sort_tile_in_wave(uint64_t* __restrict__ p, uint32_t n) {
    const int lane = threadIdx.x & (kWave - 1);
    uint64_t r[E];
#pragma unroll
    for (int s = 0; s < E; ++s) {
        const uint32_t e = (uint32_t)(E * lane + s);
        r[s] = e < n ? p[e] : ~0ull;
    }
    merges_inside<E, 2>(r);             // merges of size <= E stay inside a lane
    merges_across<E, 2 * E>(r, lane);   // sizes 2E .. 64E
#pragma unroll
    for (int s = 0; s < E; ++s) {
        const uint32_t e = (uint32_t)(E * lane + s);
        if (e < n) p[e] = r[s];
    }
}

// Four tiles per 256-thread block (one per wave).  SMALL: lists of 2..64*E entries, keys-per-lane (2/4/8/16 <= E)
// chosen per tile; otherwise one size class (lo, 64*E].
template <int E, bool SMALL>
__global__ __launch_bounds__(kBlock) void spf_sort_tiles_wave_kernel(TileLists tl, const uint32_t* __restrict__ flags,
                                                                     uint32_t* __restrict__ counters,
                                                                     uint64_t* __restrict__ pairs, uint64_t capacity,
                                                                     uint32_t lo, int RT,
                                                                     uint32_t dense_thr, uint2* __restrict__ order, int T, uint32_t dense_thr_fwd) {
    if (counters[0] > capacity) return;
    const int ob = order ? 8 * order_windows(RT) : 0;
    if ((int)blockIdx.x < ob) {
        tile_order_block((int)blockIdx.x, tl.count, flags, order, RT, T, tl.cap, dense_thr, dense_thr_fwd);
        return;
    }
    const int tile = ((int)blockIdx.x - ob) * (kBlock / kWave) + (threadIdx.x >> 6);
    if (tile >= RT) return;
    uint32_t b, n;
    tile_range(tl, tile, b, n);
    if (n <= lo || n > (uint32_t)(kWave * E)) return;
    if (SMALL) {
        if (n <= 2u * kWave) sort_tile_in_wave<2>(pairs + b, n);
        else if (n <= 4u * kWave) sort_tile_in_wave<4>(pairs + b, n);
        else if (E <= 8 || n <= 8u * kWave) sort_tile_in_wave<8>(pairs + b, n);
        else sort_tile_in_wave<16>(pairs + b, n);
    } else {
        sort_tile_in_wave<E>(pairs + b, n);
    }
}

// ---- per-tile sort, one 256-thread block per tile, E keys per thread (lists of 513 .. 256*E entries) ----------------
// The same register network with four waves on one list: every wave first sorts its 64*E-key chunk exactly as the
// wave kernel does; the last two merges (128*E, 256*E) start with steps whose partner sits in another wave -- those
// three exchanges go through LDS (slot-major, conflict-free), everything below stays lane exchanges and register
// compare-exchanges.  A single wave on such a list (16 / 32 keys per lane) is a long serial chain and leaves most of
// the chip idle when the lists are long because the tiles are few (BASELINE config 3: 2,048 tiles of ~1,000 entries).
template <int E, bool MIRROR>
__device__ __forceinline__ void block_step(uint64_t (&r)[E], uint64_t* __restrict__ s, int t, int tmask) {
#pragma unroll
    for (int k = 0; k < E; ++k) s[k * kBlock + t] = r[k];
    __syncthreads();
    const int pt = t ^ tmask;
    const bool lower = (t & (MIRROR ? (tmask + 1) >> 1 : tmask)) == 0;
#pragma unroll
    for (int k = 0; k < E; ++k) {
        const uint64_t o = s[(MIRROR ? E - 1 - k : k) * kBlock + pt];
        r[k] = ((o < r[k]) == lower) ? o : r[k];
    }
    __syncthreads();
}

template <int E>
__device__ __forceinline__ void sort_tile_in_block(uint64_t* __restrict__ p, uint32_t n, uint64_t* __restrict__ s_x) {
    const int t = threadIdx.x, lane = t & (kWave - 1);
    uint64_t r[E];
#pragma unroll
    for (int k = 0; k < E; ++k) {
        const uint32_t e = (uint32_t)(E * t + k);
        r[k] = e < n ? p[e] : ~0ull;
    }
    merges_inside<E, 2>(r);
    merges_across<E, 2 * E>(r, lane);                 // every wave: its 64*E keys sorted
    block_step<E, true>(r, s_x, t, 127);              // merge of 128*E: mirror step across the wave pair ...
    lane_steps_down<E, 32 * E>(r, lane);              // ... the rest inside the wave
    thread_steps_down<E, E / 2>(r);
    block_step<E, true>(r, s_x, t, 255);              // merge of 256*E: mirror step across the block,
    block_step<E, false>(r, s_x, t, 64);              // partner distance 64*E across the wave pair,
    lane_steps_down<E, 32 * E>(r, lane);              // the rest inside the wave
    thread_steps_down<E, E / 2>(r);
#pragma unroll
    for (int k = 0; k < E; ++k) {
        const uint32_t e = (uint32_t)(E * t + k);
        if (e < n) p[e] = r[k];
    }
}

template <int E>
__global__ __launch_bounds__(kBlock) void spf_sort_tiles_block_kernel(TileLists tl,
                                                                      const uint32_t* __restrict__ counters,
                                                                      uint64_t* __restrict__ pairs, uint64_t capacity,
                                                                      uint32_t lo) {
    __shared__ uint64_t s_x[E * kBlock];
    if (counters[0] > capacity) return;
    uint32_t b, n;
    tile_range(tl, blockIdx.x, b, n);
    if (n <= lo || n > (uint32_t)(kBlock * E)) return;
    sort_tile_in_block<E>(pairs + b, n, s_x);
}

// ---- per-tile sort, one PAIR of waves per tile (many tiles, lists up to 1024) --------------------------------------
// A launch of one wave per tile is a single round of waves (eight per SIMD on the 32-render bench step), and a SIMD
// then runs as long as the SUM of the lists it happened to get: 570 instructions for a 256-slot network, 1,320 for 512
// slots, ~3,000 for 1,024 -- the unluckiest SIMD carries +40 %.  Here a list of more than 256 entries is shared by the
// two waves of a 128-thread block (each sorts its half exactly as before, the last merge starts with one exchange
// through LDS), so every wave of the launch carries about the same load; shorter lists use the first wave only.
template <int E, bool MIRROR>
__device__ __forceinline__ void pair_step(uint64_t (&r)[E], uint64_t* __restrict__ s, int t, int tmask) {
    constexpr int NT = 2 * kWave;
#pragma unroll
    for (int k = 0; k < E; ++k) s[k * NT + t] = r[k];
    __syncthreads();
    const int pt = t ^ tmask;
    const bool lower = (t & (MIRROR ? (tmask + 1) >> 1 : tmask)) == 0;
#pragma unroll
    for (int k = 0; k < E; ++k) {
        const uint64_t o = s[(MIRROR ? E - 1 - k : k) * NT + pt];
        r[k] = ((o < r[k]) == lower) ? o : r[k];
    }
}
template <int E>
__device__ __forceinline__ void sort_tile_in_pair(uint64_t* __restrict__ p, uint32_t n, uint64_t* __restrict__ s_x) {
    const int t = threadIdx.x, lane = t & (kWave - 1);
    uint64_t r[E];
#pragma unroll
    for (int k = 0; k < E; ++k) {
        const uint32_t e = (uint32_t)(E * t + k);
        r[k] = e < n ? p[e] : ~0ull;
    }
    merges_inside<E, 2>(r);
    merges_across<E, 2 * E>(r, lane);                 // each wave: its 64*E keys sorted
    pair_step<E, true>(r, s_x, t, 127);               // merge of 128*E: mirror step across the two waves ...
    lane_steps_down<E, 32 * E>(r, lane);              // ... the rest inside the wave
    thread_steps_down<E, E / 2>(r);
#pragma unroll
    for (int k = 0; k < E; ++k) {
        const uint32_t e = (uint32_t)(E * t + k);
        if (e < n) p[e] = r[k];
    }
}
__global__ __launch_bounds__(2 * kWave) void spf_sort_tiles_pair_kernel(TileLists tl, const uint32_t* __restrict__ flags,
                                                                        uint32_t* __restrict__ counters,
                                                                        uint64_t* __restrict__ pairs, uint64_t capacity,
                                                                        int RT, uint32_t dense_thr,
                                                                        uint2* __restrict__ order, int T, uint32_t dense_thr_fwd) {
    __shared__ uint64_t s_x[8 * 2 * kWave];
    if (counters[0] > capacity) return;
    const int ob = order ? 8 * order_windows(RT) : 0;
    if ((int)blockIdx.x < ob) {
        tile_order_block((int)blockIdx.x, tl.count, flags, order, RT, T, tl.cap, dense_thr, dense_thr_fwd);
        return;
    }
    const int tile = (int)blockIdx.x - ob;
    uint32_t b, n;
    tile_range(tl, tile, b, n);
    if (n <= 1u || n > 1024u) return;
    if (n <= 4u * kWave) {                            // one wave is enough (and as fast): the second leaves
        if (threadIdx.x >= kWave) return;
        if (n <= 2u * kWave) sort_tile_in_wave<2>(pairs + b, n);
        else sort_tile_in_wave<4>(pairs + b, n);
        return;
    }
    if (n <= 8u * kWave) sort_tile_in_pair<4>(pairs + b, n, s_x);
    else sort_tile_in_pair<8>(pairs + b, n, s_x);
}

// The three size classes of the "few tiles, long lists" family in ONE launch (lists of 2 .. 2048 entries): blocks
// 0 .. RT-1 take one tile each when its list has 513 .. 2048 entries (four waves on one list: 4 or 8 keys per thread),
// blocks RT .. RT + ceil(RT/4) - 1 take four tiles each, one per wave, when their lists have 2 .. 512 entries.  The
// classes touch disjoint tiles, and each alone is a one-round kernel that leaves most of the chip idle (REF2V: 4,096
// tiles -- 30 % short lists, 70 % of 513 .. 1024, a handful longer: 15 + 23 + 12 us back to back); the long lists are
// dispatched first.  Same networks, same lists bit for bit.
// BIG: a fourth class in front of them -- lists of 2,049 .. 4,096 entries, sixteen keys per thread (32 KB of LDS, 72 VGPRs:
// five blocks per CU) -- for calls whose plan reaches that class: the reference's 10-view shape is 768 tiles of ~2,800
// entries with a few shorter ones at the image border, and as two launches the classes ran one after the other, each on
// a third of the chip (39 + 18 us).
template <bool BIG>
__global__ __launch_bounds__(kBlock) void spf_sort_tiles_mixed_kernel(TileLists tl, const uint32_t* __restrict__ flags,
                                                                      uint32_t* __restrict__ counters,
                                                                      uint64_t* __restrict__ pairs, uint64_t capacity,
                                                                      int RT, uint32_t dense_thr,
                                                                      uint2* __restrict__ order, int T, uint32_t dense_thr_fwd) {
    __shared__ uint64_t s_x[(BIG ? 16 : 8) * kBlock];
    if (counters[0] > capacity) return;
    const int ob = order ? 8 * order_windows(RT) : 0;
    if ((int)blockIdx.x < ob) {
        tile_order_block((int)blockIdx.x, tl.count, flags, order, RT, T, tl.cap, dense_thr, dense_thr_fwd);
        return;
    }
    int blk = (int)blockIdx.x - ob;
    if (BIG) {
        if (blk < RT) {
            uint32_t b, n;
            tile_range(tl, blk, b, n);
            if (n > 2048u && n <= 4096u) sort_tile_in_block<16>(pairs + b, n, s_x);
            return;
        }
        blk -= RT;
    }
    if (blk < RT) {
        uint32_t b, n;
        tile_range(tl, blk, b, n);
        if (n <= 512u || n > 2048u) return;
        if (n <= 1024u) sort_tile_in_block<4>(pairs + b, n, s_x);
        else sort_tile_in_block<8>(pairs + b, n, s_x);
        return;
    }
    const int tile = (blk - RT) * (kBlock / kWave) + (threadIdx.x >> 6);
    if (tile >= RT) return;
    uint32_t b, n;
    tile_range(tl, tile, b, n);
    if (n <= 1u || n > 512u) return;
    if (n <= 2u * kWave) sort_tile_in_wave<2>(pairs + b, n);
    else if (n <= 4u * kWave) sort_tile_in_wave<4>(pairs + b, n);
    else sort_tile_in_wave<8>(pairs + b, n);
}

// Lists longer than the LDS classes (> 16384 entries: degenerate scenes where one tile holds a large part of the
// Gaussians).  Same all-ascending network, one 1024-thread block per tile, organised around 16384-entry CHUNKS that
// fit the LDS: every chunk is first sorted in LDS; then, per merge size k = 2, 4, ... chunks, only the steps whose
// partner distance is >= one chunk go through global memory (log2(k / chunk) passes), and the rest of the merge is
// finished chunk by chunk in LDS again.  65,536 entries: 3 global passes instead of the 136 of a network that lives
// in global memory throughout.
constexpr uint32_t kBigChunk = 16384;

__device__ __forceinline__ void lds_steps(uint64_t* s, uint32_t cn, uint32_t kfirst, uint32_t klast) {
    // merges of size kfirst .. klast (powers of two, <= kBigChunk) on the cn (<= kBigChunk) keys in LDS
    const uint32_t half = kBigChunk >> 1;
    for (uint32_t k = kfirst; k <= klast; k <<= 1) {
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            const uint32_t mask = (j == (k >> 1)) ? (k - 1) : j;
            for (uint32_t i = threadIdx.x; i < half; i += 1024) {
                const uint32_t a = ((i & ~(j - 1)) << 1) | (i & (j - 1));
                const uint32_t c = a ^ mask;
                if (c < cn) {
                    const uint64_t x = s[a], y = s[c];
                    if (x > y) { s[a] = y; s[c] = x; }
                }
            }
            __syncthreads();
        }
    }
}
// the tail of a merge whose first steps were done in global memory: partner distances chunk/2 .. 1 (no mirror step)
__device__ __forceinline__ void lds_tail(uint64_t* s, uint32_t cn) {
    const uint32_t half = kBigChunk >> 1;
    for (uint32_t j = kBigChunk >> 1; j > 0; j >>= 1) {
        for (uint32_t i = threadIdx.x; i < half; i += 1024) {
            const uint32_t a = ((i & ~(j - 1)) << 1) | (i & (j - 1));
            const uint32_t c = a ^ j;
            if (c < cn) {
                const uint64_t x = s[a], y = s[c];
                if (x > y) { s[a] = y; s[c] = x; }
            }
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(1024) void spf_sort_tiles_big_kernel(TileLists tl,
                                                                  const uint32_t* __restrict__ counters,
                                                                  uint64_t* pairs, uint64_t capacity, uint32_t lo) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    uint64_t* s = reinterpret_cast<uint64_t*>(smem_raw);
    if (counters[0] > capacity) return;
    uint32_t b, n;
    tile_range(tl, blockIdx.x, b, n);
    if (n <= lo) return;
    uint64_t* p = pairs + b;
    uint32_t m = 1;
    while (m < n) m <<= 1;
    // 1) every chunk sorted on its own
    for (uint32_t c0 = 0; c0 < n; c0 += kBigChunk) {
        const uint32_t cn = min(kBigChunk, n - c0);
        for (uint32_t i = threadIdx.x; i < cn; i += 1024) s[i] = p[c0 + i];
        __syncthreads();
        lds_steps(s, cn, 2, kBigChunk);
        for (uint32_t i = threadIdx.x; i < cn; i += 1024) p[c0 + i] = s[i];
        __syncthreads();
    }
    // 2) merges across chunks
    for (uint32_t k = 2 * kBigChunk; k <= m; k <<= 1) {
        for (uint32_t j = k >> 1; j >= kBigChunk; j >>= 1) {                      // global passes
            const uint32_t mask = (j == (k >> 1)) ? (k - 1) : j;
            for (uint32_t i = threadIdx.x; i < (m >> 1); i += 1024) {
                const uint32_t a = ((i & ~(j - 1)) << 1) | (i & (j - 1));
                const uint32_t c = a ^ mask;
                if (c < n) {
                    const uint64_t x = p[a], y = p[c];
                    if (x > y) { p[a] = y; p[c] = x; }
                }
            }
            __threadfence_block();
            __syncthreads();
        }
        for (uint32_t c0 = 0; c0 < n; c0 += kBigChunk) {                          // the rest of the merge, in LDS
            const uint32_t cn = min(kBigChunk, n - c0);
            for (uint32_t i = threadIdx.x; i < cn; i += 1024) s[i] = p[c0 + i];
            __syncthreads();
            lds_tail(s, cn);
            for (uint32_t i = threadIdx.x; i < cn; i += 1024) p[c0 + i] = s[i];
            __syncthreads();
        }
    }
}

// ---- launchers ----------------------------------------------------------------------------------
// `zeroed`: the caller cleared tile_fill and counters[0..3] (together with the tile counters): the scan can then run
// with one block per render; otherwise (or when there are too many renders for every block to add up its
// predecessors) the single-block scan, which initialises both itself.
hipError_t launch_tile_scan(const SpfState& st, int R, int T, int nblk, uint32_t dense_thr, bool zeroed,
                            hipStream_t stream) {
    if (zeroed && R > 1 && R <= 256 && (long)R * T <= (1L << 18))
        spf_tile_scan_render_kernel<<<R, kScanThreads, 0, stream>>>(st.tile_count, st.tile_start, st.tile_flags,
                                                                    dense_thr, st.counters, T, st.blk_total,
                                                                    st.blk_base, nblk);
    else
        spf_tile_scan_kernel<<<1, kScanThreads, 0, stream>>>(st.tile_count, st.tile_start, st.tile_fill, st.tile_flags,
                                                             dense_thr, st.counters, R * T, st.blk_total, st.blk_base,
                                                             R * nblk);
    return hipGetLastError();
}

// (`d` may describe a chunk of the call's renders)
hipError_t launch_bin_pairs(const SpfDims& d, const SpfState& st, uint64_t capacity, int T, int tiles_x,
                            uint32_t max_tile_hint, hipStream_t stream) {
    dim3 grid((d.G + kBlock - 1) / kBlock, d.S * d.V);
    const int lds = T <= max_lds_tiles() ? 1 : 0;
    // Two views of a scene per block when a render has many blocks (measured, views per block 1 / 2 / 4: 320,000 Gaussians
    // per render 47.5 / 41.6 / 43.8 us, 500,000 at 512 x 512 54.7 / 53.5 / 54.4, 65,536 23.7 / 25.1 / 29.1 -- the longer chain
    // of a block pays only where the launch is many rounds of blocks); SPF_BIN_VIEWS=1 / 2 pins it
    const char* const want_env = getenv("SPF_BIN_VIEWS");
    const int want_vb = want_env ? atoi(want_env) : 0;
    const int nblk = (d.G + kBlock - 1) / kBlock;
    const int vb = want_vb ? (want_vb >= 2 ? 2 : 1) : (nblk >= 1024 ? 2 : 1);
    if (lds && vb == 2 && d.V >= 2 && T <= 4 * kBlock) {
        const int nvb = (d.V + 1) / 2;
        dim3 vgrid(nblk, d.S * nvb);
        spf_bin_pairs_views_kernel<2><<<vgrid, kBlock, (size_t)4 * T * sizeof(uint32_t), stream>>>(
            st.zkey, st.rect, st.tile_start, st.tile_fill, st.counters, st.pairs, capacity, st.blk_base, st.pair_off,
            d.G, T, tiles_x, d.V, nvb, max_tile_hint);
        return hipGetLastError();
    }
    spf_bin_pairs_kernel<<<grid, kBlock, lds ? 2 * sizeof(uint32_t) * T : 0, stream>>>(
        st.zkey, st.rect, st.tile_start, st.tile_fill, st.counters, st.pairs, capacity, st.blk_base, st.pair_off,
        d.G, T, tiles_x, lds, max_tile_hint);
    return hipGetLastError();
}

// Size classes: (1, 512]: one wave per tile, list in registers; (512, 1024], (1024, 2048], (2048, 4096]: one 256-thread block
// per tile, list in registers, three LDS exchanges; (4096, 8192], (8192, 16384]: one 1024-thread
// block per tile in LDS; > 16384: chunked LDS sort with a few global merge passes.  `max_tile_hint` (0 = unknown) lets the host skip empty classes.
// `RT`: tiles of this launch; `RT_call`: tiles of the whole call it is a chunk of (picks the kernel family).
// `tl`: where the lists are (packed, or direct bins)
// `order` (direct bins, or null): eight more blocks in front of the first kernel write the composite lists kernels' launch
// order there (tile_order_block).
hipError_t launch_tile_sort(const SpfState& st, const TileLists& tl, int RT, int RT_call, uint64_t capacity,
                            uint32_t max_tile_hint, const uint2* order_c, int T,
                            hipStream_t stream) {   // T: tiles per render (0: contiguous tile ranges per XCD, see xcd_map)
    uint2* order = const_cast<uint2*>(order_c);
    const int ob = order ? 8 * order_windows(RT) : 0;
    const uint32_t mx = max_tile_hint ? max_tile_hint : 0xffffffffu;
    const uint32_t thr = dense_threshold(), thr_f = dense_threshold_fwd();
    const int wgrid = (RT + kBlock / kWave - 1) / (kBlock / kWave);
    // Lists of 513 .. 2048 entries: one wave per tile (16 / 32 keys per lane) when there are enough tiles to fill the
    // chip with single waves, one 256-thread block per tile when there are not (measured: 2,048 tiles of ~1,000 entries
    // 68 -> 60 us, 4,096 tiles of ~540 entries 69 -> 52 us with blocks; 8,192 tiles of ~540 entries 55 us with waves,
    // 65 us with blocks)
    const char* force = getenv("SPF_SORT_BLOCKS");       // (tests: "0" / "1" pin one of the two families)
    const bool blocks = force ? force[0] == '1' : RT_call < 6144;
    const bool mixed = blocks && mx > 512 && !getenv("SPF_SORT_SEPARATE");    // 2 .. 2048 in one launch (see the kernel)
    if (order && !(mx > 1))      // nothing to sort: the order on its own
        spf_tile_order_kernel<<<ob, kBlock, 0, stream>>>(tl.count, st.tile_flags, order, RT, T, tl.cap, thr, thr_f);
    // (SPF_SORT_BIG_MIXED=0: the 2,049 .. 4,096 class as a launch of its own, as before)
    const char* const bm = getenv("SPF_SORT_BIG_MIXED");
    const bool big_mixed = mixed && mx > 2048 && !getenv("SPF_SORT_LDS_2K") && !(bm && bm[0] == '0');
    if (big_mixed)
        spf_sort_tiles_mixed_kernel<true><<<ob + 2 * RT + wgrid, kBlock, 0, stream>>>(tl, st.tile_flags, st.counters, st.pairs,
                                                                                      capacity, RT, thr, order, T, thr_f);
    else if (mixed)
        spf_sort_tiles_mixed_kernel<false><<<ob + RT + wgrid, kBlock, 0, stream>>>(tl, st.tile_flags, st.counters, st.pairs,
                                                                                   capacity, RT, thr, order, T, thr_f);
    // many tiles, lists of 2 .. 1024: a pair of waves per tile (C2 29.3 -> 27.7 us, C5 57.3 -> 50.1; SPF_SORT_SINGLE=1: one wave)
    // (8 px grid: lists are a quarter as long -- up to 512 entries one wave per tile, four tiles per block, is the better fit)
    const bool pairsk = !blocks && mx > 1 && mx <= 1024 && !getenv("SPF_SORT_SINGLE");
    if (pairsk)
        spf_sort_tiles_pair_kernel<<<ob + RT, 2 * kWave, 0, stream>>>(tl, st.tile_flags, st.counters, st.pairs, capacity, RT,
                                                                      thr, order, T, thr_f);
    if (mx > 1 && (mx <= 512 || blocks) && !mixed && !pairsk)  // lists of 2 .. 512: four tiles per block, one wave each, 2 / 4 / 8 keys per lane
        spf_sort_tiles_wave_kernel<8, true><<<ob + wgrid, kBlock, 0, stream>>>(tl, st.tile_flags, st.counters, st.pairs,
                                                                               capacity, 1, RT, thr, order, T, thr_f);
    if (mx > 512 && !blocks && !pairsk)     // 2 .. 1024 with one wave per tile (up to 16 keys per lane)
        spf_sort_tiles_wave_kernel<16, true><<<ob + wgrid, kBlock, 0, stream>>>(tl, st.tile_flags, st.counters, st.pairs,
                                                                                capacity, 1, RT, thr, order, T, thr_f);
    if (mx > 1024 && !blocks)    // 1025 .. 2048 with one wave per tile (32 keys per lane)
        spf_sort_tiles_wave_kernel<32, false><<<wgrid, kBlock, 0, stream>>>(tl, st.tile_flags, st.counters, st.pairs,
                                                                            capacity, 1024, RT, thr, nullptr, 0, thr_f);
    if (mx > 512 && blocks && !mixed)      // 513 .. 1024: one block per tile, 4 keys per thread
        spf_sort_tiles_block_kernel<4><<<RT, kBlock, 0, stream>>>(tl, st.counters, st.pairs, capacity, 512);
    if (mx > 1024 && blocks && !mixed)     // 1025 .. 2048: 8 keys per thread
        spf_sort_tiles_block_kernel<8><<<RT, kBlock, 0, stream>>>(tl, st.counters, st.pairs, capacity, 1024);
    // 2049 .. 4096: one 256-thread block per tile, SIXTEEN keys per thread in registers (each wave sorts its 1,024 keys
    // with lane exchanges, the last two merges start with three exchanges through LDS) -- round 5: the reference's
    // 10-view shape is 768 tiles of ~2,800 entries, ALL in this class, and the all-LDS network below took 74 us for them
    // (78 barrier-separated passes of 1,024 threads); 4097 .. 8192 stay with it
    if (mx > 2048 && !getenv("SPF_SORT_LDS_2K") && !big_mixed)
        spf_sort_tiles_block_kernel<16><<<RT, kBlock, 0, stream>>>(tl, st.counters, st.pairs, capacity, 2048);
    if (mx > 2048 && getenv("SPF_SORT_LDS_2K"))
        spf_sort_tiles_lds_kernel<1024><<<RT, 1024, 8192 * 8, stream>>>(tl, st.counters, st.pairs, capacity, 2048, 8192);
    else if (mx > 4096)
        spf_sort_tiles_lds_kernel<1024><<<RT, 1024, 8192 * 8, stream>>>(tl, st.counters, st.pairs, capacity, 4096, 8192);
    if (mx > 8192) {
        // the opt-in to 128 KB of dynamic LDS is a per-DEVICE function attribute: remember it per device
        static std::atomic<bool> attr_set[64];          // (zero-initialised; the attribute is idempotent, so two host
        int dev = 0;                                      //  threads racing here at worst both set it)
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64 || !attr_set[dev].load(std::memory_order_acquire)) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&spf_sort_tiles_lds_kernel<1024>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, 16384 * 8);
            if (e != hipSuccess) return e;
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(&spf_sort_tiles_big_kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 16384 * 8);
            if (e != hipSuccess) return e;
            if (dev >= 0 && dev < 64) attr_set[dev].store(true, std::memory_order_release);
        }
        spf_sort_tiles_lds_kernel<1024><<<RT, 1024, 16384 * 8, stream>>>(tl, st.counters, st.pairs, capacity, 8192, 16384);
    }
    if (mx > 16384)
        spf_sort_tiles_big_kernel<<<RT, 1024, 16384 * 8, stream>>>(tl, st.counters, st.pairs, capacity, 16384);
    return hipGetLastError();
}

}  // namespace spf
