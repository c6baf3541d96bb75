// 16x16-tile alpha compositing, forward and backward, for gfx950 (wave64).
//
// ONE kernel per direction; a block = one tile = 256 threads = 4 waves composites its tile in one of two forms, chosen per
// tile and per direction from the tile's mean footprint (SPF_DENSE_AREA_FWD / SPF_DENSE_AREA, include/spfsplat_hip.h):
//
// "lists" (sparse tiles: footprints of a few pixels) -- see spf_render_fwd_lists_kernel / spf_render_bwd_lists_kernel:
//   thread i scatters staged entry i's footprint into per-pixel candidate bit words, thread p then walks ITS pixel's
//   candidates; work is proportional to the (pixel, Gaussian) pairs that really interact.
//
// "rows" (dense tiles) -- fwd_rows_tile / bwd_rows_tile, device functions the lists kernels branch into (block-uniform)
// on their own LDS: wave w owns the 8x8 sub-tile (w&1, w>>1) and every DPP row (16 lanes) of a wave owns one 4x4 pixel block.
//   * the tile's depth-sorted list is staged 256 entries at a time into LDS (one coalesced 8-byte key load +
//     one 48-byte record gather per thread),
//   * while staging, each thread tests its entry against the tile's sixteen 4x4 blocks with a conservative
//     bound (record field 7, see project.hip); __ballot turns the results into one 32-bit mask per block per 32
//     staged entries,
//   * every row then walks only the set bits of ITS block's masks, so the four rows of a wave composite four
//     different Gaussians at the same time; entries are read back from LDS (four addresses per wave),
//   * per-lane early termination (T < 1e-4) is folded into row / wave ballots and a block-wide
//     __syncthreads_and, so a saturated tile stops streaming its list.
// A culled (block, entry) -- or, in the lists form, (pixel, entry) -- is one whose alpha is < 1/255 there, i.e. one the
// per-pixel loop would have skipped anyway, so results are identical to the un-culled loop; and every form evaluates
// alpha with the same expression tree (lists_power2_scalar), so the forward's output does not depend on the form.
//
// Semantics: SURVEY.md Appendix B #10/#11 (restated in oracle/splat_ref.py::composite).
#include <stdlib.h>

#include "spf_common.h"

namespace spf {

constexpr int kStage = 256;  // list entries staged per round (one per thread) of the forward's rows form
constexpr int kFwdLongRoundsMaxTiles = 768;    // calls of at most this many tiles: 512-entry rounds in the forward lists kernel

struct BlockCtx {
    int r, tx, ty, wave, lane, row, l16, beta, px, py;
    bool inside;
};

// Work decomposition shared by forward and backward: block = tile, wave = 8x8 sub-tile, DPP row (16 lanes)
// = 4x4 pixel block `beta` (row-major over the tile's 4x4 grid of blocks).
__device__ __forceinline__ void block_ctx_at(BlockCtx& c, int r, int tx, int ty, int H, int W) {
    c.r = r;
    c.ty = ty;
    c.tx = tx;
    c.wave = threadIdx.x >> 6;
    c.lane = threadIdx.x & 63;
    c.row = c.lane >> 4;
    c.l16 = c.lane & 15;
    const int bx = (c.wave & 1) * 2 + (c.row & 1), by = (c.wave >> 1) * 2 + (c.row >> 1);
    c.beta = by * 4 + bx;
    c.px = c.tx * kTile + bx * 4 + (c.l16 & 3);
    c.py = c.ty * kTile + by * 4 + (c.l16 >> 2);
    c.inside = c.px < W && c.py < H;
}

// Row totals of TEN values at once, 26 instructions (one shift ladder per value: 5 each).  A value-pairing butterfly over
// the two HIGH lane bits of a row, then plain sums over the two low ones:
//   step 1 (row_ror:8: lane l meets l ^ 8)        lanes 0-7 keep values 0-7, lanes 8-15 keep values 8, 9;
//   step 2 (row_half_mirror: l meets 7 - l)       lanes with bit 2 clear keep the lower four of their values, the others
//                                                 the upper four;
//   steps 3, 4 (quad_perm xor 2, xor 1)           plain sums: the four lanes of a quad end with the same four totals.
// "Keep" costs nothing: a DPP add writes only the banks its bank_mask names, so each half of a destination register is
// written by its own instruction from its own source (no v_cndmask -- the builtin form of the same butterfly, which needs
// two selects per step, measured 2 - 3 % SLOWER than the ladders; see DESIGN_EXPERIMENTS.md).  On return quad q = l16 >> 2
// holds z0..z3 = row totals of values 4 q .. 4 q + 3 (q = 2: values 8, 9 in z0, z1; q = 3: nothing).  asm: hipcc lowers
// a bank-masked update_dpp to three instructions; the leading s_nop covers the VALU-write -> DPP-read hazard (2 wait
// states) the compiler does not insert around inline asm, inside the block every DPP source is >= 3 instructions old.
__device__ __forceinline__ void row_sums10(float x0, float x1, float x2, float x3, float x4, float x5, float x6, float x7,
                                           float x8, float x9, float& z0, float& z1, float& z2, float& z3) {
    float y0, y1, y2, y3, y4, y5, y6, y7;
#define SPF_ROR8(d, a, bank) "v_add_f32_dpp %[" #d "], %[" #a "], %[" #a "] row_ror:8 row_mask:0xf bank_mask:" #bank "\n\t"
#define SPF_HMIR(d, a, bank) "v_add_f32_dpp %[" #d "], %[" #a "], %[" #a "] row_half_mirror row_mask:0xf bank_mask:" #bank "\n\t"
#define SPF_QUAD(d, perm) "v_add_f32_dpp %[" #d "], %[" #d "], %[" #d "] quad_perm:" perm " row_mask:0xf bank_mask:0xf\n\t"
    asm volatile("s_nop 1\n\t"
                 SPF_ROR8(y0, x0, 0x3) SPF_ROR8(y0, x8, 0xc) SPF_ROR8(y1, x1, 0x3) SPF_ROR8(y1, x9, 0xc)
                 SPF_ROR8(y2, x2, 0xf) SPF_ROR8(y3, x3, 0xf) SPF_ROR8(y4, x4, 0xf) SPF_ROR8(y5, x5, 0xf)
                 SPF_ROR8(y6, x6, 0xf) SPF_ROR8(y7, x7, 0xf)
                 SPF_HMIR(z0, y0, 0x5) SPF_HMIR(z0, y4, 0xa) SPF_HMIR(z1, y1, 0x5) SPF_HMIR(z1, y5, 0xa)
                 SPF_HMIR(z2, y2, 0x5) SPF_HMIR(z2, y6, 0xa) SPF_HMIR(z3, y3, 0x5) SPF_HMIR(z3, y7, 0xa)
                 SPF_QUAD(z0, "[2,3,0,1]") SPF_QUAD(z1, "[2,3,0,1]") SPF_QUAD(z2, "[2,3,0,1]") SPF_QUAD(z3, "[2,3,0,1]")
                 SPF_QUAD(z0, "[1,0,3,2]") SPF_QUAD(z1, "[1,0,3,2]") SPF_QUAD(z2, "[1,0,3,2]") SPF_QUAD(z3, "[1,0,3,2]")
                 : [y0] "=&v"(y0), [y1] "=&v"(y1), [y2] "=&v"(y2), [y3] "=&v"(y3), [y4] "=&v"(y4), [y5] "=&v"(y5),
                   [y6] "=&v"(y6), [y7] "=&v"(y7), [z0] "=&v"(z0), [z1] "=&v"(z1), [z2] "=&v"(z2), [z3] "=&v"(z3)
                 : [x0] "v"(x0), [x1] "v"(x1), [x2] "v"(x2), [x3] "v"(x3), [x4] "v"(x4), [x5] "v"(x5), [x6] "v"(x6),
                   [x7] "v"(x7), [x8] "v"(x8), [x9] "v"(x9));
#undef SPF_ROR8
#undef SPF_HMIR
#undef SPF_QUAD
}

template <int CTRL>
__device__ __forceinline__ uint32_t dpp_u(uint32_t x) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, CTRL, 0xf, 0xf, true);
}

// max over the 16 lanes of a row, result in every lane (xor-1, xor-2, half-mirror, mirror butterflies)
__device__ __forceinline__ uint32_t row_max_all(uint32_t x) {
    x = max(x, dpp_u<0xB1>(x));    // quad_perm [1,0,3,2]
    x = max(x, dpp_u<0x4E>(x));    // quad_perm [2,3,0,1]
    x = max(x, dpp_u<0x141>(x));   // row_half_mirror
    x = max(x, dpp_u<0x140>(x));   // row_mirror
    return x;
}

// 16-bit mask: which 4x4 blocks of tile (tx,ty) can be touched by a Gaussian at (gx,gy), squared cull radius r2
__device__ __forceinline__ uint32_t block_bits(float gx, float gy, float r2, int tx, int ty) {
    float ddx[4], ddy[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float x0 = (float)(tx * kTile + 4 * k), y0 = (float)(ty * kTile + 4 * k);
        const float dx = fmaxf(fmaxf(x0 - gx, gx - (x0 + 3.f)), 0.f);
        const float dy = fmaxf(fmaxf(y0 - gy, gy - (y0 + 3.f)), 0.f);
        ddx[k] = dx * dx;
        ddy[k] = dy * dy;
    }
    uint32_t bits = 0;
#pragma unroll
    for (int by = 0; by < 4; ++by)
#pragma unroll
        for (int bx = 0; bx < 4; ++bx)
            if (!(ddx[bx] + ddy[by] > r2)) bits |= 1u << (by * 4 + bx);
    return bits;
}

// A PLANNED call (see spf_raster_forward_render) whose plan did not hold -- counters[2] != 0: pair buffer too small, a
// list too long to have been sorted, or tiles without a kernel -- must not leave uninitialised or half-right pixels
// behind: every render kernel that runs then fills its tiles' outputs with NaN instead of rendering (both kernels
// cover every tile, so whichever was launched poisons the whole batch).  The flag also subsumes the memory-safety
// guard `D > capacity`.
__device__ __forceinline__ void poison_tile(int RT, int T, int tiles_x, int H, int W, float* __restrict__ image,
                                            float* __restrict__ depth_out, float* __restrict__ alpha_out) {
    const int vid = xcd_remap(blockIdx.x, gridDim.x);
    if (vid >= RT) return;
    const int r = vid / T, tile = vid - r * T;
    const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
    const int px = tx * kTile + (threadIdx.x & 15), py = ty * kTile + (threadIdx.x >> 4);
    if (px >= W || py >= H) return;
    const size_t P = (size_t)H * W, pix = (size_t)py * W + px;
    const float nan = __builtin_nanf("");
    float* __restrict__ img = image + (size_t)r * 3 * P;
    img[pix] = nan; img[P + pix] = nan; img[2 * P + pix] = nan;
    depth_out[(size_t)r * P + pix] = nan;
    alpha_out[(size_t)r * P + pix] = nan;
}

// Exponent of ALL compositing forms (lists and rows, forward and backward).  They stage the conic pre-multiplied: A' = -0.5*log2(e)*A, B' = -log2(e)*B,
// C' = -0.5*log2(e)*C, so that G = exp2(A' dx^2 + B' dx dy + C' dy^2) is one fma chain and one v_exp_f32.  The chain is
// spelled out so that the forward and the backward replay evaluate it identically (same hit decisions) -- whichever form
// composites a tile in either direction: a tile may go forward through lists and backward through rows (the dense
// thresholds differ), and the forward's images do not depend on the form at all (bit-identical, tested).
typedef float v2f __attribute__((ext_vector_type(2)));

constexpr float kHalfLog2e = -0.72134752044448170368f;   // -0.5 * log2(e)
constexpr float kLog2e = -1.44269504088896340736f;       // -log2(e)
// Packed form: the lists kernels stage (x, y | A', C') in one 16-byte LDS record, so the pixel offset is one
// v_pk_add_f32 and (A' dx, C' dy) one v_pk_mul_f32; then exponent = dx * (A' dx + B' dy) + (C' dy) * dy.
__device__ __forceinline__ float lists_power2(const float4& p0, float Bs, v2f fxy, v2f& dxy) {
    dxy = v2f{p0.x, p0.y} - fxy;
    const v2f u = v2f{p0.z, p0.w} * dxy;                      // (A' dx, C' dy)
    return fmaf(dxy.x, fmaf(Bs, dxy.y, u.x), u.y * dxy.y);
}
// The same expression tree with scalar instructions -- bit-identical results (the backward replay must take the
// forward's hit decisions), for the backward kernel, where the even-aligned register pairs of the packed form cost more
// moves than they save (measured: +8 us).
__device__ __forceinline__ float lists_power2_scalar(const float4& p0, float Bs, float fx, float fy) {
    const float dx = p0.x - fx, dy = p0.y - fy;
    const float ux = p0.z * dx, uy = p0.w * dy;
    return fmaf(dx, fmaf(Bs, dy, ux), uy * dy);
}

// ------------------------------------------------------------------------------------------------
// Forward of a DENSE tile ("rows"): front-to-back compositing, one 4x4 pixel block per DPP row (see the header
// comment).  Not a kernel of its own: the forward kernel below runs it for the tiles the launch order marks dense, on
// its own LDS (256 staged entries: 10.5 KB of the 18.7 KB the lists form holds) -- one launch composites every tile.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void fwd_rows_tile(float4* s_p0, float2* s_p1, float4* s_p2, uint32_t (*s_mask)[16],
                                              const float* __restrict__ rec_r, const uint64_t* __restrict__ pairs,
                                              uint32_t beg, uint32_t n, int r, int tx, int ty,
                                              const float* __restrict__ bg_all, float* __restrict__ image,
                                              float* __restrict__ depth_out, float* __restrict__ alpha_out,
                                              float* __restrict__ final_T, uint32_t* __restrict__ n_contrib, int H, int W) {
    BlockCtx c;
    block_ctx_at(c, r, tx, ty, H, W);
    float fx = (float)c.px, fy = (float)c.py;
    asm volatile("" : "+v"(fx), "+v"(fy));   // keep the converted coordinates live (no per-iteration v_cvt)

    float Tr = 1.0f, C0 = 0.f, C1 = 0.f, C2 = 0.f, Dp = 0.f;
    uint32_t last = 0;
    bool done = !c.inside;
    bool row_done = false;
    bool wave_done = __ballot(!done) == 0;

    for (uint32_t base = 0; base < n; base += kStage) {
        // ---- stage: one list entry per thread, plus the 4x4-block masks of the staged entries ----
        uint32_t bits = 0;
        const uint32_t idx = base + threadIdx.x;
        if (idx < n) {
            const uint32_t gid = (uint32_t)pairs[beg + idx];
            const float4* __restrict__ rp = reinterpret_cast<const float4*>(rec_r + (size_t)gid * kRec);
            const float4 a = rp[0], b = rp[1], cc = rp[2];
            s_p0[threadIdx.x] = make_float4(a.x, a.y, kHalfLog2e * a.z, kHalfLog2e * b.x);   // x, y | A', C'
            s_p1[threadIdx.x] = make_float2(kLog2e * a.w, b.y);                               // B', opacity
            s_p2[threadIdx.x] = make_float4(cc.x, cc.y, cc.z, b.z);
            bits = block_bits(a.x, a.y, b.w, c.tx, c.ty);
        }
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const uint64_t m = __ballot((bits >> k) & 1u);
            if (c.lane == 0) {
                s_mask[2 * c.wave][k] = (uint32_t)m;
                s_mask[2 * c.wave + 1][k] = (uint32_t)(m >> 32);
            }
        }
        __syncthreads();
        // ---- consume: every row walks the set bits of its own block's masks ----
        if (!wave_done) {
            const int nq = (int)min((uint32_t)(kStage / 32), (n - base + 31u) / 32u);
            for (int q = 0; q < nq; ++q) {
                uint32_t m = row_done ? 0u : s_mask[q][c.beta];
                while (__ballot(m != 0)) {
                    const bool act = m != 0;
                    const int bit = act ? __builtin_ctz(m) : 0;
                    m &= m - 1u;
                    const int j = q * 32 + bit;
                    const float4 p0 = s_p0[j];
                    const float2 p1 = s_p1[j];
                    const float power = lists_power2_scalar(p0, p1.x, fx, fy);            // (log2 units)
                    const float alpha = fminf(kAlphaMax, p1.y * __builtin_amdgcn_exp2f(power));
                    const float4 p2 = s_p2[j];
                    // branch-free body: predicates fold into selects, every lane runs the same ~35 instructions
                    const bool hit = act && !done && power <= 0.f && alpha >= kAlphaMin;
                    const float test_T = Tr * (1.f - alpha);
                    const bool stop = hit && test_T < kTMin;
                    const bool take = hit && !stop;
                    done = done || stop;
                    const float w = take ? alpha * Tr : 0.f;
                    C0 = fmaf(p2.x, w, C0); C1 = fmaf(p2.y, w, C1); C2 = fmaf(p2.z, w, C2); Dp = fmaf(p2.w, w, Dp);
                    Tr = take ? test_T : Tr;
                    last = take ? base + (uint32_t)j + 1u : last;
                }
                // per-row / per-wave early termination from one ballot per 32 entries
                const uint64_t alive = __ballot(!done);
                row_done = ((alive >> (c.lane & 48)) & 0xffffull) == 0;
                if (alive == 0) { wave_done = true; break; }
            }
        }
        if (__syncthreads_and(wave_done)) break;
    }
    if (c.inside) {
        const float* __restrict__ bg = bg_all + 3 * c.r;
        const size_t P = (size_t)H * W, pix = (size_t)c.py * W + c.px;
        float* __restrict__ img = image + (size_t)c.r * 3 * P;
        img[pix] = C0 + Tr * bg[0];
        img[P + pix] = C1 + Tr * bg[1];
        img[2 * P + pix] = C2 + Tr * bg[2];
        depth_out[(size_t)c.r * P + pix] = Dp;
        alpha_out[(size_t)c.r * P + pix] = 1.0f - Tr;
        final_T[(size_t)c.r * P + pix] = Tr;
        n_contrib[(size_t)c.r * P + pix] = last;
    }
}

// ------------------------------------------------------------------------------------------------
// Forward for sparse tiles ("lists"): Gaussian-parallel footprint scatter + pixel-parallel private lists.
//
//   phase A  thread i owns staged entry i and ORs bit i into the candidate words of exactly the pixels inside its
//            conservative cull disc (LDS atomics; ~9 pixels for a pixel-aligned Gaussian) -- every lane works on a
//            different Gaussian, nothing is wasted on empty (pixel, Gaussian) combinations;
//   phase B  thread p owns pixel p (x = p & 15, y = p >> 4) and walks the set bits of ITS eight 32-bit words in
//            list order, gathering each candidate from LDS and compositing it.
// A pixel outside the disc has alpha < 1/255 for that Gaussian, so the result equals the plain per-pixel loop.
// ------------------------------------------------------------------------------------------------
// Ballot of a lane predicate as the compiler keeps it (an SGPR pair).  HIP's __ballot(int) widens the predicate to
// 0 / 1 in a VGPR and compares it again (v_cndmask + v_cmp per call); in loops whose body is ~30 instructions that
// round trip is 5 % of the kernel.

// Phase A over a box that is already known (the backward needs the clipped box for its slot pool anyway): no second
// disc_box (an IEEE square root), pixel offsets stepped in float.  dx, dy differ from (float)x - gx by rounding of
// the steps only (relative 1e-7) while the disc carries 0.2 % of slack over every pixel that can reach alpha = 1/255
// (project.hip), so the candidate set stays a superset of the contributors.
#ifndef SPF_SCATTER4
#define SPF_SCATTER4 1
#endif
__device__ __forceinline__ void scatter_box(uint32_t (*s_pm)[kBlock], int i, float gx, float gy, float r2, int X0, int Y0,
                                            int xl, int yl, int bw, int bh) {
    uint32_t* __restrict__ wp = s_pm[i >> 5] + yl * kTile + xl;
    const uint32_t bit = 1u << (i & 31);
    const float dx0 = (float)(X0 + xl) - gx;
    float dy = (float)(Y0 + yl) - gy;
#if SPF_SCATTER4
    // Rows outside, FOUR columns per trip inside: a pixel-aligned footprint is 2 - 4 pixels wide, so the column loop of
    // the plain form ran once or twice per row and paid two branches and half a dozen scalar instructions per pixel for
    // it (divergent loops are waterfalls of exec-mask bookkeeping); unrolled, a pixel is its test and a predicated LDS OR.
    for (int y = 0; y < bh; ++y) {
        const float dy2 = dy * dy;
        float dx = dx0;
        for (int x = 0; x < bw; x += 4) {
            const int left = bw - x;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float d = dx + (float)k;
                if (k < left && !(fmaf(d, d, dy2) > r2)) atomicOr(&wp[x + k], bit);
            }
            dx += 4.f;
        }
        dy += 1.f;
        wp += kTile;
    }
#else
    for (int y = 0; y < bh; ++y) {
        const float dy2 = dy * dy;
        float dx = dx0;
        for (int x = 0; x < bw; ++x) {
            if (!(fmaf(dx, dx, dy2) > r2)) atomicOr(&wp[x], bit);
            dx += 1.f;
        }
        dy += 1.f;
        wp += kTile;
    }
#endif
}

// The cull disc's pixel box clipped to tile (X0, Y0): first column / row inside the tile, width, height (0: empty).
struct TileBox { int xl, yl, bw, bh; };
__device__ __forceinline__ TileBox clipped_box(float gx, float gy, float r2, int X0, int Y0) {
    TileBox t = {0, 0, 0, 0};
    const DiscBox db = disc_box_fast(gx, gy, r2);
    if (db.any) {
        t.xl = (int)fmaxf(db.xlo - (float)X0, 0.f);
        t.yl = (int)fmaxf(db.ylo - (float)Y0, 0.f);
        t.bw = max(0, (int)fminf(db.xhi - (float)X0, (float)(kTile - 1)) - t.xl + 1);
        t.bh = max(0, (int)fminf(db.yhi - (float)Y0, (float)(kTile - 1)) - t.yl + 1);
    }
    return t;
}

// STAGE = list entries staged per round: 256 (one per thread; 18.7 KB of LDS, eight blocks per CU) for calls of many tiles,
// 512 (two per thread; 37 KB) for calls of few tiles with long lists -- there the launch is one or two rounds of blocks that
// run as long as their own chain of rounds, and half as many rounds are half as many barriers and trips to memory.
// Measured (round 5, same box, forward stage): the reference's 10-view shape (768 tiles of ~2,800 entries) 98.9 -> 89.0 us;
// but BASELINE config 3 (2,048 tiles) 59.8 -> 71.6, REF2V (4,096) 66.8 -> 78.9, C2 (8,192) 66.4 -> 87.6: four blocks per
// CU instead of eight costs more than the rounds save as soon as the chip is full -- 512 only up to 768 tiles.
template <int STAGE>
__global__ __launch_bounds__(kBlock, STAGE == 256 ? 8 : 4) void spf_render_fwd_lists_kernel(   // (8 blocks per CU: <= 64 VGPRs)
    const float* __restrict__ rec, const uint64_t* __restrict__ pairs, TileLists tl,
    const uint32_t* __restrict__ tile_flags, const uint32_t* __restrict__ counters, uint64_t capacity,
    const float* __restrict__ bg_all, float* __restrict__ image, float* __restrict__ depth_out,
    float* __restrict__ alpha_out, float* __restrict__ final_T, uint32_t* __restrict__ n_contrib, int G, int H, int W,
    int T, int tiles_x, int RT, uint32_t dense_thr_arg) {
    const uint32_t dense_thr = dense_thr_arg;
    __shared__ float4 s_p0[STAGE];   // x, y | A', C'   (conic pre-scaled, see lists_power2)
    __shared__ float2 s_p1[STAGE];   // B', opacity   (8 bytes: with a float4 here the block is 20,752 bytes of LDS -- 7 per CU instead of 8)
    __shared__ float4 s_p2z[STAGE + 1];   // r, g | b, depth; record 0 is all zeros (see `next` below), entry i is record i + 1
    __shared__ uint32_t s_pm[STAGE / 32][kBlock];   // [32-entry word][pixel]: candidate bits
    float4* const s_p2 = s_p2z + 1;

    (void)capacity;
    if (counters[2] != 0u) { poison_tile(RT, T, tiles_x, H, W, image, depth_out, alpha_out); return; }
    int vid;
    uint32_t beg, n;
    bool dense_tile;
    if (!lists_tile(tl, tile_flags, dense_thr, true, RT, vid, beg, n, dense_tile)) return;
    const int r = vid / T, tile = vid - r * T;
    const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
    const int tid = threadIdx.x;
    if (tid == 0) s_p2z[0] = make_float4(0.f, 0.f, 0.f, 0.f);
    const int X0 = tx * kTile, Y0 = ty * kTile;
    // (thread -> pixel stays row-major.  Measured: 16 consecutive lanes = one 4x4 pixel block, so that the LDS gathers of
    //  a 16-lane group hit the same staged records -- 74.1 / 73.6 against 74.5 / 73.2 us: bank conflicts are not the limit)
    const int lx = tid & 15, ly = tid >> 4;
    const int pid = ly * kTile + lx;            // this thread's pixel inside the tile (row-major): its candidate column
    const int px = X0 + lx, py = Y0 + ly;
    const bool inside = px < W && py < H;
    const float* __restrict__ rec_r = rec + (size_t)r * G * kRec;
    if (dense_tile) {                                                                // (block-uniform)
        fwd_rows_tile(s_p0, s_p1, s_p2, reinterpret_cast<uint32_t (*)[16]>(&s_pm[0][0]), rec_r, pairs, beg, n, r, tx, ty,
                      bg_all, image, depth_out, alpha_out, final_T, n_contrib, H, W);
        return;
    }
    float fx = (float)px, fy = (float)py;
    asm volatile("" : "+v"(fx), "+v"(fy));
    const v2f fxy = {fx, fy};

    float Tr = 1.0f;
    v2f c01 = {0.f, 0.f}, c2d = {0.f, 0.f};      // (r, g), (b, depth) accumulators
    uint32_t last16 = 0;                         // 16 * (list position + 1) of the last contributor
    // Lane predicates that live across candidates are kept as WAVE MASKS in scalar registers (`dm`: the lanes whose
    // pixel is finished) and combined with scalar instructions; lane_ballot / inverse_ballot move between the two
    // views for free.  As per-lane bools they cost three vector instructions per candidate.
    uint64_t dm = lane_ballot(!inside);
    bool wave_done = dm == ~0ull;

    // (candidate words: every thread keeps ITS pixel's column clear -- before the first round here, afterwards right
    // after it has consumed it -- so that staging and scatter of a round need no barrier between them)
#pragma unroll
    for (int w = 0; w < STAGE / 32; ++w) s_pm[w][pid] = 0u;
    __syncthreads();
    for (uint32_t base = 0; base < n; base += STAGE) {
        const int nw = (int)((min((uint32_t)STAGE, n - base) + 31u) >> 5);
        constexpr int EPT = STAGE / kBlock;                  // entries per thread: e = tid, tid + 256
        float4 ea[EPT], eb[EPT], ec[EPT];
#pragma unroll
        for (int q = 0; q < EPT; ++q) {                      // (all gathers of the round in flight together)
            const uint32_t idx = base + q * kBlock + tid;
            ea[q] = make_float4(0.f, 0.f, 0.f, 0.f); eb[q] = make_float4(0.f, 0.f, 0.f, -1.f); ec[q] = ea[q];
            if (idx < n) {
                const uint32_t gid = (uint32_t)pairs[beg + idx];
                const float4* __restrict__ rp = reinterpret_cast<const float4*>(rec_r + (size_t)gid * kRec);
                ea[q] = rp[0]; eb[q] = rp[1]; ec[q] = rp[2];
            }
        }
#pragma unroll
        for (int q = 0; q < EPT; ++q) {
            const int e = q * kBlock + tid;
            if (base + e < n) {
                s_p0[e] = make_float4(ea[q].x, ea[q].y, kHalfLog2e * ea[q].z, kHalfLog2e * eb[q].x);
                s_p1[e] = make_float2(kLog2e * ea[q].w, eb[q].y);
                s_p2[e] = make_float4(ec[q].x, ec[q].y, ec[q].z, eb[q].z);
            }
            const TileBox tb = clipped_box(ea[q].x, ea[q].y, eb[q].w, X0, Y0);      // (no entry: r2 = -1 -> empty box)
            scatter_box(s_pm, e, ea[q].x, ea[q].y, eb[q].w, X0, Y0, tb.xl, tb.yl, tb.bw, tb.bh);
        }
        __syncthreads();
        if (!wave_done) {
            const char* __restrict__ wcol = reinterpret_cast<const char*>(&s_pm[0][pid]);   // word w: wcol + 1024 w
            uint32_t m = *reinterpret_cast<const uint32_t*>(wcol);
            // which of the later words of this pixel's column hold a candidate at all (most are empty: a pixel has ~9
            // candidates among the round's 256 entries); words >= nw are clear
            uint32_t nz = 0u;
#pragma unroll
            for (int k = 1; k < STAGE / 32; ++k)
                nz |= min(*reinterpret_cast<const uint32_t*>(wcol + 1024 * k), 1u) << k;
            int wb = 0;                                  // 512 * word = byte offset of entry 32 w in the staged arrays
            // next candidate of this lane as the byte offset of its staged record.  has = false: none left -- the
            // offset is then 16 bytes below a word's first record (v_ffbl_b32 of 0 is -1).  What is read there is only
            // used by the colour accumulation, as colour x 0: it has to be FINITE.  It is: the record before a later
            // word's first is an entry of this round, and in front of entry 0 s_p2 holds a record of zeros.  A refill
            // jumps straight to the next non-empty word: no loop.
            auto next = [&](bool& has) -> int {
                if (m == 0u && nz != 0u) {       // (a divergent `if` is skipped as a whole when no lane takes it)
                    int w;
                    asm("v_ffbl_b32 %0, %1" : "=v"(w) : "v"(nz));
                    nz &= nz - 1u;
                    wb = w << 9;
                    m = *reinterpret_cast<const uint32_t*>(wcol + 2 * wb);
                }
                int bit;
                asm("v_ffbl_b32 %0, %1" : "=v"(bit) : "v"(m));
                has = bit >= 0;
                m &= m - 1u;
                return wb + (bit << 4);
            };
            const uint32_t base16 = (base + 1u) << 4;
            // hm: the lanes that hold a candidate.  A finished pixel keeps walking its bits (its lane is masked by dm,
            // so it neither composites nor keeps the loop alive).
            auto composite = [&](uint64_t hm, int j, const float4& p0, const float2& p1, const float4& p2) {
                v2f dxy;
                const float pw = lists_power2(p0, p1.x, fxy, dxy);
                const float alpha = fminf(kAlphaMax, p1.y * __builtin_amdgcn_exp2f(pw));
                const float test_T = Tr * (1.f - alpha);
                const uint64_t hitm = hm & ~dm & lane_ballot(pw <= 0.f) & lane_ballot(alpha >= kAlphaMin);
                const uint64_t stopm = hitm & lane_ballot(test_T < kTMin);
                dm |= stopm;
                const bool take = __builtin_amdgcn_inverse_ballot_w64(hitm & ~stopm);
                const float wgt = take ? alpha * Tr : 0.f;
                const v2f ww = {wgt, wgt};
                c01 = __builtin_elementwise_fma(v2f{p2.x, p2.y}, ww, c01);      // two v_pk_fma_f32 for (r, g | b, depth)
                c2d = __builtin_elementwise_fma(v2f{p2.z, p2.w}, ww, c2d);
                Tr = take ? test_T : Tr;
                last16 = take ? (uint32_t)j + base16 : last16;
            };
            auto rec0 = [&](int j) { return *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(s_p0) + j); };
            auto rec1 = [&](int j) { return *reinterpret_cast<const float2*>(reinterpret_cast<const char*>(s_p1) + (j >> 1)); };
            auto rec2 = [&](int j) { return *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(s_p2) + j); };
            // software pipeline, unrolled by two: the LDS gathers of one candidate are in flight while the other is
            // composited, and the two register sets swap roles instead of being copied
            bool ha, hb;
            int ja = next(ha), jb;
            uint64_t ma = lane_ballot(ha) & ~dm, mb;
            float4 a0 = rec0(ja), a2 = rec2(ja), b0, b2;
            float2 a1 = rec1(ja), b1;
            while (true) {
                if (!ma) break;
                jb = next(hb);
                mb = lane_ballot(hb);
                b0 = rec0(jb); b1 = rec1(jb); b2 = rec2(jb);
                composite(ma, ja, a0, a1, a2);
                mb &= ~dm;
                if (!mb) break;
                ja = next(ha);
                ma = lane_ballot(ha);
                a0 = rec0(ja); a1 = rec1(ja); a2 = rec2(ja);
                composite(mb, jb, b0, b1, b2);
                ma &= ~dm;
            }
            wave_done = dm == ~0ull;
        }
        for (int w = 0; w < nw; ++w) s_pm[w][pid] = 0u;
        if (__syncthreads_and(wave_done)) break;
    }
    const uint32_t last = last16 >> 4;
    if (inside) {
        const float* __restrict__ bg = bg_all + 3 * r;
        const size_t P = (size_t)H * W, pix = (size_t)py * W + px;
        float* __restrict__ img = image + (size_t)r * 3 * P;
        img[pix] = c01.x + Tr * bg[0];
        img[P + pix] = c01.y + Tr * bg[1];
        img[2 * P + pix] = c2d.x + Tr * bg[2];
        depth_out[(size_t)r * P + pix] = c2d.y;
        alpha_out[(size_t)r * P + pix] = 1.0f - Tr;
        final_T[(size_t)r * P + pix] = Tr;
        n_contrib[(size_t)r * P + pix] = last;
    }
}

// ------------------------------------------------------------------------------------------------
// Backward: back-to-front replay with the same decomposition.  The ten partial gradients of a
// (block, Gaussian) pair are reduced inside the row (5 DPP adds per value, no cross-row step); lane 15 adds the
// row totals into per-entry LDS accumulators shared by the tile's 16 blocks, and when a staging round is over
// each thread writes ONE 48-byte record for its entry's (Gaussian, tile) pair -- no global atomics.
// ------------------------------------------------------------------------------------------------
constexpr int kAcc = 10;
#ifdef SPF_ROWS_CENSUS
__device__ unsigned long long g_rows_census[8];
#define SPF_CENSUS(i, v) do { if ((threadIdx.x & 63) == 0) atomicAdd(&g_rows_census[i], (unsigned long long)(v)); } while (0)
#else
#define SPF_CENSUS(i, v) do {} while (0)
#endif

template <bool DEPTH_GRAD>
__device__ __forceinline__ void flush_pair(float* __restrict__ gpair, uint32_t slot, const float* acc) {
    store_grec<DEPTH_GRAD>(gpair, slot, acc[0], acc[1], acc[2], acc[3], acc[4], acc[5], acc[6], acc[7], acc[8], acc[9]);
}

// Backward of a DENSE tile ("rows"); like fwd_rows_tile not a kernel of its own: the backward kernel below runs it for the
// tiles the launch order marks dense, on its own LDS.  RS = entries staged per round (the kernel's kRoundL <= 256: one per
// thread, threads >= RS stage nothing); s_slot / s_acc live in the lists form's slot pool.
template <bool DEPTH_GRAD, int RS>
__device__ __forceinline__ void bwd_rows_tile(float4* s_p0, float2* s_p1, float4* s_p2, uint32_t* s_slot,
                                              float (*s_acc)[kAcc], uint32_t (*s_mask)[16], uint32_t* s_wmax,
                                              const float* __restrict__ rec_r, const uint64_t* __restrict__ pairs,
                                              uint32_t beg, uint32_t n, int r, int tx, int ty,
                                              const float* __restrict__ bg_all, const float* __restrict__ final_T,
                                              const uint32_t* __restrict__ n_contrib, const float* __restrict__ dL_dimage,
                                              const float* __restrict__ dL_ddepth, const float* __restrict__ dL_dalpha,
                                              const uint2* __restrict__ pinfo, float* __restrict__ gpair, int G, int H,
                                              int W) {
    static_assert(RS % 32 == 0 && RS <= kBlock, "one staged entry per thread, whole mask words");
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, row = lane >> 4, l16 = lane & 15;
    const int bx = (wave & 1) * 2 + (row & 1), by = (wave >> 1) * 2 + (row >> 1);
    const int beta = by * 4 + bx;
    const int px = tx * kTile + bx * 4 + (l16 & 3), py = ty * kTile + by * 4 + (l16 >> 2);
    const bool inside = px < W && py < H;

    const float fx = (float)px, fy = (float)py;
    const size_t P = (size_t)H * W, pix = (size_t)py * W + px;
    // Gaussian-major index of list entry idx's (Gaussian, tile) pair: ONE 8-byte gather (rect, first pair)
    auto pair_slot = [&](uint32_t gid) -> uint32_t {
        const uint2 pi = pinfo[(size_t)r * G + gid];
        const int x0 = pi.x & 0xff, y0 = (pi.x >> 8) & 0xff, x1 = (pi.x >> 16) & 0xff;
        return pi.y + (uint32_t)((ty - y0) * (x1 - x0) + (tx - x0));
    };

    float T_final = 1.f, gI0 = 0.f, gI1 = 0.f, gI2 = 0.f, gD = 0.f, gA = 0.f;
    uint32_t ncon = 0;
    if (inside) {
        T_final = final_T[(size_t)r * P + pix];
        ncon = n_contrib[(size_t)r * P + pix];
        if (dL_dimage) {
            const float* __restrict__ gi = dL_dimage + (size_t)r * 3 * P;
            gI0 = gi[pix]; gI1 = gi[P + pix]; gI2 = gi[2 * P + pix];
        }
        if (DEPTH_GRAD) gD = dL_ddepth[(size_t)r * P + pix];
        if (dL_dalpha) gA = dL_dalpha[(size_t)r * P + pix];
    }
    const float* __restrict__ bg = bg_all + 3 * r;
    const float tail = gA - (bg[0] * gI0 + bg[1] * gI1 + bg[2] * gI2);

    const uint32_t rowmax = row_max_all(ncon);
    const uint32_t wmax = wave_max_u32(rowmax);
    if (lane == 0) s_wmax[wave] = wmax;
    __syncthreads();
    const uint32_t bmax = min(n, max(max(s_wmax[0], s_wmax[1]), max(s_wmax[2], s_wmax[3])));   // (<= n by construction; the clamp only matters for a tile a failed plan left unrendered)
    // Entries behind every pixel's last contributor get a zero record (every pair slot is written exactly once,
    // so the scratch needs no memset).
    {
        for (uint32_t idx = bmax + threadIdx.x; idx < n; idx += kBlock)
            store_grec<DEPTH_GRAD>(gpair, pair_slot((uint32_t)pairs[beg + idx]), 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f,
                                   0.f, 0.f);
    }
    if (bmax == 0) return;
    if (wave == 0) { SPF_CENSUS(0, 1); SPF_CENSUS(1, bmax); SPF_CENSUS(7, n); }

    float Tr = T_final;
    float sB = -tail * T_final;          // running "behind" scalar of the replay (as in the lists form, see its phase B)

    bool staged = false;
    for (int base = (int)((bmax - 1) / RS) * RS; base >= 0; base -= RS) {
        // flush the previous round's accumulators: ONE plain 48-byte store per (Gaussian, tile) pair
        if (staged) flush_pair<DEPTH_GRAD>(gpair, s_slot[threadIdx.x], s_acc[threadIdx.x]);
        uint32_t bits = 0;
        const uint32_t idx = (uint32_t)base + threadIdx.x;
        staged = (int)threadIdx.x < RS && idx < n && idx < bmax;
        if (staged) {
            const uint32_t gid = (uint32_t)pairs[beg + idx];
            const float4* __restrict__ rp = reinterpret_cast<const float4*>(rec_r + (size_t)gid * kRec);
            const float4 a = rp[0], b = rp[1], cc = rp[2];
            s_p0[threadIdx.x] = make_float4(a.x, a.y, kHalfLog2e * a.z, kHalfLog2e * b.x);   // as in the forward
            s_p1[threadIdx.x] = make_float2(kLog2e * a.w, b.y);
            s_p2[threadIdx.x] = make_float4(cc.x, cc.y, cc.z, b.z);
            s_slot[threadIdx.x] = pair_slot(gid);
#pragma unroll
            for (int k = 0; k < kAcc; ++k) s_acc[threadIdx.x][k] = 0.f;
            bits = block_bits(a.x, a.y, b.w, tx, ty);
        }
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const uint64_t m = __ballot((bits >> k) & 1u);
            if (lane == 0) {
                if (2 * wave < RS / 32) s_mask[2 * wave][k] = (uint32_t)m;
                if (2 * wave + 1 < RS / 32) s_mask[2 * wave + 1][k] = (uint32_t)(m >> 32);
            }
        }
        __syncthreads();
        for (int q = RS / 32 - 1; q >= 0; --q) {
            const uint32_t qbase = (uint32_t)base + (uint32_t)q * 32u;
            if (qbase >= wmax) continue;
            uint32_t m = s_mask[q][beta];
            if (qbase >= rowmax) m = 0;
            else if (qbase + 32u > rowmax) m &= (1u << (rowmax - qbase)) - 1u;
            while (__ballot(m != 0)) {
                const bool act = m != 0;
                const int bit = act ? 31 - __clz((int)m) : 0;
                m &= ~(1u << bit);
                const int j = q * 32 + bit;
                const uint32_t pos = qbase + (uint32_t)bit;
                const float4 p0 = s_p0[j];
                const float2 p1 = s_p1[j];
                const float dx = p0.x - fx, dy = p0.y - fy;
                const float power = lists_power2_scalar(p0, p1.x, fx, fy);                // (log2 units)
                const float Gv = __builtin_amdgcn_exp2f(power);
                const float alpha = fminf(kAlphaMax, p1.y * Gv);
                const bool hit = act && pos < ncon && power <= 0.f && alpha >= kAlphaMin;
                const uint64_t hb = __ballot(hit);
                SPF_CENSUS(2, 1); SPF_CENSUS(3, __popcll(__ballot(act)) >> 4); SPF_CENSUS(4, __popcll(hb)); SPF_CENSUS(5, hb != 0);
                if (hb == 0) continue;
                // (the ten partial gradients are products of two per-lane scalars, w = alpha T and u = G dL/dalpha: only those
                //  are zeroed for the lanes without a hit, the products are formed for the whole wave after the branch)
                float w = 0.f, u = 0.f;
                if (hit) {
                    const float4 p2 = s_p2[j];
                    const float inv1ma = __builtin_amdgcn_rcpf(1.f - alpha);
                    Tr = fmaf(Tr, alpha * inv1ma, Tr);       // (not Tr * inv1ma: see the lists kernel)
                    w = alpha * Tr;
                    // With w_j = alpha_j*T_j and cg_j = c_j . dL/dC (+ depth_j * dL/ddepth):
                    //   dL/dalpha_i = cg_i*T_i - (sum_{j behind i} cg_j*w_j - tail*T_final) / (1 - alpha_i)
                    // (image = C + T_final*bg and alpha_out = 1 - T_final see alpha_i only through T_final); the bracket
                    // is ONE running scalar -- the classic form carries the normalised colour behind the entry instead,
                    // four accumulators and a dot product per hit: 19 instructions where this takes 7
                    float cg = fmaf(p2.x, gI0, fmaf(p2.y, gI1, p2.z * gI2));
                    if (DEPTH_GRAD) cg = fmaf(p2.w, gD, cg);
                    const float dL_dalpha_ = fmaf(cg, Tr, -(sB * inv1ma));
                    sB = fmaf(cg, w, sB);
                    u = Gv * dL_dalpha_;
                }
                const float r_c0 = w * gI0, r_c1 = w * gI1, r_c2 = w * gI2, r_dd = DEPTH_GRAD ? w * gD : 0.f;
                // [3DGS-grad] the min(0.99, .) clamp is straight-through: dL/dG = opacity * dL/dalpha
                const float sg = p1.y * u, r_do = u;
                // v = conic * offset; dL/d(a, b, c) of the 2-D covariance directly (spf_common.h: the classic
                // sum of dL/dconic cancels in float32 for a far-off-centre anisotropic splat, v does not)
                // (from the staged A' = -0.5 log2(e) A, C', B' = -log2(e) B: 1 / -0.5 log2(e) = -2 ln 2, 1 / -log2(e) = -ln 2)
                constexpr float kInvH = -1.38629436111989061883f, kInvL = -0.69314718055994530942f;
                const float vx = fmaf(p0.z * dx, kInvH, (p1.x * dy) * kInvL), vy = fmaf(p0.w * dy, kInvH, (p1.x * dx) * kInvL);
                const float r_dx = -sg * vx, r_dy = -sg * vy;
                const float r_dA = -0.5f * r_dx * vx, r_dB = -r_dx * vy, r_dC = -0.5f * r_dy * vy;
                float z0, z1, z2, z3;                     // quad q of the row: totals of values 4 q .. 4 q + 3
                row_sums10(r_dx, r_dy, r_dA, r_dB, r_dC, r_do, r_c0, r_c1, r_c2, r_dd, z0, z1, z2, z3);
                const bool rowhit = ((hb >> (lane & 48)) & 0xffffull) != 0;
                const int quad = l16 >> 2;
                if ((l16 & 3) == 0 && quad < 3 && rowhit) {
                    // the tile's 16 blocks meet in LDS (ds_add_f32; three lanes of a row add four, four and two values);
                    // HBM sees one record per pair.  (Without a depth gradient value 9 is zero: its accumulator is not read.)
                    float* gp = s_acc[j] + 4 * quad;
                    atomicAdd(gp + 0, z0); atomicAdd(gp + 1, z1);
                    if (quad < 2) { atomicAdd(gp + 2, z2); atomicAdd(gp + 3, z3); }
                }
            }
        }
        __syncthreads();
    }
    if (staged) flush_pair<DEPTH_GRAD>(gpair, s_slot[threadIdx.x], s_acc[threadIdx.x]);
}

// ------------------------------------------------------------------------------------------------
// Backward for sparse tiles ("lists"): three phases per round, no float atomics anywhere.
//
// A round takes the next (back-to-front) entries of the tile's list, as many as fit a pool of kPool LDS slots:
// thread i owns entry hi-1-i and needs one slot per pixel of its cull-disc bounding box clipped to the tile
// (typically 9-16; a block-wide prefix sum hands out slot ranges, the accepted entries are a prefix of i).
//   phase A  the owner ORs bit i into the candidate words of the pixels inside its disc (as in the forward);
//   phase B  thread p owns pixel p: replays ITS candidates back to front (ascending bits; private list, per-lane
//            LDS gathers) and for every contributor stores the two scalars the Gaussian needs from this pixel,
//            w = alpha*T and u = G*dL/dalpha, into slot (entry, pixel-within-box): plain ds_write_b64, one writer;
//   phase C  the owner sums ITS slots against the pixel offsets and the pixels' dL/dC (kept in LDS), in registers,
//            and writes the pair's 48-byte record.
// Work is proportional to real (pixel, Gaussian) contributions and no cross-lane reduction is needed.
// ------------------------------------------------------------------------------------------------
// Round shapes: kRoundL candidate entries per round (kRoundL / 32 mask words), a pool of kPool (w, u) slots, BPC blocks
// per CU.  Many tiles (the chip runs several rounds of blocks): 192 entries / 1,536 slots (12 KB) -- ~31 KB of LDS, five
// blocks per CU; every other shape measured slower there (DESIGN_EXPERIMENTS.md).  FEW tiles with long lists are another
// regime: the launch is one round (or two) of blocks that run as long as their own chain of rounds, LDS is free, and
// longer rounds are fewer barriers per entry -- round 5, same box: the reference's 10-view shape (768 tiles of ~2,800
// entries) 176 -> 149 us with 256 / 2,560 at three blocks per CU (224 / 1,792: 169); BASELINE config 3 (2,048 tiles of
// ~1,300) 144.8 -> 139.5 us with 224 / 1,792 at four (256 / 2,560: 150); config 5 (8,192 tiles) loses with either
// (326 -> 343 / 381).  launch_render_bwd_t picks by the number of tiles.  (Requesting the next round's entries a round
// ahead, forward or backward, bought nothing in either regime: +-0 backward, +2 ... +7 us forward.)
template <bool DEPTH_GRAD, int kRoundL, int kPool, int BPC>
__global__ __launch_bounds__(kBlock, BPC) void spf_render_bwd_lists_kernel(
    const float* __restrict__ rec, const uint64_t* __restrict__ pairs, TileLists tl,
    const uint32_t* __restrict__ tile_flags, const float* __restrict__ bg_all, const float* __restrict__ final_T,
    const uint32_t* __restrict__ n_contrib, const float* __restrict__ dL_dimage, const float* __restrict__ dL_ddepth,
    const float* __restrict__ dL_dalpha, const uint2* __restrict__ pinfo,
    float* __restrict__ gpair, int G, int H, int W, int T, int tiles_x, int RT, uint32_t dense_thr_arg,
    const uint32_t* __restrict__ counters, uint64_t capacity) {
    (void)capacity;
    if (counters[2] != 0u) return;           // failed plan: nothing was rendered; the projection backward poisons the gradients
    const uint32_t dense_thr = dense_thr_arg;
    __shared__ float4 s_p0[kRoundL];                 // x, y | A', C'   (as in the forward)
    __shared__ float4 s_p1[kRoundL];                 // B', opacity, box width (int), depth
    __shared__ float4 s_p2[kRoundL];                 // r, g, b, box (int bits: xl | yl<<4 | (bw-1)<<8 | off<<12)
    __shared__ __attribute__((aligned(16))) uint32_t s_pm[kRoundL / 32][kBlock];  // [32-entry word][pixel]: candidate bits
    __shared__ float2 s_pool[kPool];                // (w, u) slots of this round's entries
    __shared__ float4 s_gI[kBlock];                 // per pixel: dL/dC (rgb), dL/ddepth
    __shared__ uint32_t s_w[4];                     // per-wave scratch (max / scan totals)
    __shared__ uint32_t s_wacc[4];                  // per-wave scratch (accepted-entry counts)
    __shared__ uint32_t s_wmax[4];                  // per-wave last contributor (read once, before the rounds)

    int vid;
    uint32_t beg, n;
    bool dense_tile;
    if (!lists_tile(tl, tile_flags, dense_thr, false, RT, vid, beg, n, dense_tile)) return;
    const int r = vid / T, tile = vid - r * T;
    const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int X0 = tx * kTile, Y0 = ty * kTile;
    if (n == 0) return;
    if (dense_tile) {                                                                // (block-uniform)
        static_assert(sizeof(float2) * kPool >= (sizeof(uint32_t) + sizeof(float) * kAcc) * kRoundL, "rows scratch fits the pool");
        uint32_t* const s_slot = reinterpret_cast<uint32_t*>(s_pool);
        bwd_rows_tile<DEPTH_GRAD, kRoundL>(s_p0, reinterpret_cast<float2*>(s_p1), s_p2, s_slot,
                                           reinterpret_cast<float (*)[kAcc]>(s_slot + kRoundL),
                                           reinterpret_cast<uint32_t (*)[16]>(&s_pm[0][0]), s_wmax,
                                           rec + (size_t)r * G * kRec, pairs, beg, n, r, tx, ty, bg_all, final_T, n_contrib,
                                           dL_dimage, dL_ddepth, dL_dalpha, pinfo, gpair, G, H, W);
        return;
    }
    const size_t P = (size_t)H * W;
    // ---- per-pixel state: thread <-> pixel in the natural order ----
    // (Rounds 1 - 3 put the pixels on lanes by the number of contributors the forward recorded -- an LDS counting sort,
    // three barriers, the state parked in LDS and picked up again.  It paid while the replay was most of the kernel; at
    // 28 % of the wave time it cost 2.4 us more than it gave on C2, and it went together with the forward's per-pixel
    // contributor count in round 4.)
    const int mypix = tid;
    float T_final = 1.f, gI0 = 0.f, gI1 = 0.f, gI2 = 0.f, gD = 0.f, gA = 0.f;
    uint32_t ncon = 0;
    {
        const int qx = X0 + (tid & 15), qy = Y0 + (tid >> 4);
        if (qx < W && qy < H) {
            const size_t qpix = (size_t)qy * W + qx;
            ncon = n_contrib[(size_t)r * P + qpix];
            T_final = final_T[(size_t)r * P + qpix];
            if (dL_dimage) {
                const float* __restrict__ gi = dL_dimage + (size_t)r * 3 * P;
                gI0 = gi[qpix]; gI1 = gi[P + qpix]; gI2 = gi[2 * P + qpix];
            }
            if (DEPTH_GRAD) gD = dL_ddepth[(size_t)r * P + qpix];
            if (dL_dalpha) gA = dL_dalpha[(size_t)r * P + qpix];
        }
        s_gI[tid] = make_float4(gI0, gI1, gI2, DEPTH_GRAD ? gD : 1.f);   // phase C's table; .w == 1 lets it fold sum(u) into a packed fma
    }
    const int lx = mypix & 15, ly = mypix >> 4;
    const int px = X0 + lx, py = Y0 + ly;
    const float* __restrict__ rec_r = rec + (size_t)r * G * kRec;
    float fx = (float)px, fy = (float)py;
    asm volatile("" : "+v"(fx), "+v"(fy));
    auto pair_slot = [&](uint32_t gid) -> uint32_t {                  // ONE 8-byte gather: (rect, first pair)
        const uint2 pi = pinfo[(size_t)r * G + gid];
        const int x0 = pi.x & 0xff, y0 = (pi.x >> 8) & 0xff, x1 = (pi.x >> 16) & 0xff;
        return pi.y + (uint32_t)((ty - y0) * (x1 - x0) + (tx - x0));
    };
    const float* __restrict__ bg = bg_all + 3 * r;
    const float tail = gA - (bg[0] * gI0 + bg[1] * gI1 + bg[2] * gI2);

    // (its own four words, not s_w: the first round writes s_w again with no barrier after this read, and a wave that
    // is held up for the length of that round's gather would then take a prefix total for a list position, i.e. run
    // with its own idea of the list's end -- seen once per ~10^6 tiles, as a barrier mismatch and a wild read)
    const uint32_t wmax = wave_max_u32(ncon);
    if (lane == 0) s_wmax[wave] = wmax;
    __syncthreads();
    const uint32_t bmax = min(n, max(max(s_wmax[0], s_wmax[1]), max(s_wmax[2], s_wmax[3])));   // (clamp: see bwd_rows_tile)
    {   // entries behind every pixel's last contributor: zero record (each pair slot is written exactly once)
        for (uint32_t idx = bmax + tid; idx < n; idx += kBlock)
            store_grec<DEPTH_GRAD>(gpair, pair_slot((uint32_t)pairs[beg + idx]), 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f,
                                   0.f, 0.f);
    }
    if (bmax == 0) return;

    float Tr = T_final;
    float sB = -tail * T_final;          // running "behind" scalar of the replay (see phase B)

    uint32_t hi = bmax;   // entries [0, hi) are still to be replayed
    while (hi > 0) {
        // ---- candidates of this round: thread i <-> entry hi-1-i; slot demand; prefix sum ----
#pragma unroll
        for (int w = 0; w < kRoundL / 32; ++w) s_pm[w][tid] = 0u;
        const bool have = tid < kRoundL && (uint32_t)tid < hi;
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a, cc = a;
        uint32_t gid = 0, pslot = 0;
        int xl = 0, yl = 0, bw = 0, bh = 0;
        if (have) {
            gid = (uint32_t)pairs[beg + (hi - 1u - (uint32_t)tid)];
            const float4* __restrict__ rp = reinterpret_cast<const float4*>(rec_r + (size_t)gid * kRec);
            a = rp[0]; b = rp[1]; cc = rp[2];
            // (the pair's slot hangs off two more gathers -- rect, pair_off: issued here, next to the record's, they
            //  cost no round trip of their own; looked up at the end of phase C they were one per round, exposed)
            if (!DEPTH_GRAD) pslot = pair_slot(gid);
            const TileBox tb = clipped_box(a.x, a.y, b.w, X0, Y0);
            xl = tb.xl; yl = tb.yl; bw = tb.bw; bh = tb.bh;
        }
        const uint32_t size = (uint32_t)(bw * bh);
        const uint32_t inc = wave_iscan_u32(size);      // DPP ladder: 7 VALU adds, no LDS permutes
        // (the barrier that closed the previous round makes s_w / s_wacc / s_pool / s_p* reusable here)
        if (lane == kWave - 1) s_w[wave] = inc;
        __syncthreads();
        uint32_t off = inc - size;
        for (int w = 0; w < wave; ++w) off += s_w[w];
        // accepted = a prefix of the threads: thread t is in iff the slot demand of threads 0..t fits the pool; the
        // per-wave counts travel with the staging barrier below (no __syncthreads_count round trip)
        const bool acc = have && off + size <= (uint32_t)kPool;
        const uint64_t accb = __ballot(acc);
        if (lane == 0) s_wacc[wave] = (uint32_t)__popcll(accb);
        if (acc) {
            s_p0[tid] = make_float4(a.x, a.y, kHalfLog2e * a.z, kHalfLog2e * b.x);
            // slot of pixel (lx, ly) = off + (ly - yl) * bw + (lx - xl) = [off - yl*bw - xl] + bw*ly + lx: the box width
            // and the (signed) bracket travel as integers in the two fields phase B has no other use for
            // (as BYTE offsets into the pool, 8 bytes per slot: the replay then needs one v_mad_i32_i24 and one add)
            // (.w: the depth for the depth-gradient kernel; otherwise free -- it carries the pair slot to phase C: 32,560
            //  bytes of LDS, i.e. an array of its own, measured 4 blocks per CU instead of 5 and 157 -> 177 us)
            s_p1[tid] = make_float4(kLog2e * a.w, b.y, __int_as_float(8 * bw), DEPTH_GRAD ? b.z : __uint_as_float(pslot));
            s_p2[tid] = make_float4(cc.x, cc.y, cc.z, __int_as_float(8 * ((int)off - yl * bw - xl)));
        }
#pragma unroll
        for (int k = 0; k < kPool / kBlock; ++k) s_pool[k * kBlock + tid] = make_float2(0.f, 0.f);
        // ---- phase A (the candidate words were cleared before the prefix-sum barrier above) ----
        if (acc) scatter_box(s_pm, tid, a.x, a.y, b.w, X0, Y0, xl, yl, bw, bh);
        __syncthreads();
        const int cnt = (int)(s_wacc[0] + s_wacc[1] + s_wacc[2] + s_wacc[3]);   // >= 1: one entry needs <= 256 slots
        // ---- phase B: ascending bits = descending list position ----
        const int nw = (cnt + 31) >> 5;
        if (hi - (uint32_t)cnt < wmax) {
            // contributors of this pixel are entries < ncon, i.e. thread indices >= hi - ncon =: jmin.  The walk starts
            // at the word that holds bit jmin (only that word needs masking) and refills are plain loads.
            const uint32_t jmin = ncon < hi ? hi - ncon : 0u;
            const char* __restrict__ wcol = reinterpret_cast<const char*>(&s_pm[0][mypix]);   // word w: wcol + 1024 w
            const int w0 = (int)min(jmin >> 5, (uint32_t)(kRoundL / 32 - 1));
            int wb = w0 << 9;                                            // 512 * word = byte offset of entry 32 w (16 B each)
            uint32_t m = *reinterpret_cast<const uint32_t*>(wcol + 2 * wb);
            m = jmin < 32u * (uint32_t)nw ? m & (~0u << (jmin & 31u)) : 0u;
            // which of the later words of this pixel's column hold a candidate at all (mean 0.7 candidates per word: most
            // are empty); words >= nw were cleared with the rest and read as empty
            uint32_t nz = 0u;
#pragma unroll
            for (int k = 1; k < kRoundL / 32; ++k)
                nz |= min(*reinterpret_cast<const uint32_t*>(wcol + 1024 * k), 1u) << k;
            nz &= ~1u << w0;
            // next candidate of this lane as the byte offset of its staged record.  has = false: none left -- the
            // offset is then 16 bytes below a word's first record (v_ffbl_b32 of 0 is -1): still inside the kernel's LDS
            // (s_p0..2 do not start it), and what is read there is not used.  A refill jumps straight to the next
            // non-empty word: no loop.
            auto next = [&](bool& has) -> int {
                if (m == 0u && nz != 0u) {       // (a divergent `if` is skipped as a whole when no lane takes it)
                    int w;
                    asm("v_ffbl_b32 %0, %1" : "=v"(w) : "v"(nz));
                    nz &= nz - 1u;
                    wb = w << 9;
                    m = *reinterpret_cast<const uint32_t*>(wcol + 2 * wb);
                }
                int bit;
                asm("v_ffbl_b32 %0, %1" : "=v"(bit) : "v"(m));
                has = bit >= 0;                  // (not `m != 0`: the compiler would take that from the carry of m - 1 and
                m &= m - 1u;                     //  then needs two more instructions to turn it into a wave mask)
                return wb + (bit << 4);
            };
            const int lx8 = 8 * lx;
            auto replay = [&](bool has, const float4& p0, const float4& p1, const float4& p2) {
                const float pw = lists_power2_scalar(p0, p1.x, fx, fy);
                const float Gv = __builtin_amdgcn_exp2f(pw);
                const float alpha = fminf(kAlphaMax, p1.y * Gv);
                if (has && pw <= 0.f && alpha >= kAlphaMin) {
                    // With w_j = alpha_j*T_j and cg_j = c_j . dL/dC (+ depth_j * dL/ddepth):
                    //   dL/dalpha_i = cg_i*T_i - (sum_{j behind i} cg_j*w_j - tail*T_final) / (1 - alpha_i)
                    // (image = C + T_final*bg and alpha_out = 1 - T_final see alpha_i only through T_final); the
                    // bracket is ONE running scalar, sB.
                    const float inv1ma = __builtin_amdgcn_rcpf(1.f - alpha);
                    // T_i = T_{i+1} / (1 - alpha_i), formed as T + T * (alpha / (1 - alpha)): the hardware reciprocal's
                    // error (1 ulp, not centred) then enters scaled by alpha instead of in full.  As T * inv1ma it walked
                    // T off by ~0.4 ulp per entry -- 1e-3 of dL/dcolour at the front of a 40,000-entry list (round 4:
                    // seen once the knife-edge masks stopped covering half of that test's pixels).
                    Tr = fmaf(Tr, alpha * inv1ma, Tr);                           // T_i: transmittance in front of i
                    float cg = fmaf(p2.x, gI0, fmaf(p2.y, gI1, p2.z * gI2));
                    if (DEPTH_GRAD) cg = fmaf(p1.w, gD, cg);
                    const float dL_dalpha_ = fmaf(cg, Tr, -(sB * inv1ma));
                    const float wgt = alpha * Tr;
                    sB = fmaf(cg, wgt, sB);
                    const int k8 = __mul24(__float_as_int(p1.z), ly) + __float_as_int(p2.w) + lx8;   // v_mad_i32_i24 + add
                    *reinterpret_cast<float2*>(reinterpret_cast<char*>(s_pool) + k8) = make_float2(wgt, Gv * dL_dalpha_);
                }
            };
            auto rec0 = [&](int j) { return *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(s_p0) + j); };
            auto rec1 = [&](int j) { return *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(s_p1) + j); };
            auto rec2 = [&](int j) { return *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(s_p2) + j); };
            // software pipeline unrolled by two (see the forward)
            // (the loop tests are carried as wave masks -- scalar registers -- so that they cost no vector instruction)
            bool ha, hb;
            int ja = next(ha), jb;
            uint64_t ma = lane_ballot(ha), mb;
            float4 a0 = rec0(ja), a1 = rec1(ja), a2 = rec2(ja), b0, b1, b2;
            while (true) {
                if (!ma) break;
                jb = next(hb);
                mb = lane_ballot(hb);
                b0 = rec0(jb); b1 = rec1(jb); b2 = rec2(jb);
                replay(ha, a0, a1, a2);
                if (!mb) break;
                ja = next(ha);
                ma = lane_ballot(ha);
                a0 = rec0(ja); a1 = rec1(ja); a2 = rec2(ja);
                replay(hb, b0, b1, b2);
            }
        }
        __syncthreads();
        // ---- phase C: packed fp32 (v_pk_fma_f32), six arithmetic instructions per slot ----
        if (acc) {
            v2f c01 = {0.f, 0.f}, c2s = {0.f, 0.f};      // (dL/dr, dL/dg), (dL/db, sum u)
            v2f s1 = {0.f, 0.f}, s2 = {0.f, 0.f};        // (sum u dx, sum u dy), (sum u dx^2, sum u dy^2)
            float sxy = 0.f, cd = 0.f;
            const float2* __restrict__ hp = s_pool + off;
            const float4* __restrict__ gp = s_gI + yl * kTile + xl;
            // The moments are taken about a LOCAL origin -- the box pixel nearest to the centre -- and shifted to the
            // centre analytically at the end: offsets stay small integers whatever the distance of the centre, so the
            // sums carry no cancellation (spf_common.h); for a centre inside its box the shift is below one pixel.
            const float bx0 = (float)(X0 + xl), by0 = (float)(Y0 + yl);
            const float ox = fminf(fmaxf(rintf(a.x), bx0), bx0 + (float)(bw - 1));
            const float oy = fminf(fmaxf(rintf(a.y), by0), by0 + (float)(bh - 1));
            const float dx0 = ox - bx0;
            v2f d = {dx0, oy - by0};
            // rows outside, columns inside: the column loop is one pointer step, one offset step and the six packed
            // accumulations per slot (a flat loop over the box pays ~7 instructions per slot for the row wrap)
            const float dxe = dx0 - (float)bw;
            const int nrow = bw > 0 ? bh : 0;            // (an empty box has no column to end the inner loop on)
            for (int y = 0; y < nrow; ++y) {
                const float2* __restrict__ hq = hp;
                const float4* __restrict__ gq = gp;
                d.x = dx0;
#pragma unroll 1
                do {
                    const float2 h = *hq;                // (w, u) of this pixel
                    const float4 gi = *gq;               // (dL/dC, dL/ddepth or 1)
                    const v2f dc = d;
                    ++hq; ++gq; d.x -= 1.f;
                    const v2f hw = {h.x, h.x}, hu = {h.y, h.y}, hwu = {h.x, h.y};
                    c01 = __builtin_elementwise_fma(hw, v2f{gi.x, gi.y}, c01);
                    if (DEPTH_GRAD) {
                        c2s = __builtin_elementwise_fma(hwu, v2f{gi.z, 1.f}, c2s);
                        cd = fmaf(h.x, gi.w, cd);
                    } else {
                        c2s = __builtin_elementwise_fma(hwu, v2f{gi.z, gi.w}, c2s);   // gi.w == 1
                    }
                    const v2f t = hu * dc;
                    s1 += t;
                    s2 = __builtin_elementwise_fma(t, dc, s2);
                    sxy = fmaf(t.x, dc.y, sxy);
                } while (d.x > dxe);                     // (offsets are small integers: exact; bw >= 1 here)
                hp += bw; gp += kTile; d.y -= 1.f;
            }
            const float o = b.y;   // [3DGS-grad] dL/dG = opacity * dL/dalpha (the 0.99 clamp is straight-through)
            // With e = origin - pixel (the loop's offsets), d0 = centre - origin, Q = conic, v = Q (d0 + e) = v0 + Q e:
            //   sum u v = v0 M0 + Q E1,   sum u v v^T = v0 v0^T M0 + v0 (Q E1)^T + (Q E1) v0^T + Q E2 Q
            // and dL/d(centre) = -o sum u v, dL/d(a, b, c) = o (1/2 sum u vx^2, sum u vx vy, 1/2 sum u vy^2).
            const float M0 = c2s.y;
            const v2f P1 = {a.z, a.w}, P2 = {a.w, b.x};                     // conic rows (A, B), (B, C)
            const float d0x = a.x - ox, d0y = a.y - oy;
            const v2f v0 = __builtin_elementwise_fma(P2, v2f{d0y, d0y}, P1 * v2f{d0x, d0x});
            const v2f L = __builtin_elementwise_fma(P2, v2f{s1.y, s1.y}, P1 * v2f{s1.x, s1.x});      // Q E1
            const v2f S1 = __builtin_elementwise_fma(v0, v2f{M0, M0}, L);
            const v2f R0 = __builtin_elementwise_fma(P2, v2f{sxy, sxy}, P1 * v2f{s2.x, s2.x});       // rows of E2 Q
            const v2f R1 = __builtin_elementwise_fma(P2, v2f{s2.y, s2.y}, P1 * v2f{sxy, sxy});
            const v2f Qx = __builtin_elementwise_fma(R1, v2f{a.w, a.w}, R0 * v2f{a.z, a.z});         // (QEQ_xx, QEQ_xy)
            const float Qyy = fmaf(b.x, R1.y, a.w * R0.y);
            const v2f Tt = __builtin_elementwise_fma(v0, v2f{M0, M0}, L + L);
            const float Sxx = fmaf(v0.x, Tt.x, Qx.x), Syy = fmaf(v0.y, Tt.y, Qyy);
            const float Sxy = fmaf(v0.x, fmaf(v0.y, M0, L.y), fmaf(v0.y, L.x, Qx.y));
            const uint32_t my_slot = DEPTH_GRAD ? pair_slot(gid) : __float_as_uint(s_p1[tid].w);
            store_grec<DEPTH_GRAD>(gpair, my_slot, -o * S1.x, -o * S1.y, 0.5f * o * Sxx, o * Sxy, 0.5f * o * Syy,
                                   M0, c01.x, c01.y, c2s.x, cd);
        }
        hi -= (uint32_t)cnt;
        __syncthreads();                         // round over: LDS scratch may be reused
    }
}

// ---- launchers ------------------------------------------------------------------------------------
uint32_t dense_threshold_fwd();
uint32_t dense_threshold() {                       // (read per call: the tests flip it)
    const char* const e = getenv("SPF_DENSE_AREA");
    return e ? (uint32_t)atoi(e) : SPF_DENSE_AREA;
}
// The forward's own threshold (>= the backward's).  Round-5 sweep (same box, C2 with footprints x 3 / 6 / 10 / 30 and C5 x 1 /
// 4): the lists FORWARD beats the rows forward up to far denser tiles than the lists backward beats the rows backward
// (x 6: forward 201 us through rows vs 125 through lists, backward 525 vs 566; x 10: 269 vs 190 and 742 vs 1,468; only at
// x 30 do the rows win the forward, 508 vs 1,228) -- with one threshold for both, every tile handed to the rows form for
// the backward's sake cost the forward 20 - 40 %.  SPF_DENSE_AREA (experiments) pins both.
uint32_t dense_threshold_fwd() {
    const char* const f = getenv("SPF_DENSE_AREA_FWD");
    const char* const e = getenv("SPF_DENSE_AREA");
    const uint32_t v = f ? (uint32_t)atoi(f) : (e ? (uint32_t)atoi(e) : SPF_DENSE_AREA_FWD), b = dense_threshold();
    return v > b ? v : b;
}

// ONE launch composites every tile: the kernel takes the "rows" or the "lists" form per tile (launch-order flag, or
// tile_flags / list length in image order).  Rounds 2 - 4 ran two kernels on forked streams and skipped one of them from
// the plan's dense-tile census (`dense_hint`); the empty or near-empty second launch cost 7 - 20 us a direction on every
// scene with a handful of dense tiles, and the fork's event edges on every exact call.
hipError_t launch_render_fwd(const SpfDims& d, const SpfInputs& in, const SpfState& st, const SpfOutputs& out,
                             uint64_t capacity, int T, int tiles_x, bool ordered, hipStream_t stream) {
    const int RT = d.S * d.V * T;
    TileLists tlo = tile_lists(st, d);
    if (ordered) tlo.order = tile_order_ptr(st, d, RT);
    const int grid = (RT + 7) / 8 * 8;
    const char* const fe = getenv("SPF_FWD_STAGE");           // ("256" / "512" pins the instantiation: experiments, tests)
    const int stage = fe ? atoi(fe) : (RT <= kFwdLongRoundsMaxTiles ? 512 : 256);
    if (stage == 512)
        spf_render_fwd_lists_kernel<512><<<grid, kBlock, 0, stream>>>(
            st.rec, st.pairs, tlo, st.tile_flags, st.counters, capacity, in.bg, out.image, out.depth,
            out.alpha, st.final_T, st.n_contrib, d.G, d.H, d.W, T, tiles_x, RT, dense_threshold_fwd());
    else
        spf_render_fwd_lists_kernel<256><<<grid, kBlock, 0, stream>>>(
            st.rec, st.pairs, tlo, st.tile_flags, st.counters, capacity, in.bg, out.image, out.depth,
            out.alpha, st.final_T, st.n_contrib, d.G, d.H, d.W, T, tiles_x, RT, dense_threshold_fwd());
    return hipGetLastError();
}

template <bool DG>
static void launch_render_bwd_t(const SpfDims& d, const SpfInputs& in, const SpfState& st, const SpfGrads& g, int T,
                                int tiles_x, int RT, int grid, uint64_t capacity, bool ordered, hipStream_t stream) {
    TileLists tlo = tile_lists(st, d);
    if (ordered) tlo.order = tile_order_ptr(st, d, RT);
    const uint2* const pinfo = reinterpret_cast<const uint2*>(st.pair_off);
    // round shape by the number of tiles (see the kernel); SPF_BWD_ROUNDS=192 / 224 / 256 pins one (experiments, tests)
    const char* const fe = getenv("SPF_BWD_ROUNDS");          // (read per call: the tests flip it)
    const int forced = fe ? atoi(fe) : 0;
    const int shape = forced ? forced : (RT <= 768 ? 256 : (RT <= 2048 ? 224 : 192));
#define SPF_BWD_LISTS(RL, PL, BPC)                                                                                      \
    spf_render_bwd_lists_kernel<DG, RL, PL, BPC><<<grid, kBlock, 0, stream>>>(                                            \
        st.rec, st.pairs, tlo, st.tile_flags, in.bg, st.final_T, st.n_contrib, g.dL_dimage, g.dL_ddepth, g.dL_dalpha,     \
        pinfo, g.gpair, d.G, d.H, d.W, T, tiles_x, RT, dense_threshold(), st.counters, capacity)
    if (shape == 256) SPF_BWD_LISTS(256, 2560, 3);
    else if (shape == 224) SPF_BWD_LISTS(224, 1792, 4);
    else SPF_BWD_LISTS(192, 1536, 5);
#undef SPF_BWD_LISTS
}

hipError_t launch_render_bwd(const SpfDims& d, const SpfInputs& in, const SpfState& st, const SpfGrads& g, int T,
                             int tiles_x, uint64_t capacity, bool ordered, hipStream_t stream) {
    const int RT = d.S * d.V * T;
    const int grid = (RT + 7) / 8 * 8;
    if (g.dL_ddepth) launch_render_bwd_t<true>(d, in, st, g, T, tiles_x, RT, grid, capacity, ordered, stream);
    else launch_render_bwd_t<false>(d, in, st, g, T, tiles_x, RT, grid, capacity, ordered, stream);
    return hipGetLastError();
}

}  // namespace spf
#ifdef SPF_ROWS_CENSUS
extern "C" int spf_debug_rows_census(unsigned long long* out, int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(spf::g_rows_census), sizeof(unsigned long long) * 8) != hipSuccess) return -1;
    if (reset) { unsigned long long z[8] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(spf::g_rows_census), z, sizeof(z)) != hipSuccess) return -1; }
    return 0;
}
#endif
