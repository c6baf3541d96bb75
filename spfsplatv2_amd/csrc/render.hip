// 16x16-tile alpha compositing, forward and backward, for gfx950 (wave64).
//
// Block = one tile = 256 threads = 4 waves; wave w owns the 8x8 sub-tile (w&1, w>>1), so a
// whole wave can drop a Gaussian with ONE scalar decision:
//   * the tile's depth-sorted list is staged 256 entries at a time into LDS (one coalesced
//     8-byte key load + one 48-byte record gather per thread),
//   * while staging, each thread tests its entry against the four sub-tiles with a conservative
//     bound (record field 7, see project.hip) and the results are turned into 64-bit wave masks
//     with __ballot -- afterwards every wave walks only the set bits of its own masks (scalar
//     s_ff1 loop), reading the entry back from LDS as a broadcast,
//   * per-lane early termination (T < 1e-4) is folded into a wave ballot and a block-wide
//     __syncthreads_and so a saturated tile stops streaming its list.
// A culled entry is one whose alpha is < 1/255 at every pixel of the sub-tile, i.e. one the
// per-pixel loop would have skipped anyway, so results are identical to the un-culled loop.
//
// Semantics: SURVEY.md Appendix B #10/#11 (restated in oracle/splat_ref.py::composite).
#include "spf_common.h"

namespace spf {

constexpr int kStage = 256;  // list entries staged per round (one per thread)

struct TileCtx {
    int r, tile, tx, ty, wave, lane, px, py;
    bool inside;
};

__device__ __forceinline__ bool tile_ctx(TileCtx& c, int RT, int T, int tiles_x, int H, int W) {
    const int vid = xcd_remap(blockIdx.x, gridDim.x);
    if (vid >= RT) return false;
    c.r = vid / T;
    c.tile = vid - c.r * T;
    c.ty = c.tile / tiles_x;
    c.tx = c.tile - c.ty * tiles_x;
    c.wave = threadIdx.x >> 6;
    c.lane = threadIdx.x & 63;
    c.px = c.tx * kTile + (c.wave & 1) * 8 + (c.lane & 7);
    c.py = c.ty * kTile + (c.wave >> 1) * 8 + (c.lane >> 3);
    c.inside = c.px < W && c.py < H;
    return true;
}

// 4-bit mask: which 8x8 sub-tiles of tile (tx,ty) can be touched by a Gaussian at (gx,gy) with
// squared cull radius r2.
__device__ __forceinline__ uint32_t subtile_bits(float gx, float gy, float r2, int tx, int ty) {
    uint32_t bits = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        const float x0 = (float)(tx * kTile + (w & 1) * 8), y0 = (float)(ty * kTile + (w >> 1) * 8);
        const float dx = fmaxf(fmaxf(x0 - gx, gx - (x0 + 7.f)), 0.f);
        const float dy = fmaxf(fmaxf(y0 - gy, gy - (y0 + 7.f)), 0.f);
        if (!(dx * dx + dy * dy > r2)) bits |= 1u << w;
    }
    return bits;
}

// ------------------------------------------------------------------------------------------------
// Forward
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void spf_render_fwd_kernel(
    const float* __restrict__ rec, const uint64_t* __restrict__ pairs, const uint32_t* __restrict__ tile_start,
    const uint32_t* __restrict__ counters, uint64_t capacity, const float* __restrict__ bg_all,
    float* __restrict__ image, float* __restrict__ depth_out, float* __restrict__ alpha_out,
    float* __restrict__ final_T, uint32_t* __restrict__ n_contrib, int G, int H, int W, int T, int tiles_x, int RT) {
    __shared__ float4 s_p0[kStage];   // x, y, A, B
    __shared__ float2 s_p1[kStage];   // C, opacity
    __shared__ float4 s_p2[kStage];   // r, g, b, depth
    __shared__ uint64_t s_mask[4][4];  // [staged chunk of 64][consumer wave]

    if (counters[0] > capacity) return;
    TileCtx c;
    if (!tile_ctx(c, RT, T, tiles_x, H, W)) return;
    const uint32_t beg = tile_start[(size_t)c.r * T + c.tile];
    const uint32_t n = tile_start[(size_t)c.r * T + c.tile + 1] - beg;
    const float* __restrict__ rec_r = rec + (size_t)c.r * G * kRec;
    const float fx = (float)c.px, fy = (float)c.py;

    float Tr = 1.0f, C0 = 0.f, C1 = 0.f, C2 = 0.f, Dp = 0.f;
    uint32_t last = 0;
    bool done = !c.inside;
    bool wave_done = __ballot(!done) == 0;

    for (uint32_t base = 0; base < n; base += kStage) {
        // ---- stage ----
        uint32_t bits = 0;
        const uint32_t idx = base + threadIdx.x;
        if (idx < n) {
            const uint32_t gid = (uint32_t)pairs[beg + idx];
            const float4* __restrict__ rp = reinterpret_cast<const float4*>(rec_r + (size_t)gid * kRec);
            const float4 a = rp[0], b = rp[1], cc = rp[2];
            s_p0[threadIdx.x] = a;
            s_p1[threadIdx.x] = make_float2(b.x, b.y);
            s_p2[threadIdx.x] = make_float4(cc.x, cc.y, cc.z, b.z);
            bits = subtile_bits(a.x, a.y, b.w, c.tx, c.ty);
        }
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const uint64_t m = __ballot((bits >> w) & 1u);
            if (c.lane == 0) s_mask[c.wave][w] = m;
        }
        __syncthreads();
        // ---- consume ----
        if (!wave_done) {
            for (int ch = 0; ch < 4; ++ch) {
                uint64_t m = readfirstlane64(s_mask[ch][c.wave]);
                while (m) {
                    const int j = ch * 64 + __builtin_ctzll(m);
                    m &= m - 1;
                    const float4 p0 = s_p0[j];
                    const float2 p1 = s_p1[j];
                    const float dx = p0.x - fx, dy = p0.y - fy;
                    const float power = -0.5f * (p0.z * dx * dx + p1.x * dy * dy) - p0.w * dx * dy;
                    const float alpha = fminf(kAlphaMax, p1.y * __expf(power));
                    const bool hit = !done && power <= 0.f && alpha >= kAlphaMin;
                    if (hit) {
                        const float test_T = Tr * (1.f - alpha);
                        if (test_T < kTMin) {
                            done = true;
                        } else {
                            const float4 p2 = s_p2[j];
                            const float w = alpha * Tr;
                            C0 += p2.x * w; C1 += p2.y * w; C2 += p2.z * w; Dp += p2.w * w;
                            Tr = test_T;
                            last = base + j + 1;
                        }
                    }
                }
                if (__ballot(!done) == 0) { wave_done = true; break; }
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
// Backward: back-to-front replay.  Per-Gaussian partial gradients are summed across the wave with
// DPP adds and leave the wave as ONE 10-lane global atomic instruction per (wave, Gaussian).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void spf_render_bwd_kernel(
    const float* __restrict__ rec, const uint64_t* __restrict__ pairs, const uint32_t* __restrict__ tile_start,
    const float* __restrict__ bg_all, const float* __restrict__ final_T, const uint32_t* __restrict__ n_contrib,
    const float* __restrict__ dL_dimage, const float* __restrict__ dL_ddepth, const float* __restrict__ dL_dalpha,
    float* __restrict__ grec, int G, int H, int W, int T, int tiles_x, int RT) {
    __shared__ float4 s_p0[kStage];
    __shared__ float2 s_p1[kStage];
    __shared__ float4 s_p2[kStage];
    __shared__ uint32_t s_gid[kStage];
    __shared__ uint64_t s_mask[4][4];
    __shared__ uint32_t s_wmax[4];

    TileCtx c;
    if (!tile_ctx(c, RT, T, tiles_x, H, W)) return;
    const uint32_t beg = tile_start[(size_t)c.r * T + c.tile];
    const uint32_t n = tile_start[(size_t)c.r * T + c.tile + 1] - beg;
    if (n == 0) return;
    const float* __restrict__ rec_r = rec + (size_t)c.r * G * kRec;
    float* __restrict__ grec_r = grec + (size_t)c.r * G * kRec;
    const float fx = (float)c.px, fy = (float)c.py;
    const size_t P = (size_t)H * W, pix = (size_t)c.py * W + c.px;

    float T_final = 1.f, gI0 = 0.f, gI1 = 0.f, gI2 = 0.f, gD = 0.f, gA = 0.f;
    uint32_t ncon = 0;
    if (c.inside) {
        T_final = final_T[(size_t)c.r * P + pix];
        ncon = n_contrib[(size_t)c.r * P + pix];
        if (dL_dimage) {
            const float* __restrict__ gi = dL_dimage + (size_t)c.r * 3 * P;
            gI0 = gi[pix]; gI1 = gi[P + pix]; gI2 = gi[2 * P + pix];
        }
        if (dL_ddepth) gD = dL_ddepth[(size_t)c.r * P + pix];
        if (dL_dalpha) gA = dL_dalpha[(size_t)c.r * P + pix];
    }
    const float* __restrict__ bg = bg_all + 3 * c.r;
    const float tail = gA - (bg[0] * gI0 + bg[1] * gI1 + bg[2] * gI2);

    const uint32_t wmax = wave_max_u32(ncon);
    if (c.lane == 0) s_wmax[c.wave] = wmax;
    __syncthreads();
    const uint32_t bmax = max(max(s_wmax[0], s_wmax[1]), max(s_wmax[2], s_wmax[3]));
    if (bmax == 0) return;

    float Tr = T_final;
    float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, accD = 0.f;   // colour/depth behind the current entry
    float last_alpha = 0.f, lc0 = 0.f, lc1 = 0.f, lc2 = 0.f, lD = 0.f;

    for (int base = (int)((bmax - 1) / kStage) * kStage; base >= 0; base -= kStage) {
        uint32_t bits = 0;
        const uint32_t idx = (uint32_t)base + threadIdx.x;
        if (idx < n && idx < bmax) {
            const uint32_t gid = (uint32_t)pairs[beg + idx];
            const float4* __restrict__ rp = reinterpret_cast<const float4*>(rec_r + (size_t)gid * kRec);
            const float4 a = rp[0], b = rp[1], cc = rp[2];
            s_p0[threadIdx.x] = a;
            s_p1[threadIdx.x] = make_float2(b.x, b.y);
            s_p2[threadIdx.x] = make_float4(cc.x, cc.y, cc.z, b.z);
            s_gid[threadIdx.x] = gid;
            bits = subtile_bits(a.x, a.y, b.w, c.tx, c.ty);
        }
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const uint64_t m = __ballot((bits >> w) & 1u);
            if (c.lane == 0) s_mask[c.wave][w] = m;
        }
        __syncthreads();
        for (int ch = 3; ch >= 0; --ch) {
            const int cbase = base + ch * 64;
            if ((uint32_t)cbase >= wmax) continue;
            uint64_t m = readfirstlane64(s_mask[ch][c.wave]);
            if ((uint32_t)cbase + 64u > wmax) m &= (1ull << (wmax - (uint32_t)cbase)) - 1ull;
            while (m) {
                const int bit = 63 - __builtin_clzll(m);
                m &= ~(1ull << bit);
                const int j = ch * 64 + bit;
                const uint32_t pos = (uint32_t)base + (uint32_t)j;
                const float4 p0 = s_p0[j];
                const float2 p1 = s_p1[j];
                const float dx = p0.x - fx, dy = p0.y - fy;
                const float power = -0.5f * (p0.z * dx * dx + p1.x * dy * dy) - p0.w * dx * dy;
                const float Gv = __expf(power);
                const float alpha = fminf(kAlphaMax, p1.y * Gv);
                const bool hit = pos < ncon && power <= 0.f && alpha >= kAlphaMin;
                if (__ballot(hit) == 0) continue;
                float r_dx = 0.f, r_dy = 0.f, r_dA = 0.f, r_dB = 0.f, r_dC = 0.f, r_do = 0.f;
                float r_c0 = 0.f, r_c1 = 0.f, r_c2 = 0.f, r_dd = 0.f;
                if (hit) {
                    const float4 p2 = s_p2[j];
                    Tr = Tr / (1.f - alpha);
                    const float w = alpha * Tr;
                    // colour accumulated behind this entry (recurrence, back to front)
                    acc0 = last_alpha * lc0 + (1.f - last_alpha) * acc0;
                    acc1 = last_alpha * lc1 + (1.f - last_alpha) * acc1;
                    acc2 = last_alpha * lc2 + (1.f - last_alpha) * acc2;
                    accD = last_alpha * lD + (1.f - last_alpha) * accD;
                    lc0 = p2.x; lc1 = p2.y; lc2 = p2.z; lD = p2.w;
                    float dL_dalpha_ = (p2.x - acc0) * gI0 + (p2.y - acc1) * gI1 + (p2.z - acc2) * gI2 +
                                       (p2.w - accD) * gD;
                    dL_dalpha_ *= Tr;
                    last_alpha = alpha;
                    // image = C + T_final * bg and alpha_out = 1 - T_final both see alpha only through
                    // T_final: dT_final/dalpha = -T_final / (1 - alpha)
                    dL_dalpha_ += (T_final / (1.f - alpha)) * tail;
                    r_c0 = w * gI0; r_c1 = w * gI1; r_c2 = w * gI2; r_dd = w * gD;
                    // [3DGS-grad] the min(0.99, .) clamp is straight-through
                    const float dL_dG = p1.y * dL_dalpha_;
                    const float s = dL_dG * Gv;
                    r_do = Gv * dL_dalpha_;
                    r_dx = -s * (p0.z * dx + p0.w * dy);
                    r_dy = -s * (p1.x * dy + p0.w * dx);
                    r_dA = -0.5f * s * dx * dx;
                    r_dB = -s * dx * dy;
                    r_dC = -0.5f * s * dy * dy;
                }
                r_dx = wave_sum_to63(r_dx); r_dy = wave_sum_to63(r_dy);
                r_dA = wave_sum_to63(r_dA); r_dB = wave_sum_to63(r_dB); r_dC = wave_sum_to63(r_dC);
                r_do = wave_sum_to63(r_do);
                r_c0 = wave_sum_to63(r_c0); r_c1 = wave_sum_to63(r_c1); r_c2 = wave_sum_to63(r_c2);
                r_dd = wave_sum_to63(r_dd);
                // gather the ten totals (lane 63) into lanes 0..9 and issue one atomic instruction
                float val = 0.f;
#define SPF_PICK(k, x) { const float t_ = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), 63)); if (c.lane == k) val = t_; }
                SPF_PICK(0, r_dx) SPF_PICK(1, r_dy) SPF_PICK(2, r_dA) SPF_PICK(3, r_dB) SPF_PICK(4, r_dC)
                SPF_PICK(5, r_do) SPF_PICK(6, r_c0) SPF_PICK(7, r_c1) SPF_PICK(8, r_c2) SPF_PICK(9, r_dd)
#undef SPF_PICK
                if (c.lane < 10) {
                    const uint32_t gid = s_gid[j];
                    atomicAdd(grec_r + (size_t)gid * kRec + c.lane, val);
                }
            }
        }
        __syncthreads();
    }
}

// ---- launchers ------------------------------------------------------------------------------------
hipError_t launch_render_fwd(const SpfDims& d, const SpfInputs& in, const SpfState& st, const SpfOutputs& out,
                             uint64_t capacity, int T, int tiles_x, hipStream_t stream) {
    const int RT = d.S * d.V * T;
    const int grid = (RT + 7) / 8 * 8;
    spf_render_fwd_kernel<<<grid, kBlock, 0, stream>>>(st.rec, st.pairs, st.tile_start, st.counters, capacity, in.bg,
                                                       out.image, out.depth, out.alpha, st.final_T, st.n_contrib, d.G,
                                                       d.H, d.W, T, tiles_x, RT);
    return hipGetLastError();
}

hipError_t launch_render_bwd(const SpfDims& d, const SpfInputs& in, const SpfState& st, const SpfGrads& g, int T,
                             int tiles_x, hipStream_t stream) {
    const int RT = d.S * d.V * T;
    const int grid = (RT + 7) / 8 * 8;
    spf_render_bwd_kernel<<<grid, kBlock, 0, stream>>>(st.rec, st.pairs, st.tile_start, in.bg, st.final_T,
                                                       st.n_contrib, g.dL_dimage, g.dL_ddepth, g.dL_dalpha, g.grec,
                                                       d.G, d.H, d.W, T, tiles_x, RT);
    return hipGetLastError();
}

}  // namespace spf
