// 8x8-tile alpha compositing, forward and backward, ONE WAVE PER TILE, for gfx950 (wave64).
//
// The 16x16 "lists" kernels (render.hip) put four waves on a tile and pay for it with block barriers: four per round
// of the backward, 2.4 rounds per tile -- 60 % of its wave-cycles are spent waiting (SQ_WAIT_ANY / SQ_WAVE_CYCLES), and
// a launch of few tiles with long lists (BASELINE config 3, the reference's 10-view shape) runs as long as its longest
// block.  Here a tile is 8x8 pixels = 64 lanes = one wave, a workgroup is one wave, and nothing in the kernel waits for
// another wave: every exchange between the "entry owner" role of a lane (lane i <-> list entry i of the round) and its
// "pixel owner" role (lane p <-> pixel p of the tile) is wave-synchronous.
//
//   phase A  lane i computes the 64-bit FOOTPRINT MASK of entry i inside the tile (bit p: pixel p's centre is inside the
//            entry's conservative cull disc; row by row from one hardware square root per row) and the wave TRANSPOSES
//            the 64 x 64 bit matrix in registers (five DPP / two permute steps): lane p then holds the candidate word of
//            pixel p -- what the 16x16 kernels build with ~9 LDS atomics per entry;
//   phase B  lane p walks the set bits of its word in list order (forward) or back to front (backward), gathering each
//            candidate's staged record from the wave's own LDS, exactly as the 16x16 kernels do;
//   phase C  (backward) lane i sums the (w, u) slots of its entry: slot of (entry, pixel p) = the entry's pool offset
//            + popcount(mask below bit p) -- the pool holds exactly the candidates, no box padding -- and writes the
//            pair's packed gradient record.
// The tile grid is 8 px everywhere in this build (SPF_TILE = 8: projection, direct bins, sort), so a (Gaussian, tile)
// pair is a (Gaussian, 8x8 quadrant) pair: ~1.3x the pairs of the 16 px grid, in lists a quarter as long.
//
// Semantics: SURVEY.md Appendix B #10/#11 (restated in oracle/splat_ref.py::composite); same arithmetic, same hit
// decisions as render.hip's lists kernels (the exponent chain is shared).
#include "spf_common.h"

namespace spf {

constexpr int kWT = 8;                 // tile edge of these kernels
constexpr float kWHalfLog2e = -0.72134752044448170368f;   // -0.5 * log2(e)
constexpr float kWLog2e = -1.44269504088896340736f;       // -log2(e)

// value of lane (l ^ MASK): DPP inside a row of 16 lanes, ds_bpermute across rows
template <int CTRL>
__device__ __forceinline__ uint32_t wdpp_mov(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, 0xf, 0xf, true);
}
template <int MASK>
__device__ __forceinline__ uint32_t wlane_xor(uint32_t v) {
    if constexpr (MASK == 1) return wdpp_mov<0xB1>(v);                          // quad_perm [1,0,3,2]
    else if constexpr (MASK == 2) return wdpp_mov<0x4E>(v);                     // quad_perm [2,3,0,1]
    else if constexpr (MASK == 4) return wdpp_mov<0x1B>(wdpp_mov<0x141>(v));    // (l ^ 7) ^ 3
    else if constexpr (MASK == 8) return wdpp_mov<0x128>(v);                    // row_ror:8
    else return (uint32_t)__shfl_xor((int)v, MASK, kWave);
}

// One step of the bit-matrix transpose on one 32-bit half: rows (lanes) and columns (bits) are split by bit S of their
// index; the off-diagonal S x S blocks change places.  `up` = this lane's row index has bit S set.
template <int S>
__device__ __forceinline__ uint32_t transpose_step(uint32_t h, bool up) {
    constexpr uint32_t m = S == 16 ? 0x0000ffffu : S == 8 ? 0x00ff00ffu : S == 4 ? 0x0f0f0f0fu : S == 2 ? 0x33333333u : 0x55555555u;
    const uint32_t ph = wlane_xor<S>(h);
    // up:  keep my columns with bit S set, take the partner's columns with bit S set moved down;  else the mirror image
    return up ? (((ph >> S) & m) | (h & ~m)) : ((h & m) | ((ph & m) << S));
}
// 64 x 64 bit-matrix transpose across the wave.  In: lane i holds row i (bit p = column p).  Out: lane p holds column p
// (bit i = row i).
__device__ __forceinline__ uint64_t wave_transpose64(uint64_t x, int lane) {
    uint32_t lo = (uint32_t)x, hi = (uint32_t)(x >> 32);
    {   // step 32: rows 0..31 keep their low word and take the low word of row + 32 as their high word; rows 32..63 the mirror
        const uint32_t plo = wlane_xor<32>(lo), phi = wlane_xor<32>(hi);
        const bool up = (lane & 32) != 0;
        const uint32_t nlo = up ? phi : lo, nhi = up ? hi : plo;
        lo = nlo; hi = nhi;
    }
    lo = transpose_step<16>(lo, (lane & 16) != 0); hi = transpose_step<16>(hi, (lane & 16) != 0);
    lo = transpose_step<8>(lo, (lane & 8) != 0);   hi = transpose_step<8>(hi, (lane & 8) != 0);
    lo = transpose_step<4>(lo, (lane & 4) != 0);   hi = transpose_step<4>(hi, (lane & 4) != 0);
    lo = transpose_step<2>(lo, (lane & 2) != 0);   hi = transpose_step<2>(hi, (lane & 2) != 0);
    lo = transpose_step<1>(lo, (lane & 1) != 0);   hi = transpose_step<1>(hi, (lane & 1) != 0);
    return ((uint64_t)hi << 32) | lo;
}

// Footprint of a Gaussian (centre (gx, gy), squared cull radius r2) inside the 8x8 tile at (X0, Y0): bit 8*y + x is set
// when pixel (X0 + x, Y0 + y) may lie inside the cull disc -- a superset of the pixels with dx^2 + dy^2 <= r2 (the row
// extents come from the hardware square root, padded like disc_box_fast), and the disc itself carries 0.2 % of slack over
// every pixel that can reach alpha = 1/255 (project.hip).  A candidate that is no contributor is skipped by phase B's
// exact test, so results do not depend on the padding.
__device__ __forceinline__ uint64_t footprint_mask8(float gx, float gy, float r2, int X0, int Y0) {
    uint32_t lo = 0u, hi = 0u;
    const float fx0 = (float)X0 - gx;             // offset of column 0
    float dy = (float)Y0 - gy;
#pragma unroll
    for (int y = 0; y < kWT; ++y) {
        const float rem = fmaf(-dy, dy, r2);      // what the row leaves for dx^2
        const float h = __builtin_amdgcn_sqrtf(fmaxf(rem, 0.f)) * 1.0001f + 1e-3f;
        const float xl = fmaxf(ceilf(-h - fx0), 0.f), xh = fminf(floorf(h - fx0), (float)(kWT - 1));
        uint32_t row = 0u;
        if (rem >= 0.f && xh >= xl) row = ((2u << (int)xh) - 1u) & ~((1u << (int)xl) - 1u);
        if (y < 4) lo |= row << (8 * y);
        else hi |= row << (8 * (y - 4));
        dy += 1.f;
    }
    return ((uint64_t)hi << 32) | lo;
}

typedef float w2f __attribute__((ext_vector_type(2)));
// the exponent of the lists kernels (render.hip::lists_power2 / lists_power2_scalar): the same expression trees, so that
// all four kernels take the same hit decisions
__device__ __forceinline__ float wave_power2(const float4& p0, float Bs, w2f fxy, w2f& dxy) {
    dxy = w2f{p0.x, p0.y} - fxy;
    const w2f u = w2f{p0.z, p0.w} * dxy;
    return fmaf(dxy.x, fmaf(Bs, dxy.y, u.x), u.y * dxy.y);
}
__device__ __forceinline__ float wave_power2_scalar(const float4& p0, float Bs, float fx, float fy) {
    const float dx = p0.x - fx, dy = p0.y - fy;
    const float ux = p0.z * dx, uy = p0.w * dy;
    return fmaf(dx, fmaf(Bs, dy, ux), uy * dy);
}

__device__ __forceinline__ void wave_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// The tile of this workgroup (one wave): image order, or the launch order (spf_common.h::lists_tile without the dense
// split -- these kernels take every tile).
__device__ __forceinline__ bool wave_tile(const TileLists& tl, int RT, int& vid, uint32_t& beg, uint32_t& n) {
    const int slot = xcd_remap(blockIdx.x, gridDim.x);
    if (tl.order) {
        const uint2 o = tl.order[slot];
        vid = (int)(o.x & kOrderTileMask);
        n = o.y;
        beg = (uint32_t)vid * tl.cap;
        return vid < RT;
    }
    vid = slot;
    if (vid >= RT) return false;
    tile_range(tl, (size_t)vid, beg, n);
    return true;
}

// ------------------------------------------------------------------------------------------------
// Forward
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kWave) void spf_render_fwd_wave_kernel(
    const float* __restrict__ rec, const uint64_t* __restrict__ pairs, TileLists tl,
    const uint32_t* __restrict__ counters, const float* __restrict__ bg_all, float* __restrict__ image,
    float* __restrict__ depth_out, float* __restrict__ alpha_out, float* __restrict__ final_T,
    uint32_t* __restrict__ n_contrib, int G, int H, int W, int T, int tiles_x, int RT) {
    // staged records of the round's 64 entries; record -1 of every array exists and is finite (see `next`)
    __shared__ float4 s_p0z[kWave + 1];    // x, y | A', C'
    __shared__ float2 s_p1z[kWave + 1];    // B', opacity
    __shared__ float4 s_p2z[kWave + 1];    // r, g | b, depth
    float4* const s_p0 = s_p0z + 1;
    float2* const s_p1 = s_p1z + 1;
    float4* const s_p2 = s_p2z + 1;

    const int lane = threadIdx.x;
    int vid;
    uint32_t beg, n;
    if (!wave_tile(tl, RT, vid, beg, n)) return;
    const int r = vid / T, tile = vid - r * T;
    const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
    const int X0 = tx * kWT, Y0 = ty * kWT;
    const int px = X0 + (lane & 7), py = Y0 + (lane >> 3);
    const bool inside = px < W && py < H;
    const size_t P = (size_t)H * W, pix = (size_t)py * W + px;
    if (counters[2] != 0u) {                     // failed plan: NaN, never uninitialised memory (render.hip::poison_tile)
        if (inside) {
            const float nan = __builtin_nanf("");
            float* __restrict__ img = image + (size_t)r * 3 * P;
            img[pix] = nan; img[P + pix] = nan; img[2 * P + pix] = nan;
            depth_out[(size_t)r * P + pix] = nan;
            alpha_out[(size_t)r * P + pix] = nan;
        }
        return;
    }
    if (lane == 0) { s_p0z[0] = make_float4(0.f, 0.f, 0.f, 0.f); s_p1z[0] = make_float2(0.f, 0.f); s_p2z[0] = make_float4(0.f, 0.f, 0.f, 0.f); }
    const float* __restrict__ rec_r = rec + (size_t)r * G * kRec;
    float fx = (float)px, fy = (float)py;
    asm volatile("" : "+v"(fx), "+v"(fy));
    const w2f fxy = {fx, fy};

    float Tr = 1.0f;
    w2f c01 = {0.f, 0.f}, c2d = {0.f, 0.f};      // (r, g), (b, depth) accumulators
    uint32_t last16 = 0;                         // 16 * (list position + 1) of the last contributor
    uint64_t dm = lane_ballot(!inside);          // lanes whose pixel is finished, as a wave mask (render.hip)

    // the first round's entries; later rounds are requested one round ahead (the chain list entry -> record is two
    // dependent global reads, and nothing but this wave's own arithmetic can hide them)
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = make_float4(0.f, 0.f, 0.f, -1.f), cc = a;
    if ((uint32_t)lane < n) {
        const uint32_t gid = (uint32_t)pairs[beg + lane];
        const float4* __restrict__ rp = reinterpret_cast<const float4*>(rec_r + (size_t)gid * kRec);
        a = rp[0]; b = rp[1]; cc = rp[2];
    }
    for (uint32_t base = 0; base < n; base += kWave) {
        const uint64_t M = footprint_mask8(a.x, a.y, b.w, X0, Y0);      // (no entry: r2 = -1 -> empty)
        wave_fence();                                                  // the previous round's gathers are done
        s_p0[lane] = make_float4(a.x, a.y, kWHalfLog2e * a.z, kWHalfLog2e * b.x);
        s_p1[lane] = make_float2(kWLog2e * a.w, b.y);
        s_p2[lane] = make_float4(cc.x, cc.y, cc.z, b.z);
        {   // next round's entries
            const uint32_t idx = base + kWave + lane;
            a = make_float4(0.f, 0.f, 0.f, 0.f); b = make_float4(0.f, 0.f, 0.f, -1.f); cc = a;
            if (idx < n) {
                const uint32_t gid = (uint32_t)pairs[beg + idx];
                const float4* __restrict__ rp = reinterpret_cast<const float4*>(rec_r + (size_t)gid * kRec);
                a = rp[0]; b = rp[1]; cc = rp[2];
            }
        }
        const uint64_t C = wave_transpose64(M, lane);                   // candidates of MY pixel, bit j = entry base + j
        wave_fence();
        uint32_t m = (uint32_t)C, m2 = (uint32_t)(C >> 32);
        int wb = 0;                                                     // byte offset of entry 32 w in the staged arrays
        // next candidate of this lane as the byte offset of its staged record; has = false: none left -- the offset is then
        // 16 bytes below a word's first record (v_ffbl_b32 of 0 is -1), where a finite record sits (entry 31, or the zero
        // record in front of entry 0)
        auto next = [&](bool& has) -> int {
            if (m == 0u && m2 != 0u) { m = m2; m2 = 0u; wb = 512; }
            int bit;
            asm("v_ffbl_b32 %0, %1" : "=v"(bit) : "v"(m));
            has = bit >= 0;
            m &= m - 1u;
            return wb + (bit << 4);
        };
        const uint32_t base16 = (base + 1u) << 4;
        auto composite = [&](uint64_t hm, int j, const float4& p0, const float2& p1, const float4& p2) {
            w2f dxy;
            const float pw = wave_power2(p0, p1.x, fxy, dxy);
            const float alpha = fminf(kAlphaMax, p1.y * __builtin_amdgcn_exp2f(pw));
            const float test_T = Tr * (1.f - alpha);
            const uint64_t hitm = hm & ~dm & lane_ballot(pw <= 0.f) & lane_ballot(alpha >= kAlphaMin);
            const uint64_t stopm = hitm & lane_ballot(test_T < kTMin);
            dm |= stopm;
            const bool take = __builtin_amdgcn_inverse_ballot_w64(hitm & ~stopm);
            const float wgt = take ? alpha * Tr : 0.f;
            const w2f ww = {wgt, wgt};
            c01 = __builtin_elementwise_fma(w2f{p2.x, p2.y}, ww, c01);
            c2d = __builtin_elementwise_fma(w2f{p2.z, p2.w}, ww, c2d);
            Tr = take ? test_T : Tr;
            last16 = take ? (uint32_t)j + base16 : last16;
        };
        auto rec0 = [&](int j) { return *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(s_p0) + j); };
        auto rec1 = [&](int j) { return *reinterpret_cast<const float2*>(reinterpret_cast<const char*>(s_p1) + (j >> 1)); };
        auto rec2 = [&](int j) { return *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(s_p2) + j); };
        // software pipeline, unrolled by two (render.hip)
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
        if (dm == ~0ull) break;
    }
    if (inside) {
        const float* __restrict__ bg = bg_all + 3 * r;
        float* __restrict__ img = image + (size_t)r * 3 * P;
        img[pix] = c01.x + Tr * bg[0];
        img[P + pix] = c01.y + Tr * bg[1];
        img[2 * P + pix] = c2d.x + Tr * bg[2];
        depth_out[(size_t)r * P + pix] = c2d.y;
        alpha_out[(size_t)r * P + pix] = 1.0f - Tr;
        final_T[(size_t)r * P + pix] = Tr;
        n_contrib[(size_t)r * P + pix] = last16 >> 4;
    }
}

// ------------------------------------------------------------------------------------------------
// Backward
// ------------------------------------------------------------------------------------------------
#ifndef SPF_WPOOL
#define SPF_WPOOL 448
#endif
constexpr int kWPool = SPF_WPOOL;      // (w, u) slots per round: 3.5 KB; LDS per wave ~7.7 KB -> 20 waves per CU

template <bool DEPTH_GRAD>
__global__ __launch_bounds__(kWave) void spf_render_bwd_wave_kernel(
    const float* __restrict__ rec, const uint64_t* __restrict__ pairs, TileLists tl, const float* __restrict__ bg_all,
    const float* __restrict__ final_T, const uint32_t* __restrict__ n_contrib, const float* __restrict__ dL_dimage,
    const float* __restrict__ dL_ddepth, const float* __restrict__ dL_dalpha, const uint2* __restrict__ pinfo,
    float* __restrict__ gpair, int G, int H, int W, int T, int tiles_x, int RT, const uint32_t* __restrict__ counters) {
    if (counters[2] != 0u) return;           // failed plan: nothing was rendered; the projection backward poisons the gradients
    __shared__ float4 s_p0z[kWave + 1];     // x, y | A', C'
    __shared__ float4 s_p1z[kWave + 1];     // B', opacity | footprint mask (two words, int bits)
    __shared__ float4 s_p2z[kWave + 1];     // r, g, b | byte offset of the entry's first pool slot (int bits)
    __shared__ float s_pd[DEPTH_GRAD ? kWave + 1 : 1];   // depth (only with an upstream depth gradient)
    __shared__ float2 s_pool[kWPool];       // (w, u) of this round's (entry, pixel) candidates, entry-major
    __shared__ float4 s_gI[kWave];          // per pixel: dL/dC (rgb), dL/ddepth (or 1)
    float4* const s_p0 = s_p0z + 1;
    float4* const s_p1 = s_p1z + 1;
    float4* const s_p2 = s_p2z + 1;

    const int lane = threadIdx.x;
    int vid;
    uint32_t beg, n;
    if (!wave_tile(tl, RT, vid, beg, n)) return;
    if (n == 0) return;
    const int r = vid / T, tile = vid - r * T;
    const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
    const int X0 = tx * kWT, Y0 = ty * kWT;
    const int lx = lane & 7, ly = lane >> 3;
    const int px = X0 + lx, py = Y0 + ly;
    const size_t P = (size_t)H * W;
    float T_final = 1.f, gI0 = 0.f, gI1 = 0.f, gI2 = 0.f, gD = 0.f, gA = 0.f;
    uint32_t ncon = 0;
    if (px < W && py < H) {
        const size_t qpix = (size_t)py * W + px;
        ncon = n_contrib[(size_t)r * P + qpix];
        T_final = final_T[(size_t)r * P + qpix];
        if (dL_dimage) {
            const float* __restrict__ gi = dL_dimage + (size_t)r * 3 * P;
            gI0 = gi[qpix]; gI1 = gi[P + qpix]; gI2 = gi[2 * P + qpix];
        }
        if (DEPTH_GRAD) gD = dL_ddepth[(size_t)r * P + qpix];
        if (dL_dalpha) gA = dL_dalpha[(size_t)r * P + qpix];
    }
    s_gI[lane] = make_float4(gI0, gI1, gI2, DEPTH_GRAD ? gD : 1.f);   // phase C's table; .w == 1 folds sum(u) into a packed fma
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
    const uint32_t wmax = wave_max_u32(ncon);
    const uint32_t bmax = min(n, wmax);          // (<= n by construction; the clamp only matters after a failed plan)
    // entries behind every pixel's last contributor: zero record (each pair slot is written exactly once, no memset)
    for (uint32_t idx = bmax + lane; idx < n; idx += kWave)
        store_grec<DEPTH_GRAD>(gpair, pair_slot((uint32_t)pairs[beg + idx]), 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f);
    if (bmax == 0) return;
    // lanes' own pixel as bit masks: the candidates of pixel p BELOW it in an entry's footprint number its pool slot
    const uint32_t below_lo = lane < 32 ? (1u << lane) - 1u : ~0u, below_hi = lane < 32 ? 0u : (1u << (lane - 32)) - 1u;

    float Tr = T_final;
    float sB = -tail * T_final;          // running "behind" scalar of the replay (render.hip, phase B)
    uint32_t hi = bmax;                  // entries [0, hi) are still to be replayed
    while (hi > 0) {
        // ---- this round: lane i <-> entry hi-1-i; footprint, slot demand, prefix sum, acceptance (a prefix of the lanes) ----
        const bool have = (uint32_t)lane < hi;
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = make_float4(0.f, 0.f, 0.f, -1.f), cc = a;
        uint32_t gid = 0, pslot = 0;
        if (have) {
            gid = (uint32_t)pairs[beg + (hi - 1u - (uint32_t)lane)];
            const float4* __restrict__ rp = reinterpret_cast<const float4*>(rec_r + (size_t)gid * kRec);
            a = rp[0]; b = rp[1]; cc = rp[2];
            pslot = pair_slot(gid);
        }
        uint64_t M = footprint_mask8(a.x, a.y, b.w, X0, Y0);
        const uint32_t size = (uint32_t)__popcll(M);
        const uint32_t inc = wave_iscan_u32(size);
        const uint32_t off = inc - size;
        const bool acc = have && inc <= (uint32_t)kWPool;              // (one entry needs <= 64 slots: lane 0 is always in)
        const int cnt = __popcll(lane_ballot(acc));
        if (!acc) M = 0ull;
        const uint32_t used = (uint32_t)__builtin_amdgcn_readlane((int)inc, cnt - 1);
        wave_fence();                                                  // the previous round is done with the LDS
        s_p0[lane] = make_float4(a.x, a.y, kWHalfLog2e * a.z, kWHalfLog2e * b.x);
        s_p1[lane] = make_float4(kWLog2e * a.w, b.y, __uint_as_float((uint32_t)M), __uint_as_float((uint32_t)(M >> 32)));
        s_p2[lane] = make_float4(cc.x, cc.y, cc.z, __int_as_float(8 * (int)off));
        if (DEPTH_GRAD) s_pd[1 + lane] = b.z;
        for (uint32_t k = lane; k < used; k += kWave) s_pool[k] = make_float2(0.f, 0.f);
        const uint64_t C = wave_transpose64(M, lane);                   // candidates of MY pixel: bit j = entry hi-1-j
        wave_fence();
        // ---- phase B: ascending bits = descending list position ----
        if (hi - (uint32_t)cnt < wmax) {
            // contributors of this pixel are entries < ncon, i.e. lane indices >= hi - ncon
            const uint32_t jmin = ncon < hi ? hi - ncon : 0u;
            const uint64_t Cm = jmin < 64u ? (C & (~0ull << jmin)) : 0ull;
            uint32_t m = (uint32_t)Cm, m2 = (uint32_t)(Cm >> 32);
            int wb = 0;
            auto next = [&](bool& has) -> int {
                if (m == 0u && m2 != 0u) { m = m2; m2 = 0u; wb = 512; }
                int bit;
                asm("v_ffbl_b32 %0, %1" : "=v"(bit) : "v"(m));
                has = bit >= 0;
                m &= m - 1u;
                return wb + (bit << 4);
            };
            auto replay = [&](bool has, int j, const float4& p0, const float4& p1, const float4& p2) {
                const float pw = wave_power2_scalar(p0, p1.x, fx, fy);
                const float Gv = __builtin_amdgcn_exp2f(pw);
                const float alpha = fminf(kAlphaMax, p1.y * Gv);
                if (has && pw <= 0.f && alpha >= kAlphaMin) {
                    // (render.hip, phase B: one running scalar; T as fma(T, alpha / (1 - alpha), T))
                    const float inv1ma = __builtin_amdgcn_rcpf(1.f - alpha);
                    Tr = fmaf(Tr, alpha * inv1ma, Tr);
                    float cg = fmaf(p2.x, gI0, fmaf(p2.y, gI1, p2.z * gI2));
                    if (DEPTH_GRAD) cg = fmaf(*reinterpret_cast<const float*>(reinterpret_cast<const char*>(s_pd + 1) + (j >> 2)), gD, cg);
                    const float dL_dalpha_ = fmaf(cg, Tr, -(sB * inv1ma));
                    const float wgt = alpha * Tr;
                    sB = fmaf(cg, wgt, sB);
                    // slot = the entry's first slot + the number of its footprint pixels below mine
                    const int k8 = __float_as_int(p2.w) + 8 * (__popc(__float_as_uint(p1.z) & below_lo) + __popc(__float_as_uint(p1.w) & below_hi));
                    *reinterpret_cast<float2*>(reinterpret_cast<char*>(s_pool) + k8) = make_float2(wgt, Gv * dL_dalpha_);
                }
            };
            auto rec0 = [&](int j) { return *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(s_p0) + j); };
            auto rec1 = [&](int j) { return *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(s_p1) + j); };
            auto rec2 = [&](int j) { return *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(s_p2) + j); };
            bool ha, hb;
            int ja = next(ha), jb;
            uint64_t ma = lane_ballot(ha), mb;
            float4 a0 = rec0(ja), a1 = rec1(ja), a2 = rec2(ja), b0, b1, b2;
            while (true) {
                if (!ma) break;
                jb = next(hb);
                mb = lane_ballot(hb);
                b0 = rec0(jb); b1 = rec1(jb); b2 = rec2(jb);
                replay(ha, ja, a0, a1, a2);
                if (!mb) break;
                ja = next(ha);
                ma = lane_ballot(ha);
                a0 = rec0(ja); a1 = rec1(ja); a2 = rec2(ja);
                replay(hb, jb, b0, b1, b2);
            }
        }
        wave_fence();
        // ---- phase C: the entry owner sums its slots (render.hip, phase C: moments about a local origin) ----
        if (acc) {
            w2f c01 = {0.f, 0.f}, c2s = {0.f, 0.f};      // (dL/dr, dL/dg), (dL/db, sum u)
            w2f s1 = {0.f, 0.f}, s2 = {0.f, 0.f};        // (sum u dx, sum u dy), (sum u dx^2, sum u dy^2)
            float sxy = 0.f, cd = 0.f;
            const float2* __restrict__ hp = s_pool + off;
            const float ox = fminf(fmaxf(rintf(a.x), (float)X0), (float)(X0 + kWT - 1));
            const float oy = fminf(fmaxf(rintf(a.y), (float)Y0), (float)(Y0 + kWT - 1));
            const float ex = ox - (float)X0, ey = oy - (float)Y0;    // origin inside the tile: offsets are small integers
            uint32_t mw = (uint32_t)M, mw2 = (uint32_t)(M >> 32);
            int pb = 0;
            while (true) {
                if (mw == 0u) {
                    if (mw2 == 0u) break;
                    mw = mw2; mw2 = 0u; pb = 32;
                }
                const int p = pb + __builtin_ctz(mw);
                mw &= mw - 1u;
                const float2 h = *hp++;                  // (w, u) of pixel p
                const float4 gi = s_gI[p];               // (dL/dC, dL/ddepth or 1)
                const w2f dc = {ex - (float)(p & 7), ey - (float)(p >> 3)};
                const w2f hw = {h.x, h.x}, hu = {h.y, h.y}, hwu = {h.x, h.y};
                c01 = __builtin_elementwise_fma(hw, w2f{gi.x, gi.y}, c01);
                if (DEPTH_GRAD) {
                    c2s = __builtin_elementwise_fma(hwu, w2f{gi.z, 1.f}, c2s);
                    cd = fmaf(h.x, gi.w, cd);
                } else {
                    c2s = __builtin_elementwise_fma(hwu, w2f{gi.z, gi.w}, c2s);   // gi.w == 1
                }
                const w2f t = hu * dc;
                s1 += t;
                s2 = __builtin_elementwise_fma(t, dc, s2);
                sxy = fmaf(t.x, dc.y, sxy);
            }
            const float o = b.y;   // [3DGS-grad] dL/dG = opacity * dL/dalpha (the 0.99 clamp is straight-through)
            // (render.hip, phase C: shift of the moments from the local origin to the centre)
            const float M0 = c2s.y;
            const w2f P1 = {a.z, a.w}, P2 = {a.w, b.x};                     // conic rows (A, B), (B, C)
            const float d0x = a.x - ox, d0y = a.y - oy;
            const w2f v0 = __builtin_elementwise_fma(P2, w2f{d0y, d0y}, P1 * w2f{d0x, d0x});
            const w2f L = __builtin_elementwise_fma(P2, w2f{s1.y, s1.y}, P1 * w2f{s1.x, s1.x});      // Q E1
            const w2f S1 = __builtin_elementwise_fma(v0, w2f{M0, M0}, L);
            const w2f R0 = __builtin_elementwise_fma(P2, w2f{sxy, sxy}, P1 * w2f{s2.x, s2.x});       // rows of E2 Q
            const w2f R1 = __builtin_elementwise_fma(P2, w2f{s2.y, s2.y}, P1 * w2f{sxy, sxy});
            const w2f Qx = __builtin_elementwise_fma(R1, w2f{a.w, a.w}, R0 * w2f{a.z, a.z});         // (QEQ_xx, QEQ_xy)
            const float Qyy = fmaf(b.x, R1.y, a.w * R0.y);
            const w2f Tt = __builtin_elementwise_fma(v0, w2f{M0, M0}, L + L);
            const float Sxx = fmaf(v0.x, Tt.x, Qx.x), Syy = fmaf(v0.y, Tt.y, Qyy);
            const float Sxy = fmaf(v0.x, fmaf(v0.y, M0, L.y), fmaf(v0.y, L.x, Qx.y));
            store_grec<DEPTH_GRAD>(gpair, pslot, -o * S1.x, -o * S1.y, 0.5f * o * Sxx, o * Sxy, 0.5f * o * Syy,
                                   M0, c01.x, c01.y, c2s.x, cd);
        }
        hi -= (uint32_t)cnt;
    }
}

// ---- launchers (called from render.hip when the tile grid is 8 px) ---------------------------------------------
hipError_t launch_render_fwd_wave(const SpfDims& d, const SpfInputs& in, const SpfState& st, const SpfOutputs& out,
                                  int T, int tiles_x, const TileLists& tl, hipStream_t stream) {
    const int RT = d.S * d.V * T;
    const int grid = (RT + 7) / 8 * 8;
    spf_render_fwd_wave_kernel<<<grid, kWave, 0, stream>>>(st.rec, st.pairs, tl, st.counters, in.bg, out.image, out.depth,
                                                           out.alpha, st.final_T, st.n_contrib, d.G, d.H, d.W, T,
                                                           tiles_x, RT);
    return hipGetLastError();
}

hipError_t launch_render_bwd_wave(const SpfDims& d, const SpfInputs& in, const SpfState& st, const SpfGrads& g, int T,
                                  int tiles_x, const TileLists& tl, hipStream_t stream) {
    const int RT = d.S * d.V * T;
    const int grid = (RT + 7) / 8 * 8;
    const uint2* const pinfo = reinterpret_cast<const uint2*>(st.pair_off);
    if (g.dL_ddepth)
        spf_render_bwd_wave_kernel<true><<<grid, kWave, 0, stream>>>(
            st.rec, st.pairs, tl, in.bg, st.final_T, st.n_contrib, g.dL_dimage, g.dL_ddepth, g.dL_dalpha, pinfo, g.gpair,
            d.G, d.H, d.W, T, tiles_x, RT, st.counters);
    else
        spf_render_bwd_wave_kernel<false><<<grid, kWave, 0, stream>>>(
            st.rec, st.pairs, tl, in.bg, st.final_T, st.n_contrib, g.dL_dimage, g.dL_ddepth, g.dL_dalpha, pinfo, g.gpair,
            d.G, d.H, d.W, T, tiles_x, RT, st.counters);
    return hipGetLastError();
}

}  // namespace spf
