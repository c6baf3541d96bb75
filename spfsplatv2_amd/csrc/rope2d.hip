// In-place 2-D rotary position embedding for gfx950.
//
// Replaces rope_2d_cuda_kernel (/root/reference/src/model/encoder/backbone/croco/curope/kernels.cu:17-82).
// tokens[B,N,H,D] (stride(H)=D, stride(D)=1, arbitrary outer strides); one head row is laid out as
//   [ u_Y (Q) | v_Y (Q) | u_X (Q) | v_X (Q) ],  Q = D/4,
// and every (u,v) pair with frequency index q is rotated by  angle = pos * fwd / base^(q/Q)
// where pos is the token's y position for the first half and its x position for the second.
//
// Mapping: one lane owns 16 bytes of consecutive frequencies (4 floats, 8 halves) of one (token, half) and walks two or
// four heads with them (the angle does not depend on the head: one evaluation per group of head rows); consecutive lanes walk q,
// then the half, so a group of lanes covers whole contiguous head rows.  The Q inverse frequencies are computed
// on the host with libm powf (exactly what the reference's CPU path evaluates,
// curope/curope.cpp:35) and travel in the kernel-argument segment -- no device powf, no table in HBM.
#include <hip/hip_bf16.h>
#include <hip/hip_fp16.h>

#include "spf_common.h"

namespace spf {

struct RopeFreq {
    float inv[64];  // fwd / base^(q/Q), q < Q <= 64
};

template <typename T>
__device__ __forceinline__ float to_f(T v);
template <> __device__ __forceinline__ float to_f<float>(float v) { return v; }
template <> __device__ __forceinline__ float to_f<__half>(__half v) { return __half2float(v); }
template <> __device__ __forceinline__ float to_f<__hip_bfloat16>(__hip_bfloat16 v) { return __bfloat162float(v); }
template <typename T>
__device__ __forceinline__ T from_f(float v);
template <> __device__ __forceinline__ float from_f<float>(float v) { return v; }
template <> __device__ __forceinline__ __half from_f<__half>(float v) { return __float2half(v); }
template <> __device__ __forceinline__ __hip_bfloat16 from_f<__hip_bfloat16>(float v) { return __float2bfloat16(v); }

// 16 bytes per lane whatever the type: 4 floats, 8 halves / bfloat16s.
template <typename T> struct Wide;
template <> struct Wide<float> {
    static constexpr int N = 4;
    static __device__ __forceinline__ void load(const float* p, float* f) {
        const float4 v = *reinterpret_cast<const float4*>(p);
        f[0] = v.x; f[1] = v.y; f[2] = v.z; f[3] = v.w;
    }
    static __device__ __forceinline__ void store(float* p, const float* f) {
        *reinterpret_cast<float4*>(p) = make_float4(f[0], f[1], f[2], f[3]);
    }
};
template <typename H> struct WideHalf {
    static constexpr int N = 8;
    struct __attribute__((aligned(16))) Pack { H h[8]; };
    static __device__ __forceinline__ void load(const H* p, float* f) {
        const Pack v = *reinterpret_cast<const Pack*>(p);
#pragma unroll
        for (int i = 0; i < 8; ++i) f[i] = to_f<H>(v.h[i]);
    }
    static __device__ __forceinline__ void store(H* p, const float* f) {
        Pack v;
#pragma unroll
        for (int i = 0; i < 8; ++i) v.h[i] = from_f<H>(f[i]);
        *reinterpret_cast<Pack*>(p) = v;
    }
};
template <> struct Wide<__half> : WideHalf<__half> {};
template <> struct Wide<__hip_bfloat16> : WideHalf<__hip_bfloat16> {};

// sin and cos of a rotation angle: Cody-Waite reduction by pi/2 in three fused steps (pi/2 = hi + mid + lo to ~72 bits;
// the fma keeps each partial product exact, so the reduced argument is good to half an ulp for |k| < 2^15) and the
// cephes minimax polynomials on [-pi/4, pi/4]: max abs error 8.9e-8 against float64 over positions 0..30000 x every
// frequency (checked on the CPU with emulated float32 fmas) -- the libm sincosf this replaces is good to 1 ulp too, at
// three times the instructions (its Payne-Hanek path for huge arguments is kept for exactly those).  The angle's
// evaluation was what made the half types compute-bound.
__device__ __forceinline__ void rope_sincos(float x, float& s, float& c) {
    if (!(fabsf(x) < 30000.f)) { sincosf(x, &s, &c); return; }           // (huge positions, inf, nan: libm)
    const float kf = rintf(x * 0.63661977236758134f);
    float r = fmaf(-kf, 1.5707963705062866f, x);
    r = fmaf(-kf, -4.371138828673793e-08f, r);
    r = fmaf(-kf, -1.7763568394002505e-15f, r);
    const float z = r * r;
    const float ps = fmaf(fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f);
    const float sr = fmaf(ps * z, r, r);
    const float pc = fmaf(fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f);
    const float cr = fmaf(pc * z, z, fmaf(-0.5f, z, 1.0f));
    const int k = (int)kf;
    const float a = (k & 1) ? cr : sr, b = (k & 1) ? sr : cr;            // quadrant: (s, c) = (sr, cr), (cr, -sr), (-sr, -cr), (-cr, sr)
    s = (k & 2) ? -a : a;
    c = ((k + 1) & 2) ? -b : b;
}

// Vector path.  One lane owns EPL = 16 bytes / sizeof(T) consecutive frequencies of one (token, half) and walks HEADS
// heads with them: the rotation angle depends on (token, half, frequency) only, so it is evaluated once per HEADS head
// rows.  Consecutive lanes walk the frequencies, then the half, so at every step of the head loop a group of 2*Q/EPL
// lanes covers one contiguous head row.  `tokens2` (may be null): a second tensor of the same shape / strides /
// positions rotated in the same launch with the same angles (q and k of one attention layer, croco/blocks.py:102-104).
// HEADS is picked per launch (launch_rope_t: four, see rope_heads).  Every token row a lane will rotate is REQUESTED
// before the angles are evaluated (the loads do not depend on them), and index arithmetic is 32-bit whenever the launch
// allows (I).
template <typename T, int HEADS, typename I>
__global__ __launch_bounds__(kBlock) void spf_rope2d_vec_kernel(T* __restrict__ tokens, T* __restrict__ tokens2,
                                                                const int64_t* __restrict__ pos, int N, int H, int D,
                                                                int64_t stride_b, int64_t stride_n, int64_t stride_h, int pos_div,
                                                                RopeFreq f, I total) {
    using W = Wide<T>;
    constexpr int EPL = W::N;
    const I item = (I)blockIdx.x * (I)kBlock + (I)threadIdx.x;
    if (item >= total) return;
    const int Q = D >> 2, QG = Q / EPL, per_chunk = 2 * QG;
    const int nchunk = (H + HEADS - 1) / HEADS;
    const I per_tok = (I)(nchunk * per_chunk);
    const I token = item / per_tok;
    const int rem = (int)(item - token * per_tok);
    const int hc = rem / per_chunk, e = rem - hc * per_chunk;
    const int x = e / QG, q0 = (e - x * QG) * EPL;
    const I b = token / (I)N, n = token - b * (I)N;
    const int64_t pword = pos[((size_t)(b / (I)pos_div) * N + n) * 2 + x];
    const int h0 = hc * HEADS;
    const size_t base = (size_t)b * stride_b + (size_t)n * stride_n + (size_t)h0 * stride_h + x * 2 * Q + q0;
    float u[HEADS][EPL], v[HEADS][EPL];
#pragma unroll
    for (int j = 0; j < HEADS; ++j)
        if (h0 + j < H) { W::load(tokens + base + j * stride_h, u[j]); W::load(tokens + base + j * stride_h + Q, v[j]); }
    const float p = (float)pword;
    float sn[EPL], cs[EPL];
#pragma unroll
    for (int k = 0; k < EPL; ++k) rope_sincos(p * f.inv[q0 + k], sn[k], cs[k]);
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        T* __restrict__ t = pass == 0 ? tokens : tokens2;
        if (!t) break;
        if (pass == 1) {
#pragma unroll
            for (int j = 0; j < HEADS; ++j)
                if (h0 + j < H) { W::load(t + base + j * stride_h, u[j]); W::load(t + base + j * stride_h + Q, v[j]); }
        }
#pragma unroll
        for (int j = 0; j < HEADS; ++j)
            if (h0 + j < H) {
                float uo[EPL], vo[EPL];
#pragma unroll
                for (int k = 0; k < EPL; ++k) {
                    uo[k] = u[j][k] * cs[k] - v[j][k] * sn[k];
                    vo[k] = v[j][k] * cs[k] + u[j][k] * sn[k];
                }
                W::store(t + base + j * stride_h, uo);
                W::store(t + base + j * stride_h + Q, vo);
            }
    }
}

// Scalar path: any D % 4 == 0, any alignment.  One lane per (token, head, half, q).
template <typename T>
__global__ __launch_bounds__(kBlock) void spf_rope2d_scalar_kernel(T* __restrict__ tokens, T* __restrict__ tokens2,
                                                                   const int64_t* __restrict__ pos, int N, int H,
                                                                   int D, int64_t stride_b, int64_t stride_n, int64_t stride_h,
                                                                   int pos_div, RopeFreq f, size_t total) {
    const size_t item = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (item >= total) return;
    const int Q = D >> 2, per_head = 2 * Q;
    const int per_tok = H * per_head;
    const size_t token = item / per_tok;
    const int rem = (int)(item - token * per_tok);
    const int h = rem / per_head, e = rem - h * per_head;
    const int x = e / Q, q = e - x * Q;
    const size_t b = token / N, n = token - b * N;
    const float p = (float)pos[((b / pos_div) * N + n) * 2 + x];
    float s, c;
    rope_sincos(p * f.inv[q], s, c);
    const size_t o = b * stride_b + n * stride_n + (size_t)h * stride_h + x * 2 * Q + q;
    for (int pass = 0; pass < 2; ++pass) {
        T* __restrict__ up = (pass == 0 ? tokens : tokens2);
        if (!up) break;
        up += o;
        const float u = to_f<T>(up[0]), v = to_f<T>(up[Q]);
        up[0] = from_f<T>(u * c - v * s);
        up[Q] = from_f<T>(v * c + u * s);
    }
}

template <typename T, int HEADS>
static void launch_rope_vec(void* tokens, void* tokens2, const int64_t* pos, int B, int N, int H, int D, int64_t sb,
                            int64_t sn, int64_t sh, int pos_div, const RopeFreq& f, hipStream_t stream) {
    constexpr int EPL = Wide<T>::N;
    const size_t total = (size_t)B * N * ((H + HEADS - 1) / HEADS) * 2 * ((D / 4) / EPL);
    const unsigned grid = (unsigned)((total + kBlock - 1) / kBlock);
    if (total < ((size_t)1 << 31))
        spf_rope2d_vec_kernel<T, HEADS, uint32_t><<<grid, kBlock, 0, stream>>>(
            static_cast<T*>(tokens), static_cast<T*>(tokens2), pos, N, H, D, sb, sn, sh, pos_div, f, (uint32_t)total);
    else
        spf_rope2d_vec_kernel<T, HEADS, size_t><<<grid, kBlock, 0, stream>>>(
            static_cast<T*>(tokens), static_cast<T*>(tokens2), pos, N, H, D, sb, sn, sh, pos_div, f, total);
}

// heads per lane (see the kernel): 4.  Measured (round 5, 50 dependent calls per HIP graph, us per call, heads 1 / 2 / 4):
// (32,258,12,64) fp16 8.4 / 7.6 / 7.2, fp32 9.4 / 7.5 / 7.6; (48,256,16,64) fp16 12.3 / 10.3 / 10.1, fp32 19.4 / 16.8 / 15.9 --
// fewer, fatter lanes win or tie everywhere, also where four heads per lane leave the chip 1.5 waves per SIMD (the small
// shapes run at the ~7 us floor of a dependent launch either way).  SPF_ROPE_HEADS=1/2/4 pins it (experiments).
static int rope_heads(size_t lanes_at_4) {
    static const int forced = getenv("SPF_ROPE_HEADS") ? atoi(getenv("SPF_ROPE_HEADS")) : 0;
    if (forced == 1 || forced == 2 || forced == 4) return forced;
    (void)lanes_at_4;
    return 4;
}

template <typename T>
static hipError_t launch_rope_t(void* tokens, void* tokens2, const int64_t* pos, int B, int N, int H, int D, int64_t sb,
                                int64_t sn, int64_t sh, int pos_div, const RopeFreq& f, hipStream_t stream) {
    const int Q = D / 4;
    constexpr int EPL = Wide<T>::N;
    const bool vec = (Q % EPL == 0) && (reinterpret_cast<uintptr_t>(tokens) % 16 == 0) &&
                     (reinterpret_cast<uintptr_t>(tokens2) % 16 == 0) && (sb % EPL == 0) && (sn % EPL == 0) &&
                     (sh % EPL == 0);
    if (vec) {
        const int heads = rope_heads((size_t)B * N * ((H + 3) / 4) * 2 * (Q / EPL));
        if (heads == 4) launch_rope_vec<T, 4>(tokens, tokens2, pos, B, N, H, D, sb, sn, sh, pos_div, f, stream);
        else if (heads == 2) launch_rope_vec<T, 2>(tokens, tokens2, pos, B, N, H, D, sb, sn, sh, pos_div, f, stream);
        else launch_rope_vec<T, 1>(tokens, tokens2, pos, B, N, H, D, sb, sn, sh, pos_div, f, stream);
    } else {
        const size_t total = (size_t)B * N * H * 2 * Q;
        spf_rope2d_scalar_kernel<T><<<(unsigned)((total + kBlock - 1) / kBlock), kBlock, 0, stream>>>(
            static_cast<T*>(tokens), static_cast<T*>(tokens2), pos, N, H, D, sb, sn, sh, pos_div, f, total);
    }
    return hipGetLastError();
}

hipError_t launch_rope2d(void* tokens, void* tokens2, const int64_t* pos, int B, int N, int H, int D, int64_t sb,
                         int64_t sn, int64_t sh, int pos_div, int dtype, float base, float fwd, hipStream_t stream) {
    RopeFreq f;
    const int Q = D / 4;
    for (int q = 0; q < 64; ++q) f.inv[q] = q < Q ? fwd / powf(base, q / float(Q)) : 0.f;
    switch (dtype) {
        case 0: return launch_rope_t<float>(tokens, tokens2, pos, B, N, H, D, sb, sn, sh, pos_div, f, stream);
        case 1: return launch_rope_t<__half>(tokens, tokens2, pos, B, N, H, D, sb, sn, sh, pos_div, f, stream);
        default: return launch_rope_t<__hip_bfloat16>(tokens, tokens2, pos, B, N, H, D, sb, sn, sh, pos_div, f, stream);
    }
}

}  // namespace spf
