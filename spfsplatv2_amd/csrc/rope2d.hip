// In-place 2-D rotary position embedding for gfx950.
//
// Replaces rope_2d_cuda_kernel (/root/reference/src/model/encoder/backbone/croco/curope/kernels.cu:17-82).
// tokens[B,N,H,D] (stride(H)=D, stride(D)=1, arbitrary outer strides); one head row is laid out as
//   [ u_Y (Q) | v_Y (Q) | u_X (Q) | v_X (Q) ],  Q = D/4,
// and every (u,v) pair with frequency index q is rotated by  angle = pos * fwd / base^(q/Q)
// where pos is the token's y position for the first half and its x position for the second.
//
// Mapping: one lane owns 4 consecutive frequencies of one (token, head, half): two 16-byte
// loads + two 16-byte stores (8-byte for half types); consecutive lanes walk q, then the half, then
// the head, so a wave touches whole contiguous head rows.  The Q inverse frequencies are computed
// on the host with libm powf (exactly what the reference's CPU path evaluates,
// curope/curope.cpp:35) and travel in the kernel-argument segment -- no device powf, no table in HBM.
#include <hip/hip_bf16.h>
#include <hip/hip_fp16.h>

#include "spf_common.h"

namespace spf {

struct RopeFreq {
    float inv[64];  // fwd / base^(q/Q), q < Q <= 64
};

template <typename T> struct Vec4;
template <> struct Vec4<float> {
    using type = float4;
    static __device__ __forceinline__ void unpack(const type& v, float* f) { f[0] = v.x; f[1] = v.y; f[2] = v.z; f[3] = v.w; }
    static __device__ __forceinline__ type pack(const float* f) { return make_float4(f[0], f[1], f[2], f[3]); }
};
template <> struct Vec4<__half> {
    struct __attribute__((aligned(8))) type { __half h[4]; };
    static __device__ __forceinline__ void unpack(const type& v, float* f) {
#pragma unroll
        for (int i = 0; i < 4; ++i) f[i] = __half2float(v.h[i]);
    }
    static __device__ __forceinline__ type pack(const float* f) {
        type v;
#pragma unroll
        for (int i = 0; i < 4; ++i) v.h[i] = __float2half(f[i]);
        return v;
    }
};
template <> struct Vec4<__hip_bfloat16> {
    struct __attribute__((aligned(8))) type { __hip_bfloat16 h[4]; };
    static __device__ __forceinline__ void unpack(const type& v, float* f) {
#pragma unroll
        for (int i = 0; i < 4; ++i) f[i] = __bfloat162float(v.h[i]);
    }
    static __device__ __forceinline__ type pack(const float* f) {
        type v;
#pragma unroll
        for (int i = 0; i < 4; ++i) v.h[i] = __float2bfloat16(f[i]);
        return v;
    }
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

// Vector path: Q % 4 == 0 and every row 4-element aligned.
template <typename T>
__global__ __launch_bounds__(kBlock) void spf_rope2d_vec_kernel(T* __restrict__ tokens,
                                                                const int64_t* __restrict__ pos, int N, int H, int D,
                                                                int64_t stride_b, int64_t stride_n, int64_t stride_h, int pos_div,
                                                                RopeFreq f, size_t total) {
    using V = Vec4<T>;
    const size_t item = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (item >= total) return;
    const int Q = D >> 2, per_head = D >> 3, q4n = Q >> 2;
    const int per_tok = H * per_head;
    const size_t token = item / per_tok;
    const int rem = (int)(item - token * per_tok);
    const int h = rem / per_head, e = rem - h * per_head;
    const int x = e / q4n, q0 = (e - x * q4n) * 4;
    const size_t b = token / N, n = token - b * N;
    const float p = (float)pos[((b / pos_div) * N + n) * 2 + x];
    T* __restrict__ up = tokens + b * stride_b + n * stride_n + (size_t)h * stride_h + x * 2 * Q + q0;
    typename V::type uv = *reinterpret_cast<const typename V::type*>(up);
    typename V::type vv = *reinterpret_cast<const typename V::type*>(up + Q);
    float u[4], v[4], uo[4], vo[4];
    V::unpack(uv, u);
    V::unpack(vv, v);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        float s, c;
        sincosf(p * f.inv[q0 + k], &s, &c);
        uo[k] = u[k] * c - v[k] * s;
        vo[k] = v[k] * c + u[k] * s;
    }
    *reinterpret_cast<typename V::type*>(up) = V::pack(uo);
    *reinterpret_cast<typename V::type*>(up + Q) = V::pack(vo);
}

// Scalar path: any D % 4 == 0, any alignment.  One lane per (token, head, half, q).
template <typename T>
__global__ __launch_bounds__(kBlock) void spf_rope2d_scalar_kernel(T* __restrict__ tokens,
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
    T* __restrict__ up = tokens + b * stride_b + n * stride_n + (size_t)h * stride_h + x * 2 * Q + q;
    const float u = to_f<T>(up[0]), v = to_f<T>(up[Q]);
    float s, c;
    sincosf(p * f.inv[q], &s, &c);
    up[0] = from_f<T>(u * c - v * s);
    up[Q] = from_f<T>(v * c + u * s);
}

template <typename T>
static hipError_t launch_rope_t(void* tokens, const int64_t* pos, int B, int N, int H, int D, int64_t sb, int64_t sn,
                                int64_t sh, int pos_div, const RopeFreq& f, hipStream_t stream) {
    const int Q = D / 4;
    const size_t align = 4 * sizeof(T);
    const bool vec = (Q % 4 == 0) && (reinterpret_cast<uintptr_t>(tokens) % align == 0) && (sb % 4 == 0) &&
                     (sn % 4 == 0) && (sh % 4 == 0);
    if (vec) {
        const size_t total = (size_t)B * N * H * (D / 8);
        spf_rope2d_vec_kernel<T><<<(unsigned)((total + kBlock - 1) / kBlock), kBlock, 0, stream>>>(
            static_cast<T*>(tokens), pos, N, H, D, sb, sn, sh, pos_div, f, total);
    } else {
        const size_t total = (size_t)B * N * H * 2 * Q;
        spf_rope2d_scalar_kernel<T><<<(unsigned)((total + kBlock - 1) / kBlock), kBlock, 0, stream>>>(
            static_cast<T*>(tokens), pos, N, H, D, sb, sn, sh, pos_div, f, total);
    }
    return hipGetLastError();
}

hipError_t launch_rope2d(void* tokens, const int64_t* pos, int B, int N, int H, int D, int64_t sb, int64_t sn,
                         int64_t sh, int pos_div, int dtype, float base, float fwd, hipStream_t stream) {
    RopeFreq f;
    const int Q = D / 4;
    for (int q = 0; q < 64; ++q) f.inv[q] = q < Q ? fwd / powf(base, q / float(Q)) : 0.f;
    switch (dtype) {
        case 0: return launch_rope_t<float>(tokens, pos, B, N, H, D, sb, sn, sh, pos_div, f, stream);
        case 1: return launch_rope_t<__half>(tokens, pos, B, N, H, D, sb, sn, sh, pos_div, f, stream);
        default: return launch_rope_t<__hip_bfloat16>(tokens, pos, B, N, H, D, sb, sn, sh, pos_div, f, stream);
    }
}

}  // namespace spf
