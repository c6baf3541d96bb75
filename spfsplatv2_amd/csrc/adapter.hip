// Fused Gaussian-adapter pre-pass for gfx950: raw network channels -> rasterizer-ready parameters, forward and
// backward, replacing the elementwise torch chain of UnifiedGaussianAdapter.forward
// (/root/reference/src/model/encoder/common/gaussian_adapter.py:122-150):
//   scales    = min(0.001 * softplus(raw[0:3]), 0.3)
//   rotations = raw[3:7] / (||raw[3:7]|| + eps)
//   harmonics = raw[7 + c*K + k] * sh_mask[k]           -> [N,3,K] (the layout the projection kernels read)
// One lane per Gaussian for the 7 geometric channels; the 3K harmonic channels are a flat, coalesced scale-copy.
#include "spf_common.h"

namespace spf {

__device__ __forceinline__ float softplus_torch(float x) { return x > 20.f ? x : log1pf(expf(x)); }

__global__ __launch_bounds__(kBlock) void spf_adapter_geom_fwd_kernel(const float* __restrict__ raw, int64_t N, int C,
                                                                     float eps, float* __restrict__ scales,
                                                                     float* __restrict__ rot) {
    const int64_t n = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (n >= N) return;
    const float* __restrict__ r = raw + n * C;
#pragma unroll
    for (int i = 0; i < 3; ++i) scales[3 * n + i] = fminf(0.001f * softplus_torch(r[i]), 0.3f);
    const float q0 = r[3], q1 = r[4], q2 = r[5], q3 = r[6];
    const float inv = 1.0f / (sqrtf(q0 * q0 + q1 * q1 + q2 * q2 + q3 * q3) + eps);
    *reinterpret_cast<float4*>(rot + 4 * n) = make_float4(q0 * inv, q1 * inv, q2 * inv, q3 * inv);
}

__global__ __launch_bounds__(kBlock) void spf_adapter_geom_bwd_kernel(const float* __restrict__ raw, int64_t N, int C,
                                                                     float eps, const float* __restrict__ g_scales,
                                                                     const float* __restrict__ g_rot,
                                                                     float* __restrict__ g_raw) {
    const int64_t n = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (n >= N) return;
    const float* __restrict__ r = raw + n * C;
    float* __restrict__ o = g_raw + n * C;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const float x = r[i];
        const float sp = softplus_torch(x);
        const float dsp = x > 20.f ? 1.f : 1.f / (1.f + expf(-x));          // softplus' = sigmoid
        const float pass = (0.001f * sp <= 0.3f) ? 1.f : 0.f;               // clamp_max passes the gradient up to the bound
        o[i] = g_scales ? g_scales[3 * n + i] * 0.001f * dsp * pass : 0.f;
    }
    const float q[4] = {r[3], r[4], r[5], r[6]};
    float g[4] = {0.f, 0.f, 0.f, 0.f};
    if (g_rot) {
        const float4 t = *reinterpret_cast<const float4*>(g_rot + 4 * n);
        g[0] = t.x; g[1] = t.y; g[2] = t.z; g[3] = t.w;
    }
    const float nrm = sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    const float d = nrm + eps, dot = g[0] * q[0] + g[1] * q[1] + g[2] * q[2] + g[3] * q[3];
    const float k = nrm > 0.f ? dot / (nrm * d * d) : 0.f;                  // r = q/(|q|+eps): dr = dq/d - q (q.dq)/(|q| d^2)
#pragma unroll
    for (int i = 0; i < 4; ++i) o[3 + i] = g[i] / d - q[i] * k;
}

// harmonics: out[n][j] = raw[n][7 + j] * mask[j % K], j < 3K  (forward), g_raw[n][7 + j] = g_sh[n][j] * mask[j % K]
__global__ __launch_bounds__(kBlock) void spf_adapter_sh_kernel(const float* __restrict__ src, int64_t src_stride,
                                                               int64_t src_off, float* __restrict__ dst,
                                                               int64_t dst_stride, int64_t dst_off,
                                                               const float* __restrict__ mask, int K, int64_t total) {
    const int K3 = 3 * K;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (int64_t)gridDim.x * kBlock) {
        const int64_t n = i / K3;
        const int j = (int)(i - n * K3);
        dst[n * dst_stride + dst_off + j] = src[n * src_stride + src_off + j] * mask[j % K];
    }
}

hipError_t launch_adapter_fwd(const float* raw, int64_t N, int K, const float* mask, float eps, float* scales,
                              float* rot, float* sh, hipStream_t stream) {
    const int C = 7 + 3 * K;
    spf_adapter_geom_fwd_kernel<<<(unsigned)((N + kBlock - 1) / kBlock), kBlock, 0, stream>>>(raw, N, C, eps, scales, rot);
    const int64_t total = N * 3 * K;
    const int64_t want = (total + kBlock - 1) / kBlock;
    const unsigned grid = (unsigned)(want < 256 * 16 ? want : 256 * 16);
    spf_adapter_sh_kernel<<<grid, kBlock, 0, stream>>>(raw, C, 7, sh, 3 * K, 0, mask, K, total);
    return hipGetLastError();
}

hipError_t launch_adapter_bwd(const float* raw, int64_t N, int K, const float* mask, float eps, const float* g_scales,
                              const float* g_rot, const float* g_sh, float* g_raw, hipStream_t stream) {
    const int C = 7 + 3 * K;
    spf_adapter_geom_bwd_kernel<<<(unsigned)((N + kBlock - 1) / kBlock), kBlock, 0, stream>>>(raw, N, C, eps, g_scales,
                                                                                            g_rot, g_raw);
    const int64_t total = N * 3 * K;
    const int64_t want = (total + kBlock - 1) / kBlock;
    const unsigned grid = (unsigned)(want < 256 * 16 ? want : 256 * 16);
    if (g_sh)
        spf_adapter_sh_kernel<<<grid, kBlock, 0, stream>>>(g_sh, 3 * K, 0, g_raw, C, 7, mask, K, total);
    else
        return hipMemset2DAsync(g_raw + 7, sizeof(float) * C, 0, sizeof(float) * 3 * K, (size_t)N, stream);
    return hipGetLastError();
}

}  // namespace spf
