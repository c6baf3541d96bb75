// Fused Gaussian-adapter pre-pass for gfx950: raw network channels -> rasterizer-ready parameters, forward and
// backward, replacing the elementwise torch chain of UnifiedGaussianAdapter.forward
// (/root/reference/src/model/encoder/common/gaussian_adapter.py:122-150):
//   scales    = min(0.001 * softplus(raw[0:3]), 0.3)
//   rotations = raw[3:7] / (||raw[3:7]|| + eps)
//   harmonics = raw[7 + c*K + k] * sh_mask[k]           -> [N,3,K] (the layout the projection kernels read), or, band
//                                                          split (K = 25): [N,3,16] and [N,3,9] (SpfDims.sh_layout 2)
// ONE pass per direction.  A block walks tiles of 64 Gaussians: the tile's raw rows (7 + 3K floats each, at the caller's
// row stride -- the encoder hands over `gaussians[..., 1:]`, a view into its 83-channel head output,
// encoder_spfsplatv2.py:261-268, which is read in place) come in through LDS with flat, coalesced loads, and every
// output leaves as a flat, coalesced store: each byte crosses HBM once.  (Round 4's two kernels -- one lane per Gaussian
// for the seven geometric channels, a flat scale-copy for the harmonics -- read every raw row twice, 28 bytes of it at a
// 328-byte stride, and needed a contiguous copy of a strided input first.)
#include "spf_common.h"

namespace spf {

constexpr int kAdRows = 64;          // Gaussians per tile
constexpr int kAdLow = 16, kAdHigh = 9;

__device__ __forceinline__ float softplus_torch(float x) { return x > 20.f ? x : log1pf(expf(x)); }
__host__ __device__ inline int adapter_row_stride(int C) { return C | 1; }     // odd: rows start in different LDS banks

// flat index e of a [rows, 3, KP] plane -> LDS offset of raw channel 7 + c*K + k0 + k inside the tile
__device__ __forceinline__ int plane_src(int e, int KP, int k0, int K, int Cs, int& k) {
    const int n = e / (3 * KP), rem = e - n * 3 * KP;
    const int c = rem / KP;
    k = k0 + rem - c * KP;
    return n * Cs + 7 + c * K + k;
}

__global__ __launch_bounds__(kBlock) void spf_adapter_fwd_kernel(const float* __restrict__ raw, int64_t stride, int64_t N,
                                                                int K, const float* __restrict__ mask, float eps,
                                                                float* __restrict__ scales, float* __restrict__ rot,
                                                                float* __restrict__ sh, float* __restrict__ sh_hi) {
    extern __shared__ __attribute__((aligned(16))) float s_ad[];
    const int C = 7 + 3 * K, Cs = adapter_row_stride(C);
    float* const s_mask = s_ad + kAdRows * Cs;
    const int tid = threadIdx.x;
    for (int k = tid; k < K; k += kBlock) s_mask[k] = mask[k];
    const int64_t ntile = (N + kAdRows - 1) / kAdRows;
    const bool vec = stride == C && (reinterpret_cast<uintptr_t>(raw) & 15) == 0 && (kAdRows * C) % 4 == 0;
    for (int64_t t = blockIdx.x; t < ntile; t += gridDim.x) {
        const int64_t n0 = t * kAdRows;
        const int rows = (int)min((int64_t)kAdRows, N - n0);
        __syncthreads();                                    // (the previous tile has been consumed; the mask is in place)
        if (vec && rows == kAdRows) {
            const float4* __restrict__ src = reinterpret_cast<const float4*>(raw + n0 * C);
            for (int i = tid; i < kAdRows * C / 4; i += kBlock) {
                const float4 v = src[i];
                const float x[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int f = 4 * i + q, r = f / C;
                    s_ad[r * Cs + (f - r * C)] = x[q];
                }
            }
        } else {
            for (int i = tid; i < rows * C; i += kBlock) {
                const int r = i / C, j = i - r * C;
                s_ad[r * Cs + j] = raw[(n0 + r) * stride + j];
            }
        }
        __syncthreads();
        // ---- geometry: 3 scale and 4 quaternion floats per Gaussian, one output float per thread ----
        if (tid < rows * 3) {
            const int r = tid / 3, k = tid - 3 * r;
            scales[n0 * 3 + tid] = fminf(0.001f * softplus_torch(s_ad[r * Cs + k]), 0.3f);
        }
        if (tid < rows * 4) {
            const float* __restrict__ q = s_ad + (tid >> 2) * Cs + 3;
            const float inv = 1.0f / (sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]) + eps);
            rot[n0 * 4 + tid] = q[tid & 3] * inv;
        }
        // ---- harmonics: flat planes ----
        if (sh_hi) {
            float* __restrict__ lo = sh + n0 * 3 * kAdLow;
            for (int e = tid; e < rows * 3 * kAdLow; e += kBlock) {
                int k;
                const int src = plane_src(e, kAdLow, 0, K, Cs, k);
                lo[e] = s_ad[src] * s_mask[k];
            }
            float* __restrict__ hi = sh_hi + n0 * 3 * kAdHigh;
            for (int e = tid; e < rows * 3 * kAdHigh; e += kBlock) {
                int k;
                const int src = plane_src(e, kAdHigh, kAdLow, K, Cs, k);
                hi[e] = s_ad[src] * s_mask[k];
            }
        } else {
            float* __restrict__ o = sh + n0 * 3 * K;
            for (int e = tid; e < rows * 3 * K; e += kBlock) {
                int k;
                const int src = plane_src(e, K, 0, K, Cs, k);
                o[e] = s_ad[src] * s_mask[k];
            }
        }
    }
}

// dL/draw [N, 7+3K] (contiguous): a tile's rows are assembled in LDS and leave as one flat store.
__global__ __launch_bounds__(kBlock) void spf_adapter_bwd_kernel(const float* __restrict__ raw, int64_t stride, int64_t N,
                                                                int K, const float* __restrict__ mask, float eps,
                                                                const float* __restrict__ g_scales,
                                                                const float* __restrict__ g_rot,
                                                                const float* __restrict__ g_sh,
                                                                const float* __restrict__ g_sh_hi, int split,
                                                                float* __restrict__ g_raw) {
    extern __shared__ __attribute__((aligned(16))) float s_ad[];
    const int C = 7 + 3 * K, Cs = adapter_row_stride(C);
    float* const s_mask = s_ad + kAdRows * Cs;
    const int tid = threadIdx.x;
    for (int k = tid; k < K; k += kBlock) s_mask[k] = mask[k];
    const int64_t ntile = (N + kAdRows - 1) / kAdRows;
    const bool vec = (reinterpret_cast<uintptr_t>(g_raw) & 15) == 0 && (kAdRows * C) % 4 == 0;
    for (int64_t t = blockIdx.x; t < ntile; t += gridDim.x) {
        const int64_t n0 = t * kAdRows;
        const int rows = (int)min((int64_t)kAdRows, N - n0);
        __syncthreads();
        // ---- geometry (one Gaussian per thread: the chain through the quaternion norm needs all four components) ----
        if (tid < rows) {
            const float* __restrict__ r = raw + (n0 + tid) * stride;
            float* __restrict__ o = s_ad + tid * Cs;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const float x = r[i];
                const float sp = softplus_torch(x);
                const float dsp = x > 20.f ? 1.f : 1.f / (1.f + expf(-x));          // softplus' = sigmoid
                const float pass = (0.001f * sp <= 0.3f) ? 1.f : 0.f;               // clamp_max passes the gradient up to the bound
                o[i] = g_scales ? g_scales[(n0 + tid) * 3 + i] * 0.001f * dsp * pass : 0.f;
            }
            const float q[4] = {r[3], r[4], r[5], r[6]};
            float g[4] = {0.f, 0.f, 0.f, 0.f};
            if (g_rot) {
                const float4 tq = *reinterpret_cast<const float4*>(g_rot + (n0 + tid) * 4);
                g[0] = tq.x; g[1] = tq.y; g[2] = tq.z; g[3] = tq.w;
            }
            const float nrm = sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
            const float d = nrm + eps, dot = g[0] * q[0] + g[1] * q[1] + g[2] * q[2] + g[3] * q[3];
            const float k = nrm > 0.f ? dot / (nrm * d * d) : 0.f;                  // r = q/(|q|+eps): dr = dq/d - q (q.dq)/(|q| d^2)
#pragma unroll
            for (int i = 0; i < 4; ++i) o[3 + i] = g[i] / d - q[i] * k;
        }
        // ---- harmonics ----
        if (split) {
            for (int e = tid; e < rows * 3 * kAdLow; e += kBlock) {
                int k;
                const int dst = plane_src(e, kAdLow, 0, K, Cs, k);
                s_ad[dst] = g_sh ? g_sh[n0 * 3 * kAdLow + e] * s_mask[k] : 0.f;
            }
            for (int e = tid; e < rows * 3 * kAdHigh; e += kBlock) {
                int k;
                const int dst = plane_src(e, kAdHigh, kAdLow, K, Cs, k);
                s_ad[dst] = g_sh_hi ? g_sh_hi[n0 * 3 * kAdHigh + e] * s_mask[k] : 0.f;     // (NULL: band 4 was not evaluated)
            }
        } else {
            for (int e = tid; e < rows * 3 * K; e += kBlock) {
                int k;
                const int dst = plane_src(e, K, 0, K, Cs, k);
                s_ad[dst] = g_sh ? g_sh[n0 * 3 * K + e] * s_mask[k] : 0.f;
            }
        }
        __syncthreads();
        if (vec && rows == kAdRows) {
            float4* __restrict__ dst = reinterpret_cast<float4*>(g_raw + n0 * C);
            for (int i = tid; i < kAdRows * C / 4; i += kBlock) {
                float x[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int f = 4 * i + q, r = f / C;
                    x[q] = s_ad[r * Cs + (f - r * C)];
                }
                dst[i] = make_float4(x[0], x[1], x[2], x[3]);
            }
        } else {
            for (int i = tid; i < rows * C; i += kBlock) {
                const int r = i / C;
                g_raw[n0 * C + i] = s_ad[r * Cs + (i - r * C)];
            }
        }
    }
}

static unsigned adapter_grid(int64_t N) {
    const int64_t ntile = (N + kAdRows - 1) / kAdRows;
    return (unsigned)(ntile < 256 * 8 ? ntile : 256 * 8);
}
static size_t adapter_lds(int K) { return sizeof(float) * ((size_t)kAdRows * adapter_row_stride(7 + 3 * K) + (size_t)K); }

hipError_t launch_adapter_fwd(const float* raw, int64_t stride, int64_t N, int K, const float* mask, float eps,
                              float* scales, float* rot, float* sh, float* sh_hi, hipStream_t stream) {
    spf_adapter_fwd_kernel<<<adapter_grid(N), kBlock, adapter_lds(K), stream>>>(raw, stride, N, K, mask, eps, scales, rot,
                                                                                 sh, sh_hi);
    return hipGetLastError();
}

hipError_t launch_adapter_bwd(const float* raw, int64_t stride, int64_t N, int K, const float* mask, float eps,
                              const float* g_scales, const float* g_rot, const float* g_sh, const float* g_sh_hi,
                              int split, float* g_raw, hipStream_t stream) {
    spf_adapter_bwd_kernel<<<adapter_grid(N), kBlock, adapter_lds(K), stream>>>(raw, stride, N, K, mask, eps, g_scales,
                                                                                 g_rot, g_sh, g_sh_hi, split, g_raw);
    return hipGetLastError();
}

}  // namespace spf
