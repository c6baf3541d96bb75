// Fused Gaussian-adapter pre-pass for gfx950: raw network channels -> rasterizer-ready parameters, forward and
// backward, replacing the elementwise torch chain of UnifiedGaussianAdapter.forward
// (/root/reference/src/model/encoder/common/gaussian_adapter.py:122-150):
//   scales    = min(0.001 * softplus(raw[0:3]), 0.3)
//   rotations = raw[3:7] / (||raw[3:7]|| + eps)
//   harmonics = raw[7 + c*K + k] * sh_mask[k]           -> [N,3,K] (the layout the projection kernels read), or, band
//                                                          split (K = 25): [N,3,16] and [N,3,9] (SpfDims.sh_layout 2)
// ONE pass per direction, no block barrier.  A WAVE walks groups of eight Gaussians: the group's raw rows (7 + 3K floats
// each, at the caller's row stride -- the encoder hands over `gaussians[..., 1:]`, a view into its 83-channel head output,
// encoder_spfsplatv2.py:261-268, which is read in place) come in with flat, coalesced loads into the wave's own 2.6 KB of
// LDS, and every output leaves as a flat, coalesced store: each byte crosses HBM once.  K is a template parameter for the
// sizes the models use (the index arithmetic is divisions by 7 + 3K and K: constants fold to a multiply and a shift; with
// run-time divisors the pass was bound by them -- 0.6 ms where the bytes need 0.25), 0 = any other K at run time.
// (Round 4's two kernels -- one lane per Gaussian for the seven geometric channels, a flat scale-copy for the harmonics --
// read every raw row twice, 28 bytes of it at a 328-byte stride, and needed a contiguous copy of a strided input first.)
#include "spf_common.h"

namespace spf {

constexpr int kAdRows = 8;           // Gaussians per wave and trip
constexpr int kAdLow = 16, kAdHigh = 9;

__device__ __forceinline__ float softplus_torch(float x) { return x > 20.f ? x : log1pf(expf(x)); }

// flat index e of a [rows, 3, KP] plane -> offset of raw channel 7 + c*K + k0 + k inside the group's rows (row stride C)
__device__ __forceinline__ int plane_src(int e, int KP, int k0, int K, int C, int& k) {
    const int n = e / (3 * KP), rem = e - n * 3 * KP;
    const int c = rem / KP;
    k = k0 + rem - c * KP;
    return n * C + 7 + c * K + k;
}

template <int KT>
__global__ __launch_bounds__(kBlock) void spf_adapter_fwd_kernel(const float* __restrict__ raw, int64_t stride, int64_t N,
                                                                int K_rt, const float* __restrict__ mask, float eps,
                                                                float* __restrict__ scales, float* __restrict__ rot,
                                                                float* __restrict__ sh, float* __restrict__ sh_hi) {
    extern __shared__ __attribute__((aligned(16))) float s_ad[];
    const int K = KT ? KT : K_rt, C = 7 + 3 * K;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float* const s_row = s_ad + wave * (kAdRows * C);                   // this wave's rows
    const float* const s_mask = s_ad + 4 * kAdRows * C;
    float* const s_mask_w = s_ad + 4 * kAdRows * C;
    for (int k = threadIdx.x; k < K; k += kBlock) s_mask_w[k] = mask[k];
    __syncthreads();                                                    // (the only one: the mask)
    const int64_t ngroup = (N + kAdRows - 1) / kAdRows;
    const bool flat = stride == C;
    for (int64_t grp = (int64_t)blockIdx.x * 4 + wave; grp < ngroup; grp += (int64_t)gridDim.x * 4) {
        const int64_t n0 = grp * kAdRows;
        const int rows = (int)min((int64_t)kAdRows, N - n0);
        const int nflt = rows * C;
        if (flat && KT && rows == kAdRows) {
            // (a full group with a compile-time row length: every load of the group is in flight before the first wait)
            const float* __restrict__ src = raw + n0 * C;
            constexpr int NF = kAdRows * (7 + 3 * KT), NT = (NF + kWave - 1) / kWave;
            float v[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) v[t] = (lane + kWave * t < NF) ? src[lane + kWave * t] : 0.f;
#pragma unroll
            for (int t = 0; t < NT; ++t)
                if (lane + kWave * t < NF) s_row[lane + kWave * t] = v[t];
        } else if (flat) {
            const float* __restrict__ src = raw + n0 * C;
            for (int i = lane; i < nflt; i += kWave) s_row[i] = src[i];
        } else {
            for (int i = lane; i < nflt; i += kWave) {
                const int r = i / C;
                s_row[i] = raw[(n0 + r) * stride + (i - r * C)];
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // ---- geometry: 3 scale and 4 quaternion floats per Gaussian, one output float per lane ----
        if (lane < rows * 3) {
            const int r = lane / 3, k = lane - 3 * r;
            scales[n0 * 3 + lane] = fminf(0.001f * softplus_torch(s_row[r * C + k]), 0.3f);
        }
        if (lane < rows * 4) {
            const float* __restrict__ q = s_row + (lane >> 2) * C + 3;
            const float inv = 1.0f / (sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]) + eps);
            rot[n0 * 4 + lane] = q[lane & 3] * inv;
        }
        // ---- harmonics: flat planes ----
        if (sh_hi) {
            float* __restrict__ lo = sh + n0 * 3 * kAdLow;
            for (int e = lane; e < rows * 3 * kAdLow; e += kWave) {
                int k;
                const int src = plane_src(e, kAdLow, 0, K, C, k);
                lo[e] = s_row[src] * s_mask[k];
            }
            float* __restrict__ hi = sh_hi + n0 * 3 * kAdHigh;
            for (int e = lane; e < rows * 3 * kAdHigh; e += kWave) {
                int k;
                const int src = plane_src(e, kAdHigh, kAdLow, K, C, k);
                hi[e] = s_row[src] * s_mask[k];
            }
        } else {
            float* __restrict__ o = sh + n0 * 3 * K;
            for (int e = lane; e < rows * 3 * K; e += kWave) {
                int k;
                const int src = plane_src(e, K, 0, K, C, k);
                o[e] = s_row[src] * s_mask[k];
            }
        }
        __builtin_amdgcn_wave_barrier();                               // (the rows are rewritten by the next trip)
    }
}

// dL/draw [N, 7+3K] (contiguous): a group's rows are assembled in the wave's LDS and leave as one flat store.
template <int KT>
__global__ __launch_bounds__(kBlock) void spf_adapter_bwd_kernel(const float* __restrict__ raw, int64_t stride, int64_t N,
                                                                int K_rt, const float* __restrict__ mask, float eps,
                                                                const float* __restrict__ g_scales,
                                                                const float* __restrict__ g_rot,
                                                                const float* __restrict__ g_sh,
                                                                const float* __restrict__ g_sh_hi, int split,
                                                                float* __restrict__ g_raw) {
    extern __shared__ __attribute__((aligned(16))) float s_ad[];
    const int K = KT ? KT : K_rt, C = 7 + 3 * K;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float* const s_row = s_ad + wave * (kAdRows * C);
    const float* const s_mask = s_ad + 4 * kAdRows * C;
    float* const s_mask_w = s_ad + 4 * kAdRows * C;
    for (int k = threadIdx.x; k < K; k += kBlock) s_mask_w[k] = mask[k];
    __syncthreads();
    const int64_t ngroup = (N + kAdRows - 1) / kAdRows;
    for (int64_t grp = (int64_t)blockIdx.x * 4 + wave; grp < ngroup; grp += (int64_t)gridDim.x * 4) {
        const int64_t n0 = grp * kAdRows;
        const int rows = (int)min((int64_t)kAdRows, N - n0);
        // ---- geometry (one Gaussian per lane: the chain through the quaternion norm needs all four components) ----
        if (lane < rows) {
            const float* __restrict__ r = raw + (n0 + lane) * stride;
            float* __restrict__ o = s_row + lane * C;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const float x = r[i];
                const float sp = softplus_torch(x);
                const float dsp = x > 20.f ? 1.f : 1.f / (1.f + expf(-x));          // softplus' = sigmoid
                const float pass = (0.001f * sp <= 0.3f) ? 1.f : 0.f;               // clamp_max passes the gradient up to the bound
                o[i] = g_scales ? g_scales[(n0 + lane) * 3 + i] * 0.001f * dsp * pass : 0.f;
            }
            const float q[4] = {r[3], r[4], r[5], r[6]};
            float g[4] = {0.f, 0.f, 0.f, 0.f};
            if (g_rot) {
                const float4 tq = *reinterpret_cast<const float4*>(g_rot + (n0 + lane) * 4);
                g[0] = tq.x; g[1] = tq.y; g[2] = tq.z; g[3] = tq.w;
            }
            const float nrm = sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
            const float d = nrm + eps, dot = g[0] * q[0] + g[1] * q[1] + g[2] * q[2] + g[3] * q[3];
            const float k = nrm > 0.f ? dot / (nrm * d * d) : 0.f;                  // r = q/(|q|+eps): dr = dq/d - q (q.dq)/(|q| d^2)
#pragma unroll
            for (int i = 0; i < 4; ++i) o[3 + i] = g[i] / d - q[i] * k;
        }
        // ---- harmonics ----
        if (split && g_sh && rows == kAdRows) {
            constexpr int NT = kAdRows * 3 * kAdLow / kWave;         // 6 coalesced loads in flight
            float v[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) v[t] = g_sh[n0 * 3 * kAdLow + lane + kWave * t];
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                int k;
                const int dst = plane_src(lane + kWave * t, kAdLow, 0, K, C, k);
                s_row[dst] = v[t] * s_mask[k];
            }
            for (int e = lane; e < rows * 3 * kAdHigh; e += kWave) {
                int k;
                const int dst = plane_src(e, kAdHigh, kAdLow, K, C, k);
                s_row[dst] = g_sh_hi ? g_sh_hi[n0 * 3 * kAdHigh + e] * s_mask[k] : 0.f;     // (NULL: band 4 was not evaluated)
            }
        } else if (split) {
            for (int e = lane; e < rows * 3 * kAdLow; e += kWave) {
                int k;
                const int dst = plane_src(e, kAdLow, 0, K, C, k);
                s_row[dst] = g_sh ? g_sh[n0 * 3 * kAdLow + e] * s_mask[k] : 0.f;
            }
            for (int e = lane; e < rows * 3 * kAdHigh; e += kWave) {
                int k;
                const int dst = plane_src(e, kAdHigh, kAdLow, K, C, k);
                s_row[dst] = g_sh_hi ? g_sh_hi[n0 * 3 * kAdHigh + e] * s_mask[k] : 0.f;     // (NULL: band 4 was not evaluated)
            }
        } else {
            for (int e = lane; e < rows * 3 * K; e += kWave) {
                int k;
                const int dst = plane_src(e, K, 0, K, C, k);
                s_row[dst] = g_sh ? g_sh[n0 * 3 * K + e] * s_mask[k] : 0.f;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        float* __restrict__ dst = g_raw + n0 * C;
        for (int i = lane; i < rows * C; i += kWave) dst[i] = s_row[i];
        __builtin_amdgcn_wave_barrier();
    }
}

static unsigned adapter_grid(int64_t N) {
    const int64_t nblk = ((N + kAdRows - 1) / kAdRows + 3) / 4;
    return (unsigned)(nblk < 256 * 16 ? (nblk < 1 ? 1 : nblk) : 256 * 16);
}
static size_t adapter_lds(int K) { return sizeof(float) * ((size_t)4 * kAdRows * (7 + 3 * K) + (size_t)K); }

#define SPF_ADAPTER_K(FN, ...)                      \
    switch (K) {                                    \
        case 25: FN<25> __VA_ARGS__; break;         \
        case 16: FN<16> __VA_ARGS__; break;         \
        case 9: FN<9> __VA_ARGS__; break;           \
        case 4: FN<4> __VA_ARGS__; break;           \
        case 1: FN<1> __VA_ARGS__; break;           \
        default: FN<0> __VA_ARGS__; break;          \
    }

hipError_t launch_adapter_fwd(const float* raw, int64_t stride, int64_t N, int K, const float* mask, float eps,
                              float* scales, float* rot, float* sh, float* sh_hi, hipStream_t stream) {
    SPF_ADAPTER_K(spf_adapter_fwd_kernel, <<<adapter_grid(N), kBlock, adapter_lds(K), stream>>>(raw, stride, N, K, mask, eps,
                                                                                                 scales, rot, sh, sh_hi))
    return hipGetLastError();
}

hipError_t launch_adapter_bwd(const float* raw, int64_t stride, int64_t N, int K, const float* mask, float eps,
                              const float* g_scales, const float* g_rot, const float* g_sh, const float* g_sh_hi,
                              int split, float* g_raw, hipStream_t stream) {
    SPF_ADAPTER_K(spf_adapter_bwd_kernel, <<<adapter_grid(N), kBlock, adapter_lds(K), stream>>>(
        raw, stride, N, K, mask, eps, g_scales, g_rot, g_sh, g_sh_hi, split, g_raw))
    return hipGetLastError();
}
#undef SPF_ADAPTER_K

}  // namespace spf
