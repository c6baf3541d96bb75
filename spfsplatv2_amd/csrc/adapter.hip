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


// flat index e of a [rows, 3, KP] plane -> offset of raw channel 7 + c*K + k0 + k inside the group's rows (row stride C)
__device__ __forceinline__ int plane_src(int e, int KP, int k0, int K, int C, int& k) {
    const int n = e / (3 * KP), rem = e - n * 3 * KP;
    const int c = rem / KP;
    k = k0 + rem - c * KP;
    return n * C + 7 + c * K + k;
}

// ---- forward ------------------------------------------------------------------------------------------------------
// The group's rows as registers: element i = lane + 64 t of the group's kAdRows * C floats (flat or row-strided source).
template <int KT, int NT>
__device__ __forceinline__ void adapter_load_rows(const float* __restrict__ raw, int64_t stride, bool flat, int64_t n0,
                                                  int rows, int lane, float (&v)[NT]) {
    constexpr int C = 7 + 3 * KT;
    const int nflt = rows * C;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int i = lane + kWave * t;
        float x = 0.f;
        if (i < nflt) {
            if (flat) {
                x = raw[n0 * C + i];
            } else {
                const int r = i / C;
                x = raw[(n0 + r) * stride + (i - r * C)];
            }
        }
        v[t] = x;
    }
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
    const int64_t gstep = (int64_t)gridDim.x * 4;
    constexpr int NT = KT ? (kAdRows * (7 + 3 * KT) + kWave - 1) / kWave : 1;
    float v[NT], vn[NT];
    int64_t grp = (int64_t)blockIdx.x * 4 + wave;
    if (KT && grp < ngroup)
        adapter_load_rows<KT ? KT : 1, NT>(raw, stride, flat, grp * kAdRows, (int)min((int64_t)kAdRows, N - grp * kAdRows), lane, v);
    for (; grp < ngroup; grp += gstep) {
        const int64_t n0 = grp * kAdRows;
        const int rows = (int)min((int64_t)kAdRows, N - n0);
        const int nflt = rows * C;
        if (KT) {
            // software pipeline: the NEXT group's loads go out before this group is touched -- a wave keeps two groups
            // (5 KB) in flight instead of alternating between waiting for loads and waiting for stores
            const int64_t nx = grp + gstep;
            if (nx < ngroup)
                adapter_load_rows<KT ? KT : 1, NT>(raw, stride, flat, nx * kAdRows, (int)min((int64_t)kAdRows, N - nx * kAdRows), lane, vn);
#pragma unroll
            for (int t = 0; t < NT; ++t)
                if (lane + kWave * t < nflt) s_row[lane + kWave * t] = v[t];
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
#pragma unroll
            for (int t = 0; t < kAdRows * 3 * kAdLow / kWave; ++t) {
                const int e = lane + kWave * t;
                int k;
                const int src = plane_src(e, kAdLow, 0, K, C, k);
                if (e < rows * 3 * kAdLow) lo[e] = s_row[src] * s_mask[k];
            }
            float* __restrict__ hi = sh_hi + n0 * 3 * kAdHigh;
#pragma unroll
            for (int t = 0; t < (kAdRows * 3 * kAdHigh + kWave - 1) / kWave; ++t) {
                const int e = lane + kWave * t;
                int k;
                const int src = plane_src(e, kAdHigh, kAdLow, K, C, k);
                if (e < rows * 3 * kAdHigh) hi[e] = s_row[src] * s_mask[k];
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
        if (KT) {
#pragma unroll
            for (int t = 0; t < NT; ++t) v[t] = vn[t];
        }
    }
}

// ---- backward -----------------------------------------------------------------------------------------------------
// dL/draw [N, 7+3K] (contiguous): a group's rows are assembled in the wave's LDS and leave as one flat store.  What a
// group reads -- its gradient planes, the seven geometric raw channels of its rows, dL/dscales, dL/drotations -- is
// requested for the NEXT group before this one is assembled.
template <int KT>
struct AdapterGradRegs {
    static constexpr int NS = KT ? (kAdRows * 3 * KT + kWave - 1) / kWave : 1;     // dense plane, or:
    static constexpr int NLO = kAdRows * 3 * kAdLow / kWave, NHI = (kAdRows * 3 * kAdHigh + kWave - 1) / kWave;
    float sh[NS > NLO + NHI ? NS : NLO + NHI];
    float raw7, gs, gr;
};
template <int KT>
__device__ __forceinline__ void adapter_load_grads(AdapterGradRegs<KT>& g, const float* __restrict__ raw, int64_t stride,
                                                   const float* __restrict__ g_scales, const float* __restrict__ g_rot,
                                                   const float* __restrict__ g_sh, const float* __restrict__ g_sh_hi,
                                                   int split, int64_t n0, int rows, int lane) {
    using R = AdapterGradRegs<KT>;
    if (split) {
#pragma unroll
        for (int t = 0; t < R::NLO; ++t) {
            const int e = lane + kWave * t;
            g.sh[t] = (g_sh && e < rows * 3 * kAdLow) ? g_sh[n0 * 3 * kAdLow + e] : 0.f;
        }
#pragma unroll
        for (int t = 0; t < R::NHI; ++t) {
            const int e = lane + kWave * t;
            g.sh[R::NLO + t] = (g_sh_hi && e < rows * 3 * kAdHigh) ? g_sh_hi[n0 * 3 * kAdHigh + e] : 0.f;   // (NULL: band 4 not evaluated)
        }
    } else {
#pragma unroll
        for (int t = 0; t < R::NS; ++t) {
            const int e = lane + kWave * t;
            g.sh[t] = (g_sh && e < rows * 3 * KT) ? g_sh[n0 * 3 * KT + e] : 0.f;
        }
    }
    const int r7 = lane / 7;
    g.raw7 = lane < rows * 7 ? raw[(n0 + r7) * stride + (lane - 7 * r7)] : 0.f;
    g.gs = (g_scales && lane < rows * 3) ? g_scales[n0 * 3 + lane] : 0.f;
    g.gr = (g_rot && lane < rows * 4) ? g_rot[n0 * 4 + lane] : 0.f;
}

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
    float* const s_geo = s_ad + 4 * kAdRows * C + ((K + 3) & ~3) + wave * 128;     // raw7 [56] | gs [24] | gr [32]
    for (int k = threadIdx.x; k < K; k += kBlock) s_mask_w[k] = mask[k];
    __syncthreads();
    const int64_t ngroup = (N + kAdRows - 1) / kAdRows;
    const int64_t gstep = (int64_t)gridDim.x * 4;
    using R = AdapterGradRegs<KT>;
    R cur, nxt;
    int64_t grp = (int64_t)blockIdx.x * 4 + wave;
    if (KT && grp < ngroup)
        adapter_load_grads<KT>(cur, raw, stride, g_scales, g_rot, g_sh, g_sh_hi, split, grp * kAdRows,
                               (int)min((int64_t)kAdRows, N - grp * kAdRows), lane);
    for (; grp < ngroup; grp += gstep) {
        const int64_t n0 = grp * kAdRows;
        const int rows = (int)min((int64_t)kAdRows, N - n0);
        if (KT) {
            const int64_t nx = grp + gstep;
            if (nx < ngroup)
                adapter_load_grads<KT>(nxt, raw, stride, g_scales, g_rot, g_sh, g_sh_hi, split, nx * kAdRows,
                                       (int)min((int64_t)kAdRows, N - nx * kAdRows), lane);
            // ---- harmonics into the rows ----
            if (split) {
#pragma unroll
                for (int t = 0; t < R::NLO; ++t) {
                    const int e = lane + kWave * t;
                    int k;
                    const int dst = plane_src(e, kAdLow, 0, K, C, k);
                    if (e < rows * 3 * kAdLow) s_row[dst] = cur.sh[t] * s_mask[k];
                }
#pragma unroll
                for (int t = 0; t < R::NHI; ++t) {
                    const int e = lane + kWave * t;
                    int k;
                    const int dst = plane_src(e, kAdHigh, kAdLow, K, C, k);
                    if (e < rows * 3 * kAdHigh) s_row[dst] = cur.sh[R::NLO + t] * s_mask[k];
                }
            } else {
#pragma unroll
                for (int t = 0; t < R::NS; ++t) {
                    const int e = lane + kWave * t;
                    int k;
                    const int dst = plane_src(e, K, 0, K, C, k);
                    if (e < rows * 3 * K) s_row[dst] = cur.sh[t] * s_mask[k];
                }
            }
            if (lane < 56) s_geo[lane] = cur.raw7;
            if (lane < 24) s_geo[56 + lane] = cur.gs;
            if (lane < 32) s_geo[80 + lane] = cur.gr;
        } else {
            for (int e = lane; e < rows * 3 * K; e += kWave) {
                int k;                                          // (the split layout is K = 25: always a template instance)
                const int dst = plane_src(e, K, 0, K, C, k);
                s_row[dst] = g_sh ? g_sh[n0 * 3 * K + e] * s_mask[k] : 0.f;
            }
            const int r7 = lane / 7;
            if (lane < rows * 7) s_geo[lane] = raw[(n0 + r7) * stride + (lane - 7 * r7)];
            if (lane < 24) s_geo[56 + lane] = (g_scales && lane < rows * 3) ? g_scales[n0 * 3 + lane] : 0.f;
            if (lane < 32) s_geo[80 + lane] = (g_rot && lane < rows * 4) ? g_rot[n0 * 4 + lane] : 0.f;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // ---- geometry (one Gaussian per lane: the chain through the quaternion norm needs all four components) ----
        if (lane < rows) {
            const float* __restrict__ r = s_geo + 7 * lane;
            float* __restrict__ o = s_row + lane * C;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const float x = r[i];
                const float sp = softplus_torch(x);
                const float dsp = x > 20.f ? 1.f : 1.f / (1.f + expf(-x));          // softplus' = sigmoid
                const float pass = (0.001f * sp <= 0.3f) ? 1.f : 0.f;               // clamp_max passes the gradient up to the bound
                o[i] = s_geo[56 + 3 * lane + i] * 0.001f * dsp * pass;
            }
            const float q[4] = {r[3], r[4], r[5], r[6]};
            const float g[4] = {s_geo[80 + 4 * lane], s_geo[81 + 4 * lane], s_geo[82 + 4 * lane], s_geo[83 + 4 * lane]};
            const float nrm = sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
            const float d = nrm + eps, dot = g[0] * q[0] + g[1] * q[1] + g[2] * q[2] + g[3] * q[3];
            const float k = nrm > 0.f ? dot / (nrm * d * d) : 0.f;                  // r = q/(|q|+eps): dr = dq/d - q (q.dq)/(|q| d^2)
#pragma unroll
            for (int i = 0; i < 4; ++i) o[3 + i] = g[i] / d - q[i] * k;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        float* __restrict__ dst = g_raw + n0 * C;
        if (KT) {
            constexpr int NF = (kAdRows * (7 + 3 * KT) + kWave - 1) / kWave;
#pragma unroll
            for (int t = 0; t < NF; ++t)
                if (lane + kWave * t < rows * C) dst[lane + kWave * t] = s_row[lane + kWave * t];
        } else {
            for (int i = lane; i < rows * C; i += kWave) dst[i] = s_row[i];
        }
        __builtin_amdgcn_wave_barrier();
        if (KT) cur = nxt;
    }
}

static unsigned adapter_grid(int64_t N) {
    const int64_t nblk = ((N + kAdRows - 1) / kAdRows + 3) / 4;
    return (unsigned)(nblk < 256 * 16 ? (nblk < 1 ? 1 : nblk) : 256 * 16);
}
static size_t adapter_lds(int K) { return sizeof(float) * ((size_t)4 * kAdRows * (7 + 3 * K) + (size_t)((K + 3) & ~3) + 4 * 128); }

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
