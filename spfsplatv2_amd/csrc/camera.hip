// Camera set-up for a batch of renders, forward and backward, as two tiny kernels (one lane per render).
//
// Replaces the per-step chain of small PyTorch ops in the reference's render_cuda
// (/root/reference/src/model/decoder/cuda_splatting.py:66-74,84-91): scale-invariant rescale of the
// camera translation, get_fov (src/geometry/projection.py:269-283), tan(fov/2), get_projection_matrix
// (cuda_splatting.py:15-42), extrinsics.inverse() and the two transposes.  On MI355X those ~100 launches of
// a few microseconds each cost more host time than the whole rasterizer costs device time.
#include "spf_common.h"

namespace spf {

template <typename T>
__device__ __forceinline__ T det3(T a, T b, T c, T d, T e, T f, T g, T h, T i) {
    return a * (e * i - f * h) - b * (d * i - f * g) + c * (d * h - e * g);
}

// General 4x4 inverse by cofactors (row-major).  Returns false if singular.  Evaluated in float64 by the forward
// (one lane per render: the cost is nil, and the float32 cofactor sums were only good to ~1e-6 of the translation).
template <typename T>
__device__ __forceinline__ bool inv4(const T* m, T* o) {
    T c[16];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int col = 0; col < 4; ++col) {
            T s[9];
            int k = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (i != r && j != col) s[k++] = m[4 * i + j];
            const T minor = det3<T>(s[0], s[1], s[2], s[3], s[4], s[5], s[6], s[7], s[8]);
            c[4 * r + col] = ((r + col) & 1) ? -minor : minor;
        }
    const T det = m[0] * c[0] + m[1] * c[1] + m[2] * c[2] + m[3] * c[3];
    const T id = T(1) / det;
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int col = 0; col < 4; ++col) o[4 * r + col] = c[4 * col + r] * id;  // adjugate = cofactor^T
    return det != T(0);
}

// (float64 like the pose: tan(fov/2) scales every pixel coordinate, and a splat whose centre lies thousands of pixels
//  outside the image -- its footprint may still cross it -- moved by 1e-3 px with the float32 acos / tan chain)
__device__ __forceinline__ void inv3(const double* m, double* o) {
    const double c00 = m[4] * m[8] - m[5] * m[7], c01 = m[5] * m[6] - m[3] * m[8], c02 = m[3] * m[7] - m[4] * m[6];
    const double id = 1.0 / (m[0] * c00 + m[1] * c01 + m[2] * c02);
    o[0] = c00 * id; o[1] = (m[2] * m[7] - m[1] * m[8]) * id; o[2] = (m[1] * m[5] - m[2] * m[4]) * id;
    o[3] = c01 * id; o[4] = (m[0] * m[8] - m[2] * m[6]) * id; o[5] = (m[2] * m[3] - m[0] * m[5]) * id;
    o[6] = c02 * id; o[7] = (m[1] * m[6] - m[0] * m[7]) * id; o[8] = (m[0] * m[4] - m[1] * m[3]) * id;
}

// un-normalised ray through normalised image coordinates (u, v)
__device__ __forceinline__ void pixel_ray(const double* Kinv, double u, double v, double* d) {
    d[0] = Kinv[0] * u + Kinv[1] * v + Kinv[2];
    d[1] = Kinv[3] * u + Kinv[4] * v + Kinv[5];
    d[2] = Kinv[6] * u + Kinv[7] * v + Kinv[8];
}
// tan(theta / 2) for the angle theta between two rays, without normalising them, acos or tan: with c = cos theta,
// tan^2(theta/2) = (1 - c) / (1 + c) = (|a||b| - a.b) / (|a||b| + a.b).  The reference's chain is acos of the dot
// product of unit rays followed by tan of half of it (projection.py:269-283, cuda_splatting.py:84-85): the same number
// (float64 here, to ~1e-16), at a fifth of the dependent instructions of the one lane that sets a render's camera up.
__device__ __forceinline__ double tan_half_angle(const double* a, const double* b) {
    const double dot = a[0] * b[0] + a[1] * b[1] + a[2] * b[2];
    const double nn = sqrt((a[0] * a[0] + a[1] * a[1] + a[2] * a[2]) * (b[0] * b[0] + b[1] * b[1] + b[2] * b[2]));
    return sqrt((nn - dot) / (nn + dot));
}

__device__ __forceinline__ void camera_fwd_one(const SpfCamera& c, int r) {
    float nr = c.near[r], fr = c.far[r];
    const float scale = c.scale_invariant ? 1.0f / nr : 1.0f;
    const double scale_d = c.scale_invariant ? 1.0 / (double)nr : 1.0;
    double A[16], B[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) A[i] = (double)c.extrinsics[16 * r + i];
    if (c.scale_invariant) {
        A[3] *= scale_d; A[7] *= scale_d; A[11] *= scale_d;
        nr = nr * scale;
        fr = fr * scale;
    }
    inv4<double>(A, B);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            c.viewmatrix[16 * r + 4 * i + j] = (float)B[4 * j + i];
            // float64 copy for the projection kernels, the world scale folded into the rows that multiply the mean
            if (c.viewmatrix64) c.viewmatrix64[16 * r + 4 * i + j] = (i < 3 ? scale_d : 1.0) * B[4 * j + i];
        }
    // field of view from the normalised intrinsics
    double K[9], Ki[9], l[3], rr[3], t[3], b[3];
#pragma unroll
    for (int i = 0; i < 9; ++i) K[i] = (double)c.intrinsics[9 * r + i];
    inv3(K, Ki);
    pixel_ray(Ki, 0.0, 0.5, l); pixel_ray(Ki, 1.0, 0.5, rr);
    pixel_ray(Ki, 0.5, 0.0, t); pixel_ray(Ki, 0.5, 1.0, b);
    const double tan_xd = tan_half_angle(l, rr), tan_yd = tan_half_angle(t, b);
    const float tan_x = (float)tan_xd, tan_y = (float)tan_yd;
    c.tanfov[2 * r] = tan_x;
    c.tanfov[2 * r + 1] = tan_y;
    if (c.view_scale) c.view_scale[r] = scale;
    // perspective matrix (column-vector form P), stored transposed
    const float top = tan_y * nr, bottom = -top, right = tan_x * nr, left = -right;
    float P[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) P[i] = 0.f;
    P[0] = (float)(1.0 / tan_xd);      // = 2 n / (right - left)
    P[5] = (float)(1.0 / tan_yd);      // = 2 n / (top - bottom)
    P[2] = (right + left) / (right - left);
    P[6] = (top + bottom) / (top - bottom);
    P[14] = 1.f;
    P[10] = fr / (fr - nr);
    P[11] = -(fr * nr) / (fr - nr);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) c.projmatrix[16 * r + 4 * i + j] = P[4 * j + i];
}

__global__ void spf_camera_fwd_kernel(SpfCamera c) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < c.R) camera_fwd_one(c, r);
}

// The decoder's first launch: camera set-up AND the clearing of the per-tile counters the projection kernel
// accumulates into (`zero`, 16-byte aligned, `nvec` uint4) -- one kernel instead of a kernel and a memset node.
__global__ __launch_bounds__(kBlock) void spf_camera_fwd_zero_kernel(SpfCamera c, uint4* __restrict__ zero,
                                                                     uint64_t nvec) {
    const uint64_t t = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    for (uint64_t i = t; i < nvec; i += (uint64_t)gridDim.x * kBlock) zero[i] = make_uint4(0u, 0u, 0u, 0u);
    for (uint64_t r = t; r < (uint64_t)c.R; r += (uint64_t)gridDim.x * kBlock) camera_fwd_one(c, (int)r);
}

// dL/dA = -B^T (dL/dB) B^T with B = A^-1 = viewmatrix^T and dL/dB = (dL/dviewmatrix)^T; then undo the
// translation rescale.
__device__ __forceinline__ void camera_bwd_one(const SpfCamera& c, int r, const float* dL_dview,
                                               float* __restrict__ dL_dext) {
    const float scale = c.scale_invariant ? 1.0f / c.near[r] : 1.0f;
    float Bt[16], Gb[16], T1[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) Bt[i] = c.viewmatrix[16 * r + i];            // B^T
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) Gb[4 * i + j] = dL_dview[4 * j + i];  // dL/dB
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) s += Bt[4 * i + k] * Gb[4 * k + j];
            T1[4 * i + j] = s;
        }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) s += T1[4 * i + k] * Bt[4 * k + j];
            if (j == 3 && i < 3) s *= scale;
            dL_dext[16 * r + 4 * i + j] = -s;
        }
}

__global__ void spf_camera_bwd_kernel(SpfCamera c, const float* __restrict__ dL_dview,
                                      float* __restrict__ dL_dext) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < c.R) camera_bwd_one(c, r, dL_dview + 16 * r, dL_dext);
}

// The decoder's last launch: sum the projection backward's per-block viewmatrix partials vpartial[r][0..nblk)[12]
// (fixed order: deterministic) and chain the result to the pose -- what spf_view_reduce_kernel + the kernel above do
// in two launches.  grid = R, block = 256: four waves take every fourth group of 64 blocks (three 16-byte loads per
// record), their totals are added in a fixed order -- deterministic, and a quarter of the dependent loads per wave that a
// single wave had (the kernel is pure latency: 1,954 records per render on the 500,000-Gaussian workload took 13.8 us).
__global__ __launch_bounds__(kBlock) void spf_camera_bwd_reduce_kernel(SpfCamera c, const float* __restrict__ vpartial,
                                                                       int nblk, float* __restrict__ dL_dext) {
    __shared__ float s_w[kBlock / kWave][12];
    __shared__ float s_dv[16];
    const int r = blockIdx.x, lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
    float acc[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) acc[k] = 0.f;
    for (int b = threadIdx.x; b < nblk; b += kBlock) {
        const float4* pp = reinterpret_cast<const float4*>(vpartial + ((size_t)r * nblk + b) * 12);
        const float4 a = pp[0], b4 = pp[1], c4 = pp[2];
        acc[0] += a.x; acc[1] += a.y; acc[2] += a.z; acc[3] += a.w;
        acc[4] += b4.x; acc[5] += b4.y; acc[6] += b4.z; acc[7] += b4.w;
        acc[8] += c4.x; acc[9] += c4.y; acc[10] += c4.z; acc[11] += c4.w;
    }
#pragma unroll
    for (int k = 0; k < 12; ++k) acc[k] = wave_sum(acc[k]);
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < 12; ++k) s_w[wave][k] = acc[k];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float tot[12];
#pragma unroll
        for (int k = 0; k < 12; ++k) tot[k] = (s_w[0][k] + s_w[1][k]) + (s_w[2][k] + s_w[3][k]);
        // partial k < 9: dL/dVm[4i+j] with k = 3i+j (i, j < 3); k = 9+j: dL/dVm[12+j]; the last column gets none
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) s_dv[4 * i + j] = j < 3 ? tot[i < 3 ? 3 * i + j : 9 + j] : 0.f;
        camera_bwd_one(c, r, s_dv, dL_dext);
    }
}

hipError_t launch_camera_fwd(const SpfCamera& c, hipStream_t stream) {
    spf_camera_fwd_kernel<<<(c.R + 63) / 64, 64, 0, stream>>>(c);
    return hipGetLastError();
}

hipError_t launch_camera_bwd(const SpfCamera& c, const float* dL_dview, float* dL_dext, hipStream_t stream) {
    spf_camera_bwd_kernel<<<(c.R + 63) / 64, 64, 0, stream>>>(c, dL_dview, dL_dext);
    return hipGetLastError();
}

hipError_t launch_camera_fwd_zero(const SpfCamera& c, void* zero, uint64_t nbytes, hipStream_t stream) {
    const uint64_t nvec = nbytes / 16;
    const uint64_t want = ((nvec > (uint64_t)c.R ? nvec : (uint64_t)c.R) + kBlock - 1) / kBlock;
    const int grid = (int)(want < 1 ? 1 : (want > 1024 ? 1024 : want));
    spf_camera_fwd_zero_kernel<<<grid, kBlock, 0, stream>>>(c, static_cast<uint4*>(zero), nvec);
    return hipGetLastError();
}

hipError_t launch_camera_bwd_reduce(const SpfCamera& c, const float* vpartial, int nblk, float* dL_dext,
                                    hipStream_t stream) {
    spf_camera_bwd_reduce_kernel<<<c.R, kBlock, 0, stream>>>(c, vpartial, nblk, dL_dext);
    return hipGetLastError();
}

}  // namespace spf
