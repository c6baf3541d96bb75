// Photometric MSE loss on the decoder's output, fused for gfx950.
//
// Replaces LossMse.forward (/root/reference/src/loss/loss_mse.py:36-51): weight * mean((prediction - image)^2), which
// eager PyTorch runs as four elementwise/reduction kernels forward and four more backward over the 25 MB image batch.
// Here: one pass forward (read both images, deterministic two-level sum) and one pass backward
// (grad = (2 * weight / N) * dL/dloss * (prediction - image), the upstream scalar read on the device: no sync).
#include "spf_common.h"

namespace spf {

constexpr int kLossBlocks = 1024;

// Sum of squared differences, two levels: every block writes its partial to partial[block]; a one-block kernel then
// adds the partials in a FIXED order and writes scale * total -> run-to-run identical.  (A single launch with a
// "last block finishes" ticket was measured 4x slower: the device-scope fence every block needs for it writes the L2
// back.)
// GRAD: the same pass also writes unit[i] = scale2 * (prediction[i] - image[i]) -- the gradient for dL/dloss = 1, which
// is what `loss.backward()` hands over.  The backward then only has to look at the upstream scalar (spf_mse_scale_kernel):
// 25 MB more written here, 75 MB less moved there.
template <bool GRAD>
__global__ __launch_bounds__(kBlock) void spf_mse_fwd_kernel(const float* __restrict__ pred,
                                                             const float* __restrict__ target, int64_t n,
                                                             float* __restrict__ partial, float scale2,
                                                             float* __restrict__ unit) {
    __shared__ float s_w[kBlock / kWave];
    const int64_t n4 = n >> 2;
    const float4* __restrict__ p4 = reinterpret_cast<const float4*>(pred);
    const float4* __restrict__ t4 = reinterpret_cast<const float4*>(target);
    float4* __restrict__ u4 = reinterpret_cast<float4*>(unit);
    float acc = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n4; i += (int64_t)gridDim.x * kBlock) {
        const float4 a = p4[i], b = t4[i];
        const float dx = a.x - b.x, dy = a.y - b.y, dz = a.z - b.z, dw = a.w - b.w;
        acc += (dx * dx + dy * dy) + (dz * dz + dw * dw);
        if (GRAD) u4[i] = make_float4(scale2 * dx, scale2 * dy, scale2 * dz, scale2 * dw);
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {      // tail (n not a multiple of 4)
        const float d = pred[(n4 << 2) + threadIdx.x] - target[(n4 << 2) + threadIdx.x];
        acc += d * d;
        if (GRAD) unit[(n4 << 2) + threadIdx.x] = scale2 * d;
    }
    const float w = wave_sum(acc);
    if ((threadIdx.x & (kWave - 1)) == 0) s_w[threadIdx.x >> 6] = w;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (s_w[0] + s_w[1]) + (s_w[2] + s_w[3]);
}

// grad[i] *= dL/dloss, in place -- and nothing at all when dL/dloss is exactly 1 (every block leaves after one scalar
// read): the usual backward of a loss costs a launch, not a pass over the images.
__global__ __launch_bounds__(kBlock) void spf_mse_scale_kernel(float* __restrict__ grad, int64_t n,
                                                               const float* __restrict__ grad_loss) {
    const float g = grad_loss[0];
    if (g == 1.0f) return;
    const int64_t n4 = n >> 2;
    float4* __restrict__ o4 = reinterpret_cast<float4*>(grad);
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n4; i += (int64_t)gridDim.x * kBlock) {
        const float4 a = o4[i];
        o4[i] = make_float4(g * a.x, g * a.y, g * a.z, g * a.w);
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) grad[(n4 << 2) + threadIdx.x] *= g;
}

__global__ __launch_bounds__(kBlock) void spf_mse_final_kernel(const float* __restrict__ partial, int nblocks,
                                                               float scale, float* __restrict__ loss) {
    __shared__ float s_w[kBlock / kWave];
    float tot = 0.f;
    for (int b = threadIdx.x; b < nblocks; b += kBlock) tot += partial[b];
    const float w = wave_sum(tot);
    if ((threadIdx.x & (kWave - 1)) == 0) s_w[threadIdx.x >> 6] = w;
    __syncthreads();
    if (threadIdx.x == 0) *loss = scale * ((s_w[0] + s_w[1]) + (s_w[2] + s_w[3]));
}

__global__ __launch_bounds__(kBlock) void spf_mse_bwd_kernel(const float* __restrict__ pred,
                                                             const float* __restrict__ target, int64_t n, float scale2,
                                                             const float* __restrict__ grad_loss,
                                                             float* __restrict__ grad_pred) {
    const float g = scale2 * grad_loss[0];
    const int64_t n4 = n >> 2;
    const float4* __restrict__ p4 = reinterpret_cast<const float4*>(pred);
    const float4* __restrict__ t4 = reinterpret_cast<const float4*>(target);
    float4* __restrict__ o4 = reinterpret_cast<float4*>(grad_pred);
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n4; i += (int64_t)gridDim.x * kBlock) {
        const float4 a = p4[i], b = t4[i];
        o4[i] = make_float4(g * (a.x - b.x), g * (a.y - b.y), g * (a.z - b.z), g * (a.w - b.w));
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const int64_t i = (n4 << 2) + threadIdx.x;
        grad_pred[i] = g * (pred[i] - target[i]);
    }
}

int mse_partial_blocks() { return kLossBlocks; }

hipError_t launch_mse_fwd(const float* pred, const float* target, int64_t n, float scale, float* partial,
                          float* loss, float scale2, float* unit_grad, hipStream_t stream) {
    const int64_t want = ((n >> 2) + kBlock - 1) / kBlock;
    const int grid = (int)(want < 1 ? 1 : (want > kLossBlocks ? kLossBlocks : want));
    if (unit_grad) spf_mse_fwd_kernel<true><<<grid, kBlock, 0, stream>>>(pred, target, n, partial, scale2, unit_grad);
    else spf_mse_fwd_kernel<false><<<grid, kBlock, 0, stream>>>(pred, target, n, partial, 0.f, nullptr);
    spf_mse_final_kernel<<<1, kBlock, 0, stream>>>(partial, grid, scale, loss);
    return hipGetLastError();
}

hipError_t launch_mse_scale(float* grad, int64_t n, const float* grad_loss, hipStream_t stream) {
    const int64_t want = ((n >> 2) + kBlock - 1) / kBlock;
    const int grid = (int)(want < 1 ? 1 : (want > 4 * kLossBlocks ? 4 * kLossBlocks : want));
    spf_mse_scale_kernel<<<grid, kBlock, 0, stream>>>(grad, n, grad_loss);
    return hipGetLastError();
}

hipError_t launch_mse_bwd(const float* pred, const float* target, int64_t n, float scale2, const float* grad_loss,
                          float* grad_pred, hipStream_t stream) {
    const int64_t want = ((n >> 2) + kBlock - 1) / kBlock;
    const int grid = (int)(want < 1 ? 1 : (want > 4 * kLossBlocks ? 4 * kLossBlocks : want));
    spf_mse_bwd_kernel<<<grid, kBlock, 0, stream>>>(pred, target, n, scale2, grad_loss, grad_pred);
    return hipGetLastError();
}

}  // namespace spf
