// development: which SIMD does wave k of a 256-thread block land on?   hipcc --offload-arch=gfx950 tools/simd_map.hip -o /tmp/simd_map && /tmp/simd_map
// (HW_REG_HW_ID, gfx9 layout: wave_id 3:0, simd_id 5:4, pipe_id 7:6, cu_id 11:8, sh_id 12, se_id 15:13)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__global__ __launch_bounds__(256) void probe(uint32_t* out, int spin) {
    __shared__ float s_pad[7936];                       // ~31 KB: five blocks per CU, like the kernel in question
    const uint32_t id = __builtin_amdgcn_s_getreg((31 << 11) | 4);
    float x = (float)threadIdx.x;
    for (int i = 0; i < spin; ++i) x = x * 1.0001f + 0.5f;      // stay resident for a while
    s_pad[threadIdx.x] = x;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = id | (s_pad[255 - threadIdx.x] == 123.f ? 1u << 31 : 0u);
}
int main() {
    const int B = 8192;
    uint32_t* d; hipMalloc(&d, B * 4 * sizeof(uint32_t));
    probe<<<B, 256>>>(d, 20000);
    hipDeviceSynchronize();
    static uint32_t h[B * 4];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    long hist[4][4] = {};
    for (int b = 0; b < B; ++b) for (int w = 0; w < 4; ++w) hist[w][(h[b * 4 + w] >> 4) & 3]++;
    printf("rows: wave index in block; columns: SIMD id\n");
    for (int w = 0; w < 4; ++w) printf("wave %d: %6ld %6ld %6ld %6ld\n", w, hist[w][0], hist[w][1], hist[w][2], hist[w][3]);
    // blocks per (se, sh, cu) for the first 40 blocks: which blocks share a CU
    for (int b = 0; b < 48; ++b) printf("block %2d: cu %2u se %u sh %u simd(w0..3) %u%u%u%u\n", b, (h[b * 4] >> 8) & 15, (h[b * 4] >> 13) & 7, (h[b * 4] >> 12) & 1,
        (h[b * 4] >> 4) & 3, (h[b * 4 + 1] >> 4) & 3, (h[b * 4 + 2] >> 4) & 3, (h[b * 4 + 3] >> 4) & 3);
    return 0;
}
