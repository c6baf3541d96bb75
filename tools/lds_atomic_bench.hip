// development: LDS cycles per wave-instruction of ds_or_b32 (no return) against ds_write_b32 / ds_read_b32 at random
// (pixel-like) addresses.   hipcc --offload-arch=gfx950 -O3 tools/lds_atomic_bench.hip -o /tmp/lab && /tmp/lab
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
template <int MODE>
__global__ __launch_bounds__(256) void k(unsigned long long* out, int iters, int spread) {
    __shared__ uint32_t s[2048];
    for (int i = threadIdx.x; i < 2048; i += 256) s[i] = 0;
    __syncthreads();
    uint32_t x = threadIdx.x * 2654435761u + blockIdx.x;
    uint32_t acc = 0;
    const int word = (threadIdx.x >> 5) * 256;          // a half-wave owns its own 256-word column block
    const long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
        x = x * 1664525u + 1013904223u;
        const uint32_t a = word + ((x >> 16) % (uint32_t)spread);
        if (MODE == 0) atomicOr(&s[a], 1u << (threadIdx.x & 31));
        if (MODE == 1) s[a] = x;
        if (MODE == 2) acc += s[a];
        if (MODE == 3) acc += atomicOr(&s[a], 1u << (threadIdx.x & 31));
    }
    __syncthreads();
    const long long t1 = clock64();
    if (acc == 0x12345u) s[0] = acc;
    if (threadIdx.x == 0) out[blockIdx.x] = (unsigned long long)(t1 - t0) + (s[1] == 77u ? 1 : 0);
}
template <int MODE>
void run(const char* name, int blocks_per_cu, int spread) {
    const int B = 256 * blocks_per_cu, iters = 2000;
    unsigned long long* d; hipMalloc(&d, B * 8);
    k<MODE><<<B, 256>>>(d, iters, spread);
    hipDeviceSynchronize();
    static unsigned long long h[4096];
    hipMemcpy(h, d, B * 8, hipMemcpyDeviceToHost);
    double s = 0; for (int i = 0; i < B; ++i) s += (double)h[i];
    // per CU: blocks_per_cu blocks x 4 waves issue `iters` DS ops each during ~mean cycles
    printf("%-28s blocks/CU %d spread %3d: %.1f cycles per wave-op per CU (block time %.0f cyc)\n", name, blocks_per_cu, spread,
           s / B / iters / (4.0 * blocks_per_cu), s / B);
    hipFree(d);
}
int main() {
    for (int bpc : {1, 4}) for (int spread : {256, 32}) {
        run<0>("ds_or_b32 (no return)", bpc, spread);
        run<3>("ds_or_rtn_b32", bpc, spread);
        run<1>("ds_write_b32", bpc, spread);
        run<2>("ds_read_b32", bpc, spread);
    }
    return 0;
}
