// Microbenchmark (debug tooling): per-CU throughput of LDS-DMA (global_load_lds_dwordx4) and of plain global_load_dwordx4
// from an L2-resident table, at 1..3 blocks of 4 waves per CU.  Build: hipcc --offload-arch=gfx950 -O3 -o dma_rate dma_rate.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

template <int MODE>   // 0: LDS-DMA, 1: register loads
__global__ __launch_bounds__(256) void k(const unsigned char* src, size_t span, int iters, unsigned long long* out, float* sink) {
    extern __shared__ unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // each block walks its own 32 KiB "stages" round a span that stays in L2
    size_t off = ((size_t)blockIdx.x * 32768) % span;
    unsigned long long t0 = __builtin_readcyclecounter();
    float4 accv = {0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const unsigned char* p = src + off + (size_t)(wave * 8 + j) * 1024 + lane * 16;
            if (MODE == 0) __builtin_amdgcn_global_load_lds(GLB_PTR(p), LDS_PTR(smem + (wave * 8 + j) * 1024), 16, 0, 0);
            else { float4 v = *(const float4*)p; accv.x += v.x; accv.y += v.y; accv.z += v.z; accv.w += v.w; }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        off += 32768 * 7;
        if (off >= span) off -= span;
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    if (MODE == 1 && accv.x == 12345.f) sink[0] = accv.x + accv.y + accv.z + accv.w;
}

int main(int argc, char** argv) {
    const size_t span = (argc > 1 ? atoi(argv[1]) : 2) * 1024 * 1024;   // per-XCD L2 is 4 MiB
    unsigned char* src; hipMalloc(&src, span + 65536); hipMemset(src, 1, span + 65536);
    unsigned long long* out; hipMalloc(&out, 8 * 4096);
    float* sink; hipMalloc(&sink, 4);
    const int iters = 200;
    for (int mode = 0; mode < 2; ++mode)
        for (int bpc = 1; bpc <= 3; ++bpc) {
            const int grid = 256 * bpc;
            for (int rep = 0; rep < 2; ++rep) {
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(256), 32768, 0, src, span, iters, out, sink);
                else hipLaunchKernelGGL(k<1>, dim3(grid), dim3(256), 32768, 0, src, span, iters, out, sink);
                hipDeviceSynchronize();
            }
            static unsigned long long h[4096];
            hipMemcpy(h, out, 8 * grid, hipMemcpyDeviceToHost);
            double s = 0; for (int i = 0; i < grid; ++i) s += h[i];
            const double clk = s / grid;            // per block, all blocks concurrent
            printf("%s span %zu MiB, %d block(s)/CU: %.0f clk per 32 KiB stage per block -> %.1f B/clk/CU\n",
                   mode == 0 ? "LDS-DMA " : "reg load", span >> 20, bpc, clk / iters, 32768.0 * bpc * iters / clk);
        }
    return 0;
}
