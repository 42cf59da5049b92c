// Microbenchmark (debug tooling): the shader clock the chip HOLDS under MFMA-dense work, measured inside the kernel as
// delta(s_memtime) / delta(s_memrealtime) x 100 MHz (guide MI355X_MICROARCH.md, "DVFS give-back" item 6).  DESIGN.md prices
// the MFMA-bound kernels against this clock (2.5 PFLOP/s is quoted at 2.4 GHz).
//   mode 0  bf16 MFMA 16x16x32 loop, operands in registers (random data), 1 wave per SIMD
//   mode 1  the same with every operand re-read from LDS by ds_read_b128 (the GEMM / attention inner-loop shape)
//   mode 2  the same, 3 waves per SIMD (the GEMM kernels' occupancy)
//   mode 3  HBM streaming copy (no MFMA)
// Two seconds of back-to-back launches warm the chip into its steady state, then one stamped launch is read back.
// Build: hipcc --offload-arch=gfx950 -O3 -o clock_probe clock_probe.hip      Run on the GPU box: ./clock_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int MODE>
__global__ __launch_bounds__(256) void probe(const float* __restrict__ seed, int iters, unsigned long long* stamps, float* sink,
                                             float4* copy_dst, const float4* copy_src, size_t copy_n) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[32768];
    const int tid = threadIdx.x, lane = tid & 63;
    unsigned long long c0 = 0, r0 = 0, c1 = 0, r1 = 0;
    if constexpr (MODE == 3) {
        c0 = __builtin_amdgcn_s_memtime();
        r0 = __builtin_amdgcn_s_memrealtime();
        float4 acc = {0, 0, 0, 0};
        for (size_t i = (size_t)blockIdx.x * 256 + tid; i < copy_n; i += (size_t)gridDim.x * 256) {
            const float4 v = copy_src[i];
            copy_dst[i] = v;
            acc.x += v.x;
        }
        c1 = __builtin_amdgcn_s_memtime();
        r1 = __builtin_amdgcn_s_memrealtime();
        if (acc.x == 1234.5f) sink[0] = acc.x;
    } else {
        bf16x8 a[4], b[4];
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 8; ++j) {
                a[i][j] = (__bf16)seed[(tid * 64 + i * 8 + j) & 4095];
                b[i][j] = (__bf16)seed[(tid * 64 + 32 + i * 8 + j) & 4095];
            }
        for (int i = tid; i < 32768 / 16; i += 256) ((bf16x8*)lds)[i] = a[i & 3];
        __syncthreads();
        f32x4 acc[4][4];
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        c0 = __builtin_amdgcn_s_memtime();
        r0 = __builtin_amdgcn_s_memrealtime();
        for (int it = 0; it < iters; ++it) {
            if constexpr (MODE >= 1) {
                const unsigned char* p = lds + ((it & 7) * 4096) + lane * 16;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    a[i] = *(const bf16x8*)(p + i * 1024);
                    b[i] = *(const bf16x8*)(p + ((i + 1) & 3) * 1024);
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        c1 = __builtin_amdgcn_s_memtime();
        r1 = __builtin_amdgcn_s_memrealtime();
        float s = 0.f;
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][3];
        if (s == 1234.5f) sink[0] = s;
    }
    if (tid == 0) {
        stamps[blockIdx.x * 2] = c1 - c0;
        stamps[blockIdx.x * 2 + 1] = r1 - r0;
    }
}

template <int MODE>
static void run(const char* name, int blocks, int iters, const float* seed, unsigned long long* stamps, float* sink, float4* dst, const float4* src,
                size_t n) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float ms = 0.f, total = 0.f;
    int launches = 0;
    while (total < 2000.f) {                       // >= 2 s of back-to-back launches on random data
        hipEventRecord(e0);
        for (int k = 0; k < 20; ++k) probe<MODE><<<blocks, 256>>>(seed, iters, stamps, sink, dst, src, n);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        total += ms;
        launches += 20;
    }
    probe<MODE><<<blocks, 256>>>(seed, iters, stamps, sink, dst, src, n);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks * 2);
    hipMemcpy(h.data(), stamps, blocks * 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    std::vector<double> ghz;
    for (int b = 0; b < blocks; ++b)
        if (h[2 * b + 1]) ghz.push_back((double)h[2 * b] / (double)h[2 * b + 1] * 0.1);   // s_memrealtime ticks at 100 MHz
    std::sort(ghz.begin(), ghz.end());
    const double per_launch_ms = total / launches;
    double tf = 0.0;
    if (MODE < 3) tf = (double)blocks * 4 * iters * 16 * (2.0 * 16 * 16 * 32) / (per_launch_ms * 1e-3) / 1e12;
    printf("%-52s clock p10 %.3f  median %.3f  p90 %.3f GHz   %.3f ms/launch", name, ghz[ghz.size() / 10], ghz[ghz.size() / 2], ghz[ghz.size() * 9 / 10],
           per_launch_ms);
    if (MODE < 3) printf("   %.0f TFLOP/s", tf);
    else printf("   %.2f TB/s (read + write)", 2.0 * n * 16 / (per_launch_ms * 1e-3) / 1e12);
    printf("\n");
}

int main() {
    float* seed;
    hipMalloc(&seed, 4096 * 4);
    std::vector<float> hs(4096);
    srand(7);
    for (auto& v : hs) v = (float)rand() / RAND_MAX * 2.f - 1.f;
    hipMemcpy(seed, hs.data(), 4096 * 4, hipMemcpyHostToDevice);
    unsigned long long* stamps;
    hipMalloc(&stamps, 4096 * 2 * 8);
    float* sink;
    hipMalloc(&sink, 4);
    const size_t n = (size_t)1 << 26;              // 1 GiB of float4 each way
    float4 *src, *dst;
    hipMalloc(&src, n * 16);
    hipMalloc(&dst, n * 16);
    hipMemset(src, 1, n * 16);
    run<0>("bf16 MFMA 16x16x32, registers, 1 wave/SIMD", 256, 20000, seed, stamps, sink, dst, src, n);
    run<1>("bf16 MFMA + ds_read_b128 operands, 1 wave/SIMD", 256, 20000, seed, stamps, sink, dst, src, n);
    run<2>("bf16 MFMA + ds_read_b128 operands, 3 waves/SIMD", 768, 20000, seed, stamps, sink, dst, src, n);
    run<3>("HBM streaming copy, 2048 blocks", 2048, 0, seed, stamps, sink, dst, src, n);
    return 0;
}
