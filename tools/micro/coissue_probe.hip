// Microbenchmark (debug tooling): do VALU instructions hide behind MFMAs on one SIMD?
//   test 1  ONE wave per SIMD: loop of { 1 x v_mfma_f32_32x32x16_bf16 (independent accumulators) + k x v_fma_f32 }, k = 0..10:
//           cycles per iteration (s_memtime) -- the guide's constants say ~32 until the VALU issue costs (4 cycles each) + the
//           MFMA's own 8 exceed 32.
//   test 2  TWO waves per SIMD (512-thread block), even waves run MFMAs only, odd waves VALU only (same instruction counts as
//           one k = 6 iteration of test 1 per loop trip): cycles per trip of each kind alone and together.
// Build: hipcc --offload-arch=gfx950 -O3 -o coissue_probe coissue_probe.hip      Run on the GPU box: ./coissue_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int K>
__global__ __launch_bounds__(256) void mix(int iters, unsigned long long* out, float* sink) {
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(0.01f * (threadIdx.x + j)); b[j] = (__bf16)(0.02f * (threadIdx.x - j)); }
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    float v[12];
    for (int i = 0; i < 12; ++i) v[i] = 0.001f * (threadIdx.x + i);
    const float m = 0.999f, ad = 0.0001f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[u], 0, 0, 0);
#pragma unroll
            for (int k = 0; k < K; ++k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[k]) : "v"(m), "v"(ad));
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][7];
    for (int i = 0; i < 12; ++i) s += v[i];
    if (s == 1234.5f) sink[0] = s;
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
}

// mode 0: even waves MFMA, odd waves VALU; 1: all waves MFMA-role only on even (odd idle); 2: odd VALU only (even idle)
__global__ __launch_bounds__(512) void split(int iters, int mode, unsigned long long* out, float* sink) {
    const int wave = threadIdx.x >> 6;
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(0.01f * (threadIdx.x + j)); b[j] = (__bf16)(0.02f * (threadIdx.x - j)); }
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    float v[6];
    for (int i = 0; i < 6; ++i) v[i] = 0.001f * (threadIdx.x + i);
    const float m = 0.999f, ad = 0.0001f;
    // waves 0-3 land on SIMDs 0-3, waves 4-7 are their partners (guide: split roles by wave number >= 4)
    const bool mfma_role = wave < 4, active = mode == 0 || (mode == 1 && mfma_role) || (mode == 2 && !mfma_role);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (active) {
        if (mfma_role) {
            for (int it = 0; it < iters; ++it)
#pragma unroll
                for (int u = 0; u < 4; ++u) acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[u], 0, 0, 0);
        } else {
            for (int it = 0; it < iters; ++it)
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int k = 0; k < 6; ++k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[k]) : "v"(m), "v"(ad));
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][7];
    for (int i = 0; i < 6; ++i) s += v[i];
    if (s == 1234.5f) sink[0] = s;
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int K>
static void run_mix(unsigned long long* d, float* sink) {
    const int iters = 2000;
    mix<K><<<256, 256>>>(iters, d, sink);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(256);
    hipMemcpy(h.data(), d, 256 * 8, hipMemcpyDeviceToHost);
    double s = 0;
    for (auto x : h) s += (double)x;
    printf("one wave/SIMD: 1 MFMA + %2d v_fma per step: %6.1f cycles per step\n", K, s / 256 / (iters * 4.0));
}

int main() {
    unsigned long long* d;
    float* sink;
    hipMalloc(&d, 256 * 8 * 8);
    hipMalloc(&sink, 4);
    run_mix<0>(d, sink); run_mix<1>(d, sink); run_mix<2>(d, sink); run_mix<3>(d, sink); run_mix<4>(d, sink); run_mix<5>(d, sink);
    run_mix<6>(d, sink); run_mix<7>(d, sink); run_mix<8>(d, sink); run_mix<10>(d, sink); run_mix<12>(d, sink);
    const char* names[3] = {"MFMA waves + VALU waves together", "MFMA waves alone", "VALU waves alone"};
    for (int mode = 0; mode < 3; ++mode) {
        const int iters = 2000;
        split<<<256, 512>>>(iters, mode, d, sink);
        hipDeviceSynchronize();
        std::vector<unsigned long long> h(256 * 8);
        hipMemcpy(h.data(), d, 256 * 8 * 8, hipMemcpyDeviceToHost);
        double sm = 0, sv = 0;
        for (int b = 0; b < 256; ++b)
            for (int w = 0; w < 8; ++w) (w < 4 ? sm : sv) += (double)h[b * 8 + w];
        printf("two waves/SIMD, %-34s: MFMA wave %6.1f cycles per 1 MFMA, VALU wave %6.1f cycles per 6 v_fma\n", names[mode],
               sm / 1024 / (iters * 4.0), sv / 1024 / (iters * 4.0));
    }
    return 0;
}
