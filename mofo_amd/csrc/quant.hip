// OCP e4m3 quantisation for the fp8 forward GEMMs (BASELINE configs[4]: "fp8 MFMA attention/MLP"; mofo_gemm op NT_FP8).
// Per-TENSOR scaling: q = sat_e4m3(x * 448 / amax(x)); the GEMM multiplies its f32 accumulators by the two inverse scales.
//   weights      the flat bf16 shadow (runtime.FlatStore) -> a flat e4m3 shadow, one scale per weight matrix: a per-1024-chunk
//                table names each chunk's matrix ("segment", -1 = not a GEMM weight), so two launches cover every matrix;
//   activations  written by the LayerNorm forward itself (layernorm.hip) with DELAYED scaling: the scale of step t comes from
//                the amax the kernel saw in step t-1; mofo_fp8_update_scales turns the collected amax values into scales.
// The reference has no fp8 path (its low-precision site is the fp16 autocast at engine_for_pretraining.py:65).
#include "common.h"
#include "../../include/mofo_hip.h"

namespace {

constexpr float E4M3_MAX = 448.0f;

// Each block walks a contiguous RUN of 1024-element chunks and keeps a running |x| max while the segment stays the same: one
// atomicMax per (block, segment) instead of one per chunk (318 M parameters = 310 k chunks: at one atomic per chunk the same
// few addresses took 1.5 ms; non-negative floats order as uints).
__global__ __launch_bounds__(256) void amax_segments_kernel(const bf16_t* __restrict__ x, const short* __restrict__ chunk_seg, int nchunks,
                                                            int per_block, float* __restrict__ amax) {
    __shared__ float red[4];
    const int c0 = blockIdx.x * per_block, c1 = min(nchunks, c0 + per_block);
    int cur = -1;
    float m = 0.f;
    auto flush = [&]() {
        if (cur < 0) return;
        const float wm = wave_max(m);
        __syncthreads();
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = wm;
        __syncthreads();
        if (threadIdx.x == 0) atomicMax((unsigned*)(amax + cur), __float_as_uint(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]))));
    };
    for (int c = c0; c < c1; ++c) {
        const int seg = chunk_seg[c];            // block-uniform
        if (seg != cur) {
            flush();
            cur = seg;
            m = 0.f;
        }
        if (seg < 0) continue;
        const u32x2 v = *(const u32x2*)(x + (size_t)c * 1024 + threadIdx.x * 4);
        m = fmaxf(m, fmaxf(fmaxf(fabsf(bf16lo_to_f32(v[0])), fabsf(bf16hi_to_f32(v[0]))), fmaxf(fabsf(bf16lo_to_f32(v[1])), fabsf(bf16hi_to_f32(v[1])))));
    }
    flush();
}

__global__ __launch_bounds__(256) void quant_segments_kernel(const bf16_t* __restrict__ x, const short* __restrict__ chunk_seg,
                                                             const float* __restrict__ amax, uint8_t* __restrict__ out,
                                                             float* __restrict__ scale_inv) {
    const int c = blockIdx.x;
    const int seg = chunk_seg[c];
    if (seg < 0) return;
    const float am = amax[seg];
    const float s = am > 0.f ? E4M3_MAX / am : 1.0f;
    if (threadIdx.x == 0 && (c == 0 || chunk_seg[c - 1] != seg)) scale_inv[seg] = am > 0.f ? am / E4M3_MAX : 1.0f;
    const u32x2 v = *(const u32x2*)(x + (size_t)c * 1024 + threadIdx.x * 4);
    *(uint32_t*)(out + (size_t)c * 1024 + threadIdx.x * 4) =
        pack4_e4m3(bf16lo_to_f32(v[0]) * s, bf16hi_to_f32(v[0]) * s, bf16lo_to_f32(v[1]) * s, bf16hi_to_f32(v[1]) * s);
}

// plain tensor: q = sat(x * scale[0]) (bf16 in), optional amax of the input for the caller's next scale
__global__ __launch_bounds__(256) void quant_bf16_kernel(const bf16_t* __restrict__ x, long long n4, const float* __restrict__ scale,
                                                         uint8_t* __restrict__ out, float* __restrict__ amax_out) {
    const float s = scale[0];
    float m = 0.f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        const u32x2 v = *(const u32x2*)(x + i * 4);
        const float a = bf16lo_to_f32(v[0]), b = bf16hi_to_f32(v[0]), c = bf16lo_to_f32(v[1]), d = bf16hi_to_f32(v[1]);
        m = fmaxf(m, fmaxf(fmaxf(fabsf(a), fabsf(b)), fmaxf(fabsf(c), fabsf(d))));
        *(uint32_t*)(out + i * 4) = pack4_e4m3(a * s, b * s, c * s, d * s);
    }
    if (amax_out) {
        m = wave_max(m);
        if ((threadIdx.x & 63) == 0) atomicMax((unsigned*)amax_out, __float_as_uint(m));
    }
}

// scales[i] = (448 / (amax[i] * margin), its inverse); amax[i] is cleared for the next step.  A site that saw nothing keeps its scale.
// amax is [n][MOFO_FP8_AMAX_STRIPES]: one wave per site folds its stripes (the LayerNorm blocks spread their atomics over them)
__global__ void update_scales_kernel(float* __restrict__ amax, float* __restrict__ scales, int n, float margin) {
    const int i = blockIdx.x, lane = threadIdx.x;
    float* row = amax + (size_t)i * MOFO_FP8_AMAX_STRIPES;
    float a = 0.f;
    for (int k = lane; k < MOFO_FP8_AMAX_STRIPES; k += 64) {
        a = fmaxf(a, row[k]);
        row[k] = 0.f;
    }
    a = wave_max(a);
    if (lane == 0 && a > 0.f && a < 3.0e38f) {
        const float s = E4M3_MAX / (a * margin);
        scales[2 * i] = s;
        scales[2 * i + 1] = 1.0f / s;
    }
}

}  // namespace

extern "C" int mofo_fp8_quantize_segments(const void* x_bf16, long long n, const short* chunk_seg, int nseg, float* amax_ws,
                                          void* out_e4m3, float* scale_inv, void* stream) {
    if (!x_bf16 || !chunk_seg || !amax_ws || !out_e4m3 || !scale_inv) MOFO_FAIL(MOFO_EINVAL, "mofo_fp8_quantize_segments: null pointer");
    if (n <= 0 || n % 1024 || nseg <= 0 || nseg > 32767) MOFO_FAIL(MOFO_EINVAL, "mofo_fp8_quantize_segments: n must be a positive multiple of 1024, 1 <= nseg <= 32767");
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(amax_ws, 0, (size_t)nseg * sizeof(float), s) != hipSuccess) MOFO_FAIL(MOFO_ERUNTIME, "mofo_fp8_quantize_segments: memset failed");
    const unsigned chunks = (unsigned)(n / 1024);
    const int per_block = (int)((chunks + 4095) / 4096);
    hipLaunchKernelGGL(amax_segments_kernel, dim3((chunks + per_block - 1) / per_block), dim3(256), 0, s, (const bf16_t*)x_bf16, chunk_seg, (int)chunks,
                       per_block, amax_ws);
    hipLaunchKernelGGL(quant_segments_kernel, dim3(chunks), dim3(256), 0, s, (const bf16_t*)x_bf16, chunk_seg, (const float*)amax_ws,
                       (uint8_t*)out_e4m3, scale_inv);
    MOFO_CHECK_LAUNCH("mofo_fp8_quantize_segments");
    return MOFO_OK;
}

extern "C" int mofo_fp8_quantize_bf16(const void* x_bf16, long long n, const float* scale, void* out_e4m3, float* amax_out, void* stream) {
    if (!x_bf16 || !scale || !out_e4m3) MOFO_FAIL(MOFO_EINVAL, "mofo_fp8_quantize_bf16: null pointer");
    if (n <= 0 || n % 4) MOFO_FAIL(MOFO_EINVAL, "mofo_fp8_quantize_bf16: n must be a positive multiple of 4");
    const long long n4 = n / 4;
    long long blocks = (n4 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(quant_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x_bf16, n4, scale,
                       (uint8_t*)out_e4m3, amax_out);
    MOFO_CHECK_LAUNCH("mofo_fp8_quantize_bf16");
    return MOFO_OK;
}

extern "C" int mofo_fp8_update_scales(float* amax, float* scales, int n, float margin, void* stream) {
    if (!amax || !scales || n <= 0 || !(margin > 0.f)) MOFO_FAIL(MOFO_EINVAL, "mofo_fp8_update_scales: bad arguments");
    hipLaunchKernelGGL(update_scales_kernel, dim3(n), dim3(64), 0, (hipStream_t)stream, amax, scales, n, margin);
    MOFO_CHECK_LAUNCH("mofo_fp8_update_scales");
    return MOFO_OK;
}
