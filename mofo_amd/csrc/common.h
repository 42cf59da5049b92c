// Shared device/host helpers for libmofo_hip.so (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define MOFO_OK 0
#define MOFO_EINVAL (-1)       // bad argument (null pointer, non-positive size)
#define MOFO_EUNSUPPORTED (-2) // shape outside what the kernels are built for
#define MOFO_ELAUNCH (-3)      // hipGetLastError() after a launch
#define MOFO_ERUNTIME (-4)     // other HIP runtime failure

typedef uint16_t bf16_t;  // storage type for bf16 in global memory / C-ABI
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

// host side -------------------------------------------------------------------------------------
void mofo_set_error(const char* fmt, ...);
#define MOFO_FAIL(code, ...)          \
    do {                              \
        mofo_set_error(__VA_ARGS__);  \
        return (code);                \
    } while (0)
#define MOFO_CHECK_LAUNCH(name)                                                          \
    do {                                                                                 \
        hipError_t e__ = hipGetLastError();                                              \
        if (e__ != hipSuccess) MOFO_FAIL(MOFO_ELAUNCH, "%s: %s", name, hipGetErrorString(e__)); \
    } while (0)

// device side -----------------------------------------------------------------------------------
__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
__device__ __forceinline__ float bf16lo_to_f32(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf16hi_to_f32(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }

// round-to-nearest-even f32 -> bf16 via the hardware cast (v_cvt_pk_bf16_f32 at -O3; keeps NaN a NaN)
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
    __bf16 b = (__bf16)f;
    return __builtin_bit_cast(bf16_t, b);
}
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
    bf16x2 v;
    v[0] = (__bf16)lo;
    v[1] = (__bf16)hi;
    return __builtin_bit_cast(uint32_t, v);
}

// four f32 -> four OCP e4m3 bytes (saturating at +-448; byte k = argument k)
__device__ __forceinline__ uint32_t pack4_e4m3(float a, float b, float c, float d) {
    a = __builtin_amdgcn_fmed3f(a, -448.0f, 448.0f);
    b = __builtin_amdgcn_fmed3f(b, -448.0f, 448.0f);
    c = __builtin_amdgcn_fmed3f(c, -448.0f, 448.0f);
    d = __builtin_amdgcn_fmed3f(d, -448.0f, 448.0f);
    int r = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
    r = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, r, true);
    return (uint32_t)r;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// erf-form GELU and its derivative (nn.GELU() default, modeling_finetune.py:35), two elements at a time in packed fp32
// (v_pk_fma_f32 / v_pk_mul_f32: two lanes' worth of FMA per issue slot).  The Gaussian cdf is an odd minimax polynomial
// 0.5 + x * P(x^2) on |x| <= 4 (degree 8 in x^2, |error| <= 4e-6 inside the clamp, <= 3.2e-5 = 1 - cdf(4) beyond it --
// two orders below the bf16 rounding of the stored result); no transcendental and no division, where the previous
// Abramowitz-Stegun form spent a v_exp + v_rcp (quarter rate each) per element and made the K=384 GELU epilogues
// VALU-bound.  The derivative adds x * pdf(x) with one v_exp per element.
typedef __attribute__((ext_vector_type(2))) float f32x2;
__device__ __forceinline__ f32x2 gelu_cdf2(f32x2 xc) {   // xc already clamped to [-4, 4]
    const f32x2 u = xc * xc;
    f32x2 a = {8.063485893e-11f, 8.063485893e-11f};
    a = __builtin_elementwise_fma(a, u, (f32x2){-7.003511440e-09f, -7.003511440e-09f});
    a = __builtin_elementwise_fma(a, u, (f32x2){2.716168348e-07f, 2.716168348e-07f});
    a = __builtin_elementwise_fma(a, u, (f32x2){-6.295016881e-06f, -6.295016881e-06f});
    a = __builtin_elementwise_fma(a, u, (f32x2){9.890821982e-05f, 9.890821982e-05f});
    a = __builtin_elementwise_fma(a, u, (f32x2){-1.133922770e-03f, -1.133922770e-03f});
    a = __builtin_elementwise_fma(a, u, (f32x2){9.877478530e-03f, 9.877478530e-03f});
    a = __builtin_elementwise_fma(a, u, (f32x2){-6.641059700e-02f, -6.641059700e-02f});
    a = __builtin_elementwise_fma(a, u, (f32x2){3.989227102e-01f, 3.989227102e-01f});
    return __builtin_elementwise_fma(xc, a, (f32x2){0.5f, 0.5f});
}
__device__ __forceinline__ f32x2 gelu_erf2(f32x2 x) {
    // the factor in front of the cdf is clamped from below as well: x * cdf(-4) would grow linearly where gelu -> 0
    const f32x2 xl = {fmaxf(x[0], -4.f), fmaxf(x[1], -4.f)};
    const f32x2 xc = {fminf(xl[0], 4.f), fminf(xl[1], 4.f)};
    return xl * gelu_cdf2(xc);
}
__device__ __forceinline__ f32x2 dgelu_erf2(f32x2 x) {
    const f32x2 xc = {__builtin_amdgcn_fmed3f(x[0], -4.f, 4.f), __builtin_amdgcn_fmed3f(x[1], -4.f, 4.f)};
    const f32x2 t = xc * xc * (f32x2){-0.72134752044448170f, -0.72134752044448170f};   // -x^2/2 * log2(e)
    const f32x2 e = {__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])};
    return __builtin_elementwise_fma(xc * (f32x2){0.39894228040143268f, 0.39894228040143268f}, e, gelu_cdf2(xc));
}
__device__ __forceinline__ float gelu_erf(float x) { return gelu_erf2((f32x2){x, x})[0]; }
__device__ __forceinline__ float dgelu_erf(float x) { return dgelu_erf2((f32x2){x, x})[0]; }

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
