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

// erf-form GELU and its derivative (nn.GELU() default, modeling_finetune.py:35).  erf is evaluated with the
// Abramowitz-Stegun 7.1.26 rational form (|error| <= 1.5e-7, far below the bf16 rounding of the result): one v_exp,
// one v_rcp and five FMAs instead of libm's erff -- the GELU epilogues were VALU-bound on erff.  The same
// exp(-x^2/2) serves the Gaussian pdf of the derivative.
__device__ __forceinline__ void gelu_parts(float x, float& cdf, float& e) {
    const float z = fabsf(x) * 0.70710678118654752f;
    e = __builtin_amdgcn_exp2f(-1.4426950408889634f * z * z);   // exp(-x^2/2)
    const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * z);
    const float poly = ((((1.061405429f * t - 1.453152027f) * t + 1.421413741f) * t - 0.284496736f) * t + 0.254829592f) * t;
    const float erf_abs = 1.0f - poly * e;
    cdf = 0.5f * (1.0f + copysignf(erf_abs, x));
}
__device__ __forceinline__ float gelu_erf(float x) {
    float cdf, e;
    gelu_parts(x, cdf, e);
    return x * cdf;
}
__device__ __forceinline__ float dgelu_erf(float x) {
    float cdf, e;
    gelu_parts(x, cdf, e);
    return cdf + x * 0.39894228040143268f * e;
}

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
