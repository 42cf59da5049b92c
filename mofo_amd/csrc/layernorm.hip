// LayerNorm forward / backward for the fp32 residual stream (gfx950).  HBM-bound: one wave per row, 16-B loads,
// statistics in fp32 by wave shuffles.  Replaces nn.LayerNorm(eps=1e-6) at modeling_finetune.py:200,206,218-219 and
// modeling_pretrain.py:51,95,123,157 and its autograd backward.
#include "common.h"
#include "../../include/mofo_hip.h"

namespace {

constexpr int MAX_IT = 4;  // D <= 1024

__device__ __forceinline__ size_t map_row(int r, int rows_in, int rows_out, int row_off) {
    return (size_t)(r / rows_in) * rows_out + row_off + (r % rows_in);
}

// XB: the residual stream is bf16 (the decoder, runtime.py) instead of f32: 8-byte loads of four elements per lane.
// Q8: the normalised row also goes out as OCP e4m3 (the A operand of the fp8 forward GEMMs, gemm.hip NT_FP8), multiplied by
// qscale[0]; the maximum |y| over ALL rows goes to amax_out's 1024 stripes (delayed scaling: it sets the NEXT step's scale).  The running maximum
// is read before the atomic, so only the waves that raise it issue one (an atomic per block on one address would cost more than the
// LayerNorm itself; round 2 sampled every 64th block instead and could miss outlier rows).
template <int NIT, bool XB, bool Q8 = false>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const void* __restrict__ xv_, int ldx, const float* __restrict__ w,
                                                      const float* __restrict__ b, float eps, int M, int D, int rows_in,
                                                      int rows_out, int row_off, bf16_t* __restrict__ y, int ldy,
                                                      float* __restrict__ mean, float* __restrict__ rstd,
                                                      uint8_t* __restrict__ y8 = nullptr, int ldy8 = 0,
                                                      const float* __restrict__ qscale = nullptr, float* __restrict__ amax_out = nullptr) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = blockIdx.x * 4 + wave;
    if (r >= M) return;
    const size_t xrow = map_row(r, rows_in, rows_out, row_off) * ldx;
    f32x4 v[NIT];
    float s = 0.f;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int c = (it * 64 + lane) * 4;
        v[it] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (c < D) {
            if constexpr (XB) {
                const u32x2 t = *(const u32x2*)((const bf16_t*)xv_ + xrow + c);
                v[it] = f32x4{bf16lo_to_f32(t[0]), bf16hi_to_f32(t[0]), bf16lo_to_f32(t[1]), bf16hi_to_f32(t[1])};
            } else {
                v[it] = *(const f32x4*)((const float*)xv_ + xrow + c);
            }
        }
        s += v[it][0] + v[it][1] + v[it][2] + v[it][3];
    }
    const float mu = wave_sum(s) / D;
    float q = 0.f;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int c = (it * 64 + lane) * 4;
        if (c < D) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float d = v[it][e] - mu;
                q += d * d;
            }
        }
    }
    const float rs = rsqrtf(wave_sum(q) / D + eps);
    if (lane == 0) {
        mean[r] = mu;
        rstd[r] = rs;
    }
    bf16_t* yr = y + (size_t)r * ldy;
    float qs = 1.f, am = 0.f;
    if constexpr (Q8) qs = qscale[0];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int c = (it * 64 + lane) * 4;
        if (c < D) {
            const f32x4 wv = *(const f32x4*)(w + c);
            const f32x4 bv = *(const f32x4*)(b + c);
            float o[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (v[it][e] - mu) * rs * wv[e] + bv[e];
            u32x2 pk = {pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3])};
            *(u32x2*)(yr + c) = pk;
            if constexpr (Q8) {
                // quantise what the bf16 consumers see (the rounded values), so both forms of the activation agree
                const float q0 = bf16lo_to_f32(pk[0]), q1 = bf16hi_to_f32(pk[0]), q2 = bf16lo_to_f32(pk[1]), q3 = bf16hi_to_f32(pk[1]);
                am = fmaxf(am, fmaxf(fmaxf(fabsf(q0), fabsf(q1)), fmaxf(fabsf(q2), fabsf(q3))));
                int rq = __builtin_amdgcn_cvt_pk_fp8_f32(__builtin_amdgcn_fmed3f(q0 * qs, -448.f, 448.f), __builtin_amdgcn_fmed3f(q1 * qs, -448.f, 448.f), 0, false);
                rq = __builtin_amdgcn_cvt_pk_fp8_f32(__builtin_amdgcn_fmed3f(q2 * qs, -448.f, 448.f), __builtin_amdgcn_fmed3f(q3 * qs, -448.f, 448.f), rq, true);
                *(uint32_t*)(y8 + (size_t)r * ldy8 + c) = (uint32_t)rq;
            }
        }
    }
    if constexpr (Q8) {
        // EVERY row contributes to the next step's scale (a sample of the rows can miss the few outlier rows a ViT's LayerNorm
        // outputs have, and values beyond the scale are silently clamped to +-448).  The running maximum is read first: once the
        // first rows have set it, almost no wave issues the atomic (a stale read only costs a redundant atomic).
        // amax_out is MOFO_FP8_AMAX_STRIPES (1024) floats: the row waves spread over the stripes (same-address atomics serialise at
        // ~0.1-0.3 us each, and the thousands of waves that start together all read a zero maximum: with 64 stripes a 10 240-row
        // LayerNorm took 60 instead of 16 us), mofo_fp8_update_scales folds the stripes
        am = wave_max(am);
        float* slot = amax_out + ((blockIdx.x * 4 + (threadIdx.x >> 6)) & (MOFO_FP8_AMAX_STRIPES - 1));
        if (lane == 0 && am > *(volatile const float*)slot) atomicMax((unsigned*)slot, __float_as_uint(am));
    }
}

// bf16 input with D <= 512 (the decoder's stream): one 16-B load per lane and row, TWO rows per wave in flight (the general
// kernel's 8-B loads and one row per wave left ~24 KB in flight per CU: 3.3 TB/s on the 77 MB of a decoder LayerNorm).
// Q8: also the e4m3 copy and the launch's max|y|, as ln_fwd_kernel<.., Q8> (8 bytes per lane and row).
template <bool Q8 = false>
__global__ __launch_bounds__(256) void ln_fwd_bf16_2row_kernel(const bf16_t* __restrict__ x, int ldx, const float* __restrict__ w,
                                                                const float* __restrict__ b, float eps, int M, int D, int rows_in,
                                                                int rows_out, int row_off, bf16_t* __restrict__ y, int ldy,
                                                                float* __restrict__ mean, float* __restrict__ rstd,
                                                                uint8_t* __restrict__ y8 = nullptr, int ldy8 = 0,
                                                                const float* __restrict__ qscale = nullptr, float* __restrict__ amax_out = nullptr) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r0 = (blockIdx.x * 4 + wave) * 2;
    if (r0 >= M) return;
    const int c = lane * 8;
    const bool on = c < D;
    float v[2][8];
    float s[2] = {0.f, 0.f};
    int rr[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        rr[k] = r0 + k < M ? r0 + k : r0;
        u32x4 t = {0u, 0u, 0u, 0u};
        if (on) t = *(const u32x4*)(x + map_row(rr[k], rows_in, rows_out, row_off) * ldx + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            v[k][2 * e] = bf16lo_to_f32(t[e]);
            v[k][2 * e + 1] = bf16hi_to_f32(t[e]);
        }
    }
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int e = 0; e < 8; ++e) s[k] += v[k][e];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        s[0] += __shfl_xor(s[0], o, 64);
        s[1] += __shfl_xor(s[1], o, 64);
    }
    const float mu[2] = {s[0] / D, s[1] / D};
    float q[2] = {0.f, 0.f};
    if (on) {
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float d = v[k][e] - mu[k];
                q[k] += d * d;
            }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        q[0] += __shfl_xor(q[0], o, 64);
        q[1] += __shfl_xor(q[1], o, 64);
    }
    const float rs[2] = {rsqrtf(q[0] / D + eps), rsqrtf(q[1] / D + eps)};
    f32x4 w0 = {0.f, 0.f, 0.f, 0.f}, w1 = w0, b0 = w0, b1 = w0;
    if (on) {
        w0 = *(const f32x4*)(w + c), w1 = *(const f32x4*)(w + c + 4);
        b0 = *(const f32x4*)(b + c), b1 = *(const f32x4*)(b + c + 4);
    }
    float qs = 1.f, am = 0.f;
    if constexpr (Q8) qs = qscale[0];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        if (k == 1 && r0 + 1 >= M) break;
        if (lane == 0) {
            mean[rr[k]] = mu[k];
            rstd[rr[k]] = rs[k];
        }
        if (on) {
            float o[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                o[e] = (v[k][e] - mu[k]) * rs[k] * w0[e] + b0[e];
                o[4 + e] = (v[k][4 + e] - mu[k]) * rs[k] * w1[e] + b1[e];
            }
            const u32x4 pk = {pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]), pack_bf16x2(o[4], o[5]), pack_bf16x2(o[6], o[7])};
            *(u32x4*)(y + (size_t)rr[k] * ldy + c) = pk;
            if constexpr (Q8) {     // what the bf16 consumers see (the rounded values), so both forms of the activation agree
                float q[8];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    q[2 * e] = bf16lo_to_f32(pk[e]);
                    q[2 * e + 1] = bf16hi_to_f32(pk[e]);
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) am = fmaxf(am, fabsf(q[e]));
                *(u32x2*)(y8 + (size_t)rr[k] * ldy8 + c) = u32x2{pack4_e4m3(q[0] * qs, q[1] * qs, q[2] * qs, q[3] * qs),
                                                                 pack4_e4m3(q[4] * qs, q[5] * qs, q[6] * qs, q[7] * qs)};
            }
        }
    }
    if constexpr (Q8) {
        am = wave_max(am);
        float* slot = amax_out + ((blockIdx.x * 4 + wave) & (MOFO_FP8_AMAX_STRIPES - 1));
        if (lane == 0 && am > *(volatile const float*)slot) atomicMax((unsigned*)slot, __float_as_uint(am));
    }
}

// Two rows per wave per iteration: both rows' loads are issued before either row's shuffle reductions, so the HBM
// latency of one row hides behind the arithmetic of the other (the one-row form ran at 2-2.8 TB/s in the step).
template <int NIT, bool XB>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const bf16_t* __restrict__ dy, int lddy, const void* __restrict__ xv_,
                                                      int ldx, const float* __restrict__ w, const float* __restrict__ mean,
                                                      const float* __restrict__ rstd, const float* __restrict__ dres,
                                                      int lddres, int M, int D, int rows_in, int rows_out, int row_off,
                                                      float* __restrict__ dx, int lddx, bf16_t* __restrict__ dxb, int lddxb,
                                                      float* __restrict__ dw, float* __restrict__ db,
                                                      const bf16_t* __restrict__ dresb, int lddresb, float* __restrict__ partial,
                                                      int dres_period, int dres_skip) {
    // dres_period > 0: the residual gradient exists only for the rows t >= dres_skip of every group of dres_period rows and is
    // stored COMPACTLY (row (r / period) * (period - skip) + r % period - skip); the other rows get no residual term.  The block
    // below the last decoder block: only its masked tokens were passed on (modeling_pretrain.py:157).
    __shared__ float red[2][4][MAX_IT * 256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f32x4 wv[NIT], aw[NIT], ab[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int c = (it * 64 + lane) * 4;
        wv[it] = c < D ? *(const f32x4*)(w + c) : f32x4{0.f, 0.f, 0.f, 0.f};
        aw[it] = f32x4{0.f, 0.f, 0.f, 0.f};
        ab[it] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const int stride = gridDim.x * 4;
    for (int r0 = blockIdx.x * 4 + wave; r0 < M; r0 += 2 * stride) {
        int rr[2] = {r0, r0 + stride};
        const bool has2 = rr[1] < M;
        if (!has2) rr[1] = r0;
        size_t xr[2];
        float mu[2], rs[2];
        f32x4 xv[2][NIT], rv[2][NIT];
        u32x2 dv[2][NIT];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            xr[k] = map_row(rr[k], rows_in, rows_out, row_off);
            mu[k] = mean[rr[k]];
            rs[k] = rstd[rr[k]];
            size_t dr = xr[k];
            bool has_res = true;
            if (dres_period > 0) {
                const int cl = rr[k] / dres_period, t = rr[k] - cl * dres_period;
                has_res = t >= dres_skip;
                dr = (size_t)cl * (dres_period - dres_skip) + (t - dres_skip);
            }
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int c = (it * 64 + lane) * 4;
                if (c < D) {
                    if constexpr (XB) {
                        const u32x2 t = *(const u32x2*)((const bf16_t*)xv_ + xr[k] * ldx + c);
                        xv[k][it] = f32x4{bf16lo_to_f32(t[0]), bf16hi_to_f32(t[0]), bf16lo_to_f32(t[1]), bf16hi_to_f32(t[1])};
                    } else {
                        xv[k][it] = *(const f32x4*)((const float*)xv_ + xr[k] * ldx + c);
                    }
                    dv[k][it] = *(const u32x2*)(dy + (size_t)rr[k] * lddy + c);
                    if (dresb) {   // residual-stream gradient kept in bf16 (one tensor instead of an f32 + a bf16 copy)
                        u32x2 rb = {0u, 0u};
                        if (has_res) rb = *(const u32x2*)(dresb + dr * lddresb + c);
                        rv[k][it] = f32x4{bf16lo_to_f32(rb[0]), bf16hi_to_f32(rb[0]), bf16lo_to_f32(rb[1]), bf16hi_to_f32(rb[1])};
                    } else {
                        rv[k][it] = (dres && has_res) ? *(const f32x4*)(dres + dr * lddres + c) : f32x4{0.f, 0.f, 0.f, 0.f};
                    }
                } else {
                    xv[k][it] = f32x4{0.f, 0.f, 0.f, 0.f};
                    dv[k][it] = u32x2{0u, 0u};
                    rv[k][it] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
            }
        }
        // both rows' statistics are reduced TOGETHER: four independent values per shuffle step instead of four dependent
        // 6-step chains (the cross-lane shuffles go through the LDS crossbar; their latency, not HBM, bounded this kernel)
        f32x4 xh[2][NIT], g[2][NIT];
        float c1[2] = {0.f, 0.f}, c2[2] = {0.f, 0.f};
        const float live1 = has2 ? 1.0f : 0.0f;   // the duplicated row of an odd tail must not count in dw / db
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const float live = k == 0 ? 1.0f : live1;
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const float d[4] = {bf16lo_to_f32(dv[k][it][0]), bf16hi_to_f32(dv[k][it][0]), bf16lo_to_f32(dv[k][it][1]), bf16hi_to_f32(dv[k][it][1])};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    xh[k][it][e] = (xv[k][it][e] - mu[k]) * rs[k];
                    g[k][it][e] = d[e] * wv[it][e];
                    c1[k] += g[k][it][e];
                    c2[k] += g[k][it][e] * xh[k][it][e];
                    aw[it][e] += live * d[e] * xh[k][it][e];
                    ab[it][e] += live * d[e];
                }
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float t0 = __shfl_xor(c1[0], o, 64), t1 = __shfl_xor(c2[0], o, 64);
            const float t2 = __shfl_xor(c1[1], o, 64), t3 = __shfl_xor(c2[1], o, 64);
            c1[0] += t0; c2[0] += t1; c1[1] += t2; c2[1] += t3;
        }
        const float invD = 1.0f / D;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            if (k == 1 && !has2) break;
            const float m1 = c1[k] * invD, m2 = c2[k] * invD;
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int c = (it * 64 + lane) * 4;
                if (c < D) {
                    f32x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = rs[k] * (g[k][it][e] - m1 - xh[k][it][e] * m2) + rv[k][it][e];
                    if (dx) *(f32x4*)(dx + xr[k] * lddx + c) = o;
                    if (dxb) {
                        u32x2 pk = {pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3])};
                        *(u32x2*)(dxb + xr[k] * lddxb + c) = pk;
                    }
                }
            }
        }
    }
    // block reduce of the parameter-gradient partials, then one atomic per column per block
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            red[0][wave][(it * 64 + lane) * 4 + e] = aw[it][e];
            red[1][wave][(it * 64 + lane) * 4 + e] = ab[it][e];
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < D; c += 256) {
        const float sw = red[0][0][c] + red[0][1][c] + red[0][2][c] + red[0][3][c];
        const float sb = red[1][0][c] + red[1][1][c] + red[1][2][c] + red[1][3][c];
        if (partial) {   // every block adding into the SAME D addresses runs at ~1/14 of the atomic rate: store, reduce later
            partial[((size_t)blockIdx.x * 2) * D + c] = sw;
            partial[((size_t)blockIdx.x * 2 + 1) * D + c] = sb;
        } else {
            atomicAdd(dw + c, sw);
            atomicAdd(db + c, sb);
        }
    }
}

// dw[c] += sum_b partial[b][0][c], db[c] += sum_b partial[b][1][c].  grid (D/64, 2, FIN_SLICES): each block sums a slice of
// the block range (4 row lanes x 64 columns, coalesced 256-B rows) and adds it atomically: FIN_SLICES adders per address
// instead of up to 1024.
constexpr int FIN_SLICES = 16;
__global__ __launch_bounds__(256) void ln_bwd_finalize_kernel(const float* __restrict__ partial, int nblocks, int D,
                                                               float* __restrict__ dw, float* __restrict__ db) {
    __shared__ float red[4][64];
    const int cl = threadIdx.x & 63, part = threadIdx.x >> 6;
    const int which = blockIdx.y;                                  // 0: dw, 1: db
    const int c = blockIdx.x * 64 + cl;
    const int per = (nblocks + FIN_SLICES - 1) / FIN_SLICES;
    const int b0 = blockIdx.z * per, b1 = min(nblocks, b0 + per);
    float s = 0.f;
    if (c < D)
        for (int b = b0 + part; b < b1; b += 4) s += partial[((size_t)b * 2 + which) * D + c];
    red[part][cl] = s;
    __syncthreads();
    if (part == 0 && c < D && b0 < b1) atomicAdd((which ? db : dw) + c, red[0][cl] + red[1][cl] + red[2][cl] + red[3][cl]);
}

// The same for up to FIN_MAX LayerNorms in ONE launch: the per-LayerNorm launch cost 6 us x 34 per step for 0.1 us of work;
// the runtime now lets a gradient bucket's LayerNorms leave their block partials in separate workspaces and reduces them
// together before the bucket's all-reduce / the optimizer needs them.
constexpr int FIN_MAX = 40;   // one launch for all 34 LayerNorms of a ViT-B step when nothing consumes gradient buckets earlier
struct FinItems {
    const float* partial[FIN_MAX];
    float* dw[FIN_MAX];
    float* db[FIN_MAX];
    int nblocks[FIN_MAX];
    int D[FIN_MAX];
};
__global__ __launch_bounds__(256) void ln_bwd_finalize_grouped_kernel(FinItems it) {
    __shared__ float red[4][64];
    const int item = blockIdx.z / FIN_SLICES, slice = blockIdx.z % FIN_SLICES;
    const float* partial = it.partial[item];
    const int nblocks = it.nblocks[item], D = it.D[item];
    const int cl = threadIdx.x & 63, part = threadIdx.x >> 6;
    const int which = blockIdx.y;
    const int c = blockIdx.x * 64 + cl;
    const int per = (nblocks + FIN_SLICES - 1) / FIN_SLICES;
    const int b0 = slice * per, b1 = min(nblocks, b0 + per);
    float s = 0.f;
    if (c < D) {
        // eight independent loads in flight per thread (the one-load-per-iteration loop read 126 MB of partials at 1.5 TB/s: 84 us
        // on the critical path in front of AdamW)
        float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        int b = b0 + part;
        for (; b + 28 < b1; b += 32) {
#pragma unroll
            for (int u = 0; u < 8; ++u) a[u] += partial[((size_t)(b + 4 * u) * 2 + which) * D + c];
        }
        for (; b < b1; b += 4) a[0] += partial[((size_t)b * 2 + which) * D + c];
        s = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
    }
    red[part][cl] = s;
    __syncthreads();
    if (part == 0 && c < D && b0 < b1) atomicAdd((which ? it.db[item] : it.dw[item]) + c, red[0][cl] + red[1][cl] + red[2][cl] + red[3][cl]);
}

// Mean pooling over a clip's tokens + fc_norm of the fine-tune / feature-extraction model (modeling_finetune.py:403-405:
// `self.fc_norm(x.mean(1))`).  Stage 1: every block adds 32 token rows of one clip into pooled[b, :] (f32 atomics: B*D
// addresses, N/32 adds each).  Stage 2: one block per clip scales by 1/N and applies LayerNorm; writes f32 and bf16.
__global__ __launch_bounds__(256) void token_sum_kernel(const float* __restrict__ x, int ldx, int N, int D, float* __restrict__ pooled) {
    const int b = blockIdx.y;
    const int r0 = blockIdx.x * 32, r1 = min(N, r0 + 32);
    for (int c = threadIdx.x * 4; c < D; c += 1024) {
        f32x4 a = {0.f, 0.f, 0.f, 0.f};
        for (int r = r0; r < r1; ++r) a += *(const f32x4*)(x + ((size_t)b * N + r) * ldx + c);
#pragma unroll
        for (int k = 0; k < 4; ++k) atomicAdd(pooled + (size_t)b * D + c + k, a[k]);
    }
}

__global__ __launch_bounds__(256) void pool_norm_kernel(const float* __restrict__ pooled, int N, int D, const float* __restrict__ w,
                                                        const float* __restrict__ bias, float eps, float* __restrict__ out_f32,
                                                        bf16_t* __restrict__ out_bf16) {
    __shared__ float red[8];
    const int b = blockIdx.x, tid = threadIdx.x;
    const float inv_n = 1.0f / (float)N;
    auto block_sum = [&](float v) {
        v = wave_sum(v);
        __syncthreads();
        if ((tid & 63) == 0) red[tid >> 6] = v;
        __syncthreads();
        return red[0] + red[1] + red[2] + red[3];
    };
    float s = 0.f;
    for (int c = tid; c < D; c += 256) s += pooled[(size_t)b * D + c] * inv_n;
    const float mu = block_sum(s) / (float)D;
    float ss = 0.f;
    for (int c = tid; c < D; c += 256) {
        const float d = pooled[(size_t)b * D + c] * inv_n - mu;
        ss += d * d;
    }
    const float rstd = rsqrtf(block_sum(ss) / (float)D + eps);
    for (int c = tid; c < D; c += 256) {
        const float y = (pooled[(size_t)b * D + c] * inv_n - mu) * rstd * w[c] + bias[c];
        out_f32[(size_t)b * D + c] = y;
        if (out_bf16) out_bf16[(size_t)b * D + c] = f32_to_bf16(y);
    }
}

}  // namespace

static int ln_check(const char* who, int M, int D, int rows_in, int rows_out) {
    if (M <= 0 || D <= 0) MOFO_FAIL(MOFO_EINVAL, "%s: bad dims M=%d D=%d", who, M, D);
    if (D % 4 || D > 1024) MOFO_FAIL(MOFO_EUNSUPPORTED, "%s: D=%d must be a multiple of 4 and <= 1024", who, D);
    if (rows_in <= 0 || rows_out < rows_in) MOFO_FAIL(MOFO_EINVAL, "%s: bad row map", who);
    return MOFO_OK;
}

extern "C" int mofo_layernorm_fwd(const void* x, int x_is_bf16, int ldx, const float* w, const float* b, float eps, int M, int D,
                                  int rows_in, int rows_out, int row_off, void* y, int ldy, float* mean, float* rstd,
                                  void* stream) {
    if (!x || !w || !b || !y || !mean || !rstd) MOFO_FAIL(MOFO_EINVAL, "mofo_layernorm_fwd: null pointer");
    int rc = ln_check("mofo_layernorm_fwd", M, D, rows_in, rows_out);
    if (rc) return rc;
    if (ldx % 4 || ldy % 4) MOFO_FAIL(MOFO_EUNSUPPORTED, "mofo_layernorm_fwd: leading dims must be multiples of 4");
    hipStream_t s = (hipStream_t)stream;
    const int nit = ceil_div(D, 256);
    if (x_is_bf16 && D % 8 == 0 && D <= 512 && ldx % 8 == 0 && ldy % 8 == 0) {
        hipLaunchKernelGGL(ln_fwd_bf16_2row_kernel<false>, dim3(ceil_div(M, 8)), dim3(256), 0, s, (const bf16_t*)x, ldx, w, b, eps, M, D, rows_in, rows_out,
                           row_off, (bf16_t*)y, ldy, mean, rstd, (uint8_t*)nullptr, 0, (const float*)nullptr, (float*)nullptr);
        MOFO_CHECK_LAUNCH("mofo_layernorm_fwd");
        return MOFO_OK;
    }
    dim3 grid(ceil_div(M, 4)), block(256);
#define GO(N_) do { if (x_is_bf16) hipLaunchKernelGGL((ln_fwd_kernel<N_, true>), grid, block, 0, s, x, ldx, w, b, eps, M, D, rows_in, rows_out, row_off, (bf16_t*)y, ldy, mean, rstd); \
                    else hipLaunchKernelGGL((ln_fwd_kernel<N_, false>), grid, block, 0, s, x, ldx, w, b, eps, M, D, rows_in, rows_out, row_off, (bf16_t*)y, ldy, mean, rstd); } while (0)
    switch (nit) { case 1: GO(1); break; case 2: GO(2); break; case 3: GO(3); break; default: GO(4); break; }
#undef GO
    MOFO_CHECK_LAUNCH("mofo_layernorm_fwd");
    return MOFO_OK;
}

extern "C" int mofo_layernorm_fwd_q(const void* x, int x_is_bf16, int ldx, const float* w, const float* b, float eps, int M, int D,
                                    int rows_in, int rows_out, int row_off, void* y, int ldy, float* mean, float* rstd,
                                    void* y_e4m3, int ldy8, const float* qscale, float* amax_out, void* stream) {
    if (!x || !w || !b || !y || !mean || !rstd || !y_e4m3 || !qscale || !amax_out) MOFO_FAIL(MOFO_EINVAL, "mofo_layernorm_fwd_q: null pointer");
    int rc = ln_check("mofo_layernorm_fwd_q", M, D, rows_in, rows_out);
    if (rc) return rc;
    if (ldx % 4 || ldy % 4 || ldy8 % 4) MOFO_FAIL(MOFO_EUNSUPPORTED, "mofo_layernorm_fwd_q: leading dims must be multiples of 4");
    hipStream_t s = (hipStream_t)stream;
    const int nit = ceil_div(D, 256);
    if (x_is_bf16 && D % 8 == 0 && D <= 512 && ldx % 8 == 0 && ldy % 8 == 0 && ldy8 % 8 == 0) {
        hipLaunchKernelGGL(ln_fwd_bf16_2row_kernel<true>, dim3(ceil_div(M, 8)), dim3(256), 0, s, (const bf16_t*)x, ldx, w, b, eps, M, D, rows_in, rows_out,
                           row_off, (bf16_t*)y, ldy, mean, rstd, (uint8_t*)y_e4m3, ldy8, qscale, amax_out);
        MOFO_CHECK_LAUNCH("mofo_layernorm_fwd_q");
        return MOFO_OK;
    }
    dim3 grid(ceil_div(M, 4)), block(256);
#define GO(N_) do { if (x_is_bf16) hipLaunchKernelGGL((ln_fwd_kernel<N_, true, true>), grid, block, 0, s, x, ldx, w, b, eps, M, D, rows_in, rows_out, row_off, (bf16_t*)y, ldy, mean, rstd, (uint8_t*)y_e4m3, ldy8, qscale, amax_out); \
                    else hipLaunchKernelGGL((ln_fwd_kernel<N_, false, true>), grid, block, 0, s, x, ldx, w, b, eps, M, D, rows_in, rows_out, row_off, (bf16_t*)y, ldy, mean, rstd, (uint8_t*)y_e4m3, ldy8, qscale, amax_out); } while (0)
    switch (nit) { case 1: GO(1); break; case 2: GO(2); break; case 3: GO(3); break; default: GO(4); break; }
#undef GO
    MOFO_CHECK_LAUNCH("mofo_layernorm_fwd_q");
    return MOFO_OK;
}

extern "C" int mofo_layernorm_bwd_blocks(int M) {
    int blocks = ceil_div(M, 8);
    return blocks > 1024 ? 1024 : (blocks < 1 ? 1 : blocks);
}

extern "C" int mofo_layernorm_bwd(const void* dy, int lddy, const void* x, int x_is_bf16, int ldx, const float* w, const float* mean,
                                  const float* rstd, const float* dres, int lddres, int M, int D, int rows_in, int rows_out,
                                  int row_off, float* dx, int lddx, void* dxb, int lddxb, float* dw, float* db,
                                  const void* dresb, int lddresb, float* partial_ws, void* stream) {
    return mofo_layernorm_bwd_partial_res(dy, lddy, x, x_is_bf16, ldx, w, mean, rstd, dres, lddres, M, D, rows_in, rows_out, row_off, dx, lddx,
                                          dxb, lddxb, dw, db, dresb, lddresb, partial_ws, 0, 0, stream);
}

extern "C" int mofo_layernorm_bwd_partial_res(const void* dy, int lddy, const void* x, int x_is_bf16, int ldx, const float* w, const float* mean,
                                              const float* rstd, const float* dres, int lddres, int M, int D, int rows_in, int rows_out,
                                              int row_off, float* dx, int lddx, void* dxb, int lddxb, float* dw, float* db,
                                              const void* dresb, int lddresb, float* partial_ws, int dres_period, int dres_skip, void* stream) {
    if (dres_period < 0 || (dres_period > 0 && (dres_skip < 0 || dres_skip >= dres_period || rows_in != rows_out || row_off != 0 || M % dres_period)))
        MOFO_FAIL(MOFO_EINVAL, "mofo_layernorm_bwd: partial residual needs 0 <= dres_skip < dres_period, M a multiple of it and no row map");
    const bool defer = partial_ws && !dw && !db;      // block partials only; mofo_layernorm_bwd_finalize reduces them later
    if (!dy || !x || !w || !mean || !rstd || (!defer && (!dw || !db))) MOFO_FAIL(MOFO_EINVAL, "mofo_layernorm_bwd: null pointer");
    if (!dx && !dxb) MOFO_FAIL(MOFO_EINVAL, "mofo_layernorm_bwd: need dx (f32) and/or dx_bf16");
    if (dres && dresb) MOFO_FAIL(MOFO_EINVAL, "mofo_layernorm_bwd: pass the residual gradient as f32 OR bf16, not both");
    int rc = ln_check("mofo_layernorm_bwd", M, D, rows_in, rows_out);
    if (rc) return rc;
    if (lddy % 4 || ldx % 4 || (dx && lddx % 4) || (dres && lddres % 4) || (dxb && lddxb % 4) || (dresb && lddresb % 4))
        MOFO_FAIL(MOFO_EUNSUPPORTED, "mofo_layernorm_bwd: leading dims must be multiples of 4");
    hipStream_t s = (hipStream_t)stream;
    const int nit = ceil_div(D, 256);
    // every block pays a fixed cost (weight load, LDS reduce, 2 D atomics); 2 rows per wave keeps >= 2 blocks per CU busy
    // at the encoder's M = 5120 while bounding the atomic traffic (1024 blocks x 2 D floats)
    const int blocks = mofo_layernorm_bwd_blocks(M);
    dim3 grid(blocks), block(256);
#define GO_(N_, XB_) hipLaunchKernelGGL((ln_bwd_kernel<N_, XB_>), grid, block, 0, s, (const bf16_t*)dy, lddy, x, ldx, w, mean, rstd, dres, lddres, M, D, rows_in, rows_out, row_off, dx, lddx, (bf16_t*)dxb, lddxb, dw, db, (const bf16_t*)dresb, lddresb, partial_ws, dres_period, dres_skip)
#define GO(N_) do { if (x_is_bf16) GO_(N_, true); else GO_(N_, false); } while (0)
    switch (nit) { case 1: GO(1); break; case 2: GO(2); break; case 3: GO(3); break; default: GO(4); break; }
#undef GO_
#undef GO
    MOFO_CHECK_LAUNCH("mofo_layernorm_bwd");
    if (partial_ws && !defer) {
        hipLaunchKernelGGL(ln_bwd_finalize_kernel, dim3(ceil_div(D, 64), 2, FIN_SLICES), dim3(256), 0, s, (const float*)partial_ws, blocks, D, dw, db);
        MOFO_CHECK_LAUNCH("mofo_layernorm_bwd(finalize)");
    }
    return MOFO_OK;
}

extern "C" int mofo_layernorm_bwd_finalize(const float* const* partials, const int* nblocks, const int* Ds, float* const* dws,
                                           float* const* dbs, int count, void* stream) {
    if (!partials || !nblocks || !Ds || !dws || !dbs) MOFO_FAIL(MOFO_EINVAL, "mofo_layernorm_bwd_finalize: null pointer");
    if (count < 1 || count > FIN_MAX) MOFO_FAIL(MOFO_EINVAL, "mofo_layernorm_bwd_finalize: count must be 1..%d", FIN_MAX);
    FinItems it;
    int dmax = 0;
    for (int i = 0; i < FIN_MAX; ++i) {
        const int k = i < count ? i : 0;
        if (!partials[k] || !dws[k] || !dbs[k] || nblocks[k] < 1 || nblocks[k] > 1024 || Ds[k] < 1)
            MOFO_FAIL(MOFO_EINVAL, "mofo_layernorm_bwd_finalize: bad item %d", k);
        it.partial[i] = partials[k]; it.dw[i] = dws[k]; it.db[i] = dbs[k]; it.nblocks[i] = nblocks[k]; it.D[i] = Ds[k];
        if (Ds[k] > dmax) dmax = Ds[k];
    }
    hipLaunchKernelGGL(ln_bwd_finalize_grouped_kernel, dim3(ceil_div(dmax, 64), 2, FIN_SLICES * count), dim3(256), 0, (hipStream_t)stream, it);
    MOFO_CHECK_LAUNCH("mofo_layernorm_bwd_finalize");
    return MOFO_OK;
}

extern "C" int mofo_token_mean_norm(const float* x, int ldx, int B, int N, int D, const float* w, const float* b, float eps,
                                    float* pooled_ws, float* out_f32, void* out_bf16, void* stream) {
    if (!x || !w || !b || !pooled_ws || !out_f32) MOFO_FAIL(MOFO_EINVAL, "mofo_token_mean_norm: null pointer");
    if (B <= 0 || N <= 0 || D <= 0 || D % 4 || ldx % 4 || ldx < D) MOFO_FAIL(MOFO_EINVAL, "mofo_token_mean_norm: bad sizes (D, ldx multiples of 4)");
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(pooled_ws, 0, (size_t)B * D * sizeof(float), s) != hipSuccess) MOFO_FAIL(MOFO_ERUNTIME, "mofo_token_mean_norm: memset failed");
    hipLaunchKernelGGL(token_sum_kernel, dim3(ceil_div(N, 32), B), dim3(256), 0, s, x, ldx, N, D, pooled_ws);
    MOFO_CHECK_LAUNCH("mofo_token_mean_norm(sum)");
    hipLaunchKernelGGL(pool_norm_kernel, dim3(B), dim3(256), 0, s, (const float*)pooled_ws, N, D, w, b, eps, out_f32, (bf16_t*)out_bf16);
    MOFO_CHECK_LAUNCH("mofo_token_mean_norm(norm)");
    return MOFO_OK;
}
