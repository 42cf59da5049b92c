// Fused multi-head self-attention for gfx950, head_dim 64 (every ViT config of the reference), bf16 in / f32 accumulate.
// Replaces modeling_finetune.py:85-95 (q*scale; q@k^T; softmax; @v) and its autograd backward; the [B,H,N,N]
// score / probability tensors of the reference are never materialised.
//
// All three kernels use v_mfma_f32_32x32x16_bf16 with the "other" sequence index on the LANE:
//   fwd / dQ : S^T[key][query] = K . Q^T    -> each lane owns one query column: row-softmax is in-lane + one half swap
//   dK,dV    : S  [query][key] = Q . K^T    -> each lane owns one key column
// so the f32 accumulator of the first product is, after a bf16 pack, directly the B operand of the second product
// (k index permuted as the accumulator rows are: element j of lane half h <-> row 16s + 8(j>>2) + 4h + (j&3));
// the matching A operand (V^T, K^T, dO^T, Q^T) comes from the row-major LDS tile through ds_read_b64_tr_b16.
// K/V (or Q/dO) tiles of 32 rows are register-staged into a 2-deep LDS ring: one barrier per tile.
#include "common.h"
#include "../../include/mofo_hip.h"
#include <stdlib.h>
#include <type_traits>

namespace {

// Debug build only (-DMOFO_ATTN_TRACE, tools/attn_trace.py): s_memtime stamps inside key tile 10 of wave 0 of every block.
#ifdef MOFO_ATTN_TRACE
__device__ unsigned long long g_attn_trace[1 << 18];
#define ATTN_STAMP(kt, slot)                                                                       \
    do {                                                                                           \
        if ((kt) == 10 && threadIdx.x == 0 && blockIdx.x < (1 << 15)) g_attn_trace[blockIdx.x * 8 + (slot)] = __builtin_readcyclecounter(); \
    } while (0)
#define FUSED_STAMP(slot)                                                                          \
    do {                                                                                           \
        if (threadIdx.x == 0 && blockIdx.x < (1 << 15)) g_attn_trace[blockIdx.x * 8 + (slot)] = __builtin_readcyclecounter(); \
    } while (0)
#else
#define ATTN_STAMP(kt, slot)
#define FUSED_STAMP(slot)
#endif

constexpr int HD = 64;
constexpr int RS = 128;              // LDS row stride in bytes (no padding)
constexpr int TILE = 32 * RS;        // one 32-row tile
// One LDS image serves BOTH the row reads (ds_read_b128, MFMA operand = rows of the tile) and the transposed reads
// (ds_read_b64_tr_b16, MFMA operand = columns): 16-B chunk c of row r is stored at chunk c ^ swz(r),
// swz(r) = ((r>>1)&1)<<2 | (r>>2)&3.  Brute-force checked conflict-free for every fragment of both kinds (the padded
// 144-B rows were 2-way conflicted on the transposed reads: 23-26 % of the LDS cycles, and these kernels are LDS-bound).
__device__ __forceinline__ int swz(int r) { return (((r >> 1) & 1) << 2) | ((r >> 2) & 3); }
constexpr float NEG_BIG = -1.0e30f;

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float fast_log2(float x) { return __builtin_amdgcn_logf(x); }
__device__ __forceinline__ int acc_row(int r, int hh) { return (r & 3) + 8 * (r >> 2) + 4 * hh; }

__device__ __forceinline__ f32x16 zero16() {
    f32x16 z;
#pragma unroll
    for (int i = 0; i < 16; ++i) z[i] = 0.f;
    return z;
}

// A operand = rows of a [32][64] LDS tile: A[row = lane&31][k = 16 ks + 8 hh + j]
__device__ __forceinline__ bf16x8 row_frag(const unsigned char* tile, int ks, int lane) {
    const int r = lane & 31;
    return *(const bf16x8*)(tile + r * RS + (((2 * ks + (lane >> 5)) ^ swz(r)) << 4));
}

// A operand = TRANSPOSE of a [32 seq][64 d] LDS tile for k-step s2 (16 seq rows) and d-tile dt (32 d):
// A[row = d = 32 dt + (lane&31)][slot (hh, j)] = tile[seq = 16 s2 + 8 (j>>2) + 4 hh + (j&3)][d]
__device__ __forceinline__ bf16x8 tr_frag(const unsigned char* tile, int s2, int dt, int lane) {
    const int g = lane >> 4, i = lane & 15, q = i >> 2, pp = i & 3, hh = lane >> 5;
    const int r0 = 16 * s2 + 4 * hh + q, r1 = r0 + 8;
    const int chunk = 4 * dt + 2 * (g & 1) + (pp >> 1), sub = (pp & 1) * 8;
    const unsigned char* a0 = tile + r0 * RS + ((chunk ^ swz(r0)) << 4) + sub;
    const unsigned char* a1 = tile + r1 * RS + ((chunk ^ swz(r1)) << 4) + sub;
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)LDS_PTR(a0));
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)LDS_PTR(a1));
    // compose as 32-bit words (element-wise 16-bit assembly made hipcc emit v_perm / v_or / v_mov chains)
    const u32x2 l2 = __builtin_bit_cast(u32x2, lo), h2 = __builtin_bit_cast(u32x2, hi);
    const u32x4 r = {l2[0], l2[1], h2[0], h2[1]};
    return __builtin_bit_cast(bf16x8, r);
}

// pack accumulator registers 8 s .. 8 s + 7 into the B fragment of k-step s
__device__ __forceinline__ bf16x8 pack_frag(const float* p, int s) {
    bf16x8 f;
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = (__bf16)p[8 * s + j];
    return f;
}

// store an accumulator tile pair D[d][seq] (seq on the lane) as bf16 rows dst[seq][d], scaled
__device__ __forceinline__ void store_T(bf16_t* dst_row, const f32x16& a0, const f32x16& a1, float mul, int hh) {
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) {
        const f32x16& a = dt ? a1 : a0;
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
            u32x2 o = {pack_bf16x2(a[4 * rg] * mul, a[4 * rg + 1] * mul), pack_bf16x2(a[4 * rg + 2] * mul, a[4 * rg + 3] * mul)};
            *(u32x2*)(dst_row + 32 * dt + 8 * rg + 4 * hh) = o;
        }
    }
}

// XCD-aware block decode.  Blocks are dealt round-robin over the 8 XCDs (private L2s), so the NX blocks that share one
// (clip, head)'s K/V (or Q/dO) must have block ids that are EQUAL mod 8, otherwise every XCD re-fetches the same K/V
// (measured: FETCH_SIZE 5x the algorithmic bytes, TA/TC 93 % busy).  1-D grid of 8 * ceil(G/8) * NX blocks:
// id -> xcd = id & 7, slot = id >> 3 -> group = (slot / NX) * 8 + xcd, x = slot % NX.  Speed only, never correctness.
__device__ __forceinline__ bool decode_block(int nx, int G, int H, int& xb, int& b, int& h) {
    const int id = blockIdx.x;
    const int xcd = id & 7, slot = id >> 3;
    const int gl = slot / nx;
    xb = slot - gl * nx;
    const int g = gl * 8 + xcd;
    if (g >= G) return false;
    b = g / H;
    h = g - b * H;
    return true;
}

// ------------------------------------------------------------------------------------------------ forward
// MODE 0: forward (writes out, lse2).  MODE 1: dQ pass of the backward (writes delta and the q part of dqkv).
// WHOLE: the sequence is short (N <= 160: the encoder's visible tokens): all K/V tiles are staged once, one barrier,
// and the tile loop runs without further loads or barriers (the streaming form spent its time in 5 load->barrier rounds).
template <int NW, int MODE, bool WHOLE>
__global__ __launch_bounds__(NW * 64) void attn_q_kernel(const bf16_t* __restrict__ qkv, int ldqkv, int nx, int G, int N, int H, float c,
                                                          float scale, bf16_t* __restrict__ out, int ldo,
                                                          float* __restrict__ lse2, const bf16_t* __restrict__ dout, int lddo,
                                                          bf16_t* __restrict__ dqkv, int lddqkv, float* __restrict__ delta) {
    constexpr int NBUF = WHOLE ? 5 : 2;
    __shared__ __attribute__((aligned(16))) unsigned char smem[NBUF * 2 * TILE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hh = lane >> 5;
    int xb, b, h;
    if (!decode_block(nx, G, H, xb, b, h)) return;
    const int D = H * HD;
    const int q0 = (xb * NW + wave) * 32;
    const bf16_t* base = qkv + (size_t)b * N * ldqkv;
    const bf16_t* qp = base + h * HD;
    const bf16_t* kp = base + D + h * HD;
    const bf16_t* vp = base + 2 * D + h * HD;
    const int qi = q0 + (lane & 31);
    const int qrow = qi < N ? qi : N - 1;
    const bool qvalid = qi < N;

    bf16x8 qf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[ks] = *(const bf16x8*)(qp + (size_t)qrow * ldqkv + 16 * ks + 8 * hh);

    bf16x8 dof[4];
    float L2 = 0.f, dl = 0.f;
    if constexpr (MODE == 1) {
        const bf16_t* dop = dout + ((size_t)b * N + qrow) * lddo + h * HD;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) dof[ks] = *(const bf16x8*)(dop + 16 * ks + 8 * hh);
        L2 = lse2[((size_t)b * H + h) * N + qrow];
        dl = delta[((size_t)b * H + h) * N + qrow];     // rowsum(dO * O), from attn_delta_kernel
    }

    f32x16 o0 = zero16(), o1 = zero16();
    float m = NEG_BIG, l = 0.f;

    const int nkt = (N + 31) >> 5;
    u32x4 kreg = {0, 0, 0, 0}, vreg = {0, 0, 0, 0};
    auto gload = [&](int kt) {
        if (tid < 256) {
            int r = kt * 32 + (tid >> 3);
            r = r < N ? r : N - 1;
            const size_t off = (size_t)r * ldqkv + (tid & 7) * 8;
            kreg = *(const u32x4*)(kp + off);
            vreg = *(const u32x4*)(vp + off);
        }
    };
    auto lwrite = [&](int buf) {
        if (tid < 256) {
            unsigned char* d = smem + buf * 2 * TILE + (tid >> 3) * RS + (((tid & 7) ^ swz(tid >> 3)) << 4);
            *(u32x4*)d = kreg;
            *(u32x4*)(d + TILE) = vreg;
        }
    };
    if constexpr (WHOLE) {
        for (int kt = 0; kt < nkt; ++kt) {
            gload(kt);
            lwrite(kt);
        }
        __syncthreads();
    } else {
        gload(0);
        lwrite(0);
        __syncthreads();
        if (nkt > 1) gload(1);
    }

    // the tile body is instantiated twice: full tiles carry NO masking code; only a ragged last tile pays for the
    // per-element key-index compare / select (the single-version loop spent ~64 VALU per tile on it)
    auto tile = [&](int kt, auto masked_tag) {
        constexpr bool MASKED = decltype(masked_tag)::value;
        const unsigned char* Kt = smem + (WHOLE ? kt : (kt & 1)) * 2 * TILE;
        const unsigned char* Vt = Kt + TILE;
        ATTN_STAMP(kt, 0);
        f32x16 s = zero16();
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag(Kt, ks, lane), qf[ks], s, 0, 0, 0);
        ATTN_STAMP(kt, 1);
        float p[16];
        if constexpr (MODE == 0) {
            // running max m is kept in RAW score units; p = exp2(c * s - c * m) is one FMA + one v_exp per element
            float mx = NEG_BIG;
            if constexpr (MASKED) {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (kt * 32 + acc_row(r, hh) >= N) s[r] = NEG_BIG;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[r]);
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float mn = fmaxf(m, mx);
            const float nmc = -mn * c;
            float ls = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                p[r] = fast_exp2(fmaf(s[r], c, nmc));
                ls += p[r];
            }
            if (__any(mn > m)) {   // wave-uniform: after the first tiles the running max rarely moves
                const float alpha = fast_exp2((m - mn) * c);
                l *= alpha;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    o0[r] *= alpha;
                    o1[r] *= alpha;
                }
            }
            l += ls;
            m = mn;
            const bf16x8 pf0 = pack_frag(p, 0), pf1 = pack_frag(p, 1);
            ATTN_STAMP(kt, 2);
            o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag(Vt, 0, 0, lane), pf0, o0, 0, 0, 0);
            o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag(Vt, 1, 0, lane), pf1, o0, 0, 0, 0);
            o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag(Vt, 0, 1, lane), pf0, o1, 0, 0, 0);
            o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag(Vt, 1, 1, lane), pf1, o1, 0, 0, 0);
        } else {
            f32x16 dp = zero16();
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag(Vt, ks, lane), dof[ks], dp, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float pr = fast_exp2(fmaf(s[r], c, -L2));
                if constexpr (MASKED) {
                    if (kt * 32 + acc_row(r, hh) >= N) pr = 0.f;
                }
                p[r] = pr * (dp[r] - dl);
            }
            const bf16x8 f0 = pack_frag(p, 0), f1 = pack_frag(p, 1);
            ATTN_STAMP(kt, 2);
            o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag(Kt, 0, 0, lane), f0, o0, 0, 0, 0);
            o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag(Kt, 1, 0, lane), f1, o0, 0, 0, 0);
            o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag(Kt, 0, 1, lane), f0, o1, 0, 0, 0);
            o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag(Kt, 1, 1, lane), f1, o1, 0, 0, 0);
        }
        ATTN_STAMP(kt, 3);
        if constexpr (!WHOLE) {
            if (kt + 1 < nkt) lwrite((kt + 1) & 1);
            ATTN_STAMP(kt, 4);
            __syncthreads();
            ATTN_STAMP(kt, 5);
            if (kt + 2 < nkt) gload(kt + 2);
        }
        ATTN_STAMP(kt, 6);
    };
    const int nfull = N >> 5;
    for (int kt = 0; kt < nfull; ++kt) tile(kt, std::false_type{});
    if (nfull < nkt) tile(nfull, std::true_type{});

    if (!qvalid) return;
    if constexpr (MODE == 0) {
        const float lt = l + __shfl_xor(l, 32, 64);
        const float inv = 1.0f / lt;
        store_T(out + ((size_t)b * N + qi) * ldo + h * HD, o0, o1, inv, hh);
        if (hh == 0) lse2[((size_t)b * H + h) * N + qi] = m * c + fast_log2(lt);
    } else {
        store_T(dqkv + ((size_t)b * N + qi) * lddqkv + h * HD, o0, o1, scale, hh);
    }
}

// ------------------------------------------------------------------------------------------------ dK, dV
template <int NW, bool WHOLE>
__global__ __launch_bounds__(NW * 64) void attn_dkv_kernel(const bf16_t* __restrict__ qkv, int ldqkv, int nx, int G, int N, int H, float c,
                                                            float scale, const bf16_t* __restrict__ dout, int lddo,
                                                            const float* __restrict__ lse2, const float* __restrict__ delta,
                                                            bf16_t* __restrict__ dqkv, int lddqkv) {
    // per buffer: Q tile, dO tile, then 32 f32 lse2 + 32 f32 delta
    constexpr int BUF = 2 * TILE + 256;
    constexpr int NBUF = WHOLE ? 5 : 2;
    __shared__ __attribute__((aligned(16))) unsigned char smem[NBUF * BUF];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hh = lane >> 5;
    int xb, b, h;
    if (!decode_block(nx, G, H, xb, b, h)) return;
    const int D = H * HD;
    const int k0 = (xb * NW + wave) * 32;
    const bf16_t* base = qkv + (size_t)b * N * ldqkv;
    const bf16_t* qp = base + h * HD;
    const bf16_t* kp = base + D + h * HD;
    const bf16_t* vp = base + 2 * D + h * HD;
    const bf16_t* dop = dout + (size_t)b * N * lddo + h * HD;
    const float* lp = lse2 + ((size_t)b * H + h) * N;
    const float* dp_ = delta + ((size_t)b * H + h) * N;
    const int ki = k0 + (lane & 31);
    const int krow = ki < N ? ki : N - 1;

    bf16x8 kf[4], vf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        kf[ks] = *(const bf16x8*)(kp + (size_t)krow * ldqkv + 16 * ks + 8 * hh);
        vf[ks] = *(const bf16x8*)(vp + (size_t)krow * ldqkv + 16 * ks + 8 * hh);
    }
    f32x16 dk0 = zero16(), dk1 = zero16(), dv0 = zero16(), dv1 = zero16();

    const int nqt = (N + 31) >> 5;
    u32x4 qreg = {0, 0, 0, 0}, oreg = {0, 0, 0, 0};
    float sreg = 0.f;
    auto gload = [&](int qt) {
        if (tid < 256) {
            int r = qt * 32 + (tid >> 3);
            r = r < N ? r : N - 1;
            qreg = *(const u32x4*)(qp + (size_t)r * ldqkv + (tid & 7) * 8);
            oreg = *(const u32x4*)(dop + (size_t)r * lddo + (tid & 7) * 8);
            if (tid < 64) {
                const int qq = qt * 32 + (tid & 31);
                // a query row beyond N must contribute nothing: lse2 = +big -> p = exp2(-big) = 0, delta = 0
                if (tid < 32) sreg = qq < N ? lp[qq] : 1.0e30f;
                else sreg = qq < N ? dp_[qq] : 0.f;
            }
        }
    };
    auto lwrite = [&](int buf) {
        if (tid < 256) {
            unsigned char* d = smem + buf * BUF + (tid >> 3) * RS + (((tid & 7) ^ swz(tid >> 3)) << 4);
            *(u32x4*)d = qreg;
            *(u32x4*)(d + TILE) = oreg;
            if (tid < 64) *(float*)(smem + buf * BUF + 2 * TILE + tid * 4) = sreg;
        }
    };
    if constexpr (WHOLE) {
        for (int qt = 0; qt < nqt; ++qt) {
            gload(qt);
            lwrite(qt);
        }
        __syncthreads();
    } else {
        gload(0);
        lwrite(0);
        __syncthreads();
        if (nqt > 1) gload(1);
    }

    for (int qt = 0; qt < nqt; ++qt) {
        const unsigned char* Qt = smem + (WHOLE ? qt : (qt & 1)) * BUF;
        const unsigned char* Ot = Qt + TILE;
        const float* Lt = (const float*)(Qt + 2 * TILE);
        const float* Dt = Lt + 32;
        f32x16 s = zero16(), dpv = zero16();
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag(Qt, ks, lane), kf[ks], s, 0, 0, 0);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) dpv = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag(Ot, ks, lane), vf[ks], dpv, 0, 0, 0);
        float p[16], ds[16];
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
            const f32x4 lv = *(const f32x4*)(Lt + 8 * rg + 4 * hh);
            const f32x4 dv = *(const f32x4*)(Dt + 8 * rg + 4 * hh);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int r = 4 * rg + e;
                const float pr = fast_exp2(s[r] * c - lv[e]);
                p[r] = pr;
                ds[r] = pr * (dpv[r] - dv[e]);
            }
        }
        const bf16x8 pf0 = pack_frag(p, 0), pf1 = pack_frag(p, 1);
        const bf16x8 sf0 = pack_frag(ds, 0), sf1 = pack_frag(ds, 1);
        dv0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag(Ot, 0, 0, lane), pf0, dv0, 0, 0, 0);
        dv0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag(Ot, 1, 0, lane), pf1, dv0, 0, 0, 0);
        dv1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag(Ot, 0, 1, lane), pf0, dv1, 0, 0, 0);
        dv1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag(Ot, 1, 1, lane), pf1, dv1, 0, 0, 0);
        dk0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag(Qt, 0, 0, lane), sf0, dk0, 0, 0, 0);
        dk0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag(Qt, 1, 0, lane), sf1, dk0, 0, 0, 0);
        dk1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag(Qt, 0, 1, lane), sf0, dk1, 0, 0, 0);
        dk1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag(Qt, 1, 1, lane), sf1, dk1, 0, 0, 0);
        if constexpr (!WHOLE) {
            if (qt + 1 < nqt) lwrite((qt + 1) & 1);
            __syncthreads();
            if (qt + 2 < nqt) gload(qt + 2);
        }
    }
    if (ki >= N) return;
    bf16_t* drow = dqkv + ((size_t)b * N + ki) * lddqkv + h * HD;
    store_T(drow + D, dk0, dk1, scale, hh);
    store_T(drow + 2 * D, dv0, dv1, 1.0f, hh);
}

// ------------------------------------------------------------------------------------------------ short sequences: ONE backward kernel
// N <= 160 (the encoder's visible tokens): delta, dQ, dK and dV of one (clip, head) in ONE block of 5 waves, Q / dO / K
// tiles resident in LDS (71 KiB -> two blocks per CU, the 384 (clip, head) pairs of B=32 run in one round).  The
// three-kernel form costs 9 + 16 + 22 us per layer for 10 GFLOP -- launch ramps and two recomputations of S and dP.
// Here every (query tile i, key tile j) pair is visited once, by wave j, in the key-on-lane orientation of
// attn_dkv_kernel (S^T, dP^T, P, dS -> dV_j, dK_j in registers: 5 MFMA products instead of 7).  dQ needs dS with the
// QUERY on the lane and is accumulated by the wave that OWNS the query tile: at step s wave j works on query tile
// i = (j + s) mod T and leaves dS_ij in its 2 KiB scratch as [key][query] bf16; after a barrier wave i picks up the
// scratch of wave j = (i - s) mod T through the transposing `ds_read_b64_tr_b16` and adds K_j^T dS_ij to its dQ_i^T
// registers; a second barrier frees the scratch.  delta = rowsum(dO * O) is computed while the tiles are staged.
constexpr int FUSED_NW = 5;
constexpr int SCR = 64;                                        // scratch row stride: 32 queries x bf16
constexpr int FUSED_SMEM = 3 * FUSED_NW * TILE + FUSED_NW * 256 + FUSED_NW * 32 * SCR;   // 72 960 B

// B operand = TRANSPOSE of a wave's dS scratch [32 keys][32 queries]: lane <-> query, k slots <-> keys in tr_frag's order
__device__ __forceinline__ bf16x8 tr_frag_scratch(const unsigned char* tile, int s2, int lane) {
    const int g = lane >> 4, i = lane & 15, q = i >> 2, pp = i & 3, hh = lane >> 5;
    const int r0 = 16 * s2 + 4 * hh + q, r1 = r0 + 8;
    const int chunk = 2 * (g & 1) + (pp >> 1), sub = (pp & 1) * 8;
    const unsigned char* a0 = tile + r0 * SCR + ((chunk ^ (swz(r0) & 3)) << 4) + sub;
    const unsigned char* a1 = tile + r1 * SCR + ((chunk ^ (swz(r1) & 3)) << 4) + sub;
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)LDS_PTR(a0));
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)LDS_PTR(a1));
    const u32x2 l2 = __builtin_bit_cast(u32x2, lo), h2 = __builtin_bit_cast(u32x2, hi);
    const u32x4 r = {l2[0], l2[1], h2[0], h2[1]};
    return __builtin_bit_cast(bf16x8, r);
}

__global__ __launch_bounds__(FUSED_NW * 64) void attn_bwd_fused_kernel(const bf16_t* __restrict__ qkv, int ldqkv, int G, int N, int H, float c,
                                                                          float scale, const bf16_t* __restrict__ out, int ldo,
                                                                          const bf16_t* __restrict__ dout, int lddo,
                                                                          const float* __restrict__ lse2, float* __restrict__ delta_out,
                                                                          bf16_t* __restrict__ dqkv, int lddqkv) {
    extern __shared__ __attribute__((aligned(16))) unsigned char fsm[];
    unsigned char* QT = fsm;                                   // [5][TILE]
    unsigned char* OT = QT + FUSED_NW * TILE;                  // dO tiles
    unsigned char* KT = OT + FUSED_NW * TILE;
    float* LD = (float*)(KT + FUSED_NW * TILE);                // [5][64]: 32 lse2 + 32 delta per query tile
    unsigned char* SC = (unsigned char*)(LD + FUSED_NW * 64);  // [5][32 x SCR] per-wave dS scratch, [key][query]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hh = lane >> 5;
    int xb, b, h;
    if (!decode_block(1, G, H, xb, b, h)) return;
    const int D = H * HD;
    const bf16_t* base = qkv + (size_t)b * N * ldqkv;
    const bf16_t* qp = base + h * HD;
    const bf16_t* kp = base + D + h * HD;
    const bf16_t* vp = base + 2 * D + h * HD;
    const bf16_t* dop = dout + (size_t)b * N * lddo + h * HD;
    const bf16_t* op = out + (size_t)b * N * ldo + h * HD;
    const float* lp = lse2 + ((size_t)b * H + h) * N;
    const int nt = (N + 31) >> 5;                              // tiles (<= 5)
    FUSED_STAMP(0);

    // ---- this wave's V rows as MFMA B fragments (key on the lane); its K fragments are re-read from the LDS tile per step
    const int ki = wave * 32 + (lane & 31);
    const int krow = ki < N ? ki : N - 1;
    const bool kvalid = ki < N;
    bf16x8 vf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) vf[ks] = *(const bf16x8*)(vp + (size_t)krow * ldqkv + 16 * ks + 8 * hh);
    // ---- stage Q, dO, K tiles; lse2 and delta per query row (all loads of all tiles issued before the first is consumed)
    if (tid < 256) {
        const int rl = tid >> 3, ch = tid & 7;
        // two batches (3 + 2 tiles): the loads of a batch are all issued before its first use; 80 staging VGPRs at once
        // would not fit beside the 3-waves-per-SIMD register budget
#pragma unroll
        for (int t0 = 0; t0 < FUSED_NW; t0 += 3) {
            constexpr int NB = 3;
            u32x4 q[NB], k[NB];
            bf16x8 g[NB], o[NB];
            float lv[NB];
#pragma unroll
            for (int u = 0; u < NB; ++u) {
                const int t = t0 + u;
                if (t < nt && t < FUSED_NW) {
                    const int row = t * 32 + rl;
                    const int r = row < N ? row : N - 1;
                    q[u] = *(const u32x4*)(qp + (size_t)r * ldqkv + ch * 8);
                    k[u] = *(const u32x4*)(kp + (size_t)r * ldqkv + ch * 8);
                    g[u] = *(const bf16x8*)(dop + (size_t)r * lddo + ch * 8);
                    o[u] = *(const bf16x8*)(op + (size_t)r * ldo + ch * 8);
                    lv[u] = lp[r];
                }
            }
#pragma unroll
            for (int u = 0; u < NB; ++u) {
                const int t = t0 + u;
                if (t < nt && t < FUSED_NW) {
                    const int row = t * 32 + rl;
                    const int off = t * TILE + rl * RS + ((ch ^ swz(rl)) << 4);
                    *(u32x4*)(QT + off) = q[u];
                    *(u32x4*)(KT + off) = k[u];
                    *(bf16x8*)(OT + off) = g[u];
                    float part = 0.f;
#pragma unroll
                    for (int j = 0; j < 8; ++j) part += (float)g[u][j] * (float)o[u][j];
                    part += __shfl_xor(part, 1, 64);
                    part += __shfl_xor(part, 2, 64);
                    part += __shfl_xor(part, 4, 64);
                    if (ch == 0) {
                        // a query row beyond N contributes nothing: lse2 = +big -> p = exp2(-big) = 0, delta = 0
                        LD[t * 64 + rl] = row < N ? lv[u] : 1.0e30f;
                        LD[t * 64 + 32 + rl] = row < N ? part : 0.f;
                        if (row < N && delta_out) delta_out[((size_t)b * H + h) * N + row] = part;
                    }
                }
            }
        }
    }
    f32x16 dk0 = zero16(), dk1 = zero16(), dv0 = zero16(), dv1 = zero16(), q0 = zero16(), q1 = zero16();
    FUSED_STAMP(1);
    __syncthreads();
    FUSED_STAMP(2);

    const bool active = wave < nt;                             // waves beyond the last tile only keep the barriers company
    unsigned char* Sw = SC + wave * (32 * SCR);
    const unsigned char* Kw = KT + wave * TILE;
    for (int st = 0; st < nt; ++st) {
        if (active) {
            int qt = wave + st;
            qt = qt >= nt ? qt - nt : qt;
            const unsigned char* Qt = QT + qt * TILE;
            const unsigned char* Ot = OT + qt * TILE;
            const float* Lt = LD + qt * 64;
            const float* Dt = Lt + 32;
            f32x16 s = zero16(), dpv = zero16();
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag(Qt, ks, lane), row_frag(Kw, ks, lane), s, 0, 0, 0);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) dpv = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag(Ot, ks, lane), vf[ks], dpv, 0, 0, 0);
            float p[16], ds[16];
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const f32x4 lv = *(const f32x4*)(Lt + 8 * rg + 4 * hh);
                const f32x4 dv = *(const f32x4*)(Dt + 8 * rg + 4 * hh);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int r = 4 * rg + e;
                    const float pr = fast_exp2(s[r] * c - lv[e]);
                    p[r] = pr;
                    ds[r] = pr * (dpv[r] - dv[e]);
                }
            }
            const bf16x8 pf0 = pack_frag(p, 0), pf1 = pack_frag(p, 1);
            const bf16x8 sf0 = pack_frag(ds, 0), sf1 = pack_frag(ds, 1);
            // dS tile to the scratch as [key = lane & 31][query]: registers 4 rg .. 4 rg + 3 are queries 8 rg + 4 hh .. + 3;
            // a key beyond N must not reach dQ (its P is not masked in this orientation)
            {
                const int kr = lane & 31;
                const u32x4 w4 = __builtin_bit_cast(u32x4, sf0), w5 = __builtin_bit_cast(u32x4, sf1);
                const u32x2 z = {0u, 0u};
                const u32x2 w[4] = {kvalid ? u32x2{w4[0], w4[1]} : z, kvalid ? u32x2{w4[2], w4[3]} : z,
                                    kvalid ? u32x2{w5[0], w5[1]} : z, kvalid ? u32x2{w5[2], w5[3]} : z};
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) *(u32x2*)(Sw + kr * SCR + ((rg ^ (swz(kr) & 3)) << 4) + 8 * hh) = w[rg];
            }
            dv0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag(Ot, 0, 0, lane), pf0, dv0, 0, 0, 0);
            dv0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag(Ot, 1, 0, lane), pf1, dv0, 0, 0, 0);
            dv1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag(Ot, 0, 1, lane), pf0, dv1, 0, 0, 0);
            dv1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag(Ot, 1, 1, lane), pf1, dv1, 0, 0, 0);
            dk0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag(Qt, 0, 0, lane), sf0, dk0, 0, 0, 0);
            dk0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag(Qt, 1, 0, lane), sf1, dk0, 0, 0, 0);
            dk1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag(Qt, 0, 1, lane), sf0, dk1, 0, 0, 0);
            dk1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag(Qt, 1, 1, lane), sf1, dk1, 0, 0, 0);
        }
        if (st == 0) FUSED_STAMP(3);
        __syncthreads();                                       // every wave's dS of this step is in its scratch
        if (st == 0) FUSED_STAMP(4);
        if (active) {
            // dQ_wave^T [d][query] += K_j^T [d][key] . dS_(wave, j) [key][query],  j = the wave that worked on this query tile
            int j = wave - st;
            j = j < 0 ? j + nt : j;
            const unsigned char* Sj = SC + j * (32 * SCR);
            const unsigned char* Kj = KT + j * TILE;
            const bf16x8 t0 = tr_frag_scratch(Sj, 0, lane), t1 = tr_frag_scratch(Sj, 1, lane);
            q0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag(Kj, 0, 0, lane), t0, q0, 0, 0, 0);
            q0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag(Kj, 1, 0, lane), t1, q0, 0, 0, 0);
            q1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag(Kj, 0, 1, lane), t0, q1, 0, 0, 0);
            q1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag(Kj, 1, 1, lane), t1, q1, 0, 0, 0);
        }
        if (st + 1 < nt) __syncthreads();                      // the scratches are free again
        if (st == 0) FUSED_STAMP(5);
    }
    FUSED_STAMP(6);
    if (!active) return;
    if (kvalid) {       // query tile `wave` has the same rows as key tile `wave`
        bf16_t* drow = dqkv + ((size_t)b * N + ki) * lddqkv + h * HD;
        store_T(drow, q0, q1, scale, hh);
        store_T(drow + D, dk0, dk1, scale, hh);
        store_T(drow + 2 * D, dv0, dv1, 1.0f, hh);
    }
}

// delta[b,h,q] = sum_d dO[q, h*64+d] * O[q, h*64+d]: 8 lanes per (token, head), 16-B loads, 3-step shuffle reduce.
// Its own kernel so that the dQ pass and the dK/dV pass (which both need it) can run concurrently on two streams.
__global__ __launch_bounds__(256) void attn_delta_kernel(const bf16_t* __restrict__ out, int ldo, const bf16_t* __restrict__ dout,
                                                         int lddo, int N, int H, long long pairs, float* __restrict__ delta) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long pr = i >> 3;
    const int ch = (int)(i & 7);
    float part = 0.f;
    if (pr < pairs) {
        const long long tok = pr / H;
        const int h = (int)(pr - tok * H);
        const bf16x8 a = *(const bf16x8*)(dout + tok * lddo + h * HD + ch * 8);
        const bf16x8 o = *(const bf16x8*)(out + tok * ldo + h * HD + ch * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) part += (float)a[j] * (float)o[j];
    }
    part += __shfl_xor(part, 1, 64);
    part += __shfl_xor(part, 2, 64);
    part += __shfl_xor(part, 4, 64);
    if (pr < pairs && ch == 0) {
        const long long tok = pr / H;
        const int h = (int)(pr - tok * H);
        const long long b = tok / N, q = tok - b * N;
        delta[((size_t)b * H + h) * N + q] = part;
    }
}

int pick_nw(int N) {
    static int forced = -1;
    if (forced < 0) {
        const char* e = getenv("MOFO_ATTN_NW");
        forced = e ? atoi(e) : 0;
    }
    const int t = (N + 31) / 32;  // 32-row wave tiles
    if (t <= 5) return 5;         // short sequences: one block per (clip, head)
    if (forced == 4 || forced == 7) return forced;
    return 4;                     // 3 blocks/CU by waves; measured 3 % faster than 7-wave blocks even with a 6 % ragged tail
}
}  // namespace

#define LAUNCH_Q(NW, MODE) do { if (N <= 160) { LAUNCH_Q_(NW, MODE, true); } else { LAUNCH_Q_(NW, MODE, false); } } while (0)
#define LAUNCH_Q_(NW, MODE, WH)                                                                                         \
    hipLaunchKernelGGL((attn_q_kernel<NW, MODE, WH>), dim3(8 * ceil_div(B * H, 8) * ceil_div(N, 32 * NW)), dim3(NW * 64), 0, s, \
                       (const bf16_t*)qkv, ldqkv, ceil_div(N, 32 * NW), B * H, N, H, c, scale, (bf16_t*)out, ldo, (float*)lse2, \
                       (const bf16_t*)dout, lddo, (bf16_t*)dqkv, lddqkv, delta)

static int check_common(const char* who, const void* qkv, int ldqkv, int B, int N, int H) {
    if (!qkv) MOFO_FAIL(MOFO_EINVAL, "%s: null qkv", who);
    if (B <= 0 || N <= 0 || H <= 0) MOFO_FAIL(MOFO_EINVAL, "%s: bad dims B=%d N=%d H=%d", who, B, N, H);
    if (ldqkv < 3 * H * 64 || ldqkv % 8) MOFO_FAIL(MOFO_EUNSUPPORTED, "%s: ldqkv=%d must be >= 3*H*64 and a multiple of 8", who, ldqkv);
    if ((long long)B * H * ((N + 127) / 128) > (1LL << 28)) MOFO_FAIL(MOFO_EUNSUPPORTED, "%s: grid too large", who);
    return MOFO_OK;
}

extern "C" int mofo_attention_fwd(const void* qkv, int ldqkv, int B, int N, int H, float scale, void* out, int ldo,
                                  float* lse2, void* stream) {
    int rc = check_common("mofo_attention_fwd", qkv, ldqkv, B, N, H);
    if (rc) return rc;
    if (!out || !lse2) MOFO_FAIL(MOFO_EINVAL, "mofo_attention_fwd: null output");
    if (ldo < H * 64 || ldo % 4) MOFO_FAIL(MOFO_EUNSUPPORTED, "mofo_attention_fwd: bad ldo=%d", ldo);
    hipStream_t s = (hipStream_t)stream;
    const float c = scale * 1.4426950408889634f;
    const void* dout = nullptr; int lddo = 0; void* dqkv = nullptr; int lddqkv = 0; float* delta = nullptr;
    switch (pick_nw(N)) {
        case 7: LAUNCH_Q(7, 0); break;
        case 5: LAUNCH_Q(5, 0); break;
        default: LAUNCH_Q(4, 0); break;
    }
    MOFO_CHECK_LAUNCH("mofo_attention_fwd");
    return MOFO_OK;
}

static int bwd_check(const char* who, const void* qkv, int ldqkv, const void* dout, int lddo, const float* lse2, const float* delta,
                     int B, int N, int H, void* dqkv, int lddqkv) {
    int rc = check_common(who, qkv, ldqkv, B, N, H);
    if (rc) return rc;
    if (!dout || !lse2 || !dqkv || !delta) MOFO_FAIL(MOFO_EINVAL, "%s: null pointer", who);
    if (lddo % 8 || lddqkv % 4 || lddqkv < 3 * H * 64) MOFO_FAIL(MOFO_EUNSUPPORTED, "%s: bad leading dims", who);
    return MOFO_OK;
}

extern "C" int mofo_attention_delta(const void* out, int ldo, const void* dout, int lddo, int B, int N, int H, float* delta, void* stream) {
    if (!out || !dout || !delta) MOFO_FAIL(MOFO_EINVAL, "mofo_attention_delta: null pointer");
    if (B <= 0 || N <= 0 || H <= 0 || ldo % 8 || lddo % 8) MOFO_FAIL(MOFO_EINVAL, "mofo_attention_delta: bad sizes");
    const long long pairs = (long long)B * N * H;
    hipLaunchKernelGGL(attn_delta_kernel, dim3((unsigned)((pairs * 8 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)out, ldo,
                       (const bf16_t*)dout, lddo, N, H, pairs, delta);
    MOFO_CHECK_LAUNCH("mofo_attention_delta");
    return MOFO_OK;
}

extern "C" int mofo_attention_bwd_dq(const void* qkv, int ldqkv, const void* dout, int lddo, const float* lse2_in, const float* delta_in,
                                     int B, int N, int H, float scale, void* dqkv, int lddqkv, void* stream) {
    int rc = bwd_check("mofo_attention_bwd_dq", qkv, ldqkv, dout, lddo, lse2_in, delta_in, B, N, H, dqkv, lddqkv);
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    const float c = scale * 1.4426950408889634f;
    float* lse2 = const_cast<float*>(lse2_in);
    float* delta = const_cast<float*>(delta_in);
    void* out = nullptr; int ldo = 0;
    switch (pick_nw(N)) {
        case 7: LAUNCH_Q(7, 1); break;
        case 5: LAUNCH_Q(5, 1); break;
        default: LAUNCH_Q(4, 1); break;
    }
    MOFO_CHECK_LAUNCH("mofo_attention_bwd_dq");
    return MOFO_OK;
}

extern "C" int mofo_attention_bwd_dkv(const void* qkv, int ldqkv, const void* dout, int lddo, const float* lse2, const float* delta,
                                      int B, int N, int H, float scale, void* dqkv, int lddqkv, void* stream) {
    int rc = bwd_check("mofo_attention_bwd_dkv", qkv, ldqkv, dout, lddo, lse2, delta, B, N, H, dqkv, lddqkv);
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    const float c = scale * 1.4426950408889634f;
#define LAUNCH_KV(NW) do { if (N <= 160) { LAUNCH_KV_(NW, true); } else { LAUNCH_KV_(NW, false); } } while (0)
#define LAUNCH_KV_(NW, WH)                                                                                             \
    hipLaunchKernelGGL((attn_dkv_kernel<NW, WH>), dim3(8 * ceil_div(B * H, 8) * ceil_div(N, 32 * NW)), dim3(NW * 64), 0, s,   \
                       (const bf16_t*)qkv, ldqkv, ceil_div(N, 32 * NW), B * H, N, H, c, scale, (const bf16_t*)dout, lddo,      \
                       (const float*)lse2, (const float*)delta, (bf16_t*)dqkv, lddqkv)
    switch (pick_nw(N)) {
        case 7: LAUNCH_KV(7); break;
        case 5: LAUNCH_KV(5); break;
        default: LAUNCH_KV(4); break;
    }
    MOFO_CHECK_LAUNCH("mofo_attention_bwd_dkv");
    return MOFO_OK;
}

extern "C" int mofo_attention_bwd(const void* qkv, int ldqkv, const void* out, int ldo, const void* dout, int lddo,
                                  const float* lse2, int B, int N, int H, float scale, void* dqkv, int lddqkv,
                                  float* delta, void* stream) {
    int rc;
    if (N <= 160 && !getenv("MOFO_ATTN_NO_FUSED_BWD")) {
        rc = bwd_check("mofo_attention_bwd", qkv, ldqkv, dout, lddo, lse2, delta, B, N, H, dqkv, lddqkv);
        if (rc) return rc;
        if (!out || ldo % 8) MOFO_FAIL(MOFO_EINVAL, "mofo_attention_bwd: bad out / ldo");
        static bool attr_set = false;
        if (!attr_set) {
            if (hipFuncSetAttribute((const void*)attn_bwd_fused_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, FUSED_SMEM) != hipSuccess)
                MOFO_FAIL(MOFO_ERUNTIME, "mofo_attention_bwd: cannot reserve %d bytes of LDS", FUSED_SMEM);
            attr_set = true;
        }
        const float c = scale * 1.4426950408889634f;
        hipLaunchKernelGGL(attn_bwd_fused_kernel, dim3(8 * ceil_div(B * H, 8)), dim3(FUSED_NW * 64), FUSED_SMEM, (hipStream_t)stream,
                           (const bf16_t*)qkv, ldqkv, B * H, N, H, c, scale, (const bf16_t*)out, ldo, (const bf16_t*)dout, lddo, lse2, delta,
                           (bf16_t*)dqkv, lddqkv);
        MOFO_CHECK_LAUNCH("mofo_attention_bwd(fused)");
        return MOFO_OK;
    }
    rc = mofo_attention_delta(out, ldo, dout, lddo, B, N, H, delta, stream);
    if (rc) return rc;
    rc = mofo_attention_bwd_dq(qkv, ldqkv, dout, lddo, lse2, delta, B, N, H, scale, dqkv, lddqkv, stream);
    if (rc) return rc;
    return mofo_attention_bwd_dkv(qkv, ldqkv, dout, lddo, lse2, delta, B, N, H, scale, dqkv, lddqkv, stream);
}

#ifdef MOFO_ATTN_TRACE
extern "C" int mofo_debug_attn_trace_read(void* dst_host, size_t bytes) {
    return hipMemcpyFromSymbol(dst_host, HIP_SYMBOL(g_attn_trace), bytes) == hipSuccess ? 0 : MOFO_ERUNTIME;
}
#endif
