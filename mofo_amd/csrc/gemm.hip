// bf16 MFMA GEMM family for gfx950 (MI355X): one kernel template, three operand-layout pairs.
//
//   NT  C[m,n] = sum_k A[m,k] * B[n,k]     forward Linear (x @ W^T); both operands reduction-contiguous
//   NN  C[m,n] = sum_k A[m,k] * B[k,n]     dgrad  (dY @ W);   B is read "reduction-strided"
//   TN  C[m,n] = sum_k A[k,m] * B[k,n]     wgrad  (dY^T @ X); both operands reduction-strided
//
// Tile 128x128x64, 256 threads = 4 waves (2x2), each wave 64x64 = 4x4 tiles of v_mfma_f32_16x16x32_bf16.
// Operands go HBM -> LDS with global_load_lds_dwordx4 (no VGPR round trip), two LDS buffers.
//   ROW operand (reduction contiguous): LDS image [128 rows][64 k] bf16 (128-B rows); 16-B chunk c of row r
//       sits at chunk position c ^ (r & 7)  -> ds_read_b128 fragment reads are bank-conflict free.
//   COL operand (reduction strided):   LDS image [64 k][128 cols] bf16 (256-B rows); 32-B unit u of k-row r
//       sits at unit position u ^ key(r), key(r) = (r&3) | ((r>>3)&1)<<2  -> ds_read_b64_tr_b16 (hardware
//       transpose read) fragment reads are bank-conflict free.
//   LDS-DMA writes are lane-linear, so both swizzles are applied to the per-lane SOURCE address.
// The MFMA is issued as mfma(Bfrag, Afrag) so that a lane ends up holding 4 CONSECUTIVE n of one m:
//   acc[mt][nt][j] = C[m0 + 16 mt + (lane & 15)][n0 + 16 nt + 4 (lane >> 4) + j]   (8-B bf16 / 16-B f32 stores)
#include "common.h"
#include "../../include/mofo_hip.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = 128 * 64 * 2;  // 16 KiB per operand tile
constexpr int OPL_ROW = 0, OPL_COL = 1;

__device__ __attribute__((aligned(256))) unsigned int g_zero_page[64];  // 256 B of zeros: source of padded k-rows

struct GemmP {
    const bf16_t* A; const bf16_t* B;
    void* C; void* C2;
    const float* bias; const float* resid; const bf16_t* aux; const float* pos; const int* row_idx;
    int M, N, K;
    int lda, ldb, ldc, ldc2, ldr, ldaux, ldpos;
    int rows_in, rows_out, row_off;
    int k_per_split;
    int atomic;
};

__device__ __forceinline__ int col_key(int krow) { return (krow & 3) | (((krow >> 3) & 1) << 2); }

// Issue the LDS-DMA loads of one operand tile.  `dim` = extent of the non-reduction index, `kend` = end of the
// reduction range of this block.
template <int LAYOUT>
__device__ __forceinline__ void stage_tile(const bf16_t* __restrict__ base, int ld, int dim, int d0, int k0, int kend,
                                           unsigned char* lds_tile, int wave, int lane) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int i = wave * 4 + j;  // wave-instruction index 0..15, 1 KiB each
        const bf16_t* src;
        if constexpr (LAYOUT == OPL_ROW) {
            const int rl = 8 * i + (lane >> 3);
            int row = d0 + rl;
            row = row < dim ? row : dim - 1;  // clamp: garbage rows are never stored
            const int gch = (lane & 7) ^ ((lane >> 3) & 7);
            src = base + (size_t)row * ld + k0 + gch * 8;
        } else {
            const int kr = 4 * i + (lane >> 4);
            const int cpos = lane & 15;
            const int gch = ((((cpos >> 1) ^ col_key(kr)) << 1) | (cpos & 1));
            int col = d0 + gch * 8;
            col = col <= dim - 8 ? col : dim - 8;
            const int k = k0 + kr;
            src = (k < kend) ? base + (size_t)k * ld + col
                             : (const bf16_t*)((const unsigned char*)g_zero_page + cpos * 16);
        }
        __builtin_amdgcn_global_load_lds(GLB_PTR(src), LDS_PTR(lds_tile + i * 1024), 16, 0, 0);
    }
}

template <int LAYOUT>
__device__ __forceinline__ bf16x8 read_frag(const unsigned char* lds_tile, int sub0, int ks, int lane) {
    if constexpr (LAYOUT == OPL_ROW) {
        const int row = sub0 + (lane & 15);
        const int kc = 4 * ks + (lane >> 4);
        return *(const bf16x8*)(lds_tile + row * 128 + ((kc ^ (row & 7)) << 4));
    } else {
        const int g = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
        const int unit = sub0 >> 4;
        const int kr0 = 32 * ks + 8 * g + q;
        const int kr1 = kr0 + 4;
        const unsigned char* a0 = lds_tile + kr0 * 256 + ((unit ^ col_key(kr0)) << 5) + 8 * pp;
        const unsigned char* a1 = lds_tile + kr1 * 256 + ((unit ^ col_key(kr1)) << 5) + 8 * pp;
        s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)LDS_PTR(a0));
        s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)LDS_PTR(a1));
        s16x8 r;
        r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
        r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
        return __builtin_bit_cast(bf16x8, r);
    }
}

template <int LA, int LB, int EPI>
__global__ __launch_bounds__(256) void gemm_kernel(GemmP p) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[4 * TILE_BYTES];  // [buf][A|B]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    // XCD-aware remap (8 XCDs, private L2s): blocks b and b+8 share an XCD, so give each XCD a contiguous
    // run of tiles with n fastest; the B panel (weights) and one A row-panel then stay L2-resident per XCD.
    const int tiles_n = (p.N + BN - 1) / BN;
    const int nwg = gridDim.x;
    int wg = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = wg & 7;
        wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (wg >> 3);
    }
    const int m0 = (wg / tiles_n) * BM, n0 = (wg % tiles_n) * BN;
    const int kbeg = blockIdx.z * p.k_per_split;
    const int kend = min(p.K, kbeg + p.k_per_split);
    const int nk = (kend - kbeg + BK - 1) / BK;

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    auto stage = [&](int t, int buf) {
        unsigned char* ta = smem + buf * 2 * TILE_BYTES;
        stage_tile<LA>(p.A, p.lda, p.M, m0, kbeg + t * BK, kend, ta, wave, lane);
        stage_tile<LB>(p.B, p.ldb, p.N, n0, kbeg + t * BK, kend, ta + TILE_BYTES, wave, lane);
    };

    if (nk > 0) stage(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int cur = 0;
    for (int t = 0; t < nk; ++t) {
        if (t + 1 < nk) stage(t + 1, cur ^ 1);
        const unsigned char* ta = smem + cur * 2 * TILE_BYTES;
        const unsigned char* tb = ta + TILE_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 af[4], bfr[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) af[i] = read_frag<LA>(ta, wm * 64 + 16 * i, ks, lane);
#pragma unroll
            for (int i = 0; i < 4; ++i) bfr[i] = read_frag<LB>(tb, wn * 64 + 16 * i, ks, lane);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        cur ^= 1;
    }

    // ------------------------------------------------------------------ epilogue
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + wm * 64 + 16 * i + (lane & 15);
        if (m >= p.M) continue;
        int orow = m;
        const float* posrow = nullptr;
        if constexpr (EPI == MOFO_EPI_POS_F32) {
            orow = (m / p.rows_in) * p.rows_out + p.row_off + (m % p.rows_in);
            posrow = p.pos + (size_t)p.row_idx[m] * p.ldpos;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + wn * 64 + 16 * j + 4 * (lane >> 4);
            if (n >= p.N) continue;
            f32x4 v = acc[i][j];
            if constexpr (EPI != MOFO_EPI_F32) {
                if (p.bias) {
                    const f32x4 b = *(const f32x4*)(p.bias + n);
                    v += b;
                }
            }
            if constexpr (EPI == MOFO_EPI_BF16) {
                u32x2 o = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
                *(u32x2*)((bf16_t*)p.C + (size_t)orow * p.ldc + n) = o;
            } else if constexpr (EPI == MOFO_EPI_BIAS_GELU) {
                u32x2 o = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
                *(u32x2*)((bf16_t*)p.C + (size_t)orow * p.ldc + n) = o;
                u32x2 g = {pack_bf16x2(gelu_erf(v[0]), gelu_erf(v[1])), pack_bf16x2(gelu_erf(v[2]), gelu_erf(v[3]))};
                *(u32x2*)((bf16_t*)p.C2 + (size_t)orow * p.ldc2 + n) = g;
            } else if constexpr (EPI == MOFO_EPI_RESID_F32) {
                const f32x4 r = *(const f32x4*)(p.resid + (size_t)m * p.ldr + n);
                v += r;
                *(f32x4*)((float*)p.C + (size_t)orow * p.ldc + n) = v;
            } else if constexpr (EPI == MOFO_EPI_POS_F32) {
                const f32x4 r = *(const f32x4*)(posrow + n);
                v += r;
                *(f32x4*)((float*)p.C + (size_t)orow * p.ldc + n) = v;
            } else if constexpr (EPI == MOFO_EPI_DGELU_BF16) {
                const u32x2 h = *(const u32x2*)(p.aux + (size_t)m * p.ldaux + n);
                v[0] *= dgelu_erf(bf16lo_to_f32(h[0]));
                v[1] *= dgelu_erf(bf16hi_to_f32(h[0]));
                v[2] *= dgelu_erf(bf16lo_to_f32(h[1]));
                v[3] *= dgelu_erf(bf16hi_to_f32(h[1]));
                u32x2 o = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
                *(u32x2*)((bf16_t*)p.C + (size_t)orow * p.ldc + n) = o;
            } else {  // MOFO_EPI_F32
                float* dst = (float*)p.C + (size_t)orow * p.ldc + n;
                if (p.atomic) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) atomicAdd(dst + e, v[e]);
                } else {
                    *(f32x4*)dst = v;
                }
            }
        }
    }
}

template <int LA, int LB, int EPI>
int launch(const GemmP& p, int splits, hipStream_t s) {
    const int tiles = ceil_div(p.M, BM) * ceil_div(p.N, BN);
    hipLaunchKernelGGL((gemm_kernel<LA, LB, EPI>), dim3(tiles, 1, splits), dim3(256), 0, s, p);
    MOFO_CHECK_LAUNCH("mofo_gemm");
    return MOFO_OK;
}

}  // namespace

extern "C" int mofo_gemm(const mofo_gemm_args* a, void* stream) {
    if (!a || !a->A || !a->B || !a->C) MOFO_FAIL(MOFO_EINVAL, "mofo_gemm: null operand");
    if (a->M <= 0 || a->N <= 0 || a->K <= 0) MOFO_FAIL(MOFO_EINVAL, "mofo_gemm: non-positive dims %d %d %d", a->M, a->N, a->K);
    const int op = a->op, epi = a->epilogue;
    // alignment / divisibility the kernels are built for
    if (a->N % 8 || a->lda % 8 || a->ldb % 8) MOFO_FAIL(MOFO_EUNSUPPORTED, "mofo_gemm: N, lda, ldb must be multiples of 8");
    if (op == MOFO_GEMM_NT && a->K % 64) MOFO_FAIL(MOFO_EUNSUPPORTED, "mofo_gemm NT: K=%d must be a multiple of 64", a->K);
    if (op == MOFO_GEMM_NN && a->K % 64) MOFO_FAIL(MOFO_EUNSUPPORTED, "mofo_gemm NN: K=%d must be a multiple of 64", a->K);
    if (op == MOFO_GEMM_TN && a->M % 8) MOFO_FAIL(MOFO_EUNSUPPORTED, "mofo_gemm TN: M=%d must be a multiple of 8", a->M);
    if (a->ldc % 4) MOFO_FAIL(MOFO_EUNSUPPORTED, "mofo_gemm: ldc must be a multiple of 4");
    int splits = a->splits < 1 ? 1 : a->splits;
    if (splits > 1 && epi != MOFO_EPI_F32) MOFO_FAIL(MOFO_EUNSUPPORTED, "mofo_gemm: split-K only with the f32 accumulate epilogue");
    if (epi == MOFO_EPI_BIAS_GELU && !a->C2) MOFO_FAIL(MOFO_EINVAL, "mofo_gemm: BIAS_GELU needs C2");
    if (epi == MOFO_EPI_RESID_F32 && !a->resid) MOFO_FAIL(MOFO_EINVAL, "mofo_gemm: RESID_F32 needs resid");
    if (epi == MOFO_EPI_DGELU_BF16 && !a->aux) MOFO_FAIL(MOFO_EINVAL, "mofo_gemm: DGELU needs aux");
    if (epi == MOFO_EPI_POS_F32 && (!a->pos || !a->row_idx || a->rows_in <= 0)) MOFO_FAIL(MOFO_EINVAL, "mofo_gemm: POS_F32 needs pos,row_idx,rows_in");

    GemmP p;
    p.A = (const bf16_t*)a->A; p.B = (const bf16_t*)a->B; p.C = a->C; p.C2 = a->C2;
    p.bias = a->bias; p.resid = a->resid; p.aux = (const bf16_t*)a->aux; p.pos = a->pos; p.row_idx = a->row_idx;
    p.M = a->M; p.N = a->N; p.K = a->K;
    p.lda = a->lda; p.ldb = a->ldb; p.ldc = a->ldc; p.ldc2 = a->ldc2; p.ldr = a->ldr; p.ldaux = a->ldaux; p.ldpos = a->ldpos;
    p.rows_in = a->rows_in; p.rows_out = a->rows_out; p.row_off = a->row_off;
    int kps = ceil_div(ceil_div(a->K, splits), BK) * BK;
    splits = ceil_div(a->K, kps);
    p.k_per_split = kps;
    p.atomic = (splits > 1 || a->accumulate) ? 1 : 0;
    hipStream_t s = (hipStream_t)stream;

#define GO(LA, LB, E) return launch<LA, LB, E>(p, splits, s)
    if (op == MOFO_GEMM_NT) {
        switch (epi) {
            case MOFO_EPI_BF16: GO(OPL_ROW, OPL_ROW, MOFO_EPI_BF16);
            case MOFO_EPI_BIAS_GELU: GO(OPL_ROW, OPL_ROW, MOFO_EPI_BIAS_GELU);
            case MOFO_EPI_RESID_F32: GO(OPL_ROW, OPL_ROW, MOFO_EPI_RESID_F32);
            case MOFO_EPI_POS_F32: GO(OPL_ROW, OPL_ROW, MOFO_EPI_POS_F32);
            case MOFO_EPI_F32: GO(OPL_ROW, OPL_ROW, MOFO_EPI_F32);
        }
    } else if (op == MOFO_GEMM_NN) {
        switch (epi) {
            case MOFO_EPI_BF16: GO(OPL_ROW, OPL_COL, MOFO_EPI_BF16);
            case MOFO_EPI_DGELU_BF16: GO(OPL_ROW, OPL_COL, MOFO_EPI_DGELU_BF16);
            case MOFO_EPI_F32: GO(OPL_ROW, OPL_COL, MOFO_EPI_F32);
        }
    } else if (op == MOFO_GEMM_TN) {
        switch (epi) {
            case MOFO_EPI_F32: GO(OPL_COL, OPL_COL, MOFO_EPI_F32);
            case MOFO_EPI_BF16: GO(OPL_COL, OPL_COL, MOFO_EPI_BF16);
        }
    }
#undef GO
    MOFO_FAIL(MOFO_EUNSUPPORTED, "mofo_gemm: op %d with epilogue %d is not built", op, epi);
}
