// bf16 MFMA GEMM family for gfx950 (MI355X): one kernel template, three operand-layout pairs.
//
//   NT  C[m,n] = sum_k A[m,k] * B[n,k]     forward Linear (x @ W^T); both operands reduction-contiguous
//   NN  C[m,n] = sum_k A[m,k] * B[k,n]     dgrad  (dY @ W);   B is read "reduction-strided"
//   TN  C[m,n] = sum_k A[k,m] * B[k,n]     wgrad  (dY^T @ X); both operands reduction-strided
//
// Tile 128x128x64, 256 threads = 4 waves (2x2), each wave 64x64 = 4x4 tiles of v_mfma_f32_16x16x32_bf16.
// Operands go HBM -> LDS with buffer_load_dwordx4 ... lds (no VGPR round trip; SRD + 32-bit lane offset + SGPR tile offset).
//   ROW operand (reduction contiguous): LDS image [128 rows][64 k] bf16 (128-B rows); 16-B chunk c of row r
//       sits at chunk position c ^ (r & 7)  -> ds_read_b128 fragment reads are bank-conflict free.
//   COL operand (reduction strided):   LDS image [64 k][128 cols] bf16 (256-B rows); 32-B unit u of k-row r
//       sits at unit position u ^ key(r), key(r) = (r&3) | ((r>>3)&1)<<2  -> ds_read_b64_tr_b16 (hardware
//       transpose read) fragment reads are bank-conflict free.
//   LDS-DMA writes are lane-linear, so both swizzles are applied to the per-lane SOURCE offset.
// The MFMA is issued as mfma(Bfrag, Afrag) so that a lane ends up holding 4 CONSECUTIVE n of one m:
//   acc[mt][nt][j] = C[m0 + 16 mt + (lane & 15)][n0 + 16 nt + 4 (lane >> 4) + j]
// EPILOGUE: the fragment layout touches 16 different rows per store instruction (32-B pieces), which made the short-K
// GEMMs of the decoder bound by the store path (~1.6 TB/s).  Every epilogue therefore stages the wave's 64x64 f32 tile
// through LDS (free after the main loop; float4 slot c4 of row r at c4 ^ (r & 15), conflict-free) and then reads /
// writes global memory in whole row segments: 16 B per lane, 8 rows x 128 B (bf16) or 4 rows x 256 B (f32) per
// wave-instruction; bias / residual / pre-activation operands are read the same way.
// GROUPED launches: up to MAXG problems of one (op, epilogue) kind in ONE grid (blockIdx -> problem by a prefix table);
// the four weight-gradient GEMMs of a transformer block are issued together so that 108+36+144+144 tiles fill 256 CUs
// without split-K (every split costs another f32 atomic pass over the gradient); three encoder blocks' twelve together are
// 1296 tiles = 5.06 per CU instead of 1.69 (tools/wgrad_group_exp.py: 89 -> 75 us per block's worth, 812 -> 963 TFLOP/s).
#include "common.h"
#include "../../include/mofo_hip.h"
#include <stdlib.h>
#include <type_traits>
#include <utility>

namespace {

constexpr int BN = 128, BK = 64;
constexpr int TILE_BYTES = 128 * 64 * 2;  // 16 KiB per operand tile
constexpr int OPL_ROW = 0, OPL_COL = 1;


struct GemmP {
    const bf16_t* A; const bf16_t* B;
    void* C; void* C2;
    const float* bias; const float* resid; const bf16_t* aux; const float* pos; const int* row_idx;
    int M, N, K;
    int lda, ldb, ldc, ldc2, ldr, ldaux, ldpos;
    int rows_in, rows_out, row_off;
    int k_per_split;
    int atomic;
    float* colsum;            // TN only: colsum[m] += sum_k A[k,m]  (bias gradient riding on the wgrad GEMM), or null
    int colsum_skip_lo, colsum_skip_hi;   // rows m in [lo,hi) are not written (the k third of the fused qkv bias)
    int rotate;               // persistent form: rotated reduction order per tile (see gemm_persistent_kernel)
    const float* a_scale_inv; const float* b_scale_inv;   // NT_FP8: per-tensor de-quantisation factors (device scalars)
    unsigned char* C8; int ldc8;          // NT_FP8 + BIAS_GELU: e4m3 copy of the activation (C2), sat(gelu * q_scale[0]); or null
    const float* q_scale; float* q_amax;  // ... its scale (device scalar) and the MOFO_FP8_AMAX_STRIPES stripes that collect max|gelu|
    int aux_nt;               // DGELU: read the saved pre-activation with non-temporal loads (set by size)
    int rotate_tile;          // the same in the one-tile-per-block kernel (on; round 1): small-grid
                              // residual GEMMs 1.39 -> 1.35 ms/step, wgrad neutral
};

constexpr int MAXG = 13;   // three transformer blocks' four weight gradients + one more (patch embed / head) in one launch (runtime.py: MOFO_WGRAD_BLOCKS)
struct GroupP {
    GemmP p[MAXG];
    int start[MAXG + 1];   // first block of each problem; start[count] = grid size
    int count;
};

// Debug build only (-DMOFO_GEMM_TRACE, tools/gemm_trace.py): per-block phase timestamps (s_memtime) + the CU the block ran on.
#ifdef MOFO_GEMM_TRACE
__device__ unsigned long long g_trace[1 << 19];
#define MOFO_TRACE(slot)                                                                          \
    do {                                                                                          \
        if (threadIdx.x == 0 && blockIdx.x < (1 << 16)) g_trace[blockIdx.x * 8 + (slot)] = __builtin_readcyclecounter(); \
    } while (0)
#define MOFO_TRACE_ID()                                                                           \
    do {                                                                                          \
        if (threadIdx.x == 0 && blockIdx.x < (1 << 16))                                           \
            g_trace[blockIdx.x * 8 + 7] = ((unsigned long long)__builtin_amdgcn_s_getreg(6164) << 32) | __builtin_amdgcn_s_getreg(63492); \
    } while (0)
// stamps inside main-loop iteration 2 of wave 0 (slots 0..5 of g_trace_it)
__device__ unsigned long long g_trace_it[1 << 19];
#define MOFO_TRACE_IT(t, slot)                                                                    \
    do {                                                                                          \
        if ((t) == 2 && threadIdx.x == 0 && blockIdx.x < (1 << 16)) g_trace_it[blockIdx.x * 8 + (slot)] = __builtin_readcyclecounter(); \
    } while (0)
#else
#define MOFO_TRACE(slot)
#define MOFO_TRACE_ID()
#define MOFO_TRACE_IT(t, slot)
#endif

__device__ __forceinline__ int col_key(int krow) { return (krow & 3) | (((krow >> 3) & 1) << 2); }

// Issue the LDS-DMA loads of one operand tile through a buffer resource (SRD): `buffer_load_dwordx4 ... offen lds` takes the per-lane part of the
// address as ONE 32-bit VGPR offset that is the same for every piece and every k-stage, the tile / piece / k position as a
// wave-uniform SGPR offset, and range-checks against the operand's extent (rows beyond M or N and k-rows beyond K read as
// zeros: no clamps, no zero page).  The first version used flat `global_load_lds` with a 64-bit per-lane address per piece: 16
// address VGPRs per operand live across the main loop, and 2-3x the issue time for the reduction-strided (COL) pieces.
// Piece i of a tile is 1 KiB: ROW rows 8i..8i+7 (8 lanes x 16 B per row), COL k-rows 4i..4i+3 (16 lanes x 16 B per row).
// ROW: voff = ((lane>>3) * ld + ((lane&7) ^ ((lane>>3)&7)) * 8) * 2 bytes.
// COL: the swizzle key of a piece's k-rows has one bit that depends on the piece, (i>>1)&1, so two offsets alternate:
//      voff[h] = ((lane>>4) * ld + gch_h * 8) * 2,  gch_h = ((((lane&15)>>1) ^ ((lane>>4) | h<<2)) << 1) | (lane&1).
template <int LAYOUT>
__device__ __forceinline__ void srd_lane_offsets(int ld, int lane, int& v0, int& v1) {
    if constexpr (LAYOUT == OPL_ROW) {
        v0 = v1 = ((lane >> 3) * ld + (((lane & 7) ^ ((lane >> 3) & 7)) << 3)) * 2;
    } else {
        const int cpos = lane & 15, kq = lane >> 4;
        const int g0 = ((((cpos >> 1) ^ kq) << 1) | (cpos & 1));
        const int g1 = ((((cpos >> 1) ^ (kq | 4)) << 1) | (cpos & 1));
        v0 = (kq * ld + g0 * 8) * 2;
        v1 = (kq * ld + g1 * 8) * 2;
    }
}
// LDS-DMA piece issued from INLINE ASM: 1 KiB = 64 lanes x 16 B, global -> LDS without a VGPR round trip, for the kernels with a
// COUNTED vmcnt pipeline (gemm8.h, gemm_k2.h, gemm_r3.h).
// WHY NOT THE BUILTIN there (found in round 5 from the disassembly): hipcc's wait-count pass knows that `raw_ptr_buffer_load_lds`
// writes LDS, and in front of every LDS read whose memory operand carries no alias scope it puts a wait for "every LDS-DMA so far" =
// `s_waitcnt vmcnt(0)`.  Plain `ds_read_b128` loads carry a scope (no wait); the transposing `ds_read_b64_tr_b16` builtin does not,
// so every kernel that reads a reduction-strided (COL) operand drained its whole LDS-DMA pipeline once per read group, whatever the
// counted `vmcnt(N)` in the source said (gemm8 TN: 15 such waits in the K loop, NN: 4; gemm_k2 NN: 1).  An LDS-DMA the compiler cannot
// see sets no such score: the counted waits in the source are then the only ones (they were designed to be sufficient: RAW / WAR rules
// in each kernel's header).  hipcc's own vmcnt bookkeeping for the loads and stores it does see stays safe: the counter is in
// order, so unknown extra operations in the queue only make its `vmcnt(N)` wait for more.
// M0 (the LDS destination base) is saved and restored inside the statement (hipcc reserves it); `s_nop 4` covers a descriptor /
// offset SGPR freshly written by a VALU (v_readfirstlane); `s_nop 0` the M0 write -> LDS-DMA hazard.
// Measured (profiles/r05_gemm_r3_ab.txt, r05_dma_asm_k2.txt): the ring kernel's weight gradients 869 -> 1 064 TFLOP/s at 4096^3 with the asm form; gemm_k2's
// NN shapes are 10 % SLOWER with it (its inserted wait only shortens a two-stage ring, and an asm statement is a scheduling barrier
// the builtin is not), so gemm_k2 and gemm8 keep the builtin (MOFO_DMA_ASM_K2 / _G8 = 1 builds them with the asm form).
// LEAN: M0 is not saved / restored and no VALU -> SGPR pad is issued -- for kernels that contain no compiler-visible use of M0 and
// whose descriptor / offsets are SALU results (kernel arguments, blockIdx and readfirstlane'd wave ids from the prologue).
#ifndef MOFO_DMA_ASM
#define MOFO_DMA_ASM 1
#endif
#ifndef MOFO_DMA_ASM_K2
#define MOFO_DMA_ASM_K2 0
#endif
#ifndef MOFO_DMA_ASM_G8
#define MOFO_DMA_ASM_G8 0
#endif
template <bool LEAN = false>
__device__ __forceinline__ void lds_dma16(__amdgpu_buffer_rsrc_t rsrc, unsigned char* lds_dst, int voff, unsigned soff) {
#if MOFO_DMA_ASM
    if constexpr (LEAN) {
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                     :
                     : "s"((unsigned)(unsigned long long)LDS_PTR(lds_dst)), "v"(voff), "s"(rsrc), "s"(soff)
                     : "memory");
    } else {
        unsigned keep;
        asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "s"((unsigned)(unsigned long long)LDS_PTR(lds_dst)), "v"(voff), "s"(rsrc), "s"(soff)
                     : "memory");
    }
#else
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, LDS_PTR(lds_dst), 16, voff, (int)soff, 0, 0);
#endif
}

// AUX = 2: non-temporal load (an operand this launch streams ONCE, e.g. the activation rows of a tall forward GEMM) so that
// it does not displace the weight panel every block of the XCD re-reads from its 4 MiB L2.
template <int LAYOUT, int NI, int AUX = 0, bool ASM = false>
__device__ __forceinline__ void stage_tile_srd(__amdgpu_buffer_rsrc_t rsrc, int v0, int v1, int ld, int d0, int k0,
                                               unsigned char* lds_tile, int wave_u) {
#pragma unroll
    for (int j = 0; j < NI; ++j) {
        const int i = wave_u * NI + j;
        if constexpr (LAYOUT == OPL_ROW) {
            // unsigned: operands reach up to 4 GiB (fill_problem keeps a tile of slack below it, so a row offset past the
            // operand -- ragged last tile -- does not wrap); the SRD range check then reads zeros
            const unsigned soff = ((unsigned)(d0 + 8 * i) * (unsigned)ld + (unsigned)k0) * 2u;
            if constexpr (ASM) lds_dma16<false>(rsrc, lds_tile + i * 1024, v0, soff);
            else __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, LDS_PTR(lds_tile + i * 1024), 16, v0, (int)soff, 0, AUX);
        } else {
            static_assert(LAYOUT == OPL_ROW || NI == 4, "piece parity below assumes 4 pieces per wave");
            const unsigned soff = ((unsigned)(k0 + 4 * i) * (unsigned)ld + (unsigned)d0) * 2u;
            if constexpr (ASM) lds_dma16<false>(rsrc, lds_tile + i * 1024, ((j >> 1) & 1) ? v1 : v0, soff);
            else __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, LDS_PTR(lds_tile + i * 1024), 16, ((j >> 1) & 1) ? v1 : v0, (int)soff, 0, AUX);
        }
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
        // compose as 32-bit words (element-wise 16-bit assembly made hipcc emit v_perm / v_or / v_mov chains)
        const u32x2 l2 = __builtin_bit_cast(u32x2, lo), h2 = __builtin_bit_cast(u32x2, hi);
        const u32x4 r = {l2[0], l2[1], h2[0], h2[1]};
        return __builtin_bit_cast(bf16x8, r);
    }
}

// EPILOGUE of one wave's (16 MI) x 64 accumulator tile, through a wave-private LDS staging area `ep` of PROWS x 64 f32.
// The staging area is wave-private and a wave's DS operations execute in order, so no block barrier is needed: the four
// waves drift apart instead of storing in step.  Per-block phase stamps (tools/gemm_trace.py) showed what the first
// version of this epilogue cost: hipcc put an `s_waitcnt vmcnt(0)` in front of every row group (the per-row bounds checks
// split the loop into basic blocks and the bias loads stayed "possibly pending"), so every store waited for the previous
// store's acknowledge, and the residual / pre-activation loads sat inside the loop, one HBM round trip per row group:
// 5k / 16k / 25k clk for the bf16 / dGELU / residual epilogues beside a 13k-clk K=384 main loop.  Hence: (1) tiles that
// lie wholly inside C take a branch-free, fully unrolled path; (2) residual / pre-activation rows are loaded in chunks of
// up to 32 VGPRs, the first chunk BEFORE the accumulators are staged, the next one before the current one is consumed.
template <int EPI, int MI, int PASSES, bool SCALED = false>
__device__ __forceinline__ void epilogue(const GemmP& p, f32x4 (&acc)[MI][4], float* ep, int mb, int nb, bool full_tile, int lane,
                                         bool stamp = true, float alpha = 1.0f) {   // mb, nb: first row / column of the wave's tile; stamp: trace builds only
    (void)stamp;                                                                     // alpha (SCALED): fp8 de-quantisation factor
    constexpr int WROWS = 16 * MI;              // rows of the wave tile
    constexpr int PROWS = WROWS / PASSES;       // ... staged per pass
    constexpr bool OUT_BF16 = (EPI == MOFO_EPI_BF16 || EPI == MOFO_EPI_BIAS_GELU || EPI == MOFO_EPI_DGELU_BF16 || EPI == MOFO_EPI_RESID_BF16);
    constexpr bool AUX_ROWS = (EPI == MOFO_EPI_DGELU_BF16 || EPI == MOFO_EPI_RESID_BF16);   // a bf16 [M,N] operand read in the epilogue
    // RESID_F32 / RESID_BF16 with rows_in > 0: the RESIDUAL operand's row of output row m is (m / rows_in) * rows_out + row_off +
    // m % rows_in (the output itself is dense): the last decoder block runs on the masked tokens only, its residual input is the
    // whole-sequence stream of the block before.  rows_in >= 256 (every real shape): one division per wave tile, which then crosses at
    // most one group boundary; smaller groups (test geometries) divide per row.
    // Built for wave tiles of <= 64 rows only (the 128 x 128 / 64 x 128 forms and gemm_k2): in the 128-row wave tiles of the MI 8 and
    // 256 x 256 kernels the mapped addressing pushed the residual prefetch chunks to scratch (784 B: ViT-L's proj / fc2 forward on
    // gemm8 fell to 0.40 / 0.61 x) -- mofo_gemm_grouped never routes a problem with a row map there.
    constexpr bool RESID_MAP = (EPI == MOFO_EPI_RESID_F32 || EPI == MOFO_EPI_RESID_BF16) && WROWS <= 64;
    const bool rmap = RESID_MAP && p.rows_in > 0;
    const int rm_seg0 = rmap ? mb / p.rows_in : 0;
    const int rm_rem0 = rmap ? mb - rm_seg0 * p.rows_in : 0;
    auto resid_row = [&](int m) -> size_t {
        if constexpr (!RESID_MAP) return (size_t)m;
        if (!rmap) return (size_t)m;
        if (p.rows_in < 256) {
            const int sg = m / p.rows_in;
            return (size_t)sg * p.rows_out + p.row_off + (m - sg * p.rows_in);
        }
        const int rr = rm_rem0 + (m - mb);
        const int wrap = rr >= p.rows_in ? 1 : 0;
        return (size_t)(rm_seg0 + wrap) * p.rows_out + p.row_off + (rr - wrap * p.rows_in);
    };

    auto stage_acc = [&](int ps) {
#pragma unroll
        for (int ii = 0; ii < MI / PASSES; ++ii) {
            const int i = ps * (MI / PASSES) + ii;
            const int r = 16 * ii + (lane & 15);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int c4 = 4 * j + (lane >> 4);
                if constexpr (SCALED) *(f32x4*)(ep + r * 64 + ((c4 ^ (r & 15)) << 2)) = acc[i][j] * alpha;
                else *(f32x4*)(ep + r * 64 + ((c4 ^ (r & 15)) << 2)) = acc[i][j];
            }
        }
        __builtin_amdgcn_wave_barrier();
    };

    if constexpr (EPI == MOFO_EPI_BF16) {
        // Plain bf16 output: bias is added and the tile rounded IN THE FRAGMENT LAYOUT (a lane holds 4 consecutive n of one m),
        // so the staging area carries bf16 -- twice the rows per pass of the f32 form (the persistent kernel: 2 passes of 32
        // rows instead of 4 of 16; every pass is a write -> read -> store latency chain), half the LDS bytes, and 8-byte
        // writes (6 LDS cycles) instead of 16-byte ones (13).  Image: 128-B rows; 16-B chunk c of row r at c ^ (r & 7), the
        // 8-B half h inside it at h ^ ((r >> 3) & 1): writes (16 rows x 8 B per lane group) and ds_read_b128 row reads are
        // both bank-conflict free; the reader un-swaps the halves by register naming (the bit is the read's index parity).
        constexpr int BROWS = (2 * PROWS < WROWS) ? 2 * PROWS : WROWS;    // rows per pass
        constexpr int BPASS = WROWS / BROWS;
        unsigned char* eb = (unsigned char*)ep;
        const int r16 = lane & 15, c4g = lane >> 4;
        f32x4 bf[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = nb + 16 * j + 4 * c4g;
            bf[j] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (p.bias && n < p.N) bf[j] = *(const f32x4*)(p.bias + n);
        }
        const int rr = lane >> 3, ch = lane & 7;
        const int n = nb + ch * 8;
        const bool ncol = n < p.N;
        auto run = [&](auto full_tag) {
            constexpr bool FULL = decltype(full_tag)::value;
#pragma unroll
        for (int ps = 0; ps < BPASS; ++ps) {
            if (ps) __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int ii = 0; ii < BROWS / 16; ++ii) {
                const int i = ps * (BROWS / 16) + ii;
                const int row = 16 * ii + r16;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    f32x4 v = acc[i][j];
                    if constexpr (SCALED) v = v * alpha;
                    v += bf[j];
                    const int c = 2 * j + (c4g >> 1), h = c4g & 1;
                    *(u32x2*)(eb + row * 128 + ((c ^ (row & 7)) << 4) + ((h ^ ((row >> 3) & 1)) << 3)) =
                        u32x2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
                }
            }
            __builtin_amdgcn_wave_barrier();
            if (ps == 0 && stamp) MOFO_TRACE(3);
            u32x4 t[BROWS / 8];
#pragma unroll
            for (int g = 0; g < BROWS / 8; ++g) {
                const int row = 8 * g + rr;
                t[g] = *(const u32x4*)(eb + row * 128 + ((ch ^ (row & 7)) << 4));
            }
#pragma unroll
            for (int g = 0; g < BROWS / 8; ++g) {
                const u32x4 o = (g & 1) ? u32x4{t[g][2], t[g][3], t[g][0], t[g][1]} : t[g];
                const int m = mb + ps * BROWS + 8 * g + rr;
                if (FULL || (ncol && m < p.M)) *(u32x4*)((bf16_t*)p.C + (size_t)m * p.ldc + n) = o;
            }
        }
        __builtin_amdgcn_wave_barrier();
        };
        if (full_tile) run(std::true_type{});
        else run(std::false_type{});
    } else if constexpr (OUT_BF16) {
        const int cg = lane & 7;
        const int n = nb + cg * 8;
        const bool ncol = n < p.N;
        f32x4 b0 = {0.f, 0.f, 0.f, 0.f}, b1 = {0.f, 0.f, 0.f, 0.f};
        if (p.bias && ncol) {
            b0 = *(const f32x4*)(p.bias + n);
            b1 = *(const f32x4*)(p.bias + n + 4);
        }
        constexpr int NIT = PROWS / 8;          // row groups (8 rows x 128 B per wave-instruction) per pass
        constexpr int NG = WROWS / 8;           // ... per wave tile; the pre-activation rows of all of them are fetched up front
        // e4m3 copy of the activation for an fp8 fc2 (the e4m3 kernels only): quantised from the f32 values with the DELAYED scale
        // q_scale[0] (set from the maximum the previous forward saw); this forward's maximum goes to the stripes of q_amax
        constexpr bool Q8OUT = SCALED && EPI == MOFO_EPI_BIAS_GELU;
        float q8s = 1.f, q8am = 0.f;
        if constexpr (Q8OUT) {
            if (p.C8) q8s = p.q_scale[0];
        }
        auto run = [&](auto full_tag) {
            constexpr bool FULL = decltype(full_tag)::value;
            u32x4 h[NG];
            if constexpr (AUX_ROWS) {
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    const int m = mb + g * 8 + (lane >> 3);
                    h[g] = u32x4{0u, 0u, 0u, 0u};
#ifndef MOFO_GEMM_NT_AUX
#define MOFO_GEMM_NT_AUX 1
#endif
                    if (FULL || (ncol && m < p.M)) {
                        // dGELU: the saved pre-activation is read here for the last time -- non-temporally, so that it does not push
                        // the gradient this kernel writes (and the next GEMM reads) out of the Infinity Cache
                        // ... when it is large (the decoder's 154 MB: 105 vs 112 us); the encoder's 31 MB is read faster with plain loads
                        // (32.1 vs 33.8 us, profiles/r05_epi_ablate.txt): p.aux_nt, set by size in fill_problem
                        if constexpr (EPI == MOFO_EPI_DGELU_BF16 && MOFO_GEMM_NT_AUX) {
                            if (p.aux_nt) h[g] = __builtin_nontemporal_load((const u32x4*)(p.aux + (size_t)m * p.ldaux + n));
                            else h[g] = *(const u32x4*)(p.aux + (size_t)m * p.ldaux + n);
                        } else h[g] = *(const u32x4*)(p.aux + resid_row(m) * p.ldaux + n);
                    }
                }
            }
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                if (g % NIT == 0) {
                    if (g) __builtin_amdgcn_wave_barrier();
                    stage_acc(g / NIT);
                    if (g == 0 && stamp) MOFO_TRACE(3);
                }
                const int r = (g % NIT) * 8 + (lane >> 3);
                const int m = mb + g * 8 + (lane >> 3);
                f32x4 v0 = *(const f32x4*)(ep + r * 64 + (((2 * cg) ^ (r & 15)) << 2));
                f32x4 v1 = *(const f32x4*)(ep + r * 64 + (((2 * cg + 1) ^ (r & 15)) << 2));
                v0 += b0;
                v1 += b1;
                if constexpr (EPI == MOFO_EPI_DGELU_BF16) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const f32x2 d = dgelu_erf2((f32x2){bf16lo_to_f32(h[g][q]), bf16hi_to_f32(h[g][q])});
                        if (q < 2) { v0[2 * q] *= d[0]; v0[2 * q + 1] *= d[1]; }
                        else       { v1[2 * q - 4] *= d[0]; v1[2 * q - 3] *= d[1]; }
                    }
                }
                if constexpr (EPI == MOFO_EPI_RESID_BF16) {   // residual add on the bf16 residual stream
                    v0 += f32x4{bf16lo_to_f32(h[g][0]), bf16hi_to_f32(h[g][0]), bf16lo_to_f32(h[g][1]), bf16hi_to_f32(h[g][1])};
                    v1 += f32x4{bf16lo_to_f32(h[g][2]), bf16hi_to_f32(h[g][2]), bf16lo_to_f32(h[g][3]), bf16hi_to_f32(h[g][3])};
                }
                const u32x4 o = {pack_bf16x2(v0[0], v0[1]), pack_bf16x2(v0[2], v0[3]), pack_bf16x2(v1[0], v1[1]), pack_bf16x2(v1[2], v1[3])};
                const bool ok = FULL || (ncol && m < p.M);
#ifndef MOFO_GEMM_NT_H1
#define MOFO_GEMM_NT_H1 1
#endif
// (tried in round 5 and not kept, profiles/r05_epi_ablate.txt / r05_epi_sc1.txt: the activation stored non-temporally too 1.03 x,
// write-through (sc1) stores 0.90 x at the decoder's shapes)
#ifndef MOFO_ABL_NO_H1      // timing-only ablation builds (tools/gemm_epi_libs.py): drop the pre-activation / the activation store
#define MOFO_ABL_NO_H1 0
#endif
#ifndef MOFO_ABL_NO_G
#define MOFO_ABL_NO_G 0
#endif
                if constexpr (EPI == MOFO_EPI_BIAS_GELU && MOFO_ABL_NO_H1) {
                    asm volatile("" ::"v"(o));
                } else if constexpr (EPI == MOFO_EPI_BIAS_GELU && MOFO_GEMM_NT_H1) {
                    // the pre-activation is not read again before the backward pass: stored non-temporally, it leaves the Infinity
                    // Cache to the activation (C2) that the next GEMM reads (308 MB of outputs per decoder fc1 for 256 MB of cache)
                    if (ok) __builtin_nontemporal_store(o, (u32x4*)((bf16_t*)p.C + (size_t)m * p.ldc + n));
                } else {
                    if (ok) *(u32x4*)((bf16_t*)p.C + (size_t)m * p.ldc + n) = o;
                }
                if constexpr (EPI == MOFO_EPI_BIAS_GELU) {
                    const f32x2 g0 = gelu_erf2((f32x2){v0[0], v0[1]}), g1 = gelu_erf2((f32x2){v0[2], v0[3]});
                    const f32x2 g2 = gelu_erf2((f32x2){v1[0], v1[1]}), g3 = gelu_erf2((f32x2){v1[2], v1[3]});
                    const u32x4 gg = {pack_bf16x2(g0[0], g0[1]), pack_bf16x2(g1[0], g1[1]), pack_bf16x2(g2[0], g2[1]), pack_bf16x2(g3[0], g3[1])};
                    if constexpr (MOFO_ABL_NO_G) asm volatile("" ::"v"(gg));
                    else if (ok) *(u32x4*)((bf16_t*)p.C2 + (size_t)m * p.ldc2 + n) = gg;
                    if constexpr (Q8OUT) {
                        if (p.C8 && ok) {
                            q8am = fmaxf(fmaxf(q8am, fmaxf(fmaxf(fabsf(g0[0]), fabsf(g0[1])), fmaxf(fabsf(g1[0]), fabsf(g1[1])))),
                                         fmaxf(fmaxf(fabsf(g2[0]), fabsf(g2[1])), fmaxf(fabsf(g3[0]), fabsf(g3[1]))));
                            *(u32x2*)(p.C8 + (size_t)m * p.ldc8 + n) = u32x2{pack4_e4m3(g0[0] * q8s, g0[1] * q8s, g1[0] * q8s, g1[1] * q8s),
                                                                              pack4_e4m3(g2[0] * q8s, g2[1] * q8s, g3[0] * q8s, g3[1] * q8s)};
                        }
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
        };
        if (full_tile) run(std::true_type{});
        else run(std::false_type{});
        if constexpr (Q8OUT) {
            if (p.C8) {      // the running maximum is read first: after the first tiles almost no wave issues the atomic (layernorm.hip)
                q8am = wave_max(q8am);
                float* slot = p.q_amax + ((blockIdx.x * 4 + (threadIdx.x >> 6)) & (MOFO_FP8_AMAX_STRIPES - 1));
                if (lane == 0 && q8am > *(volatile const float*)slot) atomicMax((unsigned*)slot, __float_as_uint(q8am));
            }
        }
    } else if (EPI == MOFO_EPI_F32 && p.atomic) {
#pragma unroll
        for (int ps = 0; ps < PASSES; ++ps) {
            const int mp = mb + ps * PROWS;
            stage_acc(ps);
            if (ps == 0 && stamp) MOFO_TRACE(3);
            // one 256-B contiguous row segment per atomic wave-instruction (full chip-wide atomic rate)
            const int n = nb + lane;
            if (n < p.N) {
                float* dst = (float*)p.C + (size_t)mp * p.ldc + n;
                const int rows = min(PROWS, p.M - mp);
                for (int r = 0; r < rows; ++r) {
                    atomicAdd(dst, ep[r * 64 + ((((lane >> 2) ^ (r & 15)) << 2) | (lane & 3))]);
                    dst += p.ldc;
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
    } else {
        const int c4 = lane & 15;
        const int n = nb + c4 * 4;
        const bool ncol = n < p.N;
        f32x4 bv = {0.f, 0.f, 0.f, 0.f};
        if (p.bias && ncol) bv = *(const f32x4*)(p.bias + n);
        constexpr int NIT = PROWS / 4;          // row groups (4 rows x 256 B per wave-instruction) per pass
        constexpr int NG = WROWS / 4;           // ... per wave tile
        constexpr int CH = NG < 8 ? NG : (SCALED ? 4 : 8);   // row groups per residual prefetch chunk (32 VGPRs; 16 beside the e4m3 kernel's 8-register fragments)
        static_assert(NG % CH == 0, "chunking");
        auto run = [&](auto full_tag) {
            constexpr bool FULL = decltype(full_tag)::value;
            f32x4 rr[2][CH];
            auto load_chunk = [&](int c, f32x4* dst) {
                if constexpr (EPI == MOFO_EPI_RESID_F32) {
#pragma unroll
                    for (int k = 0; k < CH; ++k) {
                        const int m = mb + (c * CH + k) * 4 + (lane >> 4);
                        dst[k] = f32x4{0.f, 0.f, 0.f, 0.f};
                        if (FULL || (ncol && m < p.M)) dst[k] = *(const f32x4*)(p.resid + resid_row(m) * p.ldr + n);
                    }
                }
            };
            load_chunk(0, rr[0]);
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                if (g % NIT == 0) {
                    if (g) __builtin_amdgcn_wave_barrier();
                    stage_acc(g / NIT);
                    if (g == 0 && stamp) MOFO_TRACE(3);
                }
                if (g % CH == 0 && g + CH < NG) load_chunk(g / CH + 1, rr[(g / CH + 1) & 1]);
                const int r = (g % NIT) * 4 + (lane >> 4);
                const int m = mb + g * 4 + (lane >> 4);
                const bool ok = FULL || (ncol && m < p.M);
                f32x4 v = *(const f32x4*)(ep + r * 64 + ((c4 ^ (r & 15)) << 2));
                v += bv;
                size_t orow = m;
                if constexpr (EPI == MOFO_EPI_RESID_F32) {
                    v += rr[(g / CH) & 1][g % CH];
                } else if constexpr (EPI == MOFO_EPI_POS_F32 || EPI == MOFO_EPI_POS_BF16) {
                    if (ok) {
                        orow = (size_t)(m / p.rows_in) * p.rows_out + p.row_off + (m % p.rows_in);
                        v += *(const f32x4*)(p.pos + (size_t)p.row_idx[m] * p.ldpos + n);
                    }
                }
                if constexpr (EPI == MOFO_EPI_POS_BF16) {
                    if (ok) *(u32x2*)((bf16_t*)p.C + orow * p.ldc + n) = u32x2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
                } else {
                    if (ok) *(f32x4*)((float*)p.C + orow * p.ldc + n) = v;
                }
            }
            __builtin_amdgcn_wave_barrier();
        };
        if (full_tile) run(std::true_type{});
        else run(std::false_type{});
    }
}

// VAR 0: two LDS stages (64 KiB, 2 blocks/CU).  VAR 1: ONE LDS stage (32 KiB, 3 blocks/CU by VGPRs) with register
// double-buffering: all fragments of tile t are read into VGPRs, barrier, tile t+1's LDS-DMA is issued into the same
// LDS buffer and flies while the 32 MFMAs of tile t run from registers.  More resident blocks per CU let one block's
// (HBM-bound) epilogue overlap another's main loop; the model's short reductions (K = 384..3072) need that.
// MI = 16-row MFMA sub-tiles per wave in M: MI 4 -> 128x128 block tile, MI 2 -> 64x128 (twice the blocks for the
// shapes whose 128x128 tiling gives fewer blocks than the chip has CUs, e.g. M = 5120, N = 768: 240 -> 480).
template <int LA, int LB, int EPI, int VAR, int MI>
__global__ __launch_bounds__(256, (VAR == 1 ? 3 : 2)) void gemm_kernel(GroupP G) {
    static_assert(MI == 4 || (MI == 2 && LA == OPL_ROW), "64-row tiles are built for the ROW A operand (NT / NN)");
    constexpr int BMT = 32 * MI;                       // block tile rows
    constexpr int A_BYTES = BMT * 64 * 2;              // A stage bytes (ROW: [BMT][64] ; COL: [64][128])
    constexpr int STG = A_BYTES + TILE_BYTES;          // one stage = A tile + B tile
    __shared__ __attribute__((aligned(16))) unsigned char smem[(VAR == 0 ? 2 : 1) * STG];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    MOFO_TRACE(0);
    MOFO_TRACE_ID();
    // XCD-aware order (8 XCDs with private L2s; blocks b and b+8 share an XCD): every XCD takes ONE contiguous run of the
    // whole launch's (problem, split, tile) list, and inside a problem the tiles run along the SHORTER side of the tile grid
    // first, so an XCD's run is a few full short-side stripes: it fetches each operand panel once.  (Spreading every problem
    // of a grouped weight-gradient launch over all 8 XCDs made each XCD fetch most panels of every problem: PMC on the step
    // showed 502 MB fetched per wgrad launch for 220 MB of operands.)
    int w = blockIdx.x;
    {
        const int total = G.start[G.count];
        const int q = total >> 3, r = total & 7, xcd = w & 7;
        w = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (w >> 3);
    }
    int gi = 0;
#pragma unroll
    for (int k = 1; k < MAXG; ++k)
        if (k < G.count && w >= G.start[k]) gi = k;
    const GemmP p = G.p[gi];
    const int tiles_n = (p.N + BN - 1) / BN, tiles_m = (p.M + BMT - 1) / BMT;
    const int tiles = tiles_n * tiles_m;
    int wg = w - G.start[gi];
    const int split = wg / tiles;
    wg -= split * tiles;
    const int m0 = (tiles_n <= tiles_m ? wg / tiles_n : wg % tiles_m) * BMT;
    const int n0 = (tiles_n <= tiles_m ? wg % tiles_n : wg / tiles_m) * BN;
    const int kbeg = split * p.k_per_split;
    const int kend = min(p.K, kbeg + p.k_per_split);
    const int nk = (kend - kbeg + BK - 1) / BK;

    f32x4 acc[MI][4];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // bias gradient on the wgrad GEMM: one extra MFMA per m-subtile against an all-ones fragment gives sum_k A[k,m]
    // (replaces a separate column-sum kernel that re-read every dY).  Only the first n-tile's wn==0 waves do it.
    constexpr bool CAN_COLSUM = (LA == OPL_COL && LB == OPL_COL && EPI == MOFO_EPI_F32);
    const bool do_colsum = CAN_COLSUM && p.colsum != nullptr && n0 == 0 && wn == 0;
    f32x4 accb[MI];
    bf16x8 ones;
#pragma unroll
    for (int i = 0; i < MI; ++i) accb[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = (__bf16)1.0f;

    // operand extents for the range check: rows beyond M / N and k-rows beyond this block's k-range read as zeros
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const size_t ext_a = (LA == OPL_ROW ? ((size_t)p.M - 1) * p.lda + kend : ((size_t)kend - 1) * p.lda + p.M) * 2;
    const size_t ext_b = (LB == OPL_ROW ? ((size_t)p.N - 1) * p.ldb + kend : ((size_t)kend - 1) * p.ldb + p.N) * 2;
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, (int)ext_a, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)p.B, 0, (int)ext_b, 0x00020000);
    int va0, va1, vb0, vb1;
    srd_lane_offsets<LA>(p.lda, lane, va0, va1);
    srd_lane_offsets<LB>(p.ldb, lane, vb0, vb1);
    const int krot = p.rotate_tile ? (m0 / BMT + n0 / BN) % (nk > 0 ? nk : 1) : 0;   // rotated reduction order, see gemm_persistent_kernel
    auto stage = [&](int t, int buf) {
        unsigned char* ta = smem + buf * STG;
        int kc = t + krot;
        kc = kc >= nk ? kc - nk : kc;
        stage_tile_srd<LA, (LA == OPL_ROW ? MI : 4)>(ra, va0, va1, p.lda, m0, kbeg + kc * BK, ta, wave_u);
        stage_tile_srd<LB, 4>(rb, vb0, vb1, p.ldb, n0, kbeg + kc * BK, ta + A_BYTES, wave_u);
    };

    if (nk > 0) stage(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    MOFO_TRACE(1);
    if constexpr (VAR == 0) {
        int cur = 0;
        for (int t = 0; t < nk; ++t) {
            if (t + 1 < nk) stage(t + 1, cur ^ 1);
            const unsigned char* ta = smem + cur * STG;
            const unsigned char* tb = ta + A_BYTES;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                bf16x8 af[MI], bfr[4];
#pragma unroll
                for (int i = 0; i < MI; ++i) af[i] = read_frag<LA>(ta, wm * (16 * MI) + 16 * i, ks, lane);
#pragma unroll
                for (int i = 0; i < 4; ++i) bfr[i] = read_frag<LB>(tb, wn * 64 + 16 * i, ks, lane);
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
                if constexpr (CAN_COLSUM) {
                    if (do_colsum) {
#pragma unroll
                        for (int i = 0; i < MI; ++i) accb[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, af[i], accb[i], 0, 0, 0);
                    }
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            cur ^= 1;
        }
    } else {
        const unsigned char* ta = smem;
        const unsigned char* tb = smem + A_BYTES;
        for (int t = 0; t < nk; ++t) {
            bf16x8 af[2][MI], bfr[2][4];
            MOFO_TRACE_IT(t, 0);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
                for (int i = 0; i < MI; ++i) af[ks][i] = read_frag<LA>(ta, wm * (16 * MI) + 16 * i, ks, lane);
#pragma unroll
                for (int i = 0; i < 4; ++i) bfr[ks][i] = read_frag<LB>(tb, wn * 64 + 16 * i, ks, lane);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            MOFO_TRACE_IT(t, 1);
            __builtin_amdgcn_s_barrier();          // every wave holds tile t in registers: the LDS buffer is free
            MOFO_TRACE_IT(t, 2);
            if (t + 1 < nk) stage(t + 1, 0);       // tile t+1 flies while tile t is multiplied from registers
            MOFO_TRACE_IT(t, 3);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[ks][j], af[ks][i], acc[i][j], 0, 0, 0);
                if constexpr (CAN_COLSUM) {
                    if (do_colsum) {
#pragma unroll
                        for (int i = 0; i < MI; ++i) accb[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, af[ks][i], accb[i], 0, 0, 0);
                    }
                }
            }
            MOFO_TRACE_IT(t, 4);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            MOFO_TRACE_IT(t, 5);
            __builtin_amdgcn_s_barrier();          // tile t+1 landed and is visible to every wave
            MOFO_TRACE_IT(t, 6);
        }
    }

    MOFO_TRACE(2);
    if constexpr (CAN_COLSUM) {
        if (do_colsum && lane < 16) {   // D[n][m]: every row n holds the same sum; lanes 0..15 hold m = 16 i + lane in element 0
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const int m = m0 + wm * (16 * MI) + 16 * i + lane;
                if (m < p.M && !(m >= p.colsum_skip_lo && m < p.colsum_skip_hi)) atomicAdd(p.colsum + m, accb[i][0]);
            }
        }
    }
    // ------------------------------------------------------------------ epilogue (through LDS, whole row segments)
    // VAR 0 stages the wave's whole 64x64 f32 tile (16 KiB per wave); VAR 1 has 32 KiB of LDS and stages 32 rows per pass.
    constexpr int PASSES = (VAR == 0) ? 1 : 2;
    static_assert(4 * (16 * MI / PASSES) * 64 * 4 <= (VAR == 0 ? 2 : 1) * STG, "epilogue staging must fit the main-loop LDS");
    epilogue<EPI, MI, PASSES>(p, acc, (float*)smem + wave * ((16 * MI / PASSES) * 64), m0 + wm * (16 * MI), n0 + wn * 64,
                              (m0 + BMT <= p.M) && (n0 + BN <= p.N), lane);
    MOFO_TRACE(4);
#ifdef MOFO_GEMM_TRACE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // stores acknowledged
    MOFO_TRACE(5);
#endif
}

// s_waitcnt with only the vector-memory counter set (gfx9 encoding: vmcnt = imm[15:14]:imm[3:0], expcnt imm[6:4], lgkmcnt imm[11:8])
#define G8_WAIT_VM(n)                                                                      \
    do {                                                                                   \
        __builtin_amdgcn_s_waitcnt((((n) & 15) | (7 << 4) | (15 << 8) | (((n) >> 4) << 14))); \
        asm volatile("" ::: "memory");                                                     \
    } while (0)

// Vector-memory STORES a wave issues in the epilogue of one FULL (16 MI) x 64 wave tile (epilogue<> above, branch-free path): the bf16
// forms write 8 rows x 128 B per instruction (BIAS_GELU: two outputs), the f32 forms 4 rows x 256 B; loads of the epilogue (bias,
// residual, pre-activation rows) are consumed before its last store is issued and do not stay in the queue.  POS_* also gather
// index / table rows per group (not counted: S must not exceed the truth).
template <int EPI, int MI>
constexpr int pers_epi_stores() {
    return EPI == MOFO_EPI_BIAS_GELU ? 4 * MI
         : (EPI == MOFO_EPI_BF16 || EPI == MOFO_EPI_DGELU_BF16 || EPI == MOFO_EPI_RESID_BF16) ? 2 * MI
         : 4 * MI;   // F32, RESID_F32, POS_*: (16 MI / 4) row groups
}

// VAR 2: PERSISTENT form of VAR 1 for the single-problem NT / NN GEMMs (no split-K).  Phase stamps of VAR 1 at the decoder
// shapes (K = 384: six k-iterations per tile) showed 13-18 % of a block's life spent waiting for its FIRST operand tile
// (kernel-argument fetch + LDS-DMA issue + HBM/L2 latency) with nothing else to do.  Here at most 3 x 256 blocks are
// launched and each walks tiles w, w + grid, w + 2 grid, ...: the first k-stage of the NEXT tile is issued from the last
// k-iteration of the current one and lands while the epilogue runs.  That needs an epilogue staging area apart from the
// operand stage: 16 rows x 64 f32 per wave (16 KiB per block, MI passes), 48 KiB of LDS per block -> still 3 blocks / CU.
// MI 8 (256 x 128 tiles, 64 KiB of LDS, 2 blocks / CU): 25 % fewer L1->LDS bytes and LDS fragment reads per MFMA than MI 4.
template <int LA, int LB, int EPI, int MI>
__global__ __launch_bounds__(256, MI == 8 ? 2 : 3) void gemm_persistent_kernel(GemmP p, int total) {
    static_assert(LA == OPL_ROW, "persistent form is built for the ROW A operand (NT / NN)");
    constexpr int BMT = 32 * MI;
    constexpr int A_BYTES = BMT * 64 * 2;
    constexpr int STG = A_BYTES + TILE_BYTES;
    constexpr int EP_BYTES = 4 * 16 * 64 * 4;
    __shared__ __attribute__((aligned(16))) unsigned char smem[STG + EP_BYTES];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    MOFO_TRACE_ID();
    int ti = 0;   // tile counter of this block (trace build: the third tile is stamped)
    (void)ti;
    const int tiles_n = (p.N + BN - 1) / BN;
    const int nk = p.K / BK;
    // XCD-aware order over the virtual grid of `total` one-tile blocks (grid is `total` or a multiple of 8, so a block's
    // tiles all have its own index mod 8 = its XCD): each XCD walks a contiguous run of tiles, n fastest
    auto decode = [&](int w, int& m0, int& n0) {
        const int q = total >> 3, r = total & 7, xcd = w & 7;
        const int wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (w >> 3);
        m0 = (wg / tiles_n) * BMT;
        n0 = (wg % tiles_n) * BN;
    };
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    // extents: A is [M, K] (ROW); B is [N, K] (ROW, NT) or [K, N] (COL, NN)
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, (int)((((size_t)p.M - 1) * p.lda + p.K) * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(
        (void*)p.B, 0, (int)((LB == OPL_ROW ? ((size_t)p.N - 1) * p.ldb + p.K : ((size_t)p.K - 1) * p.ldb + p.N) * 2), 0x00020000);
    int va0, va1, vb0, vb1;
    srd_lane_offsets<LA>(p.lda, lane, va0, va1);
    srd_lane_offsets<LB>(p.ldb, lane, vb0, vb1);
    // k-stage t of tile (m0, n0) is reduction chunk (t + m-tile index + n-tile index) mod nk: the co-resident blocks that share
    // an operand panel (same m-tile: the A rows; same n-tile: the weights) walk the reduction in ROTATED order.  In step they
    // all ask for the same lines at the same moment (PMC on the step: the decoder's fc1 GEMM fetches ~4x its operand bytes
    // either way -- the L2 does not merge the concurrent misses -- but rotated the requests spread over the memory channels).
    // Used where it measured faster (p.rotate, set in fill_problem).
    auto stage = [&](int m0, int n0, int t) {
        int kc = t + m0 / BMT + n0 / BN;
        kc = p.rotate ? kc % nk : t;
        stage_tile_srd<LA, MI>(ra, va0, va1, p.lda, m0, kc * BK, smem, wave_u);
        stage_tile_srd<LB, 4>(rb, vb0, vb1, p.ldb, n0, kc * BK, smem + A_BYTES, wave_u);
    };
    const unsigned char* ta = smem;
    const unsigned char* tb = smem + A_BYTES;
    float* ep = (float*)(smem + STG) + wave * (16 * 64);
    int w = blockIdx.x, m0, n0;
    decode(w, m0, n0);
    stage(m0, n0, 0);
    bool prev_full = false;     // the previous tile took the branch-free epilogue: its store count is known
    for (;;) {
        const int wnext = w + (int)gridDim.x;
        const bool has_next = wnext < total;
        int m1 = 0, n1 = 0;
        if (has_next) decode(wnext, m1, n1);
        f32x4 acc[MI][4];
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (ti == 2) MOFO_TRACE(0);
        // This tile's first k-stage was issued BEFORE the previous tile's epilogue, i.e. it is OLDER than that epilogue's stores in the
        // wave's in-order vmcnt queue: `vmcnt(S)` with S = the stores that epilogue issued waits for the stage alone and leaves the
        // stores one more k-step to be acknowledged (the wait at the end of k-step 0 covers them).  Ablation (profiles/r05_epi_ablate.txt):
        // the decoder's fc1 + GELU takes 98 us with and 78 us without its stores although they need 45 us of HBM time -- each tile paid
        // ~3 us here waiting for acknowledges.  S is exact for full tiles (a LOWER bound is all the wait needs: a smaller count only
        // waits for more); after a ragged tile (stores branched around) and for the first tile the wait is vmcnt(0).
#ifndef MOFO_PERSIST_COUNTED
#define MOFO_PERSIST_COUNTED 1
#endif
        constexpr int SE = pers_epi_stores<EPI, MI>();
        if (MOFO_PERSIST_COUNTED && LB == OPL_ROW && prev_full) G8_WAIT_VM(SE);
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (ti == 2) MOFO_TRACE(1);
        for (int t = 0; t < nk; ++t) {
            bf16x8 af[2][MI], bfr[2][4];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
                for (int i = 0; i < MI; ++i) af[ks][i] = read_frag<LA>(ta, wm * (16 * MI) + 16 * i, ks, lane);
#pragma unroll
                for (int i = 0; i < 4; ++i) bfr[ks][i] = read_frag<LB>(tb, wn * 64 + 16 * i, ks, lane);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();          // every wave holds k-stage t in registers: the LDS stage is free
            if (t + 1 < nk) stage(m0, n0, t + 1);
            else if (has_next) stage(m1, n1, 0);   // the next tile's first k-stage flies under this tile's epilogue
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[ks][j], af[ks][i], acc[i][j], 0, 0, 0);
            if (t + 1 < nk) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();      // k-stage t+1 landed and is visible to every wave
            }
        }
        if (ti == 2) MOFO_TRACE(2);
        prev_full = (m0 + BMT <= p.M) && (n0 + BN <= p.N);
        epilogue<EPI, MI, MI>(p, acc, ep, m0 + wm * (16 * MI), n0 + wn * 64, prev_full, lane, ti == 2);
        if (ti == 2) {
            MOFO_TRACE(4);
            MOFO_TRACE(5);
        }
        ++ti;
        if (!has_next) break;
        w = wnext;
        m0 = m1;
        n0 = n1;
    }
}

// VAR 3: IN-BLOCK SPLIT-K for the grids that cannot fill the chip (the encoder's N = 768 GEMMs at M = 5120: 480 tiles of
// 64 x 128 for 256 CUs -> 1.9 four-wave blocks per CU, every wave waiting on its own load->MFMA chain).  512 threads = two
// groups of four waves; both groups own the SAME 64 x 128 output tile and each walks one half of the reduction with its own
// LDS stage (the VAR 1 loop; a k-stage past the end reads zeros through the SRD), so a CU holds twice the waves for the
// same tiles.  The halves meet in LDS: a group hands the partner the 16-row tile it does not own, adds the one it receives,
// and runs the epilogue of its own 16 rows -- the epilogue is spread over all eight waves too.
template <int LA, int LB, int EPI>
__global__ __launch_bounds__(512, 2) void gemm_ksplit_kernel(GemmP p, int total) {
    static_assert(LA == OPL_ROW, "built for the ROW A operand (NT / NN)");
    constexpr int MI = 2;
    constexpr int BMT = 64;
    constexpr int A_BYTES = BMT * 64 * 2;
    constexpr int STG = A_BYTES + TILE_BYTES;                    // 24 KiB per group
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * STG];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int grp = wave >> 2, w4 = wave & 3;
    const int wm = w4 >> 1, wn = w4 & 1;
    const int tiles_n = (p.N + BN - 1) / BN;
    int m0, n0;
    {
        const int w = blockIdx.x;
        const int q = total >> 3, r = total & 7, xcd = w & 7;
        const int wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (w >> 3);
        m0 = (wg / tiles_n) * BMT;
        n0 = (wg % tiles_n) * BN;
    }
    const int nk = p.K / BK;
    const int nkh = (nk + 1) >> 1;                               // k-stages per group (the second group's last may be empty)
    const int grp_u = __builtin_amdgcn_readfirstlane(grp), w4_u = __builtin_amdgcn_readfirstlane(w4);
    const int kb = grp_u * nkh * BK;
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, (int)((((size_t)p.M - 1) * p.lda + p.K) * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(
        (void*)p.B, 0, (int)((LB == OPL_ROW ? ((size_t)p.N - 1) * p.ldb + p.K : ((size_t)p.K - 1) * p.ldb + p.N) * 2), 0x00020000);
    int va0, va1, vb0, vb1;
    srd_lane_offsets<LA>(p.lda, lane, va0, va1);
    srd_lane_offsets<LB>(p.ldb, lane, vb0, vb1);
    unsigned char* gs = smem + grp_u * STG;
    // NT: both operands are k-contiguous rows -- a k offset past K inside the LAST row would be in range, so an empty
    // trailing k-stage is skipped explicitly (nk odd); NN's B rows past K are out of range by themselves.
    const int krot = p.rotate_tile ? (m0 / BMT + n0 / BN) % nkh : 0;     // rotated reduction order, see gemm_persistent_kernel
    auto kof = [&](int t) {
        int kc = t + krot;
        kc = kc >= nkh ? kc - nkh : kc;
        return kb + kc * BK;
    };
    auto stage = [&](int t) {
        const int k0 = kof(t);
        if (k0 < p.K) {
            stage_tile_srd<LA, MI>(ra, va0, va1, p.lda, m0, k0, gs, w4_u);
            stage_tile_srd<LB, 4>(rb, vb0, vb1, p.ldb, n0, k0, gs + A_BYTES, w4_u);
        }
    };
    f32x4 acc[MI][4];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    stage(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const unsigned char* ta = gs;
    const unsigned char* tb = gs + A_BYTES;
    for (int t = 0; t < nkh; ++t) {
        const bool live = kof(t) < p.K;                          // group-uniform
        bf16x8 af[2][MI], bfr[2][4];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int i = 0; i < MI; ++i) af[ks][i] = read_frag<LA>(ta, wm * (16 * MI) + 16 * i, ks, lane);
#pragma unroll
            for (int i = 0; i < 4; ++i) bfr[ks][i] = read_frag<LB>(tb, wn * 64 + 16 * i, ks, lane);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (t + 1 < nkh) stage(t + 1);
        if (live) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[ks][j], af[ks][i], acc[i][j], 0, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    // exchange: group g keeps row tile i = g of every wave tile and gives away i = 1 - g (4 f32x4 per lane -> 16 KiB per group)
    {
        f32x4* mine = (f32x4*)gs;
        const f32x4* theirs = (const f32x4*)(smem + (1 - grp_u) * STG);
        const int slot = w4 * 64 + lane;
#pragma unroll
        for (int j = 0; j < 4; ++j) mine[j * 256 + slot] = grp ? acc[0][j] : acc[1][j];
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const f32x4 o = theirs[j * 256 + slot];
            if (grp) acc[1][j] += o;
            else acc[0][j] += o;
        }
        __syncthreads();                                         // the exchange area becomes the epilogue staging area
    }
    f32x4 own[1][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) own[0][j] = grp ? acc[1][j] : acc[0][j];
    epilogue<EPI, 1, 1>(p, own, (float*)smem + wave * (16 * 64), m0 + wm * 32 + grp * 16, n0 + wn * 64,
                        (m0 + BMT <= p.M) && (n0 + BN <= p.N), lane, false);
}

// NT on OCP e4m3 operands (BASELINE configs[4]: "fp8 MFMA attention/MLP"; here the four forward Linears of a block).  One byte
// per element, so a 128-row x 128-BYTE operand tile has exactly the bf16 ROW image's geometry (128-B rows, 16-B chunk c of row r
// at c ^ (r & 7)) with twice the reduction depth; the MFMA is the block-scaled v_mfma_scale_f32_16x16x128_f8f6f4 with unit
// (e8m0 = 127) block scales -- 2x the bf16 rate -- and lane l feeds row l & 15, reduction bytes 32 (l >> 4) .. + 31 (two
// ds_read_b128).  LDS-DMA staging, the shared epilogue with the per-tensor de-quantisation factor a_scale_inv * b_scale_inv applied to
// the f32 accumulators.  (Rounds 2-4 ran a two-stage one-tile-per-block kernel here: 3-24 % slower, profiles/r05_gemm_fp8_ab.txt.)
typedef __attribute__((ext_vector_type(8))) int i32x8;
template <int NI>
__device__ __forceinline__ void stage_tile_bytes(__amdgpu_buffer_rsrc_t rsrc, int voff, int ld, int d0, int k0, unsigned char* lds_tile, int wave_u) {
#pragma unroll
    for (int j = 0; j < NI; ++j) {
        const int i = wave_u * NI + j;
        const unsigned soff = (unsigned)(d0 + 8 * i) * (unsigned)ld + (unsigned)k0;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, LDS_PTR(lds_tile + i * 1024), 16, voff, (int)soff, 0, 0);
    }
}
__device__ __forceinline__ i32x8 read_frag_fp8(const unsigned char* lds_tile, int sub0, int lane) {
    const int row = sub0 + (lane & 15);
    const int kc = 2 * (lane >> 4);
    const u32x4 lo = *(const u32x4*)(lds_tile + row * 128 + ((kc ^ (row & 7)) << 4));
    const u32x4 hi = *(const u32x4*)(lds_tile + row * 128 + (((kc + 1) ^ (row & 7)) << 4));
    i32x8 r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3]; r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    return r;
}
// The bf16 persistent kernel's schedule on one-byte operands.  A 128-B LDS row holds 128 reduction
// elements instead of 64 and one v_mfma_scale_f32_16x16x128_f8f6f4 (8 passes) does the work of four bf16 16x16x32 MFMAs (4 passes
// each): a k-stage costs the same LDS bytes, fragment registers (8 VGPRs per 16-row sub-tile) and matrix-pipe cycles as a bf16
// k-stage and covers TWICE the reduction.  One LDS stage + register double buffering (3 blocks / CU at MI <= 4), the next tile's
// first k-stage issued from the last k-iteration, counted wait at the tile start, wave-private epilogue staging: see
// gemm_persistent_kernel.  Every forward Linear of a block runs here under MOFO_FP8=1: qkv (BF16), proj / fc2 (RESID_*), fc1
// (BIAS_GELU, which also writes the e4m3 copy of its activation for fc2).
template <int EPI, int MI>
__global__ __launch_bounds__(256, 3) void gemm_fp8_persistent_kernel(GemmP p, int total) {
    static_assert(MI == 2 || MI == 4, "64- and 128-row tiles (256-row tiles: slower on 7 of 8 ViT-L shapes and 4-22 spilled VGPRs)");
    constexpr int BMT = 32 * MI;
    constexpr int A_BYTES = BMT * 128;
    constexpr int STG = A_BYTES + 128 * 128;
    constexpr int EP_BYTES = 4 * 16 * 64 * 4;
    __shared__ __attribute__((aligned(16))) unsigned char smem[STG + EP_BYTES];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int tiles_n = (p.N + BN - 1) / BN;
    const int nk = p.K / 128;
    auto decode = [&](int w, int& m0, int& n0) {
        const int q = total >> 3, r = total & 7, xcd = w & 7;
        const int wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (w >> 3);
        m0 = (wg / tiles_n) * BMT;
        n0 = (wg % tiles_n) * BN;
    };
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, (int)(((size_t)p.M - 1) * p.lda + p.K), 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)p.B, 0, (int)(((size_t)p.N - 1) * p.ldb + p.K), 0x00020000);
    const int va = (lane >> 3) * p.lda + (((lane & 7) ^ ((lane >> 3) & 7)) << 4);
    const int vb = (lane >> 3) * p.ldb + (((lane & 7) ^ ((lane >> 3) & 7)) << 4);
    auto stage = [&](int m0, int n0, int t) {
        int kc = t + m0 / BMT + n0 / BN;
        kc = p.rotate ? kc % nk : t;
        stage_tile_bytes<MI>(ra, va, p.lda, m0, kc * 128, smem, wave_u);
        stage_tile_bytes<4>(rb, vb, p.ldb, n0, kc * 128, smem + A_BYTES, wave_u);
    };
    const unsigned char* ta = smem;
    const unsigned char* tb = smem + A_BYTES;
    float* ep = (float*)(smem + STG) + wave * (16 * 64);
    const float alpha = p.a_scale_inv[0] * p.b_scale_inv[0];
    int w = blockIdx.x, m0, n0;
    decode(w, m0, n0);
    stage(m0, n0, 0);
    bool prev_full = false;
    for (;;) {
        const int wnext = w + (int)gridDim.x;
        const bool has_next = wnext < total;
        int m1 = 0, n1 = 0;
        if (has_next) decode(wnext, m1, n1);
        f32x4 acc[MI][4];
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        // the stores of the previous (full) tile's epilogue are younger than this tile's first k-stage: a LOWER bound of their count
        // is all the wait needs (the e4m3 copy of the GELU output adds stores and an atomic that are not counted)
        constexpr int SE = pers_epi_stores<EPI, MI>();
        if (prev_full) G8_WAIT_VM(SE);
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        for (int t = 0; t < nk; ++t) {
            i32x8 af[MI], bfr[4];
#pragma unroll
            for (int i = 0; i < MI; ++i) af[i] = read_frag_fp8(ta, wm * (16 * MI) + 16 * i, lane);
#pragma unroll
            for (int i = 0; i < 4; ++i) bfr[i] = read_frag_fp8(tb, wn * 64 + 16 * i, lane);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();          // every wave holds k-stage t in registers: the LDS stage is free
            if (t + 1 < nk) stage(m0, n0, t + 1);
            else if (has_next) stage(m1, n1, 0);   // the next tile's first k-stage flies under this tile's epilogue
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(bfr[j], af[i], acc[i][j], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
            if (t + 1 < nk) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();      // k-stage t+1 landed and is visible to every wave
            }
        }
        prev_full = (m0 + BMT <= p.M) && (n0 + BN <= p.N);
        epilogue<EPI, MI, MI, true>(p, acc, ep, m0 + wm * (16 * MI), n0 + wn * 64, prev_full, lane, false, alpha);
        if (!has_next) break;
        w = wnext;
        m0 = m1;
        n0 = n1;
    }
}

#include "gemm8.h"
#include "gemm_k2.h"
#include "gemm_r3.h"
#include "gemm_r4.h"

// Which main-loop form per (layouts, epilogue, grid), from A/B timing of kernel classes inside the ViT-B B=32 step on MI355X
// (MOFO_GEMM_VARIANT=0|1|2 forces one form; profiles/):
//   wgrad (TN)            VAR 1: 2.30 ms/step vs 2.69 for VAR 0 (3 blocks per CU; fits 168 VGPRs since the SRD staging)
//   NT / NN, all epilogues VAR 2 (persistent): dgrad 2.06 vs 2.20 (VAR 1) / 2.36 (VAR 0); dGELU 0.98 / 1.05 / 1.14;
//                          GELU 0.98 / 1.04 / 1.07; bf16 0.62 / 0.65 / 0.71
//   ... except the f32-residual epilogue on grids that give a persistent block a single tile (encoder proj / fc2:
//       480 64-row tiles): the two-stage VAR 0 is 5-8 % faster there; and the 2-launch pos-embedding epilogue (VAR 0).
static int forced_variant() {
    static int forced = -2;
    if (forced == -2) {
        const char* e = getenv("MOFO_GEMM_VARIANT");
        forced = e ? atoi(e) : -1;
    }
    return forced;
}

// which (layouts, epilogue) pairs the 256 x 256 counted-vmcnt kernel (gemm8.h) is built for
template <int LA, int LB, int EPI>
constexpr bool gemm8_built() {
    if (LA == OPL_ROW && LB == OPL_ROW)
        return EPI == MOFO_EPI_BF16 || EPI == MOFO_EPI_BIAS_GELU || EPI == MOFO_EPI_RESID_F32 || EPI == MOFO_EPI_RESID_BF16 || EPI == MOFO_EPI_F32;
    if (LA == OPL_ROW && LB == OPL_COL) return EPI == MOFO_EPI_BF16 || EPI == MOFO_EPI_DGELU_BF16;
    if (LA == OPL_COL && LB == OPL_COL) return EPI == MOFO_EPI_F32;
    return false;
}
static bool gemm8_has(int op, int epi) {
    if (op == MOFO_GEMM_NT) return epi == MOFO_EPI_BF16 || epi == MOFO_EPI_BIAS_GELU || epi == MOFO_EPI_RESID_F32 || epi == MOFO_EPI_RESID_BF16 || epi == MOFO_EPI_F32;
    if (op == MOFO_GEMM_NN) return epi == MOFO_EPI_BF16 || epi == MOFO_EPI_DGELU_BF16;
    if (op == MOFO_GEMM_TN) return epi == MOFO_EPI_F32;
    return false;
}

// launches per kernel family since the last reset (mofo_gemm_route_counts): tests assert that a model-level parity run really went
// through the shape-routed forms it is meant to cover
enum { ROUTE_TILE = 0, ROUTE_PERSIST = 1, ROUTE_PERSIST8 = 2, ROUTE_KSPLIT = 3, ROUTE_GEMM8 = 4, ROUTE_FP8 = 5, ROUTE_K2 = 6, ROUTE_R3 = 7, ROUTE_R4 = 8, ROUTE_N = 9 };
static long long g_route[ROUTE_N];
#define ROUTE(k) __atomic_fetch_add(&g_route[k], 1LL, __ATOMIC_RELAXED)

template <int LA, int LB, int EPI>
int launch(const GroupP& g, int mi, hipStream_t s) {
    if constexpr (EPI == MOFO_EPI_RESID_F32 || EPI == MOFO_EPI_RESID_BF16) {
        // the residual row map is compiled into the epilogues of <= 64-row wave tiles only (epilogue<>: RESID_MAP); mofo_gemm_grouped
        // never routes a mapped problem to the 128-row wave tiles -- a future routing change or a direct caller gets an error, not
        // residual rows read from the wrong place
        if (mi == 16 || mi == 8)
            for (int i = 0; i < g.count; ++i)
                if (g.p[i].rows_in > 0) MOFO_FAIL(MOFO_EUNSUPPORTED, "mofo_gemm: a residual row map (rows_in > 0) needs the 128 x 128 / 64 x 128 tile forms");
    }
    if (mi == 16) {
        if constexpr (gemm8_built<LA, LB, EPI>()) {
            const int total = g.start[g.count];
            const char* e = getenv("MOFO_GEMM8_GRID");   // persistent blocks (tests force a few so that every block walks several tiles)
            const int cap = e && atoi(e) > 0 ? atoi(e) : 256;
            hipLaunchKernelGGL((gemm8_kernel<LA, LB, EPI>), dim3(total < cap ? total : cap), dim3(512), 0, s, g, total);
            ROUTE(ROUTE_GEMM8);
            MOFO_CHECK_LAUNCH("mofo_gemm(gemm8)");
            return MOFO_OK;
        } else {
            MOFO_FAIL(MOFO_EUNSUPPORTED, "mofo_gemm: the 256-tile kernel is not built for this (op, epilogue)");
        }
    }
    if (mi == 32) {   // one 128 x 128 tile per CU, two K-halves with two-stage rings (gemm_k2.h)
        if constexpr (LA == OPL_ROW) {
            const int total = g.start[1];
            const char* e = getenv("MOFO_GEMM_K2_STAG");         // read per call (A/B in one process)
            const GemmP& pk = g.p[0];
            if (!e || atoi(e) != 0) hipLaunchKernelGGL((gemm_k2_kernel<LA, LB, EPI, true>), dim3(total), dim3(512), 0, s, pk, total);
            else hipLaunchKernelGGL((gemm_k2_kernel<LA, LB, EPI, false>), dim3(total), dim3(512), 0, s, pk, total);
            ROUTE(ROUTE_K2);
            MOFO_CHECK_LAUNCH("mofo_gemm(k2)");
            return MOFO_OK;
        } else {
            MOFO_FAIL(MOFO_EUNSUPPORTED, "mofo_gemm: the one-tile-per-CU split-K kernel is built for NT / NN");
        }
    }
    const int forced = forced_variant();
    const dim3 grid(g.start[g.count]), block(256);
    if constexpr (LA == OPL_ROW) {
        const GemmP& p = g.p[0];
        const int total = g.start[1];
        // persistent form: one problem, no split-K / accumulate
        const bool can_persist = g.count == 1 && !p.atomic && p.k_per_split >= p.K && EPI != MOFO_EPI_POS_F32 && EPI != MOFO_EPI_POS_BF16;
        int var = forced >= 0 ? forced : ((EPI == MOFO_EPI_RESID_F32 && total <= 768) || !can_persist ? 0 : 2);
        if (var == 2 && !can_persist) var = 1;
        // grids of at most two 64-row tiles per CU with a reduction worth splitting: in-block split-K (VAR 3)
        if (mi == 2 && can_persist && total <= 512 && p.K >= 1536 && (forced < 0 || forced == 3)) {
            hipLaunchKernelGGL((gemm_ksplit_kernel<LA, LB, EPI>), dim3(total), dim3(512), 0, s, p, total);
            ROUTE(ROUTE_KSPLIT);
        } else if (mi == 8) {
            const dim3 pgrid(total < 512 ? total : 512);
            hipLaunchKernelGGL((gemm_persistent_kernel<LA, LB, EPI, 8>), pgrid, block, 0, s, p, total);
            ROUTE(ROUTE_PERSIST8);
        } else if (var == 2) {
            ROUTE(ROUTE_PERSIST);
            const dim3 pgrid(total < 768 ? total : 768);
            if (mi == 2) hipLaunchKernelGGL((gemm_persistent_kernel<LA, LB, EPI, 2>), pgrid, block, 0, s, p, total);
            else hipLaunchKernelGGL((gemm_persistent_kernel<LA, LB, EPI, 4>), pgrid, block, 0, s, p, total);
        } else if (mi == 2) {
            ROUTE(ROUTE_TILE);
            if (var == 3) var = 1;
            if (var == 0) hipLaunchKernelGGL((gemm_kernel<LA, LB, EPI, 0, 2>), grid, block, 0, s, g);
            else hipLaunchKernelGGL((gemm_kernel<LA, LB, EPI, 1, 2>), grid, block, 0, s, g);
        } else {
            ROUTE(ROUTE_TILE);
            if (var == 3) var = 1;
            if (var == 0) hipLaunchKernelGGL((gemm_kernel<LA, LB, EPI, 0, 4>), grid, block, 0, s, g);
            else hipLaunchKernelGGL((gemm_kernel<LA, LB, EPI, 1, 4>), grid, block, 0, s, g);
        }
    } else {
        ROUTE(ROUTE_TILE);
        const int var = forced >= 0 ? (forced != 0) : 1;
        if (var == 0) hipLaunchKernelGGL((gemm_kernel<LA, LB, EPI, 0, 4>), grid, block, 0, s, g);
        else hipLaunchKernelGGL((gemm_kernel<LA, LB, EPI, 1, 4>), grid, block, 0, s, g);
    }
    MOFO_CHECK_LAUNCH("mofo_gemm");
    return MOFO_OK;
}

}  // namespace

static int fill_problem(const mofo_gemm_args* a, GemmP& p, int bm, int bn, int& blocks) {
    if (!a->A || !a->B || !a->C) MOFO_FAIL(MOFO_EINVAL, "mofo_gemm: null operand");
    if (a->M <= 0 || a->N <= 0 || a->K <= 0) MOFO_FAIL(MOFO_EINVAL, "mofo_gemm: non-positive dims %d %d %d", a->M, a->N, a->K);
    const int op = a->op, epi = a->epilogue;
    // alignment / divisibility the kernels are built for
    if (a->N % 8 || a->lda % 8 || a->ldb % 8) MOFO_FAIL(MOFO_EUNSUPPORTED, "mofo_gemm: N, lda, ldb must be multiples of 8");
    if (op == MOFO_GEMM_NT && a->K % 64) MOFO_FAIL(MOFO_EUNSUPPORTED, "mofo_gemm NT: K=%d must be a multiple of 64", a->K);
    if (op == MOFO_GEMM_NT_FP8) {
        if (a->K % 128 || a->lda % 16 || a->ldb % 16) MOFO_FAIL(MOFO_EUNSUPPORTED, "mofo_gemm NT_FP8: K must be a multiple of 128, lda / ldb of 16");
        if (!a->a_scale_inv || !a->b_scale_inv) MOFO_FAIL(MOFO_EINVAL, "mofo_gemm NT_FP8: needs the device scalars a_scale_inv, b_scale_inv");
        if (epi != MOFO_EPI_BF16 && epi != MOFO_EPI_BIAS_GELU && epi != MOFO_EPI_RESID_F32 && epi != MOFO_EPI_RESID_BF16)
            MOFO_FAIL(MOFO_EUNSUPPORTED, "mofo_gemm NT_FP8: epilogues BF16, BIAS_GELU, RESID_F32, RESID_BF16 only");
        if (a->C8 && (epi != MOFO_EPI_BIAS_GELU || !a->q_scale || !a->q_amax || a->ldc8 % 8))
            MOFO_FAIL(MOFO_EINVAL, "mofo_gemm NT_FP8: the e4m3 activation copy (C8) rides on BIAS_GELU and needs q_scale, q_amax, ldc8 a multiple of 8");
        if ((a->splits > 1) || a->accumulate) MOFO_FAIL(MOFO_EUNSUPPORTED, "mofo_gemm NT_FP8: no split-K / accumulate");
    }
    if (a->C8 && op != MOFO_GEMM_NT_FP8) MOFO_FAIL(MOFO_EINVAL, "mofo_gemm: the e4m3 activation copy (C8) exists for op NT_FP8 + BIAS_GELU only");
    if (op == MOFO_GEMM_NN && a->K % 64) MOFO_FAIL(MOFO_EUNSUPPORTED, "mofo_gemm NN: K=%d must be a multiple of 64", a->K);
    if (op == MOFO_GEMM_TN && a->M % 8) MOFO_FAIL(MOFO_EUNSUPPORTED, "mofo_gemm TN: M=%d must be a multiple of 8", a->M);
    const bool out_bf16 = (epi == MOFO_EPI_BF16 || epi == MOFO_EPI_BIAS_GELU || epi == MOFO_EPI_DGELU_BF16 || epi == MOFO_EPI_RESID_BF16 ||
                           epi == MOFO_EPI_POS_BF16);
    if (a->ldc % (out_bf16 ? 8 : 4)) MOFO_FAIL(MOFO_EUNSUPPORTED, "mofo_gemm: ldc must be a multiple of %d", out_bf16 ? 8 : 4);
    int splits = a->splits < 1 ? 1 : a->splits;
    if (splits > 1 && epi != MOFO_EPI_F32) MOFO_FAIL(MOFO_EUNSUPPORTED, "mofo_gemm: split-K only with the f32 accumulate epilogue");
    if (epi == MOFO_EPI_BIAS_GELU && (!a->C2 || a->ldc2 % 8)) MOFO_FAIL(MOFO_EINVAL, "mofo_gemm: BIAS_GELU needs C2 (ldc2 multiple of 8)");
    if (epi == MOFO_EPI_RESID_F32 && (!a->resid || a->ldr % 4)) MOFO_FAIL(MOFO_EINVAL, "mofo_gemm: RESID_F32 needs resid (ldr multiple of 4)");
    if (epi == MOFO_EPI_DGELU_BF16 && (!a->aux || a->ldaux % 8)) MOFO_FAIL(MOFO_EINVAL, "mofo_gemm: DGELU needs aux (ldaux multiple of 8)");
    if (epi == MOFO_EPI_RESID_BF16 && (!a->aux || a->ldaux % 8)) MOFO_FAIL(MOFO_EINVAL, "mofo_gemm: RESID_BF16 needs the bf16 residual in aux (ldaux multiple of 8)");
    if ((epi == MOFO_EPI_POS_F32 || epi == MOFO_EPI_POS_BF16) && (!a->pos || !a->row_idx || a->rows_in <= 0 || a->ldpos % 4)) MOFO_FAIL(MOFO_EINVAL, "mofo_gemm: POS_F32 needs pos,row_idx,rows_in");
    if ((epi == MOFO_EPI_RESID_F32 || epi == MOFO_EPI_RESID_BF16) && a->rows_in != 0 &&
        (a->rows_in < 0 || a->rows_out < a->rows_in || a->row_off < 0 || a->row_off + a->rows_in > a->rows_out || a->M % a->rows_in))
        MOFO_FAIL(MOFO_EINVAL, "mofo_gemm: residual row map needs 0 < rows_in <= rows_out - row_off and M a multiple of rows_in");
    {
        // the operands are addressed through UNSIGNED 32-bit buffer offsets (SRD extent, per-lane and per-tile byte offsets; every
        // offset is formed in unsigned arithmetic and the descriptor's num_records is unsigned): an operand image of 4 GiB or more is
        // refused instead of wrapping (split the rows / the reduction on the caller's side).  ViT-L, 32 frames, 256 clips per GPU: the
        // decoder's fc1 output is 3.3 GB, its qkv 2.5 GB.
        const long long ra = op == MOFO_GEMM_TN ? a->K : a->M, ca = op == MOFO_GEMM_TN ? a->M : a->K;
        const bool ntlike = op == MOFO_GEMM_NT || op == MOFO_GEMM_NT_FP8;
        const long long rb = ntlike ? a->N : a->K, cb = ntlike ? a->K : a->N;
        const long long esz = op == MOFO_GEMM_NT_FP8 ? 1 : 2;
        const long long ext_a = ((ra - 1) * a->lda + ca) * esz, ext_b = ((rb - 1) * a->ldb + cb) * esz;
        // + slack: offsets of rows past the end (a ragged last tile; the k-stages the ring kernels issue past the end of the reduction)
        // are formed before the range check drops them, and must not wrap into the operand
        const long long slack = 1024LL * (a->lda > a->ldb ? a->lda : a->ldb) * 2;
        if (ext_a + slack >= (1LL << 32) || ext_b + slack >= (1LL << 32))
            MOFO_FAIL(MOFO_EUNSUPPORTED, "mofo_gemm: operand extent %lld / %lld bytes is beyond the 4 GiB the 32-bit buffer offsets address",
                      ext_a, ext_b);
        if (a->lda < ca || a->ldb < cb) MOFO_FAIL(MOFO_EINVAL, "mofo_gemm: leading dimension smaller than the row length");
    }
    p.A = (const bf16_t*)a->A; p.B = (const bf16_t*)a->B; p.C = a->C; p.C2 = a->C2;
    p.bias = a->bias; p.resid = a->resid; p.aux = (const bf16_t*)a->aux; p.pos = a->pos; p.row_idx = a->row_idx;
    p.M = a->M; p.N = a->N; p.K = a->K;
    p.lda = a->lda; p.ldb = a->ldb; p.ldc = a->ldc; p.ldc2 = a->ldc2; p.ldr = a->ldr; p.ldaux = a->ldaux; p.ldpos = a->ldpos;
    p.rows_in = a->rows_in; p.rows_out = a->rows_out; p.row_off = a->row_off;
    const int kps = ceil_div(ceil_div(a->K, splits), BK) * BK;
    splits = ceil_div(a->K, kps);
    p.k_per_split = kps;
    p.atomic = (splits > 1 || a->accumulate) ? 1 : 0;
    p.a_scale_inv = a->a_scale_inv;
    p.b_scale_inv = a->b_scale_inv;
    p.C8 = op == MOFO_GEMM_NT_FP8 ? (unsigned char*)a->C8 : nullptr;
    p.ldc8 = a->ldc8;
    p.q_scale = a->q_scale;
    p.q_amax = a->q_amax;
    p.colsum = a->colsum;
    p.colsum_skip_lo = a->colsum_skip_lo;
    p.colsum_skip_hi = a->colsum_skip_hi;
    {
        // rotated reduction order (gemm_persistent_kernel): measured per class on one box -- fc1 + GELU forward (12 / 24 n-tiles
        // share an A panel, long epilogue) 126 -> 106 us (decoder), 38.0 -> 37.2 (encoder); neutral for the other wide-N
        // GEMMs; 5-10 % SLOWER for the N = 384 ones (3 n-tiles: little to de-duplicate, and the block's own row panel is no
        // longer streamed in order).  (The MOFO_GEMM_ROTATE / _ROTATE_TILE overrides were retired in round 5.)
        p.rotate = ((op == MOFO_GEMM_NT || op == MOFO_GEMM_NT_FP8) && epi == MOFO_EPI_BIAS_GELU);
        p.rotate_tile = 1;
        p.aux_nt = (long long)a->M * a->N * 2 >= (96LL << 20);
    }
    if (a->colsum && !(op == MOFO_GEMM_TN && epi == MOFO_EPI_F32)) MOFO_FAIL(MOFO_EUNSUPPORTED, "mofo_gemm: colsum rides on TN + F32 (wgrad) only");
    blocks = ceil_div(a->M, bm) * ceil_div(a->N, bn) * splits;
    return MOFO_OK;
}

static int dispatch(int op, int epi, const GroupP& g, int mi, hipStream_t s) {
#define GO(LA, LB, E) return launch<LA, LB, E>(g, mi, s)
    if (op == MOFO_GEMM_NT) {
        switch (epi) {
            case MOFO_EPI_BF16: GO(OPL_ROW, OPL_ROW, MOFO_EPI_BF16);
            case MOFO_EPI_BIAS_GELU: GO(OPL_ROW, OPL_ROW, MOFO_EPI_BIAS_GELU);
            case MOFO_EPI_RESID_F32: GO(OPL_ROW, OPL_ROW, MOFO_EPI_RESID_F32);
            case MOFO_EPI_POS_F32: GO(OPL_ROW, OPL_ROW, MOFO_EPI_POS_F32);
            case MOFO_EPI_POS_BF16: GO(OPL_ROW, OPL_ROW, MOFO_EPI_POS_BF16);
            case MOFO_EPI_RESID_BF16: GO(OPL_ROW, OPL_ROW, MOFO_EPI_RESID_BF16);
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
    } else if (op == MOFO_GEMM_NT_FP8 && g.count == 1) {
        const GemmP& p = g.p[0];
        const int total = g.start[1];
        const dim3 pgrid(total < 768 ? total : 768), block(256);
#define GO8(E)                                                                                                           \
    do {                                                                                                                 \
        if (mi == 2) hipLaunchKernelGGL((gemm_fp8_persistent_kernel<E, 2>), pgrid, block, 0, s, p, total);               \
        else hipLaunchKernelGGL((gemm_fp8_persistent_kernel<E, 4>), pgrid, block, 0, s, p, total);                       \
    } while (0)
        if (epi == MOFO_EPI_BF16) GO8(MOFO_EPI_BF16);
        else if (epi == MOFO_EPI_BIAS_GELU) GO8(MOFO_EPI_BIAS_GELU);
        else if (epi == MOFO_EPI_RESID_F32) GO8(MOFO_EPI_RESID_F32);
        else if (epi == MOFO_EPI_RESID_BF16) GO8(MOFO_EPI_RESID_BF16);
        else MOFO_FAIL(MOFO_EUNSUPPORTED, "mofo_gemm NT_FP8: epilogue %d is not built", epi);
#undef GO8
        ROUTE(ROUTE_FP8);
        MOFO_CHECK_LAUNCH("mofo_gemm(fp8)");
        return MOFO_OK;
    }
#undef GO
    MOFO_FAIL(MOFO_EUNSUPPORTED, "mofo_gemm: op %d with epilogue %d is not built", op, epi);
}

// Shapes routed to the 256 x 256 kernel by default, from same-process, STEADY-STATE A/B timings on MI355X (tools/gemm8_ab.py,
// profiles/r03_gemm8_ab.txt; 20 ms of the same variant before every timed block -- in 2-ms bursts the 256-tile kernel reads 20-35 % low):
//   * any NT / NN problem that gives a block several tiles of a deep reduction: 8192^2 x 4096 1.45 x, 16384 x 4096 x 4096 1.44 x,
//     8192^3 1.61 x (NN 1.60 x), 4096^2 x 16384 1.12 x; 4096^3 0.98 x
//   * the forward (NT) GEMMs of the ViT-L / 32-frame step (BASELINE configs[4]: 10 240 encoder and 100 352 decoder rows): enc qkv
//     1.09 x, proj 1.05 x, fc1 + GELU 1.05 x, fc2 1.13 x, dec qkv 1.23 x, fc2 1.11 x, fc1 + GELU 1.02 x
//   * not the ViT-B step (5 120 / 50 176 rows, K = 384 / 768, N = 384 / 768 / 1152): 0.47-0.99 x -- too few, too short tiles for
//     one block per CU, nor the dgrad (NN) GEMMs of ViT-L (0.88-1.03 x), nor weight gradients (tile counts).
static bool gemm8_wanted(const mofo_gemm_args* a, int count) {
    if (count != 1 || a[0].op == MOFO_GEMM_TN || a[0].N % 256) return false;
    const long long t256 = (long long)ceil_div(a[0].M, 256) * ceil_div(a[0].N, 256);
    const int K = a[0].K;
    if ((t256 >= 1024 && K >= 4096) || (t256 >= 256 && K >= 16384)) return true;
    return a[0].op == MOFO_GEMM_NT && t256 >= 160 && (K >= 1024 || (K >= 512 && t256 >= 2000));
}


// ---- the 256 x 128 ring kernel (gemm_r3.h): host side.  Built for the weight gradients (TN, f32 out; split-K / accumulate with f32
// atomics; fused bias-gradient column sums).  MOFO_GEMM_R3 = 0: never, 1: wherever legal, unset: by shape (r3_wanted).
static bool r3_legal(const mofo_gemm_args* a, int count) {
    if (count < 1 || count > MAXR) return false;
    for (int i = 0; i < count; ++i)
        if (a[i].op != MOFO_GEMM_TN || a[i].epilogue != MOFO_EPI_F32 || a[i].bias) return false;
    return true;
}
// Which weight-gradient groups go to the ring kernel by default, from the same-process A/B at the step's shapes (tools/gemm_r3_ab.py,
// profiles/r05_gemm_r3_ab.txt).  While all 256 CUs hold a unit the ring kernel and the 128 x 128 kernel (three blocks per CU) run at
// the same rate on the encoder's shapes (1 157 vs 1 135 TFLOP/s: both sit on the CU's fill path), so the
// choice is a question of ROUNDS: 216 units per encoder block on 256 slots against 432 tiles on 768.
//   1 block 0.84 rounds: 0.94-0.98 x | 2 blocks 1.69: 1.12-1.22 x | 3 blocks 2.53: 0.94-0.97 x | 6 blocks 5.06 with the tail dealt in
//   chunks: 1.02 x | 7 blocks 5.91: 1.04 x (the 128 x 128 kernel takes at most three blocks per launch)
// Whole tiles only: the decoder's 384-wide outputs pad 256-row tiles by 14 % (0.65-0.92 x, not routed).
static bool r3_rounds(const mofo_gemm_args* a, int count, double* rounds, int tm = 256) {
    long long units = 0;
    for (int i = 0; i < count; ++i) {
        if (a[i].M % tm || a[i].N % 128 || a[i].K < 2048) return false;
        units += (long long)(a[i].M / tm) * (a[i].N / 128) * (a[i].splits < 1 ? 1 : a[i].splits);
    }
    *rounds = (double)units / 256.0;
    return true;
}
static bool r3_wanted(const mofo_gemm_args* a, int count, int tm = 256) {
    double r;
    if (!r3_rounds(a, count, &r, tm)) return false;
    const int fl = (int)r;
    const double frac = r - fl;
    if (fl >= 4) return true;                                 // many rounds: at most one in five is partly filled, or its units are dealt in chunks
    return fl >= 1 && (frac == 0.0 || frac >= 0.65);
}
// the last, partial round of a run is dealt in chunks (f32 atomics onto zeroed destinations) only when it is a small share of the
// launch: every unit of it costs its sharers an atomic pass over its 128 KiB (1.3 TB/s chip-wide; 1 block: 0.52 x, 6 blocks: 1.02 x)
static int r3_tail_auto(const mofo_gemm_args* a, int count, int tm = 256) {
    double r;
    if (!r3_rounds(a, count, &r, tm)) return 0;
    const int fl = (int)r;
    return fl >= 4 && r - fl > 0.0 && r - fl <= 0.2;
}
static int r3_mode() {
    const char* e = getenv("MOFO_GEMM_R3");      // read per call: A/B switches inside one process
    return e ? atoi(e) : -1;
}
// tile rows of the ring kernel a weight-gradient group goes to: 384 (gemm_r4.h) when every output is whole 384 x 128 tiles (ViT-B:
// every width is a multiple of 384; same-process A/B, tools/wgrad_dec_ab.py, profiles/r06_wgrad_dec_ab.txt: encoder groups of 3 / 5 / 7
// blocks 1.10 / 1.14 / 1.10 x gemm_r3), else 256 (gemm_r3.h: ViT-L's 1 024).  MOFO_GEMM_R4=0 keeps gemm_r3 everywhere.
static int ring_tm(const mofo_gemm_args* a, int count) {
    const char* e = getenv("MOFO_GEMM_R4");      // read per call: A/B switches inside one process.  0: never gemm_r4; unset / 1: by shape
    if (e && atoi(e) == 0) return R3_TM;
    for (int i = 0; i < count; ++i)
        if (a[i].M % R4_TM || a[i].N % R3_TN) return R3_TM;
    return R4_TM;
}
static int r3_fill(const mofo_gemm_args* a, int count, R3Group& g, int tm = R3_TM) {
    g.count = count;
    g.start[0] = 0;
    for (int i = 0; i < count; ++i) {
        GemmP full;
        int blocks = 0;
        const int rc = fill_problem(&a[i], full, tm, R3_TN, blocks);
        if (rc) return rc;
        R3Prob& p = g.p[i];
        p.A = full.A; p.B = full.B; p.C = full.C; p.colsum = full.colsum;
        p.M = full.M; p.N = full.N; p.K = full.K; p.lda = full.lda; p.ldb = full.ldb; p.ldc = full.ldc;
        p.k_per_split = full.k_per_split; p.atomic = full.atomic; p.skip_lo = full.colsum_skip_lo; p.skip_hi = full.colsum_skip_hi;
        g.start[i + 1] = g.start[i] + blocks;
    }
    for (int i = count; i < MAXR; ++i) {
        g.p[i] = g.p[0];
        g.start[i + 1] = g.start[count];
    }
    const char* e = getenv("MOFO_GEMM_R3_TAIL");     // 0 / 1 forces plain rounds / tail chunks (tests, A/B); unset: by shape
    g.tail = e ? atoi(e) : r3_tail_auto(a, count, tm);
    g.slices = 0;
    g.slab_stride = 0;
    return MOFO_OK;
}
static int r3_grid() {
    const char* e = getenv("MOFO_GEMM_R3_GRID");     // blocks (a multiple of 8; tests force a few so that every block walks several units)
    int nb = e && atoi(e) > 0 ? atoi(e) : 256;
    nb = (nb + 7) / 8 * 8;
    return nb;
}
// which problems of the launch receive f32 atomics from SEVERAL blocks (units of a run's tail, or split / accumulating problems):
// the caller zeroes those destinations (or accumulates on purpose).  Conservative: every problem that owns a tail unit is flagged.
static void r3_plan(const R3Group& g, int nb, int* shared) {
    const int total = g.start[g.count], nbx = nb >> 3;
    for (int i = 0; i < g.count; ++i) shared[i] = g.p[i].atomic ? 1 : 0;
    if (!g.tail) return;
    const int q = total >> 3, r = total & 7;
    for (int xcd = 0; xcd < 8; ++xcd) {
        const int xbeg = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
        const int xlen = q + (xcd < r ? 1 : 0);
        for (int u = (xlen / nbx) * nbx; u < xlen; ++u) {
            int gi = 0;
            for (int k = 1; k < g.count; ++k)
                if (xbeg + u >= g.start[k]) gi = k;
            shared[gi] = 1;
        }
    }
}
static int r3_launch(const mofo_gemm_args* a, int count, hipStream_t s) {
    R3Group g;
    const int rc = r3_fill(a, count, g);
    if (rc) return rc;
    const int total = g.start[count];
    hipLaunchKernelGGL((gemm_r3_kernel<OPL_COL, OPL_COL, MOFO_EPI_F32>), dim3(r3_grid()), dim3(512), 0, s, g, total);
    ROUTE(ROUTE_R3);
    MOFO_CHECK_LAUNCH("mofo_gemm(r3)");
    return MOFO_OK;
}

// ---- the 384 x 128 ring kernel (gemm_r4.h) and the SLICED weight-gradient launch.
static bool r4_tiles(const mofo_gemm_args* a, int count) {
    for (int i = 0; i < count; ++i)
        if (a[i].M % R4_TM || a[i].N % R3_TN) return false;
    return true;
}
static int r4_launch(const mofo_gemm_args* a, int count, hipStream_t s) {
    R3Group g;
    const int rc = r3_fill(a, count, g, R4_TM);
    if (rc) return rc;
    hipLaunchKernelGGL((gemm_r4_kernel<MOFO_EPI_F32>), dim3(r3_grid()), dim3(512), 0, s, g, g.start[count]);
    ROUTE(ROUTE_R4);
    MOFO_CHECK_LAUNCH("mofo_gemm(r4)");
    return MOFO_OK;
}

// SLICED weight gradients (include/mofo_hip.h: mofo_gemm_wgrad_sliced).  tile rows: 384 when every output is whole 384 x 128 tiles
// (MOFO_WGRAD_TILE=256 forces gemm_r3's 256-row tiles: A/B), else 256 (ragged tiles allowed).
static int sliced_tile(const mofo_gemm_args* a, int count) {
    const char* e = getenv("MOFO_WGRAD_TILE");
    if (e && atoi(e) == 256) return R3_TM;
    return r4_tiles(a, count) ? R4_TM : R3_TM;
}
static int sliced_check(const mofo_gemm_args* a, int count, int slices, long long* stride) {
    if (!a || count < 1 || count > MAXR) MOFO_FAIL(MOFO_EINVAL, "mofo_gemm_wgrad_sliced: count must be 1..%d", MAXR);
    if (slices < 1 || slices > 8) MOFO_FAIL(MOFO_EINVAL, "mofo_gemm_wgrad_sliced: slices must be 1..8");
    long long tot = 0;
    for (int i = 0; i < count; ++i) {
        if (a[i].op != MOFO_GEMM_TN || a[i].epilogue != MOFO_EPI_F32 || a[i].bias || a[i].splits > 1)
            MOFO_FAIL(MOFO_EUNSUPPORTED, "mofo_gemm_wgrad_sliced: every problem must be TN + F32 without bias / splits");
        if (a[i].M <= 0 || a[i].N <= 0 || a[i].K <= 0 || a[i].N % 8) MOFO_FAIL(MOFO_EINVAL, "mofo_gemm_wgrad_sliced: bad dims");
        tot += (long long)a[i].M * a[i].N;
    }
    *stride = tot;
    return MOFO_OK;
}
extern "C" long long mofo_gemm_wgrad_sliced_ws(const mofo_gemm_args* a, int count, int slices) {
    long long stride = 0;
    const int rc = sliced_check(a, count, slices, &stride);
    if (rc) return rc;
    return stride * slices;
}
extern "C" int mofo_gemm_wgrad_sliced(const mofo_gemm_args* a, int count, int slices, float* ws, long long ws_floats, void* stream) {
    long long stride = 0;
    int rc = sliced_check(a, count, slices, &stride);
    if (rc) return rc;
    if (!ws || ws_floats < stride * slices || ((uintptr_t)ws & 15))
        MOFO_FAIL(MOFO_EINVAL, "mofo_gemm_wgrad_sliced: workspace of %lld floats (16-byte aligned) needed, got %lld", stride * slices, ws_floats);
    hipStream_t s = (hipStream_t)stream;
    const int tm = sliced_tile(a, count);
    R3Group g;
    SlabReduceP rp;
    g.count = count;
    g.start[0] = 0;
    rp.start[0] = 0;
    long long off = 0;
    int accumulate = -1;
    for (int i = 0; i < count; ++i) {
        GemmP full;
        int blocks = 0;
        mofo_gemm_args one = a[i];
        one.splits = 1;
        one.accumulate = 0;
        rc = fill_problem(&one, full, tm, R3_TN, blocks);
        if (rc) return rc;
        if (a[i].ldc % 4 || ((uintptr_t)a[i].C & 15)) MOFO_FAIL(MOFO_EUNSUPPORTED, "mofo_gemm_wgrad_sliced: C must be 16-byte aligned with ldc a multiple of 4");
        if (accumulate >= 0 && accumulate != (a[i].accumulate ? 1 : 0)) MOFO_FAIL(MOFO_EINVAL, "mofo_gemm_wgrad_sliced: problems must agree on accumulate");
        accumulate = a[i].accumulate ? 1 : 0;
        R3Prob& p = g.p[i];
        p.A = full.A; p.B = full.B; p.colsum = full.colsum;
        p.C = ws + off;                     // slab 0 of this problem, dense rows of N
        p.M = full.M; p.N = full.N; p.K = full.K; p.lda = full.lda; p.ldb = full.ldb; p.ldc = full.N;
        p.k_per_split = ceil_div(ceil_div(full.K, slices), R4_KH) * R4_KH;     // (a slice past the end of a short reduction stores zeros)
        p.atomic = 0; p.skip_lo = full.colsum_skip_lo; p.skip_hi = full.colsum_skip_hi;
        g.start[i + 1] = g.start[i] + blocks;
        rp.C[i] = (float*)a[i].C; rp.off[i] = off; rp.M[i] = a[i].M; rp.N[i] = a[i].N; rp.ldc[i] = a[i].ldc;
        rp.start[i + 1] = rp.start[i] + (int)(((long long)a[i].M * a[i].N / 4 + 255) / 256);
        off += (long long)a[i].M * a[i].N;
    }
    for (int i = count; i < MAXR; ++i) {
        g.p[i] = g.p[0];
        g.start[i + 1] = g.start[count];
        rp.C[i] = rp.C[0]; rp.off[i] = 0; rp.M[i] = 0; rp.N[i] = 8; rp.ldc[i] = 8;
        rp.start[i + 1] = rp.start[count];
    }
    g.tail = 0;
    g.slices = slices;
    g.slab_stride = stride;
    const int total = slices * g.start[count];
    if (tm == R4_TM) {
        hipLaunchKernelGGL((gemm_r4_kernel<MOFO_EPI_F32>), dim3(r3_grid()), dim3(512), 0, s, g, total);
        ROUTE(ROUTE_R4);
    } else {
        hipLaunchKernelGGL((gemm_r3_kernel<OPL_COL, OPL_COL, MOFO_EPI_F32>), dim3(r3_grid()), dim3(512), 0, s, g, total);
        ROUTE(ROUTE_R3);
    }
    MOFO_CHECK_LAUNCH("mofo_gemm_wgrad_sliced(ring)");
    rp.count = count; rp.slices = slices; rp.accumulate = accumulate; rp.slab_stride = stride; rp.ws = ws;
    switch (slices) {
#define SLAB_REDUCE(S) case S: hipLaunchKernelGGL(wgrad_slab_reduce_kernel<S>, dim3(rp.start[count]), dim3(256), 0, s, rp); break
        SLAB_REDUCE(1); SLAB_REDUCE(2); SLAB_REDUCE(3); SLAB_REDUCE(4); SLAB_REDUCE(5); SLAB_REDUCE(6); SLAB_REDUCE(7); SLAB_REDUCE(8);
#undef SLAB_REDUCE
    }
    MOFO_CHECK_LAUNCH("mofo_gemm_wgrad_sliced(reduce)");
    return MOFO_OK;
}

extern "C" int mofo_gemm_grouped_plan(const mofo_gemm_args* a, int count, int* shared) {
    if (!a || !shared || count < 1) MOFO_FAIL(MOFO_EINVAL, "mofo_gemm_grouped_plan: null argument");
    const int mode = r3_mode();
    const int tm = ring_tm(a, count);
    if (mode != 0 && r3_legal(a, count) && (mode == 1 || r3_wanted(a, count, tm))) {
        R3Group g;
        const int rc = r3_fill(a, count, g, tm);
        if (rc) return rc;
        r3_plan(g, r3_grid(), shared);
        return 1;
    }
    for (int i = 0; i < count; ++i) shared[i] = (a[i].splits > 1 || a[i].accumulate) ? 1 : 0;
    return 0;
}

extern "C" int mofo_gemm_grouped(const mofo_gemm_args* a, int count, void* stream) {
    if (!a || count < 1) MOFO_FAIL(MOFO_EINVAL, "mofo_gemm_grouped: count must be 1..%d", MAXG);
    {
        const int mode = r3_mode();
        const int tm = ring_tm(a, count);
        if (mode != 0 && r3_legal(a, count) && (mode == 1 || r3_wanted(a, count, tm)))
            return tm == R4_TM ? r4_launch(a, count, (hipStream_t)stream) : r3_launch(a, count, (hipStream_t)stream);
    }
    if (count > MAXG) MOFO_FAIL(MOFO_EINVAL, "mofo_gemm_grouped: count must be 1..%d (1..%d for weight-gradient groups on the 256 x 128 ring kernel)", MAXG, MAXR);
    GroupP g;
    g.count = count;
    g.start[0] = 0;
    // 64-row tiles when 128x128 tiling would leave the 256 CUs with fewer than ~1.5 blocks each (NT / NN only)
    int mi = 4;
    if (a[0].op != MOFO_GEMM_TN) {
        long long t128 = 0;
        for (int i = 0; i < count; ++i) t128 += (long long)ceil_div(a[i].M, 128) * ceil_div(a[i].N, BN);
        if (t128 < 400) mi = 2;
        // 256-row tiles (persistent form only: one problem, no split-K / accumulate, not the pos epilogue).  Tile for tile
        // they are 10-15 % SLOWER than 128-row tiles (2 blocks per CU overlap less than 3: dec.fc1 dgrad 64 -> 74 us), so they
        // are used only where they repair the grid quantisation: the encoder's fc1 / fc2-dgrad GEMMs are 960 128-row tiles
        // for 768 resident blocks (1.25 rounds, the second one a quarter full) but 480 256-row tiles for 512 (0.94 of one round):
        // 36.6 -> 32.8 us and 35.9 -> 33.9 us alone, encoder step 4.99 -> 4.88 ms.  MOFO_GEMM_MI8=0 / 1 turns them off / forces them.
        static int mi8 = -2;
        if (mi8 == -2) {
            const char* e8 = getenv("MOFO_GEMM_MI8");
            mi8 = e8 ? atoi(e8) : -1;
        }
        if (mi != 2 && mi8 != 0 && count == 1 && a[0].splits <= 1 && !a[0].accumulate && a[0].epilogue != MOFO_EPI_POS_F32 &&
            a[0].epilogue != MOFO_EPI_POS_BF16 && a[0].op != MOFO_GEMM_NT_FP8) {   // (e4m3: 256-row tiles measured slower on 7 of 8 ViT-L shapes, profiles/r05_gemm_fp8_ab.txt)
            const long long t256 = (long long)ceil_div(a[0].M, 256) * ceil_div(a[0].N, BN);
            const double eff4 = (double)t128 / (double)(ceil_div((int)t128, 768) * 768);
            const double eff8 = (double)t256 / (double)(ceil_div((int)t256, 512) * 512);
            if (mi8 == 1 || eff8 > 1.25 * eff4) mi = 8;
        }
    }
    // a residual row map (RESID_F32 / RESID_BF16 with rows_in > 0) exists in the epilogues of the <= 64-row wave tiles only
    bool row_mapped = false;
    for (int i = 0; i < count; ++i)
        if ((a[i].epilogue == MOFO_EPI_RESID_F32 || a[i].epilogue == MOFO_EPI_RESID_BF16) && a[i].rows_in > 0) row_mapped = true;
    if (row_mapped && mi == 8) mi = 4;
    // 256 x 256 counted-vmcnt kernel (gemm8.h).  MOFO_GEMM8 = 0: never, 1: wherever it is built and legal, unset: by shape.
    {
        const char* e = getenv("MOFO_GEMM8");    // read per call: A/B switches inside one process (tools/gemm8_ab.py)
        const int mode = e ? atoi(e) : -1;
        bool legal = mode != 0 && !row_mapped && gemm8_has(a[0].op, a[0].epilogue);
        for (int i = 0; legal && i < count; ++i) {
            // NT / NN stream whole K-tile PAIRS of a k-contiguous operand: a K-tile past the end would read the next row, not zeros
            if (a[i].op != MOFO_GEMM_TN && (a[i].K % 128 || a[i].splits > 1)) legal = false;
            if (a[i].colsum) legal = false;      // the fused bias-gradient column sums ride on the 128 x 128 TN kernel only
        }
        if (legal && (mode == 1 || gemm8_wanted(a, count))) mi = 16;
    }
    // One 128 x 128 tile per CU with the reduction halved over two wave groups (gemm_k2.h): NT / NN problems whose 128 x 128 tiling
    // gives at most one tile per CU and whose reduction is worth a two-stage ring -- the encoder's N = 768 GEMMs at 5 120 rows.
    // MOFO_GEMM_K2 = 0: never, 1: wherever legal, unset: by shape.
    if (mi != 16) {
        const char* e = getenv("MOFO_GEMM_K2");     // read per call: A/B switches inside one process
        const int mode = e ? atoi(e) : -1;
        const bool legal = mode != 0 && count == 1 && (a[0].op == MOFO_GEMM_NT || a[0].op == MOFO_GEMM_NN) && a[0].splits <= 1 && !a[0].accumulate &&
                           a[0].K % 64 == 0;
        if (legal) {
            // same-process A/B at the ViT-B step's shapes (tools/gemm_k2_ab.py, profiles/r04_gemm_k2_ab.txt): 5120 x 768 x 3072 NT
            // 34.9 -> 32.5 us, NN 32.4 -> 29.2, K = 2304 NN 25.7 -> 23.5; at K = 768 (six k-stages per group) the old forms are at par
            const long long t128 = (long long)ceil_div(a[0].M, 128) * ceil_div(a[0].N, BN);
            if (mode == 1 ? t128 <= 4096 : (t128 <= 256 && a[0].K >= 1536)) mi = 32;
        }
    }
    for (int i = 0; i < count; ++i) {
        if (a[i].op != a[0].op || a[i].epilogue != a[0].epilogue) MOFO_FAIL(MOFO_EINVAL, "mofo_gemm_grouped: problems must share op and epilogue");
        int blocks = 0;
        const int rc = fill_problem(&a[i], g.p[i], mi == 16 ? 256 : (mi == 32 ? 128 : 32 * mi), mi == 16 ? 256 : BN, blocks);
        if (rc) return rc;
        g.start[i + 1] = g.start[i] + blocks;
    }
    for (int i = count; i < MAXG; ++i) {
        g.p[i] = g.p[0];
        g.start[i + 1] = g.start[count];
    }
    return dispatch(a[0].op, a[0].epilogue, g, mi, (hipStream_t)stream);
}

#ifdef MOFO_GEMM_TRACE
extern "C" int mofo_debug_trace_read(void* dst_host, size_t bytes) {
    return hipMemcpyFromSymbol(dst_host, HIP_SYMBOL(g_trace), bytes) == hipSuccess ? 0 : MOFO_ERUNTIME;
}
extern "C" int mofo_debug_trace8_read(void* dst_host, size_t bytes) {
    return hipMemcpyFromSymbol(dst_host, HIP_SYMBOL(g8_trace), bytes) == hipSuccess ? 0 : MOFO_ERUNTIME;
}
extern "C" int mofo_debug_trace_it_read(void* dst_host, size_t bytes) {
    return hipMemcpyFromSymbol(dst_host, HIP_SYMBOL(g_trace_it), bytes) == hipSuccess ? 0 : MOFO_ERUNTIME;
}
#endif

extern "C" int mofo_gemm_route_counts(long long* out, int n, int reset) {
    if ((!out && n > 0) || n < 0 || n > ROUTE_N) MOFO_FAIL(MOFO_EINVAL, "mofo_gemm_route_counts: out[0..n), n <= %d", (int)ROUTE_N);
    for (int k = 0; k < n; ++k) out[k] = __atomic_load_n(&g_route[k], __ATOMIC_RELAXED);
    if (reset)
        for (int k = 0; k < ROUTE_N; ++k) __atomic_store_n(&g_route[k], 0LL, __ATOMIC_RELAXED);
    return MOFO_OK;
}

extern "C" int mofo_gemm(const mofo_gemm_args* a, void* stream) {
    if (!a) MOFO_FAIL(MOFO_EINVAL, "mofo_gemm: null args");
    return mofo_gemm_grouped(a, 1, stream);
}
