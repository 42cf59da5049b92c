// gemm_r3: the 256 x 128 x 64 bf16 MFMA GEMM on a THREE-STAGE LDS ring with a counted vmcnt and REGISTER double-buffered fragments --
// one barrier per k-step (included into gemm.hip's anonymous namespace after gemm_k2.h; reuses the two LDS images, read_frag,
// G8_WAIT_VM and epilogue<>).  Built first for the grouped weight gradients (TN, f32 out): modeling_finetune.py:44-51,84,96's autograd.
//
// Why a third family ("The fill-rate wall"; round-4 review item 1).  The 128 x 128 kernels hold ONE k-stage in
// flight per block and wait `vmcnt(0)` + barrier every 64-deep step: a block alternates between waiting and multiplying, three
// co-resident blocks hide part of it, and a CU takes in 33-35 B/clk with 64 FLOP per staged byte.  gemm8 (256 x 256, 128 FLOP/B, four
// half-tiles always in flight) does not have that problem, but its wave tile leaves no registers for a second fragment set, so its
// LDS reads sit in a "load section" of their own between two barriers -- and for the reduction-strided operands of a weight gradient
// (twice the read instructions: ds_read_b64_tr_b16) that section is twice the MFMA section.  Here:
//   * tile 256 x 128, 8 waves as 4 (M) x 2 (N), wave tile 64 x 64 = acc[4][4] of v_mfma_f32_16x16x32_bf16 (64 accumulator VGPRs):
//     85 FLOP per staged byte, and room for TWO half-step fragment sets (64 VGPRs);
//   * fragments are read ONE K-SUBSTEP AHEAD: while the 16 MFMAs of substep ks run from one set, the other set is read from LDS, one
//     fragment pair behind every four MFMAs -- the LDS reads ride inside the MFMA stream of the wave instead of in front of it;
//   * ring of three 48-KiB stages (A 256 x 64, B 128 x 64): while step t is multiplied, stage t + 1 is being read, stages t + 2 and
//     t + 3 are in flight (96 KiB per CU across the barrier; `s_waitcnt vmcnt(6)`, never 0 inside the stream);
//   * ONE barrier per k-step, between its two substeps.  Order inside step t:  16 MFMAs of (t, ks 0) with the reads of (t, ks 1) |
//     lgkmcnt(0) | vmcnt(6) | barrier B_t | issue stage t + 3 into the buffer of stage t | 16 MFMAs of (t, ks 1) with the reads of
//     (t + 1, ks 0).
//     RAW: a wave waits for its OWN six pieces of stage t + 1 (counted), then the barrier, then anybody reads that stage.
//     WAR: the buffer of stage t is restaged after B_t; every wave finished reading it (lgkmcnt(0)) before it arrived at B_t.
// WORK LIST: units = (problem, split, 256 x 128 tile) in the XCD-aware order of the other kernels (an XCD label b & 7 walks one
// contiguous run of the list).  One block per CU walks its XCD's run round by round (unit j, j + 32, ...): the 32 blocks of a label run
// neighbouring tiles of one problem in k-lock-step, which is what lets one L2 fetch of an operand panel serve all of them.  The LAST,
// partial round of a run is not left to a fraction of the CUs: its units are cut into chunks of R3_CH k-steps and the chunks are dealt
// evenly over the label's blocks (a unit shared by several blocks is summed with f32 atomics onto a destination the caller has
// zeroed -- mofo_gemm_grouped_plan says which problems that concerns).  432 / 864 / 1296 / 2592 tiles of a 1 / 2 / 3 / 6-block
// encoder group are 0.84 / 1.69 / 2.53 / 5.06 rounds of 256: without the tail split every launch would run at 84 % fill.

#ifndef R3_NO_DMA
#define R3_NO_DMA 0
#endif
#ifndef R3_NO_MFMA
#define R3_NO_MFMA 0
#endif
#ifndef R3_NO_READ
#define R3_NO_READ 0
#endif
#ifndef R3_PRIO
#define R3_PRIO 1
#endif
#ifndef R3_LEAN
#define R3_LEAN 1
#endif
#ifndef R3_SGB
#define R3_SGB 1
#endif
constexpr int R3_TM = 256, R3_TN = 128;
constexpr int R3_A = R3_TM * 64 * 2;           // 32 KiB: ROW [256][64 k] or COL two [64 k][128] sub-images
constexpr int R3_B = R3_TN * 64 * 2;           // 16 KiB
constexpr int R3_STG = R3_A + R3_B;            // 48 KiB
constexpr int R3_RING = 3 * R3_STG;            // 144 KiB
constexpr int R3_CH = 4;                       // k-steps per chunk of the tail split
constexpr int MAXR = 32;                       // problems per launch (the compact descriptor below keeps 32 of them under 4 KiB of kernel arguments)

struct R3Prob {
    const bf16_t* A; const bf16_t* B; void* C; float* colsum;
    int M, N, K, lda, ldb, ldc;
    int k_per_split, atomic, skip_lo, skip_hi;
};
struct R3Group {
    R3Prob p[MAXR];
    int start[MAXR + 1];       // first unit of each problem; start[count] = units of the launch (SLICED: of one slice)
    int count;
    int tail;                  // 1: cut the last partial round of every XCD run into chunks (else: plain rounds)
    // SLICED (slices > 0; mofo_gemm_wgrad_sliced): the list is SLICE-major -- unit w = (slice w / U, problem, tile) with U = start[count]
    // -- and slice s of EVERY problem is reduction rows [s k_per_split, (s + 1) k_per_split).  With slices = 8 the XCD label x = blockIdx & 7
    // owns exactly slice x: all the rounds of an XCD stream ONE row range of the operands through its L2 (each operand byte is fetched
    // by one XCD only), and a round's 32 units are neighbouring tiles of one or two problems in k-lock-step.  Results are PARTIAL sums:
    // plainly stored to slab `slice` of a workspace (p.C + slice * slab_stride, dense rows of N), summed by wgrad_slab_reduce_kernel.
    int slices;
    long long slab_stride;     // f32 elements between two slices' slabs
};

struct R3Seg {
    int gi, m0, n0;
    int k0, kend;              // first reduction row of the segment; end of the UNIT's reduction range (operand extent for the range check)
    int nk;                    // k-steps of the segment
    int atomic;
    int slice;                 // SLICED: which slab the result goes to (else 0)
};

// unit `wg` (index into the launch's (problem, split, tile) list) -> its whole-unit segment.  TM: tile rows (256: gemm_r3, 384: gemm_r4);
// KS: reduction rows per k-step of the kernel's ring (64 / 32)
template <int TM, int KS>
__device__ __forceinline__ R3Seg ring_unit(const R3Group& G, int wg) {
    int slice = 0;
    if (G.slices > 0) {
        const int U = G.start[G.count];
        slice = wg / U;
        wg -= slice * U;
    }
    int gi = 0;
#pragma nounroll
    for (int k = 1; k < G.count; ++k)
        if (wg >= G.start[k]) gi = k;
    const R3Prob& p = G.p[gi];
    const int tiles_n = (p.N + R3_TN - 1) / R3_TN, tiles_m = (p.M + TM - 1) / TM;
    const int tiles = tiles_n * tiles_m;
    wg -= G.start[gi];
    int split = slice;
    if (G.slices <= 0) {
        split = wg / tiles;
        wg -= split * tiles;
    }
    R3Seg s;
    s.gi = gi;
    s.m0 = (tiles_n <= tiles_m ? wg / tiles_n : wg % tiles_m) * TM;
    s.n0 = (tiles_n <= tiles_m ? wg % tiles_n : wg / tiles_m) * R3_TN;
    s.k0 = split * p.k_per_split;
    s.kend = min(p.K, s.k0 + p.k_per_split);
    s.nk = s.kend > s.k0 ? (s.kend - s.k0 + KS - 1) / KS : 0;
    s.atomic = p.atomic;
    s.slice = slice;
    return s;
}
__device__ __forceinline__ R3Seg r3_unit(const R3Group& G, int wg) { return ring_unit<R3_TM, BK>(G, wg); }

template <int LA, int LB, int EPI>
__global__ __launch_bounds__(512, 2) void gemm_r3_kernel(R3Group G, int total) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[R3_RING];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;

    // ---- this block's share of the work list.  Label x = blockIdx & 7 (blocks b and b + 8 share an XCD) owns units [xbeg, xbeg + xlen);
    // the label's nbx blocks walk them round by round; the units past the last full round are the tail.
    const int nbx = (int)gridDim.x >> 3, jx = (int)blockIdx.x >> 3, xcd = (int)blockIdx.x & 7;
    const int q = total >> 3, r = total & 7;
    const int xbeg = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    const int xlen = q + (xcd < r ? 1 : 0);
    const int rounds = G.tail ? xlen / nbx : (xlen + nbx - 1) / nbx;     // rounds of whole units
    const int tail0 = rounds * nbx;                                        // first tail unit of the run (G.tail only)
    // tail: chunk range [clo, chi) of the run's tail units
    int clo = 0, chi = 0;
    if (G.tail && tail0 < xlen) {
        int ctot = 0;
        for (int u = tail0; u < xlen; ++u) ctot += (r3_unit(G, xbeg + u).nk + R3_CH - 1) / R3_CH;
        clo = (int)(((long long)ctot * jx) / nbx);
        chi = (int)(((long long)ctot * (jx + 1)) / nbx);
    }

    // per-lane source offsets of the LDS-DMA pieces (the tile / piece / k position goes into the wave-uniform SGPR offset)
    auto lane_off = [&](int layout, int ld) -> int {
        if (layout == OPL_ROW) return ((lane >> 3) * ld + (((lane & 7) ^ ((lane >> 3) & 7)) << 3)) * 2;
        // COL image [64 k][128 cols]: piece i = k-rows 4i .. 4i+3; a wave stages pieces 2 wave, 2 wave + 1 of a (sub-)image, so the
        // piece-dependent bit of the swizzle key, (i >> 1) & 1, is wave & 1 for all of them
        const int cpos = lane & 15, kq = lane >> 4;
        const int key = kq | ((wave & 1) << 2);
        const int g = ((((cpos >> 1) ^ key) << 1) | (cpos & 1));
        return (kq * ld + g * 8) * 2;
    };
    bf16x8 ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = (__bf16)1.0f;
    constexpr bool CAN_COLSUM = (LA == OPL_COL && LB == OPL_COL && EPI == MOFO_EPI_F32);

    // ---- the next segment of this block: whole units round by round, then its chunks of the run's last units
    int rd = 0, ut = tail0, c0 = 0;
    auto next_seg = [&](R3Seg& s) -> bool {
        while (rd < rounds) {
            const int u = rd * nbx + jx;
            ++rd;
            if (u >= xlen) continue;
            s = r3_unit(G, xbeg + u);
            return true;
        }
        while (clo < chi && ut < xlen && c0 < chi) {
            s = r3_unit(G, xbeg + ut);
            ++ut;
            const int nch = (s.nk + R3_CH - 1) / R3_CH;
            const int lo = max(clo, c0) - c0, hi = min(chi, c0 + nch) - c0;
            c0 += nch;
            if (lo >= hi) continue;
            const int ks0 = lo * R3_CH, ks1 = min(s.nk, hi * R3_CH);
            if (ks0 > 0 || ks1 < s.nk) s.atomic = 1;   // the unit is shared: summed with f32 atomics onto a zeroed destination
            s.k0 += ks0 * BK;
            s.nk = ks1 - ks0;
            return true;
        }
        return false;
    };

    // what staging needs of a segment (wave-uniform except va / vb)
    struct Ctx {
        __amdgpu_buffer_rsrc_t ra, rb;
        int lda, ldb, va, vb, m0, n0, k0;
    };
    auto make_ctx = [&](const R3Seg& sg) -> Ctx {
        const R3Prob& p = G.p[sg.gi];
        Ctx c;
        const size_t ext_a = (LA == OPL_ROW ? ((size_t)p.M - 1) * p.lda + sg.kend : ((size_t)sg.kend - 1) * p.lda + p.M) * 2;
        const size_t ext_b = (LB == OPL_ROW ? ((size_t)p.N - 1) * p.ldb + sg.kend : ((size_t)sg.kend - 1) * p.ldb + p.N) * 2;
        c.ra = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, (int)ext_a, 0x00020000);
        c.rb = __builtin_amdgcn_make_buffer_rsrc((void*)p.B, 0, (int)ext_b, 0x00020000);
        c.lda = p.lda;
        c.ldb = p.ldb;
        c.va = lane_off(LA, p.lda);
        c.vb = lane_off(LB, p.ldb);
        c.m0 = sg.m0;
        c.n0 = sg.n0;
        c.k0 = sg.k0;
        return c;
    };
    // piece IDX (0-3: A, 4-5: B) of the six 1-KiB pieces of k-step t of segment c that this wave stages (A: 32 pieces, B: 16, 8 waves)
    auto piece = [&](auto idx_tag, const Ctx& c, int t, int buf) {
        constexpr int IDX = decltype(idx_tag)::value;
#if R3_NO_DMA
        return;                                    // timing-only ablation build (wrong results)
#endif
        unsigned char* dst = smem + buf * R3_STG;
        const int kk = c.k0 + t * BK;
        if constexpr (IDX < 4) {
            if constexpr (LA == OPL_ROW) {
                const int i = wave * 4 + IDX;
                const unsigned soff = ((unsigned)(c.m0 + 8 * i) * (unsigned)c.lda + (unsigned)kk) * 2u;
                lds_dma16<R3_LEAN != 0>(c.ra, dst + i * 1024, c.va, soff);
            } else {
                constexpr int h = IDX >> 1;
                const int i = wave * 2 + (IDX & 1);
                const unsigned soff = ((unsigned)(kk + 4 * i) * (unsigned)c.lda + (unsigned)(c.m0 + 128 * h)) * 2u;
                lds_dma16<R3_LEAN != 0>(c.ra, dst + h * 16384 + i * 1024, c.va, soff);
            }
        } else {
            const int i = wave * 2 + (IDX - 4);
            const unsigned soff = LB == OPL_ROW ? ((unsigned)(c.n0 + 8 * i) * (unsigned)c.ldb + (unsigned)kk) * 2u
                                                : ((unsigned)(kk + 4 * i) * (unsigned)c.ldb + (unsigned)c.n0) * 2u;
            lds_dma16<R3_LEAN != 0>(c.rb, dst + R3_A + i * 1024, c.vb, soff);
        }
    };
    auto pieces_a = [&](const Ctx& c, int t, int buf) {
        piece(std::integral_constant<int, 0>{}, c, t, buf);
        piece(std::integral_constant<int, 1>{}, c, t, buf);
        piece(std::integral_constant<int, 2>{}, c, t, buf);
        piece(std::integral_constant<int, 3>{}, c, t, buf);
    };
    auto pieces_b = [&](const Ctx& c, int t, int buf) {
        piece(std::integral_constant<int, 4>{}, c, t, buf);
        piece(std::integral_constant<int, 5>{}, c, t, buf);
    };
    auto read_a = [&](const unsigned char* st, int i, int ks) -> bf16x8 {
        if constexpr (LA == OPL_ROW) return read_frag<OPL_ROW>(st, wm * 64 + 16 * i, ks, lane);
        else return read_frag<OPL_COL>(st + (wm >> 1) * 16384, (wm & 1) * 64 + 16 * i, ks, lane);
    };
    auto read_b = [&](const unsigned char* st, int j, int ks) -> bf16x8 { return read_frag<LB>(st + R3_A, wn * 64 + 16 * j, ks, lane); };

    // Fragment sets by k-substep: fa / fb[0] hold ks = 0 of the step being multiplied (or of the next one), fa / fb[1] hold ks = 1.
    // Half-step (t, 0) multiplies set 0 while set 1 is read from stage t; half-step (t, 1) multiplies set 1 while set 0 is read from
    // stage t + 1.  The barrier sits BETWEEN the halves: by then this wave has read all of stage t (its buffer is restaged right
    // after the barrier) and its six pieces of stage t + 1 have landed.
    bf16x8 fa[2][4], fb[2][4];
    f32x4 acc[4][4], accb[4];
    auto zero_acc = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            accb[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };

    // TRIED AND NOT KEPT (commit cf2a863 has the code; profiles/r05_gemm_r3_ab.txt the numbers):
    //  * the stream running on into the next segment (its stages 0-2 issued from this one's last steps, results stored straight from
    //    the accumulators because the ring is never idle): 0.96-0.98 x -- fragment-layout stores are 64-B segments, they hold the
    //    in-order vmcnt queue longer than the drain they save, and f32 atomics in 64-B segments run at a fraction of the 256-B-row rate;
    //  * de-phasing the two waves of a SIMD (bursts after MFMAs 2, 6, 10, 14 of a half for waves 4-7): 0.95 x;
    //  * pieces issued together behind the barrier, reads spread one pair per MFMA group: 0.96 x; no s_setprio, no sched_group_barrier: +-1 %.
    R3Seg cs;
    while (next_seg(cs)) {
        const Ctx cur = make_ctx(cs);
        const int nk = cs.nk;
        const R3Prob& p = G.p[cs.gi];
        const bool do_colsum = CAN_COLSUM && p.colsum != nullptr && cs.n0 == 0 && wn == 0;
        int bt = 0;                                // ring buffer of the current stage t
        zero_acc();
        {
            // ---- prologue: stages 0, 1 and the A pieces of stage 2 in flight; ks = 0 fragments of step 0 on their way to registers
            pieces_a(cur, 0, 0);
            pieces_b(cur, 0, 0);
            pieces_a(cur, 1, 1);
            pieces_b(cur, 1, 1);
            pieces_a(cur, 2, 2);
            G8_WAIT_VM(10);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                fa[0][i] = read_a(smem, i, 0);
                fb[0][i] = read_b(smem, i, 0);
#if R3_NO_READ
                fa[1][i] = read_a(smem, i, 1);
                fb[1][i] = read_b(smem, i, 1);
#endif
            }
        }
        // 16 (20) MFMAs of one k-substep from set KS with four BURSTS of other work behind MFMAs 4, 8, 12, 16: burst 0 = the B fragment
        // reads of the OTHER set from `src` (8 ds_read_b64_tr_b16), burst 1 = its A fragment reads, bursts 2, 3 = the LDS-DMA pieces
        // `dma(2 / 3)` -- the issue cost of a piece (60-185 clk of the wave's instruction stream, MI355X_MICROARCH.md) stays out of the
        // head of the half.  Ablation builds (profiles/r05_gemm_r3_ablate.txt) show the cost of a k-step to be close to the SUM of its
        // parts -- 0.45 us of MFMAs + 0.17 of fragment reads + 0.27 of piece issue against 0.93 measured: a wave issues in order, and
        // while it issues reads or pieces it issues no MFMA.
        auto half = [&](auto ks_tag, auto cs_tag, const unsigned char* src, auto dma) {
            constexpr int KS = decltype(ks_tag)::value;
            constexpr int PH = 0;
            constexpr bool CS = decltype(cs_tag)::value;
            auto burst = [&](int b) {
#if !R3_NO_READ
                if (b == 0) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) fb[KS ^ 1][j] = read_b(src, j, KS ^ 1);
                } else if (b == 1) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) fa[KS ^ 1][j] = read_a(src, j, KS ^ 1);
                }
#endif
                if (b >= 2) dma(b);
                __builtin_amdgcn_sched_barrier(0);
            };
#if R3_PRIO
            __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
            for (int n = 0; n < 16; ++n) {
                const int i = n >> 2, j = n & 3;
#if R3_NO_MFMA
                asm volatile("" ::"v"(fb[KS][j]), "v"(fa[KS][i]));    // timing-only ablation build: operands kept alive
#else
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[KS][j], fa[KS][i], acc[i][j], 0, 0, 0);
                if constexpr (CS)
                    if (j == 3) accb[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, fa[KS][i], accb[i], 0, 0, 0);
#endif
                if (((n + 1 + PH) & 3) == 0) {
                    __builtin_amdgcn_sched_barrier(0);
                    burst(((n + 1 + PH) >> 2) - 1);
                }
            }
#if R3_PRIO
            __builtin_amdgcn_s_setprio(0);
#endif
        };
        // one k-step.  Pieces: B of stage t + 2 in half (t, 0), A of stage t + 3 in half (t, 1).  Stages past the segment's last k-step are
        // issued all the same (ONE straight-line loop body: no tail variants, the accumulators stay in place): past the unit's reduction
        // range the hardware range check drops them, inside it (a tail chunk that ends before its unit does) they fetch three stages
        // nobody multiplies.
        // vmcnt at B_t: stage t + 1's last pieces (its B pieces, issued in half (t - 1, 0)) have A(t + 2) and B(t + 2) behind them = 6.
        auto step = [&](auto cs_tag, int t) {
            const int bn = bt == 2 ? 0 : bt + 1;   // buffer of stage t + 1
            const int bp = bt == 0 ? 2 : bt - 1;   // buffer of stage t + 2 (= of stage t - 1)
            half(std::integral_constant<int, 0>{}, cs_tag, smem + bt * R3_STG, [&](int g) {
                if (g == 2) piece(std::integral_constant<int, 4>{}, cur, t + 2, bp);
                else piece(std::integral_constant<int, 5>{}, cur, t + 2, bp);
            });
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // set 1 is in registers: this wave has read all of stage t
            G8_WAIT_VM(6);                         // stage t + 1 landed (this wave's pieces); six younger pieces keep flying
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();          // B_t: stage t + 1 is visible to every wave, the buffer of stage t is free
            __builtin_amdgcn_sched_barrier(0);
            half(std::integral_constant<int, 1>{}, cs_tag, smem + bn * R3_STG, [&](int g) {
                if (g == 2) {
                    piece(std::integral_constant<int, 0>{}, cur, t + 3, bt);
                    piece(std::integral_constant<int, 1>{}, cur, t + 3, bt);
                } else {
                    piece(std::integral_constant<int, 2>{}, cur, t + 3, bt);
                    piece(std::integral_constant<int, 3>{}, cur, t + 3, bt);
                }
            });
            __builtin_amdgcn_sched_barrier(0);
            bt = bn;
        };
        auto loops = [&](auto cs_tag) {
            for (int t = 0; t < nk; ++t) step(cs_tag, t);
        };
        if constexpr (CAN_COLSUM) {
            if (do_colsum) loops(std::true_type{});
            else loops(std::false_type{});
        } else {
            loops(std::false_type{});
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // the look-ahead reads of the step after the last one
        G8_WAIT_VM(0);                             // ... and the stages issued past the end have landed (or were dropped)
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();              // every wave is done with the ring: it becomes the epilogue's staging area
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (CAN_COLSUM) {
            if (do_colsum && lane < 16) {          // D[n][m]: every row n holds the same sum; lanes 0..15 hold m = 16 i + lane in element 0
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int m = cs.m0 + wm * 64 + 16 * i + lane;
                    if (m < p.M && !(m >= p.skip_lo && m < p.skip_hi)) atomicAdd(p.colsum + m, accb[i][0]);
                }
            }
        }
        const bool full = (cs.m0 + R3_TM <= p.M) && (cs.n0 + R3_TN <= p.N);
        // ---- results through the (drained) ring: the shared epilogue stages the wave's 64 x 64 f32 tile in LDS and leaves in whole
        // 256-B row segments (plain stores, or one 256-B row per f32 atomic instruction for a shared / split / accumulating unit)
        {
            GemmP pe = {};
            pe.C = (float*)p.C + (long long)cs.slice * G.slab_stride;
            pe.M = p.M;
            pe.N = p.N;
            pe.ldc = p.ldc;
            pe.atomic = cs.atomic;
            epilogue<EPI, 4, 1>(pe, acc, (float*)smem + wave * (64 * 64), cs.m0 + wm * 64, cs.n0 + wn * 64, full, lane, false);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();          // the staging area becomes the next segment's ring
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}
