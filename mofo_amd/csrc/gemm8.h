// gemm8: the 256 x 256 x 64 bf16 MFMA GEMM with a COUNTED-vmcnt pipeline (included into gemm.hip's anonymous namespace, after
// epilogue<>; it reuses GemmP / GroupP, the two LDS images and their fragment reads, and every epilogue).
//
// Why a second family.  The 128 x 128 kernels keep ONE k-stage in flight per block: every k-step ends in `s_waitcnt vmcnt(0)` +
// barrier, so a block pays a full L2 round trip per 64-deep step (profiles/r02_gemm_loop_anatomy.txt: 2 838 clk per k-step for 682
// clk of MFMA) and only three co-resident blocks per CU hide part of it.  Here ONE block of 8 waves owns the CU, the operand ring
// (2 x 64 KiB) always has four 16-KiB half-tiles in flight across the barriers (`s_waitcnt vmcnt(8)`, never 0 inside the stream),
// and the two 4-wave groups run half a phase apart, so that one group's 16-MFMA cluster covers the other group's LDS reads and
// LDS-DMA issue (guide: cdna_hip_programming.md section 5, "The 256^2 8-phase template").
//
// Geometry: 512 threads = 8 waves as 2 (M) x 4 (N); a wave owns 128 x 64 of C = acc[8][4] tiles of v_mfma_f32_16x16x32_bf16,
// issued as mfma(Bfrag, Afrag) like the other kernels (a lane holds 4 consecutive n of one m -> the shared epilogues apply).
// One K-tile (64 deep) = 4 phases of 16 MFMAs per wave: phase q = rows 32 q .. 32 q + 31 of the wave (i = 2q, 2q+1) x all four
// column tiles; the B fragments (32 VGPRs) are read once per K-tile, the A fragments (16 VGPRs) per phase.
// LDS ring: 2 buffers x {Ah0, Ah1, Bh0, Bh1}, each a 16-KiB half-tile of 128 rows (or columns) x 64 k, INTERLEAVED so that every
// half has one short reading window per K-tile:
//   Ah(h) = tile rows  wr*128 + h*64 + [0,64)   for wr = 0,1      (the waves' i = 4h .. 4h+3: read in phases 2h, 2h+1)
//   Bh(h) = tile cols  wc*64  + h*32 + [0,32)   for wc = 0..3     (the waves' j = 2h, 2h+1:   read in phase 0)
//   ds_read_b128 per phase: P0 8 (B) + 4 (A), P1 4, P2 4, P3 4
// A half is restaged two phases after its last reading phase at the earliest (WAR: the staggered group's reads retire one barrier
// later), and read one phase after the counted wait that retires it at the earliest (RAW: each wave waits for its own pieces,
// then a barrier).  Staging slots while K-tile t is multiplied (half of K-tile -> phase):
//   P0: Ah0(t+1)   P1: Ah1(t+1)   P2: Bh0(t+2)   P3: Bh1(t+2)
// and the waits, each covering what the NEXT phase starts to read, with 3-4 younger half-tiles left in flight:
//   P1: vmcnt(8) -> Ah1(t)      P3: vmcnt(6) -> Bh0(t+1), Bh1(t+1), Ah0(t+1)
// PERSISTENT: a block walks tiles w, w + grid, ...; the staging stream runs on into the next tile (its first two K-tiles' halves
// are issued from the last two K-tiles of the current one), the epilogue has its own 32 KiB of LDS (16 rows x 64 f32 per wave and
// pass, 8 passes), and the first K-tile after an epilogue waits with vmcnt(8 + S): the epilogue's S stores sit in the same
// in-order counter between the prefetched and the new loads (MI355X_MICROARCH.md: loads, stores and LDS-DMA count together).
// The last two K-tiles of a block's stream wait with vmcnt(0) (nothing younger is issued any more).

#ifndef G8_DMA_IN_M
#define G8_DMA_IN_M 0
#endif
constexpr int G8_HALF = 16384;                 // one half-tile: 128 rows x 64 k bf16 (ROW) or 64 k x 128 cols (COL)
constexpr int G8_BUF = 4 * G8_HALF;            // Ah0 Ah1 Bh0 Bh1
constexpr int G8_RING = 2 * G8_BUF;            // 128 KiB
constexpr int G8_EPB = 8 * 4096;               // epilogue staging: 16 rows x 64 f32 per wave
constexpr int G8_TM = 256, G8_TN = 256;

struct G8Tile {
    int gi, m0, n0, kbeg, nk;                  // problem, tile origin, first reduction index, K-tiles
};

// per-lane byte offset of a DMA piece's source (the piece / tile / k position goes into the wave-uniform SGPR offset)
template <int LAYOUT, bool IS_A>
__device__ __forceinline__ int g8_lane_offset(int ld, int lane, int wave) {
    if constexpr (LAYOUT == OPL_ROW) {
        return ((lane >> 3) * ld + (((lane & 7) ^ ((lane >> 3) & 7)) << 3)) * 2;
    } else {
        // COL image [64 k][128 cols]: piece i = k-rows 4i .. 4i+3; lane -> k-row kq of the piece, 16-B chunk position cpos.  The
        // 32-B unit at position u of k-row r holds logical unit u ^ key(r), key(r) = (r & 3) | ((r >> 3) & 1) << 2; a wave stages
        // pieces 2 wave, 2 wave + 1, so (r >> 3) & 1 = wave & 1 for both.
        const int cpos = lane & 15, kq = lane >> 4;
        const int key = kq | ((wave & 1) << 2);
        const int g = ((((cpos >> 1) ^ key) << 1) | (cpos & 1));      // logical 8-column chunk 0..15 of the half
        const int lc = g * 8;
        const int col = IS_A ? ((lc >> 6) * 128 + (lc & 63)) : ((lc >> 5) * 64 + (lc & 31));   // interleaved halves (+ h * 64 / 32, uniform)
        return (kq * ld + col) * 2;
    }
}

// the two 1-KiB pieces of half-tile h that this wave stages (16 pieces per half, 8 waves)
template <int LAYOUT, bool IS_A>
__device__ __forceinline__ void g8_stage_half(__amdgpu_buffer_rsrc_t r, int v, int ld, int d0, int k0, int h, unsigned char* dst, int wave_u) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int i = wave_u * 2 + j;
        unsigned soff;
        if constexpr (LAYOUT == OPL_ROW) {
            const int rb = IS_A ? ((i >> 3) * 128 + h * 64 + (i & 7) * 8) : ((i >> 2) * 64 + h * 32 + (i & 3) * 8);
            soff = ((unsigned)(d0 + rb) * (unsigned)ld + (unsigned)k0) * 2u;
        } else {
            soff = ((unsigned)(k0 + 4 * i) * (unsigned)ld + (unsigned)(d0 + (IS_A ? h * 64 : h * 32))) * 2u;
        }
#if MOFO_DMA_ASM_G8
        lds_dma16<false>(r, dst + i * 1024, v, soff);       // inline asm: see lds_dma16 in gemm.hip
#else
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r, LDS_PTR(dst + i * 1024), 16, v, (int)soff, 0, 0);
#endif
    }
}

// A LOWER bound on the vector-memory operations (loads AND stores: one in-order counter) a wave issues in the epilogue of one FULL
// 128 x 64 wave tile -- the vmcnt budget of the first K-tile after it: `vmcnt(8 + S)` retires the prefetched pieces that sit BEHIND
// the epilogue's S operations in the counter only if the wave really issued >= S of them.  A larger S waits for LESS, so S must never
// exceed the true count; a smaller S over-waits -- and what it then waits for are the epilogue's own first STORES, an HBM round trip
// (the f32-residual epilogue interleaves its 32 residual loads with its 32 stores in chunks of 8: with S = 32 the wait would cover
// its first 8 stores).
// Counts per epilogue of gemm.hip, FULL path (no row group is branched around there; every load below is a separate dwordx4 / dwordx2):
//   BF16 16 stores | BIAS_GELU 16 + 16 stores | DGELU_BF16, RESID_BF16 16 aux loads + 16 stores | F32 32 stores (the accumulating
//   form: 128 atomics) | RESID_F32 32 residual loads + 32 stores | POS_F32 / POS_BF16 32 position-row loads + 32 stores (+ index loads).
// Ragged tiles take the bounds-checked path, whose row groups may be branched around: the caller falls back to S = 0 after them.
template <int EPI>
constexpr int g8_epi_stores() {
    return EPI == MOFO_EPI_BF16 ? 16
         : EPI == MOFO_EPI_BIAS_GELU || EPI == MOFO_EPI_DGELU_BF16 || EPI == MOFO_EPI_RESID_BF16 ? 32
         : EPI == MOFO_EPI_RESID_F32 || EPI == MOFO_EPI_POS_F32 || EPI == MOFO_EPI_POS_BF16 ? 64
         : 32;   // F32
}

#ifdef MOFO_GEMM_TRACE
// debug build only (tools/gemm8_trace.py): s_memtime stamps of one K-tile (the 7th of a block's stream), wave 0 (group 0) and wave 4 (group 1)
__device__ unsigned long long g8_trace[256 * 64];
#define G8_STAMP(slot)                                                                                                   \
    do {                                                                                                                 \
        if (trace_on && (threadIdx.x & 255) == 0 && blockIdx.x < 256)                                                    \
            g8_trace[blockIdx.x * 64 + (threadIdx.x >> 8) * 32 + (slot)] = __builtin_readcyclecounter();                   \
    } while (0)
#else
#define G8_STAMP(slot)
#endif

template <int LA, int LB, int EPI>
__global__ __launch_bounds__(512, 2) void gemm8_kernel(GroupP G, int total) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[G8_RING + G8_EPB];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;

    // ---- tile list: the XCD-aware order of the other kernels over the launch's (problem, split, tile) list
    auto decode = [&](int w) -> G8Tile {
        G8Tile t;
        const int q = total >> 3, r = total & 7, xcd = w & 7;
        int wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (w >> 3);
        int gi = 0;
#pragma unroll
        for (int k = 1; k < MAXG; ++k)
            if (k < G.count && wg >= G.start[k]) gi = k;
        const GemmP& p = G.p[gi];
        const int tiles_n = (p.N + G8_TN - 1) / G8_TN, tiles_m = (p.M + G8_TM - 1) / G8_TM;
        const int tiles = tiles_n * tiles_m;
        wg -= G.start[gi];
        const int split = wg / tiles;
        wg -= split * tiles;
        t.gi = gi;
        t.m0 = (tiles_n <= tiles_m ? wg / tiles_n : wg % tiles_m) * G8_TM;
        t.n0 = (tiles_n <= tiles_m ? wg % tiles_n : wg / tiles_m) * G8_TN;
        t.kbeg = split * p.k_per_split;
        const int kend = min(p.K, t.kbeg + p.k_per_split);
        t.nk = (kend - t.kbeg + BK - 1) / BK;
        t.nk += t.nk & 1;                     // K-tiles come in pairs (buffer parity at compile time); a K-tile past the end reads zeros
        return t;
    };
    struct Ctx {                               // what staging needs of a tile (wave-uniform except va / vb)
        __amdgpu_buffer_rsrc_t ra, rb;
        int lda, ldb, va, vb, m0, n0, kbeg;
    };
    auto make_ctx = [&](const G8Tile& t) -> Ctx {
        const GemmP& p = G.p[t.gi];
        Ctx c;
        const int kend = min(p.K, t.kbeg + p.k_per_split);
        const size_t ext_a = (LA == OPL_ROW ? ((size_t)p.M - 1) * p.lda + kend : ((size_t)kend - 1) * p.lda + p.M) * 2;
        const size_t ext_b = (LB == OPL_ROW ? ((size_t)p.N - 1) * p.ldb + kend : ((size_t)kend - 1) * p.ldb + p.N) * 2;
        c.ra = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, (int)ext_a, 0x00020000);
        c.rb = __builtin_amdgcn_make_buffer_rsrc((void*)p.B, 0, (int)ext_b, 0x00020000);
        c.lda = p.lda;
        c.ldb = p.ldb;
        c.va = g8_lane_offset<LA, true>(p.lda, lane, wave);
        c.vb = g8_lane_offset<LB, false>(p.ldb, lane, wave);
        c.m0 = t.m0;
        c.n0 = t.n0;
        c.kbeg = t.kbeg;
        return c;
    };

    int w = blockIdx.x;
    G8Tile tc = decode(w);
    Ctx cur = make_ctx(tc);
    // half `sel` (0: Bh0, 1: Ah0, 2: Bh1, 3: Ah1) of K-tile kk of the tile described by c, into ring buffer `buf`
    auto stage = [&](const Ctx& c, int sel, int kk, int buf) {
        unsigned char* base = smem + buf * G8_BUF;
        const int k0 = c.kbeg + kk * BK;
        if (sel & 1) g8_stage_half<LA, true>(c.ra, c.va, c.lda, c.m0, k0, sel >> 1, base + (sel >> 1) * G8_HALF, wave);
        else g8_stage_half<LB, false>(c.rb, c.vb, c.ldb, c.n0, k0, sel >> 1, base + (2 + (sel >> 1)) * G8_HALF, wave);
    };

    // ---- prologue of the stream: K-tile 0 whole, the h0 halves of K-tile 1 (the order the steady state would have issued them in)
#if G8_DMA_IN_M
    stage(cur, 0, 0, 0);
    stage(cur, 1, 0, 0);
    stage(cur, 2, 0, 0);
    stage(cur, 3, 0, 0);
    stage(cur, 0, 1, 1);
    stage(cur, 1, 1, 1);
    stage(cur, 2, 1, 1);
    G8_WAIT_VM(8);                             // Bh0(0), Ah0(0), Bh1(0) landed (this wave's pieces)
#else
    stage(cur, 0, 0, 0);
    stage(cur, 2, 0, 0);
    stage(cur, 1, 0, 0);
    stage(cur, 3, 0, 0);
    stage(cur, 0, 1, 1);
    stage(cur, 2, 1, 1);
    G8_WAIT_VM(6);                             // Bh0(0), Bh1(0), Ah0(0) landed (this wave's pieces)
#endif
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);

    float* ep = (float*)(smem + G8_RING) + wave * (16 * 64);
    bool first_tile = true;
    bool prev_full = false;                    // the previous tile took the branch-free (full-tile) epilogue: its store count is known
    for (;;) {
        const int wnext = w + (int)gridDim.x;
        const bool has_next = wnext < total;
        G8Tile tn = tc;
        if (has_next) tn = decode(wnext);
        const Ctx nxt = make_ctx(tn);
        const GemmP& p = G.p[tc.gi];
        const int nk = tc.nk;

        f32x4 acc[8][4];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

        if (wr == 1) __builtin_amdgcn_s_barrier();          // stagger: waves 4-7 run half a phase behind waves 0-3

        // one K-tile (ring buffer BUF) = 4 phases; phase q multiplies the wave's rows 32 q .. 32 q + 31 with all four column tiles
        auto ktile = [&](auto buf_tag, int kt) {
            constexpr int BUF = decltype(buf_tag)::value;
            const unsigned char* bb = smem + BUF * G8_BUF;
            // which tile do K-tiles kt+1 / kt+2 belong to (the stream runs on into the next tile)
            const bool o1 = kt + 1 >= nk, o2 = kt + 2 >= nk;
            const bool e1 = !o1 || has_next, e2 = !o2 || has_next;
            const bool tail = !e2;                          // last two K-tiles of the stream: nothing younger to leave in flight
            const bool after_epi = (kt == 0) && !first_tile && prev_full;   // else: plain W1 / W3 (waits for the epilogue's stores too)
#ifdef MOFO_GEMM_TRACE
            const bool trace_on = first_tile && kt == 6;
#endif
            bf16x8 bfr[4][2], af[2][2];
            auto read_a = [&](int q) {
#pragma unroll
                for (int ii = 0; ii < 2; ++ii)
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) af[ii][ks] = read_frag<LA>(bb + (q >> 1) * G8_HALF, wr * 64 + (q & 1) * 32 + 16 * ii, ks, lane);
            };
            auto mfma4 = [&](int q, int ks, int ii) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[2 * q + ii][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j][ks], af[ii][ks], acc[2 * q + ii][j], 0, 0, 0);
            };
            // the 16 MFMAs of phase q; `dma` (the phase's two LDS-DMA pieces) is issued after the first four
            auto mfma_q = [&](int q, auto dma) {
                __builtin_amdgcn_s_setprio(1);
                mfma4(q, 0, 0);
#if G8_DMA_IN_M
                __builtin_amdgcn_sched_barrier(0);
                dma();
                __builtin_amdgcn_sched_barrier(0);
#endif
                mfma4(q, 0, 1);
                mfma4(q, 1, 0);
                mfma4(q, 1, 1);
                __builtin_amdgcn_s_setprio(0);
            };
            auto lsec = [&](auto dma) {
#if !G8_DMA_IN_M
                dma();
#endif
            };
            int ph = 0;
            (void)ph;
            auto mid = [&]() {
                __builtin_amdgcn_sched_barrier(0);
                G8_STAMP(4 * ph + 1);          // load section done (reads and DMA issued, counted wait passed)
                __builtin_amdgcn_s_barrier();
                G8_STAMP(4 * ph + 2);          // past the barrier: MFMA section starts
            };
            auto end = [&]() {
                __builtin_amdgcn_sched_barrier(0);
                G8_STAMP(4 * ph + 3);          // MFMAs issued
                __builtin_amdgcn_s_barrier();
                ++ph;
                G8_STAMP(4 * ph);              // past the barrier: next load section starts
            };
            G8_STAMP(0);
            constexpr int SE = g8_epi_stores<EPI>();
#if G8_DMA_IN_M
            // slots (issued inside the MFMA cluster: a half may be restaged ONE phase after its last reading phase):
            //   M(P0): Ah1(t+1)   M(P1): Bh0(t+2)   M(P2): Ah0(t+2)   M(P3): Bh1(t+2)
            auto d0 = [&]() { if (e1) stage(o1 ? nxt : cur, 3, o1 ? kt + 1 - nk : kt + 1, BUF ^ 1); };
            auto d1 = [&]() { if (e2) stage(o2 ? nxt : cur, 0, o2 ? kt + 2 - nk : kt + 2, BUF); };
            auto d2 = [&]() { if (e2) stage(o2 ? nxt : cur, 1, o2 ? kt + 2 - nk : kt + 2, BUF); };
            auto d3 = [&]() { if (e2) stage(o2 ? nxt : cur, 2, o2 ? kt + 2 - nk : kt + 2, BUF); };
            constexpr int W1 = 8, W1E = (8 + SE > 63 ? 63 : 8 + SE), W3 = 6, W3E = (6 + SE > 63 ? 63 : 6 + SE);
#else
            //   L(P0): Ah0(t+1)   L(P1): Ah1(t+1)   L(P2): Bh0(t+2)   L(P3): Bh1(t+2)
            auto d0 = [&]() { if (e1) stage(o1 ? nxt : cur, 1, o1 ? kt + 1 - nk : kt + 1, BUF ^ 1); };
            auto d1 = [&]() { if (e1) stage(o1 ? nxt : cur, 3, o1 ? kt + 1 - nk : kt + 1, BUF ^ 1); };
            auto d2 = [&]() { if (e2) stage(o2 ? nxt : cur, 0, o2 ? kt + 2 - nk : kt + 2, BUF); };
            auto d3 = [&]() { if (e2) stage(o2 ? nxt : cur, 2, o2 ? kt + 2 - nk : kt + 2, BUF); };
            constexpr int W1 = 8, W1E = (8 + SE > 63 ? 63 : 8 + SE), W3 = 6, W3E = 6;
#endif
            // ---------------- P0: reads Bh0, Bh1 (all four column tiles) and rows 0-31 of Ah0
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) bfr[j][ks] = read_frag<LB>(bb + (2 + (j >> 1)) * G8_HALF, wc * 32 + 16 * (j & 1), ks, lane);
            read_a(0);
            lsec(d0);
            mid();
            mfma_q(0, d0);
            end();
            // ---------------- P1: rows 32-63 of Ah0; waits for Ah1(kt), which P2 reads
            read_a(1);
            lsec(d1);
            if (tail) G8_WAIT_VM(0);
            else if (after_epi) G8_WAIT_VM(W1E);
            else G8_WAIT_VM(W1);
            mid();
            mfma_q(1, d1);
            end();
            // ---------------- P2: rows 0-31 of Ah1
            read_a(2);
            lsec(d2);
            mid();
            mfma_q(2, d2);
            end();
            // ---------------- P3: rows 32-63 of Ah1; waits for Bh0, Bh1, Ah0 of K-tile kt+1
            read_a(3);
            lsec(d3);
            if (tail) G8_WAIT_VM(0);
            else if (after_epi) G8_WAIT_VM(W3E);
            else G8_WAIT_VM(W3);
            mid();
            mfma_q(3, d3);
            end();
        };
        for (int kt = 0; kt < nk; kt += 2) {
            ktile(std::integral_constant<int, 0>{}, kt);
            ktile(std::integral_constant<int, 1>{}, kt + 1);
        }
        if (wr == 0) __builtin_amdgcn_s_barrier();          // re-align the two groups for the epilogue
        __builtin_amdgcn_sched_barrier(0);

        const bool full_tile = (tc.m0 + G8_TM <= p.M) && (tc.n0 + G8_TN <= p.N);
        epilogue<EPI, 8, 8>(p, acc, ep, tc.m0 + wr * 128, tc.n0 + wc * 64, full_tile, lane, false);
        if (!has_next) break;
        first_tile = false;
        prev_full = full_tile;
        w = wnext;
        tc = tn;
        cur = nxt;
    }
}
