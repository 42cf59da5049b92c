// gemm_r4: the 384 x 128 bf16 MFMA weight-gradient GEMM (TN: both operands reduction-strided, f32 out) on a ring of FOUR 32-deep
// LDS stages with a counted vmcnt -- one barrier per 32-deep stage (included into gemm.hip's anonymous namespace after gemm_r3.h;
// reuses R3Group / ring_unit, the COL LDS image, lds_dma16, G8_WAIT_VM and epilogue<>).
//
// Why a tile of 384 rows (round-5 review, item 1).  Every width of the decoder (384, 1 152, 1 536) is a multiple of 384, none of
// 256: gemm_r3's 256-row tiles pad the decoder's weight gradients by 17 %, so its four blocks ran on the one-stage 128 x 128 form
// (0.29 of the matrix peak, 30 % MFMA-busy) while the encoder's ran on the ring (0.42, 49 %).  The encoder's widths (768, 2 304,
// 3 072) are multiples of 384 as well.  Against gemm_r3 per k-row of the reduction: 96 FLOP per staged byte (85), 10 fragment reads
// and 4 LDS-DMA pieces per 24 MFMAs of a wave (16 and 6 per 32).
//   * tile 384 x 128, 8 waves as 4 (M) x 2 (N), wave tile 96 x 64 = acc[6][4] of v_mfma_f32_16x16x32_bf16 (96 accumulator VGPRs);
//   * a stage is 32 reduction rows: A as three [32 k][128 col] COL images + B as one = 32 KiB; ring of four = 128 KiB.  A 64-deep
//     stage would be 64 KiB: two of them leave ONE stage in flight, three do not fit 160 KiB;
//   * the A fragments are SINGLE-buffered: the 4 MFMAs of row group i are the only readers of fa[i], so stage h + 1's fa[i] is read
//     into the same registers right behind them; the B fragments (read by every row group) are double-buffered.  56 fragment
//     VGPRs instead of 80 -- the kernel has to fit 256 (two waves per SIMD);
//   * fragment addresses are per-lane constants (ring laid out [image][buffer]: the buffer and the second read of a fragment are
//     16-bit immediates): no address arithmetic in the MFMA stream (gemm_r3 spends two v_add3 per fragment pair).
// Order inside step h (frags(h) in registers, stage h + 1 landed and visible, stages h + 2, h + 3 in flight, buffer h & 3 free):
//     24 MFMAs of stage h in six row groups; behind group i: the reads of stage h + 1's fa[i] (group 0: its four B fragments too);
//     behind groups 2..5: the wave's four LDS-DMA pieces of stage h + 4 into buffer h & 3 | lgkmcnt(0) | vmcnt(8) | barrier b_h.
//     RAW: a wave waits for its OWN pieces of stage h + 2 (8 younger ones keep flying), then the barrier, then anybody reads it (in
//          step h + 1).  WAR: buffer h & 3 was last read in step h - 1; every wave finished those reads (lgkmcnt(0)) before b_(h-1).
// WORK LIST: gemm_r3's (rounds of an XCD's run, tail chunks of R4_CH stages, or the SLICED form: R3Group).

constexpr int R4_TM = 384, R4_KH = 32;
constexpr int R4_IMG = R4_KH * 128 * 2;        // 8 KiB: one [32 k][128 cols] COL image
constexpr int R4_STG = 4 * R4_IMG;             // 32 KiB per stage: images A0, A1, A2, B
#ifndef R4_NB
#define R4_NB 4                                // ring buffers (4: 128 KiB, three stages in flight; 5: 160 KiB, four)
#endif
constexpr int R4_RING = R4_NB * R4_STG;
constexpr int R4_ISTR = R4_NB * R4_IMG;        // the ring is laid out [image][buffer]: the buffers of ONE image are 8 KiB apart, so a
                                               // fragment has ONE address register and the buffer is a 16-bit immediate (<= 4 x 8 KiB + 1 KiB)
constexpr int R4_CH = 8;                       // stages per chunk of the tail split (256 reduction rows, like gemm_r3's)
#ifndef R4_PRIO
#define R4_PRIO 1
#endif
// placement of the four LDS-DMA pieces behind the row groups: 0 = groups 2, 3, 4, 5; 1 = groups 0, 1, 2, 3 for waves 4-7 (the SIMD
// partners of waves 0-3 issue their pieces while the others multiply): 0.95 x, profiles/r06_gemm_r4_ablate.txt
#ifndef R4_DMA_SKEW
#define R4_DMA_SKEW 0
#endif
// TRIED AND NOT KEPT (profiles/r06_gemm_r4_ablate.txt): fa[4], fa[5] double-buffered too, so that the last fragment read of a step is issued
// behind row group 3 and the lgkmcnt(0) in front of the barrier finds it done -- 0.97-1.00 x and 58 spilled registers; a fifth ring
// buffer (160 KiB, four stages in flight) -- 1.00 x: the L2 -> LDS path delivers ~21-23 B/clk per CU whatever is in flight.
// timing-only ablation builds (wrong results; profiles/r06_gemm_r4_ablate.txt): drop the LDS-DMA pieces / the fragment reads / the MFMAs /
// the per-step barrier from the main loop
#ifndef R4_NO_DMA
#define R4_NO_DMA 0
#endif
#ifndef R4_NO_READ
#define R4_NO_READ 0
#endif
#ifndef R4_NO_MFMA
#define R4_NO_MFMA 0
#endif
#ifndef R4_NO_BARRIER
#define R4_NO_BARRIER 0
#endif

template <class F, int... Is>
__device__ __forceinline__ void r4_static_for(F&& f, std::integer_sequence<int, Is...>) {
    (f(std::integral_constant<int, Is>{}), ...);
}

template <int EPI>
__global__ __launch_bounds__(512, 2) void gemm_r4_kernel(R3Group G, int total) {
    static_assert(EPI == MOFO_EPI_F32, "built for the weight gradients (TN, f32 out)");
    __shared__ __attribute__((aligned(16))) unsigned char smem[R4_RING];
    typedef s16x4 __attribute__((address_space(3)))* LdsFrag;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;

    // ---- this block's share of the work list (gemm_r3.h: label x = blockIdx & 7 owns one contiguous run, walked round by round; the
    // units past the last full round are dealt in chunks when G.tail)
    const int nbx = (int)gridDim.x >> 3, jx = (int)blockIdx.x >> 3, xcd = (int)blockIdx.x & 7;
    const int q = total >> 3, r = total & 7;
    const int xbeg = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    const int xlen = q + (xcd < r ? 1 : 0);
    const int rounds = G.tail ? xlen / nbx : (xlen + nbx - 1) / nbx;
    const int tail0 = rounds * nbx;
    int clo = 0, chi = 0;
    if (G.tail && tail0 < xlen) {
        int ctot = 0;
        for (int u = tail0; u < xlen; ++u) ctot += (ring_unit<R4_TM, R4_KH>(G, xbeg + u).nk + R4_CH - 1) / R4_CH;
        clo = (int)(((long long)ctot * jx) / nbx);
        chi = (int)(((long long)ctot * (jx + 1)) / nbx);
    }
    int rd = 0, ut = tail0, c0 = 0;
    auto next_seg = [&](R3Seg& s) -> bool {
        while (rd < rounds) {
            const int u = rd * nbx + jx;
            ++rd;
            if (u >= xlen) continue;
            s = ring_unit<R4_TM, R4_KH>(G, xbeg + u);
            return true;
        }
        while (clo < chi && ut < xlen && c0 < chi) {
            s = ring_unit<R4_TM, R4_KH>(G, xbeg + ut);
            ++ut;
            const int nch = (s.nk + R4_CH - 1) / R4_CH;
            const int lo = max(clo, c0) - c0, hi = min(chi, c0 + nch) - c0;
            c0 += nch;
            if (lo >= hi) continue;
            const int ks0 = lo * R4_CH, ks1 = min(s.nk, hi * R4_CH);
            if (ks0 > 0 || ks1 < s.nk) s.atomic = 1;   // the unit is shared: summed with f32 atomics onto a zeroed destination
            s.k0 += ks0 * R4_KH;
            s.nk = ks1 - ks0;
            return true;
        }
        return false;
    };

    // ---- per-lane constants
    // LDS-DMA source offset: a wave stages piece `wave` (k-rows 4 wave .. 4 wave + 3) of each of the stage's four images; lane ->
    // k-row kq of the piece, 16-B chunk position cpos; the 32-B unit at position u of k-row r holds logical unit u ^ col_key(r)
    auto lane_off = [&](int ld) -> int {
        const int cpos = lane & 15, kq = lane >> 4;
        const int key = kq | (((wave >> 1) & 1) << 2);
        const int g = ((((cpos >> 1) ^ key) << 1) | (cpos & 1));
        return (kq * ld + g * 8) * 2;
    };
    // fragment read addresses (LDS byte address of the FIRST ds_read_b64_tr_b16 of a fragment in ring buffer 0; the second read is
    // 4 k-rows = 1 024 B further, buffer b is b * 32 KiB further): read_frag<OPL_COL> of gemm.hip with ks = 0, resolved once
    const unsigned lds0 = (unsigned)(unsigned long long)LDS_PTR(smem);
    unsigned aaddr[6], baddr[4];
    {
        const int g = lane >> 4, qq = (lane & 15) >> 2, pp = lane & 3;
        const int kr0 = 8 * g + qq;
        const unsigned rowpart = lds0 + kr0 * 256 + 8 * pp;
        const int key = col_key(kr0);
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int row = wm * 96 + 16 * i;
            aaddr[i] = rowpart + (row >> 7) * R4_ISTR + ((((row & 127) >> 4) ^ key) << 5);
            asm volatile("" : "+v"(aaddr[i]));     // opaque: kept in a register, not re-derived per read
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            baddr[j] = rowpart + 3 * R4_ISTR + ((((wn * 64 + 16 * j) >> 4) ^ key) << 5);
            asm volatile("" : "+v"(baddr[j]));
        }
    }
    auto rd_frag = [&](unsigned base, int imm) -> bf16x8 {
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LdsFrag)(uintptr_t)(base + imm));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LdsFrag)(uintptr_t)(base + imm + 1024));
        const u32x2 l2 = __builtin_bit_cast(u32x2, lo), h2 = __builtin_bit_cast(u32x2, hi);
        const u32x4 w = {l2[0], l2[1], h2[0], h2[1]};
        return __builtin_bit_cast(bf16x8, w);
    };
    bf16x8 ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = (__bf16)1.0f;

    struct Ctx {
        __amdgpu_buffer_rsrc_t ra, rb;
        int lda, ldb, va, vb, m0, n0, k0;
    };
    auto make_ctx = [&](const R3Seg& sg) -> Ctx {
        const R3Prob& p = G.p[sg.gi];
        Ctx c;
        const size_t ext_a = (((size_t)sg.kend - 1) * p.lda + p.M) * 2;
        const size_t ext_b = (((size_t)sg.kend - 1) * p.ldb + p.N) * 2;
        c.ra = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, (int)ext_a, 0x00020000);
        c.rb = __builtin_amdgcn_make_buffer_rsrc((void*)p.B, 0, (int)ext_b, 0x00020000);
        c.lda = p.lda;
        c.ldb = p.ldb;
        c.va = lane_off(p.lda);
        c.vb = lane_off(p.ldb);
        c.m0 = sg.m0;
        c.n0 = sg.n0;
        c.k0 = sg.k0;
        return c;
    };
    // piece IDX (0-2: A sub-image IDX, 3: B) of stage t of segment c into ring buffer `buf`
    auto piece = [&](auto idx_tag, const Ctx& c, int t, int buf) {
        constexpr int IDX = decltype(idx_tag)::value;
#if R4_NO_DMA
        if (t >= R4_NB) return;
#endif
        unsigned char* dst = smem + IDX * R4_ISTR + buf * R4_IMG + wave * 1024;
        const unsigned kr = (unsigned)(c.k0 + t * R4_KH + 4 * wave);
        if constexpr (IDX < 3) lds_dma16<true>(c.ra, dst, c.va, (kr * (unsigned)c.lda + (unsigned)(c.m0 + 128 * IDX)) * 2u);
        else lds_dma16<true>(c.rb, dst, c.vb, (kr * (unsigned)c.ldb + (unsigned)c.n0) * 2u);
    };
    auto stage_all = [&](const Ctx& c, int t, int buf) {
        piece(std::integral_constant<int, 0>{}, c, t, buf);
        piece(std::integral_constant<int, 1>{}, c, t, buf);
        piece(std::integral_constant<int, 2>{}, c, t, buf);
        piece(std::integral_constant<int, 3>{}, c, t, buf);
    };

    bf16x8 fa[6], fb[2][4];
    f32x4 acc[6][4], accb[6];
    constexpr bool CAN_COLSUM = true;

    R3Seg cs;
    while (next_seg(cs)) {
        const Ctx cur = make_ctx(cs);
        const int nk = cs.nk;
        const R3Prob& p = G.p[cs.gi];
        const bool do_colsum = CAN_COLSUM && p.colsum != nullptr && cs.n0 == 0 && wn == 0;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            accb[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        // ---- prologue: stages 0 .. NB-1 in flight; stages 0 and 1 landed; frags(0) in registers in EVERY wave (then buffer 0 is free)
#pragma unroll
        for (int b = 0; b < R4_NB; ++b) stage_all(cur, b, b);
        G8_WAIT_VM(4 * (R4_NB - 2));
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 4; ++j) fb[0][j] = rd_frag(baddr[j], 0);
#pragma unroll
        for (int i = 0; i < 6; ++i) fa[i] = rd_frag(aaddr[i], 0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);

        // one step: multiplies stage h (ring buffer H % NB) from registers, reads stage h + 1's fragments, issues stage h + NB into
        // the buffer of stage h
        auto step = [&](auto h_tag, auto cs_tag, int h) {
            constexpr int H = decltype(h_tag)::value;
            constexpr bool CS = decltype(cs_tag)::value;
            constexpr int B = H % R4_NB, BN = (H + 1) % R4_NB;     // buffers of stages h, h + 1
            constexpr int P = H & 1, PN = P ^ 1;                    // double-buffered fragment set multiplied / read
            constexpr int IMM = BN * R4_IMG;
#if R4_PRIO
            __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                const bf16x8 a_i = fa[i];
#pragma unroll
#if R4_NO_MFMA
                for (int j = 0; j < 4; ++j) asm volatile("" ::"v"(fb[P][j]), "v"(a_i));
#else
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[P][j], a_i, acc[i][j], 0, 0, 0);
                if constexpr (CS) accb[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, a_i, accb[i], 0, 0, 0);
#endif
                __builtin_amdgcn_sched_barrier(0);
#if !R4_NO_READ
                if (i == 0) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) fb[PN][j] = rd_frag(baddr[j], IMM);
                }
                fa[i] = rd_frag(aaddr[i], IMM);
#else
                if (i == 0) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) fb[PN][j] = fb[P][j];
                }
#endif
#if R4_DMA_SKEW
                const int slot = wave >= 4 ? i : i - 2;
#else
                const int slot = i - 2;
#endif
                if (slot == 0) piece(std::integral_constant<int, 0>{}, cur, h + R4_NB, B);
                else if (slot == 1) piece(std::integral_constant<int, 1>{}, cur, h + R4_NB, B);
                else if (slot == 2) piece(std::integral_constant<int, 2>{}, cur, h + R4_NB, B);
                else if (slot == 3) piece(std::integral_constant<int, 3>{}, cur, h + R4_NB, B);
                __builtin_amdgcn_sched_barrier(0);
            }
#if R4_PRIO
            __builtin_amdgcn_s_setprio(0);
#endif
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // stage h + 1 is in registers: this wave is done with its buffer
            G8_WAIT_VM(4 * (R4_NB - 2));                            // stage h + 2 landed (this wave's pieces); the younger ones keep flying
            __builtin_amdgcn_sched_barrier(0);
#if !R4_NO_BARRIER
            __builtin_amdgcn_s_barrier();                           // b_h
#endif
            __builtin_amdgcn_sched_barrier(0);
        };
        // the buffer index has period NB, the fragment parity period 2: the loop body is PERIOD steps with both at compile time
        constexpr int PERIOD = (R4_NB & 1) ? 2 * R4_NB : R4_NB;
        auto body = [&](auto cs_tag, int h) {
            r4_static_for([&](auto it) { step(it, cs_tag, h + decltype(it)::value); }, std::make_integer_sequence<int, PERIOD>{});
        };
        auto rest = [&](auto cs_tag, int h) {
            r4_static_for([&](auto it) { if (h + decltype(it)::value < nk) step(it, cs_tag, h + decltype(it)::value); },
                          std::make_integer_sequence<int, PERIOD - 1>{});
        };
        auto loops = [&](auto cs_tag) {
            int h = 0;
            for (; h + PERIOD <= nk; h += PERIOD) body(cs_tag, h);
            rest(cs_tag, h);
        };
        if (do_colsum) loops(std::true_type{});
        else loops(std::false_type{});
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        G8_WAIT_VM(0);                             // the stages issued past the end have landed (or were dropped by the range check)
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();              // every wave is done with the ring: it becomes the epilogue's staging area
        __builtin_amdgcn_sched_barrier(0);
        // (an opaque copy of the lane id: what the epilogue derives from it is computed HERE, per segment, instead of being hoisted in
        // front of the work loop and carried across the main loop in 39 spilled registers)
        int lane_e = lane;
        asm volatile("" : "+v"(lane_e));
        if (do_colsum && lane_e < 16) {            // D[n][m]: every row n holds the same sum; lanes 0..15 hold m = 16 i + lane in element 0
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                const int m = cs.m0 + wm * 96 + 16 * i + lane_e;
                if (m < p.M && !(m >= p.skip_lo && m < p.skip_hi)) atomicAdd(p.colsum + m, accb[i][0]);
            }
        }
        const bool full = (cs.m0 + R4_TM <= p.M) && (cs.n0 + R3_TN <= p.N);
        {
            GemmP pe = {};
            pe.C = (float*)p.C + (long long)cs.slice * G.slab_stride;
            pe.M = p.M;
            pe.N = p.N;
            pe.ldc = p.ldc;
            pe.atomic = cs.atomic;
            // the wave's 96 x 64 f32 tile through 48 staged rows at a time (12 KiB per wave of the drained ring)
            epilogue<EPI, 6, 2>(pe, acc, (float*)smem + wave * (48 * 64), cs.m0 + wm * 96, cs.n0 + wn * 64, full, lane_e, false);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();          // the staging area becomes the next segment's ring
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// Sum of the slices' partial weight gradients (SLICED launches of gemm_r3 / gemm_r4): C[m, n] (+)= sum_s slab[s][off + m N + n].
// One float4 per thread, every slice's load in flight before the first add; HBM-bound (slices + 1 floats moved per element).
struct SlabReduceP {
    float* C[MAXR];
    long long off[MAXR];       // f32 element offset of the problem inside a slab
    int M[MAXR], N[MAXR], ldc[MAXR];
    int start[MAXR + 1];       // first block of each problem (256 threads x one float4)
    int count, slices, accumulate;
    long long slab_stride;
    const float* ws;
};
template <int S>
__global__ __launch_bounds__(256) void wgrad_slab_reduce_kernel(SlabReduceP P) {
    int gi = 0;
#pragma nounroll
    for (int k = 1; k < P.count; ++k)
        if ((int)blockIdx.x >= P.start[k]) gi = k;
    const int N = P.N[gi];
    const long long e = ((long long)((int)blockIdx.x - P.start[gi]) * 256 + threadIdx.x) * 4;
    if (e >= (long long)P.M[gi] * N) return;
    const float* src = P.ws + P.off[gi] + e;
    f32x4 v[S];
#pragma unroll
    for (int s = 0; s < S; ++s) v[s] = __builtin_nontemporal_load((const f32x4*)(src + (long long)s * P.slab_stride));
    f32x4 sum = v[0];
#pragma unroll
    for (int s = 1; s < S; ++s) sum += v[s];
    const int m = (int)(e / N), n = (int)(e - (long long)m * N);
    f32x4* dst = (f32x4*)(P.C[gi] + (long long)m * P.ldc[gi] + n);
    if (P.accumulate) sum += *dst;
    *dst = sum;
}
