// gemm_k2: ONE 128 x 128 tile per CU, eight waves = two 4-wave groups that each walk ONE HALF of the reduction through their own
// two-stage LDS ring with a COUNTED vmcnt (included into gemm.hip's anonymous namespace after gemm8.h; reuses GemmP, the two LDS
// images, stage_tile_srd / read_frag, G8_WAIT_VM and every epilogue).
//
// Which problem it is for (tools/gemm_geometry.py, profiles/r04_gemm_geometry.txt): the encoder's N = 768 GEMMs at 5 120 token rows
// (fc2 forward, fc1 / qkv dgrad: K = 3072 / 2304; proj and its dgrad: K = 768) give 240 tiles of 128 x 128 -- less than one per CU.
// The forms that carried them (64 x 128 tiles, one k-stage in flight, `vmcnt(0)` + barrier per k-step, two blocks per CU) stream
// 1.5x the bytes through each CU's L1 -> LDS path (25.3 us floor at K = 3072 against 16.9 for a 128 x 128 tile) and pay a full
// L2 round trip per 64-deep step.  With one tile per CU there is nothing to co-schedule, so the block spends the CU's LDS on DEPTH
// instead of on co-resident blocks:
//   * 128 KiB = 2 groups x 2 stages x (A 16 KiB + B 16 KiB); a group reads ALL fragments of stage t into registers (the VAR 1 idea),
//     which frees that stage's buffer after ONE barrier, issues stage t + 2 into it and multiplies from registers: stages t + 1 and
//     t + 2 are in flight while stage t is multiplied, and the wait before the next step is `vmcnt(8)` -- stage t + 1 landed, the
//     eight pieces of t + 2 still flying (never vmcnt(0) inside the stream);
//   * the two groups run HALF A STEP apart (STAG: group 1 starts one barrier late, group 0 ends one barrier late): one group's
//     LDS-DMA issue + 32 MFMAs beside the other's fragment reads on every SIMD (two waves per SIMD, one of each group);
//   * the halves meet in LDS as in the 64-row split-K kernel: a group hands over the 32 rows of every wave tile it does not own,
//     adds the 32 it receives and runs the shared epilogue on its own rows -- the epilogue is spread over all eight waves.
// A group only ever reads buffers its OWN waves staged, so the RAW rule is the plain one: every wave waits for its own pieces
// (counted), then a barrier, then the reads; WAR: a buffer is restaged after the barrier that follows the `lgkmcnt(0)` of its reads.
// The barriers are workgroup-wide (the other group takes part, half a step off): both groups execute the same count.

template <int LA, int LB, int EPI, bool STAG>
__global__ __launch_bounds__(512, 2) void gemm_k2_kernel(GemmP p, int total) {
    static_assert(LA == OPL_ROW, "built for the ROW A operand (NT / NN)");
    constexpr int MI = 4;
    constexpr int BMT = 128;
    constexpr int A_BYTES = BMT * 64 * 2;
    constexpr int STG = A_BYTES + TILE_BYTES;                    // 32 KiB per stage
    constexpr int RING = 2 * STG;                                // per group
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * RING];
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
    unsigned char* gs = smem + grp_u * RING;
    // rotated reduction order (see gemm_persistent_kernel): 0 none, 1 by m-tile + n-tile, 2 by m-tile, 3 by n-tile
    const int krot = (p.rotate_tile == 1 ? (m0 / BMT + n0 / BN) : p.rotate_tile == 2 ? m0 / BMT : p.rotate_tile == 3 ? n0 / BN : 0) % nkh;
    auto kof = [&](int t) {
        int kc = t + krot;
        kc = kc >= nkh ? kc - nkh : kc;
        return kb + kc * BK;
    };
    // NT: both operands are k-contiguous rows -- a k offset past K inside the LAST row would be in range, so an empty trailing
    // k-stage is skipped explicitly (nk odd); group-uniform.  Returns whether the eight pieces were issued (the vmcnt budget).
    auto stage = [&](int t, int buf) -> bool {
        const int k0 = kof(t);
        if (k0 >= p.K) return false;
        unsigned char* dst = gs + buf * STG;
        stage_tile_srd<LA, MI, 0, MOFO_DMA_ASM_K2 != 0>(ra, va0, va1, p.lda, m0, k0, dst, w4_u);    // builtin pieces: see lds_dma16 in gemm.hip
        stage_tile_srd<LB, 4, 0, MOFO_DMA_ASM_K2 != 0>(rb, vb0, vb1, p.ldb, n0, k0, dst + A_BYTES, w4_u);
        return true;
    };
    f32x4 acc[MI][4];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    stage(0, 0);
    const bool s1 = nkh > 1 && stage(1, 1);
    if (s1) G8_WAIT_VM(8);
    else G8_WAIT_VM(0);
    __builtin_amdgcn_s_barrier();                                // stage 0 landed and visible to the group
    if (STAG && grp_u == 1) __builtin_amdgcn_s_barrier();        // stagger: group 1 runs half a step behind group 0
    __builtin_amdgcn_sched_barrier(0);

    for (int t = 0; t < nkh; ++t) {
        const unsigned char* ta = gs + (t & 1) * STG;
        const unsigned char* tb = ta + A_BYTES;
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
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();                            // every wave of the group holds stage t in registers: its buffer is free
        __builtin_amdgcn_sched_barrier(0);
        const bool s2 = (t + 2 < nkh) && stage(t + 2, t & 1);    // flies for two steps
        if (live) {
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[ks][j], af[ks][i], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (s2) G8_WAIT_VM(8);                                   // stage t + 1 landed (this wave's pieces); stage t + 2 keeps flying
        else G8_WAIT_VM(0);
        __builtin_amdgcn_s_barrier();                            // ... and is visible to every wave of the group
        __builtin_amdgcn_sched_barrier(0);
    }
    if (STAG && grp_u == 0) __builtin_amdgcn_s_barrier();        // re-align the groups
    __syncthreads();
    // exchange: group g keeps row tiles i = 2g, 2g + 1 of every wave tile and gives the other two away (8 f32x4 per lane -> 32 KiB per group)
    {
        f32x4* mine = (f32x4*)gs;
        const f32x4* theirs = (const f32x4*)(smem + (1 - grp_u) * RING);
        const int slot = w4 * 64 + lane;
#pragma unroll
        for (int ii = 0; ii < 2; ++ii)
#pragma unroll
            for (int j = 0; j < 4; ++j) mine[(ii * 4 + j) * 256 + slot] = grp ? acc[ii][j] : acc[2 + ii][j];
        __syncthreads();
#pragma unroll
        for (int ii = 0; ii < 2; ++ii)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x4 o = theirs[(ii * 4 + j) * 256 + slot];
                if (grp) acc[2 + ii][j] += o;
                else acc[ii][j] += o;
            }
        __syncthreads();                                         // the exchange area becomes the epilogue staging area
    }
    f32x4 own[2][4];
#pragma unroll
    for (int ii = 0; ii < 2; ++ii)
#pragma unroll
        for (int j = 0; j < 4; ++j) own[ii][j] = grp ? acc[2 + ii][j] : acc[ii][j];
    epilogue<EPI, 2, 1>(p, own, (float*)smem + wave * (32 * 64), m0 + wm * 64 + grp * 32, n0 + wn * 64,
                        (m0 + BMT <= p.M) && (n0 + BN <= p.N), lane, false);
}
