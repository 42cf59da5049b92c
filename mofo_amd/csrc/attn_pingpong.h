// Ping-pong forms of the streaming attention kernels (included into attention.hip's anonymous namespace; reuses its LDS image,
// fragment readers and store helpers).
//
// The 4-wave kernels run two UNRELATED blocks per CU: the two waves that share a SIMD drift through the same program and meet in
// their MFMA sections as often as not (profiles/r02_pmc_issue_attention_after.txt: dK/dV matrix pipe 39 % busy at 1.73 waves per
// SIMD, both MFMA groups of a tile stretched 2.4-3.3 x by the partner).  Here ONE block of 7-8 waves owns the CU and the two waves
// of a SIMD (wave w of group 0 = waves 0-3, wave w + 4 of group 1) run the SAME four-section tile program exactly one section
// apart, separated by raw s_barriers:
//
//        group 0:   R(t)   | M1(t) | V(t)   | M2(t)  | R(t+1) | ...
//        group 1:   M2(t-1)| R(t)  | M1(t)  | V(t)   | M2(t)  | ...
//
//   R  : the tile's row fragments LDS -> registers, and the accumulators' INITIAL values straight from LDS (below)
//   M1 : 8 MFMAs (S and dP)                      M2 : 8 MFMAs (dV and dK)
//   V  : the exp2 / dS arithmetic + the transposed fragments of M2
// so every section pairs one wave's matrix work with its partner's vector / LDS work (guide: MI355X_MICROARCH.md "Two waves per
// SIMD", cdna_hip_programming.md T16).
//
// Vector diet of the dK/dV pass (what the V section has to hide behind 8 MFMAs of the partner):
//   * dP - delta: the wave's V fragments are held NEGATED and the dP accumulator starts at +delta (read from the staged tile
//     straight into the accumulator registers): acc = delta - dO.V^T = -(dP - delta).  Exact; the sign is returned when dK is
//     stored (dK accumulates -dS^T Q).
//   * exp2(c S - lse2): with FOLD_S the wave's K fragments are held as bf16(-c K) and the S accumulator starts at +lse2:
//     acc = lse2 - c Q.K^T, p = exp2(-acc) (the negation is the instruction's source modifier).  Rounds c K instead of K to bf16:
//     same order as the bf16 rounding of P itself (tests: test_attention_fwd_bwd tolerances unchanged).
//   Left per 32 x 32 tile and lane: 16 v_exp, 16 v_mul, 16 pack conversions.

// hipcc's instruction selection orders an MFMA (no side effects) only by its data dependences: without these pins the MFMAs of a
// section drift across the s_barrier / s_setprio that delimit it (seen in the .s: 3 of 8 MFMAs left in their section).  An empty
// asm volatile that takes the accumulator as a read-write operand ties the chain to the barriers around it (the asm statements and
// the barriers are ordered among themselves); it emits no instruction.
#define PP_PIN1(a) asm volatile("" : "+v"(a))
#define PP_PIN2(a, b) asm volatile("" : "+v"(a), "+v"(b))
#define PP_PIN4(a, b, c_, d) asm volatile("" : "+v"(a), "+v"(b), "+v"(c_), "+v"(d))

template <int NW, bool FOLD_S>
__global__ __launch_bounds__(NW * 64, 2) void attn_dkv_pp_kernel(const bf16_t* __restrict__ qkv, int ldqkv, int nx, int G, int N, int H, float c,
                                                                  float scale, const bf16_t* __restrict__ dout, int lddo,
                                                                  const float* __restrict__ lse2, const float* __restrict__ delta,
                                                                  bf16_t* __restrict__ dqkv, int lddqkv) {
    static_assert(NW >= 5 && NW <= 8, "two groups: waves 0-3 and 4..NW-1");
    constexpr int BUF = 2 * TILE + 256;       // Q tile, dO tile, 32 f32 lse2 + 32 f32 delta
    __shared__ __attribute__((aligned(16))) unsigned char smem[4 * BUF];
    const int tid = threadIdx.x, lane = tid & 63, hh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2;
    int xb, b, h;
    if (!decode_block(nx, G, H, xb, b, h)) return;      // block-uniform
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
        u32x4 kraw = *(const u32x4*)(kp + (size_t)krow * ldqkv + 16 * ks + 8 * hh);
        u32x4 vraw = *(const u32x4*)(vp + (size_t)krow * ldqkv + 16 * ks + 8 * hh);
        vraw ^= u32x4{0x80008000u, 0x80008000u, 0x80008000u, 0x80008000u};            // -V (exact)
        vf[ks] = __builtin_bit_cast(bf16x8, vraw);
        if constexpr (FOLD_S) {
            u32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = pack_bf16x2(-c * bf16lo_to_f32(kraw[e]), -c * bf16hi_to_f32(kraw[e]));
            kf[ks] = __builtin_bit_cast(bf16x8, o);
        } else {
            kf[ks] = __builtin_bit_cast(bf16x8, kraw);
        }
    }
    f32x16 dk0 = zero16(), dk1 = zero16(), dv0 = zero16(), dv1 = zero16();

    const int nqt = (N + 31) >> 5;
    const int npair = (nqt + 1) >> 1;
    // LDS-DMA staging of one PAIR of query tiles (the 4-wave kernel's scheme): waves 0..3 each move one 8-row piece of the Q and dO
    // tiles of both tiles; wave 0 also the (lse2 | delta) rows; rows beyond the sequence read the pad values
    auto dma_pair = [&](int pr, int pb) {
        if (wave < 4) {
            const unsigned lds0 = (unsigned)(size_t)LDS_PTR(smem) + (unsigned)(2 * pb) * BUF;
            const int rl = wave * 8 + (lane >> 3);
            const int ch = (lane & 7) ^ swz(rl);
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                int r = (2 * pr + u) * 32 + rl;
                r = r < N ? r : N - 1;
                dma_b128(qp + (size_t)r * ldqkv + ch * 8, lds0 + u * BUF + wave * 1024);
                dma_b128(dop + (size_t)r * lddo + ch * 8, lds0 + u * BUF + TILE + wave * 1024);
            }
            if (wave == 0) {
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int qq = (2 * pr + u) * 32 + (lane & 31);
                    const float* src = lane < 32 ? (qq < N ? lp + qq : g_pad_row) : (qq < N ? dp_ + qq : g_pad_row + 1);
                    dma_b32(src, lds0 + u * BUF + 2 * TILE);
                }
            }
        }
    };
    auto bar = [&]() {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };

    dma_pair(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    bar();
    if (grp == 1) bar();                                   // group 1 runs one section behind group 0

    // one query tile = four sections; `first` / `last` (compile-time) say where in its pair the tile sits
    auto qtile = [&](int qt, auto buf_tag, bool issue_dma, int next_pr, bool drain) {
        constexpr int TB = decltype(buf_tag)::value;       // tile buffer 0..3
        const unsigned char* Qt = smem + TB * BUF;
        const unsigned char* Ot = Qt + TILE;
        const float* Lt = (const float*)(Qt + 2 * TILE);
        const float* Dt = Lt + 32;
        // ---------------- R: row fragments and the accumulators' initial values
        if (issue_dma) dma_pair(next_pr, (TB >> 1) ^ 1);   // the other pair buffer: every wave is past its last read of it
        bf16x8 qa[4], oa[4];
        f32x16 s, dpv;
        f32x4 lvs[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) qa[ks] = row_frag(Qt, ks, lane);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) oa[ks] = row_frag(Ot, ks, lane);
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
            const f32x4 lv = *(const f32x4*)(Lt + 8 * rg + 4 * hh);
            const f32x4 dv = *(const f32x4*)(Dt + 8 * rg + 4 * hh);
            lvs[rg] = lv;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                s[4 * rg + e] = FOLD_S ? lv[e] : 0.f;
                dpv[4 * rg + e] = dv[e];
            }
        }
        PP_PIN2(s, dpv);
        bar();
        // ---------------- M1: S and dP (alternating: no MFMA waits for its predecessor's result)
        PP_PIN2(s, dpv);
        ATTN_PRIO(1);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qa[ks], kf[ks], s, 0, 0, 0);
            dpv = __builtin_amdgcn_mfma_f32_32x32x16_bf16(oa[ks], vf[ks], dpv, 0, 0, 0);
        }
        ATTN_PRIO(0);
        PP_PIN2(s, dpv);
        bar();
        PP_PIN2(s, dpv);
        // ---------------- V: transposed fragments for M2 in flight, then p = exp2(.), -dS = p * (delta - dP)
        bf16x8 ot[4], qt4[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            ot[i] = tr_frag(Ot, i & 1, i >> 1, lane);
            qt4[i] = tr_frag(Qt, i & 1, i >> 1, lane);
        }
        float p[16], ds[16];
#pragma unroll
        for (int rg = 0; rg < 4; ++rg)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int r = 4 * rg + e;
                const float pr = FOLD_S ? fast_exp2(-s[r]) : fast_exp2(fmaf(s[r], c, -lvs[rg][e]));
                p[r] = pr;
                ds[r] = pr * dpv[r];
            }
        bf16x8 pf0 = pack_frag(p, 0), pf1 = pack_frag(p, 1);
        bf16x8 sf0 = pack_frag(ds, 0), sf1 = pack_frag(ds, 1);
        PP_PIN4(pf0, pf1, sf0, sf1);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this tile's last LDS reads have returned before the barrier:
        bar();                                               // the other group may restage the buffer behind it
        // ---------------- M2: dV += dO^T P, dK -= Q^T dS
        PP_PIN4(dk0, dk1, dv0, dv1);
        ATTN_PRIO(1);
        dv0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ot[0], pf0, dv0, 0, 0, 0);
        dv1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ot[2], pf0, dv1, 0, 0, 0);
        dk0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qt4[0], sf0, dk0, 0, 0, 0);
        dk1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qt4[2], sf0, dk1, 0, 0, 0);
        dv0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ot[1], pf1, dv0, 0, 0, 0);
        dv1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ot[3], pf1, dv1, 0, 0, 0);
        dk0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qt4[1], sf1, dk0, 0, 0, 0);
        dk1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qt4[3], sf1, dk1, 0, 0, 0);
        ATTN_PRIO(0);
        PP_PIN4(dk0, dk1, dv0, dv1);
        if (drain) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the staging waves' pieces of the next pair have landed
        bar();
    };
    auto pair = [&](int pr, auto pb_tag) {
        constexpr int PB = decltype(pb_tag)::value;
        const bool more = pr + 1 < npair;
        const bool two = 2 * pr + 1 < nqt;
        // The next pair goes into the other pair buffer.  It is issued by waves 0-3 (group 0) in the R section of this pair's FIRST
        // tile: group 0 is past the barrier that closed its M2 of the previous pair, and group 1 -- one section behind -- is in
        // its last M2 of the previous pair, whose LDS reads were retired (lgkmcnt(0)) before the barrier in front of it.
        qtile(2 * pr, std::integral_constant<int, 2 * PB>{}, more, pr + 1, more && !two);
        if (two) qtile(2 * pr + 1, std::integral_constant<int, 2 * PB + 1>{}, false, 0, more);
    };
    int pr = 0;
    for (; pr + 1 < npair; pr += 2) {
        pair(pr, std::integral_constant<int, 0>{});
        pair(pr + 1, std::integral_constant<int, 1>{});
    }
    if (pr < npair) pair(pr, std::integral_constant<int, 0>{});
    if (grp == 0) bar();                                   // the barrier count of the two groups is the same again
    if (ki >= N) return;
    bf16_t* drow = dqkv + ((size_t)b * N + ki) * lddqkv + h * HD;
    store_T(drow + D, dk0, dk1, -scale, hh);               // dK accumulated with the sign of (delta - dP)
    store_T(drow + 2 * D, dv0, dv1, 1.0f, hh);
}
