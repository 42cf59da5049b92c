/* libmofo_hip.so -- C-ABI of the MI355X-native MOFO / VideoMAE pretraining hot path.
 *
 * The reference (Moohnai/MOFO) has NO FFI / plugin layer for this path: it is plain Python on torch ops
 * (SURVEY.md 8b).  This header therefore DEFINES the boundary a maintainer would bind (ctypes stub in
 * INTEGRATION.md); each entry cites the reference arithmetic (file:line under /root/reference) it replaces.
 *
 * Contract for every entry:
 *   - plain device pointers + explicit sizes / leading dimensions (in ELEMENTS) + a hipStream_t passed as void*;
 *   - returns 0 (MOFO_OK) or a negative MOFO_E* code; never throws; mofo_last_error() has the text;
 *   - never allocates, frees or synchronises; kernels are enqueued on the given stream;
 *   - bf16 tensors are uint16_t bit patterns; "f32" = float; index tensors are int32.
 * Built for gfx950 only.
 */
#ifndef MOFO_HIP_H
#define MOFO_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* 2 (round 5): mofo_attention_delta_zero_dq and mofo_attention_bwd_onepass are gone (round 4), mofo_gemm_grouped_plan is new and
 * mofo_gemm_grouped takes up to 32 weight-gradient problems; INTEGRATION.md lists every change of the exported set. */
/* 3 (round 5): mofo_gemm_args grew (C8, ldc8, q_scale, q_amax: the e4m3 copy of fc1's activation); MOFO_GEMM_NT_FP8 takes the
 * residual epilogues; new entries mofo_attention_fwd_q8, mofo_adamw_q8. */
/* 4 (round 6): new entries mofo_gemm_wgrad_sliced / mofo_gemm_wgrad_sliced_ws (the weight gradients of a whole pass in ONE launch, the token
 * reduction sliced over the XCDs, no atomics); mofo_gemm_route_counts reports a ninth family (the 384 x 128 ring kernel). */
#define MOFO_ABI_VERSION 4

/* ---- library ---- */
int mofo_version(void);
const char* mofo_last_error(void);

/* ---- GEMM family: modeling_finetune.py:45-49 (Mlp fc1/fc2), :84 (qkv F.linear), :96 (proj),
 *      modeling_pretrain.py:124,157 (decoder head), :228,256 (encoder_to_decoder), and the Conv3d of
 *      modeling_finetune.py:238-247 as a dot product per tubelet; plus their autograd dgrad / wgrad. ---- */
enum { MOFO_GEMM_NT = 0, /* C[m,n] = sum_k A[m,k] B[n,k]  forward  */
       MOFO_GEMM_NN = 1, /* C[m,n] = sum_k A[m,k] B[k,n]  dgrad    */
       MOFO_GEMM_TN = 2, /* C[m,n] = sum_k A[k,m] B[k,n]  wgrad    */
       MOFO_GEMM_NT_FP8 = 3 /* NT on e4m3 operands (see below)       */ };
/* MOFO_GEMM_NT_FP8 (op 3): the NT form on OCP e4m3 operands (1 byte per element, lda / ldb in bytes, K a multiple of 128),
 * block-scaled MFMA (v_mfma_scale_f32_16x16x128_f8f6f4, unit block scales: 2x the bf16 MFMA rate), f32 accumulate; the
 * per-tensor de-quantisation factors are read on the device.  Epilogues BF16, BIAS_GELU (optionally with an e4m3 copy of the
 * activation, C8), RESID_F32, RESID_BF16: the four forward Linears of a transformer block.  BASELINE configs[4]'s fp8 forward. */
enum { MOFO_EPI_BF16 = 0,       /* C(bf16) = acc (+bias[n])                                            */
       MOFO_EPI_BIAS_GELU = 1,  /* h = acc+bias; C(bf16)=h; C2(bf16)=gelu_erf(h)   (fc1 + nn.GELU)      */
       MOFO_EPI_RESID_F32 = 2,  /* C(f32) = resid(f32) + acc (+bias)   (proj / fc2 + residual add)      */
       MOFO_EPI_POS_F32 = 3,    /* C(f32)[rowmap(m)] = acc (+bias) + pos[row_idx[m]]  (patch-embed / e2d)*/
       MOFO_EPI_DGELU_BF16 = 4, /* C(bf16) = acc * gelu_erf'(aux[m,n])   (backward through nn.GELU)     */
       MOFO_EPI_F32 = 5,        /* C(f32) = acc, or += (atomic) when splits>1 or accumulate!=0          */
       MOFO_EPI_RESID_BF16 = 6, /* C(bf16) = aux(bf16)[m,n] + acc (+bias): residual add on a bf16 residual stream (decoder) */
       MOFO_EPI_POS_BF16 = 7    /* POS_F32 with a bf16 destination (the decoder's bf16 input stream)    */ };
typedef struct mofo_gemm_args {
    int op, epilogue;
    int M, N, K;                  /* C is MxN, reduction length K */
    const void* A; int lda;       /* bf16 */
    const void* B; int ldb;       /* bf16 */
    void* C; int ldc;
    void* C2; int ldc2;           /* BIAS_GELU second output */
    const float* bias;            /* [N] or NULL */
    const float* resid; int ldr;  /* RESID_F32 */
    const void* aux; int ldaux;   /* DGELU: pre-activation h (bf16); RESID_BF16: the residual (bf16) */
    const float* pos; int ldpos;  /* POS_F32: table rows */
    const int* row_idx;           /* POS_F32: [M] table row per output row */
    int rows_in, rows_out, row_off; /* POS_F32 / POS_BF16: out row = (m / rows_in) * rows_out + row_off + m % rows_in.
                                     * RESID_F32 / RESID_BF16 with rows_in > 0 (M % rows_in == 0): the RESIDUAL operand's row of
                                     * output row m is that expression (C itself stays dense) -- the last decoder block works on the masked
                                     * tokens only while its residual input is the whole-sequence stream (modeling_pretrain.py:157) */
    int splits;                   /* split the reduction over gridDim.z (F32 epilogue only) */
    int accumulate;               /* F32 epilogue: add into C instead of overwrite */
    float* colsum;                /* TN + F32 only (wgrad): colsum[m] += sum_k A[k,m] = the bias gradient, fused; or NULL */
    int colsum_skip_lo, colsum_skip_hi; /* rows m in [lo,hi) of colsum are left untouched (k third of the fused qkv bias) */
    const float* a_scale_inv;     /* NT_FP8: device scalars; C = epilogue(a_scale_inv[0] * b_scale_inv[0] * sum_k A8 B8) */
    const float* b_scale_inv;
    void* C8; int ldc8;           /* NT_FP8 + BIAS_GELU (ABI 3): e4m3 copy of the activation C2, sat(gelu * q_scale[0]), the A operand of an */
    const float* q_scale;         /* fp8 fc2; q_scale = device scalar (delayed scaling), q_amax = MOFO_FP8_AMAX_STRIPES floats that        */
    float* q_amax;                /* collect max|gelu| of this launch (folded by mofo_fp8_update_scales); C8 = NULL: no copy              */
} mofo_gemm_args;
int mofo_gemm(const mofo_gemm_args* args, void* stream);
/* up to 13 problems (32 for TN + F32 weight-gradient groups) of ONE (op, epilogue) kind in one launch (e.g. the four weight-gradient GEMMs of three transformer blocks
 * and the patch embed's) */
int mofo_gemm_grouped(const mofo_gemm_args* args, int count, void* stream);
/* Weight-gradient groups (every problem TN + F32; up to 32 of them) may be routed to the 256 x 128 ring kernel, which deals the last,
 * partial round of tiles to ALL compute units in chunks of the reduction: a tile shared by several workgroups is summed with f32
 * atomics, so its destination must hold zeros (or the running sum) before the launch.  This host-only query tells the caller which:
 * shared[i] = 1 when problem i's C may receive atomic adds (always so for splits > 1 / accumulate), 0 when every element of C is
 * plainly stored exactly once.  Returns 1 when the group goes to the ring kernel, 0 otherwise, < 0 on error.  Same routing
 * decision as mofo_gemm_grouped with the same arguments and environment. */
int mofo_gemm_grouped_plan(const mofo_gemm_args* args, int count, int* shared);
/* SLICED weight-gradient group: the autograd of the Linears of modeling_finetune.py:44-51,84,96 at the decoder widths of
 * modeling_pretrain.py:292-314, where the reduction runs over all B * 1568 token rows.  Up to 32 problems, every one TN + F32
 * (C[m,n] (+)= sum_k A[k,m] B[k,n]; colsum as in mofo_gemm_grouped; bias = NULL; splits <= 1; all problems agree on accumulate).
 * The reduction of EVERY problem is cut into `slices` (1..8) contiguous row ranges; slice s of all problems is computed by the
 * workgroups of one XCD label (blockIdx % 8 == s for 8 slices), which therefore streams one row range of the operands through its
 * L2: every operand byte is fetched by one XCD only.  The slices' partial products are plainly stored to the caller's workspace
 * `ws` (mofo_gemm_wgrad_sliced_ws(...) floats, 16-byte aligned; contents undefined afterwards) and a second kernel of the same call
 * sums them into C (overwrite, or += when accumulate != 0): NO destination needs zeroing and no float atomics touch C, so results
 * are bit-identical from launch to launch.  Tile: 384 x 128 (gemm_r4) when every M is a multiple of 384 and every N of 128, else
 * 256 x 128 (ragged tiles allowed).  Slice s of a problem is rows [s k, (s + 1) k) with k = ceil(K / slices) rounded up to 32.
 * mofo_gemm_wgrad_sliced_ws returns the workspace size in floats (< 0: error code). */
long long mofo_gemm_wgrad_sliced_ws(const mofo_gemm_args* args, int count, int slices);
int mofo_gemm_wgrad_sliced(const mofo_gemm_args* args, int count, int slices, float* ws, long long ws_floats, void* stream);
/* Diagnostics (host side, no device work): launches per main-loop family since the last reset -- out[0] one tile per block,
 * [1] persistent 64/128-row tiles, [2] persistent 256-row tiles, [3] in-block split-K, [4] the 256 x 256 counted-vmcnt kernel,
 * [5] e4m3, [6] one 128 x 128 tile per CU with the reduction halved over two wave groups, [7] the 256 x 128 three-stage ring kernel, [8] the 384 x 128 four-stage ring kernel.  Lets a parity test assert that the shape-routed form it is meant to cover really ran. */
int mofo_gemm_route_counts(long long* out, int n, int reset);

/* ---- column sums: bias gradients (autograd of the `+ bias` in the Linears above). out[n] (+)= sum_m X[m,n] ---- */
int mofo_colsum_bf16(const void* X, int ldx, int M, int N, float* out, void* stream);   /* out must be zeroed */

/* ---- LayerNorm (eps inside sqrt, biased variance): modeling_finetune.py:200,206,218-219; modeling_pretrain.py:51,95,123,157.
 * Row r of the (M x D) problem reads/writes x row  (r / rows_in) * rows_out + row_off + r % rows_in  (pass rows_in=M,
 * rows_out=M, row_off=0 for the identity map; the decoder's final norm uses the last N_mask rows of every clip). ----
 * x (the residual stream) is f32, or bf16 when x_is_bf16 != 0 (the decoder's stream). */
int mofo_layernorm_fwd(const void* x, int x_is_bf16, int ldx, const float* w, const float* b, float eps, int M, int D,
                       int rows_in, int rows_out, int row_off,
                       void* y_bf16, int ldy, float* mean, float* rstd, void* stream);
/* the same, and the normalised rows ALSO as OCP e4m3 (y_e4m3 [M, ldy8] = sat(y * qscale[0])): the A operand of the fp8 forward
 * GEMMs (MOFO_GEMM_NT_FP8).  amax_out = MOFO_FP8_AMAX_STRIPES floats (zeroed by the caller / by mofo_fp8_update_scales): max|y| over ALL rows, the
 * blocks' atomic maxima spread over the stripes; mofo_fp8_update_scales folds them into the next step's scale. */
#define MOFO_FP8_AMAX_STRIPES 1024
int mofo_layernorm_fwd_q(const void* x, int x_is_bf16, int ldx, const float* w, const float* b, float eps, int M, int D,
                         int rows_in, int rows_out, int row_off, void* y_bf16, int ldy, float* mean, float* rstd,
                         void* y_e4m3, int ldy8, const float* qscale, float* amax_out, void* stream);
/* dx = dres + LN'(dy).  The incoming residual-stream gradient is dres (f32) OR dres_bf16 (bf16) OR neither; the result
 * goes to dx (f32) and/or dx_bf16 (at least one).  dw/db accumulate (+=): with partial_ws the per-block partials are stored
 * and summed by a second tiny kernel (one writer per address, deterministic); without it every block adds atomically. */
int mofo_layernorm_bwd(const void* dy_bf16, int lddy, const void* x, int x_is_bf16, int ldx, const float* w,
                       const float* mean, const float* rstd, const float* dres, int lddres, int M, int D,
                       int rows_in, int rows_out, int row_off,
                       float* dx, int lddx, void* dx_bf16, int lddxb, float* dw, float* db,
                       const void* dres_bf16, int lddres_bf16,
                       float* partial_ws /* >= 2 * mofo_layernorm_bwd_blocks(M) * D floats of scratch (2*1024*D always suffices), or NULL: NULL falls back to contended atomics */,
                       void* stream);
/* The same with a PARTIAL residual gradient: of every group of dres_period rows only the rows t >= dres_skip carry one, stored
 * compactly (row (r / period) * (period - skip) + r % period - skip of dres / dres_bf16); the others get none.  The decoder block
 * below the last one: only its masked tokens were passed on (modeling_pretrain.py:157).  dres_period = 0: mofo_layernorm_bwd. */
int mofo_layernorm_bwd_partial_res(const void* dy, int lddy, const void* x, int x_is_bf16, int ldx, const float* w, const float* mean,
                                   const float* rstd, const float* dres, int lddres, int M, int D, int rows_in, int rows_out, int row_off,
                                   float* dx, int lddx, void* dx_bf16, int lddxb, float* dw, float* db, const void* dres_bf16, int lddres_bf16,
                                   float* partial_ws, int dres_period, int dres_skip, void* stream);
/* Deferred reduction: called with dw = db = NULL (and a partial_ws of its own) mofo_layernorm_bwd leaves only the
 * mofo_layernorm_bwd_blocks(M) block partials in partial_ws; mofo_layernorm_bwd_finalize adds the partials of up to 40
 * LayerNorms to their dw / db in one launch (34 per-LayerNorm reduction launches per ViT-B step become 1). */
int mofo_layernorm_bwd_blocks(int M);
int mofo_layernorm_bwd_finalize(const float* const* partials, const int* nblocks, const int* Ds, float* const* dws,
                                float* const* dbs, int count, void* stream);

/* ---- multi-head self-attention core: modeling_finetune.py:85-95 (q*scale, q@k^T, softmax, @v).
 * qkv is the fused projection output, bf16 [B*N, 3*H*64] (q | k | v, each head-major x 64); head_dim is 64 in every
 * configuration of the reference.  out bf16 [B*N, H*64]; lse2 f32 [B,H,N] = log2-sum-exp2 of (scale*log2e*scores). ---- */
int mofo_attention_fwd(const void* qkv, int ldqkv, int B, int N, int H, float scale,
                       void* out, int ldo, float* lse2, void* stream);
/* dqkv bf16 [B*N, 3*H*64] (same layout as qkv); delta f32 [B,H,N] is scratch written by the call. */
int mofo_attention_bwd(const void* qkv, int ldqkv, const void* out, int ldo, const void* dout, int lddo,
                       const float* lse2, int B, int N, int H, float scale,
                       void* dqkv, int lddqkv, float* delta, void* stream);

/* the three parts of mofo_attention_bwd as separate entries, so that a caller can run the dQ pass and the dK/dV pass
 * (independent once delta is known; each alone keeps the MFMA pipe ~35 % busy) concurrently on two streams:
 *   delta[b,h,q] = sum_d dout[q,h,d] * out[q,h,d];  dq -> first third of dqkv;  dk, dv -> second and third thirds. */
int mofo_attention_delta(const void* out, int ldo, const void* dout, int lddo, int B, int N, int H, float* delta, void* stream);
int mofo_attention_bwd_dq(const void* qkv, int ldqkv, const void* dout, int lddo, const float* lse2, const float* delta,
                          int B, int N, int H, float scale, void* dqkv, int lddqkv, void* stream);
int mofo_attention_bwd_dkv(const void* qkv, int ldqkv, const void* dout, int lddo, const float* lse2, const float* delta,
                           int B, int N, int H, float scale, void* dqkv, int lddqkv, void* stream);
/* Query RANGES: only the query rows q_begin .. N - 1 of every clip are worked on -- the last decoder block, whose visible-token
 * outputs nothing reads (modeling_pretrain.py:157 keeps x[:, -return_token_num:]; the visible tokens are rows 0 .. n_vis - 1 of a
 * clip, modeling_pretrain.py:259).  `out` / `dout` hold those N - q_begin rows per clip COMPACTLY: bf16 [B * (N - q_begin), H*64];
 * qkv, dqkv, lse2 and delta keep whole-sequence indices (keys and values are still all N rows; dq rows below q_begin are NOT
 * written: clear them).  q_begin = 0 is exactly the entries above. */
int mofo_attention_fwd_range(const void* qkv, int ldqkv, int B, int N, int H, float scale, int q_begin,
                             void* out, int ldo, float* lse2, void* stream);
/* mofo_attention_fwd_range that ALSO writes the output rows as OCP e4m3 (out_e4m3 [B * (N - q_begin), ldo8 bytes] = sat(O * q_scale[0])):
 * the A operand of an fp8 proj GEMM (BASELINE configs[4]).  q_scale: device scalar (delayed scaling); q_amax: MOFO_FP8_AMAX_STRIPES
 * floats collecting max|O| of this launch (mofo_fp8_update_scales folds them into the next step's scale). */
int mofo_attention_fwd_q8(const void* qkv, int ldqkv, int B, int N, int H, float scale, int q_begin, void* out, int ldo, float* lse2,
                          void* out_e4m3, int ldo8, const float* q_scale, float* q_amax, void* stream);
int mofo_attention_delta_range(const void* out, int ldo, const void* dout, int lddo, int B, int N, int H, int q_begin, float* delta,
                               void* stream);
int mofo_attention_bwd_dq_range(const void* qkv, int ldqkv, const void* dout, int lddo, const float* lse2, const float* delta,
                                int B, int N, int H, float scale, int q_begin, void* dqkv, int lddqkv, void* stream);
/* (the dK/dV pass also CLEARS the dq columns of the rows 0 .. q_begin - 1 of every clip: no dQ pass writes them, the qkv dgrad reads them) */
int mofo_attention_bwd_dkv_range(const void* qkv, int ldqkv, const void* dout, int lddo, const float* lse2, const float* delta,
                                 int B, int N, int H, float scale, int q_begin, void* dqkv, int lddqkv, void* stream);
/* The dQ pass that also computes delta = rowsum(dO * O) of its query rows (out: laid out like dout) and WRITES it to delta_out for the
 * dK/dV pass: call it first, then mofo_attention_bwd_dkv[_range] on the same stream; mofo_attention_delta[_range] is then not needed. */
int mofo_attention_bwd_dq_delta_range(const void* qkv, int ldqkv, const void* out, int ldo, const void* dout, int lddo, const float* lse2,
                                      float* delta_out, int B, int N, int H, float scale, int q_begin, void* dqkv, int lddqkv, void* stream);

/* ---- RCCL communicator: the stand-alone route to the ONE collective of the path -- the per-step gradient
 * all-reduce of run_mae_pretraining.py:225-227 (DDP) over the group of utils.py:289-294 -- for callers that bind this library
 * without torch.distributed (mofo_amd/dist.py itself uses torch.distributed's "nccl" = RCCL backend).  librccl is opened at
 * run time.  Rank 0 calls mofo_comm_unique_id and hands the 128 bytes to the other ranks by its own means (file, env, TCP);
 * every rank then calls mofo_comm_init (a collective) with HIP's current device = its GPU.  The handle is owned by the library.
 * mofo_comm_allreduce_f32: in-place SUM over the ranks, enqueued on `stream` (pre-scale by 1/world for DDP's mean). */
int mofo_comm_unique_id(void* id128);
int mofo_comm_init(const void* id128, int rank, int world, void** comm_out);
int mofo_comm_allreduce_f32(void* comm, float* buf, long long n, void* stream);
int mofo_comm_destroy(void* comm);

/* zero the listed 1024-element chunks of a flat f32 buffer (optimizer.zero_grad() over the gradient ranges the next backward
 * accumulates into; the ranges it overwrites are skipped).  chunk_ids: device int32 [n]. */
int mofo_zero_chunks(float* base, const int* chunk_ids, int n, void* stream);

/* ---- OCP e4m3 quantisation for MOFO_GEMM_NT_FP8 (per-tensor scales; BASELINE configs[4]).
 * segments: x is a flat bf16 buffer of n = 1024 * chunks elements; chunk_seg[c] names the tensor ("segment", 0..nseg-1) chunk c
 * belongs to, or -1 (left untouched).  Per segment: amax -> out = sat_e4m3(x * 448 / amax), scale_inv[seg] = amax / 448.
 * amax_ws: nseg floats of scratch.  (The runtime quantises the whole bf16 weight shadow with it: two launches per step.) */
int mofo_fp8_quantize_segments(const void* x_bf16, long long n, const short* chunk_seg, int nseg, float* amax_ws,
                               void* out_e4m3, float* scale_inv, void* stream);
/* out = sat_e4m3(x * scale[0]) for a plain bf16 tensor; amax_out (or NULL) collects max|x| by atomic max */
int mofo_fp8_quantize_bf16(const void* x_bf16, long long n, const float* scale, void* out_e4m3, float* amax_out, void* stream);
/* delayed scaling: a_i = max over amax[i][0 .. MOFO_FP8_AMAX_STRIPES); scales[2i] = 448 / (a_i * margin), scales[2i+1] = its inverse;
 * the stripes are cleared.  a_i == 0 keeps the old pair */
int mofo_fp8_update_scales(float* amax, float* scales, int n, float margin, void* stream);
/* weights, delayed scaling through the optimizer (mofo_adamw_q8): once per step BEFORE its first update launch --
 * scale[i] = 448 / amax[i], scale_inv[i] = amax[i] / 448, amax[i] = 0; amax[i] == 0 keeps the old pair.  The gate words are those of
 * the update that follows (mofo_adamw_gated): a declined update leaves the e4m3 shadow untouched, and so does this its scales */
int mofo_fp8_roll_scales(float* amax, float* scale, float* scale_inv, int n, const float* gate_finite, const int* gate_zero,
                         const float* gate_one, void* stream);

/* ---- masks -> index lists: replaces the boolean gathers x[~mask] / pos[mask] of modeling_pretrain.py:90,261-262
 * and engine_for_pretraining.py:63 (which cost a device->host sync in the reference).  mask: uint8 [B,N], 1 = masked.
 * vis_idx [B,n_vis], msk_idx [B,N-n_vis]: ascending token ids per clip.  status[0] |= 1 if a clip's count differs. ---- */
int mofo_mask_to_indices(const uint8_t* mask, int B, int N, int n_vis, int* vis_idx, int* msk_idx, int* status, void* stream);
/* Device-side tube masks (the distribution of masking_generator.py:3-24, NOT numpy's random stream):
 * mask u8 [B, frames * patches_per_frame], 1 = masked; per clip one pattern with exactly n_mask masked patches per frame,
 * repeated over the frames; clip c of the call uses the key stream (seed, counter + c) -- pass a running clip counter. */
int mofo_tube_masks(unsigned seed, unsigned counter, int B, int frames, int patches_per_frame, int n_mask, uint8_t* mask, void* stream);

/* ---- on-device ingest (the step BEFORE the path): frames uint8 [B,H,W,T*3] = the reference's Stack()
 * output (transforms.py:346-360) -> clips f32 [B,3,T,H,W] = ((u/255) - mean_c) / std_c, i.e. ToTorchFormatTensor(div=True)
 * (transforms.py:363-382) + GroupNormalize (datasets.py:12-14) + the view/transpose at kinetics.py:492-493 (any T).  Bit-exact with
 * the reference's fp32 arithmetic; cuts the H2D copy from 9.63 MB to 2.41 MB per clip. ---- */
int mofo_ingest_u8(const uint8_t* frames, int B, int T, int H, int W, float* clips, void* stream);

/* ---- tubelet gather for PatchEmbed over VISIBLE tokens only: modeling_finetune.py:238-248 + modeling_pretrain.py:90.
 * clips f32 [B,C,T,H,W]; token id = t*(H/p)*(W/p) + h*(W/p) + w; out bf16 [B*n_tok, C*pt*p*p], column order (c,pt,ph,pw)
 * = Conv3d weight order.  The GEMM (NT, POS_F32 epilogue) against proj.weight.view(D,-1) finishes PatchEmbed + pos. ---- */
int mofo_patch_gather(const float* clips, int B, int C, int T, int H, int W, int pt, int p,
                      const int* tok_idx, int n_tok, void* out_bf16, int ldo, void* stream);

/* ---- decoder input assembly: modeling_pretrain.py:260-263.  Writes the masked half of x_full ([B,N,D], f32 or bf16):
 * x_full[b, n_vis + j] = mask_token + pos[msk_idx[b,j]]  (the visible half is the e2d GEMM's POS_F32 / POS_BF16 epilogue). ---- */
int mofo_fill_mask_tokens(const float* mask_token, const float* pos, int ldpos, const int* msk_idx,
                          int B, int N, int n_vis, int D, void* x_full, int x_is_bf16, void* stream);
/* ---- decoder block 0, shared masked rows.  Its input rows of the masked tokens are mask_token + pos[j] (modeling_pretrain.py:259-262):
 * they depend on the position only, and so do their LayerNorm-1 and qkv rows (modeling_finetune.py:216-219).  A caller computes those
 * once per position next to the visible rows -- "cat" layout [B * n_vis visible rows | N position rows] x W bf16 -- and
 *   mofo_dec0_gather : full[b, r] = r < n_vis ? cat[b * n_vis + r] : cat[B * n_vis + msk_idx[b, r - n_vis]]      ([B * N, W])
 *   mofo_dec0_reduce : its adjoint for gradients (visible rows copied, position rows = f32 sum over the clips that mask that position)
 *   mofo_dec0_inverse: inv[b, j] = slot of position j in clip b's ASCENDING masked list (as mofo_mask_to_indices writes it), or -1
 *                      (what mofo_dec0_reduce reads). ---- */
int mofo_dec0_inverse(const int* msk_idx, int B, int N, int n_vis, int* inv, void* stream);
int mofo_dec0_gather(const void* cat, int ldcat, const int* msk_idx, int B, int N, int n_vis, int W, void* full, int ldfull, void* stream);
int mofo_dec0_reduce(const void* full, int ldfull, const int* inv, int B, int N, int n_vis, int W, void* cat, int ldcat, void* stream);
/* backward of the assembly: d_e2d(bf16)[b*n_vis + j] = dx_full[b, j];  d_mask_token[d] += sum over masked rows.
 * dx_full is f32 (dx_is_bf16 = 0) or bf16 (1).  partial_ws: mofo_assemble_bwd_blocks(B, N) * D floats of scratch for the
 * blocks' column sums (a second small launch adds them), or NULL: every block then adds to d_mask_token with atomics
 * (same-address adds serialise at ~115 ns each: 95 us instead of ~20 at ViT-B, B = 32). */
int mofo_assemble_bwd_blocks(int B, int N);
int mofo_assemble_bwd(const void* dx_full, int dx_is_bf16, int B, int N, int n_vis, int D, void* d_e2d_bf16, float* d_mask_token,
                      float* partial_ws, void* stream);
/* Deferred form: mofo_assemble_bwd with d_mask_token = NULL only leaves the block partials in partial_ws; this adds them to
 * d_mask_token (+=).  d(mask_token) is needed by nothing before the optimizer, so the caller can take the tiny reduce launch
 * off the activation-gradient chain (it used to wait ~180 us for a CU slot behind a weight-gradient launch). */
int mofo_assemble_bwd_finalize(const float* partial_ws, int B, int N, int D, float* d_mask_token, void* stream);

/* ---- reconstruction target + MSE: engine_for_pretraining.py:43-63 (un-normalise, patchify (p0 p1 p2) c,
 * per-(token,channel) standardise with UNBIASED var and 1e-6 after the sqrt, gather masked) and :27,67 (nn.MSELoss).
 * pred bf16 [B*n_msk, C*pt*p*p] (feature order (pt,ph,pw,c)); row_loss f32 [B*n_msk] scratch; loss f32 [1];
 * dpred bf16 (may be NULL) = grad_scale * 2*(pred-target)/numel.  normalize=0 -> raw pixel targets (:59-60). ---- */
int mofo_target_mse(const float* clips, int B, int C, int T, int H, int W, int pt, int p,
                    const int* msk_idx, int n_msk, const void* pred, int ldp, int normalize, float grad_scale,
                    float* row_loss, float* loss, void* dpred, int lddp, void* target_out_f32, void* stream);

/* ---- ingest fused into the two kernels that read the clip: the same results as mofo_ingest_u8
 * followed by mofo_patch_gather / mofo_target_mse, bit for bit, computed straight from the uint8 frame stack
 * [B,H,W,T*3] (transforms.py:346-360) -- the f32 clip is never materialised (C = 3; target: tubelet 2, patch 16). ---- */
int mofo_patch_gather_u8(const uint8_t* frames, int B, int T, int H, int W, int pt, int p,
                         const int* tok_idx, int n_tok, void* out_bf16, int ldo, void* stream);
int mofo_target_mse_u8(const uint8_t* frames, int B, int T, int H, int W, int pt, int p,
                       const int* msk_idx, int n_msk, const void* pred, int ldp, int normalize, float grad_scale,
                       float* row_loss, float* loss, void* dpred, int lddp, void* stream);

/* ---- "next" rows, same boundary rules.
 * Reconstruction video of the inference / visualisation script, run_videomae_vis.py:150-180: every token standardised per
 * channel over its 512 pixels (unbiased variance, 1e-6 after the sqrt), masked tokens replaced by the model's predictions
 * (pred [B*n_msk, 1536], bf16 or f32, rows in the order of msk_idx), multiplied back by the token's own std and mean.
 * rec / masked / ori: f32 [B,3,T,H,W] in [0,1] pixel units; masked (rec on visible tokens, 0 on masked ones) and ori
 * (clips * std + mean) may be NULL.  msk_idx: ascending per clip, as mofo_mask_to_indices writes it. ---- */
int mofo_reconstruct(const float* clips, int B, int C, int T, int H, int W, int pt, int p, const int* msk_idx, int n_msk,
                     const void* pred, int pred_is_bf16, int ldp, float* rec, float* masked, float* ori, void* stream);
/* Token mean pooling + fc_norm of the fine-tune / feature-extraction model, modeling_finetune.py:403-405
 * (`self.fc_norm(x.mean(1))`): x f32 [B*N, D] -> out_f32 [B, D] (+ bf16 copy for the head GEMM, may be NULL).
 * pooled_ws: f32 [B, D] scratch (zeroed by the call). ---- */
int mofo_token_mean_norm(const float* x, int ldx, int B, int N, int D, const float* w, const float* b, float eps,
                         float* pooled_ws, float* out_f32, void* out_bf16, void* stream);

/* ---- optimizer side on FLAT buffers: utils.py:376-388 (global grad L2 norm), torch.nn.utils.clip_grad_norm_
 * (utils.py:359), torch.optim.AdamW as configured by optim_factory.py:91-127 (two param groups: decayed / not decayed).
 * n is a multiple of 1024; chunk c covers elements [1024c, 1024c+1024); chunk_group[c] (0/1) selects (lr0,wd0) or
 * (lr1,wd1).  partial f32 [>= 1024] scratch.  If max_norm > 0 the gradient is scaled by min(1, max_norm/(norm+1e-6))
 * with norm read from grad_norm[0] on the device (no host sync).  When no clipping is wanted the norm is not needed
 * BEFORE the update: pass norm_partial (f32 [>= 2048] scratch) and norm_out and mofo_adamw leaves the global L2 norm of
 * the gradients it has just read in norm_out[0] (saves mofo_sumsq's extra pass over them); both NULL otherwise.
 * Range by range (data parallelism: a range of the flat buffers is updated as soon as ITS all-reduce has landed, while
 * later ranges are still on the wire): call mofo_adamw on sub-ranges (offsets multiples of 1024) with norm_partial pointing
 * at consecutive slices of mofo_adamw_blocks(n_range) floats and norm_out NULL, then mofo_norm_finalize over all slices. ---- */
int mofo_sumsq(const float* g, long long n, float* partial, float* out_norm, void* stream);
int mofo_adamw(float* p, const float* g, float* m, float* v, void* p_bf16, long long n, const uint8_t* chunk_group,
               float lr0, float wd0, float lr1, float wd1, float beta1, float beta2, float eps, int step,
               const float* grad_norm, float max_norm, float grad_mult, float* norm_partial, float* norm_out, void* stream);
/* mofo_adamw behind a device-side gate: the kernel leaves p, m, v and p_bf16 untouched unless gate_finite[0] is finite,
 * gate_zero[0] == 0 and gate_one[0] == 1.0f (each pointer may be NULL = not checked).  A skipped update writes NaN into its
 * norm_partial slots, so the norm reduced behind it (norm_out / mofo_norm_finalize) reads NaN, never a previous step's value.
 * The gate reflects the LAST forward only, and the caller's step counter (bias correction) is the caller's business: the drop-in
 * engine aborts on the non-finite loss it reads right after (engine_for_pretraining.py:168-170), as the reference does.  The host enqueues
 * backward and the update before it reads the loss; the reference stops before backward on a non-finite loss
 * (engine_for_pretraining.py:168-176), so a bad step must not reach the parameters. */
int mofo_adamw_gated(float* p, const float* g, float* m, float* v, void* p_bf16, long long n, const uint8_t* chunk_group,
                     float lr0, float wd0, float lr1, float wd1, float beta1, float beta2, float eps, int step,
                     const float* grad_norm, float max_norm, float grad_mult, float* norm_partial, float* norm_out,
                     const float* gate_finite, const int* gate_zero, const float* gate_one, void* stream);
/* mofo_adamw_gated that also writes the OCP e4m3 shadow of the fp8 forward's GEMM weights (BASELINE configs[4]) from the pass that
 * holds the new weight anyway: chunk_seg[n / 1024] = weight-matrix index of each 1024-element chunk or -1 (as mofo_fp8_quantize_segments),
 * p_e4m3[i] = sat(bf16(p[i]) * w_scale[seg]) with the DELAYED scale of mofo_fp8_roll_scales, w_amax[seg] = max |bf16(p)| of the new
 * values (atomic max; rolled into the next step's scale).  p_bf16 is required.  Range-by-range calls pass the slices of every
 * per-element / per-chunk array and the WHOLE w_scale / w_amax arrays. */
int mofo_adamw_q8(float* p, const float* g, float* m, float* v, void* p_bf16, long long n, const uint8_t* chunk_group,
                  float lr0, float wd0, float lr1, float wd1, float beta1, float beta2, float eps, int step,
                  const float* grad_norm, float max_norm, float grad_mult, float* norm_partial, float* norm_out,
                  const float* gate_finite, const int* gate_zero, const float* gate_one, const short* chunk_seg,
                  const float* w_scale, float* w_amax, void* p_e4m3, void* stream);
int mofo_adamw_blocks(long long n);
int mofo_norm_finalize(const float* partial, int count, float* out_norm, void* stream);
int mofo_cast_bf16(const float* src, void* dst_bf16, long long n, void* stream);
/* the widening cast (exact).  With mofo_cast_bf16 the two ends of the OPTIONAL bf16 gradient transport of the data-parallel exchange
 * (SURVEY.md 8e "Collective": 188 instead of 377 MB per rank and step; mofo_amd/dist.py, MOFO_GRAD_BF16=1): a range of the flat f32
 * gradient buffer is narrowed into a wire buffer, all-reduced there and widened back before AdamW reads it. */
int mofo_cast_f32(const void* src_bf16, float* dst, long long n, void* stream);

#ifdef __cplusplus
}
#endif
#endif
