// Token plumbing kernels (all HBM-bound, integer/byte + copy work; gfx950):
//   mask -> ascending index lists, tubelet gather of visible patches, decoder input assembly and its backward,
//   column sums (bias gradients).
#include "common.h"
#include "../../include/mofo_hip.h"

namespace {

// ---------------------------------------------------------------------------------------------- mask -> indices
// one 256-thread block per clip; thread t owns elements [t*E, t*E+E); block-wide exclusive scan of visible counts.
__global__ __launch_bounds__(256) void mask_idx_kernel(const uint8_t* __restrict__ mask, int N, int n_vis,
                                                       int* __restrict__ vis_idx, int* __restrict__ msk_idx,
                                                       int* __restrict__ status) {
    __shared__ int wsum[4];
    const int b = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int E = (N + 255) / 256;
    const int beg = t * E, end = min(N, beg + E);
    const uint8_t* m = mask + (size_t)b * N;
    int cnt = 0;
    for (int i = beg; i < end; ++i) cnt += (m[i] == 0);
    int inc = cnt;  // inclusive scan inside the wave
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(inc, o, 64);
        if (lane >= o) inc += v;
    }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    int base = 0;
    for (int w = 0; w < wave; ++w) base += wsum[w];
    const int total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    if (t == 0 && total != n_vis) atomicOr(status, 1);
    int v = base + inc - cnt;               // visible tokens before `beg`
    int k = (beg < N ? beg : N) - v;        // masked tokens before `beg`
    const int n_msk = N - n_vis;
    int* vo = vis_idx + (size_t)b * n_vis;
    int* mo = msk_idx + (size_t)b * n_msk;
    for (int i = beg; i < end; ++i) {
        if (m[i] == 0) {
            if (v < n_vis) vo[v] = i;
            ++v;
        } else {
            if (k < n_msk) mo[k] = i;
            ++k;
        }
    }
}

// ---------------------------------------------------------------------------------------------- device-side tube masks
// the tube mask of masking_generator.py:3-24 drawn ON the device -- one per-frame pattern with exactly
// n_mask of the P = height x width patches masked, repeated over the F temporal slots -- so that no mask crosses PCIe.  The
// reference shuffles with numpy's global Mersenne-Twister stream inside DataLoader workers; a device kernel cannot continue that
// stream, so this is a generator of its own (the host classes in masking_generator.py stay the bit-exact drop-in): patch i of
// clip c gets the 32-bit key mix32(seed, counter + c, i), the n_mask SMALLEST keys (ties: lower index first) are masked.  Every
// subset of size n_mask is equally likely up to the quality of the integer mixer; oracle/pretrain_oracle.py restates it in numpy.
__device__ __forceinline__ uint32_t mix32(uint32_t x) {          // "lowbias32" integer finaliser
    x ^= x >> 16;
    x *= 0x7feb352du;
    x ^= x >> 15;
    x *= 0x846ca68bu;
    x ^= x >> 16;
    return x;
}
__global__ __launch_bounds__(256) void tube_mask_kernel(uint32_t seed, uint32_t counter, int F, int P, int n_mask, uint8_t* __restrict__ mask) {
    __shared__ uint32_t key[1024];
    __shared__ uint8_t pat[1024];
    const int c = blockIdx.x;
    const uint32_t base = mix32(seed ^ mix32(counter + (uint32_t)c + 0x9e3779b9u));
    for (int i = threadIdx.x; i < P; i += 256) key[i] = mix32(base + 0x85ebca6bu * (uint32_t)(i + 1));
    __syncthreads();
    for (int i = threadIdx.x; i < P; i += 256) {
        const uint32_t k = key[i];
        int rank = 0;                                            // patches that sort before patch i
        for (int j = 0; j < P; ++j) rank += (key[j] < k) || (key[j] == k && j < i);
        pat[i] = rank < n_mask;
    }
    __syncthreads();
    uint8_t* m = mask + (size_t)c * F * P;
    for (int i = threadIdx.x; i < F * P; i += 256) m[i] = pat[i % P];
}

// ---------------------------------------------------------------------------------------------- tubelet gather
// Pixel sources.  F32: the model's input contract, clips f32 [B][C][T][H][W] ImageNet-normalised.  U8 (SURVEY.md 8f rank 3,
// "ingest fused into the K1 / K12 reads"): the reference's Stack() output per clip, uint8 [B][H][W][T*3], normalised on the
// fly with exactly the operations of mofo_ingest_u8 -- ((u / 255) - mean_c) / std_c -- so the fused path is bit-identical
// to ingest + the f32 kernels while the f32 clip (4x the bytes) is never written or read.
__constant__ float c_src_mean[3] = {0.485f, 0.456f, 0.406f};
__constant__ float c_src_std[3] = {0.229f, 0.224f, 0.225f};
struct SrcF32 {
    const float* clips; int C, T, H, W;
    __device__ __forceinline__ f32x4 load4(int b, int c, int t, int y, int x0) const {
        return *(const f32x4*)(clips + ((((size_t)b * C + c) * T + t) * H + y) * W + x0);
    }
};

// uint8 source, tubelet 2 x patch 16: one wave per token; lane -> image row p1 = lane >> 2, pixels (lane & 3) * 4 + k.  The 6
// bytes a tubelet needs of one pixel (2 frames x 3 channels) are contiguous in the frame stack: one bounds-checked 8-byte
// buffer load per pixel, then six packed stores -- the same (c, p0, p1, p2) columns the f32 kernel writes.
__global__ __launch_bounds__(256) void patch_gather_u8_kernel(const uint8_t* __restrict__ frames, uint32_t frame_bytes, int T, int H, int W,
                                                              const int* __restrict__ tok_idx, int n_tok, int rows,
                                                              bf16_t* __restrict__ out, int ldo) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = blockIdx.x * 4 + wave;
    if (row >= rows) return;
    const int b = row / n_tok;
    const int tok = tok_idx[row];
    const int gw = W >> 4, gh = H >> 4;
    const int tw = tok % gw, th = (tok / gw) % gh, tt = tok / (gw * gh);
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)frames, 0, (int)frame_bytes, 0x00020000);
    uint64_t px[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t addr = (uint32_t)(((b * H + th * 16 + (lane >> 2)) * W + tw * 16 + (lane & 3) * 4 + k) * (T * 3) + tt * 6);
        const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rsrc, (int)(addr & ~3u), 0, 0);
        px[k] = (((uint64_t)v[1] << 32) | v[0]) >> ((addr & 3u) * 8);
    }
    bf16_t* orow = out + (size_t)row * ldo;
#pragma unroll
    for (int i = 0; i < 6; ++i) {            // float4 #(i*64 + lane) of the row: c = i >> 1, p0 = i & 1
        const int c = i >> 1, p0 = i & 1;
        float v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = ((float)(uint32_t)((px[k] >> (8 * (p0 * 3 + c))) & 0xff) / 255.0f - c_src_mean[c]) / c_src_std[c];
        u32x2 pk = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
        *(u32x2*)(orow + (i * 64 + lane) * 4) = pk;
    }
}

// one wave per token row; lane e handles float4 #e of the row: row layout (c, p0, p1, p2) so that float4 #e is 16 B
// of one 16-pixel image row segment (64 B contiguous in the clip).
template <class SRC>
__global__ __launch_bounds__(256) void patch_gather_kernel(SRC src, int pt, int p, const int* __restrict__ tok_idx, int n_tok,
                                                           int rows, bf16_t* __restrict__ out, int ldo) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = blockIdx.x * 4 + wave;
    if (row >= rows) return;
    const int b = row / n_tok;
    const int tok = tok_idx[row];
    const int gw = src.W / p, gh = src.H / p;
    const int tw = tok % gw, th = (tok / gw) % gh, tt = tok / (gw * gh);
    const int q4 = p >> 2;            // float4 per segment
    const int n4 = src.C * pt * p * q4;   // float4 per row
    bf16_t* orow = out + (size_t)row * ldo;
    for (int e = lane; e < n4; e += 64) {
        const int seg = e / q4, qq = e - seg * q4;
        const int c = seg / (pt * p), rem = seg - c * (pt * p);
        const int p0 = rem / p, p1 = rem - p0 * p;
        const f32x4 v = src.load4(b, c, tt * pt + p0, th * p + p1, tw * p + qq * 4);
        u32x2 pk = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
        *(u32x2*)(orow + e * 4) = pk;
    }
}

// ---------------------------------------------------------------------------------------------- decoder assembly
template <bool OUTB>   // OUTB: the decoder's input stream is bf16
__global__ __launch_bounds__(256) void fill_mask_kernel(const float* __restrict__ mask_token, const float* __restrict__ pos,
                                                        int ldpos, const int* __restrict__ msk_idx, int N, int n_vis, int D,
                                                        int rows, void* __restrict__ x_full) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = blockIdx.x * 4 + wave;
    if (row >= rows) return;
    const int n_msk = N - n_vis;
    const int b = row / n_msk, j = row - b * n_msk;
    const float* pr = pos + (size_t)msk_idx[row] * ldpos;
    const size_t drow = ((size_t)b * N + n_vis + j) * D;
    for (int c = lane * 4; c < D; c += 256) {
        const f32x4 v = *(const f32x4*)(mask_token + c) + *(const f32x4*)(pr + c);
        if constexpr (OUTB) *(u32x2*)((bf16_t*)x_full + drow + c) = u32x2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
        else *(f32x4*)((float*)x_full + drow + c) = v;
    }
}

// 256 threads = CT column threads (one float4 each, CT = D/4 rounded up to a power-of-two divisor of 256) x RL row lanes;
// a block covers RB rows; the masked rows' column sums are reduced over the row lanes in LDS, one atomic per column.
// RB rows per block: fewer blocks => fewer adders on the same D addresses of d_mask_token (contended same-address f32 atomics
// run ~14x below the atomic rate), but 256 rows per block left 196 blocks for the 50 176 rows of ViT-B, B=32 -- less than one
// per CU, 78 us for 38.5 MB.  64 rows: 784 blocks, 784 adds per address (~10 us of serialised atomics, under the read time).
constexpr int RB = 64;
template <bool BF16IN>
__global__ __launch_bounds__(256) void assemble_bwd_kernel(const void* __restrict__ dxv, int N, int n_vis, int D, int rows, int ct,
                                                           bf16_t* __restrict__ d_e2d, float* __restrict__ d_mask_token) {
    __shared__ f32x4 red[256];
    const int cthr = threadIdx.x % ct, rl = threadIdx.x / ct, nrl = 256 / ct;
    const int c = cthr * 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const int r0 = blockIdx.x * RB;
    if (c < D) {
        for (int r = r0 + rl; r < min(rows, r0 + RB); r += nrl) {
            const int b = r / N, j = r - b * N;
            f32x4 v;
            u32x2 raw = {0u, 0u};
            if constexpr (BF16IN) {
                raw = *(const u32x2*)((const bf16_t*)dxv + (size_t)r * D + c);
                v = f32x4{bf16lo_to_f32(raw[0]), bf16hi_to_f32(raw[0]), bf16lo_to_f32(raw[1]), bf16hi_to_f32(raw[1])};
            } else {
                v = *(const f32x4*)((const float*)dxv + (size_t)r * D + c);
                raw = u32x2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
            }
            if (j < n_vis) {
                *(u32x2*)(d_e2d + ((size_t)b * n_vis + j) * D + c) = raw;
            } else {
                acc += v;
            }
        }
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    if (rl == 0 && c < D) {
        for (int k = 1; k < nrl; ++k) acc += red[k * ct + cthr];
#pragma unroll
        for (int e = 0; e < 4; ++e) atomicAdd(d_mask_token + c + e, acc[e]);
    }
}

// The same with one WAVE per row (16-B loads, four rows in flight per wave) and NO same-address atomics: the masked rows' column
// sums of a block go to partial[block][D], a second small launch adds the partials (16 adders per address).  With atomics from
// every block the kernel took 78 us (196 blocks) / 95 us (784 blocks) for a 38.5 MB read at ViT-B, B = 32: ~115 ns per
// serialised add on each of the D addresses, whatever the block count.
template <bool BF16IN>
__global__ __launch_bounds__(256) void assemble_bwd_rows_kernel(const void* __restrict__ dxv, int N, int n_vis, int D, int rows,
                                                                bf16_t* __restrict__ d_e2d, float* __restrict__ partial) {
    __shared__ float red[4][512];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane * 8;
    const bool on = c < D;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const int r0 = blockIdx.x * RB;
    const int rend = min(rows, r0 + RB);
    for (int rb = r0 + wave; rb < rend; rb += 16) {
        u32x4 raw[4];
        f32x4 lo[4], hi[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int r = rb + 4 * u;
            if (on && r < rend) {
                if constexpr (BF16IN) {
                    raw[u] = *(const u32x4*)((const bf16_t*)dxv + (size_t)r * D + c);
                } else {
                    lo[u] = *(const f32x4*)((const float*)dxv + (size_t)r * D + c);
                    hi[u] = *(const f32x4*)((const float*)dxv + (size_t)r * D + c + 4);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int r = rb + 4 * u;
            if (on && r < rend) {
                if constexpr (BF16IN) {
                    lo[u] = f32x4{bf16lo_to_f32(raw[u][0]), bf16hi_to_f32(raw[u][0]), bf16lo_to_f32(raw[u][1]), bf16hi_to_f32(raw[u][1])};
                    hi[u] = f32x4{bf16lo_to_f32(raw[u][2]), bf16hi_to_f32(raw[u][2]), bf16lo_to_f32(raw[u][3]), bf16hi_to_f32(raw[u][3])};
                } else {
                    raw[u] = u32x4{pack_bf16x2(lo[u][0], lo[u][1]), pack_bf16x2(lo[u][2], lo[u][3]), pack_bf16x2(hi[u][0], hi[u][1]),
                                   pack_bf16x2(hi[u][2], hi[u][3])};
                }
                const int b = r / N, j = r - b * N;
                if (j < n_vis) {
                    *(u32x4*)(d_e2d + ((size_t)b * n_vis + j) * D + c) = raw[u];
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        acc[e] += lo[u][e];
                        acc[4 + e] += hi[u][e];
                    }
                }
            }
        }
    }
    if (on) {
#pragma unroll
        for (int e = 0; e < 8; ++e) red[wave][c + e] = acc[e];
    }
    __syncthreads();
    for (int k = threadIdx.x; k < D; k += 256) partial[(size_t)blockIdx.x * D + k] = red[0][k] + red[1][k] + red[2][k] + red[3][k];
}

// out[c] += sum_b partial[b][c]: grid (D/64, COL_SLICES), 4 block lanes x 64 columns per block
constexpr int COL_SLICES = 16;
__global__ __launch_bounds__(256) void add_partials_kernel(const float* __restrict__ partial, int nblocks, int D, float* __restrict__ out) {
    __shared__ float red[4][64];
    const int cl = threadIdx.x & 63, part = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    const int per = (nblocks + COL_SLICES - 1) / COL_SLICES;
    const int b0 = blockIdx.y * per, b1 = min(nblocks, b0 + per);
    float s = 0.f;
    if (c < D)
        for (int b = b0 + part; b < b1; b += 4) s += partial[(size_t)b * D + c];
    red[part][cl] = s;
    __syncthreads();
    if (part == 0 && c < D && b0 < b1) atomicAdd(out + c, red[0][cl] + red[1][cl] + red[2][cl] + red[3][cl]);
}

// ---------------------------------------------------------------------------------------------- uint8 ingest
// frames: the reference's Stack() output per clip, uint8 [B][H][W][T*3] (frame-major, then r,g,b) -- transforms.py:346-360;
// clips: f32 [B][3][T][H][W] = ((u / 255) - mean_c) / std_c, i.e. ToTorchFormatTensor(div=True) (transforms.py:363-382),
// GroupNormalize (datasets.py:12-14) and the view/transpose of kinetics.py:492-493, with the same IEEE fp32 operations.
// One thread per pixel column w: VEC-byte loads of its T*3 bytes (16 when T*3 is a multiple of 16 -- 16 / 32 frames --, else
// 8 / 4 / 1), stores coalesced over w in each (c,t) plane.
__constant__ float c_in_mean[3] = {0.485f, 0.456f, 0.406f};
__constant__ float c_in_std[3] = {0.229f, 0.224f, 0.225f};
template <int VEC>
__global__ __launch_bounds__(256) void ingest_u8_kernel(const uint8_t* __restrict__ frames, int T, int H, int W,
                                                        float* __restrict__ clips) {
    const int w = blockIdx.x * 256 + threadIdx.x;
    const int h = blockIdx.y, b = blockIdx.z;
    if (w >= W) return;
    const int nb = T * 3;
    const uint8_t* src = frames + (((size_t)b * H + h) * W + w) * nb;
    float* dst = clips + (size_t)b * 3 * T * H * W + (size_t)h * W + w;
    const size_t plane = (size_t)H * W;
    for (int k = 0; k < nb; k += VEC) {
        uint32_t v[VEC >= 4 ? VEC / 4 : 1];
        if constexpr (VEC == 16) {
            const u32x4 q = *(const u32x4*)(src + k);
            v[0] = q[0]; v[1] = q[1]; v[2] = q[2]; v[3] = q[3];
        } else if constexpr (VEC == 8) {
            const u32x2 q = *(const u32x2*)(src + k);
            v[0] = q[0]; v[1] = q[1];
        } else if constexpr (VEC == 4) {
            v[0] = *(const uint32_t*)(src + k);
        } else {
            v[0] = src[k];
        }
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            const int idx = k + e, t = idx / 3, c = idx - 3 * t;
            const float u = (float)((v[e >> 2] >> (8 * (e & 3))) & 0xffu);
            dst[((size_t)c * T + t) * plane] = (u / 255.0f - c_in_mean[c]) / c_in_std[c];
        }
    }
}

// ---------------------------------------------------------------------------------------------- column sums
// block = 256 threads = 32 row lanes x 8 column groups of 8 bf16 (16 B): 64 columns x `rows_per_block` rows per block.
__global__ __launch_bounds__(256) void colsum_bf16_kernel(const bf16_t* __restrict__ X, int ldx, int M, int N,
                                                          int rows_per_block, float* __restrict__ out) {
    __shared__ float red[32][65];
    const int cg = threadIdx.x & 7, rl = threadIdx.x >> 3;
    const int c0 = blockIdx.x * 64 + cg * 8;
    const int rbeg = blockIdx.y * rows_per_block, rend = min(M, rbeg + rows_per_block);
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (c0 < N) {
        for (int r = rbeg + rl; r < rend; r += 32) {
            const u32x4 v = *(const u32x4*)(X + (size_t)r * ldx + c0);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                acc[2 * e] += bf16lo_to_f32(v[e]);
                acc[2 * e + 1] += bf16hi_to_f32(v[e]);
            }
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) red[rl][cg * 8 + e] = acc[e];
    __syncthreads();
    if (threadIdx.x < 64) {
        float s = 0.f;
#pragma unroll
        for (int r = 0; r < 32; ++r) s += red[r][threadIdx.x];
        const int c = blockIdx.x * 64 + threadIdx.x;
        if (c < N) atomicAdd(out + c, s);
    }
}

}  // namespace

extern "C" int mofo_ingest_u8(const uint8_t* frames, int B, int T, int H, int W, float* clips, void* stream) {
    if (!frames || !clips) MOFO_FAIL(MOFO_EINVAL, "mofo_ingest_u8: null pointer");
    if (B <= 0 || T <= 0 || H <= 0 || W <= 0) MOFO_FAIL(MOFO_EINVAL, "mofo_ingest_u8: bad sizes");
    if (H > 65535 || B > 65535) MOFO_FAIL(MOFO_EUNSUPPORTED, "mofo_ingest_u8: grid too large");
    const dim3 grid(ceil_div(W, 256), H, B);
    const int nb = T * 3;
    if (nb % 16 == 0) hipLaunchKernelGGL(ingest_u8_kernel<16>, grid, dim3(256), 0, (hipStream_t)stream, frames, T, H, W, clips);
    else if (nb % 8 == 0) hipLaunchKernelGGL(ingest_u8_kernel<8>, grid, dim3(256), 0, (hipStream_t)stream, frames, T, H, W, clips);
    else if (nb % 4 == 0) hipLaunchKernelGGL(ingest_u8_kernel<4>, grid, dim3(256), 0, (hipStream_t)stream, frames, T, H, W, clips);
    else hipLaunchKernelGGL(ingest_u8_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, frames, T, H, W, clips);
    MOFO_CHECK_LAUNCH("mofo_ingest_u8");
    return MOFO_OK;
}

extern "C" int mofo_tube_masks(unsigned seed, unsigned counter, int B, int frames, int patches_per_frame, int n_mask, uint8_t* mask, void* stream) {
    if (!mask) MOFO_FAIL(MOFO_EINVAL, "mofo_tube_masks: null pointer");
    if (B <= 0 || frames <= 0 || patches_per_frame <= 0 || patches_per_frame > 1024 || n_mask < 0 || n_mask > patches_per_frame)
        MOFO_FAIL(MOFO_EINVAL, "mofo_tube_masks: bad sizes (1 <= patches per frame <= 1024, 0 <= n_mask <= patches per frame)");
    hipLaunchKernelGGL(tube_mask_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, (uint32_t)seed, (uint32_t)counter, frames, patches_per_frame,
                       n_mask, mask);
    MOFO_CHECK_LAUNCH("mofo_tube_masks");
    return MOFO_OK;
}

extern "C" int mofo_mask_to_indices(const uint8_t* mask, int B, int N, int n_vis, int* vis_idx, int* msk_idx, int* status,
                                    void* stream) {
    if (!mask || !vis_idx || !msk_idx || !status) MOFO_FAIL(MOFO_EINVAL, "mofo_mask_to_indices: null pointer");
    if (B <= 0 || N <= 0 || n_vis <= 0 || n_vis >= N) MOFO_FAIL(MOFO_EINVAL, "mofo_mask_to_indices: bad sizes B=%d N=%d n_vis=%d", B, N, n_vis);
    hipLaunchKernelGGL(mask_idx_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, mask, N, n_vis, vis_idx, msk_idx, status);
    MOFO_CHECK_LAUNCH("mofo_mask_to_indices");
    return MOFO_OK;
}

extern "C" int mofo_patch_gather(const float* clips, int B, int C, int T, int H, int W, int pt, int p, const int* tok_idx,
                                 int n_tok, void* out, int ldo, void* stream) {
    if (!clips || !tok_idx || !out) MOFO_FAIL(MOFO_EINVAL, "mofo_patch_gather: null pointer");
    if (B <= 0 || C <= 0 || T <= 0 || H <= 0 || W <= 0 || pt <= 0 || p <= 0 || n_tok <= 0) MOFO_FAIL(MOFO_EINVAL, "mofo_patch_gather: bad sizes");
    if (p % 4 || W % 4 || T % pt || H % p || W % p || ldo % 4 || ldo < C * pt * p * p)
        MOFO_FAIL(MOFO_EUNSUPPORTED, "mofo_patch_gather: patch %d / tubelet %d do not tile %dx%dx%d or ldo too small", p, pt, T, H, W);
    const int rows = B * n_tok;
    hipLaunchKernelGGL(patch_gather_kernel<SrcF32>, dim3(ceil_div(rows, 4)), dim3(256), 0, (hipStream_t)stream, SrcF32{clips, C, T, H, W}, pt, p,
                       tok_idx, n_tok, rows, (bf16_t*)out, ldo);
    MOFO_CHECK_LAUNCH("mofo_patch_gather");
    return MOFO_OK;
}

extern "C" int mofo_patch_gather_u8(const uint8_t* frames, int B, int T, int H, int W, int pt, int p, const int* tok_idx,
                                    int n_tok, void* out, int ldo, void* stream) {
    if (!frames || !tok_idx || !out) MOFO_FAIL(MOFO_EINVAL, "mofo_patch_gather_u8: null pointer");
    if (B <= 0 || T <= 0 || H <= 0 || W <= 0 || pt <= 0 || p <= 0 || n_tok <= 0) MOFO_FAIL(MOFO_EINVAL, "mofo_patch_gather_u8: bad sizes");
    if (pt != 2 || p != 16) MOFO_FAIL(MOFO_EUNSUPPORTED, "mofo_patch_gather_u8: built for tubelet 2, patch 16 (got %d,%d)", pt, p);
    if (T % 2 || H % 16 || W % 16 || ldo % 4 || ldo < 1536)
        MOFO_FAIL(MOFO_EINVAL, "mofo_patch_gather_u8: 2x16x16 tubelets do not tile %dx%dx%d or ldo too small", T, H, W);
    const size_t bytes = (size_t)B * H * W * T * 3;
    if (bytes > 0x7fffffffu) MOFO_FAIL(MOFO_EUNSUPPORTED, "mofo_patch_gather_u8: frame stack of %zu bytes exceeds one 2 GiB buffer descriptor", bytes);
    const int rows = B * n_tok;
    hipLaunchKernelGGL(patch_gather_u8_kernel, dim3(ceil_div(rows, 4)), dim3(256), 0, (hipStream_t)stream, frames, (uint32_t)bytes, T, H, W,
                       tok_idx, n_tok, rows, (bf16_t*)out, ldo);
    MOFO_CHECK_LAUNCH("mofo_patch_gather_u8");
    return MOFO_OK;
}

extern "C" int mofo_fill_mask_tokens(const float* mask_token, const float* pos, int ldpos, const int* msk_idx, int B, int N,
                                     int n_vis, int D, void* x_full, int x_is_bf16, void* stream) {
    if (!mask_token || !pos || !msk_idx || !x_full) MOFO_FAIL(MOFO_EINVAL, "mofo_fill_mask_tokens: null pointer");
    if (B <= 0 || N <= n_vis || n_vis < 0 || D <= 0 || D % 4 || ldpos % 4) MOFO_FAIL(MOFO_EINVAL, "mofo_fill_mask_tokens: bad sizes");
    const int rows = B * (N - n_vis);
    if (x_is_bf16)
        hipLaunchKernelGGL(fill_mask_kernel<true>, dim3(ceil_div(rows, 4)), dim3(256), 0, (hipStream_t)stream, mask_token, pos, ldpos,
                           msk_idx, N, n_vis, D, rows, x_full);
    else
        hipLaunchKernelGGL(fill_mask_kernel<false>, dim3(ceil_div(rows, 4)), dim3(256), 0, (hipStream_t)stream, mask_token, pos, ldpos,
                           msk_idx, N, n_vis, D, rows, x_full);
    MOFO_CHECK_LAUNCH("mofo_fill_mask_tokens");
    return MOFO_OK;
}

// ---------------------------------------------------------------------------------------------- decoder block 0: shared masked rows
// In the FIRST decoder block the input rows of the masked tokens are mask_token + pos[j] (modeling_pretrain.py:259-262): they depend
// on the POSITION j only, not on the clip, and so do their LayerNorm-1 and qkv rows.  The runtime computes those once per position
// (N rows) beside the B * n_vis visible rows ("cat" layout: [B * n_vis visible rows | N position rows]) and
//   * dec0_gather spreads them to the whole-sequence layout the attention kernels read:
//       full[b, r] = r < n_vis ? cat[b * n_vis + r] : cat[B * n_vis + msk_idx[b, r - n_vis]]
//   * dec0_reduce is its adjoint for the qkv gradient:  cat[b * n_vis + r] = full[b, r];
//       cat[B * n_vis + j] = sum over the clips b in which position j is masked of full[b, n_vis + slot_b(j)]   (f32 sum, one rounding)
//     with slot_b(j) from the inverse table dec0_inverse builds (inv[b, j] = slot or -1).  No atomics: one thread per (row, 16-B chunk).
namespace {
// one thread per (clip, position): binary search in the clip's ASCENDING masked list (what mofo_mask_to_indices writes) -- every entry
// of inv is written by exactly one thread, nothing to clear first
__global__ __launch_bounds__(256) void dec0_inverse_kernel(const int* __restrict__ msk_idx, int n_msk, int N, int total, int* __restrict__ inv) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int b = i / N, j = i - b * N;
    const int* row = msk_idx + (size_t)b * n_msk;
    int lo = 0, hi = n_msk;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (row[mid] < j) lo = mid + 1;
        else hi = mid;
    }
    inv[i] = (lo < n_msk && row[lo] == j) ? lo : -1;
}

__global__ __launch_bounds__(256) void dec0_gather_kernel(const bf16_t* __restrict__ cat, int ldcat, const int* __restrict__ msk_idx, int B, int N,
                                                          int n_vis, int cpr, bf16_t* __restrict__ full, int ldfull) {
    const long long total = (long long)B * N * cpr;
    for (long long id = (long long)blockIdx.x * 256 + threadIdx.x; id < total; id += (long long)gridDim.x * 256) {
        const int row = (int)(id / cpr), c = (int)(id - (long long)row * cpr);
        const int b = row / N, r = row - b * N;
        const int src = r < n_vis ? b * n_vis + r : B * n_vis + msk_idx[(size_t)b * (N - n_vis) + (r - n_vis)];
        *(u32x4*)(full + (size_t)row * ldfull + c * 8) = *(const u32x4*)(cat + (size_t)src * ldcat + c * 8);
    }
}

__global__ __launch_bounds__(256) void dec0_reduce_kernel(const bf16_t* __restrict__ full, int ldfull, const int* __restrict__ inv, int B, int N,
                                                          int n_vis, int cpr, bf16_t* __restrict__ cat, int ldcat) {
    const long long vis_chunks = (long long)B * n_vis * cpr, total = vis_chunks + (long long)N * cpr;
    for (long long id = (long long)blockIdx.x * 256 + threadIdx.x; id < total; id += (long long)gridDim.x * 256) {
        if (id < vis_chunks) {
            const int row = (int)(id / cpr), c = (int)(id - (long long)row * cpr);
            const int b = row / n_vis, r = row - b * n_vis;
            *(u32x4*)(cat + (size_t)row * ldcat + c * 8) = *(const u32x4*)(full + ((size_t)b * N + r) * ldfull + c * 8);
        } else {
            const long long q = id - vis_chunks;
            const int j = (int)(q / cpr), c = (int)(q - (long long)j * cpr);
            float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            // eight clips per round: the eight slot loads, then the eight row loads are independent (a thread's latency chain is
            // B / 8 * 2 round trips instead of 2 B); the sum runs in clip order whatever the grouping
            for (int b0 = 0; b0 < B; b0 += 8) {
                int slot[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) slot[u] = b0 + u < B ? inv[(size_t)(b0 + u) * N + j] : -1;
                u32x4 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    v[u] = slot[u] >= 0 ? *(const u32x4*)(full + ((size_t)(b0 + u) * N + n_vis + slot[u]) * ldfull + c * 8) : u32x4{0u, 0u, 0u, 0u};
#pragma unroll
                for (int u = 0; u < 8; ++u)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        acc[2 * e] += bf16lo_to_f32(v[u][e]);
                        acc[2 * e + 1] += bf16hi_to_f32(v[u][e]);
                    }
            }
            *(u32x4*)(cat + ((size_t)B * n_vis + j) * ldcat + c * 8) =
                u32x4{pack_bf16x2(acc[0], acc[1]), pack_bf16x2(acc[2], acc[3]), pack_bf16x2(acc[4], acc[5]), pack_bf16x2(acc[6], acc[7])};
        }
    }
}
}  // namespace

static int dec0_check(const char* who, int B, int N, int n_vis, int W, int lda, int ldb) {
    if (B <= 0 || N <= 0 || n_vis <= 0 || n_vis >= N) MOFO_FAIL(MOFO_EINVAL, "%s: need 0 < n_vis < N and B > 0", who);
    if (W <= 0 || W % 8 || lda % 8 || ldb % 8 || lda < W || ldb < W) MOFO_FAIL(MOFO_EUNSUPPORTED, "%s: row width and leading dims must be multiples of 8", who);
    return MOFO_OK;
}

extern "C" int mofo_dec0_inverse(const int* msk_idx, int B, int N, int n_vis, int* inv, void* stream) {
    if (!msk_idx || !inv) MOFO_FAIL(MOFO_EINVAL, "mofo_dec0_inverse: null pointer");
    if (B <= 0 || N <= 0 || n_vis <= 0 || n_vis >= N) MOFO_FAIL(MOFO_EINVAL, "mofo_dec0_inverse: need 0 < n_vis < N and B > 0");
    const int total = B * N;
    hipLaunchKernelGGL(dec0_inverse_kernel, dim3(ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream, msk_idx, N - n_vis, N, total, inv);
    MOFO_CHECK_LAUNCH("mofo_dec0_inverse");
    return MOFO_OK;
}

extern "C" int mofo_dec0_gather(const void* cat, int ldcat, const int* msk_idx, int B, int N, int n_vis, int W, void* full, int ldfull, void* stream) {
    if (!cat || !msk_idx || !full) MOFO_FAIL(MOFO_EINVAL, "mofo_dec0_gather: null pointer");
    int rc = dec0_check("mofo_dec0_gather", B, N, n_vis, W, ldcat, ldfull);
    if (rc) return rc;
    const long long total = (long long)B * N * (W / 8);
    const int blocks = (int)(total / 256 + 1 < 8192 ? total / 256 + 1 : 8192);
    hipLaunchKernelGGL(dec0_gather_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)cat, ldcat, msk_idx, B, N, n_vis, W / 8,
                       (bf16_t*)full, ldfull);
    MOFO_CHECK_LAUNCH("mofo_dec0_gather");
    return MOFO_OK;
}

extern "C" int mofo_dec0_reduce(const void* full, int ldfull, const int* inv, int B, int N, int n_vis, int W, void* cat, int ldcat, void* stream) {
    if (!full || !inv || !cat) MOFO_FAIL(MOFO_EINVAL, "mofo_dec0_reduce: null pointer");
    int rc = dec0_check("mofo_dec0_reduce", B, N, n_vis, W, ldcat, ldfull);
    if (rc) return rc;
    const long long total = ((long long)B * n_vis + N) * (W / 8);
    const int blocks = (int)(total / 256 + 1 < 8192 ? total / 256 + 1 : 8192);
    hipLaunchKernelGGL(dec0_reduce_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)full, ldfull, inv, B, N, n_vis, W / 8,
                       (bf16_t*)cat, ldcat);
    MOFO_CHECK_LAUNCH("mofo_dec0_reduce");
    return MOFO_OK;
}

extern "C" int mofo_assemble_bwd_blocks(int B, int N) { return B > 0 && N > 0 ? ceil_div(B * N, RB) : 0; }

extern "C" int mofo_assemble_bwd_finalize(const float* partial_ws, int B, int N, int D, float* d_mask_token, void* stream) {
    if (!partial_ws || !d_mask_token || B <= 0 || N <= 0 || D <= 0) MOFO_FAIL(MOFO_EINVAL, "mofo_assemble_bwd_finalize: bad arguments");
    hipLaunchKernelGGL(add_partials_kernel, dim3(ceil_div(D, 64), COL_SLICES), dim3(256), 0, (hipStream_t)stream, partial_ws, ceil_div(B * N, RB), D,
                       d_mask_token);
    MOFO_CHECK_LAUNCH("mofo_assemble_bwd_finalize");
    return MOFO_OK;
}

extern "C" int mofo_assemble_bwd(const void* dx_full, int dx_is_bf16, int B, int N, int n_vis, int D, void* d_e2d,
                                 float* d_mask_token, float* partial_ws, void* stream) {
    if (!dx_full || !d_e2d || (!d_mask_token && !partial_ws)) MOFO_FAIL(MOFO_EINVAL, "mofo_assemble_bwd: null pointer");
    if (!d_mask_token && !(D % 8 == 0 && D <= 512)) MOFO_FAIL(MOFO_EUNSUPPORTED, "mofo_assemble_bwd: the deferred form needs D a multiple of 8, <= 512");
    if (B <= 0 || N <= n_vis || n_vis <= 0 || D <= 0 || D % 4 || D > 1024) MOFO_FAIL(MOFO_EINVAL, "mofo_assemble_bwd: bad sizes");
    const int rows = B * N;
    hipStream_t s = (hipStream_t)stream;
    if (partial_ws && D % 8 == 0 && D <= 512) {
        const int nb = ceil_div(rows, RB);
        if (dx_is_bf16)
            hipLaunchKernelGGL(assemble_bwd_rows_kernel<true>, dim3(nb), dim3(256), 0, s, dx_full, N, n_vis, D, rows, (bf16_t*)d_e2d, partial_ws);
        else
            hipLaunchKernelGGL(assemble_bwd_rows_kernel<false>, dim3(nb), dim3(256), 0, s, dx_full, N, n_vis, D, rows, (bf16_t*)d_e2d, partial_ws);
        MOFO_CHECK_LAUNCH("mofo_assemble_bwd");
        if (!d_mask_token) return MOFO_OK;       // deferred: the caller adds the block partials later (mofo_assemble_bwd_finalize)
        hipLaunchKernelGGL(add_partials_kernel, dim3(ceil_div(D, 64), COL_SLICES), dim3(256), 0, s, (const float*)partial_ws, nb, D, d_mask_token);
        MOFO_CHECK_LAUNCH("mofo_assemble_bwd(partials)");
        return MOFO_OK;
    }
    int ct = 1;
    while (ct * 4 < D) ct *= 2;     // column threads: power of two >= D/4
    if (ct > 256) MOFO_FAIL(MOFO_EUNSUPPORTED, "mofo_assemble_bwd: D=%d too wide", D);
    if (dx_is_bf16)
        hipLaunchKernelGGL(assemble_bwd_kernel<true>, dim3(ceil_div(rows, RB)), dim3(256), 0, s, dx_full, N, n_vis, D,
                           rows, ct, (bf16_t*)d_e2d, d_mask_token);
    else
        hipLaunchKernelGGL(assemble_bwd_kernel<false>, dim3(ceil_div(rows, RB)), dim3(256), 0, s, dx_full, N, n_vis, D,
                           rows, ct, (bf16_t*)d_e2d, d_mask_token);
    MOFO_CHECK_LAUNCH("mofo_assemble_bwd");
    return MOFO_OK;
}

extern "C" int mofo_colsum_bf16(const void* X, int ldx, int M, int N, float* out, void* stream) {
    if (!X || !out) MOFO_FAIL(MOFO_EINVAL, "mofo_colsum_bf16: null pointer");
    if (M <= 0 || N <= 0 || N % 8 || ldx % 8) MOFO_FAIL(MOFO_EUNSUPPORTED, "mofo_colsum_bf16: N and ldx must be multiples of 8");
    const int cb = ceil_div(N, 64);
    int rsplit = ceil_div(1024, cb);
    int rpb = ceil_div(ceil_div(M, rsplit), 32) * 32;
    if (rpb < 256) rpb = 256;
    rsplit = ceil_div(M, rpb);
    hipLaunchKernelGGL(colsum_bf16_kernel, dim3(cb, rsplit), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)X, ldx, M, N, rpb, out);
    MOFO_CHECK_LAUNCH("mofo_colsum_bf16");
    return MOFO_OK;
}
