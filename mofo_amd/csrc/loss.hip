// Fused reconstruction-target builder + MSE loss + d(loss)/d(pred)  (gfx950, HBM-bound).
// Replaces engine_for_pretraining.py:43-63 (un-normalise with the ImageNet constants, patchify `(p0 p1 p2) c`,
// per-(token,channel) standardise over 512 pixels with the UNBIASED variance and 1e-6 added after the sqrt, gather the
// masked tokens) and :27,67 (nn.MSELoss, mean over every element) in ONE pass over the masked 90 % of the clip:
// the [B,1568,1536] f32 target tensor of the reference is never written.
// One wave per masked token: lane e and e+64 each own one float4 (4 pixels of one image-row segment) per channel,
// i.e. 12 consecutive prediction features (4 pixels x 3 channels, channel fastest).
#include "common.h"
#include "../../include/mofo_hip.h"

namespace {

__constant__ float c_mean[3] = {0.485f, 0.456f, 0.406f};  // IMAGENET_DEFAULT_MEAN (engine_for_pretraining.py:45)
__constant__ float c_std[3] = {0.229f, 0.224f, 0.225f};   // IMAGENET_DEFAULT_STD  (:46)

// pixel sources of the target builder: the f32 clip, or the uint8 frame stack normalised on the fly (see tokens.hip)
struct PixF32 {
    const float* clips; int T, H, W;
    static constexpr bool kBytes = false;
    __device__ __forceinline__ f32x4 load4(int b, int c, int t, int y, int x0) const {
        return *(const f32x4*)(clips + ((((size_t)b * 3 + c) * T + t) * H + y) * W + x0);
    }
};
// uint8 frame stack [B][H][W][T*3]: the 6 bytes a tubelet needs of one pixel (2 frames x 3 channels) are contiguous, so
// ONE bounds-checked 8-byte buffer load per pixel serves all six values (byte p0*3 + c of the returned word).
struct PixU8 {
    const uint8_t* frames; int T, H, W;
    static constexpr bool kBytes = true;
    __device__ __forceinline__ uint64_t load6(__amdgpu_buffer_rsrc_t rsrc, int b, int tt, int y, int x) const {
        const uint32_t addr = (uint32_t)(((b * H + y) * W + x) * (T * 3) + tt * 6);
        const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rsrc, (int)(addr & ~3u), 0, 0);
        return (((uint64_t)v[1] << 32) | v[0]) >> ((addr & 3u) * 8);
    }
    __device__ __forceinline__ f32x4 load4(int, int, int, int, int) const { return f32x4{0.f, 0.f, 0.f, 0.f}; }
};

template <class PIX>
__global__ __launch_bounds__(256) void target_mse_kernel(PIX pix, uint32_t frame_bytes,
                                                         const int* __restrict__ msk_idx, int n_msk, int rows,
                                                         const bf16_t* __restrict__ pred, int ldp, int normalize, float gs,
                                                         float* __restrict__ row_loss, bf16_t* __restrict__ dpred, int lddp,
                                                         float* __restrict__ target_out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = blockIdx.x * 4 + wave;
    if (row >= rows) return;
    const int b = row / n_msk;
    const int tok = msk_idx[row];
    const int gw = pix.W >> 4, gh = pix.H >> 4;
    const int tw = tok % gw, th = (tok / gw) % gh, tt = tok / (gw * gh);

    // the prediction row does not depend on the statistics below: issue its loads first so they fly under the pixel math
    u32x2 w[2][3];
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int i = 0; i < 3; ++i) w[q][i] = *(const u32x2*)(pred + (size_t)row * ldp + (q * 64 + lane) * 12 + 4 * i);
    // lane -> image row p1 = lane >> 2, pixels (lane & 3) * 4 + k of the 16x16 patch, in both frames q = p0 of the tubelet
    float u[3][2][4];
    if constexpr (PIX::kBytes) {
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)pix.frames, 0, (int)frame_bytes, 0x00020000);
        uint64_t px[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) px[k] = pix.load6(rsrc, b, tt, th * 16 + (lane >> 2), tw * 16 + (lane & 3) * 4 + k);
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    // ToTorchFormatTensor + GroupNormalize exactly as mofo_ingest_u8, then the target's un-normalise
                    const float x = ((float)(uint32_t)((px[k] >> (8 * (q * 3 + c))) & 0xff) / 255.0f - c_mean[c]) / c_std[c];
                    u[c][q][k] = x * c_std[c] + c_mean[c];
                }
    } else {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int e = q * 64 + lane;          // float4 index inside the channel's 512 pixels
                const int seg = e >> 2, qq = e & 3;   // seg = p0*16 + p1
                const int p0 = seg >> 4, p1 = seg & 15;
                const f32x4 v = pix.load4(b, c, tt * 2 + p0, th * 16 + p1, tw * 16 + qq * 4);
#pragma unroll
                for (int k = 0; k < 4; ++k) u[c][q][k] = v[k] * c_std[c] + c_mean[c];
            }
        }
    }
    if (normalize) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float s = 0.f;
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int k = 0; k < 4; ++k) s += u[c][q][k];
            const float mu = wave_sum(s) * (1.0f / 512.0f);
            float ss = 0.f;
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float d = u[c][q][k] - mu;
                    ss += d * d;
                }
            const float var = wave_sum(ss) * (1.0f / 511.0f);
            const float inv = 1.0f / (sqrtf(var) + 1e-6f);
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int k = 0; k < 4; ++k) u[c][q][k] = (u[c][q][k] - mu) * inv;
        }
    }
    float acc = 0.f;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int e = q * 64 + lane;
        float pv[12];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            pv[4 * i + 0] = bf16lo_to_f32(w[q][i][0]);
            pv[4 * i + 1] = bf16hi_to_f32(w[q][i][0]);
            pv[4 * i + 2] = bf16lo_to_f32(w[q][i][1]);
            pv[4 * i + 3] = bf16hi_to_f32(w[q][i][1]);
        }
        float d[12];
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float t = u[c][q][k];
                const float df = pv[k * 3 + c] - t;
                d[k * 3 + c] = df;
                acc += df * df;
                if (target_out) target_out[(size_t)row * 1536 + e * 12 + k * 3 + c] = t;
            }
        if (dpred) {
            bf16_t* dp = dpred + (size_t)row * lddp + e * 12;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                u32x2 o = {pack_bf16x2(d[4 * i] * gs, d[4 * i + 1] * gs), pack_bf16x2(d[4 * i + 2] * gs, d[4 * i + 3] * gs)};
                *(u32x2*)(dp + 4 * i) = o;
            }
        }
    }
    acc = wave_sum(acc);
    if (lane == 0) row_loss[row] = acc;
}

// deterministic final reduction in double: loss = sum(row_loss) / numel
__global__ __launch_bounds__(1024) void loss_reduce_kernel(const float* __restrict__ row_loss, int rows, double inv_numel,
                                                           float* __restrict__ loss) {
    __shared__ double red[16];
    // four independent partial sums per thread: the loads of a thread's 44 rows (B = 32) are all in flight at once
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    int i = threadIdx.x;
    for (; i + 3 * 1024 < rows; i += 4 * 1024) {
        const float a = row_loss[i], b = row_loss[i + 1024], c = row_loss[i + 2048], d = row_loss[i + 3072];
        s0 += (double)a; s1 += (double)b; s2 += (double)c; s3 += (double)d;
    }
    for (; i < rows; i += 1024) s0 += (double)row_loss[i];
    double s = (s0 + s1) + (s2 + s3);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int i = 0; i < 16; ++i) t += red[i];
        loss[0] = (float)(t * inv_numel);
    }
}

// Reconstruction video of the inference / visualisation path (run_videomae_vis.py:150-180): one wave per token (ALL tokens).
// Every token is standardised per channel over its 512 pixels exactly like the training target; masked tokens take the
// model's prediction instead, and everything is multiplied back by the token's own (std + 1e-6) and mean:
//   rec = (masked ? pred : (u - mu) / sd) * sd + mu,   u = x * imagenet_std + imagenet_mean,   sd = sqrt(var_unbiased) + 1e-6
// `masked_out` = rec on visible tokens, 0 on masked ones (:163-167,180); `ori_out` = u (:152).  The masked tokens of a clip
// are given as the ascending index list the training path already builds; a token finds its prediction row by bisection.
template <bool PRED_BF16>
__global__ __launch_bounds__(256) void reconstruct_kernel(const float* __restrict__ clips, int T, int H, int W, int N,
                                                          const int* __restrict__ msk_idx, int n_msk, int rows,
                                                          const void* __restrict__ pred, int ldp,
                                                          float* __restrict__ rec_out, float* __restrict__ masked_out,
                                                          float* __restrict__ ori_out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = blockIdx.x * 4 + wave;
    if (row >= rows) return;
    const int b = row / N, tok = row - b * N;
    // bisection in msk_idx[b, :] (wave-uniform)
    int lo = 0, hi = n_msk;
    const int* mi = msk_idx + (size_t)b * n_msk;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (mi[mid] < tok) lo = mid + 1;
        else hi = mid;
    }
    const bool is_masked = lo < n_msk && mi[lo] == tok;
    const size_t prow = (size_t)b * n_msk + lo;
    const int gw = W >> 4, gh = H >> 4;
    const int tw = tok % gw, th = (tok / gw) % gh, tt = tok / (gw * gh);
    const size_t cb = (size_t)b * 3 * T * H * W;

    float u[3][2][4];
    size_t off[3][2];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int e = q * 64 + lane;
            const int seg = e >> 2, qq = e & 3;
            const int p0 = seg >> 4, p1 = seg & 15;
            off[c][q] = cb + (((size_t)c * T + (tt * 2 + p0)) * H + (th * 16 + p1)) * W + tw * 16 + qq * 4;
            const f32x4 v = *(const f32x4*)(clips + off[c][q]);
#pragma unroll
            for (int k = 0; k < 4; ++k) u[c][q][k] = v[k] * c_std[c] + c_mean[c];
        }
    }
    float mu[3], sd[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float s = 0.f;
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int k = 0; k < 4; ++k) s += u[c][q][k];
        mu[c] = wave_sum(s) * (1.0f / 512.0f);
        float ss = 0.f;
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float d = u[c][q][k] - mu[c];
                ss += d * d;
            }
        sd[c] = sqrtf(wave_sum(ss) * (1.0f / 511.0f)) + 1e-6f;
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int e = q * 64 + lane;
        float pv[12];
        if (is_masked) {
            if constexpr (PRED_BF16) {
                const bf16_t* pp = (const bf16_t*)pred + prow * ldp + e * 12;
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    const u32x2 w = *(const u32x2*)(pp + 4 * i);
                    pv[4 * i + 0] = bf16lo_to_f32(w[0]);
                    pv[4 * i + 1] = bf16hi_to_f32(w[0]);
                    pv[4 * i + 2] = bf16lo_to_f32(w[1]);
                    pv[4 * i + 3] = bf16hi_to_f32(w[1]);
                }
            } else {
                const float* pp = (const float*)pred + prow * ldp + e * 12;
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    const f32x4 w = *(const f32x4*)(pp + 4 * i);
#pragma unroll
                    for (int k = 0; k < 4; ++k) pv[4 * i + k] = w[k];
                }
            }
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            f32x4 r, m, o;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float z = is_masked ? pv[k * 3 + c] : (u[c][q][k] - mu[c]) / sd[c];
                r[k] = z * sd[c] + mu[c];
                m[k] = is_masked ? 0.f : r[k];
                o[k] = u[c][q][k];
            }
            *(f32x4*)(rec_out + off[c][q]) = r;
            if (masked_out) *(f32x4*)(masked_out + off[c][q]) = m;
            if (ori_out) *(f32x4*)(ori_out + off[c][q]) = o;
        }
    }
}

}  // namespace

extern "C" int mofo_target_mse(const float* clips, int B, int C, int T, int H, int W, int pt, int p, const int* msk_idx,
                               int n_msk, const void* pred, int ldp, int normalize, float grad_scale, float* row_loss,
                               float* loss, void* dpred, int lddp, void* target_out, void* stream) {
    if (!clips || !msk_idx || !pred || !row_loss || !loss) MOFO_FAIL(MOFO_EINVAL, "mofo_target_mse: null pointer");
    if (C != 3 || pt != 2 || p != 16) MOFO_FAIL(MOFO_EUNSUPPORTED, "mofo_target_mse: built for 3 channels, tubelet 2, patch 16 (got %d,%d,%d)", C, pt, p);
    if (B <= 0 || n_msk <= 0 || T % 2 || H % 16 || W % 16 || ldp % 4 || ldp < 1536 || (dpred && (lddp % 4 || lddp < 1536)))
        MOFO_FAIL(MOFO_EINVAL, "mofo_target_mse: bad sizes");
    const int rows = B * n_msk;
    const double numel = (double)rows * 1536.0;
    const float gs = (float)(2.0 * (double)grad_scale / numel);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(target_mse_kernel<PixF32>, dim3(ceil_div(rows, 4)), dim3(256), 0, s, PixF32{clips, T, H, W}, 0u, msk_idx, n_msk, rows,
                       (const bf16_t*)pred, ldp, normalize, gs, row_loss, (bf16_t*)dpred, lddp, (float*)target_out);
    MOFO_CHECK_LAUNCH("mofo_target_mse");
    hipLaunchKernelGGL(loss_reduce_kernel, dim3(1), dim3(1024), 0, s, (const float*)row_loss, rows, 1.0 / numel, loss);
    MOFO_CHECK_LAUNCH("mofo_target_mse(reduce)");
    return MOFO_OK;
}

extern "C" int mofo_reconstruct(const float* clips, int B, int C, int T, int H, int W, int pt, int p, const int* msk_idx, int n_msk,
                                const void* pred, int pred_is_bf16, int ldp, float* rec, float* masked, float* ori, void* stream) {
    if (!clips || !rec || (n_msk > 0 && (!msk_idx || !pred))) MOFO_FAIL(MOFO_EINVAL, "mofo_reconstruct: null pointer");
    if (C != 3 || pt != 2 || p != 16) MOFO_FAIL(MOFO_EUNSUPPORTED, "mofo_reconstruct: built for 3 channels, tubelet 2, patch 16 (got %d,%d,%d)", C, pt, p);
    if (B <= 0 || n_msk < 0 || T % 2 || H % 16 || W % 16 || (n_msk > 0 && (ldp % 4 || ldp < 1536))) MOFO_FAIL(MOFO_EINVAL, "mofo_reconstruct: bad sizes");
    const int N = (T / 2) * (H / 16) * (W / 16);
    if (n_msk > N) MOFO_FAIL(MOFO_EINVAL, "mofo_reconstruct: n_msk %d > tokens %d", n_msk, N);
    const int rows = B * N;
    hipStream_t s = (hipStream_t)stream;
    if (pred_is_bf16)
        hipLaunchKernelGGL(reconstruct_kernel<true>, dim3(ceil_div(rows, 4)), dim3(256), 0, s, clips, T, H, W, N, msk_idx, n_msk, rows, pred, ldp, rec, masked, ori);
    else
        hipLaunchKernelGGL(reconstruct_kernel<false>, dim3(ceil_div(rows, 4)), dim3(256), 0, s, clips, T, H, W, N, msk_idx, n_msk, rows, pred, ldp, rec, masked, ori);
    MOFO_CHECK_LAUNCH("mofo_reconstruct");
    return MOFO_OK;
}

extern "C" int mofo_target_mse_u8(const uint8_t* frames, int B, int T, int H, int W, int pt, int p, const int* msk_idx, int n_msk,
                                  const void* pred, int ldp, int normalize, float grad_scale, float* row_loss, float* loss,
                                  void* dpred, int lddp, void* stream) {
    if (!frames || !msk_idx || !pred || !row_loss || !loss) MOFO_FAIL(MOFO_EINVAL, "mofo_target_mse_u8: null pointer");
    if (pt != 2 || p != 16) MOFO_FAIL(MOFO_EUNSUPPORTED, "mofo_target_mse_u8: built for tubelet 2, patch 16 (got %d,%d)", pt, p);
    if (B <= 0 || n_msk <= 0 || T % 2 || H % 16 || W % 16 || ldp % 4 || ldp < 1536 || (dpred && (lddp % 4 || lddp < 1536)))
        MOFO_FAIL(MOFO_EINVAL, "mofo_target_mse_u8: bad sizes");
    const size_t bytes = (size_t)B * H * W * T * 3;
    if (bytes > 0x7fffffffu) MOFO_FAIL(MOFO_EUNSUPPORTED, "mofo_target_mse_u8: frame stack of %zu bytes exceeds one 2 GiB buffer descriptor", bytes);
    const int rows = B * n_msk;
    const double numel = (double)rows * 1536.0;
    const float gs = (float)(2.0 * (double)grad_scale / numel);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(target_mse_kernel<PixU8>, dim3(ceil_div(rows, 4)), dim3(256), 0, s, PixU8{frames, T, H, W}, (uint32_t)bytes, msk_idx, n_msk, rows,
                       (const bf16_t*)pred, ldp, normalize, gs, row_loss, (bf16_t*)dpred, lddp, (float*)nullptr);
    MOFO_CHECK_LAUNCH("mofo_target_mse_u8");
    hipLaunchKernelGGL(loss_reduce_kernel, dim3(1), dim3(1024), 0, s, (const float*)row_loss, rows, 1.0 / numel, loss);
    MOFO_CHECK_LAUNCH("mofo_target_mse_u8(reduce)");
    return MOFO_OK;
}
