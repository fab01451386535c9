// Index-driven HBM kernels that glue the GEMMs of the trunk together: multimodal splice (token-embedding gather +
// image-feature rows), patch / conv im2col in NHWC, SAM window partition / unpartition, decomposed rel-pos tables,
// adapter channel gate, CLIP class-token assembly.  All 16 B per lane where the layout allows.
//
// Reference sites: medplib_arch.py:296-527 (prepare_inputs_labels_for_multimodal), image_encoder.py:299-345
// (window_partition/unpartition), :381-421 (add_decomposed_rel_pos), :18-56 (Adapter_Layer), :424-455 (PatchEmbed),
// HF CLIPVisionEmbeddings.
#include "common.h"

namespace {

// ---- multimodal splice: out[r, :] = src_code[r] >= 0 ? embed[src_code[r]] : (src_code[r] == PAD ? 0 : feats[-1 - src_code[r]])
constexpr int64_t SPLICE_PAD = INT64_MIN;
__global__ void splice_rows_kernel(const bf16_t* __restrict__ embed, const bf16_t* __restrict__ feats,
                                   const int64_t* __restrict__ src, bf16_t* __restrict__ out, int64_t rows, int dim) {
  const int per_row = dim / 8;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= rows * per_row) return;
  const int64_t r = idx / per_row;
  const int c = (int)(idx % per_row) * 8;
  const int64_t code = src[r];
  bf16x8 v;
  if (code == SPLICE_PAD) {
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (bf16_t)0.f;
  } else if (code >= 0) {
    v = *reinterpret_cast<const bf16x8*>(embed + code * dim + c);
  } else {
    v = *reinterpret_cast<const bf16x8*>(feats + (-1 - code) * dim + c);
  }
  *reinterpret_cast<bf16x8*>(out + r * dim + c) = v;
}

// ---- patch embedding im2col (non-overlapping p x p patches, NCHW fp32/bf16 image -> [B*gh*gw, Kpad] bf16,
//      column order (c, py, px) = Conv2d weight.view(out, -1) order; columns >= 3*p*p are zero padding)
template <typename TIN>
__global__ void patch_im2col_kernel(const TIN* __restrict__ img, bf16_t* __restrict__ out, int B, int C, int H, int W, int p,
                                    int Kpad) {
  const int gh = H / p, gw = W / p;
  const int64_t total = (int64_t)B * gh * gw * Kpad;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int k = (int)(idx % Kpad);
  const int64_t t = idx / Kpad;
  const int gx = (int)(t % gw), gy = (int)((t / gw) % gh), b = (int)(t / ((int64_t)gw * gh));
  float v = 0.f;
  if (k < C * p * p) {
    const int c = k / (p * p), py = (k / p) % p, px = k % p;
    v = ld_f(img, (((int64_t)b * C + c) * H + gy * p + py) * W + gx * p + px);
  }
  out[idx] = (bf16_t)v;
}

// ---- generic NHWC tap-gather im2col: out[(b,oy,ox), t*C + c] = x[b, oy*sy + dy[t], ox*sx + dx[t], c] (0 if outside)
struct Taps { int n; int dy[16]; int dx[16]; };
__global__ void im2col_nhwc_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ out, int B, int H, int W, int C, int OH,
                                   int OW, int sy, int sx, Taps taps) {
  const int per_pix = taps.n * (C / 8);
  const int64_t total = (int64_t)B * OH * OW * per_pix;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int q = (int)(idx % per_pix);
  const int64_t pix = idx / per_pix;
  const int t = q / (C / 8), c = (q % (C / 8)) * 8;
  const int ox = (int)(pix % OW), oy = (int)((pix / OW) % OH), b = (int)(pix / ((int64_t)OW * OH));
  const int iy = oy * sy + taps.dy[t], ix = ox * sx + taps.dx[t];
  bf16x8 v;
  if (iy >= 0 && iy < H && ix >= 0 && ix < W) {
    v = *reinterpret_cast<const bf16x8*>(x + (((int64_t)b * H + iy) * W + ix) * C + c);
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (bf16_t)0.f;
  }
  *reinterpret_cast<bf16x8*>(out + pix * ((int64_t)taps.n * C) + (int64_t)t * C + c) = v;
}

// ---- strided scatter with optional add: dst[b, oy*sy+py, ox*sx+px, :] = src[(b,oy,ox), :] (+ add[same dst index])
__global__ void scatter_parity_kernel(const bf16_t* __restrict__ src, const bf16_t* __restrict__ add, bf16_t* __restrict__ dst,
                                      int B, int OH, int OW, int C, int sy, int sx, int py, int px, int DH, int DW) {
  const int per_pix = C / 8;
  const int64_t total = (int64_t)B * OH * OW * per_pix;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int c = (int)(idx % per_pix) * 8;
  const int64_t pix = idx / per_pix;
  const int ox = (int)(pix % OW), oy = (int)((pix / OW) % OH), b = (int)(pix / ((int64_t)OW * OH));
  const int64_t d = (((int64_t)b * DH + oy * sy + py) * DW + ox * sx + px) * C + c;
  bf16x8 v = *reinterpret_cast<const bf16x8*>(src + pix * C + c);
  if (add) {
    const bf16x8 a = *reinterpret_cast<const bf16x8*>(add + d);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (bf16_t)((float)v[j] + (float)a[j]);
  }
  *reinterpret_cast<bf16x8*>(dst + d) = v;
}

// ---- SAM window partition (zero pad) and unpartition (+ shortcut add)
__global__ void window_partition_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ win, int B, int H, int W, int C,
                                        int ws, int nwy, int nwx) {
  const int per_tok = C / 8;
  const int64_t total = (int64_t)B * nwy * nwx * ws * ws * per_tok;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int c = (int)(idx % per_tok) * 8;
  int64_t t = idx / per_tok;
  const int wx_in = (int)(t % ws); t /= ws;
  const int wy_in = (int)(t % ws); t /= ws;
  const int wxi = (int)(t % nwx); t /= nwx;
  const int wyi = (int)(t % nwy);
  const int b = (int)(t / nwy);
  const int y = wyi * ws + wy_in, xx = wxi * ws + wx_in;
  bf16x8 v;
  if (y < H && xx < W) v = *reinterpret_cast<const bf16x8*>(x + (((int64_t)b * H + y) * W + xx) * C + c);
  else {
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (bf16_t)0.f;
  }
  *reinterpret_cast<bf16x8*>(win + (idx / per_tok) * C + c) = v;
}
__global__ void window_unpartition_add_kernel(const bf16_t* __restrict__ win, const bf16_t* __restrict__ shortcut,
                                              bf16_t* __restrict__ out, int B, int H, int W, int C, int ws, int nwy, int nwx) {
  const int per_tok = C / 8;
  const int64_t total = (int64_t)B * H * W * per_tok;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int c = (int)(idx % per_tok) * 8;
  const int64_t tok = idx / per_tok;
  const int xx = (int)(tok % W), y = (int)((tok / W) % H), b = (int)(tok / ((int64_t)W * H));
  const int64_t wtok = ((((int64_t)b * nwy + y / ws) * nwx + xx / ws) * ws + y % ws) * ws + xx % ws;
  const bf16x8 v = *reinterpret_cast<const bf16x8*>(win + wtok * C + c);
  const bf16x8 s = *reinterpret_cast<const bf16x8*>(shortcut + tok * C + c);
  bf16x8 o;
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] = (bf16_t)((float)v[j] + (float)s[j]);
  *reinterpret_cast<bf16x8*>(out + tok * C + c) = o;
}

// ---- decomposed rel-pos tables: rel_h[bh, q, kh] = sum_c q[b, q, h, c] * rel_pos_h[(qy - kh) + (hh-1), c] ; same for w
//      q lives in a fused [Bw, S, 3*H*D] qkv buffer (D == 64).  A block takes 8 (bh, q) pairs: both tables ((2*hh-1) + (2*ww-1) rows of
//      64 floats, row stride 65 so the 32 threads of a pair read 32 different banks) and the 8 query vectors sit in LDS; thread
//      (pair, k) owns one output and walks the 64 channels in order.  (One wave per pair with a wave reduction per output took
//      117 us per SAM layer.)
constexpr int RP_PAIRS = 8, RP_MAXK = 32, RP_STRIDE = 65, RP_ITERS = 8;
__global__ __launch_bounds__(256) void relpos_tables_kernel(const bf16_t* __restrict__ qkv, int64_t ld, const float* __restrict__ rph,
                                                            const float* __restrict__ rpw, float* __restrict__ rel_h,
                                                            float* __restrict__ rel_w, int Bw, int H, int hh, int ww) {
  extern __shared__ float rp_smem[];
  const int S = hh * ww, D = 64;
  const int nh = 2 * hh - 1, nw = 2 * ww - 1;
  float* th = rp_smem;                               // [nh][65]
  float* tw = th + nh * RP_STRIDE;                   // [nw][65]
  float* qs = tw + nw * RP_STRIDE;                   // [8][64]
  for (int i = threadIdx.x; i < nh * D; i += 256) th[(i / D) * RP_STRIDE + (i % D)] = rph[i];
  for (int i = threadIdx.x; i < nw * D; i += 256) tw[(i / D) * RP_STRIDE + (i % D)] = rpw[i];
  const int64_t total = (int64_t)Bw * H * S;
  const int p = threadIdx.x >> 5, k = threadIdx.x & 31;
  for (int it = 0; it < RP_ITERS; ++it) {            // the staged tables serve RP_ITERS groups of 8 pairs
    const int64_t pair0 = ((int64_t)blockIdx.x * RP_ITERS + it) * RP_PAIRS;
    if (pair0 >= total) break;
    __syncthreads();                                 // tables staged / previous group's query vectors consumed
    for (int i = threadIdx.x; i < RP_PAIRS * D; i += 256) {
      const int64_t wq = pair0 + i / D;
      float v = 0.f;
      if (wq < total) {
        const int q = (int)(wq % S), bh = (int)(wq / S), b = bh / H, h = bh % H;
        v = (float)qkv[((int64_t)b * S + q) * ld + (int64_t)h * D + (i % D)];
      }
      qs[i] = v;
    }
    __syncthreads();
    const int64_t wq = pair0 + p;
    if (wq < total && k < hh + ww) {
      const int q = (int)(wq % S);
      const int qy = q / ww, qx = q % ww;
      const float* qv = qs + p * D;
      const float* row = k < hh ? th + (qy - k + hh - 1) * RP_STRIDE : tw + (qx - (k - hh) + ww - 1) * RP_STRIDE;
      float acc = 0.f;
#pragma unroll 16
      for (int c = 0; c < D; ++c) acc = fmaf(qv[c], row[c], acc);
      if (k < hh) rel_h[wq * hh + k] = acc;
      else rel_w[wq * ww + (k - hh)] = acc;
    }
  }
}

// ---- adapter: per-image token mean (bf16 -> fp32) and channel gate.  One block per (image, 64 channels); the tokens are striped
//      over 16 waves and combined in a fixed order (a thread per (image, channel) walking all tokens took 61 us on 6144 threads)
__global__ __launch_bounds__(1024) void token_mean_kernel(const bf16_t* __restrict__ x, float* __restrict__ out, int B, int T, int C) {
  __shared__ float part[16][64];
  const int cb = (C + 63) / 64;
  const int b = blockIdx.x / cb;
  const int c = (blockIdx.x % cb) * 64 + (threadIdx.x & 63);
  const int w = threadIdx.x >> 6;
  float s0 = 0.f, s1 = 0.f;
  if (c < C) {
    int t = w;
    for (; t + 16 < T; t += 32) { s0 += (float)x[((int64_t)b * T + t) * C + c]; s1 += (float)x[((int64_t)b * T + t + 16) * C + c]; }
    for (; t < T; t += 16) s0 += (float)x[((int64_t)b * T + t) * C + c];
  }
  part[w][threadIdx.x & 63] = s0 + s1;
  __syncthreads();
  if (w == 0 && c < C) {
    float tsum = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) tsum += part[k][threadIdx.x];
    out[(int64_t)b * C + c] = tsum / (float)T;
  }
}
__global__ void scale_channels_kernel(const bf16_t* __restrict__ x, const float* __restrict__ gate, bf16_t* __restrict__ y, int B,
                                      int T, int C) {
  const int per_tok = C / 8;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)B * T * per_tok) return;
  const int c = (int)(idx % per_tok) * 8;
  const int64_t tok = idx / per_tok;
  const int b = (int)(tok / T);
  const bf16x8 v = *reinterpret_cast<const bf16x8*>(x + tok * C + c);
  bf16x8 o;
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] = (bf16_t)(gate[(int64_t)b * C + c + j] * (float)v[j]);
  *reinterpret_cast<bf16x8*>(y + tok * C + c) = o;
}

// ---- CLIP embeddings: out[b, 0, :] = cls + pos[0]; out[b, 1+i, :] = patch[b, i, :] + pos[1+i]
__global__ void clip_embed_kernel(const bf16_t* __restrict__ patch, const bf16_t* __restrict__ cls, const bf16_t* __restrict__ pos,
                                  bf16_t* __restrict__ out, int B, int NP, int C) {
  const int per_tok = C / 8;
  const int64_t total = (int64_t)B * (NP + 1) * per_tok;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int c = (int)(idx % per_tok) * 8;
  const int64_t tok = idx / per_tok;
  const int i = (int)(tok % (NP + 1)), b = (int)(tok / (NP + 1));
  const bf16x8 a = (i == 0) ? *reinterpret_cast<const bf16x8*>(cls + c)
                            : *reinterpret_cast<const bf16x8*>(patch + ((int64_t)b * NP + i - 1) * C + c);
  const bf16x8 p = *reinterpret_cast<const bf16x8*>(pos + (int64_t)i * C + c);
  bf16x8 o;
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] = (bf16_t)((float)a[j] + (float)p[j]);
  *reinterpret_cast<bf16x8*>(out + tok * C + c) = o;
}

// ---- strided row copy: dst[r, :] = src[map(r), :] with src row = (r / rows_per_batch) * src_batch_rows + r % rows_per_batch + src_row0
//      (drop the CLS token: clip_encoder.py:33-34)
__global__ void copy_rows_kernel(const bf16_t* __restrict__ src, bf16_t* __restrict__ dst, int64_t rows, int dim, int rows_per_batch,
                                 int src_batch_rows, int src_row0) {
  const int per_row = dim / 8;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= rows * per_row) return;
  const int64_t r = idx / per_row;
  const int c = (int)(idx % per_row) * 8;
  const int64_t sr = (r / rows_per_batch) * src_batch_rows + r % rows_per_batch + src_row0;
  *reinterpret_cast<bf16x8*>(dst + r * dim + c) = *reinterpret_cast<const bf16x8*>(src + sr * dim + c);
}


// ---- nn.AdaptiveAvgPool1d over the token axis of a token-major tensor: out[n, i, c] = mean_{t in [floor(i*Lin/Lout),
//      ceil((i+1)*Lin/Lout))} x[n, t, c]  (TokenCompressor 576 -> 256, MaskTokenEncoder 441 -> 64; medplib_arch.py:67-108).
//      fp32 accumulation, one bf16 rounding.
__global__ void adaptive_avgpool_tokens_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ out, int n, int Lin, int Lout, int C) {
  const int per_row = C / 8;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)n * Lout * per_row) return;
  const int c = (int)(idx % per_row) * 8;
  const int i = (int)((idx / per_row) % Lout);
  const int b = (int)(idx / ((int64_t)per_row * Lout));
  const int t0 = (int)(((int64_t)i * Lin) / Lout);
  const int t1 = (int)((((int64_t)(i + 1)) * Lin + Lout - 1) / Lout);
  float acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = 0.f;
  for (int t = t0; t < t1; ++t) {
    const bf16x8 v = *reinterpret_cast<const bf16x8*>(x + ((int64_t)b * Lin + t) * C + c);
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] += (float)v[j];
  }
  const float inv = 1.f / (float)(t1 - t0);
  bf16x8 o;
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] = (bf16_t)(acc[j] * inv);
  *reinterpret_cast<bf16x8*>(out + ((int64_t)b * Lout + i) * C + c) = o;
}

// ---- first layer of MaskTokenEncoder: Conv2d(1, CO, k3, s2, p1) + GELU on a single-channel mask image, NHWC bf16 output
//      (medplib_arch.py:84-85).  One thread per (pixel, 8 output channels): 9 taps x 8 FMAs, HBM-bound on the output.
template <typename TIN>
__global__ void conv3x3s2_c1_gelu_kernel(const TIN* __restrict__ img, const float* __restrict__ w, const float* __restrict__ bias,
                                         bf16_t* __restrict__ out, int n, int H, int W, int OH, int OW, int CO) {
  const int per_pix = CO / 8;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)n * OH * OW * per_pix) return;
  const int co = (int)(idx % per_pix) * 8;
  const int64_t pix = idx / per_pix;
  const int ox = (int)(pix % OW), oy = (int)((pix / OW) % OH), b = (int)(pix / ((int64_t)OW * OH));
  float acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = bias[co + j];
#pragma unroll
  for (int ky = 0; ky < 3; ++ky)
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const int iy = oy * 2 - 1 + ky, ix = ox * 2 - 1 + kx;
      if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
      // the reference casts the mask to the model dtype first (medplib_arch.py:103-104)
      const float v = (float)(bf16_t)ld_f(img, ((int64_t)b * H + iy) * W + ix);
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] = fmaf(v, w[(co + j) * 9 + ky * 3 + kx], acc[j]);
    }
  bf16x8 o;
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] = (bf16_t)gelu_erf(acc[j]);
  *reinterpret_cast<bf16x8*>(out + pix * CO + co) = o;
}


// ---- region prompts: extract_region_feature (medplib_arch.py:580-613).  For mask m: mean over its sampled points of the bilinear
//      (grid_sample, align_corners=True, zero padding) read-out of the sample's [h, w, C] feature map at normalised (x, y).
//      Mirrors the reference's rounding points: the sample is computed in fp32, rounded to the model dtype (bf16), the mean is
//      accumulated in fp32 and rounded once.  One thread per (mask, 8 channels).
__global__ void region_point_mean_kernel(const bf16_t* __restrict__ fmap, const float* __restrict__ xy, const int64_t* __restrict__ offsets,
                                         const int* __restrict__ map_index, bf16_t* __restrict__ out, int n_masks, int h, int w, int C) {
  const int per_row = C / 8;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)n_masks * per_row) return;
  const int c = (int)(idx % per_row) * 8;
  const int m = (int)(idx / per_row);
  const bf16_t* fm = fmap + (int64_t)map_index[m] * h * w * C;
  const int64_t p0 = offsets[m], p1 = offsets[m + 1];
  float acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = 0.f;
  for (int64_t p = p0; p < p1; ++p) {
    // grid = 2 * coord - 1; align_corners=True unnormalisation ((g + 1) / 2) * (size - 1)
    const float gx = 2.f * xy[2 * p] - 1.f, gy = 2.f * xy[2 * p + 1] - 1.f;
    const float ix = ((gx + 1.f) * 0.5f) * (float)(w - 1), iy = ((gy + 1.f) * 0.5f) * (float)(h - 1);
    const float fx = floorf(ix), fy = floorf(iy);
    const int x0 = (int)fx, y0 = (int)fy, x1 = x0 + 1, y1 = y0 + 1;
    const float wx1 = ix - fx, wy1 = iy - fy, wx0 = 1.f - wx1, wy0 = 1.f - wy1;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = 0.f;
    auto corner = [&](int yy, int xx, float wt) {
      if (yy < 0 || yy >= h || xx < 0 || xx >= w) return;
      const bf16x8 t = *reinterpret_cast<const bf16x8*>(fm + ((int64_t)yy * w + xx) * C + c);
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] += (float)t[j] * wt;
    };
    corner(y0, x0, wy0 * wx0); corner(y0, x1, wy0 * wx1); corner(y1, x0, wy1 * wx0); corner(y1, x1, wy1 * wx1);
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] += (float)(bf16_t)v[j];
  }
  const float inv = p1 > p0 ? 1.f / (float)(p1 - p0) : 0.f;        // empty mask: mean of nothing = NaN -> nan_to_num -> 0
  bf16x8 o;
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] = (bf16_t)(acc[j] * inv);
  *reinterpret_cast<bf16x8*>(out + (int64_t)m * C + c) = o;
}

}  // namespace

#define GRID1D(n) dim3((unsigned)mp_cdiv((n), 256)), dim3(256), 0, stream

extern "C" int mp_splice_rows_bf16(const void* embed, const void* feats, const int64_t* src_code, void* out, int64_t rows, int dim,
                                   hipStream_t stream) {
  MP_REQUIRE(dim % 8 == 0, MP_ERR_SHAPE, "mp_splice_rows_bf16: dim must be a multiple of 8");
  const int64_t n = rows * (dim / 8);
  if (n == 0) return MP_OK;
  hipLaunchKernelGGL(splice_rows_kernel, GRID1D(n), (const bf16_t*)embed, (const bf16_t*)feats, src_code, (bf16_t*)out, rows, dim);
  return mp_check_launch("mp_splice_rows_bf16");
}

extern "C" int mp_patch_im2col(const void* img, int img_dtype, void* out, int B, int C, int H, int W, int patch, int k_padded,
                               hipStream_t stream) {
  MP_REQUIRE(H % patch == 0 && W % patch == 0 && k_padded >= C * patch * patch, MP_ERR_SHAPE, "mp_patch_im2col: bad shape");
  const int64_t n = (int64_t)B * (H / patch) * (W / patch) * k_padded;
  if (n == 0) return MP_OK;
  if (img_dtype == MP_F32)
    hipLaunchKernelGGL(patch_im2col_kernel<float>, GRID1D(n), (const float*)img, (bf16_t*)out, B, C, H, W, patch, k_padded);
  else if (img_dtype == MP_BF16)
    hipLaunchKernelGGL(patch_im2col_kernel<bf16_t>, GRID1D(n), (const bf16_t*)img, (bf16_t*)out, B, C, H, W, patch, k_padded);
  else MP_REQUIRE(false, MP_ERR_DTYPE, "mp_patch_im2col: bad dtype");
  return mp_check_launch("mp_patch_im2col");
}

extern "C" int mp_im2col_nhwc_bf16(const void* x, void* out, int B, int H, int W, int C, int OH, int OW, int stride_y, int stride_x,
                                   int n_taps, const int* dy, const int* dx, hipStream_t stream) {
  MP_REQUIRE(C % 8 == 0 && n_taps >= 1 && n_taps <= 16, MP_ERR_SHAPE, "mp_im2col_nhwc_bf16: bad shape");
  Taps t{};
  t.n = n_taps;
  for (int i = 0; i < n_taps; ++i) { t.dy[i] = dy[i]; t.dx[i] = dx[i]; }   // dy/dx are HOST arrays
  const int64_t n = (int64_t)B * OH * OW * n_taps * (C / 8);
  if (n == 0) return MP_OK;
  hipLaunchKernelGGL(im2col_nhwc_kernel, GRID1D(n), (const bf16_t*)x, (bf16_t*)out, B, H, W, C, OH, OW, stride_y, stride_x, t);
  return mp_check_launch("mp_im2col_nhwc_bf16");
}

extern "C" int mp_scatter_parity_bf16(const void* src, const void* add, void* dst, int B, int OH, int OW, int C, int sy, int sx,
                                      int py, int px, int DH, int DW, hipStream_t stream) {
  MP_REQUIRE(C % 8 == 0, MP_ERR_SHAPE, "mp_scatter_parity_bf16: bad shape");
  const int64_t n = (int64_t)B * OH * OW * (C / 8);
  if (n == 0) return MP_OK;
  hipLaunchKernelGGL(scatter_parity_kernel, GRID1D(n), (const bf16_t*)src, (const bf16_t*)add, (bf16_t*)dst, B, OH, OW, C, sy, sx,
                     py, px, DH, DW);
  return mp_check_launch("mp_scatter_parity_bf16");
}

extern "C" int mp_window_partition_bf16(const void* x, void* win, int B, int H, int W, int C, int ws, hipStream_t stream) {
  MP_REQUIRE(C % 8 == 0 && ws > 0, MP_ERR_SHAPE, "mp_window_partition_bf16: bad shape");
  const int nwy = (H + ws - 1) / ws, nwx = (W + ws - 1) / ws;
  const int64_t n = (int64_t)B * nwy * nwx * ws * ws * (C / 8);
  if (n == 0) return MP_OK;
  hipLaunchKernelGGL(window_partition_kernel, GRID1D(n), (const bf16_t*)x, (bf16_t*)win, B, H, W, C, ws, nwy, nwx);
  return mp_check_launch("mp_window_partition_bf16");
}

extern "C" int mp_window_unpartition_add_bf16(const void* win, const void* shortcut, void* out, int B, int H, int W, int C, int ws,
                                              hipStream_t stream) {
  MP_REQUIRE(C % 8 == 0 && ws > 0, MP_ERR_SHAPE, "mp_window_unpartition_add_bf16: bad shape");
  const int nwy = (H + ws - 1) / ws, nwx = (W + ws - 1) / ws;
  const int64_t n = (int64_t)B * H * W * (C / 8);
  if (n == 0) return MP_OK;
  hipLaunchKernelGGL(window_unpartition_add_kernel, GRID1D(n), (const bf16_t*)win, (const bf16_t*)shortcut, (bf16_t*)out, B, H, W,
                     C, ws, nwy, nwx);
  return mp_check_launch("mp_window_unpartition_add_bf16");
}

extern "C" int mp_relpos_tables_bf16(const void* qkv, int64_t ld, const float* rel_pos_h, const float* rel_pos_w, float* rel_h,
                                     float* rel_w, int Bw, int heads, int hh, int ww, int head_dim, hipStream_t stream) {
  MP_REQUIRE(head_dim == 64, MP_ERR_SHAPE, "mp_relpos_tables_bf16: head_dim must be 64");
  MP_REQUIRE(hh > 0 && ww > 0 && hh + ww <= RP_MAXK, MP_ERR_SHAPE, "mp_relpos_tables_bf16: hh + ww must be <= %d", RP_MAXK);
  const int64_t pairs = (int64_t)Bw * heads * hh * ww;
  if (pairs == 0) return MP_OK;
  const size_t smem = (size_t)(((2 * hh - 1) + (2 * ww - 1)) * RP_STRIDE + RP_PAIRS * 64) * sizeof(float);
  hipLaunchKernelGGL(relpos_tables_kernel, dim3((unsigned)mp_cdiv(pairs, RP_PAIRS * RP_ITERS)), dim3(256), smem, stream, (const bf16_t*)qkv, ld,
                     rel_pos_h, rel_pos_w, rel_h, rel_w, Bw, heads, hh, ww);
  return mp_check_launch("mp_relpos_tables_bf16");
}

extern "C" int mp_token_mean_bf16(const void* x, float* out, int B, int T, int C, hipStream_t stream) {
  const int64_t n = (int64_t)B * C;
  if (n == 0) return MP_OK;
  hipLaunchKernelGGL(token_mean_kernel, dim3((unsigned)(B * ((C + 63) / 64))), dim3(1024), 0, stream, (const bf16_t*)x, out, B, T, C);
  return mp_check_launch("mp_token_mean_bf16");
}

extern "C" int mp_scale_channels_bf16(const void* x, const float* gate, void* y, int B, int T, int C, hipStream_t stream) {
  MP_REQUIRE(C % 8 == 0, MP_ERR_SHAPE, "mp_scale_channels_bf16: bad shape");
  const int64_t n = (int64_t)B * T * (C / 8);
  if (n == 0) return MP_OK;
  hipLaunchKernelGGL(scale_channels_kernel, GRID1D(n), (const bf16_t*)x, gate, (bf16_t*)y, B, T, C);
  return mp_check_launch("mp_scale_channels_bf16");
}

extern "C" int mp_clip_embed_bf16(const void* patch, const void* cls, const void* pos, void* out, int B, int n_patches, int C,
                                  hipStream_t stream) {
  MP_REQUIRE(C % 8 == 0, MP_ERR_SHAPE, "mp_clip_embed_bf16: bad shape");
  const int64_t n = (int64_t)B * (n_patches + 1) * (C / 8);
  if (n == 0) return MP_OK;
  hipLaunchKernelGGL(clip_embed_kernel, GRID1D(n), (const bf16_t*)patch, (const bf16_t*)cls, (const bf16_t*)pos, (bf16_t*)out, B,
                     n_patches, C);
  return mp_check_launch("mp_clip_embed_bf16");
}

extern "C" int mp_copy_rows_bf16(const void* src, void* dst, int64_t rows, int dim, int rows_per_batch, int src_batch_rows,
                                 int src_row0, hipStream_t stream) {
  MP_REQUIRE(dim % 8 == 0 && rows_per_batch > 0, MP_ERR_SHAPE, "mp_copy_rows_bf16: bad shape");
  const int64_t n = rows * (dim / 8);
  if (n == 0) return MP_OK;
  hipLaunchKernelGGL(copy_rows_kernel, GRID1D(n), (const bf16_t*)src, (bf16_t*)dst, rows, dim, rows_per_batch, src_batch_rows,
                     src_row0);
  return mp_check_launch("mp_copy_rows_bf16");
}

extern "C" int mp_adaptive_avgpool_tokens_bf16(const void* x, void* out, int n, int len_in, int len_out, int C, hipStream_t stream) {
  MP_REQUIRE(n > 0 && len_in > 0 && len_out > 0 && C % 8 == 0, MP_ERR_SHAPE, "mp_adaptive_avgpool_tokens_bf16: bad shape");
  const int64_t total = (int64_t)n * len_out * (C / 8);
  hipLaunchKernelGGL(adaptive_avgpool_tokens_kernel, dim3((unsigned)mp_cdiv(total, 256)), dim3(256), 0, stream, (const bf16_t*)x,
                     (bf16_t*)out, n, len_in, len_out, C);
  return mp_check_launch("mp_adaptive_avgpool_tokens_bf16");
}

extern "C" int mp_conv3x3s2_c1_gelu_bf16(const void* img, int img_dtype, const float* w, const float* bias, void* out, int n, int H,
                                         int W, int CO, hipStream_t stream) {
  MP_REQUIRE(n > 0 && H > 0 && W > 0 && CO % 8 == 0, MP_ERR_SHAPE, "mp_conv3x3s2_c1_gelu_bf16: bad shape");
  MP_REQUIRE(img_dtype == MP_BF16 || img_dtype == MP_F32, MP_ERR_DTYPE, "mp_conv3x3s2_c1_gelu_bf16: bad image dtype");
  const int OH = (H + 2 - 3) / 2 + 1, OW = (W + 2 - 3) / 2 + 1;
  const int64_t total = (int64_t)n * OH * OW * (CO / 8);
  const dim3 grid((unsigned)mp_cdiv(total, 256)), blk(256);
  if (img_dtype == MP_F32)
    hipLaunchKernelGGL(conv3x3s2_c1_gelu_kernel<float>, grid, blk, 0, stream, (const float*)img, w, bias, (bf16_t*)out, n, H, W, OH, OW, CO);
  else
    hipLaunchKernelGGL(conv3x3s2_c1_gelu_kernel<bf16_t>, grid, blk, 0, stream, (const bf16_t*)img, w, bias, (bf16_t*)out, n, H, W, OH, OW, CO);
  return mp_check_launch("mp_conv3x3s2_c1_gelu_bf16");
}

extern "C" int mp_region_point_mean_bf16(const void* fmap, const float* xy, const int64_t* offsets, const int* map_index, void* out,
                                         int n_masks, int h, int w, int C, hipStream_t stream) {
  MP_REQUIRE(n_masks >= 0 && h > 0 && w > 0 && C % 8 == 0, MP_ERR_SHAPE, "mp_region_point_mean_bf16: bad shape");
  if (n_masks == 0) return MP_OK;
  const int64_t total = (int64_t)n_masks * (C / 8);
  hipLaunchKernelGGL(region_point_mean_kernel, dim3((unsigned)mp_cdiv(total, 256)), dim3(256), 0, stream, (const bf16_t*)fmap, xy,
                     offsets, map_index, (bf16_t*)out, n_masks, h, w, C);
  return mp_check_launch("mp_region_point_mean_bf16");
}

// ---- rows of the LAST decoder layer that something reads (round 5: the MLP of that layer runs on those rows only) ----
namespace {
// out[r, :] = src[idx[r], :] (bf16, 16-byte pieces); gather: out compact / src strided; scatter: the reverse
__global__ void gather_rows_bf16_kernel(const bf16_t* __restrict__ src, int64_t lds_, const int64_t* __restrict__ idx, bf16_t* __restrict__ out,
                                        int64_t ldo, int64_t n, int dim8, int scatter) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * dim8) return;
  const int64_t r = i / dim8;
  const int c = (int)(i - r * dim8) * 8;
  const int64_t row = idx[r];
  if (scatter) *reinterpret_cast<bf16x8*>(out + row * ldo + c) = *reinterpret_cast<const bf16x8*>(src + r * lds_ + c);
  else *reinterpret_cast<bf16x8*>(out + r * ldo + c) = *reinterpret_cast<const bf16x8*>(src + row * lds_ + c);
}
// Per expert: the slots (in slot order) whose token is marked in `needed` -> compacted slot_token_out[e, 0 .. kept_out[e])
__global__ __launch_bounds__(1024) void moe_filter_slots_kernel(const int* __restrict__ slot_token, const int* __restrict__ kept,
                                                               const uint8_t* __restrict__ needed, int* __restrict__ slot_token_out,
                                                               int* __restrict__ kept_out, int cap) {
  __shared__ int wtot[16];
  __shared__ int s_base;
  const int e = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int n = min(kept[e], cap);
  if (tid == 0) s_base = 0;
  __syncthreads();
  for (int b0 = 0; b0 < n; b0 += 1024) {
    const int s = b0 + tid;
    const int tok = s < n ? slot_token[(int64_t)e * cap + s] : -1;
    const bool f = tok >= 0 && needed[tok] != 0;
    const unsigned long long m = __ballot(f);
    const int before = __popcll(m & ((1ull << lane) - 1ull));
    if (lane == 0) wtot[wv] = __popcll(m);
    __syncthreads();
    int off = s_base;
    for (int w = 0; w < wv; ++w) off += wtot[w];
    if (f) slot_token_out[(int64_t)e * cap + off + before] = tok;
    __syncthreads();
    if (tid == 0) { int t = 0; for (int w = 0; w < 16; ++w) t += wtot[w]; s_base += t; }
    __syncthreads();
  }
  if (tid == 0) kept_out[e] = s_base;
}
}  // namespace

extern "C" int mp_gather_rows_bf16(const void* src, int64_t ld_src, const int64_t* idx, void* out, int64_t ld_out, int64_t n_rows, int dim,
                                   int scatter, hipStream_t stream) {
  MP_REQUIRE(dim % 8 == 0 && ld_src % 8 == 0 && ld_out % 8 == 0, MP_ERR_SHAPE, "mp_gather_rows_bf16: dim and strides must be multiples of 8");
  const int64_t n = n_rows * (dim / 8);
  if (n == 0) return MP_OK;
  hipLaunchKernelGGL(gather_rows_bf16_kernel, dim3((unsigned)mp_cdiv(n, 256)), dim3(256), 0, stream, (const bf16_t*)src, ld_src, idx, (bf16_t*)out, ld_out,
                     n_rows, dim / 8, scatter);
  return mp_check_launch("mp_gather_rows_bf16");
}

extern "C" int mp_moe_filter_slots(const int* slot_token, const int* kept, const uint8_t* needed, int* slot_token_out, int* kept_out, int n_experts,
                                   int capacity, hipStream_t stream) {
  MP_REQUIRE(n_experts >= 1 && capacity >= 1 && slot_token && kept && needed && slot_token_out && kept_out, MP_ERR_ARG, "mp_moe_filter_slots: bad arguments");
  hipLaunchKernelGGL(moe_filter_slots_kernel, dim3((unsigned)n_experts), dim3(1024), 0, stream, slot_token, kept, needed, slot_token_out, kept_out, capacity);
  return mp_check_launch("mp_moe_filter_slots");
}
