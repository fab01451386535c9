// fp32 row / elementwise kernels (forward + backward) for the trainable tail: LayerNorm (also LayerNorm2d in NHWC),
// row softmax, activations, broadcast add, column sums (bias grads), ConvTranspose2d(k=2,s=2) pixel shuffle, gathers.
// Reference sites: nn.LayerNorm / LayerNorm2d (modeling/common.py:31-45), transformer.py:218-244 softmax,
// mask_decoder.py:53-59 (ConvTranspose2d 2x2/s2 + LayerNorm2d + GELU), MedPLIB.py:461 (SEG-row gather).
#include "common.h"

namespace {

// ---------------- LayerNorm fwd/bwd (one wave per row; dim <= 4096) ----------------
__global__ __launch_bounds__(256) void ln_fwd_f32_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                         const float* __restrict__ b, float* __restrict__ y,
                                                         float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                         int64_t rows, int dim, float eps) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* xr = x + row * dim;
  float s = 0.f;
  for (int i = lane; i < dim; i += 64) s += xr[i];
  const float mean = wave_sum(s) / (float)dim;
  float q = 0.f;
  for (int i = lane; i < dim; i += 64) { const float d = xr[i] - mean; q += d * d; }
  const float rstd = 1.f / sqrtf(wave_sum(q) / (float)dim + eps);
  for (int i = lane; i < dim; i += 64) y[row * dim + i] = (xr[i] - mean) * rstd * w[i] + b[i];
  if (lane == 0) {
    if (mean_out) mean_out[row] = mean;
    if (rstd_out) rstd_out[row] = rstd;
  }
}

// dx = rstd * (g - mean(g) - xhat * mean(g*xhat)),  g = dy * w ; dw / db come from ln_bwd_wb_f32_kernel below.
__global__ __launch_bounds__(256) void ln_bwd_f32_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                         const float* __restrict__ w, const float* __restrict__ mean,
                                                         const float* __restrict__ rstd, float* __restrict__ dx,
                                                         float* __restrict__ dw, float* __restrict__ db, int64_t rows,
                                                         int dim) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* xr = x + row * dim;
  const float* dyr = dy + row * dim;
  const float mu = mean[row], rs = rstd[row];
  float sg = 0.f, sgx = 0.f;
  for (int i = lane; i < dim; i += 64) {
    const float xh = (xr[i] - mu) * rs, g = dyr[i] * w[i];
    sg += g; sgx += g * xh;
  }
  sg = wave_sum(sg) / (float)dim;
  sgx = wave_sum(sgx) / (float)dim;
  for (int i = lane; i < dim; i += 64) {
    const float xh = (xr[i] - mu) * rs, g = dyr[i] * w[i];
    dx[row * dim + i] = rs * (g - sg - xh * sgx);
  }
}
// dw[c] += sum_r dy[r,c] * xhat[r,c];  db[c] += sum_r dy[r,c]: one block per 64 columns, rows striped over 16 waves, partials
// combined in wave order (float atomics from every row made LayerNorm weight gradients irreproducible)
__global__ __launch_bounds__(1024) void ln_bwd_wb_f32_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                             const float* __restrict__ mean, const float* __restrict__ rstd,
                                                             float* __restrict__ dw, float* __restrict__ db, int64_t rows, int dim) {
  __shared__ float pw[16][64], pb[16][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);
  const int w = threadIdx.x >> 6;
  float sw = 0.f, sb = 0.f;
  if (c < dim)
    for (int64_t r = w; r < rows; r += 16) {
      const float d = dy[r * dim + c];
      sw += d * ((x[r * dim + c] - mean[r]) * rstd[r]);
      sb += d;
    }
  pw[w][threadIdx.x & 63] = sw; pb[w][threadIdx.x & 63] = sb;
  __syncthreads();
  if (w == 0 && c < dim) {
    float tw = 0.f, tb = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) { tw += pw[k][threadIdx.x]; tb += pb[k][threadIdx.x]; }
    if (dw) dw[c] += tw;
    if (db) db[c] += tb;
  }
}

// ---------------- row softmax fwd/bwd (one wave per row; cols <= 4096) ----------------
__global__ __launch_bounds__(256) void softmax_fwd_f32_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t rows,
                                                              int cols, float scale) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* xr = x + row * cols;
  float m = -INFINITY;
  for (int i = lane; i < cols; i += 64) m = fmaxf(m, xr[i] * scale);
  m = wave_max(m);
  float s = 0.f;
  for (int i = lane; i < cols; i += 64) s += expf(xr[i] * scale - m);
  s = wave_sum(s);
  const float inv = 1.f / s;
  for (int i = lane; i < cols; i += 64) y[row * cols + i] = expf(xr[i] * scale - m) * inv;
}
// dx = scale * p * (dp - sum(dp * p))
__global__ __launch_bounds__(256) void softmax_bwd_f32_kernel(const float* __restrict__ p, const float* __restrict__ dp,
                                                              float* __restrict__ dx, int64_t rows, int cols, float scale) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  float s = 0.f;
  for (int i = lane; i < cols; i += 64) s += p[row * cols + i] * dp[row * cols + i];
  s = wave_sum(s);
  for (int i = lane; i < cols; i += 64) dx[row * cols + i] = scale * p[row * cols + i] * (dp[row * cols + i] - s);
}

// ---------------- elementwise ----------------
// y = a + b[i % period]
__global__ void add_f32_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ y, int64_t n,
                               int64_t period) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) y[i] = a[i] + b[i % period];
}
// act forward: 1 relu, 2 gelu
__global__ void act_fwd_f32_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t n, int act) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) y[i] = act == 1 ? fmaxf(x[i], 0.f) : gelu_erf(x[i]);
}
// act backward from the PRE-activation input x: dx = dy * act'(x)
__global__ void act_bwd_f32_kernel(const float* __restrict__ dy, const float* __restrict__ x, float* __restrict__ dx, int64_t n,
                                   int act) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (act == 1) dx[i] = x[i] > 0.f ? dy[i] : 0.f;
  else if (act == 2) dx[i] = dy[i] * gelu_erf_grad(x[i]);
  else { const float s = x[i]; dx[i] = dy[i] * s * (1.f - s); }  // act==3: x holds the sigmoid OUTPUT
}
// out[c] (+)= sum_r x[r, c]   (bias gradients); one block per 64 columns, rows striped over 16 waves with four independent
// accumulators per thread (the launches are latency-bound: 4 waves walking 2048 rows one dependent add at a time took 17 us);
// the 16 partials are combined in a fixed order, so the result does not depend on scheduling
__global__ __launch_bounds__(1024) void colsum_f32_kernel(const float* __restrict__ x, float* __restrict__ out, int64_t rows,
                                                          int cols, int accumulate) {
  __shared__ float part[16][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);
  const int w = threadIdx.x >> 6;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (c < cols) {
    int64_t r = w;
    for (; r + 48 < rows; r += 64) {
      s0 += x[r * cols + c]; s1 += x[(r + 16) * cols + c]; s2 += x[(r + 32) * cols + c]; s3 += x[(r + 48) * cols + c];
    }
    for (; r < rows; r += 16) s0 += x[r * cols + c];
  }
  part[w][threadIdx.x & 63] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (w == 0 && c < cols) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += part[k][threadIdx.x];
    out[c] = accumulate ? out[c] + t : t;
  }
}

// ---------------- ConvTranspose2d(k=2, s=2) pixel shuffle ----------------
// GEMM output G[b*h*w + i*w + j, co*4 + kh*2 + kw]  <->  Y[b, 2i+kh, 2j+kw, co] (NHWC), + bias[co] on the way forward.
__global__ void convt2x2_shuffle_fwd_kernel(const float* __restrict__ G, const float* __restrict__ bias, float* __restrict__ Y,
                                            int B, int h, int w, int Co) {
  const int64_t n = (int64_t)B * h * w * Co * 4;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n) return;
  // idx enumerates Y in NHWC order
  const int co = (int)(idx % Co);
  int64_t t = idx / Co;
  const int X = (int)(t % (2 * w)); t /= 2 * w;
  const int Yy = (int)(t % (2 * h));
  const int b = (int)(t / (2 * h));
  const int i = Yy >> 1, kh = Yy & 1, j = X >> 1, kw = X & 1;
  Y[idx] = G[(((int64_t)b * h + i) * w + j) * (Co * 4) + co * 4 + kh * 2 + kw] + (bias ? bias[co] : 0.f);
}
__global__ void convt2x2_shuffle_bwd_kernel(const float* __restrict__ dY, float* __restrict__ dG, int B, int h, int w, int Co) {
  const int64_t n = (int64_t)B * h * w * Co * 4;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n) return;
  // idx enumerates dG
  const int q = (int)(idx % (Co * 4));
  const int64_t pix = idx / (Co * 4);
  const int co = q >> 2, kh = (q >> 1) & 1, kw = q & 1;
  const int j = (int)(pix % w);
  const int i = (int)((pix / w) % h);
  const int b = (int)(pix / ((int64_t)w * h));
  dG[idx] = dY[((((int64_t)b * 2 * h) + 2 * i + kh) * (2 * w) + 2 * j + kw) * Co + co];
}

// ---------------- gathers / casts ----------------
// out[r, :] = (float) src[idx[r], :]   (bf16 hidden-state rows -> fp32)
__global__ void gather_rows_bf16_f32_kernel(const bf16_t* __restrict__ src, const int64_t* __restrict__ idx, float* __restrict__ out,
                                            int64_t n_rows, int dim, int64_t ld) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_rows * dim) return;
  const int64_t r = i / dim;
  const int c = (int)(i % dim);
  out[i] = (float)src[idx[r] * ld + c];
}
// out[r, :] = src[idx[r], :]   fp32 rows (expand_embedding, MedPLIB.py:292-308)
__global__ void gather_rows_f32_kernel(const float* __restrict__ src, const int64_t* __restrict__ idx, float* __restrict__ out,
                                       int64_t n_rows, int64_t dim) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_rows * dim) return;
  out[i] = src[idx[i / dim] * dim + i % dim];
}
__global__ void scale_f32_kernel(float* __restrict__ x, int64_t n, float s) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) x[i] *= s;
}

}  // namespace

#define GRID1D(n) dim3((unsigned)mp_cdiv((n), 256)), dim3(256), 0, stream

extern "C" int mp_layernorm_fwd_f32(const float* x, const float* w, const float* b, float* y, float* mean, float* rstd,
                                    int64_t rows, int dim, float eps, hipStream_t stream) {
  MP_REQUIRE(dim > 0, MP_ERR_SHAPE, "mp_layernorm_fwd_f32: bad dim");
  if (rows == 0) return MP_OK;
  hipLaunchKernelGGL(ln_fwd_f32_kernel, dim3((unsigned)mp_cdiv(rows, 4)), dim3(256), 0, stream, x, w, b, y, mean, rstd, rows,
                     dim, eps);
  return mp_check_launch("mp_layernorm_fwd_f32");
}
extern "C" int mp_layernorm_bwd_f32(const float* dy, const float* x, const float* w, const float* mean, const float* rstd,
                                    float* dx, float* dw_accum, float* db_accum, int64_t rows, int dim, hipStream_t stream) {
  MP_REQUIRE(dim > 0, MP_ERR_SHAPE, "mp_layernorm_bwd_f32: bad dim");
  if (rows == 0) return MP_OK;
  hipLaunchKernelGGL(ln_bwd_f32_kernel, dim3((unsigned)mp_cdiv(rows, 4)), dim3(256), 0, stream, dy, x, w, mean, rstd, dx,
                     dw_accum, db_accum, rows, dim);
  if (dw_accum || db_accum)
    hipLaunchKernelGGL(ln_bwd_wb_f32_kernel, dim3((unsigned)mp_cdiv(dim, 64)), dim3(1024), 0, stream, dy, x, mean, rstd, dw_accum, db_accum,
                       rows, dim);
  return mp_check_launch("mp_layernorm_bwd_f32");
}
extern "C" int mp_softmax_fwd_f32(const float* x, float* y, int64_t rows, int cols, float scale, hipStream_t stream) {
  if (rows == 0 || cols == 0) return MP_OK;
  hipLaunchKernelGGL(softmax_fwd_f32_kernel, dim3((unsigned)mp_cdiv(rows, 4)), dim3(256), 0, stream, x, y, rows, cols, scale);
  return mp_check_launch("mp_softmax_fwd_f32");
}
extern "C" int mp_softmax_bwd_f32(const float* p, const float* dp, float* dx, int64_t rows, int cols, float scale,
                                  hipStream_t stream) {
  if (rows == 0 || cols == 0) return MP_OK;
  hipLaunchKernelGGL(softmax_bwd_f32_kernel, dim3((unsigned)mp_cdiv(rows, 4)), dim3(256), 0, stream, p, dp, dx, rows, cols,
                     scale);
  return mp_check_launch("mp_softmax_bwd_f32");
}
extern "C" int mp_add_f32(const float* a, const float* b, float* y, int64_t n, int64_t period, hipStream_t stream) {
  MP_REQUIRE(period > 0, MP_ERR_ARG, "mp_add_f32: period must be positive");
  if (n == 0) return MP_OK;
  hipLaunchKernelGGL(add_f32_kernel, GRID1D(n), a, b, y, n, period);
  return mp_check_launch("mp_add_f32");
}
extern "C" int mp_act_fwd_f32(const float* x, float* y, int64_t n, int act, hipStream_t stream) {
  MP_REQUIRE(act == 1 || act == 2, MP_ERR_ARG, "mp_act_fwd_f32: act must be 1 (relu) or 2 (gelu)");
  if (n == 0) return MP_OK;
  hipLaunchKernelGGL(act_fwd_f32_kernel, GRID1D(n), x, y, n, act);
  return mp_check_launch("mp_act_fwd_f32");
}
extern "C" int mp_act_bwd_f32(const float* dy, const float* x, float* dx, int64_t n, int act, hipStream_t stream) {
  MP_REQUIRE(act >= 1 && act <= 3, MP_ERR_ARG, "mp_act_bwd_f32: bad act");
  if (n == 0) return MP_OK;
  hipLaunchKernelGGL(act_bwd_f32_kernel, GRID1D(n), dy, x, dx, n, act);
  return mp_check_launch("mp_act_bwd_f32");
}
extern "C" int mp_colsum_f32(const float* x, float* out, int64_t rows, int cols, int accumulate, hipStream_t stream) {
  if (cols == 0) return MP_OK;
  hipLaunchKernelGGL(colsum_f32_kernel, dim3((unsigned)mp_cdiv(cols, 64)), dim3(1024), 0, stream, x, out, rows, cols,
                     accumulate);
  return mp_check_launch("mp_colsum_f32");
}
extern "C" int mp_convt2x2_shuffle_fwd_f32(const float* G, const float* bias, float* Y, int B, int h, int w, int Co,
                                           hipStream_t stream) {
  const int64_t n = (int64_t)B * h * w * Co * 4;
  if (n == 0) return MP_OK;
  hipLaunchKernelGGL(convt2x2_shuffle_fwd_kernel, GRID1D(n), G, bias, Y, B, h, w, Co);
  return mp_check_launch("mp_convt2x2_shuffle_fwd_f32");
}
extern "C" int mp_convt2x2_shuffle_bwd_f32(const float* dY, float* dG, int B, int h, int w, int Co, hipStream_t stream) {
  const int64_t n = (int64_t)B * h * w * Co * 4;
  if (n == 0) return MP_OK;
  hipLaunchKernelGGL(convt2x2_shuffle_bwd_kernel, GRID1D(n), dY, dG, B, h, w, Co);
  return mp_check_launch("mp_convt2x2_shuffle_bwd_f32");
}
extern "C" int mp_gather_rows_bf16_to_f32(const void* src, int64_t ld, const int64_t* idx, float* out, int64_t n_rows, int dim,
                                          hipStream_t stream) {
  const int64_t n = n_rows * dim;
  if (n == 0) return MP_OK;
  hipLaunchKernelGGL(gather_rows_bf16_f32_kernel, GRID1D(n), (const bf16_t*)src, idx, out, n_rows, dim, ld);
  return mp_check_launch("mp_gather_rows_bf16_to_f32");
}
extern "C" int mp_gather_rows_f32(const float* src, const int64_t* idx, float* out, int64_t n_rows, int64_t dim,
                                  hipStream_t stream) {
  const int64_t n = n_rows * dim;
  if (n == 0) return MP_OK;
  hipLaunchKernelGGL(gather_rows_f32_kernel, GRID1D(n), src, idx, out, n_rows, dim);
  return mp_check_launch("mp_gather_rows_f32");
}
extern "C" int mp_scale_f32(float* x, int64_t n, float s, hipStream_t stream) {
  if (n == 0) return MP_OK;
  hipLaunchKernelGGL(scale_f32_kernel, GRID1D(n), x, n, s);
  return mp_check_launch("mp_scale_f32");
}
