// HBM-bound row kernels of the bf16 trunk (Llama / CLIP / SAM-encoder): RMSNorm, LayerNorm, RoPE, SwiGLU gate,
// casts and broadcast adds.  All loads/stores are 16 B per lane (bf16x8); statistics in fp32.
//
// Reference semantics: HF-4.31 LlamaRMSNorm (fp32 variance, cast, then weight),
// nn.LayerNorm (CLIP eps 1e-5, SAM eps 1e-6 build_sam.py:91), half-split RoPE, LlamaMLP silu(gate)*up.
#include "common.h"

namespace {

constexpr int ROW_THREADS = 256;
constexpr int MAXC = 4;  // register-resident chunks per thread: dim <= 256*8*4 = 8192

__global__ __launch_bounds__(ROW_THREADS) void rmsnorm_bf16_kernel(const bf16_t* __restrict__ x, const float* __restrict__ w,
                                                                   bf16_t* __restrict__ y, int dim, float eps,
                                                                   int64_t ldx, int64_t ldy) {
  __shared__ float red[16];
  const int64_t row = blockIdx.x;
  const bf16_t* xr = x + row * ldx;
  bf16_t* yr = y + row * ldy;
  bf16x8 v[MAXC];
  float ss = 0.f;
#pragma unroll
  for (int c = 0; c < MAXC; ++c) {
    const int i = (c * ROW_THREADS + threadIdx.x) * 8;
    if (i < dim) {
      v[c] = *reinterpret_cast<const bf16x8*>(xr + i);
#pragma unroll
      for (int j = 0; j < 8; ++j) { const float f = (float)v[c][j]; ss += f * f; }
    }
  }
  ss = block_sum(ss, red);
  const float rs = rsqrtf(ss / (float)dim + eps);
#pragma unroll
  for (int c = 0; c < MAXC; ++c) {
    const int i = (c * ROW_THREADS + threadIdx.x) * 8;
    if (i < dim) {
      bf16x8 o;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const bf16_t t = (bf16_t)((float)v[c][j] * rs);  // HF: normalised value is cast to the input dtype first
        o[j] = (bf16_t)(w[i + j] * (float)t);
      }
      *reinterpret_cast<bf16x8*>(yr + i) = o;
    }
  }
}

// LayerNorm with one WAVE per row (dim <= 2048: CLIP 1024, SAM 768 / 256): the two reductions are wave shuffles, no LDS and no
// workgroup barrier; four rows per 256-thread block.  Same arithmetic order per lane as the block version (two-pass variance).
__global__ __launch_bounds__(256) void layernorm_bf16_wave_kernel(const bf16_t* __restrict__ x, const float* __restrict__ w,
                                                                  const float* __restrict__ b, bf16_t* __restrict__ y, int64_t rows,
                                                                  int dim, float eps, int64_t ldx, int64_t ldy) {
  constexpr int WC = 4;                                   // chunks of 64 lanes x 8 elements
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const bf16_t* xr = x + row * ldx;
  bf16_t* yr = y + row * ldy;
  bf16x8 v[WC];
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < WC; ++c) {
    const int i = (c * 64 + lane) * 8;
    if (i < dim) {
      v[c] = *reinterpret_cast<const bf16x8*>(xr + i);
#pragma unroll
      for (int j = 0; j < 8; ++j) s += (float)v[c][j];
    }
  }
  const float mean = wave_sum(s) / (float)dim;
  float q = 0.f;
#pragma unroll
  for (int c = 0; c < WC; ++c) {
    const int i = (c * 64 + lane) * 8;
    if (i < dim) {
#pragma unroll
      for (int j = 0; j < 8; ++j) { const float d = (float)v[c][j] - mean; q += d * d; }
    }
  }
  const float rs = rsqrtf(wave_sum(q) / (float)dim + eps);
#pragma unroll
  for (int c = 0; c < WC; ++c) {
    const int i = (c * 64 + lane) * 8;
    if (i < dim) {
      const f32x4 w0 = *reinterpret_cast<const f32x4*>(w + i), w1 = *reinterpret_cast<const f32x4*>(w + i + 4);
      f32x4 b0 = {0.f, 0.f, 0.f, 0.f}, b1 = {0.f, 0.f, 0.f, 0.f};
      if (b) { b0 = *reinterpret_cast<const f32x4*>(b + i); b1 = *reinterpret_cast<const f32x4*>(b + i + 4); }
      bf16x8 o;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        o[j] = (bf16_t)(((float)v[c][j] - mean) * rs * w0[j] + b0[j]);
        o[4 + j] = (bf16_t)(((float)v[c][4 + j] - mean) * rs * w1[j] + b1[j]);
      }
      *reinterpret_cast<bf16x8*>(yr + i) = o;
    }
  }
}

__global__ __launch_bounds__(ROW_THREADS) void layernorm_bf16_kernel(const bf16_t* __restrict__ x, const float* __restrict__ w,
                                                                     const float* __restrict__ b, bf16_t* __restrict__ y,
                                                                     int dim, float eps, int64_t ldx, int64_t ldy) {
  __shared__ float red[16];
  const int64_t row = blockIdx.x;
  const bf16_t* xr = x + row * ldx;
  bf16_t* yr = y + row * ldy;
  bf16x8 v[MAXC];
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < MAXC; ++c) {
    const int i = (c * ROW_THREADS + threadIdx.x) * 8;
    if (i < dim) {
      v[c] = *reinterpret_cast<const bf16x8*>(xr + i);
#pragma unroll
      for (int j = 0; j < 8; ++j) s += (float)v[c][j];
    }
  }
  const float mean = block_sum(s, red) / (float)dim;
  float q = 0.f;
#pragma unroll
  for (int c = 0; c < MAXC; ++c) {
    const int i = (c * ROW_THREADS + threadIdx.x) * 8;
    if (i < dim) {
#pragma unroll
      for (int j = 0; j < 8; ++j) { const float d = (float)v[c][j] - mean; q += d * d; }
    }
  }
  const float rs = rsqrtf(block_sum(q, red) / (float)dim + eps);
#pragma unroll
  for (int c = 0; c < MAXC; ++c) {
    const int i = (c * ROW_THREADS + threadIdx.x) * 8;
    if (i < dim) {
      bf16x8 o;
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = (bf16_t)(((float)v[c][j] - mean) * rs * w[i + j] + (b ? b[i + j] : 0.f));
      *reinterpret_cast<bf16x8*>(yr + i) = o;
    }
  }
}

// RoPE in place on the q and k thirds of a fused [T, 3*H*D] qkv buffer (D = 128 -> half = 64).
// One thread handles 8 consecutive dims of the low half and the matching 8 of the high half.
__global__ void rope_qk_bf16_kernel(bf16_t* __restrict__ qkv, const float* __restrict__ cos_t, const float* __restrict__ sin_t,
                                    int64_t T, int S, int H, int D, int64_t ld, int pos0) {
  const int half = D / 2;
  const int per_head = half / 8;                 // threads per head
  const int64_t per_tok = (int64_t)2 * H * per_head;  // q and k
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= T * per_tok) return;
  const int64_t tok = idx / per_tok;
  int r = (int)(idx % per_tok);
  const int which = r / (H * per_head);  // 0 = q, 1 = k
  r %= H * per_head;
  const int h = r / per_head, c = (r % per_head) * 8;
  const int pos = (int)(tok % S) + pos0;
  bf16_t* base = qkv + tok * ld + (int64_t)which * H * D + (int64_t)h * D;
  bf16x8 lo = *reinterpret_cast<bf16x8*>(base + c);
  bf16x8 hi = *reinterpret_cast<bf16x8*>(base + half + c);
  const float* cs = cos_t + (int64_t)pos * half + c;
  const float* sn = sin_t + (int64_t)pos * half + c;
  bf16x8 olo, ohi;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float a = (float)lo[j], b = (float)hi[j];
    olo[j] = (bf16_t)(a * cs[j] - b * sn[j]);
    ohi[j] = (bf16_t)(b * cs[j] + a * sn[j]);
  }
  *reinterpret_cast<bf16x8*>(base + c) = olo;
  *reinterpret_cast<bf16x8*>(base + half + c) = ohi;
}

// Decode step (one new token per sequence): RoPE at the cached length read from DEVICE memory (so the launch arguments do not
// change from token to token and the step can live in a HIP graph), q rotated in place, the rotated k and the v appended to the
// KV cache at that position.  One thread per (sequence, head, 8 rotary pairs).
__global__ void decode_rope_append_kernel(bf16_t* __restrict__ qkv, int64_t ld, const float* __restrict__ cos_t,
                                          const float* __restrict__ sin_t, bf16_t* __restrict__ ck, bf16_t* __restrict__ cv,
                                          const int* __restrict__ pos_dev, int B, int H, int D, int64_t c_sb, int64_t c_ss) {
  const int half = D / 2, per_head = half / 8;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= B * H * per_head) return;
  const int c = (idx % per_head) * 8, h = (idx / per_head) % H, b = idx / (per_head * H);
  const int pos = pos_dev[0];
  const float* cs = cos_t + (int64_t)pos * half + c;
  const float* sn = sin_t + (int64_t)pos * half + c;
  bf16_t* q = qkv + (int64_t)b * ld + (int64_t)h * D;
  bf16_t* k = q + (int64_t)H * D;
  const bf16_t* v = k + (int64_t)H * D;
  bf16_t* kd = ck + b * c_sb + (int64_t)pos * c_ss + (int64_t)h * D;
  bf16_t* vd = cv + b * c_sb + (int64_t)pos * c_ss + (int64_t)h * D;
  const bf16x8 qlo = *reinterpret_cast<const bf16x8*>(q + c), qhi = *reinterpret_cast<const bf16x8*>(q + half + c);
  const bf16x8 klo = *reinterpret_cast<const bf16x8*>(k + c), khi = *reinterpret_cast<const bf16x8*>(k + half + c);
  bf16x8 oql, oqh, okl, okh;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float a = (float)qlo[j], bq = (float)qhi[j], ka = (float)klo[j], kb = (float)khi[j];
    oql[j] = (bf16_t)(a * cs[j] - bq * sn[j]); oqh[j] = (bf16_t)(bq * cs[j] + a * sn[j]);
    okl[j] = (bf16_t)(ka * cs[j] - kb * sn[j]); okh[j] = (bf16_t)(kb * cs[j] + ka * sn[j]);
  }
  *reinterpret_cast<bf16x8*>(q + c) = oql; *reinterpret_cast<bf16x8*>(q + half + c) = oqh;
  *reinterpret_cast<bf16x8*>(kd + c) = okl; *reinterpret_cast<bf16x8*>(kd + half + c) = okh;
  *reinterpret_cast<bf16x8*>(vd + c) = *reinterpret_cast<const bf16x8*>(v + c);
  *reinterpret_cast<bf16x8*>(vd + half + c) = *reinterpret_cast<const bf16x8*>(v + half + c);
}
__global__ void advance_ints_kernel(int* p, int n, int delta) {
  if ((int)threadIdx.x < n) p[threadIdx.x] += delta;
}

// out[T, F] = silu(gu[T, 0:F]) * gu[T, F:2F]
__global__ void swiglu_bf16_kernel(const bf16_t* __restrict__ gu, bf16_t* __restrict__ out, int64_t T, int F, int64_t ldgu,
                                   int64_t ldo) {
  const int64_t per_row = F / 8;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= T * per_row) return;
  const int64_t row = idx / per_row;
  const int c = (int)(idx % per_row) * 8;
  const bf16x8 g = *reinterpret_cast<const bf16x8*>(gu + row * ldgu + c);
  const bf16x8 u = *reinterpret_cast<const bf16x8*>(gu + row * ldgu + F + c);
  bf16x8 o;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float gv = (float)g[j];
    o[j] = (bf16_t)(gv * mp_sigmoid_fast(gv) * (float)u[j]);
  }
  *reinterpret_cast<bf16x8*>(out + row * ldo + c) = o;
}

__global__ void cast_f32_bf16_kernel(const float* __restrict__ x, bf16_t* __restrict__ y, int64_t n) {
  const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i + 3 < n) {
    const float4 v = *reinterpret_cast<const float4*>(x + i);
    bf16x4 o = {(bf16_t)v.x, (bf16_t)v.y, (bf16_t)v.z, (bf16_t)v.w};
    *reinterpret_cast<bf16x4*>(y + i) = o;
  } else {
    for (int64_t k = i; k < n; ++k) y[k] = (bf16_t)x[k];
  }
}
__global__ void cast_bf16_f32_kernel(const bf16_t* __restrict__ x, float* __restrict__ y, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) y[i] = (float)x[i];
}

// y[r, :] = x[r, :] + addend[(r % period), :]   (position-embedding add; period = rows of the addend)
__global__ void add_rows_bf16_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ addend, bf16_t* __restrict__ y,
                                     int64_t rows, int dim, int64_t period) {
  const int64_t per_row = dim / 8;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= rows * per_row) return;
  const int64_t row = idx / per_row;
  const int c = (int)(idx % per_row) * 8;
  const bf16x8 a = *reinterpret_cast<const bf16x8*>(x + row * dim + c);
  const bf16x8 b = *reinterpret_cast<const bf16x8*>(addend + (row % period) * dim + c);
  bf16x8 o;
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] = (bf16_t)((float)a[j] + (float)b[j]);
  *reinterpret_cast<bf16x8*>(y + row * dim + c) = o;
}

// y = a + b + c (c optional), elementwise bf16 (fp32 add, single rounding)
__global__ void add3_bf16_kernel(const bf16_t* __restrict__ a, const bf16_t* __restrict__ b, const bf16_t* __restrict__ c,
                                 bf16_t* __restrict__ y, int64_t n) {
  const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 8;
  if (i >= n) return;
  const bf16x8 va = *reinterpret_cast<const bf16x8*>(a + i);
  const bf16x8 vb = *reinterpret_cast<const bf16x8*>(b + i);
  bf16x8 vc;
  if (c) vc = *reinterpret_cast<const bf16x8*>(c + i);
  bf16x8 o;
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] = (bf16_t)((float)va[j] + (float)vb[j] + (c ? (float)vc[j] : 0.f));
  *reinterpret_cast<bf16x8*>(y + i) = o;
}

// greedy decoding: index of the maximum of each row (first index on exact ties, like torch.argmax on CPU)
__global__ __launch_bounds__(256) void argmax_rows_kernel(const float* __restrict__ x, int64_t ld, int cols, int64_t* __restrict__ out) {
  __shared__ float sv[256];
  __shared__ int si[256];
  const float* r = x + (int64_t)blockIdx.x * ld;
  float bv = -INFINITY; int bi = 0x7fffffff;
  for (int i = threadIdx.x; i < cols; i += 256) {
    const float v = r[i];
    if (v > bv || (v == bv && i < bi)) { bv = v; bi = i; }
  }
  sv[threadIdx.x] = bv; si[threadIdx.x] = bi;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (threadIdx.x < off) {
      const float v = sv[threadIdx.x + off]; const int i = si[threadIdx.x + off];
      if (v > sv[threadIdx.x] || (v == sv[threadIdx.x] && i < si[threadIdx.x])) { sv[threadIdx.x] = v; si[threadIdx.x] = i; }
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) out[blockIdx.x] = si[0];
}

// The decode step's form (up to 32768 columns: the 32267-entry vocabulary): 1024 threads, every thread's <= 8 float4 loads
// requested at once, wave reductions by shuffles, one pass through LDS for the sixteen waves.  The 256-thread kernel above walks 125
// dependent trips per thread: 37 us of a 3.3 ms decode step for 128 KB of logits; this one ~5 us.  Same result (first index on ties).
__device__ __forceinline__ void argmax_take(float& bv, int& bi, float v, int i) {
  if (v > bv || (v == bv && i < bi)) { bv = v; bi = i; }
}
__global__ __launch_bounds__(1024) void argmax_rows_wide_kernel(const float* __restrict__ x, int64_t ld, int cols, int64_t* __restrict__ out) {
  __shared__ float sv[16];
  __shared__ int si[16];
  const float* r = x + (int64_t)blockIdx.x * ld;
  const int tid = threadIdx.x;
  float bv = -INFINITY; int bi = 0x7fffffff;
  if ((reinterpret_cast<uintptr_t>(r) & 15) == 0) {          // (wave-uniform) 16-byte pieces + up to three single columns at the end
    const int n4 = cols >> 2;
    float4 v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int i4 = tid + k * 1024;
      v[k] = float4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
      if (i4 < n4) v[k] = *reinterpret_cast<const float4*>(r + 4 * i4);
    }
    const int it = 4 * n4 + tid;
    const float vt = (tid < 3 && it < cols) ? r[it] : -INFINITY;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int i = 4 * (tid + k * 1024);
      if (tid + k * 1024 < n4) { argmax_take(bv, bi, v[k].x, i); argmax_take(bv, bi, v[k].y, i + 1); argmax_take(bv, bi, v[k].z, i + 2); argmax_take(bv, bi, v[k].w, i + 3); }
    }
    if (tid < 3 && it < cols) argmax_take(bv, bi, vt, it);
  } else {                                                   // a row that does not start on 16 bytes: single columns, eight requests at a time
    for (int k0 = 0; k0 < 32; k0 += 8) {
      float v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) { const int i = tid + (k0 + k) * 1024; v[k] = i < cols ? r[i] : -INFINITY; }
#pragma unroll
      for (int k = 0; k < 8; ++k) { const int i = tid + (k0 + k) * 1024; if (i < cols) argmax_take(bv, bi, v[k], i); }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(bv, o, 64); const int oi = __shfl_xor(bi, o, 64);
    argmax_take(bv, bi, ov, oi);
  }
  if ((tid & 63) == 0) { sv[tid >> 6] = bv; si[tid >> 6] = bi; }
  __syncthreads();
  if (tid == 0) {
    for (int w = 1; w < 16; ++w) argmax_take(bv, bi, sv[w], si[w]);
    out[blockIdx.x] = bi;
  }
}

}  // namespace

extern "C" int mp_argmax_rows_f32(const float* x, int64_t ld, int64_t rows, int cols, int64_t* out, hipStream_t stream) {
  MP_REQUIRE(cols > 0, MP_ERR_SHAPE, "mp_argmax_rows_f32: bad shape");
  if (rows == 0) return MP_OK;
  if (cols <= 32768)
    hipLaunchKernelGGL(argmax_rows_wide_kernel, dim3((unsigned)rows), dim3(1024), 0, stream, x, ld, cols, out);
  else
    hipLaunchKernelGGL(argmax_rows_kernel, dim3((unsigned)rows), dim3(256), 0, stream, x, ld, cols, out);
  return mp_check_launch("mp_argmax_rows_f32");
}

extern "C" int mp_rmsnorm_bf16(const void* x, int64_t ldx, const float* w, void* y, int64_t ldy, int64_t rows, int dim,
                               float eps, hipStream_t stream) {
  MP_REQUIRE(dim % 8 == 0 && dim <= ROW_THREADS * 8 * MAXC && ldx % 8 == 0 && ldy % 8 == 0, MP_ERR_SHAPE,
             "mp_rmsnorm_bf16: dim=%d unsupported", dim);
  if (rows == 0) return MP_OK;
  hipLaunchKernelGGL(rmsnorm_bf16_kernel, dim3((unsigned)rows), dim3(ROW_THREADS), 0, stream, (const bf16_t*)x, w,
                     (bf16_t*)y, dim, eps, ldx, ldy);
  return mp_check_launch("mp_rmsnorm_bf16");
}

extern "C" int mp_layernorm_bf16(const void* x, int64_t ldx, const float* w, const float* b, void* y, int64_t ldy,
                                 int64_t rows, int dim, float eps, hipStream_t stream) {
  MP_REQUIRE(dim % 8 == 0 && dim <= ROW_THREADS * 8 * MAXC && ldx % 8 == 0 && ldy % 8 == 0, MP_ERR_SHAPE,
             "mp_layernorm_bf16: dim=%d unsupported", dim);
  if (rows == 0) return MP_OK;
  if (dim <= 2048) {
    hipLaunchKernelGGL(layernorm_bf16_wave_kernel, dim3((unsigned)mp_cdiv(rows, 4)), dim3(256), 0, stream, (const bf16_t*)x, w, b,
                       (bf16_t*)y, rows, dim, eps, ldx, ldy);
    return mp_check_launch("mp_layernorm_bf16");
  }
  hipLaunchKernelGGL(layernorm_bf16_kernel, dim3((unsigned)rows), dim3(ROW_THREADS), 0, stream, (const bf16_t*)x, w, b,
                     (bf16_t*)y, dim, eps, ldx, ldy);
  return mp_check_launch("mp_layernorm_bf16");
}

extern "C" int mp_rope_qk_bf16(void* qkv, int64_t ld, const float* cos_t, const float* sin_t, int64_t tokens, int seq,
                               int heads, int head_dim, int pos_offset, hipStream_t stream) {
  MP_REQUIRE(head_dim % 16 == 0 && ld % 8 == 0 && seq > 0, MP_ERR_SHAPE, "mp_rope_qk_bf16: bad shape");
  const int64_t n = tokens * 2 * heads * (head_dim / 16);
  if (n == 0) return MP_OK;
  hipLaunchKernelGGL(rope_qk_bf16_kernel, dim3((unsigned)mp_cdiv(n, 256)), dim3(256), 0, stream, (bf16_t*)qkv, cos_t, sin_t,
                     tokens, seq, heads, head_dim, ld, pos_offset);
  return mp_check_launch("mp_rope_qk_bf16");
}

extern "C" int mp_decode_rope_append_bf16(void* qkv, int64_t ld, const float* cos_t, const float* sin_t, void* cache_k, void* cache_v,
                                          const int* pos_dev, int B, int heads, int head_dim, int64_t cache_batch_stride,
                                          int64_t cache_seq_stride, hipStream_t stream) {
  MP_REQUIRE(head_dim % 16 == 0 && ld % 8 == 0 && pos_dev != nullptr, MP_ERR_SHAPE, "mp_decode_rope_append_bf16: bad shape");
  const int n = B * heads * (head_dim / 16);
  if (n == 0) return MP_OK;
  hipLaunchKernelGGL(decode_rope_append_kernel, dim3((unsigned)mp_cdiv(n, 256)), dim3(256), 0, stream, (bf16_t*)qkv, ld, cos_t, sin_t,
                     (bf16_t*)cache_k, (bf16_t*)cache_v, pos_dev, B, heads, head_dim, cache_batch_stride, cache_seq_stride);
  return mp_check_launch("mp_decode_rope_append_bf16");
}

extern "C" int mp_advance_ints(int* p, int n, int delta, hipStream_t stream) {
  MP_REQUIRE(n >= 0 && n <= 64, MP_ERR_SHAPE, "mp_advance_ints: n <= 64");
  if (n == 0) return MP_OK;
  hipLaunchKernelGGL(advance_ints_kernel, dim3(1), dim3(64), 0, stream, p, n, delta);
  return mp_check_launch("mp_advance_ints");
}

extern "C" int mp_swiglu_bf16(const void* gu, int64_t ldgu, void* out, int64_t ldo, int64_t rows, int ff,
                              hipStream_t stream) {
  MP_REQUIRE(ff % 8 == 0 && ldgu % 8 == 0 && ldo % 8 == 0, MP_ERR_SHAPE, "mp_swiglu_bf16: bad shape");
  const int64_t n = rows * (ff / 8);
  if (n == 0) return MP_OK;
  hipLaunchKernelGGL(swiglu_bf16_kernel, dim3((unsigned)mp_cdiv(n, 256)), dim3(256), 0, stream, (const bf16_t*)gu,
                     (bf16_t*)out, rows, ff, ldgu, ldo);
  return mp_check_launch("mp_swiglu_bf16");
}

extern "C" int mp_cast_f32_to_bf16(const float* x, void* y, int64_t n, hipStream_t stream) {
  if (n == 0) return MP_OK;
  hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3((unsigned)mp_cdiv(mp_cdiv(n, 4), 256)), dim3(256), 0, stream, x,
                     (bf16_t*)y, n);
  return mp_check_launch("mp_cast_f32_to_bf16");
}

extern "C" int mp_cast_bf16_to_f32(const void* x, float* y, int64_t n, hipStream_t stream) {
  if (n == 0) return MP_OK;
  hipLaunchKernelGGL(cast_bf16_f32_kernel, dim3((unsigned)mp_cdiv(n, 256)), dim3(256), 0, stream, (const bf16_t*)x, y, n);
  return mp_check_launch("mp_cast_bf16_to_f32");
}

extern "C" int mp_add_rows_bf16(const void* x, const void* addend, void* y, int64_t rows, int dim, int64_t period,
                                hipStream_t stream) {
  MP_REQUIRE(dim % 8 == 0 && period > 0, MP_ERR_SHAPE, "mp_add_rows_bf16: bad shape");
  const int64_t n = rows * (dim / 8);
  if (n == 0) return MP_OK;
  hipLaunchKernelGGL(add_rows_bf16_kernel, dim3((unsigned)mp_cdiv(n, 256)), dim3(256), 0, stream, (const bf16_t*)x,
                     (const bf16_t*)addend, (bf16_t*)y, rows, dim, period);
  return mp_check_launch("mp_add_rows_bf16");
}

extern "C" int mp_add3_bf16(const void* a, const void* b, const void* c, void* y, int64_t n, hipStream_t stream) {
  MP_REQUIRE(n % 8 == 0, MP_ERR_SHAPE, "mp_add3_bf16: n must be a multiple of 8");
  if (n == 0) return MP_OK;
  hipLaunchKernelGGL(add3_bf16_kernel, dim3((unsigned)mp_cdiv(n / 8, 256)), dim3(256), 0, stream, (const bf16_t*)a,
                     (const bf16_t*)b, (const bf16_t*)c, (bf16_t*)y, n);
  return mp_check_launch("mp_add3_bf16");
}
