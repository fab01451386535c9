// Row / elementwise kernels of the decoder BACKWARD (LoRA training; the autograd of HF-4.31 LlamaRMSNorm,
// LlamaMLP's silu(gate) * up, the filtered cross-entropy of medplib_moe_llama.py:392-408, and peft's LoRA adapters,
// train_ds_medplib.py:262-303).  All HBM-bound; the GEMMs of the backward are the forward's NT kernel on transposed weight copies.
//
//   mp_rmsnorm_bwd_bf16        dx = rs * (dy*w - xhat * mean(dy*w*xhat)) (+ add), xhat = x*rs              one block per row
//   mp_swiglu_pair_fwd/bwd     act = silu(g)*u on the gate|up GEMM output with gate / up interleaved in blocks of 32 columns
//                              (the layout of the fused weights, ops.swiglu_interleave); d_g, d_u in the same layout
//   mp_tn_skinny_f32           out[n, j] = scale * sum_t X[t, n] * G[t, j], j < R <= 32: the adapters' weight gradients (a column
//                              block per workgroup, the token axis split over its waves and combined in a fixed order)
//   mp_ce_rows_bwd             d_logits = g * (softmax(logits) - onehot(label)) for the supervised rows, written as bf16 [n, ldo]
//   mp_scatter_rows_f32_bf16   out[rows[i], :] = bf16(g[i, :])  (backward of the row gather in front of text_hidden_fcs / lm_head)
//   mp_dropout_bf16            y = x * keep / (1 - p) with keep from the stateless hash generator (peft lora_dropout on the adapter input)
#include "common.h"
#include <algorithm>
#include <stdlib.h>
#include "gemm_common.h"

int mp_device_cus();            // gemm256_bf16.hip (cached per device)

namespace {

constexpr int RB_THREADS = 256, RB_MAXC = 4;

__global__ __launch_bounds__(RB_THREADS) void rmsnorm_bwd_kernel(const bf16_t* __restrict__ x, const float* __restrict__ w,
                                                                 const bf16_t* __restrict__ dy, const bf16_t* __restrict__ add,
                                                                 bf16_t* __restrict__ dx, int dim, float eps, int64_t ldx, int64_t ldy,
                                                                 int64_t lda, int64_t ldo, float* __restrict__ rs_out) {
  __shared__ float red[16];
  const int64_t row = blockIdx.x;
  const bf16_t* xr = x + row * ldx;
  const bf16_t* gr = dy + row * ldy;
  bf16x8 xv[RB_MAXC], gv[RB_MAXC];
  float ss = 0.f;
#pragma unroll
  for (int c = 0; c < RB_MAXC; ++c) {
    const int i = (c * RB_THREADS + threadIdx.x) * 8;
    if (i < dim) {
      xv[c] = *reinterpret_cast<const bf16x8*>(xr + i);
      gv[c] = *reinterpret_cast<const bf16x8*>(gr + i);
#pragma unroll
      for (int j = 0; j < 8; ++j) { const float f = (float)xv[c][j]; ss += f * f; }
    }
  }
  ss = block_sum(ss, red);
  const float rs = rsqrtf(ss / (float)dim + eps);
  if (rs_out && threadIdx.x == 0) rs_out[row] = rs;
  float dot = 0.f;
#pragma unroll
  for (int c = 0; c < RB_MAXC; ++c) {
    const int i = (c * RB_THREADS + threadIdx.x) * 8;
    if (i < dim) {
#pragma unroll
      for (int j = 0; j < 8; ++j) dot += (float)gv[c][j] * w[i + j] * ((float)xv[c][j] * rs);
    }
  }
  dot = block_sum(dot, red) / (float)dim;
#pragma unroll
  for (int c = 0; c < RB_MAXC; ++c) {
    const int i = (c * RB_THREADS + threadIdx.x) * 8;
    if (i < dim) {
      bf16x8 av;
      if (add) av = *reinterpret_cast<const bf16x8*>(add + row * lda + i);
      bf16x8 o;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float xh = (float)xv[c][j] * rs;
        float v = rs * ((float)gv[c][j] * w[i + j] - xh * dot);
        if (add) v += (float)av[j];
        o[j] = (bf16_t)v;
      }
      *reinterpret_cast<bf16x8*>(dx + row * ldo + i) = o;
    }
  }
}

// gate|up interleaved in blocks of 32: column blk*64 + j = gate channel blk*32 + j, column blk*64 + 32 + j = its up channel
// (counts / cap, optional: the rows are capacity slabs [E, cap, .] of which only the first counts[e] of slab e hold routed tokens -- the rest is skipped)
__global__ void swiglu_pair_fwd_kernel(const bf16_t* __restrict__ gu, bf16_t* __restrict__ act, int64_t T, int ff, int64_t ldact,
                                       const int* __restrict__ counts, int cap) {
  const int per_row = ff / 8;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= T * per_row) return;
  const int64_t t = idx / per_row;
  if (counts && (int)(t % cap) >= counts[t / cap]) return;
  const int c = (int)(idx % per_row) * 8;                 // act channel (8 consecutive, inside one block of 32)
  const int64_t col = (int64_t)(c >> 5) * 64 + (c & 31);
  const bf16x8 g = *reinterpret_cast<const bf16x8*>(gu + t * 2 * ff + col);
  const bf16x8 u = *reinterpret_cast<const bf16x8*>(gu + t * 2 * ff + col + 32);
  bf16x8 o;
#pragma unroll
  for (int j = 0; j < 8; ++j) { const float gf = (float)g[j]; o[j] = (bf16_t)(gf * mp_sigmoid_fast(gf) * (float)u[j]); }
  *reinterpret_cast<bf16x8*>(act + t * ldact + c) = o;
}

__global__ void swiglu_pair_bwd_kernel(const bf16_t* __restrict__ gu, const bf16_t* __restrict__ dact, bf16_t* __restrict__ dgu, int64_t T,
                                       int ff, const int* __restrict__ counts, int cap) {
  const int per_row = ff / 8;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= T * per_row) return;
  const int64_t t = idx / per_row;
  if (counts && (int)(t % cap) >= counts[t / cap]) return;
  const int c = (int)(idx % per_row) * 8;
  const int64_t col = (int64_t)(c >> 5) * 64 + (c & 31);
  const bf16x8 g = *reinterpret_cast<const bf16x8*>(gu + t * 2 * ff + col);
  const bf16x8 u = *reinterpret_cast<const bf16x8*>(gu + t * 2 * ff + col + 32);
  const bf16x8 d = *reinterpret_cast<const bf16x8*>(dact + t * ff + c);
  bf16x8 dg, du;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float gf = (float)g[j], uf = (float)u[j], df = (float)d[j];
    const float sg = mp_sigmoid_fast(gf);
    du[j] = (bf16_t)(df * gf * sg);
    dg[j] = (bf16_t)(df * uf * sg * (1.f + gf * (1.f - sg)));
  }
  *reinterpret_cast<bf16x8*>(dgu + t * 2 * ff + col) = dg;
  *reinterpret_cast<bf16x8*>(dgu + t * 2 * ff + col + 32) = du;
}

// lora_dropout's keep decision: one splitmix64 value per 4 consecutive elements (16 bits each: keep iff bits >= p * 65536), so the
// generator costs a quarter of a hash per element (the 64-bit finaliser is ~25 VALU operations; at one per element it made the dropout
// pass VALU-bound).  Index = position in the contiguous [tokens, features] tensor; every kernel that needs the mask calls this.
__device__ __forceinline__ uint64_t hash64(uint64_t k) {
  k ^= k >> 30; k *= 0xbf58476d1ce4e5b9ull;
  k ^= k >> 27; k *= 0x94d049bb133111ebull;
  k ^= k >> 31;
  return k;
}
__device__ __forceinline__ uint64_t dropout_bits4(uint64_t seed, uint64_t i4) { return hash64(seed * 0x100000001b3ull + i4); }   // i4 = index / 4
__device__ __forceinline__ unsigned dropout_thresh(float p) { return (unsigned)(p * 65536.f); }
// The keep decisions of 8 consecutive elements (i4 = index of the first / 4) as a byte: bit j set = element j is kept.  Round 5: the
// forward writes these bytes ([tokens, features / 8], 1/16 of the tensor) and the two backward kernels that need the same mask read them
// instead of hashing again: two splitmix64 values + eight 16-bit compares per 8 elements made the backward's weight-gradient pass over the
// adapter input VALU-bound (54 us against 26 for the [5112, 11008] input of the down adapter, scripts/r05_skinny_bench.py).
__device__ __forceinline__ unsigned dropout_keep8(uint64_t seed, uint64_t i4, unsigned th) {
  unsigned m = 0;
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const uint64_t bits = dropout_bits4(seed, i4 + q);
#pragma unroll
    for (int j = 0; j < 4; ++j) m |= ((((unsigned)(bits >> (16 * j)) & 0xffffu) >= th) ? 1u : 0u) << (q * 4 + j);
  }
  return m;
}
__device__ __forceinline__ bf16x8 dropout_apply8(bf16x8 v, unsigned m, float keep_scale) {
  bf16x8 o;
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] = (bf16_t)(((m >> j) & 1u) ? (float)v[j] * keep_scale : 0.f);
  return o;
}
// out[n, j] = scale * sum_t X[t, n] * G[t, j]   (j < R <= 32).  Pass 1: a workgroup owns 256 columns x a chunk of 256 token rows;
// the chunk's G rows sit in LDS as fp32 (every lane reads the same row: broadcast), a thread owns 4 columns (8-byte loads of X) and
// R accumulators per column, the 4 waves take the chunk's rows round-robin and meet in LDS in wave order -> partial[chunk][n][R].
// Pass 2 adds the chunks in ascending order (fixed summation order: bit-reproducible).
constexpr int TN_MAXR = 32, TN_CHUNK = 256;
template <int R>
__global__ __launch_bounds__(256) void tn_skinny_partial_kernel(const bf16_t* __restrict__ X, int64_t ldx, const bf16_t* __restrict__ G,
                                                                int64_t ldg, float* __restrict__ partial, int64_t T, int N, const int* __restrict__ rows_dev) {
  if (rows_dev) T = min(T, (int64_t)*rows_dev);
  __shared__ float gs[TN_CHUNK][R];
  __shared__ float red[3][64][4 * R + 1];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n0 = blockIdx.x * 256 + lane * 4;
  const int64_t t0 = (int64_t)blockIdx.y * TN_CHUNK;
  const int rows = (int)max((int64_t)0, min((int64_t)TN_CHUNK, T - t0));
  for (int i = threadIdx.x; i < rows * (R / 8); i += 256) {
    const int rr = i / (R / 8), c8 = (i % (R / 8)) * 8;
    const bf16x8 gv = *reinterpret_cast<const bf16x8*>(G + (t0 + rr) * ldg + c8);
#pragma unroll
    for (int j = 0; j < 8; ++j) gs[rr][c8 + j] = (float)gv[j];
  }
  __syncthreads();
  float acc[4][R];
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int j = 0; j < R; ++j) acc[c][j] = 0.f;
  const bool full = n0 + 4 <= N;
  for (int rr = wave; rr < rows; rr += 4) {
    float xv[4] = {0.f, 0.f, 0.f, 0.f};
    const bf16_t* xp = X + (t0 + rr) * ldx + n0;
    if (full) {
      const bf16x4 v = *reinterpret_cast<const bf16x4*>(xp);
#pragma unroll
      for (int c = 0; c < 4; ++c) xv[c] = (float)v[c];
    } else {
#pragma unroll
      for (int c = 0; c < 4; ++c) if (n0 + c < N) xv[c] = (float)xp[c];
    }
#pragma unroll
    for (int j4 = 0; j4 < R; j4 += 4) {
      const f32x4 g4 = *reinterpret_cast<const f32x4*>(&gs[rr][j4]);
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[c][j4 + j] = fmaf(xv[c], g4[j], acc[c][j4 + j]);
    }
  }
  if (wave > 0) {
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int j = 0; j < R; ++j) red[wave - 1][lane][c * R + j] = acc[c][j];
  }
  __syncthreads();
  if (wave == 0) {
    float* po = partial + ((int64_t)blockIdx.y * N) * R;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      if (n0 + c >= N) continue;
#pragma unroll
      for (int j = 0; j < R; ++j)
        po[(int64_t)(n0 + c) * R + j] = ((acc[c][j] + red[0][lane][c * R + j]) + red[1][lane][c * R + j]) + red[2][lane][c * R + j];
    }
  }
}

// The same partial sums on the matrix cores (round 2): at R = 16 the scalar form above issues T * N * R fmas -- VALU-bound, 91 us on the
// gate|up gradient [5112, 22016] where the read alone is ~50 us.  Both MFMA operands are reductions over the TOKEN axis, i.e. transposes of
// the row-major X and G tiles: a workgroup stages 64 tokens x 256 columns of X (coalesced 16-byte loads, lora_dropout applied on the way
// when p > 0, so the forward need not store its dropped activations) and the 64 x R tile of G in XOR-swizzled LDS images and reads
// both as transposed fragments with ds_read_b64_tr_b16 (the attention backward's X^T recipe: lane (fr, fq), element e <-> token
// fq * 4 + (e & 3) + (e >> 2) * 16 of a 32-token block -- the same assignment on both operands).  A wave owns 64 columns x R; chunks of
// 256 tokens per workgroup as before, added by tn_skinny_reduce_kernel in ascending order.
typedef __attribute__((ext_vector_type(4))) short tn_s16x4;
typedef __attribute__((ext_vector_type(8))) short tn_s16x8;
template <int RB>                                           // row bytes of the tile
__device__ __forceinline__ bf16x8 tn_tr_frag(const char* tile, int kp, int n, int fr, int fq) {
  const int row = kp * 32 + fq * 4 + (fr >> 2);
  const int cb = n * 32 + (fr & 3) * 8;
  const char* p0 = tile + row * RB + ((((cb >> 4) ^ (row & 7)) << 4) | (cb & 15));
  const tn_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) tn_s16x4*)(p0));
  const tn_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) tn_s16x4*)(p0 + 16 * RB));
  const tn_s16x8 both = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8, both);
}
template <int RG>                                           // 16-wide rank groups: R <= 16 * RG
__global__ __launch_bounds__(256) void tn_skinny_mfma_kernel(const bf16_t* __restrict__ X, int64_t ldx, const bf16_t* __restrict__ G, int64_t ldg,
                                                             float* __restrict__ partial, int64_t T, int N, int R, float p, uint64_t seed, const int* __restrict__ rows_dev,
                                                             const uint8_t* __restrict__ keep_bits, int64_t ld_bits) {
  if (rows_dev) T = min(T, (int64_t)*rows_dev);        // device-side row count (an expert's routed rows): chunks beyond it write zeros
  __shared__ __attribute__((aligned(16))) char xt[64 * 512];      // [64 tokens][256 columns] bf16, 16-byte chunk c of row r at (c ^ (r & 7))
  __shared__ __attribute__((aligned(16))) char gt[64 * 128];      // [64 tokens][64 columns] bf16 (columns >= 16 * RG never read)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fq = lane >> 4;
  const int n_base = blockIdx.x * 256;
  const int64_t t0 = (int64_t)blockIdx.y * TN_CHUNK;
  const int rows = (int)max((int64_t)0, min((int64_t)TN_CHUNK, T - t0));
  const int nsteps = (rows + 63) / 64;
  const float keep_scale = 1.f / (1.f - p);
  const unsigned th = dropout_thresh(p);
  f32x4 acc[4][RG];
#pragma unroll
  for (int nf = 0; nf < 4; ++nf)
#pragma unroll
    for (int jf = 0; jf < RG; ++jf) acc[nf][jf] = f32x4{0.f, 0.f, 0.f, 0.f};
  // staging roles: X chunk (row i * 8 + tid / 32, 16-byte chunk tid % 32); G chunk idx = tid (+ 256): (row idx / (2 RG), chunk idx % (2 RG))
  const int xr = tid >> 5, xc = tid & 31;
  const int xn = n_base + xc * 8;
  bf16x8 xv[8], gv[(64 * 2 * RG + 255) / 256];
  unsigned km[8];                                          // p > 0: the keep bits of xv[i], applied when the step is written to LDS
  auto load_step = [&](int step) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int r = step * 64 + i * 8 + xr;
      xv[i] = bf16x8{};
      km[i] = 0;
      if (r < rows && xn < N) {
        xv[i] = *reinterpret_cast<const bf16x8*>(X + (t0 + r) * ldx + xn);
        // the mask does not depend on the loaded values: it is read / hashed while they travel, and applied a step later (round 5: applied
        // here it made every step wait for its own loads before the previous step's MFMAs — 54 us against 26 without dropout)
        if (p > 0.f) km[i] = keep_bits ? keep_bits[(t0 + r) * ld_bits + (xn >> 3)]
                                       : dropout_keep8(seed, ((uint64_t)(t0 + r) * (uint64_t)N + (uint64_t)xn) >> 2, th);
      }
    }
#pragma unroll
    for (int u = 0; u < (64 * 2 * RG + 255) / 256; ++u) {
      const int idx = tid + u * 256;
      const int r = step * 64 + idx / (2 * RG), c = idx % (2 * RG);
      gv[u] = bf16x8{};
      if (idx < 64 * 2 * RG && r < rows) gv[u] = *reinterpret_cast<const bf16x8*>(G + (t0 + r) * ldg + c * 8);
    }
  };
  if (nsteps > 0) load_step(0);
  for (int step = 0; step < nsteps; ++step) {
    __syncthreads();                                       // the previous step's fragment reads are done
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int r = i * 8 + xr;
      *reinterpret_cast<bf16x8*>(xt + r * 512 + ((xc ^ (r & 7)) << 4)) = p > 0.f ? dropout_apply8(xv[i], km[i], keep_scale) : xv[i];
    }
#pragma unroll
    for (int u = 0; u < (64 * 2 * RG + 255) / 256; ++u) {
      const int idx = tid + u * 256;
      const int r = idx / (2 * RG), c = idx % (2 * RG);
      if (idx < 64 * 2 * RG) *reinterpret_cast<bf16x8*>(gt + r * 128 + ((c ^ (r & 7)) << 4)) = gv[u];
    }
    __syncthreads();
    if (step + 1 < nsteps) load_step(step + 1);            // in flight under this step's MFMAs
#pragma unroll
    for (int kp = 0; kp < 2; ++kp) {
      bf16x8 gf[RG];
#pragma unroll
      for (int jf = 0; jf < RG; ++jf) gf[jf] = tn_tr_frag<128>(gt, kp, jf, fr, fq);
#pragma unroll
      for (int nf = 0; nf < 4; ++nf) {
        const bf16x8 xf = tn_tr_frag<512>(xt, kp, wave * 4 + nf, fr, fq);
#pragma unroll
        for (int jf = 0; jf < RG; ++jf) acc[nf][jf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf, gf[jf], acc[nf][jf], 0, 0, 0);
      }
    }
  }
  // acc[nf][jf][r] = sum_t X[t, n] G[t, j] with n = n_base + (wave * 4 + nf) * 16 + fq * 4 + r, j = jf * 16 + fr
  float* po = partial + ((int64_t)blockIdx.y * N) * R;
#pragma unroll
  for (int nf = 0; nf < 4; ++nf)
#pragma unroll
    for (int jf = 0; jf < RG; ++jf)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = n_base + (wave * 4 + nf) * 16 + fq * 4 + r, j = jf * 16 + fr;
        if (n < N && j < R) po[(int64_t)n * R + j] = acc[nf][jf][r];
      }
}

__global__ void tn_skinny_reduce_kernel(const float* __restrict__ partial, float* __restrict__ out, int64_t NR, int chunks, float scale) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= NR) return;
  float s = 0.f;
  for (int c = 0; c < chunks; ++c) s += partial[(int64_t)c * NR + i];
  out[i] = scale * s;
}

// d_logits[i, v] = g * (exp(logit - lse_i) - [v == label_i]); one block per supervised row; bf16 output, columns >= V zeroed up to ldo
__global__ __launch_bounds__(256) void ce_rows_bwd_kernel(const float* __restrict__ logits, const int64_t* __restrict__ labels,
                                                          const float* __restrict__ gscale, float gconst, bf16_t* __restrict__ out, int V,
                                                          int64_t ldl, int64_t ldo) {
  __shared__ float red[16];
  const int64_t row = blockIdx.x;
  const float* lr = logits + row * ldl;
  float mx = -INFINITY;
  for (int v = threadIdx.x; v < V; v += 256) mx = fmaxf(mx, lr[v]);
  mx = block_max(mx, red);
  float s = 0.f;
  for (int v = threadIdx.x; v < V; v += 256) s += expf(lr[v] - mx);
  s = block_sum(s, red);
  const float g = gconst * (gscale ? gscale[0] : 1.f);
  const int64_t lab = labels[row];
  const float inv = 1.f / s;
  for (int v = threadIdx.x; v < ldo; v += 256) {
    float d = 0.f;
    if (v < V) d = g * (expf(lr[v] - mx) * inv - (v == lab ? 1.f : 0.f));
    out[row * ldo + v] = (bf16_t)d;
  }
}

__global__ void scatter_rows_f32_bf16_kernel(const float* __restrict__ g, const int64_t* __restrict__ rows, bf16_t* __restrict__ out, int64_t n,
                                             int d) {
  const int per_row = d / 4;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n * per_row) return;
  const int64_t i = idx / per_row;
  const int c = (int)(idx % per_row) * 4;
  const f32x4 v = *reinterpret_cast<const f32x4*>(g + i * d + c);
  *reinterpret_cast<bf16x4*>(out + rows[i] * d + c) = bf16x4{(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
}

__global__ void dropout_bf16_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y, int64_t n, float p, uint64_t seed) {
  const int64_t i8 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 8;
  if (i8 >= n) return;
  const float keep_scale = 1.f / (1.f - p);
  const unsigned th = dropout_thresh(p);
  if (i8 + 8 <= n) {
    const bf16x8 v = *reinterpret_cast<const bf16x8*>(x + i8);
    bf16x8 o;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const uint64_t bits = dropout_bits4(seed, (uint64_t)(i8 >> 2) + h);
#pragma unroll
      for (int j = 0; j < 4; ++j) o[h * 4 + j] = (bf16_t)(((unsigned)(bits >> (16 * j)) & 0xffffu) >= th ? (float)v[h * 4 + j] * keep_scale : 0.f);
    }
    *reinterpret_cast<bf16x8*>(y + i8) = o;
  } else {
    for (int64_t i = i8; i < n; ++i) {
      const uint64_t bits = dropout_bits4(seed, (uint64_t)(i >> 2));
      y[i] = (bf16_t)(((unsigned)(bits >> (16 * (i & 3))) & 0xffffu) >= th ? (float)x[i] * keep_scale : 0.f);
    }
  }
}

// LoRA down-projection of one (fused) adapter group with the dropout inline: t[token, j] = bf16(sum_k drop(x)[token, k] A[j, k]) for the 64
// padded rank columns (zeros beyond 16 * RG), and optionally the dropped activations themselves (the wgrad's operand).  t is written as
// the K-EXTENSION of the projection's input: the base GEMM then runs over [x | t] and [W | scaling B] (K + 64 deep) and produces
// x W^T + scaling (drop(x) A^T) B^T in one pass -- no [tokens, out] round trip for the adapter branch (peft lora.Linear.forward:
// result + lora_B(lora_A(dropout(x))) * scaling).  A workgroup owns 16 tokens; its 8 waves split K and are summed in wave order through
// LDS (fixed order); a lane's x fragment is 16 bytes of one token row = the MFMA operand layout, so x goes from global memory straight
// into v_mfma_f32_16x16x32_bf16 (issued as (A rows, tokens): a lane ends up with four consecutive rank columns of a token).  Same
// mask as mp_dropout_bf16 over the contiguous [tokens, K] tensor (index token * K + k).
template <int RG>
__global__ __launch_bounds__(512) void lora_down_kernel(const bf16_t* __restrict__ x, int64_t ldx, const bf16_t* __restrict__ A, int64_t lda,
                                                        bf16_t* __restrict__ t, int64_t ldt, bf16_t* __restrict__ xd, int64_t ldxd, int T, int K,
                                                        float p, uint64_t seed, float alpha, const int* __restrict__ rows_dev) {
  if (rows_dev) T = min(T, *rows_dev);                     // device-side row count (an expert's routed rows on its capacity slab)
  if ((int)blockIdx.x * 16 >= T) return;
  __shared__ float red[8][RG][256];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fq = lane >> 4;
  const int tok0 = blockIdx.x * 16;
  const int nk = K / 64;                                     // 64-deep K blocks; a lane reads 32 contiguous bytes of its row per block
  const int ks0 = (int)((int64_t)wave * nk / 8), ks1 = (int)((int64_t)(wave + 1) * nk / 8);
  const float keep_scale = 1.f / (1.f - p);
  const unsigned th = dropout_thresh(p);
  f32x4 acc[RG];
#pragma unroll
  for (int rg = 0; rg < RG; ++rg) acc[rg] = f32x4{0.f, 0.f, 0.f, 0.f};
  const bool live = tok0 + fr < T;
  const int tok = min(tok0 + fr, T - 1);
  // MFMA K slot (fq, j) of the block's first / second step holds k = kb + fq * 16 + j resp. + 8 + j -- the same assignment for x and A, so
  // the dot products are over the same k; with it a token row's 128 bytes are read by four lanes x 32 contiguous bytes (whole lines)
  constexpr int U = 4;                                       // K blocks whose loads are in flight together
  for (int ks = ks0; ks < ks1; ks += U) {
    const int n = min(U, ks1 - ks);
    bf16x8 af[U][RG][2], xv[U][2];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (u < n) {
        const int k = (ks + u) * 64 + fq * 16;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
          for (int rg = 0; rg < RG; ++rg) af[u][rg][h] = *reinterpret_cast<const bf16x8*>(A + (int64_t)(rg * 16 + fr) * lda + k + h * 8);
          xv[u][h] = *reinterpret_cast<const bf16x8*>(x + (int64_t)tok * ldx + k + h * 8);
        }
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (u < n) {
        const int k = (ks + u) * 64 + fq * 16;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          if (p > 0.f) {
            const uint64_t i4 = ((uint64_t)tok * (uint64_t)K + (uint64_t)(k + h * 8)) >> 2;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
              const uint64_t bits = dropout_bits4(seed, i4 + q);
#pragma unroll
              for (int j = 0; j < 4; ++j)
                xv[u][h][q * 4 + j] = (bf16_t)(((unsigned)(bits >> (16 * j)) & 0xffffu) >= th ? (float)xv[u][h][q * 4 + j] * keep_scale : 0.f);
            }
            if (xd && live) *reinterpret_cast<bf16x8*>(xd + (int64_t)tok * ldxd + k + h * 8) = xv[u][h];
          }
#pragma unroll
          for (int rg = 0; rg < RG; ++rg) acc[rg] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[u][rg][h], xv[u][h], acc[rg], 0, 0, 0);
        }
      }
    }
  }
#pragma unroll
  for (int rg = 0; rg < RG; ++rg) *reinterpret_cast<f32x4*>(&red[wave][rg][lane * 4]) = acc[rg];
  __syncthreads();
  // RG x 64 lane-quads to finish (summing the 8 waves in order); everyone helps with the zeros beyond the rank columns
  for (int q = tid; q < RG * 64; q += 512) {
    const int rg = q / 64, ln = q & 63;
    f32x4 sum = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int w = 0; w < 8; ++w) sum += *reinterpret_cast<const f32x4*>(&red[w][rg][ln * 4]);
    sum *= alpha;
    const int tk = tok0 + (ln & 15);
    if (tk < T) *reinterpret_cast<bf16x4*>(t + (int64_t)tk * ldt + rg * 16 + (ln >> 4) * 4) = bf16x4{(bf16_t)sum[0], (bf16_t)sum[1], (bf16_t)sum[2], (bf16_t)sum[3]};
  }
  constexpr int ZC = (64 - 16 * RG) / 4;                    // zero quads per token beyond the rank columns
  for (int q = tid; q < 16 * ZC; q += 512) {
    const int tk = tok0 + q / (ZC > 0 ? ZC : 1), c = 16 * RG + (q % (ZC > 0 ? ZC : 1)) * 4;
    if (ZC > 0 && tk < T) *reinterpret_cast<bf16x4*>(t + (int64_t)tk * ldt + c) = bf16x4{(bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f};
  }
}

// Round 5: the backward's two products over the SAME operand in one pass.  For an adapter with output gradient dY [T, N] the backward needs
// dB = dY^T t (tn_skinny_mfma_kernel: [N, R], a reduction over tokens) and dt = dY B (lora_down_staged_kernel: [T, R], a reduction over
// columns) — two kernels that stage the identical [64 tokens x 256 columns] tile of dY in the identical LDS image, i.e. dY was read twice
// (225 MB each time for the fused gate|up adapter).  Here the workgroup of tn_skinny_mfma_kernel (256 columns x a chunk of 256 tokens, four
// steps of 64) also keeps the [16 RG x 256] piece of B^T for ITS columns in LDS and, per step, multiplies the tile row-major against it: the
// 64 x 16 RG partial of dt over this column block goes to dt_partial[column block][token][16 RG] (fp32), which lora_down_finish_kernel adds
// over the column blocks in ascending order (fixed order: bit-reproducible), scales and rounds — the finish launch the split-K form needs
// anyway.  No dropout on this side (dY is a gradient).  dB: tn_skinny_mfma_kernel's arithmetic, bit for bit.
template <int RG>
__global__ __launch_bounds__(256) void tn_skinny_down_mfma_kernel(const bf16_t* __restrict__ X, int64_t ldx, const bf16_t* __restrict__ G, int64_t ldg,
                                                                  float* __restrict__ partial, const bf16_t* __restrict__ Bt, int64_t ldb,
                                                                  float* __restrict__ dt_partial, int64_t T, int N, int R) {
  __shared__ __attribute__((aligned(16))) char xt[64 * 512];      // [64 tokens][256 columns] bf16, 16-byte chunk c of row r at (c ^ (r & 7))
  __shared__ __attribute__((aligned(16))) char gt[64 * 128];      // [64 tokens][64 columns] bf16 (columns >= 16 * RG never read)
  __shared__ __attribute__((aligned(16))) char at[16 * RG * 512]; // [16 RG rank rows of B^T][this block's 256 columns]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fq = lane >> 4;
  const int n_base = blockIdx.x * 256;
  const int64_t t0 = (int64_t)blockIdx.y * TN_CHUNK;
  const int rows = (int)max((int64_t)0, min((int64_t)TN_CHUNK, T - t0));
  const int nsteps = (rows + 63) / 64;
  f32x4 acc[4][RG];
#pragma unroll
  for (int nf = 0; nf < 4; ++nf)
#pragma unroll
    for (int jf = 0; jf < RG; ++jf) acc[nf][jf] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int xr = tid >> 5, xc = tid & 31;
  const int xn = n_base + xc * 8;
  // B^T piece of this column block (columns beyond N: zeros), once per workgroup
#pragma unroll
  for (int u = 0; u < (16 * RG * 32 + 255) / 256; ++u) {
    const int idx = tid + u * 256;
    if (idx < 16 * RG * 32) {
      const int r = idx >> 5, c = idx & 31;
      bf16x8 v = bf16x8{};
      if (n_base + c * 8 < N) v = *reinterpret_cast<const bf16x8*>(Bt + (int64_t)r * ldb + n_base + c * 8);
      *reinterpret_cast<bf16x8*>(at + r * 512 + ((c ^ (r & 7)) << 4)) = v;
    }
  }
  bf16x8 xv[8], gv[(64 * 2 * RG + 255) / 256];
  auto load_step = [&](int step) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int r = step * 64 + i * 8 + xr;
      xv[i] = bf16x8{};
      if (r < rows && xn < N) xv[i] = *reinterpret_cast<const bf16x8*>(X + (t0 + r) * ldx + xn);
    }
#pragma unroll
    for (int u = 0; u < (64 * 2 * RG + 255) / 256; ++u) {
      const int idx = tid + u * 256;
      const int r = step * 64 + idx / (2 * RG), c = idx % (2 * RG);
      gv[u] = bf16x8{};
      if (idx < 64 * 2 * RG && r < rows) gv[u] = *reinterpret_cast<const bf16x8*>(G + (t0 + r) * ldg + c * 8);
    }
  };
  if (nsteps > 0) load_step(0);
  for (int step = 0; step < nsteps; ++step) {
    __syncthreads();                                       // the previous step's fragment reads are done (and, first time, `at` is written)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int r = i * 8 + xr;
      *reinterpret_cast<bf16x8*>(xt + r * 512 + ((xc ^ (r & 7)) << 4)) = xv[i];
    }
#pragma unroll
    for (int u = 0; u < (64 * 2 * RG + 255) / 256; ++u) {
      const int idx = tid + u * 256;
      const int r = idx / (2 * RG), c = idx % (2 * RG);
      if (idx < 64 * 2 * RG) *reinterpret_cast<bf16x8*>(gt + r * 128 + ((c ^ (r & 7)) << 4)) = gv[u];
    }
    __syncthreads();
    if (step + 1 < nsteps) load_step(step + 1);            // in flight under this step's MFMAs
#pragma unroll
    for (int kp = 0; kp < 2; ++kp) {
      bf16x8 gf[RG];
#pragma unroll
      for (int jf = 0; jf < RG; ++jf) gf[jf] = tn_tr_frag<128>(gt, kp, jf, fr, fq);
#pragma unroll
      for (int nf = 0; nf < 4; ++nf) {
        const bf16x8 xf = tn_tr_frag<512>(xt, kp, wave * 4 + nf, fr, fq);
#pragma unroll
        for (int jf = 0; jf < RG; ++jf) acc[nf][jf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf, gf[jf], acc[nf][jf], 0, 0, 0);
      }
    }
    // dt partial of this step's 64 tokens over this block's 256 columns: wave w owns tokens w * 16 .. + 15 (lora_down_staged_kernel's reads)
    f32x4 dacc[RG];
#pragma unroll
    for (int rg = 0; rg < RG; ++rg) dacc[rg] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
      const int c = kk * 4 + fq;
      const int xrow = wave * 16 + fr;
      const bf16x8 xf = *reinterpret_cast<const bf16x8*>(xt + xrow * 512 + ((c ^ (xrow & 7)) << 4));
#pragma unroll
      for (int rg = 0; rg < RG; ++rg) {
        const int arow = rg * 16 + fr;
        const bf16x8 af = *reinterpret_cast<const bf16x8*>(at + arow * 512 + ((c ^ (arow & 7)) << 4));
        dacc[rg] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, xf, dacc[rg], 0, 0, 0);
      }
    }
    const int64_t tok = t0 + step * 64 + wave * 16 + fr;
    if (tok < T) {
#pragma unroll
      for (int rg = 0; rg < RG; ++rg)
        *reinterpret_cast<f32x4*>(dt_partial + ((int64_t)blockIdx.x * T + tok) * (16 * RG) + rg * 16 + fq * 4) = dacc[rg];
    }
  }
  float* po = partial + ((int64_t)blockIdx.y * N) * R;
#pragma unroll
  for (int nf = 0; nf < 4; ++nf)
#pragma unroll
    for (int jf = 0; jf < RG; ++jf)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = n_base + (wave * 4 + nf) * 16 + fq * 4 + r, j = jf * 16 + fr;
        if (n < N && j < R) po[(int64_t)n * R + j] = acc[nf][jf][r];
      }
}

// The same down-projection with the x tile STAGED THROUGH LDS (K % 256 == 0): the kernel above feeds the MFMA from global memory with
// fragment-shaped loads (16 bytes of sixteen different rows per instruction: nothing coalesces, 2.5-3 TB/s); here a workgroup loads a
// [64 tokens x 256] tile with whole-row 16-byte accesses (the dropout and the optional store of the dropped values happen on the way,
// also coalesced), writes it and the [16 RG x 256] piece of A into XOR-swizzled LDS images and reads both as plain ds_read_b128
// fragments.  One token group per wave; K is split over gridDim.y workgroups (80 token blocks alone would leave two thirds of the chip
// idle) whose fp32 partials are added in split order by lora_down_finish_kernel (scaling, rounding, zero padding to 64 columns).
template <int RG>
__global__ __launch_bounds__(256) void lora_down_staged_kernel(const bf16_t* __restrict__ x, int64_t ldx, const bf16_t* __restrict__ A, int64_t lda,
                                                               float* __restrict__ partial, bf16_t* __restrict__ xd, int64_t ldxd, int T, int K,
                                                               float p, uint64_t seed, const int* __restrict__ rows_dev,
                                                               uint8_t* __restrict__ keep_bits, int64_t ld_bits) {
  __shared__ __attribute__((aligned(16))) char xt[64 * 512];            // [64 tokens][256 k] bf16, chunk c of row r at c ^ (r & 7)
  __shared__ __attribute__((aligned(16))) char at[16 * RG * 512];       // [16 RG rank rows][256 k]
  if (rows_dev) T = min(T, *rows_dev);
  const int tok0 = blockIdx.x * 64;
  if (tok0 >= T) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fq = lane >> 4;
  const int nsteps_all = K / 256;
  const int st0 = (int)((int64_t)blockIdx.y * nsteps_all / gridDim.y), st1 = (int)((int64_t)(blockIdx.y + 1) * nsteps_all / gridDim.y);
  const float keep_scale = 1.f / (1.f - p);
  const unsigned th = dropout_thresh(p);
  const int xr = tid >> 5, xc = tid & 31;                                // staging role: row i * 8 + xr, 16-byte chunk xc
  f32x4 acc[RG];
#pragma unroll
  for (int rg = 0; rg < RG; ++rg) acc[rg] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8 xv[8], av[(16 * RG * 32 + 255) / 256];
  unsigned km[8];                                          // p > 0: the keep bits of xv[i], applied when the step is written to LDS
  int k_held = 0;                                          // the K offset of the step held in xv (for the optional store of the dropped values)
  auto load_step = [&](int st) {
    const int k = st * 256 + xc * 8;
    k_held = k;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int tok = tok0 + i * 8 + xr;
      xv[i] = bf16x8{};
      km[i] = 0;
      if (tok < T) {
        xv[i] = *reinterpret_cast<const bf16x8*>(x + (int64_t)tok * ldx + k);
        if (p > 0.f) {                                     // hashed while the loads travel, applied a step later (see tn_skinny_mfma_kernel)
          km[i] = dropout_keep8(seed, ((uint64_t)tok * (uint64_t)K + (uint64_t)k) >> 2, th);
          if (keep_bits) keep_bits[(int64_t)tok * ld_bits + (k >> 3)] = (uint8_t)km[i];  // (each (token, chunk) belongs to exactly one workgroup of the K split)
        }
      }
    }
#pragma unroll
    for (int u = 0; u < (16 * RG * 32 + 255) / 256; ++u) {
      const int idx = tid + u * 256;
      av[u] = bf16x8{};
      if (idx < 16 * RG * 32) av[u] = *reinterpret_cast<const bf16x8*>(A + (int64_t)(idx >> 5) * lda + st * 256 + (idx & 31) * 8);
    }
  };
  if (st0 < st1) load_step(st0);
  for (int st = st0; st < st1; ++st) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int r = i * 8 + xr;
      const bf16x8 v = p > 0.f ? dropout_apply8(xv[i], km[i], keep_scale) : xv[i];
      *reinterpret_cast<bf16x8*>(xt + r * 512 + ((xc ^ (r & 7)) << 4)) = v;
      if (p > 0.f && xd && tok0 + r < T) *reinterpret_cast<bf16x8*>(xd + (int64_t)(tok0 + r) * ldxd + k_held) = v;
    }
#pragma unroll
    for (int u = 0; u < (16 * RG * 32 + 255) / 256; ++u) {
      const int idx = tid + u * 256;
      if (idx < 16 * RG * 32) {
        const int r = idx >> 5, c = idx & 31;
        *reinterpret_cast<bf16x8*>(at + r * 512 + ((c ^ (r & 7)) << 4)) = av[u];
      }
    }
    __syncthreads();
    if (st + 1 < st1) load_step(st + 1);
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
      const int c = kk * 4 + fq;
      const int xrow = wave * 16 + fr;
      const bf16x8 xf = *reinterpret_cast<const bf16x8*>(xt + xrow * 512 + ((c ^ (xrow & 7)) << 4));
#pragma unroll
      for (int rg = 0; rg < RG; ++rg) {
        const int arow = rg * 16 + fr;
        const bf16x8 af = *reinterpret_cast<const bf16x8*>(at + arow * 512 + ((c ^ (arow & 7)) << 4));
        acc[rg] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, xf, acc[rg], 0, 0, 0);
      }
    }
  }
  // acc[rg][r] = partial t[token tok0 + wave * 16 + fr][rank rg * 16 + fq * 4 + r]
  const int tok = tok0 + wave * 16 + fr;
  if (tok < T) {
#pragma unroll
    for (int rg = 0; rg < RG; ++rg)
      *reinterpret_cast<f32x4*>(partial + ((int64_t)blockIdx.y * T + tok) * (16 * RG) + rg * 16 + fq * 4) = acc[rg];
  }
}

__global__ void lora_down_finish_kernel(const float* __restrict__ partial, bf16_t* __restrict__ t, int64_t ldt, int T, int RW, int splits, float alpha,
                                        const int* __restrict__ rows_dev) {
  if (rows_dev) T = min(T, *rows_dev);
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;      // (token, 4-column group of the 64)
  if (idx >= (int64_t)T * 16) return;
  const int tok = (int)(idx >> 4), c = (int)(idx & 15) * 4;
  f32x4 sum = {0.f, 0.f, 0.f, 0.f};
  if (c < RW) {
    // eight partials in flight per thread, added in ascending order (mp_tn_skinny_down_f32 brings one partial per 256-column block: 86 of them)
    const float* src = partial + (int64_t)tok * RW + c;
    const int64_t stride = (int64_t)T * RW;
    for (int s0 = 0; s0 < splits; s0 += 8) {
      f32x4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = (s0 + u < splits) ? *reinterpret_cast<const f32x4*>(src + (s0 + u) * stride) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int u = 0; u < 8; ++u) if (s0 + u < splits) sum += v[u];
    }
  }
  sum *= alpha;
  *reinterpret_cast<bf16x4*>(t + (int64_t)tok * ldt + c) = bf16x4{(bf16_t)sum[0], (bf16_t)sum[1], (bf16_t)sum[2], (bf16_t)sum[3]};
}

// Backward of the adapter branch into the projection's input gradient, in one pass over it: out = dx + dropout(bf16(dt A)) with the
// forward's mask -- the thin GEMM (dt A^T^T -> [tokens, in]), the dropout pass over its output and the add were six passes over a
// [tokens, in] tensor; this is one read and one write.  A lane owns 8 consecutive input features of a 512-wide chunk (16-byte accesses,
// 1 KiB per wave per token row) and keeps their rank vectors in registers (AT [in, 64]: a feature's rank values are contiguous, so two
// ranks at a time go through v_dot2c_f32_bf16 against the token's (uniform) dt pairs); the four waves of a workgroup take different
// token ranges of the same chunk.  Rounding points as the three kernels it replaces: the product rounded to bf16 (the GEMM's output),
// scaled by 1 / (1 - p) and rounded (mp_dropout_bf16), added to dx and rounded (mp_add3_bf16).
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
// SW: the SwiGLU backward in the same pass (the down projection's input gradient is consumed by exactly one kernel, mp_swiglu_pair_bwd_bf16):
// dx + dropout(bf16(dt A)) is rounded to bf16 exactly as the stored form was, then d gate / d up of the same 8 channels are written to
// dgu [tokens, 2 K] (gate|up interleaved in blocks of 32) from gu — one pass over d_act instead of a read-modify-write and a read.
template <int RP, bool SW = false>                           // rank pairs: R / 2 in {4, 8, 16}
__global__ __launch_bounds__(256) void lora_up_add_kernel(const bf16_t* __restrict__ dt, int64_t lddt, const bf16_t* __restrict__ AT,
                                                          const bf16_t* __restrict__ dx, int64_t lddx, bf16_t* __restrict__ out, int64_t ldo,
                                                          int T, int K, float p, uint64_t seed, int tpw, const int* __restrict__ rows_dev,
                                                          const uint8_t* __restrict__ keep_bits, int64_t ld_bits,
                                                          const bf16_t* __restrict__ gu = nullptr, bf16_t* __restrict__ dgu = nullptr) {
  if (rows_dev) T = min(T, *rows_dev);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if ((int)(blockIdx.y * 4 + wave) * tpw >= T) return;
  const int k0 = (blockIdx.x * 64 + lane) * 8;
  const bool col_live = k0 < K;
  const int kc = col_live ? k0 : 0;
  bf16x2_t a[8][RP];
#pragma unroll
  for (int j = 0; j < 8; ++j)
#pragma unroll
    for (int q = 0; q < RP / 4; ++q) {
      const bf16x8 v = *reinterpret_cast<const bf16x8*>(AT + (int64_t)(kc + j) * 64 + q * 8);
#pragma unroll
      for (int r = 0; r < 4; ++r) a[j][q * 4 + r] = bf16x2_t{v[2 * r], v[2 * r + 1]};
    }
  const float keep_scale = 1.f / (1.f - p);
  const unsigned th = dropout_thresh(p);
  const int t_beg = (blockIdx.y * 4 + wave) * tpw, t_end = min(T, t_beg + tpw);
  constexpr int U = 4;                                       // tokens whose loads are in flight together
  for (int tb = t_beg; tb < t_end; tb += U) {
    const int n = min(U, t_end - tb);
    bf16x8 dv[U], dtv[U][RP / 4], gv[SW ? U : 1], uv[SW ? U : 1];
    unsigned km[U];
    const int64_t gcol = (int64_t)(kc >> 5) * 64 + (kc & 31);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      km[u] = 0xffu;
      if (u < n) {
        dv[u] = *reinterpret_cast<const bf16x8*>(dx + (int64_t)(tb + u) * lddx + kc);
        if (p > 0.f && keep_bits) km[u] = keep_bits[(int64_t)(tb + u) * ld_bits + (kc >> 3)];
        if constexpr (SW) {
          gv[u] = *reinterpret_cast<const bf16x8*>(gu + (int64_t)(tb + u) * 2 * K + gcol);
          uv[u] = *reinterpret_cast<const bf16x8*>(gu + (int64_t)(tb + u) * 2 * K + gcol + 32);
        }
#pragma unroll
        for (int q = 0; q < RP / 4; ++q) dtv[u][q] = *reinterpret_cast<const bf16x8*>(dt + (int64_t)(tb + u) * lddt + q * 8);   // same address in every lane
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (u < n) {
        float acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = 0.f;
#pragma unroll
        for (int q = 0; q < RP / 4; ++q)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const bf16x2_t d2 = bf16x2_t{dtv[u][q][2 * r], dtv[u][q][2 * r + 1]};
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] = __builtin_amdgcn_fdot2_f32_bf16(a[j][q * 4 + r], d2, acc[j], false);
          }
        bf16x8 o;
        const unsigned m = (p > 0.f && !keep_bits) ? dropout_keep8(seed, ((uint64_t)(tb + u) * (uint64_t)K + (uint64_t)kc) >> 2, th) : km[u];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          float v = (float)(bf16_t)acc[j];
          if (p > 0.f) v = ((m >> j) & 1u) ? (float)(bf16_t)(v * keep_scale) : 0.f;
          o[j] = (bf16_t)((float)dv[u][j] + v);
        }
        if constexpr (SW) {
          bf16x8 dg, du;
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const float gf = (float)gv[u][j], uf = (float)uv[u][j], df = (float)o[j];
            const float sg = mp_sigmoid_fast(gf);
            du[j] = (bf16_t)(df * gf * sg);
            dg[j] = (bf16_t)(df * uf * sg * (1.f + gf * (1.f - sg)));
          }
          if (col_live) {
            *reinterpret_cast<bf16x8*>(dgu + (int64_t)(tb + u) * 2 * K + gcol) = dg;
            *reinterpret_cast<bf16x8*>(dgu + (int64_t)(tb + u) * 2 * K + gcol + 32) = du;
          }
        } else {
          if (col_live) *reinterpret_cast<bf16x8*>(out + (int64_t)(tb + u) * ldo + kc) = o;
        }
      }
    }
  }
}

// Round 5: mp_lora_up_add_bf16 folded into the RMSNorm backward that reads its result.  The gate|up adapter's input gradient
// d_h2' = bf16(d_h2 + dropout(bf16(dt A))) has ONE reader, the post-attention norm's backward: here that kernel forms d_h2' from d_h2, the
// token's dt row and the lane's eight A^T rank vectors (held in registers: a workgroup walks rows, 512 threads = one 16-byte chunk each at
// dim 4096) on its way in — one read-modify-write pass over [T, d] less per layer.  BIT-IDENTICAL with the two kernels: the adapter term is
// lora_up_add_kernel's dot2 chain and rounding points; the two row reductions reproduce rmsnorm_bwd_kernel's order (its thread t sums chunk
// t, then chunk 256 + t, element by element; waves 4-7 here continue the chains waves 0-3 started, lane for lane, so the wave and block sums
// see the same operands in the same order).
template <int RP>
__global__ __launch_bounds__(1024) void rmsnorm_bwd_up_kernel(const bf16_t* __restrict__ x, const float* __restrict__ w, const bf16_t* __restrict__ dy,
                                                              const bf16_t* __restrict__ add, bf16_t* __restrict__ dx, float eps, int64_t ldx, int64_t ldy,
                                                              int64_t lda, int64_t ldo, int rows, const bf16_t* __restrict__ dt, int64_t lddt,
                                                              const bf16_t* __restrict__ AT, float p, uint64_t seed, const uint8_t* __restrict__ keep_bits,
                                                              int64_t ld_bits) {
  // FOUR rows per trip: threads 256 s .. 256 s + 255 are rmsnorm_bwd_kernel's 256-thread row (thread t: the 16-byte chunks t and 256 + t, its
  // two block sums over four waves), so a trip costs that kernel's four barriers for four rows and a CU has 96 KB of loads in flight.  A^T sits
  // in LDS as [chunk c][channel j][rank piece q][thread]: 16 bytes per thread and (c, j, q), consecutive threads consecutive pieces
  // (conflict-free), 128 KiB at R = 16, the norm weight behind it.  (Rank vectors in registers, one row per 512-thread workgroup: 64 registers
  // per lane, one workgroup per CU, six barriers per row to reproduce the summation order — 77-95 us against 78 for the two kernels.)
  constexpr int dim = 4096, NQ = RP / 4;
  extern __shared__ __attribute__((aligned(16))) char up_lds[];
  uint4* alds = reinterpret_cast<uint4*>(up_lds);            // [2][8][NQ][256]
  float* wlds = reinterpret_cast<float*>(up_lds + 2 * 8 * NQ * 256 * 16);      // [dim]
  __shared__ float red[4][4];
  const int tid = threadIdx.x, lane = tid & 63, slot = tid >> 8, t = tid & 255, ws = (tid >> 6) & 3;
  for (int idx = tid; idx < 2 * 8 * NQ * 256; idx += 1024) {
    const int c = idx / (8 * NQ * 256), j = (idx / (NQ * 256)) & 7, q = (idx / 256) % NQ, tt = idx & 255;
    alds[idx] = *reinterpret_cast<const uint4*>(AT + (int64_t)((c * 256 + tt) * 8 + j) * 64 + q * 8);
  }
  for (int idx = tid; idx < dim; idx += 1024) wlds[idx] = w[idx];
  const float keep_scale = 1.f / (1.f - p);
  const unsigned th = dropout_thresh(p);
  __syncthreads();
  for (int base = blockIdx.x * 4; base < rows; base += gridDim.x * 4) {
    const int row = base + slot;
    const bool live = row < rows;
    bf16x8 xv[2], gv[2], av[2], dtv[NQ];
    unsigned m[2] = {0xffu, 0xffu};
#pragma unroll
    for (int c = 0; c < 2; ++c) { xv[c] = bf16x8{}; gv[c] = bf16x8{}; av[c] = bf16x8{}; }
#pragma unroll
    for (int q = 0; q < NQ; ++q) dtv[q] = bf16x8{};
    if (live) {
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const int i = (c * 256 + t) * 8;
        xv[c] = *reinterpret_cast<const bf16x8*>(x + (int64_t)row * ldx + i);
        gv[c] = *reinterpret_cast<const bf16x8*>(dy + (int64_t)row * ldy + i);
        if (add) av[c] = *reinterpret_cast<const bf16x8*>(add + (int64_t)row * lda + i);
        if (p > 0.f) m[c] = keep_bits ? keep_bits[(int64_t)row * ld_bits + c * 256 + t] : dropout_keep8(seed, ((uint64_t)row * (uint64_t)dim + (uint64_t)i) >> 2, th);
      }
#pragma unroll
      for (int q = 0; q < NQ; ++q) dtv[q] = *reinterpret_cast<const bf16x8*>(dt + (int64_t)row * lddt + q * 8);      // the same address in every lane
    }
    // ---- d_h2' = d_h2 + dropout(bf16(dt A)): lora_up_add_kernel's arithmetic (rank pairs ascending per channel), written over gv
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float acc = 0.f;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
          const uint4 piece = alds[((c * 8 + j) * NQ + q) * 256 + t];
          const unsigned pr[4] = {piece.x, piece.y, piece.z, piece.w};
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const bf16x2_t d2 = bf16x2_t{dtv[q][2 * r], dtv[q][2 * r + 1]};
            acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, pr[r]), d2, acc, false);
          }
        }
        float v = (float)(bf16_t)acc;
        if (p > 0.f) v = ((m[c] >> j) & 1u) ? (float)(bf16_t)(v * keep_scale) : 0.f;
        gv[c][j] = (bf16_t)((float)gv[c][j] + v);
      }
    // ---- from here on: rmsnorm_bwd_kernel on (xv, gv), per row slot
    float ss = 0.f;
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int j = 0; j < 8; ++j) { const float f = (float)xv[c][j]; ss += f * f; }
    ss = wave_sum(ss);
    __syncthreads();                                         // the previous trip's red[] reads are done
    if (lane == 0) red[slot][ws] = ss;
    __syncthreads();
    ss = ((0.f + red[slot][0]) + red[slot][1]) + red[slot][2] + red[slot][3];
    const float rs = rsqrtf(ss / (float)dim + eps);
    float dot = 0.f;
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int j = 0; j < 8; ++j) dot += (float)gv[c][j] * wlds[(c * 256 + t) * 8 + j] * ((float)xv[c][j] * rs);
    dot = wave_sum(dot);
    __syncthreads();
    if (lane == 0) red[slot][ws] = dot;
    __syncthreads();
    dot = (((0.f + red[slot][0]) + red[slot][1]) + red[slot][2] + red[slot][3]) / (float)dim;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const int i = (c * 256 + t) * 8;
      bf16x8 o;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float xh = (float)xv[c][j] * rs;
        float v = rs * ((float)gv[c][j] * wlds[i + j] - xh * dot);
        if (add) v += (float)av[c][j];
        o[j] = (bf16_t)v;
      }
      if (live) *reinterpret_cast<bf16x8*>(dx + (int64_t)row * ldo + i) = o;
    }
  }
}

// Round 5: the dense MLP's backward between its two input-gradient GEMMs as ONE kernel.  mp_lora_up_add_swiglu_bwd_bf16 writes
// d gate|up [T, 2 ff] (225 MB at 7B) and mp_tn_skinny_down_f32 reads it straight back for the gate|up adapter's dB = d_gu^T t and dt = d_gu B.
// Here the workgroup of tn_skinny_down_mfma_kernel (256 columns of d_gu = 128 channels x a chunk of 256 tokens) PRODUCES its tile instead of
// loading it: a thread owns one 8-channel group (its A^T rank vectors in registers) and four token rows per step, forms
// d_act' = bf16(d_act + dropout(bf16(dt_down A_down))) and d gate / d up exactly as lora_up_add_kernel<., true> does, stores them to d_gu (the
// main input-gradient GEMM reads it) AND into the LDS image the two MFMA products read.  Same bits as the two kernels for d_gu, dB and dt.
template <int RP, int RG>
__global__ __launch_bounds__(256) void swiglu_bwd_skinny_kernel(const bf16_t* __restrict__ dact, int64_t lddact, const bf16_t* __restrict__ gu, bf16_t* __restrict__ dgu,
                                                                const bf16_t* __restrict__ dtd, int64_t lddtd, const bf16_t* __restrict__ ATd, float p, uint64_t seed,
                                                                const uint8_t* __restrict__ keep_bits, int64_t ld_bits,
                                                                const bf16_t* __restrict__ G, int64_t ldg, float* __restrict__ partial, const bf16_t* __restrict__ Bt,
                                                                int64_t ldb, float* __restrict__ dt_partial, int64_t T, int ff, int R) {
  __shared__ __attribute__((aligned(16))) char xt[64 * 512];      // [64 tokens][256 columns of d_gu] bf16, 16-byte chunk c of row r at (c ^ (r & 7))
  __shared__ __attribute__((aligned(16))) char gt[64 * 128];      // [64 tokens][64 columns] of the forward's t (columns >= 16 * RG never read)
  __shared__ __attribute__((aligned(16))) char at[16 * RG * 512]; // [16 RG rank rows of B^T][this block's 256 columns]
  const int N = 2 * ff;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fq = lane >> 4;
  const int n_base = blockIdx.x * 256;                            // d_gu column of the tile = 4 blocks of (32 gate | 32 up) = channels blockIdx.x * 128 ..
  const int64_t t0 = (int64_t)blockIdx.y * TN_CHUNK;
  const int rows = (int)max((int64_t)0, min((int64_t)TN_CHUNK, T - t0));
  const int nsteps = (rows + 63) / 64;
  // producer role: channel group o (8 channels), token rows pr + 16 i of a step
  const int po = tid & 15, pr = tid >> 4;
  const int ch = blockIdx.x * 128 + po * 8;
  const bool ch_live = ch < ff;
  const int chc = ch_live ? ch : 0;
  const int64_t gcol = (int64_t)(chc >> 5) * 64 + (chc & 31);     // gate column of the group in d_gu / gu; its up column is + 32
  const int cg = (po >> 2) * 8 + (po & 3), cu = cg + 4;           // the two 16-byte chunks of the tile row the group fills
  bf16x2_t a[8][RP];
#pragma unroll
  for (int j = 0; j < 8; ++j)
#pragma unroll
    for (int q = 0; q < RP / 4; ++q) {
      const bf16x8 v = *reinterpret_cast<const bf16x8*>(ATd + (int64_t)(chc + j) * 64 + q * 8);
#pragma unroll
      for (int r = 0; r < 4; ++r) a[j][q * 4 + r] = bf16x2_t{v[2 * r], v[2 * r + 1]};
    }
  const float keep_scale = 1.f / (1.f - p);
  const unsigned th = dropout_thresh(p);
  f32x4 acc[4][RG];
#pragma unroll
  for (int nf = 0; nf < 4; ++nf)
#pragma unroll
    for (int jf = 0; jf < RG; ++jf) acc[nf][jf] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int u = 0; u < (16 * RG * 32 + 255) / 256; ++u) {          // B^T piece of this column block, once per workgroup
    const int idx = tid + u * 256;
    if (idx < 16 * RG * 32) {
      const int r = idx >> 5, c = idx & 31;
      bf16x8 v = bf16x8{};
      if (n_base + c * 8 < N) v = *reinterpret_cast<const bf16x8*>(Bt + (int64_t)r * ldb + n_base + c * 8);
      *reinterpret_cast<bf16x8*>(at + r * 512 + ((c ^ (r & 7)) << 4)) = v;
    }
  }
  bf16x8 dv[4], gv[4], uv[4], dtv[4][RP / 4], tv[(64 * 2 * RG + 255) / 256];
  unsigned km[4];
  auto load_step = [&](int step) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = step * 64 + pr + 16 * i;
      dv[i] = bf16x8{}; gv[i] = bf16x8{}; uv[i] = bf16x8{}; km[i] = 0xffu;
#pragma unroll
      for (int q = 0; q < RP / 4; ++q) dtv[i][q] = bf16x8{};
      if (r < rows && ch_live) {
        const int64_t tok = t0 + r;
        dv[i] = *reinterpret_cast<const bf16x8*>(dact + tok * lddact + chc);
        gv[i] = *reinterpret_cast<const bf16x8*>(gu + tok * N + gcol);
        uv[i] = *reinterpret_cast<const bf16x8*>(gu + tok * N + gcol + 32);
#pragma unroll
        for (int q = 0; q < RP / 4; ++q) dtv[i][q] = *reinterpret_cast<const bf16x8*>(dtd + tok * lddtd + q * 8);
        if (p > 0.f) km[i] = keep_bits ? keep_bits[tok * ld_bits + (chc >> 3)] : dropout_keep8(seed, ((uint64_t)tok * (uint64_t)ff + (uint64_t)chc) >> 2, th);
      }
    }
#pragma unroll
    for (int u = 0; u < (64 * 2 * RG + 255) / 256; ++u) {
      const int idx = tid + u * 256;
      const int r = step * 64 + idx / (2 * RG), c = idx % (2 * RG);
      tv[u] = bf16x8{};
      if (idx < 64 * 2 * RG && r < rows) tv[u] = *reinterpret_cast<const bf16x8*>(G + (t0 + r) * ldg + c * 8);
    }
  };
  if (nsteps > 0) load_step(0);
  for (int step = 0; step < nsteps; ++step) {
    // ---- produce the step's tile rows from the operands requested a step ago (lora_up_add_kernel<RP, true>'s arithmetic), one token row
    //      at a time straight into LDS and d_gu (all four in registers first cost 256 registers per lane: one wave per SIMD)
    __syncthreads();                                       // the previous step's fragment reads are done (and, first time, `at` is written)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float ac[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) ac[j] = 0.f;
#pragma unroll
      for (int q = 0; q < RP / 4; ++q)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const bf16x2_t d2 = bf16x2_t{dtv[i][q][2 * r], dtv[i][q][2 * r + 1]};
#pragma unroll
          for (int j = 0; j < 8; ++j) ac[j] = __builtin_amdgcn_fdot2_f32_bf16(a[j][q * 4 + r], d2, ac[j], false);
        }
      bf16x8 dgs, dus;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float v = (float)(bf16_t)ac[j];
        if (p > 0.f) v = ((km[i] >> j) & 1u) ? (float)(bf16_t)(v * keep_scale) : 0.f;
        const float df = (float)(bf16_t)((float)dv[i][j] + v);
        const float gf = (float)gv[i][j], uf = (float)uv[i][j];
        const float sg = mp_sigmoid_fast(gf);
        dus[j] = (bf16_t)(df * gf * sg);
        dgs[j] = (bf16_t)(df * uf * sg * (1.f + gf * (1.f - sg)));
      }
      const int r = pr + 16 * i;
      const int64_t tok = t0 + step * 64 + r;
      const bool cell_live = step * 64 + r < rows && ch_live;
      *reinterpret_cast<bf16x8*>(xt + r * 512 + ((cg ^ (r & 7)) << 4)) = cell_live ? dgs : bf16x8{};
      *reinterpret_cast<bf16x8*>(xt + r * 512 + ((cu ^ (r & 7)) << 4)) = cell_live ? dus : bf16x8{};
      if (cell_live) {
        *reinterpret_cast<bf16x8*>(dgu + tok * N + gcol) = dgs;
        *reinterpret_cast<bf16x8*>(dgu + tok * N + gcol + 32) = dus;
      }
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int u = 0; u < (64 * 2 * RG + 255) / 256; ++u) {
      const int idx = tid + u * 256;
      const int r = idx / (2 * RG), c = idx % (2 * RG);
      if (idx < 64 * 2 * RG) *reinterpret_cast<bf16x8*>(gt + r * 128 + ((c ^ (r & 7)) << 4)) = tv[u];
    }
    __syncthreads();
    if (step + 1 < nsteps) load_step(step + 1);            // in flight under this step's MFMAs and the next step's production
#pragma unroll
    for (int kp = 0; kp < 2; ++kp) {
      bf16x8 gf[RG];
#pragma unroll
      for (int jf = 0; jf < RG; ++jf) gf[jf] = tn_tr_frag<128>(gt, kp, jf, fr, fq);
#pragma unroll
      for (int nf = 0; nf < 4; ++nf) {
        const bf16x8 xf = tn_tr_frag<512>(xt, kp, wave * 4 + nf, fr, fq);
#pragma unroll
        for (int jf = 0; jf < RG; ++jf) acc[nf][jf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf, gf[jf], acc[nf][jf], 0, 0, 0);
      }
    }
    f32x4 dacc[RG];
#pragma unroll
    for (int rg = 0; rg < RG; ++rg) dacc[rg] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
      const int c = kk * 4 + fq;
      const int xrow = wave * 16 + fr;
      const bf16x8 xf = *reinterpret_cast<const bf16x8*>(xt + xrow * 512 + ((c ^ (xrow & 7)) << 4));
#pragma unroll
      for (int rg = 0; rg < RG; ++rg) {
        const int arow = rg * 16 + fr;
        const bf16x8 af = *reinterpret_cast<const bf16x8*>(at + arow * 512 + ((c ^ (arow & 7)) << 4));
        dacc[rg] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, xf, dacc[rg], 0, 0, 0);
      }
    }
    const int64_t tok = t0 + step * 64 + wave * 16 + fr;
    if (tok < T) {
#pragma unroll
      for (int rg = 0; rg < RG; ++rg)
        *reinterpret_cast<f32x4*>(dt_partial + ((int64_t)blockIdx.x * T + tok) * (16 * RG) + rg * 16 + fq * 4) = dacc[rg];
    }
  }
  float* pout = partial + ((int64_t)blockIdx.y * N) * R;
#pragma unroll
  for (int nf = 0; nf < 4; ++nf)
#pragma unroll
    for (int jf = 0; jf < RG; ++jf)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = n_base + (wave * 4 + nf) * 16 + fq * 4 + r, j = jf * 16 + fr;
        if (n < N && j < R) pout[(int64_t)n * R + j] = acc[nf][jf][r];
      }
}

// The inverse of lora_pack for gradients: one adapter's slices of the fused group's padded gradients (dB [W, R] rows rows[o] columns k0..,
// dA^T [fin, R] columns k0..) ADDED into the parameter-shaped gradient tensors (lora_B.grad [fout, r], lora_A.grad [r, fin]) -- the engine's
// flat buffer; one launch instead of a gather, a transpose-copy and two adds per adapter (192 adapters' worth of 5-us kernels per step).
__global__ void lora_grad_unpack_kernel(const float* __restrict__ dB, const float* __restrict__ dAT, const int64_t* __restrict__ rows, int R, int k0, int r,
                                        int fin, int fout, float* __restrict__ gB, float* __restrict__ gA) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t nb = (int64_t)fout * r, na = (int64_t)r * fin;
  if (idx < nb) {
    const int o = (int)(idx / r), j = (int)(idx % r);
    gB[idx] += dB[rows[o] * R + k0 + j];
  } else if (idx < nb + na) {
    const int64_t e = idx - nb;
    const int i = (int)(e / fin), c = (int)(e % fin);
    gA[e] += dAT[(int64_t)c * R + k0 + i];
  }
}

// lora_grad_unpack_kernel reading the weight-gradient kernels' CHUNK PARTIALS directly (dBp [chunksB][W * R], dATp [chunksA][fin * R], each to be
// summed in ascending chunk order and scaled: exactly tn_skinny_reduce_kernel's arithmetic) — the two reduce launches per adapter group go away
// (128 launches of ~8 us per LoRA step); eight chunk loads in flight per thread.
__global__ void lora_grad_unpack_partials_kernel(const float* __restrict__ dBp, const float* __restrict__ dATp, int chunksB, int chunksA, float scaleB,
                                                 float scaleA, const int64_t* __restrict__ rows, int R, int k0, int r, int fin, int fout, int W,
                                                 float* __restrict__ gB, float* __restrict__ gA) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t nb = (int64_t)fout * r, na = (int64_t)r * fin;
  const float* src; int64_t stride; int chunks; float scale; float* dst;
  if (idx < nb) {
    const int o = (int)(idx / r), j = (int)(idx % r);
    src = dBp + rows[o] * R + k0 + j; stride = (int64_t)W * R; chunks = chunksB; scale = scaleB; dst = gB + idx;
  } else if (idx < nb + na) {
    const int64_t e = idx - nb;
    const int i = (int)(e / fin), c = (int)(e % fin);
    src = dATp + (int64_t)c * R + k0 + i; stride = (int64_t)fin * R; chunks = chunksA; scale = scaleA; dst = gA + e;
  } else return;
  float s = 0.f;
  for (int c0 = 0; c0 < chunks; c0 += 8) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = (c0 + u < chunks) ? src[(int64_t)(c0 + u) * stride] : 0.f;
#pragma unroll
    for (int u = 0; u < 8; ++u) if (c0 + u < chunks) s += v[u];
  }
  *dst += scale * s;
}

// One adapter's fp32 parameters written into the padded bf16 GEMM operands of its group (both orientations): A [64, fin] rows
// k0..k0+r-1, A^T [fin, 64] columns k0.., B [W, 64] rows rows[o] columns k0.., B^T [64, W] rows k0.. columns rows[o]
__global__ void lora_pack_kernel(const float* __restrict__ a, const float* __restrict__ b, const int64_t* __restrict__ rows, bf16_t* __restrict__ A,
                                 bf16_t* __restrict__ AT, bf16_t* __restrict__ B, bf16_t* __restrict__ BT, int r, int fin, int fout, int k0,
                                 int W, float bscale, bf16_t* __restrict__ Bx, int64_t ldbx, float xscale) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t na = (int64_t)r * fin, nb = (int64_t)fout * r;
  if (idx < na) {
    const int i = (int)(idx / fin), c = (int)(idx % fin);
    const bf16_t v = (bf16_t)a[idx];
    A[(int64_t)(k0 + i) * fin + c] = v;
    AT[(int64_t)c * 64 + k0 + i] = v;
  } else if (idx < na + nb) {
    const int64_t e = idx - na;
    const int o = (int)(e / r), j = (int)(e % r);
    const int64_t ro = rows[o];
    const bf16_t v = (bf16_t)(b[e] * bscale);
    B[ro * 64 + k0 + j] = v;
    BT[(int64_t)(k0 + j) * W + ro] = v;
    if (Bx) Bx[ro * ldbx + k0 + j] = (bf16_t)(b[e] * xscale);      // the K-extension columns of [W | scaling B] (mp_lora_down_bf16)
  }
}

// Every adapter of the decoder in ONE launch (blockIdx.y = adapter): the training step re-packs all of them once per step (96 launches of the
// kernel above at r = 8 on gate / up / down of 32 layers).  The descriptor table lives on the device and is rebuilt only when a pointer moves.
struct LoraPackDesc {
  const float* a; const float* b; const int64_t* rows; bf16_t* A; bf16_t* AT; bf16_t* B; bf16_t* BT; bf16_t* Bx;
  int64_t ldbx;
  int r, fin, fout, k0, W;
  float bscale, xscale;
  int pad;
};
static_assert(sizeof(LoraPackDesc) == 104, "LoraPackDesc is packed by medplib_amd/model/llama_lora.py as 104 bytes");
__global__ void lora_pack_batched_kernel(const LoraPackDesc* __restrict__ descs) {
  const LoraPackDesc g = descs[blockIdx.y];
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t na = (int64_t)g.r * g.fin, nb = (int64_t)g.fout * g.r;
  if (idx < na) {
    const int i = (int)(idx / g.fin), c = (int)(idx % g.fin);
    const bf16_t v = (bf16_t)g.a[idx];
    g.A[(int64_t)(g.k0 + i) * g.fin + c] = v;
    g.AT[(int64_t)c * 64 + g.k0 + i] = v;
  } else if (idx < na + nb) {
    const int64_t e = idx - na;
    const int o = (int)(e / g.r), j = (int)(e % g.r);
    const int64_t ro = g.rows[o];
    const bf16_t v = (bf16_t)(g.b[e] * g.bscale);
    g.B[ro * 64 + g.k0 + j] = v;
    g.BT[(int64_t)(g.k0 + j) * g.W + ro] = v;
    if (g.Bx) g.Bx[ro * g.ldbx + g.k0 + j] = (bf16_t)(g.b[e] * g.xscale);
  }
}

// ---- MoE layer backward (top-1; DeepSpeed MOELayer + top1gating autograd) ----
// combine backward: out[t] = residual[t] + w[t] * y[e_t, slot_t]  =>  d_y[e_t, slot_t] = w[t] * d_out[t],  d_w[t] = <d_out[t], y[e_t, slot_t]>
// (dropped tokens: nothing).  One wave per token; d_y is pre-zeroed.
__global__ __launch_bounds__(256) void moe_combine_bwd_kernel(const bf16_t* __restrict__ dout, const bf16_t* __restrict__ y,
                                                              const int* __restrict__ expert, const int* __restrict__ slot,
                                                              const float* __restrict__ weight, bf16_t* __restrict__ dy, float* __restrict__ dw,
                                                              int64_t T, int d, int capacity, int top_k) {
  const int lane = threadIdx.x & 63;
  const int64_t en = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);        // entry = choice * T + token (top_k choices per token)
  if (en >= T * top_k) return;
  const int64_t t = en % T;
  const int sl = slot[en];
  if (sl < 0) { if (lane == 0) dw[en] = 0.f; return; }
  const int64_t row = ((int64_t)expert[en] * capacity + sl) * d;
  const float w = weight[en];
  float acc = 0.f;
  for (int i = lane * 8; i < d; i += 512) {
    const bf16x8 g = *reinterpret_cast<const bf16x8*>(dout + t * d + i);
    const bf16x8 yv = *reinterpret_cast<const bf16x8*>(y + row + i);
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) { acc = fmaf((float)g[j], (float)yv[j], acc); o[j] = (bf16_t)(w * (float)g[j]); }
    *reinterpret_cast<bf16x8*>(dy + row + i) = o;
  }
  acc = wave_sum(acc);
  if (lane == 0) dw[en] = acc;
}

// gate backward: the combine weight is the softmax probability of the chosen expert (kept tokens) and
// l_aux = E * sum_e mean_t(p[t, e]) * (counts_e / T)  =>  g[t, j] = dw[t] [j == e_t, kept] + c_aux * E * counts_j / T^2,
// d_logits[t, j] = p_j (g_j - sum_k p_k g_k)
__global__ void moe_gate_bwd_kernel(const float* __restrict__ gates, const int* __restrict__ expert, const int* __restrict__ slot,
                                    const float* __restrict__ dw, const long long* __restrict__ counts, const float* __restrict__ c_aux,
                                    float aux_coef, float* __restrict__ dlogits, int64_t T, int E, int top_k) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= T) return;
  const float ca = (c_aux ? c_aux[0] : 0.f) * aux_coef * (float)E / ((float)T * (float)T);
  const int e1 = expert[t];
  const bool kept1 = slot[t] >= 0;
  int e2 = -1;
  float dg1 = kept1 ? dw[t] : 0.f, dg2 = 0.f;
  if (top_k == 2) {
    // top2gating: w1 = g1 / D, w2 = g2 / D with g = the (kept) choices' probabilities and D = max(g1 + g2, eps):
    // d g1 = g2 (dw1 - dw2) / D^2, d g2 = g1 (dw2 - dw1) / D^2 (both zero when one choice was dropped: the other weight is 1)
    e2 = expert[T + t];
    const bool kept2 = slot[T + t] >= 0;
    const float g1 = kept1 ? gates[t * E + e1] : 0.f, g2 = kept2 ? gates[t * E + e2] : 0.f;
    const float D = g1 + g2;
    const float dw1 = kept1 ? dw[t] : 0.f, dw2 = kept2 ? dw[T + t] : 0.f;
    if (D > 1.1920929e-07f) { dg1 = g2 * (dw1 - dw2) / (D * D); dg2 = g1 * (dw2 - dw1) / (D * D); }
    else { dg1 = 0.f; dg2 = 0.f; }
    if (!kept1) dg1 = 0.f;
    if (!kept2) dg2 = 0.f;
  }
  float dot = 0.f;
  for (int j = 0; j < E; ++j) {
    const float g = (j == e1 ? dg1 : 0.f) + (j == e2 ? dg2 : 0.f) + ca * (float)counts[j];
    dot += gates[t * E + j] * g;
  }
  for (int j = 0; j < E; ++j) {
    const float g = (j == e1 ? dg1 : 0.f) + (j == e2 ? dg2 : 0.f) + ca * (float)counts[j];
    dlogits[t * E + j] = gates[t * E + j] * (g - dot);
  }
}

// d_x[t, :] += sum_e d_logits[t, e] * wg[e, :]   (the gate's input gradient; wg fp32 [E, d])
__global__ void moe_gate_dgrad_kernel(const float* __restrict__ dlogits, const float* __restrict__ wg, bf16_t* __restrict__ dx, int64_t T, int d,
                                      int E) {
  const int per_row = d / 8;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= T * per_row) return;
  const int64_t t = idx / per_row;
  const int c = (int)(idx % per_row) * 8;
  bf16x8 v = *reinterpret_cast<const bf16x8*>(dx + t * d + c);
  float acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = (float)v[j];
  for (int e = 0; e < E; ++e) {
    const float g = dlogits[t * E + e];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = fmaf(g, wg[(int64_t)e * d + c + j], acc[j]);
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = (bf16_t)acc[j];
  *reinterpret_cast<bf16x8*>(dx + t * d + c) = v;
}

// embedding-table gradient: out[ids[u], :] = sum over the rows of segment u (rows_sorted[seg[u] .. seg[u+1])) of g[row, :], added in
// list order (the host sorts the token rows by id): no atomics, duplicates of a token id across the batch included
__global__ void embed_grad_kernel(const bf16_t* __restrict__ g, const int64_t* __restrict__ rows_sorted, const int64_t* __restrict__ seg,
                                  const int64_t* __restrict__ ids, float* __restrict__ out, int d) {
  const int64_t u = blockIdx.x;
  const int64_t b = seg[u], e = seg[u + 1];
  float* o = out + ids[u] * d;
  for (int c = threadIdx.x * 4; c < d; c += blockDim.x * 4) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int64_t k = b; k < e; ++k) {
      const bf16x4 v = *reinterpret_cast<const bf16x4*>(g + rows_sorted[k] * d + c);
      acc += f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
    }
    *reinterpret_cast<f32x4*>(o + c) = acc;
  }
}

// RMSNorm weight gradient: dw[c] = sum_t dy[t, c] * bf16(x[t, c] * rs[t]) (the normalised value as the forward rounded it).
// A block owns 256 columns x a chunk of 256 rows -> partial[chunk][c]; the chunks are added in ascending order by tn_skinny_reduce.
__global__ __launch_bounds__(256) void rmsnorm_wgrad_partial_kernel(const bf16_t* __restrict__ x, int64_t ldx, const bf16_t* __restrict__ dy,
                                                                    int64_t ldy, const float* __restrict__ rs, float* __restrict__ partial,
                                                                    int64_t T, int dim) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  const int64_t t0 = (int64_t)blockIdx.y * 256, t1 = min(T, t0 + 256);
  if (c >= dim) return;
  float acc = 0.f;
  for (int64_t t = t0; t < t1; ++t) acc = fmaf((float)dy[t * ldy + c], (float)(bf16_t)((float)x[t * ldx + c] * rs[t]), acc);
  partial[(int64_t)blockIdx.y * dim + c] = acc;
}

// GELU on a bf16 tensor with the GEMM epilogue's own function, and its derivative (mm_projector training, train_stage2.sh):
// d/dx gelu(x) = Phi(x) + x phi(x)
__global__ void gelu_fwd_bf16_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y, int64_t n) {
  const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 8;
  if (i >= n) return;
  const bf16x8 v = *reinterpret_cast<const bf16x8*>(x + i);
  bf16x8 o;
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] = (bf16_t)gelu_erf_fast((float)v[j]);
  *reinterpret_cast<bf16x8*>(y + i) = o;
}
__global__ void gelu_bwd_bf16_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy, bf16_t* __restrict__ dx, int64_t n) {
  const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 8;
  if (i >= n) return;
  const bf16x8 v = *reinterpret_cast<const bf16x8*>(x + i);
  const bf16x8 g = *reinterpret_cast<const bf16x8*>(dy + i);
  bf16x8 o;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float xf = (float)v[j];
    const float cdf = 0.5f * (1.f + erff(xf * 0.70710678118654752f));
    const float pdf = 0.3989422804014327f * __expf(-0.5f * xf * xf);
    o[j] = (bf16_t)((float)g[j] * (cdf + xf * pdf));
  }
  *reinterpret_cast<bf16x8*>(dx + i) = o;
}

// extract_region_feature backward (medplib_arch.py:580-613; region_fea_adapter in --sft_modules): the forward is, per mask m, the
// mean over its sampled points of the bilinear (align_corners=True, zero padding) read-out of feature map map_index[m].
// Pass 1: wt[m, pixel] = (1 / n_m) * sum over m's points of the bilinear weight that point puts on the pixel (a thread per (m, pixel)
// walks m's points: no atomics).  Pass 2: d_fmap[j, pixel, :] = sum over the masks of map j of wt[m, pixel] * d_out[m, :].
__global__ void region_weights_kernel(const float* __restrict__ xy, const int64_t* __restrict__ offsets, float* __restrict__ wt, int n_masks,
                                      int h, int w) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)n_masks * h * w) return;
  const int pix = (int)(idx % (h * w)), m = (int)(idx / (h * w));
  const int py = pix / w, px = pix % w;
  const int64_t p0 = offsets[m], p1 = offsets[m + 1];
  float acc = 0.f;
  for (int64_t p = p0; p < p1; ++p) {
    const float gx = 2.f * xy[2 * p] - 1.f, gy = 2.f * xy[2 * p + 1] - 1.f;
    const float ix = ((gx + 1.f) * 0.5f) * (float)(w - 1), iy = ((gy + 1.f) * 0.5f) * (float)(h - 1);
    const float fx = floorf(ix), fy = floorf(iy);
    const int x0 = (int)fx, y0 = (int)fy;
    const float wx1 = ix - fx, wy1 = iy - fy;
    const float wxs = px == x0 ? 1.f - wx1 : (px == x0 + 1 ? wx1 : 0.f);
    const float wys = py == y0 ? 1.f - wy1 : (py == y0 + 1 ? wy1 : 0.f);
    acc += wxs * wys;
  }
  wt[idx] = p1 > p0 ? acc / (float)(p1 - p0) : 0.f;
}
__global__ void region_fmap_grad_kernel(const float* __restrict__ wt, const bf16_t* __restrict__ dout, const int* __restrict__ map_index,
                                        bf16_t* __restrict__ dfmap, int n_maps, int n_masks, int hw, int C) {
  const int per_row = C / 8;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)n_maps * hw * per_row) return;
  const int c = (int)(idx % per_row) * 8;
  const int pix = (int)((idx / per_row) % hw), j = (int)(idx / ((int64_t)per_row * hw));
  float acc[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) acc[k] = 0.f;
  for (int m = 0; m < n_masks; ++m) {
    if (map_index[m] != j) continue;
    const float wv = wt[(int64_t)m * hw + pix];
    if (wv == 0.f) continue;
    const bf16x8 g = *reinterpret_cast<const bf16x8*>(dout + (int64_t)m * C + c);
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] = fmaf(wv, (float)g[k], acc[k]);
  }
  bf16x8 o;
#pragma unroll
  for (int k = 0; k < 8; ++k) o[k] = (bf16_t)acc[k];
  *reinterpret_cast<bf16x8*>(dfmap + ((int64_t)j * hw + pix) * C + c) = o;
}

// ---- MaskTokenEncoder backward pieces (`mask_encoder` in --sft_modules, scripts/train_medplib_icl.sh:12; medplib_arch.py:80-108) ----
// layer 1 without its GELU (the training forward keeps the pre-activation): Conv2d(1, CO, k3, s2, p1) on a single-channel image
template <typename TIN>
__global__ void conv3x3s2_c1_pre_kernel(const TIN* __restrict__ img, const float* __restrict__ w, const float* __restrict__ bias,
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
  for (int ky = 0; ky < 3; ++ky)
    for (int kx = 0; kx < 3; ++kx) {
      const int iy = oy * 2 - 1 + ky, ix = ox * 2 - 1 + kx;
      if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
      const float v = (float)(bf16_t)(float)img[((int64_t)b * H + iy) * W + ix];
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] = fmaf(v, w[(co + j) * 9 + ky * 3 + kx], acc[j]);
    }
  bf16x8 o;
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] = (bf16_t)acc[j];
  *reinterpret_cast<bf16x8*>(out + pix * CO + co) = o;
}
// its weight / bias gradient: block (q, co), q = tap 0..8 or 9 = bias; fixed-order block reduction over all output pixels
template <typename TIN>
__global__ __launch_bounds__(256) void conv3x3s2_c1_wgrad_kernel(const TIN* __restrict__ img, const bf16_t* __restrict__ dpre,
                                                                 float* __restrict__ dw, float* __restrict__ db, int n, int H, int W, int OH,
                                                                 int OW, int CO) {
  __shared__ float red[16];
  const int q = blockIdx.x, co = blockIdx.y;
  const int ky = q / 3, kx = q % 3;
  float acc = 0.f;
  const int64_t total = (int64_t)n * OH * OW;
  for (int64_t pix = threadIdx.x; pix < total; pix += 256) {
    const float g = (float)dpre[pix * CO + co];
    if (q == 9) { acc += g; continue; }
    const int ox = (int)(pix % OW), oy = (int)((pix / OW) % OH), b = (int)(pix / ((int64_t)OW * OH));
    const int iy = oy * 2 - 1 + ky, ix = ox * 2 - 1 + kx;
    if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
    acc = fmaf(g, (float)(bf16_t)(float)img[((int64_t)b * H + iy) * W + ix], acc);
  }
  acc = block_sum(acc, red);
  if (threadIdx.x == 0) { if (q == 9) db[co] = acc; else dw[co * 9 + q] = acc; }
}
// AdaptiveAvgPool1d over tokens, backward: d_x[b, t, :] = sum over the outputs i whose window [t0_i, t1_i) holds t of d_out[b, i, :] / size_i
__global__ void adaptive_avgpool_tokens_bwd_kernel(const bf16_t* __restrict__ dout, bf16_t* __restrict__ dx, int n, int Lin, int Lout, int C) {
  const int per_row = C / 8;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)n * Lin * per_row) return;
  const int c = (int)(idx % per_row) * 8;
  const int t = (int)((idx / per_row) % Lin);
  const int b = (int)(idx / ((int64_t)per_row * Lin));
  float acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = 0.f;
  const int i_lo = max(0, (int)(((int64_t)t * Lout) / Lin) - 1), i_hi = min(Lout - 1, (int)((((int64_t)t + 1) * Lout + Lin - 1) / Lin));
  for (int i = i_lo; i <= i_hi; ++i) {
    const int t0 = (int)(((int64_t)i * Lin) / Lout), t1 = (int)((((int64_t)(i + 1)) * Lin + Lout - 1) / Lout);
    if (t < t0 || t >= t1) continue;
    const float inv = 1.f / (float)(t1 - t0);
    const bf16x8 g = *reinterpret_cast<const bf16x8*>(dout + ((int64_t)b * Lout + i) * C + c);
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = fmaf((float)g[j], inv, acc[j]);
  }
  bf16x8 o;
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] = (bf16_t)acc[j];
  *reinterpret_cast<bf16x8*>(dx + ((int64_t)b * Lin + t) * C + c) = o;
}
// col2im of a k3 / s2 / p1 convolution in gather form: d_x[b, y, x, :] = sum over the taps (ky, kx) that read this pixel of
// d_cols[(b, oy, ox), ky*3+kx, :],  oy = (y + 1 - ky) / 2 (when even and in range), ox likewise
__global__ void col2im_k3s2p1_kernel(const bf16_t* __restrict__ dcols, bf16_t* __restrict__ dx, int n, int H, int W, int C, int OH, int OW) {
  const int per_pix = C / 8;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)n * H * W * per_pix) return;
  const int c = (int)(idx % per_pix) * 8;
  const int64_t pix = idx / per_pix;
  const int x = (int)(pix % W), y = (int)((pix / W) % H), b = (int)(pix / ((int64_t)W * H));
  float acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = 0.f;
  for (int ky = 0; ky < 3; ++ky) {
    const int ty = y + 1 - ky;
    if (ty < 0 || (ty & 1) || ty / 2 >= OH) continue;
    for (int kx = 0; kx < 3; ++kx) {
      const int tx = x + 1 - kx;
      if (tx < 0 || (tx & 1) || tx / 2 >= OW) continue;
      const int64_t orow = ((int64_t)b * OH + ty / 2) * OW + tx / 2;
      const bf16x8 g = *reinterpret_cast<const bf16x8*>(dcols + orow * (9 * (int64_t)C) + (int64_t)(ky * 3 + kx) * C + c);
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] += (float)g[j];
    }
  }
  bf16x8 o;
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] = (bf16_t)acc[j];
  *reinterpret_cast<bf16x8*>(dx + pix * C + c) = o;
}

}  // namespace

#define GRID1D(n) dim3((unsigned)mp_cdiv((n), 256)), dim3(256), 0, stream

extern "C" int mp_rmsnorm_bwd_bf16(const void* x, int64_t ldx, const float* w, const void* dy, int64_t ldy, const void* add, int64_t lda,
                                   void* dx, int64_t ldo, int64_t rows, int dim, float eps, float* rs_out, hipStream_t stream) {
  MP_REQUIRE(dim % 8 == 0 && dim <= RB_THREADS * 8 * RB_MAXC && ldx % 8 == 0 && ldy % 8 == 0 && ldo % 8 == 0 && (!add || lda % 8 == 0), MP_ERR_SHAPE,
             "mp_rmsnorm_bwd_bf16: dim=%d unsupported", dim);
  if (rows == 0) return MP_OK;
  hipLaunchKernelGGL(rmsnorm_bwd_kernel, dim3((unsigned)rows), dim3(RB_THREADS), 0, stream, (const bf16_t*)x, w, (const bf16_t*)dy,
                     (const bf16_t*)add, (bf16_t*)dx, dim, eps, ldx, ldy, lda, ldo, rs_out);
  return mp_check_launch("mp_rmsnorm_bwd_bf16");
}

extern "C" int mp_rmsnorm_bwd_up_bf16(const void* x, int64_t ldx, const float* w, const void* dy, int64_t ldy, const void* add, int64_t lda, void* dx,
                                     int64_t ldo, int64_t rows, int dim, float eps, const void* dt, int64_t lddt, const void* AT, int R, float p,
                                     uint64_t seed, const uint8_t* keep_bits, int64_t ld_bits, hipStream_t stream) {
  MP_REQUIRE(dim == 4096 && ldx % 8 == 0 && ldy % 8 == 0 && ldo % 8 == 0 && (!add || lda % 8 == 0) && lddt % 8 == 0 && rows < (1ll << 31), MP_ERR_SHAPE,
             "mp_rmsnorm_bwd_up_bf16: dim must be 4096 (got %d), strides multiples of 8", dim);
  MP_REQUIRE((R == 8 || R == 16) && p >= 0.f && p < 1.f && dt && AT && (!keep_bits || (p > 0.f && ld_bits * 8 >= dim)), MP_ERR_ARG,
             "mp_rmsnorm_bwd_up_bf16: R in {8, 16}, 0 <= p < 1, keep_bits only with p > 0");
  if (rows == 0) return MP_OK;
  MP_REQUIRE(R <= 16, MP_ERR_SHAPE, "mp_rmsnorm_bwd_up_bf16: R = %d: the rank vectors of R = 32 do not fit the LDS (use mp_lora_up_add_bf16 + mp_rmsnorm_bwd_bf16)", R);
  const int lds = 2 * 8 * (R / 8) * 256 * 16 + 4096 * 4;    // [2][8][R / 8][256] 16-byte pieces of A^T + the norm weight
  {
    static bool attr_set[64][2] = {};
    int devi = 0;
    (void)hipGetDevice(&devi);
    const int k = R == 8 ? 0 : 1;
    if (devi >= 0 && devi < 64 && !attr_set[devi][k]) {
      if (k == 0) (void)hipFuncSetAttribute((const void*)rmsnorm_bwd_up_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
      else (void)hipFuncSetAttribute((const void*)rmsnorm_bwd_up_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
      attr_set[devi][k] = true;
    }
  }
  const dim3 grid((unsigned)std::min<int64_t>((rows + 3) / 4, mp_device_cus())), blk(1024);
#define MP_GO(RP) hipLaunchKernelGGL((rmsnorm_bwd_up_kernel<RP>), grid, blk, lds, stream, (const bf16_t*)x, w, (const bf16_t*)dy, (const bf16_t*)add, (bf16_t*)dx, eps, ldx, ldy, lda, ldo, (int)rows, (const bf16_t*)dt, lddt, (const bf16_t*)AT, p, seed, keep_bits, ld_bits)
  if (R == 8) MP_GO(4); else MP_GO(8);
#undef MP_GO
  return mp_check_launch("mp_rmsnorm_bwd_up_bf16");
}

extern "C" int mp_swiglu_pair_fwd_bf16(const void* gu, void* act, int64_t ldact, int64_t tokens, int ff, const int* counts, int cap, hipStream_t stream) {
  MP_REQUIRE(ff % 32 == 0 && ldact >= ff && ldact % 8 == 0 && (!counts || cap > 0), MP_ERR_SHAPE, "mp_swiglu_pair_fwd_bf16: ff %% 32 != 0 or bad ldact / cap");
  const int64_t n = tokens * (ff / 8);
  if (n == 0) return MP_OK;
  hipLaunchKernelGGL(swiglu_pair_fwd_kernel, GRID1D(n), (const bf16_t*)gu, (bf16_t*)act, tokens, ff, ldact, counts, cap);
  return mp_check_launch("mp_swiglu_pair_fwd_bf16");
}

extern "C" int mp_swiglu_pair_bwd_bf16(const void* gu, const void* dact, void* dgu, int64_t tokens, int ff, const int* counts, int cap, hipStream_t stream) {
  MP_REQUIRE(ff % 32 == 0, MP_ERR_SHAPE, "mp_swiglu_pair_bwd_bf16: ff %% 32 != 0");
  const int64_t n = tokens * (ff / 8);
  if (n == 0) return MP_OK;
  hipLaunchKernelGGL(swiglu_pair_bwd_kernel, GRID1D(n), (const bf16_t*)gu, (const bf16_t*)dact, (bf16_t*)dgu, tokens, ff, counts, cap);
  return mp_check_launch("mp_swiglu_pair_bwd_bf16");
}

extern "C" int mp_tn_skinny_f32(const void* X, int64_t ldx, const void* G, int64_t ldg, float* out, float* partial, int64_t partial_floats,
                                int64_t tokens, int N, int R, float scale, float p, uint64_t seed, const int* rows_dev, const uint8_t* keep_bits,
                                int64_t ld_bits, hipStream_t stream) {
  MP_REQUIRE(!keep_bits || (p > 0.f && ld_bits * 8 >= N), MP_ERR_ARG, "mp_tn_skinny_f32: keep_bits come with p > 0 and a row stride of >= N / 8 bytes");
  MP_REQUIRE(N > 0 && tokens > 0 && (R == 8 || R == 16 || R == 32) && ldg % 8 == 0 && ldx % 4 == 0 && p >= 0.f && p < 1.f, MP_ERR_SHAPE,
             "mp_tn_skinny_f32: R must be 8, 16 or 32; ldx %% 4, ldg %% 8; 0 <= p < 1");
  const int chunks = (int)mp_cdiv(tokens, TN_CHUNK);
  MP_REQUIRE(partial && partial_floats >= (int64_t)chunks * N * R, MP_ERR_WORKSPACE, "mp_tn_skinny_f32: partial needs %lld floats",
             (long long)((int64_t)chunks * N * R));
  const dim3 grid((unsigned)mp_cdiv(N, 256), (unsigned)chunks);
  static int use_mfma = -1;
  if (use_mfma < 0) { const char* e = getenv("MP_TN_SKINNY_MFMA"); use_mfma = (e && atoi(e) == 0) ? 0 : 1; }     // 0: the scalar kernel (A/B)
  const bool mfma = ldx % 8 == 0 && N % 8 == 0 && (use_mfma || p > 0.f);
  MP_REQUIRE(mfma || p == 0.f, MP_ERR_ARG, "mp_tn_skinny_f32: inline dropout needs ldx %% 8 == 0 and N %% 8 == 0");
  if (mfma) {
    // G must be readable for 16 columns per rank group (the padded [tokens, 64] adapter tensors are)
    if (R <= 16) hipLaunchKernelGGL(tn_skinny_mfma_kernel<1>, grid, dim3(256), 0, stream, (const bf16_t*)X, ldx, (const bf16_t*)G, ldg, partial, tokens, N, R, p, seed, rows_dev, keep_bits, ld_bits);
    else hipLaunchKernelGGL(tn_skinny_mfma_kernel<2>, grid, dim3(256), 0, stream, (const bf16_t*)X, ldx, (const bf16_t*)G, ldg, partial, tokens, N, R, p, seed, rows_dev, keep_bits, ld_bits);
  } else if (R == 8) hipLaunchKernelGGL(tn_skinny_partial_kernel<8>, grid, dim3(256), 0, stream, (const bf16_t*)X, ldx, (const bf16_t*)G, ldg, partial, tokens, N, rows_dev);
  else if (R == 16) hipLaunchKernelGGL(tn_skinny_partial_kernel<16>, grid, dim3(256), 0, stream, (const bf16_t*)X, ldx, (const bf16_t*)G, ldg, partial, tokens, N, rows_dev);
  else hipLaunchKernelGGL(tn_skinny_partial_kernel<32>, grid, dim3(256), 0, stream, (const bf16_t*)X, ldx, (const bf16_t*)G, ldg, partial, tokens, N, rows_dev);
  const int64_t NR = (int64_t)N * R;
  if (out)                                                  // out == NULL: the caller consumes the chunk partials itself (mp_lora_grad_unpack_partials_f32)
    hipLaunchKernelGGL(tn_skinny_reduce_kernel, dim3((unsigned)mp_cdiv(NR, 256)), dim3(256), 0, stream, partial, out, NR, chunks, scale);
  return mp_check_launch("mp_tn_skinny_f32");
}

extern "C" int mp_tn_skinny_down_f32(const void* X, int64_t ldx, const void* G, int64_t ldg, float* out, float* partial, int64_t partial_floats,
                                     const void* Bt, int64_t ldb, void* dt, int64_t lddt, float* dt_partial, int64_t dt_partial_floats, int64_t tokens,
                                     int N, int R, float scale, float alpha, hipStream_t stream) {
  MP_REQUIRE(N > 0 && tokens > 0 && tokens < (1ll << 31) && (R == 8 || R == 16 || R == 32) && ldg % 8 == 0 && ldx % 8 == 0 && N % 8 == 0 && ldb % 8 == 0 && lddt % 4 == 0,
             MP_ERR_SHAPE, "mp_tn_skinny_down_f32: R must be 8, 16 or 32; ldx, ldg, ldb, N %% 8; lddt %% 4");
  const int chunks = (int)mp_cdiv(tokens, TN_CHUNK), blocks = (int)mp_cdiv(N, 256), rg = (R + 15) / 16;
  MP_REQUIRE(partial && partial_floats >= (int64_t)chunks * N * R, MP_ERR_WORKSPACE, "mp_tn_skinny_down_f32: partial needs %lld floats",
             (long long)((int64_t)chunks * N * R));
  MP_REQUIRE(dt_partial && dt_partial_floats >= (int64_t)blocks * tokens * 16 * rg, MP_ERR_WORKSPACE, "mp_tn_skinny_down_f32: dt_partial needs %lld floats",
             (long long)((int64_t)blocks * tokens * 16 * rg));
  const dim3 grid((unsigned)blocks, (unsigned)chunks);
  if (rg == 1) hipLaunchKernelGGL(tn_skinny_down_mfma_kernel<1>, grid, dim3(256), 0, stream, (const bf16_t*)X, ldx, (const bf16_t*)G, ldg, partial, (const bf16_t*)Bt, ldb, dt_partial, tokens, N, R);
  else hipLaunchKernelGGL(tn_skinny_down_mfma_kernel<2>, grid, dim3(256), 0, stream, (const bf16_t*)X, ldx, (const bf16_t*)G, ldg, partial, (const bf16_t*)Bt, ldb, dt_partial, tokens, N, R);
  hipLaunchKernelGGL(lora_down_finish_kernel, GRID1D(tokens * 16), dt_partial, (bf16_t*)dt, lddt, (int)tokens, 16 * rg, blocks, alpha, (const int*)nullptr);
  const int64_t NR = (int64_t)N * R;
  if (out)
    hipLaunchKernelGGL(tn_skinny_reduce_kernel, dim3((unsigned)mp_cdiv(NR, 256)), dim3(256), 0, stream, partial, out, NR, chunks, scale);
  return mp_check_launch("mp_tn_skinny_down_f32");
}

extern "C" int mp_swiglu_bwd_skinny_f32(const void* dact, int64_t lddact, const void* gu, void* dgu, const void* dt_down, int64_t lddtd, const void* AT_down,
                                        int R_down, float p, uint64_t seed, const uint8_t* keep_bits, int64_t ld_bits, const void* t_gu, int64_t ldg,
                                        float* out, float* partial, int64_t partial_floats, const void* Bt_gu, int64_t ldb, void* dt_gu, int64_t lddt,
                                        float* dt_partial, int64_t dt_partial_floats, int64_t tokens, int ff, int R_gu, float scale, float alpha,
                                        hipStream_t stream) {
  MP_REQUIRE(tokens > 0 && tokens < (1ll << 31) && ff > 0 && ff % 32 == 0 && (R_down == 8 || R_down == 16) && (R_gu == 8 || R_gu == 16 || R_gu == 32), MP_ERR_SHAPE,
             "mp_swiglu_bwd_skinny_f32: ff %% 32 == 0, R_down in {8, 16}, R_gu in {8, 16, 32} (got ff %d, R %d / %d)", ff, R_down, R_gu);
  MP_REQUIRE(lddact % 8 == 0 && lddtd % 8 == 0 && ldg % 8 == 0 && ldb % 8 == 0 && lddt % 4 == 0 && p >= 0.f && p < 1.f && dact && gu && dgu && dt_down && AT_down
                 && t_gu && Bt_gu && dt_gu && (!keep_bits || (p > 0.f && ld_bits * 8 >= ff)), MP_ERR_ARG, "mp_swiglu_bwd_skinny_f32: bad strides / p / null operand");
  const int N = 2 * ff;
  const int chunks = (int)mp_cdiv(tokens, TN_CHUNK), blocks = (int)mp_cdiv(N, 256), rg = (R_gu + 15) / 16;
  MP_REQUIRE(partial && partial_floats >= (int64_t)chunks * N * R_gu, MP_ERR_WORKSPACE, "mp_swiglu_bwd_skinny_f32: partial needs %lld floats",
             (long long)((int64_t)chunks * N * R_gu));
  MP_REQUIRE(dt_partial && dt_partial_floats >= (int64_t)blocks * tokens * 16 * rg, MP_ERR_WORKSPACE, "mp_swiglu_bwd_skinny_f32: dt_partial needs %lld floats",
             (long long)((int64_t)blocks * tokens * 16 * rg));
  const dim3 grid((unsigned)blocks, (unsigned)chunks);
#define MP_GO(RP, RGG) hipLaunchKernelGGL((swiglu_bwd_skinny_kernel<RP, RGG>), grid, dim3(256), 0, stream, (const bf16_t*)dact, lddact, (const bf16_t*)gu, (bf16_t*)dgu, \
    (const bf16_t*)dt_down, lddtd, (const bf16_t*)AT_down, p, seed, keep_bits, ld_bits, (const bf16_t*)t_gu, ldg, partial, (const bf16_t*)Bt_gu, ldb, dt_partial, tokens, ff, R_gu)
  if (R_down == 8) { if (rg == 1) MP_GO(4, 1); else MP_GO(4, 2); }
  else { if (rg == 1) MP_GO(8, 1); else MP_GO(8, 2); }
#undef MP_GO
  hipLaunchKernelGGL(lora_down_finish_kernel, GRID1D(tokens * 16), dt_partial, (bf16_t*)dt_gu, lddt, (int)tokens, 16 * rg, blocks, alpha, (const int*)nullptr);
  const int64_t NR = (int64_t)N * R_gu;
  if (out)
    hipLaunchKernelGGL(tn_skinny_reduce_kernel, dim3((unsigned)mp_cdiv(NR, 256)), dim3(256), 0, stream, partial, out, NR, chunks, scale);
  return mp_check_launch("mp_swiglu_bwd_skinny_f32");
}

extern "C" int mp_ce_rows_bwd(const float* logits, int64_t ldl, const int64_t* labels, const float* gscale, float gconst, void* dlogits,
                              int64_t ldo, int64_t rows, int V, hipStream_t stream) {
  MP_REQUIRE(V > 0 && ldo >= V, MP_ERR_SHAPE, "mp_ce_rows_bwd: bad shape");
  if (rows == 0) return MP_OK;
  hipLaunchKernelGGL(ce_rows_bwd_kernel, dim3((unsigned)rows), dim3(256), 0, stream, logits, labels, gscale, gconst, (bf16_t*)dlogits, V, ldl, ldo);
  return mp_check_launch("mp_ce_rows_bwd");
}

extern "C" int mp_scatter_rows_f32_bf16(const float* g, const int64_t* rows, void* out, int64_t n, int dim, hipStream_t stream) {
  MP_REQUIRE(dim % 4 == 0, MP_ERR_SHAPE, "mp_scatter_rows_f32_bf16: dim %% 4 != 0");
  const int64_t m = n * (dim / 4);
  if (m == 0) return MP_OK;
  hipLaunchKernelGGL(scatter_rows_f32_bf16_kernel, GRID1D(m), g, rows, (bf16_t*)out, n, dim);
  return mp_check_launch("mp_scatter_rows_f32_bf16");
}

extern "C" int mp_dropout_bf16(const void* x, void* y, int64_t n, float p, uint64_t seed, hipStream_t stream) {
  MP_REQUIRE(p >= 0.f && p < 1.f, MP_ERR_ARG, "mp_dropout_bf16: p must be in [0, 1)");
  if (n == 0) return MP_OK;
  hipLaunchKernelGGL(dropout_bf16_kernel, GRID1D(mp_cdiv(n, 8)), (const bf16_t*)x, (bf16_t*)y, n, p, seed);
  return mp_check_launch("mp_dropout_bf16");
}

extern "C" int mp_lora_down_bf16(const void* x, int64_t ldx, const void* A, int64_t lda, void* t, int64_t ldt, void* xd, int64_t ldxd, int tokens,
                                 int K, int R, float p, uint64_t seed, float alpha, const int* rows_dev, float* partial, int64_t partial_floats,
                                 uint8_t* keep_bits, int64_t ld_bits, hipStream_t stream) {
  MP_REQUIRE(!keep_bits || (p > 0.f && ld_bits * 8 >= K), MP_ERR_ARG, "mp_lora_down_bf16: keep_bits come with p > 0 and a row stride of >= K / 8 bytes");
  MP_REQUIRE(tokens >= 0 && K > 0 && K % 64 == 0 && R > 0 && R <= 64, MP_ERR_SHAPE, "mp_lora_down_bf16: K %% 64 == 0 and 0 < R <= 64 (got K %d, R %d)", K, R);
  MP_REQUIRE(ldx % 8 == 0 && lda % 8 == 0 && ldt % 4 == 0 && (!xd || ldxd % 8 == 0) && p >= 0.f && p < 1.f, MP_ERR_ARG, "mp_lora_down_bf16: bad strides / p");
  if (tokens == 0) return MP_OK;
  const int rg = (R + 15) / 16;
  // staged form (coalesced x loads through LDS) when K is a multiple of 256 and the caller brought room for the K-split partials
  const int splits = (int)std::min<int64_t>(std::min<int64_t>(8, K / 256), std::max<int64_t>(1, 1024 / mp_cdiv(tokens, 64)));     // ~1024 workgroups
  static int staged = -1;
  if (staged < 0) { const char* e = getenv("MP_LORA_DOWN_STAGED"); staged = (e && atoi(e) == 0) ? 0 : 1; }                // 0: fragment-shaped loads (A/B)
  if (staged && K % 256 == 0 && partial && partial_floats >= (int64_t)splits * tokens * 16 * rg) {
    const dim3 grid((unsigned)mp_cdiv(tokens, 64), (unsigned)splits), blk(256);
#define MP_GO(RG) hipLaunchKernelGGL((lora_down_staged_kernel<RG>), grid, blk, 0, stream, (const bf16_t*)x, ldx, (const bf16_t*)A, lda, partial, (bf16_t*)xd, ldxd, tokens, K, p, seed, rows_dev, keep_bits, ld_bits)
    switch (rg) { case 1: MP_GO(1); break; case 2: MP_GO(2); break; case 3: MP_GO(3); break; default: MP_GO(4); }
#undef MP_GO
    hipLaunchKernelGGL(lora_down_finish_kernel, GRID1D((int64_t)tokens * 16), partial, (bf16_t*)t, ldt, tokens, 16 * rg, splits, alpha, rows_dev);
    return mp_check_launch("mp_lora_down_bf16(staged)");
  }
  MP_REQUIRE(!keep_bits, MP_ERR_ARG, "mp_lora_down_bf16: keep_bits need the staged form (K %% 256 == 0 and the partial workspace)");
  const dim3 grid((unsigned)mp_cdiv(tokens, 16)), blk(512);
#define MP_GO(RG) hipLaunchKernelGGL((lora_down_kernel<RG>), grid, blk, 0, stream, (const bf16_t*)x, ldx, (const bf16_t*)A, lda, (bf16_t*)t, ldt, (bf16_t*)xd, ldxd, tokens, K, p, seed, alpha, rows_dev)
  switch (rg) { case 1: MP_GO(1); break; case 2: MP_GO(2); break; case 3: MP_GO(3); break; default: MP_GO(4); }
#undef MP_GO
  return mp_check_launch("mp_lora_down_bf16");
}

extern "C" int mp_lora_up_add_bf16(const void* dt, int64_t lddt, const void* AT, const void* dx, int64_t lddx, void* out, int64_t ldo, int tokens,
                                   int K, int R, float p, uint64_t seed, const int* rows_dev, const uint8_t* keep_bits, int64_t ld_bits,
                                   hipStream_t stream) {
  MP_REQUIRE(!keep_bits || (p > 0.f && ld_bits * 8 >= K), MP_ERR_ARG, "mp_lora_up_add_bf16: keep_bits come with p > 0 and a row stride of >= K / 8 bytes");
  MP_REQUIRE(tokens >= 0 && K > 0 && K % 8 == 0 && (R == 8 || R == 16 || R == 32), MP_ERR_SHAPE, "mp_lora_up_add_bf16: K %% 8 == 0, R in {8, 16, 32} (got K %d, R %d)", K, R);
  MP_REQUIRE(lddt % 8 == 0 && lddx % 8 == 0 && ldo % 8 == 0 && p >= 0.f && p < 1.f, MP_ERR_ARG, "mp_lora_up_add_bf16: bad strides / p");
  if (tokens == 0) return MP_OK;
  // tokens per wave: enough waves to hide the loads' latency (~4 per SIMD) without re-reading the chunk's rank vectors too often
  const int64_t chunks = mp_cdiv(K, 512);
  const int tpw = (int)std::min<int64_t>(32, std::max<int64_t>(8, (chunks * tokens / 4096 + 3) / 4 * 4));
  const dim3 grid((unsigned)chunks, (unsigned)mp_cdiv(tokens, 4 * tpw)), blk(256);
#define MP_GO(RP) hipLaunchKernelGGL((lora_up_add_kernel<RP>), grid, blk, 0, stream, (const bf16_t*)dt, lddt, (const bf16_t*)AT, (const bf16_t*)dx, lddx, (bf16_t*)out, ldo, tokens, K, p, seed, tpw, rows_dev, keep_bits, ld_bits)
  switch (R) { case 8: MP_GO(4); break; case 16: MP_GO(8); break; default: MP_GO(16); }
#undef MP_GO
  return mp_check_launch("mp_lora_up_add_bf16");
}

extern "C" int mp_lora_up_add_swiglu_bwd_bf16(const void* dt, int64_t lddt, const void* AT, const void* dact, int64_t lddact, const void* gu, void* dgu,
                                              int tokens, int ff, int R, float p, uint64_t seed, const uint8_t* keep_bits, int64_t ld_bits,
                                              hipStream_t stream) {
  MP_REQUIRE(!keep_bits || (p > 0.f && ld_bits * 8 >= ff), MP_ERR_ARG, "mp_lora_up_add_swiglu_bwd_bf16: keep_bits come with p > 0 and a row stride of >= ff / 8 bytes");
  MP_REQUIRE(tokens >= 0 && ff > 0 && ff % 32 == 0 && (R == 8 || R == 16 || R == 32), MP_ERR_SHAPE, "mp_lora_up_add_swiglu_bwd_bf16: ff %% 32 == 0, R in {8, 16, 32} (got ff %d, R %d)", ff, R);
  MP_REQUIRE(lddt % 8 == 0 && lddact % 8 == 0 && p >= 0.f && p < 1.f && gu && dgu, MP_ERR_ARG, "mp_lora_up_add_swiglu_bwd_bf16: bad strides / p / null");
  if (tokens == 0) return MP_OK;
  const int64_t chunks = mp_cdiv(ff, 512);
  const int tpw = (int)std::min<int64_t>(32, std::max<int64_t>(8, (chunks * tokens / 4096 + 3) / 4 * 4));
  const dim3 grid((unsigned)chunks, (unsigned)mp_cdiv(tokens, 4 * tpw)), blk(256);
#define MP_GO(RP) hipLaunchKernelGGL((lora_up_add_kernel<RP, true>), grid, blk, 0, stream, (const bf16_t*)dt, lddt, (const bf16_t*)AT, (const bf16_t*)dact, lddact, (bf16_t*)nullptr, (int64_t)0, tokens, ff, p, seed, tpw, (const int*)nullptr, keep_bits, ld_bits, (const bf16_t*)gu, (bf16_t*)dgu)
  switch (R) { case 8: MP_GO(4); break; case 16: MP_GO(8); break; default: MP_GO(16); }
#undef MP_GO
  return mp_check_launch("mp_lora_up_add_swiglu_bwd_bf16");
}

extern "C" int mp_lora_grad_unpack_f32(const float* dB, const float* dAT, const int64_t* rows, int R, int k0, int r, int fin, int fout, float* gB,
                                       float* gA, hipStream_t stream) {
  MP_REQUIRE(R > 0 && r > 0 && k0 >= 0 && k0 + r <= R && fin > 0 && fout > 0 && dB && dAT && rows && gB && gA, MP_ERR_ARG, "mp_lora_grad_unpack_f32: bad arguments");
  const int64_t n = (int64_t)fout * r + (int64_t)r * fin;
  hipLaunchKernelGGL(lora_grad_unpack_kernel, GRID1D(n), dB, dAT, rows, R, k0, r, fin, fout, gB, gA);
  return mp_check_launch("mp_lora_grad_unpack_f32");
}

extern "C" int mp_lora_grad_unpack_partials_f32(const float* dBp, const float* dATp, int chunksB, int chunksA, float scaleB, float scaleA,
                                                const int64_t* rows, int R, int k0, int r, int fin, int fout, int W, float* gB, float* gA,
                                                hipStream_t stream) {
  MP_REQUIRE(R > 0 && r > 0 && k0 >= 0 && k0 + r <= R && fin > 0 && fout > 0 && W >= 1 && chunksB >= 1 && chunksA >= 1 && dBp && dATp && rows && gB && gA,
             MP_ERR_ARG, "mp_lora_grad_unpack_partials_f32: bad arguments");
  const int64_t n = (int64_t)fout * r + (int64_t)r * fin;
  hipLaunchKernelGGL(lora_grad_unpack_partials_kernel, GRID1D(n), dBp, dATp, chunksB, chunksA, scaleB, scaleA, rows, R, k0, r, fin, fout, W, gB, gA);
  return mp_check_launch("mp_lora_grad_unpack_partials_f32");
}

extern "C" int mp_lora_pack(const float* a, const float* b, const int64_t* rows, void* A, void* AT, void* B, void* BT, int r, int fin, int fout,
                            int k0, int W, float bscale, void* Bx, int64_t ldbx, float xscale, hipStream_t stream) {
  MP_REQUIRE(r > 0 && k0 >= 0 && k0 + r <= 64 && fin > 0 && fout > 0 && W >= fout, MP_ERR_SHAPE, "mp_lora_pack: bad shape");
  const int64_t n = (int64_t)r * fin + (int64_t)fout * r;
  hipLaunchKernelGGL(lora_pack_kernel, GRID1D(n), a, b, rows, (bf16_t*)A, (bf16_t*)AT, (bf16_t*)B, (bf16_t*)BT, r, fin, fout, k0, W, bscale,
                     (bf16_t*)Bx, ldbx, xscale);
  return mp_check_launch("mp_lora_pack");
}

extern "C" int mp_lora_pack_batched(const void* descs, int n, int64_t max_elems, hipStream_t stream) {
  MP_REQUIRE(descs && n >= 1 && max_elems >= 1, MP_ERR_ARG, "mp_lora_pack_batched: bad arguments");
  hipLaunchKernelGGL(lora_pack_batched_kernel, dim3((unsigned)mp_cdiv(max_elems, 256), (unsigned)n), dim3(256), 0, stream, (const LoraPackDesc*)descs);
  return mp_check_launch("mp_lora_pack_batched");
}

extern "C" int mp_moe_combine_bwd_bf16(const void* dout, const void* y, const int* expert, const int* slot, const float* weight, void* dy,
                                       float* dw, int64_t tokens, int dim, int capacity, int top_k, hipStream_t stream) {
  MP_REQUIRE(dim % 8 == 0 && capacity >= 0 && (top_k == 1 || top_k == 2), MP_ERR_SHAPE, "mp_moe_combine_bwd_bf16: bad shape");
  if (tokens == 0) return MP_OK;
  hipLaunchKernelGGL(moe_combine_bwd_kernel, dim3((unsigned)mp_cdiv(tokens * top_k, 4)), dim3(256), 0, stream, (const bf16_t*)dout, (const bf16_t*)y,
                     expert, slot, weight, (bf16_t*)dy, dw, tokens, dim, capacity, top_k);
  return mp_check_launch("mp_moe_combine_bwd_bf16");
}

extern "C" int mp_moe_gate_bwd_f32(const float* gates, const int* expert, const int* slot, const float* dw, const long long* exp_counts,
                                   const float* c_aux, float aux_coef, float* dlogits, int64_t tokens, int n_experts, int top_k,
                                   hipStream_t stream) {
  MP_REQUIRE(n_experts >= 1 && n_experts <= 8 && (top_k == 1 || top_k == 2), MP_ERR_SHAPE, "mp_moe_gate_bwd_f32: experts <= 8, top_k 1 or 2");
  if (tokens == 0) return MP_OK;
  hipLaunchKernelGGL(moe_gate_bwd_kernel, GRID1D(tokens), gates, expert, slot, dw, exp_counts, c_aux, aux_coef, dlogits, tokens, n_experts, top_k);
  return mp_check_launch("mp_moe_gate_bwd_f32");
}

extern "C" int mp_moe_gate_dgrad_bf16(const float* dlogits, const float* wg, void* dx, int64_t tokens, int dim, int n_experts, hipStream_t stream) {
  MP_REQUIRE(dim % 8 == 0 && n_experts >= 1 && n_experts <= 8, MP_ERR_SHAPE, "mp_moe_gate_dgrad_bf16: bad shape");
  const int64_t n = tokens * (dim / 8);
  if (n == 0) return MP_OK;
  hipLaunchKernelGGL(moe_gate_dgrad_kernel, GRID1D(n), dlogits, wg, (bf16_t*)dx, tokens, dim, n_experts);
  return mp_check_launch("mp_moe_gate_dgrad_bf16");
}

extern "C" int mp_embed_grad_f32(const void* g, const int64_t* rows_sorted, const int64_t* seg, const int64_t* ids, float* out, int64_t n_unique,
                                 int dim, hipStream_t stream) {
  MP_REQUIRE(dim % 4 == 0, MP_ERR_SHAPE, "mp_embed_grad_f32: dim %% 4 != 0");
  if (n_unique == 0) return MP_OK;
  hipLaunchKernelGGL(embed_grad_kernel, dim3((unsigned)n_unique), dim3(256), 0, stream, (const bf16_t*)g, rows_sorted, seg, ids, out, dim);
  return mp_check_launch("mp_embed_grad_f32");
}

extern "C" int mp_rmsnorm_wgrad_f32(const void* x, int64_t ldx, const void* dy, int64_t ldy, const float* rs, float* dw, float* partial,
                                    int64_t partial_floats, int64_t rows, int dim, hipStream_t stream) {
  const int chunks = (int)mp_cdiv(rows, 256);
  MP_REQUIRE(rows > 0 && dim > 0 && partial && partial_floats >= (int64_t)chunks * dim, MP_ERR_WORKSPACE,
             "mp_rmsnorm_wgrad_f32: partial needs %lld floats", (long long)((int64_t)chunks * dim));
  hipLaunchKernelGGL(rmsnorm_wgrad_partial_kernel, dim3((unsigned)mp_cdiv(dim, 256), (unsigned)chunks), dim3(256), 0, stream, (const bf16_t*)x, ldx,
                     (const bf16_t*)dy, ldy, rs, partial, rows, dim);
  hipLaunchKernelGGL(tn_skinny_reduce_kernel, dim3((unsigned)mp_cdiv(dim, 256)), dim3(256), 0, stream, partial, dw, (int64_t)dim, chunks, 1.f);
  return mp_check_launch("mp_rmsnorm_wgrad_f32");
}

extern "C" int mp_gelu_fwd_bf16(const void* x, void* y, int64_t n, hipStream_t stream) {
  MP_REQUIRE(n % 8 == 0, MP_ERR_SHAPE, "mp_gelu_fwd_bf16: n %% 8 != 0");
  if (n == 0) return MP_OK;
  hipLaunchKernelGGL(gelu_fwd_bf16_kernel, GRID1D(n / 8), (const bf16_t*)x, (bf16_t*)y, n);
  return mp_check_launch("mp_gelu_fwd_bf16");
}

extern "C" int mp_gelu_bwd_bf16(const void* x, const void* dy, void* dx, int64_t n, hipStream_t stream) {
  MP_REQUIRE(n % 8 == 0, MP_ERR_SHAPE, "mp_gelu_bwd_bf16: n %% 8 != 0");
  if (n == 0) return MP_OK;
  hipLaunchKernelGGL(gelu_bwd_bf16_kernel, GRID1D(n / 8), (const bf16_t*)x, (const bf16_t*)dy, (bf16_t*)dx, n);
  return mp_check_launch("mp_gelu_bwd_bf16");
}

extern "C" int mp_region_point_mean_bwd_bf16(const float* xy, const int64_t* offsets, const int* map_index, const void* dout, void* dfmap,
                                             float* wt, int n_maps, int n_masks, int h, int w, int C, hipStream_t stream) {
  MP_REQUIRE(n_maps > 0 && n_masks >= 0 && h > 0 && w > 0 && C % 8 == 0 && wt, MP_ERR_SHAPE, "mp_region_point_mean_bwd_bf16: bad shape");
  if (n_masks > 0) {
    hipLaunchKernelGGL(region_weights_kernel, GRID1D((int64_t)n_masks * h * w), xy, offsets, wt, n_masks, h, w);
  }
  hipLaunchKernelGGL(region_fmap_grad_kernel, GRID1D((int64_t)n_maps * h * w * (C / 8)), wt, (const bf16_t*)dout, map_index, (bf16_t*)dfmap, n_maps,
                     n_masks, h * w, C);
  return mp_check_launch("mp_region_point_mean_bwd_bf16");
}

extern "C" int mp_conv3x3s2_c1_pre_bf16(const void* img, int img_dtype, const float* w, const float* bias, void* out, int n, int H, int W, int CO,
                                        hipStream_t stream) {
  MP_REQUIRE(CO % 8 == 0 && (img_dtype == MP_BF16 || img_dtype == MP_F32), MP_ERR_SHAPE, "mp_conv3x3s2_c1_pre_bf16: bad arguments");
  const int OH = (H + 2 - 3) / 2 + 1, OW = (W + 2 - 3) / 2 + 1;
  const int64_t total = (int64_t)n * OH * OW * (CO / 8);
  if (total == 0) return MP_OK;
  if (img_dtype == MP_F32) hipLaunchKernelGGL(conv3x3s2_c1_pre_kernel<float>, GRID1D(total), (const float*)img, w, bias, (bf16_t*)out, n, H, W, OH, OW, CO);
  else hipLaunchKernelGGL(conv3x3s2_c1_pre_kernel<bf16_t>, GRID1D(total), (const bf16_t*)img, w, bias, (bf16_t*)out, n, H, W, OH, OW, CO);
  return mp_check_launch("mp_conv3x3s2_c1_pre_bf16");
}

extern "C" int mp_conv3x3s2_c1_wgrad_f32(const void* img, int img_dtype, const void* dpre, float* dw, float* db, int n, int H, int W, int CO,
                                         hipStream_t stream) {
  MP_REQUIRE(CO > 0 && (img_dtype == MP_BF16 || img_dtype == MP_F32), MP_ERR_SHAPE, "mp_conv3x3s2_c1_wgrad_f32: bad arguments");
  const int OH = (H + 2 - 3) / 2 + 1, OW = (W + 2 - 3) / 2 + 1;
  if (img_dtype == MP_F32) hipLaunchKernelGGL(conv3x3s2_c1_wgrad_kernel<float>, dim3(10, CO), dim3(256), 0, stream, (const float*)img, (const bf16_t*)dpre, dw, db, n, H, W, OH, OW, CO);
  else hipLaunchKernelGGL(conv3x3s2_c1_wgrad_kernel<bf16_t>, dim3(10, CO), dim3(256), 0, stream, (const bf16_t*)img, (const bf16_t*)dpre, dw, db, n, H, W, OH, OW, CO);
  return mp_check_launch("mp_conv3x3s2_c1_wgrad_f32");
}

extern "C" int mp_adaptive_avgpool_tokens_bwd_bf16(const void* dout, void* dx, int n, int len_in, int len_out, int C, hipStream_t stream) {
  MP_REQUIRE(C % 8 == 0 && len_in > 0 && len_out > 0, MP_ERR_SHAPE, "mp_adaptive_avgpool_tokens_bwd_bf16: bad shape");
  const int64_t total = (int64_t)n * len_in * (C / 8);
  if (total == 0) return MP_OK;
  hipLaunchKernelGGL(adaptive_avgpool_tokens_bwd_kernel, GRID1D(total), (const bf16_t*)dout, (bf16_t*)dx, n, len_in, len_out, C);
  return mp_check_launch("mp_adaptive_avgpool_tokens_bwd_bf16");
}

extern "C" int mp_col2im_k3s2p1_bf16(const void* dcols, void* dx, int n, int H, int W, int C, hipStream_t stream) {
  MP_REQUIRE(C % 8 == 0, MP_ERR_SHAPE, "mp_col2im_k3s2p1_bf16: bad shape");
  const int OH = (H + 2 - 3) / 2 + 1, OW = (W + 2 - 3) / 2 + 1;
  const int64_t total = (int64_t)n * H * W * (C / 8);
  if (total == 0) return MP_OK;
  hipLaunchKernelGGL(col2im_k3s2p1_kernel, GRID1D(total), (const bf16_t*)dcols, (bf16_t*)dx, n, H, W, C, OH, OW);
  return mp_check_launch("mp_col2im_k3s2p1_bf16");
}
