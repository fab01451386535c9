// bf16 GEMV for the single-token decode steps of evaluate() (HF generate with a KV cache, MedPLIB.py:592-606): y[m, n] = x[m, :] . W[n, :]
// for M <= 8 rows.  One decode step reads every LLM weight once, so this kernel is bound by the HBM stream of W (13.2 GB per
// token at 7B): a wave owns four consecutive output columns (four W rows of K bf16), streams them with 16-byte loads — eight in
// flight per lane — against the x chunk it holds in registers, reduces across the wave, and applies the same fused epilogues as the
// GEMM (bias, activation, residual, SwiGLU pairing over the interleaved gate|up rows).  MoE decode: `w_index[m]` (device) selects
// the expert's weight matrix per row and `row_scale[m]` / `row_keep[m]` carry the top-1 gate probability and the capacity drop.
#include "gemm_common.h"

namespace {

constexpr int GV_MAXM = 8;

struct GemvArgs {
  const bf16_t* x; int64_t ldx;
  const bf16_t* W; int64_t ldw; int64_t strideW;
  void* y; int64_t ldy;
  const float* bias; const bf16_t* residual; int64_t ldr;
  const int* w_index;        // [M] device: weight matrix per row (null = one shared matrix)
  const float* row_scale;    // [M] device or null
  const int* row_keep;       // [M] device or null: rows with row_keep[m] < 0 get scale 0 (capacity-dropped tokens)
  int M, N, K, act, out_f32;
  float alpha;
};

// The weight stream: every byte is read once per token by ONE wave, so the loads are marked non-temporal (streaming lines do not displace
// the x rows, norm weights and KV cache other kernels of the decode step re-read from L2; MP_GEMV_NT=0 at build time restores the default
// policy for an A/B).
#ifndef MP_GEMV_NT
#define MP_GEMV_NT 1
#endif
#ifndef MP_GEMV_PIPE
#define MP_GEMV_PIPE 1        // 0 at build time: the norm-folded GEMVs request a trip only after the previous one was consumed (round 4; A/B)
#endif
__device__ __forceinline__ bf16x8 gv_ldw(const bf16_t* p) {
#if MP_GEMV_NT
  return __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(p));
#else
  return *reinterpret_cast<const bf16x8*>(p);
#endif
}

__device__ __forceinline__ float gv_dot8(const bf16x8 a, const bf16x8 b, float acc) {
#pragma unroll
  for (int j = 0; j < 8; ++j) acc = fmaf((float)a[j], (float)b[j], acc);
  return acc;
}

// SHARED: all M rows use the same W (W rows are loaded once and reused for every x row)
template <int M, bool SWIGLU>
__global__ __launch_bounds__(256) void gemv_shared_kernel(GemvArgs g) {
  const int lane = threadIdx.x & 63;
  const int wid = blockIdx.x * 4 + (threadIdx.x >> 6);
  // columns of this wave: plain = 4 consecutive; SWIGLU = gate rows {g0, g0+1} and their up rows {g0+32, g0+33}
  int rows[4];
  if (SWIGLU) {
    const int pair = wid * 2;                               // output column pair index
    const int blk = pair >> 5, j = pair & 31;
    rows[0] = blk * 64 + j; rows[1] = rows[0] + 1; rows[2] = rows[0] + 32; rows[3] = rows[0] + 33;
  } else {
#pragma unroll
    for (int r = 0; r < 4; ++r) rows[r] = wid * 4 + r;
  }
  if (rows[0] >= g.N) return;
  const bf16_t* wp[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) wp[r] = g.W + (int64_t)min(rows[r], g.N - 1) * g.ldw;
  float acc[M][4];
#pragma unroll
  for (int m = 0; m < M; ++m)
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[m][r] = 0.f;
  const int iters = g.K / 512;                              // 64 lanes x 8 elements per step
  int k = lane * 8;
  // (four steps per trip for the one-row case, 16 loads in flight, was measured too: o_proj 11.0 -> 11.2 us — no change)
  int it = 0;
  for (; it + 1 < iters; it += 2, k += 1024) {              // two steps per trip: 8 weight loads in flight per lane
    bf16x8 w0[4], w1[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) { w0[r] = gv_ldw(wp[r] + k); w1[r] = gv_ldw(wp[r] + k + 512); }
#pragma unroll
    for (int m = 0; m < M; ++m) {
      const bf16x8 x0 = *reinterpret_cast<const bf16x8*>(g.x + (int64_t)m * g.ldx + k);
      const bf16x8 x1 = *reinterpret_cast<const bf16x8*>(g.x + (int64_t)m * g.ldx + k + 512);
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[m][r] = gv_dot8(x1, w1[r], gv_dot8(x0, w0[r], acc[m][r]));
    }
  }
  if (it < iters) {
    bf16x8 w0[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) w0[r] = gv_ldw(wp[r] + k);
#pragma unroll
    for (int m = 0; m < M; ++m) {
      const bf16x8 x0 = *reinterpret_cast<const bf16x8*>(g.x + (int64_t)m * g.ldx + k);
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[m][r] = gv_dot8(x0, w0[r], acc[m][r]);
    }
  }
  k = iters * 512 + lane * 8;                                // K tail (K % 512 != 0; K % 8 == 0)
  if (k < g.K) {
#pragma unroll
    for (int m = 0; m < M; ++m) {
      const bf16x8 x0 = *reinterpret_cast<const bf16x8*>(g.x + (int64_t)m * g.ldx + k);
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[m][r] = gv_dot8(x0, gv_ldw(wp[r] + k), acc[m][r]);
    }
  }
#pragma unroll
  for (int m = 0; m < M; ++m)
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[m][r] = wave_sum(acc[m][r]);
  if (lane != 0) return;
#pragma unroll
  for (int m = 0; m < M; ++m) {
    if (SWIGLU) {
      bf16_t* yo = reinterpret_cast<bf16_t*>(g.y) + (int64_t)m * g.ldy;
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        const float gf = (float)(bf16_t)(acc[m][p] * g.alpha), uf = (float)(bf16_t)(acc[m][2 + p] * g.alpha);
        const int col = (rows[p] >> 6) * 32 + (rows[p] & 31);
        if (rows[p] < g.N) yo[col] = (bf16_t)(gf * mp_sigmoid_fast(gf) * uf);
      }
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = rows[r];
        if (n >= g.N) continue;
        float v = apply_act(acc[m][r] * g.alpha + (g.bias ? g.bias[n] : 0.f), g.act);
        if (g.out_f32) {
          if (g.residual) v += (float)g.residual[(int64_t)m * g.ldr + n];
          reinterpret_cast<float*>(g.y)[(int64_t)m * g.ldy + n] = v;
        } else {
          v = (float)(bf16_t)v;                              // same rounding points as the 256x256 GEMM epilogue
          if (g.residual) v += (float)g.residual[(int64_t)m * g.ldr + n];
          reinterpret_cast<bf16_t*>(g.y)[(int64_t)m * g.ldy + n] = (bf16_t)v;
        }
      }
    }
  }
}

// INDEXED: every row picks its own weight matrix (MoE experts); rows are processed one after the other
template <bool SWIGLU>
__global__ __launch_bounds__(256) void gemv_indexed_kernel(GemvArgs g) {
  const int lane = threadIdx.x & 63;
  const int wid = blockIdx.x * 4 + (threadIdx.x >> 6);
  int rows[4];
  if (SWIGLU) {
    const int pair = wid * 2;
    const int blk = pair >> 5, j = pair & 31;
    rows[0] = blk * 64 + j; rows[1] = rows[0] + 1; rows[2] = rows[0] + 32; rows[3] = rows[0] + 33;
  } else {
#pragma unroll
    for (int r = 0; r < 4; ++r) rows[r] = wid * 4 + r;
  }
  if (rows[0] >= g.N) return;
  for (int m = 0; m < g.M; ++m) {
    const bf16_t* Wm = g.W + (int64_t)g.w_index[m] * g.strideW;
    float scale = g.row_scale ? g.row_scale[m] : 1.f;
    if (g.row_keep && g.row_keep[m] < 0) scale = 0.f;
    const bf16_t* wp[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) wp[r] = Wm + (int64_t)min(rows[r], g.N - 1) * g.ldw;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    int k = lane * 8;
    // four steps per trip: 16 weight loads in flight per lane (one step per trip left every trip waiting on its own 4 loads: the down
    // projection's 21.5 trips per row were 21.5 dependent round trips); same accumulation order
    for (; k + 3 * 512 < g.K; k += 4 * 512) {
      bf16x8 xs[4], ws[4][4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        xs[u] = *reinterpret_cast<const bf16x8*>(g.x + (int64_t)m * g.ldx + k + u * 512);
#pragma unroll
        for (int r = 0; r < 4; ++r) ws[u][r] = gv_ldw(wp[r] + k + u * 512);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r] = gv_dot8(xs[u], ws[u][r], acc[r]);
    }
    for (; k < g.K; k += 512) {
      const bf16x8 x0 = *reinterpret_cast<const bf16x8*>(g.x + (int64_t)m * g.ldx + k);
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[r] = gv_dot8(x0, gv_ldw(wp[r] + k), acc[r]);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] = wave_sum(acc[r]);
    if (lane != 0) continue;
    bf16_t* yo = reinterpret_cast<bf16_t*>(g.y) + (int64_t)m * g.ldy;
    if (SWIGLU) {
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        const float gf = (float)(bf16_t)acc[p], uf = (float)(bf16_t)acc[2 + p];
        const int col = (rows[p] >> 6) * 32 + (rows[p] & 31);
        if (rows[p] < g.N) yo[col] = (bf16_t)(gf * mp_sigmoid_fast(gf) * uf);
      }
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = rows[r];
        if (n >= g.N) continue;
        float v = scale * (float)(bf16_t)acc[r];            // combine: gate weight on the bf16-rounded expert output, then the residual
        if (g.residual) v += (float)g.residual[(int64_t)m * g.ldr + n];
        yo[n] = (bf16_t)v;
      }
    }
  }
}

// The shared core of the norm-folded GEMVs: every wave normalises the M rows (HF rounding points, rmsnorm_bf16_kernel's summation order),
// parks them in its own LDS region and streams its four weight rows against them in a RUNTIME loop, two steps per trip (8 weight loads in
// flight).  Unrolled with the rows in registers the compiler kept ~320 values live in the RoPE form (256 VGPRs + 63 AGPRs, one wave per
// SIMD, 42 us instead of 24); this form takes ~100 VGPRs, four waves per SIMD.  acc[m][r] = the wave-reduced dot products.
template <int M, int NCH>
__device__ __forceinline__ void gv_norm_rows_dot(const GemvArgs& g, const float* __restrict__ nw, float eps, const bf16_t* const (&wp)[4],
                                                 int lane, int wv, float (&acc)[M][4]) {
  __shared__ __attribute__((aligned(16))) bf16_t hsh[4][M][NCH * 512];
#pragma unroll
  for (int m = 0; m < M; ++m) {
    const bf16_t* xr = g.x + (int64_t)m * g.ldx;
    bf16x8 hx[NCH];
    float part[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
      hx[k] = *reinterpret_cast<const bf16x8*>(xr + (k * 64 + lane) * 8);
#pragma unroll
      for (int j = 0; j < 8; ++j) { const float f = (float)hx[k][j]; part[k & 3] += f * f; }
    }
    float ss = 0.f;
#pragma unroll
    for (int w4 = 0; w4 < 4; ++w4) ss += wave_sum(part[w4]);
    const float rs = rsqrtf(ss / (float)g.K + eps);
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
      const int i = (k * 64 + lane) * 8;
      const f32x4 w0 = *reinterpret_cast<const f32x4*>(nw + i), w1 = *reinterpret_cast<const f32x4*>(nw + i + 4);
      bf16x8 o;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const bf16_t t = (bf16_t)((float)hx[k][j] * rs);
        o[j] = (bf16_t)((j < 4 ? w0[j & 3] : w1[j & 3]) * (float)t);
      }
      *reinterpret_cast<bf16x8*>(&hsh[wv][m][i]) = o;        // lane l reads back exactly the chunks it wrote: no barrier needed
    }
  }
#pragma unroll
  for (int m = 0; m < M; ++m)
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[m][r] = 0.f;
  int kk = lane * 8;
#pragma unroll 1
  for (int it = 0; it + 1 < NCH; it += 2, kk += 1024) {
    bf16x8 w0[4], w1[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) { w0[r] = gv_ldw(wp[r] + kk); w1[r] = gv_ldw(wp[r] + kk + 512); }
#pragma unroll
    for (int m = 0; m < M; ++m) {
      const bf16x8 x0 = *reinterpret_cast<const bf16x8*>(&hsh[wv][m][kk]), x1 = *reinterpret_cast<const bf16x8*>(&hsh[wv][m][kk + 512]);
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[m][r] = gv_dot8(x1, w1[r], gv_dot8(x0, w0[r], acc[m][r]));
    }
  }
  if (NCH & 1) {
#pragma unroll
    for (int m = 0; m < M; ++m) {
      const bf16x8 x0 = *reinterpret_cast<const bf16x8*>(&hsh[wv][m][kk]);
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[m][r] = gv_dot8(x0, gv_ldw(wp[r] + kk), acc[m][r]);
    }
  }
#pragma unroll
  for (int m = 0; m < M; ++m)
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[m][r] = wave_sum(acc[m][r]);
}

// Round 4, K a multiple of 2048: the WORKGROUP normalises the rows once — rmsnorm_bf16_kernel's own 256-thread row (thread t owns the
// 16-byte chunks c * 256 + t, block_sum in its order: the same bits by construction) — into one LDS copy its four waves share, instead
// of every wave normalising them for itself: a quarter of the L2 reads of x and the norm weights (3072 waves x 24 KB = 74 MB per
// qkv launch at 7B, all on the same lines, now 18 MB), a quarter of the LDS.  And the first trip of the weight stream is requested BEFORE
// the norm, so the prologue runs under the HBM latency of the first eight loads instead of in front of it.  Must be called by all 256
// threads (`wp` must be valid addresses for waves without work, which skip the epilogue).
template <int M, int NCH, bool PIPE_OK>
__device__ __forceinline__ void gv_norm_rows_dot_block(const GemvArgs& g, const float* __restrict__ nw, float eps, const bf16_t* const (&wp)[4],
                                                       int lane, float (&acc)[M][4]) {
  static_assert(NCH % 4 == 0, "block form: K a multiple of 2048");
  constexpr int NC = NCH / 4;
  __shared__ __attribute__((aligned(16))) bf16_t hsh[M][NCH * 512];
  __shared__ float red[16];
  int kk = lane * 8;
  constexpr bool PIPE = MP_GEMV_PIPE && PIPE_OK && NCH >= 8;
  bf16x8 w0[4], w1[4], u0[4], u1[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) { w0[r] = gv_ldw(wp[r] + kk); w1[r] = gv_ldw(wp[r] + kk + 512); }
  if constexpr (PIPE) {
#pragma unroll
    for (int r = 0; r < 4; ++r) { u0[r] = gv_ldw(wp[r] + kk + 1024); u1[r] = gv_ldw(wp[r] + kk + 1536); }
  }
#pragma unroll
  for (int m = 0; m < M; ++m) {
    const bf16_t* xr = g.x + (int64_t)m * g.ldx;
    bf16x8 hx[NC];
    float ss = 0.f;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      hx[c] = *reinterpret_cast<const bf16x8*>(xr + (c * 256 + (int)threadIdx.x) * 8);
#pragma unroll
      for (int j = 0; j < 8; ++j) { const float f = (float)hx[c][j]; ss += f * f; }
    }
    ss = block_sum(ss, red);
    const float rs = rsqrtf(ss / (float)g.K + eps);
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const int i = (c * 256 + (int)threadIdx.x) * 8;
      const f32x4 n0 = *reinterpret_cast<const f32x4*>(nw + i), n1 = *reinterpret_cast<const f32x4*>(nw + i + 4);
      bf16x8 o;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const bf16_t t = (bf16_t)((float)hx[c][j] * rs);
        o[j] = (bf16_t)((j < 4 ? n0[j & 3] : n1[j & 3]) * (float)t);
      }
      *reinterpret_cast<bf16x8*>(&hsh[m][i]) = o;
    }
  }
  __syncthreads();
#pragma unroll
  for (int m = 0; m < M; ++m)
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[m][r] = 0.f;
  if constexpr (PIPE) {
    // Round 5: the trips overlap.  With a trip's eight loads per lane requested only after the previous trip had been multiplied, the 768
    // workgroups of the qkv launch (one round of the chip, all started together) moved in lockstep through four dependent round trips with
    // the HBM queue draining between them (5.0 TB/s).  Two register sets: trips 0 and 1 are requested before the norm, trip t + 2 as soon as
    // trip t has been multiplied, so 8-16 loads per lane are in flight for the whole row.  The same products in the same order.  20.1 ->
    // 19.25 us per qkv launch (profiles/r05a / r05b_decode_timeline_dense.md); the dense gate|up launch (1376 workgroups = 1.3 rounds,
    // their waves already out of step) measured 30.5 -> 30.9 with it at three waves per SIMD instead of four, so only the qkv form asks for it.
    constexpr int T = NCH / 2;                               // trips of two 512-element steps (NCH % 4 == 0: T is even)
#pragma unroll 1
    for (int t = 0; t < T; t += 2, kk += 2048) {
#pragma unroll
      for (int m = 0; m < M; ++m) {
        const bf16x8 x0 = *reinterpret_cast<const bf16x8*>(&hsh[m][kk]), x1 = *reinterpret_cast<const bf16x8*>(&hsh[m][kk + 512]);
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[m][r] = gv_dot8(x1, w1[r], gv_dot8(x0, w0[r], acc[m][r]));
      }
      if (t + 2 < T) {
#pragma unroll
        for (int r = 0; r < 4; ++r) { w0[r] = gv_ldw(wp[r] + kk + 2048); w1[r] = gv_ldw(wp[r] + kk + 2560); }
      }
      __builtin_amdgcn_sched_barrier(0);                     // (one register set's conversions at a time)
#pragma unroll
      for (int m = 0; m < M; ++m) {
        const bf16x8 x0 = *reinterpret_cast<const bf16x8*>(&hsh[m][kk + 1024]), x1 = *reinterpret_cast<const bf16x8*>(&hsh[m][kk + 1536]);
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[m][r] = gv_dot8(x1, u1[r], gv_dot8(x0, u0[r], acc[m][r]));
      }
      if (t + 2 < T) {
#pragma unroll
        for (int r = 0; r < 4; ++r) { u0[r] = gv_ldw(wp[r] + kk + 3072); u1[r] = gv_ldw(wp[r] + kk + 3584); }
      }
    }
  } else {
#pragma unroll
    for (int m = 0; m < M; ++m) {                            // the first trip: its weights are already on their way
      const bf16x8 x0 = *reinterpret_cast<const bf16x8*>(&hsh[m][kk]), x1 = *reinterpret_cast<const bf16x8*>(&hsh[m][kk + 512]);
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[m][r] = gv_dot8(x1, w1[r], gv_dot8(x0, w0[r], acc[m][r]));
    }
    kk += 1024;
#pragma unroll 1
    for (int it = 2; it + 1 < NCH; it += 2, kk += 1024) {
#pragma unroll
      for (int r = 0; r < 4; ++r) { w0[r] = gv_ldw(wp[r] + kk); w1[r] = gv_ldw(wp[r] + kk + 512); }
#pragma unroll
      for (int m = 0; m < M; ++m) {
        const bf16x8 x0 = *reinterpret_cast<const bf16x8*>(&hsh[m][kk]), x1 = *reinterpret_cast<const bf16x8*>(&hsh[m][kk + 512]);
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[m][r] = gv_dot8(x1, w1[r], gv_dot8(x0, w0[r], acc[m][r]));
      }
    }
  }
#pragma unroll
  for (int m = 0; m < M; ++m)
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[m][r] = wave_sum(acc[m][r]);
}

template <int M, int NCH, bool PIPE_OK = false>
__device__ __forceinline__ void gv_norm_rows_dot_any(const GemvArgs& g, const float* __restrict__ nw, float eps, const bf16_t* const (&wp)[4],
                                                     int lane, int wv, float (&acc)[M][4]) {
  if constexpr (NCH % 4 == 0) gv_norm_rows_dot_block<M, NCH, PIPE_OK>(g, nw, eps, wp, lane, acc);
  else gv_norm_rows_dot<M, NCH>(g, nw, eps, wp, lane, wv, acc);
}

// RMSNorm folded into the GEMV (decode steps: input_layernorm -> qkv projection, final norm -> lm_head).  Every wave normalises the
// row itself — K bf16 (8 KB at 7B) from L2 and two wave reductions, nothing beside the 4 x K weight stream it is about to read — and
// keeps the normalised row in registers; the stand-alone norm kernel's launch (5 us of a 130 us decode layer) disappears.  BIT-IDENTICAL
// with mp_rmsnorm_bf16 followed by mp_gemv_bf16: lane l holds the 16-byte chunks q = k*64 + l, the chunk thread (c = q >> 8, t = q & 255)
// of rmsnorm_bf16_kernel owns, so the sum of squares is accumulated per "virtual wave" (q >> 6) & 3 in that kernel's order (the same
// construction as rmsnorm_gate_kernel, ce_moe.hip), the HF rounding points are the same, and chunk q is also what the GEMV's lane l
// multiplies in its k-th step.
template <int M, int NCH, bool SWIGLU = false>
__global__ __launch_bounds__(256) void gemv_rmsnorm_kernel(GemvArgs g, const float* __restrict__ nw, float eps) {
  const int lane = threadIdx.x & 63;
  const int wid = blockIdx.x * 4 + (threadIdx.x >> 6);
  int rows[4];
  if (SWIGLU) {                                             // as gemv_shared_kernel: gate rows {g0, g0+1} and their up rows {g0+32, g0+33}
    const int pair = wid * 2;
    const int blk = pair >> 5, j = pair & 31;
    rows[0] = blk * 64 + j; rows[1] = rows[0] + 1; rows[2] = rows[0] + 32; rows[3] = rows[0] + 33;
  } else {
#pragma unroll
    for (int r = 0; r < 4; ++r) rows[r] = wid * 4 + r;
  }
  const bf16_t* wp[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) wp[r] = g.W + (int64_t)min(rows[r], g.N - 1) * g.ldw;
  float acc[M][4];
  gv_norm_rows_dot_any<M, NCH>(g, nw, eps, wp, lane, threadIdx.x >> 6, acc);      // (waves beyond N take part: the block form has barriers)
  if (lane != 0 || rows[0] >= g.N) return;
#pragma unroll
  for (int m = 0; m < M; ++m) {
    if (SWIGLU) {                                           // the shared kernel's SwiGLU epilogue, rounding points included
      bf16_t* yo = reinterpret_cast<bf16_t*>(g.y) + (int64_t)m * g.ldy;
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        const float gf = (float)(bf16_t)(acc[m][p] * g.alpha), uf = (float)(bf16_t)(acc[m][2 + p] * g.alpha);
        const int col = (rows[p] >> 6) * 32 + (rows[p] & 31);
        if (rows[p] < g.N) yo[col] = (bf16_t)(gf * mp_sigmoid_fast(gf) * uf);
      }
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = rows[r];
        if (n >= g.N) continue;
        const float v = acc[m][r] * g.alpha;
        if (g.out_f32) reinterpret_cast<float*>(g.y)[(int64_t)m * g.ldy + n] = v;
        else reinterpret_cast<bf16_t*>(g.y)[(int64_t)m * g.ldy + n] = (bf16_t)v;
      }
    }
  }
}

// The decode step's q|k|v projection with BOTH neighbours folded in: input_layernorm in front (gemv_rmsnorm_kernel's prologue) and RoPE at
// the device-side position + the KV-cache append behind it (mp_decode_rope_append_bf16's arithmetic: the projection rounded to bf16, then
// lo' = fma(lo, cos, -(hi * sin)), hi' = fma(lo, sin, hi * cos) in fp32, rounded to bf16 — the forms that kernel compiles to).  A wave of the
// q and k thirds owns the rotary pairs (c, c + 1) and (c + D/2, c + D/2 + 1) of one head instead of four consecutive columns, so both
// partners of a rotation sit in the same lane-0 registers; a wave of the v third keeps four consecutive columns.  Outputs: the rotated q in
// the q third of `y` (the k and v thirds of y are NOT written), rotated k and v at cache position pos_dev[0].  Same bits as the three launches.
struct RopeArgs {
  const float* cos_t; const float* sin_t;    // [positions, D/2]
  bf16_t* ck; bf16_t* cv;                    // caches [B, max_len, H, D]
  const int* pos_dev;
  int H, D;
  int64_t c_sb, c_ss;
};

template <int M, int NCH>
__global__ __launch_bounds__(256) void gemv_rmsnorm_rope_kernel(GemvArgs g, const float* __restrict__ nw, float eps, RopeArgs ra) {
  // the normalised row goes through a per-wave LDS region and the weight stream runs as a RUNTIME loop, two steps per trip (8 loads in
  // flight) like gemv_shared_kernel: with the row in registers and the loop unrolled the compiler kept ~320 values live (256 VGPRs + 63
  // AGPRs, one wave per SIMD) and the launch took 42 us against the 26.5 of the three launches it replaces
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int wid = blockIdx.x * 4 + wv;
  const int d = ra.H * ra.D, half = ra.D / 2;
  const int per_region = d / 4;                            // waves per third
  const bool idle = wid >= 3 * per_region;                 // (takes part in the block-wide norm, computes a valid wave's rows, stores nothing)
  const int widc = idle ? 3 * per_region - 1 : wid;
  const int region = widc / per_region, idx = widc % per_region;
  const int per_head = ra.D / 4;
  const int head = idx / per_head, c = (idx % per_head) * 2;
  const int row0 = region < 2 ? region * d + head * ra.D + c : 2 * d + idx * 4;          // wave-uniform
  const int step2 = region < 2 ? half : 2;                                               // rows: row0, +1, +step2, +step2 + 1
  const bf16_t* wp[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) wp[r] = g.W + (int64_t)(row0 + (r & 1) + (r >> 1) * step2) * g.ldw;
  float acc[M][4];
  gv_norm_rows_dot_any<M, NCH, true>(g, nw, eps, wp, lane, wv, acc);      // overlapped trips: one round of workgroups in lockstep (see the core)
  if (lane != 0 || idle) return;
  const int pos = ra.pos_dev[0];
#pragma unroll
  for (int m = 0; m < M; ++m) {
    float v[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = (float)(bf16_t)(acc[m][r] * g.alpha);        // the projection's own bf16 rounding
    if (region == 2) {
      bf16_t* vd = ra.cv + m * ra.c_sb + (int64_t)pos * ra.c_ss + idx * 4;
#pragma unroll
      for (int r = 0; r < 4; ++r) vd[r] = (bf16_t)v[r];
      continue;
    }
    const float* cs = ra.cos_t + (int64_t)pos * half + c;
    const float* sn = ra.sin_t + (int64_t)pos * half + c;
    bf16_t o[4];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const float lo = v[p], hi = v[2 + p];
      o[p] = (bf16_t)fmaf(lo, cs[p], -(hi * sn[p]));
      o[2 + p] = (bf16_t)fmaf(lo, sn[p], hi * cs[p]);
    }
    bf16_t* dst = region == 0 ? reinterpret_cast<bf16_t*>(g.y) + (int64_t)m * g.ldy + head * ra.D
                              : ra.ck + m * ra.c_sb + (int64_t)pos * ra.c_ss + head * ra.D;
    dst[c] = o[0]; dst[c + 1] = o[1]; dst[half + c] = o[2]; dst[half + c + 1] = o[3];
  }
}

// K-SPLIT form for the narrow projections of a decode step (N <= 4096: o_proj and the down projections, one row of x or the indexed
// expert form).  With a wave per four W rows those launches are 256 workgroups of four waves — one wave per SIMD, 8-16 KB of loads in
// flight per wave and nothing to hide their latency behind: the o projection streamed at 3.2 TB/s, the expert down projection at 4.2
// (gate|up, 1376 workgroups: 5.9).  Here a workgroup of SIXTEEN waves owns the same sixteen W rows: wave (rg, kp) takes the four rows
// of row group rg and the 512-element steps s = kp, kp + 4, kp + 8, ... of K, so a CU has four times the loads in flight and four waves
// per SIMD; the four K parts of a row are added in ascending kp order through LDS (a fixed order: deterministic, but not the
// single-wave order of gemv_shared_kernel — the two forms agree to fp32 rounding, not bit for bit).
// Round 5, measured and dropped: the same kernel with EVERY weight load of a row requested before the first multiply (five steps of 4 x 16
// bytes per lane up front, the x row through LDS; a CU's 352 KB share of the down projection fits its register file): 21.4 us against this
// form's 18.8 on the 90 MB down projection, bit-identical.  A launch here is ~4 us of fixed cost (dispatch, first latency, reduction,
// epilogue) + bytes / 6.3 TB/s — qkv 16.0 + 3.3, o 5.3 + 3.9, gate|up 28.6 + 1.9, down 14.3 + 4.5 (profiles/r05a_decode_timeline_dense.md) —
// and the eight loads per lane it keeps in flight already cover the latency; deeper queues only lengthen the register-file fill.
__global__ __launch_bounds__(1024) void gemv_ksplit_kernel(GemvArgs g) {
  __shared__ float red[4][4][4];                            // [kp][rg][r]
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int rg = wv & 3, kp = wv >> 2;
  const int row0 = (blockIdx.x * 4 + rg) * 4;
  for (int m = 0; m < g.M; ++m) {
    const bf16_t* Wm = g.W + (g.w_index ? (int64_t)g.w_index[m] * g.strideW : 0);
    const bf16_t* wp[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) wp[r] = Wm + (int64_t)min(row0 + r, g.N - 1) * g.ldw;
    const bf16_t* xr = g.x + (int64_t)m * g.ldx;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    int k = lane * 8 + kp * 512;
    for (; k + 2048 < g.K; k += 4096) {                     // two of this wave's steps per trip: 8 weight loads in flight per lane
      const bf16x8 x0 = *reinterpret_cast<const bf16x8*>(xr + k), x1 = *reinterpret_cast<const bf16x8*>(xr + k + 2048);
      bf16x8 w0[4], w1[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) { w0[r] = gv_ldw(wp[r] + k); w1[r] = gv_ldw(wp[r] + k + 2048); }
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[r] = gv_dot8(x1, w1[r], gv_dot8(x0, w0[r], acc[r]));
    }
    if (k < g.K) {
      const bf16x8 x0 = *reinterpret_cast<const bf16x8*>(xr + k);
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[r] = gv_dot8(x0, gv_ldw(wp[r] + k), acc[r]);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] = wave_sum(acc[r]);
    if (m > 0) __syncthreads();                             // the previous row's partials have been read
    if (lane == 0) {
#pragma unroll
      for (int r = 0; r < 4; ++r) red[kp][rg][r] = acc[r];
    }
    __syncthreads();
    if (kp != 0 || lane >= 4) continue;
    const int n = row0 + lane;                               // lane r of the kp = 0 wave finishes row r of its group
    if (n >= g.N) continue;
    const float a = ((red[0][rg][lane] + red[1][rg][lane]) + red[2][rg][lane]) + red[3][rg][lane];
    if (g.w_index) {                                        // gemv_indexed_kernel's epilogue: combine weight on the bf16-rounded expert output, then the residual
      float scale = g.row_scale ? g.row_scale[m] : 1.f;
      if (g.row_keep && g.row_keep[m] < 0) scale = 0.f;
      float v = scale * (float)(bf16_t)a;
      if (g.residual) v += (float)g.residual[(int64_t)m * g.ldr + n];
      reinterpret_cast<bf16_t*>(g.y)[(int64_t)m * g.ldy + n] = (bf16_t)v;
    } else {                                                // gemv_shared_kernel's
      float v = apply_act(a * g.alpha + (g.bias ? g.bias[n] : 0.f), g.act);
      if (g.out_f32) {
        if (g.residual) v += (float)g.residual[(int64_t)m * g.ldr + n];
        reinterpret_cast<float*>(g.y)[(int64_t)m * g.ldy + n] = v;
      } else {
        v = (float)(bf16_t)v;
        if (g.residual) v += (float)g.residual[(int64_t)m * g.ldr + n];
        reinterpret_cast<bf16_t*>(g.y)[(int64_t)m * g.ldy + n] = (bf16_t)v;
      }
    }
  }
}

template <int M>
void launch_shared(const GemvArgs& g, dim3 grid, hipStream_t s) {
  if (g.act == ACT_SWIGLU_PAIR) hipLaunchKernelGGL((gemv_shared_kernel<M, true>), grid, dim3(256), 0, s, g);
  else hipLaunchKernelGGL((gemv_shared_kernel<M, false>), grid, dim3(256), 0, s, g);
}

}  // namespace

extern "C" int mp_gemv_bf16(const void* x, int64_t ldx, const void* W, int64_t ldw, int64_t strideW, void* y, int64_t ldy, const float* bias,
                            const void* residual, int64_t ldr, const int* w_index, const float* row_scale, const int* row_keep, int M, int N,
                            int K, int act, int out_dtype, float alpha, hipStream_t stream) {
  MP_REQUIRE(M >= 1 && M <= GV_MAXM && N > 0 && K > 0 && K % 8 == 0 && ldx % 8 == 0 && ldw % 8 == 0, MP_ERR_SHAPE,
             "mp_gemv_bf16: 1 <= M <= 8, K %% 8 == 0 (M=%d N=%d K=%d)", M, N, K);
  MP_REQUIRE(out_dtype == MP_BF16 || out_dtype == MP_F32, MP_ERR_DTYPE, "mp_gemv_bf16: bad out dtype");
  MP_REQUIRE(act >= 0 && act <= 5, MP_ERR_ARG, "mp_gemv_bf16: bad activation %d", act);
  MP_REQUIRE(act != ACT_SWIGLU_PAIR || (N % 64 == 0 && out_dtype == MP_BF16 && !bias && !residual), MP_ERR_ARG,
             "mp_gemv_bf16: SWIGLU_PAIR needs N %% 64 == 0, bf16 output, no bias / residual");
  MP_REQUIRE(!w_index || (out_dtype == MP_BF16 && !bias && (act == ACT_NONE || act == ACT_SWIGLU_PAIR)), MP_ERR_ARG,
             "mp_gemv_bf16: the indexed (expert) form takes no bias and only NONE / SWIGLU_PAIR");
  GemvArgs g{(const bf16_t*)x, ldx, (const bf16_t*)W, ldw, strideW, y, ldy, bias, (const bf16_t*)residual, ldr, w_index, row_scale, row_keep,
             M, N, K, act, out_dtype == MP_F32, alpha};
  const int waves = act == ACT_SWIGLU_PAIR ? N / 4 : (int)mp_cdiv(N, 4);
  const dim3 grid((unsigned)mp_cdiv(waves, 4));
  // narrow projections with a long K (o_proj, the down projections of a decode step): sixteen waves per sixteen rows, K split four ways
  static int ksplit = -1;
  if (ksplit < 0) { const char* e = getenv("MP_GEMV_KSPLIT"); ksplit = (e && atoi(e) == 0) ? 0 : 1; }     // 0: the one-wave-per-four-rows kernels (A/B)
  if (ksplit && act != ACT_SWIGLU_PAIR && N <= 4096 && K >= 4096 && (w_index || M == 1)) {
    hipLaunchKernelGGL(gemv_ksplit_kernel, grid, dim3(1024), 0, stream, g);
    return mp_check_launch("mp_gemv_bf16(ksplit)");
  }
  if (w_index) {
    if (act == ACT_SWIGLU_PAIR) hipLaunchKernelGGL(gemv_indexed_kernel<true>, grid, dim3(256), 0, stream, g);
    else hipLaunchKernelGGL(gemv_indexed_kernel<false>, grid, dim3(256), 0, stream, g);
    return mp_check_launch("mp_gemv_bf16(indexed)");
  }
  switch (M) {
    case 1: launch_shared<1>(g, grid, stream); break;
    case 2: launch_shared<2>(g, grid, stream); break;
    case 3: launch_shared<3>(g, grid, stream); break;
    case 4: launch_shared<4>(g, grid, stream); break;
    default: {      // 5..8 rows: two passes of <= 4 over the same weights
      GemvArgs a = g; a.M = 4; launch_shared<4>(a, grid, stream);
      GemvArgs b = g; b.M = M - 4; b.x = g.x + 4 * ldx;
      b.y = g.out_f32 ? (void*)(reinterpret_cast<float*>(y) + 4 * ldy) : (void*)(reinterpret_cast<bf16_t*>(y) + 4 * ldy);
      if (g.residual) b.residual = g.residual + 4 * ldr;
      switch (b.M) { case 1: launch_shared<1>(b, grid, stream); break; case 2: launch_shared<2>(b, grid, stream); break;
                     case 3: launch_shared<3>(b, grid, stream); break; default: launch_shared<4>(b, grid, stream); }
    }
  }
  return mp_check_launch("mp_gemv_bf16");
}

extern "C" int mp_gemv_rmsnorm_bf16(const void* x, int64_t ldx, const float* norm_w, float eps, const void* W, int64_t ldw, void* y,
                                    int64_t ldy, int M, int N, int K, int act, int out_dtype, hipStream_t stream) {
  MP_REQUIRE(M >= 1 && M <= 2 && N > 0 && K >= 512 && K % 512 == 0 && K <= 8192 && ldx % 8 == 0 && ldw % 8 == 0, MP_ERR_SHAPE,
             "mp_gemv_rmsnorm_bf16: 1 <= M <= 2, K a multiple of 512 up to 8192 (M=%d N=%d K=%d)", M, N, K);
  MP_REQUIRE(out_dtype == MP_BF16 || out_dtype == MP_F32, MP_ERR_DTYPE, "mp_gemv_rmsnorm_bf16: bad out dtype");
  MP_REQUIRE(norm_w != nullptr, MP_ERR_ARG, "mp_gemv_rmsnorm_bf16: norm weight missing");
  MP_REQUIRE(act == ACT_NONE || (act == ACT_SWIGLU_PAIR && N % 64 == 0 && out_dtype == MP_BF16), MP_ERR_ARG,
             "mp_gemv_rmsnorm_bf16: activation NONE, or SWIGLU_PAIR with N %% 64 == 0 and bf16 output");
  GemvArgs g{(const bf16_t*)x, ldx, (const bf16_t*)W, ldw, 0, y, ldy, nullptr, nullptr, 0, nullptr, nullptr, nullptr, M, N, K, act,
             out_dtype == MP_F32, 1.f};
  const bool sw = act == ACT_SWIGLU_PAIR;
  const dim3 grid((unsigned)mp_cdiv(sw ? N / 4 : mp_cdiv(N, 4), 4));
#define MP_GVN(MM, NC) do { if (sw) hipLaunchKernelGGL((gemv_rmsnorm_kernel<MM, NC, true>), grid, dim3(256), 0, stream, g, norm_w, eps); \
                            else hipLaunchKernelGGL((gemv_rmsnorm_kernel<MM, NC, false>), grid, dim3(256), 0, stream, g, norm_w, eps); } while (0)
#define MP_GVN_M(NC) do { if (M == 1) MP_GVN(1, NC); else MP_GVN(2, NC); } while (0)
  switch (K / 512) {
    case 1: MP_GVN_M(1); break;
    case 2: MP_GVN_M(2); break;
    case 4: MP_GVN_M(4); break;
    case 8: MP_GVN_M(8); break;
    case 16: MP_GVN_M(16); break;
    default: MP_REQUIRE(false, MP_ERR_SHAPE, "mp_gemv_rmsnorm_bf16: K / 512 must be 1, 2, 4, 8 or 16 (K=%d)", K);
  }
#undef MP_GVN_M
#undef MP_GVN
  return mp_check_launch("mp_gemv_rmsnorm_bf16");
}

extern "C" int mp_gemv_rmsnorm_rope_append_bf16(const void* x, int64_t ldx, const float* norm_w, float eps, const void* W_qkv, int64_t ldw,
                                                void* qkv, int64_t ldy, const float* cos_t, const float* sin_t, void* cache_k, void* cache_v,
                                                const int* pos_dev, int M, int heads, int head_dim, int K, int64_t cache_batch_stride,
                                                int64_t cache_seq_stride, hipStream_t stream) {
  MP_REQUIRE(M >= 1 && M <= 2 && heads > 0 && head_dim % 16 == 0 && K >= 512 && K % 512 == 0 && K <= 8192 && ldx % 8 == 0 && ldw % 8 == 0,
             MP_ERR_SHAPE, "mp_gemv_rmsnorm_rope_append_bf16: 1 <= M <= 2, head_dim %% 16 == 0, K a multiple of 512 up to 8192");
  MP_REQUIRE(norm_w && cos_t && sin_t && cache_k && cache_v && pos_dev, MP_ERR_ARG, "mp_gemv_rmsnorm_rope_append_bf16: null operand");
  const int N = 3 * heads * head_dim;
  GemvArgs g{(const bf16_t*)x, ldx, (const bf16_t*)W_qkv, ldw, 0, qkv, ldy, nullptr, nullptr, 0, nullptr, nullptr, nullptr, M, N, K, ACT_NONE,
             0, 1.f};
  RopeArgs ra{cos_t, sin_t, (bf16_t*)cache_k, (bf16_t*)cache_v, pos_dev, heads, head_dim, cache_batch_stride, cache_seq_stride};
  const dim3 grid((unsigned)mp_cdiv(N / 4, 4));
#define MP_GVR(MM, NC) hipLaunchKernelGGL((gemv_rmsnorm_rope_kernel<MM, NC>), grid, dim3(256), 0, stream, g, norm_w, eps, ra)
#define MP_GVR_M(NC) do { if (M == 1) MP_GVR(1, NC); else MP_GVR(2, NC); } while (0)
  switch (K / 512) {
    case 1: MP_GVR_M(1); break;
    case 2: MP_GVR_M(2); break;
    case 4: MP_GVR_M(4); break;
    case 8: MP_GVR_M(8); break;
    case 16: MP_GVR_M(16); break;
    default: MP_REQUIRE(false, MP_ERR_SHAPE, "mp_gemv_rmsnorm_rope_append_bf16: K / 512 must be 1, 2, 4, 8 or 16 (K=%d)", K);
  }
#undef MP_GVR_M
#undef MP_GVR
  return mp_check_launch("mp_gemv_rmsnorm_rope_append_bf16");
}
