// The trainable fp32 mask tail as ONE launch per direction: text_hidden_fcs -> two-way transformer -> hypernetwork /
// IoU heads forward, and the whole chain rule of the same backward, each as a PROGRAM of tile operations that one persistent grid
// walks phase by phase with a grid barrier between phases.  Reference sites: TwoWayTransformer / TwoWayAttentionBlock / Attention
// (model/segment_anything_med2d/modeling/transformer.py:62-106,151-182,185-244), MaskDecoder.predict_masks (mask_decoder.py:113-153),
// text_hidden_fcs (model/MedPLIB.py:152-164).
//
// Why a program and not ~550 launches: the tail is 0.3 GFLOP per prompt spread over GEMMs of 6 x 256 and 256 x 256 rows — every
// launch is latency-bound, and with the decoder adapters training the tail sits on the step's critical path.  The host
// (medplib_amd/tail_program.py) lowers the module graph ONCE into a table of 256-byte op descriptors (seven op types: GEMM with fused
// operand adds / bias / residual / ReLU / ReLU-mask / second accumulate target / column sums, fixed-order REDUCE, LayerNorm forward and
// backward, the attention core forward and backward per (prompt, head), a 2-D broadcast copy), assigns phases by buffer hazards, and
// this kernel executes it: workgroup b takes tiles b, b + G, ... of the phase's combined tile list.  Every sum runs in a fixed order
// (split-K partials and column sums are combined by REDUCE ops, no float atomics): results are bit-reproducible.
//
// Inter-workgroup visibility follows the microarchitecture guide's XCD-hierarchical barrier: every wave drains its stores, the last
// workgroup of an XCD issues the agent-scope release (L2 write-back) and meets the other XCDs on a top counter, everyone polls ONE word
// relaxed with s_sleep, then ONE agent-scope acquire (L1 invalidate) + __syncthreads covers the workgroup.  Data produced inside the launch is never read through const __restrict__
// pointers (no scalar-cache path).  The spin is bounded: a barrier that cannot complete within tens of seconds sets sync[1] and every workgroup leaves
// the launch (no trap); the host reads sync[1] back and falls back to the op-by-op tail.
#include "common.h"

int mp_device_cus();            // gemm256_bf16.hip (cached per device)

namespace {

enum { OP_GEMM = 1, OP_REDUCE = 2, OP_LN_FWD = 3, OP_LN_BWD = 4, OP_ATTN_FWD = 5, OP_ATTN_BWD = 6, OP_COPY2D = 7 };
enum { F_TRANS_A = 1, F_TRANS_B = 2, F_RELU = 4, F_ACCUM = 8, F_CS_ACCUM = 16, F_A_BF16 = 32 };

struct TailOp {
  int type, flags, ntiles, tile_begin;
  int M, N, K, i0, i1, i2, i3, pad0;
  float f0, f1, f2, f3;
  int64_t ld[12];
  uint64_t p[12];
};
static_assert(sizeof(TailOp) == 256, "TailOp is packed by medplib_amd/tail_program.py as 256 bytes");

struct SlotArgs { uint64_t base[8]; };       // the kernel argument
struct Slots { const uint64_t* base; };       // ... copied to LDS once: operand slots are indexed dynamically (a kernel-argument array would go to scratch)

constexpr int POOL = 16384;                 // floats of LDS per workgroup (64 KB, thirty-two of them hold the barrier's context and the slot table)

__device__ __forceinline__ float* ptr(const Slots& s, uint64_t a) {
  return a ? (float*)(s.base[a >> 56] + (a & 0x00FFFFFFFFFFFFFFull)) : nullptr;
}
__device__ __forceinline__ int wrap(int r, int rows) { return rows > 0 ? ((rows & (rows - 1)) == 0 ? (r & (rows - 1)) : r % rows) : r; }
// a wave-uniform index laundered into a VGPR: the load that uses it takes the vector path (in-launch produced data must not go through the
// scalar cache, which the agent-scope acquire does not refresh)
__device__ __forceinline__ int64_t vidx(int64_t i) { asm volatile("" : "+v"(i)); return i; }

// ---------------------------------------------------------------------------------------------------------------- GEMM
// C[M,N] = epi(alpha * op(A [+ A2]) op(B [+ B2])): 64 x 64 tile, K slabs of 64 through LDS, v_mfma_f32_32x32x2_f32 (an fmaf chain over k in
// ascending order).  tile -> (tm, tn, split).  splits > 1: raw partials to C + split * ld[8] (and column-sum partials to p9 + split * M); the
// epilogue then belongs to the REDUCE op that follows.  Operands whose contiguous dimension is 16-byte addressable are staged with 16-byte
// loads (four per thread, operand and slab, all in flight together); anything else takes the scalar form (sixteen 4-byte loads).
__device__ __forceinline__ bool al16(const void* p) { return ((uintptr_t)p & 15) == 0; }

// K <= 16 with op(A) = A^T: an outer-product update (the weight gradient of a layer that saw only a handful of rows, e.g. text_hidden_fcs:
// dW [4096, 4096] += dy^T x over 8 rows).  No LDS: thread (ty, tx) owns a 4 x 4 block, reads its A / B columns for every k with 16-byte loads and
// streams C with 16-byte read-modify-writes — the op is the 2 x 64 MB of C traffic, nothing else.
__device__ void gemm_smallk_tile(const TailOp& g, const Slots& S, int tile) {
  const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
  const int tiles_m = (g.M + 63) / 64;
  const int tm = tile % tiles_m, tn = tile / tiles_m;
  const int gm = tm * 64 + ty * 4, gn = tn * 64 + tx * 4;
  const float* A = ptr(S, g.p[0]); const float* B = ptr(S, g.p[1]);
  float* C = ptr(S, g.p[2]);
  float acc[4][4] = {};
  float cs[4] = {0.f, 0.f, 0.f, 0.f};
  const bool ok = gm < g.M && gn < g.N;                    // M, N multiples of 4 (checked by the caller): a block is inside or outside
  for (int kb = 0; kb < g.K; kb += 8) {                     // eight k at a time: sixteen 16-byte requests in flight
    f32x4 av[8], bv[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      av[k] = (f32x4){0.f, 0.f, 0.f, 0.f}; bv[k] = av[k];
      if (kb + k < g.K && ok) {
        av[k] = *(const f32x4*)(A + (int64_t)(kb + k) * g.ld[0] + gm);
        bv[k] = *(const f32x4*)(B + (int64_t)(kb + k) * g.ld[1] + gn);
      }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        cs[i] += av[k][i];
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(av[k][i], bv[k][j], acc[i][j]);
      }
  }
  if (!ok) return;
  float* cso = ptr(S, g.p[9]);
  if (cso && tn == 0 && tx == 0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) cso[gm + i] = (g.flags & F_CS_ACCUM) ? cso[gm + i] + cs[i] : cs[i];
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    f32x4* c = (f32x4*)(C + (int64_t)(gm + i) * g.ld[2] + gn);
    f32x4 v = {g.f0 * acc[i][0], g.f0 * acc[i][1], g.f0 * acc[i][2], g.f0 * acc[i][3]};
    if (g.flags & F_ACCUM) { const f32x4 o = *c; v += o; }
    *c = v;
  }
}

__device__ void gemm_tile(const TailOp& g, const Slots& S, int tile, float* smem) {
  constexpr int BM = 64, BN = 64, BK = 64, NL = BK / 4;
  const bool tA = g.flags & F_TRANS_A, tB = g.flags & F_TRANS_B;
  const float* A = ptr(S, g.p[0]); const float* B = ptr(S, g.p[1]);
  const float* A2 = ptr(S, g.p[4]); const float* B2 = ptr(S, g.p[5]);
  const int64_t lda = g.ld[0], ldb = g.ld[1], lda2 = g.ld[6], ldb2 = g.ld[7];
  const int splits = g.i0 > 1 ? g.i0 : 1;
  // 16-byte staging: row strides and bases on 16 bytes, and the contiguous extent a multiple of 4 (a 16-byte group is inside or outside)
  const bool a_bf16 = g.flags & F_A_BF16;                   // A holds bf16 values (the upsampler's tokens): scalar staging, converted on the way
  const bool vecA = !a_bf16 && !(lda & 3) && al16(A) && !((tA ? g.M : g.K) & 3) && (!A2 || (!(lda2 & 3) && al16(A2)));
  const bool vecB = !(ldb & 3) && al16(B) && !((tB ? g.K : g.N) & 3) && (!B2 || (!(ldb2 & 3) && al16(B2)));
  if (g.K <= 16 && tA && !tB && splits == 1 && vecA && vecB && !A2 && !B2 && !g.p[3] && !g.p[6] && !g.p[7] && !g.p[8] && !(g.flags & F_RELU) &&
      !(g.ld[2] & 3) && g.ld[9] <= 1 && !a_bf16 && al16(ptr(S, g.p[2]))) {
    gemm_smallk_tile(g, S, tile);
    return;
  }
  float (*sA)[BM + 4] = (float (*)[BM + 4])smem;
  float (*sB)[BN + 4] = (float (*)[BN + 4])(smem + BK * (BM + 4));
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int tiles_m = (g.M + BM - 1) / BM, tiles_n = (g.N + BN - 1) / BN;
  const int tm = tile % tiles_m, tn = (tile / tiles_m) % tiles_n, sp = tile / (tiles_m * tiles_n);
  const int m0 = tm * BM, n0 = tn * BN;
  const int a2_rows = g.i1, b2_rows = g.i2;
  int kbeg = 0, kend = g.K;
  if (splits > 1) {
    const int per = ((g.K + splits - 1) / splits + BK - 1) / BK * BK;
    kbeg = sp * per;
    kend = min(g.K, kbeg + per);
  }
  f32x16 macc = {};
  const int qm = (wv >> 1) * 32, qn = (wv & 1) * 32;
  float ra[NL], rb[NL];
  float asum[4] = {0.f, 0.f, 0.f, 0.f};                    // column sums of the transposed A operand: sum over k of A[k, m]
  float* cs_out = (tA && tn == 0) ? ptr(S, g.p[9]) : nullptr;
  // one operand's slab into registers.  kc = contiguous along k (non-transposed A / transposed B): scalar form element (x = id / 64, k = id % 64),
  // vector form group (x = id4 / 16, k = 4 (id4 % 16)); otherwise contiguous along x: scalar (k = id / 64, x = id % 64), vector (k = id4 / 16, x = 4 (id4 % 16))
  auto load = [&](const float* P, const float* P2, int64_t ld, int64_t ld2, int rows2, bool kc, bool vec, int x0, int X, int k0, float* r, bool bf) __attribute__((always_inline)) {
    if (vec) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int id4 = tid + i * 256, c4 = id4 & 15, hi = id4 >> 4;
        const int x = x0 + (kc ? hi : 4 * c4), k = k0 + (kc ? 4 * c4 : hi);
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (x < X && k < kend) {
          const int row = kc ? x : k, col = kc ? k : x;
          v = *(const f32x4*)(P + (int64_t)row * ld + col);
          if (P2) v += *(const f32x4*)(P2 + (int64_t)wrap(row, rows2) * ld2 + col);
        }
        r[4 * i] = v[0]; r[4 * i + 1] = v[1]; r[4 * i + 2] = v[2]; r[4 * i + 3] = v[3];
      }
    } else {
#pragma unroll
      for (int i = 0; i < NL; ++i) {
        const int id = tid + i * 256;
        const int x = x0 + (kc ? id / BK : (id & 63)), k = k0 + (kc ? (id & (BK - 1)) : (id >> 6));
        float v = 0.f;
        if (x < X && k < kend) {
          const int row = kc ? x : k, col = kc ? k : x;
          v = bf ? (float)((const bf16_t*)P)[(int64_t)row * ld + col] : P[(int64_t)row * ld + col];
          if (P2) v += P2[(int64_t)wrap(row, rows2) * ld2 + col];
        }
        r[i] = v;
      }
    }
  };
  auto store = [&](float (*s)[BM + 4], bool kc, bool vec, const float* r) __attribute__((always_inline)) {
    if (vec) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int id4 = tid + i * 256, c4 = id4 & 15, hi = id4 >> 4;
        if (kc) {
#pragma unroll
          for (int j = 0; j < 4; ++j) s[4 * c4 + j][hi] = r[4 * i + j];
        } else {
          *(f32x4*)&s[hi][4 * c4] = (f32x4){r[4 * i], r[4 * i + 1], r[4 * i + 2], r[4 * i + 3]};
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < NL; ++i) {
        const int id = tid + i * 256;
        if (kc) s[id & (BK - 1)][id / BK] = r[i];
        else s[id >> 6][id & 63] = r[i];
      }
    }
  };
  auto gload = [&](int k0) __attribute__((always_inline)) {
    load(A, A2, lda, lda2, a2_rows, !tA, vecA, m0, g.M, k0, ra, a_bf16);
    load(B, B2, ldb, ldb2, b2_rows, tB, vecB, n0, g.N, k0, rb, false);
  };
  if (kbeg < kend) gload(kbeg);
  for (int k0 = kbeg; k0 < kend; k0 += BK) {
    store(sA, !tA, vecA, ra);
    store(sB, tB, vecB, rb);
    if (cs_out) {                                          // this thread's columns: vector form 4 (tid % 16) .. + 3, scalar form tid % 64
      if (vecA) {
#pragma unroll
        for (int i = 0; i < NL; ++i) asum[i & 3] += ra[i];
      } else {
#pragma unroll
        for (int i = 0; i < NL; ++i) asum[0] += ra[i];
      }
    }
    __syncthreads();
    if (k0 + BK < kend) gload(k0 + BK);
#pragma unroll
    for (int k = 0; k < BK; k += 2)
      macc = __builtin_amdgcn_mfma_f32_32x32x2f32(sA[k + (lane >> 5)][qm + (lane & 31)], sB[k + (lane >> 5)][qn + (lane & 31)], macc, 0, 0, 0);
    __syncthreads();
  }
  if (cs_out) {                                            // partial rows: 16 thread groups (vector) or 4 waves (scalar), combined in ascending order
    const int nparts = vecA ? 16 : 4;
    if (vecA) {
#pragma unroll
      for (int j = 0; j < 4; ++j) smem[(tid >> 4) * 64 + 4 * (tid & 15) + j] = asum[j];
    } else {
      smem[wv * 64 + lane] = asum[0];
    }
    __syncthreads();
    if (tid < 64 && m0 + tid < g.M) {
      float t = 0.f;
      for (int q = 0; q < nparts; ++q) t += smem[q * 64 + tid];
      float* o = cs_out + (splits > 1 ? (int64_t)sp * g.M : 0) + m0 + tid;
      *o = (splits == 1 && (g.flags & F_CS_ACCUM)) ? *o + t : t;
    }
    __syncthreads();
  }
  float* C = ptr(S, g.p[2]);
  const int64_t ldc = g.ld[2];
  const int64_t csc = g.ld[9] > 0 ? g.ld[9] : 1;            // column stride of C (a weight gradient written straight into a [Cin, Cout, 2, 2] tensor)
  const int gn = n0 + qn + (lane & 31);
  if (splits > 1) {
    C += (int64_t)sp * g.ld[8];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int gm = m0 + qm + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      if (gm < g.M && gn < g.N) C[(int64_t)gm * ldc + gn] = macc[r];
    }
    return;
  }
  const float* bias = ptr(S, g.p[3]); const float* R = ptr(S, g.p[6]); const float* MK = ptr(S, g.p[7]);
  float* C2 = ptr(S, g.p[8]);
  const float bv = (bias && gn < g.N) ? bias[gn] : 0.f;
  float rv[16], mv[16], c2v[16], cv[16];                   // the epilogue's operand reads, all issued before the first use
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int gm = m0 + qm + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    const bool in = gm < g.M && gn < g.N;
    rv[r] = (R && in) ? R[(int64_t)gm * g.ld[3] + gn] : 0.f;
    mv[r] = (MK && in) ? MK[(int64_t)gm * g.ld[4] + gn] : 1.f;
    c2v[r] = (C2 && in) ? C2[(int64_t)gm * g.ld[5] + gn] : 0.f;
    cv[r] = ((g.flags & F_ACCUM) && in) ? C[(int64_t)gm * ldc + gn * csc] : 0.f;
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int gm = m0 + qm + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    if (gm >= g.M || gn >= g.N) continue;
    float v = g.f0 * macc[r] + bv + rv[r];
    if (g.flags & F_RELU) v = fmaxf(v, 0.f);
    if (!(mv[r] > 0.f)) v = 0.f;
    if (C2) C2[(int64_t)gm * g.ld[5] + gn] = c2v[r] + v;
    C[(int64_t)gm * ldc + gn * csc] = cv[r] + v;
  }
}

// ---------------------------------------------------------------------------------------------------------------- REDUCE
// out[r, c] = epi(alpha * sum_{s < K} in[s * ld[2] + r * ld[0] + c]), r < M, c < N, the sum in ascending s: split-K finish, column sums of the
// split weight gradients, LayerNorm weight / bias gradients from the per-tile partials, token gradients over the prompts.  1024 elements per tile;
// the loads of eight consecutive s are issued together (a loop with one load in flight pays the fabric latency K times).
__device__ void reduce_tile(const TailOp& g, const Slots& S, int tile) {
  const float* in = ptr(S, g.p[0]); float* out = ptr(S, g.p[1]);
  const float* bias = ptr(S, g.p[2]); const float* R = ptr(S, g.p[3]); const float* MK = ptr(S, g.p[4]);
  const int64_t total = (int64_t)g.M * g.N;
  const int64_t ocs = g.ld[5] > 0 ? g.ld[5] : 1;          // column stride of `out`
  const float* src[4]; float acc[4]; int rr[4], cc[4]; bool ok[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int64_t e = (int64_t)tile * 1024 + j * 256 + threadIdx.x;
    ok[j] = e < total;
    const int64_t ee = ok[j] ? e : 0;
    rr[j] = (int)(ee / g.N); cc[j] = (int)(ee % g.N);
    src[j] = in + (int64_t)rr[j] * g.ld[0] + cc[j];
    acc[j] = 0.f;
  }
  for (int s0 = 0; s0 < g.K; s0 += 8) {
    float v[4][8];
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j][u] = (ok[j] && s0 + u < g.K) ? src[j][(int64_t)(s0 + u) * g.ld[2]] : 0.f;
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int j = 0; j < 4; ++j) if (s0 + u < g.K) acc[j] += v[j][u];
  }
  float rv[4], mv[4], ov[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    rv[j] = (R && ok[j]) ? R[(int64_t)rr[j] * g.ld[3] + cc[j]] : 0.f;
    mv[j] = (MK && ok[j]) ? MK[(int64_t)rr[j] * g.ld[4] + cc[j]] : 1.f;
    ov[j] = ((g.flags & F_ACCUM) && ok[j]) ? out[(int64_t)rr[j] * g.ld[1] + cc[j] * ocs] : 0.f;
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    if (!ok[j]) continue;
    float v = g.f0 * acc[j];
    if (bias) v += bias[cc[j]];
    v += rv[j];
    if (g.flags & F_RELU) v = fmaxf(v, 0.f);
    if (!(mv[j] > 0.f)) v = 0.f;
    out[(int64_t)rr[j] * g.ld[1] + cc[j] * ocs] = ov[j] + v;
  }
}

// ---------------------------------------------------------------------------------------------------------------- LayerNorm
// 16 rows per tile, one wave per row (four rows each, all four rows' values requested before the first reduction); the arithmetic of
// ln_fwd_f32_kernel / ln_bwd_f32_kernel (small_ops_f32.hip).  dim <= 512, a multiple of 64.
__device__ void ln_fwd_tile(const TailOp& g, const Slots& S, int tile) {
  const float* x = ptr(S, g.p[0]); const float* w = ptr(S, g.p[1]); const float* b = ptr(S, g.p[2]);
  float* y = ptr(S, g.p[3]); float* mean_out = ptr(S, g.p[4]); float* rstd_out = ptr(S, g.p[5]);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, dim = g.N, nq = dim >> 6;
  float xv[4][8];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int row = tile * 16 + wv * 4 + q;
#pragma unroll
    for (int u = 0; u < 8; ++u) xv[q][u] = (row < g.M && u < nq) ? x[(int64_t)row * g.ld[0] + lane + 64 * u] : 0.f;
  }
  float wr[8], br[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) { wr[u] = u < nq ? w[lane + 64 * u] : 0.f; br[u] = u < nq ? b[lane + 64 * u] : 0.f; }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int row = tile * 16 + wv * 4 + q;
    if (row >= g.M) break;
    float s = 0.f;
#pragma unroll
    for (int u = 0; u < 8; ++u) if (u < nq) s += xv[q][u];
    const float mean = wave_sum(s) / (float)dim;
    float v = 0.f;
#pragma unroll
    for (int u = 0; u < 8; ++u) if (u < nq) { const float d = xv[q][u] - mean; v += d * d; }
    const float rstd = 1.f / sqrtf(wave_sum(v) / (float)dim + g.f0);
    float* yr = y + (int64_t)row * g.ld[1];
#pragma unroll
    for (int u = 0; u < 8; ++u) if (u < nq) yr[lane + 64 * u] = (xv[q][u] - mean) * rstd * wr[u] + br[u];
    if (lane == 0) { mean_out[row] = mean; rstd_out[row] = rstd; }
  }
}
// dx = rstd * (g - mean(g) - xhat * mean(g * xhat)), g = dy * w; the tile's partial dw / db (sum over its 16 rows, row order, then wave order) to
// p6 + tile * 2 N.
__device__ void ln_bwd_tile(const TailOp& g, const Slots& S, int tile, float* smem) {
  const float* dy = ptr(S, g.p[0]); const float* x = ptr(S, g.p[1]); const float* w = ptr(S, g.p[2]);
  const float* mean = ptr(S, g.p[3]); const float* rstd = ptr(S, g.p[4]);
  float* dx = ptr(S, g.p[5]); float* part = ptr(S, g.p[6]);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, dim = g.N, nq = dim >> 6;
  float pw[8], pb[8], wr[8], xv[4][8], dv[4][8], mu[4], rs[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int row = tile * 16 + wv * 4 + q;
    const bool in = row < g.M;
    mu[q] = in ? mean[vidx(row)] : 0.f; rs[q] = in ? rstd[vidx(row)] : 0.f;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      xv[q][u] = (in && u < nq) ? x[(int64_t)row * g.ld[1] + lane + 64 * u] : 0.f;
      dv[q][u] = (in && u < nq) ? dy[(int64_t)row * g.ld[0] + lane + 64 * u] : 0.f;
    }
  }
#pragma unroll
  for (int u = 0; u < 8; ++u) { pw[u] = pb[u] = 0.f; wr[u] = u < nq ? w[lane + 64 * u] : 0.f; }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int row = tile * 16 + wv * 4 + q;
    if (row >= g.M) break;
    float sg = 0.f, sgx = 0.f;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (u >= nq) break;
      const float xh = (xv[q][u] - mu[q]) * rs[q], d = dv[q][u], gg = d * wr[u];
      sg += gg; sgx += gg * xh;
      pw[u] += d * xh; pb[u] += d;
    }
    sg = wave_sum(sg) / (float)dim;
    sgx = wave_sum(sgx) / (float)dim;
    float* dxr = dx + (int64_t)row * g.ld[2];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (u >= nq) break;
      const float xh = (xv[q][u] - mu[q]) * rs[q], gg = dv[q][u] * wr[u];
      dxr[lane + 64 * u] = rs[q] * (gg - sg - xh * sgx);
    }
  }
  if (part) {
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (u >= nq) break;
      smem[wv * 2 * dim + lane + 64 * u] = pw[u];
      smem[wv * 2 * dim + dim + lane + 64 * u] = pb[u];
    }
    __syncthreads();
    for (int c = threadIdx.x; c < 2 * dim; c += 256)
      part[(int64_t)tile * 2 * dim + c] = ((smem[c] + smem[2 * dim + c]) + smem[4 * dim + c]) + smem[6 * dim + c];
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------------------------- attention core
// One (prompt, head) per tile, everything in LDS: softmax(scale * Q K^T) V with Nq, Nk in {6, 256} and d in {16, 32} here (general within the
// pool: 2 (Nq + Nk)(d + 1) + 2 Nq Nk <= POOL, checked by the host).  P [b, h, Nq, Nk] is kept for the backward.
// rows x d floats from global rows (stride ld) into LDS rows of d + 1; 16-byte loads when addressable, eight requests in flight per thread
__device__ __forceinline__ void lds_rows(float* dst, const float* src, int rows, int d, int64_t ld) {
  if (!(d & 3) && !(ld & 3) && al16(src)) {
    const int d4 = d >> 2, total = rows * d4;
    for (int e0 = threadIdx.x; e0 < total; e0 += 256 * 8) {
      f32x4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int e = e0 + u * 256;
        if (e < total) { const int i = e / d4, t = e - i * d4; v[u] = *(const f32x4*)(src + (int64_t)i * ld + 4 * t); }
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int e = e0 + u * 256;
        if (e < total) {
          const int i = e / d4, t = e - i * d4;
          float* o = dst + i * (d + 1) + 4 * t;
          o[0] = v[u][0]; o[1] = v[u][1]; o[2] = v[u][2]; o[3] = v[u][3];
        }
      }
    }
    return;
  }
  const int total = rows * d;
  for (int e0 = threadIdx.x; e0 < total; e0 += 256 * 8) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) { const int e = e0 + u * 256; if (e < total) { const int i = e / d, t = e - i * d; v[u] = src[(int64_t)i * ld + t]; } }
#pragma unroll
    for (int u = 0; u < 8; ++u) { const int e = e0 + u * 256; if (e < total) { const int i = e / d, t = e - i * d; dst[i * (d + 1) + t] = v[u]; } }
  }
}
__device__ void attn_fwd_tile(const TailOp& g, const Slots& S, int tile, float* smem) {
  const int Nq = g.M, Nk = g.N, d = g.K, H = g.i0, dp = d + 1;
  const int b = tile / H, h = tile - b * H;
  const float* Q = ptr(S, g.p[0]) + b * g.ld[4] + h * d; const float* K = ptr(S, g.p[1]) + b * g.ld[5] + h * d;
  const float* V = ptr(S, g.p[2]) + b * g.ld[6] + h * d;
  float* O = ptr(S, g.p[3]) + b * g.ld[7] + h * d; float* P = ptr(S, g.p[4]) + (int64_t)tile * Nq * Nk;
  float* sQ = smem; float* sK = sQ + Nq * dp; float* sV = sK + Nk * dp; float* sS = sV + Nk * dp;
  lds_rows(sQ, Q, Nq, d, g.ld[0]); lds_rows(sK, K, Nk, d, g.ld[1]); lds_rows(sV, V, Nk, d, g.ld[2]);
  __syncthreads();
  for (int e = threadIdx.x; e < Nq * Nk; e += 256) {
    const int i = e / Nk, j = e - i * Nk;
    float a = 0.f;
    for (int t = 0; t < d; ++t) a = fmaf(sQ[i * dp + t], sK[j * dp + t], a);
    sS[e] = a * g.f0;
  }
  __syncthreads();
  if (Nk <= 32) {
    for (int i = threadIdx.x; i < Nq; i += 256) {
      float m = -INFINITY;
      for (int j = 0; j < Nk; ++j) m = fmaxf(m, sS[i * Nk + j]);
      float s = 0.f;
      for (int j = 0; j < Nk; ++j) { const float ex = expf(sS[i * Nk + j] - m); sS[i * Nk + j] = ex; s += ex; }
      const float inv = 1.f / s;
      for (int j = 0; j < Nk; ++j) sS[i * Nk + j] *= inv;
    }
  } else {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int i = wv; i < Nq; i += 4) {
      float m = -INFINITY;
      for (int j = lane; j < Nk; j += 64) m = fmaxf(m, sS[i * Nk + j]);
      m = wave_max(m);
      float s = 0.f;
      for (int j = lane; j < Nk; j += 64) { const float ex = expf(sS[i * Nk + j] - m); sS[i * Nk + j] = ex; s += ex; }
      const float inv = 1.f / wave_sum(s);
      for (int j = lane; j < Nk; j += 64) sS[i * Nk + j] *= inv;
    }
  }
  __syncthreads();
  for (int e = threadIdx.x; e < Nq * Nk; e += 256) P[e] = sS[e];
  for (int e = threadIdx.x; e < Nq * d; e += 256) {
    const int i = e / d, t = e - i * d;
    float a = 0.f;
    for (int j = 0; j < Nk; ++j) a = fmaf(sS[i * Nk + j], sV[j * dp + t], a);
    O[(int64_t)i * g.ld[3] + t] = a;
  }
  __syncthreads();
}
// dV = P^T dO, dP = dO V^T, dS = scale * P (dP - rowsum(dP P)), dQ = dS K, dK = dS^T Q  (AttentionCoreFn.backward, autograd_ops.py)
__device__ void attn_bwd_tile(const TailOp& g, const Slots& S, int tile, float* smem) {
  const int Nq = g.M, Nk = g.N, d = g.K, H = g.i0, dp = d + 1;
  const int b = tile / H, h = tile - b * H;
  const float* Q = ptr(S, g.p[0]) + b * g.ld[4] + h * d; const float* K = ptr(S, g.p[1]) + b * g.ld[5] + h * d;
  const float* V = ptr(S, g.p[2]) + b * g.ld[6] + h * d; const float* P = ptr(S, g.p[3]) + (int64_t)tile * Nq * Nk;
  const float* dO = ptr(S, g.p[4]) + b * g.ld[7] + h * d;
  float* dQ = ptr(S, g.p[5]) + (int64_t)b * Nq * g.ld[8] + h * d; float* dK = ptr(S, g.p[6]) + (int64_t)b * Nk * g.ld[9] + h * d;
  float* dV = ptr(S, g.p[7]) + (int64_t)b * Nk * g.ld[10] + h * d;
  float* sQ = smem; float* sdO = sQ + Nq * dp; float* sK = sdO + Nq * dp; float* sV = sK + Nk * dp;
  float* sP = sV + Nk * dp; float* sdS = sP + Nq * Nk;
  lds_rows(sQ, Q, Nq, d, g.ld[0]); lds_rows(sdO, dO, Nq, d, g.ld[3]); lds_rows(sK, K, Nk, d, g.ld[1]); lds_rows(sV, V, Nk, d, g.ld[2]);
  for (int e0 = threadIdx.x; e0 < Nq * Nk; e0 += 256 * 8) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = (e0 + u * 256 < Nq * Nk) ? P[e0 + u * 256] : 0.f;
#pragma unroll
    for (int u = 0; u < 8; ++u) if (e0 + u * 256 < Nq * Nk) sP[e0 + u * 256] = v[u];
  }
  __syncthreads();
  for (int e = threadIdx.x; e < Nk * d; e += 256) {
    const int j = e / d, t = e - j * d;
    float a = 0.f;
    for (int i = 0; i < Nq; ++i) a = fmaf(sP[i * Nk + j], sdO[i * dp + t], a);
    dV[(int64_t)j * g.ld[10] + t] = a;
  }
  for (int e = threadIdx.x; e < Nq * Nk; e += 256) {
    const int i = e / Nk, j = e - i * Nk;
    float a = 0.f;
    for (int t = 0; t < d; ++t) a = fmaf(sdO[i * dp + t], sV[j * dp + t], a);
    sdS[e] = a;
  }
  __syncthreads();
  if (Nk <= 32) {
    for (int i = threadIdx.x; i < Nq; i += 256) {
      float s = 0.f;
      for (int j = 0; j < Nk; ++j) s += sP[i * Nk + j] * sdS[i * Nk + j];
      for (int j = 0; j < Nk; ++j) sdS[i * Nk + j] = g.f0 * sP[i * Nk + j] * (sdS[i * Nk + j] - s);
    }
  } else {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int i = wv; i < Nq; i += 4) {
      float s = 0.f;
      for (int j = lane; j < Nk; j += 64) s += sP[i * Nk + j] * sdS[i * Nk + j];
      s = wave_sum(s);
      for (int j = lane; j < Nk; j += 64) sdS[i * Nk + j] = g.f0 * sP[i * Nk + j] * (sdS[i * Nk + j] - s);
    }
  }
  __syncthreads();
  for (int e = threadIdx.x; e < Nq * d; e += 256) {
    const int i = e / d, t = e - i * d;
    float a = 0.f;
    for (int j = 0; j < Nk; ++j) a = fmaf(sdS[i * Nk + j], sK[j * dp + t], a);
    dQ[(int64_t)i * g.ld[8] + t] = a;
  }
  for (int e = threadIdx.x; e < Nk * d; e += 256) {
    const int j = e / d, t = e - j * d;
    float a = 0.f;
    for (int i = 0; i < Nq; ++i) a = fmaf(sdS[i * Nk + j], sQ[i * dp + t], a);
    dK[(int64_t)j * g.ld[9] + t] = a;
  }
  __syncthreads();
}

// ---------------------------------------------------------------------------------------------------------------- COPY2D
// out[r, c] = A[r % a_rows, c] + B[r % b_rows, c] (either may be absent: zero), r < M, c < N: broadcast adds, token assembly, zero fill.
__device__ void copy2d_tile(const TailOp& g, const Slots& S, int tile) {
  const float* A = ptr(S, g.p[0]); const float* B = ptr(S, g.p[1]); float* out = ptr(S, g.p[2]);
  const int64_t total = (int64_t)g.M * g.N;
  float av[4], bv[4], ov[4]; int64_t oi[4]; bool ok[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int64_t e = (int64_t)tile * 1024 + j * 256 + threadIdx.x;
    ok[j] = e < total;
    const int64_t ee = ok[j] ? e : 0;
    const int r = (int)(ee / g.N), c = (int)(ee % g.N);
    oi[j] = (int64_t)r * g.ld[2] + c;
    av[j] = (A && ok[j]) ? A[(int64_t)wrap(r, g.i1) * g.ld[0] + c] : 0.f;
    bv[j] = (B && ok[j]) ? B[(int64_t)wrap(r, g.i2) * g.ld[1] + c] : 0.f;
    ov[j] = ((g.flags & F_ACCUM) && ok[j]) ? out[oi[j]] : 0.f;
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) if (ok[j]) out[oi[j]] = ov[j] + (av[j] + bv[j]);
}

// ---------------------------------------------------------------------------------------------------------------- the grid barrier
// XCD-hierarchical (microarchitecture guide, price list row "barrier-xcd"): a workgroup arrives on the counter of the XCD it is physically on
// (HW_REG_XCC_ID — membership is COUNTED in a prologue, nothing is assumed about placement); the last arriver of an XCD issues ONE agent-scope
// release (buffer_wbl2: the write-back of that XCD's L2 covers every workgroup of the XCD, all of which drained their stores before arriving),
// arrives on the top counter, waits for the other XCDs, acquires, and publishes the XCD's generation word; the others poll that word and acquire
// (their own CU's L1).  Eight write-backs and eight pollers of the top word per barrier instead of 256 of each.
// sync words: [0] top counter, [1] give-up flag, [2] prologue counter, [8..15] per-XCD arrivals, [16..23] per-XCD generation, [24..31] members.
#define RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT
// Bounded poll.  A barrier that cannot complete (a workgroup of the grid never became resident within tens of seconds) sets the give-up flag
// and returns false; every other poller sees the flag within 256 polls and returns false too, so the launch ENDS (no trap: the context stays
// usable) with sync[1] != 0, which the host reads back and treats as "this launch's results are void" (medplib_amd/tail_program.py falls
// back to the op-by-op tail from then on).
__device__ __forceinline__ bool spin_until(unsigned* w, unsigned target, unsigned* flag) {
  unsigned spins = 0;
  while (__hip_atomic_load(w, RLX_AGENT) < target) {
    __builtin_amdgcn_s_sleep(2);
    ++spins;
    if ((spins & 255u) == 0u && __hip_atomic_load(flag, RLX_AGENT) != 0u) return false;
    if (spins > (1u << 24)) {
      __hip_atomic_store(flag, 1u, RLX_AGENT);
      return false;
    }
  }
  return true;
}
struct BarrierCtx { unsigned xcc, members, nx, dead; };
__device__ bool barrier_prologue(unsigned* sync, BarrierCtx* ctx) {
  if (threadIdx.x == 0) {
    unsigned x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    x &= 7u;
    __hip_atomic_fetch_add(sync + 24 + x, 1u, RLX_AGENT);
    // the arrival is a RELEASE (the membership increment above has completed before it is visible) and the counts are read behind an
    // ACQUIRE: whoever sees all gridDim.x arrivals sees every membership increment (two relaxed atomics on different words carry no order)
    __hip_atomic_fetch_add(sync + 2, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    const bool ok = spin_until(sync + 2, gridDim.x, sync + 1);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    unsigned nx = 0;
    for (int i = 0; i < 8; ++i) nx += __hip_atomic_load(sync + 24 + i, RLX_AGENT) != 0u;
    ctx->xcc = x; ctx->members = __hip_atomic_load(sync + 24 + x, RLX_AGENT); ctx->nx = nx; ctx->dead = ok ? 0u : 1u;
  }
  __syncthreads();
  return ctx->dead == 0u;
}
__device__ bool grid_barrier(unsigned* sync, unsigned epoch, BarrierCtx* ctx) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // every storing wave drains: its stores are in the XCD's L2
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned x = ctx->xcc;
    bool ok;
    const unsigned old = __hip_atomic_fetch_add(sync + 8 + x, 1u, RLX_AGENT);
    if (old + 1u == ctx->members * epoch) {                // the last of this XCD
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the post-write-back wait where the compiler cannot drop it (guide, pitfall 12)
      __hip_atomic_fetch_add(sync, 1u, RLX_AGENT);
      ok = spin_until(sync, ctx->nx * epoch, sync + 1);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      __hip_atomic_store(sync + 16 + x, epoch, RLX_AGENT);
    } else {
      ok = spin_until(sync + 16 + x, epoch, sync + 1);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    if (!ok) ctx->dead = 1u;
  }
  __syncthreads();
  return ctx->dead == 0u;
}

__global__ __launch_bounds__(256) void tail_program_kernel(const TailOp* ops, const int* phase_ops, const int* phase_tiles, int n_phases,
                                                           SlotArgs SA, unsigned* sync, unsigned long long* stamps) {
  __shared__ float smem[POOL - 32];
  __shared__ BarrierCtx bctx;
  __shared__ uint64_t s_slots[8];
  if (threadIdx.x < 8) s_slots[threadIdx.x] = SA.base[threadIdx.x];
  const Slots S{s_slots};
  if (!barrier_prologue(sync, &bctx)) return;
  for (int ph = 0; ph < n_phases; ++ph) {
    const int ob = phase_ops[ph], oe = phase_ops[ph + 1], nt = phase_tiles[ph];
    for (int t = blockIdx.x; t < nt; t += gridDim.x) {
      int o = ob;
      while (o + 1 < oe && t >= ops[o + 1].tile_begin) ++o;
      const TailOp& g = ops[o];
      const int tile = t - g.tile_begin;
      switch (g.type) {
        case OP_GEMM: gemm_tile(g, S, tile, smem); break;
        case OP_REDUCE: reduce_tile(g, S, tile); break;
        case OP_LN_FWD: ln_fwd_tile(g, S, tile); break;
        case OP_LN_BWD: ln_bwd_tile(g, S, tile, smem); break;
        case OP_ATTN_FWD: attn_fwd_tile(g, S, tile, smem); break;
        case OP_ATTN_BWD: attn_bwd_tile(g, S, tile, smem); break;
        case OP_COPY2D: copy2d_tile(g, S, tile); break;
        default: break;
      }
    }
    if (ph + 1 < n_phases && !grid_barrier(sync, (unsigned)(ph + 1), &bctx)) return;    // gave up: sync[1] is set, the results are void
    if (stamps && blockIdx.x == 0 && threadIdx.x == 0) stamps[ph] = wall_clock64();
  }
}

}  // namespace

// ops: n_ops descriptors on the device; phase_ops [n_phases + 1] / phase_tiles [n_phases] on the device; slots: EIGHT base addresses on the HOST
// (slot 0 must be 0: absolute operand addresses); sync: 128 zeroable bytes on the device (barrier counters, give-up flag at word 1); stamps: n_phases x 8 bytes
// on the device or null (a 100 MHz stamp per phase end, workgroup 0); grid: workgroups (<= compute units: every workgroup must be resident).
extern "C" int mp_tail_program_run(const void* ops, const int* phase_ops, const int* phase_tiles, int n_phases, const uint64_t* slots,
                                   void* sync, void* stamps, int grid, hipStream_t stream) {
  MP_REQUIRE(ops && phase_ops && phase_tiles && slots && sync, MP_ERR_ARG, "mp_tail_program_run: null argument");
  MP_REQUIRE(n_phases >= 1 && grid >= 1, MP_ERR_ARG, "mp_tail_program_run: n_phases, grid >= 1");
  MP_REQUIRE(slots[0] == 0, MP_ERR_ARG, "mp_tail_program_run: slot 0 is the absolute address space (base 0)");
  // every workgroup must be resident for the barrier: at most one per compute unit of THIS device (mp_device_cus caches per device), and the
  // occupancy query must admit at least one (64 KB of LDS per workgroup; the query is cached per device as well)
  const int cus = mp_device_cus();
  static int per_cu[64] = {};
  int dev = 0;
  (void)hipGetDevice(&dev);
  const int di = dev >= 0 && dev < 64 ? dev : 0;
  if (per_cu[di] == 0) {
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, tail_program_kernel, 256, 0) != hipSuccess) nb = 0;
    per_cu[di] = nb > 0 ? nb : -1;
  }
  MP_REQUIRE(per_cu[di] >= 1, MP_ERR_LAUNCH, "mp_tail_program_run: the occupancy query admits no resident workgroup of the program kernel");
  if (grid > cus) grid = cus;
  SlotArgs S;
  for (int i = 0; i < 8; ++i) S.base[i] = slots[i];
  if (hipMemsetAsync(sync, 0, 128, stream) != hipSuccess) { mp_set_error("mp_tail_program_run: memset failed"); return MP_ERR_LAUNCH; }
  hipLaunchKernelGGL(tail_program_kernel, dim3((unsigned)grid), dim3(256), 0, stream, (const TailOp*)ops, phase_ops, phase_tiles, n_phases, S,
                     (unsigned*)sync, (unsigned long long*)stamps);
  return mp_check_launch("mp_tail_program_run");
}
