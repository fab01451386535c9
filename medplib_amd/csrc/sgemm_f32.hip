// Generic fp32 GEMM for the small, trainable tail of the path (SAM-Med2D mask decoder + text_hidden_fcs, forward and
// backward): C = act(alpha * op(A) op(B) + bias) with NN / NT / TN operand forms, arbitrary sizes, two-level batching
// (e.g. batch x heads) through element strides.  Reference sites: two-way transformer Attention/MLP
// (model/segment_anything_med2d/modeling/transformer.py:185-244), hypernetwork MLPs and ConvTranspose2d-as-GEMM
// (mask_decoder.py:53-65,141-148), text_hidden_fcs (model/MedPLIB.py:152-164).
//
// These problems are tiny (<= 0.3 GFLOP per prompt) and latency-bound; exact fp32 FMA accumulation keeps the parity
// tolerance against the fp32 CPU oracle tight.  64x64x16 tiles, 256 threads, 4x4 register micro-tile, LDS staged,
// optional split-K with fp32 atomics for the skinny (M <= 64) weight-streaming cases.
#include "common.h"

namespace {

struct SgemmArgs {
  const float* A; const float* B; float* C;
  const float* bias;   // [N] or null (added before activation)
  int M, N, K;
  int64_t lda, ldb, ldc;
  int transA, transB;  // op(A) = A^T if transA (A stored [K,M]); op(B) = B^T if transB (B stored [N,K])
  int nb0, nb1;        // batch dims; blockIdx.z = b0 * nb1 + b1
  int64_t sA0, sA1, sB0, sB1, sC0, sC1;
  float alpha, beta;
  int act;             // 0 none, 1 relu, 2 gelu(erf), 3 sigmoid
  int split_k;         // >1: gridDim.y encodes (tile_n, split) and C is accumulated with atomics (beta must be 1, C pre-initialised)
};

// BK = the K depth of one staged slab.  BK = 64 keeps 32 loads per thread in flight and walks K in a quarter of the steps of BK = 16:
// measured over the LoRA step's 768 launches 13.3 -> 12.6 ms (rocprofv3), i.e. the per-slab load latency is NOT what these launches lose
// their time to -- the 504 launches with K >= 64 average 21 us because a 2048 x 256 x 256 product is 128 workgroups of 4 x 4 register
// tiles (LDS-read bound, ~10 % of the fp32 FMA peak), the 264 shorter ones 7 us.  Accumulation order over k is unchanged (ascending), so
// results are bit-identical between the two depths.  MP_SGEMM_BK=16 selects the first version (A/B).
// Round 3: the inner product runs on the matrix cores' f32 form (v_mfma_f32_32x32x2_f32: f32 in, f32 accumulate, the f32 vector rate,
// and — MI355X_MICROARCH.md — bitwise an fmaf chain over k in ascending order, i.e. EXACTLY what the register-tile loop computes).  What it
// buys is not peak but operand traffic: a wave owns a 32 x 32 quadrant and reads ONE A and ONE B float per lane per MFMA (4096 flops)
// instead of eight LDS floats per 32 flops, which is what held the 4 x 4 register tiles at ~10 % of the f32 FMA rate.  Same staging, same
// accumulation order, same results bit for bit; MP_SGEMM_MFMA=0 selects the register-tile loop (A/B).
typedef __attribute__((ext_vector_type(16))) float f32x16;
template <int BK, bool MFMA>
__global__ __launch_bounds__(256) void sgemm_kernel(SgemmArgs g) {
  constexpr int BM = 64, BN = 64, NL = BK / 4;             // NL loads per thread and operand per slab
  __shared__ float sA[BK][BM + 4];
  __shared__ float sB[BK][BN + 4];
  const int tid = threadIdx.x;
  const int tx = tid & 15, ty = tid >> 4;
  const int z = blockIdx.z, b0 = z / g.nb1, b1 = z % g.nb1;
  const float* A = g.A + b0 * g.sA0 + b1 * g.sA1;
  const float* B = g.B + b0 * g.sB0 + b1 * g.sB1;
  float* C = g.C + b0 * g.sC0 + b1 * g.sC1;
  const int tiles_n = (g.N + BN - 1) / BN;
  const int tn = blockIdx.y % tiles_n, sp = blockIdx.y / tiles_n;
  const int m0 = blockIdx.x * BM, n0 = tn * BN;
  int kbeg = 0, kend = g.K;
  if (g.split_k > 1) {
    const int per = ((g.K + g.split_k - 1) / g.split_k + BK - 1) / BK * BK;
    kbeg = sp * per;
    kend = min(g.K, kbeg + per);
    if (kbeg >= kend) return;
  }
  float acc[4][4] = {};
  f32x16 macc = {};
  const int lane = tid & 63, wv = tid >> 6;
  const int qm = (wv >> 1) * 32, qn = (wv & 1) * 32;           // this wave's quadrant of the 64 x 64 tile (MFMA form)
  // global -> register prefetch of the NEXT K-slab overlaps the FMA loop on the current one
  float ra[NL], rb[NL];
  auto gload = [&](int k0) {
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      const int id = tid + i * 256;  // 0 .. 64 * BK - 1
      int m, k;
      if (g.transA) { m = id & 63; k = id >> 6; } else { k = id & (BK - 1); m = id / BK; }
      const int gm = m0 + m, gk = k0 + k;
      ra[i] = (gm < g.M && gk < kend) ? (g.transA ? A[(int64_t)gk * g.lda + gm] : A[(int64_t)gm * g.lda + gk]) : 0.f;
      int n, kb;
      if (g.transB) { kb = id & (BK - 1); n = id / BK; } else { n = id & 63; kb = id >> 6; }
      const int gn = n0 + n, gkb = k0 + kb;
      rb[i] = (gn < g.N && gkb < kend) ? (g.transB ? B[(int64_t)gn * g.ldb + gkb] : B[(int64_t)gkb * g.ldb + gn]) : 0.f;
    }
  };
  auto lstore = [&]() {
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      const int id = tid + i * 256;
      int m, k;
      if (g.transA) { m = id & 63; k = id >> 6; } else { k = id & (BK - 1); m = id / BK; }
      sA[k][m] = ra[i];
      int n, kb;
      if (g.transB) { kb = id & (BK - 1); n = id / BK; } else { n = id & 63; kb = id >> 6; }
      sB[kb][n] = rb[i];
    }
  };
  if (kbeg < kend) gload(kbeg);
  for (int k0 = kbeg; k0 < kend; k0 += BK) {
    lstore();
    __syncthreads();
    if (k0 + BK < kend) gload(k0 + BK);
    if constexpr (MFMA) {
      // lane l: A[m = qm + (l & 31)][k + (l >> 5)], B[k + (l >> 5)][n = qn + (l & 31)]; rows / columns beyond M / N and k beyond K were
      // staged as zeros
#pragma unroll
      for (int k = 0; k < BK; k += 2)
        macc = __builtin_amdgcn_mfma_f32_32x32x2f32(sA[k + (lane >> 5)][qm + (lane & 31)], sB[k + (lane >> 5)][qn + (lane & 31)], macc, 0, 0, 0);
    } else {
#pragma unroll
      for (int k = 0; k < BK; ++k) {
        float a[4], b[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] = sA[k][ty * 4 + i];
#pragma unroll
        for (int j = 0; j < 4; ++j) b[j] = sB[k][tx * 4 + j];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(a[i], b[j], acc[i][j]);
      }
    }
    __syncthreads();
  }
  auto finish = [&](int gm, int gn, float sum) {
    float* c = C + (int64_t)gm * g.ldc + gn;
    if (g.split_k > 1) { atomicAdd(c, g.alpha * sum); return; }
    float v = g.alpha * sum;
    if (g.beta != 0.f) v += g.beta * (*c);
    if (g.bias) v += g.bias[gn];
    if (g.act == 1) v = fmaxf(v, 0.f);
    else if (g.act == 2) v = gelu_erf(v);
    else if (g.act == 3) v = 1.f / (1.f + expf(-v));
    *c = v;
  };
  if constexpr (MFMA) {
    // accumulator register r of lane l: row (r & 3) + 8 (r >> 2) + 4 (l >> 5), column l & 31 of the quadrant
    const int gn = n0 + qn + (lane & 31);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int gm = m0 + qm + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      if (gm < g.M && gn < g.N) finish(gm, gn, macc[r]);
    }
    return;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int gm = m0 + ty * 4 + i;
    if (gm >= g.M) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int gn = n0 + tx * 4 + j;
      if (gn >= g.N) continue;
      float* c = C + (int64_t)gm * g.ldc + gn;
      if (g.split_k > 1) {
        atomicAdd(c, g.alpha * acc[i][j]);
        continue;
      }
      float v = g.alpha * acc[i][j];
      if (g.beta != 0.f) v += g.beta * (*c);
      if (g.bias) v += g.bias[gn];
      if (g.act == 1) v = fmaxf(v, 0.f);
      else if (g.act == 2) v = gelu_erf(v);
      else if (g.act == 3) v = 1.f / (1.f + expf(-v));
      *c = v;
    }
  }
}

}  // namespace

extern "C" int mp_sgemm_f32(const float* A, int64_t lda, int transA, const float* B, int64_t ldb, int transB, float* C,
                            int64_t ldc, const float* bias, int M, int N, int K, float alpha, float beta, int act,
                            int nb0, int nb1, int64_t sA0, int64_t sA1, int64_t sB0, int64_t sB1, int64_t sC0, int64_t sC1,
                            int split_k, hipStream_t stream) {
  MP_REQUIRE(M >= 0 && N >= 0 && K >= 0 && nb0 >= 1 && nb1 >= 1, MP_ERR_SHAPE, "mp_sgemm_f32: bad shape");
  MP_REQUIRE(act >= 0 && act <= 3, MP_ERR_ARG, "mp_sgemm_f32: bad activation");
  MP_REQUIRE(split_k <= 1 || (beta == 1.f && bias == nullptr && act == 0), MP_ERR_ARG,
             "mp_sgemm_f32: split_k needs beta=1 (pre-initialised C), no bias, no activation");
  if (M == 0 || N == 0) return MP_OK;
  SgemmArgs g{A, B, C, bias, M, N, K, lda, ldb, ldc, transA, transB, nb0, nb1, sA0, sA1, sB0, sB1, sC0, sC1,
              alpha, beta, act, split_k < 1 ? 1 : split_k};
  dim3 grid((unsigned)mp_cdiv(M, 64), (unsigned)(mp_cdiv(N, 64) * g.split_k), (unsigned)(nb0 * nb1));
  static int bk = -1;
  if (bk < 0) { const char* e = getenv("MP_SGEMM_BK"); bk = (e && atoi(e) == 16) ? 16 : 64; }      // 16: the first version (A/B)
  const int k_unit = g.split_k > 1 ? (K + g.split_k - 1) / g.split_k : K;
  static int mfma = -1;
  if (mfma < 0) { const char* e = getenv("MP_SGEMM_MFMA"); mfma = (e && atoi(e) == 0) ? 0 : 1; }
  const bool deep = bk == 64 && k_unit >= 64;
  if (mfma) {
    if (deep) hipLaunchKernelGGL((sgemm_kernel<64, true>), grid, dim3(256), 0, stream, g);
    else hipLaunchKernelGGL((sgemm_kernel<16, true>), grid, dim3(256), 0, stream, g);
  } else {
    if (deep) hipLaunchKernelGGL((sgemm_kernel<64, false>), grid, dim3(256), 0, stream, g);
    else hipLaunchKernelGGL((sgemm_kernel<16, false>), grid, dim3(256), 0, stream, g);
  }
  return mp_check_launch("mp_sgemm_f32");
}
