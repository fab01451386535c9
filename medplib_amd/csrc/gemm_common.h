// Shared argument block and epilogue helpers of the bf16 MFMA GEMM kernels (gemm_bf16.hip: 128x128 tiles,
// gemm256_bf16.hip: 256x256 ping-pong tiles).
#pragma once
#include "common.h"

enum { ACT_NONE = 0, ACT_RELU = 1, ACT_GELU = 2, ACT_QUICK_GELU = 3, ACT_SILU = 4,
       // fused SwiGLU: W holds gate/up rows interleaved in blocks of 32 ([g0..31 | u0..31 | g32..63 | u32..63 | ...]); the epilogue
       // writes silu(gate) * up, so C is [M, N/2] (256x256 kernel only)
       ACT_SWIGLU_PAIR = 5,
       // fused RoPE on the q and k thirds of a fused qkv projection (256x256 kernel only): within every 128-wide head the W rows
       // are interleaved in blocks of 32 like the SwiGLU pairs ([lo 0..31 | hi 0..31 | lo 32..63 | hi 32..63], lo = dims 0..63,
       // hi = dims 64..127), so the two members of a rotary pair sit in fragments j and j+2 of the same lane; the rotated values
       // are stored at their STANDARD column positions, so C has the reference's layout.  The v third is a plain store.
       ACT_ROPE_QK = 6 };

struct GemmArgs {
  const bf16_t* A;  int64_t lda;
  const bf16_t* W;  int64_t ldw;
  void* C;          int64_t ldc;
  const float* bias;          // [N] or null
  const bf16_t* residual;     // [M,N] (ldr) or null, added after activation
  int64_t ldr;
  const int* m_dev;           // optional device-side row count (overrides M when non-null)
  int M, N, K;
  int act;
  int out_f32;
  float alpha;                // scale applied to the accumulator before bias
  // batching (blockIdx.y): element strides
  int64_t sA, sW, sC, sR, sBias;
  int m_dev_stride;
  int group_m;                // M-tiles per group in the L2-friendly tile order (1 = plain column-major order)
  // tail split-K of the 256x256 kernel (see gemm256_bf16.hip): fp32 partial workspace + per-tile arrival tickets, or null
  float* ws; int* tickets;
  int n_cu;                   // compute units the tile waves are counted against
  int nbatch;                 // batch count (the flat work decode walks all batches)
  int max_split;              // upper bound on the tail split factor (1 = off)
  long long tail_wait;        // 320-row kernel: shader cycles a split-tail unit waits for its siblings before the tile falls back to its last unit
  // MoE expert GEMMs fused with dispatch / combine (mp_gemm_bf16_nt_batched_rows): per batch b, A row r is read from row
  // a_rows[b * rows_stride + r] of the (shared) A matrix; C row r is written to row c_rows[b * rows_stride + r] of the (shared) C
  // matrix, scaled by c_scale[that row] before the residual (indexed by the same destination row) is added.  null = identity.
  const int* a_rows; const int* c_rows; const float* c_scale;
  int rows_stride;
  // ACT_ROPE_QK: cos / sin tables [positions, 64] fp32, position of row m = m % rope_seq + rope_pos0
  const float* rope_cos; const float* rope_sin;
  int rope_seq, rope_pos0;
  int ep8;              // 256x256 kernel: 1 = the 8-byte epilogue of round 1 (MP_GEMM_EP8=1, A/B runs only)
  // ACT_SWIGLU_PAIR with keep_gu: the rounded gate|up values are ALSO stored (interleaved [M, N] layout, ld_gu) -- training keeps them for
  // the backward, and the stand-alone SwiGLU pass over the [tokens, 2 ff] tensor disappears (mp_gemm_swiglu_keep_bf16)
  bf16_t* keep_gu; int64_t ld_gu;
  // folded input norms (320-row kernel, RoPE and SwiGLU families only): the accumulator of output row r is multiplied by a_scale[source row of
  // A row r] before the family's first rounding — rstd of an RMSNorm whose weight is folded into W's columns (mp_gemm_qkv_rope_scaled_bf16,
  // mp_gemm_bf16_nt_batched_rows_scaled).  null = 1.
  const float* a_scale;
};

// GELU(erf) without erff: gelu(x) = relu(x) - a Phi(-a), a = |x|, log2 Phi(-a) fitted by a degree-5 polynomial (minimax on the absolute
// error of a Phi(-a) over [0, 12]: 4.8e-7 in fp32, far below a bf16 ulp; the leading coefficient is negative, so the tail
// underflows to 0 for any |x|).  One v_exp_f32 and six FMAs; erff inlined 128 times per thread does not fit the register file.
static __device__ __forceinline__ float gelu_erf_fast(float x) {
  const float xp = fmaxf(x, 0.f);
  const float a = fabsf(x);
  float q = fmaf(-0.0004733088717330247f, a, 0.007084553129971027f);
  q = fmaf(q, a, -0.05182736739516258f);
  q = fmaf(q, a, -0.4599924683570862f);
  q = fmaf(q, a, -1.1507878303527832f);
  q = fmaf(q, a, -1.000037670135498f);
  return fmaf(-a, __builtin_amdgcn_exp2f(q), xp);
}

static __device__ __forceinline__ float apply_act(float v, int act) {
  switch (act) {
    case ACT_RELU: return fmaxf(v, 0.f);
    case ACT_GELU: return gelu_erf_fast(v);
    case ACT_QUICK_GELU: return v * mp_sigmoid_fast(v, 1.702f);
    case ACT_SILU: return v * mp_sigmoid_fast(v);
    default: return v;
  }
}


// implemented in gemm256_bf16.hip
int mp_launch_gemm256(const GemmArgs& g, int batch, hipStream_t stream);

// Two neighbouring fragments' bf16x4 row pieces (fragment j: columns j*16 + fq*4 .. +3, fragment j+1: 16 columns further) become one
// 16-byte piece per lane: v_permlane16_swap exchanges the odd 16-lane rows of the first operand with the even rows of the second (lane
// = fq*16 + fr, so a row of lanes IS an fq), after which lane (fr, fq) holds the EIGHT consecutive columns
// (fq & 1) * 16 + (fq >> 1) * 8 .. +7 of the pair's 32 -- same matrix row as before.  Stores (and residual loads) are then 16 bytes per
// lane and 64 contiguous bytes per matrix row per instruction instead of 8 / 32: half the store instructions (the epilogue is
// store-ISSUE bound, MI355X_MICROARCH.md T21).  Must be executed by all 64 lanes.
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2_t;
static __device__ __forceinline__ bf16x8 pair_swap16(bf16x4 a, bf16x4 b) {
  const u32x2_t ua = __builtin_bit_cast(u32x2_t, a), ub = __builtin_bit_cast(u32x2_t, b);
  const u32x2_t r0 = __builtin_amdgcn_permlane16_swap(ua[0], ub[0], false, false);
  const u32x2_t r1 = __builtin_amdgcn_permlane16_swap(ua[1], ub[1], false, false);
  typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
  return __builtin_bit_cast(bf16x8, (u32x4_t{r0[0], r1[0], r0[1], r1[1]}));
}
static __device__ __forceinline__ int pair_col8(int fq) { return (fq & 1) * 16 + (fq >> 1) * 8; }
static __device__ __forceinline__ bf16x4 round4(const f32x4& v) { return bf16x4{(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]}; }
static __device__ __forceinline__ bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

