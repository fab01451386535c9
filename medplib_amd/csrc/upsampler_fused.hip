// Fused SAM-Med2D mask-decoder upsampler (inference form), bf16 in / bf16 out, one pass over HBM:
//   ConvTranspose2d(256->64, k2 s2) -> LayerNorm2d(64, eps 1e-6) -> GELU -> ConvTranspose2d(64->32, k2 s2) -> GELU
//   [-> optional hypernetwork product  mask[b,y,x] = sum_c hyper[b,c] * up[b,c,y,x]]
// Reference: `output_upscaling` + `hyper_in @ upscaled_embedding` (model/segment_anything_med2d/modeling/mask_decoder.py:53-59,
// 141-148).  The reference runs 5 kernels that write and re-read the [B,64,2h,2w] and [B,32,4h,4w] intermediates; here every
// input token is read once and every output pixel written once (k = s = 2: no overlap between the 4x4 output patches of
// different tokens), so the algorithmic traffic is  in + weights + out  (3.29 MB at the 256-px geometry,
// 50.5 MB at the 1024-px geometry, batch 8, bf16).
//
// Both transposed convolutions are GEMMs over tokens (MFMA 16x16x32 bf16):
//   G1[token, sub*64 + c]        = X[token, :256] . W1[:, c, kh, kw]        sub = kh*2+kw        (K = 256, N = 256)
//   G2[(token,sub), sub2*32+c2]  = gelu(LN_c(G1[token, sub, :])) . W2[:, c2, kh2, kw2]            (K = 64,  N = 128)
// Persistent 1024-thread workgroups, one per CU, each specialised on one kh (row parity of the first ConvT): its half of the
// packed W1 (64 KiB) and W2 (16 KiB) stay in LDS.  The kh = 0 / kh = 1 partners of a token range sit on the same XCD so the
// second read of the tokens is an L2 hit.  Each of the 16 waves walks 16-token groups (16 consecutive tokens of one image row)
// on its own -- no workgroup barrier after the weight staging; four waves per SIMD let the MFMA phases of one wave run under
// the LayerNorm / GELU VALU phases of the others.  A wave computes both kw sub-pixels, so LayerNorm2d is a 16-lane DPP
// reduction and, after the second ConvT, lane (fr, fq) owns for channel c2 = fr (+16) the 16 consecutive output pixels of
// tokens 4fq..4fq+3 on output lines 2kh and 2kh+1: they leave as two 16-byte stores per line straight from registers.  The
// only LDS round trip is the per-wave C-layout -> A-layout transposition of the 32x64 intermediate (4 KiB per wave).
// LDS: 64 K (W1 half) + 16 K (W2) + 16 x 4 K = 144 KiB.
// The kernel is bounded by VALU issue, not HBM: 768 GELUs per token; GELU is evaluated as
//   gelu(x) = max(x,0) - |x| * Phi(-|x|),   Phi(-a) = 2^q5(a)   (degree-5 fit, |abs error| < 5e-7, all-packed v_pk_fma_f32)
#include <algorithm>
#include <stdlib.h>

#include "common.h"

namespace {

constexpr int UP_THREADS = 1024;
constexpr int UP_WAVES = UP_THREADS / 64;
constexpr int W1H_BYTES = 128 * 256 * 2;       // [n = kw*64 + c][k = 256] bf16, 512-B rows, 16-B chunks XOR-swizzled by (n & 7)
constexpr int W2_BYTES = 128 * 64 * 2;         // [n2 = sub2*32 + c2][k = 64] bf16, 128-B rows, chunks XOR-swizzled by (n2 & 7)
constexpr int T_WAVE_BYTES = 32 * 64 * 2;      // per-wave transposition buffer [kw*16 + token][64 ch] bf16 (swizzled)
constexpr int UP_LDS = W1H_BYTES + W2_BYTES + UP_WAVES * T_WAVE_BYTES;   // 147456
// Round 4, BOTH = true (the shipped forward): a workgroup serves BOTH row parities of its tokens — wave w takes (group w >> 1, kh = w & 1) — so
// a CU pulls each token once (64 KiB per CU and pass instead of 128: a CU ingests ~24 KB/us whatever the source, and the sixteenth wave's
// tokens used to land 5.5-8 us after the first's).  All of W1 (128 KiB) + W2 (16 KiB) stay in LDS; what is left is 1 KiB per wave, so the
// C-layout -> A-layout transposition of the 32 x 64 intermediate goes through it in four rounds of [16 tokens][32 channels].
constexpr int W1F_BYTES = 256 * 256 * 2;
constexpr int T1K_BYTES = 16 * 32 * 2;
constexpr int UP_LDS2 = W1F_BYTES + W2_BYTES + UP_WAVES * T1K_BYTES;    // 163840 = all of the CU's LDS

typedef __attribute__((ext_vector_type(2))) float f32x2;

struct UpArgs {
  const bf16_t* src;      // [B, h*w, 256]
  const bf16_t* w1p;      // [256][256]  packed: row n1 = sub*64 + c, col k = input channel
  const float* b1;        // [64]
  const float* lnw; const float* lnb;   // [64]
  const bf16_t* w2p;      // [128][64]   packed: row n2 = sub2*32 + c2, col k = channel of the first ConvT
  const float* b2;        // [32]
  const float* hyper;     // [B, 32] or null
  bf16_t* up;             // [B, 32, 4h, 4w] or null
  float* mask;            // [B, 4h, 4w] or null
  int B, h, w;
  float eps;
  int skew;               // MP_UPS_SKEW: waves sharing a SIMD start (wave >> 2) * skew * 64 clocks apart (0 = together)
  int early;              // BOTH: waves 0 .. early-1 of a workgroup request their first group's tokens before the staging wait (0 = all)
  long long* dbg;         // ABL & 4 (scripts/ups_lab.hip only): per (workgroup, wave) 16 s_memrealtime stamps (100 MHz) of the first pass
};

__device__ __forceinline__ f32x2 splat2(float v) { return f32x2{v, v}; }

// GELU(erf), two values per lane so every multiply-add is a v_pk_fma_f32.  gelu(x) = relu(x) - a Phi(-a), a = |x| = 2 relu(x) - x,
// Phi(-a) = 2^q(a).  One v_exp_f32 per value instead of exp + rcp.
template <int ABL>
__device__ __forceinline__ f32x2 gelu2(f32x2 x) {
  if constexpr (ABL & 1) return x;
  if constexpr ((ABL & 16777216) != 0) {     // lab: the same arithmetic as scalar instructions (build the lab with -fno-slp-vectorize)
    f32x2 r;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const float xi = x[i], xp = fmaxf(xi, 0.f), a = fmaf(2.f, xp, -xi);
      float q = fmaf(-0.0248856320977211f, a, -0.4988200068473816f);
      q = fmaf(q, a, -1.1292459964752197f);
      q = fmaf(q, a, -1.0035316944122314f);
      r[i] = fmaf(-a, __builtin_amdgcn_exp2f(q), xp);
    }
    return r;
  }
  // Round 4: log2 Phi(-a) by a DEGREE-3 polynomial (minimax on the absolute error of a Phi(-a) over [0, 12]: 5.5e-5 — a bf16 ulp of the
  // values this feeds is 4e-3 at 1 and 6e-5 at 0.01; both GELUs of this kernel are rounded to bf16 immediately; the leading coefficient
  // is negative, so the tail still underflows to 0).  Two packed FMAs fewer per pair than the degree-5 fit of the GEMM epilogues
  // (gemm_common.h keeps that one: 4.8e-7): the kernel is VALU-issue bound and a GELU pair was 11 of its instructions.
  const f32x2 xp = __builtin_elementwise_max(x, splat2(0.f));
  const f32x2 a = __builtin_elementwise_fma(splat2(2.f), xp, -x);
  f32x2 q = __builtin_elementwise_fma(splat2(-0.0248856320977211f), a, splat2(-0.4988200068473816f));
  q = __builtin_elementwise_fma(q, a, splat2(-1.1292459964752197f));
  q = __builtin_elementwise_fma(q, a, splat2(-1.0035316944122314f));
  const f32x2 e = {__builtin_amdgcn_exp2f(q.x), __builtin_amdgcn_exp2f(q.y)};
  return __builtin_elementwise_fma(-a, e, xp);
}

// all-reduce (sum) over the 16 lanes of a DPP row for four independent values: 16 v_add_f32_dpp, no LDS traffic.  The four
// chains are interleaved so every DPP source was written >= 3 instructions earlier (the VALU-write -> DPP-read hazard needs
// two wait states, which the leading s_nop covers for the first step).
__device__ __forceinline__ void row16_sum4(float& v0, float& v1, float& v2, float& v3) {
#define MP_DPP4(ctrl)                                                                 \
  "v_add_f32_dpp %0, %0, %0 " ctrl " row_mask:0xf bank_mask:0xf bound_ctrl:1\n"       \
  "v_add_f32_dpp %1, %1, %1 " ctrl " row_mask:0xf bank_mask:0xf bound_ctrl:1\n"       \
  "v_add_f32_dpp %2, %2, %2 " ctrl " row_mask:0xf bank_mask:0xf bound_ctrl:1\n"       \
  "v_add_f32_dpp %3, %3, %3 " ctrl " row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
  asm("s_nop 1\n" MP_DPP4("quad_perm:[1,0,3,2]") MP_DPP4("quad_perm:[2,3,0,1]") MP_DPP4("row_half_mirror") MP_DPP4("row_mirror")
      : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));
#undef MP_DPP4
}

// Round 4: the key is (n & 15), not (n & 7).  A ds_read_b128 is served in groups of 16 lanes that here hold 16 consecutive rows n (512-byte
// rows: every row starts on bank 0) and two neighbouring chunk indices; with a 3-bit key their 16 x 16 bytes fell into ONE aligned
// 128-byte window = half of the 64 banks: a guaranteed 2-way conflict on every W1 fragment read (GEMM1 re-reads the whole 64 KiB half
// per 16-token group: 1 MiB per CU and pass at 128 instead of 256 B/clk — the timeline in profiles/r04_upsampler_timeline.txt shows the
// GEMM1 phase of a wave taking 2.3-4.3 us).  With four key bits the group covers all 16 chunk positions of the 256-byte bank span.
__device__ __forceinline__ int w1_off(int n, int c) { return n * 512 + ((c ^ (n & 15)) << 4); }      // c = 16-B chunk 0..31

template <bool WITH_UP, bool WITH_MASK, int ABL = 0, bool BOTH = false>
__global__ __launch_bounds__(UP_THREADS, 1) void upsample_fused_kernel(UpArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sW1 = smem;
  char* sW2 = smem + (BOTH ? W1F_BYTES : W1H_BYTES);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fq = lane >> 4;
  bf16_t* tw = reinterpret_cast<bf16_t*>(BOTH ? smem + W1F_BYTES + W2_BYTES + wave * T1K_BYTES : smem + W1H_BYTES + W2_BYTES + wave * T_WAVE_BYTES);
  // T1: GEMM1 issued transposed too — see the LayerNorm section (lab bit 1048576 = the earlier form: GEMM1 as tokens x channels, the
  // intermediate transposed through LDS)
  constexpr bool T1 = BOTH && (ABL & (1048576 | 131072 | 262144)) == 0;
  float* sCst = reinterpret_cast<float*>(smem + W1F_BYTES + W2_BYTES);      // T1: b1 | ln weight | ln bias, 64 floats each (the transposition buffers are unused)
  long long ts[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define UPS_STAMP(i) do { if constexpr ((ABL & 4) != 0) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); ts[i] = __builtin_amdgcn_s_memrealtime(); } } while (0)
  UPS_STAMP(0);

  // workgroup -> (kh, group set): partners share blockIdx % 8, i.e. the XCD and its L2
  int kh, gset;
  int n_gsets = gridDim.x >> 1;
  if constexpr (BOTH) { kh = wave & 1; gset = blockIdx.x; n_gsets = gridDim.x; }
  else if ((gridDim.x & 15) == 0) { const int loc = blockIdx.x >> 3; kh = loc & 1; gset = (loc >> 1) * 8 + (blockIdx.x & 7); }
  else { kh = blockIdx.x & 1; gset = blockIdx.x >> 1; }

  const int tokens_per_img = a.h * a.w;
  const int64_t n_tokens = (int64_t)a.B * tokens_per_img;
  const int64_t n_groups = n_tokens / 16;                  // w % 16 == 0: groups are whole
  const int nw = blockDim.x >> 6;                          // waves in this launch (1..16: small problems spread over more CUs)
  const int64_t stride = (int64_t)n_gsets * (BOTH ? (nw >> 1) : nw);
  // Round 4: a CU's token reads from HBM run at ~10 B/clk (a wave's 8 KiB land ~0.3 us after the previous wave's: the timeline shows the
  // sixteen waves' tokens arriving between 3.3 and 7.2 us), and the kh = 0 / kh = 1 partners of a group set want the SAME tokens at the
  // same time — both wait for HBM.  The kh = 1 workgroup therefore takes its groups half a turn ahead: in the first half of the pass the
  // partners fetch different tokens from HBM, in the second half each reads what the other has already brought into their XCD's L2.
  int64_t grp = BOTH ? (int64_t)gset * (nw >> 1) + (wave >> 1) : (int64_t)gset * nw + ((wave + (kh ? (nw >> 1) : 0)) % nw);

  // ---- stage W1 (this kh's half: 64 instructions of 1 KiB = 2 rows each; BOTH: all 128) and W2 (16 x 1 KiB = 8 rows each)
  //      by LDS-DMA; the LDS image is lane-linear, so the XOR swizzle sits on the source address
  if constexpr (BOTH && (ABL & 262144) != 0) {
    // Lab only (bit 262144; measured SLOWER, 22.6 vs 17.1 us: the DMA's scattered source — 16 rows x 64 bytes per instruction — makes the
    // staging 2-3 x longer, and GEMM1 does not speed up, i.e. its fragment reads were not bank-conflicted after the 4-bit key):
    // FRAGMENT-MAJOR weights: one 1 KiB LDS block per MFMA operand fragment — block (n-block, kk) holds, lane-linear, exactly the
    // 16 bytes lane (fr, fq) feeds to the MFMA (row n-block*16 + fr, K elements kk*32 + fq*8 ..+7) — so a fragment read is ONE ds_read_b128
    // of 1 KiB of CONTIGUOUS LDS (every 16-lane service group reads 256 consecutive bytes: all 64 banks once, whatever the grouping is),
    // with the same base register for all blocks (the block is an immediate offset).  The row-major image with an XOR swizzle (rounds 1-3,
    // and the 4-bit key tried first this round) measured ~40 B/clk on these reads: the timeline showed the four waves of a SIMD passing
    // GEMM1 strictly one after another, 2.5-3 us each — a saturated LDS, 6 x below its rate.  The scatter moves to the DMA's SOURCE side
    // (16 rows x 64 bytes per instruction from L2), which does not care.
    for (int j = wave; j < 128; j += nw)                      // block j = n-block (j >> 3) x kk (j & 7)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a.w1p + (int64_t)((j >> 3) * 16 + fr) * 256 + ((j & 7) * 4 + fq) * 8),
                                       (__attribute__((address_space(3))) void*)(sW1 + j * 1024), 16, 0, 0);
    for (int j = wave; j < 16; j += nw)                       // block j = n2-block (j >> 1) x kk (j & 1)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a.w2p + (int64_t)((j >> 1) * 16 + fr) * 64 + ((j & 1) * 4 + fq) * 8),
                                       (__attribute__((address_space(3))) void*)(sW2 + j * 1024), 16, 0, 0);
  } else {
  for (int j = wave; j < (BOTH ? 128 : 64); j += nw) {
    const int n = 2 * j + (lane >> 5), pc = lane & 31;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a.w1p + (int64_t)((BOTH ? 0 : kh * 128) + n) * 256 + ((pc ^ (n & 15)) << 3)),
                                     (__attribute__((address_space(3))) void*)(sW1 + j * 1024), 16, 0, 0);
  }
  if constexpr (T1) {
    // W2's K axis permuted inside every 32-channel step: a lane of the transposed GEMM1 result owns channels 16jj + 4fq + r of the step
    // (jj = 0, 1; r = 0..3), so element e of its GEMM2 operand is channel 32kk + 16(e >> 2) + 4fq + (e & 3), and the W2 fragment must
    // carry the same channel in the same element.  4-byte DMA pieces (two channels) place them: 64 instructions of 256 bytes = 2 rows.
    for (int j = wave; j < 64; j += nw) {
      const int n = 2 * j + (lane >> 5), c = ((lane & 31) >> 2) ^ (n & 7), d = lane & 3;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a.w2p + (int64_t)n * 64 + (c >> 2) * 32 + (d >> 1) * 16 + (c & 3) * 4 + (d & 1) * 2),
                                       (__attribute__((address_space(3))) void*)(sW2 + j * 256), 4, 0, 0);
    }
    if (wave == 0) { sCst[lane] = a.b1[lane]; sCst[64 + lane] = a.lnw[lane]; sCst[128 + lane] = a.lnb[lane]; }
  } else
  for (int j = wave; j < 16; j += nw) {
    const int n = 8 * j + (lane >> 3), pc = lane & 7;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a.w2p + (int64_t)n * 64 + ((pc ^ (n & 7)) << 3)),
                                     (__attribute__((address_space(3))) void*)(sW2 + j * 1024), 16, 0, 0);
  }
  }
  // per-lane constants (channel = j*16 + fr)
  float b1v[4], lwv[4], lbv[4], b2v[2];
#pragma unroll
  for (int j = 0; j < 4; ++j) { b1v[j] = a.b1[j * 16 + fr]; lwv[j] = a.lnw[j * 16 + fr]; lbv[j] = a.lnb[j * 16 + fr]; }
#pragma unroll
  for (int p = 0; p < 2; ++p) b2v[p] = a.b2[p * 16 + fr];
  const int tokens_per_img0 = a.h * a.w;
  (void)tokens_per_img0;
  bf16x8 xa0[8];
  bool have_tokens = false;
  auto load_tokens = [&](int64_t g, bf16x8 (&xa)[8]) {
    const int64_t t0 = g * 16;
    if constexpr ((ABL & 512) != 0) {       // lab (WRONG RESULTS): the kh = 1 partner reads other tokens — do the duplicate reads cost time?
      const bf16_t* xrow = a.src + ((t0 + (kh ? n_tokens / 2 : 0)) % n_tokens + fr) * 256;
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) xa[kk] = *reinterpret_cast<const bf16x8*>(xrow + (kk * 4 + fq) * 8);
    } else if constexpr ((ABL & 1024) != 0) {   // lab (WRONG RESULTS): each load instruction covers 1 KiB of contiguous memory
      const bf16_t* xrow = a.src + t0 * 256;
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) xa[kk] = *reinterpret_cast<const bf16x8*>(xrow + kk * 512 + lane * 8);
    } else if constexpr ((ABL & 8388608) != 0) {   // lab: non-temporal token loads
      const bf16_t* xrow = a.src + (t0 + fr) * 256;
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) xa[kk] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(xrow + (kk * 4 + fq) * 8));
    } else {
      const bf16_t* xrow = a.src + (t0 + fr) * 256;
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) xa[kk] = *reinterpret_cast<const bf16x8*>(xrow + (kk * 4 + fq) * 8);
    }
  };
  if constexpr (BOTH && (ABL & 2048) == 0) {
    // Round 4: the HBM sat idle for the ~2.5 us the weights take to arrive (they come from L2 for all but the first workgroup of an XCD),
    // and the tokens' 16.8 MB then needed 5-6 us.  Now: every wave has ISSUED its share of the weight DMA (a barrier that waits for no
    // data), then issues the token loads of its first group — behind all weight requests in the CU's queue, not between them (round 2
    // tried "tokens before the staging wait" without that barrier: the last waves' weights queued behind the first waves' tokens and the
    // staging barrier waited for both: 10 % slower) — then waits for the weights only (a counted vmcnt: the eight youngest requests are
    // the tokens; memory returns in order) and meets the others.  The tokens keep streaming in across that second barrier.
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (grp < n_groups && (a.early <= 0 || wave < a.early)) { load_tokens(grp, xa0); have_tokens = true; }
    __builtin_amdgcn_sched_barrier(0);
    if (have_tokens) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    UPS_STAMP(1);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("" ::: "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    UPS_STAMP(1);
    __syncthreads();
  }
  UPS_STAMP(2);
  // At the benchmark geometry every wave makes exactly ONE pass, so without this all sixteen waves of a CU walk the phases (token read,
  // GEMM1, LayerNorm + GELU, transposition, GEMM2, GELU + stores) in lockstep and the four waves of a SIMD want the same unit at the same
  // time.  Starting the k-th wave of each SIMD k * skew * 64 clocks late puts them in different phases: the matrix pipe of one runs under
  // the VALU phase of another.
  if (a.skew > 0) {
    const int k = wave >> 2;
    for (int i = 0; i < k; ++i) __builtin_amdgcn_s_sleep(1);
    for (int i = 0; i < k * (a.skew - 1); ++i) __builtin_amdgcn_s_sleep(1);
  }

  if constexpr (BOTH && (ABL & 4194304) != 0) {     // lab: the k-th wave of a SIMD runs at priority 3 - k (oldest task first: stores start earlier)
    switch (wave >> 2) {
      case 0: __builtin_amdgcn_s_setprio(3); break;
      case 1: __builtin_amdgcn_s_setprio(2); break;
      case 2: __builtin_amdgcn_s_setprio(1); break;
      default: break;
    }
  }
  const int OW = 4 * a.w, OH = 4 * a.h;
  const unsigned up_lane_off = 2u * ((unsigned)(4 * fq) * (unsigned)(OH * OW) + 4u * (unsigned)fr);   // BYTES (a 32-bit offset beside a scalar base: < 2^32 for maps up to 8192 x 8192 pixels)
  // one (group, kh) task; the first task of a wave gets its tokens from the prologue (xa0), later ones load them here
  auto task_body = [&](const int64_t grp, const bf16x8 (&xa)[8]) __attribute__((always_inline)) {
    const int64_t t0 = grp * 16;
    // (wave-uniform by construction; said explicitly so that everything derived from them — the output bases above all — lives in scalar registers)
    const int b = __builtin_amdgcn_readfirstlane((int)(t0 / tokens_per_img));
    const int ti = __builtin_amdgcn_readfirstlane((int)(t0 % tokens_per_img));
    const int irow = __builtin_amdgcn_readfirstlane(ti / a.w), j0 = __builtin_amdgcn_readfirstlane(ti % a.w);
    // four waves per SIMD hide this latency; a register prefetch would cost 32 VGPRs across the whole body.  Measured: issuing the
    // first group's loads before the weight-staging wait (one pass per wave at the 1024-px geometry) is 10 % SLOWER (24.7 vs
    // 22.5 us, same box) -- 16 waves x 8 KiB of token reads queue in front of the 80 KiB of weight DMA every wave waits for.
    float h0 = 0.f, h1 = 0.f;
    if (WITH_MASK) { h0 = a.hyper[(int64_t)b * 32 + fr]; h1 = a.hyper[(int64_t)b * 32 + 16 + fr]; }
    if constexpr ((ABL & 4) != 0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); UPS_STAMP(3); }
    // ---------------- GEMM1: 16 tokens x (2 kw x 64 channels) ----------------
    f32x4 acc1[2][4];                       // initialised with the bias of the lane's channel j*16 + fr (round 4: one packed add per value pair less)
#pragma unroll
    for (int kw = 0; kw < 2; ++kw)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if constexpr (T1) acc1[kw][j] = *reinterpret_cast<const f32x4*>(sCst + j * 16 + 4 * fq);
        else acc1[kw][j] = f32x4{b1v[j], b1v[j], b1v[j], b1v[j]};
      }
    // GEMM1 of a wave is a chain of LDS round trips (~190 clocks each under load) with 16 clocks of matrix work per fragment, and the four
    // waves of a SIMD pass it one after another (timeline: 2.5-3 us per wave with the compiler's own order, which keeps TWO reads in flight:
    // ds_read x2, wait, MFMA, wait, MFMA).
    if constexpr (BOTH && (T1 || (WITH_UP && !WITH_MASK)) && (ABL & 8192) == 0) {
      // four fragment reads in flight, then their four MFMAs, as a scheduling hint (sched_group_barrier) rather than a fence: with
      // sched_barrier(0) fences the allocator spilled 9-25 registers and the kernel got slower; this form compiles to 121 VGPRs without
      // scratch.  With GEMM1 transposed (T1) every instantiation takes this form: left to its own order the compiler hoists the fragment
      // reads of the whole K loop in the mask forms (268 registers spilled).
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) {
#pragma unroll
        for (int kw = 0; kw < 2; ++kw) {
          bf16x8 wb[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) wb[j] = *reinterpret_cast<const bf16x8*>(sW1 + w1_off(kh * 128 + kw * 64 + j * 16 + fr, kk * 4 + fq));
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc1[kw][j] = T1 ? __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[j], xa[kk], acc1[kw][j], 0, 0, 0)
                             : __builtin_amdgcn_mfma_f32_16x16x32_bf16(xa[kk], wb[j], acc1[kw][j], 0, 0, 0);
        }
#pragma unroll
        for (int g2 = 0; g2 < 2; ++g2) {
          __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
        }
      }
    } else {
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) {
#pragma unroll
        for (int kw = 0; kw < 2; ++kw)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const bf16x8 wb = (BOTH && (ABL & 262144) != 0)
                ? *reinterpret_cast<const bf16x8*>(sW1 + (((kh * 8 + kw * 4 + j) * 8 + kk) * 1024) + lane * 16)
                : *reinterpret_cast<const bf16x8*>(sW1 + w1_off((BOTH ? kh * 128 : 0) + kw * 64 + j * 16 + fr, kk * 4 + fq));
            acc1[kw][j] = T1 ? __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb, xa[kk], acc1[kw][j], 0, 0, 0)
                             : __builtin_amdgcn_mfma_f32_16x16x32_bf16(xa[kk], wb, acc1[kw][j], 0, 0, 0);
          }
        if constexpr ((ABL & 524288) != 0) { if (kk == 0) { asm volatile("" :: "v"(acc1[1][3])); UPS_STAMP(1); } if (kk == 3) { asm volatile("" :: "v"(acc1[1][3])); UPS_STAMP(2); } }
      }
    }
    // ---------------- + bias, LayerNorm2d over the 64 channels, GELU (C layout: channel = j*16 + fr, token = fq*4 + r) ----
    if constexpr ((ABL & 4) != 0) { asm volatile("" :: "v"(acc1[1][3])); UPS_STAMP(4); }
    bf16x8 ya[2][2];                          // GEMM2's A fragments: the intermediate back in [row = token][8 consecutive channels]
    if constexpr (T1) {
      // GEMM1 transposed: the lane owns ONE token (column fr) and channels j*16 + 4fq + r.  The LayerNorm sums are 16 in-lane values plus two
      // cross-row steps (the four fq rows of a token), and the normalised values already sit where GEMM2's operand wants them (W2's K axis
      // is permuted to match at staging): no transposition through LDS, no 64 DPP adds.
      f32x4 lwt[4], lbt[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) { lwt[j] = *reinterpret_cast<const f32x4*>(sCst + 64 + j * 16 + 4 * fq); lbt[j] = *reinterpret_cast<const f32x4*>(sCst + 128 + j * 16 + 4 * fq); }
      f32x2 v[2][4][2];
      float s[2];
#pragma unroll
      for (int kw = 0; kw < 2; ++kw) {
        f32x2 s2 = splat2(0.f);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          v[kw][j][0] = f32x2{acc1[kw][j][0], acc1[kw][j][1]}; v[kw][j][1] = f32x2{acc1[kw][j][2], acc1[kw][j][3]};
          s2 += v[kw][j][0]; s2 += v[kw][j][1];
        }
        s[kw] = s2.x + s2.y;
      }
#pragma unroll
      for (int kw = 0; kw < 2; ++kw) s[kw] += __shfl_xor(s[kw], 16);
#pragma unroll
      for (int kw = 0; kw < 2; ++kw) s[kw] += __shfl_xor(s[kw], 32);
      float q[2];
#pragma unroll
      for (int kw = 0; kw < 2; ++kw) {
        const f32x2 mean = splat2(s[kw] * (1.f / 64.f));
        f32x2 q2 = splat2(0.f);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int p = 0; p < 2; ++p) { v[kw][j][p] -= mean; q2 = __builtin_elementwise_fma(v[kw][j][p], v[kw][j][p], q2); }
        q[kw] = q2.x + q2.y;
      }
#pragma unroll
      for (int kw = 0; kw < 2; ++kw) q[kw] += __shfl_xor(q[kw], 16);
#pragma unroll
      for (int kw = 0; kw < 2; ++kw) q[kw] += __shfl_xor(q[kw], 32);
#pragma unroll
      for (int kw = 0; kw < 2; ++kw) {
        const f32x2 rstd = splat2(__builtin_amdgcn_rsqf(fmaf(q[kw], 1.f / 64.f, a.eps)));
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int p = 0; p < 2; ++p) {
            const f32x2 y = gelu2<ABL>(__builtin_elementwise_fma(v[kw][j][p] * rstd, f32x2{lwt[j][2 * p], lwt[j][2 * p + 1]}, f32x2{lbt[j][2 * p], lbt[j][2 * p + 1]}));
            ya[kw][j >> 1][(j & 1) * 4 + 2 * p] = (bf16_t)y[0];
            ya[kw][j >> 1][(j & 1) * 4 + 2 * p + 1] = (bf16_t)y[1];
          }
      }
    } else
#pragma unroll
    for (int kw = 0; kw < 2; ++kw) {
      f32x2 v[2][4];                          // [token pair rp][j]: tokens r = 2rp, 2rp+1 -> packed arithmetic
      f32x2 s[2] = {splat2(0.f), splat2(0.f)};
#pragma unroll
      for (int rp = 0; rp < 2; ++rp)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          v[rp][j] = f32x2{acc1[kw][j][2 * rp], acc1[kw][j][2 * rp + 1]};
          s[rp] += v[rp][j];
        }
      {
        float s0 = s[0].x, s1 = s[0].y, s2 = s[1].x, s3 = s[1].y;
        row16_sum4(s0, s1, s2, s3);
        s[0] = f32x2{s0, s1}; s[1] = f32x2{s2, s3};
      }
      f32x2 q[2] = {splat2(0.f), splat2(0.f)};
#pragma unroll
      for (int rp = 0; rp < 2; ++rp) {
        const f32x2 mean = s[rp] * splat2(1.f / 64.f);
#pragma unroll
        for (int j = 0; j < 4; ++j) { v[rp][j] -= mean; q[rp] = __builtin_elementwise_fma(v[rp][j], v[rp][j], q[rp]); }
      }
      {
        float q0 = q[0].x, q1 = q[0].y, q2 = q[1].x, q3 = q[1].y;
        row16_sum4(q0, q1, q2, q3);
        q[0] = f32x2{q0, q1}; q[1] = f32x2{q2, q3};
      }
      bf16_t yb[2][4][2];                    // BOTH: this kw's 16 values per lane, rounded, until their round through the 1 KiB buffer
#pragma unroll
      for (int rp = 0; rp < 2; ++rp) {
        const f32x2 var = __builtin_elementwise_fma(q[rp], splat2(1.f / 64.f), splat2(a.eps));
        const f32x2 rstd = {__builtin_amdgcn_rsqf(var.x), __builtin_amdgcn_rsqf(var.y)};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const f32x2 y = gelu2<ABL>(__builtin_elementwise_fma(v[rp][j] * rstd, splat2(lwv[j]), splat2(lbv[j])));
          if constexpr (BOTH) { yb[rp][j][0] = (bf16_t)y[0]; yb[rp][j][1] = (bf16_t)y[1]; }
          else {
            const int ch = j * 16 + fr;       // transposition buffer [kw*16 + token][64 ch], 16-B chunks XOR-swizzled by (token & 7)
#pragma unroll
            for (int e = 0; e < 2; ++e) {
              const int tk = fq * 4 + 2 * rp + e;
              tw[(kw * 16 + tk) * 64 + ((((ch >> 3) ^ (tk & 7)) << 3) | (ch & 7))] = (bf16_t)y[e];
            }
          }
        }
      }
      if constexpr (BOTH) {
        // two rounds per kw through [16 tokens][32 channels] (64-byte rows): channels 32kk .. 32kk+31 = this lane's j = 2kk, 2kk+1.  A wave's
        // LDS operations execute in order, so the rounds need no wait between them — only the compiler must keep the order.  The 16-byte
        // chunk c of row t sits at c ^ key(t >> 2), key = (0, 3, 2, 1): a ds_read_b128 service group holds rows of all four t >> 2 classes
        // with two neighbouring c, and this key sends them to sixteen different (row mod 4, chunk) slots of the 256-byte bank span.
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
#pragma unroll
          for (int jj = 0; jj < 2; ++jj)
#pragma unroll
            for (int rp = 0; rp < 2; ++rp)
#pragma unroll
              for (int e = 0; e < 2; ++e) {
                const int cl = jj * 16 + fr;
                tw[(fq * 4 + 2 * rp + e) * 32 + ((((cl >> 3) ^ ((4 - fq) & 3)) << 3) | (cl & 7))] = yb[rp][2 * kk + jj][e];
              }
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          ya[kw][kk] = *reinterpret_cast<const bf16x8*>(tw + fr * 32 + ((fq ^ ((4 - (fr >> 2)) & 3)) << 3));
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
          __builtin_amdgcn_wave_barrier();
        }
      }
    }
    if constexpr (!BOTH) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int kw = 0; kw < 2; ++kw)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
          ya[kw][kk] = *reinterpret_cast<const bf16x8*>(tw + (kw * 16 + fr) * 64 + (((kk * 4 + fq) ^ (fr & 7)) << 3));
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      __builtin_amdgcn_wave_barrier();          // the transposition buffer may be overwritten by the next group
    }
    UPS_STAMP(5);
    // ---------------- GEMM2 per output line 2kh + kh2: [2 kw x 16 tokens] x [kw2, half] (64 of the 128 columns), K = 64 ----------------
    // Round 4 (BOTH): GEMM2 is issued TRANSPOSED — MFMA(W2 fragment, intermediate) instead of MFMA(intermediate, W2 fragment); the operand
    // registers are the same — so a lane owns ONE token (column fr) and the four channels 4fq + r of every 16-channel block.  The 16 lanes of a
    // row then hold the 16 consecutive tokens of one (channel, output line): their 4 pixels x 2 bytes are one whole 128-byte line, written
    // by one 8-byte store per lane — whole-line stores instead of 16-byte pieces scattered over 16 channel planes (the lab measured the same
    // bytes as whole lines 1.6 us faster per launch), and the hypernetwork product needs two cross-row adds instead of 64 DPP adds.
    if constexpr (BOTH && (ABL & 131072) == 0) {
      const f32x4 b2t[2] = {*reinterpret_cast<const f32x4*>(a.b2 + 4 * fq), *reinterpret_cast<const f32x4*>(a.b2 + 16 + 4 * fq)};
      f32x4 ht[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
      if (WITH_MASK) { ht[0] = *reinterpret_cast<const f32x4*>(a.hyper + (int64_t)b * 32 + 4 * fq); ht[1] = *reinterpret_cast<const f32x4*>(a.hyper + (int64_t)b * 32 + 16 + 4 * fq); }
      // (the two output lines as a real loop in the up-only form: unrolled, the allocator overlaps their accumulators and spills 17 registers;
      //  the mask forms spill least when unrolled — measured per instantiation with -Rpass-analysis=kernel-resource-usage)
      constexpr int KH2_UNROLL = (WITH_UP && !WITH_MASK) ? 1 : 2;
#pragma unroll KH2_UNROLL
      for (int kh2 = 0; kh2 < 2; ++kh2) {
        // acc2[kw][q], q = kw2*2 + half: row n2 = (kh2*4 + q)*16 + 4fq + r -> channel c2 = half*16 + 4fq + r; column = token fr
        f32x4 acc2[2][4];
#pragma unroll
        for (int kw = 0; kw < 2; ++kw)
#pragma unroll
          for (int qn = 0; qn < 4; ++qn) acc2[kw][qn] = b2t[qn & 1];
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
          for (int qn = 0; qn < 4; ++qn) {
            const int n2 = (kh2 * 4 + qn) * 16 + fr;
            const bf16x8 wb = (ABL & 262144) != 0 ? *reinterpret_cast<const bf16x8*>(sW2 + (((kh2 * 4 + qn) * 2 + kk) * 1024) + lane * 16)
                                                  : *reinterpret_cast<const bf16x8*>(sW2 + n2 * 128 + (((kk * 4 + fq) ^ (fr & 7)) << 4));
#pragma unroll
            for (int kw = 0; kw < 2; ++kw) acc2[kw][qn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb, ya[kw][kk], acc2[kw][qn], 0, 0, 0);
          }
        const int64_t line = (int64_t)4 * irow + 2 * kh + kh2;
        float m4[4] = {0.f, 0.f, 0.f, 0.f};               // this token's four pixels (kw*2 + kw2) of the line, summed over the lane's channels
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          f32x2 g[2][2][2];                               // [kw][kw2][channel pair]
#pragma unroll
          for (int kw = 0; kw < 2; ++kw)
#pragma unroll
            for (int kw2 = 0; kw2 < 2; ++kw2) {
              const f32x4 c = acc2[kw][kw2 * 2 + half];
              g[kw][kw2][0] = gelu2<ABL>(f32x2{c[0], c[1]});
              g[kw][kw2][1] = gelu2<ABL>(f32x2{c[2], c[3]});
            }
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const bf16x4 o = {(bf16_t)g[0][0][r >> 1][r & 1], (bf16_t)g[0][1][r >> 1][r & 1], (bf16_t)g[1][0][r >> 1][r & 1], (bf16_t)g[1][1][r >> 1][r & 1]};
            if (WITH_UP) {
              // wave-uniform base (scalar registers) + one per-lane offset for all sixteen stores of a task: the lane's own four channels
              // 4fq + r sit 4fq planes above the base, its token 4 fr pixels along the line
              bf16_t* dst = (ABL & 33554432) != 0      // lab: the earlier form, the whole address per lane and store
                  ? a.up + (((int64_t)b * 32 + half * 16 + 4 * fq + r) * OH + line) * OW + 4 * (j0 + fr)
                  : reinterpret_cast<bf16_t*>(reinterpret_cast<char*>(a.up + (((int64_t)b * 32 + half * 16 + r) * OH + line) * OW + 4 * j0) + up_lane_off);
              if constexpr ((ABL & 2) != 0) { asm volatile("" :: "v"(o)); }
              else if constexpr ((ABL & 4096) != 0) *reinterpret_cast<bf16x4*>(dst) = o;
              else __builtin_nontemporal_store(o, reinterpret_cast<bf16x4*>(dst));     // streaming: see the note at the 16-byte form below
            }
            if (WITH_MASK) {      // the reference multiplies the bf16-rounded upscaled embedding: keep that rounding point
#pragma unroll
              for (int i = 0; i < 4; ++i) m4[i] = fmaf(ht[half][r], (float)o[i], m4[i]);
            }
          }
        }
        if constexpr ((ABL & 4) != 0) { if (kh2 == 0) UPS_STAMP(6); }
        if (WITH_MASK) {
#pragma unroll
          for (int i = 0; i < 4; ++i) { m4[i] += __shfl_xor(m4[i], 16); m4[i] += __shfl_xor(m4[i], 32); }
          if (fq == 0) *reinterpret_cast<float4*>(a.mask + ((int64_t)b * OH + line) * OW + 4 * (j0 + fr)) = float4{m4[0], m4[1], m4[2], m4[3]};
        }
      }
    } else
#pragma unroll
    for (int kh2 = 0; kh2 < 2; ++kh2) {
      // acc2[kw][q], q = kw2*2 + half: n2 = (kh2*4 + q)*16 + fr -> c2 = half*16 + fr; acc2[..][r] is token fq*4 + r
      f32x4 acc2[2][4];
#pragma unroll
      for (int kw = 0; kw < 2; ++kw)
#pragma unroll
        for (int qn = 0; qn < 4; ++qn) acc2[kw][qn] = f32x4{b2v[qn & 1], b2v[qn & 1], b2v[qn & 1], b2v[qn & 1]};   // bias of channel (qn & 1) * 16 + fr
      if constexpr (BOTH && (ABL & 8192) != 0) {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
          bf16x8 wb2[4];
#pragma unroll
          for (int qn = 0; qn < 4; ++qn)
            wb2[qn] = *reinterpret_cast<const bf16x8*>(sW2 + ((kh2 * 4 + qn) * 16 + fr) * 128 + (((kk * 4 + fq) ^ (fr & 7)) << 4));
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int qn = 0; qn < 4; ++qn)
#pragma unroll
            for (int kw = 0; kw < 2; ++kw) acc2[kw][qn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ya[kw][kk], wb2[qn], acc2[kw][qn], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
      } else {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
          for (int qn = 0; qn < 4; ++qn) {
            const int n2 = (kh2 * 4 + qn) * 16 + fr;
            const bf16x8 wb = *reinterpret_cast<const bf16x8*>(sW2 + n2 * 128 + (((kk * 4 + fq) ^ (fr & 7)) << 4));
#pragma unroll
            for (int kw = 0; kw < 2; ++kw) acc2[kw][qn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ya[kw][kk], wb, acc2[kw][qn], 0, 0, 0);
          }
      }
      // ---------------- + bias, GELU, store: lane owns pixels 16fq .. 16fq+15 (= r*4 + kw*2 + kw2) of this line ----------------
      float msum[16];
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        bf16_t px[16];
#pragma unroll
        for (int kw = 0; kw < 2; ++kw)
#pragma unroll
          for (int kw2 = 0; kw2 < 2; ++kw2) {
            const f32x4 c = acc2[kw][kw2 * 2 + half];
            const f32x2 g01 = gelu2<ABL>(f32x2{c[0], c[1]});
            const f32x2 g23 = gelu2<ABL>(f32x2{c[2], c[3]});
            px[0 * 4 + kw * 2 + kw2] = (bf16_t)g01.x;
            px[1 * 4 + kw * 2 + kw2] = (bf16_t)g01.y;
            px[2 * 4 + kw * 2 + kw2] = (bf16_t)g23.x;
            px[3 * 4 + kw * 2 + kw2] = (bf16_t)g23.y;
          }
        if (WITH_UP) {
          bf16_t* dst = a.up + (((int64_t)b * 32 + half * 16 + fr) * OH + 4 * irow + 2 * kh + kh2) * OW + 4 * j0 + 16 * fq;
          bf16x8 lo, hi;
#pragma unroll
          for (int i = 0; i < 8; ++i) { lo[i] = px[i]; hi[i] = px[8 + i]; }
          if constexpr ((ABL & 2) != 0) { asm volatile("" :: "v"(lo), "v"(hi)); }
          else if constexpr ((ABL & 8) != 0) {            // lab: non-temporal stores
            __builtin_nontemporal_store(lo, reinterpret_cast<bf16x8*>(dst));
            __builtin_nontemporal_store(hi, reinterpret_cast<bf16x8*>(dst + 8));
          } else if constexpr ((ABL & 64) != 0) {         // lab: write-through (sc1) stores
            typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
            asm volatile("global_store_dwordx4 %0, %1, off sc1\n\tglobal_store_dwordx4 %0, %2, off offset:16 sc1"
                         :: "v"(dst), "v"(__builtin_bit_cast(u32x4, lo)), "v"(__builtin_bit_cast(u32x4, hi)) : "memory");
          } else if constexpr ((ABL & 128) != 0) {        // lab: sc0 sc1 stores
            typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
            asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\tglobal_store_dwordx4 %0, %2, off offset:16 sc0 sc1"
                         :: "v"(dst), "v"(__builtin_bit_cast(u32x4, lo)), "v"(__builtin_bit_cast(u32x4, hi)) : "memory");
          } else if constexpr ((ABL & 256) != 0) {        // lab: sc1 nt stores
            typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
            asm volatile("global_store_dwordx4 %0, %1, off sc1 nt\n\tglobal_store_dwordx4 %0, %2, off offset:16 sc1 nt"
                         :: "v"(dst), "v"(__builtin_bit_cast(u32x4, lo)), "v"(__builtin_bit_cast(u32x4, hi)) : "memory");
          } else if constexpr ((ABL & 32) != 0) {         // lab (WRONG RESULTS): the same bytes as whole 128-byte lines, 1 KiB contiguous per instruction
            bf16_t* d2 = a.up + ((int64_t)(grp * 2 + kh) * 4096 + (kh2 * 2 + half) * 1024 + lane * 8);
            typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
            if constexpr ((ABL & 16384) != 0)
              asm volatile("global_store_dwordx4 %0, %1, off sc1\n\tglobal_store_dwordx4 %0, %2, off offset:1024 sc1"
                           :: "v"(d2), "v"(__builtin_bit_cast(u32x4, lo)), "v"(__builtin_bit_cast(u32x4, hi)) : "memory");
            else if constexpr ((ABL & 32768) != 0)
              asm volatile("global_store_dwordx4 %0, %1, off sc1 nt\n\tglobal_store_dwordx4 %0, %2, off offset:1024 sc1 nt"
                           :: "v"(d2), "v"(__builtin_bit_cast(u32x4, lo)), "v"(__builtin_bit_cast(u32x4, hi)) : "memory");
            else if constexpr ((ABL & 65536) != 0)
              asm volatile("global_store_dwordx4 %0, %1, off nt\n\tglobal_store_dwordx4 %0, %2, off offset:1024 nt"
                           :: "v"(d2), "v"(__builtin_bit_cast(u32x4, lo)), "v"(__builtin_bit_cast(u32x4, hi)) : "memory");
            else {
              *reinterpret_cast<bf16x8*>(d2) = lo;
              *reinterpret_cast<bf16x8*>(d2 + 512) = hi;
            }
          } else if constexpr (BOTH && (ABL & 4096) == 0) {
            // non-temporal: the 33.5 MB of output are never read by this kernel and must all reach HBM before the next kernel of the
            // stream starts (the release at a kernel's end writes the L2's dirty lines back: ~5 us for what 32 MiB of L2 hold); lines marked
            // streaming leave the L2 earlier, i.e. during the kernel (measured 21.9 -> 20.9 us per launch, back to back)
            __builtin_nontemporal_store(lo, reinterpret_cast<bf16x8*>(dst));
            __builtin_nontemporal_store(hi, reinterpret_cast<bf16x8*>(dst + 8));
          } else {
            *reinterpret_cast<bf16x8*>(dst) = lo;
            *reinterpret_cast<bf16x8*>(dst + 8) = hi;
          }
        }
        if (WITH_MASK) {      // the reference multiplies the bf16-rounded upscaled embedding: keep that rounding point
          const float hv = half ? h1 : h0;
#pragma unroll
          for (int i = 0; i < 16; ++i) msum[i] = half ? fmaf(hv, (float)px[i], msum[i]) : hv * (float)px[i];
        }
      }
      if constexpr ((ABL & 4) != 0) { if (kh2 == 0) UPS_STAMP(6); }
      if (WITH_MASK) {
#pragma unroll
        for (int i = 0; i < 16; i += 4) row16_sum4(msum[i], msum[i + 1], msum[i + 2], msum[i + 3]);
        if (fr == 0) {
          float* dst = a.mask + ((int64_t)b * OH + 4 * irow + 2 * kh + kh2) * OW + 4 * j0 + 16 * fq;
#pragma unroll
          for (int i = 0; i < 16; i += 4) *reinterpret_cast<float4*>(dst + i) = float4{msum[i], msum[i + 1], msum[i + 2], msum[i + 3]};
        }
      }
    }
    if constexpr ((ABL & 4) != 0) {
      UPS_STAMP(7);
      if (lane == 0 && a.dbg) {
        long long* d = a.dbg + ((int64_t)blockIdx.x * UP_WAVES + wave) * 8;
#pragma unroll
        for (int i = 0; i < 8; ++i) d[i] = ts[i];
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the stores have left: the true end of this wave
      if (lane == 0 && a.dbg) a.dbg[((int64_t)gridDim.x * UP_WAVES + (int64_t)blockIdx.x * UP_WAVES + wave) * 8] = __builtin_amdgcn_s_memrealtime();
    }
  };
  if (have_tokens) { task_body(grp, xa0); grp += stride; }
  for (; grp < n_groups; grp += stride) {
    bf16x8 xa[8];
    load_tokens(grp, xa);
    task_body(grp, xa);
  }
#undef UPS_STAMP
}

// ------------------------------------------------------------------ backward ---------------------------------------------------
// Training form (round 3): the forward above with the hypernetwork product is differentiated by ONE kernel that recomputes the
// forward of its 16-token group (same instruction sequence, same bf16 rounding points: no intermediate was kept) and walks the chain
//   dmask -> d gelu2 -> [dW2, db2 operands] -> da1 = dy2 . W2 -> d gelu1 -> d LayerNorm2d -> [dW1, db1 operands] -> dx = dy1 . W1
// with the two data-gradient products on MFMA (bf16 operands, fp32 accumulation) and GELU' / LayerNorm' in registers.  The weight
// gradients are `tn` products over ALL tokens; the kernel writes their operands (dy1, a1, dy2: 8 MB at the model's 16 x 16 geometry,
// batch 8) and the host runs them as two deterministic GEMMs; bias / LayerNorm / hypernetwork gradients leave as one partial row per
// (group, kh) task that a fixed-order column sum finishes.  Nothing is accumulated with atomics: the step stays bit-reproducible.
// One wave per (group, kh) task as in the forward, at most four waves per workgroup (one per SIMD: ~300 live VGPRs).
constexpr int UPB_WAVES = 4;
constexpr int TB_WAVE_BYTES = T_WAVE_BYTES;        // per-wave LDS, 4 KiB, used in turn for a1 [32][64], dy2 [32][64] per output line and dy1 [16][128] (bf16)
constexpr int UPB_LDS = W1H_BYTES + W2_BYTES + UPB_WAVES * TB_WAVE_BYTES;

struct UpBwdArgs {
  UpArgs f;               // the forward's operands (up / mask unused)
  const bf16_t* w1t;      // [256 cin][256 n1]  = w1p transposed
  const bf16_t* w2t;      // [64 ch][128 n2]    = w2p transposed
  const float* dmask;     // [B, 4h, 4w]
  float* dx;              // [2 kh][B*h*w][256]
  float* dy1;             // [B*h*w][256]            column n1 = (kh*2 + kw)*64 + ch
  float* a1;              // [tasks*32][64]          task = group*2 + kh, row = kw*16 + token
  float* dy2;             // [tasks*32][128]         column n2 = (kh2*2 + kw2)*32 + c2
  float* part;            // [tasks][256]            db1[64] | dlnw[64] | dlnb[64] | db2[32] | dhyper[32]
};

// d/dx of x Phi(x) = Phi(x) + x phi(x), with Phi(-a) = 2^q5(a) from the forward's own fit (|error of Phi| < 2e-5: far below the bf16
// rounding of the operands the gradient is multiplied into) and phi by one v_exp_f32
__device__ __forceinline__ float gelu_grad(float x) {
  const float a = fabsf(x);
  float q = fmaf(-0.0004733088717330247f, a, 0.007084553129971027f);
  q = fmaf(q, a, -0.05182736739516258f);
  q = fmaf(q, a, -0.4599924683570862f);
  q = fmaf(q, a, -1.1507878303527832f);
  q = fmaf(q, a, -1.000037670135498f);
  const float tail = __builtin_amdgcn_exp2f(q);                       // Phi(-|x|)
  const float cdf = x >= 0.f ? 1.f - tail : tail;
  const float pdf = 0.3989422804014327f * __builtin_amdgcn_exp2f(-0.7213475204444817f * x * x);
  return fmaf(x, pdf, cdf);
}

__device__ __forceinline__ float sum_over_fq(float v) {        // the four 16-lane rows of the wave hold partial sums of the same channel
  v += __shfl_xor(v, 16);
  v += __shfl_xor(v, 32);
  return v;
}

__global__ __launch_bounds__(64 * UPB_WAVES, 1) void upsample_fused_bwd_kernel(UpBwdArgs g) {
  const UpArgs& a = g.f;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sW1 = smem;
  char* sW2 = smem + W1H_BYTES;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fq = lane >> 4;
  bf16_t* tw = reinterpret_cast<bf16_t*>(smem + W1H_BYTES + W2_BYTES + wave * TB_WAVE_BYTES);

  int kh, gset;
  const int n_gsets = gridDim.x >> 1;
  if ((gridDim.x & 15) == 0) { const int loc = blockIdx.x >> 3; kh = loc & 1; gset = (loc >> 1) * 8 + (blockIdx.x & 7); }
  else { kh = blockIdx.x & 1; gset = blockIdx.x >> 1; }
  const int tokens_per_img = a.h * a.w;
  const int64_t n_tokens = (int64_t)a.B * tokens_per_img;
  const int64_t n_groups = n_tokens / 16;
  const int nw = blockDim.x >> 6;
  const int64_t stride = (int64_t)n_gsets * nw;
  int64_t grp = (int64_t)gset * nw + wave;

  for (int j = wave; j < 64; j += nw) {
    const int n = 2 * j + (lane >> 5), pc = lane & 31;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a.w1p + (int64_t)(kh * 128 + n) * 256 + ((pc ^ (n & 15)) << 3)),
                                     (__attribute__((address_space(3))) void*)(sW1 + j * 1024), 16, 0, 0);
  }
  for (int j = wave; j < 16; j += nw) {
    const int n = 8 * j + (lane >> 3), pc = lane & 7;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a.w2p + (int64_t)n * 64 + ((pc ^ (n & 7)) << 3)),
                                     (__attribute__((address_space(3))) void*)(sW2 + j * 1024), 16, 0, 0);
  }
  float b1v[4], lwv[4], lbv[4], b2v[2];
#pragma unroll
  for (int j = 0; j < 4; ++j) { b1v[j] = a.b1[j * 16 + fr]; lwv[j] = a.lnw[j * 16 + fr]; lbv[j] = a.lnb[j * 16 + fr]; }
#pragma unroll
  for (int p = 0; p < 2; ++p) b2v[p] = a.b2[p * 16 + fr];
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  const int OW = 4 * a.w, OH = 4 * a.h;
  for (; grp < n_groups; grp += stride) {
    const int64_t t0 = grp * 16;
    const int64_t task = grp * 2 + kh;
    const int b = (int)(t0 / tokens_per_img);
    const int ti = (int)(t0 % tokens_per_img);
    const int irow = ti / a.w, j0 = ti % a.w;
    bf16x8 xa[8];
    {
      const bf16_t* xrow = a.src + (t0 + fr) * 256;
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) xa[kk] = *reinterpret_cast<const bf16x8*>(xrow + (kk * 4 + fq) * 8);
    }
    const float hv[2] = {a.hyper[(int64_t)b * 32 + fr], a.hyper[(int64_t)b * 32 + 16 + fr]};
    // ---------------- recompute: GEMM1 -> + bias, LayerNorm2d, GELU (the forward's own sequence) ----------------
    f32x4 acc1[2][4];                       // initialised with the bias of the lane's channel j*16 + fr (round 4: one packed add per value pair less)
#pragma unroll
    for (int kw = 0; kw < 2; ++kw)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc1[kw][j] = f32x4{b1v[j], b1v[j], b1v[j], b1v[j]};
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
#pragma unroll
      for (int kw = 0; kw < 2; ++kw)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const bf16x8 wb = *reinterpret_cast<const bf16x8*>(sW1 + w1_off(kw * 64 + j * 16 + fr, kk * 4 + fq));
          acc1[kw][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xa[kk], wb, acc1[kw][j], 0, 0, 0);
        }
    }
    f32x2 zh[2][2][4], g1[2][2][4], rs[2][2];           // normalised value, gelu'(z), 1 / sigma   per [kw][token pair][j]
    float* a1row = g.a1 + task * (32 * 64);
#pragma unroll
    for (int kw = 0; kw < 2; ++kw) {
      f32x2 v[2][4];
      f32x2 s[2] = {splat2(0.f), splat2(0.f)};
#pragma unroll
      for (int rp = 0; rp < 2; ++rp)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          v[rp][j] = f32x2{acc1[kw][j][2 * rp], acc1[kw][j][2 * rp + 1]};
          s[rp] += v[rp][j];
        }
      {
        float s0 = s[0].x, s1 = s[0].y, s2 = s[1].x, s3 = s[1].y;
        row16_sum4(s0, s1, s2, s3);
        s[0] = f32x2{s0, s1}; s[1] = f32x2{s2, s3};
      }
      f32x2 q[2] = {splat2(0.f), splat2(0.f)};
#pragma unroll
      for (int rp = 0; rp < 2; ++rp) {
        const f32x2 mean = s[rp] * splat2(1.f / 64.f);
#pragma unroll
        for (int j = 0; j < 4; ++j) { v[rp][j] -= mean; q[rp] = __builtin_elementwise_fma(v[rp][j], v[rp][j], q[rp]); }
      }
      {
        float q0 = q[0].x, q1 = q[0].y, q2 = q[1].x, q3 = q[1].y;
        row16_sum4(q0, q1, q2, q3);
        q[0] = f32x2{q0, q1}; q[1] = f32x2{q2, q3};
      }
#pragma unroll
      for (int rp = 0; rp < 2; ++rp) {
        const f32x2 var = __builtin_elementwise_fma(q[rp], splat2(1.f / 64.f), splat2(a.eps));
        const f32x2 rstd = {__builtin_amdgcn_rsqf(var.x), __builtin_amdgcn_rsqf(var.y)};
        rs[kw][rp] = rstd;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          zh[kw][rp][j] = v[rp][j] * rstd;
          const f32x2 z = __builtin_elementwise_fma(zh[kw][rp][j], splat2(lwv[j]), splat2(lbv[j]));
          const f32x2 y = gelu2<0>(z);
          g1[kw][rp][j] = f32x2{gelu_grad(z.x), gelu_grad(z.y)};
          const int ch = j * 16 + fr;
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            const int tk = fq * 4 + 2 * rp + e;
            const bf16_t yb = (bf16_t)y[e];
            tw[(kw * 16 + tk) * 64 + ((((ch >> 3) ^ (tk & 7)) << 3) | (ch & 7))] = yb;
            a1row[(kw * 16 + tk) * 64 + ch] = (float)yb;       // dW2's operand: what GEMM2 multiplied
          }
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    bf16x8 ya[2][2];
#pragma unroll
    for (int kw = 0; kw < 2; ++kw)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
        ya[kw][kk] = *reinterpret_cast<const bf16x8*>(tw + (kw * 16 + fr) * 64 + (((kk * 4 + fq) ^ (fr & 7)) << 3));
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // ---------------- per output line: recompute GEMM2, d gelu2, and da1 += dy2 . W2 ----------------
    f32x4 da1[2][4];
#pragma unroll
    for (int kw = 0; kw < 2; ++kw)
#pragma unroll
      for (int j = 0; j < 4; ++j) da1[kw][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    float dhp[2] = {0.f, 0.f}, db2p[2] = {0.f, 0.f};
    float* dy2row = g.dy2 + task * (32 * 128);
#pragma unroll 1
    for (int kh2 = 0; kh2 < 2; ++kh2) {
      f32x4 acc2[2][4];
#pragma unroll
      for (int kw = 0; kw < 2; ++kw)
#pragma unroll
        for (int qn = 0; qn < 4; ++qn) acc2[kw][qn] = f32x4{b2v[qn & 1], b2v[qn & 1], b2v[qn & 1], b2v[qn & 1]};   // bias of channel (qn & 1) * 16 + fr
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int qn = 0; qn < 4; ++qn) {
          const int n2 = (kh2 * 4 + qn) * 16 + fr;
          const bf16x8 wb = *reinterpret_cast<const bf16x8*>(sW2 + n2 * 128 + (((kk * 4 + fq) ^ (fr & 7)) << 4));
#pragma unroll
          for (int kw = 0; kw < 2; ++kw) acc2[kw][qn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ya[kw][kk], wb, acc2[kw][qn], 0, 0, 0);
        }
      float dm[16];                          // the 16 pixels of this line the lane row fq owns: pixel = r*4 + kw*2 + kw2
      {
        const float* dmp = g.dmask + ((int64_t)b * OH + 4 * irow + 2 * kh + kh2) * OW + 4 * j0 + 16 * fq;
#pragma unroll
        for (int i = 0; i < 16; i += 4) {
          const float4 t = *reinterpret_cast<const float4*>(dmp + i);
          dm[i] = t.x; dm[i + 1] = t.y; dm[i + 2] = t.z; dm[i + 3] = t.w;
        }
      }
#pragma unroll
      for (int kw = 0; kw < 2; ++kw)
#pragma unroll
        for (int qn = 0; qn < 4; ++qn) {
          const int kw2 = qn >> 1, half = qn & 1;
          const f32x4 c = acc2[kw][qn];
          const f32x2 g01 = gelu2<0>(f32x2{c[0], c[1]});
          const f32x2 g23 = gelu2<0>(f32x2{c[2], c[3]});
          const float a2[4] = {(float)(bf16_t)g01.x, (float)(bf16_t)g01.y, (float)(bf16_t)g23.x, (float)(bf16_t)g23.y};
          const int n2l = qn * 16 + fr;       // column within this line's 64
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float dmv = dm[r * 4 + kw * 2 + kw2];
            dhp[half] = fmaf(dmv, a2[r], dhp[half]);
            const float d = dmv * hv[half] * gelu_grad(c[r]);
            db2p[half] += d;
            const int tk = fq * 4 + r;
            dy2row[(kw * 16 + tk) * 128 + kh2 * 64 + n2l] = d;
            tw[(kw * 16 + tk) * 64 + ((((n2l >> 3) ^ (tk & 7)) << 3) | (n2l & 7))] = (bf16_t)d;
          }
        }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      bf16x8 dya[2][2];
#pragma unroll
      for (int kw = 0; kw < 2; ++kw)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
          dya[kw][kk] = *reinterpret_cast<const bf16x8*>(tw + (kw * 16 + fr) * 64 + (((kk * 4 + fq) ^ (fr & 7)) << 3));
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      // da1[(kw, token)][ch] += sum_n2 dy2[(kw, token)][n2] * W2p[n2][ch]: B operand = W2^T rows (16 KiB, L1 / L2 resident)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const bf16x8 wb = *reinterpret_cast<const bf16x8*>(g.w2t + (j * 16 + fr) * 128 + kh2 * 64 + kk * 32 + fq * 8);
#pragma unroll
          for (int kw = 0; kw < 2; ++kw) da1[kw][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dya[kw][kk], wb, da1[kw][j], 0, 0, 0);
        }
    }
    // ---------------- d gelu1, d LayerNorm2d -> dy1 ----------------
    float db1p[4] = {0.f, 0.f, 0.f, 0.f}, dlwp[4] = {0.f, 0.f, 0.f, 0.f}, dlbp[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kw = 0; kw < 2; ++kw) {
      f32x2 dzh[2][4];
      f32x2 s1[2] = {splat2(0.f), splat2(0.f)}, s2[2] = {splat2(0.f), splat2(0.f)};
#pragma unroll
      for (int rp = 0; rp < 2; ++rp)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const f32x2 dz = f32x2{da1[kw][j][2 * rp], da1[kw][j][2 * rp + 1]} * g1[kw][rp][j];
          const f32x2 t = dz * zh[kw][rp][j];
          dlwp[j] += t.x + t.y;
          dlbp[j] += dz.x + dz.y;
          dzh[rp][j] = dz * splat2(lwv[j]);
          s1[rp] += dzh[rp][j];
          s2[rp] = __builtin_elementwise_fma(dzh[rp][j], zh[kw][rp][j], s2[rp]);
        }
      {
        float s0 = s1[0].x, sa = s1[0].y, sb = s1[1].x, sc = s1[1].y;
        row16_sum4(s0, sa, sb, sc);
        s1[0] = f32x2{s0, sa}; s1[1] = f32x2{sb, sc};
        float u0 = s2[0].x, ua = s2[0].y, ub = s2[1].x, uc = s2[1].y;
        row16_sum4(u0, ua, ub, uc);
        s2[0] = f32x2{u0, ua}; s2[1] = f32x2{ub, uc};
      }
#pragma unroll
      for (int rp = 0; rp < 2; ++rp) {
        const f32x2 m1 = s1[rp] * splat2(1.f / 64.f), m2 = s2[rp] * splat2(1.f / 64.f);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const f32x2 d = rs[kw][rp] * (dzh[rp][j] - m1 - zh[kw][rp][j] * m2);
          db1p[j] += d.x + d.y;
          const int ch = j * 16 + fr, n1l = kw * 64 + ch;
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            const int tk = fq * 4 + 2 * rp + e;
            g.dy1[(t0 + tk) * 256 + kh * 128 + n1l] = d[e];
            tw[tk * 128 + ((((n1l >> 3) ^ (tk & 7)) << 3) | (n1l & 7))] = (bf16_t)d[e];
          }
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    bf16x8 d1a[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) d1a[kk] = *reinterpret_cast<const bf16x8*>(tw + fr * 128 + (((kk * 4 + fq) ^ (fr & 7)) << 3));
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // ---------------- dx[token][cin] (this kh's share) = sum_n1 dy1[token][n1] * W1p[n1][cin]: B operand = W1^T rows (L2) ----------------
    float* dxrow = g.dx + ((int64_t)kh * n_tokens + t0) * 256;
#pragma unroll 2
    for (int nf = 0; nf < 16; ++nf) {
      f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        const bf16x8 wb = *reinterpret_cast<const bf16x8*>(g.w1t + (int64_t)(nf * 16 + fr) * 256 + kh * 128 + kk * 32 + fq * 8);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(d1a[kk], wb, acc, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) dxrow[(fq * 4 + r) * 256 + nf * 16 + fr] = acc[r];
    }
    // ---------------- this task's partial row: db1 | dlnw | dlnb | db2 | dhyper ----------------
    float* prow = g.part + task * 256;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float x1 = sum_over_fq(db1p[j]), x2 = sum_over_fq(dlwp[j]), x3 = sum_over_fq(dlbp[j]);
      if (fq == 0) { prow[j * 16 + fr] = x1; prow[64 + j * 16 + fr] = x2; prow[128 + j * 16 + fr] = x3; }
    }
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const float x1 = sum_over_fq(db2p[half]), x2 = sum_over_fq(dhp[half]);
      if (fq == 0) { prow[192 + half * 16 + fr] = x1; prow[224 + half * 16 + fr] = x2; }
    }
  }
}

}  // namespace

extern "C" int mp_mask_upsample_fused_bf16(const void* src, const void* w1_packed, const float* b1, const float* ln_w,
                                           const float* ln_b, const void* w2_packed, const float* b2, const float* hyper, void* up,
                                           float* mask, int B, int h, int w, float ln_eps, hipStream_t stream) {
  MP_REQUIRE(B > 0 && h > 0 && w > 0 && w % 16 == 0, MP_ERR_SHAPE, "mp_mask_upsample_fused_bf16: token-grid width must be a multiple of 16");
  MP_REQUIRE(up != nullptr || mask != nullptr, MP_ERR_ARG, "mp_mask_upsample_fused_bf16: nothing to produce");
  MP_REQUIRE(mask == nullptr || hyper != nullptr, MP_ERR_ARG, "mp_mask_upsample_fused_bf16: mask output needs hyper");
  static int skew = -1;
  if (skew < 0) { const char* e = getenv("MP_UPS_SKEW"); skew = e ? atoi(e) : 0; }
  // which waves request their first group's tokens before the staging wait: the first FOUR of a 16-wave workgroup (one per SIMD).  The four
  // waves of a SIMD pass their tasks one after another (the timeline of section 3.2), so only the first of them needs its tokens at the
  // barrier; the other twelve loads of a CU used to share the pipe with the weight DMA and held the staging barrier back from ~2.9 to ~5.3 us
  // (17.2 -> 16.3 us per launch; 8 early waves 16.4, 2: 16.5, all 16: 17.2).  Small launches (2-4 waves per workgroup): all of them.
  static int early_env = -2;
  if (early_env == -2) { const char* e = getenv("MP_UPS_EARLY"); early_env = e ? atoi(e) : -1; }
  UpArgs a{(const bf16_t*)src, (const bf16_t*)w1_packed, b1, ln_w, ln_b, (const bf16_t*)w2_packed, b2, hyper, (bf16_t*)up, mask,
           B, h, w, ln_eps, skew, 0, nullptr};
  const int64_t groups = mp_cdiv((int64_t)B * h * w, 16);
  // Round 4: one (group, row parity) task per wave and pass, BOTH parities of a group in the same workgroup (neighbouring waves), up to 256
  // workgroups: 2 waves per workgroup for the model's own 16 x 16 token maps (128 groups at batch 8 -> 128 workgroups), 16 waves from 2048
  // groups on.  MP_UPS_NW overrides the minimum (sweep), MP_UPS_SPLIT=1 selects the round-1..3 arrangement (one parity per workgroup; A/B).
  static int nw_env = -1, split_env = -1;
  if (nw_env < 0) { const char* e = getenv("MP_UPS_NW"); nw_env = e ? atoi(e) : 2; }
  if (split_env < 0) { const char* e = getenv("MP_UPS_SPLIT"); split_env = (e && atoi(e) == 1) ? 1 : 0; }
  static int abl = -1;
  if (abl < 0) { const char* e = getenv("MP_UPS_ABLATE"); abl = e ? atoi(e) : 0; }       // scripts/upsampler_bench.py only
  if (split_env) {
    int nw = (int)std::min<int64_t>(UP_WAVES, mp_cdiv(groups, 128));
    if (nw_env > 0 && groups >= 2 && groups <= 128 * UP_WAVES) nw = std::min<int>(UP_WAVES, std::max(nw, nw_env));
    const int64_t gsets = mp_cdiv(groups, nw);
    const int grid = 2 * (int)(gsets < 128 ? gsets : 128);
    auto launch = [&](auto kern) {
      (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, UP_LDS);
      hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * nw), UP_LDS, stream, a);
    };
    if (up && mask) launch(upsample_fused_kernel<true, true>);
    else if (up) launch(upsample_fused_kernel<true, false>);
    else launch(upsample_fused_kernel<false, true>);
  } else {
    int nw = (int)std::min<int64_t>(UP_WAVES, 2 * mp_cdiv(groups, 256));
    nw = std::min<int>(UP_WAVES, std::max(nw, 2 * std::max(1, nw_env / 2)));
    nw = (int)std::min<int64_t>(nw, 2 * groups);
    const int grid = (int)std::min<int64_t>(256, mp_cdiv(groups, nw / 2));
    a.early = early_env >= 0 ? early_env : (nw >= 8 ? 4 : 0);
    auto launch = [&](auto kern) {
      // function attributes are per DEVICE: one flag per (instantiation of this lambda's operator(), device) — a process that drives a second
      // GPU sets it there too (round-4 advisor)
      static bool attr_set[64] = {};
      int dev = 0;
      (void)hipGetDevice(&dev);
      if (dev < 0 || dev >= 64 || !attr_set[dev]) {
        (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, UP_LDS2);
        if (dev >= 0 && dev < 64) attr_set[dev] = true;
      }
      hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * nw), UP_LDS2, stream, a);
    };
    if (up && !mask && abl == 1) launch(upsample_fused_kernel<true, false, 1, true>);
    else if (up && !mask && abl == 2) launch(upsample_fused_kernel<true, false, 2, true>);
    else if (up && !mask && abl == 3) launch(upsample_fused_kernel<true, false, 3, true>);
    else if (up && mask) launch(upsample_fused_kernel<true, true, 0, true>);
    else if (up) launch(upsample_fused_kernel<true, false, 0, true>);
    else launch(upsample_fused_kernel<false, true, 0, true>);
  }
  return mp_check_launch("mp_mask_upsample_fused_bf16");
}

extern "C" int mp_mask_upsample_fused_bwd_bf16(const void* src, const void* w1_packed, const float* b1, const float* ln_w, const float* ln_b,
                                               const void* w2_packed, const float* b2, const float* hyper, const void* w1_t, const void* w2_t,
                                               const float* dmask, float* dx2, float* dy1, float* a1, float* dy2, float* part,
                                               int B, int h, int w, float ln_eps, hipStream_t stream) {
  MP_REQUIRE(B > 0 && h > 0 && w > 0 && w % 16 == 0, MP_ERR_SHAPE, "mp_mask_upsample_fused_bwd_bf16: token-grid width must be a multiple of 16");
  MP_REQUIRE(src && w1_packed && w2_packed && w1_t && w2_t && hyper && dmask && dx2 && dy1 && a1 && dy2 && part, MP_ERR_ARG,
             "mp_mask_upsample_fused_bwd_bf16: null operand");
  UpBwdArgs g{UpArgs{(const bf16_t*)src, (const bf16_t*)w1_packed, b1, ln_w, ln_b, (const bf16_t*)w2_packed, b2, hyper, nullptr, nullptr,
                     B, h, w, ln_eps, 0, 0, nullptr},
              (const bf16_t*)w1_t, (const bf16_t*)w2_t, dmask, dx2, dy1, a1, dy2, part};
  const int64_t groups = (int64_t)B * h * w / 16;
  const int nw = (int)std::min<int64_t>(UPB_WAVES, std::max<int64_t>(2, mp_cdiv(groups, 128)));
  const int64_t gsets = mp_cdiv(groups, nw);
  const int grid = 2 * (int)(gsets < 128 ? gsets : 128);
  (void)hipFuncSetAttribute((const void*)upsample_fused_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, UPB_LDS);
  hipLaunchKernelGGL(upsample_fused_bwd_kernel, dim3(grid), dim3(64 * nw), UPB_LDS, stream, g);
  return mp_check_launch("mp_mask_upsample_fused_bwd_bf16");
}

// The four bf16 operand images of the two ConvTranspose2d weights in ONE launch (the training step re-packs them every step: the weights train).
// w [Cin, Cout, 2, 2] fp32 (the reference layout, mask_decoder.py:53-59) -> packed [(kh, kw, cout), cin] and its transpose [cin, (kh, kw, cout)].
namespace {
__global__ void upsampler_pack_kernel(const float* __restrict__ w1, const float* __restrict__ w2, bf16_t* __restrict__ w1p, bf16_t* __restrict__ w2p,
                                      bf16_t* __restrict__ w1t, bf16_t* __restrict__ w2t, int ci1, int co1, int ci2, int co2) {
  const int64_t n1 = (int64_t)ci1 * co1 * 4, n2 = (int64_t)ci2 * co2 * 4;
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const float* w; bf16_t* wp; bf16_t* wt; int ci, co;
  if (i < n1) { w = w1; wp = w1p; wt = w1t; ci = ci1; co = co1; }
  else if (i < n1 + n2) { i -= n1; w = w2; wp = w2p; wt = w2t; ci = ci2; co = co2; }
  else return;
  // i enumerates the source [ci][co][kh*2+kw]
  const int q = (int)(i & 3), o = (int)((i >> 2) % co), c = (int)((i >> 2) / co);
  const bf16_t v = (bf16_t)w[i];
  const int j = q * co + o;                               // packed row (kh, kw, cout)
  wp[(int64_t)j * ci + c] = v;
  if (wt) wt[(int64_t)c * (4 * co) + j] = v;
}
}  // namespace

extern "C" int mp_upsampler_pack_bf16(const float* w1, const float* w2, void* w1p, void* w2p, void* w1t, void* w2t, int ci1, int co1, int ci2,
                                      int co2, hipStream_t stream) {
  MP_REQUIRE(w1 && w2 && w1p && w2p && ci1 > 0 && co1 > 0 && ci2 > 0 && co2 > 0, MP_ERR_ARG, "mp_upsampler_pack_bf16: null / bad shape");
  const int64_t n = (int64_t)ci1 * co1 * 4 + (int64_t)ci2 * co2 * 4;
  hipLaunchKernelGGL(upsampler_pack_kernel, dim3((unsigned)mp_cdiv(n, 256)), dim3(256), 0, stream, w1, w2, (bf16_t*)w1p, (bf16_t*)w2p, (bf16_t*)w1t,
                     (bf16_t*)w2t, ci1, co1, ci2, co2);
  return mp_check_launch("mp_upsampler_pack_bf16");
}
