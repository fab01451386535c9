// Flash-style fused attention forward (bf16 in/out, fp32 softmax statistics) for gfx950.
//
// One kernel covers the three trunk attentions of the reference path:
//   * Llama causal self-attention with key padding (HF-4.31 eager semantics: masked scores get zero weight, softmax in
//     fp32; padded QUERY rows still produce values)                       — medplib_moe_llama.py:127-135, SURVEY A.1
//   * CLIP ViT-L self-attention, no mask, D = 64                              — clip_encoder.py:41-60, SURVEY A.2
//   * SAM-Med2D ViT-B window/global attention with decomposed relative-position bias
//     (scores = (q*scale) k^T + rel_h[q, k_row] + rel_w[q, k_col])            — image_encoder.py:280-296, 381-421
//
// Geometry: 256 threads = 4 waves; a block owns 64 query rows (16 per wave) of one (batch, head); K/V stream through
// LDS in 64-key tiles.  QK^T and PV both run on v_mfma_f32_16x16x32_bf16.  K is staged row-major with XOR-swizzled
// 16-B chunks (conflict-free ds_read_b128 B-fragments); V is staged row-major and consumed through the gfx950
// hardware transpose read ds_read_b64_tr_b16 (VT_SCALAR=true keeps a scalar-transposed staging as a cross-check path).
// The [S,S] score matrix is never materialised (the reference materialises [B,32,S,S] fp32).
#include "common.h"
#include <stdlib.h>

namespace {

typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(2))) float f32x2;

struct AttnArgs {
  const bf16_t* Q; const bf16_t* K; const bf16_t* V;
  bf16_t* O;
  int64_t q_sb, q_ss;   // element strides: batch, sequence (head stride is D)
  int64_t k_sb, k_ss;
  int64_t v_sb, v_ss;
  int64_t o_sb, o_ss;
  const uint8_t* key_valid;  // [B, Sk] or null
  const float* rel_h;        // [B*H, Sq, kh] or null
  const float* rel_w;        // [B*H, Sq, kw] or null
  int kh, kw;
  int B, H, Sq, Sk;
  int causal;
  float scale;
  const int* sk_dev;         // optional: number of valid keys read from device memory (<= Sk); decode steps inside a HIP graph
  float* lse2;               // optional [B*H, Sq]: row log-sum-exp of the scaled scores in the log2 domain (for the backward; v2 kernel)
  int bh_chunk;              // v2 kernel: (batch, head) pairs per chunk of the workgroup order (0 = all: the plain order)
};

template <int D>
__device__ __forceinline__ int k_off(int r, int c) {  // byte offset of 16-B chunk c of row r in a [64][D] bf16 tile
  // XOR over ALL chunk bits of the row (D = 128: r & 15; D = 64: r & 7).  With r & 7 at D = 128 the two 8-lane halves of a
  // ds_read_b128 lane group (fq = 0 rows {0-3, 12-15}, fq = 1 rows {4-11}) landed on the same eight 16-byte slots: a 2-way conflict
  // on every K fragment read (SQ_LDS_BANK_CONFLICT, scripts/attn_pmc.sh).
  return r * (D * 2) + ((c ^ (r & (D / 8 - 1))) << 4);
}

template <int D, bool VT_SCALAR>
__global__ __launch_bounds__(256) void attn_fwd_kernel(AttnArgs a) {
  if (a.sk_dev) a.Sk = min(a.Sk, a.sk_dev[0]);
  constexpr int KT = 64;            // keys per tile
  constexpr int CH = D / 8;         // 16-B chunks per row
  constexpr int NF = D / 16;        // output fragments along D
  constexpr int KS = D / 32;        // k-steps for QK^T
  // LDS: K tile [64][D] | V tile ([64][D] row-major, or V^T [D][64+8] when VT_SCALAR) | P [4 waves][16][64]
  constexpr int VT_LD = KT + 8;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sK = smem;
  char* sV = smem + KT * D * 2;
  constexpr int V_BYTES = VT_SCALAR ? D * VT_LD * 2 : KT * D * 2;
  bf16_t* sP = reinterpret_cast<bf16_t*>(smem + KT * D * 2 + V_BYTES);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fq = lane >> 4;
  const int bh = blockIdx.y, b = bh / a.H, h = bh % a.H;
  const int q0 = blockIdx.x * 64;
  const int qw0 = q0 + wave * 16;

  const bf16_t* Qb = a.Q + b * a.q_sb + (int64_t)h * D;
  const bf16_t* Kb = a.K + b * a.k_sb + (int64_t)h * D;
  const bf16_t* Vb = a.V + b * a.v_sb + (int64_t)h * D;
  const uint8_t* kv = a.key_valid ? a.key_valid + (int64_t)b * a.Sk : nullptr;

  // Q fragments (A operand): row = qw0 + fr, k = kk*32 + fq*8 .. +8
  bf16x8 qf[KS];
  {
    const int qr = min(qw0 + fr, a.Sq - 1);
#pragma unroll
    for (int kk = 0; kk < KS; ++kk)
      qf[kk] = *reinterpret_cast<const bf16x8*>(Qb + (int64_t)qr * a.q_ss + kk * 32 + fq * 8);
  }

  f32x4 o[NF];
#pragma unroll
  for (int n = 0; n < NF; ++n) o[n] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m_run[4], l_run[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) { m_run[r] = -INFINITY; l_run[r] = 0.f; }

  int n_tiles = (a.Sk + KT - 1) / KT;
  if (a.causal) n_tiles = min(n_tiles, (q0 + 64 + KT - 1) / KT);

  const float* relh = a.rel_h ? a.rel_h + (int64_t)bh * a.Sq * a.kh : nullptr;
  const float* relw = a.rel_w ? a.rel_w + (int64_t)bh * a.Sq * a.kw : nullptr;

  // K/V tiles travel global -> registers -> LDS; the loads for tile t+1 are issued right after tile t has been written
  // to LDS, so their latency is covered by the QK^T / softmax / PV work of tile t.
  constexpr int NCH = (KT * CH) / 256;     // 16-B chunks per thread per operand per tile
  bf16x8 kreg[NCH], vreg[NCH];
  auto gload = [&](int t) {
    const int k0 = t * KT;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int id = tid + i * 256;
      const int r = id / CH, c = id % CH;
      const int kr = min(k0 + r, a.Sk - 1);
      kreg[i] = *reinterpret_cast<const bf16x8*>(Kb + (int64_t)kr * a.k_ss + c * 8);
      vreg[i] = *reinterpret_cast<const bf16x8*>(Vb + (int64_t)kr * a.v_ss + c * 8);
    }
  };
  gload(0);

  for (int t = 0; t < n_tiles; ++t) {
    const int k0 = t * KT;
    __syncthreads();  // previous tile's LDS reads complete
    // ---- stage K and V tiles ----
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int id = tid + i * 256;
      const int r = id / CH, c = id % CH;
      *reinterpret_cast<bf16x8*>(sK + k_off<D>(r, c)) = kreg[i];
      if constexpr (VT_SCALAR) {
        bf16_t* vt = reinterpret_cast<bf16_t*>(sV);
#pragma unroll
        for (int j = 0; j < 8; ++j) vt[(c * 8 + j) * VT_LD + r] = vreg[i][j];
      } else {
        *reinterpret_cast<bf16x8*>(sV + r * (D * 2) + c * 16) = vreg[i];
      }
    }
    __syncthreads();
    if (t + 1 < n_tiles) gload(t + 1);

    // ---- S = Q K^T (16 q rows x 64 keys per wave) ----
    f32x4 s[4];
#pragma unroll
    for (int n = 0; n < 4; ++n) s[n] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) {
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        const bf16x8 kf = *reinterpret_cast<const bf16x8*>(sK + k_off<D>(n * 16 + fr, kk * 4 + fq));
        s[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qf[kk], kf, s[n], 0, 0, 0);
      }
    }

    // ---- scale, bias, mask, online softmax.  S layout: key = n*16 + fr, q row = fq*4 + r ----
    float alpha[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int qi = qw0 + fq * 4 + r;
      const int qc = min(qi, a.Sq - 1);
      float mx = -INFINITY;
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        const int kj = k0 + n * 16 + fr;
        float v = s[n][r] * a.scale;
        if (relh) {
          const int kc = min(kj, a.Sk - 1);
          v += relh[(int64_t)qc * a.kh + kc / a.kw] + relw[(int64_t)qc * a.kw + kc % a.kw];
        }
        bool ok = kj < a.Sk;
        if (a.causal) ok = ok && (kj <= qi);
        if (kv) ok = ok && (kv[min(kj, a.Sk - 1)] != 0);
        v = ok ? v : -INFINITY;
        s[n][r] = v;
        mx = fmaxf(mx, v);
      }
#pragma unroll
      for (int off = 1; off < 16; off <<= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
      const float m_new = fmaxf(m_run[r], mx);
      const float m_safe = (m_new == -INFINITY) ? 0.f : m_new;
      alpha[r] = __expf(m_run[r] - m_safe);
      float rs = 0.f;
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        const float p = __expf(s[n][r] - m_safe);
        s[n][r] = p;
        rs += p;
      }
#pragma unroll
      for (int off = 1; off < 16; off <<= 1) rs += __shfl_xor(rs, off, 64);
      l_run[r] = l_run[r] * alpha[r] + rs;
      m_run[r] = m_new;
    }
#pragma unroll
    for (int n = 0; n < NF; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r) o[n][r] *= alpha[r];

    // ---- P (bf16) -> LDS in A-operand order: sP[wave][row][key] ----
    bf16_t* pw = sP + wave * 16 * KT;
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r) pw[(fq * 4 + r) * KT + n * 16 + fr] = (bf16_t)s[n][r];
    // sP[wave] is written and read by this wave only: DS operations of one wave execute in order, a compiler fence suffices
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();

    // ---- O += P V : A = P[16 x 64 keys], B = V[keys x D] ----
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const bf16x8 pf = *reinterpret_cast<const bf16x8*>(pw + fr * KT + kk * 32 + fq * 8);
#pragma unroll
      for (int n = 0; n < NF; ++n) {
        bf16x8 vf;
        if constexpr (VT_SCALAR) {
          const bf16_t* vt = reinterpret_cast<const bf16_t*>(sV);
          vf = *reinterpret_cast<const bf16x8*>(vt + (n * 16 + fr) * VT_LD + kk * 32 + fq * 8);
        } else {
          // hardware transpose read: 16-lane group fq covers keys kk*32 + fq*8 + {0..7}; lane p=fr supplies the
          // 8-byte row segment (key = base + p/4, cols n*16 + (p%4)*4 ..+3) and receives column fr.
          const int key0 = kk * 32 + fq * 8 + (fr >> 2);
          const int col = n * 16 + (fr & 3) * 4;
          const char* p0 = sV + key0 * (D * 2) + col * 2;
          const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (__attribute__((address_space(3))) s16x4*)(p0));
          const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (__attribute__((address_space(3))) s16x4*)(p0 + 4 * (D * 2)));
          s16x8 both = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
          vf = __builtin_bit_cast(bf16x8, both);
        }
        o[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pf, vf, o[n], 0, 0, 0);
      }
    }
  }

  // ---- normalise and store ----
  bf16_t* Ob = a.O + b * a.o_sb + (int64_t)h * D;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int qi = qw0 + fq * 4 + r;
    if (qi >= a.Sq) continue;
    const float inv = l_run[r] > 0.f ? 1.f / l_run[r] : 0.f;
#pragma unroll
    for (int n = 0; n < NF; ++n) Ob[(int64_t)qi * a.o_ss + n * 16 + fr] = (bf16_t)(o[n][r] * inv);
  }
}


// =====================================================================================================================
// v2 (default): the transposed formulation.  S^T = K Q^T (keys on the MFMA M axis, queries on N) and O^T = V^T P^T, so
//   * the fp32 scores of a (key-fragment pair, query fragment) are, lane for lane, the B operand of the PV MFMA once rounded to
//     bf16 — P never goes through LDS (the MFMA k index is an arbitrary bijection of the 32 keys as long as the V^T fragment uses
//     the same one: k = fq*8 + t  <->  key fq*4 + t of the even fragment (t < 4), key 16 + fq*4 + t - 4 of the odd one);
//   * a lane owns ONE query column (fr) of each of its wave's two query fragments: running max / sum / rescale factor are one
//     scalar per lane and fragment, the row max needs two cross-row exchanges (fq) instead of four, and the row sum is kept as
//     a per-lane partial until the end;
//   * each wave covers 32 queries, so every K / V^T fragment read from LDS feeds two MFMAs.
// K / V tiles (64 keys) arrive by LDS-DMA into a two-stage ring (K XOR-swizzled via the source address, V row-major for the
// hardware transpose read); one barrier per tile.  exp is v_exp_f32 on scores pre-multiplied by scale*log2(e); interior tiles
// (no causal diagonal, no padding, no bias) skip every mask test.
// V tile swizzle of the v2 kernel.  The hardware transpose read (ds_read_b64_tr_b16) of a 32-lane group touches 8 consecutive V
// rows x 32 bytes; with plain row-major rows of D*2 = 256 B (D = 128) all 8 rows start on the same LDS bank: an 8-way conflict on
// every one of the 64 transpose reads of a tile (SQ_LDS_BANK_CONFLICT = 78 % of the LDS-array cycles, scripts/attn_pmc.sh).  The
// tile is stored with its 32-byte pair-blocks XORed by a per-row key (the row index in units of 256 B, modulo the pair-blocks per
// row) — on the DMA SOURCE address (the LDS image of an LDS-DMA is lane-linear; pairs of 16-byte chunks stay adjacent, so the global
// read still walks whole 32-byte pieces of the same two cache lines) and on the read address: the same involution on both sides.
template <int D>
__device__ __forceinline__ int v_key(int row) {
  // D = 64 (CLIP, rows of 128 B: a 4-way conflict) keeps the plain layout: the swizzled tile measured 41 vs 33 us there, while
  // D = 128 went 75.8 -> 73.6 us at S = 639 and 160 -> 119 us at S = 1316 with SQ_LDS_BANK_CONFLICT 14.4 M -> 0 per launch
  if constexpr (D != 128) return 0;
  constexpr int PB = D * 2 / 32;                                   // 32-byte pair-blocks per row
  return row & (PB - 1);
}

// GENERAL = false: no key-padding mask and no rel-pos bias (the un-padded Llama batch, CLIP) -- the masked tiles (causal diagonal, last
// partial key tile) then need no per-element index arithmetic: both tests compare a per-lane constant with a wave-uniform threshold.
// KT = keys per LDS stage.  64 (default): two stages, one tile in flight while one is computed.  32 (round 3 experiment, MP_ATTN_KT=32):
// the same LDS as FOUR half-size stages, so three tiles are in flight — 96 keys of prefetch instead of 64 — behind a counted vmcnt; the
// per-tile bookkeeping (two row-max shuffles, one rescale exp per query fragment, a barrier) runs twice as often.  MEASURED SLOWER on every
// shape (scripts/attn_bench.py, same box, twice): Llama causal S = 639 76.6 vs 63.1 us, CLIP 37.1 vs 32.5, S = 1316 130 vs 109 — the
// "DMA wait" of the round-2 ablation is not a prefetch-depth problem: what a deeper ring buys is less than what twice the barriers and
// online-softmax bookkeeping cost.  Kept as the A/B switch that produced the numbers; all results are identical to KT = 64's up to the
// online softmax's rescale points (tests/test_gpu_trunk_kernels.py passes with either).
// Round 3 (TUNED): the kernel is VALU-issue bound (~800 non-MFMA instructions per 64-key tile and wave against 64 MFMAs; ISA census in
// DESIGN section 10), so the tuned form removes instructions, not latency:
//   * the scores are multiplied by scale*log2(e) once, right after QK^T: fmaxf on the result of a multiply needs no canonicalisation
//     (on raw MFMA outputs it costs a v_max x,x,x per operand: 3 instructions per value instead of 1/2), the compiler folds the chain
//     into v_max3_f32, and the exponent is a subtraction.  (An inline-asm v_max3 did the same with 16 instructions fewer, but asm reads
//     of MFMA results and asm writes feeding v_permlane swaps are outside the compiler's hazard bookkeeping — see max_over_rows);
//   * the cross-row (fq) max by v_permlane16_swap / v_permlane32_swap + v_max (max_over_rows) instead of two ds_bpermute round trips through LDS;
//   * the accumulator rescale unconditional (the wave-uniform skip made the compiler copy all 64 accumulator registers around the branch:
//     32 v_mov_b64 per tile whether or not the branch was taken — as many issue slots as the 32 v_pk_mul it saved);
//   * causal self-attention (Sq == Sk): the key bound kj < Sk is implied by kj <= qi, one compare per value on the diagonal tiles;
//   * the tiles a wave skips (all keys in its queries' future) run in a trailing loop instead of a `continue`: the accumulators stopped
//     being a two-way merge at the loop header, which had cost 69 v_mov per tile on the back edge;
//   * DMA source pointers advanced by a scalar offset per whole tile; at D = 128 one V^T address register per d-block.
// Llama causal S = 639: 62.6 -> 53.4 us (501 TF/s of the causal half's flops), S = 1316 110 -> 91 us; CLIP 33.4 -> 29.1 us.
// What did NOT help (measured, removed; commit 44be988 holds the code): pipelining the loop INSIDE a wave — QK^T of tile t+1 in one basic
// block with the exponentials of tile t, interleaved by sched_group_barrier (69.0 us, and 37.0 on CLIP where no register pressure
// confounds it), and the in-tile variant (second half of the exponentials behind the first PV MFMAs: 55.2 vs 54.5).  Per 64-key tile
// the time matches MFMA + VALU issue of the two resident waves added, however they are arranged: fewer instructions is what pays.
// All-reduce (max) over the wave's four 16-lane rows (same fr), every lane gets the result; all 64 lanes must execute.  One asm block
// with its own wait states, for two reasons found the hard way (tests/test_gpu_trunk_kernels.py::test_attention_row_max_spans_all_lane_rows):
//   * with the builtins, `max(swap(x, x)[0], swap(x, x)[1])` was compiled to swap, copy, swap — the compiler folded the max away (also
//     with an opaque copy as second operand), which leaves ROW 0's max broadcast to all four rows.  A smaller-than-true max is still a
//     valid softmax shift as long as the four rows agree on it, which they did, so every tolerance test passed; a dominant key in
//     rows 1-3 overflowed 2^(s - m);
//   * a VALU write of a swap operand needs two wait states before v_permlane*_swap reads it; the compiler inserts them only for
//     instructions it emitted itself.
// v_permlane16_swap: odd rows of vdst <-> even rows of src; v_permlane32_swap: rows 2,3 of vdst <-> rows 0,1 of src.
__device__ __forceinline__ float max_over_rows(float v) {
  float t;
  asm("v_mov_b32 %1, %0\n\t"
      "s_nop 1\n\t"
      "v_permlane16_swap_b32 %0, %1\n\t"
      "s_nop 1\n\t"
      "v_max_f32 %0, %0, %1\n\t"
      "v_mov_b32 %1, %0\n\t"
      "s_nop 1\n\t"
      "v_permlane32_swap_b32 %0, %1\n\t"
      "s_nop 1\n\t"
      "v_max_f32 %0, %0, %1"
      : "+v"(v), "=&v"(t));
  return v;
}

template <int D, bool GENERAL, int KT, bool TUNED = true>
__global__ __launch_bounds__(256, 2) void attn_fwd2_kernel(AttnArgs a) {
  if (a.sk_dev) a.Sk = min(a.Sk, a.sk_dev[0]);
  constexpr int CH = D / 8, NF = D / 16, KS = D / 32;
  constexpr int NST = 128 / KT;                   // stages of the K / V ring (2 x 64 keys or 4 x 32 keys: the same bytes)
  constexpr int NKF = KT / 16, NKP = KT / 32;     // key fragments / key-fragment pairs per tile
  constexpr int TILE_BYTES = KT * D * 2;          // one operand, one stage
  constexpr int RPI = 1024 / (D * 2);             // rows per 1-KiB DMA instruction
  constexpr int IPW = (TILE_BYTES / 1024) / 4;    // DMA instructions per wave per operand per tile
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fq = lane >> 4;
  // grid = (batch*heads, query blocks): the dispatcher walks x fastest, so with the query-block index REVERSED every (batch, head)'s
  // heaviest causal block (most key tiles) starts first and the 2-tile blocks fill the tail (longest-processing-time order)
  // Round 2: ... inside CHUNKS of 64 (batch, head) pairs.  With all of a query-block rank's 256 (batch, head) pairs ahead of the next rank, the
  // five blocks of one (batch, head) ran a whole round of workgroups apart and each re-fetched its K / V (327 KB) through an L2 that had seen
  // 255 other heads in between; now they are 64 workgroup ids apart -- the same XCD (ids equal modulo 8), the same round -- and an XCD
  // holds the K / V of the 8 heads of a chunk it is working on (2.6 MB of its 4 MiB L2).  MP_ATTN_CHUNK=0 restores the plain order (A/B).
  int bh, qrank;
  {
    const int BH = gridDim.x, nqb = gridDim.y;
    const int L = blockIdx.y * BH + blockIdx.x;               // dispatch order
    const int CH = a.bh_chunk > 0 ? a.bh_chunk : BH;
    const int nch = (BH + CH - 1) / CH;
    const int chunk = min(L / (CH * nqb), nch - 1);
    const int cs = min(CH, BH - chunk * CH);
    const int rem = L - chunk * CH * nqb;
    qrank = rem / cs;
    bh = chunk * CH + rem % cs;
  }
  const int b = bh / a.H, h = bh % a.H;
  const int q0 = (a.causal ? (int)(gridDim.y - 1 - qrank) : qrank) * 128;
  const int qw0 = q0 + wave * 32;

  const bf16_t* Qb = a.Q + b * a.q_sb + (int64_t)h * D;
  const bf16_t* Kb = a.K + b * a.k_sb + (int64_t)h * D;
  const bf16_t* Vb = a.V + b * a.v_sb + (int64_t)h * D;
  const uint8_t* kv = (GENERAL && a.key_valid) ? a.key_valid + (int64_t)b * a.Sk : nullptr;
  const float* relh = (GENERAL && a.rel_h) ? a.rel_h + (int64_t)bh * a.Sq * a.kh : nullptr;
  const float* relw = (GENERAL && a.rel_w) ? a.rel_w + (int64_t)bh * a.Sq * a.kw : nullptr;

  int n_tiles = (a.Sk + KT - 1) / KT;
  if (a.causal) n_tiles = min(n_tiles, (min(q0 + 128, a.Sq) + KT - 1) / KT);

  // ---- DMA: instruction j of an operand fills LDS bytes [j*1024, +1024) = rows j*RPI .. +RPI-1; wave w issues j = w*IPW + i
  const int dma_row = lane / CH, dma_c = lane % CH;
  // TUNED: the per-lane source addresses of a WHOLE tile differ from tile 0's by a wave-uniform offset, so they are kept as pointers and
  // advanced by one scalar product per tile (2 VALU per DMA instruction instead of ~6: a 64-bit multiply-add per lane); only the last,
  // partial tile clamps rows and takes the general form
  const bf16_t* kbase[IPW];
  const bf16_t* vbase[IPW];
  if constexpr (TUNED) {
#pragma unroll
    for (int i = 0; i < IPW; ++i) {
      const int row = (wave * IPW + i) * RPI + dma_row;
      kbase[i] = Kb + (int64_t)row * a.k_ss + ((dma_c ^ (row & (CH - 1))) << 3);
      vbase[i] = Vb + (int64_t)row * a.v_ss + ((dma_c ^ (v_key<D>(row) << 1)) << 3);
    }
  }
  auto issue = [&](int t, int stage) {
    const int k0 = t * KT;
    char* sK = smem + stage * 2 * TILE_BYTES;
    char* sV = sK + TILE_BYTES;
    if (TUNED && k0 + KT <= a.Sk) {
      const int64_t ko = (int64_t)k0 * a.k_ss, vo = (int64_t)k0 * a.v_ss;          // wave-uniform
#pragma unroll
      for (int i = 0; i < IPW; ++i) {
        const int j = wave * IPW + i;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(kbase[i] + ko),
                                         (__attribute__((address_space(3))) void*)(sK + j * 1024), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(vbase[i] + vo),
                                         (__attribute__((address_space(3))) void*)(sV + j * 1024), 16, 0, 0);
      }
      return;
    }
#pragma unroll
    for (int i = 0; i < IPW; ++i) {
      const int j = wave * IPW + i;
      const int row = j * RPI + dma_row;
      const int kr = min(k0 + row, a.Sk - 1);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(Kb + (int64_t)kr * a.k_ss + ((dma_c ^ (row & (CH - 1))) << 3)),
                                       (__attribute__((address_space(3))) void*)(sK + j * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(Vb + (int64_t)kr * a.v_ss + ((dma_c ^ (v_key<D>(row) << 1)) << 3)),
                                       (__attribute__((address_space(3))) void*)(sV + j * 1024), 16, 0, 0);
    }
  };
  // Q fragments (B operand): query = qw0 + j*16 + fr, k = kk*32 + fq*8 .. +8 — requested BEFORE the ring's first tiles, so every counted
  // wait on the tile DMAs below also covers them (vmcnt retires in order)
  bf16x8 qf[2][KS];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int qr = min(qw0 + j * 16 + fr, a.Sq - 1);
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) qf[j][kk] = *reinterpret_cast<const bf16x8*>(Qb + (int64_t)qr * a.q_ss + kk * 32 + fq * 8);
  }
#pragma unroll
  for (int p_ = 0; p_ < NST - 1; ++p_)
    if (p_ < n_tiles) issue(p_, p_);

  f32x4 o[NF][2];
#pragma unroll
  for (int n = 0; n < NF; ++n)
#pragma unroll
    for (int j = 0; j < 2; ++j) o[n][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  // V^T fragment reads (swizzled tile, see v_key): this lane's key row inside a 32-key half, the address of d-block 0 and the
  // signed address step of each key bit (+/- 32, 64, 128 bytes)
  constexpr int VBITS = (D == 128) ? 3 : 2;                    // log2(pair-blocks per row) = log2(NF)
  const int v_row = fq * 4 + (fr >> 2);
  const int v_k = v_key<D>(v_row);                             // v_key(row + 16) == v_key(row + 32) == v_key(row)
  const int v_a0 = v_row * (D * 2) + (v_k << 5) + (fr & 3) * 8;
  int v_dlt[VBITS];
#pragma unroll
  for (int bit = 0; bit < VBITS; ++bit) v_dlt[bit] = ((v_k >> bit) & 1) ? -(32 << bit) : (32 << bit);
  float m_run[2] = {-INFINITY, -INFINITY}, l_part[2] = {0.f, 0.f};
  const float c2 = a.scale * 1.44269504088896340736f;          // exponent of 2 per unit of raw score (softmax in the log2 domain)
  const float inv_scale = 1.f / a.scale;                       // the rel-pos bias is added to the UNSCALED score, so it is pre-divided

  // tile t has landed when at most the DMAs of the tiles issued after it are outstanding (2 * IPW instructions per tile and wave)
  auto land_and_refill = [&](int t) {
    {
      const int later = min(NST - 2, n_tiles - 1 - t);      // wave-uniform
      if (NST > 2 && later == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * 2 * IPW) : "memory");
      else if (NST > 2 && later == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * IPW) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();                       // tile t is visible to every wave; nobody still reads the stage tile t + NST - 1 goes to
    if (t + NST - 1 < n_tiles) issue(t + NST - 1, (t + NST - 1) % NST);
  };
  // Causal: the tiles whose every key is in the future of this wave's 32 queries (k0 > qw0 + 31) need only this wave's share of the
  // workgroup's DMA / barrier protocol.  They are the LAST tiles, so they get their own trailing loop: a `continue` inside the main loop
  // made the 64 accumulator registers a two-way merge at the loop header, which the register allocator resolved with 69 v_mov per tile
  // on the back edge (ISA census, round 3).
  const int my_tiles = (TUNED && a.causal) ? min(n_tiles, (qw0 + 31) / KT + 1) : n_tiles;
  for (int t = 0; t < my_tiles; ++t) {
    const int k0 = t * KT;
    land_and_refill(t);
    if (!TUNED && a.causal && k0 > qw0 + 31) continue;            // wave-uniform: every key of this tile is in the future of this wave's queries
    const char* sK = smem + (t % NST) * 2 * TILE_BYTES;
    const char* sV = sK + TILE_BYTES;

    // ---- S^T = K Q^T: NKF key fragments x 2 query fragments ----
    f32x4 s[NKF][2];
#pragma unroll
    for (int kf = 0; kf < NKF; ++kf)
#pragma unroll
      for (int j = 0; j < 2; ++j) s[kf][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kk = 0; kk < KS; ++kk)
#pragma unroll
      for (int kf = 0; kf < NKF; ++kf) {
        const bf16x8 kfr = *reinterpret_cast<const bf16x8*>(sK + k_off<D>(kf * 16 + fr, kk * 4 + fq));
#pragma unroll
        for (int j = 0; j < 2; ++j) s[kf][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kfr, qf[j][kk], s[kf][j], 0, 0, 0);
      }

    // ---- online softmax; s[kf][j][r]: key = k0 + kf*16 + fq*4 + r, query = qw0 + j*16 + fr ----
    const bool masked = (kv != nullptr) || (relh != nullptr) || (k0 + KT > a.Sk) || (a.causal && k0 + KT - 1 > qw0);
    bf16x8 pb[NKP][2];                       // [key-fragment pair][query fragment]: B operands of the PV MFMAs
    // TUNED: log2-domain scores from here on.  The multiply sits in the SAME basic block as the fmaxf that reads it (each branch below
    // starts with it): only then does the compiler know the operand is canonical and drop the v_max x,x,x in front of every max.
    const float pre = TUNED ? c2 : 1.f;
    constexpr float LOG2E = 1.44269504088896340736f;
    const float mx_scale = TUNED ? 1.f : c2;                   // what the row max still has to be multiplied by
    const float rel_scale = TUNED ? LOG2E : inv_scale;         // the rel-pos bias joins the score in the score's current units
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int qi = qw0 + j * 16 + fr;
      float mx = -INFINITY;
      if (!masked) {
        // interior tile (non-TUNED: the max is taken on the raw scores — c2 > 0 commutes with max — and the scaling rides in the exp's fma)
#pragma unroll
        for (int kf = 0; kf < NKF; ++kf) {
          if constexpr (TUNED) s[kf][j] *= pre;
          mx = fmaxf(fmaxf(mx, fmaxf(s[kf][j][0], s[kf][j][1])), fmaxf(s[kf][j][2], s[kf][j][3]));
        }
      } else if (TUNED && !GENERAL && a.causal && a.Sq <= a.Sk) {
        // causal self-attention: kj <= qi < Sq <= Sk, so the key bound is implied (rows qi >= Sq are never stored)
        const int dl = fq * 4 - fr, cb = qw0 - k0 + j * 16;
#pragma unroll
        for (int kf = 0; kf < NKF; ++kf)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float v = (dl <= cb - kf * 16 - r) ? s[kf][j][r] * pre : -INFINITY;
            s[kf][j][r] = v;
            mx = fmaxf(mx, v);
          }
      } else if (!GENERAL) {
        // key kj = k0 + kf*16 + fq*4 + r, query qi = qw0 + j*16 + fr:
        //   kj < Sk    <=>  fq*4      <= (Sk - 1 - k0) - kf*16 - r
        //   kj <= qi   <=>  fq*4 - fr <= (qw0 - k0 + j*16) - kf*16 - r
        const int fb = fq * 4, bb = a.Sk - 1 - k0;
        const int dl = a.causal ? fq * 4 - fr : 0, cb = a.causal ? qw0 - k0 + j * 16 : 1 << 20;
#pragma unroll
        for (int kf = 0; kf < NKF; ++kf)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const bool ok = (fb <= bb - kf * 16 - r) && (dl <= cb - kf * 16 - r);
            const float v = ok ? s[kf][j][r] * pre : -INFINITY;
            s[kf][j][r] = v;
            mx = fmaxf(mx, v);
          }
      } else {
        const int qc = min(qi, a.Sq - 1);
#pragma unroll
        for (int kf = 0; kf < NKF; ++kf)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int kj = k0 + kf * 16 + fq * 4 + r;
            float v = s[kf][j][r] * pre;
            if (relh) {
              const int kc = min(kj, a.Sk - 1);
              v += (relh[(int64_t)qc * a.kh + kc / a.kw] + relw[(int64_t)qc * a.kw + kc % a.kw]) * rel_scale;
            }
            bool ok = kj < a.Sk;
            if (a.causal) ok = ok && (kj <= qi);
            if (kv) ok = ok && (kv[min(kj, a.Sk - 1)] != 0);
            v = ok ? v : -INFINITY;
            s[kf][j][r] = v;
            mx = fmaxf(mx, v);
          }
      }
      mx *= mx_scale;                                    // -inf stays -inf
      if constexpr (TUNED) mx = max_over_rows(mx);
      else {
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      }
      const float m_new = fmaxf(m_run[j], mx);
      const float m_safe = (m_new == -INFINITY) ? 0.f : m_new;
      const float alpha = __builtin_amdgcn_exp2f(m_run[j] - m_safe);
      m_run[j] = m_new;
      f32x2 rs2 = {0.f, 0.f};
#pragma unroll
      for (int kf = 0; kf < NKF; ++kf)
#pragma unroll
        for (int hp = 0; hp < 2; ++hp) {
          // p = 2^(s*c2 - m): one v_exp_f32 per value; a masked score is -inf -> p = 0
          const f32x2 sv = {s[kf][j][2 * hp], s[kf][j][2 * hp + 1]};
          const f32x2 e = TUNED ? sv - f32x2{m_safe, m_safe} : __builtin_elementwise_fma(sv, f32x2{c2, c2}, f32x2{-m_safe, -m_safe});
          const f32x2 p = {__builtin_amdgcn_exp2f(e.x), __builtin_amdgcn_exp2f(e.y)};
          rs2 += p;
          pb[kf >> 1][j][(kf & 1) * 4 + 2 * hp] = (bf16_t)p.x;
          pb[kf >> 1][j][(kf & 1) * 4 + 2 * hp + 1] = (bf16_t)p.y;
        }
      l_part[j] = fmaf(l_part[j], alpha, rs2.x + rs2.y);      // explicit: the contraction must not depend on the instantiation
      // rescale the accumulators only when some lane's running max moved (wave-uniform test)
      if (TUNED || __builtin_amdgcn_ballot_w64(alpha != 1.f)) {
#pragma unroll
        for (int n = 0; n < NF; ++n) o[n][j] *= alpha;
      }
    }

    // ---- O^T += V^T P^T: A = V^T fragment (d rows) by the hardware transpose read, B = P^T from registers ----
    // The d-blocks are walked in Gray-code order so that the swizzled address of block n, vrow_base + ((n ^ key) << 5), follows from
    // the previous one by ONE add of a per-lane delta (the bit that flips): one running address register instead of NF of them.
    if constexpr (TUNED && D == 128) {
      // one address register per d-block (block n of this lane's key row: (n ^ key) << 5), advanced to this stage by ONE add each; the
      // key pair and the +16-row half ride in the instructions' offset fields — 8 VALU per tile instead of the ~36 of the running
      // Gray-code address below (which saved 7 registers when the kernel sat at the 256-register limit; it no longer does).  D = 128 only:
      // at D = 64 (CLIP, SAM) the block-major order measured 3 us SLOWER (32.1 vs 29.1 us, CLIP shape)
      const int stage_off = (t % NST) * 2 * TILE_BYTES + TILE_BYTES;
#pragma unroll
      for (int n = 0; n < NF; ++n) {
        const char* p0 = smem + stage_off + (v_a0 ^ (n << 5));
#pragma unroll
        for (int kp = 0; kp < NKP; ++kp) {
          const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p0 + kp * 32 * (D * 2)));
          const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p0 + kp * 32 * (D * 2) + 16 * (D * 2)));
          const s16x8 both = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
          const bf16x8 vf = __builtin_bit_cast(bf16x8, both);
#pragma unroll
          for (int j = 0; j < 2; ++j) o[n][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pb[kp][j], o[n][j], 0, 0, 0);
        }
      }
    } else
#pragma unroll
    for (int kp = 0; kp < NKP; ++kp) {
      int va = v_a0 + kp * 32 * (D * 2);
#pragma unroll
      for (int g = 0; g < NF; ++g) {
        const int n = g ^ (g >> 1);
        if (g > 0) {
          const int bit = __builtin_ctz(g);
          va += ((n >> bit) & 1) ? v_dlt[bit] : -v_dlt[bit];
          asm volatile("" : "+v"(va));               // keep ONE running register (the compiler would otherwise pre-compute all NF addresses)
        }
        const char* p0 = sV + va;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p0));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p0 + 16 * (D * 2)));
        const s16x8 both = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        const bf16x8 vf = __builtin_bit_cast(bf16x8, both);
#pragma unroll
        for (int j = 0; j < 2; ++j) o[n][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pb[kp][j], o[n][j], 0, 0, 0);
      }
    }
  }
  for (int t = my_tiles; t < n_tiles; ++t) land_and_refill(t);

  // ---- normalise and store: o[n][j][r] = O[query qw0 + j*16 + fr][d = n*16 + fq*4 + r] ----
  bf16_t* Ob = a.O + b * a.o_sb + (int64_t)h * D;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    float l = l_part[j];
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
    const int qi = qw0 + j * 16 + fr;
    if (qi >= a.Sq) continue;
    const float inv = l > 0.f ? 1.f / l : 0.f;
    if (a.lse2 && fq == 0) a.lse2[(int64_t)bh * a.Sq + qi] = l > 0.f ? m_run[j] + __builtin_amdgcn_logf(l) : INFINITY;   // v_log_f32 = log2
#pragma unroll
    for (int n = 0; n < NF; ++n) {
      bf16x4 v;
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = (bf16_t)(o[n][j][r] * inv);
      *reinterpret_cast<bf16x4*>(Ob + (int64_t)qi * a.o_ss + n * 16 + fq * 4) = v;
    }
  }
}

template <int D>
int launch_attn2(const AttnArgs& a, hipStream_t stream) {
  constexpr int LDS = 4 * 64 * D * 2;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)attn_fwd2_kernel<D, false, 64, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    (void)hipFuncSetAttribute((const void*)attn_fwd2_kernel<D, true, 64, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    (void)hipFuncSetAttribute((const void*)attn_fwd2_kernel<D, false, 64, false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    (void)hipFuncSetAttribute((const void*)attn_fwd2_kernel<D, true, 64, false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    (void)hipFuncSetAttribute((const void*)attn_fwd2_kernel<D, false, 32>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    attr = true;
  }
  static int kt32 = -1;
  if (kt32 < 0) { const char* e = getenv("MP_ATTN_KT"); kt32 = (e && atoi(e) == 32) ? 1 : 0; }      // 32: four half-size stages (A/B)
  static int chunk = -1;
  if (chunk < 0) { const char* e = getenv("MP_ATTN_CHUNK"); chunk = e ? atoi(e) : 64; }
  AttnArgs ac = a;
  ac.bh_chunk = chunk;
  dim3 grid(a.B * a.H, (a.Sq + 127) / 128);
  static int plain = -1;
  if (plain < 0) { const char* e = getenv("MP_ATTN_PLAIN"); plain = (e && atoi(e) == 0) ? 0 : 1; }      // 0: always the general kernel (A/B)
  static int tuned = -1;
  if (tuned < 0) { const char* e = getenv("MP_ATTN_TUNED"); tuned = (e && atoi(e) == 0) ? 0 : 1; }       // 0: the round-2 instruction stream (A/B)
  if (plain && !a.key_valid && !a.rel_h) {
    if (kt32) hipLaunchKernelGGL((attn_fwd2_kernel<D, false, 32>), grid, dim3(256), LDS, stream, ac);
    else if (tuned) hipLaunchKernelGGL((attn_fwd2_kernel<D, false, 64, true>), grid, dim3(256), LDS, stream, ac);
    else hipLaunchKernelGGL((attn_fwd2_kernel<D, false, 64, false>), grid, dim3(256), LDS, stream, ac);
  } else if (tuned) hipLaunchKernelGGL((attn_fwd2_kernel<D, true, 64, true>), grid, dim3(256), LDS, stream, ac);
  else hipLaunchKernelGGL((attn_fwd2_kernel<D, true, 64, false>), grid, dim3(256), LDS, stream, ac);
  return mp_check_launch("mp_attention_fwd_bf16(v2)");
}

}  // namespace

// =====================================================================================================================
// Decode attention (one query per sequence against the KV cache): no MFMA tile fits a single row, and the 128-query kernel
// walks the key tiles of a head serially on one CU (~25 us at 700 keys, 32 CUs busy).  Flash-decoding layout instead: the keys of
// a (sequence, head) are split over NS workgroups (all 256 CUs stream K / V); inside a workgroup CH = D/8 lanes share a key row
// (16-byte loads, partial dot, shuffle reduction), scores sit in LDS, a block-wide max / sum, then the same lanes accumulate
// p * V.  Every split leaves (max, sum, unnormalised o[D]) in the workspace, takes a ticket, and the last arriver of the head
// merges the NS partials in split order.  fp32 throughout, P rounded to bf16 before the PV product like the tiled kernels.
// Workspace / tickets: the registered split-K scratch of the GEMM (mp_gemm_set_workspace); without it NS = 1.
void mp_gemm_split_workspace(hipStream_t stream, float** ws, int** tickets, int64_t* bytes);

namespace {

constexpr int DEC_MAX_CHUNK = 2048;         // keys per split held in LDS

template <int D>
__global__ __launch_bounds__(256) void attn_decode_kernel(AttnArgs a, float* __restrict__ ws, int* __restrict__ tickets, int NS) {
  constexpr int CH = D / 8;               // lanes per key row
  constexpr int KPB = 256 / CH;           // keys per block-iteration
  __shared__ float sc[DEC_MAX_CHUNK];
  __shared__ float red[16];
  __shared__ float part[KPB][D];
  __shared__ int s_ticket;
  const int Sk = a.sk_dev ? min(a.Sk, a.sk_dev[0]) : a.Sk;
  const int tid = threadIdx.x;
  const int c = tid % CH, kg = tid / CH;
  const int bh = blockIdx.x, b = bh / a.H, h = bh % a.H;
  const int split = blockIdx.y;
  int chunk = (Sk + NS - 1) / NS;
  chunk = ((chunk + KPB - 1) / KPB) * KPB;
  const int k_lo = min(Sk, split * chunk), k_hi = min(Sk, k_lo + chunk);
  const bf16_t* Kb = a.K + b * a.k_sb + (int64_t)h * D;
  const bf16_t* Vb = a.V + b * a.v_sb + (int64_t)h * D;
  const bf16x8 qv = *reinterpret_cast<const bf16x8*>(a.Q + b * a.q_sb + (int64_t)h * D + c * 8);
  float q[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) q[j] = (float)qv[j];
  // ---- short splits (<= 8 key rows per lane group = 128 keys at D = 128: every decode step of evaluate()): ALL K rows and ALL V rows of
  //      the split are requested up front — the V rows travel while the scores, the block max and the exponentials are computed — instead
  //      of one dependent round trip to memory per pair of rows (3 + 3 of them at 80-112 keys per split).  Same arithmetic in the same
  //      order as the general loops below (scores per key, the pairs (key, key + KPB) of the PV accumulation ascending).
  constexpr int NR = 8;
  const bool short_split = (k_hi - k_lo) <= NR * KPB;
  bf16x8 vv[NR];
  float mx = -INFINITY;
  if (short_split) {
    bf16x8 kr[NR];
#pragma unroll
    for (int i = 0; i < NR; ++i) {
      const int key = k_lo + i * KPB + kg;
      kr[i] = bf16x8{};
      if (key < k_hi) kr[i] = *reinterpret_cast<const bf16x8*>(Kb + (int64_t)key * a.k_ss + c * 8);
    }
#pragma unroll
    for (int i = 0; i < NR; ++i) {
      const int key = k_lo + i * KPB + kg;
      vv[i] = bf16x8{};
      if (key < k_hi) vv[i] = *reinterpret_cast<const bf16x8*>(Vb + (int64_t)key * a.v_ss + c * 8);
    }
#pragma unroll
    for (int i = 0; i < NR; ++i) {
      const int key = k_lo + i * KPB + kg;
      float d0 = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) d0 = fmaf(q[j], (float)kr[i][j], d0);
#pragma unroll
      for (int off = 1; off < CH; off <<= 1) d0 += __shfl_xor(d0, off, 64);
      if (key < k_hi) { d0 *= a.scale; if (c == 0) sc[key - k_lo] = d0; mx = fmaxf(mx, d0); }
    }
  } else
  for (int k0 = k_lo; k0 < k_hi; k0 += 2 * KPB) {
    const int key0 = k0 + kg, key1 = k0 + KPB + kg;
    bf16x8 kv0 = {}, kv1 = {};
    if (key0 < k_hi) kv0 = *reinterpret_cast<const bf16x8*>(Kb + (int64_t)key0 * a.k_ss + c * 8);
    if (key1 < k_hi) kv1 = *reinterpret_cast<const bf16x8*>(Kb + (int64_t)key1 * a.k_ss + c * 8);
    float d0 = 0.f, d1 = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) { d0 = fmaf(q[j], (float)kv0[j], d0); d1 = fmaf(q[j], (float)kv1[j], d1); }
#pragma unroll
    for (int off = 1; off < CH; off <<= 1) { d0 += __shfl_xor(d0, off, 64); d1 += __shfl_xor(d1, off, 64); }
    if (key0 < k_hi) { d0 *= a.scale; if (c == 0) sc[key0 - k_lo] = d0; mx = fmaxf(mx, d0); }
    if (key1 < k_hi) { d1 *= a.scale; if (c == 0) sc[key1 - k_lo] = d1; mx = fmaxf(mx, d1); }
  }
  mx = wave_max(mx);
  if ((tid & 63) == 0) red[tid >> 6] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  float sum = 0.f;
  for (int k = tid; k < k_hi - k_lo; k += 256) {
    const float p = __expf(sc[k] - mx);
    sum += p;
    sc[k] = (float)(bf16_t)p;               // P is rounded to bf16 before PV (HF: softmax(...).to(q.dtype))
  }
  sum = block_sum(sum, red);                // (also orders the sc[] writes before the reads below)
  // ---- unnormalised o = P V over this split
  float o[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] = 0.f;
  if (short_split) {
#pragma unroll
    for (int i = 0; i < NR; i += 2) {
      const int key = k_lo + i * KPB + kg, key1 = key + KPB;
      if (key < k_hi) {
        const float p0 = sc[key - k_lo], p1 = key1 < k_hi ? sc[key1 - k_lo] : 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = fmaf(p1, (float)vv[i + 1][j], fmaf(p0, (float)vv[i][j], o[j]));
      }
    }
  } else
  for (int key = k_lo + kg; key < k_hi; key += 2 * KPB) {
    const int key1 = key + KPB;
    const bf16x8 v0 = *reinterpret_cast<const bf16x8*>(Vb + (int64_t)key * a.v_ss + c * 8);
    bf16x8 v1 = {};
    float p1 = 0.f;
    if (key1 < k_hi) { v1 = *reinterpret_cast<const bf16x8*>(Vb + (int64_t)key1 * a.v_ss + c * 8); p1 = sc[key1 - k_lo]; }
    const float p0 = sc[key - k_lo];
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = fmaf(p1, (float)v1[j], fmaf(p0, (float)v0[j], o[j]));
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) part[kg][c * 8 + j] = o[j];
  __syncthreads();
  float t = 0.f;
  if (tid < D) {
#pragma unroll 4
    for (int g2 = 0; g2 < KPB; ++g2) t += part[g2][tid];
  }
  bf16_t* Op = a.O + b * a.o_sb + (int64_t)h * D;
  if (NS == 1) {
    if (tid < D) Op[tid] = (bf16_t)(sum > 0.f ? t / sum : 0.f);
    return;
  }
  // ---- partial (max, sum, o[D]) to the workspace; last arriver of the head merges in split order
  float* wsh = ws + (int64_t)bh * NS * (D + 2);
  float* mine = wsh + split * (D + 2);
  if (tid < D) __hip_atomic_store(mine + 2 + tid, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (tid == 0) { __hip_atomic_store(mine, mx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); __hip_atomic_store(mine + 1, sum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) s_ticket = __hip_atomic_fetch_add(tickets + bh, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  if (s_ticket != NS - 1) return;
  if (tid == 0) __hip_atomic_store(tickets + bh, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  // Round 4: the merge requests EVERYTHING it reads in one go — the NS (max, sum) headers by NS threads into LDS, each output thread's NS
  // partial values into registers — and only then combines them, in the same ascending split order with the same operations (same
  // bits).  Before, two runtime loops of agent-scope loads made 2 x NS dependent trips to the memory-side coherence point (the partials
  // are sc1 data of other XCDs' workgroups): ~10 of the kernel's 14 us per layer and token at NS = 8.
  constexpr int MAX_NS = 32;                               // launch_attn_decode: NS <= 16 by choice, more only for caches beyond 31 k keys
  if (NS > MAX_NS) {
    // beyond 32 splits (caches past ~31 k keys): the loop form — dependent trips to the coherence point, slow but correct, same ascending
    // order (round-4 advisor: the register form alone refused such caches)
    if (tid < D) {
      float M = -INFINITY;
      for (int sp = 0; sp < NS; ++sp) M = fmaxf(M, __hip_atomic_load(wsh + sp * (D + 2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
      float L = 0.f, acc = 0.f;
      for (int sp = 0; sp < NS; ++sp) {
        const float ms = __hip_atomic_load(wsh + sp * (D + 2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const float w = (ms == -INFINITY) ? 0.f : __expf(ms - M);
        L += w * __hip_atomic_load(wsh + sp * (D + 2) + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        acc += w * __hip_atomic_load(wsh + sp * (D + 2) + 2 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      Op[tid] = (bf16_t)(L > 0.f ? acc / L : 0.f);
    }
    return;
  }
  float* s_ms = sc;                                        // (the score array is dead by now)
  float* s_sum = sc + MAX_NS;
  if (tid < NS) {
    s_ms[tid] = __hip_atomic_load(wsh + tid * (D + 2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_sum[tid] = __hip_atomic_load(wsh + tid * (D + 2) + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  float pv[MAX_NS];
#pragma unroll
  for (int sp = 0; sp < MAX_NS; ++sp)
    pv[sp] = (tid < D && sp < NS) ? __hip_atomic_load(wsh + sp * (D + 2) + 2 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.f;
  __syncthreads();
  if (tid < D) {
    float M = -INFINITY;
    for (int sp = 0; sp < NS; ++sp) M = fmaxf(M, s_ms[sp]);
    float L = 0.f, acc = 0.f;
#pragma unroll
    for (int sp = 0; sp < MAX_NS; ++sp)
      if (sp < NS) {
        const float ms = s_ms[sp];
        const float w = (ms == -INFINITY) ? 0.f : __expf(ms - M);
        L += w * s_sum[sp];
        acc += w * pv[sp];
      }
    Op[tid] = (bf16_t)(L > 0.f ? acc / L : 0.f);
  }
}

template <int D>
int launch_attn_decode(const AttnArgs& a, hipStream_t stream) {
  float* ws = nullptr; int* tickets = nullptr; int64_t bytes = 0;
  mp_gemm_split_workspace(stream, &ws, &tickets, &bytes);
  int NS = 1;
  if (ws && a.B * a.H <= 256) {
    NS = 256 / (a.B * a.H);
    NS = NS < 1 ? 1 : (NS > 16 ? 16 : NS);
    while (NS > 1 && (int64_t)a.B * a.H * NS * (D + 2) * 4 > bytes) --NS;
  }
  while ((a.Sk + NS - 1) / NS + 64 > DEC_MAX_CHUNK) ++NS;               // a split's scores must fit the LDS array
  if (NS > 1 && !ws) return -1;
  if ((int64_t)a.B * a.H * NS * (D + 2) * 4 > bytes) { mp_set_error("mp_attention_fwd_bf16(decode): %d keys in %d splits do not fit the split workspace", a.Sk, NS); return MP_ERR_WORKSPACE; }
  hipLaunchKernelGGL((attn_decode_kernel<D>), dim3(a.B * a.H, NS), dim3(256), 0, stream, a, ws, tickets, NS);
  return mp_check_launch("mp_attention_fwd_bf16(decode)");
}

}  // namespace

namespace {

template <int D, bool VT_SCALAR>
int launch_attn(const AttnArgs& a, hipStream_t stream) {
  constexpr int KT = 64, VT_LD = KT + 8;
  constexpr int V_BYTES = VT_SCALAR ? D * VT_LD * 2 : KT * D * 2;
  constexpr int LDS = KT * D * 2 + V_BYTES + 4 * 16 * KT * 2;
  dim3 grid((a.Sq + 63) / 64, a.B * a.H);
  hipLaunchKernelGGL((attn_fwd_kernel<D, VT_SCALAR>), grid, dim3(256), LDS, stream, a);
  return mp_check_launch("mp_attention_fwd_bf16");
}

}  // namespace

extern "C" int mp_attention_fwd_bf16(const void* Q, int64_t q_sb, int64_t q_ss, const void* K, int64_t k_sb, int64_t k_ss,
                                     const void* V, int64_t v_sb, int64_t v_ss, void* O, int64_t o_sb, int64_t o_ss,
                                     const uint8_t* key_valid, const float* rel_h, const float* rel_w, int kh, int kw,
                                     int B, int H, int Sq, int Sk, int D, int causal, float scale, int variant,
                                     const int* sk_dev, hipStream_t stream) {
  MP_REQUIRE(D == 64 || D == 128, MP_ERR_SHAPE, "mp_attention_fwd_bf16: head_dim %d unsupported (64/128)", D);
  MP_REQUIRE(B > 0 && H > 0 && Sq > 0 && Sk > 0, MP_ERR_SHAPE, "mp_attention_fwd_bf16: bad shape");
  MP_REQUIRE((q_ss % 8 == 0) && (k_ss % 8 == 0) && (v_ss % 8 == 0), MP_ERR_SHAPE,
             "mp_attention_fwd_bf16: sequence strides must be multiples of 8 elements");
  MP_REQUIRE((rel_h == nullptr) == (rel_w == nullptr), MP_ERR_ARG, "mp_attention_fwd_bf16: rel_h/rel_w must come together");
  MP_REQUIRE(!rel_h || (kh > 0 && kw > 0 && kh * kw == Sk), MP_ERR_SHAPE, "mp_attention_fwd_bf16: kh*kw must equal Sk");
  AttnArgs a{(const bf16_t*)Q, (const bf16_t*)K, (const bf16_t*)V, (bf16_t*)O, q_sb, q_ss, k_sb, k_ss, v_sb, v_ss,
             o_sb, o_ss, key_valid, rel_h, rel_w, kh, kw, B, H, Sq, Sk, causal, scale, sk_dev};
  MP_REQUIRE(o_ss % 4 == 0, MP_ERR_SHAPE, "mp_attention_fwd_bf16: output sequence stride must be a multiple of 4 elements");
  // variant 0 = transposed-formulation kernel (default); 1 = first-generation kernel with scalar-transposed V; 2 = first-generation
  // kernel with the hardware transpose read (both kept as cross-checks)
  if (variant == 0 && Sq == 1 && !causal && !key_valid && !rel_h && Sk <= 16 * (DEC_MAX_CHUNK - 64)) {   // single-query decode step
    const int rc = D == 64 ? launch_attn_decode<64>(a, stream) : launch_attn_decode<128>(a, stream);
    if (rc >= 0) return rc;                                                           // (-1: needs the workspace -> tiled kernel)
  }
  // (short rel-pos attention — SAM windows / global blocks, S <= 256 — stays on the 64-query-row kernel: more workgroups for the
  // same work and a per-row bias lookup; measured 24 vs 37 us on the global blocks)
  if (variant == 0 && !(rel_h && Sq <= 256)) return D == 64 ? launch_attn2<64>(a, stream) : launch_attn2<128>(a, stream);
  if (variant == 0) variant = 2;
  if (D == 64) return variant == 1 ? launch_attn<64, true>(a, stream) : launch_attn<64, false>(a, stream);
  return variant == 1 ? launch_attn<128, true>(a, stream) : launch_attn<128, false>(a, stream);
}

// Forward that also returns the row log-sum-exp (log2 domain) the backward needs; transposed-formulation kernel only (no rel-pos).
extern "C" int mp_attention_fwd_lse_bf16(const void* Q, int64_t q_sb, int64_t q_ss, const void* K, int64_t k_sb, int64_t k_ss,
                                         const void* V, int64_t v_sb, int64_t v_ss, void* O, int64_t o_sb, int64_t o_ss,
                                         const uint8_t* key_valid, int B, int H, int Sq, int Sk, int D, int causal, float scale,
                                         float* lse2, hipStream_t stream) {
  MP_REQUIRE(D == 64 || D == 128, MP_ERR_SHAPE, "mp_attention_fwd_lse_bf16: head_dim %d unsupported (64/128)", D);
  MP_REQUIRE(B > 0 && H > 0 && Sq > 0 && Sk > 0 && lse2, MP_ERR_SHAPE, "mp_attention_fwd_lse_bf16: bad shape");
  MP_REQUIRE((q_ss % 8 == 0) && (k_ss % 8 == 0) && (v_ss % 8 == 0) && (o_ss % 4 == 0), MP_ERR_SHAPE,
             "mp_attention_fwd_lse_bf16: sequence strides must be multiples of 8 (inputs) / 4 (output) elements");
  AttnArgs a{(const bf16_t*)Q, (const bf16_t*)K, (const bf16_t*)V, (bf16_t*)O, q_sb, q_ss, k_sb, k_ss, v_sb, v_ss,
             o_sb, o_ss, key_valid, nullptr, nullptr, 0, 0, B, H, Sq, Sk, causal, scale, nullptr, lse2};
  return D == 64 ? launch_attn2<64>(a, stream) : launch_attn2<128>(a, stream);
}
