// Flash-attention BACKWARD for the Llama causal self-attention (bf16, D = 128 / 64), the first piece of the decoder backward that
// LoRA training needs (HF-4.31 LlamaAttention eager semantics, SURVEY A.1 — the autograd of
// softmax(q k^T / sqrt(D) + mask, fp32) v).  Nothing [S, S]-sized is materialised.
//
// One templated kernel, two instantiations — the transposed formulation of attn_fwd2_kernel (attention.hip) carries over.  A wave
// owns 32 RESIDENT columns (128 per workgroup) and the workgroup streams 64-row tiles of two operands X, Y through LDS:
//     T1 = X R1^T      T2 = Y R2^T            (MFMA accumulators: lane = resident column, registers = streamed rows)
//     P  = 2^(T1 c - L[query])                 dS = P o (T2 - delta[query])
//     Out1^T += X^T dS                         Out2^T += Y^T P          (X^T / Y^T by ds_read_b64_tr_b16, dS / P straight from registers)
//   dQ      : resident = (Q, dO) of a query block, streamed = (K, V) tiles; query index = lane;            Out1 = dQ
//   dK / dV : resident = (K, V) of a key block,   streamed = (Q, dO) tiles; query index = accumulator row; Out1 = dK, Out2 = dV
// L = the forward's row log-sum-exp in the log2 domain (mp_attention_fwd_lse_bf16), delta = rowsum(dO o O) (mp_attention_delta_bf16).
// Every output element has exactly one owner: no atomics, bit-reproducible.  Up to 256 accumulator registers -> one wave per SIMD.
//
// build-flags: -mllvm -amdgpu-mfma-vgpr-form
// (medplib_amd/build.py reads the line above.)  With up to 512 registers per wave the compiler's default puts the MFMA destinations in
// AGPRs; every score / dP value the VALU then touches, and everything spilled around the 256-register accumulator sets, travels through
// v_accvgpr_read / _write: 854 of the dK/dV kernel's ~1170 VALU instructions per tile were such moves (rocprofv3: 9.1 VALU instructions
// per MFMA).  The VGPR form keeps the products where the VALU reads them; ~100 moves remain in the whole kernel.
//
// Round-2 accounting (B 8, H 32, S 639, D 128; rocprofv3 counters, scripts/attn_bwd_pmc.sh): 374 -> 272 us for the whole backward (VGPR-form
// MFMAs -4 %, conflict-free swizzle + interior-tile path -10 %, delta inside the dQ kernel -5 %, one-compare diagonal tiles -5 %, two dQ
// workgroups per CU -2 %).  What is left per streamed tile of the dK/dV kernel: 2048 MFMA-pipe cycles, ~3000 VALU cycles (754 instructions:
// exp2, fma, two bf16 roundings per element and the register traffic around the 256 accumulator + resident registers), ~4000 LDS-pipe cycles
// for the workgroup (every wave reads the whole X and Y tile twice, once row-major, once transposed: 64 KB per wave and tile), and with one
// wave per SIMD the three follow each other instead of overlapping: ~12 K cycles per tile against the 2 K of the MFMA pipe.  A four-stage
// ring (MP_ATTN_BWD_STAGES=4) does not help (the DMA wait is 6 % of the time); halving the resident columns per wave to fit two waves per
// SIMD doubles the LDS traffic per MFMA, which is already level with the MFMA time.
#include "common.h"

namespace {

typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;

struct AttnBwdArgs {
  const bf16_t* Q; const bf16_t* K; const bf16_t* V; const bf16_t* dO;
  bf16_t* dQ; bf16_t* dK; bf16_t* dV;
  const float* lse2; const float* delta;            // [B*H, Sq]
  int64_t q_sb, q_ss, k_sb, k_ss, v_sb, v_ss, do_sb, do_ss;
  int64_t dq_sb, dq_ss, dk_sb, dk_ss, dv_sb, dv_ss;
  const uint8_t* key_valid;                          // [B, Sk] or null
  int B, H, Sq, Sk, causal;
  float scale;
  const bf16_t* O; int64_t o_sb, o_ss;              // optional forward output: the dQ kernel then computes delta itself and writes it to delta_out
  float* delta_out;
  const float* rope_cos; const float* rope_sin;      // optional [positions, D / 2] fp32: dQ and dK are stored rotated (see the store)
};

// XOR key of a tile row's 16-byte chunks.  A tile is read BOTH ways: row-major (ds_read_b128: 16 lanes = 16 consecutive rows, one chunk
// each) and transposed (ds_read_b64_tr_b16: 32 lanes = 8 consecutive rows x one 32-byte chunk pair).  D = 128 (16 chunks per 256-byte row):
// the key is a bijection of r & 15 (16 distinct slots for the row-major read) whose upper three bits are r & 7 (8 distinct chunk pairs for
// the transposed read); bit 0 = bit 3 of r swaps the halves of a pair, uniformly over the 8 rows of a transposed read.  The previous key
// r & 7 made rows r and r + 8 share a slot: a 2-way conflict on every T1 / T2 fragment read (SQ_LDS_BANK_CONFLICT = 47 % of the LDS cycles).
template <int D>
__device__ __forceinline__ int sw_key(int r) {
  if constexpr (D == 128) return ((r & 7) << 1) | ((r >> 3) & 1);
  return r & 7;
}

template <int D>
__device__ __forceinline__ int sw_off(int r, int c) {  // byte offset of 16-B chunk c of row r in a [64][D] bf16 tile, XOR-swizzled
  return r * (D * 2) + ((c ^ sw_key<D>(r)) << 4);
}

// A fragment of the TRANSPOSE of a swizzled [64][D] tile: M = d rows n*16 .. +15, K = the 32 streamed rows kp*32 .. +31 in the
// order the accumulator packing below uses (attn_fwd2_kernel's V^T read, with the chunk swizzle applied per lane)
template <int D>
__device__ __forceinline__ bf16x8 tr_frag(const char* tile, int kp, int n, int fr, int fq) {
  const int row = kp * 32 + fq * 4 + (fr >> 2);
  const int cb = n * 32 + (fr & 3) * 8;                               // byte column of this lane's 8-byte piece
  const char* p0 = tile + row * (D * 2) + ((((cb >> 4) ^ sw_key<D>(row)) << 4) | (cb & 15));   // sw_key(row + 16) == sw_key(row)
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p0));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p0 + 16 * (D * 2)));
  const s16x8 both = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8, both);
}

template <int D, bool DKV, int NST, int OCC>
__global__ __launch_bounds__(256, OCC) void attn_bwd_kernel(AttnBwdArgs a) {
  constexpr int TT = 64, CH = D / 8, NF = D / 16, KS = D / 32;
  constexpr int TILE_BYTES = TT * D * 2;
  constexpr int PD = NST - 1;                     // prefetch distance in tiles: the ring holds the tile in use + PD tiles in flight
  constexpr int STAT_BYTES = 1024;                // per stage: 4 x [16 lse | 16 delta | 32 unused] floats (dK/dV kernel only)
  constexpr int RPI = 1024 / (D * 2);             // rows per 1-KiB DMA instruction
  constexpr int IPW = (TILE_BYTES / 1024) / 4;    // DMA instructions per wave per operand per tile
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fq = lane >> 4;
  // grid = (batch*heads, resident blocks), x walked first by the dispatcher: under a causal mask the blocks with the most streamed
  // tiles start first (dQ: the LAST query block, so its index is reversed; dK/dV: the first key block, natural order)
  const int bh = blockIdx.x, b = bh / a.H, h = bh % a.H;
  const int blk = (!DKV && a.causal) ? (int)(gridDim.y - 1 - blockIdx.y) : (int)blockIdx.y;
  const int r0 = blk * 128;                       // first resident column of the workgroup
  const int rw0 = r0 + wave * 32;                 // ... of this wave
  const int Sres = DKV ? a.Sk : a.Sq;             // resident axis length
  const int Sstr = DKV ? a.Sq : a.Sk;             // streamed axis length

  const bf16_t* Qb = a.Q + b * a.q_sb + (int64_t)h * D;
  const bf16_t* Kb = a.K + b * a.k_sb + (int64_t)h * D;
  const bf16_t* Vb = a.V + b * a.v_sb + (int64_t)h * D;
  const bf16_t* Gb = a.dO + b * a.do_sb + (int64_t)h * D;
  const uint8_t* kv = a.key_valid ? a.key_valid + (int64_t)b * a.Sk : nullptr;
  const float* Lb = a.lse2 + (int64_t)bh * a.Sq;
  const float* Db = a.delta + (int64_t)bh * a.Sq;

  // streamed operands X (scores) and Y (dP); resident operands R1, R2
  const bf16_t* Xb = DKV ? Qb : Kb;  const int64_t x_ss = DKV ? a.q_ss : a.k_ss;
  const bf16_t* Yb = DKV ? Gb : Vb;  const int64_t y_ss = DKV ? a.do_ss : a.v_ss;
  const bf16_t* R1b = DKV ? Kb : Qb; const int64_t r1_ss = DKV ? a.k_ss : a.q_ss;
  const bf16_t* R2b = DKV ? Vb : Gb; const int64_t r2_ss = DKV ? a.v_ss : a.do_ss;

  // tile range of the streamed axis
  int t_begin = 0, t_end = (Sstr + TT - 1) / TT;
  if (a.causal) {
    if (DKV) t_begin = r0 / TT;                                   // queries before the block's first key see none of its keys
    else t_end = min(t_end, (min(r0 + 128, a.Sq) + TT - 1) / TT); // keys after the block's last query are in its future
  }

  const int dma_row = lane / CH, dma_c = lane % CH;
  char* const sStat = smem + NST * 2 * TILE_BYTES;
  auto issue = [&](int t, int stage) {
    const int s0 = t * TT;
    char* sX = smem + stage * 2 * TILE_BYTES;
    char* sY = sX + TILE_BYTES;
    if (DKV) {
      // the streamed queries' log-sum-exp and delta ride the same DMA queue (a register load here would sit BEHIND the tile DMAs in the
      // in-order vmcnt queue and its wait would drain the whole prefetch): wave w brings rows 16w .. 16w+15, lanes 0-15 lse, 16-31 delta
      const int qrow = min(s0 + wave * 16 + (lane & 15), a.Sq - 1);
      const float* src = ((lane & 16) ? Db : Lb) + qrow;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(sStat + stage * STAT_BYTES + wave * 256), 4, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < IPW; ++i) {
      const int j = wave * IPW + i;
      const int row = j * RPI + dma_row;
      const int sr = min(s0 + row, Sstr - 1);
      const int c = (dma_c ^ sw_key<D>(row)) << 3;                // the swizzle lives on the source address
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(Xb + (int64_t)sr * x_ss + c),
                                       (__attribute__((address_space(3))) void*)(sX + j * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(Yb + (int64_t)sr * y_ss + c),
                                       (__attribute__((address_space(3))) void*)(sY + j * 1024), 16, 0, 0);
    }
  };
  const int n_t = t_end - t_begin;
#pragma unroll
  for (int i = 0; i < PD; ++i)
    if (i < n_t) issue(t_begin + i, i);

  // resident fragments (B operands): column = rw0 + j*16 + fr, k = kk*32 + fq*8 .. +8
  bf16x8 r1f[2][KS], r2f[2][KS];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int rr = min(rw0 + j * 16 + fr, Sres - 1);
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) {
      r1f[j][kk] = *reinterpret_cast<const bf16x8*>(R1b + (int64_t)rr * r1_ss + kk * 32 + fq * 8);
      r2f[j][kk] = *reinterpret_cast<const bf16x8*>(R2b + (int64_t)rr * r2_ss + kk * 32 + fq * 8);
    }
  }
  // per-lane query statistics when the query is the resident index
  float Lq[2] = {0.f, 0.f}, Dq[2] = {0.f, 0.f};
  if (!DKV) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int qi = min(rw0 + j * 16 + fr, a.Sq - 1);
      Lq[j] = Lb[qi];
      if (a.O == nullptr) { Dq[j] = Db[qi]; continue; }
      // delta = rowsum(dO o O) from the resident dO fragments (r2f: this lane holds d = kk*32 + fq*8 .. +8 of its query) and the same
      // slices of O; the four fq lanes of a query add up.  Saves the separate pass over O and dO (mp_attention_delta_bf16, 36 us at S = 639).
      const bf16_t* orow = a.O + b * a.o_sb + (int64_t)qi * a.o_ss + (int64_t)h * D;
      float acc = 0.f;
#pragma unroll
      for (int kk = 0; kk < KS; ++kk) {
        const bf16x8 ov = *reinterpret_cast<const bf16x8*>(orow + kk * 32 + fq * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc = fmaf((float)ov[e], (float)r2f[j][kk][e], acc);
      }
      acc += __shfl_xor(acc, 16, 64);
      acc += __shfl_xor(acc, 32, 64);
      Dq[j] = acc;
      if (fq == 0 && rw0 + j * 16 + fr < a.Sq) a.delta_out[(int64_t)bh * a.Sq + qi] = acc;
    }
  }

  f32x4 o1[NF][2], o2[DKV ? NF : 1][2];
#pragma unroll
  for (int n = 0; n < NF; ++n)
#pragma unroll
    for (int j = 0; j < 2; ++j) o1[n][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int n = 0; n < (DKV ? NF : 1); ++n)
#pragma unroll
    for (int j = 0; j < 2; ++j) o2[n][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const float c2 = a.scale * 1.44269504088896340736f;

  constexpr int PER_TILE = 2 * IPW + (DKV ? 1 : 0);     // DMA instructions of one tile in a lane's vmcnt queue
  // Everything issued so far (the prologue tiles, the resident fragments) lands HERE, by a wait the compiler's waitcnt pass can see:
  // otherwise it keeps the resident fragments "possibly in flight" at the loop header and drains the DMA queue with a vmcnt(0) at their
  // first use in EVERY iteration.
  __builtin_amdgcn_s_waitcnt(0x0F70);                   // vmcnt(0), expcnt / lgkmcnt untouched
  for (int i = 0; i < n_t; ++i) {
    const int t = t_begin + i;
    const int s0 = t * TT;
    const int stage = i % NST;
    // tile t must have landed; the up to PD - 1 tiles issued after it may stay in flight (loads retire in order).  Iteration 0 also
    // waits (compiler-inserted) for the resident fragments, which were issued after the whole prologue.
    const int later = min(PD - 1, n_t - 1 - i);
    if (NST > 2 && later >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PER_TILE) : "memory");
    else if (NST > 2 && later == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER_TILE) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // raw s_barrier: __syncthreads() carries a release fence that drains the in-flight DMA of the later tiles with vmcnt(0).  The ring is
    // written by DMA only and every LDS read of the previous tile has been consumed by an MFMA of this wave already.
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();          // tile t has landed for every wave; nobody still reads the stage of tile t - 1
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("" ::: "memory");
    if (i + PD < n_t) issue(t + PD, (i + PD) % NST);
    if (a.causal) {                        // wave-uniform skips of tiles that cannot interact with this wave's columns
      if (DKV) { if (s0 + TT - 1 < rw0) continue; }          // every query of the tile precedes every key of the wave
      else { if (s0 > rw0 + 31) continue; }                  // every key of the tile follows every query of the wave
    }
    const char* sX = smem + stage * 2 * TILE_BYTES;
    const char* sY = sX + TILE_BYTES;

    // ---- T1 = X R1^T, T2 = Y R2^T: 4 streamed fragments x 2 resident fragments ----
    f32x4 s[4][2], dp[4][2];
#pragma unroll
    for (int f = 0; f < 4; ++f)
#pragma unroll
      for (int j = 0; j < 2; ++j) { s[f][j] = f32x4{0.f, 0.f, 0.f, 0.f}; dp[f][j] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
    for (int kk = 0; kk < KS; ++kk)
#pragma unroll
      for (int f = 0; f < 4; ++f) {
        const bf16x8 xf = *reinterpret_cast<const bf16x8*>(sX + sw_off<D>(f * 16 + fr, kk * 4 + fq));
        const bf16x8 yf = *reinterpret_cast<const bf16x8*>(sY + sw_off<D>(f * 16 + fr, kk * 4 + fq));
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          s[f][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf, r1f[j][kk], s[f][j], 0, 0, 0);
          dp[f][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(yf, r2f[j][kk], dp[f][j], 0, 0, 0);
        }
      }

    // ---- P and dS; element (f, j, r): streamed index s0 + f*16 + fq*4 + r, resident index rw0 + j*16 + fr ----
    bf16x8 pb[2][2], dsb[2][2];            // [streamed-fragment pair][resident fragment]: B operands of the output MFMAs
    // interior tile (wave-uniform): no key-padding mask, every streamed / resident index in range, the whole 64 x 32 block on the visible
    // side of the causal diagonal -- no index arithmetic, compares or selects (they were a quarter of the kernel's time)
    bool interior = (kv == nullptr) && (s0 + TT <= Sstr) && (rw0 + 32 <= Sres);
    if (a.causal) interior = interior && (DKV ? (s0 >= rw0 + 31) : (s0 + TT - 1 <= rw0));
    if (interior) {
#pragma unroll
      for (int f = 0; f < 4; ++f) {
        f32x4 Lr = {0.f, 0.f, 0.f, 0.f}, Dr = {0.f, 0.f, 0.f, 0.f};
        if (DKV) {
          const float* st = reinterpret_cast<const float*>(sStat + stage * STAT_BYTES + f * 256);
          Lr = *reinterpret_cast<const f32x4*>(st + fq * 4);
          Dr = *reinterpret_cast<const f32x4*>(st + 16 + fq * 4);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float L = DKV ? Lr[r] : Lq[j], dl = DKV ? Dr[r] : Dq[j];
            const float p = __builtin_amdgcn_exp2f(fmaf(s[f][j][r], c2, -L));
            const float ds = p * (dp[f][j][r] - dl);
            pb[f >> 1][j][(f & 1) * 4 + r] = (bf16_t)p;
            dsb[f >> 1][j][(f & 1) * 4 + r] = (bf16_t)ds;
          }
      }
    } else if (a.causal && (kv == nullptr) && (s0 + TT <= Sstr) && (rw0 + 32 <= Sres)) {
      // diagonal tile, everything in range: key <= query is ONE compare of a per-lane constant against a wave-uniform threshold
      //   dK/dV: key = rw0 + j*16 + fr, query = s0 + f*16 + fq*4 + r   <=>   fr - fq*4 <= (s0 - rw0) + (f - j)*16 + r
      //   dQ   : key = s0 + f*16 + fq*4 + r, query = rw0 + j*16 + fr   <=>   fq*4 - fr <= (rw0 - s0) + (j - f)*16 - r
      const int dl = DKV ? (fr - fq * 4) : (fq * 4 - fr);
      const int base = DKV ? (s0 - rw0) : (rw0 - s0);
#pragma unroll
      for (int f = 0; f < 4; ++f) {
        f32x4 Lr = {0.f, 0.f, 0.f, 0.f}, Dr = {0.f, 0.f, 0.f, 0.f};
        if (DKV) {
          const float* st = reinterpret_cast<const float*>(sStat + stage * STAT_BYTES + f * 256);
          Lr = *reinterpret_cast<const f32x4*>(st + fq * 4);
          Dr = *reinterpret_cast<const f32x4*>(st + 16 + fq * 4);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const bool ok = dl <= base + (DKV ? (f - j) * 16 + r : (j - f) * 16 - r);
            const float L = DKV ? Lr[r] : Lq[j], dl2 = DKV ? Dr[r] : Dq[j];
            const float p = ok ? __builtin_amdgcn_exp2f(fmaf(s[f][j][r], c2, -L)) : 0.f;
            const float ds = p * (dp[f][j][r] - dl2);
            pb[f >> 1][j][(f & 1) * 4 + r] = (bf16_t)p;
            dsb[f >> 1][j][(f & 1) * 4 + r] = (bf16_t)ds;
          }
      }
    } else {
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      f32x4 Lr = {0.f, 0.f, 0.f, 0.f}, Dr = {0.f, 0.f, 0.f, 0.f};
      if (DKV) {
        const float* st = reinterpret_cast<const float*>(sStat + stage * STAT_BYTES + f * 256);
        Lr = *reinterpret_cast<const f32x4*>(st + fq * 4);
        Dr = *reinterpret_cast<const f32x4*>(st + 16 + fq * 4);
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int ri = rw0 + j * 16 + fr;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int si = s0 + f * 16 + fq * 4 + r;
          const int qi = DKV ? si : ri, kj = DKV ? ri : si;
          bool ok = (kj < a.Sk) && (qi < a.Sq);
          if (a.causal) ok = ok && (kj <= qi);
          if (kv) ok = ok && (kv[min(kj, a.Sk - 1)] != 0);
          const float L = DKV ? Lr[r] : Lq[j], dl = DKV ? Dr[r] : Dq[j];
          const float p = ok ? __builtin_amdgcn_exp2f(fmaf(s[f][j][r], c2, -L)) : 0.f;
          const float ds = p * (dp[f][j][r] - dl);
          pb[f >> 1][j][(f & 1) * 4 + r] = (bf16_t)p;
          dsb[f >> 1][j][(f & 1) * 4 + r] = (bf16_t)ds;
        }
      }
    }
    }

    // ---- Out1^T += X^T dS, Out2^T += Y^T P ----
#pragma unroll
    for (int kp = 0; kp < 2; ++kp)
#pragma unroll
      for (int n = 0; n < NF; ++n) {
        const bf16x8 xt = tr_frag<D>(sX, kp, n, fr, fq);
#pragma unroll
        for (int j = 0; j < 2; ++j) o1[n][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xt, dsb[kp][j], o1[n][j], 0, 0, 0);
        if (DKV) {
          const bf16x8 yt = tr_frag<D>(sY, kp, n, fr, fq);
#pragma unroll
          for (int j = 0; j < 2; ++j) o2[n][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(yt, pb[kp][j], o2[n][j], 0, 0, 0);
        }
      }
  }

  // ---- store: o[n][j][r] = Out[resident rw0 + j*16 + fr][d = n*16 + fq*4 + r]; dQ and dK carry the score scale ----
  bf16_t* O1 = DKV ? a.dK + b * a.dk_sb + (int64_t)h * D : a.dQ + b * a.dq_sb + (int64_t)h * D;
  const int64_t o1_ss = DKV ? a.dk_ss : a.dq_ss;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int ri = rw0 + j * 16 + fr;
    if (ri >= Sres) continue;
    if (a.rope_cos) {
      // Round 5: the transpose of RoPE applied to dQ / dK on the way out (the LoRA backward ran mp_rope_qk_bf16 with -sin over the fused
      // [T, 3 d] gradient right after this kernel: a read-modify-write of 2/3 of it per layer).  Dims d and d + D/2 of a row are fragments
      // n and n + NF/2 of the same lane and register, the position is the resident row.  Same values as the two kernels: the gradient is
      // rounded to bf16 first (what this kernel stored), then lo' = lo cos - hi sin, hi' = hi cos + lo sin with every product and sum
      // rounded to fp32 on its own (what rope_qk_bf16_kernel compiles to: checked on the ties where a fused form differs), rounded to bf16.
      const float* cs_row = a.rope_cos + (int64_t)ri * (D / 2);
      const float* sn_row = a.rope_sin + (int64_t)ri * (D / 2);
#pragma unroll
      for (int n = 0; n < NF / 2; ++n) {
        const f32x4 cs = *reinterpret_cast<const f32x4*>(cs_row + n * 16 + fq * 4), sn = *reinterpret_cast<const f32x4*>(sn_row + n * 16 + fq * 4);
        bf16x4 vlo, vhi;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float lo = (float)(bf16_t)(o1[n][j][r] * a.scale), hi = (float)(bf16_t)(o1[n + NF / 2][j][r] * a.scale);
          vlo[r] = (bf16_t)__fsub_rn(__fmul_rn(lo, cs[r]), __fmul_rn(hi, sn[r]));      // rope_qk_bf16_kernel rounds both products (no fused form)
          vhi[r] = (bf16_t)__fadd_rn(__fmul_rn(hi, cs[r]), __fmul_rn(lo, sn[r]));
        }
        *reinterpret_cast<bf16x4*>(O1 + (int64_t)ri * o1_ss + n * 16 + fq * 4) = vlo;
        *reinterpret_cast<bf16x4*>(O1 + (int64_t)ri * o1_ss + D / 2 + n * 16 + fq * 4) = vhi;
      }
    }
#pragma unroll
    for (int n = 0; n < NF; ++n) {
      bf16x4 v;
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = (bf16_t)(o1[n][j][r] * a.scale);
      if (!a.rope_cos) *reinterpret_cast<bf16x4*>(O1 + (int64_t)ri * o1_ss + n * 16 + fq * 4) = v;
      if (DKV) {
        bf16x4 w;
#pragma unroll
        for (int r = 0; r < 4; ++r) w[r] = (bf16_t)o2[n][j][r];
        *reinterpret_cast<bf16x4*>(a.dV + b * a.dv_sb + (int64_t)h * D + (int64_t)ri * a.dv_ss + n * 16 + fq * 4) = w;
      }
    }
  }
}

// delta[bh, q] = sum_d dO[b, q, h, d] * O[b, q, h, d]   (fp32): one wave per (bh, q)
template <int D>
__global__ __launch_bounds__(256) void attn_delta_kernel(const bf16_t* __restrict__ O, int64_t o_sb, int64_t o_ss, const bf16_t* __restrict__ dO,
                                                         int64_t do_sb, int64_t do_ss, float* __restrict__ delta, int B, int H, int Sq) {
  const int lane = threadIdx.x & 63;
  const int64_t w = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (w >= (int64_t)B * H * Sq) return;
  const int q = (int)(w % Sq), bh = (int)(w / Sq), b = bh / H, h = bh % H;
  float acc = 0.f;
  for (int i = lane * 2; i < D; i += 128) {
    const bf16x2 ov = *reinterpret_cast<const bf16x2*>(O + b * o_sb + (int64_t)q * o_ss + (int64_t)h * D + i);
    const bf16x2 gv = *reinterpret_cast<const bf16x2*>(dO + b * do_sb + (int64_t)q * do_ss + (int64_t)h * D + i);
    acc = fmaf((float)ov[0], (float)gv[0], acc);
    acc = fmaf((float)ov[1], (float)gv[1], acc);
  }
  acc = wave_sum(acc);
  if (lane == 0) delta[w] = acc;
}

template <int D, int NST>
int launch_bwd_n(const AttnBwdArgs& a, hipStream_t stream) {
  constexpr int LDS = NST * 2 * 64 * D * 2 + NST * 1024;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)attn_bwd_kernel<D, false, NST, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    (void)hipFuncSetAttribute((const void*)attn_bwd_kernel<D, false, NST, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    (void)hipFuncSetAttribute((const void*)attn_bwd_kernel<D, true, NST, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    attr = true;
  }
  // the dQ kernel (one accumulator set) fits 256 registers with 17 spilled: two workgroups per CU hide some of the LDS / DMA latency that a
  // single wave per SIMD exposes -- 279 -> 272.5 us for the whole backward at B 8, S 639, same box.  MP_ATTN_BWD_OCC2=0: one per CU (A/B).
  static int occ2 = -1;
  if (occ2 < 0) { const char* e = getenv("MP_ATTN_BWD_OCC2"); occ2 = (e && atoi(e) == 0) ? 0 : 1; }
  if (occ2 && NST == 2) hipLaunchKernelGGL((attn_bwd_kernel<D, false, NST, 2>), dim3(a.B * a.H, (a.Sq + 127) / 128), dim3(256), LDS, stream, a);
  else hipLaunchKernelGGL((attn_bwd_kernel<D, false, NST, 1>), dim3(a.B * a.H, (a.Sq + 127) / 128), dim3(256), LDS, stream, a);
  hipLaunchKernelGGL((attn_bwd_kernel<D, true, NST, 1>), dim3(a.B * a.H, (a.Sk + 127) / 128), dim3(256), LDS, stream, a);
  return mp_check_launch("mp_attention_bwd_bf16");
}

// ring depth: both kernels hold one wave per SIMD (register-bound), so one workgroup owns the CU's LDS and a four-stage ring (132 KB at
// D = 128, three tiles in flight instead of one) fits.  Measured at B 8, S 639: 322-325 us against the two-stage ring's 308-312 us -- the
// wait for the next tile is not what the kernels lose their time to (no-DMA ablation: -6 %).  MP_ATTN_BWD_STAGES=4 selects it (A/B).
template <int D>
int launch_bwd(const AttnBwdArgs& a, hipStream_t stream) {
  static int nst = -1;
  if (nst < 0) { const char* e = getenv("MP_ATTN_BWD_STAGES"); nst = (e && atoi(e) == 4) ? 4 : 2; }
  return nst == 2 ? launch_bwd_n<D, 2>(a, stream) : launch_bwd_n<D, 4>(a, stream);
}

}  // namespace

extern "C" int mp_attention_delta_bf16(const void* O, int64_t o_sb, int64_t o_ss, const void* dO, int64_t do_sb, int64_t do_ss, float* delta,
                                       int B, int H, int Sq, int D, hipStream_t stream) {
  MP_REQUIRE(D == 64 || D == 128, MP_ERR_SHAPE, "mp_attention_delta_bf16: head_dim %d unsupported (64/128)", D);
  MP_REQUIRE(B > 0 && H > 0 && Sq > 0 && o_ss % 2 == 0 && do_ss % 2 == 0, MP_ERR_SHAPE, "mp_attention_delta_bf16: bad shape");
  const int64_t waves = (int64_t)B * H * Sq;
  if (D == 64) hipLaunchKernelGGL(attn_delta_kernel<64>, dim3((unsigned)mp_cdiv(waves, 4)), dim3(256), 0, stream, (const bf16_t*)O, o_sb, o_ss,
                                  (const bf16_t*)dO, do_sb, do_ss, delta, B, H, Sq);
  else hipLaunchKernelGGL(attn_delta_kernel<128>, dim3((unsigned)mp_cdiv(waves, 4)), dim3(256), 0, stream, (const bf16_t*)O, o_sb, o_ss,
                          (const bf16_t*)dO, do_sb, do_ss, delta, B, H, Sq);
  return mp_check_launch("mp_attention_delta_bf16");
}

extern "C" int mp_attention_bwd_bf16(const void* Q, int64_t q_sb, int64_t q_ss, const void* K, int64_t k_sb, int64_t k_ss, const void* V,
                                     int64_t v_sb, int64_t v_ss, const void* dO, int64_t do_sb, int64_t do_ss, const float* lse2,
                                     const float* delta, void* dQ, int64_t dq_sb, int64_t dq_ss, void* dK, int64_t dk_sb, int64_t dk_ss,
                                     void* dV, int64_t dv_sb, int64_t dv_ss, const uint8_t* key_valid, int B, int H, int Sq, int Sk, int D,
                                     int causal, float scale, hipStream_t stream) {
  MP_REQUIRE(D == 64 || D == 128, MP_ERR_SHAPE, "mp_attention_bwd_bf16: head_dim %d unsupported (64/128)", D);
  MP_REQUIRE(B > 0 && H > 0 && Sq > 0 && Sk > 0, MP_ERR_SHAPE, "mp_attention_bwd_bf16: bad shape");
  MP_REQUIRE(q_ss % 8 == 0 && k_ss % 8 == 0 && v_ss % 8 == 0 && do_ss % 8 == 0 && dq_ss % 4 == 0 && dk_ss % 4 == 0 && dv_ss % 4 == 0, MP_ERR_SHAPE,
             "mp_attention_bwd_bf16: input sequence strides must be multiples of 8 elements, output ones of 4");
  MP_REQUIRE(lse2 && delta, MP_ERR_ARG, "mp_attention_bwd_bf16: needs the forward's log-sum-exp and delta");
  AttnBwdArgs a{(const bf16_t*)Q, (const bf16_t*)K, (const bf16_t*)V, (const bf16_t*)dO, (bf16_t*)dQ, (bf16_t*)dK, (bf16_t*)dV, lse2, delta,
                q_sb, q_ss, k_sb, k_ss, v_sb, v_ss, do_sb, do_ss, dq_sb, dq_ss, dk_sb, dk_ss, dv_sb, dv_ss, key_valid, B, H, Sq, Sk, causal, scale,
                nullptr, 0, 0, nullptr, nullptr, nullptr};
  return D == 64 ? launch_bwd<64>(a, stream) : launch_bwd<128>(a, stream);
}

extern "C" int mp_attention_bwd_fused_bf16(const void* Q, int64_t q_sb, int64_t q_ss, const void* K, int64_t k_sb, int64_t k_ss, const void* V,
                                           int64_t v_sb, int64_t v_ss, const void* O, int64_t o_sb, int64_t o_ss, const void* dO, int64_t do_sb,
                                           int64_t do_ss, const float* lse2, float* delta_ws, void* dQ, int64_t dq_sb, int64_t dq_ss, void* dK,
                                           int64_t dk_sb, int64_t dk_ss, void* dV, int64_t dv_sb, int64_t dv_ss, const uint8_t* key_valid, int B,
                                           int H, int Sq, int Sk, int D, int causal, float scale, const float* rope_cos, const float* rope_sin,
                                           hipStream_t stream) {
  MP_REQUIRE(D == 64 || D == 128, MP_ERR_SHAPE, "mp_attention_bwd_fused_bf16: head_dim %d unsupported (64/128)", D);
  MP_REQUIRE((rope_cos == nullptr) == (rope_sin == nullptr) && (!rope_cos || Sq == Sk), MP_ERR_ARG, "mp_attention_bwd_fused_bf16: rope_cos / rope_sin come together (self-attention)");
  MP_REQUIRE(B > 0 && H > 0 && Sq > 0 && Sk > 0, MP_ERR_SHAPE, "mp_attention_bwd_fused_bf16: bad shape");
  MP_REQUIRE(q_ss % 8 == 0 && k_ss % 8 == 0 && v_ss % 8 == 0 && do_ss % 8 == 0 && o_ss % 8 == 0 && dq_ss % 4 == 0 && dk_ss % 4 == 0 && dv_ss % 4 == 0,
             MP_ERR_SHAPE, "mp_attention_bwd_fused_bf16: input sequence strides must be multiples of 8 elements, output ones of 4");
  MP_REQUIRE(lse2 && delta_ws && O, MP_ERR_ARG, "mp_attention_bwd_fused_bf16: needs the forward's output, its log-sum-exp and a [B*H, Sq] fp32 workspace");
  AttnBwdArgs a{(const bf16_t*)Q, (const bf16_t*)K, (const bf16_t*)V, (const bf16_t*)dO, (bf16_t*)dQ, (bf16_t*)dK, (bf16_t*)dV, lse2, delta_ws,
                q_sb, q_ss, k_sb, k_ss, v_sb, v_ss, do_sb, do_ss, dq_sb, dq_ss, dk_sb, dk_ss, dv_sb, dv_ss, key_valid, B, H, Sq, Sk, causal, scale,
                (const bf16_t*)O, o_sb, o_ss, delta_ws, rope_cos, rope_sin};
  return D == 64 ? launch_bwd<64>(a, stream) : launch_bwd<128>(a, stream);
}
