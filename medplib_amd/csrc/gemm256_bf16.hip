// 256x256x64 "ping-pong" bf16 MFMA GEMM for gfx950 (large trunk GEMMs: Llama qkv / o / gate|up / down, experts).
//
// 512 threads = 8 waves as 2 (M) x 4 (N); each wave owns a 128x64 output tile = 8x4 fragments of
// v_mfma_f32_16x16x32_bf16 (128 accumulator registers).  One workgroup per CU (128 KiB LDS: 2 stages x (A 32 KiB + B 32 KiB)),
// two waves per SIMD: waves w and w+4 share a SIMD, and the two M-halves (wave groups) run ONE BARRIER OUT OF PHASE, so on
// every SIMD one wave is in a 32-MFMA segment while its partner is in its LDS-read / DMA-issue segment:
//
//   group 0:        LOAD(h) | B | MFMA(h) | B | LOAD(h+1) | B | MFMA(h+1) | B ...
//   group 1:   B  |  LOAD(h) | B | MFMA(h) | B | LOAD(h+1) | B | MFMA(h+1) ...
//
// A K-tile (64 deep) is two segments = the two 64x64 halves of the wave tile (rows 0-63, then 64-127; the B fragments stay
// in registers for both): 16 + 8 ds_read_b128 and 4 barriers per K-tile.  (Four 16-MFMA quadrant segments with 8 barriers
// per K-tile, the schedule this kernel had first, measured 6-7 % slower on every Llama shape: the segment hand-off, not the
// barrier count of the skeleton, is what costs.)
// Operands arrive by LDS-DMA (global_load_lds_dwordx4, lane-linear LDS image, XOR swizzle applied on the SOURCE address) as a
// continuous stream: a stage is recycled region by region as soon as its last reader phase has retired (details at the kernel).
// The MFMAs are issued as (W fragment, A fragment), which transposes the accumulator fragments: a lane ends up owning four
// CONSECUTIVE COLUMNS of a row, so the epilogue (bias / activation / residual / SwiGLU pairing / MoE row scatter) stores 8- or
// 16-byte row pieces straight from registers — no LDS round trip.
// Work decode: a flat unit id over all batches with XCD chunking and a grouped tile order, plus tail split-K (at the kernel).
// Tried and dropped: BK = 32 with 4 stages (half cache lines per DMA row), 32x32x16 MFMAs (dependent-accumulator distance 2),
// per-tile DMA bursts with vmcnt(0) (the "v1" schedule this file started with), an LDS-transposed epilogue with 16-byte stores
// (slower than the register-direct one by 5-20 % depending on K), and persistent workgroups with the next tile's prologue
// issued before the epilogue (+2-4 % on multi-wave shapes in isolation, -2.3 % on the training step: workgroups that never
// leave the CUs starve the concurrent SAM-encoder and mask-tail streams).  Round 2: the buffer-descriptor form of the LDS-DMA
// (raw_ptr_buffer_load_lds: 32-bit lane offsets computed once, the K advance in the scalar offset, no VALU per piece) measured
// 0.5-1 % SLOWER on every Llama shape; 16-byte write-through stores / loads for the tail split-K partials changed nothing (the
// 64 MB + 64 MB of partial traffic per N = 4096 launch is the cost, not the store width); L2 hit rate 80 % on the multi-wave
// shapes = what a 4 x 8 block of co-resident tiles can reach (scripts/gemm_l2_pmc.sh).
// Round 2, what DID pay: the epilogue.  bench.py's per-shape samples showed the N = 4096 projections 20-30 us slower in the model than
// alone; the residual epilogue was the difference (scripts/gemm_sustained.py: o 158 -> 188 us with a residual): eight dependent round
// trips to memory, one per fragment row.  All residual pieces are now requested up front (the operand fragments' registers are dead by
// then) and every bf16 output path stores 16-byte pieces built with v_permlane16_swap (pair_swap16): o 188 -> 165 us, down 373 -> 357
// alone; in the model o 192 -> 172, down (expert, combine folded in) 397 -> 374, qkv + RoPE 416 -> 407; same box A/B of the whole
// step (MP_GEMM_EP8=1 restores the 8-byte epilogue): 74.6 -> 72.7 ms.
#include "gemm_common.h"
#include <stdlib.h>
#include <algorithm>
#include <map>
#include <mutex>
#include <utility>

namespace {

constexpr int BM2 = 256, BN2 = 256, BK2 = 64, NT2 = 512;
constexpr int STAGE_BYTES = 65536, OP_BYTES = 32768;

__device__ __forceinline__ int lds_off2(int r, int c) { return r * 128 + ((c ^ (r & 7)) << 4); }

// raw s_barrier (no fence: a __syncthreads() would drain the in-flight LDS-DMA with vmcnt(0)); the empty asm statements are
// compiler-only memory fences so no LDS access is moved across the barrier at IR level, sched_barrier pins the machine schedule
#define MP_BAR()                           \
  do {                                     \
    asm volatile("" ::: "memory");         \
    __builtin_amdgcn_sched_barrier(0);     \
    __builtin_amdgcn_s_barrier();          \
    __builtin_amdgcn_sched_barrier(0);     \
    asm volatile("" ::: "memory");         \
  } while (0)

// Epilogue for TRANSPOSED accumulators (v3): the MFMAs are issued as (W fragment, A fragment), so acc[i][j][r] is
// C[row = i*16 + fr, col = j*16 + fq*4 + r] of the wave tile — a lane owns FOUR CONSECUTIVE COLUMNS of one row per fragment.
// Outputs leave straight from registers as 8-byte (bf16) / 16-byte (fp32) pieces, four fragments completing each 128-byte row
// segment; bias / activation / residual / SwiGLU pairing are per-lane register work with the same rounding points as the LDS
// epilogue (activation result rounded to bf16 before the residual add or the silu(gate)*up product).  No LDS, no barrier: the
// operand ring is free for whatever comes next.
__device__ __forceinline__ void gemm256_epilogue_t(const GemmArgs& g, f32x4 (&acc)[8][4], int batch, int M, int N, int m0, int n0,
                                                   int wr, int wc, int fr, int fq) {
  if (g.c_rows) {
    // combine folded into the epilogue (top-1 MoE down projection): out[token] = residual[token] + weight[token] * bf16(acc) with
    // the same rounding points as the separate combine kernel (expert output rounded to bf16 first)
    bf16_t* Cb = reinterpret_cast<bf16_t*>(g.C);
    const int cw = n0 + wc * 64;
    if (!g.ep8 && (g.ldc & 7) == 0 && (N & 7) == 0 && aligned16(Cb) && (!g.residual || ((g.ldr & 7) == 0 && aligned16(g.residual)))) {
      // 16-byte pieces (pair_swap16); the row indices, scales and residual pieces of the whole wave tile are fetched before the first one
      // is used (eight dependent round trips to memory otherwise: +25 us per launch on the N = 4096 shapes)
      int orow[8];
      float sc[8];
      bf16x8 rv[8][2];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int row = m0 + wr * 128 + i * 16 + fr;
        orow[i] = row < M ? g.c_rows[batch * g.rows_stride + row] : -1;
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        sc[i] = (orow[i] >= 0 && g.c_scale) ? g.c_scale[orow[i]] : 1.f;
#pragma unroll
        for (int jp = 0; jp < 2; ++jp) {
          const int col = cw + jp * 32 + pair_col8(fq);
          rv[i][jp] = bf16x8{};
          if (g.residual && orow[i] >= 0 && col < N) rv[i][jp] = *reinterpret_cast<const bf16x8*>(g.residual + (int64_t)orow[i] * g.ldr + col);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
#pragma unroll
        for (int jp = 0; jp < 2; ++jp) {
          const int col = cw + jp * 32 + pair_col8(fq);
          const bf16x8 p = pair_swap16(round4(acc[i][2 * jp]), round4(acc[i][2 * jp + 1]));
          bf16x8 o;
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            float v = (float)p[e] * sc[i];
            if (g.residual) v += (float)rv[i][jp][e];
            o[e] = (bf16_t)v;
          }
          if (orow[i] >= 0 && col < N) *reinterpret_cast<bf16x8*>(Cb + (int64_t)orow[i] * g.ldc + col) = o;
        }
      }
      return;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int row = m0 + wr * 128 + i * 16 + fr;
      const int orow = row < M ? g.c_rows[batch * g.rows_stride + row] : 0;
      const float sc = (row < M && g.c_scale) ? g.c_scale[orow] : 1.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int col = cw + j * 16 + fq * 4;
        if (row < M && col + 4 <= N) {
          f32x4 v = {(float)(bf16_t)acc[i][j][0], (float)(bf16_t)acc[i][j][1], (float)(bf16_t)acc[i][j][2], (float)(bf16_t)acc[i][j][3]};
          v *= sc;
          if (g.residual) {
            const bf16x4 rv = *reinterpret_cast<const bf16x4*>(g.residual + (int64_t)orow * g.ldr + col);
            v += f32x4{(float)rv[0], (float)rv[1], (float)rv[2], (float)rv[3]};
          }
          *reinterpret_cast<bf16x4*>(Cb + (int64_t)orow * g.ldc + col) = bf16x4{(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    return;
  }
  const float* bias0 = g.bias ? g.bias + batch * g.sBias : nullptr;
  const bf16_t* R = g.residual ? g.residual + batch * g.sR : nullptr;
  const int cw = n0 + wc * 64;                                  // first column of the wave tile
  // ---- pass 1: alpha, bias (in place)
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    f32x4 bv = {0.f, 0.f, 0.f, 0.f};
    if (bias0) {
#pragma unroll
      for (int r = 0; r < 4; ++r) { const int col = cw + j * 16 + fq * 4 + r; bv[r] = col < N ? bias0[col] : 0.f; }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i][j] = acc[i][j] * g.alpha + bv;
  }
  // ---- pass 2: activation (one fully unrolled sweep per activation kind; the switch stays outside the sweeps)
#define MP_ACT_SWEEP(EXPR)                                                          \
  _Pragma("unroll") for (int i = 0; i < 8; ++i) _Pragma("unroll") for (int j = 0; j < 4; ++j) {                    \
    _Pragma("unroll") for (int r = 0; r < 4; ++r) { const float v = acc[i][j][r]; acc[i][j][r] = (EXPR); }         \
    __builtin_amdgcn_sched_barrier(0); /* one fragment's temporaries at a time (erff is register hungry) */         \
  }
  switch (g.act) {
    case ACT_RELU: MP_ACT_SWEEP(fmaxf(v, 0.f)) break;
    case ACT_GELU: MP_ACT_SWEEP(gelu_erf_fast(v)) break;
    case ACT_QUICK_GELU: MP_ACT_SWEEP(v * mp_sigmoid_fast(v, 1.702f)) break;
    case ACT_SILU: MP_ACT_SWEEP(v * mp_sigmoid_fast(v)) break;
    default: break;
  }
#undef MP_ACT_SWEEP
  // ---- pass 3: stores
  if (g.out_f32) {
    float* Cf = reinterpret_cast<float*>(g.C) + batch * g.sC;
    const bool vec = (g.ldc & 3) == 0 && (!R || (g.ldr & 3) == 0);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int row = m0 + wr * 128 + i * 16 + fr;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int col = cw + j * 16 + fq * 4;
        f32x4 v = acc[i][j];
        if (row < M && col < N) {
          if (vec && col + 4 <= N) {
            if (R) {
              const bf16x4 rv = *reinterpret_cast<const bf16x4*>(R + (int64_t)row * g.ldr + col);
              v += f32x4{(float)rv[0], (float)rv[1], (float)rv[2], (float)rv[3]};
            }
            *reinterpret_cast<f32x4*>(Cf + (int64_t)row * g.ldc + col) = v;
          } else {
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (col + r < N) Cf[(int64_t)row * g.ldc + col + r] = v[r] + (R ? (float)R[(int64_t)row * g.ldr + col + r] : 0.f);
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);            // one row of fragments at a time: keeps the address / residual registers bounded
    }
    return;
  }
  bf16_t* Cb = reinterpret_cast<bf16_t*>(g.C) + batch * g.sC;
  if (g.act == ACT_SWIGLU_PAIR) {
    // W rows are [gate 0..31 | up 0..31 | gate 32..63 | ...]: fragments j = 0,1 are gate columns, j = 2,3 the matching up columns;
    // gate and up are rounded to bf16 first, like the unfused GEMM + SwiGLU kernel pair
    const int half_n = N >> 1;
    if ((!g.ep8 || g.keep_gu) && (g.ldc & 7) == 0 && (half_n & 7) == 0 && aligned16(Cb)) {      // (keep_gu exists in the 16-byte form only: its entry point checks the alignment)
      const int col = (cw >> 1) + pair_col8(fq);               // the lane's eight output columns (pair_swap16 of the j = 0, 1 pieces)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int row = m0 + wr * 128 + i * 16 + fr;
        bf16x4 o[2];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float gf = (float)(bf16_t)acc[i][j][r];
            const float uf = (float)(bf16_t)acc[i][j + 2][r];
            o[j][r] = (bf16_t)(gf * mp_sigmoid_fast(gf) * uf);
          }
        const bf16x8 p = pair_swap16(o[0], o[1]);
        if (row < M && col < half_n) *reinterpret_cast<bf16x8*>(Cb + (int64_t)row * g.ldc + col) = p;
        if (g.keep_gu) {                                     // the bf16 gate / up values the product was formed from, in the GEMM's own column order
          const bf16x8 pg = pair_swap16(round4(acc[i][0]), round4(acc[i][1])), pu = pair_swap16(round4(acc[i][2]), round4(acc[i][3]));
          if (row < M && cw + 64 <= N) {
            *reinterpret_cast<bf16x8*>(g.keep_gu + (int64_t)row * g.ld_gu + cw + pair_col8(fq)) = pg;
            *reinterpret_cast<bf16x8*>(g.keep_gu + (int64_t)row * g.ld_gu + cw + 32 + pair_col8(fq)) = pu;
          }
        }
        if (i & 1) __builtin_amdgcn_sched_barrier(0);
      }
      return;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int row = m0 + wr * 128 + i * 16 + fr;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int col = (cw >> 1) + j * 16 + fq * 4;
        bf16x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float gf = (float)(bf16_t)acc[i][j][r];
          const float uf = (float)(bf16_t)acc[i][j + 2][r];
          o[r] = (bf16_t)(gf * mp_sigmoid_fast(gf) * uf);
        }
        if (row < M && col + 4 <= half_n) *reinterpret_cast<bf16x4*>(Cb + (int64_t)row * g.ldc + col) = o;
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    return;
  }
  if (g.act == ACT_ROPE_QK && n0 < (N / 3) * 2) {
    // q / k tile of the fused qkv projection (N = 3 * hidden, hidden % 256 == 0: a tile never straddles two thirds).  W rows are
    // interleaved per 128-wide head (gemm_common.h), so this wave's 64 columns are [lo 32 dims | hi 32 dims] of ONE head: fragments
    // j = 0,1 hold x_i, fragments j + 2 the matching x_{i+64}.  Same arithmetic and rounding points as the stand-alone kernel
    // (mp_rope_qk_bf16: the projection output is rounded to bf16 first, the rotation runs in fp32, one rounding per output).
    const int head0 = (cw >> 7) << 7;                 // first column of the head (standard layout)
    const int blk = (cw >> 6) & 1;                    // which 32-dim block of lo / hi this wave holds
    if (!g.ep8 && (g.ldc & 7) == 0 && aligned16(Cb)) {
      // 16-byte pieces: the lo pieces of fragments 0, 1 pair up, and the hi pieces of fragments 2, 3; the cos / sin rows of four matrix
      // rows are fetched together (one round trip to L2 per half instead of one per row)
      const int c8 = blk * 32 + pair_col8(fq);        // the lane's eight rotary indices after the swap
#pragma unroll
      for (int ih = 0; ih < 2; ++ih) {
        f32x4 cs[4][2], sn[4][2];
#pragma unroll
        for (int ii = 0; ii < 4; ++ii) {
          const int row = m0 + wr * 128 + (ih * 4 + ii) * 16 + fr;
          const int pos = (row < M ? row : 0) % g.rope_seq + g.rope_pos0;
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const int dim = blk * 32 + j * 16 + fq * 4;
            cs[ii][j] = *reinterpret_cast<const f32x4*>(g.rope_cos + (int64_t)pos * 64 + dim);
            sn[ii][j] = *reinterpret_cast<const f32x4*>(g.rope_sin + (int64_t)pos * 64 + dim);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ii = 0; ii < 4; ++ii) {
          const int i = ih * 4 + ii;
          const int row = m0 + wr * 128 + i * 16 + fr;
          bf16x4 olo[2], ohi[2];
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float a = (float)(bf16_t)acc[i][j][r], b = (float)(bf16_t)acc[i][j + 2][r];
              olo[j][r] = (bf16_t)(a * cs[ii][j][r] - b * sn[ii][j][r]);
              ohi[j][r] = (bf16_t)(b * cs[ii][j][r] + a * sn[ii][j][r]);
            }
          const bf16x8 plo = pair_swap16(olo[0], olo[1]), phi = pair_swap16(ohi[0], ohi[1]);
          if (row < M) {
            *reinterpret_cast<bf16x8*>(Cb + (int64_t)row * g.ldc + head0 + c8) = plo;
            *reinterpret_cast<bf16x8*>(Cb + (int64_t)row * g.ldc + head0 + 64 + c8) = phi;
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      return;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int row = m0 + wr * 128 + i * 16 + fr;
      const int pos = (row < M ? row : 0) % g.rope_seq + g.rope_pos0;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int dim = blk * 32 + j * 16 + fq * 4;   // rotary index i in [0, 64)
        const f32x4 cs = *reinterpret_cast<const f32x4*>(g.rope_cos + (int64_t)pos * 64 + dim);
        const f32x4 sn = *reinterpret_cast<const f32x4*>(g.rope_sin + (int64_t)pos * 64 + dim);
        bf16x4 olo, ohi;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float a = (float)(bf16_t)acc[i][j][r], b = (float)(bf16_t)acc[i][j + 2][r];
          olo[r] = (bf16_t)(a * cs[r] - b * sn[r]);
          ohi[r] = (bf16_t)(b * cs[r] + a * sn[r]);
        }
        if (row < M) {
          *reinterpret_cast<bf16x4*>(Cb + (int64_t)row * g.ldc + head0 + dim) = olo;
          *reinterpret_cast<bf16x4*>(Cb + (int64_t)row * g.ldc + head0 + 64 + dim) = ohi;
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    return;
  }
  if (!g.ep8 && (g.ldc & 7) == 0 && (N & 7) == 0 && aligned16(Cb) && (!R || ((g.ldr & 7) == 0 && aligned16(R)))) {
    // 16-byte pieces (pair_swap16).  All sixteen residual pieces of the lane are requested before the first is used: the operand
    // fragments' 64 registers are dead here, and eight dependent round trips to memory (one per fragment row, the first version) cost
    // 25-30 us per launch on the N = 4096 projections.  The activation result is rounded to bf16 before the residual add (HF's own
    // order: act(x) is a bf16 tensor).
    bf16x8 rv[8][2];
    if (R) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int row = m0 + wr * 128 + i * 16 + fr;
#pragma unroll
        for (int jp = 0; jp < 2; ++jp) {
          const int col = cw + jp * 32 + pair_col8(fq);
          rv[i][jp] = bf16x8{};
          if (row < M && col < N) rv[i][jp] = *reinterpret_cast<const bf16x8*>(R + (int64_t)row * g.ldr + col);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int row = m0 + wr * 128 + i * 16 + fr;
#pragma unroll
      for (int jp = 0; jp < 2; ++jp) {
        const int col = cw + jp * 32 + pair_col8(fq);
        bf16x8 p = pair_swap16(round4(acc[i][2 * jp]), round4(acc[i][2 * jp + 1]));
        if (R) {
#pragma unroll
          for (int e = 0; e < 8; ++e) p[e] = (bf16_t)((float)p[e] + (float)rv[i][jp][e]);
        }
        if (row < M && col < N) *reinterpret_cast<bf16x8*>(Cb + (int64_t)row * g.ldc + col) = p;
      }
    }
    return;
  }
  const bool vec = (g.ldc & 3) == 0 && (!R || (g.ldr & 3) == 0);
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int row = m0 + wr * 128 + i * 16 + fr;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int col = cw + j * 16 + fq * 4;
      // the activation result is rounded to bf16 before the residual add (HF's own order: act(x) is a bf16 tensor)
      f32x4 v = {(float)(bf16_t)acc[i][j][0], (float)(bf16_t)acc[i][j][1], (float)(bf16_t)acc[i][j][2], (float)(bf16_t)acc[i][j][3]};
      if (row < M && col < N) {
        if (vec && col + 4 <= N) {
          if (R) {
            const bf16x4 rv = *reinterpret_cast<const bf16x4*>(R + (int64_t)row * g.ldr + col);
            v += f32x4{(float)rv[0], (float)rv[1], (float)rv[2], (float)rv[3]};
          }
          *reinterpret_cast<bf16x4*>(Cb + (int64_t)row * g.ldc + col) = bf16x4{(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (col + r < N) Cb[(int64_t)row * g.ldc + col + r] = (bf16_t)(v[r] + (R ? (float)R[(int64_t)row * g.ldr + col + r] : 0.f));
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}


// =====================================================================================================================
// The kernel ("v3" keeps the name it had among the variants that were tried): BK = 64, 128-B rows = whole cache lines per DMA
// row, 2 stages, with a CONTINUOUS DMA stream: a stage is recycled region by region as soon as its last reader segment has
// retired, and one counted wait (vmcnt(6)) per K-tile retires it -- the memory pipe never drains.
//   reads:  H0  B (all 64 columns) + A rows 0-63        H1  A rows 64-127
//   DMA  :  H0  A rows 64-127 of tile t+1 (2 pieces)     H1  A rows 0-63 + B of tile t+2 (6 pieces)
// (B and the A rows 0-63 are free after H0's reads, the A rows 64-127 after H1's.)  Every LOAD segment ends with lgkmcnt(0)
// before its barrier, so a region's reads are retired by all waves before the barrier that precedes the first DMA into it;
// within a segment the fragment reads are issued first and the DMA pieces behind them (+2 %).
// Where the time goes (8192^3, one box, scripts/gemm_ab.py with MP_GEMM_ABLATE): everything 750 us; skeleton + epilogue alone
// 99; + MFMA only 498; + fragment reads only 297; + DMA only 549 (380 when every workgroup fetches the same tile, 295 when it
// is also the same K-tile: the LDS-DMA path tops out at ~64 B/clk/CU and the real L2 access pattern delivers ~31); MFMA +
// reads 579, MFMA + DMA 588, reads + DMA 568.  The L2 -> LDS stream is the longest single leg; leading-dimension padding
// (channel conflicts), more DMA in flight (80 KiB) and a second counted wait changed nothing; other splits of the eight DMA
// pieces over the two load segments are slower than 2 + 6 (4 + 4: -3.5 %, 0 + 8: -1...-4 %);
// anything placed inside the MFMA segments costs: closing the segment's barrier 2-8 MFMAs early (so the partner starts while
// this wave finishes) -12 %, two of the six DMA pieces interleaved with the MFMAs -4 %.
// Work decode (flat 1-D grid over all batches) and TAIL SPLIT-K.  T = tiles of all batches (from the effective, device-side row
// counts).  The first full = floor(T / CUs) * CUs tiles are whole-K units, XCD-chunked so that the units co-resident on one XCD
// cover neighbouring tiles.  The remaining rem = T - full tiles would occupy rem of the CUs for a whole tile-time (Llama's
// N = 4096 GEMMs at 5112 rows: 320 tiles = 1.25 waves -> 62 % of the machine); instead each is cut into S = floor(CUs / rem)
// K-ranges that run side by side, so the tail costs 1/S tile-time.  A split unit writes its fp32 accumulators to the
// workspace in register order (coalesced 16-byte stores), takes a ticket, and the LAST arriver of a tile sums the S partials
// in ascending split order (its own from registers) -- a fixed order, so the result does not depend on the arrival order --
// and runs the normal epilogue.
constexpr int MAX_FLAT_BATCH = 8;

template <int ABL>
__global__ __launch_bounds__(NT2, 1) void gemm256v3_bf16_nt_kernel(GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int N = g.N;
  const int tiles_n = (N + BN2 - 1) / BN2;
  // ---- tile counts per batch (block-uniform scalar work)
  int T = 0;
  int pre[MAX_FLAT_BATCH];
#pragma unroll
  for (int b = 0; b < MAX_FLAT_BATCH; ++b) {
    pre[b] = T;
    if (b < g.nbatch) {
      const int Mb = g.m_dev ? min(g.M, g.m_dev[b * g.m_dev_stride]) : g.M;
      T += ((Mb + BM2 - 1) / BM2) * tiles_n;
    }
  }
  const int C = g.n_cu;
  const int full = (T / C) * C, rem = T - full;
  const int nt_all = g.K / BK2;
  int S = 1;
  if (g.ws && rem > 0) S = max(1, min(min(C / rem, g.max_split), nt_all / 4));
  int bid = blockIdx.x;
  if (bid >= full + rem * S) return;
  int flat, split = 0;
  if (bid < full) {
    const int q = full >> 3, xcd = bid & 7, loc = bid >> 3;      // full is a multiple of the CU count, hence of 8
    flat = xcd * q + loc;
  } else {
    const int r = bid - full;
    if (S == 1 && (rem & 7) == 0) {
      // an unsplit tail (rem > CUs / 2: Llama qkv 192, expert gate|up 184 tiles): XCD-chunked like the full waves, so the units of one
      // XCD are neighbours in the grouped tile order instead of every 8th tile (bid & 7 == r & 7: full is a multiple of 8)
      flat = full + (r & 7) * (rem >> 3) + (r >> 3);
    } else {
      flat = full + r % rem;
      split = r / rem;
    }
  }
  const bool is_split = (bid >= full) && S > 1;
  int batch = 0, pbase = 0;
#pragma unroll
  for (int b = 1; b < MAX_FLAT_BATCH; ++b) if (b < g.nbatch && flat >= pre[b]) { batch = b; pbase = pre[b]; }
  const int lid = flat - pbase;
  const bf16_t* __restrict__ A = g.A + batch * g.sA;
  const bf16_t* __restrict__ W = g.W + batch * g.sW;
  const int M = g.m_dev ? min(g.M, g.m_dev[batch * g.m_dev_stride]) : g.M;
  const int tiles_m = (M + BM2 - 1) / BM2;
  const int GROUP_M = g.group_m;
  const int per_group = GROUP_M * tiles_n;
  const int grp = lid / per_group;
  const int first_m = grp * GROUP_M;
  const int gsz = min(tiles_m - first_m, GROUP_M);
  const int tm = first_m + (lid % per_group) % gsz, tn = (lid % per_group) / gsz;
  const int m0 = tm * BM2, n0 = tn * BN2;
  // K range of this unit (in BK2 tiles)
  int kt0 = 0, nt = nt_all;
  if (is_split) {
    const int base = nt_all / S, extra = nt_all % S;
    kt0 = split * base + min(split, extra);
    nt = base + (split < extra ? 1 : 0);
  }

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int fr = lane & 15, fq = lane >> 4;

  // DMA instruction j of an operand covers rows 8j..8j+7.  A: wave w issues j = w + 8*i, so i = 0,2 are the quadrant-0 rows
  // (0-63 of each M half) and i = 1,3 the quadrant-1 rows.  B: j = 4*w + i.
  const int sub_row = lane >> 3;
  const int src_c = (lane & 7) ^ sub_row;
  const bf16_t* a_src[4];
  const bf16_t* w_src[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int arow = (wave + 8 * i) * 8 + sub_row;
    const int wrow = (wave * 4 + i) * 8 + sub_row;
    int ar = min(m0 + arow, M - 1);
    if (g.a_rows) ar = g.a_rows[batch * g.rows_stride + ar];        // gather: the dispatch is folded into the operand fetch
    a_src[i] = A + (int64_t)ar * g.lda + src_c * 8 + (int64_t)kt0 * BK2;
    w_src[i] = W + (int64_t)min(n0 + wrow, N - 1) * g.ldw + src_c * 8 + (int64_t)kt0 * BK2;
  }
  auto dma_a = [&](int i, int t) {
    if constexpr (ABL & 4) return;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a_src[i] + (int64_t)t * BK2),
                                     (__attribute__((address_space(3))) void*)(smem + (t & 1) * STAGE_BYTES + (wave + 8 * i) * 1024), 16, 0, 0);
  };
  auto dma_w = [&](int i, int t) {
    if constexpr (ABL & 4) return;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(w_src[i] + (int64_t)t * BK2),
                                     (__attribute__((address_space(3))) void*)(smem + (t & 1) * STAGE_BYTES + OP_BYTES + (wave * 4 + i) * 1024), 16, 0, 0);
  };

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8 fa[4][2], fb0[2][2], fb1[2][2];
  if constexpr (ABL & 2) {
#pragma unroll
    for (int i = 0; i < 4; ++i) for (int kk = 0; kk < 2; ++kk) for (int e = 0; e < 8; ++e) {
      fa[i][kk][e] = (bf16_t)(float)(lane + i);
      if (i < 2) { fb0[i][kk][e] = (bf16_t)(float)(lane - i); fb1[i][kk][e] = (bf16_t)(float)(lane - 2 * i); }
    }
  }
  auto load_a = [&](int qm, const char* st) {
    if constexpr (ABL & 2) return;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
        fa[i][kk] = *reinterpret_cast<const bf16x8*>(st + lds_off2(wr * 128 + (qm * 4 + i) * 16 + fr, kk * 4 + fq));
  };
#define MP_LOAD_B(FB, QN, ST)                                                                                \
  do {                                                                                                      \
    if constexpr (!(ABL & 2)) {                                                                             \
      _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                         \
        _Pragma("unroll") for (int kk = 0; kk < 2; ++kk)                                                    \
          FB[j][kk] = *reinterpret_cast<const bf16x8*>((ST) + OP_BYTES + lds_off2(wc * 64 + ((QN) * 2 + j) * 16 + fr, kk * 4 + fq)); \
    }                                                                                                       \
  } while (0)
// MFMAs [LO, HI) of a 32-MFMA segment: the 64 x 64 half QM of the wave tile, K = 64; column half QNX (fragments FBX) first
#define MP_MFMA_RANGE(QM, QNX, FBX, QNY, FBY, LO, HI)                                                       \
  _Pragma("unroll") for (int idx = (LO); idx < (HI); ++idx) {                                               \
    const int h = idx >> 4, kk = (idx >> 3) & 1, i = (idx >> 1) & 3, j = idx & 1;                           \
    const int qn = h ? (QNY) : (QNX);                                                                       \
    const bf16x8 bfrag = h ? FBY[j][kk] : FBX[j][kk];                                                       \
    if constexpr (ABL & 1) { asm volatile("" :: "v"(fa[i][kk]), "v"(bfrag)); }                              \
    else acc[(QM) * 4 + i][qn * 2 + j] =                                                                    \
        __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfrag, fa[i][kk], acc[(QM) * 4 + i][qn * 2 + j], 0, 0, 0);  \
  }
#define MP_MFMA_32(QM, QNX, FBX, QNY, FBY)                                                                  \
  do {                                                                                                      \
    __builtin_amdgcn_s_setprio(1);                                                                          \
    MP_MFMA_RANGE(QM, QNX, FBX, QNY, FBY, 0, 32)                                                            \
    __builtin_amdgcn_s_setprio(0);                                                                          \
    MP_BAR();                                                                                               \
  } while (0)
#define MP_LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

  // ---- prologue: all of tile 0 and what the (virtual) tile -1 would have issued for tile 1
  // (the W pieces first: with gathered A rows -- the experts' dispatch -- the A addresses wait for the row-index loads, and the W fetch
  //  can be in flight meanwhile)
#pragma unroll
  for (int i = 0; i < 4; ++i) dma_w(i, 0);
#pragma unroll
  for (int i = 0; i < 4; ++i) dma_a(i, 0);
  if (nt > 1) {
    dma_a(0, 1); dma_a(2, 1);
#pragma unroll
    for (int i = 0; i < 4; ++i) dma_w(i, 1);
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  MP_BAR();
  if (wr == 1) MP_BAR();

  for (int t = 0; t < nt; ++t) {
    const char* st = smem + (t & 1) * STAGE_BYTES;
    const bool n1 = (t + 1 < nt), n2 = (t + 2 < nt);
    // ---------------- H0: rows 0-63 of the wave tile x all 64 columns (32 MFMAs) ----------------
    MP_LOAD_B(fb0, 0, st);
    MP_LOAD_B(fb1, 1, st);
    load_a(0, st);
    __builtin_amdgcn_sched_barrier(0);                 // fragment reads first: the DMA issue then overlaps their latency
    if (n1) { dma_a(1, t + 1); dma_a(3, t + 1); }
    MP_LGKM0();
    MP_BAR();
    MP_MFMA_32(0, 0, fb0, 1, fb1);
    // ---------------- H1: rows 64-127 (32 MFMAs) ----------------
    load_a(1, st);
    __builtin_amdgcn_sched_barrier(0);
    if (n2) {
      dma_a(0, t + 2); dma_a(2, t + 2);
#pragma unroll
      for (int i = 0; i < 4; ++i) dma_w(i, t + 2);
      asm volatile("s_waitcnt vmcnt(6)" ::: "memory");     // everything up to the A rows 64-127 of tile t+1 has landed
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    MP_LGKM0();
    MP_BAR();
    MP_MFMA_32(1, 1, fb1, 0, fb0);
  }
  if (wr == 0) MP_BAR();
  if (is_split) {
    const int tail = flat - full;
    // Partials travel WRITE-THROUGH (16-byte buffer stores / loads with sc1: written to / read from the memory-side coherence point),
    // so no L2 write-back or invalidate is needed for the other XCDs to see them -- an agent-scope fence per wave costs an L2-wide
    // flush each and made the tail slower than the unsplit tile.  One accumulator fragment (f32x4) per store: 8-byte sc1 stores
    // (the first version: 64-bit relaxed atomics) cost 2.7 x the time per byte of 16-byte ones (MI355X_MICROARCH.md, stores table).
    typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
    constexpr int SLAB = BM2 * BN2 * 4;                  // one unit's partial tile, bytes
    float* tile_ws = g.ws + ((int64_t)tail * S) * (BM2 * BN2);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(tile_ws, 0, S * SLAB, 0x00020000);
    const int my_off = split * SLAB + tid * 16;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[i][j]), rs, my_off + (i * 4 + j) * (NT2 * 16), 0, 16);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // every partial of this wave has reached the coherence point
    __syncthreads();
    int* flag = reinterpret_cast<int*>(smem);
    if (tid == 0) *flag = __hip_atomic_fetch_add(g.tickets + tail, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const int ticket = *flag;
    __syncthreads();                                   // smem is reused by the epilogue
    if (ticket != S - 1) return;
    if (tid == 0) __hip_atomic_store(g.tickets + tail, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // self-resetting
    // sum the S partials in ascending split order (own one re-read from the workspace: the order, hence the rounding, is then
    // independent of which unit arrived last)
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int sp = 0; sp < S; ++sp) {
      const int base = sp * SLAB + tid * 16;
#pragma unroll
      for (int ip = 0; ip < 2; ++ip) {                 // 16 x 16-byte loads in flight, then their adds
        u32x4 t[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) t[q] = __builtin_amdgcn_raw_buffer_load_b128(rs, base + (ip * 16 + q) * (NT2 * 16), 0, 16);
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[ip * 4 + (q >> 2)][q & 3] += __builtin_bit_cast(f32x4, t[q]);
      }
    }
  }
  gemm256_epilogue_t(g, acc, batch, M, N, m0, n0, wr, wc, fr, fq);
}


}  // namespace

// split-K workspaces, registered by the host: >= n_cu * 256 KiB of fp32 partials + >= 384 tickets each (the 320-row kernel keeps three words per tail tile).  One workspace serves
// one stream at a time, so a stream that runs GEMMs concurrently with others registers its own (mp_gemm_set_stream_workspace);
// launches on any other stream of that device use the device's default entry (mp_gemm_set_workspace).  The table is keyed by
// (device, stream) with no cap on either; it is only a directory of caller-owned buffers (the library never allocates).
struct SplitWs { float* ws; int* tickets; int64_t bytes; };
static std::mutex g_split_mu;
static std::map<std::pair<int, hipStream_t>, SplitWs> g_split;          // stream == nullptr: the device's default entry

static int current_device() {
  int dev = 0;
  (void)hipGetDevice(&dev);
  return dev;
}

static int register_split_ws(hipStream_t stream, void* ws, int64_t ws_bytes, int* tickets) {
  std::lock_guard<std::mutex> lk(g_split_mu);
  const auto key = std::make_pair(current_device(), stream);
  if (ws) g_split[key] = SplitWs{(float*)ws, tickets, ws_bytes};
  else g_split.erase(key);
  return MP_OK;
}

extern "C" int mp_gemm_set_workspace(void* ws, int64_t ws_bytes, int* tickets, int n_tickets) {
  MP_REQUIRE(ws == nullptr || (tickets != nullptr && n_tickets >= 384), MP_ERR_ARG, "mp_gemm_set_workspace: need >= 384 zeroed int tickets");
  return register_split_ws(nullptr, ws, ws_bytes, tickets);
}

extern "C" int mp_gemm_set_stream_workspace(hipStream_t stream, void* ws, int64_t ws_bytes, int* tickets, int n_tickets) {
  MP_REQUIRE(ws == nullptr || (tickets != nullptr && n_tickets >= 384), MP_ERR_ARG, "mp_gemm_set_stream_workspace: need >= 384 zeroed int tickets");
  MP_REQUIRE(stream != nullptr, MP_ERR_ARG, "mp_gemm_set_stream_workspace: the default entry is mp_gemm_set_workspace");
  return register_split_ws(stream, ws, ws_bytes, tickets);
}

void mp_gemm_split_workspace(hipStream_t stream, float** ws, int** tickets, int64_t* bytes) {
  std::lock_guard<std::mutex> lk(g_split_mu);
  const int dev = current_device();
  auto it = g_split.find(std::make_pair(dev, stream));
  if (it == g_split.end()) it = g_split.find(std::make_pair(dev, (hipStream_t) nullptr));
  if (it == g_split.end()) { *ws = nullptr; *tickets = nullptr; *bytes = 0; return; }
  *ws = it->second.ws; *tickets = it->second.tickets; *bytes = it->second.bytes;
}

// Whether `stream` has its own registered workspace, i.e. the host declared that it runs GEMMs CONCURRENTLY with the device's primary
// stream.  The 320-row kernel's cooperative tail (units that WAIT for their siblings) is only deadlock-free while a single kernel on
// the device waits at a time: two such kernels on two streams can each hold CUs the other's missing units need.
bool mp_gemm_stream_registered(hipStream_t stream) {
  if (!stream) return false;
  std::lock_guard<std::mutex> lk(g_split_mu);
  return g_split.find(std::make_pair(current_device(), stream)) != g_split.end();
}

// CU count of the CURRENT device (immutable per device; cached per device id, not in a process-wide static)
int mp_device_cus() {
  static std::mutex mu;
  static std::map<int, int> cus;
  const int dev = current_device();
  std::lock_guard<std::mutex> lk(mu);
  auto it = cus.find(dev);
  if (it != cus.end()) return it->second;
  int n = 0;
  if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
  cus[dev] = n;
  return n;
}

int mp_launch_gemm256(const GemmArgs& g, int batch, hipStream_t stream) {
  static int abl = -1;
  if (abl < 0) {
    const char* e = getenv("MP_GEMM_ABLATE");          // scripts/gemm_ab.py only: 1 = no MFMA, 2 = no LDS fragment reads, 4 = no DMA
    abl = e ? atoi(e) : 0;
    if (abl != 1 && abl != 2 && abl != 4) abl = 0;
#define MP_ATTR(A) (void)hipFuncSetAttribute((const void*)gemm256v3_bf16_nt_kernel<A>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE_BYTES)
    MP_ATTR(0); MP_ATTR(1); MP_ATTR(2); MP_ATTR(4);
#undef MP_ATTR
  }
  MP_REQUIRE(batch <= MAX_FLAT_BATCH, MP_ERR_SHAPE, "256x256 GEMM: at most %d batches per launch (got %d)", MAX_FLAT_BATCH, batch);
  const int tiles = (int)(mp_cdiv(g.M, BM2) * mp_cdiv(g.N, BN2));
  const dim3 blk(NT2);
  const int n_cu = std::min(mp_device_cus(), 256);   // workspace / ticket sizing
  static int max_split = -1;                         // environment knob, read once (immutable afterwards)
  if (max_split < 0) {
    const char* e = getenv("MP_GEMM_MAX_SPLIT");     // 1 = no tail split (A/B), default 8
    max_split = (e && atoi(e) >= 1) ? atoi(e) : 8;
  }
  GemmArgs gf = g;
  static int ep8 = -1;
  if (ep8 < 0) { const char* e = getenv("MP_GEMM_EP8"); ep8 = (e && atoi(e) == 1) ? 1 : 0; }
  gf.n_cu = n_cu; gf.nbatch = batch; gf.max_split = max_split; gf.ep8 = ep8;
  int64_t ws_bytes = 0;
  mp_gemm_split_workspace(stream, &gf.ws, &gf.tickets, &ws_bytes);
  if (!gf.ws || ws_bytes < (int64_t)n_cu * BM2 * BN2 * 4 || gf.out_f32) { gf.ws = nullptr; gf.tickets = nullptr; gf.max_split = 1; }
  const dim3 fgrid(tiles * batch + (gf.max_split > 1 ? n_cu : 0));   // surplus workgroups (the units of a split tail; device-side row counts leave more) exit at once
#define MP_GO(A) hipLaunchKernelGGL((gemm256v3_bf16_nt_kernel<A>), fgrid, blk, 2 * STAGE_BYTES, stream, gf)
  switch (abl) {
    case 1: MP_GO(1); break;
    case 2: MP_GO(2); break;
    case 4: MP_GO(4); break;
    default: MP_GO(0);
  }
#undef MP_GO
  return mp_check_launch("mp_gemm_bf16_nt(256)");
}
