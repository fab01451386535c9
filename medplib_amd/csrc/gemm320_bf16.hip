// 320x256x64 bf16 MFMA GEMM for gfx950: the 256x256 ping-pong kernel's schedule (gemm256_bf16.hip) on a taller tile, for the shapes whose
// 256-row tiling leaves a short last wave.  The decoder runs M = batch x sequence = 8 x 639 = 5112 rows: 20 row tiles of 256 x 16 column
// tiles = 320 tiles = 1.25 waves on 256 CUs for every N = 4096 projection (o_proj, down_proj, three of the four dgrad GEMMs of LoRA training),
// 3.75 waves for qkv; 16 row tiles of 320 make that exactly 1 and 3 waves of 1.25 x the work each, without the split-K tail's fp32
// partials going through memory (64 MB written + 48 MB read per N = 4096 launch).  mp_gemm_bf16_nt picks this kernel per call from the
// two wave counts (use_320 in gemm_bf16.hip).
//
// 512 threads = 8 waves as 4 (M) x 2 (N); a wave owns an 80 x 128 output tile = 5 x 8 fragments of v_mfma_f32_16x16x32_bf16 (160
// accumulator registers).  LDS: 2 stages x (A 40 KiB + B 32 KiB) = 144 KiB, one workgroup per CU, two waves per SIMD.  The two N halves
// (waves 0-3 / 4-7, one of each per SIMD) run one barrier out of phase like the 256x256 kernel's M halves, so one wave of a SIMD is in
// a 40-MFMA segment while its partner reads fragments and issues DMA.  A K-tile is two segments = the two 64-column halves of the wave
// tile; the A fragments (all 80 rows) stay in registers for both, the B fragments are read per half:
//   reads:  H0  A (5 fragments x 2) + B columns 0-63 (4 x 2)        H1  B columns 64-127 (4 x 2)
//   DMA  :  H0  B columns 64-127 of tile t+1 (2 pieces per wave)     H1  A (5 pieces) + B columns 0-63 (2) of tile t+2
// (A and the B columns 0-63 of a stage are free after H0's reads, the B columns 64-127 after H1's) -- the split along N rather than
// M makes the DMA piece counts integral (40 A pieces of 8 rows over 8 waves; a split along M would need 2.5 per wave and region).
// (Issuing two of the five A pieces one segment later -- 4 + 5 DMA pieces per K-tile instead of 2 + 7, the load segments' fragment reads being 18
// + 8 -- measured the same on every shape: the DMA issue split is not what holds this tile ~7 % below the 256x256 one per flop.)
// One counted wait (vmcnt(7)) per K-tile; the operand stream never drains.  The DMA goes through buffer descriptors (buffer_load ... lds:
// one 32-bit lane offset per operand, everything else in the scalar offset): 2 address registers instead of 18 -- with 160 accumulators,
// 40 A-fragment and 32 B-fragment registers there is no room for per-piece 64-bit pointers (the launcher checks the operands fit 2 GiB).
// The MFMAs are issued as (W fragment, A fragment) = transposed accumulators (a lane owns four consecutive columns of a row); the
// epilogue pairs neighbouring fragments with v_permlane16_swap into 16-byte pieces (gemm_common.h, pair_swap16).
// Round 2, late: ONE INSTANTIATION PER EPILOGUE FAMILY (plain / RoPE / SwiGLU / MoE combine).  With the RoPE and the plain store paths in
// one function the register allocator spilled 362 VGPRs and the plain path wrote and re-read its accumulators through scratch: a K sweep
// at one wave of tiles (scripts/gemm_ksweep.py) showed 35 us of fixed cost per launch, 28 of them the epilogue (K = 128: 38.1 us with, 10.8
// without it); each family alone compiles to 241-253 VGPRs with no spill and the fixed cost is 10 us -- below the 256x256 kernel's 14.
// The kernel then beat the 256-row tiling on every dense decoder shape (o_proj 131 vs 180 us, down 301 vs 362, qkv+RoPE 355 vs 383, CLIP
// fc1 43.5 vs 72 on the 128x128 kernel), so the selection model in gemm_bf16.hip was recalibrated from the K sweeps.
// Batched expert calls too: flat tile decode over the experts' device-side row counts, the MoE dispatch folded into the A fetch (one
// 32-bit lane offset per DMA piece instead of one per operand: 5 registers), SwiGLU pairing and the combine scatter as epilogue families,
// and the 256x256 kernel's tail split-K (batched calls only; dense calls keep whole waves and one accumulation order per shape).
// bf16 output only.
#include "gemm_common.h"
#include <stdlib.h>
#include <algorithm>

int mp_device_cus();
void mp_gemm_split_workspace(hipStream_t stream, float** ws, int** tickets, int64_t* bytes);
bool mp_gemm_stream_registered(hipStream_t stream);
bool mp_gemm_policy_whole_tiles();
long long mp_gemm_tail_wait_value();

namespace {

constexpr int BM3 = 320, BN3 = 256, BK3 = 64, NT3 = 512;
constexpr int A_BYTES = BM3 * 128, B_BYTES = BN3 * 128, STAGE3 = A_BYTES + B_BYTES;     // 40960 + 32768 = 73728

__device__ __forceinline__ int lds_off3(int r, int c) { return r * 128 + ((c ^ (r & 7)) << 4); }

#define MP3_BAR()                          \
  do {                                     \
    asm volatile("" ::: "memory");         \
    __builtin_amdgcn_sched_barrier(0);     \
    __builtin_amdgcn_s_barrier();          \
    __builtin_amdgcn_sched_barrier(0);     \
    asm volatile("" ::: "memory");         \
  } while (0)
#define MP3_LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

// Epilogue: alpha / bias / activation (rounded to bf16) then the residual add, or the RoPE pairing of a q / k tile (same arithmetic and
// rounding points as gemm256_epilogue_t; 16-byte pieces only: the launcher guarantees the alignment).
// ROPE: the kernel is instantiated once per epilogue family -- with both in one function the register allocator sized the epilogue for the
// RoPE path and spilled 362 VGPRs; the PLAIN path then wrote and re-read its accumulators through scratch (K sweep at 256 tiles: 35 us of
// fixed cost per launch, 28 of them the epilogue).
// EPI: 0 = alpha / bias / QuickGELU / residual, 1 = RoPE pairing (fused qkv), 2 = SwiGLU pairing (gate|up), 3 = MoE combine (row scatter)
constexpr int EPI_PLAIN = 0, EPI_ROPE = 1, EPI_SWIGLU = 2, EPI_COMBINE = 3, EPI_RELU = 4, EPI_GELU = 5;
// EPI_RELU / EPI_GELU (round 6): the PLAIN family with ONE other activation sweep each — what the SAM-Med2D encoder's GEMMs need (Adapter.spatial:
// ReLU, mlp.lin1: erf-GELU).  One sweep per instantiation: with two or three in one function the allocator spills ~300 VGPRs, each alone none.
// `items`: which of the wave tile's 20 (fragment row i, fragment pair) items this call finishes — bit i * 4 + pair, pair = 2 consecutive
// fragments = 32 columns; all twenty for an unsplit tile, a unit's share of them in the cooperative fix-up of a split tail tile (every
// unit reduces and stores a share).  The SwiGLU family pairs fragments j and j + 2, so its items come as the two pairs of a column half.
template <int EPI>
__device__ __forceinline__ void gemm320_epilogue(const GemmArgs& g, f32x4 (&acc)[5][8], int batch, int M, int N, int m0, int n0, int wr, int wc,
                                                 int fr, int fq, unsigned items = 0xfffffu) {
  bf16_t* Cb = reinterpret_cast<bf16_t*>(g.C) + batch * g.sC;
  const int cw = n0 + wc * 128;
  const int c8 = pair_col8(fq);                             // the lane's eight columns inside a fragment pair's 32 (gemm_common.h)
  if constexpr (EPI == EPI_SWIGLU) {
    // W rows are [gate 0..31 | up 0..31 | gate 32..63 | up 32..63 | ...]: of the wave's 8 fragments 0,1 / 4,5 are gate columns and 2,3 / 6,7
    // the matching up columns; gate and up are rounded to bf16 first, like the unfused GEMM + SwiGLU kernel pair (and gemm256_epilogue_t)
    // folded post-attention norm: rstd of the token behind A row `row` (null: 1.f) multiplies the fp32 accumulators before their bf16 rounding
    float rsc[5] = {1.f, 1.f, 1.f, 1.f, 1.f};
    if (g.a_scale) {
#pragma unroll
      for (int i = 0; i < 5; ++i) {
        const int row = m0 + wr * 80 + i * 16 + fr;
        if (row < M) rsc[i] = g.a_scale[g.a_rows ? g.a_rows[batch * g.rows_stride + row] : row];
      }
    }
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      const int row = m0 + wr * 80 + i * 16 + fr;
#pragma unroll
      for (int jb = 0; jb < 2; ++jb) {
        if (!((items >> (i * 4 + jb * 2)) & 1u)) continue;
        bf16x4 o[2];
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            acc[i][jb * 4 + jj][r] *= rsc[i];
            acc[i][jb * 4 + jj + 2][r] *= rsc[i];
            const float gf = (float)(bf16_t)acc[i][jb * 4 + jj][r];
            const float uf = (float)(bf16_t)acc[i][jb * 4 + jj + 2][r];
            o[jj][r] = (bf16_t)(gf * mp_sigmoid_fast(gf) * uf);
          }
        const bf16x8 p = pair_swap16(o[0], o[1]);
        if (row < M) *reinterpret_cast<bf16x8*>(Cb + (int64_t)row * g.ldc + (cw >> 1) + jb * 32 + c8) = p;
        if (g.keep_gu) {                                   // the bf16 gate / up values the product was formed from, in the GEMM's own column order (training keeps them)
          const bf16x8 pg = pair_swap16(round4(acc[i][jb * 4]), round4(acc[i][jb * 4 + 1])), pu = pair_swap16(round4(acc[i][jb * 4 + 2]), round4(acc[i][jb * 4 + 3]));
          if (row < M) {
            *reinterpret_cast<bf16x8*>(g.keep_gu + (int64_t)row * g.ld_gu + cw + jb * 64 + c8) = pg;
            *reinterpret_cast<bf16x8*>(g.keep_gu + (int64_t)row * g.ld_gu + cw + jb * 64 + 32 + c8) = pu;
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    return;
  }
  if constexpr (EPI == EPI_COMBINE) {
    // combine folded into the epilogue (top-1 MoE down projection): out[token] = residual[token] + weight[token] * bf16(acc), the rounding
    // points of the separate combine kernel (and of gemm256_epilogue_t).  Row indices and weights of the wave tile first, then per column
    // half all residual pieces before the first is used.
    bf16_t* Cs = reinterpret_cast<bf16_t*>(g.C);          // the shared [tokens, N] output
    int orow[5];
    float sc[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      const int row = m0 + wr * 80 + i * 16 + fr;
      orow[i] = row < M ? g.c_rows[batch * g.rows_stride + row] : -1;
    }
#pragma unroll
    for (int i = 0; i < 5; ++i) sc[i] = (orow[i] >= 0 && g.c_scale) ? g.c_scale[orow[i]] : 1.f;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      bf16x8 rv[5][2];
#pragma unroll
      for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int jp = 0; jp < 2; ++jp) {
          const int col = cw + (h * 2 + jp) * 32 + c8;
          rv[i][jp] = bf16x8{};
          if (g.residual && orow[i] >= 0 && ((items >> (i * 4 + h * 2 + jp)) & 1u))
            rv[i][jp] = *reinterpret_cast<const bf16x8*>(g.residual + (int64_t)orow[i] * g.ldr + col);
        }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int jp = 0; jp < 2; ++jp) {
          const int col = cw + (h * 2 + jp) * 32 + c8;
          const bf16x8 p = pair_swap16(round4(acc[i][(h * 2 + jp) * 2]), round4(acc[i][(h * 2 + jp) * 2 + 1]));
          bf16x8 o;
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            float v = (float)p[e] * sc[i];
            if (g.residual) v += (float)rv[i][jp][e];
            o[e] = (bf16_t)v;
          }
          if (orow[i] >= 0 && ((items >> (i * 4 + h * 2 + jp)) & 1u)) *reinterpret_cast<bf16x8*>(Cs + (int64_t)orow[i] * g.ldc + col) = o;
        }
      __builtin_amdgcn_sched_barrier(0);
    }
    return;
  }
  if constexpr (EPI == EPI_ROPE) {
    // mp_gemm_qkv_rope_bf16 (no bias / residual / alpha).  q and k tiles: the wave's 128 columns are one head, fragments 0,1 | 2,3 hold
    // [lo 0..31 | hi 0..31], fragments 4,5 | 6,7 [lo 32..63 | hi 32..63].  v tiles (n0 >= 2N/3) go through the SAME loop with the rotation
    // switched off and the standard column order -- a second, plain store path beside this one in the same function made the register
    // allocator spill 212 VGPRs (~28 us per wave of tiles), each path alone spills none.
    // One fragment row (16 tile rows) at a time; the cos / sin pieces of the NEXT row are requested before this row's arithmetic, so the
    // five table round trips overlap the work and only 64 registers sit beside the accumulators.
    const bool is_v = n0 >= (N / 3) * 2;                  // wave-uniform
    const int head0 = (cw >> 7) << 7;
    f32x4 cs[2][2][2], sn[2][2][2];                       // [buffer][jb][j]
    float rsc[2] = {1.f, 1.f};                            // folded input norm: rstd of the fragment row's token (null: 1.f)
    auto fetch = [&](int i, int buf) {
      const int row = m0 + wr * 80 + i * 16 + fr;
      const int pos = (row < M ? row : 0) % g.rope_seq + g.rope_pos0;
      if (g.a_scale) rsc[buf] = g.a_scale[row < M ? row : 0];
#pragma unroll
      for (int jb = 0; jb < 2; ++jb)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
          const int dim = jb * 32 + jj * 16 + fq * 4;
          cs[buf][jb][jj] = *reinterpret_cast<const f32x4*>(g.rope_cos + (int64_t)pos * 64 + dim);
          sn[buf][jb][jj] = *reinterpret_cast<const f32x4*>(g.rope_sin + (int64_t)pos * 64 + dim);
        }
    };
    fetch(0, 0);
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      if (i + 1 < 5) fetch(i + 1, (i + 1) & 1);
      __builtin_amdgcn_sched_barrier(0);
      const int row = m0 + wr * 80 + i * 16 + fr;
#pragma unroll
      for (int jb = 0; jb < 2; ++jb) {
        bf16x4 olo[2], ohi[2];
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
          const bf16x4 qa = round4(acc[i][jb * 4 + jj] * rsc[i & 1]), qb = round4(acc[i][jb * 4 + jj + 2] * rsc[i & 1]);   // the projection output's own rounding point
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float a = (float)qa[r], b = (float)qb[r];
            const bf16_t rlo = (bf16_t)(a * cs[i & 1][jb][jj][r] - b * sn[i & 1][jb][jj][r]);
            const bf16_t rhi = (bf16_t)(b * cs[i & 1][jb][jj][r] + a * sn[i & 1][jb][jj][r]);
            olo[jj][r] = is_v ? qa[r] : rlo;
            ohi[jj][r] = is_v ? qb[r] : rhi;
          }
        }
        const bf16x8 plo = pair_swap16(olo[0], olo[1]), phi = pair_swap16(ohi[0], ohi[1]);
        // q / k: lo half at head0 + jb*32, hi half 64 further; v: fragments 4jb, 4jb+1 are columns cw + 64jb .. +31, fragments 4jb+2, +3 the next 32
        const int col_lo = (is_v ? cw + jb * 64 : head0 + jb * 32) + c8;
        const int col_hi = (is_v ? cw + jb * 64 + 32 : head0 + 64 + jb * 32) + c8;
        if (row < M) {
          *reinterpret_cast<bf16x8*>(Cb + (int64_t)row * g.ldc + col_lo) = plo;
          *reinterpret_cast<bf16x8*>(Cb + (int64_t)row * g.ldc + col_hi) = phi;
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    return;
  }
  // ---- alpha, bias, activation in place (fp32)
  const float* bias0 = g.bias ? g.bias + batch * g.sBias : nullptr;
  if (g.alpha != 1.f || bias0) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      f32x4 bv = {0.f, 0.f, 0.f, 0.f};
      if (bias0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) { const int col = cw + j * 16 + fq * 4 + r; bv[r] = col < N ? bias0[col] : 0.f; }
      }
#pragma unroll
      for (int i = 0; i < 5; ++i) acc[i][j] = acc[i][j] * g.alpha + bv;
    }
  }
#define MP3_ACT_SWEEP(EXPR)                                                          \
  _Pragma("unroll") for (int i = 0; i < 5; ++i) _Pragma("unroll") for (int j = 0; j < 8; ++j) {                    \
    _Pragma("unroll") for (int r = 0; r < 4; ++r) { const float v = acc[i][j][r]; acc[i][j][r] = (EXPR); }         \
    __builtin_amdgcn_sched_barrier(0);                                                                             \
  }
  if constexpr (EPI == EPI_RELU) { MP3_ACT_SWEEP(fmaxf(v, 0.f)) }
  else if constexpr (EPI == EPI_GELU) { MP3_ACT_SWEEP(gelu_erf_fast(v)) }
  else if (g.act == ACT_QUICK_GELU) { MP3_ACT_SWEEP(v * mp_sigmoid_fast(v, 1.702f)) }
#undef MP3_ACT_SWEEP
  // ---- stores, in two column halves of the wave tile (the residual pieces of a half are all requested before the first is used)
  const bf16_t* R = g.residual ? g.residual + batch * g.sR : nullptr;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    bf16x8 rv[5][2];
    if (R) {
#pragma unroll
      for (int i = 0; i < 5; ++i) {
        const int row = m0 + wr * 80 + i * 16 + fr;
#pragma unroll
        for (int jp = 0; jp < 2; ++jp) {
          const int col = cw + (h * 2 + jp) * 32 + c8;
          rv[i][jp] = bf16x8{};
          if (row < M && col < N && ((items >> (i * 4 + h * 2 + jp)) & 1u)) rv[i][jp] = *reinterpret_cast<const bf16x8*>(R + (int64_t)row * g.ldr + col);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      const int row = m0 + wr * 80 + i * 16 + fr;
#pragma unroll
      for (int jp = 0; jp < 2; ++jp) {
        const int col = cw + (h * 2 + jp) * 32 + c8;
        bf16x8 p = pair_swap16(round4(acc[i][(h * 2 + jp) * 2]), round4(acc[i][(h * 2 + jp) * 2 + 1]));
        if (R) {
#pragma unroll
          for (int e = 0; e < 8; ++e) p[e] = (bf16_t)((float)p[e] + (float)rv[i][jp][e]);
        }
        if (row < M && col < N && ((items >> (i * 4 + h * 2 + jp)) & 1u)) *reinterpret_cast<bf16x8*>(Cb + (int64_t)row * g.ldc + col) = p;
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

constexpr int MAX_FLAT_BATCH3 = 8;
template <int EPI>
__global__ __launch_bounds__(NT3, 1) void gemm320_bf16_nt_kernel(GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int N = g.N;
  const int tiles_n = N / BN3;
  // ---- tile counts per batch from the effective (device-side) row counts; the grid is sized for the host-side bound, surplus workgroups exit
  int T = 0;
  int pre[MAX_FLAT_BATCH3];
#pragma unroll
  for (int b = 0; b < MAX_FLAT_BATCH3; ++b) {
    pre[b] = T;
    if (b < g.nbatch) {
      const int Mb = g.m_dev ? min(g.M, g.m_dev[b * g.m_dev_stride]) : g.M;
      T += ((Mb + BM3 - 1) / BM3) * tiles_n;
    }
  }
  // ---- work decode.  The first full = floor(T / CUs) * CUs tiles are whole-K units, XCD-chunked (workgroup b runs on XCD b & 7: an XCD
  // walks a contiguous chunk of the grouped tile order).  TAIL SPLIT-K as in the 256x256 kernel: when the remaining rem tiles would occupy
  // at most half the CUs for a whole tile-time, each is cut into S K-ranges (S * rem <= CUs units); the fp32 partials meet in the registered
  // workspace (write-through 16-byte stores, a ticket per tile, the last arriver sums them in split order and runs the epilogue).
  const int C = g.n_cu;
  const int full = (T / C) * C, rem = T - full;
  const int nt_all = g.K / BK3;
  int S = 1;
  if (g.ws && rem > 0 && rem * 2 <= C) S = max(1, min(min(min(C / rem, g.max_split), nt_all / 4), EPI == EPI_SWIGLU ? 10 : 20));   // <= the fix-up items of a tile
  const int bid = blockIdx.x;
  if (bid >= full + rem * S) return;
  int flat, split = 0;
  if (bid < full) {
    flat = (bid & 7) * (full >> 3) + (bid >> 3);            // full is a multiple of the CU count, hence of 8
  } else {
    const int r = bid - full;
    if (S == 1 && (rem & 7) == 0) {
      flat = full + (r & 7) * (rem >> 3) + (r >> 3);        // an unsplit tail: XCD-chunked like the full waves
    } else {
      flat = full + r % rem;
      split = r / rem;
    }
  }
  const bool is_split = (bid >= full) && S > 1;
  int batch = 0, pbase = 0;
#pragma unroll
  for (int b = 1; b < MAX_FLAT_BATCH3; ++b) if (b < g.nbatch && flat >= pre[b]) { batch = b; pbase = pre[b]; }
  const int lid = flat - pbase;
  const int M = g.m_dev ? min(g.M, g.m_dev[batch * g.m_dev_stride]) : g.M;
  const int tiles_m = (M + BM3 - 1) / BM3;
  const int GROUP_M = g.group_m;
  const int per_group = GROUP_M * tiles_n;
  const int grp = lid / per_group;
  const int first_m = grp * GROUP_M;
  const int gsz = min(tiles_m - first_m, GROUP_M);
  const int tm = first_m + (lid % per_group) % gsz, tn = (lid % per_group) / gsz;
  const int m0 = tm * BM3, n0 = tn * BN3;
  int kt0 = 0, nt = nt_all;                               // K range of this unit (in BK3 tiles)
  if (is_split) {
    const int base = nt_all / S, extra = nt_all % S;
    kt0 = split * base + min(split, extra);
    nt = base + (split < extra ? 1 : 0);
  }
  const int k_byte0 = kt0 * (BK3 * 2);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wc = wave >> 2, wr = wave & 3;               // wave group = N half (waves w and w + 4 share a SIMD)
  const int fr = lane & 15, fq = lane >> 4;

  // DMA piece p of an operand covers its rows 8p..8p+7 (LDS image: p * 1024 + lane * 16, XOR swizzle on the source chunk).
  //   A: 40 pieces, wave w issues p = w + 8 i (i < 5)
  //   B: columns 0-63 of both wave groups = rows {0-63, 128-191} = pieces {0-7, 16-23}; columns 64-127 = pieces {8-15, 24-31};
  //      wave w issues the entries 2w, 2w+1 of each list
  const int sub_row = lane >> 3;
  const int src_c = (lane & 7) ^ sub_row;
  // Buffer-descriptor DMA: ONE lane offset per operand (the lane's row within a piece + its swizzled 16-byte chunk); the piece's first
  // row, the tile origin and the K advance are a scalar offset.  Rows beyond M are beyond the descriptor's range and read as zeros (no
  // clamping, so the A pieces are affine in i); N is a multiple of 256 (launcher), so no W row is out of range.
  // A: one 32-bit lane offset PER PIECE (the row of the piece this lane fetches + its swizzled chunk): with the MoE dispatch folded into
  // the operand fetch (a_rows) the rows of a piece are arbitrary rows of the shared activation matrix; only the K advance is scalar.
  const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(g.A + batch * g.sA), 0,
                                          g.a_rows ? 0x7ffffff0 : (int)(((int64_t)(M - 1) * g.lda + g.K) * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(g.W + batch * g.sW), 0, (int)(((int64_t)(N - 1) * g.ldw + g.K) * 2), 0x00020000);
  const int w_lane = (int)((sub_row * g.ldw + src_c * 8) * 2);
  const int w_row_bytes = (int)(g.ldw * 2);
  int a_off[5];
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    int row = m0 + (wave + 8 * i) * 8 + sub_row;                  // rows beyond M: past the descriptor's range (zeros), or any valid row (gather)
    if (g.a_rows) row = g.a_rows[batch * g.rows_stride + min(row, M - 1)];
    a_off[i] = (int)(((int64_t)row * g.lda + src_c * 8) * 2);
  }
  int b_piece[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int e = 2 * wave + (k & 1);                    // entry 0..15 of the list
    b_piece[k] = (e & 7) + (e >> 3) * 16 + (k >> 1) * 8; // k = 0,1: columns 0-63 (region 0); k = 2,3: columns 64-127 (region 1)
  }
  auto dma_a = [&](int i, int t) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, (__attribute__((address_space(3))) void*)(smem + (t & 1) * STAGE3 + (wave + 8 * i) * 1024), 16,
                                             a_off[i], k_byte0 + t * (BK3 * 2), 0, 0);
  };
  auto dma_b = [&](int k, int t) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (__attribute__((address_space(3))) void*)(smem + (t & 1) * STAGE3 + A_BYTES + b_piece[k] * 1024), 16,
                                             w_lane, (n0 + b_piece[k] * 8) * w_row_bytes + k_byte0 + t * (BK3 * 2), 0, 0);
  };

  f32x4 acc[5][8];
#pragma unroll
  for (int i = 0; i < 5; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8 fa[5][2], fb[4][2];
  // fragment read addresses: one lane base per operand and K half (row wr * 80 + fr resp. wc * 128 + fr; (row & 7) == (fr & 7) for every
  // fragment row, so the swizzle term is the lane's own), everything else is an immediate
  const int a_rd0 = (wr * 80 + fr) * 128 + ((fq ^ (fr & 7)) << 4), a_rd1 = (wr * 80 + fr) * 128 + (((4 + fq) ^ (fr & 7)) << 4);
  const int b_rd0 = A_BYTES + (wc * 128 + fr) * 128 + ((fq ^ (fr & 7)) << 4), b_rd1 = A_BYTES + (wc * 128 + fr) * 128 + (((4 + fq) ^ (fr & 7)) << 4);
  auto load_a = [&](const char* st) {
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      fa[i][0] = *reinterpret_cast<const bf16x8*>(st + a_rd0 + i * 2048);
      fa[i][1] = *reinterpret_cast<const bf16x8*>(st + a_rd1 + i * 2048);
    }
  };
  auto load_b = [&](int hn, const char* st) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      fb[j][0] = *reinterpret_cast<const bf16x8*>(st + b_rd0 + (hn * 64 + j * 16) * 128);
      fb[j][1] = *reinterpret_cast<const bf16x8*>(st + b_rd1 + (hn * 64 + j * 16) * 128);
    }
  };
// the 40 MFMAs of one segment: column half HN of the wave tile, K = 64
// MP3_PRIO (A/B builds, scripts/r04_prio_ab.sh): 0 = priority raised around every MFMA segment (the shipped schedule), 1 = no priority
// changes, 2 = the second-dispatched N half (waves 4-7) at priority 1 for the whole K loop, no per-segment flips (MI355X_MICROARCH.md,
// "static priority for the younger half")
#ifndef MP3_PRIO
#define MP3_PRIO 0
#endif
#define MP3_MFMA_40(HN)                                                                                     \
  do {                                                                                                      \
    if (MP3_PRIO == 0) __builtin_amdgcn_s_setprio(1);                                                       \
    _Pragma("unroll") for (int kk = 0; kk < 2; ++kk)                                                        \
      _Pragma("unroll") for (int i = 0; i < 5; ++i)                                                         \
        _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                       \
          acc[i][(HN) * 4 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j][kk], fa[i][kk], acc[i][(HN) * 4 + j], 0, 0, 0); \
    if (MP3_PRIO == 0) __builtin_amdgcn_s_setprio(0);                                                       \
    MP3_BAR();                                                                                              \
  } while (0)

  // ---- prologue: all of tile 0, and what the (virtual) tile -1 would have issued for tile 1 (A + B columns 0-63)
#pragma unroll
  for (int i = 0; i < 5; ++i) dma_a(i, 0);
#pragma unroll
  for (int k = 0; k < 4; ++k) dma_b(k, 0);
  if (nt > 1) {
#pragma unroll
    for (int i = 0; i < 5; ++i) dma_a(i, 1);
    dma_b(0, 1); dma_b(1, 1);
    asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  MP3_BAR();
  if (wc == 1) MP3_BAR();
  if (MP3_PRIO == 2 && wc == 1) __builtin_amdgcn_s_setprio(1);

  for (int t = 0; t < nt; ++t) {
    const char* st = smem + (t & 1) * STAGE3;
    const bool n1 = (t + 1 < nt), n2 = (t + 2 < nt);
    // ---------------- H0: all 80 rows x columns 0-63 of the wave tile (40 MFMAs) ----------------
    load_b(0, st);
    load_a(st);
    __builtin_amdgcn_sched_barrier(0);
    if (n1) { dma_b(2, t + 1); dma_b(3, t + 1); }
    MP3_LGKM0();
    MP3_BAR();
    MP3_MFMA_40(0);
    // ---------------- H1: columns 64-127 (40 MFMAs) ----------------
    load_b(1, st);
    __builtin_amdgcn_sched_barrier(0);
    if (n2) {
#pragma unroll
      for (int i = 0; i < 5; ++i) dma_a(i, t + 2);
      dma_b(0, t + 2); dma_b(1, t + 2);
      asm volatile("s_waitcnt vmcnt(7)" ::: "memory");     // everything up to the B columns 64-127 of tile t+1 has landed
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    MP3_LGKM0();
    MP3_BAR();
    MP3_MFMA_40(1);
  }
  if (wc == 0) MP3_BAR();
  if (MP3_PRIO == 2) __builtin_amdgcn_s_setprio(0);
  unsigned items = 0xfffffu;
  if (is_split) {
    // COOPERATIVE FIX-UP of a split tail tile (round 3; before: the last arriver summed all S partials alone — S x 320 KiB through one
    // CU, 15-20 us in which the other S - 1 CUs of the tile had nothing left to do).  Every unit stores its fp32 partial in register
    // order (write-through 16-byte stores: they reach the memory-side coherence point, no L2 flush is needed for the other XCDs to see
    // them), announces itself on the tile's arrival counter and waits for the other S - 1 (they are all resident or about to be: the
    // units of a tile are consecutive workgroups of the launch's last wave, and nothing they could wait for depends on them).  Then
    // unit `split` sums, in ascending split order whoever arrived when (so the rounding does not depend on the order), the wave-tile
    // items it % S == split of ALL partials — an item = one fragment row x one fragment pair = 2 of the 40 accumulator fragments (a column half = 4 for the SwiGLU family) — and
    // runs the epilogue on exactly those.  The departure counter lets the last unit out re-arm both counters for the next launch.
    // Progress (round 4): the wait is bounded and has a way out that finishes the tile (below) — nothing here needs the units to be co-resident.
    typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
    constexpr int SLAB = BM3 * BN3 * 4;                  // one unit's partial tile, bytes
    const int tail = flat - full;
    float* tile_ws = g.ws + ((int64_t)tail * S) * (BM3 * BN3);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(tile_ws, 0, S * SLAB, 0x00020000);
    const int my_off = split * SLAB + tid * 16;
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j)
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[i][j]), rs, my_off + (i * 8 + j) * (NT3 * 16), 0, 16);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // every partial of this wave has reached the coherence point
    __syncthreads();
    int* arrive = g.tickets + tail;
    int* depart = g.tickets + 128 + tail;                // rem <= 128 (the split rule), the registered ticket array holds 384
    int* mode = g.tickets + 256 + tail;                  // the tile's decision: 0 = open, 1 = cooperative, 2 = last unit finishes alone
    // ROUND 4: FORWARD PROGRESS NO LONGER RESTS ON CO-RESIDENCY.  A unit waits a BOUNDED time (g.tail_wait shader cycles, ~60 us) for its
    // siblings; whoever first stops waiting — because all S have arrived, or because its time is up — settles the tile's mode with ONE
    // compare-and-swap, and every unit follows that one decision:
    //   COOP (proposed only by a unit that SAW all S arrivals): every unit reduces and stores its share, as in round 3;
    //   LAST (proposed by a unit whose wait ran out): a unit leaves at once; the LAST unit to pass the tile's departure counter — by then
    //        every partial has been stored, because a unit stores before it arrives — sums all S partials and finishes the whole tile.
    // Both paths add the partials in ascending split order, so the tile's bits do not depend on which path ran or on who arrived when.
    // A sibling that is scheduled late (another process on the GPU, a CU mask, a second waiting kernel the host-side rule did not see)
    // now costs one tile a slower fix-up instead of trapping the process.  Visibility: the partials are 16-byte sc1 (write-through)
    // stores drained by s_waitcnt vmcnt(0) before the arrival, and read back with sc1 loads — the guide's valid hand-off form R1
    // ("sc1 payload -> asm vmcnt(0) -> flag", MI355X_MICROARCH.md, inter-workgroup visibility): no L1 line of them exists anywhere and no
    // L2 keeps one, so neither a release write-back nor an acquire invalidate has anything to do.
    constexpr int MODE_COOP = 1, MODE_LAST = 2;
    int* s_flag = reinterpret_cast<int*>(smem);          // the K loop's stages are dead: every wave is past its last fragment read
    if (tid == 0) {
      __hip_atomic_fetch_add(arrive, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const long long t0 = __builtin_amdgcn_s_memtime();
      int m = 0, seen = 0;
      for (;;) {
        seen = __hip_atomic_load(arrive, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (seen >= S) break;
        m = __hip_atomic_load(mode, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (m) break;
        if (__builtin_amdgcn_s_memtime() - t0 > g.tail_wait) break;
        __builtin_amdgcn_s_sleep(8);
      }
      if (!m) {
        int expected = 0;
        const int want = seen >= S ? MODE_COOP : MODE_LAST;
        m = __hip_atomic_compare_exchange_strong(mode, &expected, want, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ? want : expected;
      }
      int last = 0;
      if (m == MODE_LAST) last = __hip_atomic_fetch_add(depart, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == S - 1;
      s_flag[0] = m | (last << 8);
    }
    __syncthreads();
    const int decided = s_flag[0];
    __syncthreads();
    const bool coop = (decided & 0xff) == MODE_COOP;
    if (!coop && !(decided >> 8)) return;                // mode LAST and not the last one out: the partial is stored, nothing else to do
    // the share: work items it % S == split (an item = a fragment pair, 20 per tile; the SwiGLU family's come as column halves, 10) —
    // or, for the unit that finishes the tile alone, all of them
    constexpr int N_ITEMS = (EPI == EPI_SWIGLU) ? 10 : 20;
    items = 0;
#pragma unroll
    for (int it = 0; it < N_ITEMS; ++it)
      if (!coop || it % S == split) items |= (EPI == EPI_SWIGLU) ? (3u << (2 * it)) : (1u << it);
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int sp = 0; sp < S; ++sp) {
      const int base = sp * SLAB + tid * 16;
#pragma unroll
      for (int it = 0; it < 20; ++it) {
        if (!((items >> it) & 1u)) continue;             // wave-uniform
        const int i = it >> 2, j0 = (it & 3) * 2;
        u32x4 t[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) t[q] = __builtin_amdgcn_raw_buffer_load_b128(rs, base + (i * 8 + j0 + q) * (NT3 * 16), 0, 16);
#pragma unroll
        for (int q = 0; q < 2; ++q) acc[i][j0 + q] += __builtin_bit_cast(f32x4, t[q]);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                     // every wave of this unit has read what it needs of the partials
    if (tid == 0) {
      // last one out re-arms the tile's three words (self-resetting): in COOP mode the departure counter is taken here, after the reads;
      // in LAST mode this unit IS the last one out
      const bool rearm = coop ? (__hip_atomic_fetch_add(depart, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == S - 1) : true;
      if (rearm) {
        __hip_atomic_store(depart, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(mode, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(arrive, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
  // the epilogue's lane coordinates are derived again from the thread id (opaque to the optimiser), so nothing of the epilogue's address
  // arithmetic is kept in registers across the K loop
  int tid2 = threadIdx.x;
  asm volatile("" : "+v"(tid2));
  gemm320_epilogue<EPI>(g, acc, batch, M, N, m0, n0, (tid2 >> 6) & 3, tid2 >> 8, tid2 & 15, (tid2 >> 4) & 3, items);
}

}  // namespace

// Whether the 320-row tiling is eligible for this call (dense bf16-out, aligned, offsets fit 32 bits) -- the choice between the two
// tilings is use_320() in gemm_bf16.hip.
bool mp_gemm320_eligible(const GemmArgs& g, int batch) {
  if (batch < 1 || batch > MAX_FLAT_BATCH3 || g.out_f32) return false;
  if (g.keep_gu && (batch != 1 || (g.ld_gu & 7) || (reinterpret_cast<uintptr_t>(g.keep_gu) & 15))) return false;
  // (the other activations' sweeps over 40 fragments do not fit beside 160 accumulators: the allocator then spills accumulators inside the K loop)
  if (!(g.act == ACT_NONE || g.act == ACT_QUICK_GELU || g.act == ACT_ROPE_QK || g.act == ACT_SWIGLU_PAIR || g.act == ACT_RELU || g.act == ACT_GELU)) return false;
  if ((g.act == ACT_RELU || g.act == ACT_GELU) && (g.c_rows || g.a_rows)) return false;      // the EPI_RELU / EPI_GELU families: dense / plain batched calls
  if (g.act == ACT_ROPE_QK && (batch != 1 || g.m_dev || g.a_rows || g.c_rows)) return false;
  if (g.act == ACT_SWIGLU_PAIR && (g.c_rows || g.residual || g.bias || g.alpha != 1.f || ((g.N >> 1) & 7))) return false;
  if (g.c_rows && (g.act != ACT_NONE || g.bias || g.alpha != 1.f)) return false;
  // M >= 1024 for the decoder's calls (fewer rows are few tiles: the smaller kernels fill the machine better).  The frozen towers' whole-tile
  // policy counts CU x time instead of latency (ops.throughput_tiles), and there a 512-row call on two row tiles beats the 128 x 128 kernel's
  // split-K units several times over (the SAM adapter's K = 6912 convolution: 6 tiles x 186 us against 240 units x ~20 us + their partials)
  if (g.a_scale && !(g.act == ACT_ROPE_QK || g.act == ACT_SWIGLU_PAIR)) return false;
  if (g.N % BN3 || g.K % BK3 || g.M < (mp_gemm_policy_whole_tiles() ? BM3 : 1024)) return false;
  if ((g.ldc & 7) || (reinterpret_cast<uintptr_t>(g.C) & 15) || (g.sC & 7)) return false;
  if (g.residual && ((g.ldr & 7) || (reinterpret_cast<uintptr_t>(g.residual) & 15) || (g.sR & 7))) return false;
  // 32-bit DMA offsets inside one batch's operand; a gathered A is addressed through the whole shared matrix, whose row count the call does
  // not carry: every routed token is a row of it, so batch * M * 4 rows bound it for any capacity factor >= 1/4
  const int64_t a_rows_bound = g.a_rows ? (int64_t)batch * g.M * 4 : (int64_t)g.M + 320;
  if (a_rows_bound * g.lda * 2 >= (1ll << 31) || (int64_t)g.N * g.ldw * 2 >= (1ll << 31)) return false;
  return true;
}

// Round 6, late: a DENSE call of at most half a wave of tiles (116 <= tiles <= 128 at 256 CUs: the N = 4096 projections at 2556 rows = BASELINE configs[1],
// the dense VQA forward at batch 4: o_proj and down_proj are 8 x 16 = 128 tiles) may cut every tile in two or three along K with the cooperative
// fix-up, instead of holding half the CUs for a whole tile-time (or 160 of them on 256-row tiles).  One rule for the selection model
// (use_320, gemm_bf16.hip) and the launcher.  K >= 8192 only (the fix-up's 2 x 84 MB of partials cost ~25 us whatever K is), M >= 1024 (a tile row
// mostly valid), never for the families that cannot split.  MP_GEMM320_SUBWAVE=0: off (A/B).  Returns the split factor (1: no split).
int mp_gemm320_subwave_split(const GemmArgs& g, int batch) {
  static int on = -1;
  if (on < 0) { const char* e = getenv("MP_GEMM320_SUBWAVE"); on = (e && atoi(e) == 0) ? 0 : 1; }
  if (!on || batch != 1 || g.m_dev || g.act == ACT_ROPE_QK || g.M < 1024 || g.K < 8192) return 1;      // K = 4096 (o_proj) measured neutral: 96.3 against 97.9 us
  if (mp_gemm_policy_whole_tiles()) return 1;
  const int C = std::min(mp_device_cus(), 256);
  const int64_t tiles = (int64_t)mp_cdiv(g.M, BM3) * (g.N / BN3);
  if (tiles * 2 > C || tiles * 4 <= C) return 1;
  // ... and only when the units then fill the chip (>= 90 % of the CUs): 96 tiles cut in two are 192 units, and measured SLOWER (down_proj at 1917
  // rows: 164.6-167.3 us) than what the 256-row kernel already does there with ITS tail rule — 8 x 16 = 128 tiles of 256 rows cut in two = 256
  // units, 158-161 us.  The 256-row kernel's rule (rem * 2 <= CUs) stops at 128 tiles; at 2556 rows it has 160 whole tiles on 160 CUs (218-223 us)
  // and this rule takes over: 128 tiles of 320 rows cut in two, 184-195 us (down_proj); 97.9 -> 96.3 for o_proj (K = 4096: neutral, not taken).
  // profiles/r06_subwave_bench.txt, profiles/r06_subwave_grid.txt
  const int64_t S = std::min<int64_t>(C / tiles, g.K / BK3 / 4);
  if (S < 2 || S * tiles * 10 < (int64_t)C * 9) return 1;
  return (int)S;
}

int mp_launch_gemm320(const GemmArgs& g0, int batch, hipStream_t stream) {
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)gemm320_bf16_nt_kernel<EPI_PLAIN>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE3);
    (void)hipFuncSetAttribute((const void*)gemm320_bf16_nt_kernel<EPI_ROPE>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE3);
    (void)hipFuncSetAttribute((const void*)gemm320_bf16_nt_kernel<EPI_SWIGLU>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE3);
    (void)hipFuncSetAttribute((const void*)gemm320_bf16_nt_kernel<EPI_COMBINE>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE3);
    (void)hipFuncSetAttribute((const void*)gemm320_bf16_nt_kernel<EPI_RELU>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE3);
    (void)hipFuncSetAttribute((const void*)gemm320_bf16_nt_kernel<EPI_GELU>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE3);
    attr = true;
  }
  GemmArgs g = g0;
  g.nbatch = batch;
  g.n_cu = std::min(mp_device_cus(), 256);
  static int max_split = -1;
  if (max_split < 0) { const char* e = getenv("MP_GEMM320_MAX_SPLIT"); max_split = (e && atoi(e) >= 1) ? atoi(e) : 10; }    // 1 = no tail split (A/B); 16 measured 357 us against 349 at 10 (down projection, 2500 + 2612 rows): more units, more partial traffic
  // dense calls keep whole waves and one accumulation order per shape (the selection model counts whole waves; a gemm() must not differ
  // from the kept-gate|up form of the same product by a split's extra fp32 rounding); the tail split serves the batched expert calls
  // round 3: with the cooperative fix-up a dense call's tail may split too — only behind at least one whole wave (small calls keep the
  // single accumulation order the bit-equality tests of fused against unfused paths rely on): the dense gate|up of the LoRA step is 16 x 86
  // = 1376 tiles = 5.375 waves, whose 96 tail tiles are cut in two instead of holding 96 CUs for a whole tile-time.  MP_GEMM320_DENSE_SPLIT=0: A/B.
  static int dense_split = -1;
  if (dense_split < 0) { const char* e = getenv("MP_GEMM320_DENSE_SPLIT"); dense_split = (e && atoi(e) == 0) ? 0 : 1; }
  const int64_t dense_tiles = (int64_t)mp_cdiv(g.M, BM3) * (g.N / BN3);
  g.max_split = (batch > 1 || g.m_dev) ? max_split : ((dense_split && (dense_tiles > g.n_cu || mp_gemm320_subwave_split(g, batch) > 1)) ? max_split : 1);
  if (g.act == ACT_ROPE_QK) g.max_split = 1;            // the RoPE family's epilogue has no item mask (its calls are dense qkv projections)
  // Units of a split tail WAIT for their siblings, so only one such kernel may be in flight on the device: a second one on a concurrent
  // stream can hold the CUs the first one's unscheduled units need while waiting for CUs the first one holds (B = 16 with the towers
  // started ahead: the CLIP qkv projection, 348 tiles cut into 92 x 2 units, beside the experts' down projection -> ~1 s stalls per
  // step).  Streams the host registered as concurrent (mp_gemm_set_stream_workspace) therefore never split; the primary stream does.
  // The frozen towers ask for whole tiles outright (tile policy 3), so that their results do not depend on which stream a step ran them on.
  // Round 4: that rule is now about SPEED only (two waiting kernels slow each other down to the bounded wait); a unit whose siblings do not
  // show up within g.tail_wait cycles leaves the tile to its last unit, so progress does not depend on what else holds the CUs.
  if (mp_gemm_stream_registered(stream) || mp_gemm_policy_whole_tiles()) g.max_split = 1;
  g.tail_wait = mp_gemm_tail_wait_value();
  int64_t ws_bytes = 0;
  mp_gemm_split_workspace(stream, &g.ws, &g.tickets, &ws_bytes);
  if (!g.ws || ws_bytes < (int64_t)g.n_cu * BM3 * BN3 * 4) { g.ws = nullptr; g.tickets = nullptr; g.max_split = 1; }   // (registration checks the 384 tickets)
  // the host-side bound on the tile count plus one unit per CU for a split tail; workgroups beyond the device-side unit count exit at once.
  // Round 6: a launch that cannot split (max_split 1: the towers' whole-tile policy, registered side streams, the RoPE family, one-wave dense
  // calls) gets no surplus — 256 workgroups that exit at once still each wait for a CU with 144 KB of LDS free, which beside another stream's
  // GEMM tiles means behind them.
  const int64_t host_tiles = (int64_t)mp_cdiv(g.M, BM3) * (g.N / BN3) * batch;
  const int64_t host_rem = host_tiles % g.n_cu;
  const bool may_split = g.max_split > 1 && (g.m_dev || (host_rem > 0 && host_rem * 2 <= g.n_cu));     // (the kernel's own rule, where the host knows the row counts)
  const dim3 grid((unsigned)(host_tiles + (may_split ? g.n_cu : 0)));
#define MP3_GO(E) hipLaunchKernelGGL(gemm320_bf16_nt_kernel<E>, grid, dim3(NT3), 2 * STAGE3, stream, g)
  if (g.act == ACT_ROPE_QK) MP3_GO(EPI_ROPE);
  else if (g.act == ACT_SWIGLU_PAIR) MP3_GO(EPI_SWIGLU);
  else if (g.c_rows) MP3_GO(EPI_COMBINE);
  else if (g.act == ACT_RELU) MP3_GO(EPI_RELU);
  else if (g.act == ACT_GELU) MP3_GO(EPI_GELU);
  else MP3_GO(EPI_PLAIN);
#undef MP3_GO
  return mp_check_launch("mp_gemm_bf16_nt(320)");
}
