// bf16 MFMA GEMM for gfx950:  C[M,N] = epilogue(A[M,K] · W[N,K]^T)      ("NT": both operands K-contiguous,
// which is nn.Linear's natural layout: activations [tokens, in], weight [out, in]).
//
// Replaces the cuBLAS calls behind every nn.Linear / conv-as-GEMM on the reference hot path
// (HF LlamaAttention/LlamaMLP q,k,v,o,gate,up,down; CLIP fc1/fc2/qkv/out; mm_projector
// model/medplib/model/multimodal_projector/builder.py:39-46; SAM qkv/proj/MLP image_encoder.py:273-296).
//
// Tile: 128x128x64, 256 threads = 4 waves (2x2), each wave 64x64 = 4x4 fragments of
// v_mfma_f32_16x16x32_bf16.  Operands staged global -> VGPR -> LDS (16 B per lane, XOR-swizzled 16-B chunks so the
// ds_read_b128 fragment reads are bank-conflict free), double-buffered LDS, one barrier per K-tile, next tile's
// global loads in flight under the current tile's MFMAs.  XCD-aware block remap keeps one W panel per L2.
#include "common.h"
#include "gemm_common.h"
#include <stdlib.h>
#include <algorithm>

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int NT = 256;

// byte offset of 16-B chunk `c` (0..7) of row `r` in a [rows][64] bf16 LDS tile, XOR-swizzled
__device__ __forceinline__ int lds_off(int r, int c) { return r * 128 + ((c ^ (r & 7)) << 4); }

template <bool GLDS>
__global__ __launch_bounds__(NT, 2) void gemm_bf16_nt_kernel(GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // [buf][A|W][128*64 bf16 = 16 KiB]
  char* sA0 = smem;
  char* sW0 = smem + 2 * 16384;

  const int batch = blockIdx.y;
  const bf16_t* __restrict__ A = g.A + batch * g.sA;
  const bf16_t* __restrict__ W = g.W + batch * g.sW;
  const int M = g.m_dev ? min(g.M, g.m_dev[batch * g.m_dev_stride]) : g.M;
  const int N = g.N, K = g.K;

  const int tiles_m = (g.M + BM - 1) / BM;
  const int tiles_n = (N + BN - 1) / BN;
  const int nwg = tiles_m * tiles_n;
  // split-K (host-chosen, g.max_split > 1 only for launches with few tiles and a long K, e.g. the SAM adapter convolutions:
  // 24 tiles x 108 K-tiles): unit = split * nwg + tile; partial sums meet in the workspace exactly like the 256x256 kernel's
  // tail split (gemm256_bf16.hip): register-order fp32 partials by agent-scope stores, a ticket per tile, the last arriver sums
  // the S partials in ascending split order and runs the epilogue.
  const int S = g.max_split > 1 ? g.max_split : 1;
  const int split = blockIdx.x / nwg;
  // bijective XCD remap: blocks that land on the same XCD (bid % 8) get a contiguous range of tile ids
  int bid = blockIdx.x - split * nwg;
  const int tile_id = bid;
  {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, loc = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
  }
  // grouped order: ids walk GROUP_M consecutive M-tiles before stepping N, so the ~32-64 workgroups that are co-resident on
  // one XCD cover a GROUP_M x (32/GROUP_M) block of tiles and share both their A and their W panels through that XCD's L2
  const int GROUP_M = g.group_m;
  const int per_group = GROUP_M * tiles_n;
  const int grp = bid / per_group;
  const int first_m = grp * GROUP_M;
  const int gsz = min(tiles_m - first_m, GROUP_M);
  const int tm = first_m + (bid % per_group) % gsz, tn = (bid % per_group) / gsz;
  const int m0 = tm * BM, n0 = tn * BN;
  if (m0 >= M) return;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  int kt0 = 0, nt = K / BK;
  if (S > 1) {
    const int base = nt / S, extra = nt % S;
    kt0 = split * base + min(split, extra);
    nt = base + (split < extra ? 1 : 0);
  }
  const int fr = lane & 15;   // fragment row (A) / col (W) within 16
  const int fq = lane >> 4;   // k-chunk selector 0..3

  auto compute = [&](int buf) {
    const char* sa = sA0 + buf * 16384;
    const char* sw = sW0 + buf * 16384;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      bf16x8 fa[4], fb[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int r = wm * 64 + i * 16 + fr;
        fa[i] = *reinterpret_cast<const bf16x8*>(sa + lds_off(r, kk * 4 + fq));
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int r = wn * 64 + j * 16 + fr;
        fb[j] = *reinterpret_cast<const bf16x8*>(sw + lds_off(r, kk * 4 + fq));
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);   // (W, A): transposed accumulators
    }
  };

  if constexpr (GLDS) {
    // Direct global -> LDS DMA (global_load_lds_dwordx4): each wave-instruction fills 1 KiB of LDS lane-linearly, so the
    // XOR swizzle lives on the SOURCE address: LDS chunk position p = instr*64 + lane holds (row p/8, logical chunk
    // (p%8) ^ (row&7)) — the same image lds_off() reads.  Wave w owns instructions w*4 .. w*4+3 of each operand.
    const int sub_row = lane >> 3;                      // 0..7 within the 8-row group of one instruction
    const int src_c = (lane & 7) ^ sub_row;             // logical 16-B chunk this lane fetches
    const bf16_t* a_src[4];
    const bf16_t* w_src[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = (wave * 4 + i) * 8 + sub_row;
      int ar = min(m0 + row, M - 1);
      if (g.a_rows) ar = g.a_rows[batch * g.rows_stride + ar];
      a_src[i] = A + (int64_t)ar * g.lda + src_c * 8 + (int64_t)kt0 * BK;
      w_src[i] = W + (int64_t)min(n0 + row, N - 1) * g.ldw + src_c * 8 + (int64_t)kt0 * BK;
    }
    auto issue = [&](int t, int buf) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int lbase = buf * 16384 + (wave * 4 + i) * 1024;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a_src[i] + (int64_t)t * BK),
                                         (__attribute__((address_space(3))) void*)(sA0 + lbase), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(w_src[i] + (int64_t)t * BK),
                                         (__attribute__((address_space(3))) void*)(sW0 + lbase), 16, 0, 0);
      }
    };
    issue(0, 0);
    for (int t = 0; t < nt; ++t) {
      __syncthreads();                       // tile t landed (vmcnt(0) + barrier); everyone is done reading buf[(t+1)&1]
      if (t + 1 < nt) issue(t + 1, (t + 1) & 1);
      compute(t & 1);
    }
  } else {
    // register-staged variant (global -> VGPR -> ds_write_b128)
    const int ld_row = tid >> 3;   // 0..31  (+32*i)
    const int ld_c = tid & 7;      // 16-B chunk within the 64-wide K slab
    const bf16_t* a_ptr[4];
    const bf16_t* w_ptr[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int ar = min(m0 + ld_row + 32 * i, M - 1);
      if (g.a_rows) ar = g.a_rows[batch * g.rows_stride + ar];
      a_ptr[i] = A + (int64_t)ar * g.lda + ld_c * 8 + (int64_t)kt0 * BK;
      w_ptr[i] = W + (int64_t)min(n0 + ld_row + 32 * i, N - 1) * g.ldw + ld_c * 8 + (int64_t)kt0 * BK;
    }
    bf16x8 ra[4], rw[4];
    auto gload = [&](int t) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        ra[i] = *reinterpret_cast<const bf16x8*>(a_ptr[i] + (int64_t)t * BK);
        rw[i] = *reinterpret_cast<const bf16x8*>(w_ptr[i] + (int64_t)t * BK);
      }
    };
    auto lstore = [&](int buf) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int r = ld_row + 32 * i;
        *reinterpret_cast<bf16x8*>(sA0 + buf * 16384 + lds_off(r, ld_c)) = ra[i];
        *reinterpret_cast<bf16x8*>(sW0 + buf * 16384 + lds_off(r, ld_c)) = rw[i];
      }
    };
    gload(0);
    lstore(0);
    __syncthreads();
    for (int t = 0; t < nt; ++t) {
      if (t + 1 < nt) gload(t + 1);
      compute(t & 1);
      if (t + 1 < nt) lstore((t & 1) ^ 1);
      __syncthreads();
    }
  }

  if (S > 1) {
    __syncthreads();                                     // all LDS reads of the main loop are done: smem[0] becomes the flag
    unsigned long long* wsu = reinterpret_cast<unsigned long long*>(g.ws) + ((int64_t)(batch * nwg + tile_id) * S) * (BM * BN / 2);
    unsigned long long* mine = wsu + (int64_t)split * (BM * BN / 2);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const unsigned long long v = (unsigned long long)__float_as_uint(acc[i][j][2 * h]) |
                                       ((unsigned long long)__float_as_uint(acc[i][j][2 * h + 1]) << 32);
          __hip_atomic_store(mine + ((i * 4 + j) * 2 + h) * NT + tid, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int* flag = reinterpret_cast<int*>(smem);
    int* ticket_p = g.tickets + batch * nwg + tile_id;
    if (tid == 0) *flag = __hip_atomic_fetch_add(ticket_p, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const int ticket = *flag;
    if (ticket != S - 1) return;
    if (tid == 0) __hip_atomic_store(ticket_p, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int sp = 0; sp < S; ++sp) {
      const unsigned long long* part = wsu + (int64_t)sp * (BM * BN / 2);
      unsigned long long t[32];
#pragma unroll
      for (int q = 0; q < 32; ++q) t[q] = __hip_atomic_load(part + q * NT + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
      for (int q = 0; q < 32; ++q) {
        const int i = q >> 3, j = (q >> 1) & 3, h = q & 1;
        acc[i][j][2 * h] += __uint_as_float((unsigned)(t[q] & 0xffffffffull));
        acc[i][j][2 * h + 1] += __uint_as_float((unsigned)(t[q] >> 32));
      }
    }
  }
  // epilogue.  The MFMAs are issued as (W fragment, A fragment), so acc[i][j][r] = C[row = i*16 + fr, col = j*16 + fq*4 + r] of the
  // wave tile: a lane owns four consecutive columns of one row per fragment and stores them as one 8-byte (bf16) / 16-byte (fp32)
  // piece.  Same rounding points as before: activation result rounded to the output type, residual added after.
  if (g.c_rows) {      // combine folded into the epilogue (see gemm256_bf16.hip): out[token] = residual[token] + weight[token] * bf16(acc)
    bf16_t* Co = reinterpret_cast<bf16_t*>(g.C);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = m0 + wm * 64 + i * 16 + fr;
      if (row >= M) continue;
      const int orow = g.c_rows[batch * g.rows_stride + row];
      const float sc = g.c_scale ? g.c_scale[orow] : 1.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int col = n0 + wn * 64 + j * 16 + fq * 4;
        if (col + 4 > N) continue;
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = sc * (float)(bf16_t)acc[i][j][r];
        if (g.residual) {
          const bf16x4 rv = *reinterpret_cast<const bf16x4*>(g.residual + (int64_t)orow * g.ldr + col);
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] += (float)rv[r];
        }
        *reinterpret_cast<bf16x4*>(Co + (int64_t)orow * g.ldc + col) = bf16x4{(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
      }
    }
    return;
  }
  const float* bias = g.bias ? g.bias + batch * g.sBias : nullptr;
  const bf16_t* R = g.residual ? g.residual + batch * g.sR : nullptr;
  bf16_t* Cb = g.out_f32 ? nullptr : reinterpret_cast<bf16_t*>(g.C) + batch * g.sC;
  float* Cf = g.out_f32 ? reinterpret_cast<float*>(g.C) + batch * g.sC : nullptr;
  const bool vec = (g.ldc & 3) == 0 && (!R || (g.ldr & 3) == 0);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int col = n0 + wn * 64 + j * 16 + fq * 4;
    if (col >= N) continue;
    float bv[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) bv[r] = (bias && col + r < N) ? bias[col + r] : 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = m0 + wm * 64 + i * 16 + fr;
      if (row >= M) continue;
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = apply_act(acc[i][j][r] * g.alpha + bv[r], g.act);
      if (vec && col + 4 <= N) {
        if (R) {
          const bf16x4 rv = *reinterpret_cast<const bf16x4*>(R + (int64_t)row * g.ldr + col);
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] += (float)rv[r];
        }
        if (Cf) *reinterpret_cast<f32x4*>(Cf + (int64_t)row * g.ldc + col) = f32x4{v[0], v[1], v[2], v[3]};
        else *reinterpret_cast<bf16x4*>(Cb + (int64_t)row * g.ldc + col) = bf16x4{(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (col + r >= N) continue;
          const float f = v[r] + (R ? (float)R[(int64_t)row * g.ldr + col + r] : 0.f);
          if (Cf) Cf[(int64_t)row * g.ldc + col + r] = f;
          else Cb[(int64_t)row * g.ldc + col + r] = (bf16_t)f;
        }
      }
    }
  }
}

}  // namespace

static int gemm_variant() {
  static int v = -1;
  if (v < 0) {
    // 2 (default) = auto: 256x256 ping-pong kernel for large problems, 128x128 LDS-DMA kernel otherwise;
    // 1 = always 128x128 LDS-DMA staging; 0 = 128x128 register staging (A/B reference)
    const char* e = getenv("MP_GEMM_VARIANT");
    v = (e && e[0] >= '0' && e[0] <= '2') ? (e[0] - '0') : 2;
  }
  return v;
}

static int gemm_group_m() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("MP_GEMM_GROUP_M");
    v = (e && atoi(e) >= 1) ? atoi(e) : 4;
  }
  return v;
}

static bool use_256_rule(const GemmArgs& g, int batch) {
  if (g.act == ACT_SWIGLU_PAIR || g.act == ACT_ROPE_QK) return true;   // the paired epilogues exist in the 256x256 kernel only
  if (batch > 8) return false;                        // its flat work decode walks at most 8 batches
  if (gemm_variant() != 2) return false;
  const int64_t tiles = mp_cdiv(g.M, 256) * mp_cdiv(g.N, 256) * batch;
  static int min_tiles = -1, min_n = -1;
  if (min_tiles < 0) {
    const char* e = getenv("MP_GEMM256_MIN_TILES");
    min_tiles = (e && atoi(e) >= 1) ? atoi(e) : 128;
    const char* f = getenv("MP_GEMM256_MIN_N");
    min_n = (f && atoi(f) >= 1) ? atoi(f) : 1024;
  }
  // short K with a small second tile wave (CLIP fc1 / mm_projector.0: 304 / 288 tiles, 16 K-tiles): the tail split leaves 4-K-tile
  // units that are all prologue and epilogue; the 128x128 kernel measured 67 vs 79 us there
  static int shortk = -1;
  if (shortk < 0) { const char* e = getenv("MP_GEMM_SHORTK_RULE"); shortk = (e && atoi(e) == 0) ? 0 : 1; }
  if (shortk && g.K <= 1024 && tiles > 256 && (tiles % 256) > 0 && (tiles % 256) < 128) return false;
  // few tiles but a long K (CLIP fc2: 4616 x 1024 x 4096 = 76 tiles): the tail split cuts each tile into K-ranges of >= 16 K-tiles that
  // fill the machine (3 x 76 units): 64 vs 77 us on the 128x128 kernel (scripts/tower_ab.sh)
  if (g.M >= 1024 && g.N >= min_n && tiles >= 64 && tiles < min_tiles && g.K >= 4096 && !g.m_dev && !g.out_f32) return true;
  return g.M >= 1024 && g.N >= min_n && tiles >= min_tiles;
}

// what the RoPE epilogue of the 320-row kernel costs on top of the plain one, in microseconds per wave of tiles (see gemm320_bf16.hip)
#define MP_GEMM320_ROPE_EXTRA_US 6.0
// implemented in gemm256_bf16.hip (cached per device) / gemm320_bf16.hip
int mp_device_cus();
bool mp_gemm320_eligible(const GemmArgs& g, int batch);
int mp_launch_gemm320(const GemmArgs& g, int batch, hipStream_t stream);
int mp_gemm320_subwave_split(const GemmArgs& g, int batch);
bool mp_gemm_stream_registered(hipStream_t stream);

// 320-row tiles, or the kernel the call would otherwise get?  Modelled time in microseconds, from K sweeps at one full wave of tiles
// (scripts/gemm_ksweep.py, same box): a wave of 256x256 tiles costs 8.5 + 1.45 per 64-deep K step (prologue + epilogue, then 1480-1530
// TFLOP/s in the loop), a wave of 320x256 tiles 4.5 + 1.685 per step (1590-1640 TFLOP/s in the loop; the fixed part was 30 us until the
// kernel was split by epilogue family, see gemm320_bf16.hip); a residual epilogue adds ~8 to either, QuickGELU ~6.  The 256 tiling pays
// floor(T / C) whole waves plus a tail (a whole wave when more than half the CUs have a tile, else 1 / S of one for the S-way split
// plus ~0.3 for the partials' round trip through memory); the 320 tiling has no tail split: all its waves are whole.  Where the 256x256
// rule does not apply (short K with a small second wave, narrow N) the alternative is the 128x128 kernel at the ~560 TFLOP/s it reaches
// on such shapes (CLIP fc1: 72 us against 43.5 on 320-row tiles).  MP_GEMM320 = 0 never, 2 whenever eligible, default 1 = by this model.
static thread_local int g_tile_policy = -1;          // mp_gemm_tile_policy(): -1 = the process default (MP_GEMM320, else 1)
static bool use_320(const GemmArgs& g, int batch, hipStream_t stream = nullptr) {
  static int env_mode = -1;
  if (env_mode < 0) { const char* e = getenv("MP_GEMM320"); env_mode = (e && e[0] >= '0' && e[0] <= '2') ? e[0] - '0' : 1; }
  const int mode = g_tile_policy >= 0 ? g_tile_policy : env_mode;
  if (mode == 0 || gemm_variant() != 2 || !mp_gemm320_eligible(g, batch)) return false;
  if (mode >= 2) return true;
  const int C = std::min(mp_device_cus(), 256);
  const int64_t t256 = mp_cdiv(g.M, 256) * mp_cdiv(g.N, 256), t320 = mp_cdiv(g.M, 320) * (g.N / 256);
  if (t320 * 2 < C && (mp_gemm_stream_registered(stream) || mp_gemm320_subwave_split(g, batch) <= 1)) return false;   // fewer workgroups than half the CUs: the smaller tiles (or a K split) fill the machine better
  const double k = g.K / 64.0;
  const double epi = (g.residual ? 8.0 : 0.0) + (g.act == ACT_QUICK_GELU ? 6.0 : 0.0);
  const double rope320 = g.act == ACT_ROPE_QK ? MP_GEMM320_ROPE_EXTRA_US : 0.0;
  const double w320 = (double)mp_cdiv(t320, C);          // dense calls run whole waves (the kernel's tail split is for the batched expert calls)
  double c320 = w320 * (4.5 + epi + rope320 + 1.685 * k);
  // at most half a wave of tiles and a long K: every tile cut S ways with the cooperative fix-up (mp_gemm320_subwave_split; the primary stream only,
  // like every split).  MP_GEMM320_SUBWAVE_FIX_US: the fix-up's price in the model (partials written and read back: ~2 x 84 MB at S = 2)
  if (!mp_gemm_stream_registered(stream)) {
    const int S = mp_gemm320_subwave_split(g, batch);
    if (S > 1) {
      static double fix_us = -1.0;
      if (fix_us < 0) { const char* e = getenv("MP_GEMM320_SUBWAVE_FIX_US"); fix_us = (e && atof(e) > 0) ? atof(e) : 28.0; }
      c320 = std::min(c320, 4.5 + epi + 1.685 * k / S + fix_us);
    }
  }
  double other;
  if (use_256_rule(g, batch)) {
    const int64_t rem = t256 % C;
    double w256 = (double)(t256 / C);
    if (rem > 0) {
      const int S = (int)std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(C / rem, 8), g.K / 64 / 4));
      w256 += (rem * 2 > C || S == 1) ? 1.0 : 1.0 / S + 0.3;
    }
    other = w256 * (8.5 + epi + 1.45 * k);
  } else {
    other = 2.0 * g.M * g.N * g.K / 560e6;
  }
  return c320 < 0.98 * other;
}

// Batched calls (the MoE expert projections: per-expert device-side row counts, rows gathered / scattered through the routing tables).  The
// host does not know the row counts, so there is no wave model here: the 320-row tiles take the eligible calls with a long K and N >= 8192
// (gate|up: N = 22016, K = 4096: 86 column tiles, so a row tile more or less moves the wave count by a few percent) -- their K loop runs
// ~5 % faster and a wave of tiles costs 4 us less in prologue and epilogue than a wave of 256x256 tiles: 665-677 us against 722-787 for
// E = 2 at 5112 tokens (scripts/expert_gemm_ab.py).  The down projection (N = 4096: 16 column tiles) stays on 256x256 tiles: 2556 + 2556
// rows are 8 + 8 row tiles of 320 = exactly one wave (308 us against 365), but 2500 + 2612 are 8 + 9 = 1.06 waves and this kernel had no
// tail split then (531 us against 380); with the tail split the kernel has since got (16 tiles cut 8 ways) it measures 375-381 against
// 380-383: the routing is never balanced to the row, so there is nothing to win and the rule stays N >= 8192.  MP_GEMM320_BATCHED=0: never (A/B).
static bool use_320_batched(const GemmArgs& g, int batch) {
  static int env_b = -1, env_mode = -1;
  if (env_b < 0) { const char* e = getenv("MP_GEMM320_BATCHED"); env_b = (e && atoi(e) == 0) ? 0 : 1; }
  if (env_mode < 0) { const char* e = getenv("MP_GEMM320"); env_mode = (e && e[0] >= '0' && e[0] <= '2') ? e[0] - '0' : 1; }
  const int mode = g_tile_policy >= 0 ? g_tile_policy : env_mode;
  if (mode == 0 || gemm_variant() != 2 || !mp_gemm320_eligible(g, batch)) return false;
  if (mode >= 2) return true;
  if (!env_b) return false;
  if (g.K >= 2048 && g.N >= 8192) return true;
  // round 3: the experts' down projection (N = 4096, K = 11008, combine epilogue) too.  8 + 9 row tiles of 320 are 272 tiles = one wave
  // + 16 tail tiles; the tail is now cut 10 ways with a COOPERATIVE fix-up (every unit reduces and stores a share of the tile instead
  // of the last arriver summing S x 320 KiB alone, gemm320_bf16.hip), which is what the rule above was waiting for.  MP_GEMM320_DOWN=0: A/B.
  static int env_d = -1;
  if (env_d < 0) { const char* e = getenv("MP_GEMM320_DOWN"); env_d = (e && atoi(e) == 0) ? 0 : 1; }
  return env_d && g.c_rows && g.K >= 8192 && g.N >= 2048;
}

// which kernel the last bf16 GEMM entry of this thread dispatched to (bench.py attributes its HIP-event samples per kernel)
static thread_local int g_last_gemm_kernel = 0;
static bool use_256(const GemmArgs& g, int batch) {
  const bool big = use_256_rule(g, batch);
  g_last_gemm_kernel = big ? 256 : 128;
  return big;
}
extern "C" int mp_gemm_last_kernel(void) { return g_last_gemm_kernel; }
// policy 3 (the frozen towers): the 320-row kernel's tails never split, whichever stream the call is on (gemm320_bf16.hip: mp_launch_gemm320)
bool mp_gemm_policy_whole_tiles() { return g_tile_policy == 3; }
// mp_gemm_tail_wait(): how long (shader cycles) a unit of the 320-row kernel's split tail waits for its siblings before the tile falls
// back to "the last unit finishes alone".  Default 150 k cycles (~60-80 us: several K-ranges of the longest split the decoder runs);
// 0 = never wait (every tile decides at once: the test of the fallback path), MP_GEMM320_TAIL_WAIT overrides the default.
static long long g_tail_wait = -1;
long long mp_gemm_tail_wait_value() {
  if (g_tail_wait < 0) { const char* e = getenv("MP_GEMM320_TAIL_WAIT"); g_tail_wait = (e && atoll(e) >= 0) ? atoll(e) : 150000; }
  return g_tail_wait;
}
extern "C" int64_t mp_gemm_tail_wait(int64_t cycles) {
  const long long prev = mp_gemm_tail_wait_value();
  if (cycles >= 0) g_tail_wait = cycles;
  return prev;
}
extern "C" int mp_gemm_tile_policy(int mode) {
  MP_REQUIRE(mode >= -1 && mode <= 3, MP_ERR_ARG, "mp_gemm_tile_policy: mode must be -1 (default), 0 (256-row tiles only), 1 (by the wave model), 2 (320-row tiles whenever eligible) or 3 (as 2, tails never split)");
  g_tile_policy = mode;
  return MP_OK;
}

// implemented in gemm256_bf16.hip: the registered split-K scratch (mp_gemm_set_workspace)
void mp_gemm_split_workspace(hipStream_t stream, float** ws, int** tickets, int64_t* bytes);

// split-K factor of a 128x128 launch: only when the tile count leaves most of the machine idle, K is long enough to amortise the
// partial-sum round trip, the row count is known on the host and the scratch is registered
static int split128(const GemmArgs& g, int batch, hipStream_t stream, float** ws, int** tickets) {
  static int enabled = -1;
  if (enabled < 0) { const char* e = getenv("MP_GEMM_MAX_SPLIT"); enabled = (e && atoi(e) == 1) ? 0 : 1; }
  if (!enabled || g.m_dev) return 1;
  int64_t bytes = 0;
  mp_gemm_split_workspace(stream, ws, tickets, &bytes);
  if (!*ws) return 1;
  const int64_t tiles = mp_cdiv(g.M, BM) * mp_cdiv(g.N, BN) * batch;
  const int nt = g.K / BK;
  if (tiles > 128 || nt < 8) return 1;
  int S = (int)std::min<int64_t>(std::min<int64_t>(256 / tiles, nt / 4), 16);
  while (S > 1 && tiles * S * (int64_t)BM * BN * 4 > bytes) --S;
  return S < 2 ? 1 : S;
}

static void launch_gemm(const GemmArgs& g, dim3 grid, hipStream_t stream) {
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)gemm_bf16_nt_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    (void)hipFuncSetAttribute((const void*)gemm_bf16_nt_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    attr_set = true;
  }
  if (gemm_variant() >= 1) hipLaunchKernelGGL(gemm_bf16_nt_kernel<true>, grid, dim3(NT), 65536, stream, g);
  else hipLaunchKernelGGL(gemm_bf16_nt_kernel<false>, grid, dim3(NT), 65536, stream, g);
}

// C-ABI: see include/medplib_hip.h
extern "C" int mp_gemm_bf16_nt(const void* A, int64_t lda, const void* W, int64_t ldw, void* C, int64_t ldc,
                               const float* bias, const void* residual, int64_t ldr, int M, int N, int K, int act,
                               int out_dtype, float alpha, const int* m_dev, hipStream_t stream) {
  MP_REQUIRE(M >= 0 && N > 0 && K > 0, MP_ERR_SHAPE, "mp_gemm_bf16_nt: bad shape M=%d N=%d K=%d", M, N, K);
  MP_REQUIRE(K % BK == 0, MP_ERR_SHAPE, "mp_gemm_bf16_nt: K=%d must be a multiple of %d (pad on the host)", K, BK);
  MP_REQUIRE(lda % 8 == 0 && ldw % 8 == 0, MP_ERR_SHAPE, "mp_gemm_bf16_nt: lda/ldw must be multiples of 8");
  MP_REQUIRE(out_dtype == MP_BF16 || out_dtype == MP_F32, MP_ERR_DTYPE, "mp_gemm_bf16_nt: bad out dtype %d", out_dtype);
  MP_REQUIRE(act >= 0 && act <= 5, MP_ERR_ARG, "mp_gemm_bf16_nt: bad activation %d", act);
  MP_REQUIRE(act != ACT_SWIGLU_PAIR || (N % 64 == 0 && out_dtype == MP_BF16 && residual == nullptr && ldc % 8 == 0), MP_ERR_ARG,
             "mp_gemm_bf16_nt: SWIGLU_PAIR needs N %% 64 == 0, bf16 output [M, N/2] with ldc %% 8 == 0 and no residual");
  if (M == 0) return MP_OK;
  GemmArgs g{};
  g.A = (const bf16_t*)A; g.lda = lda; g.W = (const bf16_t*)W; g.ldw = ldw; g.C = C; g.ldc = ldc;
  g.bias = bias; g.residual = (const bf16_t*)residual; g.ldr = ldr; g.m_dev = m_dev; g.M = M; g.N = N; g.K = K;
  g.act = act; g.out_f32 = (out_dtype == MP_F32); g.alpha = alpha;
  g.sA = g.sW = g.sC = g.sR = g.sBias = 0; g.m_dev_stride = 0; g.group_m = gemm_group_m();
  if (use_320(g, 1, stream)) { g_last_gemm_kernel = 320; return mp_launch_gemm320(g, 1, stream); }
  if (use_256(g, 1)) return mp_launch_gemm256(g, 1, stream);
  const int tiles = (int)(mp_cdiv(M, BM) * mp_cdiv(N, BN));
  g.max_split = split128(g, 1, stream, &g.ws, &g.tickets);
  launch_gemm(g, dim3(tiles * g.max_split, 1), stream);
  return mp_check_launch("mp_gemm_bf16_nt");
}

// Fused qkv projection + RoPE (LlamaAttention: q_proj / k_proj / v_proj + apply_rotary_pos_emb): C[M, 3*hidden] =
// A[M, K] @ Wi[3*hidden, K]^T with the rotation of the q and k thirds done in the epilogue.  Wi = the fused qkv weight with the rows
// of every q / k head interleaved in blocks of 32 (ACT_ROPE_QK, gemm_common.h); C comes out in the standard layout.
extern "C" int mp_gemm_qkv_rope_bf16(const void* A, int64_t lda, const void* Wi, int64_t ldw, void* C, int64_t ldc, const float* cos_t,
                                     const float* sin_t, int M, int N, int K, int seq, int pos_offset, int head_dim, hipStream_t stream) {
  MP_REQUIRE(M >= 0 && N > 0 && K > 0 && K % BK == 0, MP_ERR_SHAPE, "mp_gemm_qkv_rope_bf16: K must be a multiple of %d", BK);
  MP_REQUIRE(head_dim == 128 && N % 3 == 0 && (N / 3) % 256 == 0, MP_ERR_SHAPE,
             "mp_gemm_qkv_rope_bf16: head_dim 128 and hidden %% 256 == 0 (got head_dim %d, N %d)", head_dim, N);
  MP_REQUIRE(lda % 8 == 0 && ldw % 8 == 0 && ldc % 4 == 0 && cos_t && sin_t && seq > 0, MP_ERR_ARG, "mp_gemm_qkv_rope_bf16: bad arguments");
  if (M == 0) return MP_OK;
  GemmArgs g{};
  g.A = (const bf16_t*)A; g.lda = lda; g.W = (const bf16_t*)Wi; g.ldw = ldw; g.C = C; g.ldc = ldc;
  g.M = M; g.N = N; g.K = K; g.act = ACT_ROPE_QK; g.out_f32 = 0; g.alpha = 1.f;
  g.group_m = gemm_group_m();
  g.rope_cos = cos_t; g.rope_sin = sin_t; g.rope_seq = seq; g.rope_pos0 = pos_offset;
  if (use_320(g, 1)) { g_last_gemm_kernel = 320; return mp_launch_gemm320(g, 1, stream); }
  (void)use_256(g, 1);
  return mp_launch_gemm256(g, 1, stream);
}

// The same with the input RMSNorm folded in (config.fold_input_norm): A = the RAW residual stream, Wi = the interleaved qkv weight with the norm
// weight multiplied into its columns, row_scale [M] = rstd of every row (mp_rmsnorm_gate_rstd_bf16 with n_experts = 0): the epilogue multiplies
// the fp32 accumulators by rstd before the projection's bf16 rounding and the rotation.  320-row kernel only: a shape it does not take is an error
// (the caller keeps the unfolded path for those).
extern "C" int mp_gemm_qkv_rope_scaled_bf16(const void* A, int64_t lda, const void* Wi, int64_t ldw, void* C, int64_t ldc, const float* cos_t,
                                            const float* sin_t, const float* row_scale, int M, int N, int K, int seq, int pos_offset, int head_dim,
                                            hipStream_t stream) {
  MP_REQUIRE(M >= 0 && N > 0 && K > 0 && K % BK == 0, MP_ERR_SHAPE, "mp_gemm_qkv_rope_scaled_bf16: K must be a multiple of %d", BK);
  MP_REQUIRE(head_dim == 128 && N % 3 == 0 && (N / 3) % 256 == 0, MP_ERR_SHAPE, "mp_gemm_qkv_rope_scaled_bf16: head_dim 128 and hidden %% 256 == 0");
  MP_REQUIRE(lda % 8 == 0 && ldw % 8 == 0 && ldc % 4 == 0 && cos_t && sin_t && row_scale && seq > 0, MP_ERR_ARG, "mp_gemm_qkv_rope_scaled_bf16: bad arguments");
  if (M == 0) return MP_OK;
  GemmArgs g{};
  g.A = (const bf16_t*)A; g.lda = lda; g.W = (const bf16_t*)Wi; g.ldw = ldw; g.C = C; g.ldc = ldc;
  g.M = M; g.N = N; g.K = K; g.act = ACT_ROPE_QK; g.out_f32 = 0; g.alpha = 1.f;
  g.group_m = gemm_group_m();
  g.rope_cos = cos_t; g.rope_sin = sin_t; g.rope_seq = seq; g.rope_pos0 = pos_offset;
  g.a_scale = row_scale;
  MP_REQUIRE(gemm_variant() == 2 && mp_gemm320_eligible(g, 1), MP_ERR_SHAPE, "mp_gemm_qkv_rope_scaled_bf16: M=%d N=%d K=%d is not a 320-row-kernel shape (M >= 1024, N %% 256 == 0)", M, N, K);
  g_last_gemm_kernel = 320;
  return mp_launch_gemm320(g, 1, stream);
}

// gate|up projection of a TRAINING forward: act = silu(gate) * up from the fused epilogue AND the bf16 gate|up values themselves (the
// backward's operands), one launch instead of GEMM + mp_swiglu_pair_fwd_bf16 (which re-read the [tokens, 2 ff] tensor).
extern "C" int mp_gemm_swiglu_keep_bf16(const void* A, int64_t lda, const void* W, int64_t ldw, void* act_out, int64_t ld_act, void* gu_out,
                                        int64_t ld_gu, int M, int N, int K, hipStream_t stream) {
  MP_REQUIRE(M >= 0 && N > 0 && K > 0 && K % BK == 0 && N % 64 == 0, MP_ERR_SHAPE, "mp_gemm_swiglu_keep_bf16: K %% %d == 0 and N %% 64 == 0", BK);
  MP_REQUIRE(lda % 8 == 0 && ldw % 8 == 0 && ld_act % 8 == 0 && ld_gu % 8 == 0 && act_out && gu_out &&
                 (reinterpret_cast<uintptr_t>(act_out) & 15) == 0 && (reinterpret_cast<uintptr_t>(gu_out) & 15) == 0,
             MP_ERR_ARG, "mp_gemm_swiglu_keep_bf16: strides must be multiples of 8, outputs 16-byte aligned");
  if (M == 0) return MP_OK;
  GemmArgs g{};
  g.A = (const bf16_t*)A; g.lda = lda; g.W = (const bf16_t*)W; g.ldw = ldw; g.C = act_out; g.ldc = ld_act;
  g.M = M; g.N = N; g.K = K; g.act = ACT_SWIGLU_PAIR; g.out_f32 = 0; g.alpha = 1.f;
  g.group_m = gemm_group_m();
  g.keep_gu = (bf16_t*)gu_out; g.ld_gu = ld_gu;
  if (use_320(g, 1)) { g_last_gemm_kernel = 320; return mp_launch_gemm320(g, 1, stream); }
  (void)use_256(g, 1);
  return mp_launch_gemm256(g, 1, stream);
}

// batched variant: `batch` independent problems at fixed element strides (expert GEMMs: one launch over all experts,
// with per-expert device-side row counts m_dev[b]).
extern "C" int mp_gemm_bf16_nt_batched(const void* A, int64_t lda, int64_t strideA, const void* W, int64_t ldw,
                                       int64_t strideW, void* C, int64_t ldc, int64_t strideC, const float* bias,
                                       int64_t strideBias, int batch, int M, int N, int K, int act, int out_dtype,
                                       const int* m_dev, hipStream_t stream) {
  MP_REQUIRE(M >= 0 && N > 0 && K > 0 && batch > 0, MP_ERR_SHAPE, "mp_gemm_bf16_nt_batched: bad shape");
  MP_REQUIRE(K % BK == 0, MP_ERR_SHAPE, "mp_gemm_bf16_nt_batched: K=%d must be a multiple of %d", K, BK);
  MP_REQUIRE(lda % 8 == 0 && ldw % 8 == 0 && strideA % 8 == 0 && strideW % 8 == 0, MP_ERR_SHAPE,
             "mp_gemm_bf16_nt_batched: strides must be multiples of 8");
  MP_REQUIRE(out_dtype == MP_BF16 || out_dtype == MP_F32, MP_ERR_DTYPE, "mp_gemm_bf16_nt_batched: bad out dtype");
  MP_REQUIRE(act >= 0 && act <= 5, MP_ERR_ARG, "mp_gemm_bf16_nt_batched: bad activation %d", act);
  MP_REQUIRE(act != ACT_SWIGLU_PAIR || (N % 64 == 0 && out_dtype == MP_BF16 && ldc % 8 == 0), MP_ERR_ARG,
             "mp_gemm_bf16_nt_batched: SWIGLU_PAIR needs N %% 64 == 0 and a bf16 [M, N/2] output");
  if (M == 0) return MP_OK;
  GemmArgs g{};
  g.A = (const bf16_t*)A; g.lda = lda; g.W = (const bf16_t*)W; g.ldw = ldw; g.C = C; g.ldc = ldc;
  g.bias = bias; g.residual = nullptr; g.ldr = 0; g.m_dev = m_dev; g.M = M; g.N = N; g.K = K;
  g.act = act; g.out_f32 = (out_dtype == MP_F32); g.alpha = 1.f;
  g.sA = strideA; g.sW = strideW; g.sC = strideC; g.sR = 0; g.sBias = strideBias; g.m_dev_stride = 1; g.group_m = gemm_group_m();
  if (use_320_batched(g, batch)) { g_last_gemm_kernel = 320; return mp_launch_gemm320(g, batch, stream); }
  if (use_256(g, batch)) return mp_launch_gemm256(g, batch, stream);
  const int tiles = (int)(mp_cdiv(M, BM) * mp_cdiv(N, BN));
  launch_gemm(g, dim3(tiles, batch), stream);
  return mp_check_launch("mp_gemm_bf16_nt_batched");
}

// batched variant with a batched residual: C[b] = bf16(A[b] W[b]^T) + R[b] (the per-expert LoRA delta added onto the expert projection's
// output in training, llama_lora.py)
extern "C" int mp_gemm_bf16_nt_batched_res(const void* A, int64_t lda, int64_t strideA, const void* W, int64_t ldw, int64_t strideW, void* C,
                                           int64_t ldc, int64_t strideC, const void* residual, int64_t ldr, int64_t strideR, int batch, int M,
                                           int N, int K, const int* m_dev, hipStream_t stream) {
  MP_REQUIRE(M >= 0 && N > 0 && K > 0 && batch > 0 && K % BK == 0, MP_ERR_SHAPE, "mp_gemm_bf16_nt_batched_res: bad shape");
  MP_REQUIRE(lda % 8 == 0 && ldw % 8 == 0 && strideA % 8 == 0 && strideW % 8 == 0 && residual, MP_ERR_SHAPE,
             "mp_gemm_bf16_nt_batched_res: strides must be multiples of 8, residual required");
  if (M == 0) return MP_OK;
  GemmArgs g{};
  g.A = (const bf16_t*)A; g.lda = lda; g.W = (const bf16_t*)W; g.ldw = ldw; g.C = C; g.ldc = ldc;
  g.bias = nullptr; g.residual = (const bf16_t*)residual; g.ldr = ldr; g.m_dev = m_dev; g.M = M; g.N = N; g.K = K;
  g.act = ACT_NONE; g.out_f32 = 0; g.alpha = 1.f;
  g.sA = strideA; g.sW = strideW; g.sC = strideC; g.sR = strideR; g.sBias = 0; g.m_dev_stride = 1; g.group_m = gemm_group_m();
  if (use_320_batched(g, batch)) { g_last_gemm_kernel = 320; return mp_launch_gemm320(g, batch, stream); }
  if (use_256(g, batch)) return mp_launch_gemm256(g, batch, stream);
  const int tiles = (int)(mp_cdiv(M, BM) * mp_cdiv(N, BN));
  launch_gemm(g, dim3(tiles, batch), stream);
  return mp_check_launch("mp_gemm_bf16_nt_batched_res");
}

// Expert GEMMs with the MoE dispatch / combine folded in (top-1 routing): per expert b, A row r comes from row a_rows[b*rows_stride+r]
// of the shared [tokens, K] activation matrix (a_rows null = A is [batch, M, K] as in the plain batched call), and — when c_rows is
// given — C row r goes to row c_rows[b*rows_stride+r] of the shared [tokens, N] output as residual[row] + c_scale[row] * bf16(acc).
extern "C" int mp_gemm_bf16_nt_batched_rows(const void* A, int64_t lda, int64_t strideA, const int* a_rows, const void* W, int64_t ldw,
                                            int64_t strideW, void* C, int64_t ldc, int64_t strideC, const int* c_rows,
                                            const float* c_scale, const void* residual, int64_t ldr, int rows_stride, int batch, int M,
                                            int N, int K, int act, const int* m_dev, hipStream_t stream) {
  MP_REQUIRE(M >= 0 && N > 0 && K > 0 && batch > 0, MP_ERR_SHAPE, "mp_gemm_bf16_nt_batched_rows: bad shape");
  MP_REQUIRE(K % BK == 0 && lda % 8 == 0 && ldw % 8 == 0 && strideW % 8 == 0, MP_ERR_SHAPE, "mp_gemm_bf16_nt_batched_rows: K %% 64, strides %% 8");
  MP_REQUIRE(act == ACT_NONE || act == ACT_SWIGLU_PAIR, MP_ERR_ARG, "mp_gemm_bf16_nt_batched_rows: activation must be none or SWIGLU_PAIR");
  MP_REQUIRE(act != ACT_SWIGLU_PAIR || (N % 64 == 0 && ldc % 8 == 0 && !c_rows), MP_ERR_ARG, "mp_gemm_bf16_nt_batched_rows: bad SWIGLU_PAIR use");
  MP_REQUIRE(!c_rows || (N % 4 == 0 && ldc % 4 == 0 && (!residual || ldr % 4 == 0)), MP_ERR_SHAPE, "mp_gemm_bf16_nt_batched_rows: scatter needs N, ldc, ldr %% 4 == 0");
  MP_REQUIRE(c_rows || (!c_scale && !residual), MP_ERR_ARG, "mp_gemm_bf16_nt_batched_rows: c_scale / residual come with c_rows");
  if (M == 0) return MP_OK;
  GemmArgs g{};
  g.A = (const bf16_t*)A; g.lda = lda; g.W = (const bf16_t*)W; g.ldw = ldw; g.C = C; g.ldc = ldc;
  g.bias = nullptr; g.residual = (const bf16_t*)residual; g.ldr = ldr; g.m_dev = m_dev; g.M = M; g.N = N; g.K = K;
  g.act = act; g.out_f32 = 0; g.alpha = 1.f;
  g.sA = a_rows ? 0 : strideA; g.sW = strideW; g.sC = c_rows ? 0 : strideC; g.sR = 0; g.sBias = 0; g.m_dev_stride = 1;
  g.group_m = gemm_group_m();
  g.a_rows = a_rows; g.c_rows = c_rows; g.c_scale = c_scale; g.rows_stride = rows_stride;
  if (use_320_batched(g, batch)) { g_last_gemm_kernel = 320; return mp_launch_gemm320(g, batch, stream); }
  if (use_256(g, batch)) return mp_launch_gemm256(g, batch, stream);
  const int tiles = (int)(mp_cdiv(M, BM) * mp_cdiv(N, BN));
  launch_gemm(g, dim3(tiles, batch), stream);
  return mp_check_launch("mp_gemm_bf16_nt_batched_rows");
}

// The expert gate|up projection with the post-attention RMSNorm folded in: A = the RAW residual stream [tokens, K] gathered through a_rows, W = the
// interleaved gate|up weights with the norm weight multiplied into their columns, a_row_scale [tokens] = rstd (mp_rmsnorm_gate_rstd_bf16).  The
// SwiGLU epilogue multiplies the fp32 accumulators by rstd[token of the row] before their bf16 rounding.  SWIGLU_PAIR, 320-row kernel only.
extern "C" int mp_gemm_bf16_nt_batched_rows_scaled(const void* A, int64_t lda, const int* a_rows, const float* a_row_scale, const void* W, int64_t ldw,
                                                   int64_t strideW, void* C, int64_t ldc, int64_t strideC, int rows_stride, int batch, int M, int N,
                                                   int K, const int* m_dev, hipStream_t stream) {
  MP_REQUIRE(M >= 0 && N > 0 && K > 0 && batch > 0 && a_rows && a_row_scale, MP_ERR_ARG, "mp_gemm_bf16_nt_batched_rows_scaled: a_rows and a_row_scale required");
  MP_REQUIRE(K % BK == 0 && lda % 8 == 0 && ldw % 8 == 0 && strideW % 8 == 0 && N % 64 == 0 && ldc % 8 == 0, MP_ERR_SHAPE, "mp_gemm_bf16_nt_batched_rows_scaled: bad strides");
  if (M == 0) return MP_OK;
  GemmArgs g{};
  g.A = (const bf16_t*)A; g.lda = lda; g.W = (const bf16_t*)W; g.ldw = ldw; g.C = C; g.ldc = ldc;
  g.m_dev = m_dev; g.M = M; g.N = N; g.K = K; g.act = ACT_SWIGLU_PAIR; g.out_f32 = 0; g.alpha = 1.f;
  g.sA = 0; g.sW = strideW; g.sC = strideC; g.m_dev_stride = 1; g.group_m = gemm_group_m();
  g.a_rows = a_rows; g.rows_stride = rows_stride; g.a_scale = a_row_scale;
  MP_REQUIRE(gemm_variant() == 2 && mp_gemm320_eligible(g, batch), MP_ERR_SHAPE,
             "mp_gemm_bf16_nt_batched_rows_scaled: M=%d N=%d K=%d batch=%d is not a 320-row-kernel shape (M >= 1024, N %% 256 == 0)", M, N, K, batch);
  g_last_gemm_kernel = 320;
  return mp_launch_gemm320(g, batch, stream);
}

// Whether the two folded-norm GEMM entry points take a call of this size (the host keeps the unfolded path otherwise)
extern "C" int mp_gemm_fold_ok(int M, int N, int K) { return (gemm_variant() == 2 && M >= 1024 && N % 256 == 0 && K % 64 == 0) ? 1 : 0; }
