// The frozen SAM-Med2D image encoder's own kernels (ViT-B @ 256 px: 16 x 16 tokens, 768 channels, 12 heads of 64, windows of 14;
// model/segment_anything_med2d/modeling/image_encoder.py:18-56 Adapter_Layer, :165-238 Block, :241-296 Attention, :299-345 window
// partition / unpartition, :348-421 decomposed relative position).  The encoder runs beside the decoder on a side stream, where what it costs
// the step is the CU x time of its workgroups: round 6 replaces ~350 generic launches (window partition of the activations, a rel-pos table
// kernel, the general attention kernel on padded windows, unpartition + add, token mean, two fp32 GEMMs, a channel scale, five im2col and
// four scatter launches, three LayerNorm / add passes per block) by the kernels of this file — twelve launches per block with the GEMMs.
//
//  * sam_attn_kernel: window AND global attention on the un-partitioned [B * 256, 3 * 768] qkv tensor.  The reference pads every 16 x 16 map
//    to 28 x 28 and partitions it into four 14 x 14 windows AFTER norm1, so 528 of a map's 784 window tokens are zero vectors whose q / k / v
//    rows equal the qkv bias: they are never computed here (the qkv and proj GEMMs run on 2048 rows instead of 6272) — a padded KEY is the
//    bf16-rounded bias row, exactly what the GEMM epilogue produced for it, and a padded QUERY's output is cropped by window_unpartition, so
//    it has no workgroup.  The decomposed rel-pos bias is computed in the kernel (q . Rh[qy - ky + n - 1], q . Rw[qx - kx + n - 1], fp32,
//    the channel order of the table kernel it replaces).  One workgroup per (image, head) stages the map's k / v rows once; its eight waves walk the
//    18 (window) or 16 (global) query tiles, each reading its window's keys through the window -> map index; the softmax is single-pass.
//  * sam_ln_colsum_kernel: norm2 and the per-image column sums of its output (the adapter's AdaptiveAvgPool2d) in one pass.
//  * sam_gate_kernel: mean -> Linear(768, 192) -> ReLU -> Linear(192, 768) -> Sigmoid, one workgroup per image (transposed weights).
//  * sam_im2col_scaled_kernel: the 3 x 3 / stride 2 im2col of (gate * x) — the channel scale is applied on the way.
//  * sam_im2col_parity4_kernel: the four output-parity tap gathers of ConvTranspose2d(k 4, s 2, p 1) in one launch.
//  * sam_block_tail_kernel: x + relu(convT) -> Adapter.norm -> shortcut + mlp + adapter -> the NEXT block's norm1, one pass over the rows.
// Rounding points are those of the launches replaced (every intermediate tensor of the old path was bf16 and is rounded to bf16 here).
#include "common.h"

namespace {

typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;

struct SamAttnArgs {
  const bf16_t* qkv; int64_t ld;      // [B * G * G, 3 * H * 64], image order
  const float* bias;                   // [3 * H * 64]
  const float* rph; const float* rpw;  // [2 N - 1, 64] each
  bf16_t* out; int64_t ldo;            // [B * G * G, H * 64], image order
  int B, H, G, nwin;                   // nwin windows per side (1 = global)
  float scale;
};

__device__ __forceinline__ int k_off64(int r, int c) { return r * 128 + ((c ^ (r & 7)) << 4); }   // 16-byte chunk c of key row r, XOR-swizzled

// One workgroup of 8 waves per (image, head): the map's 256 k / v rows are staged ONCE (+ one pad row = the bf16 qkv bias), every 16-query tile of
// every window reads its window's keys through the window -> map index (a padded key is the pad row).  NKF: key fragments of 16 per window
// (NKF * 16 >= N * N); N: window side (14), or the map side (16) for global attention.
constexpr int SAM_WAVES = 8, SAM_TOK = 256, SAM_PAD = 256;
template <int NKF, int N>
__global__ __launch_bounds__(512) void sam_attn_kernel(SamAttnArgs a) {
  constexpr int S = N * N;                     // keys of a window
  constexpr int KR = NKF * 16;
  constexpr int KP = (KR + 31) / 32 * 32;      // keys of the PV product (k steps of 32): P columns beyond S are zero, their V rows the pad row
  constexpr int NT = 2 * N - 1;                // rows of a rel-pos table
  constexpr int TS = 68;                       // float stride of a table row / a staged query row (16-byte aligned rows)
  constexpr int RS = 33;                       // float stride of a query's [rel_h | rel_w] row
  constexpr int PW = 16 * KP * 2;              // bytes of a wave's P region
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sK = smem;                             // [257][64] bf16, swizzled
  char* sV = sK + (SAM_TOK + 1) * 128;         // [257][64] bf16
  char* sPall = sV + (SAM_TOK + 1) * 128;      // per wave: P [16][KP] bf16; before P is written the same bytes hold the wave's fp32 query rows and its rel rows
  float* sT = reinterpret_cast<float*>(sPall + SAM_WAVES * PW);      // th [NT][68] | tw [NT][68]
  unsigned short* sYX = reinterpret_cast<unsigned short*>(sT + 2 * NT * TS);     // [KP]: key j of a window -> (j / N) | (j % N) << 8; 0xffff beyond S (a table: the
                                                                                 // constant divisions, hoisted out of the tile loop, cost ~100 live registers at N = 14)
  static_assert(PW >= (16 * TS + 16 * RS) * 4, "the query staging must fit the wave's P region");

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fq = lane >> 4;
  const int C = a.H * 64;
  const int b = blockIdx.x / a.H, h = blockIdx.x % a.H;
  const int64_t tok0 = (int64_t)b * a.G * a.G;

  // ---- stage the map's K and V rows, the pad row, the two tables
  {
    bf16x8 kreg[4], vreg[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int id = tid + i * 512, j = id >> 3, c = id & 7;
      const bf16_t* row = a.qkv + (tok0 + j) * a.ld + (int64_t)h * 64 + c * 8;
      kreg[i] = *reinterpret_cast<const bf16x8*>(row + C);
      vreg[i] = *reinterpret_cast<const bf16x8*>(row + 2 * C);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int id = tid + i * 512, j = id >> 3, c = id & 7;
      *reinterpret_cast<bf16x8*>(sK + k_off64(j, c)) = kreg[i];
      *reinterpret_cast<bf16x8*>(sV + j * 128 + c * 16) = vreg[i];
    }
    if (tid < 16) {                             // a padded token: zero input row -> the projection's bias, rounded like the GEMM epilogue rounds
      const int c = tid & 7;
      const float* bk = a.bias + (tid < 8 ? C : 2 * C) + h * 64 + c * 8;
      bf16x8 v;
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = (bf16_t)bk[e];
      if (tid < 8) *reinterpret_cast<bf16x8*>(sK + k_off64(SAM_PAD, c)) = v;
      else *reinterpret_cast<bf16x8*>(sV + SAM_PAD * 128 + c * 16) = v;
    }
    for (int i = tid; i < NT * 64; i += 512) {
      sT[(i >> 6) * TS + (i & 63)] = a.rph[i];
      sT[(NT + (i >> 6)) * TS + (i & 63)] = a.rpw[i];
    }
    if (tid < KP) sYX[tid] = tid < S ? (unsigned short)((tid / N) | ((tid % N) << 8)) : (unsigned short)0xffff;
  }
  __syncthreads();                              // the only workgroup-wide synchronisation: everything below is per wave

  // tiles of 16 queries: window w has ceil(valid queries / 16) of them (scalars, not an indexed array: no scratch)
  int te0 = 0, te1 = 0, te2 = 0, te3 = 0;
  {
    int acc = 0;
    for (int w = 0; w < 4; ++w) {
      if (w < a.nwin * a.nwin) {
        const int vy = min(N, a.G - (w / a.nwin) * N), vx = min(N, a.G - (w % a.nwin) * N);
        acc += (vy * vx + 15) / 16;
      }
      if (w == 0) te0 = acc; else if (w == 1) te1 = acc; else if (w == 2) te2 = acc; else te3 = acc;
    }
  }
  const int n_tiles = te3;
  float* sQ = reinterpret_cast<float*>(sPall + wave * PW);               // [16][68]
  float* sR = sQ + 16 * TS;                                              // [16][33]: rel_h[0..N) | rel_w[0..N)
  bf16_t* pw = reinterpret_cast<bf16_t*>(sPall + wave * PW);

  for (int t = wave; t < n_tiles; t += SAM_WAVES) {
    const int w = t < te0 ? 0 : (t < te1 ? 1 : (t < te2 ? 2 : 3));
    const int wy = w / a.nwin, wx = w % a.nwin;
    const int vy = min(N, a.G - wy * N), vx = min(N, a.G - wx * N), nq = vy * vx;
    const int qi0 = (t - (w == 0 ? 0 : (w == 1 ? te0 : (w == 2 ? te1 : te2)))) * 16;
    // key (ky, kx) of the window -> its row in the staged map (the pad row for a padded key and beyond the window's S keys)
    auto krow = [&](unsigned yx) {
      const int ty = wy * N + (int)(yx & 255u), tx = wx * N + (int)(yx >> 8);
      return (ty < a.G && tx < a.G) ? ty * a.G + tx : SAM_PAD;
    };
    // ---- the tile's queries: MFMA fragments in registers, fp32 rows in LDS for the rel-pos dot products
    bf16x8 qf[2];
    const int qc = min(qi0 + fr, nq - 1);
    const int iy = qc / vx, ix = qc % vx;
    {
      const bf16_t* qrow = a.qkv + (tok0 + (wy * N + iy) * a.G + wx * N + ix) * a.ld + (int64_t)h * 64;
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        qf[kk] = *reinterpret_cast<const bf16x8*>(qrow + kk * 32 + fq * 8);
        f32x4 lo, hi;
#pragma unroll
        for (int e = 0; e < 4; ++e) { lo[e] = (float)qf[kk][e]; hi[e] = (float)qf[kk][4 + e]; }
        *reinterpret_cast<f32x4*>(sQ + fr * TS + kk * 32 + fq * 8) = lo;
        *reinterpret_cast<f32x4*>(sQ + fr * TS + kk * 32 + fq * 8 + 4) = hi;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // ---- rel rows: lane (q = fr, group fq) takes k = fq, fq + 4, ... of the 2 N outputs of query q; the query row in registers, 64 channels ascending
    {
      f32x4 qv[16];
#pragma unroll
      for (int c = 0; c < 16; ++c) qv[c] = *reinterpret_cast<const f32x4*>(sQ + fr * TS + c * 4);
#pragma unroll 1
      for (int i = 0; i < (2 * N + 3) / 4; ++i) {
        const int k = fq + 4 * i;
        if (k < 2 * N) {
          const float* row = k < N ? sT + (iy - k + N - 1) * TS : sT + (NT + ix - (k - N) + N - 1) * TS;
          float acc = 0.f;
#pragma unroll
          for (int c = 0; c < 16; ++c) {
            const f32x4 rv = *reinterpret_cast<const f32x4*>(row + c * 4);
            acc = fmaf(qv[c][0], rv[0], acc); acc = fmaf(qv[c][1], rv[1], acc); acc = fmaf(qv[c][2], rv[2], acc); acc = fmaf(qv[c][3], rv[3], acc);
          }
          sR[fr * RS + k] = acc;
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();

    // ---- S = Q K^T: s[n][r] = score of query row fq * 4 + r against key n * 16 + fr
    f32x4 s[NKF];
#pragma unroll
    for (int n = 0; n < NKF; ++n) s[n] = f32x4{0.f, 0.f, 0.f, 0.f};
    unsigned yx[NKF];
#pragma unroll
    for (int n = 0; n < NKF; ++n) yx[n] = sYX[n * 16 + fr];
#pragma unroll
    for (int n = 0; n < NKF; ++n) {
      const int r = krow(yx[n]);
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        const bf16x8 kf = *reinterpret_cast<const bf16x8*>(sK + k_off64(r, kk * 4 + fq));
        s[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qf[kk], kf, s[n], 0, 0, 0);
      }
      if ((n & 3) == 3) __builtin_amdgcn_sched_barrier(0);       // at most four key fragments' reads in flight: the scheduler otherwise hoists all of them (spills at N = 14)
    }
    // ---- scale, bias, single-pass softmax
    float inv_l[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float* rr = sR + (fq * 4 + r) * RS;
      float mx = -INFINITY;
#pragma unroll
      for (int n = 0; n < NKF; ++n) {
        const int kj = n * 16 + fr;
        float v = s[n][r] * a.scale;
        if (kj < S) v += rr[yx[n] & 255u] + rr[N + (yx[n] >> 8)];
        else v = -INFINITY;
        s[n][r] = v;
        mx = fmaxf(mx, v);
      }
#pragma unroll
      for (int off = 1; off < 16; off <<= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
      float rs = 0.f;
#pragma unroll
      for (int n = 0; n < NKF; ++n) {
        const float p = __expf(s[n][r] - mx);
        s[n][r] = p;
        rs += p;
      }
#pragma unroll
      for (int off = 1; off < 16; off <<= 1) rs += __shfl_xor(rs, off, 64);
      inv_l[r] = 1.f / rs;
      __builtin_amdgcn_sched_barrier(0);           // one query row's rel reads at a time
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();               // every rel-row read of the wave is done: its bytes become P

    // ---- P (bf16) in A-operand order, then O = P V through the hardware transpose read of V
#pragma unroll
    for (int n = 0; n < NKF; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r) pw[(fq * 4 + r) * KP + n * 16 + fr] = (bf16_t)s[n][r];
    if constexpr (KP > KR) {
#pragma unroll
      for (int r = 0; r < 4; ++r) pw[(fq * 4 + r) * KP + KR + fr] = (bf16_t)0.f;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    f32x4 o[4];
#pragma unroll
    for (int n = 0; n < 4; ++n) o[n] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kk = 0; kk < KP / 32; ++kk) {
      const bf16x8 pf = *reinterpret_cast<const bf16x8*>(pw + fr * KP + kk * 32 + fq * 8);
      const int key0 = kk * 32 + fq * 8 + (fr >> 2);
      const char* v0 = sV + krow(sYX[key0]) * 128 + (fr & 3) * 8;
      const char* v1 = sV + krow(sYX[key0 + 4]) * 128 + (fr & 3) * 8;
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(v0 + n * 32));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(v1 + n * 32));
        const s16x8 both = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        o[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pf, __builtin_bit_cast(bf16x8, both), o[n], 0, 0, 0);
      }
      if (kk & 1) __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int qi = qi0 + fq * 4 + r;
      if (qi >= nq) continue;
      const int oy = qi / vx, ox = qi % vx;
      bf16_t* orow = a.out + (tok0 + (wy * N + oy) * a.G + wx * N + ox) * a.ldo + (int64_t)h * 64;
#pragma unroll
      for (int n = 0; n < 4; ++n) orow[n * 16 + fr] = (bf16_t)(o[n][r] * inv_l[r]);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();               // the P reads are done before the next tile's query rows overwrite the region
  }
}

// ------------------------------------------------------------------------------------------------------------------ row kernels (768 channels)
// One wave per row, the lane / chunk assignment and the arithmetic order of layernorm_bf16_wave_kernel (norm_elementwise.hip): lane l holds
// elements (c * 64 + l) * 8 .. + 8 of chunk c.
constexpr int WC = 2;                           // 768 = 64 lanes x 8 + 32 lanes x 8
struct Row768 { bf16x8 v[WC]; };
__device__ __forceinline__ bool chunk_on(int c, int lane, int dim) { return (c * 64 + lane) * 8 < dim; }
__device__ __forceinline__ Row768 load_row(const bf16_t* p, int lane, int dim) {
  Row768 r;
#pragma unroll
  for (int c = 0; c < WC; ++c) {
    if (chunk_on(c, lane, dim)) r.v[c] = *reinterpret_cast<const bf16x8*>(p + (c * 64 + lane) * 8);
    else {
#pragma unroll
      for (int j = 0; j < 8; ++j) r.v[c][j] = (bf16_t)0.f;
    }
  }
  return r;
}
__device__ __forceinline__ void store_row(bf16_t* p, const Row768& r, int lane, int dim) {
#pragma unroll
  for (int c = 0; c < WC; ++c)
    if (chunk_on(c, lane, dim)) *reinterpret_cast<bf16x8*>(p + (c * 64 + lane) * 8) = r.v[c];
}
__device__ __forceinline__ Row768 layernorm_row(const Row768& x, const float* __restrict__ w, const float* __restrict__ b, float eps, int lane, int dim) {
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < WC; ++c)
    if (chunk_on(c, lane, dim)) {
#pragma unroll
      for (int j = 0; j < 8; ++j) s += (float)x.v[c][j];
    }
  const float mean = wave_sum(s) / (float)dim;
  float q = 0.f;
#pragma unroll
  for (int c = 0; c < WC; ++c)
    if (chunk_on(c, lane, dim)) {
#pragma unroll
      for (int j = 0; j < 8; ++j) { const float d = (float)x.v[c][j] - mean; q += d * d; }
    }
  const float rs = rsqrtf(wave_sum(q) / (float)dim + eps);
  Row768 y;
#pragma unroll
  for (int c = 0; c < WC; ++c) {
    const int i = (c * 64 + lane) * 8;
    if (chunk_on(c, lane, dim)) {
      const f32x4 w0 = *reinterpret_cast<const f32x4*>(w + i), w1 = *reinterpret_cast<const f32x4*>(w + i + 4);
      const f32x4 b0 = *reinterpret_cast<const f32x4*>(b + i), b1 = *reinterpret_cast<const f32x4*>(b + i + 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        y.v[c][j] = (bf16_t)(((float)x.v[c][j] - mean) * rs * w0[j] + b0[j]);
        y.v[c][4 + j] = (bf16_t)(((float)x.v[c][4 + j] - mean) * rs * w1[j] + b1[j]);
      }
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) y.v[c][j] = (bf16_t)0.f;
    }
  }
  return y;
}

// y = LayerNorm(x [+ addend[row % period]]) (the sum, rounded to bf16, is written to xsum when an addend is given: the patch embedding + pos_embed)
__global__ __launch_bounds__(256) void sam_add_ln_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ addend, int period, bf16_t* __restrict__ xsum,
                                                         const float* __restrict__ w, const float* __restrict__ b, float eps, bf16_t* __restrict__ y, int rows, int dim) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  Row768 v = load_row(x + (int64_t)row * dim, lane, dim);
  if (addend) {
    const Row768 p = load_row(addend + (int64_t)(row % period) * dim, lane, dim);
#pragma unroll
    for (int c = 0; c < WC; ++c)
#pragma unroll
      for (int j = 0; j < 8; ++j) v.v[c][j] = (bf16_t)((float)v.v[c][j] + (float)p.v[c][j]);
    store_row(xsum + (int64_t)row * dim, v, lane, dim);
  }
  store_row(y + (int64_t)row * dim, layernorm_row(v, w, b, eps, lane, dim), lane, dim);
}

// xn = LayerNorm(x); part[b][k][c] = sum of xn over the 16 rows of slab k of image b (T = 256 tokens -> 16 slabs; fixed order: the four rows of
// a wave in row order, then the four waves in wave order)
__global__ __launch_bounds__(256) void sam_ln_colsum_kernel(const bf16_t* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b, float eps,
                                                            bf16_t* __restrict__ xn, float* __restrict__ part, int rows, int dim) {
  __shared__ float red[4][768];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float acc[WC][8];
#pragma unroll
  for (int c = 0; c < WC; ++c)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[c][j] = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = blockIdx.x * 16 + wave * 4 + i;
    if (row < rows) {
      const Row768 y = layernorm_row(load_row(x + (int64_t)row * dim, lane, dim), w, b, eps, lane, dim);
      store_row(xn + (int64_t)row * dim, y, lane, dim);
#pragma unroll
      for (int c = 0; c < WC; ++c)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[c][j] += (float)y.v[c][j];
    }
  }
#pragma unroll
  for (int c = 0; c < WC; ++c)
    if (chunk_on(c, lane, dim)) {
#pragma unroll
      for (int j = 0; j < 8; ++j) red[wave][(c * 64 + lane) * 8 + j] = acc[c][j];
    }
  __syncthreads();
  for (int c = threadIdx.x; c < dim; c += 256) part[(int64_t)blockIdx.x * dim + c] = ((red[0][c] + red[1][c]) + red[2][c]) + red[3][c];
}

// gate[b][c] = sigmoid(sum_j w2t[j][c] relu(sum_c' w1t[c'][j] mean_b[c'])), mean_b[c] = sum_k part[b][k][c] / T.  One workgroup of 1024 threads per image;
// the weights come TRANSPOSED (w1t [C, Hd], w2t [Hd, C]) so that consecutive threads read consecutive addresses.  The hidden layer is summed in four
// quarters of the channel range by four thread groups and combined in a fixed order.
__global__ __launch_bounds__(1024) void sam_gate_kernel(const float* __restrict__ part, int slabs, int T, const float* __restrict__ w1t, const float* __restrict__ w2t,
                                                        float* __restrict__ gate, int C, int Hd) {
  __shared__ float pooled[1024];
  __shared__ float hq[4][256];
  __shared__ float hid[256];
  const int b = blockIdx.x, t = threadIdx.x;
  if (t < C) {
    float s = 0.f;
    for (int k = 0; k < slabs; ++k) s += part[((int64_t)b * slabs + k) * C + t];
    pooled[t] = s / (float)T;
  }
  __syncthreads();
  const int q = t / Hd, j = t - q * Hd, cq = C / 4;
  if (q < 4) {
    float acc = 0.f;
    const float* wp = w1t + (int64_t)q * cq * Hd + j;
#pragma unroll 8
    for (int c = 0; c < cq; ++c) acc = fmaf(wp[(int64_t)c * Hd], pooled[q * cq + c], acc);
    hq[q][j] = acc;
  }
  __syncthreads();
  if (t < Hd) hid[t] = fmaxf(((hq[0][t] + hq[1][t]) + hq[2][t]) + hq[3][t], 0.f);
  __syncthreads();
  if (t < C) {
    float acc = 0.f;
#pragma unroll 8
    for (int jj = 0; jj < Hd; ++jj) acc = fmaf(w2t[(int64_t)jj * C + t], hid[jj], acc);
    gate[(int64_t)b * C + t] = sigmoidf_(acc);
  }
}

// cols[(b, oy, ox), t * C + c] = bf16(gate[b, c] * x[b, 2 oy + dy_t, 2 ox + dx_t, c]) for the nine taps of a 3 x 3 / stride 2 / pad 1 convolution
__global__ void sam_im2col_scaled_kernel(const bf16_t* __restrict__ x, const float* __restrict__ gate, bf16_t* __restrict__ out, int B, int G, int C) {
  const int OH = G / 2, per_pix = 9 * (C / 8);
  const int64_t total = (int64_t)B * OH * OH * per_pix;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int q = (int)(idx % per_pix);
  const int64_t pix = idx / per_pix;
  const int t = q / (C / 8), c = (q % (C / 8)) * 8;
  const int ox = (int)(pix % OH), oy = (int)((pix / OH) % OH), b = (int)(pix / ((int64_t)OH * OH));
  const int iy = oy * 2 + t / 3 - 1, ix = ox * 2 + t % 3 - 1;
  bf16x8 o;
  if (iy >= 0 && iy < G && ix >= 0 && ix < G) {
    const bf16x8 v = *reinterpret_cast<const bf16x8*>(x + (((int64_t)b * G + iy) * G + ix) * C + c);
    const f32x4 g0 = *reinterpret_cast<const f32x4*>(gate + (int64_t)b * C + c), g1 = *reinterpret_cast<const f32x4*>(gate + (int64_t)b * C + c + 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) { o[j] = (bf16_t)(g0[j] * (float)v[j]); o[4 + j] = (bf16_t)(g1[j] * (float)v[4 + j]); }
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (bf16_t)0.f;
  }
  *reinterpret_cast<bf16x8*>(out + pix * ((int64_t)9 * C) + (int64_t)t * C + c) = o;
}

// ConvTranspose2d(k 4, s 2, p 1) as four stride-1 GEMMs, one per output parity (py, px): output row 2 m + py takes kernel rows ky with
// 2 iy - 1 + ky = 2 m + py: py = 0 -> (ky 1, iy m), (ky 3, iy m - 1); py = 1 -> (ky 0, iy m + 1), (ky 2, iy m).  cols[cls][(b, m, n), t * C + c] =
// s1[b, m + dy, n + dx, c] with tap t = 2 * (row tap) + (column tap) in that order (the weight packing of model/sam.py:_convt_parity_taps).
__global__ void sam_im2col_parity4_kernel(const bf16_t* __restrict__ s1, bf16_t* __restrict__ out, int B, int Hh, int C) {
  const int per_pix = 4 * (C / 8);
  const int64_t per_cls = (int64_t)B * Hh * Hh * per_pix;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= 4 * per_cls) return;
  const int cls = (int)(idx / per_cls);
  const int64_t id = idx % per_cls;
  const int q = (int)(id % per_pix);
  const int64_t pix = id / per_pix;
  const int t = q / (C / 8), c = (q % (C / 8)) * 8;
  const int n = (int)(pix % Hh), m = (int)((pix / Hh) % Hh), b = (int)(pix / ((int64_t)Hh * Hh));
  const int py = cls >> 1, px = cls & 1, ty = t >> 1, tx = t & 1;
  const int dy = py == 0 ? (ty == 0 ? 0 : -1) : (ty == 0 ? 1 : 0);
  const int dx = px == 0 ? (tx == 0 ? 0 : -1) : (tx == 0 ? 1 : 0);
  const int iy = m + dy, ix = n + dx;
  bf16x8 v;
  if (iy >= 0 && iy < Hh && ix >= 0 && ix < Hh) v = *reinterpret_cast<const bf16x8*>(s1 + (((int64_t)b * Hh + iy) * Hh + ix) * C + c);
  else {
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (bf16_t)0.f;
  }
  *reinterpret_cast<bf16x8*>(out + (int64_t)cls * B * Hh * Hh * (4 * C) + pix * ((int64_t)4 * C) + (int64_t)t * C + c) = v;
}

// The end of a block, one wave per token row:  t = bf16(xn + y4[parity of the token][its half-resolution pixel])  (x + spatial(x), Adapter_Layer)
//   ad = LayerNorm(t; Adapter.norm)      x' = bf16(x + mlp + ad)      h = LayerNorm(x'; the NEXT block's norm1)  (skipped when nw is null)
__global__ __launch_bounds__(256) void sam_block_tail_kernel(const bf16_t* __restrict__ y4, const bf16_t* __restrict__ xn, const bf16_t* __restrict__ x,
                                                             const bf16_t* __restrict__ mlp, const float* __restrict__ aw, const float* __restrict__ ab, float aeps,
                                                             const float* __restrict__ nw, const float* __restrict__ nb, float neps, bf16_t* __restrict__ xout,
                                                             bf16_t* __restrict__ hout, int B, int G, int dim) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= B * G * G) return;
  const int xx = row % G, yy = (row / G) % G, b = row / (G * G), Hh = G / 2;
  const int cls = (yy & 1) * 2 + (xx & 1);
  const int64_t src = (int64_t)cls * B * Hh * Hh + ((int64_t)b * Hh + (yy >> 1)) * Hh + (xx >> 1);
  Row768 t = load_row(y4 + src * dim, lane, dim);
  const Row768 n2 = load_row(xn + (int64_t)row * dim, lane, dim);
#pragma unroll
  for (int c = 0; c < WC; ++c)
#pragma unroll
    for (int j = 0; j < 8; ++j) t.v[c][j] = (bf16_t)((float)t.v[c][j] + (float)n2.v[c][j]);
  const Row768 ad = layernorm_row(t, aw, ab, aeps, lane, dim);
  const Row768 xs = load_row(x + (int64_t)row * dim, lane, dim), m = load_row(mlp + (int64_t)row * dim, lane, dim);
  Row768 xo;
#pragma unroll
  for (int c = 0; c < WC; ++c)
#pragma unroll
    for (int j = 0; j < 8; ++j) xo.v[c][j] = (bf16_t)((float)xs.v[c][j] + (float)m.v[c][j] + (float)ad.v[c][j]);
  store_row(xout + (int64_t)row * dim, xo, lane, dim);
  if (nw) store_row(hout + (int64_t)row * dim, layernorm_row(xo, nw, nb, neps, lane, dim), lane, dim);
}

template <int NKF, int N>
int launch_sam_attn(const SamAttnArgs& a, hipStream_t stream) {
  constexpr int KR = NKF * 16, KP = (KR + 31) / 32 * 32, NT = 2 * N - 1;
  constexpr int bytes = 2 * (SAM_TOK + 1) * 128 + SAM_WAVES * 16 * KP * 2 + 2 * NT * 68 * 4 + KP * 2;
  static_assert(bytes <= 160 * 1024, "one workgroup per CU");
  static bool attr[64] = {};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev >= 0 && dev < 64 && !attr[dev]) {
    (void)hipFuncSetAttribute((const void*)sam_attn_kernel<NKF, N>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    attr[dev] = true;
  }
  hipLaunchKernelGGL((sam_attn_kernel<NKF, N>), dim3((unsigned)(a.B * a.H)), dim3(512), bytes, stream, a);
  return mp_check_launch("mp_sam_attention_bf16");
}

}  // namespace

// C-ABI: see include/medplib_hip.h
extern "C" int mp_sam_attention_bf16(const void* qkv, int64_t ld_qkv, const float* qkv_bias, const float* rel_pos_h, const float* rel_pos_w, void* out,
                                     int64_t ld_out, int B, int heads, int grid, int window, float scale, hipStream_t stream) {
  MP_REQUIRE(qkv && qkv_bias && rel_pos_h && rel_pos_w && out && B > 0 && heads > 0, MP_ERR_ARG, "mp_sam_attention_bf16: null / empty argument");
  MP_REQUIRE(ld_qkv % 8 == 0 && ld_qkv >= 3 * heads * 64 && ld_out >= heads * 64, MP_ERR_SHAPE, "mp_sam_attention_bf16: head_dim is 64; row strides must cover the heads (ld_qkv %% 8 == 0)");
  MP_REQUIRE((reinterpret_cast<uintptr_t>(qkv) & 15) == 0 && (reinterpret_cast<uintptr_t>(qkv_bias) & 15) == 0, MP_ERR_ARG, "mp_sam_attention_bf16: 16-byte aligned operands");
  MP_REQUIRE(grid == 16 && (window == 14 || window == 0), MP_ERR_SHAPE,
             "mp_sam_attention_bf16: built for the SAM-Med2D geometry (16 x 16 tokens, windows of 14 or global); got grid %d window %d", grid, window);
  SamAttnArgs a{(const bf16_t*)qkv, ld_qkv, qkv_bias, rel_pos_h, rel_pos_w, (bf16_t*)out, ld_out, B, heads, grid, 1, scale};
  if (window == 14) {
    a.nwin = 2;                              // 196, 28, 28 and 4 valid queries: 13 + 2 + 2 + 1 tiles of 16
    return launch_sam_attn<13, 14>(a, stream);
  }
  return launch_sam_attn<16, 16>(a, stream);
}

extern "C" int mp_sam_add_layernorm_bf16(const void* x, const void* addend, int period, void* xsum, const float* w, const float* b, float eps, void* y,
                                         int rows, int dim, hipStream_t stream) {
  MP_REQUIRE(x && w && b && y && rows >= 0 && dim % 8 == 0 && dim <= 1024 && dim > 0, MP_ERR_SHAPE, "mp_sam_add_layernorm_bf16: dim %% 8 == 0, dim <= 1024");
  MP_REQUIRE(!addend || (xsum && period > 0), MP_ERR_ARG, "mp_sam_add_layernorm_bf16: an addend needs its period and the sum's output");
  if (rows == 0) return MP_OK;
  hipLaunchKernelGGL(sam_add_ln_kernel, dim3((unsigned)mp_cdiv(rows, 4)), dim3(256), 0, stream, (const bf16_t*)x, (const bf16_t*)addend, period, (bf16_t*)xsum, w, b,
                     eps, (bf16_t*)y, rows, dim);
  return mp_check_launch("mp_sam_add_layernorm_bf16");
}

extern "C" int mp_sam_layernorm_colsum_bf16(const void* x, const float* w, const float* b, float eps, void* xn, float* part, int rows, int dim, hipStream_t stream) {
  MP_REQUIRE(x && w && b && xn && part && rows > 0 && rows % 16 == 0 && dim == 768, MP_ERR_SHAPE, "mp_sam_layernorm_colsum_bf16: rows %% 16 == 0, dim == 768");
  hipLaunchKernelGGL(sam_ln_colsum_kernel, dim3((unsigned)(rows / 16)), dim3(256), 0, stream, (const bf16_t*)x, w, b, eps, (bf16_t*)xn, part, rows, dim);
  return mp_check_launch("mp_sam_layernorm_colsum_bf16");
}

extern "C" int mp_sam_channel_gate_f32(const float* part, int slabs, int tokens, const float* w1t, const float* w2t, float* gate, int B, int C, int hidden,
                                       hipStream_t stream) {
  MP_REQUIRE(part && w1t && w2t && gate && B > 0 && slabs > 0 && tokens > 0 && C > 0 && C <= 1024 && C % 4 == 0 && hidden > 0 && hidden <= 256, MP_ERR_SHAPE,
             "mp_sam_channel_gate_f32: C <= 1024, C %% 4 == 0, hidden <= 256");
  hipLaunchKernelGGL(sam_gate_kernel, dim3((unsigned)B), dim3(1024), 0, stream, part, slabs, tokens, w1t, w2t, gate, C, hidden);
  return mp_check_launch("mp_sam_channel_gate_f32");
}

extern "C" int mp_sam_im2col_scaled_bf16(const void* x, const float* gate, void* cols, int B, int grid, int C, hipStream_t stream) {
  MP_REQUIRE(x && gate && cols && B > 0 && grid > 0 && grid % 2 == 0 && C % 8 == 0, MP_ERR_SHAPE, "mp_sam_im2col_scaled_bf16: even grid, C %% 8 == 0");
  const int64_t n = (int64_t)B * (grid / 2) * (grid / 2) * 9 * (C / 8);
  hipLaunchKernelGGL(sam_im2col_scaled_kernel, dim3((unsigned)mp_cdiv(n, 256)), dim3(256), 0, stream, (const bf16_t*)x, gate, (bf16_t*)cols, B, grid, C);
  return mp_check_launch("mp_sam_im2col_scaled_bf16");
}

extern "C" int mp_sam_im2col_parity4_bf16(const void* s1, void* cols4, int B, int half, int C, hipStream_t stream) {
  MP_REQUIRE(s1 && cols4 && B > 0 && half > 0 && C % 8 == 0, MP_ERR_SHAPE, "mp_sam_im2col_parity4_bf16: C %% 8 == 0");
  const int64_t n = (int64_t)4 * B * half * half * 4 * (C / 8);
  hipLaunchKernelGGL(sam_im2col_parity4_kernel, dim3((unsigned)mp_cdiv(n, 256)), dim3(256), 0, stream, (const bf16_t*)s1, (bf16_t*)cols4, B, half, C);
  return mp_check_launch("mp_sam_im2col_parity4_bf16");
}

extern "C" int mp_sam_block_tail_bf16(const void* y4, const void* xn, const void* x, const void* mlp, const float* ad_w, const float* ad_b, float ad_eps,
                                      const float* next_w, const float* next_b, float next_eps, void* x_out, void* h_out, int B, int grid, int dim,
                                      hipStream_t stream) {
  MP_REQUIRE(y4 && xn && x && mlp && ad_w && ad_b && x_out && B > 0 && grid > 0 && grid % 2 == 0 && dim % 8 == 0 && dim <= 1024, MP_ERR_SHAPE,
             "mp_sam_block_tail_bf16: even grid, dim %% 8 == 0, dim <= 1024");
  MP_REQUIRE(!next_w || (next_b && h_out), MP_ERR_ARG, "mp_sam_block_tail_bf16: the next norm needs its bias and an output");
  hipLaunchKernelGGL(sam_block_tail_kernel, dim3((unsigned)mp_cdiv((int64_t)B * grid * grid, 4)), dim3(256), 0, stream, (const bf16_t*)y4, (const bf16_t*)xn,
                     (const bf16_t*)x, (const bf16_t*)mlp, ad_w, ad_b, ad_eps, next_w, next_b, next_eps, (bf16_t*)x_out, (bf16_t*)h_out, B, grid, dim);
  return mp_check_launch("mp_sam_block_tail_bf16");
}
