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
// One persistent workgroup per CU keeps the packed W1 (128 KiB) in LDS and walks 32-token groups; wave `w` owns sub-pixel
// `w` of the first ConvT, so LayerNorm2d is a 64-channel reduction inside the wave (4 fragments x 16 lanes).  The 4x4
// output patches of 16 consecutive tokens of one image row are staged in LDS as [c2][y][64 x] and leave as whole 128-B rows.
#include "common.h"

namespace {

constexpr int UP_THREADS = 256;
constexpr int W1_BYTES = 256 * 256 * 2;        // [n1 = sub*64 + c][k = 256] bf16, 512-B rows, 16-B chunks XOR-swizzled by (n1 & 7)
constexpr int T_BYTES = 4 * 16 * 64 * 2;       // per-wave transposition buffer [16 tokens][64 ch] bf16 (+ swizzle)
constexpr int OUT_BYTES = 32 * 4 * 64 * 2;     // staging [c2][yy][64 x] bf16
constexpr int UP_LDS = W1_BYTES + T_BYTES + OUT_BYTES;   // 131072 + 8192 + 16384 = 155648

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
};

__device__ __forceinline__ int w1_off(int n, int c) { return n * 512 + ((c ^ (n & 7)) << 4); }      // c = 16-B chunk 0..31

__global__ __launch_bounds__(UP_THREADS, 1) void upsample_fused_kernel(UpArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sW1 = smem;
  char* sT = smem + W1_BYTES;
  bf16_t* sOut = reinterpret_cast<bf16_t*>(smem + W1_BYTES + T_BYTES);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fq = lane >> 4;

  // ---- stage W1 once per workgroup (16 KiB per pass of 256 threads x 4 x 16 B)
  for (int id = tid; id < 256 * 32; id += UP_THREADS) {
    const int n = id >> 5, c = id & 31;
    *reinterpret_cast<bf16x8*>(sW1 + w1_off(n, c)) = *reinterpret_cast<const bf16x8*>(a.w1p + (int64_t)n * 256 + c * 8);
  }
  // per-lane constants: bias / LN params of this lane's 4 channels (c = j*16 + fr), W2 fragments (this wave's K slice is all 64)
  float b1v[4], lwv[4], lbv[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) { b1v[j] = a.b1[j * 16 + fr]; lwv[j] = a.lnw[j * 16 + fr]; lbv[j] = a.lnb[j * 16 + fr]; }
  bf16x8 w2f[8][2];           // B fragments of GEMM2: n2 = nf*16 + fr, k chunk kk*4 + fq
#pragma unroll
  for (int nf = 0; nf < 8; ++nf)
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
      w2f[nf][kk] = *reinterpret_cast<const bf16x8*>(a.w2p + (int64_t)(nf * 16 + fr) * 64 + (kk * 4 + fq) * 8);
  float b2v[2];
#pragma unroll
  for (int p = 0; p < 2; ++p) b2v[p] = a.b2[p * 16 + fr];
  __syncthreads();

  const int tokens_per_img = a.h * a.w;
  const int64_t n_tokens = (int64_t)a.B * tokens_per_img;
  const int64_t n_groups = (n_tokens + 15) / 16;
  const int kh = wave >> 1, kw = wave & 1;     // this wave's sub-pixel of the first ConvT
  const int OW = 4 * a.w, OH = 4 * a.h;

  for (int64_t grp = blockIdx.x; grp < n_groups; grp += gridDim.x) {
    const int64_t t0 = grp * 16;
    const int b = (int)(t0 / tokens_per_img);
    const int ti = (int)(t0 % tokens_per_img);
    const int irow = ti / a.w, j0 = ti % a.w;          // 16 consecutive tokens of one image row (w % 16 == 0)
    // ---------------- GEMM1: 16 tokens x 64 channels of sub-pixel `wave` ----------------
    f32x4 acc1[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc1[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const bf16_t* xrow = a.src + (min(t0 + fr, n_tokens - 1)) * 256;
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
      const bf16x8 xa = *reinterpret_cast<const bf16x8*>(xrow + (kk * 4 + fq) * 8);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const bf16x8 wb = *reinterpret_cast<const bf16x8*>(sW1 + w1_off(wave * 64 + j * 16 + fr, kk * 4 + fq));
        acc1[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xa, wb, acc1[j], 0, 0, 0);
      }
    }
    // ---------------- + bias, LayerNorm2d over the 64 channels, GELU (C layout: channel = j*16 + fr, token = fq*4 + r) ----
    bf16_t* tw = reinterpret_cast<bf16_t*>(sT + wave * (16 * 64 * 2));
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float v[4], s = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) { v[j] = acc1[j][r] + b1v[j]; s += v[j]; }
#pragma unroll
      for (int off = 1; off < 16; off <<= 1) s += __shfl_xor(s, off, 64);
      const float mean = s * (1.f / 64.f);
      float q = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) { const float d = v[j] - mean; q += d * d; }
#pragma unroll
      for (int off = 1; off < 16; off <<= 1) q += __shfl_xor(q, off, 64);
      const float rstd = 1.f / sqrtf(q * (1.f / 64.f) + a.eps);
      const int tk = fq * 4 + r;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float y = gelu_erf((v[j] - mean) * rstd * lwv[j] + lbv[j]);
        // transposition buffer [token][64 ch], 16-B chunks XOR-swizzled by (token & 7): A-operand reads are conflict-light
        const int ch = j * 16 + fr;
        tw[tk * 64 + ((((ch >> 3) ^ (tk & 7)) << 3) | (ch & 7))] = (bf16_t)y;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // ---------------- GEMM2: [16 tokens] x [128 = sub2*32 + c2], K = 64 ----------------
    f32x4 acc2[8];
#pragma unroll
    for (int nf = 0; nf < 8; ++nf) acc2[nf] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const bf16x8 ya = *reinterpret_cast<const bf16x8*>(tw + fr * 64 + (((kk * 4 + fq) ^ (fr & 7)) << 3));
#pragma unroll
      for (int nf = 0; nf < 8; ++nf) acc2[nf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ya, w2f[nf][kk], acc2[nf], 0, 0, 0);
    }
    // ---------------- + bias, GELU, [hyper dot], stage as [c2][yy][x] ----------------
    __syncthreads();                                  // previous group's staging buffer has been drained
    const float* hy = a.hyper ? a.hyper + (int64_t)b * 32 : nullptr;
#pragma unroll
    for (int nf = 0; nf < 8; ++nf) {
      const int sub2 = nf >> 1, c2 = (nf & 1) * 16 + fr;
      const int kh2 = sub2 >> 1, kw2 = sub2 & 1;
      const int yy = 2 * kh + kh2;
      const float hv = hy ? hy[c2] : 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int tk = fq * 4 + r;
        const float y = gelu_erf(acc2[nf][r] + b2v[nf & 1]);
        const int xx = 4 * tk + 2 * kw + kw2;
        sOut[(c2 * 4 + yy) * 64 + xx] = (bf16_t)y;
        if (a.mask) {
          // the reference multiplies the bf16-rounded upscaled embedding; keep that rounding point
          float m = hv * (float)(bf16_t)y;
#pragma unroll
          for (int off = 1; off < 16; off <<= 1) m += __shfl_xor(m, off, 64);
          acc2[nf][r] = m;                             // partial over this fragment's 16 channels (same value in the 16 lanes)
        }
      }
    }
    if (a.mask && fr == 0) {
#pragma unroll
      for (int sub2 = 0; sub2 < 4; ++sub2)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int tk = fq * 4 + r;
          if (t0 + tk < n_tokens) {
            const int yy = 2 * kh + (sub2 >> 1), xx = 4 * (j0 + tk) + 2 * kw + (sub2 & 1);
            a.mask[((int64_t)b * OH + 4 * irow + yy) * OW + xx] = acc2[2 * sub2][r] + acc2[2 * sub2 + 1][r];
          }
        }
    }
    __syncthreads();
    if (a.up) {
      // 128 rows (c2, yy) of 64 pixels = 128 B each: 8 lanes per row, 16 B per lane
      const int valid_x = (int)min((int64_t)64, 4 * (n_tokens - t0));
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int id = it * UP_THREADS + tid;
        const int row = id >> 3, ch = (id & 7) * 8;
        const int c2 = row >> 2, yy = row & 3;
        if (ch < valid_x)
          *reinterpret_cast<bf16x8*>(a.up + (((int64_t)b * 32 + c2) * OH + 4 * irow + yy) * OW + 4 * j0 + ch) =
              *reinterpret_cast<const bf16x8*>(sOut + row * 64 + ch);
      }
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
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)upsample_fused_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, UP_LDS);
    attr_set = true;
  }
  UpArgs a{(const bf16_t*)src, (const bf16_t*)w1_packed, b1, ln_w, ln_b, (const bf16_t*)w2_packed, b2, hyper, (bf16_t*)up, mask,
           B, h, w, ln_eps};
  const int64_t groups = mp_cdiv((int64_t)B * h * w, 16);
  const int grid = (int)(groups < 256 ? groups : 256);
  hipLaunchKernelGGL(upsample_fused_kernel, dim3(grid), dim3(UP_THREADS), UP_LDS, stream, a);
  return mp_check_launch("mp_mask_upsample_fused_bf16");
}
