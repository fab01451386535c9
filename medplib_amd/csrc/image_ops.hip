// Image preprocessing in front of the hot path, on the device: the byte arithmetic of the reference's per-sample CPU pipeline
// (datasets/LazySupervisedDataset.py:535-556 running in DataLoader workers).
//
//   ResizeLongestSide.apply_image (model/segment_anything/utils/transforms.py:25-34) = PIL Image.resize(BILINEAR) = Pillow's
//   ImagingResample 8-bits-per-channel path (src/libImaging/Resample.c): two separable passes, horizontal first, an 8-bit
//   intermediate, per-output-pixel windows [xmin, xmin + n) with coefficients computed in double, normalised, rounded to 22-bit
//   fixed point; acc = (1 << 21) + sum(pixel * coeff); out = clip8(acc >> 22).  BIT-EXACT integer work.
//
//   mp_pil_bilinear_coeffs    host: the window bounds + fixed-point coefficients of one axis (precompute_coeffs +
//                             normalize_coeffs_8bpc); pure C double arithmetic so the values are Pillow's, bit for bit
//   mp_resample_axis_u8       one pass over a [outer, len, inner] uint8 array (horizontal: outer = H, inner = C; vertical:
//                             outer = 1, inner = W*C)
//   mp_image_table_pad_chw    uint8 HWC -> float / bf16 CHW through a per-channel 256-entry value table (the host fills it in the
//                             reference's own op order: SAM `(x - pixel_mean) / pixel_std` :484, CLIP rescale + normalise) with
//                             the centre padding of pad_tensor_channelwise (:446-477) folded in
//
// All three are HBM-bound byte kernels: one thread per output element, reads coalesced along the innermost axis.
#include "common.h"
#include <math.h>

namespace {

constexpr int PRECISION_BITS = 32 - 8 - 2;

__global__ void resample_axis_u8_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, int64_t outer, int in_len,
                                        int out_len, int64_t inner, const int* __restrict__ bounds, const int* __restrict__ coefs,
                                        int ksize) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t total = outer * out_len * inner;
  if (idx >= total) return;
  const int64_t i = idx % inner;
  const int xx = (int)((idx / inner) % out_len);
  const int64_t o = idx / (inner * out_len);
  const int xmin = bounds[2 * xx], n = bounds[2 * xx + 1];
  const int* __restrict__ k = coefs + (int64_t)xx * ksize;
  const uint8_t* __restrict__ p = src + (o * in_len + xmin) * inner + i;
  int acc = 1 << (PRECISION_BITS - 1);
  for (int x = 0; x < n; ++x) acc += (int)p[(int64_t)x * inner] * k[x];
  acc >>= PRECISION_BITS;                                   // arithmetic shift, then clip8
  dst[idx] = (uint8_t)min(max(acc, 0), 255);
}

// Horizontal pass of an HWC image (inner = C <= 4): one workgroup per row, the row staged through LDS with 4-byte loads so the
// source is read from HBM exactly once, coalesced, whatever the window length (a 12-megapixel photograph resized to 336 has
// 25-tap windows).  Same arithmetic, same order of accumulation as the generic kernel.
__global__ __launch_bounds__(256) void resample_rows_lds_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, int in_len,
                                                                int out_len, int C, const int* __restrict__ bounds,
                                                                const int* __restrict__ coefs, int ksize) {
  extern __shared__ __attribute__((aligned(16))) uint8_t row[];
  const int64_t o = blockIdx.x;
  const int row_bytes = in_len * C;
  const uint8_t* __restrict__ p = src + o * row_bytes;
  if ((row_bytes & 3) == 0) {
    const uint32_t* __restrict__ p4 = reinterpret_cast<const uint32_t*>(p);
    uint32_t* r4 = reinterpret_cast<uint32_t*>(row);
    for (int i = threadIdx.x; i < (row_bytes >> 2); i += 256) r4[i] = p4[i];
  } else {
    for (int i = threadIdx.x; i < row_bytes; i += 256) row[i] = p[i];
  }
  __syncthreads();
  uint8_t* __restrict__ q = dst + o * (int64_t)out_len * C;
  for (int e = threadIdx.x; e < out_len * C; e += 256) {
    const int xx = e / C, c = e - xx * C;
    const int xmin = bounds[2 * xx], n = bounds[2 * xx + 1];
    const int* __restrict__ k = coefs + (int64_t)xx * ksize;
    const uint8_t* t = row + xmin * C + c;
    int acc = 1 << (PRECISION_BITS - 1);
    for (int x = 0; x < n; ++x) acc += (int)t[x * C] * k[x];
    acc >>= PRECISION_BITS;
    q[e] = (uint8_t)min(max(acc, 0), 255);
  }
}

template <typename TOUT>
__global__ void image_table_pad_chw_kernel(const uint8_t* __restrict__ src, int h, int w, int C, const float* __restrict__ table,
                                           const float* __restrict__ pad, TOUT* __restrict__ dst, int size_h, int size_w, int top,
                                           int left) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t total = (int64_t)C * size_h * size_w;
  if (idx >= total) return;
  const int x = (int)(idx % size_w), y = (int)((idx / size_w) % size_h), c = (int)(idx / ((int64_t)size_w * size_h));
  const int sy = y - top, sx = x - left;
  float v = pad[c];
  if (sy >= 0 && sy < h && sx >= 0 && sx < w) v = table[c * 256 + src[((int64_t)sy * w + sx) * C + c]];
  dst[idx] = (TOUT)v;
}

}  // namespace

// ICL overlay mode (ICLLazySupervisedDataset._overlay_mask, :46-50): where the example's mask is set, pixel = trunc(clip(pixel * 0.45f +
// tint * 0.55f, 0, 255)) in float32 with the two products rounded before the add (numpy evaluates `a * 0.45 + c * 0.55` as three
// separate float32 operations: no fused multiply-add here either).  One thread per pixel, 3 channels.
__global__ void overlay_mask_u8_kernel(const uint8_t* __restrict__ img, const uint8_t* __restrict__ mask, uint8_t* __restrict__ out,
                                       int64_t n_pixels, float t0, float t1, float t2) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_pixels) return;
  const bool on = mask[i] > 0;
  const float tint[3] = {t0, t1, t2};
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const uint8_t p = img[i * 3 + c];
    const float v = __fadd_rn(__fmul_rn((float)p, 0.45f), __fmul_rn(tint[c], 0.55f));
    out[i * 3 + c] = on ? (uint8_t)fminf(fmaxf(v, 0.f), 255.f) : p;
  }
}

extern "C" int mp_pil_bilinear_ksize(int in_size, int out_size) {
  if (in_size <= 0 || out_size <= 0) return 0;
  double filterscale = (double)((float)in_size - 0.0f) / out_size;
  if (filterscale < 1.0) filterscale = 1.0;
  return (int)ceil(1.0 * filterscale) * 2 + 1;
}

extern "C" int mp_pil_bilinear_coeffs(int in_size, int out_size, int* bounds, int* coefs, int ksize) {
  MP_REQUIRE(in_size > 0 && out_size > 0 && bounds && coefs, MP_ERR_ARG, "mp_pil_bilinear_coeffs: bad arguments");
  MP_REQUIRE(ksize == mp_pil_bilinear_ksize(in_size, out_size), MP_ERR_SHAPE, "mp_pil_bilinear_coeffs: ksize must be %d",
             mp_pil_bilinear_ksize(in_size, out_size));
  // Resample.c precompute_coeffs over the box (0, in_size), bilinear filter (support 1.0)
  const float in0 = 0.0f, in1 = (float)in_size;
  double scale, filterscale;
  filterscale = scale = (double)(in1 - in0) / out_size;
  if (filterscale < 1.0) filterscale = 1.0;
  const double support = 1.0 * filterscale;
  const double ss = 1.0 / filterscale;
  double* k = new double[ksize];
  for (int xx = 0; xx < out_size; ++xx) {
    const double center = in0 + (xx + 0.5) * scale;
    double ww = 0.0;
    int xmin = (int)(center - support + 0.5);
    if (xmin < 0) xmin = 0;
    int xmax = (int)(center + support + 0.5);
    if (xmax > in_size) xmax = in_size;
    xmax -= xmin;
    for (int x = 0; x < xmax; ++x) {
      double a = (x + xmin - center + 0.5) * ss;
      if (a < 0.0) a = -a;
      const double wgt = a < 1.0 ? 1.0 - a : 0.0;
      k[x] = wgt;
      ww += wgt;
    }
    for (int x = 0; x < xmax; ++x)
      if (ww != 0.0) k[x] /= ww;
    for (int x = 0; x < ksize; ++x) {
      const double v = x < xmax ? k[x] : 0.0;
      coefs[(int64_t)xx * ksize + x] = v < 0 ? (int)(-0.5 + v * (1 << PRECISION_BITS)) : (int)(0.5 + v * (1 << PRECISION_BITS));
    }
    bounds[2 * xx] = xmin;
    bounds[2 * xx + 1] = xmax;
  }
  delete[] k;
  return MP_OK;
}

extern "C" int mp_resample_axis_u8(const void* src, void* dst, int64_t outer, int in_len, int out_len, int64_t inner,
                                   const int* bounds, const int* coefs, int ksize, hipStream_t stream) {
  MP_REQUIRE(outer > 0 && in_len > 0 && out_len > 0 && inner > 0 && ksize > 0, MP_ERR_SHAPE, "mp_resample_axis_u8: bad shape");
  const int64_t total = outer * out_len * inner;
  if (outer > 1 && inner <= 4 && (int64_t)in_len * inner <= 65536) {      // row pass of an interleaved image: LDS-staged rows
    hipLaunchKernelGGL(resample_rows_lds_kernel, dim3((unsigned)outer), dim3(256), (size_t)((in_len * inner + 15) & ~15), stream,
                       (const uint8_t*)src, (uint8_t*)dst, in_len, out_len, (int)inner, bounds, coefs, ksize);
    return mp_check_launch("mp_resample_axis_u8(rows)");
  }
  hipLaunchKernelGGL(resample_axis_u8_kernel, dim3((unsigned)mp_cdiv(total, 256)), dim3(256), 0, stream, (const uint8_t*)src,
                     (uint8_t*)dst, outer, in_len, out_len, inner, bounds, coefs, ksize);
  return mp_check_launch("mp_resample_axis_u8");
}

extern "C" int mp_image_table_pad_chw(const void* src, int h, int w, int C, const float* table, const float* pad, void* dst,
                                      int size_h, int size_w, int top, int left, int out_dtype, hipStream_t stream) {
  MP_REQUIRE(h > 0 && w > 0 && C > 0 && size_h >= h && size_w >= w && top >= 0 && left >= 0 && top + h <= size_h && left + w <= size_w,
             MP_ERR_SHAPE, "mp_image_table_pad_chw: the %d x %d image does not fit the %d x %d canvas at (%d, %d)", h, w, size_h,
             size_w, top, left);
  MP_REQUIRE(out_dtype == MP_BF16 || out_dtype == MP_F32, MP_ERR_DTYPE, "mp_image_table_pad_chw: bad out dtype");
  const int64_t total = (int64_t)C * size_h * size_w;
  const dim3 grid((unsigned)mp_cdiv(total, 256)), blk(256);
  if (out_dtype == MP_F32)
    hipLaunchKernelGGL(image_table_pad_chw_kernel<float>, grid, blk, 0, stream, (const uint8_t*)src, h, w, C, table, pad, (float*)dst,
                       size_h, size_w, top, left);
  else
    hipLaunchKernelGGL(image_table_pad_chw_kernel<bf16_t>, grid, blk, 0, stream, (const uint8_t*)src, h, w, C, table, pad,
                       (bf16_t*)dst, size_h, size_w, top, left);
  return mp_check_launch("mp_image_table_pad_chw");
}

extern "C" int mp_overlay_mask_u8(const void* img, const void* mask, void* out, int64_t n_pixels, float tint_r, float tint_g,
                                  float tint_b, hipStream_t stream) {
  MP_REQUIRE(n_pixels >= 0, MP_ERR_SHAPE, "mp_overlay_mask_u8: bad size");
  if (n_pixels == 0) return MP_OK;
  hipLaunchKernelGGL(overlay_mask_u8_kernel, dim3((unsigned)mp_cdiv(n_pixels, 256)), dim3(256), 0, stream, (const uint8_t*)img,
                     (const uint8_t*)mask, (uint8_t*)out, n_pixels, tint_r, tint_g, tint_b);
  return mp_check_launch("mp_overlay_mask_u8");
}
