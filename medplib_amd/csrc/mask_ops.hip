// HBM-bound kernels of the mask head: postprocess_masks resize (bilinear, align_corners=False, with the reference's
// Python-slice "crop"), the four mask losses in one fused pass (forward + backward), and the threshold / IoU counting
// used by validation.
//
// Reference sites: model/MedPLIB.py:682-701 (postprocess_masks), :26-124 + :515-559 (MaskIoULoss, FocalLoss, dice_loss,
// sigmoid_ce_loss and their combination), train_ds_medplib.py:702-719,750-772 and model/eval/vqa_infer.py:565-588
// (sigmoid > 0.1 threshold, IoU = |and|/|or|, Dice = 2 IoU / (1 + IoU)).
#include "common.h"

namespace {

// ---------------- bilinear resize (PyTorch upsample_bilinear2d, align_corners=False, no antialias) ----------------
struct Lerp { int i0, i1; float w0, w1; };
__device__ __forceinline__ Lerp src_index(int dst, float scale, int in_size) {
  float s = scale * ((float)dst + 0.5f) - 0.5f;
  if (s < 0.f) s = 0.f;
  int i0 = (int)s;
  if (i0 > in_size - 1) i0 = in_size - 1;
  const int i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
  float l1 = s - (float)i0;
  l1 = fminf(fmaxf(l1, 0.f), 1.f);
  return Lerp{i0, i1, 1.f - l1, l1};
}

// in: [n, IH, IW] (full low-res map), sampled from the crop window (y0, x0, ch, cw); out: [n, OH, OW]
template <typename TIN>
__global__ void bilinear_fwd_kernel(const TIN* __restrict__ in, float* __restrict__ out, int n, int IH, int IW, int y0, int x0,
                                    int ch, int cw, int OH, int OW) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)n * OH * OW) return;
  const int ox = (int)(idx % OW);
  const int oy = (int)((idx / OW) % OH);
  const int m = (int)(idx / ((int64_t)OW * OH));
  const float sh = (float)ch / (float)OH, sw = (float)cw / (float)OW;
  const Lerp ly = src_index(oy, sh, ch), lx = src_index(ox, sw, cw);
  const TIN* p = in + (int64_t)m * IH * IW;
  const float v00 = ld_f(p, (int64_t)(y0 + ly.i0) * IW + x0 + lx.i0), v01 = ld_f(p, (int64_t)(y0 + ly.i0) * IW + x0 + lx.i1);
  const float v10 = ld_f(p, (int64_t)(y0 + ly.i1) * IW + x0 + lx.i0), v11 = ld_f(p, (int64_t)(y0 + ly.i1) * IW + x0 + lx.i1);
  out[idx] = ly.w0 * (lx.w0 * v00 + lx.w1 * v01) + ly.w1 * (lx.w0 * v10 + lx.w1 * v11);
}

// d_in = transpose of the resize as a GATHER: one thread per input pixel of the crop window walks the (few) output rows / columns
// whose two taps can touch it, tests them with the same src_index() the forward uses, and adds in a fixed (oy, ox) order — no float
// atomics, so the gradient is bit-reproducible (PyTorch's upsample_bilinear2d_backward scatters with atomicAdd).
__device__ __forceinline__ void tap_range(int i, float scale, int out_size, int* lo, int* hi) {
  // outputs o with floor(src(o)) in {i-1, i}: src(o) = scale*(o+0.5)-0.5 in [i-1, i+1)  ->  o in [(i-0.5)/scale-0.5, (i+1.5)/scale-0.5)
  const float inv = 1.f / scale;
  int a = (int)floorf(((float)i - 0.5f) * inv - 0.5f) - 1;
  int b = (int)ceilf(((float)i + 1.5f) * inv - 0.5f) + 1;
  *lo = a < 0 ? 0 : a;
  *hi = b > out_size - 1 ? out_size - 1 : b;
}
__global__ void bilinear_bwd_kernel(const float* __restrict__ dout, float* __restrict__ din, int n, int IH, int IW, int y0,
                                    int x0, int ch, int cw, int OH, int OW) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)n * ch * cw) return;
  const int ix = (int)(idx % cw);
  const int iy = (int)((idx / cw) % ch);
  const int m = (int)(idx / ((int64_t)cw * ch));
  const float sh = (float)ch / (float)OH, sw = (float)cw / (float)OW;
  int oy0, oy1, ox0, ox1;
  tap_range(iy, sh, OH, &oy0, &oy1);
  tap_range(ix, sw, OW, &ox0, &ox1);
  // the first / last input row also collect the clamped outputs (src < 0 -> 0, i0 clamped to in_size - 1)
  if (iy == 0) oy0 = 0;
  if (iy == ch - 1) oy1 = OH - 1;
  if (ix == 0) ox0 = 0;
  if (ix == cw - 1) ox1 = OW - 1;
  const float* g = dout + (int64_t)m * OH * OW;
  float acc = 0.f;
  for (int oy = oy0; oy <= oy1; ++oy) {
    const Lerp ly = src_index(oy, sh, ch);
    float wy = 0.f;
    if (ly.i0 == iy) wy += ly.w0;
    if (ly.i1 == iy) wy += ly.w1;                 // i1 == i0 at the last row: both taps land on it, like the scatter
    if (ly.i0 != iy && ly.i1 != iy) continue;
    float row = 0.f;
    for (int ox = ox0; ox <= ox1; ++ox) {
      const Lerp lx = src_index(ox, sw, cw);
      float wx = 0.f;
      if (lx.i0 == ix) wx += lx.w0;
      if (lx.i1 == ix) wx += lx.w1;
      if (lx.i0 == ix || lx.i1 == ix) row += g[(int64_t)oy * OW + ox] * wx;
    }
    acc += wy * row;
  }
  din[(int64_t)m * IH * IW + (int64_t)(y0 + iy) * IW + x0 + ix] += acc;      // each input pixel has exactly one owner thread
}

// ---------------- fused mask losses ----------------
constexpr int LOSS_BLOCKS = 64;   // partial-sum blocks per mask (fixed -> deterministic reduction order)
constexpr int NSUM = 6;           // bce, p, g, pg, focal_pos, focal_neg
constexpr float FOCAL_ALPHA = 0.25f;

__device__ __forceinline__ void loss_terms(float x, float g, float* t) {
  const float p = 1.f / (1.f + expf(-x));
  t[0] = fmaxf(x, 0.f) - x * g + log1pf(expf(-fabsf(x)));                 // BCE-with-logits
  t[1] = p;
  t[2] = g;
  t[3] = p * g;
  const float omp = 1.f - p;
  t[4] = -FOCAL_ALPHA * g * omp * omp * logf(p + 1e-12f);                    // gamma = 2
  t[5] = -(1.f - FOCAL_ALPHA) * (1.f - g) * p * p * logf(omp + 1e-12f);
}

// offsets (optional, [n+1] element offsets into the flat pred / gt buffers) make the batch ragged: masks of different H x W
// in one launch (the reference loops over masks, MedPLIB.py:515-559); null = n masks of HW elements each
__global__ __launch_bounds__(256) void mask_loss_partial_kernel(const float* __restrict__ pred, const float* __restrict__ gt,
                                                                float* __restrict__ partial, int64_t HW,
                                                                const int64_t* __restrict__ offsets) {
  __shared__ float red[16];
  const int m = blockIdx.y;
  const int64_t base = offsets ? offsets[m] : (int64_t)m * HW;
  if (offsets) HW = offsets[m + 1] - base;
  const float* x = pred + base;
  const float* g = gt + base;
  float acc[NSUM] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < HW; i += (int64_t)LOSS_BLOCKS * 256) {
    float t[NSUM];
    loss_terms(x[i], g[i], t);
#pragma unroll
    for (int k = 0; k < NSUM; ++k) acc[k] += t[k];
  }
#pragma unroll
  for (int k = 0; k < NSUM; ++k) {
    const float s = block_sum(acc[k], red);
    if (threadIdx.x == 0) partial[((int64_t)m * LOSS_BLOCKS + blockIdx.x) * NSUM + k] = s;
  }
}

// One block; thread m finishes mask m.  stats[m] = {S_p, S_g, S_pg, J (iou), q (pred_iou), -, -, -}.
// out[10] follows the reference's dict order (MedPLIB.py:561-572):
//   loss, ce_loss, mask_bce_loss, mask_dice_loss, mask_loss, unscale_bce, unscale_dice, unscale_mask_loss, unscale_iou, unscale_focal
__global__ __launch_bounds__(256) void mask_loss_finalize_kernel(const float* __restrict__ partial, const float* __restrict__ pred_iou,
                                                                 const float* __restrict__ ce_loss, float* __restrict__ stats,
                                                                 float* __restrict__ out, int n, int64_t HW, float w_ce, float w_bce,
                                                                 float w_dice, float w_iou, float w_focal,
                                                                 const int64_t* __restrict__ offsets) {
  __shared__ float red[16];
  float l_bce = 0.f, l_dice = 0.f, l_iou = 0.f, l_focal = 0.f;
  for (int m = threadIdx.x; m < n; m += 256) {
    float s[NSUM] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int b = 0; b < LOSS_BLOCKS; ++b)
#pragma unroll
      for (int k = 0; k < NSUM; ++k) s[k] += partial[((int64_t)m * LOSS_BLOCKS + b) * NSUM + k];
    const float N = offsets ? (float)(offsets[m + 1] - offsets[m]) : (float)HW;
    const float bce = (s[0] / N) / (1.f + 1e-8f);
    const float dice = 1.f - (2.f * s[3] + 1e-6f) / (s[1] + s[2] + 1e-6f);
    const float J = (s[3] + 1e-7f) / (s[1] + s[2] - s[3] + 1e-7f);
    const float q = pred_iou[m];
    const float iou_l = (J - q) * (J - q);
    const float focal = (s[4] + s[5]) / (N + 1e-12f);
    l_bce += bce; l_dice += dice; l_iou += iou_l; l_focal += focal;
    float* st = stats + (int64_t)m * 8;
    st[0] = s[1]; st[1] = s[2]; st[2] = s[3]; st[3] = J; st[4] = q;
  }
  l_bce = block_sum(l_bce, red);
  l_dice = block_sum(l_dice, red);
  l_iou = block_sum(l_iou, red);
  l_focal = block_sum(l_focal, red);
  if (threadIdx.x == 0) {
    const float cn = 1.f / ((float)n + 1e-8f);
    const float u_bce = l_bce * cn, u_dice = l_dice * cn, u_iou = l_iou * cn, u_focal = l_focal * cn;
    const float ce = (ce_loss ? ce_loss[0] : 0.f) * w_ce;
    const float mask_bce = w_bce * u_bce, mask_dice = w_dice * u_dice, mask_iou = w_iou * u_iou, mask_focal = w_focal * u_focal;
    const float mask_loss = mask_bce + mask_dice + mask_iou + mask_focal;
    out[0] = ce + mask_loss; out[1] = ce; out[2] = mask_bce; out[3] = mask_dice; out[4] = mask_loss;
    out[5] = u_bce; out[6] = u_dice; out[7] = u_bce + u_dice + u_iou + u_focal; out[8] = u_iou; out[9] = u_focal;
  }
}

// d(loss)/d(pred) for upstream gradient *gscale (device scalar) on out[0]; also d(loss)/d(pred_iou)
__global__ void mask_loss_bwd_kernel(const float* __restrict__ pred, const float* __restrict__ gt, const float* __restrict__ stats,
                                     const float* __restrict__ gscale, float* __restrict__ dpred, float* __restrict__ dpred_iou,
                                     int n, int64_t HW, float w_bce, float w_dice, float w_iou, float w_focal,
                                     const int64_t* __restrict__ offsets) {
  const int m = blockIdx.y;
  const int64_t base = offsets ? offsets[m] : (int64_t)m * HW;
  if (offsets) HW = offsets[m + 1] - base;
  const float gs = gscale ? gscale[0] : 1.f;
  const float cn = gs / ((float)n + 1e-8f);
  const float* st = stats + (int64_t)m * 8;
  const float Sp = st[0], Sg = st[1], I = st[2], J = st[3], q = st[4];
  const float N = (float)HW;
  const float U = Sp + Sg + 1e-6f;
  const float V = Sp + Sg - I + 1e-7f;
  if (blockIdx.x == 0 && threadIdx.x == 0 && dpred_iou) dpred_iou[m] = cn * w_iou * (-2.f * (J - q));
  const float c_bce = cn * w_bce / (N * (1.f + 1e-8f));
  const float c_dice = cn * w_dice;
  const float c_iou = cn * w_iou * 2.f * (J - q);
  const float c_focal = cn * w_focal / (N + 1e-12f);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < HW; i += (int64_t)gridDim.x * blockDim.x) {
    const float x = pred[base + i], g = gt[base + i];
    const float p = 1.f / (1.f + expf(-x));
    const float omp = 1.f - p;
    const float dp_dx = p * omp;
    float d = c_bce * (p - g);
    // dice: D = (2I+e)/(U), dD/dp = (2g U - (2I+e)) / U^2 ; loss = 1 - D
    d += c_dice * (-(2.f * g * U - (2.f * I + 1e-6f)) / (U * U)) * dp_dx;
    // iou: J = (I+e7)/V ; dJ/dp = (g V - (I+e7)(1-g)) / V^2
    d += c_iou * ((g * V - (I + 1e-7f) * (1.f - g)) / (V * V)) * dp_dx;
    // focal
    const float dpos = -FOCAL_ALPHA * g * (-2.f * omp * logf(p + 1e-12f) + omp * omp / (p + 1e-12f));
    const float dneg = -(1.f - FOCAL_ALPHA) * (1.f - g) * (2.f * p * logf(omp + 1e-12f) - p * p / (omp + 1e-12f));
    d += c_focal * (dpos + dneg) * dp_dx;
    dpred[base + i] = d;
  }
}

// ---------------- threshold + IoU counts (integer, exact) ----------------
// bin = sigmoid(x) > thr ; counts[m] = {sum(bin), sum(gt!=0), sum(bin & gt), sum(bin | gt)}
template <typename TIN>
__global__ __launch_bounds__(256) void threshold_iou_kernel(const TIN* __restrict__ pred, const float* __restrict__ gt,
                                                            uint8_t* __restrict__ bin_out, unsigned long long* __restrict__ counts,
                                                            int64_t HW, float thr) {
  __shared__ unsigned int sc[4];
  if (threadIdx.x < 4) sc[threadIdx.x] = 0;
  __syncthreads();
  const int m = blockIdx.y;
  unsigned int c0 = 0, c1 = 0, c2 = 0, c3 = 0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < HW; i += (int64_t)gridDim.x * 256) {
    const float x = ld_f(pred, (int64_t)m * HW + i);
    const float s = 1.f / (1.f + expf(-x));
    const bool b = s > thr;
    const bool g = gt ? (gt[(int64_t)m * HW + i] != 0.f) : false;
    if (bin_out) bin_out[(int64_t)m * HW + i] = b ? 1 : 0;
    c0 += b; c1 += g; c2 += (b && g); c3 += (b || g);
  }
  atomicAdd(&sc[0], c0); atomicAdd(&sc[1], c1); atomicAdd(&sc[2], c2); atomicAdd(&sc[3], c3);
  __syncthreads();
  if (threadIdx.x < 4) atomicAdd(counts + (int64_t)m * 4 + threadIdx.x, (unsigned long long)sc[threadIdx.x]);
}

}  // namespace

extern "C" int mp_bilinear_resize_fwd(const void* in, int in_dtype, float* out, int n, int in_h, int in_w, int crop_y0,
                                      int crop_x0, int crop_h, int crop_w, int out_h, int out_w, hipStream_t stream) {
  MP_REQUIRE(n >= 0 && crop_h > 0 && crop_w > 0 && out_h > 0 && out_w > 0, MP_ERR_SHAPE, "mp_bilinear_resize_fwd: bad shape");
  MP_REQUIRE(crop_y0 >= 0 && crop_x0 >= 0 && crop_y0 + crop_h <= in_h && crop_x0 + crop_w <= in_w, MP_ERR_SHAPE,
             "mp_bilinear_resize_fwd: crop window outside the input");
  const int64_t total = (int64_t)n * out_h * out_w;
  if (total == 0) return MP_OK;
  dim3 grid((unsigned)mp_cdiv(total, 256));
  if (in_dtype == MP_F32)
    hipLaunchKernelGGL(bilinear_fwd_kernel<float>, grid, dim3(256), 0, stream, (const float*)in, out, n, in_h, in_w, crop_y0,
                       crop_x0, crop_h, crop_w, out_h, out_w);
  else if (in_dtype == MP_BF16)
    hipLaunchKernelGGL(bilinear_fwd_kernel<bf16_t>, grid, dim3(256), 0, stream, (const bf16_t*)in, out, n, in_h, in_w, crop_y0,
                       crop_x0, crop_h, crop_w, out_h, out_w);
  else MP_REQUIRE(false, MP_ERR_DTYPE, "mp_bilinear_resize_fwd: bad dtype %d", in_dtype);
  return mp_check_launch("mp_bilinear_resize_fwd");
}

extern "C" int mp_bilinear_resize_bwd(const float* dout, float* din_zeroed, int n, int in_h, int in_w, int crop_y0, int crop_x0,
                                      int crop_h, int crop_w, int out_h, int out_w, hipStream_t stream) {
  MP_REQUIRE(crop_y0 >= 0 && crop_x0 >= 0 && crop_y0 + crop_h <= in_h && crop_x0 + crop_w <= in_w && crop_h > 0 && crop_w > 0,
             MP_ERR_SHAPE, "mp_bilinear_resize_bwd: crop window outside the input");
  const int64_t total = (int64_t)n * crop_h * crop_w;            // one thread per input pixel of the crop window (gather)
  if (total == 0 || out_h == 0 || out_w == 0) return MP_OK;
  hipLaunchKernelGGL(bilinear_bwd_kernel, dim3((unsigned)mp_cdiv(total, 256)), dim3(256), 0, stream, dout, din_zeroed, n, in_h,
                     in_w, crop_y0, crop_x0, crop_h, crop_w, out_h, out_w);
  return mp_check_launch("mp_bilinear_resize_bwd");
}

extern "C" size_t mp_mask_losses_workspace(int n_masks) { return (size_t)n_masks * LOSS_BLOCKS * NSUM * sizeof(float); }

extern "C" int mp_mask_losses_fwd(const float* pred, const float* gt, const float* pred_iou, const float* ce_loss, int n_masks,
                                  int64_t hw, const int64_t* offsets, float w_ce, float w_bce, float w_dice, float w_iou, float w_focal,
                                  float* stats, float* out10, void* workspace, size_t workspace_bytes, hipStream_t stream) {
  MP_REQUIRE(n_masks > 0 && (hw > 0 || offsets), MP_ERR_SHAPE, "mp_mask_losses_fwd: need at least one mask (n=%d)", n_masks);
  MP_REQUIRE(workspace_bytes >= mp_mask_losses_workspace(n_masks), MP_ERR_WORKSPACE, "mp_mask_losses_fwd: workspace too small");
  float* partial = (float*)workspace;
  hipLaunchKernelGGL(mask_loss_partial_kernel, dim3(LOSS_BLOCKS, n_masks), dim3(256), 0, stream, pred, gt, partial, hw, offsets);
  hipLaunchKernelGGL(mask_loss_finalize_kernel, dim3(1), dim3(256), 0, stream, partial, pred_iou, ce_loss, stats, out10, n_masks,
                     hw, w_ce, w_bce, w_dice, w_iou, w_focal, offsets);
  return mp_check_launch("mp_mask_losses_fwd");
}

extern "C" int mp_mask_losses_bwd(const float* pred, const float* gt, const float* stats, const float* grad_scale, float* dpred,
                                  float* dpred_iou, int n_masks, int64_t hw, const int64_t* offsets, float w_bce, float w_dice,
                                  float w_iou, float w_focal, hipStream_t stream) {
  MP_REQUIRE(n_masks > 0 && (hw > 0 || offsets), MP_ERR_SHAPE, "mp_mask_losses_bwd: bad shape");
  const int bx = offsets ? 128 : (int)(mp_cdiv(hw, 256) < 128 ? mp_cdiv(hw, 256) : 128);
  hipLaunchKernelGGL(mask_loss_bwd_kernel, dim3(bx, n_masks), dim3(256), 0, stream, pred, gt, stats, grad_scale, dpred,
                     dpred_iou, n_masks, hw, w_bce, w_dice, w_iou, w_focal, offsets);
  return mp_check_launch("mp_mask_losses_bwd");
}

extern "C" int mp_mask_threshold_iou(const void* pred, int pred_dtype, const float* gt, uint8_t* bin_out,
                                     unsigned long long* counts_zeroed, int n_masks, int64_t hw, float threshold,
                                     hipStream_t stream) {
  MP_REQUIRE(n_masks > 0 && hw > 0, MP_ERR_SHAPE, "mp_mask_threshold_iou: bad shape");
  const int bx = (int)(mp_cdiv(hw, 256) < 64 ? mp_cdiv(hw, 256) : 64);
  if (pred_dtype == MP_F32)
    hipLaunchKernelGGL(threshold_iou_kernel<float>, dim3(bx, n_masks), dim3(256), 0, stream, (const float*)pred, gt, bin_out,
                       counts_zeroed, hw, threshold);
  else if (pred_dtype == MP_BF16)
    hipLaunchKernelGGL(threshold_iou_kernel<bf16_t>, dim3(bx, n_masks), dim3(256), 0, stream, (const bf16_t*)pred, gt, bin_out,
                       counts_zeroed, hw, threshold);
  else MP_REQUIRE(false, MP_ERR_DTYPE, "mp_mask_threshold_iou: bad dtype %d", pred_dtype);
  return mp_check_launch("mp_mask_threshold_iou");
}
