// AdamW + global-norm clipping for the trainable tail (flat fp32 buffers), replacing DeepSpeed's bf16/ZeRO-2 optimizer
// step for this path (train_ds_medplib.py:383-420: AdamW betas (0.9,0.95), weight_decay 0, gradient_clipping 1.0).
// Semantics follow DeepSpeed FusedAdam (adam_w_mode): decoupled decay, bias-corrected moments; clip as in
// DeepSpeed's unscale_and_clip: g /= max(1, (||g|| + 1e-6) / max_norm).
#include "common.h"

namespace {

// sum of squares in two stages with a FIXED reduction order (a float atomicAdd across blocks made the clip factor, hence every
// parameter, differ from run to run in the last bits): SUMSQ_BLOCKS partials into out[1 .. SUMSQ_BLOCKS], then one block adds them
// in index order onto out[0]
constexpr int SUMSQ_BLOCKS = 256;
// Round 5: four 16-byte loads in flight per thread and four independent accumulators (the first form walked its elements one dependent 4-byte
// load at a time: 219 us for the LoRA step's 33.5 M gradients = 0.6 TB/s; this form reads them at the copy rate).  The order of the additions is
// still a pure function of (n, grid): bit-reproducible from run to run.
__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ x, int64_t n, float* __restrict__ out) {
  __shared__ float red[16];
  float s = 0.f;
  const int64_t tid = (int64_t)blockIdx.x * 256 + threadIdx.x, nthr = (int64_t)gridDim.x * 256;
  int64_t done = 0;
  if ((reinterpret_cast<uintptr_t>(x) & 15) == 0) {
    const f32x4* x4 = reinterpret_cast<const f32x4*>(x);
    const int64_t n4 = n >> 2, sweep = nthr * 4;
    f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0, a2 = a0, a3 = a0;
    int64_t i = tid;
    for (; i + 3 * nthr < n4; i += sweep) {
      const f32x4 v0 = x4[i], v1 = x4[i + nthr], v2 = x4[i + 2 * nthr], v3 = x4[i + 3 * nthr];
      a0 += v0 * v0; a1 += v1 * v1; a2 += v2 * v2; a3 += v3 * v3;
    }
    for (; i < n4; i += nthr) { const f32x4 v = x4[i]; a0 += v * v; }
    const f32x4 t = (a0 + a1) + (a2 + a3);
    s = (t[0] + t[1]) + (t[2] + t[3]);
    done = n4 << 2;
  }
  for (int64_t i = done + tid; i < n; i += nthr) s += x[i] * x[i];
  s = block_sum(s, red);
  if (threadIdx.x == 0) out[1 + blockIdx.x] = s;
}
__global__ __launch_bounds__(64) void sumsq_final_kernel(float* __restrict__ out, int blocks) {
  if (threadIdx.x == 0) {
    float s = out[0];
    for (int i = 0; i < blocks; ++i) s += out[1 + i];
    out[0] = s;
  }
}

__global__ void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                             int64_t n, float lr, float b1, float b2, float eps, float wd, float bc1, float bc2_sqrt,
                             float max_norm, const float* __restrict__ sumsq, float grad_scale) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float clip = 1.f;
  if (sumsq && max_norm > 0.f) {
    const float c = (sqrtf(sumsq[0]) * grad_scale + 1e-6f) / max_norm;
    if (c > 1.f) clip = c;
  }
  const float gi = g[i] * grad_scale / clip;
  const float mi = b1 * m[i] + (1.f - b1) * gi;
  const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
  m[i] = mi; v[i] = vi;
  float pi = p[i] * (1.f - lr * wd);
  const float denom = sqrtf(vi) / bc2_sqrt + eps;
  pi -= (lr / bc1) * (mi / denom);
  p[i] = pi;
}

}  // namespace

extern "C" int mp_sumsq_accum_f32(const float* x, int64_t n, float* out_accum, hipStream_t stream) {
  if (n == 0) return MP_OK;
  const int blocks = (int)(mp_cdiv(n, 256) < SUMSQ_BLOCKS ? mp_cdiv(n, 256) : SUMSQ_BLOCKS);
  hipLaunchKernelGGL(sumsq_kernel, dim3(blocks), dim3(256), 0, stream, x, n, out_accum);
  hipLaunchKernelGGL(sumsq_final_kernel, dim3(1), dim3(64), 0, stream, out_accum, blocks);
  return mp_check_launch("mp_sumsq_accum_f32");
}

extern "C" int mp_adamw_step_f32(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1,
                                 float beta2, float eps, float weight_decay, int step, float max_norm, const float* grad_sumsq,
                                 float grad_scale, hipStream_t stream) {
  MP_REQUIRE(step >= 1, MP_ERR_ARG, "mp_adamw_step_f32: step counts from 1");
  if (n == 0) return MP_OK;
  const float bc1 = 1.f - powf(beta1, (float)step);
  const float bc2_sqrt = sqrtf(1.f - powf(beta2, (float)step));
  hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)mp_cdiv(n, 256)), dim3(256), 0, stream, param, grad, exp_avg, exp_avg_sq, n, lr, beta1,
                     beta2, eps, weight_decay, bc1, bc2_sqrt, max_norm, grad_sumsq, grad_scale);
  return mp_check_launch("mp_adamw_step_f32");
}
