// Shared device/host helpers for the medplib_amd HIP library (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

#define MP_OK 0
#define MP_ERR_SHAPE (-1)
#define MP_ERR_DTYPE (-2)
#define MP_ERR_WORKSPACE (-3)
#define MP_ERR_LAUNCH (-4)
#define MP_ERR_ARG (-5)

#define MP_BF16 0
#define MP_F32 1

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

// error plumbing (host side, defined in capi.cpp)
extern "C" const char* mp_last_error_string();
void mp_set_error(const char* fmt, ...);
int mp_check_launch(const char* what);

#define MP_REQUIRE(cond, code, ...)        \
  do {                                     \
    if (!(cond)) {                         \
      mp_set_error(__VA_ARGS__);           \
      return (code);                       \
    }                                      \
  } while (0)

static inline int64_t mp_cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

#ifdef __HIPCC__
__device__ __forceinline__ float bf2f(bf16_t v) { return (float)v; }
__device__ __forceinline__ bf16_t f2bf(float v) { return (bf16_t)v; }  // RNE

// wave64 all-reduce helpers
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
// block all-reduce (sum) for blockDim.x <= 1024, multiple of 64. `red` holds >= 16 floats of LDS.
__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if (lane == 0) red[w] = v;
  __syncthreads();
  float t = 0.f;
  for (int i = 0; i < nw; ++i) t += red[i];
  return t;
}
__device__ __forceinline__ float block_max(float v, float* red) {
  v = wave_max(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if (lane == 0) red[w] = v;
  __syncthreads();
  float t = red[0];
  for (int i = 1; i < nw; ++i) t = fmaxf(t, red[i]);
  return t;
}

template <typename T> __device__ __forceinline__ float ld_f(const T* p, int64_t i);
template <> __device__ __forceinline__ float ld_f<float>(const float* p, int64_t i) { return p[i]; }
template <> __device__ __forceinline__ float ld_f<bf16_t>(const bf16_t* p, int64_t i) { return (float)p[i]; }
template <typename T> __device__ __forceinline__ void st_f(T* p, int64_t i, float v);
template <> __device__ __forceinline__ void st_f<float>(float* p, int64_t i, float v) { p[i] = v; }
template <> __device__ __forceinline__ void st_f<bf16_t>(bf16_t* p, int64_t i, float v) { p[i] = (bf16_t)v; }

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float gelu_erf_grad(float x) {
  const float cdf = 0.5f * (1.f + erff(x * 0.70710678118654752440f));
  const float pdf = 0.39894228040143267794f * __expf(-0.5f * x * x);
  return cdf + x * pdf;
}
__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + __expf(-x)); }
// sigmoid(s x) of the bf16 trunk's activations (SiLU / SwiGLU: s = 1, QuickGELU: s = 1.702): v_rcp_f32 (1 ulp) instead of the correctly
// rounded fp32 division (ten instructions per element in the GEMM epilogues); every kernel that evaluates these activations uses this
// one expression, so the fused and unfused paths stay bit-identical with each other.  The fp32 mask tail keeps sigmoidf_.
__device__ __forceinline__ float mp_sigmoid_fast(float x, float s = 1.f) { return __builtin_amdgcn_rcpf(1.f + __expf(-s * x)); }
#endif
