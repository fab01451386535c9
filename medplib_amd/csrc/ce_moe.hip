// Cross-entropy over supervised rows and the DeepSpeed-style MoE routing kernels (gate, top-1 routing with capacity /
// random-token-selection, dispatch, combine).  Index outputs (expert id, slot) are integer-exact.
//
// Reference sites: medplib_moe_llama.py:388-421 (fp32 logits, shift, all-ignored-row filter, mean CE, + aux loss);
// DeepSpeed 0.13.1 `deepspeed.moe.sharded_moe.{TopKGate, top1gating, MOELayer}` called from medplib_moe_llama.py:604-614
// (third-party, restated in SURVEY.md Appendix A.3 — parity unpinned by any reference test).
// The reference dispatches/combines with dense one-hot einsums ("sec,sm->ecm"); here both are index gathers.
#include "common.h"

namespace {

constexpr int MAXE = 8;

// ---------------- CE ----------------
__global__ __launch_bounds__(256) void ce_rows_kernel(const float* __restrict__ logits, int64_t ld, const int64_t* __restrict__ labels,
                                                      int V, float* __restrict__ row_loss) {
  __shared__ float red[16];
  const int64_t r = blockIdx.x;
  const float* x = logits + r * ld;
  float m = -INFINITY;
  for (int i = threadIdx.x; i < V; i += 256) m = fmaxf(m, x[i]);
  m = block_max(m, red);
  float s = 0.f;
  for (int i = threadIdx.x; i < V; i += 256) s += expf(x[i] - m);
  s = block_sum(s, red);
  if (threadIdx.x == 0) row_loss[r] = logf(s) + m - x[labels[r]];
}

// out[0] = (sum(x[0:n]) / n) * scale + (add ? add_scale * sum(add[0:n_add]) : 0)     (n == 0 -> NaN like torch's mean of empty)
__global__ __launch_bounds__(256) void mean_plus_kernel(const float* __restrict__ x, int64_t n, float scale, const float* __restrict__ add,
                                                        int n_add, float add_scale, float* __restrict__ out) {
  __shared__ float red[16];
  float s = 0.f;
  for (int64_t i = threadIdx.x; i < n; i += 256) s += x[i];
  s = block_sum(s, red);
  float a = 0.f;
  for (int i = threadIdx.x; i < n_add; i += 256) a += add[i];
  a = block_sum(a, red);
  if (threadIdx.x == 0) out[0] = (s / (float)n) * scale + (add ? add_scale * a : 0.f);
}

// ---------------- MoE gate: logits = x.float() @ wg.float()^T ; gates = softmax(logits) ----------------
__global__ __launch_bounds__(256) void moe_gate_kernel(const bf16_t* __restrict__ x, int64_t ldx, const float* __restrict__ wg,
                                                       float* __restrict__ logits, float* __restrict__ gates, int64_t T, int d, int E) {
  const int lane = threadIdx.x & 63;
  const int64_t tok = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (tok >= T) return;
  float acc[MAXE];
#pragma unroll
  for (int e = 0; e < MAXE; ++e) acc[e] = 0.f;
  const bf16_t* xr = x + tok * ldx;
  for (int i = lane * 8; i < d; i += 64 * 8) {
    const bf16x8 v = *reinterpret_cast<const bf16x8*>(xr + i);
#pragma unroll
    for (int e = 0; e < MAXE; ++e) {
      if (e < E) {
        const float* w = wg + (int64_t)e * d + i;
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[e] = fmaf((float)v[j], w[j], acc[e]);
      }
    }
  }
  float mx = -INFINITY;
#pragma unroll
  for (int e = 0; e < MAXE; ++e)
    if (e < E) { acc[e] = wave_sum(acc[e]); mx = fmaxf(mx, acc[e]); }
  if (lane == 0) {
    float s = 0.f, p[MAXE];
#pragma unroll
    for (int e = 0; e < MAXE; ++e)
      if (e < E) { p[e] = expf(acc[e] - mx); s += p[e]; }
#pragma unroll
    for (int e = 0; e < MAXE; ++e)
      if (e < E) { logits[tok * E + e] = acc[e]; gates[tok * E + e] = p[e] / s; }
  }
}

// ---------------- top-1 routing (single block, 1024 threads) ----------------
// expert[s] = argmax gates[s]; capacity drop: if an expert is over capacity keep the `capacity` tokens with the largest
// uniform draws (DeepSpeed RTS; first-come when no draws are supplied); slot[s] = rank among KEPT tokens of that expert
// in token order (cumsum(mask1) - 1), -1 if dropped; weight[s] = gates[s, expert] (not renormalised).
// l_aux = E * sum_e mean_s(gates[:,e]) * mean_s(mask1[:,e])   (computed BEFORE dropping).
__global__ __launch_bounds__(1024) void moe_route_top1_kernel(const float* __restrict__ gates, const float* __restrict__ rts, int T,
                                                              int E, int capacity, int* __restrict__ expert, int* __restrict__ slot,
                                                              float* __restrict__ weight, int* __restrict__ kept_counts,
                                                              long long* __restrict__ exp_counts, float* __restrict__ l_aux) {
  __shared__ float red[16];
  __shared__ int cnt_sh[MAXE];
  __shared__ int scan[16][MAXE];     // per-wave kept totals
  const int tid = threadIdx.x;
  if (tid < MAXE) cnt_sh[tid] = 0;
  __syncthreads();
  float me[MAXE];
  int cnt[MAXE];
#pragma unroll
  for (int e = 0; e < MAXE; ++e) { me[e] = 0.f; cnt[e] = 0; }
  for (int s = tid; s < T; s += 1024) {
    int best = 0;
    float bv = gates[(int64_t)s * E];
    for (int e = 0; e < E; ++e) {
      const float g = gates[(int64_t)s * E + e];
      if (e > 0 && g > bv) { bv = g; best = e; }
#pragma unroll
      for (int k = 0; k < MAXE; ++k) if (k == e) me[k] += g;
    }
    expert[s] = best;
    weight[s] = bv;
#pragma unroll
    for (int k = 0; k < MAXE; ++k) if (k == best) cnt[k] += 1;
  }
  float aux = 0.f;
  for (int e = 0; e < E; ++e) {
    float m = 0.f; int c = 0;
#pragma unroll
    for (int k = 0; k < MAXE; ++k) if (k == e) { m = me[k]; c = cnt[k]; }
    const float msum = block_sum(m, red);
    if (c) atomicAdd(&cnt_sh[e], c);
    __syncthreads();
    aux += (msum / (float)T) * ((float)cnt_sh[e] / (float)T);
  }
  if (tid == 0) {
    l_aux[0] = aux * (float)E;
    for (int e = 0; e < E; ++e) exp_counts[e] = cnt_sh[e];
  }
  __syncthreads();
  // keep decision (global memory `slot` temporarily holds the keep flag)
  for (int s = tid; s < T; s += 1024) {
    const int e = expert[s];
    int keep = 1;
    if (cnt_sh[e] > capacity) {
      if (rts) {
        int rank = 0;
        const float u = rts[(int64_t)s * E + e];
        for (int t = 0; t < T; ++t) {
          if (expert[t] != e) continue;
          const float ut = rts[(int64_t)t * E + e];
          rank += (ut > u) || (ut == u && t < s);
        }
        keep = rank < capacity;
      }
      // without RTS draws the selection is first-come: rank == the token-order prefix count computed by the scan below, so
      // every token stays flagged here and the slots >= capacity are dropped after the scan (no O(T^2) pass)
    }
    slot[s] = keep;
  }
  __syncthreads();
  // exclusive scan of kept tokens per expert in token order: thread `tid` owns a contiguous chunk; wave-level shuffle scan of
  // the per-thread totals, then the 16 wave totals are combined through LDS
  const int chunk = (T + 1023) / 1024;
  const int s0 = tid * chunk, s1 = min(T, s0 + chunk);
  const int lane = tid & 63, wv = tid >> 6;
  int loc[MAXE];
#pragma unroll
  for (int e = 0; e < MAXE; ++e) loc[e] = 0;
  for (int s = s0; s < s1; ++s)
    if (slot[s]) {
      const int e = expert[s];
#pragma unroll
      for (int k = 0; k < MAXE; ++k) if (k == e) loc[k] += 1;
    }
  int base[MAXE];
#pragma unroll
  for (int e = 0; e < MAXE; ++e) {
    base[e] = 0;
    if (e < E) {
      int v = loc[e];
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const int n = __shfl_up(v, off, 64);
        if (lane >= off) v += n;
      }
      if (lane == 63) scan[wv][e] = v;        // wave total
      base[e] = v - loc[e];                   // exclusive prefix within the wave
    }
  }
  __syncthreads();
#pragma unroll
  for (int e = 0; e < MAXE; ++e)
    if (e < E)
      for (int w = 0; w < wv; ++w) base[e] += scan[w][e];
  for (int s = s0; s < s1; ++s) {
    if (slot[s]) {
      const int e = expert[s];
      int v = 0;
#pragma unroll
      for (int k = 0; k < MAXE; ++k) if (k == e) { v = base[k]; base[k] += 1; }
      slot[s] = (v < capacity) ? v : -1;
    } else {
      slot[s] = -1;
    }
  }
  if (tid == 1023)
    for (int e = 0; e < E; ++e) kept_counts[e] = min(base[e], capacity);
}

// buf[expert[s], slot[s], :] = x[s, :]
__global__ void moe_dispatch_kernel(const bf16_t* __restrict__ x, int64_t ldx, const int* __restrict__ expert, const int* __restrict__ slot,
                                    bf16_t* __restrict__ buf, int64_t T, int d, int capacity) {
  const int per_row = d / 8;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= T * per_row) return;
  const int64_t s = idx / per_row;
  const int c = (int)(idx % per_row) * 8;
  const int sl = slot[s];
  if (sl < 0) return;
  *reinterpret_cast<bf16x8*>(buf + ((int64_t)expert[s] * capacity + sl) * d + c) = *reinterpret_cast<const bf16x8*>(x + s * ldx + c);
}

// out[s, :] = residual[s, :] + weight[s] * y[expert[s], slot[s], :]     (dropped tokens: residual only)
__global__ void moe_combine_kernel(const bf16_t* __restrict__ y, const int* __restrict__ expert, const int* __restrict__ slot,
                                   const float* __restrict__ weight, const bf16_t* __restrict__ residual, bf16_t* __restrict__ out,
                                   int64_t T, int d, int capacity) {
  const int per_row = d / 8;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= T * per_row) return;
  const int64_t s = idx / per_row;
  const int c = (int)(idx % per_row) * 8;
  const int sl = slot[s];
  bf16x8 r;
  if (residual) r = *reinterpret_cast<const bf16x8*>(residual + s * d + c);
  bf16x8 o;
  if (sl >= 0) {
    const bf16x8 v = *reinterpret_cast<const bf16x8*>(y + ((int64_t)expert[s] * capacity + sl) * d + c);
    const float w = weight[s];
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (bf16_t)((residual ? (float)r[j] : 0.f) + w * (float)v[j]);
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = residual ? r[j] : (bf16_t)0.f;
  }
  *reinterpret_cast<bf16x8*>(out + s * d + c) = o;
}

}  // namespace

extern "C" int mp_cross_entropy_rows_f32(const float* logits, int64_t ld, const int64_t* labels, int64_t n_rows, int vocab,
                                         float* row_loss, hipStream_t stream) {
  MP_REQUIRE(vocab > 0, MP_ERR_SHAPE, "mp_cross_entropy_rows_f32: bad vocab");
  if (n_rows == 0) return MP_OK;
  hipLaunchKernelGGL(ce_rows_kernel, dim3((unsigned)n_rows), dim3(256), 0, stream, logits, ld, labels, vocab, row_loss);
  return mp_check_launch("mp_cross_entropy_rows_f32");
}

extern "C" int mp_mean_plus_f32(const float* x, int64_t n, float scale, const float* add, int n_add, float add_scale, float* out,
                                hipStream_t stream) {
  hipLaunchKernelGGL(mean_plus_kernel, dim3(1), dim3(256), 0, stream, x, n, scale, add, n_add, add_scale, out);
  return mp_check_launch("mp_mean_plus_f32");
}

extern "C" int mp_moe_gate_bf16(const void* x, int64_t ldx, const float* wg, float* logits, float* gates, int64_t tokens, int dim,
                                int n_experts, hipStream_t stream) {
  MP_REQUIRE(n_experts >= 1 && n_experts <= MAXE && dim % 8 == 0 && ldx % 8 == 0, MP_ERR_SHAPE, "mp_moe_gate_bf16: bad shape (E<=8)");
  if (tokens == 0) return MP_OK;
  hipLaunchKernelGGL(moe_gate_kernel, dim3((unsigned)mp_cdiv(tokens, 4)), dim3(256), 0, stream, (const bf16_t*)x, ldx, wg, logits,
                     gates, tokens, dim, n_experts);
  return mp_check_launch("mp_moe_gate_bf16");
}

extern "C" int mp_moe_route_top1(const float* gates, const float* rts_uniform, int tokens, int n_experts, int capacity, int* expert,
                                 int* slot, float* weight, int* kept_counts, long long* exp_counts, float* l_aux, hipStream_t stream) {
  MP_REQUIRE(n_experts >= 1 && n_experts <= MAXE && tokens > 0 && capacity >= 0, MP_ERR_SHAPE, "mp_moe_route_top1: bad shape");
  hipLaunchKernelGGL(moe_route_top1_kernel, dim3(1), dim3(1024), 0, stream, gates, rts_uniform, tokens, n_experts, capacity, expert,
                     slot, weight, kept_counts, exp_counts, l_aux);
  return mp_check_launch("mp_moe_route_top1");
}

extern "C" int mp_moe_dispatch_bf16(const void* x, int64_t ldx, const int* expert, const int* slot, void* buf, int64_t tokens, int dim,
                                    int capacity, hipStream_t stream) {
  MP_REQUIRE(dim % 8 == 0 && ldx % 8 == 0, MP_ERR_SHAPE, "mp_moe_dispatch_bf16: bad shape");
  const int64_t n = tokens * (dim / 8);
  if (n == 0) return MP_OK;
  hipLaunchKernelGGL(moe_dispatch_kernel, dim3((unsigned)mp_cdiv(n, 256)), dim3(256), 0, stream, (const bf16_t*)x, ldx, expert, slot,
                     (bf16_t*)buf, tokens, dim, capacity);
  return mp_check_launch("mp_moe_dispatch_bf16");
}

extern "C" int mp_moe_combine_bf16(const void* y, const int* expert, const int* slot, const float* weight, const void* residual,
                                   void* out, int64_t tokens, int dim, int capacity, hipStream_t stream) {
  MP_REQUIRE(dim % 8 == 0, MP_ERR_SHAPE, "mp_moe_combine_bf16: bad shape");
  const int64_t n = tokens * (dim / 8);
  if (n == 0) return MP_OK;
  hipLaunchKernelGGL(moe_combine_kernel, dim3((unsigned)mp_cdiv(n, 256)), dim3(256), 0, stream, (const bf16_t*)y, expert, slot, weight,
                     (const bf16_t*)residual, (bf16_t*)out, tokens, dim, capacity);
  return mp_check_launch("mp_moe_combine_bf16");
}
