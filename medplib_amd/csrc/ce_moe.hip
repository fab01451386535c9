// Cross-entropy over supervised rows and the DeepSpeed-style MoE routing kernels (gate, top-1 routing with capacity /
// random-token-selection, dispatch, combine).  Index outputs (expert id, slot) are integer-exact.
//
// Reference sites: medplib_moe_llama.py:388-421 (fp32 logits, shift, all-ignored-row filter, mean CE, + aux loss);
// DeepSpeed 0.13.1 `deepspeed.moe.sharded_moe.{TopKGate, top1gating, MOELayer}` called from medplib_moe_llama.py:604-614
// (third-party, restated in SURVEY.md Appendix A.3 — parity unpinned by any reference test).
// The reference dispatches/combines with dense one-hot einsums ("sec,sm->ecm"); here both are index gathers.
#include "common.h"
#include <algorithm>

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
// one token's gate by one wave: fp32 logits and their softmax (lane 0 writes them)
// the end of a token's gate: wave-reduce the E partial logits, softmax on lane 0 (shared by every form of the gate so they agree to the bit)
__device__ __forceinline__ void moe_gate_finish(float (&acc)[MAXE], int E, int lane, float* __restrict__ logits_out, float* __restrict__ gates_out) {
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
      if (e < E) {
        if (logits_out) logits_out[e] = acc[e];
        gates_out[e] = p[e] / s;
      }
  }
}

__device__ __forceinline__ void moe_gate_token(const bf16_t* __restrict__ xr, const float* __restrict__ wg, int d, int E, int lane,
                                               float* __restrict__ logits_out, float* __restrict__ gates_out) {
  float acc[MAXE];
#pragma unroll
  for (int e = 0; e < MAXE; ++e) acc[e] = 0.f;
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
  moe_gate_finish(acc, E, lane, logits_out, gates_out);
}

__global__ __launch_bounds__(256) void moe_gate_kernel(const bf16_t* __restrict__ x, int64_t ldx, const float* __restrict__ wg,
                                                       float* __restrict__ logits, float* __restrict__ gates, int64_t T, int d, int E) {
  const int lane = threadIdx.x & 63;
  const int64_t tok = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (tok >= T) return;
  moe_gate_token(x + tok * ldx, wg, d, E, lane, logits + tok * E, gates + tok * E);
}

// ---------------- post-attention RMSNorm + MoE gate in one pass (one WAVE per token row) ----------------
// h = rmsnorm(x) * w (HF LlamaRMSNorm rounding points) and the gate of mp_moe_gate_bf16 on that h, while the row is still in
// registers: the separate gate kernel re-read all of h (42 MB per layer at the 7B shape).  Both results are BIT-IDENTICAL with the
// two stand-alone kernels: lane l holds the 16-byte chunks q = k*64 + l — the chunk a thread (c = q >> 8, t = q & 255) of
// rmsnorm_bf16_kernel's 256-thread block owns — so the sum of squares is accumulated per "virtual wave" w = (q >> 6) & 3 in that
// kernel's order (c ascending, then the wave butterfly, then red[0] + red[1] + red[2] + red[3]); and q = k*64 + l is also the chunk
// moe_gate_token's lane l reads in its k-th iteration, so the fp32 logits follow the same fma sequence.
// LDSW: the norm weight and the gate matrix ((1 + E) * dim floats) are staged in LDS once per workgroup, which then walks rows with
// stride gridDim.x * 4 -- read per row from global memory they were 48 KB of L1 traffic beside the row's own 8 KB (32.7 us per launch at
// the 7B shape against 17.8 for the RMSNorm alone); same values, same order of operations.
// (Requesting the NEXT row's chunks before this row's arithmetic -- a wave walks 2-3 rows here -- measured 22.5 -> 32.9 us on the same box:
// the second row buffer's registers cost more occupancy than the overlap returns.)
template <int NCH, bool LDSW>      // 16-byte chunks per lane: dim = NCH * 512
__global__ __launch_bounds__(512) void rmsnorm_gate_kernel(const bf16_t* __restrict__ x, int64_t ldx, const float* __restrict__ w, float eps,
                                                           bf16_t* __restrict__ h, int64_t ldh, const float* __restrict__ wg, int E,
                                                           float* __restrict__ logits, float* __restrict__ gates, int64_t T, float* __restrict__ rstd = nullptr) {
  extern __shared__ __attribute__((aligned(16))) float wsh[];
  constexpr int dim = NCH * 512;
  const int lane = threadIdx.x & 63;
  // 4 or 8 waves per workgroup (blockDim).  Round 4 tried 8 (639 workgroups at the 7B shape: every row its own wave, 16-24 rows in flight per CU
  // instead of 8 waves per CU walking 2-3 rows each): 27.3 us against 22.7 with 4 (scripts/r04_rg_ab.sh, cache-warm rows; the RMSNorm alone 14.8) —
  // more staging traffic (48 KB of weights per workgroup) and no gain from the extra rows in flight.  4 stays; MP_RG_WAVES=8 selects the other.
  const int nwv = blockDim.x >> 6, step4 = blockDim.x * 4;
  if constexpr (LDSW) {
    for (int i = threadIdx.x * 4; i < dim; i += step4) *reinterpret_cast<f32x4*>(wsh + i) = *reinterpret_cast<const f32x4*>(w + i);
    for (int i = threadIdx.x * 4; i < E * dim; i += step4) *reinterpret_cast<f32x4*>(wsh + dim + i) = *reinterpret_cast<const f32x4*>(wg + i);
    __syncthreads();
  }
  const float* wn = LDSW ? wsh : w;
  const float* wgs = LDSW ? wsh + dim : wg;
  for (int64_t tok = (int64_t)blockIdx.x * nwv + (threadIdx.x >> 6); tok < T; tok += (int64_t)gridDim.x * nwv) {
    const bf16_t* xr = x + tok * ldx;
    bf16x8 v[NCH];
    float part[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
      v[k] = *reinterpret_cast<const bf16x8*>(xr + (k * 64 + lane) * 8);
#pragma unroll
      for (int j = 0; j < 8; ++j) { const float f = (float)v[k][j]; part[k & 3] += f * f; }
    }
    float ss = 0.f;
#pragma unroll
    for (int wv = 0; wv < 4; ++wv) ss += wave_sum(part[wv]);
    const float rs = rsqrtf(ss / (float)dim + eps);
    // the folded-norm form (mp_rmsnorm_gate_rstd_bf16): the consumer GEMM takes the raw rows and applies rs in its epilogue, so the normalised
    // row is never written — only its scale; the gate is still computed from the HF-rounded h values, so the routing does not move
    if (rstd && lane == 0) rstd[tok] = rs;
    if (!h && E == 0) continue;
    float acc[MAXE];
#pragma unroll
    for (int e = 0; e < MAXE; ++e) acc[e] = 0.f;
    bf16_t* hr = h + tok * ldh;
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
      const int i = (k * 64 + lane) * 8;
      const f32x4 w0 = *reinterpret_cast<const f32x4*>(wn + i), w1 = *reinterpret_cast<const f32x4*>(wn + i + 4);
      bf16x8 o;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const bf16_t t = (bf16_t)((float)v[k][j] * rs);          // HF: the normalised value is cast to the input dtype first
        o[j] = (bf16_t)((j < 4 ? w0[j & 3] : w1[j & 3]) * (float)t);
      }
      if (h) *reinterpret_cast<bf16x8*>(hr + i) = o;
#pragma unroll
      for (int e = 0; e < MAXE; ++e) {
        if (e < E) {
          const float* wr = wgs + (int64_t)e * dim + i;
          const f32x4 g0 = *reinterpret_cast<const f32x4*>(wr), g1 = *reinterpret_cast<const f32x4*>(wr + 4);
#pragma unroll
          for (int j = 0; j < 8; ++j) acc[e] = fmaf((float)o[j], j < 4 ? g0[j & 3] : g1[j & 3], acc[e]);
        }
      }
    }
    if (E == 0) continue;
    float mx = -INFINITY;
#pragma unroll
    for (int e = 0; e < MAXE; ++e)
      if (e < E) { acc[e] = wave_sum(acc[e]); mx = fmaxf(mx, acc[e]); }
    if (lane == 0) {
      float sum = 0.f, pr[MAXE];
#pragma unroll
      for (int e = 0; e < MAXE; ++e)
        if (e < E) { pr[e] = expf(acc[e] - mx); sum += pr[e]; }
#pragma unroll
      for (int e = 0; e < MAXE; ++e)
        if (e < E) {
          if (logits) logits[tok * E + e] = acc[e];
          gates[tok * E + e] = pr[e] / sum;
        }
    }
  }
}

// ---------------- top-1 routing (single block, 1024 threads) ----------------
// expert[s] = argmax gates[s]; capacity drop: if an expert is over capacity keep the `capacity` tokens with the largest
// uniform draws (DeepSpeed RTS; first-come when no draws are supplied); slot[s] = rank among KEPT tokens of that expert
// in token order (cumsum(mask1) - 1), -1 if dropped; weight[s] = gates[s, expert] (not renormalised).
// l_aux = E * sum_e mean_s(gates[:,e]) * mean_s(mask1[:,e])   (computed BEFORE dropping).
template <bool FAST>      // FAST: T <= 16 * 1024 -- a thread keeps its chunk's choices in registers (no second trip through global memory)
__global__ __launch_bounds__(1024) void moe_route_top1_kernel(const float* __restrict__ gates, const float* __restrict__ rts, int T,
                                                              int E, int capacity, int* __restrict__ expert, int* __restrict__ slot,
                                                              float* __restrict__ weight, int* __restrict__ kept_counts,
                                                              long long* __restrict__ exp_counts, float* __restrict__ l_aux,
                                                              int* __restrict__ slot_token) {
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
  // FAST: the thread walks its own contiguous chunk (the same chunk the scan below owns) and remembers expert and keep flag per token,
  // so the common case -- no expert over capacity -- never re-reads expert[] / slot[] from memory between the phases (26 -> ~10 us at
  // 5112 tokens); otherwise tokens are taken 1024 apart and the later phases go through global memory.
  const int chunk = (T + 1023) / 1024;
  const int s0 = min(T, tid * chunk), s1 = min(T, s0 + chunk);
  int ex[FAST ? 16 : 1];
  if constexpr (FAST) {
    // expert by expert with the whole chunk's gate values of that expert in flight together: E round trips to memory instead of one per
    // token (a runtime-length token loop is not unrolled, and each of its iterations waited for its own loads)
    float bvv[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { bvv[i] = 0.f; ex[i] = 0; }
    for (int e = 0; e < E; ++e) {
      float gcol[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) gcol[i] = (s0 + i < s1) ? gates[(int64_t)(s0 + i) * E + e] : 0.f;
      float msum = 0.f;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        if (s0 + i < s1) {
          if (e == 0 || gcol[i] > bvv[i]) { bvv[i] = gcol[i]; ex[i] = e; }
          msum += gcol[i];
        }
      }
#pragma unroll
      for (int k = 0; k < MAXE; ++k) if (k == e) me[k] = msum;
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if (s0 + i < s1) {
        expert[s0 + i] = ex[i];
        weight[s0 + i] = bvv[i];
#pragma unroll
        for (int k = 0; k < MAXE; ++k) if (k == ex[i]) cnt[k] += 1;
      }
    }
  } else {
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
  }
  float aux = 0.f;
  for (int e = 0; e < E; ++e) {
    float m = 0.f; int c = 0;
#pragma unroll
    for (int k = 0; k < MAXE; ++k) if (k == e) { m = me[k]; c = cnt[k]; }
    const float msum = block_sum(m, red);
    // one LDS atomic per WAVE (16 of them), not per thread: up to 1024 atomics on a single LDS word serialise
    int cw = c;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) cw += __shfl_xor(cw, off, 64);
    if ((tid & 63) == 0 && cw) atomicAdd(&cnt_sh[e], cw);
    __syncthreads();
    aux += (msum / (float)T) * ((float)cnt_sh[e] / (float)T);
  }
  if (tid == 0) {
    l_aux[0] = aux * (float)E;
    for (int e = 0; e < E; ++e) exp_counts[e] = cnt_sh[e];
  }
  __syncthreads();
  // keep decision (global memory `slot` temporarily holds the keep flag).  Without RTS draws the selection is first-come: the
  // rank is the token-order prefix count computed by the scan below, so every token stays flagged and slots >= capacity are
  // dropped after the scan.  With draws (DeepSpeed use_rts): an over-capacity expert keeps its `capacity` LARGEST draws (ties:
  // lower token index first, the order torch.topk's stable sort yields) -- found by an 8-bit-per-pass radix select over the
  // order-preserving integer image of the fp32 draws (4 histogram passes over T values instead of an O(T^2) rank count).
  const int lane = tid & 63, wv = tid >> 6;
  bool over = false;                                         // block-uniform: some expert was chosen by more tokens than it can hold
  for (int e = 0; e < E; ++e) over |= cnt_sh[e] > capacity;
  const bool via_memory = !FAST || (rts && over);            // keep flags travel through slot[] (the draws' selection writes them)
  if (via_memory) {
    for (int s = tid; s < T; s += 1024) slot[s] = 1;
    __syncthreads();
  }
  if (rts && over) {
    __shared__ unsigned hist[256];
    __shared__ unsigned bin_suf[256], bin_tot[4];
    __shared__ int bin_hi[4];
    __shared__ unsigned sel_bin, sel_need;
    __shared__ int eq_scan[16];
    for (int e = 0; e < E; ++e) {
      if (cnt_sh[e] <= capacity) continue;                 // block-uniform
      unsigned prefix = 0, pmask = 0, need = (unsigned)capacity;
      if (capacity == 0) {
        for (int s = tid; s < T; s += 1024) if (expert[s] == e) slot[s] = 0;
        __syncthreads();
        continue;
      }
      // FAST: the keys of this thread's own chunk tokens that chose expert e, loaded ONCE (16 loads in flight) — the four passes, the tie
      // count and the keep decision below then run from registers; re-reading expert[] / rts[] in every pass was one dependent round
      // trip to memory per token per pass (the draws' selection: 46 us at 5112 tokens with it)
      unsigned keyr[FAST ? 16 : 1];
      unsigned minem = 0;                                    // bit i: chunk token i chose expert e
      if constexpr (FAST) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          keyr[i] = 0;
          if (s0 + i < s1 && ex[i] == e) {
            const unsigned b = __float_as_uint(rts[(int64_t)(s0 + i) * E + e]);
            keyr[i] = b ^ ((b >> 31) ? 0xffffffffu : 0x80000000u);
            minem |= 1u << i;
          }
        }
      }
      for (int pass = 0; pass < 4; ++pass) {
        const int shift = 24 - 8 * pass;
        if (tid < 256) hist[tid] = 0;
        __syncthreads();
        if constexpr (FAST) {
#pragma unroll
          for (int i = 0; i < 16; ++i)
            if (((minem >> i) & 1u) && (keyr[i] & pmask) == prefix) atomicAdd(&hist[(keyr[i] >> shift) & 255u], 1u);
        } else
        for (int s = tid; s < T; s += 1024) {
          if (expert[s] != e) continue;
          const unsigned b = __float_as_uint(rts[(int64_t)s * E + e]);
          const unsigned key = b ^ ((b >> 31) ? 0xffffffffu : 0x80000000u);
          if ((key & pmask) == prefix) atomicAdd(&hist[(key >> shift) & 255u], 1u);
        }
        __syncthreads();
        // the bin that holds the need-th largest key: the LARGEST b with suffix(b) = hist[b] + ... + hist[255] >= need (bin 0 if none), and
        // what is still needed inside it, need - suffix(b + 1).  Four waves scan the 256 bins (thread t <-> bin t) instead of one thread
        // walking them: the serial walk was 256 dependent LDS reads per pass, four passes per over-capacity expert.
        unsigned suf = 0;
        if (tid < 256) {
          suf = hist[tid];                                    // inclusive suffix sum within the wave (this lane and the lanes above it)
#pragma unroll
          for (int off = 1; off < 64; off <<= 1) {
            const unsigned n = __shfl_down(suf, off, 64);
            if (lane + off < 64) suf += n;
          }
          if (lane == 0) bin_tot[wv] = suf;                   // the wave's total
        }
        __syncthreads();
        if (tid < 256) {
          for (int w = wv + 1; w < 4; ++w) suf += bin_tot[w];
          const unsigned long long m = __ballot(suf >= need);
          if (lane == 0) bin_hi[wv] = m ? 63 - __builtin_clzll(m) + 64 * wv : -1;
          bin_suf[tid] = suf;
        }
        __syncthreads();
        if (tid == 0) {
          int b = 0;
          for (int w = 3; w >= 0; --w) if (bin_hi[w] >= 0) { b = bin_hi[w]; break; }
          const unsigned above = b < 255 ? bin_suf[b + 1] : 0u;
          sel_bin = (unsigned)b; sel_need = need - above;
        }
        __syncthreads();
        prefix |= sel_bin << shift; pmask |= 0xffu << shift; need = sel_need;
        __syncthreads();
      }
      // prefix = key of the capacity-th largest draw; `need` of the tokens holding exactly that key are kept, in token order
      int eq = 0;
      if constexpr (FAST) {
#pragma unroll
        for (int i = 0; i < 16; ++i) eq += (((minem >> i) & 1u) && keyr[i] == prefix);
      } else
      for (int s = s0; s < s1; ++s) {
        if (expert[s] != e) continue;
        const unsigned b = __float_as_uint(rts[(int64_t)s * E + e]);
        eq += ((b ^ ((b >> 31) ? 0xffffffffu : 0x80000000u)) == prefix);
      }
      int v = eq;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const int n = __shfl_up(v, off, 64);
        if (lane >= off) v += n;
      }
      if (lane == 63) eq_scan[wv] = v;
      __syncthreads();
      int rank = v - eq;
      for (int w = 0; w < wv; ++w) rank += eq_scan[w];
      if constexpr (FAST) {
#pragma unroll
        for (int i = 0; i < 16; ++i)
          if ((minem >> i) & 1u) {
            int keep = keyr[i] > prefix;
            if (keyr[i] == prefix) { keep = rank < (int)need; ++rank; }
            slot[s0 + i] = keep;
          }
      } else
      for (int s = s0; s < s1; ++s) {
        if (expert[s] != e) continue;
        const unsigned b = __float_as_uint(rts[(int64_t)s * E + e]);
        const unsigned key = b ^ ((b >> 31) ? 0xffffffffu : 0x80000000u);
        int keep = key > prefix;
        if (key == prefix) { keep = rank < (int)need; ++rank; }
        slot[s] = keep;
      }
      __syncthreads();
    }
  }
  __syncthreads();
  // exclusive scan of kept tokens per expert in token order: thread `tid` owns a contiguous chunk; wave-level shuffle scan of
  // the per-thread totals, then the 16 wave totals are combined through LDS
  int loc[MAXE];
#pragma unroll
  for (int e = 0; e < MAXE; ++e) loc[e] = 0;
  unsigned keepm = 0;                                        // keep flag of chunk token i in bit i (FAST)
  for (int s = s0; s < s1; ++s) {
    const bool kp = via_memory ? slot[s] != 0 : true;
    if (kp) {
      int e = 0;
      if constexpr (FAST) {
#pragma unroll
        for (int i = 0; i < 16; ++i) if (i == s - s0) e = ex[i];
        keepm |= 1u << (s - s0);
      } else {
        e = expert[s];
      }
#pragma unroll
      for (int k = 0; k < MAXE; ++k) if (k == e) loc[k] += 1;
    }
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
    bool kp;
    int e = 0;
    if constexpr (FAST) {
      kp = (keepm >> (s - s0)) & 1u;
#pragma unroll
      for (int i = 0; i < 16; ++i) if (i == s - s0) e = ex[i];
    } else {
      kp = slot[s] != 0;
      e = expert[s];
    }
    if (kp) {
      int v = 0;
#pragma unroll
      for (int k = 0; k < MAXE; ++k) if (k == e) { v = base[k]; base[k] += 1; }
      slot[s] = (v < capacity) ? v : -1;
      if (slot_token && v < capacity) slot_token[(int64_t)e * capacity + v] = s;      // inverse map: the expert GEMMs gather / scatter by it
    } else {
      slot[s] = -1;
    }
  }
  if (tid == 1023)
    for (int e = 0; e < E; ++e) kept_counts[e] = min(base[e], capacity);
}

// ---------------- top-1 routing for a handful of tokens (T <= 64: the decode steps): one wave, lane = token ----------------
// Same outputs as moe_route_top1_kernel; the 1024-thread kernel spends ~25 us in block-wide reductions, which at 32 layers is a
// fifth of a decode step.
__device__ __forceinline__ void moe_route_top1_small_body(const float* __restrict__ gates, const float* __restrict__ rts, int T,
                                                               int E, int capacity, int* __restrict__ expert, int* __restrict__ slot,
                                                                  float* __restrict__ weight, int* __restrict__ kept_counts,
                                                                  long long* __restrict__ exp_counts, float* __restrict__ l_aux,
                                                                  int* __restrict__ slot_token, int s) {
  const bool live = s < T;
  int best = 0;
  float bv = live ? gates[(int64_t)s * E] : 0.f;
  float aux = 0.f;
  for (int e = 1; e < E; ++e) {
    const float g = live ? gates[(int64_t)s * E + e] : 0.f;
    if (g > bv) { bv = g; best = e; }
  }
  for (int e = 0; e < E; ++e) {
    const float me = wave_sum(live ? gates[(int64_t)s * E + e] : 0.f) / (float)T;
    const unsigned long long m = __ballot(live && best == e);
    aux += me * ((float)__popcll(m) / (float)T);
    if (s == 0) exp_counts[e] = __popcll(m);
  }
  if (s == 0) l_aux[0] = aux * (float)E;
  // keep decision: over-capacity experts keep their `capacity` largest draws (ties: lower token first) or, without draws, the
  // first `capacity` tokens
  int keep = live ? 1 : 0;
  unsigned long long same = 0;                      // the live tokens that chose this lane's expert
  for (int e = 0; e < E; ++e) {
    const unsigned long long m = __ballot(live && best == e);
    if (best == e) same = m;
  }
  if (live && __popcll(same) > capacity) {
    int rank = 0;
    if (rts) {
      const float u = rts[(int64_t)s * E + best];
      for (int t = 0; t < T; ++t) {
        if (!((same >> t) & 1ull)) continue;
        const float ut = rts[(int64_t)t * E + best];
        rank += (ut > u) || (ut == u && t < s);
      }
    } else {
      rank = __popcll(same & ((1ull << s) - 1ull));
    }
    keep = rank < capacity;
  }
  // slots = rank among the KEPT tokens of the same expert in token order
  int my_slot = -1;
  for (int e = 0; e < E; ++e) {
    const unsigned long long km = __ballot(live && keep && best == e);
    if (live && keep && best == e) my_slot = __popcll(km & ((1ull << s) - 1ull));
    if (s == 0) kept_counts[e] = min(__popcll(km), capacity);
  }
  if (live) {
    expert[s] = best; weight[s] = bv; slot[s] = my_slot;
    if (slot_token && my_slot >= 0) slot_token[(int64_t)best * capacity + my_slot] = s;
  }
}

__global__ __launch_bounds__(64) void moe_route_top1_small_kernel(const float* __restrict__ gates, const float* __restrict__ rts, int T,
                                                                  int E, int capacity, int* __restrict__ expert, int* __restrict__ slot,
                                                                  float* __restrict__ weight, int* __restrict__ kept_counts,
                                                                  long long* __restrict__ exp_counts, float* __restrict__ l_aux,
                                                                  int* __restrict__ slot_token) {
  moe_route_top1_small_body(gates, rts, T, E, capacity, expert, slot, weight, kept_counts, exp_counts, l_aux, slot_token, threadIdx.x);
}

// ---------------- decode rows (T <= 8): post-attention RMSNorm + gate + top-1 routing in ONE launch ----------------
// The three kernels it replaces cost 4.8 + 11.6 + 4.9 us per layer of a decode step, almost all of it launch latency.  Same
// arithmetic, same order: the norm is rmsnorm_bf16_kernel's 256-thread row (norm_elementwise.hip), the gate moe_gate_token, the
// routing moe_route_top1_small_body -- the fused and the separate path produce identical bits.
__global__ __launch_bounds__(256) void decode_norm_gate_route_kernel(const bf16_t* __restrict__ x, int64_t ldx, const float* __restrict__ ln_w,
                                                                     float eps, const float* __restrict__ wg, bf16_t* __restrict__ h,
                                                                     int64_t ldh, const float* __restrict__ rts, int T, int d, int E,
                                                                     int capacity, float* __restrict__ gates_out, int* __restrict__ expert,
                                                                     int* __restrict__ slot, float* __restrict__ weight,
                                                                     int* __restrict__ kept_counts, long long* __restrict__ exp_counts,
                                                                     float* __restrict__ l_aux) {
  __shared__ float red[16];
  __shared__ float gates_sh[8 * MAXE];
  constexpr int NC = 4;                                // dim <= 256 * 8 * 4
  // The decode steps' shape (d = 4096, E <= 2, T <= 4 rows): a row's gate is ONE wave's serial chain of 8 dependent trips to memory in
  // moe_gate_token (a runtime loop: x chunk, then the experts' weight chunks).  Here the wave requests its 8 x E weight chunks when the
  // kernel starts — they travel under the RMSNorm — and takes the normalised row from LDS instead of reading it back from global
  // memory; the fma chain per expert is the same (chunks ascending, elements ascending), so the bits are moe_gate_token's.
  constexpr int FK = 8;                                // 16-byte x chunks per lane at d = 4096
  __shared__ __attribute__((aligned(16))) bf16_t h_sh[4 * 4096];
  const bool fast = (d == FK * 512) && E <= 2 && T <= 4;
  const int lane_f = threadIdx.x & 63, wave_f = threadIdx.x >> 6;
  f32x4 gw[2][FK][2];
  if (fast && wave_f < T) {
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
      for (int k = 0; k < FK; ++k) {
        const float* wp_ = wg + (int64_t)(e < E ? e : 0) * d + (k * 64 + lane_f) * 8;
        gw[e][k][0] = *reinterpret_cast<const f32x4*>(wp_); gw[e][k][1] = *reinterpret_cast<const f32x4*>(wp_ + 4);
      }
  }
  for (int row = 0; row < T; ++row) {
    const bf16_t* xr = x + row * ldx;
    bf16_t* hr = h + row * ldh;
    bf16x8 v[NC];
    float ss = 0.f;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const int i = (c * 256 + threadIdx.x) * 8;
      if (i < d) {
        v[c] = *reinterpret_cast<const bf16x8*>(xr + i);
#pragma unroll
        for (int j = 0; j < 8; ++j) { const float f = (float)v[c][j]; ss += f * f; }
      }
    }
    ss = block_sum(ss, red);
    const float rs = rsqrtf(ss / (float)d + eps);
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const int i = (c * 256 + threadIdx.x) * 8;
      if (i < d) {
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const bf16_t t = (bf16_t)((float)v[c][j] * rs);
          o[j] = (bf16_t)(ln_w[i + j] * (float)t);
        }
        *reinterpret_cast<bf16x8*>(hr + i) = o;
        if (fast) *reinterpret_cast<bf16x8*>(h_sh + row * 4096 + i) = o;
      }
    }
  }
  __syncthreads();                                     // the normed rows (global / LDS) are visible to the whole workgroup
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (fast) {
    if (wave < T) {
      float acc[MAXE];
#pragma unroll
      for (int e = 0; e < MAXE; ++e) acc[e] = 0.f;
#pragma unroll
      for (int k = 0; k < FK; ++k) {
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(h_sh + wave * 4096 + (k * 64 + lane) * 8);
#pragma unroll
        for (int e = 0; e < 2; ++e)
          if (e < E) {
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[e] = fmaf((float)v[j], j < 4 ? gw[e][k][0][j & 3] : gw[e][k][1][j & 3], acc[e]);
          }
      }
      moe_gate_finish(acc, E, lane, nullptr, gates_sh + wave * E);
    }
  } else
  for (int row = wave; row < T; row += 4) moe_gate_token(h + row * ldh, wg, d, E, lane, nullptr, gates_sh + row * E);
  __syncthreads();
  if (gates_out)
    for (int i = threadIdx.x; i < T * E; i += 256) gates_out[i] = gates_sh[i];
  if (wave == 0)
    moe_route_top1_small_body(gates_sh, rts, T, E, capacity, expert, slot, weight, kept_counts, exp_counts, l_aux, nullptr, lane);
}

// DeepSpeed "residual MoE" (MoE(use_residual=True), deepspeed/moe/layer.py forward): out = x + (moe * c0 + mlp * c1) with
// c = softmax(coefficient(h)) over two logits; bf16 rounding points of the bf16 module: the softmax result, each product, their sum,
// the decoder layer's residual add.
__global__ void moe_residual_mix_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ moe, const bf16_t* __restrict__ mlp,
                                        const bf16_t* __restrict__ coef, int64_t ldcoef, bf16_t* __restrict__ out, int64_t T, int d) {
  const int per_row = d / 8;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= T * per_row) return;
  const int64_t s = idx / per_row;
  const int c = (int)(idx % per_row) * 8;
  const float l0 = (float)coef[s * ldcoef], l1 = (float)coef[s * ldcoef + 1];
  const float mx = fmaxf(l0, l1);
  const float e0 = expf(l0 - mx), e1 = expf(l1 - mx);
  const float c0 = (float)(bf16_t)(e0 / (e0 + e1)), c1 = (float)(bf16_t)(e1 / (e0 + e1));
  const bf16x8 xv = *reinterpret_cast<const bf16x8*>(x + s * d + c);
  const bf16x8 a = *reinterpret_cast<const bf16x8*>(moe + s * d + c);
  const bf16x8 b = *reinterpret_cast<const bf16x8*>(mlp + s * d + c);
  bf16x8 o;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float t0 = (float)(bf16_t)((float)a[j] * c0), t1 = (float)(bf16_t)((float)b[j] * c1);
    o[j] = (bf16_t)((float)xv[j] + (float)(bf16_t)(t0 + t1));
  }
  *reinterpret_cast<bf16x8*>(out + s * d + c) = o;
}

// buf[expert[j*T + s], slot[j*T + s], :] = x[s, :]   for the top_k choices j of token s
__global__ void moe_dispatch_kernel(const bf16_t* __restrict__ x, int64_t ldx, const int* __restrict__ expert, const int* __restrict__ slot,
                                    bf16_t* __restrict__ buf, int64_t ldbuf, int64_t T, int d, int capacity, int top_k) {
  const int per_row = d / 8;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= T * top_k * per_row) return;
  const int64_t en = idx / per_row;              // entry = choice * T + token
  const int64_t s = en % T;
  const int c = (int)(idx % per_row) * 8;
  const int sl = slot[en];
  if (sl < 0) return;
  *reinterpret_cast<bf16x8*>(buf + ((int64_t)expert[en] * capacity + sl) * ldbuf + c) = *reinterpret_cast<const bf16x8*>(x + s * ldx + c);
}

// out[s, :] = residual[s, :] + sum_j weight[j*T + s] * y[expert[j*T + s], slot[j*T + s], :]     (dropped choices contribute nothing)
__global__ void moe_combine_kernel(const bf16_t* __restrict__ y, const int* __restrict__ expert, const int* __restrict__ slot,
                                   const float* __restrict__ weight, const bf16_t* __restrict__ residual, bf16_t* __restrict__ out,
                                   int64_t T, int d, int capacity, int top_k) {
  const int per_row = d / 8;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= T * per_row) return;
  const int64_t s = idx / per_row;
  const int c = (int)(idx % per_row) * 8;
  float acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = 0.f;
  for (int k = 0; k < top_k; ++k) {
    const int64_t en = (int64_t)k * T + s;
    const int sl = slot[en];
    if (sl < 0) continue;
    const bf16x8 v = *reinterpret_cast<const bf16x8*>(y + ((int64_t)expert[en] * capacity + sl) * d + c);
    const float w = weight[en];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = fmaf(w, (float)v[j], acc[j]);
  }
  bf16x8 o;
  if (residual) {
    const bf16x8 r = *reinterpret_cast<const bf16x8*>(residual + s * d + c);
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (bf16_t)((float)r[j] + acc[j]);
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (bf16_t)acc[j];
  }
  *reinterpret_cast<bf16x8*>(out + s * d + c) = o;
}

// out[s, :] = x[s, :] for the tokens no expert took (slot < 0): the residual-only rows when combine is fused into the expert
// down-projection's epilogue (which writes every routed token's row exactly once)
__global__ void moe_fill_dropped_kernel(const bf16_t* __restrict__ x, const int* __restrict__ slot, bf16_t* __restrict__ out, int64_t T,
                                        int d) {
  const int per_row = d / 8;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= T * per_row) return;
  const int64_t s = idx / per_row;
  if (slot[s] >= 0) return;
  const int c = (int)(idx % per_row) * 8;
  *reinterpret_cast<bf16x8*>(out + s * d + c) = *reinterpret_cast<const bf16x8*>(x + s * d + c);
}

// ---------------- top-2 routing (single block, 1024 threads) ----------------
// DeepSpeed 0.13.1 top2gating: e1 = argmax gates; e2 = argmax over e != e1 of logits + noise (Gumbel
// draws, `top2_2nd_expert_sampling`; null = no noise); loc1 = cumsum(mask1) - 1, loc2 = cumsum(mask2) - 1 + sum(mask1);
// l_aux = E^2 * mean_e(mean_s gates * mean_s mask1); exp_counts = sum(mask1 + mask2) before dropping; a choice is dropped when
// its location >= capacity (first-come by token position); the two surviving gate values are renormalised to sum to 1
// (denominator clamped at fp32 eps).  Entry layout: first choices at [0, T), second choices at [T, 2T).
__global__ __launch_bounds__(1024) void moe_route_top2_kernel(const float* __restrict__ gates, const float* __restrict__ logits,
                                                              const float* __restrict__ noise, int T, int E, int capacity,
                                                              int* __restrict__ expert, int* __restrict__ slot, float* __restrict__ weight,
                                                              int* __restrict__ kept_counts, long long* __restrict__ exp_counts,
                                                              float* __restrict__ l_aux) {
  __shared__ float red[16];
  __shared__ int tot1[MAXE], tot2[MAXE];
  __shared__ int scan1[16][MAXE], scan2[16][MAXE];
  const int tid = threadIdx.x;
  if (tid < MAXE) { tot1[tid] = 0; tot2[tid] = 0; }
  __syncthreads();
  // each thread owns a contiguous chunk of tokens (token order = cumsum order)
  const int chunk = (T + 1023) / 1024;
  const int s0 = min(T, tid * chunk), s1 = min(T, s0 + chunk);
  const int lane = tid & 63, wv = tid >> 6;
  float me[MAXE];
  int c1[MAXE], c2[MAXE];
#pragma unroll
  for (int e = 0; e < MAXE; ++e) { me[e] = 0.f; c1[e] = 0; c2[e] = 0; }
  for (int s = s0; s < s1; ++s) {
    int b1 = 0;
    float bv = gates[(int64_t)s * E];
    for (int e = 0; e < E; ++e) {
      const float g = gates[(int64_t)s * E + e];
      if (e > 0 && g > bv) { bv = g; b1 = e; }
#pragma unroll
      for (int k = 0; k < MAXE; ++k) if (k == e) me[k] += g;
    }
    int b2 = -1;
    float b2v = -INFINITY;
    for (int e = 0; e < E; ++e) {
      if (e == b1) continue;
      const float v = logits[(int64_t)s * E + e] + (noise ? noise[(int64_t)s * E + e] : 0.f);
      if (b2 < 0 || v > b2v) { b2v = v; b2 = e; }
    }
    if (b2 < 0) b2 = b1;                       // E == 1: degenerate, second choice dropped below
    expert[s] = b1; expert[T + s] = b2;
#pragma unroll
    for (int k = 0; k < MAXE; ++k) { if (k == b1) c1[k] += 1; if (k == b2 && E > 1) c2[k] += 1; }
  }
  // l_aux and totals
  float aux = 0.f;
  for (int e = 0; e < E; ++e) {
    float m = 0.f; int a = 0, b = 0;
#pragma unroll
    for (int k = 0; k < MAXE; ++k) if (k == e) { m = me[k]; a = c1[k]; b = c2[k]; }
    const float msum = block_sum(m, red);
    // one LDS atomic per wave and total (see moe_route_top1_kernel): per-thread atomics on one word serialise
    int aw = a, bw = b;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { aw += __shfl_xor(aw, off, 64); bw += __shfl_xor(bw, off, 64); }
    if ((tid & 63) == 0) { if (aw) atomicAdd(&tot1[e], aw); if (bw) atomicAdd(&tot2[e], bw); }
    __syncthreads();
    aux += (msum / (float)T) * ((float)tot1[e] / (float)T);
  }
  if (tid == 0) {
    l_aux[0] = aux * (float)E;               // mean_e(me*ce) * E * E
    for (int e = 0; e < E; ++e) { exp_counts[e] = (long long)tot1[e] + tot2[e]; kept_counts[e] = min(capacity, tot1[e] + tot2[e]); }
  }
  // exclusive prefix of the per-thread counts (token order) for both choices
  int base1[MAXE], base2[MAXE];
#pragma unroll
  for (int e = 0; e < MAXE; ++e) {
    base1[e] = 0; base2[e] = 0;
    if (e < E) {
      int v1 = c1[e], v2 = c2[e];
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const int n1 = __shfl_up(v1, off, 64), n2 = __shfl_up(v2, off, 64);
        if (lane >= off) { v1 += n1; v2 += n2; }
      }
      if (lane == 63) { scan1[wv][e] = v1; scan2[wv][e] = v2; }
      base1[e] = v1 - c1[e]; base2[e] = v2 - c2[e];
    }
  }
  __syncthreads();
#pragma unroll
  for (int e = 0; e < MAXE; ++e)
    if (e < E) {
      for (int w = 0; w < wv; ++w) { base1[e] += scan1[w][e]; base2[e] += scan2[w][e]; }
      base2[e] += tot1[e];                    // second choices sit behind ALL first choices of that expert
    }
  for (int s = s0; s < s1; ++s) {
    const int e1 = expert[s], e2 = expert[T + s];
    int l1 = 0, l2 = 0;
#pragma unroll
    for (int k = 0; k < MAXE; ++k) {
      if (k == e1) { l1 = base1[k]; base1[k] += 1; }
      if (k == e2 && E > 1) { l2 = base2[k]; base2[k] += 1; }
    }
    const bool k1 = l1 < capacity, k2 = (E > 1) && l2 < capacity;
    float g1 = k1 ? gates[(int64_t)s * E + e1] : 0.f;
    float g2 = k2 ? gates[(int64_t)s * E + e2] : 0.f;
    const float den = fmaxf(g1 + g2, 1.1920929e-07f);
    slot[s] = k1 ? l1 : -1; slot[T + s] = k2 ? l2 : -1;
    weight[s] = g1 / den; weight[T + s] = g2 / den;
  }
}

// ---------------- counter-based random draws for the gate (RTS uniforms, top-2 Gumbel noise) ----------------
// out[i] = U(0,1) (mode 0) or Gumbel(0,1) = -log(-log U) (mode 1) from a SplitMix64 hash of (seed, offset + i): stateless,
// reproducible for a given (seed, offset), independent of the launch geometry.  DeepSpeed draws these from torch's generator
// (sharded_moe.py: uniform_map / gumbel_rsample); the streams cannot match bit for bit, only in distribution.
__global__ void gate_noise_kernel(float* __restrict__ out, int64_t n, unsigned long long seed, unsigned long long offset, int mode) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  unsigned long long z = seed * 0x9E3779B97F4A7C15ull + (offset + (unsigned long long)i + 1ull) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z = z ^ (z >> 31);
  const float u = ((float)(z >> 40) + 0.5f) * (1.f / 16777216.f);      // 24 random bits, strictly inside (0, 1)
  out[i] = mode == 1 ? -logf(-logf(u)) : u;
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

static int rmsnorm_gate_launch(const void* x, int64_t ldx, const float* ln_w, float eps, void* h, int64_t ldh, const float* wg,
                                    float* logits, float* gates, int64_t tokens, int dim, int n_experts, float* rstd, hipStream_t stream) {
  MP_REQUIRE(n_experts >= 0 && n_experts <= MAXE && ldx % 8 == 0 && ldh % 8 == 0, MP_ERR_SHAPE, "mp_rmsnorm_gate_bf16: bad shape (E <= %d)", MAXE);
  MP_REQUIRE(dim == 2048 || dim == 4096 || dim == 8192, MP_ERR_SHAPE, "mp_rmsnorm_gate_bf16: dim %d (2048, 4096 or 8192)", dim);
  MP_REQUIRE(n_experts == 0 || (wg != nullptr && gates != nullptr), MP_ERR_ARG, "mp_rmsnorm_gate_bf16: gate outputs missing");
  if (tokens == 0) return MP_OK;
  const size_t lds = (size_t)(1 + n_experts) * dim * sizeof(float);
  const bool stage = lds <= 65536;                          // (dim 4096 with E <= 3, dim 2048 with E <= 7; else the weights stay in global memory)
  static int rg_waves = -1;
  if (rg_waves < 0) { const char* e = getenv("MP_RG_WAVES"); rg_waves = (e && atoi(e) == 8) ? 8 : 4; }       // 8: measured slower (see the kernel)
  const int nwv = (stage && tokens >= 2048) ? rg_waves : 4;
  const dim3 grid((unsigned)std::min<int64_t>(mp_cdiv(tokens, nwv), stage ? (nwv == 8 ? 768 : 512) : (1 << 30))), blk(64 * nwv);
#define MP_RG(N)                                                                                                                                   \
  do {                                                                                                                                             \
    if (stage) hipLaunchKernelGGL((rmsnorm_gate_kernel<N, true>), grid, blk, lds, stream, (const bf16_t*)x, ldx, ln_w, eps, (bf16_t*)h, ldh, wg,   \
                                  n_experts, logits, gates, tokens, rstd);                                                                               \
    else hipLaunchKernelGGL((rmsnorm_gate_kernel<N, false>), grid, blk, 0, stream, (const bf16_t*)x, ldx, ln_w, eps, (bf16_t*)h, ldh, wg,          \
                            n_experts, logits, gates, tokens, rstd);                                                                                     \
  } while (0)
  if (dim == 2048) MP_RG(4); else if (dim == 4096) MP_RG(8); else MP_RG(16);
#undef MP_RG
  return mp_check_launch("mp_rmsnorm_gate_bf16");
}

extern "C" int mp_rmsnorm_gate_bf16(const void* x, int64_t ldx, const float* ln_w, float eps, void* h, int64_t ldh, const float* wg, float* logits,
                                    float* gates, int64_t tokens, int dim, int n_experts, hipStream_t stream) {
  MP_REQUIRE(h != nullptr, MP_ERR_ARG, "mp_rmsnorm_gate_bf16: h is null (mp_rmsnorm_gate_rstd_bf16 is the form without the normalised rows)");
  return rmsnorm_gate_launch(x, ldx, ln_w, eps, h, ldh, wg, logits, gates, tokens, dim, n_experts, nullptr, stream);
}

// The folded-norm form: rstd [tokens] fp32 = 1 / sqrt(mean(x^2) + eps) instead of the normalised rows (the consumer GEMM reads x itself, its
// weights carry ln_w, its epilogue multiplies by rstd); logits / gates exactly as mp_rmsnorm_gate_bf16 computes them (from the bf16 h values,
// which stay in registers).  n_experts = 0: rstd alone (wg unused).
extern "C" int mp_rmsnorm_gate_rstd_bf16(const void* x, int64_t ldx, const float* ln_w, float eps, const float* wg, float* logits, float* gates,
                                         float* rstd, int64_t tokens, int dim, int n_experts, hipStream_t stream) {
  MP_REQUIRE(rstd != nullptr && ln_w != nullptr, MP_ERR_ARG, "mp_rmsnorm_gate_rstd_bf16: rstd and ln_w required");
  return rmsnorm_gate_launch(x, ldx, ln_w, eps, nullptr, 8, wg, logits, gates, tokens, dim, n_experts, rstd, stream);
}

extern "C" int mp_moe_route_top1(const float* gates, const float* rts_uniform, int tokens, int n_experts, int capacity, int* expert,
                                 int* slot, float* weight, int* kept_counts, long long* exp_counts, float* l_aux, int* slot_token,
                                 hipStream_t stream) {
  MP_REQUIRE(n_experts >= 1 && n_experts <= MAXE && tokens > 0 && capacity >= 0, MP_ERR_SHAPE, "mp_moe_route_top1: bad shape");
  if (tokens <= 64) {
    hipLaunchKernelGGL(moe_route_top1_small_kernel, dim3(1), dim3(64), 0, stream, gates, rts_uniform, tokens, n_experts, capacity, expert,
                       slot, weight, kept_counts, exp_counts, l_aux, slot_token);
    return mp_check_launch("mp_moe_route_top1(small)");
  }
  if (tokens <= 16 * 1024)
    hipLaunchKernelGGL(moe_route_top1_kernel<true>, dim3(1), dim3(1024), 0, stream, gates, rts_uniform, tokens, n_experts, capacity, expert,
                       slot, weight, kept_counts, exp_counts, l_aux, slot_token);
  else
    hipLaunchKernelGGL(moe_route_top1_kernel<false>, dim3(1), dim3(1024), 0, stream, gates, rts_uniform, tokens, n_experts, capacity, expert,
                       slot, weight, kept_counts, exp_counts, l_aux, slot_token);
  return mp_check_launch("mp_moe_route_top1");
}

extern "C" int mp_decode_norm_gate_route(const void* x, int64_t ldx, const float* ln_w, float eps, const float* wg, void* h, int64_t ldh,
                                        const float* rts_uniform, int tokens, int dim, int n_experts, int capacity, float* gates,
                                        int* expert, int* slot, float* weight, int* kept_counts, long long* exp_counts, float* l_aux,
                                        hipStream_t stream) {
  MP_REQUIRE(tokens >= 1 && tokens <= 8 && n_experts >= 1 && n_experts <= MAXE && dim % 8 == 0 && dim <= 8192 && ldx % 8 == 0 &&
                 ldh % 8 == 0 && capacity >= 0,
             MP_ERR_SHAPE, "mp_decode_norm_gate_route: tokens <= 8, experts <= %d, dim %% 8 == 0 and <= 8192", MAXE);
  hipLaunchKernelGGL(decode_norm_gate_route_kernel, dim3(1), dim3(256), 0, stream, (const bf16_t*)x, ldx, ln_w, eps, wg, (bf16_t*)h, ldh,
                     rts_uniform, tokens, dim, n_experts, capacity, gates, expert, slot, weight, kept_counts, exp_counts, l_aux);
  return mp_check_launch("mp_decode_norm_gate_route");
}

extern "C" int mp_moe_residual_mix_bf16(const void* x, const void* moe, const void* mlp, const void* coef, int64_t ldcoef, void* out,
                                       int64_t tokens, int dim, hipStream_t stream) {
  MP_REQUIRE(dim % 8 == 0 && ldcoef >= 2, MP_ERR_SHAPE, "mp_moe_residual_mix_bf16: bad shape");
  const int64_t n = tokens * (dim / 8);
  if (n == 0) return MP_OK;
  hipLaunchKernelGGL(moe_residual_mix_kernel, dim3((unsigned)mp_cdiv(n, 256)), dim3(256), 0, stream, (const bf16_t*)x, (const bf16_t*)moe,
                     (const bf16_t*)mlp, (const bf16_t*)coef, ldcoef, (bf16_t*)out, tokens, dim);
  return mp_check_launch("mp_moe_residual_mix_bf16");
}

extern "C" int mp_moe_fill_dropped_bf16(const void* x, const int* slot, void* out, int64_t tokens, int dim, hipStream_t stream) {
  MP_REQUIRE(dim % 8 == 0, MP_ERR_SHAPE, "mp_moe_fill_dropped_bf16: bad shape");
  const int64_t n = tokens * (dim / 8);
  if (n == 0) return MP_OK;
  hipLaunchKernelGGL(moe_fill_dropped_kernel, dim3((unsigned)mp_cdiv(n, 256)), dim3(256), 0, stream, (const bf16_t*)x, slot, (bf16_t*)out,
                     tokens, dim);
  return mp_check_launch("mp_moe_fill_dropped_bf16");
}

extern "C" int mp_moe_route_top2(const float* gates, const float* logits, const float* noise, int tokens, int n_experts, int capacity,
                                 int* expert, int* slot, float* weight, int* kept_counts, long long* exp_counts, float* l_aux,
                                 hipStream_t stream) {
  MP_REQUIRE(n_experts >= 1 && n_experts <= MAXE && tokens > 0 && capacity >= 0, MP_ERR_SHAPE, "mp_moe_route_top2: bad shape");
  hipLaunchKernelGGL(moe_route_top2_kernel, dim3(1), dim3(1024), 0, stream, gates, logits, noise, tokens, n_experts, capacity, expert,
                     slot, weight, kept_counts, exp_counts, l_aux);
  return mp_check_launch("mp_moe_route_top2");
}

extern "C" int mp_gate_noise_f32(float* out, int64_t n, uint64_t seed, uint64_t offset, int gumbel, hipStream_t stream) {
  MP_REQUIRE(n >= 0, MP_ERR_SHAPE, "mp_gate_noise_f32: bad size");
  if (n == 0) return MP_OK;
  hipLaunchKernelGGL(gate_noise_kernel, dim3((unsigned)mp_cdiv(n, 256)), dim3(256), 0, stream, out, n, (unsigned long long)seed,
                     (unsigned long long)offset, gumbel ? 1 : 0);
  return mp_check_launch("mp_gate_noise_f32");
}

extern "C" int mp_moe_dispatch_bf16(const void* x, int64_t ldx, const int* expert, const int* slot, void* buf, int64_t ldbuf, int64_t tokens, int dim,
                                    int capacity, int top_k, hipStream_t stream) {
  MP_REQUIRE(dim % 8 == 0 && ldx % 8 == 0 && ldbuf % 8 == 0 && ldbuf >= dim && top_k >= 1 && top_k <= 2, MP_ERR_SHAPE, "mp_moe_dispatch_bf16: bad shape");
  const int64_t n = tokens * top_k * (dim / 8);
  if (n == 0) return MP_OK;
  hipLaunchKernelGGL(moe_dispatch_kernel, dim3((unsigned)mp_cdiv(n, 256)), dim3(256), 0, stream, (const bf16_t*)x, ldx, expert, slot,
                     (bf16_t*)buf, ldbuf, tokens, dim, capacity, top_k);
  return mp_check_launch("mp_moe_dispatch_bf16");
}

extern "C" int mp_moe_combine_bf16(const void* y, const int* expert, const int* slot, const float* weight, const void* residual,
                                   void* out, int64_t tokens, int dim, int capacity, int top_k, hipStream_t stream) {
  MP_REQUIRE(dim % 8 == 0 && top_k >= 1 && top_k <= 2, MP_ERR_SHAPE, "mp_moe_combine_bf16: bad shape");
  const int64_t n = tokens * (dim / 8);
  if (n == 0) return MP_OK;
  hipLaunchKernelGGL(moe_combine_kernel, dim3((unsigned)mp_cdiv(n, 256)), dim3(256), 0, stream, (const bf16_t*)y, expert, slot, weight,
                     (const bf16_t*)residual, (bf16_t*)out, tokens, dim, capacity, top_k);
  return mp_check_launch("mp_moe_combine_bf16");
}
