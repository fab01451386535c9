// Flash-style fused attention forward (bf16 in/out, fp32 softmax statistics) for gfx950.
//
// One kernel covers the three trunk attentions of the reference path:
//   * Llama causal self-attention with key padding (HF-4.31 eager semantics: masked scores get zero weight, softmax in
//     fp32; padded QUERY rows still produce values)                       — medplib_moe_llama.py:127-135, SURVEY A.1
//   * CLIP ViT-L self-attention, no mask, D = 64                              — clip_encoder.py:41-60, SURVEY A.2
//   * SAM-Med2D ViT-B window/global attention with decomposed relative-position bias
//     (scores = (q*scale) k^T + rel_h[q, k_row] + rel_w[q, k_col])            — image_encoder.py:280-296, 381-421
//
// Geometry: 256 threads = 4 waves; a block owns 64 query rows (16 per wave) of one (batch, head); K/V stream through
// LDS in 64-key tiles.  QK^T and PV both run on v_mfma_f32_16x16x32_bf16.  K is staged row-major with XOR-swizzled
// 16-B chunks (conflict-free ds_read_b128 B-fragments); V is staged row-major and consumed through the gfx950
// hardware transpose read ds_read_b64_tr_b16 (VT_SCALAR=true keeps a scalar-transposed staging as a cross-check path).
// The [S,S] score matrix is never materialised (the reference materialises [B,32,S,S] fp32).
#include "common.h"

namespace {

typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;

struct AttnArgs {
  const bf16_t* Q; const bf16_t* K; const bf16_t* V;
  bf16_t* O;
  int64_t q_sb, q_ss;   // element strides: batch, sequence (head stride is D)
  int64_t k_sb, k_ss;
  int64_t v_sb, v_ss;
  int64_t o_sb, o_ss;
  const uint8_t* key_valid;  // [B, Sk] or null
  const float* rel_h;        // [B*H, Sq, kh] or null
  const float* rel_w;        // [B*H, Sq, kw] or null
  int kh, kw;
  int B, H, Sq, Sk;
  int causal;
  float scale;
};

template <int D>
__device__ __forceinline__ int k_off(int r, int c) {  // byte offset of 16-B chunk c of row r in a [64][D] bf16 tile
  return r * (D * 2) + ((c ^ (r & 7)) << 4);
}

template <int D, bool VT_SCALAR>
__global__ __launch_bounds__(256) void attn_fwd_kernel(AttnArgs a) {
  constexpr int KT = 64;            // keys per tile
  constexpr int CH = D / 8;         // 16-B chunks per row
  constexpr int NF = D / 16;        // output fragments along D
  constexpr int KS = D / 32;        // k-steps for QK^T
  // LDS: K tile [64][D] | V tile ([64][D] row-major, or V^T [D][64+8] when VT_SCALAR) | P [4 waves][16][64]
  constexpr int VT_LD = KT + 8;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sK = smem;
  char* sV = smem + KT * D * 2;
  constexpr int V_BYTES = VT_SCALAR ? D * VT_LD * 2 : KT * D * 2;
  bf16_t* sP = reinterpret_cast<bf16_t*>(smem + KT * D * 2 + V_BYTES);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fq = lane >> 4;
  const int bh = blockIdx.y, b = bh / a.H, h = bh % a.H;
  const int q0 = blockIdx.x * 64;
  const int qw0 = q0 + wave * 16;

  const bf16_t* Qb = a.Q + b * a.q_sb + (int64_t)h * D;
  const bf16_t* Kb = a.K + b * a.k_sb + (int64_t)h * D;
  const bf16_t* Vb = a.V + b * a.v_sb + (int64_t)h * D;
  const uint8_t* kv = a.key_valid ? a.key_valid + (int64_t)b * a.Sk : nullptr;

  // Q fragments (A operand): row = qw0 + fr, k = kk*32 + fq*8 .. +8
  bf16x8 qf[KS];
  {
    const int qr = min(qw0 + fr, a.Sq - 1);
#pragma unroll
    for (int kk = 0; kk < KS; ++kk)
      qf[kk] = *reinterpret_cast<const bf16x8*>(Qb + (int64_t)qr * a.q_ss + kk * 32 + fq * 8);
  }

  f32x4 o[NF];
#pragma unroll
  for (int n = 0; n < NF; ++n) o[n] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m_run[4], l_run[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) { m_run[r] = -INFINITY; l_run[r] = 0.f; }

  int n_tiles = (a.Sk + KT - 1) / KT;
  if (a.causal) n_tiles = min(n_tiles, (q0 + 64 + KT - 1) / KT);

  const float* relh = a.rel_h ? a.rel_h + (int64_t)bh * a.Sq * a.kh : nullptr;
  const float* relw = a.rel_w ? a.rel_w + (int64_t)bh * a.Sq * a.kw : nullptr;

  // K/V tiles travel global -> registers -> LDS; the loads for tile t+1 are issued right after tile t has been written
  // to LDS, so their latency is covered by the QK^T / softmax / PV work of tile t.
  constexpr int NCH = (KT * CH) / 256;     // 16-B chunks per thread per operand per tile
  bf16x8 kreg[NCH], vreg[NCH];
  auto gload = [&](int t) {
    const int k0 = t * KT;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int id = tid + i * 256;
      const int r = id / CH, c = id % CH;
      const int kr = min(k0 + r, a.Sk - 1);
      kreg[i] = *reinterpret_cast<const bf16x8*>(Kb + (int64_t)kr * a.k_ss + c * 8);
      vreg[i] = *reinterpret_cast<const bf16x8*>(Vb + (int64_t)kr * a.v_ss + c * 8);
    }
  };
  gload(0);

  for (int t = 0; t < n_tiles; ++t) {
    const int k0 = t * KT;
    __syncthreads();  // previous tile's LDS reads complete
    // ---- stage K and V tiles ----
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int id = tid + i * 256;
      const int r = id / CH, c = id % CH;
      *reinterpret_cast<bf16x8*>(sK + k_off<D>(r, c)) = kreg[i];
      if constexpr (VT_SCALAR) {
        bf16_t* vt = reinterpret_cast<bf16_t*>(sV);
#pragma unroll
        for (int j = 0; j < 8; ++j) vt[(c * 8 + j) * VT_LD + r] = vreg[i][j];
      } else {
        *reinterpret_cast<bf16x8*>(sV + r * (D * 2) + c * 16) = vreg[i];
      }
    }
    __syncthreads();
    if (t + 1 < n_tiles) gload(t + 1);

    // ---- S = Q K^T (16 q rows x 64 keys per wave) ----
    f32x4 s[4];
#pragma unroll
    for (int n = 0; n < 4; ++n) s[n] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) {
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        const bf16x8 kf = *reinterpret_cast<const bf16x8*>(sK + k_off<D>(n * 16 + fr, kk * 4 + fq));
        s[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qf[kk], kf, s[n], 0, 0, 0);
      }
    }

    // ---- scale, bias, mask, online softmax.  S layout: key = n*16 + fr, q row = fq*4 + r ----
    float alpha[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int qi = qw0 + fq * 4 + r;
      const int qc = min(qi, a.Sq - 1);
      float mx = -INFINITY;
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        const int kj = k0 + n * 16 + fr;
        float v = s[n][r] * a.scale;
        if (relh) {
          const int kc = min(kj, a.Sk - 1);
          v += relh[(int64_t)qc * a.kh + kc / a.kw] + relw[(int64_t)qc * a.kw + kc % a.kw];
        }
        bool ok = kj < a.Sk;
        if (a.causal) ok = ok && (kj <= qi);
        if (kv) ok = ok && (kv[min(kj, a.Sk - 1)] != 0);
        v = ok ? v : -INFINITY;
        s[n][r] = v;
        mx = fmaxf(mx, v);
      }
#pragma unroll
      for (int off = 1; off < 16; off <<= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
      const float m_new = fmaxf(m_run[r], mx);
      const float m_safe = (m_new == -INFINITY) ? 0.f : m_new;
      alpha[r] = __expf(m_run[r] - m_safe);
      float rs = 0.f;
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        const float p = __expf(s[n][r] - m_safe);
        s[n][r] = p;
        rs += p;
      }
#pragma unroll
      for (int off = 1; off < 16; off <<= 1) rs += __shfl_xor(rs, off, 64);
      l_run[r] = l_run[r] * alpha[r] + rs;
      m_run[r] = m_new;
    }
#pragma unroll
    for (int n = 0; n < NF; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r) o[n][r] *= alpha[r];

    // ---- P (bf16) -> LDS in A-operand order: sP[wave][row][key] ----
    bf16_t* pw = sP + wave * 16 * KT;
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r) pw[(fq * 4 + r) * KT + n * 16 + fr] = (bf16_t)s[n][r];
    // sP[wave] is written and read by this wave only: DS operations of one wave execute in order, a compiler fence suffices
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();

    // ---- O += P V : A = P[16 x 64 keys], B = V[keys x D] ----
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const bf16x8 pf = *reinterpret_cast<const bf16x8*>(pw + fr * KT + kk * 32 + fq * 8);
#pragma unroll
      for (int n = 0; n < NF; ++n) {
        bf16x8 vf;
        if constexpr (VT_SCALAR) {
          const bf16_t* vt = reinterpret_cast<const bf16_t*>(sV);
          vf = *reinterpret_cast<const bf16x8*>(vt + (n * 16 + fr) * VT_LD + kk * 32 + fq * 8);
        } else {
          // hardware transpose read: 16-lane group fq covers keys kk*32 + fq*8 + {0..7}; lane p=fr supplies the
          // 8-byte row segment (key = base + p/4, cols n*16 + (p%4)*4 ..+3) and receives column fr.
          const int key0 = kk * 32 + fq * 8 + (fr >> 2);
          const int col = n * 16 + (fr & 3) * 4;
          const char* p0 = sV + key0 * (D * 2) + col * 2;
          const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (__attribute__((address_space(3))) s16x4*)(p0));
          const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (__attribute__((address_space(3))) s16x4*)(p0 + 4 * (D * 2)));
          s16x8 both = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
          vf = __builtin_bit_cast(bf16x8, both);
        }
        o[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pf, vf, o[n], 0, 0, 0);
      }
    }
  }

  // ---- normalise and store ----
  bf16_t* Ob = a.O + b * a.o_sb + (int64_t)h * D;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int qi = qw0 + fq * 4 + r;
    if (qi >= a.Sq) continue;
    const float inv = l_run[r] > 0.f ? 1.f / l_run[r] : 0.f;
#pragma unroll
    for (int n = 0; n < NF; ++n) Ob[(int64_t)qi * a.o_ss + n * 16 + fr] = (bf16_t)(o[n][r] * inv);
  }
}

template <int D, bool VT_SCALAR>
int launch_attn(const AttnArgs& a, hipStream_t stream) {
  constexpr int KT = 64, VT_LD = KT + 8;
  constexpr int V_BYTES = VT_SCALAR ? D * VT_LD * 2 : KT * D * 2;
  constexpr int LDS = KT * D * 2 + V_BYTES + 4 * 16 * KT * 2;
  dim3 grid((a.Sq + 63) / 64, a.B * a.H);
  hipLaunchKernelGGL((attn_fwd_kernel<D, VT_SCALAR>), grid, dim3(256), LDS, stream, a);
  return mp_check_launch("mp_attention_fwd_bf16");
}

}  // namespace

extern "C" int mp_attention_fwd_bf16(const void* Q, int64_t q_sb, int64_t q_ss, const void* K, int64_t k_sb, int64_t k_ss,
                                     const void* V, int64_t v_sb, int64_t v_ss, void* O, int64_t o_sb, int64_t o_ss,
                                     const uint8_t* key_valid, const float* rel_h, const float* rel_w, int kh, int kw,
                                     int B, int H, int Sq, int Sk, int D, int causal, float scale, int variant,
                                     hipStream_t stream) {
  MP_REQUIRE(D == 64 || D == 128, MP_ERR_SHAPE, "mp_attention_fwd_bf16: head_dim %d unsupported (64/128)", D);
  MP_REQUIRE(B > 0 && H > 0 && Sq > 0 && Sk > 0, MP_ERR_SHAPE, "mp_attention_fwd_bf16: bad shape");
  MP_REQUIRE((q_ss % 8 == 0) && (k_ss % 8 == 0) && (v_ss % 8 == 0), MP_ERR_SHAPE,
             "mp_attention_fwd_bf16: sequence strides must be multiples of 8 elements");
  MP_REQUIRE((rel_h == nullptr) == (rel_w == nullptr), MP_ERR_ARG, "mp_attention_fwd_bf16: rel_h/rel_w must come together");
  MP_REQUIRE(!rel_h || (kh > 0 && kw > 0 && kh * kw == Sk), MP_ERR_SHAPE, "mp_attention_fwd_bf16: kh*kw must equal Sk");
  AttnArgs a{(const bf16_t*)Q, (const bf16_t*)K, (const bf16_t*)V, (bf16_t*)O, q_sb, q_ss, k_sb, k_ss, v_sb, v_ss,
             o_sb, o_ss, key_valid, rel_h, rel_w, kh, kw, B, H, Sq, Sk, causal, scale};
  if (D == 64) return variant == 1 ? launch_attn<64, true>(a, stream) : launch_attn<64, false>(a, stream);
  return variant == 1 ? launch_attn<128, true>(a, stream) : launch_attn<128, false>(a, stream);
}
