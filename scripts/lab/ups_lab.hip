// Lab bench for the fused mask-decoder upsampler (not part of the library): times kernel variants at the 1024-px geometry (64 x 64 tokens,
// batch 8) with rotating buffers, and prints the per-wave phase timeline of the shipped kernel from s_memrealtime stamps (ABL & 4).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off scripts/lab/ups_lab.hip medplib_amd/csrc/capi.cpp -o scripts/lab/ups_lab
#include "../../medplib_amd/csrc/upsampler_fused.hip"
#include <stdio.h>
#include <string.h>
#include <limits.h>
#include <vector>
#include <algorithm>
#include <random>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

static std::vector<uint16_t> rnd_bf16(size_t n, float scale, unsigned seed) {
  std::mt19937 g(seed); std::normal_distribution<float> d(0.f, scale);
  std::vector<uint16_t> v(n);
  for (auto& x : v) { float f = d(g); uint32_t u = __builtin_bit_cast(uint32_t, f); x = (uint16_t)((u + 0x7fff + ((u >> 16) & 1)) >> 16); }
  return v;
}

static int g_early = 4;
int main(int argc, char** argv) {
  if (getenv("UPS_LAB_EARLY")) g_early = atoi(getenv("UPS_LAB_EARLY"));
  const int B = 8, G = argc > 1 ? atoi(argv[1]) : 64, NBUF = 12;
  const size_t tokens = (size_t)B * G * G;
  std::vector<bf16_t*> src(NBUF), up(NBUF);
  for (int i = 0; i < NBUF; ++i) {
    auto h = rnd_bf16(tokens * 256, 1.f, 100 + i);
    CK(hipMalloc(&src[i], tokens * 256 * 2)); CK(hipMemcpy(src[i], h.data(), tokens * 256 * 2, hipMemcpyHostToDevice));
    CK(hipMalloc(&up[i], tokens * 16 * 32 * 2));
  }
  auto hw1 = rnd_bf16(256 * 256, 0.06f, 1), hw2 = rnd_bf16(128 * 64, 0.12f, 2);
  bf16_t *w1, *w2; float *b1, *lw, *lb, *b2;
  CK(hipMalloc(&w1, 256 * 256 * 2)); CK(hipMemcpy(w1, hw1.data(), 256 * 256 * 2, hipMemcpyHostToDevice));
  CK(hipMalloc(&w2, 128 * 64 * 2)); CK(hipMemcpy(w2, hw2.data(), 128 * 64 * 2, hipMemcpyHostToDevice));
  std::vector<float> ones(64, 1.f), zeros(64, 0.f), bias(64, 0.03f);
  CK(hipMalloc(&b1, 256)); CK(hipMemcpy(b1, bias.data(), 256, hipMemcpyHostToDevice));
  CK(hipMalloc(&lw, 256)); CK(hipMemcpy(lw, ones.data(), 256, hipMemcpyHostToDevice));
  CK(hipMalloc(&lb, 256)); CK(hipMemcpy(lb, zeros.data(), 256, hipMemcpyHostToDevice));
  CK(hipMalloc(&b2, 128)); CK(hipMemcpy(b2, bias.data(), 128, hipMemcpyHostToDevice));
  hipStream_t st; CK(hipStreamCreate(&st));
  auto time_it = [&](const char* name, auto fn) {
    for (int i = 0; i < 30; ++i) fn(i);
    CK(hipStreamSynchronize(st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int reps = 600;
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < reps; ++i) fn(i);
    CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / reps, bytes = tokens * 256.0 * 2 + (256 * 256 + 128 * 64) * 2 + tokens * 16.0 * 32 * 2;
    printf("%-44s %7.2f us  %6.1f GB/s  frac %.4f\n", name, us, bytes / us / 1e3, bytes / (us * 1e-6) / 8e12);
  };
  const bool pmc_mode = argc > 2 && !strcmp(argv[2], "pmc");
  if (pmc_mode) {                       // under rocprofv3 --pmc: the shipped kernel only, 40 launches on rotating buffers
    for (int i = 0; i < 40; ++i)
      if (mp_mask_upsample_fused_bf16(src[i % NBUF], w1, b1, lw, lb, w2, b2, nullptr, up[i % NBUF], nullptr, B, G, G, 1e-6f, st)) { printf("%s\n", mp_last_error_string()); exit(1); }
    CK(hipStreamSynchronize(st));
    return 0;
  }
  time_it("library entry (shipped kernel)", [&](int i) {
    if (mp_mask_upsample_fused_bf16(src[i % NBUF], w1, b1, lw, lb, w2, b2, nullptr, up[i % NBUF], nullptr, B, G, G, 1e-6f, st)) { printf("%s\n", mp_last_error_string()); exit(1); }
  });
  // kernel variants launched directly.  split = the round-1..3 arrangement (one row parity per workgroup), both = round 4 (both parities
  // of a group in one workgroup).  ABL bits: 1 no GELU, 2 no stores, 8 non-temporal stores
  const int64_t groups0 = (int64_t)tokens / 16;
  auto direct_split = [&](auto kern, const char* name) {
    const int nw0 = (int)std::min<int64_t>(UP_WAVES, mp_cdiv(groups0, 128));
    const int64_t gsets0 = mp_cdiv(groups0, nw0);
    const int grid0 = 2 * (int)(gsets0 < 128 ? gsets0 : 128);
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, UP_LDS));
    time_it(name, [&](int i) {
      UpArgs a0{src[i % NBUF], w1, b1, lw, lb, w2, b2, nullptr, up[i % NBUF], nullptr, B, G, G, 1e-6f, 0, g_early, nullptr};
      hipLaunchKernelGGL(kern, dim3(grid0), dim3(64 * nw0), UP_LDS, st, a0);
    });
  };
  const char* nwe = getenv("UPS_LAB_NW");
  const int nwB = nwe ? atoi(nwe) : (int)std::min<int64_t>(UP_WAVES, std::max<int64_t>(2, 2 * mp_cdiv(groups0, 256)));
  const int gridB = (int)std::min<int64_t>(256, mp_cdiv(groups0, nwB / 2));
  printf("both: %d waves per workgroup, %d workgroups\n", nwB, gridB);
  auto direct_both = [&](auto kern, const char* name) {
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, UP_LDS2));
    time_it(name, [&](int i) {
      UpArgs a0{src[i % NBUF], w1, b1, lw, lb, w2, b2, nullptr, up[i % NBUF], nullptr, B, G, G, 1e-6f, 0, g_early, nullptr};
      hipLaunchKernelGGL(kern, dim3(gridB), dim3(64 * nwB), UP_LDS2, st, a0);
    });
  };
  direct_split(upsample_fused_kernel<true, false, 0, false>, "split: full kernel (round 3 arrangement)");
  direct_both(upsample_fused_kernel<true, false, 0, true>, "both: one group per wave, 16 waves");
  direct_both(upsample_fused_kernel<true, false, 1048576, true>, "both:   GEMM1 not transposed, LDS transposition");
  direct_both(upsample_fused_kernel<true, false, 33554432, true>, "both:   store addresses computed per lane and store");
  direct_both(upsample_fused_kernel<true, false, 0, true>, "both: one group per wave, 16 waves (again)");
  direct_both(upsample_fused_kernel<true, false, 33554432, true>, "both:   store addresses per lane and store (again)");
  direct_both(upsample_fused_kernel<true, false, 16777216, true>, "both:   GELU as scalar f32 instructions");
  direct_both(upsample_fused_kernel<true, false, 8388608, true>, "both:   non-temporal token loads");
  direct_both(upsample_fused_kernel<true, false, 4194304, true>, "both:   priority 3 - k for the k-th wave of a SIMD");
  direct_both(upsample_fused_kernel<true, false, 4194304 + 2, true>, "both:   same, no stores");
  direct_both(upsample_fused_kernel<true, false, 4096, true>, "both:   plain stores");
  direct_both(upsample_fused_kernel<true, false, 262144, true>, "both:   fragment-major weights in LDS");
  direct_both(upsample_fused_kernel<true, false, 2, true>, "both:   no stores");
  direct_both(upsample_fused_kernel<true, false, 1, true>, "both:   no GELU");
  direct_both(upsample_fused_kernel<true, false, 3, true>, "both:   no GELU, no stores");
  time_it("library entry again (after the direct launches)", [&](int i) {
    if (mp_mask_upsample_fused_bf16(src[i % NBUF], w1, b1, lw, lb, w2, b2, nullptr, up[i % NBUF], nullptr, B, G, G, 1e-6f, st)) { printf("%s\n", mp_last_error_string()); exit(1); }
  });
  direct_both(upsample_fused_kernel<true, false, 0, true>, "both: full kernel, direct, again");
  {   // the same through a captured graph of 24 launches (bench.py's method)
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    for (int i = 0; i < 24; ++i)
      if (mp_mask_upsample_fused_bf16(src[i % NBUF], w1, b1, lw, lb, w2, b2, nullptr, up[i % NBUF], nullptr, B, G, G, 1e-6f, st)) { printf("%s\n", mp_last_error_string()); exit(1); }
    CK(hipStreamEndCapture(st, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int i = 0; i < 5; ++i) CK(hipGraphLaunch(ge, st));
    CK(hipStreamSynchronize(st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < 40; ++i) CK(hipGraphLaunch(ge, st));
    CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-44s %7.2f us\n", "library entry, graph of 24 x 40 replays", ms * 1e3 / (40 * 24));
  }
  // correctness of `both` against `split` on one input (bitwise: same arithmetic, different work split)
  {
    bf16_t *o1, *o2; const size_t ob = tokens * 16 * 32 * 2;
    CK(hipMalloc(&o1, ob)); CK(hipMalloc(&o2, ob)); CK(hipMemset(o1, 0, ob)); CK(hipMemset(o2, 0xff, ob));
    UpArgs a1{src[0], w1, b1, lw, lb, w2, b2, nullptr, o1, nullptr, B, G, G, 1e-6f, 0, g_early, nullptr};
    const int nw0 = (int)std::min<int64_t>(UP_WAVES, mp_cdiv(groups0, 128));
    const int grid0 = 2 * (int)std::min<int64_t>(128, mp_cdiv(groups0, nw0));
    hipLaunchKernelGGL((upsample_fused_kernel<true, false, 0, false>), dim3(grid0), dim3(64 * nw0), UP_LDS, st, a1);
    a1.up = o2;
    CK(hipFuncSetAttribute((const void*)upsample_fused_kernel<true, false, 1048576, true>, hipFuncAttributeMaxDynamicSharedMemorySize, UP_LDS2));
    hipLaunchKernelGGL((upsample_fused_kernel<true, false, 1048576, true>), dim3(gridB), dim3(64 * nwB), UP_LDS2, st, a1);
    CK(hipStreamSynchronize(st));
    std::vector<uint16_t> h1(ob / 2), h2(ob / 2);
    CK(hipMemcpy(h1.data(), o1, ob, hipMemcpyDeviceToHost)); CK(hipMemcpy(h2.data(), o2, ob, hipMemcpyDeviceToHost));
    size_t diff = 0; for (size_t i = 0; i < h1.size(); ++i) diff += h1[i] != h2[i];
    printf("both (untransposed GEMM1) vs split: %zu of %zu output values differ\n", diff, h1.size());
    // T1 sums the LayerNorm statistics in another order: compare by value
    CK(hipMemset(o2, 0xff, ob));
    CK(hipFuncSetAttribute((const void*)upsample_fused_kernel<true, false, 0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, UP_LDS2));
    hipLaunchKernelGGL((upsample_fused_kernel<true, false, 0, true>), dim3(gridB), dim3(64 * nwB), UP_LDS2, st, a1);
    CK(hipStreamSynchronize(st));
    CK(hipMemcpy(h2.data(), o2, ob, hipMemcpyDeviceToHost));
    auto f = [](uint16_t u) { uint32_t w = (uint32_t)u << 16; return __builtin_bit_cast(float, w); };
    double worst = 0, worst_rel = 0; size_t nd = 0, bad = 0;
    for (size_t i = 0; i < h1.size(); ++i) {
      const float x = f(h1[i]), y = f(h2[i]);
      if (!(y == y)) { ++bad; continue; }
      nd += h1[i] != h2[i];
      const double d = fabs((double)x - y);
      worst = std::max(worst, d); worst_rel = std::max(worst_rel, d / (fabs((double)x) + 0.05));
    }
    printf("shipped (GEMM1 transposed) vs split: %zu differ, %zu NaN, worst abs %.4g, worst rel(+0.05) %.4g\n", nd, bad, worst, worst_rel);
  }
  // copy floor: the same bytes through a trivial kernel
  {
    const size_t n16 = tokens * 256 * 2 / 16 * 3 / 2;      // 25.2 MB read + 25.2 MB written = 50.3 MB
    (void)n16;
  }
  auto timeline = [&](auto kern, const char* tname) {
  UpArgs a{src[0], w1, b1, lw, lb, w2, b2, nullptr, up[0], nullptr, B, G, G, 1e-6f, 0, g_early, nullptr};
  const int nw = nwB, grid = gridB;
  const size_t nstamp = (size_t)grid * UP_WAVES * 8 * 2;
  long long* dbg; CK(hipMalloc(&dbg, nstamp * 8)); CK(hipMemset(dbg, 0, nstamp * 8));
  a.dbg = dbg;
  CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, UP_LDS2));
  for (int rep = 0; rep < 3; ++rep) {
    a.src = src[(rep + 3) % NBUF]; a.up = up[(rep + 3) % NBUF];
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * nw), UP_LDS2, st, a);
    CK(hipStreamSynchronize(st));
  }
  std::vector<long long> h(nstamp);
  CK(hipMemcpy(h.data(), dbg, nstamp * 8, hipMemcpyDeviceToHost));
  long long t0 = LLONG_MAX, tend = 0;
  const int W = grid * UP_WAVES;
  for (int w = 0; w < W; ++w) { if (h[w * 8]) t0 = std::min(t0, h[w * 8]); tend = std::max(tend, h[(size_t)(W + w) * 8]); }
  printf("timeline [%s] (us from the first wave's start; s_memrealtime 100 MHz); grid %d x %d waves; kernel span %.2f us\n", tname, grid, nw, (tend - t0) * 0.01);
  const char* names[8] = {"start", "weights landed (own)", "after barrier", "tokens landed", "GEMM1 issued", "LN+GELU1+transp", "line 0 stored", "line 1 stored"};
  for (int i = 0; i < 8; ++i) {
    std::vector<double> v;
    for (int w = 0; w < W; ++w) if (h[w * 8]) v.push_back((h[w * 8 + i] - t0) * 0.01);
    std::sort(v.begin(), v.end());
    printf("  %-22s min %6.2f  p10 %6.2f  median %6.2f  p90 %6.2f  max %6.2f\n", names[i], v.front(), v[v.size() / 10], v[v.size() / 2], v[v.size() * 9 / 10], v.back());
  }
  {
    std::vector<double> v;
    for (int w = 0; w < W; ++w) if (h[w * 8]) v.push_back((h[(size_t)(W + w) * 8] - t0) * 0.01);
    std::sort(v.begin(), v.end());
    printf("  %-22s min %6.2f  p10 %6.2f  median %6.2f  p90 %6.2f  max %6.2f\n", "stores drained", v.front(), v[v.size() / 10], v[v.size() / 2], v[v.size() * 9 / 10], v.back());
  }
  for (int blk : {0}) {
    if (nw > 16) break;
    printf("  workgroup %d, per wave:", blk);
    for (int i = 0; i < 8; ++i) printf(" [%s]", names[i]);
    printf(" [drained]\n");
    for (int w = 0; w < nw; ++w) {
      printf("    wave %2d:", w);
      for (int i = 0; i < 8; ++i) printf(" %6.2f", (h[((size_t)blk * UP_WAVES + w) * 8 + i] - t0) * 0.01);
      printf(" %6.2f\n", (h[(size_t)(W + blk * UP_WAVES + w) * 8] - t0) * 0.01);
    }
  }
    CK(hipFree(dbg));
  };
  timeline(upsample_fused_kernel<true, false, 4, true>, "both: full");
  timeline(upsample_fused_kernel<true, false, 4 + 4194304, true>, "both: priority by wave age");
  timeline(upsample_fused_kernel<true, false, 4 + 524288, true>, "both: full; columns 2, 3 = after K step 0 / K step 3 of GEMM1");

  return 0;
}
