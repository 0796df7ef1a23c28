// tools/edge_ab.hip - A/B micro-benchmark of the product edge kernel k_edge16v (2 waves per SIMD, resident latents,
// GEMM-phase priority, nontemporal streams) on a synthetic receiver-sorted edge list of the TGV3D-8k x 8 size
// (E = 1.097 M, N = 64 k); built twice, with -DLB_GEMM_INTERLEAVE=0 (round 3's block loop: the operand split as one
// 12-instruction block between MFMA phases) and =1 (round 4: one split instruction per MFMA slot):
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize -Ilagrangebench_amd/csrc [-DLB_GEMM_INTERLEAVE=0] tools/edge_ab.hip -o tools/bin/edge_ab_i{0,1}
// Timing: warm clocks (as many untimed launches as timed ones first - after an idle gap the first ~50 ms run 15-20 % slow).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../lagrangebench_amd/csrc/lb_edge16.hip"
#include "../lagrangebench_amd/csrc/lb_edge16v.hip"
#define EW_NO_LAUNCHER
#include "../lagrangebench_amd/csrc/lb_edge16w.hip"

thread_local std::string g_lb_err;
int lb_fail(int code, const char*, ...) { return code; }

template <typename F>
static float time_it(F launch, int iters) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  for (int i = 0; i < 20; ++i) launch();  // (the clocks are warm: main() runs 6 s of launches first)
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0, 0);
  for (int i = 0; i < iters; ++i) launch();
  (void)hipEventRecord(e1, 0);
  (void)hipEventSynchronize(e1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  return 1e3f * ms / iters;
}
int main(int argc, char** argv) {
  const int64_t E = argc > 1 ? atoll(argv[1]) : 1037000, N = argc > 2 ? atoll(argv[2]) : 64000;
  const int iters = argc > 3 ? atoi(argv[3]) : 200;
  std::vector<int> s(E), r(E), rp(N + 1);
  srand(1);
  for (int64_t k = 0; k < E; ++k) {
    r[k] = (int)(k * N / E);
    s[k] = (int)((r[k] + (rand() % 400) - 200 + N) % N);
  }
  {
    int64_t k = 0;
    rp[0] = 0;
    for (int64_t g = 0; g < N; ++g) {
      while (k < E && r[k] == g) ++k;
      rp[g + 1] = (int)k;
    }
  }
  lb_ctrl c{};
  c.overflow_step = -1;
  c.n_edges_total = (int)E;
  lb_ctrl* dc;
  int *ds, *dr, *drp;
  float *psr, *w, *vec;
  struct { int64_t E, N; float *elat, *elat0, *agg, *part; } b{};
  b.E = E;
  b.N = N;
  (void)hipMalloc(&dc, sizeof(c));
  (void)hipMemcpy(dc, &c, sizeof(c), hipMemcpyHostToDevice);
  (void)hipMalloc(&ds, E * 4);
  (void)hipMalloc(&dr, E * 4);
  (void)hipMalloc(&drp, (N + 1) * 4);
  (void)hipMemcpy(ds, s.data(), E * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(dr, r.data(), E * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(drp, rp.data(), (N + 1) * 4, hipMemcpyHostToDevice);
  (void)hipMalloc(&b.elat, (E + 32) * 512);
  (void)hipMalloc(&b.elat0, (E + 32) * 512);
  (void)hipMalloc(&psr, N * 1024);
  const size_t aggpart_bytes = (size_t)N * 512 + (size_t)(E / 16 + 2) * 1024;  // agg | part in one allocation (round 5)
  (void)hipMalloc(&b.agg, aggpart_bytes);
  b.part = b.agg + (size_t)N * 128;
  (void)hipMalloc(&w, 2 * 65536);
  (void)hipMalloc(&vec, 4 * 512);
  {
    std::vector<float> h((size_t)(E + 32) * 128);  // incl. the padding rows of the last (blocked) tile
    for (auto& x : h) x = (rand() % 2001 - 1000) * 1e-3f;
    (void)hipMemcpy(b.elat0, h.data(), (E + 32) * 512, hipMemcpyHostToDevice);
    (void)hipMemcpy(psr, h.data(), N * 1024 < E * 512 ? N * 1024 : E * 512, hipMemcpyHostToDevice);
    std::vector<float> hw(32768), packed(32768), hv(512);
    for (auto& x : hw) x = (rand() % 2001 - 1000) * 1e-4f;
    for (auto& x : hv) x = 0.5f + (rand() % 1001) * 1e-3f;
    // LayerNorm scale 1e-3, offset 0: the latents are updated IN PLACE, e += LayerNorm(..), and with an O(1) increment per
    // launch they leave the fp16 range within a measurement - every wave's range probe then raises the LARGE flag with a
    // same-address atomic and the launch reads 350 us instead of 250 (the flags are printed per line: they must read 0)
    for (int i = 128; i < 256; ++i) hv[i] *= 1e-3f;
    for (int i = 256; i < 384; ++i) hv[i] = 0.f;
    lb_pack_weight16h(hw.data(), 128, 128, 128, packed.data(), 128);
    lb_pack_weight16h(hw.data() + 16384, 128, 128, 128, packed.data() + 16384, 128);
    (void)hipMemcpy(w, packed.data(), 2 * 65536, hipMemcpyHostToDevice);
    (void)hipMemcpy(vec, hv.data(), 4 * 512, hipMemcpyHostToDevice);
  }
  lb_edge16_args a{};
  a.ctrl = dc;
  a.senders = ds;
  a.receivers = dr;
  a.elat = b.elat;
  a.psr = psr;
  a.w0p = w;
  a.w1p = w + 16384;
  a.b1 = vec;
  a.ln_s = vec + 128;
  a.ln_o = vec + 256;
  a.b0 = vec + 384;
  a.fused = 1;
  a.row_ptr = drp;
  a.agg = b.agg;
  a.part = b.part;
  a.aggpart_bytes = (int64_t)aggpart_bytes;
  const double bytes = (double)E * 1032 + (double)N * 1536;
  printf("E=%lld N=%lld algorithmic bytes %.3f GB (8 TB/s: %.0f us, 6.29 TB/s copy ceiling: %.0f us)\n", (long long)E,
         (long long)N, bytes * 1e-9, bytes / 8e6, bytes / 6.29e6);

  (void)hipMemcpy(b.elat, b.elat0, (E + 32) * 512, hipMemcpyDeviceToDevice);
  const char* only = argc > 4 ? argv[4] : nullptr;  // run only the report lines whose name contains this (clock / power sampling)
  auto report = [&](const char* name, auto launch) {
    if (only && !strstr(name, only)) return;
    (void)hipMemcpy(b.elat, b.elat0, (E + 32) * 512, hipMemcpyDeviceToDevice);
    (void)hipMemcpy(dc, &c, sizeof(c), hipMemcpyHostToDevice);  // range-guard flags cleared
    const float us = time_it(launch, iters);
    lb_ctrl cc{};
    (void)hipMemcpy(&cc, dc, sizeof(cc), hipMemcpyDeviceToHost);
    printf("%-64s %8.1f us  %5.2f TB/s  frac(8TB/s) %.3f  guard flags %d\n", name, us, bytes / us * 1e-6, bytes / us * 1e-6 / 8.0,
           cc.math_flags);
    if (cc.math_flags) {  // what tripped the guard: the latents after the run
      static std::vector<float> he;
      he.resize((size_t)E * 128);
      (void)hipMemcpy(he.data(), b.elat, he.size() * 4, hipMemcpyDeviceToHost);
      double mx = 0;
      size_t nan = 0;
      for (float x : he) {
        if (x != x) ++nan; else mx = std::max(mx, (double)fabsf(x));
      }
      printf("      (latents after the run: max |e| %.3e, %zu NaN)\n", mx, nan);
    }
    fflush(stdout);
  };
  printf("LB_GEMM_INTERLEAVE=%d\n", (int)LB_GEMM_INTERLEAVE);
  {  // the GPU takes SECONDS of sustained work after process start to reach its steady clock (the first measurements of a
     // process read 350 us for a kernel that runs 255 us five seconds later): 6 s of the product kernel first
    // (the SKIP variant: it does not store the latents.  The kernel updates them IN PLACE, e += LayerNorm(..): over
    // thousands of launches they outgrow the fp16 range and every wave's range probe then raises the guard flag with a
    // same-address atomic - 350 us per launch instead of 255.  Every measurement below restarts from the same latents.)
    auto k = [&] { hipLaunchKernelGGL((k_edge16v<2, false, true, 0, true, true, 1, false>), dim3(256), dim3(512), 0, 0, a); };
    for (int rep = 0; rep < 20; ++rep) {
      for (int i = 0; i < 1000; ++i) k();
      (void)hipDeviceSynchronize();
    }
  }
  // ---- k_edge16w (round 5: deferred epilogue) must reproduce k_edge16v bit for bit: latents, agg, part
  {
    std::vector<char> ref_e((size_t)(E + 32) * 512), ref_a(aggpart_bytes), got_e(ref_e.size()), got_a(ref_a.size());
    auto run = [&](auto launch, std::vector<char>& oe, std::vector<char>& oa) {
      (void)hipMemcpy(b.elat, b.elat0, (E + 32) * 512, hipMemcpyDeviceToDevice);
      (void)hipMemset(b.agg, 0x5a, aggpart_bytes);
      (void)hipMemcpy(dc, &c, sizeof(c), hipMemcpyHostToDevice);
      launch();
      (void)hipDeviceSynchronize();
      (void)hipMemcpy(oe.data(), b.elat, oe.size(), hipMemcpyDeviceToHost);
      (void)hipMemcpy(oa.data(), b.agg, oa.size(), hipMemcpyDeviceToHost);
    };
    auto fdiff = [&](const char* what, const float* r, const float* g, size_t n, size_t row_floats) {
      double mx = 0, ref_mx = 0;
      size_t nbad = 0, first = (size_t)-1, only_ref = 0, only_got = 0;
      for (size_t i = 0; i < n; ++i) {
        uint32_t ur, ug;
        memcpy(&ur, r + i, 4);
        memcpy(&ug, g + i, 4);
        if (ur == ug) continue;
        if (ur == 0x5a5a5a5au) { ++only_got; if (first == (size_t)-1) first = i; continue; }
        if (ug == 0x5a5a5a5au) { ++only_ref; if (first == (size_t)-1) first = i; continue; }
        ++nbad;
        if (first == (size_t)-1) first = i;
        mx = std::max(mx, (double)fabsf(r[i] - g[i]));
      }
      for (size_t i = 0; i < n; ++i) {
        uint32_t ur;
        memcpy(&ur, r + i, 4);
        if (ur != 0x5a5a5a5au && r[i] == r[i]) ref_mx = std::max(ref_mx, (double)fabsf(r[i]));
      }
      {  // rows with a LARGE difference (a bug, not re-association): which tiles / lanes
        size_t shown = 0, rows_bad = 0, last_row = (size_t)-1;
        size_t hist[40] = {0};
        for (size_t i = 0; i < n; ++i) {
          uint32_t ur, ug;
          memcpy(&ur, r + i, 4);
          memcpy(&ug, g + i, 4);
          if (ur == ug || ur == 0x5a5a5a5au || ug == 0x5a5a5a5au) continue;
          if (fabsf(r[i] - g[i]) < 1e-3f) continue;
          const size_t row = i / row_floats;
          if (row != last_row) {
            ++rows_bad;
            last_row = row;
            if (what[0] == 'p') {  // part[tile][slot]: iteration of the tile inside its wave's walk (XCD-eighths, stride 256)
              const size_t tile = row / 2, ntl = (size_t)((E + 15) / 16);
              const size_t xcd = tile * 8 / ntl, t_lo = ntl * xcd / 8;
              const size_t itn = (tile - t_lo) / 256;
              if (itn < 40) ++hist[itn];
            }
            if (shown < 6) {
              printf("      bad row %zu col %zu ref %.6f got %.6f\n", row, i % row_floats, r[i], g[i]);
              ++shown;
            }
          }
        }
        if (rows_bad) {
          printf("      %zu rows with |diff| > 1e-3;", rows_bad);
          if (what[0] == 'p') {
            printf(" by iteration of the wave's walk:");
            for (int q = 0; q < 40; ++q) printf(" %zu", hist[q]);
          }
          printf("\n");
        }
      }
      printf("    %-8s differing values %zu (max |diff| %.3e of max |ref| %.3e), written only by ref %zu / only by this %zu",
             what, nbad, mx, ref_mx, only_ref, only_got);
      if (first != (size_t)-1) printf(", first at row %zu col %zu", first / row_floats, first % row_floats);
      printf("\n");
    };
    auto check = [&](const char* name, auto launch, bool skip) {
      run(launch, got_e, got_a);
      printf("  check %s\n", name);
      if (!skip) fdiff("latents", (const float*)ref_e.data(), (const float*)got_e.data(), (size_t)E * 128, 128);
      fdiff("agg", (const float*)ref_a.data(), (const float*)got_a.data(), (size_t)N * 128, 128);
      fdiff("part", (const float*)ref_a.data() + (size_t)N * 128, (const float*)got_a.data() + (size_t)N * 128,
            (size_t)(E / 16 + 2) * 256, 128);
    };
#define KW(SKIP, G, DEFER, ABL, WPS, PF) \
  [&] { hipLaunchKernelGGL((k_edge16w<SKIP, true, G, false, DEFER, ABL, WPS, PF>), dim3(256), dim3(WPS * 256), 0, 0, a); }
    run([&] { hipLaunchKernelGGL((k_edge16v<2, false, false, 0, true, true, 1, false>), dim3(256), dim3(512), 0, 0, a); }, ref_e, ref_a);
    check("k_edge16w<deep prefetch, defer nothing>", KW(false, 1, 0, 0, 2, 2), false);
    check("k_edge16w<loads at top, defer scan>", KW(false, 1, 1, 0, 2, 0), false);
    a.reverse = 1;
    check("k_edge16w<deep prefetch, defer nothing>, reverse walk", KW(false, 1, 0, 0, 2, 2), false);
    a.reverse = 0;
    check("k_edge16w<loads at top, defer nothing>", KW(false, 1, 0, 0, 2, 0), false);
    run([&] { hipLaunchKernelGGL((k_edge16v<2, false, true, 0, true, true, 1, false>), dim3(256), dim3(512), 0, 0, a); }, ref_e, ref_a);
    check("k_edge16w<deep prefetch, defer nothing> last layer", KW(true, 1, 0, 0, 2, 2), true);
    check("k_edge16w<loads at top, defer all> last layer", KW(true, 1, 2, 0, 2, 0), true);
  }
  {  // timing: LayerNorm scale and offset 0 - the in-place latents do not evolve from launch to launch (with any other
     // choice they leave the fp16 range within a few hundred launches, the range guard fires in every wave and its
     // same-address atomics add 60 us per launch: the 'guard flags' column must read 0)
    std::vector<float> z(256, 0.f);
    (void)hipMemcpy(vec + 128, z.data(), 1024, hipMemcpyHostToDevice);
  }
  for (int round = 0; round < 2; ++round) {
  report("k_edge16w<loads at top, defer nothing>, contiguous chunk per wave",
         [&] { hipLaunchKernelGGL((k_edge16w<false, true, 1, false, 0, 0, 2, 0, 1>), dim3(256), dim3(512), 0, 0, a); });
  report("k_edge16w<loads at top, defer all> last layer, contiguous chunk per wave",
         [&] { hipLaunchKernelGGL((k_edge16w<true, true, 1, false, 2, 0, 2, 0, 1>), dim3(256), dim3(512), 0, 0, a); });
  report("k_edge16w<deep prefetch, defer nothing>", KW(false, 1, 0, 0, 2, 2));
  report("k_edge16w<deep prefetch, defer nothing, GEMM priority>", KW(false, 1, 0, 2, 2, 2));
  report("k_edge16w<loads at top, defer nothing>", KW(false, 1, 0, 0, 2, 0));
  report("k_edge16w<loads at top, defer scan> storing", KW(false, 1, 1, 0, 2, 0));
  report("k_edge16w<loads at top, defer all> storing", KW(false, 1, 2, 0, 2, 0));
  {
    int flip = 0;
    report("k_edge16w<deep prefetch, defer nothing>, alternating walk direction", [&] {
      a.reverse = (flip++) & 1;
      hipLaunchKernelGGL((k_edge16w<false, true, 1, false, 0, 0, 2, 2>), dim3(256), dim3(512), 0, 0, a);
    });
    a.reverse = 0;
  }
  report("k_edge16w<deep prefetch, defer nothing> last layer", KW(true, 1, 0, 0, 2, 2));
  report("k_edge16w<loads at top, defer all> last layer", KW(true, 1, 2, 0, 2, 0));
#define K(SKIP, ABL, G) [&] { hipLaunchKernelGGL((k_edge16v<2, false, SKIP, ABL, true, true, G, false>), dim3(256), dim3(512), 0, 0, a); }
  report("k_edge16v product (guard rows)", K(false, 0, 1));
  report("k_edge16v product (no guard)", K(false, 0, 0));
  report("k_edge16v last layer (no edge-latent store)", K(true, 0, 1));
  report("  compute only (no loads, no stores)", K(false, 7, 1));
  report("  no GEMMs (memory + VALU)", K(false, 8, 1));
  report("  no LayerNorm / scan", K(false, 16, 1));
  report("  no psr gathers", K(false, 1, 1));
  report("  no edge-latent loads", K(false, 2, 1));
  report("  no stores", K(false, 4, 1));
  report("  no GEMMs, no LayerNorm / scan (memory + little VALU)", K(false, 24, 1));
  report("  no GEMMs, no gathers", K(false, 9, 1));
  {  // out of place: the latents do not evolve from launch to launch
    static float* other = nullptr;
    if (!other) (void)hipMalloc(&other, (E + 32) * 512);
    a.elat_out = other;
    report("k_edge16v product, out of place", K(false, 0, 1));
    a.elat_out = nullptr;
  }
  }
  return 0;
}
