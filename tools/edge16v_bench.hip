// tools/edge16v_bench.hip - micro-benchmark + cross-check of the processor edge kernels:
// k_edge16n (round 1) vs the k_edge16v variants (round 2), on a synthetic receiver-sorted edge list of
// the TGV3D-8k x 8 size (E = 1.097 M, N = 64 k).  Prints us per launch, the HBM roofline fraction for
// the algorithmic bytes E*1032 + N*1536 and the max difference of every variant's outputs from
// k_edge16n's on identical inputs.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize tools/edge16v_bench.hip -o tools/bin/edge16v_bench
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../lagrangebench_amd/csrc/lb_edge16.hip"
#include "../lagrangebench_amd/csrc/lb_edge16v.hip"

thread_local std::string g_lb_err;
int lb_fail(int code, const char*, ...) { return code; }

struct Bufs {
  int64_t E, N;
  float *elat, *elat0, *agg, *part;
  std::vector<float> h_elat, h_agg, h_part;
};

template <typename F>
static float time_it(F launch, int iters) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  launch();
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0, 0);
  for (int i = 0; i < iters; ++i) launch();
  (void)hipEventRecord(e1, 0);
  (void)hipEventSynchronize(e1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  return 1e3f * ms / iters;
}

template <typename F>
static void run_once(Bufs& b, F launch, std::vector<float>& elat, std::vector<float>& agg, std::vector<float>& part) {
  (void)hipMemcpy(b.elat, b.elat0, (b.E + 32) * 512, hipMemcpyDeviceToDevice);
  (void)hipMemset(b.agg, 0, b.N * 512);
  (void)hipMemset(b.part, 0, (b.E / 16 + 2) * 1024);
  launch();
  (void)hipDeviceSynchronize();
  elat.resize(b.E * 128);
  agg.resize(b.N * 128);
  part.resize((b.E / 16 + 2) * 256);
  (void)hipMemcpy(elat.data(), b.elat, b.E * 512, hipMemcpyDeviceToHost);
  (void)hipMemcpy(agg.data(), b.agg, b.N * 512, hipMemcpyDeviceToHost);
  (void)hipMemcpy(part.data(), b.part, (b.E / 16 + 2) * 1024, hipMemcpyDeviceToHost);
}

static double maxdiff(const std::vector<float>& a, const std::vector<float>& b, double* ref_max) {
  double d = 0, m = 0;
  for (size_t i = 0; i < a.size(); ++i) {
    if (std::isnan(a[i]) || std::isnan(b[i])) return 1e30;
    d = std::max(d, (double)std::fabs(a[i] - b[i]));
    m = std::max(m, (double)std::fabs(b[i]));
  }
  *ref_max = m;
  return d;
}

int main(int argc, char** argv) {
  const int64_t E = argc > 1 ? atoll(argv[1]) : 1097000, N = argc > 2 ? atoll(argv[2]) : 64000;
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
  Bufs b{};
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
  (void)hipMalloc(&b.agg, N * 512);
  (void)hipMalloc(&b.part, (E / 16 + 2) * 1024);
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
  const double bytes = (double)E * 1032 + (double)N * 1536;
  printf("E=%lld N=%lld algorithmic bytes %.3f GB (8 TB/s: %.0f us, 6.29 TB/s copy ceiling: %.0f us)\n", (long long)E,
         (long long)N, bytes * 1e-9, bytes / 8e6, bytes / 6.29e6);

  auto base = [&] { hipLaunchKernelGGL(k_edge16n, dim3(256), dim3(E16N_THREADS), 0, 0, a); };
  // bring the clocks up first (the first measurements after idle read ~25 % slow)
  (void)hipMemcpy(b.elat, b.elat0, (E + 32) * 512, hipMemcpyDeviceToDevice);
  for (int i = 0; i < 2000; ++i) base();
  (void)hipDeviceSynchronize();
  std::vector<float> re, ra, rpart;
  run_once(b, base, re, ra, rpart);
  auto report = [&](const char* name, auto launch, bool check) {
    double dmax[3] = {0, 0, 0}, m[3] = {1, 1, 1};
    if (check) {
      std::vector<float> e2, a2, p2;
      run_once(b, launch, e2, a2, p2);
      dmax[0] = maxdiff(e2, re, &m[0]);
      dmax[1] = maxdiff(a2, ra, &m[1]);
      dmax[2] = maxdiff(p2, rpart, &m[2]);
    }
    (void)hipMemcpy(b.elat, b.elat0, (E + 32) * 512, hipMemcpyDeviceToDevice);
    const float us = time_it(launch, iters);
    printf("%-44s %8.1f us  %5.2f TB/s  frac(8TB/s) %.3f", name, us, bytes / us * 1e-6, bytes / us * 1e-6 / 8.0);
    if (check)
      printf("   rel diff vs k_edge16n: elat %.2e agg %.2e part %.2e", dmax[0] / m[0], dmax[1] / m[1], dmax[2] / m[2]);
    printf("\n");
    fflush(stdout);
  };
#define V(W, R, ABL) [&] { hipLaunchKernelGGL((k_edge16v<W, R, false, ABL>), dim3(256), dim3(W * 256), 0, 0, a); }
  report("k_edge16n (round 1, 3 waves/SIMD)", base, true);
  report("k_edge16v<3 waves, resident e>", V(3, false, 0), true);
#define VP(W) [&] { hipLaunchKernelGGL((k_edge16v<W, false, false, 0, true>), dim3(256), dim3(W * 256), 0, 0, a); }
  report("k_edge16v<3 waves, resident e, GEMM priority>", VP(3), true);
  report("k_edge16v<3 waves, resident e> again", V(3, false, 0), true);
  report("k_edge16v<2 waves, resident e>", V(2, false, 0), true);
  report("k_edge16v<2 waves, resident e, GEMM priority>", VP(2), true);
#define LP(W, ABL) [&] { hipLaunchKernelGGL((k_edge16l<W, false, ABL>), dim3(256), dim3(W * 256), 0, 0, a); }
  report("k_edge16l<2 waves, late prefetch, priority>", LP(2, 0), true);
  report("  k_edge16l no stores", LP(2, 4), false);
  report("  k_edge16l compute only", LP(2, 7), false);
  report("  k_edge16l no GEMMs", LP(2, 8), false);
  report("k_edge16v<1 wave, resident e>", V(1, false, 0), true);
  report("k_edge16v<4 waves, reload e>", V(4, true, 0), true);
  report("k_edge16v<3 waves, reload e>", V(3, true, 0), true);
#define P(W, ABL) [&] { hipLaunchKernelGGL((k_edge16p<W, false, ABL>), dim3(256), dim3(W * 256), 0, 0, a); }
  report("k_edge16p<2 waves, full prefetch>", P(2, 0), true);




  printf("--- ablation of k_edge16p<2 waves>\n");
  report("  no loads", P(2, 3), false);
  report("  no stores", P(2, 4), false);
  report("  no loads, no stores (compute only)", P(2, 7), false);
  report("  no GEMMs (memory + VALU)", P(2, 8), false);
  printf("--- ablation of <4 waves, reload e>\n");
  report("  no psr gathers", V(4, true, 1), false);
  report("  no e loads", V(4, true, 2), false);
  report("  no loads", V(4, true, 3), false);
  report("  no stores", V(4, true, 4), false);
  report("  no loads, no stores (compute only)", V(4, true, 7), false);
  report("  no GEMMs (memory + VALU)", V(4, true, 8), false);
  report("  no GEMMs, no loads/stores (VALU + LDS fill)", V(4, true, 15), false);
  printf("--- ablation of <3 waves, resident e>\n");
  report("  no loads, no stores (compute only)", V(3, false, 7), false);
  report("  no GEMMs (memory + VALU)", V(3, false, 8), false);
  report("  no stores", V(3, false, 4), false);
  report("  no loads", V(3, false, 3), false);
  report("  no GEMMs, no loads/stores (VALU + LDS fill)", V(3, false, 15), false);
  report("k_edge16n again", base, false);
  return 0;
}
