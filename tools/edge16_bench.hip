// tools/edge16_bench.hip - ablation micro-benchmark of k_edge16<PROC, F16> (see tools/edge_bench.hip).
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../lagrangebench_amd/csrc/lb_edge16.hip"

template <bool F16, int ABL>
static float run(lb_edge16_args a, int iters) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((k_edge16<true, F16, ABL>), dim3(256), dim3(E16_THREADS), 0, 0, a);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0, 0);
  for (int i = 0; i < iters; ++i)
    hipLaunchKernelGGL((k_edge16<true, F16, ABL>), dim3(256), dim3(E16_THREADS), 0, 0, a);
  (void)hipEventRecord(e1, 0);
  (void)hipEventSynchronize(e1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  return 1e3f * ms / iters;
}

template <bool F16>
static void sweep(lb_edge16_args a) {
  const int it = 20;
  printf("%s\n", F16 ? "--- f16x2" : "--- f32");
  printf("full                       %8.1f us\n", run<F16, 0>(a, it));
  printf("no scan                    %8.1f us\n", run<F16, 64>(a, it));
  printf("no gather                  %8.1f us\n", run<F16, 1>(a, it));
  printf("no gather, no e load       %8.1f us\n", run<F16, 3>(a, it));
  printf("no stores                  %8.1f us\n", run<F16, 4>(a, it));
  printf("no loads/stores            %8.1f us\n", run<F16, 7>(a, it));
  printf("no loads/stores/scan       %8.1f us\n", run<F16, 71>(a, it));
  printf("no loads/stores/scan/LN    %8.1f us\n", run<F16, 79>(a, it));
  printf("no LN                      %8.1f us\n", run<F16, 8>(a, it));
  printf("no GEMMs                   %8.1f us\n", run<F16, 48>(a, it));
  printf("no GEMMs, no scan          %8.1f us\n", run<F16, 112>(a, it));
  printf("no GEMMs/scan/LN (memory)  %8.1f us\n", run<F16, 120>(a, it));
}

int main(int argc, char** argv) {
  const int64_t E = argc > 1 ? atoll(argv[1]) : 1097000, N = argc > 2 ? atoll(argv[2]) : 64000;
  std::vector<int> s(E), r(E), rp(N + 1);
  for (int64_t k = 0; k < E; ++k) {
    r[k] = (int)(k * N / E);
    s[k] = (int)((r[k] + (rand() % 400) - 200 + N) % N);
  }
  rp[0] = 0;
  {
    int64_t k = 0;
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
  float *elat, *psr, *w, *vec, *agg, *part;
  (void)hipMalloc(&dc, sizeof(c));
  (void)hipMemcpy(dc, &c, sizeof(c), hipMemcpyHostToDevice);
  (void)hipMalloc(&ds, E * 4);
  (void)hipMalloc(&dr, E * 4);
  (void)hipMalloc(&drp, (N + 1) * 4);
  (void)hipMemcpy(ds, s.data(), E * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(dr, r.data(), E * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(drp, rp.data(), (N + 1) * 4, hipMemcpyHostToDevice);
  (void)hipMalloc(&elat, E * 512);
  (void)hipMalloc(&psr, N * 1024);
  (void)hipMalloc(&agg, N * 512);
  (void)hipMalloc(&part, (E / 16 + 2) * 1024);
  (void)hipMalloc(&w, 2 * 65536);
  (void)hipMalloc(&vec, 4 * 512);
  std::vector<float> h(E * 128);
  for (auto& x : h) x = (rand() % 2001 - 1000) * 1e-3f;
  (void)hipMemcpy(elat, h.data(), E * 512, hipMemcpyHostToDevice);
  (void)hipMemcpy(psr, h.data(), N * 1024 < E * 512 ? N * 1024 : E * 512, hipMemcpyHostToDevice);
  std::vector<float> hw(32768);
  for (auto& x : hw) x = (rand() % 2001 - 1000) * 1e-4f;
  std::vector<float> packed(32768);
  lb_pack_weight16h(hw.data(), 128, 128, 128, packed.data(), 128);
  lb_pack_weight16h(hw.data() + 16384, 128, 128, 128, packed.data() + 16384, 128);
  lb_edge16_args a{};
  a.ctrl = dc;
  a.senders = ds;
  a.receivers = dr;
  a.elat = elat;
  a.psr = psr;
  a.w0p = w;
  a.w1p = w + 16384;
  a.b1 = vec;
  a.ln_s = vec + 128;
  a.ln_o = vec + 256;
  a.b0 = vec + 384;
  a.fused = 1;
  a.row_ptr = drp;
  a.agg = agg;
  a.part = part;
  (void)hipMemcpy(vec, hw.data(), 4 * 512, hipMemcpyHostToDevice);
  printf("E=%lld N=%lld  (fp32 MFMA bound %.0f us, f16x2 MFMA bound ~%.0f us, e r/w at 6.3 TB/s %.0f us)\n", (long long)E,
         (long long)N, (double)((E + 15) / 16) * 512 * 32 / 1024 / 2.4e3,
         (double)((E + 15) / 16) * 192 * 16 / 1024 / 2.4e3, E * 1024.0 / 6.3e6);
  (void)hipMemcpy(w, packed.data(), 2 * 65536, hipMemcpyHostToDevice);
  sweep<true>(a);
  lb_pack_weight16(hw.data(), 128, 128, 128, packed.data());
  lb_pack_weight16(hw.data() + 16384, 128, 128, 128, packed.data() + 16384);
  (void)hipMemcpy(w, packed.data(), 2 * 65536, hipMemcpyHostToDevice);
  sweep<false>(a);
  return 0;
}
