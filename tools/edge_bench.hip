// tools/edge_bench.hip - ablation micro-benchmark of the processor edge-MLP kernel.
// Build + run (GPU box):
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -I lagrangebench_amd/csrc \
//         tools/edge_bench.hip lagrangebench_amd/csrc/lb_api.o lagrangebench_amd/csrc/lb_state.o \
//         lagrangebench_amd/csrc/lb_neighbor.o -o gpurun_out/edge_bench && gpurun_out/edge_bench
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../lagrangebench_amd/csrc/lb_gns.hip"

template <int ABL>
static float run(lb_edge_args a, int iters) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL((k_edge_mlp<true, ABL>), dim3(256), dim3(EDGE_THREADS), 0, 0, a);
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  for (int i = 0; i < iters; ++i)
    hipLaunchKernelGGL((k_edge_mlp<true, ABL>), dim3(256), dim3(EDGE_THREADS), 0, 0, a);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  return 1e3f * ms / iters;
}

int main(int argc, char** argv) {
  const int64_t E = argc > 1 ? atoll(argv[1]) : 1097000, N = argc > 2 ? atoll(argv[2]) : 64000;
  const int iters = 20;
  std::vector<int> s(E), r(E);
  for (int64_t k = 0; k < E; ++k) {
    r[k] = (int)(k * N / E);
    s[k] = (int)((r[k] + (rand() % 400) - 200 + N) % N);
  }
  lb_ctrl c{};
  c.overflow_step = -1;
  c.n_edges_total = (int)E;
  lb_ctrl* dc;
  int *ds, *dr;
  float *elat, *msg, *psr, *w, *vec;
  hipMalloc(&dc, sizeof(c));
  hipMemcpy(dc, &c, sizeof(c), hipMemcpyHostToDevice);
  hipMalloc(&ds, E * 4);
  hipMalloc(&dr, E * 4);
  hipMemcpy(ds, s.data(), E * 4, hipMemcpyHostToDevice);
  hipMemcpy(dr, r.data(), E * 4, hipMemcpyHostToDevice);
  hipMalloc(&elat, E * 512);
  hipMalloc(&msg, E * 512);
  hipMalloc(&psr, N * 1024);
  hipMalloc(&w, 2 * 65536);
  hipMalloc(&vec, 3 * 512);
  std::vector<float> h(E * 128);
  for (auto& x : h) x = (rand() % 2001 - 1000) * 1e-3f;
  hipMemcpy(elat, h.data(), E * 512, hipMemcpyHostToDevice);
  hipMemcpy(psr, h.data(), N * 1024 < E * 512 ? N * 1024 : E * 512, hipMemcpyHostToDevice);
  std::vector<float> hw(32768);
  for (auto& x : hw) x = (rand() % 2001 - 1000) * 1e-4f;
  hipMemcpy(w, hw.data(), 2 * 65536, hipMemcpyHostToDevice);
  hipMemcpy(vec, hw.data(), 3 * 512, hipMemcpyHostToDevice);
  lb_edge_args a{};
  a.ctrl = dc;
  a.senders = ds;
  a.receivers = dr;
  a.elat = elat;
  a.msg = msg;
  a.psr = psr;
  a.w0p = w;
  a.w1p = w + 16384;
  a.b1 = vec;
  a.ln_s = vec + 128;
  a.ln_o = vec + 256;
  const double ideal = (double)((E + 31) / 32) * 512 * 64 / 1024 / 2.4e3;  // us at 2.4 GHz
  printf("E=%lld N=%lld  ideal MFMA-bound %.1f us\n", (long long)E, (long long)N, ideal);
  printf("full                      %8.1f us\n", run<0>(a, iters));
  printf("no gather                 %8.1f us\n", run<1>(a, iters));
  printf("no gather, no e load      %8.1f us\n", run<3>(a, iters));
  printf("no stores                 %8.1f us\n", run<4>(a, iters));
  printf("no loads, no stores       %8.1f us\n", run<7>(a, iters));
  printf("no loads/stores/LN        %8.1f us\n", run<15>(a, iters));
  printf("no LN                     %8.1f us\n", run<8>(a, iters));
  printf("no GEMM2                  %8.1f us\n", run<16>(a, iters));
  printf("no GEMM1, no GEMM2        %8.1f us\n", run<48>(a, iters));
  printf("only GEMMs (15)           %8.1f us\n", run<15>(a, iters));
  return 0;
}
