// tools/lin_bench.hip - micro-benchmark + self-check of the training step's tall-skinny fp32-MFMA GEMMs (lb_lin32.h):
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -Ilagrangebench_amd/csrc -Iinclude tools/lin_bench.hip -o tools/bin/lin_bench
//   tools/bin/lin_bench [rows=62000] [node_rows=8000]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "lb_lin32.h"
#include "lin32g_experiment.h"

#define CK(x)                                                                  \
  do {                                                                         \
    hipError_t e_ = (x);                                                       \
    if (e_ != hipSuccess) {                                                    \
      printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__);    \
      exit(1);                                                                 \
    }                                                                          \
  } while (0)

int main(int argc, char** argv) {
  const int64_t rows = argc > 1 ? atoll(argv[1]) : 62000;
  const int64_t nrows = argc > 2 ? atoll(argv[2]) : 8000;
  float *X, *W, *Y, *M, *Wp;
  lb_pack_ent* tab;
  CK(hipMalloc(&X, rows * 256 * 4));
  CK(hipMalloc(&Y, rows * 256 * 4));
  CK(hipMalloc(&M, rows * 256 * 4));
  CK(hipMalloc(&W, 256 * 256 * 4));
  CK(hipMalloc(&Wp, 4 * 256 * 256 * 4));
  CK(hipMalloc(&tab, sizeof(lb_pack_ent)));
  std::vector<float> hx((size_t)rows * 256), hw(256 * 256), hm((size_t)rows * 256);
  srand(5);
  for (auto& v : hx) v = (rand() % 2001 - 1000) * 1e-3f;
  for (auto& v : hm) v = (rand() % 2001 - 1000) * 1e-3f;
  for (auto& v : hw) v = (rand() % 2001 - 1000) * 1e-3f;
  CK(hipMemcpy(X, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(M, hm.data(), hm.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(W, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
  CK(hipFuncSetAttribute((const void*)k_lin32<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
  CK(hipFuncSetAttribute((const void*)k_lin32<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
  CK(hipFuncSetAttribute((const void*)k_lin32f<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
  CK(hipFuncSetAttribute((const void*)k_lin32f<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
  CK(hipFuncSetAttribute((const void*)k_lin32f<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
  CK(hipFuncSetAttribute((const void*)k_lin32f<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
  CK(hipFuncSetAttribute((const void*)k_lin32f<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
  // gather indices (mode 5): two "node" rows per row, nodes = rows / 14
  const int64_t nodes = std::max<int64_t>(rows / 14, 1);
  std::vector<int32_t> hi1(rows), hi2(rows);
  for (int64_t i = 0; i < rows; ++i) { hi1[i] = rand() % nodes; hi2[i] = (int32_t)(i * nodes / rows); }
  int32_t *I1, *I2;
  CK(hipMalloc(&I1, rows * 4));
  CK(hipMalloc(&I2, rows * 4));
  CK(hipMemcpy(I1, hi1.data(), rows * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(I2, hi2.data(), rows * 4, hipMemcpyHostToDevice));
  float* Y2;
  CK(hipMalloc(&Y2, rows * 128 * 4 * 2));
  int bad = 0;
  double err_floor = 1.0;   // the error of an entry is measured against sum |x w| + err_floor
  CK(hipFuncSetAttribute((const void*)k_lin32g, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
  CK(hipFuncSetAttribute((const void*)k_lin32h<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
  CK(hipFuncSetAttribute((const void*)k_lin32h<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
  CK(hipFuncSetAttribute((const void*)k_lin32h<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
  CK(hipFuncSetAttribute((const void*)k_lin32h<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
  CK(hipFuncSetAttribute((const void*)k_lin32h<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
  lb_pack_ent_h* tabh;
  float* wsc;
  CK(hipMalloc(&tabh, sizeof(lb_pack_ent_h)));
  CK(hipMalloc(&wsc, 64));
  // mode: 0 plain, 1 bias + relu, 2 mask, 3 accum, 4 LayerNorm + residual, 5 two gathers + bias + relu (4, 5: fast only);  fast: use k_lin32f
  auto run = [&](const char* name, int NR, int NO, int trans, int mode, int64_t r, int fast) {
    const int nob = NO <= 16 ? 1 : 8, nj = (NR + 15) / 16;
    lb_pack_ent pe{0, 0, NR, NO, trans ? NR : NO, trans, nj, nob};
    CK(hipMemcpy(tab, &pe, sizeof(pe), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_pack_w, dim3(16, 1), dim3(256), 0, 0, W, Wp, tab);
    lb_lin_args a{};
    a.X = X; a.ldx = NR; a.NR = NR; a.Wp = Wp; a.NJ = nj; a.Y = Y; a.ldy = NO; a.NO = NO; a.rows = r;
    if (mode == 1) { a.bias = W; a.relu = 1; }
    if (mode == 2) { a.mask = M; a.ldm = NO; }
    if (mode == 3) a.accum = 1;
    if (mode == 5) { a.bias = W; a.gat1 = M; a.gidx1 = I1; a.gat2 = M + 128; a.gidx2 = I2; }
    if (mode == 4) {
      a.bias = W; a.ln_scale = W + 128; a.ln_offset = W + 256; a.resid = M; a.Yln = Y2; a.Y2 = Y2 + r * 128; a.ln_d = 128;
    }
    if (fast == 2) hipLaunchKernelGGL(k_pack_w32, dim3(16), dim3(256), 0, 0, W, Wp, NR, NO, trans ? NR : NO, trans);
    if (fast == 3) {   // f16x2: the operand as fp16 hi / lo fragments times a power of two
      lb_pack_ent_h ph{0, 0, NR, NO, trans ? NR : NO, trans, NR / 32, nob, 0};
      CK(hipMemcpy(tabh, &ph, sizeof(ph), hipMemcpyHostToDevice));
      hipLaunchKernelGGL(k_pack_wh, dim3(16, 1), dim3(256), 0, 0, W, Wp, wsc, tabh);
      a.wsc = wsc;
    }
    const size_t lds = fast == 2 ? (size_t)(NR / 8) * 4 * 64 * 16 : (size_t)nj * nob * 64 * 16 + (fast ? 96 * 16 : 0);
    const int64_t tiles = (r + 15) / 16;
    const int grid = (int)std::min<int64_t>(fast == 2 ? (r + 31) / 32 : tiles, 256);
    auto go = [&] {
      if (fast == 2) hipLaunchKernelGGL(k_lin32g, dim3(grid), dim3(512), lds, 0, a);
      else if (fast == 3) {
        if (mode == 5) hipLaunchKernelGGL((k_lin32h<4>), dim3(grid), dim3(512), lds, 0, a, a);
        else if (mode == 4) hipLaunchKernelGGL((k_lin32h<3>), dim3(grid), dim3(512), lds, 0, a, a);
        else if (mode == 2) hipLaunchKernelGGL((k_lin32h<1>), dim3(grid), dim3(512), lds, 0, a, a);
        else if (mode == 3) hipLaunchKernelGGL((k_lin32h<2>), dim3(grid), dim3(512), lds, 0, a, a);
        else hipLaunchKernelGGL((k_lin32h<0>), dim3(grid), dim3(512), lds, 0, a, a);
      } else if (fast) {
        if (mode == 5) hipLaunchKernelGGL((k_lin32f<4>), dim3(grid), dim3(512), lds, 0, a);
        else if (mode == 4) hipLaunchKernelGGL((k_lin32f<3>), dim3(grid), dim3(512), lds, 0, a);
        else if (mode == 2) hipLaunchKernelGGL((k_lin32f<1>), dim3(grid), dim3(512), lds, 0, a);
        else if (mode == 3) hipLaunchKernelGGL((k_lin32f<2>), dim3(grid), dim3(512), lds, 0, a);
        else hipLaunchKernelGGL((k_lin32f<0>), dim3(grid), dim3(512), lds, 0, a);
      } else if (nob == 1) hipLaunchKernelGGL((k_lin32<1>), dim3(grid), dim3(512), lds, 0, a);
      else hipLaunchKernelGGL((k_lin32<8>), dim3(grid), dim3(512), lds, 0, a);
    };
    // ---- check (sampled rows, fp64 reference)
    std::vector<float> y0;
    if (mode == 3) {
      CK(hipMemcpy(Y, hm.data(), (size_t)r * NO * 4, hipMemcpyHostToDevice));
    } else {
      CK(hipMemset(Y, 0xff, (size_t)r * NO * 4));
    }
    go();
    CK(hipDeviceSynchronize());
    std::vector<float> hy((size_t)r * NO);
    CK(hipMemcpy(hy.data(), Y, hy.size() * 4, hipMemcpyDeviceToHost));
    double worst = 0;
    std::vector<float> hy1, hy2;
    if (mode == 4) {
      hy1.resize((size_t)r * 128); hy2.resize((size_t)r * 128);
      CK(hipMemcpy(hy1.data(), Y2, hy1.size() * 4, hipMemcpyDeviceToHost));
      CK(hipMemcpy(hy2.data(), Y2 + r * 128, hy2.size() * 4, hipMemcpyDeviceToHost));
    }
    for (int64_t s = 0; s < 400; ++s) {
      const int64_t row = s < 40 ? s : (s < 80 ? r - 1 - (s - 40) : (int64_t)((double)rand() / RAND_MAX * (r - 1)));
      for (int m = 0; m < NO; ++m) {
        double acc = 0, mag = 0;
        for (int k = 0; k < NR; ++k) {
          const double w = trans ? hw[(size_t)m * NR + k] : hw[(size_t)k * NO + m];
          acc += (double)hx[(size_t)row * NR + k] * w;
          mag += std::fabs((double)hx[(size_t)row * NR + k] * w);
        }
        if (mode == 1) acc = std::max(acc + hw[m], 0.0);
        if (mode == 2) acc = hm[(size_t)row * NO + m] > 0.f ? acc : 0.0;
        if (mode == 3) acc += hm[(size_t)row * NO + m];
        if (mode == 5)
          acc = std::max(acc + hm[(size_t)hi1[row] * 128 + m] + hm[128 + (size_t)hi2[row] * 128 + m] + hw[m], 0.0);
        const double err = std::fabs(acc - hy[(size_t)row * NO + m]) / (mag + err_floor);
        worst = std::max(worst, err);
      }
      if (mode == 4) {  // LayerNorm of (z + bias) from the device's z, in fp64
        double x[128], mean = 0, var = 0;
        for (int m = 0; m < 128; ++m) { x[m] = (double)hy[(size_t)row * 128 + m] + hw[m]; mean += x[m]; }
        mean /= 128;
        for (int m = 0; m < 128; ++m) var += (x[m] - mean) * (x[m] - mean);
        var /= 128;
        for (int m = 0; m < 128; ++m) {
          const double y = hw[128 + m] * ((x[m] - mean) / std::sqrt(var + 1e-5)) + hw[256 + m];
          worst = std::max(worst, std::fabs(y - hy1[(size_t)row * 128 + m]) / 8.0);
          worst = std::max(worst, std::fabs(y + hm[(size_t)row * 128 + m] - hy2[(size_t)row * 128 + m]) / 8.0);
        }
      }
    }
    const double tol = err_floor < 1.0 ? 4e-6 : 2e-6;
    if (!(worst < tol)) ++bad;
    for (int i = 0; i < 50; ++i) go();
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, 0));
    const int it = 200;
    for (int i = 0; i < it; ++i) go();
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = 1e3 * ms / it, tf = 2.0 * r * NR * NO / (us * 1e-6) / 1e12;
    printf("%-8s %-40s rows %7lld  %8.2f us  %6.1f TFLOP/s (%.2f of 157)  err %.1e %s\n", fast == 3 ? "lin32h" : fast == 2 ? "lin32g" : (fast ? "lin32f" : "lin32"), name,
           (long long)r, us, tf, tf / 157.3, worst, worst < tol ? "" : "WRONG");
  };
  for (int rep = 0; rep < 2; ++rep) {
    for (int fast = 0; fast < 2; ++fast) {
      run("Y = X W      128 x 128", 128, 128, 0, 0, rows, fast);
      run("Y = X W      128 x 128 + bias + relu", 128, 128, 0, 1, rows, fast);
      run("dX = dY W^T  128 x 128 * mask", 128, 128, 1, 2, rows, fast);
      run("dX += dY W^T 128 x 128", 128, 128, 1, 3, rows, fast);
      if (fast) run("Y = LN(X W + b) + resid  128 x 128", 128, 128, 0, 4, rows, fast);
      if (fast) run("Y = LN(X W + b) + resid  (odd rows)", 128, 128, 0, 4, nrows * 3 + 5, fast);
      if (fast) run("Y = relu(X W + G1[i1] + G2[i2] + b)", 128, 128, 0, 5, rows, fast);
      run("Y = X W      256 x 128 (node sized)", 256, 128, 0, 1, nrows, fast);
      run("Y = X W      128 x 128 (node sized)", 128, 128, 0, 0, nrows, fast);
      run("Y = X W      128 x 128 (odd rows)", 128, 128, 0, 3, nrows * 3 + 5, fast);
    }
    // f16x2 (k_lin32h): same shapes, same fp64 check
    run("Y = X W      128 x 128", 128, 128, 0, 0, rows, 3);
    run("Y = X W      128 x 128 + bias + relu", 128, 128, 0, 1, rows, 3);
    run("dX = dY W^T  128 x 128 * mask", 128, 128, 1, 2, rows, 3);
    run("dX += dY W^T 128 x 128", 128, 128, 1, 3, rows, 3);
    run("Y = LN(X W + b) + resid  128 x 128", 128, 128, 0, 4, rows, 3);
    run("Y = LN(X W + b) + resid  (odd rows)", 128, 128, 0, 4, nrows * 3 + 5, 3);
    run("Y = relu(X W + G1[i1] + G2[i2] + b)", 128, 128, 0, 5, rows, 3);
    run("Y = X W      256 x 128 (node sized)", 256, 128, 0, 1, nrows, 3);
    run("Y = X W      128 x 128 (node sized)", 128, 128, 0, 0, nrows, 3);
    run("Y = X W      128 x 128 (odd rows)", 128, 128, 0, 3, nrows * 3 + 5, 3);
    run("Y = X W      128 x 128 [32x32x2 experiment]", 128, 128, 0, 0, rows, 2);
    run("Y = X W + b, relu [32x32x2 experiment]", 128, 128, 0, 1, rows, 2);
    run("Y = X W      256 x 128 [32x32x2 experiment]", 256, 128, 0, 1, nrows * 3 + 5, 2);
    run("Y = X W      32 x 128", 32, 128, 0, 1, nrows, false);
    run("Y = X W      3 x 128 (decoder back)", 3, 128, 1, 0, nrows, false);
    run("Y = X W      128 x 3 (decoder)", 128, 3, 0, 1, nrows, false);
  }
  // ---- magnitudes (k_lin32h scales every row chunk and every matrix into the fp16 range; the fp32 kernels do not care):
  // rows of X times 10^u, u uniform in [-9, 3], every seventh row zero, W times 1e-6; error relative to sum |x w| alone
  {
    for (int64_t r = 0; r < rows; ++r) {
      const double u = -9.0 + 12.0 * (double)rand() / RAND_MAX;
      const float f = (r % 7 == 3) ? 0.f : (float)std::pow(10.0, u);
      for (int k = 0; k < 256; ++k) hx[(size_t)r * 256 + k] *= f;
    }
    for (auto& v : hw) v *= 1e-6f;
    CK(hipMemcpy(X, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(W, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
    err_floor = 1e-300;
    for (int fast : {1, 3}) {
      run("Y = X W      128 x 128 [rows x 1e-9 .. 1e3, W x 1e-6]", 128, 128, 0, 0, rows, fast);
      run("dX = dY W^T  128 x 128 * mask [same magnitudes]", 128, 128, 1, 2, rows, fast);
      run("Y = X W      256 x 128 [same magnitudes]", 256, 128, 0, 0, nrows, fast);
    }
  }
  printf(bad ? "FAILED: %d wrong results\n" : "all results match the fp64 reference (%d wrong)\n", bad);
  return bad ? 1 : 0;
}
