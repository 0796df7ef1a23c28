// permlane_check.hip - what do gfx950's v_permlane16_swap / v_permlane32_swap and DPP row_ror do to a lane pattern?
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned u2 __attribute__((ext_vector_type(2)));
__global__ void k(int* out) {
  const int l = threadIdx.x;
  unsigned x = l;
  u2 r = __builtin_amdgcn_permlane16_swap(x, x, false, false);
  out[l] = r[0];
  out[64 + l] = r[1];
  u2 s = __builtin_amdgcn_permlane32_swap(x, x, false, false);
  out[128 + l] = s[0];
  out[192 + l] = s[1];
  out[256 + l] = __builtin_amdgcn_update_dpp(0, l, 0x128, 0xF, 0xF, false);
  out[320 + l] = __builtin_amdgcn_update_dpp(0, l, 0x121, 0xF, 0xF, false);
}
int main() {
  int* d;
  hipMalloc(&d, 384 * 4);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  int h[384];
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  const char* names[6] = {"p16.vdst", "p16.src", "p32.vdst", "p32.src", "ror8", "ror1"};
  for (int a = 0; a < 6; ++a) {
    printf("%s:", names[a]);
    for (int l = 0; l < 64; ++l) printf(" %d", h[a * 64 + l]);
    printf("\n");
  }
  return 0;
}
