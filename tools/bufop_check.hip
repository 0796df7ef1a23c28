// bufop_check.hip - do raw-buffer sc1 loads / stores through a make_buffer_rsrc descriptor address what we think?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__global__ void k(float* src, float* dst, int n4) {
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(src, 0, 0x7fffffff, 0x00020000);
  __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(dst, 0, 0x7fffffff, 0x00020000);
  for (int64_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(i * 16), 0, 16);
    __builtin_amdgcn_raw_buffer_store_b128(v, rd, (int)(((n4 - 1 - i)) * 16), 0, 16);
  }
}
int main() {
  const int n4 = 1 << 20;
  std::vector<float> h(n4 * 4), o(n4 * 4);
  for (int i = 0; i < n4 * 4; ++i) h[i] = (float)i;
  float *s, *d;
  hipMalloc(&s, n4 * 16);
  hipMalloc(&d, n4 * 16);
  hipMemcpy(s, h.data(), n4 * 16, hipMemcpyHostToDevice);
  hipMemset(d, 0, n4 * 16);
  hipLaunchKernelGGL(k, dim3(256), dim3(256), 0, 0, s, d, n4);
  hipMemcpy(o.data(), d, n4 * 16, hipMemcpyDeviceToHost);
  long bad = 0;
  for (int i = 0; i < n4; ++i)
    for (int j = 0; j < 4; ++j) bad += o[(size_t)(n4 - 1 - i) * 4 + j] != h[(size_t)i * 4 + j];
  printf("bad = %ld of %d\n", bad, n4 * 4);
  return 0;
}
