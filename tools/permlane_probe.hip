// v_permlane32_swap semantics of the builtin's two results (gfx950): prints which result has the upper half moved down
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* o) {
  const unsigned p = threadIdx.x;  // lane id
  auto r = __builtin_amdgcn_permlane32_swap(p, p, false, false);
  o[threadIdx.x] = r[0];
  o[threadIdx.x + 64] = r[1];
}
int main() {
  unsigned* d; hipMalloc(&d, 128 * 4);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  unsigned h[128]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  printf("r[0]: lane0=%u lane31=%u lane32=%u lane63=%u\n", h[0], h[31], h[32], h[63]);
  printf("r[1]: lane0=%u lane31=%u lane32=%u lane63=%u\n", h[64], h[95], h[96], h[127]);
  return 0;
}
