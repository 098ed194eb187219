#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const float* base, int nbytes, int D, int soff, float* out) {
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, nbytes, 0x00020000);
  const int lane = threadIdx.x, n = lane & 15, rq = lane >> 4;
  const int voff = (n * D + rq * 4) * 4;
  f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0));
  const float* p = base + (soff + voff) / 4;
  out[lane * 2] = v.x; out[lane * 2 + 1] = p[0];
}
int main() {
  const int D = 784, n = 64 * D * 10;
  std::vector<float> h(n); for (int i = 0; i < n; ++i) h[i] = (float)i;
  float *d, *o; hipMalloc(&d, n * 4); hipMalloc(&o, 512); hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
  for (int soff : {0, 64, 3072, 64 * D * 4 * 3 + 128}) {
    k<<<1, 64>>>(d, n * 4, D, soff, o); float r[128]; hipMemcpy(r, o, 512, hipMemcpyDeviceToHost);
    int bad = 0; for (int l = 0; l < 64; ++l) if (r[2 * l] != r[2 * l + 1]) { if (bad < 6) printf("  soff %d lane %d: buf %.0f plain %.0f\n", soff, l, r[2 * l], r[2 * l + 1]); ++bad; }
    printf("soff=%d: %d bad lanes\n", soff, bad);
  }
  return 0;
}
