// Probe (diagnostic): issue rate of v_mfma_f32_4x4x1_16B_f32 (two alternating accumulator chains, as the 4-column step
// kernel issues them) and of v_mfma_f32_16x16x4_f32, one and two waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o tools/mfma_rate_probe tools/mfma_rate_probe.hip && tools/mfma_rate_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int KIND, int CH> __global__ void k(float* out, unsigned long long* cyc, int n) {
  f32x4 acc[CH];
  for (int c = 0; c < CH; ++c) acc[c] = f32x4{0, 0, 0, 0};
  float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < n; ++i) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        if (KIND == 0) acc[c] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[c], 0, 0, 0);
        else acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[c], 0, 0, 0);
      }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0; for (int c = 0; c < CH; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int KIND, int CH> void run(const char* name, int nthreads) {
  float* out; (void)hipMalloc(&out, 4 * 1024 * 256); unsigned long long* cyc; (void)hipMalloc(&cyc, 8);
  const int n = 200;
  unsigned long long best = ~0ull, h;
  for (int r = 0; r < 3; ++r) { k<KIND, CH><<<1, nthreads>>>(out, cyc, n); (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost); if (h < best) best = h; }
  printf("%-28s %d chains, %d waves per SIMD: %.2f ticks per MFMA per wave\n", name, CH, nthreads / 256, (double)best / (n * 16.0 * CH));
  (void)hipFree(out); (void)hipFree(cyc);
}
int main() {
  run<0, 1>("v_mfma_f32_4x4x1_16B_f32", 256); run<0, 2>("v_mfma_f32_4x4x1_16B_f32", 256); run<0, 4>("v_mfma_f32_4x4x1_16B_f32", 256);
  run<0, 2>("v_mfma_f32_4x4x1_16B_f32", 512); run<0, 4>("v_mfma_f32_4x4x1_16B_f32", 512);
  run<1, 1>("v_mfma_f32_16x16x4_f32", 256); run<1, 2>("v_mfma_f32_16x16x4_f32", 256); run<1, 4>("v_mfma_f32_16x16x4_f32", 256);
  run<1, 2>("v_mfma_f32_16x16x4_f32", 512); run<1, 4>("v_mfma_f32_16x16x4_f32", 512);
  return 0;
}
