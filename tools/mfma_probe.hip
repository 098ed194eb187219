// Probe (diagnostic, not shipped): semantics and timing of the f32 MFMA shapes on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void k_layout(const float* A, const float* B, float* D) {
  const int l = threadIdx.x;
  f32x4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f32_4x4x1f32(A[l], B[l], c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) D[l * 4 + r] = c[r];
}
// fused check: c + a*b where a*b rounding matters
__global__ void k_fused(float a, float b, float cin, float* out) {
  f32x4 c = {cin, cin, cin, cin};
  c = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0);
  f32x4 d = {cin, cin, cin, cin};
  d = __builtin_amdgcn_mfma_f32_16x16x4f32(a, threadIdx.x < 16 ? b : 0.f, d, 0, 0, 0);
  if (threadIdx.x == 0) { out[0] = c[0]; out[1] = __builtin_fmaf(a, b, cin); out[2] = a * b + cin; out[3] = d[0]; }
}
template <int MODE> __global__ void k_time(float a, float b, unsigned long long* t, float* sink, int iters) {
  f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0, c4 = c0, c5 = c0, c6 = c0, c7 = c0;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {
    if (MODE == 0) {  // 4x4x1 dependent chain
#pragma unroll
      for (int j = 0; j < 16; ++j) c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c0, 0, 0, 0);
    } else if (MODE == 1) {  // 4x4x1, 8 independent chains
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c3, 0, 0, 0);
        c4 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c4, 0, 0, 0); c5 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c5, 0, 0, 0);
        c6 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c6, 0, 0, 0); c7 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c7, 0, 0, 0);
      }
    } else if (MODE == 2) {  // 16x16x4 dependent chain
#pragma unroll
      for (int j = 0; j < 16; ++j) c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c0, 0, 0, 0);
    } else {  // 16x16x4, 8 independent
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c3, 0, 0, 0);
        c4 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c4, 0, 0, 0); c5 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c5, 0, 0, 0);
        c6 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c6, 0, 0, 0); c7 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c7, 0, 0, 0);
      }
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  f32x4 s = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7;
  sink[threadIdx.x] = s[0] + s[1] + s[2] + s[3];
  if (threadIdx.x == 0) t[0] = t1 - t0;
}
int main() {
  std::vector<float> A(64), B(64), D(256);
  for (int l = 0; l < 64; ++l) { A[l] = 1 + l; B[l] = 1 + 1000 * l; }  // a = lane id, b = 100*lane id
  float *dA, *dB, *dD; hipMalloc(&dA, 256); hipMalloc(&dB, 256); hipMalloc(&dD, 1024);
  hipMemcpy(dA, A.data(), 256, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 256, hipMemcpyHostToDevice);
  k_layout<<<1, 64>>>(dA, dB, dD); hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost);
  // decode: D[l][r] = A[la]*B[lb]: find la, lb
  printf("4x4x1 layout: lane l reg r -> (a-lane, b-lane)\n");
  for (int l : {0, 1, 2, 3, 4, 5, 6, 17, 34, 63}) {
    printf("  lane %2d:", l);
    for (int r = 0; r < 4; ++r) { long v = lround(D[l * 4 + r]); long lb = v / 100; long la = 0; for (int x = 1; x <= 64; ++x) if (x * 100L * ((v / 100) / x) == v && v / 100 / x <= 64 && (v/100)%x==0) {} 
      // brute force
      int fa=-1, fb=-1; for (int a=0;a<64;++a) for (int b=0;b<64;++b) if ((long)(1+a)*(1+1000L*b)==v) {fa=a; fb=b;}
      printf("  r%d=(%d,%d)", r, fa, fb); (void)lb; (void)la; }
    printf("\n");
  }
  float* dout; hipMalloc(&dout, 16); float out[4];
  float a = 1.0f + ldexpf(1, -12), b = 1.0f + ldexpf(1, -12), c = -1.0f;  // a*b = 1 + 2^-11 + 2^-24
  k_fused<<<1, 64>>>(a, b, c, dout); hipMemcpy(out, dout, 16, hipMemcpyDeviceToHost);
  printf("fused check: mfma4x4x1=%a fmaf=%a mul+add=%a mfma16x16x4=%a\n", out[0], out[1], out[2], out[3]);
  unsigned long long* dt; hipMalloc(&dt, 8); float* sink; hipMalloc(&sink, 1024); unsigned long long ht;
  const int iters = 1000;
  k_time<0><<<1, 64>>>(1.f, 1.f, dt, sink, iters); hipMemcpy(&ht, dt, 8, hipMemcpyDeviceToHost); printf("4x4x1 dependent: %.1f cyc/mfma\n", ht / (16.0 * iters));
  k_time<1><<<1, 64>>>(1.f, 1.f, dt, sink, iters); hipMemcpy(&ht, dt, 8, hipMemcpyDeviceToHost); printf("4x4x1 independent: %.1f cyc/mfma\n", ht / (16.0 * iters));
  k_time<2><<<1, 64>>>(1.f, 1.f, dt, sink, iters); hipMemcpy(&ht, dt, 8, hipMemcpyDeviceToHost); printf("16x16x4 dependent: %.1f cyc/mfma\n", ht / (16.0 * iters));
  k_time<3><<<1, 64>>>(1.f, 1.f, dt, sink, iters); hipMemcpy(&ht, dt, 8, hipMemcpyDeviceToHost); printf("16x16x4 independent: %.1f cyc/mfma\n", ht / (16.0 * iters));
  k_time<1><<<1, 128>>>(1.f, 1.f, dt, sink, iters); hipMemcpy(&ht, dt, 8, hipMemcpyDeviceToHost); printf("4x4x1 independent, 2 waves/WG (diff SIMDs): %.1f cyc/mfma per wave\n", ht / (16.0 * iters));
  k_time<1><<<1, 512>>>(1.f, 1.f, dt, sink, iters); hipMemcpy(&ht, dt, 8, hipMemcpyDeviceToHost); printf("4x4x1 independent, 8 waves/WG (2/SIMD): %.1f cyc/mfma per wave\n", ht / (16.0 * iters));
  k_time<3><<<1, 512>>>(1.f, 1.f, dt, sink, iters); hipMemcpy(&ht, dt, 8, hipMemcpyDeviceToHost); printf("16x16x4 independent, 8 waves/WG (2/SIMD): %.1f cyc/mfma per wave\n", ht / (16.0 * iters));
  return 0;
}
