// Diagnostic: perform_step with W = 0 and constant biases, so every k_j == b2 and u is analytic.
#include "../localregneuralde.jl_amd/csrc/lrnde_kernels.hip"
#include <vector>
#include <cstdio>
int main(int argc, char** argv) {
  int D = argc > 1 ? atoi(argv[1]) : 16, H = argc > 2 ? atoi(argv[2]) : 16, B = 16;
  lrnde_model_desc d{D, H, 1, 1};
  lrnde_ctx* c = nullptr;
  if (lrnde_create(&c, &d, 0, nullptr)) return 1;
  size_t np = lrnde_param_count(&d);
  std::vector<float> hp(np, 0.f);
  // b2[i] = 1 + i  (last D entries)
  for (int i = 0; i < D; ++i) hp[np - D + i] = 1.0f + i;
  float *p, *u, *k1, *uo, *k7;
  size_t n = (size_t)B * D;
  hipMalloc(&p, np * 4); hipMalloc(&u, n * 4); hipMalloc(&k1, n * 4); hipMalloc(&uo, n * 4); hipMalloc(&k7, n * 4);
  hipMemcpy(p, hp.data(), np * 4, hipMemcpyHostToDevice);
  std::vector<float> hu(n), hk(n);
  for (size_t i = 0; i < n; ++i) { hu[i] = 100.f * (i / D) ; hk[i] = 1.0f + (i % D); }
  hipMemcpy(u, hu.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(k1, hk.data(), n * 4, hipMemcpyHostToDevice);
  lrnde_set_params(c, p, np);
  float ee, re, rs;
  int rc = lrnde_perform_step(c, u, k1, B, 0.f, 1.0f, 1e-3f, 1e-3f, uo, k7, &ee, &re, &rs);
  std::vector<float> ho(n), h7(n);
  hipMemcpy(ho.data(), uo, n * 4, hipMemcpyDeviceToHost); hipMemcpy(h7.data(), k7, n * 4, hipMemcpyDeviceToHost);
  printf("rc=%d eest=%g\n", rc, ee);
  for (int s : {0, 11, 12, 15}) { printf("sample %2d: u-uprev =", s); for (int i = 0; i < 8 && i < D; ++i) printf(" %8.4f", ho[s * D + i] - hu[s * D + i]); printf("   k7 ="); for (int i = 0; i < 8 && i < D; ++i) printf(" %6.3f", h7[s * D + i]); printf("\n"); }
  return 0;
}
