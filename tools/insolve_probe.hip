// Diagnostic (not shipped): the shipped kernels (no stamps) inside a real adaptive solve that is stopped by maxiters, so
// that every launch but the trailing ones is a full step: HIP-event time of the solve / attempted steps, next to the
// timing hook's back-to-back figure.  Build twice to A/B a compile-time switch (e.g. -DLRNDE_NO_PRELOAD).
#include "../localregneuralde.jl_amd/csrc/lrnde_kernels.hip"
#include <vector>
#include <cstdio>
#include <algorithm>
int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 512, NST = argc > 2 ? atoi(argv[2]) : 200;
  lrnde_model_desc d{784, 100, 1, 1};
  lrnde_ctx* c = nullptr;
  if (lrnde_create(&c, &d, 0, nullptr)) return 1;
  size_t np = lrnde_param_count(&d);
  std::vector<float> hp(np);
  for (size_t i = 0; i < np; ++i) hp[i] = 0.05f * (float)((i * 2654435761u) % 1000) / 1000.f - 0.025f;
  float *p, *u, *us, *k1;
  const size_t n = (size_t)B * 784;
  hipMalloc(&p, np * 4); hipMalloc(&u, n * 4); hipMalloc(&us, n * 4 * 2); hipMalloc(&k1, n * 4);
  hipMemcpy(p, hp.data(), np * 4, hipMemcpyHostToDevice);
  std::vector<float> hu(n, 0.5f);
  hipMemcpy(u, hu.data(), n * 4, hipMemcpyHostToDevice);
  lrnde_set_params(c, p, np);
  lrnde_rhs(c, u, 0.f, B, k1);
  std::vector<float> hook, insolve;
  lrnde_last_solve_kernel_ms(c, nullptr, nullptr);  // arms the solve's event pair
  for (int rep = 0; rep < 7; ++rep) {
    float usl = 0;
    lrnde_bench_step(c, u, k1, B, 0.f, 0.02f, 1.4e-8f, 1.4e-8f, 100, &usl);
    hook.push_back(usl);
    lrnde_solve_opts o{1.4e-8f, 1.4e-8f, NST, 0, 0, 0};
    lrnde_stats st; float ts[4]; const float sv[1] = {500.0f};
    lrnde_solve(c, u, B, 0.f, 500.0f, &o, sv, 1, us, ts, 2, &st, nullptr, 0);  // MaxIters after NST attempted steps
    float ms = 0; int launches = 0;
    lrnde_last_solve_kernel_ms(c, &ms, &launches);
    insolve.push_back(ms * 1000.f / (float)(st.naccept + st.nreject));
    if (rep == 0) printf("attempted steps %d (launches %d)\n", st.naccept + st.nreject, launches);
  }
  std::sort(hook.begin(), hook.end()); std::sort(insolve.begin(), insolve.end());
  printf("B=%d  timing hook %.2f us/launch (median of 7; min %.2f)   inside a solve %.2f us/attempted step (median; min %.2f; incl. 2 init + trailing launches over %d steps)\n",
         B, hook[3], hook[0], insolve[3], insolve[0], NST);
  return 0;
}
