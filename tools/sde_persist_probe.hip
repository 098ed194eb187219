// Diagnostic (not shipped): library source with -DLRNDE_SDE_STAMPS; runs the adaptive Euler-Heun solve at the MNIST-SDE shape
// (state 32, hidden 64, B = 512) as the ONE cooperative launch and prints where workgroup 0 spends the cycles of a step.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -DLRNDE_SDE_STAMPS \
//         -I include tools/sde_persist_probe.hip -o tools/tmp/sde_persist_probe -L/opt/rocm/lib -lrccl
#include "../localregneuralde.jl_amd/csrc/lrnde_kernels.hip"
#include <vector>
#include <cstdio>
#include <cmath>
int main() {
  const int D = 32, H = 64, B = 512, nfine = 256;
  lrnde_model_desc d{D, H, 0, LRNDE_ACT_TANH};
  lrnde_sde* s = nullptr;
  if (lrnde_sde_create(&s, &d, 1, 0, nullptr)) return 1;
  const size_t npd = lrnde_param_count(&d), npg = (size_t)D * D + D, n = (size_t)B * D;
  std::vector<float> hp(npd), hg(npg), hu(n), hW((size_t)(nfine + 1) * n, 0.f);
  unsigned st = 12345u;
  auto rnd = [&]() { st = st * 1664525u + 1013904223u; return (float)((st >> 8) & 0xffff) / 65536.f - 0.5f; };
  for (auto& v : hp) v = 0.6f * rnd();
  for (auto& v : hg) v = 0.1f * rnd();
  for (auto& v : hu) v = 2.f * rnd();
  const float sh = sqrtf(1.0f / nfine);
  for (int i = 1; i <= nfine; ++i) for (size_t e = 0; e < n; ++e) hW[(size_t)i * n + e] = hW[(size_t)(i - 1) * n + e] + 3.4f * sh * rnd();
  float *p, *g, *u, *W, *ue;
  hipMalloc(&p, npd * 4); hipMalloc(&g, npg * 4); hipMalloc(&u, n * 4); hipMalloc(&ue, n * 4); hipMalloc(&W, hW.size() * 4);
  hipMemcpy(p, hp.data(), npd * 4, hipMemcpyHostToDevice); hipMemcpy(g, hg.data(), npg * 4, hipMemcpyHostToDevice);
  hipMemcpy(u, hu.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(W, hW.data(), hW.size() * 4, hipMemcpyHostToDevice);
  if (lrnde_sde_set_params(s, p, npd, g, npg)) return 1;
  lrnde_sde_adapt_opts o{0.14f, 0.14f, 1.0f / 6.0f, 0.01f, 0.9f, 0.2f, 1.125f, 7.0f / 50.0f, 2.0f / 25.0f, 10000};
  lrnde_stats stt;
  for (int rep = 0; rep < 3; ++rep) {
    const int rc = lrnde_sde_solve_adaptive(s, u, W, nfine, B, 0.f, 1.f, &o, ue, &stt, nullptr, 0);
    if (rc) { printf("rc=%d\n", rc); return 1; }
  }
  hipDeviceSynchronize();
  static unsigned long long v[32][16];
  hipMemcpyFromSymbol(v, HIP_SYMBOL(g_sde_stamps), sizeof(v));
  printf("%d accepted + %d rejected steps; workgroup 0, cycles per phase (steps 1..%d):\n", stt.naccept, stt.nreject, 12);
  const char* nm[] = {"dW from the path, x tile, barrier", "round 1 (f, g at u)", "round 2 (f, g at tmp)", "round 3 (f at K, g at utilde), norm",
                      "publish + poll (grid barrier)", "controller, barrier", "next step's setup"};
  for (int it = 1; it <= 12 && it + 1 < stt.naccept + stt.nreject; ++it) {
    printf("  step %2d:", it);
    for (int i = 0; i < 6; ++i) printf(" %6llu", v[it][i + 1] - v[it][i]);
    printf(" %6llu | total %llu | round 1: dense1 %llu, diffusion %llu, barrier %llu, dense2 %llu, algebra + put %llu, barrier %llu\n", v[it + 1][0] - v[it][6], v[it + 1][0] - v[it][0],
           v[it][8] - v[it][1], v[it][9] - v[it][8], v[it][10] - v[it][9], v[it][11] - v[it][10], v[it][12] - v[it][11], v[it][2] - v[it][12]);
  }
  printf("columns:"); for (int i = 0; i < 7; ++i) printf(" [%s]", nm[i]); printf("\n");
  return 0;
}
