// Diagnostic (not shipped): library source with -DLRNDE_SBF_STAMPS; the NeuralDSDE layer's forward + pullback at the MNIST-SDE
// shape, then where workgroup 0 of the three pullback kernels spends its time (100-MHz clock).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -DLRNDE_SBF_STAMPS \
//         -I include tools/sde_bwd_probe.hip -o tools/tmp/sde_bwd_probe -L/opt/rocm/lib -lrccl
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
  float *p, *g, *u, *W, *z, *us, *du, *dx, *dpf, *dpg;
  hipMalloc(&p, npd * 4); hipMalloc(&g, npg * 4); hipMalloc(&u, n * 4); hipMalloc(&z, n * 4); hipMalloc(&W, hW.size() * 4);
  hipMalloc(&us, 4 * n * 4); hipMalloc(&du, 4 * n * 4); hipMalloc(&dx, n * 4); hipMalloc(&dpf, npd * 4); hipMalloc(&dpg, npg * 4);
  hipMemcpy(p, hp.data(), npd * 4, hipMemcpyHostToDevice); hipMemcpy(g, hg.data(), npg * 4, hipMemcpyHostToDevice);
  hipMemcpy(u, hu.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(z, hu.data(), n * 4, hipMemcpyHostToDevice);
  hipMemcpy(W, hW.data(), hW.size() * 4, hipMemcpyHostToDevice);
  for (int i = 0; i < 4; ++i) hipMemcpy(du + i * n, hu.data(), n * 4, hipMemcpyHostToDevice);
  if (lrnde_sde_set_params(s, p, npd, g, npg)) return 1;
  lrnde_sde_adapt_opts o{0.14f, 0.14f, 1.0f / 6.0f, 0.0f, 0.9f, 0.2f, 1.125f, 7.0f / 50.0f, 2.0f / 25.0f, 10000};
  lrnde_stats stt;
  float ts[8], reg, t1u; int ns, nf, ng;
  for (int rep = 0; rep < 3; ++rep) {
    int rc = lrnde_sde_node_forward_record(s, u, W, nfine, B, 0.f, 1.f, &o, LRNDE_MODE_UNBIASED, 0.4f, z, -1, nullptr, 0, us, ts, 4, &ns, &reg, &nf, &ng, &stt, &t1u);
    if (rc) { printf("forward rc=%d\n", rc); return 1; }
    rc = lrnde_sde_node_backward_recorded(s, B, du, ns, 2.0f, dx, dpf, dpg);
    if (rc) { printf("backward rc=%d\n", rc); return 1; }
  }
  hipDeviceSynchronize();
  static unsigned long long v[4][8];
  hipMemcpyFromSymbol(v, HIP_SYMBOL(g_sbf_stamps), sizeof(v));
  const char* kn[] = {"k_sde_eh_reg_fused_r", "k_sde_bwd_hist_gemm", "k_sde_eh_bwd_fused_r (sweep)"};
  printf("%d recorded steps; workgroup 0, microseconds: setup | body | finish\n", stt.naccept);
  for (int k = 0; k < 3; ++k)
    printf("  %-30s %7.2f | %7.2f | %7.2f\n", kn[k], (v[k][1] - v[k][0]) / 100.0, (v[k][2] - v[k][1]) / 100.0, (v[k][3] - v[k][2]) / 100.0);
  return 0;
}
