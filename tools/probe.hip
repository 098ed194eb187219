// Diagnostic probe (not shipped): builds the library source with -DLRNDE_STAMPS and prints where
// workgroup 0 of k_rhs spends its cycles, plus the shader clock it ran at.
#include "../localregneuralde.jl_amd/csrc/lrnde_kernels.hip"
#include <vector>
#include <cstdio>
int main(int argc, char** argv) {
  int B = argc > 1 ? atoi(argv[1]) : 512;
  lrnde_model_desc d{784, 100, 1, 1};
  lrnde_ctx* c = nullptr;
  if (lrnde_create(&c, &d, 0, nullptr)) return 1;
  size_t np = lrnde_param_count(&d);
  std::vector<float> hp(np);
  for (size_t i = 0; i < np; ++i) hp[i] = 0.05f * (float)((i * 2654435761u) % 1000) / 1000.f - 0.025f;
  float *p, *u, *du;
  hipMalloc(&p, np * 4); hipMalloc(&u, (size_t)B * 784 * 4); hipMalloc(&du, (size_t)B * 784 * 4);
  hipMemcpy(p, hp.data(), np * 4, hipMemcpyHostToDevice);
  std::vector<float> hu((size_t)B * 784, 0.5f);
  hipMemcpy(u, hu.data(), hu.size() * 4, hipMemcpyHostToDevice);
  lrnde_set_params(c, p, np);
  for (int it = 0; it < 200; ++it) lrnde_rhs(c, u, 0.1f, B, du);
  hipDeviceSynchronize();
  unsigned long long st[64];
  hipMemcpyFromSymbol(st, HIP_SYMBOL(g_stamps), sizeof(st));
  const char* names[] = {"entry->xtile", "dense1 mfma", "barrier", "epilogue1", "dense2 mfma", "barrier"};
  double clk = (double)(st[12] - st[0]) / ((double)(st[13] - st[1]) / 100.0);  // MHz
  printf("B=%d  total %llu cycles, %.2f us, shader clock %.0f MHz\n", B, st[12] - st[0], (st[13] - st[1]) / 100.0, clk);
  for (int i = 0; i < 6; ++i) printf("  %-14s %8llu cycles\n", names[i], st[2 * (i + 1)] - st[2 * i]);
  {  // one full step through the bench hook
    float* k1; hipMalloc(&k1, (size_t)B * 784 * 4);
    lrnde_rhs(c, u, 0.f, B, k1);
    float us = 0;
    lrnde_bench_step(c, u, k1, B, 0.f, 0.02f, 1.4e-8f, 1.4e-8f, 50, &us);
    hipDeviceSynchronize();
    hipMemcpyFromSymbol(st, HIP_SYMBOL(g_stamps), sizeof(st));
    const char* sn[] = {"prologue", "x2 combine", "stage k2", "stage k3", "stage k4", "stage k5", "stage k6", "stage k7+err", "block reduce"};
    printf("step kernel: %.1f us/launch; wave-0 phase cycles:\n", us);
    printf("  %-14s %8llu cycles\n", "launch init", st[2 * 10] - st[2 * 9]);
    for (int i = 0; i < 9; ++i) printf("  %-14s %8llu cycles\n", sn[i], st[2 * (11 + i)] - st[2 * (10 + i)]);
    printf("  total          %8llu cycles = %.1f us\n", st[2 * 19] - st[2 * 10], (st[2 * 19 + 1] - st[2 * 10 + 1]) / 100.0);
  }
  {  // the same phases of the last FULL step of a real adaptive solve (controller prologue instead of the bench hook's)
    lrnde_solve_opts o{1.4e-8f, 1.4e-8f, 24, 0, 0, 0};
    lrnde_stats stt;
    float* us; hipMalloc(&us, (size_t)B * 784 * 4 * 2);
    float ts[4]; const float sv[1] = {5.0f};
    lrnde_solve(c, u, B, 0.f, 5.0f, &o, sv, 1, us, ts, 2, &stt, nullptr, 0);  // stops at maxiters: every launch but the trailing ones is a full step
    hipDeviceSynchronize();
    hipMemcpyFromSymbol(st, HIP_SYMBOL(g_stamps), sizeof(st));
    const char* sn[] = {"prologue", "x2 combine", "stage k2", "stage k3", "stage k4", "stage k5", "stage k6", "stage k7+err", "block reduce"};
    printf("step kernel inside a solve (naccept %d nreject %d): wave-0 phase cycles:\n", stt.naccept, stt.nreject);
    printf("  %-14s %8llu cycles\n", "launch init", st[2 * 10] - st[2 * 9]);
    for (int i = 0; i < 9; ++i) printf("  %-14s %8llu cycles\n", sn[i], st[2 * (11 + i)] - st[2 * (10 + i)]);
    printf("  total          %8llu cycles = %.1f us\n", st[2 * 19] - st[2 * 10], (st[2 * 19 + 1] - st[2 * 10 + 1]) / 100.0);
    unsigned long long ps[8];
    hipMemcpyFromSymbol(ps, HIP_SYMBOL(g_pstamps), sizeof(ps));
    printf("  inside the prologue (last launch that ran it; cycles): entry->control block here %llu, ->partials summed %llu, ->eest %llu, ->decision %llu, ->broadcast written %llu;  launch entry -> prologue entry %lld\n",
           ps[1] - ps[0], ps[2] - ps[1], ps[3] - ps[2], ps[4] - ps[3], ps[5] - ps[4], (long long)(ps[0] - st[2 * 9]));
  }
  unsigned long long ws[64];
  hipMemcpyFromSymbol(ws, HIP_SYMBOL(g_wstamps), sizeof(ws));
  printf("  per-wave cycles: dense1 | epilogue1 | dense2   (start offsets vs wave0)\n");
  for (int w = 0; w < 8; ++w)
    printf("   wave %d: start %+6lld  dense1 %6llu  epi1 %6llu  dense2 %6llu\n", w, (long long)(ws[w * 8] - ws[0]),
           ws[w * 8 + 1] - ws[w * 8], ws[w * 8 + 2] - ws[w * 8 + 1], ws[w * 8 + 4] - ws[w * 8 + 3]);
  for (int w = 0; w < 8; ++w) printf("   wave %d: dense2 (last f-eval): chain+finish %llu cyc/tile, epilogue post %llu cyc/tile (%llu tiles)\n", w, ws[w*8+5]/(ws[w*8+7]?ws[w*8+7]:1), ws[w*8+6]/(ws[w*8+7]?ws[w*8+7]:1), ws[w*8+7]);
  return 0;
}
