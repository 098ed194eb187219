// Diagnostic (not shipped): library source with -DLRNDE_STAMPS; runs the layer's forward + continuous adjoint at the MNIST
// shape and prints where workgroup 0 of a stage-5 launch of the adjoint loop (k_vjp_q_pg) spends its cycles.
#include "../localregneuralde.jl_amd/csrc/lrnde_kernels.hip"
#include <vector>
#include <cstdio>
int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 512;
  lrnde_model_desc d{784, 100, 1, 1};
  lrnde_ctx* c = nullptr;
  if (lrnde_create(&c, &d, 0, nullptr)) return 1;
  size_t np = lrnde_param_count(&d);
  std::vector<float> hp(np);
  for (size_t i = 0; i < np; ++i) hp[i] = 0.05f * (float)((i * 2654435761u) % 1000) / 1000.f - 0.025f;
  const size_t n = (size_t)B * 784;
  float *p, *u, *du, *dx, *dp;
  hipMalloc(&p, np * 4); hipMalloc(&u, n * 4); hipMalloc(&du, n * 4); hipMalloc(&dx, n * 4); hipMalloc(&dp, np * 4);
  hipMemcpy(p, hp.data(), np * 4, hipMemcpyHostToDevice);
  std::vector<float> hu(n), hd(n);
  for (size_t i = 0; i < n; ++i) { hu[i] = (float)((i * 40503u) % 997) / 997.f; hd[i] = 1e-6f * (float)((int)((i * 7919u) % 13) - 6); }
  hipMemcpy(u, hu.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(du, hd.data(), n * 4, hipMemcpyHostToDevice);
  lrnde_set_params(c, p, np);
  lrnde_solve_opts o{1.4e-8f, 1.4e-8f, 10000, 0, 0, 0};
  lrnde_stats sf, sb;
  for (int rep = 0; rep < 3; ++rep) {
    int rc = lrnde_node_backward(c, u, B, 0.f, 1.f, &o, LRNDE_MODE_UNBIASED, LRNDE_REG_ERROR_ESTIMATE, 0.4f, du, 2.5f, dx, dp, &sf, &sb);
    if (rc) { printf("rc=%d %s\n", rc, lrnde_last_error(c)); return 1; }
  }
  hipDeviceSynchronize();
  unsigned long long v[16];
  hipMemcpyFromSymbol(v, HIP_SYMBOL(g_vstamps), sizeof(v));
  const char* nm[] = {"entry -> LDS init, stream start", "resolve (control block, barrier)", "operand tiles (record interp, stage lambda)",
                      "phase 1 GEMM (W1 y)", "epilogue 1 (tanh, act')", "phase 2 GEMM (W2^T lam)", "epilogue 2 (dpre)", "phase 3 GEMM (W1^T dpre) + store"};
  printf("forward: %d accepted; adjoint: %d accepted + %d rejected, nf %d\n", sf.naccept, sb.naccept, sb.nreject, sb.nf);
  printf("stage-%d launch of the adjoint loop, workgroup 0 (cycles):\n", LRNDE_STAMP_STAGE);
  for (int i = 0; i < 8; ++i) printf("  %-46s %8llu\n", nm[i], v[i + 1] - v[i]);
  printf("  total %llu cycles\n", v[8] - v[0]);
  // every workgroup of that launch on the 100-MHz clock: the VJP workgroups (B/4) and the parameter-gradient tiles behind them
  static unsigned long long w[1024][2];
  hipMemcpyFromSymbol(w, HIP_SYMBOL(g_wgstamps), sizeof(w));
  const int nv = (B + 3) / 4;
  unsigned long long t0 = ~0ull;
  int nw = 0;
  for (int i = 0; i < 1024; ++i) if (w[i][1]) { if (w[i][0] < t0) t0 = w[i][0]; nw = i + 1; }
  auto stat = [&](int lo, int hi, const char* what) {
    if (hi <= lo) return;
    double s0 = 1e30, s1 = 0, e0 = 1e30, e1 = 0, dsum = 0;
    for (int i = lo; i < hi; ++i) {
      const double st = (w[i][0] - t0) / 100.0, en = (w[i][1] - t0) / 100.0;
      s0 = st < s0 ? st : s0; s1 = st > s1 ? st : s1; e0 = en < e0 ? en : e0; e1 = en > e1 ? en : e1; dsum += en - st;
    }
    printf("  %-22s %4d workgroups: start %.2f .. %.2f us, end %.2f .. %.2f us, mean duration %.2f us\n", what, hi - lo, s0, s1, e0, e1, dsum / (hi - lo));
  };
  printf("workgroups of the stamped launch (stage %d):\n", LRNDE_STAMP_STAGE);
  stat(0, nv < nw ? nv : nw, "VJP");
  stat(nv, nw, "parameter-gradient tiles");
  return 0;
}
