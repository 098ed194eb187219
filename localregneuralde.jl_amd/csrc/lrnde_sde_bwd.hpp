// lrnde_sde_bwd.hpp — gradient path of the NeuralDSDE layer: the reference differentiates the SDE solve
// w.r.t. (x, ps) with TrackerAdjoint, i.e. by taping the solver's own arithmetic (src/layers/neural_sde.jl:12;
// test/runtests.jl:361-365, 386-397), and `reg_val` w.r.t. ps only (the integrator — uprev, dW, dt — is built under
// CRC.@non_differentiable, neural_sde.jl:42).  Here: the reverse sweep of the fixed-grid Euler-Heun solve with the
// caller's Brownian increments (discretise-then-differentiate, exactly what a tape of src/perform_step.jl:172-191 gives),
// and the reverse sweep of one local Euler-Heun step's EEst*dt (:172-206) w.r.t. the parameters.  The vector-Jacobian
// products are the library's own (lrnde_vjp on the drift and the diffusion context); these are the elementwise pieces.
// Included by lrnde_kernels.hip inside extern "C"-free file scope, after launch_vjp.

namespace {

// the one-launch reverse sweep (lrnde_sde_bwd_fused.hpp; launcher in lrnde_sde_node.hpp): what it walks — K steps given as
// (start index, length) on a grid of interval h, the state at each step's start, and either the Brownian PATH (dW formed as a
// difference) or the array of increments itself (dw_direct: the fixed-grid solve)
struct SdeSweepSrc {
  int K; const int2* im; float h; const float* x; const float* rec_u; const float* W; int dw_direct;
  int nseries; const int* ser_k; const float* ser_theta;   // host arrays: which step a cotangent's state was taken in, and where
};
int sde_sweep_fused_core(lrnde_sde* s, const SdeSweepSrc& r, int B, const float* du_series, float* dx, float* dp_drift, float* dp_diff,
                         bool sync_after = true, struct SdeNodeRecord* reg = nullptr, float w_reg = 0.f, bool* reg_done = nullptr);
bool sde_bwd_fused_ok(const lrnde_sde* s, int nseries);

// forward pieces of a step recomputed for the backward sweep (src/perform_step.jl:175,179,183) and the cotangent seeds
// of its second half:  tmp = (u + dt*du1) + L*dW ;  fb2 = (dt/2) ub ;  gb2 = (dW/2) ub     (ub = cotangent of u_{n+1})
__global__ void k_sdeb_seed(size_t n, const float* u, const float* du1, const float* L, const float* dW, const float* ub, float dt,
                            float* tmp, float* fb2, float* gb2) {
  const float hdt = dt / 2.0f;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float k = u[i] + dt * du1[i];
    tmp[i] = k + L[i] * dW[i];
    fb2[i] = hdt * ub[i];
    gb2[i] = (0.5f * dW[i]) * ub[i];
  }
}
// tmpb = dtmp_f + dtmp_g ;  du1b = (dt/2) ub + dt tmpb ;  Lb = (dW/2) ub + dW tmpb ;  up = ub + tmpb
__global__ void k_sdeb_mid(size_t n, const float* dtf, const float* dtg, const float* dW, const float* ub, float dt, float* du1b,
                           float* Lb, float* up) {
  const float hdt = dt / 2.0f;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float tb = dtf[i] + dtg[i];
    du1b[i] = hdt * ub[i] + dt * tb;
    Lb[i] = (0.5f * dW[i]) * ub[i] + dW[i] * tb;
    up[i] = ub[i] + tb;
  }
}
// ub_n = up + du_f + du_g
__global__ void k_sdeb_end(size_t n, const float* up, const float* duf, const float* dug, float* ub) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) ub[i] = (up[i] + duf[i]) + dug[i];
}
// acc += a + b (parameter cotangents of the two evaluations of one model in a step)
__global__ void k_sdeb_acc(size_t n, float* acc, const float* a, const float* b) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc[i] = acc[i] + (a[i] + b[i]);
}

// ---- local step's regulariser: reverse sweep of EEst*dt (src/perform_step.jl:193-205), uprev / dW / dt constant ----
// forward intermediates -> the seeds.  In: u, unew, du1, du2, L, g3, dW.  Out: du2b = (dt/2) Edb, du1b0 = -(dt/2) Edb,
// g3b = ggpb / sqdt, Lb0 = -ggpb / sqdt, unb = cotangent of u_new through the residual's scale.
__global__ void k_sder_seed(size_t n, const float* u, const float* un, const float* du1, const float* du2, const float* L,
                            const float* g3, const float* dW, float dt, float sqdt, float abstol, float reltol, float delta,
                            float eest, float nf, float* du2b, float* du1b0, float* g3b, float* Lb0, float* unb) {
  const float hdt = dt / 2.0f;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float Ed = (dt * (du2[i] - du1[i])) / 2.0f;
    const float ggp = (g3[i] - L[i]) / sqdt;
    const float w2 = dW[i] * dW[i];
    const float En = (ggp * w2) / 2.0f;
    const float sc = abstol + fmaxf_(__builtin_fabsf(u[i]), __builtin_fabsf(un[i])) * reltol;
    const float num = delta * Ed + En;
    const float r = num / sc;
    const float rb = (eest > 0.f) ? dt * r / (nf * eest) : 0.f;   // reg = dt * sqrt(mean r^2)
    const float numb = rb / sc;
    const float scb = -rb * num / (sc * sc);
    unb[i] = (__builtin_fabsf(un[i]) > __builtin_fabsf(u[i])) ? scb * reltol * (un[i] >= 0.f ? 1.f : -1.f) : 0.f;
    const float Edb = delta * numb, ggpb = numb * w2 * 0.5f;
    du2b[i] = hdt * Edb; du1b0[i] = -hdt * Edb;
    g3b[i] = ggpb / sqdt; Lb0[i] = -ggpb / sqdt;
  }
}
// utilde = u + L*sqdt (:196), K = u + dt*du1 (:175)
__global__ void k_sder_points(size_t n, const float* u, const float* du1, const float* L, float dt, float sqdt, float* ut, float* K) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    ut[i] = u[i] + L[i] * sqdt;
    K[i] = u[i] + dt * du1[i];
  }
}
// after the VJPs at utilde (-> utb), K (-> Kb) and tmp (-> tmpb = dtf + dtg with seeds from unb):
//   du1b = du1b0 + dt Kb + (dt/2) unb + dt tmpb ;  Lb = Lb0 + sqdt utb + (dW/2) unb + dW tmpb
__global__ void k_sder_join(size_t n, const float* du1b0, const float* Lb0, const float* utb, const float* Kb, const float* unb,
                            const float* dtf, const float* dtg, const float* dW, float dt, float sqdt, float* du1b, float* Lb) {
  const float hdt = dt / 2.0f;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float tb = dtf[i] + dtg[i];
    du1b[i] = ((du1b0[i] + dt * Kb[i]) + hdt * unb[i]) + dt * tb;
    Lb[i] = ((Lb0[i] + sqdt * utb[i]) + (0.5f * dW[i]) * unb[i]) + dW[i] * tb;
  }
}
__global__ void k_sdeb_acc1(size_t n, float* acc, const float* a) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc[i] = acc[i] + a[i];
}

// ---- Milstein step (src/perform_step.jl:108-170, diagonal noise, Ito): u' = K + L dW + Dgj J,  K = u + dt du1,
//      tmp = K + sqdt L, Dgj = (g(tmp) - L) / sqdt, J = dW^2/2 - dt/2.  Reverse sweep for a cotangent ub of u' ----
__global__ void k_mil_tmp(size_t n, const float* u, const float* du1, const float* L, float dt, float sqdt, float* tmp) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) tmp[i] = (u[i] + dt * du1[i]) + sqdt * L[i];
}
// gtb = cotangent of g(tmp) = ub J / sqdt ;  Lb0 = ub dW - ub J / sqdt
__global__ void k_mil_seed(size_t n, const float* dW, const float* ub, float dt, float sqdt, float* gtb, float* Lb0) {
  const float hdt = dt / 2.0f;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float J = (0.5f * dW[i]) * dW[i] - hdt;
    const float gb = (ub[i] * J) / sqdt;
    gtb[i] = gb;
    Lb0[i] = ub[i] * dW[i] - gb;
  }
}
// tb = cotangent of tmp (from the VJP of g at tmp): Kb = ub + tb -> up ; du1b = dt Kb ; Lb = Lb0 + sqdt tb
__global__ void k_mil_mid(size_t n, const float* ub, const float* tb, const float* Lb0, float dt, float sqdt, float* du1b, float* Lb, float* up) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float kb = ub[i] + tb[i];
    up[i] = kb;
    du1b[i] = dt * kb;
    Lb[i] = Lb0[i] + sqdt * tb[i];
  }
}
// regulariser of the Milstein step: reg = dt sqrt(mean r^2), r = (u' - u) / (abstol + max(|u|, |u'|) reltol)  (:166-169, the
// 4-argument residual :218-220); cotangent of u' with u constant
__global__ void k_mil_reg_seed(size_t n, const float* u, const float* un, float dt, float abstol, float reltol, float eest, float nf, float* unb) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float sc = abstol + fmaxf_(__builtin_fabsf(u[i]), __builtin_fabsf(un[i])) * reltol;
    const float num = un[i] - u[i];
    const float r = num / sc;
    const float rb = (eest > 0.f) ? dt * r / (nf * eest) : 0.f;
    float b = rb / sc;
    if (__builtin_fabsf(un[i]) > __builtin_fabsf(u[i])) b += (-rb * num / (sc * sc)) * reltol * (un[i] >= 0.f ? 1.f : -1.f);
    unb[i] = b;
  }
}

inline int sde_nb(size_t n) { int nb = (int)((n + 255) / 256); return nb > 1024 ? 1024 : (nb < 1 ? 1 : nb); }

// workspace of the SDE backward entry points: `cnt` state-sized vectors + the parameter-sized ones
int sde_bwd_ws(lrnde_sde* s, size_t n, size_t Pf, size_t Pg2, float** st, int cnt, float** gpf, float** gpg) {
  lrnde_ctx* c = s->drift;
  const size_t need = (size_t)cnt * n + 2 * Pf + 2 * Pg2;
  if (s->bwd_n < need) {
    if (s->bwd_ws) HIPCHK(c, hipFree(s->bwd_ws));
    s->bwd_ws = nullptr; s->bwd_n = 0;
    HIPCHK(c, hipMalloc(&s->bwd_ws, sizeof(float) * need));
    s->bwd_n = need;
  }
  for (int i = 0; i < cnt; ++i) st[i] = s->bwd_ws + (size_t)i * n;
  gpf[0] = s->bwd_ws + (size_t)cnt * n; gpf[1] = gpf[0] + Pf;
  gpg[0] = gpf[1] + Pf; gpg[1] = gpg[0] + Pg2;
  return LRNDE_OK;
}

}  // namespace

extern "C" {

// Pullback of lrnde_sde_solve_fixed (Euler-Heun) for  loss = <du_end, u_traj[nsteps-1]>: dx = d loss / d u0 (B x D),
// dp_drift (flat drift parameters), dp_diff ([vec(Wg); bg]) — all device.  u_traj / dW: what the forward call took and
// returned.  Per step, newest first: two f-evals (du1, L recomputed), four vector-Jacobian products, elementwise joins.
int lrnde_sde_solve_fixed_backward(lrnde_sde* s, const float* u0, const float* u_traj, const float* dW, int32_t B, float t0,
                                   float dt, int32_t nsteps, const float* du_end, float* dx, float* dp_drift, float* dp_diff) {
  int rc = sde_check(s, u0, dW, u_traj, B, dt);
  if (rc) return rc;
  lrnde_ctx* c = s->drift; lrnde_ctx* cg = s->diff;
  if (!du_end || !dx || !dp_drift || !dp_diff || nsteps <= 0) return fail(c, LRNDE_BADARG, "null pointer / nsteps");
  if (cg->stream != c->stream) return fail(c, LRNDE_BADARG, "drift and diffusion contexts must share a stream");
  const int D = c->desc.state_dim;
  const size_t n = (size_t)B * D, Pf = lrnde_param_count(&c->desc), Pg2 = lrnde_param_count(&cg->desc);
  const size_t Pg = (size_t)D * D + (s->diff_bias ? D : 0), goff = (size_t)D * D + D;  // [vec(I); 0] precede [vec(Wg); bg] in the expanded form
  if (sde_bwd_fused_ok(s, 1)) {
    // the whole sweep in one launch: step k = (k, 1) on a grid of interval dt, increments given directly, one cotangent on the
    // end state of the last step
    std::vector<int2> im(nsteps);
    for (int k = 0; k < nsteps; ++k) im[k] = make_int2(k, 1);
    const int sk = nsteps - 1; const float sth = 1.0f;
    SdeSweepSrc src{nsteps, im.data(), dt, u0, u_traj, dW, 1, 1, &sk, &sth};
    return sde_sweep_fused_core(s, src, B, du_end, dx, dp_drift, dp_diff);
  }
  float* v[12]; float* gpf[2]; float* gpg[2];
  if ((rc = sde_bwd_ws(s, n, Pf, Pg2, v, 12, gpf, gpg))) return rc;
  float *du1 = v[0], *L = v[1], *tmp = v[2], *fb2 = v[3], *gb2 = v[4], *dtf = v[5], *dtg = v[6], *du1b = v[7], *Lb = v[8],
        *up = v[9], *duf = v[10], *dug = v[11];
  const int nb = sde_nb(n);
  HIPCHK(c, hipMemcpyAsync(dx, du_end, sizeof(float) * n, hipMemcpyDeviceToDevice, c->stream));  // dx doubles as ub
  HIPCHK(c, hipMemsetAsync(dp_drift, 0, sizeof(float) * Pf, c->stream));
  HIPCHK(c, hipMemsetAsync(dp_diff, 0, sizeof(float) * Pg, c->stream));
  for (int i = nsteps - 1; i >= 0; --i) {
    const float t = t0 + (float)i * dt;
    const float* u = (i == 0) ? u0 : u_traj + (size_t)(i - 1) * n;
    const float* w = dW + (size_t)i * n;
    if ((rc = lrnde_rhs(c, u, t, B, du1))) return rc;
    if ((rc = lrnde_rhs(cg, u, t, B, L))) return rc;
    hipLaunchKernelGGL(k_sdeb_seed, dim3(nb), dim3(256), 0, c->stream, n, u, (const float*)du1, (const float*)L, w, (const float*)dx, dt, tmp, fb2, gb2);
    if ((rc = launch_vjp(c, tmp, nullptr, 0.f, 0.f, t + dt, fb2, B, dtf, gpf[0]))) return rc;
    if ((rc = launch_vjp(cg, tmp, nullptr, 0.f, 0.f, t + dt, gb2, B, dtg, gpg[0]))) return rc;
    hipLaunchKernelGGL(k_sdeb_mid, dim3(nb), dim3(256), 0, c->stream, n, (const float*)dtf, (const float*)dtg, w, (const float*)dx, dt, du1b, Lb, up);
    if ((rc = launch_vjp(c, u, nullptr, 0.f, 0.f, t, du1b, B, duf, gpf[1]))) return rc;
    if ((rc = launch_vjp(cg, u, nullptr, 0.f, 0.f, t, Lb, B, dug, gpg[1]))) return rc;
    hipLaunchKernelGGL(k_sdeb_end, dim3(nb), dim3(256), 0, c->stream, n, (const float*)up, (const float*)duf, (const float*)dug, dx);
    hipLaunchKernelGGL(k_sdeb_acc, dim3(sde_nb(Pf)), dim3(256), 0, c->stream, Pf, dp_drift, (const float*)gpf[0], (const float*)gpf[1]);
    hipLaunchKernelGGL(k_sdeb_acc, dim3(sde_nb(Pg)), dim3(256), 0, c->stream, Pg, dp_diff, (const float*)(gpg[0] + goff), (const float*)(gpg[1] + goff));
    HIPCHK(c, hipGetLastError());
  }
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return LRNDE_OK;
}

// reverse sweep of ONE Milstein step from the cotangent in `ub` (overwritten with the cotangent of u when want_dx), parameter
// cotangents ADDED to dp_drift / dp_diff.  v: 9 state-sized scratch vectors.
static int sde_rkmil_step_sweep(lrnde_sde* s, const float* u, const float* w, int B, float t, float dt, float* ub, float** v,
                                float** gpf, float** gpg, float* dp_drift, float* dp_diff) {
  lrnde_ctx* c = s->drift; lrnde_ctx* cg = s->diff;
  const int D = c->desc.state_dim;
  const size_t n = (size_t)B * D, Pf = lrnde_param_count(&c->desc);
  const size_t Pg = (size_t)D * D + (s->diff_bias ? D : 0), goff = (size_t)D * D + D;
  float *du1 = v[0], *L = v[1], *tmp = v[2], *gtb = v[3], *Lb0 = v[4], *tb = v[5], *du1b = v[6], *Lb = v[7], *up = v[8];
  float *duf = v[3], *dug = v[5];   // gtb / tb are free again by then
  const int nb = sde_nb(n);
  const float sqdt = sqrtf(dt);
  int rc;
  if ((rc = lrnde_rhs(c, u, t, B, du1))) return rc;
  if ((rc = lrnde_rhs(cg, u, t, B, L))) return rc;
  hipLaunchKernelGGL(k_mil_tmp, dim3(nb), dim3(256), 0, c->stream, n, u, (const float*)du1, (const float*)L, dt, sqdt, tmp);
  hipLaunchKernelGGL(k_mil_seed, dim3(nb), dim3(256), 0, c->stream, n, w, (const float*)ub, dt, sqdt, gtb, Lb0);
  if ((rc = launch_vjp(cg, tmp, nullptr, 0.f, 0.f, t, gtb, B, tb, gpg[0]))) return rc;     // g(tmp, p, t): :138
  hipLaunchKernelGGL(k_mil_mid, dim3(nb), dim3(256), 0, c->stream, n, (const float*)ub, (const float*)tb, (const float*)Lb0, dt, sqdt, du1b, Lb, up);
  if ((rc = launch_vjp(c, u, nullptr, 0.f, 0.f, t, du1b, B, duf, gpf[0]))) return rc;
  if ((rc = launch_vjp(cg, u, nullptr, 0.f, 0.f, t, Lb, B, dug, gpg[1]))) return rc;
  hipLaunchKernelGGL(k_sdeb_end, dim3(nb), dim3(256), 0, c->stream, n, (const float*)up, (const float*)duf, (const float*)dug, ub);
  hipLaunchKernelGGL(k_sdeb_acc1, dim3(sde_nb(Pf)), dim3(256), 0, c->stream, Pf, dp_drift, (const float*)gpf[0]);
  hipLaunchKernelGGL(k_sdeb_acc, dim3(sde_nb(Pg)), dim3(256), 0, c->stream, Pg, dp_diff, (const float*)(gpg[0] + goff), (const float*)(gpg[1] + goff));
  HIPCHK(c, hipGetLastError());
  return LRNDE_OK;
}

// Pullback of lrnde_sde_solve_fixed(which = 1, Milstein): the reference tapes whatever n.solver is
// (src/layers/neural_sde.jl:12,68-69); same contract as lrnde_sde_solve_fixed_backward.
int lrnde_sde_solve_fixed_backward_rkmil(lrnde_sde* s, const float* u0, const float* u_traj, const float* dW, int32_t B, float t0,
                                         float dt, int32_t nsteps, const float* du_end, float* dx, float* dp_drift, float* dp_diff) {
  int rc = sde_check(s, u0, dW, u_traj, B, dt);
  if (rc) return rc;
  lrnde_ctx* c = s->drift; lrnde_ctx* cg = s->diff;
  if (!du_end || !dx || !dp_drift || !dp_diff || nsteps <= 0) return fail(c, LRNDE_BADARG, "null pointer / nsteps");
  if (cg->stream != c->stream) return fail(c, LRNDE_BADARG, "drift and diffusion contexts must share a stream");
  const int D = c->desc.state_dim;
  const size_t n = (size_t)B * D, Pf = lrnde_param_count(&c->desc), Pg2 = lrnde_param_count(&cg->desc);
  const size_t Pg = (size_t)D * D + (s->diff_bias ? D : 0);
  float* v[9]; float* gpf[2]; float* gpg[2];
  if ((rc = sde_bwd_ws(s, n, Pf, Pg2, v, 9, gpf, gpg))) return rc;
  HIPCHK(c, hipMemcpyAsync(dx, du_end, sizeof(float) * n, hipMemcpyDeviceToDevice, c->stream));
  HIPCHK(c, hipMemsetAsync(dp_drift, 0, sizeof(float) * Pf, c->stream));
  HIPCHK(c, hipMemsetAsync(dp_diff, 0, sizeof(float) * Pg, c->stream));
  for (int i = nsteps - 1; i >= 0; --i) {
    const float* u = (i == 0) ? u0 : u_traj + (size_t)(i - 1) * n;
    if ((rc = sde_rkmil_step_sweep(s, u, dW + (size_t)i * n, B, t0 + (float)i * dt, dt, dx, v, gpf, gpg, dp_drift, dp_diff))) return rc;
  }
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return LRNDE_OK;
}

// d (EEst*dt) / d (p_drift, p_diffusion) of one local Milstein step (src/perform_step.jl:108-170) with uprev, dW, dt constant
int lrnde_sde_rkmil_reg_grad(lrnde_sde* s, const float* uprev, const float* dW, int32_t B, float t, float dt, float abstol,
                             float reltol, float* dp_drift, float* dp_diff, float* reg_val_host) {
  lrnde_ctx* c0 = s ? s->drift : nullptr;
  if (!s || !dp_drift || !dp_diff) return c0 ? fail(c0, LRNDE_BADARG, "null pointer") : LRNDE_BADARG;
  int rc = sde_check(s, uprev, dW, dp_drift, B, dt);
  if (rc) return rc;
  lrnde_ctx* c = s->drift; lrnde_ctx* cg = s->diff;
  if (cg->stream != c->stream) return fail(c, LRNDE_BADARG, "drift and diffusion contexts must share a stream");
  const int D = c->desc.state_dim;
  const size_t n = (size_t)B * D, Pf = lrnde_param_count(&c->desc), Pg2 = lrnde_param_count(&cg->desc);
  const size_t Pg = (size_t)D * D + (s->diff_bias ? D : 0);
  float* v[11]; float* gpf[2]; float* gpg[2];
  if ((rc = sde_bwd_ws(s, n, Pf, Pg2, v, 11, gpf, gpg))) return rc;
  float *un = v[9], *unb = v[10];
  float ee = 0.f, rv = 0.f;
  if ((rc = sde_step_impl(s, 1, uprev, dW, B, t, dt, abstol, reltol, 0.f, un, &ee, &rv))) return rc;
  if (reg_val_host) *reg_val_host = rv;
  hipLaunchKernelGGL(k_mil_reg_seed, dim3(sde_nb(n)), dim3(256), 0, c->stream, n, uprev, (const float*)un, dt, abstol, reltol, ee, (float)n, unb);
  HIPCHK(c, hipMemsetAsync(dp_drift, 0, sizeof(float) * Pf, c->stream));
  HIPCHK(c, hipMemsetAsync(dp_diff, 0, sizeof(float) * Pg, c->stream));
  if ((rc = sde_rkmil_step_sweep(s, uprev, dW, B, t, dt, unb, v, gpf, gpg, dp_drift, dp_diff))) return rc;   // (the cotangent of uprev is discarded: a constant)
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return LRNDE_OK;
}

// d (EEst*dt) / d (p_drift, p_diffusion) of one local Euler-Heun step (`_perform_step(integrator, ::LambaEulerHeunConstantCache, p)`,
// src/perform_step.jl:172-206) with uprev, dW, dt constant — the regulariser's gradient; there is none w.r.t. uprev
// (neural_sde.jl:42; test/runtests.jl:392: `gs_x === nothing`).  Six vector-Jacobian products.
int lrnde_sde_euler_heun_reg_grad(lrnde_sde* s, const float* uprev, const float* dW, int32_t B, float t, float dt, float abstol,
                                  float reltol, float delta, float* dp_drift, float* dp_diff, float* reg_val_host) {
  lrnde_ctx* c0 = s ? s->drift : nullptr;
  if (!s || !dp_drift || !dp_diff) return c0 ? fail(c0, LRNDE_BADARG, "null pointer") : LRNDE_BADARG;
  lrnde_ctx* c = s->drift; lrnde_ctx* cg = s->diff;
  const int D = c->desc.state_dim;
  const size_t n = (size_t)B * D, Pf = lrnde_param_count(&c->desc), Pg2 = lrnde_param_count(&cg->desc);
  const size_t Pg = (size_t)D * D + (s->diff_bias ? D : 0), goff = (size_t)D * D + D;
  float* v[20]; float* gpf[2]; float* gpg[2];
  int rc = sde_check(s, uprev, dW, dp_drift, B, dt);
  if (rc) return rc;
  if (cg->stream != c->stream) return fail(c, LRNDE_BADARG, "drift and diffusion contexts must share a stream");
  if ((rc = sde_bwd_ws(s, n, Pf, Pg2, v, 20, gpf, gpg))) return rc;
  float *un = v[0], *du1 = v[1], *L = v[2], *ut = v[3], *K = v[4], *du2 = v[5], *g3 = v[6], *du2b = v[7], *du1b0 = v[8], *g3b = v[9],
        *Lb0 = v[10], *unb = v[11], *utb = v[12], *Kb = v[13], *tmp = v[14], *fb2 = v[15], *gb2 = v[16], *dtf = v[17], *dtg = v[18],
        *du1b = v[19];
  float* Lb = tmp;  // tmp is free once its two VJPs are enqueued (stream order)
  // the step itself for EEst (and u_new); bit-identical to lrnde_sde_euler_heun_step
  float ee = 0.f, rv = 0.f;
  if ((rc = sde_step_impl(s, 0, uprev, dW, B, t, dt, abstol, reltol, delta, un, &ee, &rv))) return rc;
  if (reg_val_host) *reg_val_host = rv;
  const float sqdt = sqrtf(dt);
  const int nb = sde_nb(n);
  if ((rc = lrnde_rhs(c, uprev, t, B, du1))) return rc;
  if ((rc = lrnde_rhs(cg, uprev, t, B, L))) return rc;
  hipLaunchKernelGGL(k_sder_points, dim3(nb), dim3(256), 0, c->stream, n, uprev, (const float*)du1, (const float*)L, dt, sqdt, ut, K);
  if ((rc = lrnde_rhs(c, K, t + dt, B, du2))) return rc;
  if ((rc = lrnde_rhs(cg, ut, t, B, g3))) return rc;
  hipLaunchKernelGGL(k_sder_seed, dim3(nb), dim3(256), 0, c->stream, n, uprev, (const float*)un, (const float*)du1, (const float*)du2,
                     (const float*)L, (const float*)g3, dW, dt, sqdt, abstol, reltol, delta, ee, (float)n, du2b, du1b0, g3b, Lb0, unb);
  HIPCHK(c, hipMemsetAsync(dp_drift, 0, sizeof(float) * Pf, c->stream));
  HIPCHK(c, hipMemsetAsync(dp_diff, 0, sizeof(float) * Pg, c->stream));
  // g at utilde, f at K
  if ((rc = launch_vjp(cg, ut, nullptr, 0.f, 0.f, t, g3b, B, utb, gpg[0]))) return rc;
  if ((rc = launch_vjp(c, K, nullptr, 0.f, 0.f, t + dt, du2b, B, Kb, gpf[0]))) return rc;
  // u_new's cotangent through f, g at tmp (seeds as in the solve's sweep with ub := unb)
  hipLaunchKernelGGL(k_sdeb_seed, dim3(nb), dim3(256), 0, c->stream, n, uprev, (const float*)du1, (const float*)L, dW, (const float*)unb, dt, tmp, fb2, gb2);
  if ((rc = launch_vjp(c, tmp, nullptr, 0.f, 0.f, t + dt, fb2, B, dtf, gpf[1]))) return rc;
  if ((rc = launch_vjp(cg, tmp, nullptr, 0.f, 0.f, t + dt, gb2, B, dtg, gpg[1]))) return rc;
  hipLaunchKernelGGL(k_sdeb_acc, dim3(sde_nb(Pf)), dim3(256), 0, c->stream, Pf, dp_drift, (const float*)gpf[0], (const float*)gpf[1]);
  hipLaunchKernelGGL(k_sdeb_acc, dim3(sde_nb(Pg)), dim3(256), 0, c->stream, Pg, dp_diff, (const float*)(gpg[0] + goff), (const float*)(gpg[1] + goff));
  hipLaunchKernelGGL(k_sder_join, dim3(nb), dim3(256), 0, c->stream, n, (const float*)du1b0, (const float*)Lb0, (const float*)utb, (const float*)Kb,
                     (const float*)unb, (const float*)dtf, (const float*)dtg, dW, dt, sqdt, du1b, Lb);
  // f, g at uprev: only their parameter cotangents count (uprev is constant)
  if ((rc = launch_vjp(c, uprev, nullptr, 0.f, 0.f, t, du1b, B, dtf, gpf[0]))) return rc;
  if ((rc = launch_vjp(cg, uprev, nullptr, 0.f, 0.f, t, Lb, B, dtg, gpg[0]))) return rc;
  hipLaunchKernelGGL(k_sdeb_acc1, dim3(sde_nb(Pf)), dim3(256), 0, c->stream, Pf, dp_drift, (const float*)gpf[0]);
  hipLaunchKernelGGL(k_sdeb_acc1, dim3(sde_nb(Pg)), dim3(256), 0, c->stream, Pg, dp_diff, (const float*)(gpg[0] + goff));
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return LRNDE_OK;
}

}  // extern "C"
