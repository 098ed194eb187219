// VCAB3 / VCABM3 as the NeuralODE layer's global solver — the `solver` choices "vcab3" / "vcabm3" of
// experiments/src/construct.jl:154-164 (`_ode_solver`), handed to `solve(prob, n.solver; ...)` at src/layers/neural_ode.jl:51.
// Included by lrnde_kernels.hip (inside its anonymous namespace, after the vector helpers vec_axpy / vec_norm).
//
// UPSTREAM-RECALL, parity unpinned: OrdinaryDiffEq's adams_bashforth_moulton_perform_step.jl and adams_utils.jl are not in
// /root/reference.  What is restated is the published algorithm they implement — Hairer, Norsett, Wanner, "Solving ODEs I",
// III.5: variable step size Adams methods in phi_j(n), phi*_j(n) = beta_j(n) phi_j(n), g_j(n) with the recurrences
// (5.9)/(5.10) — with what is recalled of the package around it: order 3; the first two steps are Bogacki-Shampine 3(2)
// steps with their own embedded error estimate; VCAB3 = three-term predictor, one evaluation per step, error
// dt * g_3 * phi_3(n+1); VCABM3 = two-term predictor, corrector + dt * g_2 * phi_2(n+1), a second evaluation, error
// dt * (g_3 - g_2) * phi_3(n+1) (book's indices from 0; the code counts from 1); a rejected step changes nothing but dt;
// PI controller with the order-3 exponents (beta1 = 7/30, beta2 = 2/15; gamma, qmin, qmax as for Tsit5); initial dt by
// ode_determine_initdt with order 3; saveat / dense output by the cubic Hermite interpolant on (u_n, f_n, u_{n+1}, f_{n+1})
// (the package's default for methods without an interpolant of their own).
//
// The Hermite interpolant is kept in the dense record's polynomial form [uprev, k1, P2, P3, P4] (lrnde_math.hpp): with
// D = (u_{n+1} - u_n)/dt it is y(th) = u_n + dt*(th*f_n + th^2*(P2 + th*P3)), P2 = 3D - 2 f_n - f_{n+1},
// P3 = f_n + f_{n+1} - 2D, P4 = 0 — so a recorded Adams forward feeds the SAME continuous adjoint as a Tsit5 one
// (lrnde_node_backward_recorded).  Deviation, stated: the reversed solve is the handle's Tsit5 adjoint whatever the forward's
// method was; the reference would hand n.solver to the adjoint problem too.
//
// Host-driven: per attempted step one fused elementwise launch before the evaluation, one after it (which also leaves the
// error norm's 256 block sums), the evaluation itself (k_rhs_q / k_rhs), and one read-back — the secondary solver of the
// layer, not the metric's path.  Arithmetic: operation for operation what the CPU oracle's adams_solve does (tests/test_gpu_adams.py: equal bits).

struct AdamsPre { const float *uprev, *k1, *kprev, *sp2; float *s2, *s3, *u; float b2, b3, dt, g2, g3; int three, only_s2; size_t n; };
__global__ void k_adams_pre(AdamsPre a) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < a.n; i += (size_t)gridDim.x * blockDim.x) {
    const float k1 = a.k1[i];
    const float p2 = k1 - a.kprev[i];
    const float v2 = a.b2 * p2;
    a.s2[i] = v2;
    if (a.only_s2) continue;   // second start-up step: phi*_1 for the first multistep step
    const float p3 = p2 - a.sp2[i];
    const float v3 = a.b3 * p3;
    a.s3[i] = v3;
    float sm = k1 + a.g2 * v2;
    if (a.three) sm = sm + a.g3 * v3;
    a.u[i] = a.uprev[i] + a.dt * sm;
  }
}

// phi_j(n+1) from the new evaluation, the corrector (VCABM3), the error estimate's residual and its per-block fp64 sums
struct AdamsPost { const float *du, *k1, *s2, *s3, *uprev; float* u; float cu, ce, abstol, reltol; int moulton; size_t n; double* part; };
__global__ __launch_bounds__(256) void k_adams_post(AdamsPost a) {
  __shared__ double red[4];
  double acc = 0.0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < a.n; i += (size_t)gridDim.x * blockDim.x) {
    const float q2 = a.du[i] - a.k1[i];
    const float q3 = q2 - a.s2[i];
    const float q4 = q3 - a.s3[i];
    float u = a.u[i];
    if (a.moulton) { u = u + a.cu * q3; a.u[i] = u; }
    const float ut = a.ce * q4;
    const float sc = a.abstol + fmaxf_(__builtin_fabsf(a.uprev[i]), __builtin_fabsf(u)) * a.reltol;
    const float r = ut / sc;
    acc += (double)(r * r);
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) a.part[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// the step's Hermite interpolant in the record's polynomial form; t / dt of the step into the record's time arrays
struct AdamsRec { const float *uprev, *u, *k1, *kend; float *P2, *P3; float dt; size_t n; float* rec_t; float* rec_dt; int idx; float tprev; };
__global__ void k_adams_hermite(AdamsRec a) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < a.n; i += (size_t)gridDim.x * blockDim.x) {
    const float d = (a.u[i] - a.uprev[i]) / a.dt;
    const float k1 = a.k1[i], ke = a.kend[i];
    a.P2[i] = (3.0f * d - 2.0f * k1) - ke;
    a.P3[i] = (k1 + ke) - 2.0f * d;
  }
  if (a.rec_t && blockIdx.x == 0 && threadIdx.x == 0) { a.rec_t[a.idx] = a.tprev; a.rec_dt[a.idx] = a.dt; }
}
struct AdamsEval { const float *y0, *k1, *P2, *P3, *P4; float* out; float th, ddt; size_t n; };
__global__ void k_adams_eval(AdamsEval a) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < a.n; i += (size_t)gridDim.x * blockDim.x)
    a.out[i] = lrnde::tsit5_rec_eval(a.y0[i], a.k1[i], a.P2[i], a.P3[i], a.P4[i], a.th, a.ddt);
}

inline int adams_grid(size_t n) { int nb = (int)((n + 255) / 256); return nb > 2048 ? 2048 : (nb < 1 ? 1 : nb); }

// lrnde_solve for c->solver_alg = 1 (VCAB3) | 2 (VCABM3): same arguments, same outputs, same side effects on the handle
// (save slots, tail copy of sol.u[end], dense record when a recorded forward asked for one)
int adams_solve(lrnde_ctx* c, const float* u0, int32_t B, float t0, float t1, const lrnde_solve_opts* o,
                const float* saveat_host, int32_t nsave, float* u_saved, float* t_saved_host, int32_t cap_saved,
                lrnde_stats* st, lrnde_trace_row* trace_host, int32_t cap_trace) {
  int rc;
  const size_t n = (size_t)B * c->desc.state_dim;
  const bool moulton = c->solver_alg == 2;
  const float abstol = o->abstol, reltol = o->reltol;
  const float gamma = 0.9f, qmin = 0.2f, qmax = 10.0f, qoldinit = 1e-4f;
  const float beta1 = (float)(7.0 / 30.0), beta2 = (float)(2.0 / 15.0);
  // workspace: 16 vectors of the state's size in the adjoint's allocation (a backward pass overwrites them; nothing of the
  // forward lives there once this returns)
  AdjVec v;
  if ((rc = adj_alloc(c, 2 * n, v))) return rc;
  float* w = c->adj;
  float *uprev = w, *u = w + n, *k1 = w + 2 * n, *kprev = w + 3 * n, *kend = w + 4 * n, *du = w + 5 * n;
  float *sp2 = w + 6 * n, *s2 = w + 7 * n, *s3 = w + 8 * n, *P2 = w + 9 * n, *P3 = w + 10 * n, *P4 = w + 11 * n;
  float *tmp = w + 12 * n, *kb2 = w + 13 * n, *kb3 = w + 14 * n, *interp = w + 15 * n;
  hipStream_t sq = c->stream;
  const int nb = adams_grid(n);
  HIPCHK(c, hipMemsetAsync(P4, 0, sizeof(float) * n, sq));
  HIPCHK(c, hipMemcpyAsync(uprev, u0, sizeof(float) * n, hipMemcpyDeviceToDevice, sq));
  int nsaved = 0, isave = 0, ntrace = 0;
  auto push_save = [&](float tt, const float* uu) -> int {
    if (nsaved >= cap_saved) return fail(c, LRNDE_CAPACITY, "save buffer too small (%d)", cap_saved);
    HIPCHK(c, hipMemcpyAsync(u_saved + (size_t)nsaved * n, uu, sizeof(float) * n, hipMemcpyDeviceToDevice, sq));
    if (t_saved_host) t_saved_host[nsaved] = tt;
    ++nsaved;
    return LRNDE_OK;
  };
  float t = t0;
  const float dtmax = t1 - t0;
  const float dtmin = fmaxf(eps_f(t1), eps_f(t0));
  // ode_determine_initdt with order 3; k1 = f(u0, t0) is the first step's fsalfirst
  float dt;
  {
    float d0, d1, d2;
    if ((rc = lrnde_rhs(c, uprev, t0, B, k1))) return rc;
    if ((rc = vec_norm(c, uprev, nullptr, uprev, nullptr, abstol, reltol, n, 0, &d0))) return rc;
    if ((rc = vec_norm(c, k1, nullptr, uprev, nullptr, abstol, reltol, n, 0, &d1))) return rc;
    float dt0 = ((double)d0 < 1e-5 || (double)d1 < 1e-5) ? 1e-6f : (d0 / d1) / 100.0f;
    dt0 = fminf(dt0, dtmax);
    const float one = 1.0f; const float* kk[1] = {k1};
    if ((rc = vec_axpy(c, tmp, uprev, dt0, 1, kk, &one, n))) return rc;
    if ((rc = lrnde_rhs(c, tmp, t0 + dt0, B, kb2))) return rc;
    if ((rc = vec_norm(c, kb2, k1, uprev, nullptr, abstol, reltol, n, 0, &d2))) return rc;
    d2 = d2 / dt0;
    const float maxd = fmaxf(d1, d2);
    float dt1;
    if ((double)maxd <= 1e-15) dt1 = fmaxf(1e-6f, dt0 * 1e-3f);
    else {
      const float l10 = (float)log10((double)maxd);
      const float e = (-(2.0f + l10)) / 3.0f;
      dt1 = (float)pow(10.0, (double)e);
    }
    dt = fminf(fminf(100.0f * dt0, dt1), dtmax);
  }
  st->nf = 3; st->dt_init = dt;
  float qold = qoldinit, q11 = 1.0f, dtpropose = dt;
  float h1 = 0.0f, h2 = 0.0f;
  int accept = 0, iter = 0, status = LRNDE_OK;
  if (o->save_start && (rc = push_save(t0, u0))) return rc;
  while (isave < nsave && saveat_host[isave] <= t0) ++isave;
  const float a21 = 0.5f, a32 = 0.75f;
  const float a4[3] = {(float)(2.0 / 9.0), (float)(1.0 / 3.0), (float)(4.0 / 9.0)};
  const float bt[4] = {(float)(5.0 / 72.0), (float)(-1.0 / 12.0), (float)(-1.0 / 9.0), 0.125f};
  const float c21 = 0.5f, c22 = (float)(1.0 / 6.0), c23 = (float)(1.0 / 12.0);
  int launches = 0;
  while (t < t1) {
    if (iter > 0) {
      if (accept) {
        std::swap(uprev, u);
        float* sw = kprev; kprev = k1; k1 = kend; kend = sw;
        std::swap(sp2, s2);
        dt = dtpropose;
      } else {
        dt = dt / fminf(1.0f / qmin, q11 / gamma);
      }
    }
    ++iter;
    dt = fminf(dtmax, dt);
    dt = fmaxf(dt, dtmin);
    dt = fminf(fabsf(dt), fabsf(t1 - t));
    if (iter > o->maxiters) { status = LRNDE_MAXITERS; break; }
    if (dt != dt) { status = LRNDE_DT_NAN; break; }
    if (fabsf(dt) <= fabsf(dtmin)) { status = LRNDE_DT_LESS_THAN_MIN; break; }
    const int nacc = st->naccept;
    const float b2 = nacc >= 1 ? dt / h1 : 0.0f;
    const float b3 = nacc >= 2 ? b2 * ((dt + h1) / (h1 + h2)) : 0.0f;
    float eest;
    if (nacc < 2) {   // Bogacki-Shampine 3(2)
      const float* k_1[1] = {k1};
      if ((rc = vec_axpy(c, tmp, uprev, dt, 1, k_1, &a21, n))) return rc;
      if ((rc = lrnde_rhs(c, tmp, t + 0.5f * dt, B, kb2))) return rc;
      const float* k_2[1] = {kb2};
      if ((rc = vec_axpy(c, tmp, uprev, dt, 1, k_2, &a32, n))) return rc;
      if ((rc = lrnde_rhs(c, tmp, t + 0.75f * dt, B, kb3))) return rc;
      const float* k_3[3] = {k1, kb2, kb3};
      if ((rc = vec_axpy(c, u, uprev, dt, 3, k_3, a4, n))) return rc;
      if ((rc = lrnde_rhs(c, u, t + dt, B, kend))) return rc;
      st->nf += 3;
      const float* k_4[4] = {k1, kb2, kb3, kend};
      if ((rc = vec_axpy(c, tmp, nullptr, dt, 4, k_4, bt, n))) return rc;
      if (nacc == 1) {
        AdamsPre p{};
        p.uprev = uprev; p.k1 = k1; p.kprev = kprev; p.sp2 = sp2; p.s2 = s2; p.s3 = s3; p.u = u; p.b2 = b2; p.only_s2 = 1; p.n = n;
        hipLaunchKernelGGL(k_adams_pre, dim3(nb), dim3(256), 0, sq, p);
        HIPCHK(c, hipGetLastError());
      }
      if ((rc = vec_norm(c, tmp, nullptr, uprev, u, abstol, reltol, n, 0, &eest))) return rc;
      launches += 9;
    } else {
      const float r1 = dt / (dt + h1);
      const float g2 = c21;
      const float g3 = c21 - r1 * c22;
      const float c32 = c22 - r1 * c23;
      const float r2 = dt / ((dt + h1) + h2);
      const float g4 = g3 - r2 * c32;
      AdamsPre p{};
      p.uprev = uprev; p.k1 = k1; p.kprev = kprev; p.sp2 = sp2; p.s2 = s2; p.s3 = s3; p.u = u;
      p.b2 = b2; p.b3 = b3; p.dt = dt; p.g2 = g2; p.g3 = g3; p.three = moulton ? 0 : 1; p.only_s2 = 0; p.n = n;
      hipLaunchKernelGGL(k_adams_pre, dim3(nb), dim3(256), 0, sq, p);
      HIPCHK(c, hipGetLastError());
      float* dnew = moulton ? du : kend;   // VCAB3: the evaluation at the new state is fsallast itself
      if ((rc = lrnde_rhs(c, u, t + dt, B, dnew))) return rc;
      st->nf += 1;
      AdamsPost q{};
      q.du = dnew; q.k1 = k1; q.s2 = s2; q.s3 = s3; q.uprev = uprev; q.u = u;
      q.cu = dt * g3; q.ce = moulton ? dt * (g4 - g3) : dt * g4; q.abstol = abstol; q.reltol = reltol; q.moulton = moulton ? 1 : 0;
      q.n = n; q.part = c->adj_part;
      hipLaunchKernelGGL(k_adams_post, dim3(256), dim3(256), 0, sq, q);
      HIPCHK(c, hipGetLastError());
      if (moulton) { if ((rc = lrnde_rhs(c, u, t + dt, B, kend))) return rc; st->nf += 1; }
      if ((rc = norm_readback(c, n, 0, &eest))) return rc;
      launches += moulton ? 4 : 3;
    }
    if (eest != eest) { status = LRNDE_DT_NAN; st->eest_last = eest; break; }
    const float ttmp = t + dt;
    float q;
    if (eest == 0.0f) q = 1.0f / qmax;
    else {
      if (o->exact_pow) { q11 = (float)pow((double)eest, (double)beta1); q = q11 / (float)pow((double)qold, (double)beta2); }
      else { q11 = fastpow(eest, beta1); q = q11 / fastpow(qold, beta2); }
      q = fmaxf(1.0f / qmax, fminf(1.0f / qmin, q / gamma));
    }
    accept = (eest <= 1.0f);
    if (trace_host && ntrace < cap_trace) { trace_host[ntrace].t = t; trace_host[ntrace].dt = dt; trace_host[ntrace].eest = eest; trace_host[ntrace].accepted = accept; ++ntrace; }
    st->eest_last = eest;
    if (accept) {
      const int idx = st->naccept;
      st->naccept++;
      const float dtnew = dt / q;
      qold = fmaxf(eest, qoldinit);
      const float tprev = t;
      t = (fabsf(ttmp - t1) < 100.0f * eps_f(fmaxf(fabsf(t), fabsf(t1)))) ? t1 : ttmp;
      dtpropose = fmaxf(fminf(dtmax, dtnew), fmaxf(eps_f(t), dtmin));
      h2 = h1; h1 = dt;
      bool need_poly = c->dense_on;
      for (int i = isave; i < nsave && saveat_host[i] <= t; ++i) if (saveat_host[i] != t) need_poly = true;
      const float *ry0 = uprev, *rk1 = k1, *rP2 = P2, *rP3 = P3, *rP4 = P4;
      if (need_poly) {
        AdamsRec r{};
        r.uprev = uprev; r.u = u; r.k1 = k1; r.kend = kend; r.P2 = P2; r.P3 = P3; r.dt = dt; r.n = n;
        if (c->dense_on) {
          if (idx >= c->dense_cap) return fail(c, LRNDE_CAPACITY, "dense record too small (%d steps)", c->dense_cap);
          float* slot = c->dense + (size_t)idx * REC_ARRAYS * n;
          HIPCHK(c, hipMemcpyAsync(slot, uprev, sizeof(float) * n, hipMemcpyDeviceToDevice, sq));
          HIPCHK(c, hipMemcpyAsync(slot + n, k1, sizeof(float) * n, hipMemcpyDeviceToDevice, sq));
          HIPCHK(c, hipMemsetAsync(slot + 4 * n, 0, sizeof(float) * n, sq));
          r.P2 = slot + 2 * n; r.P3 = slot + 3 * n; r.rec_t = c->dense_t; r.rec_dt = c->dense_dt; r.idx = idx; r.tprev = tprev;
          ry0 = slot; rk1 = slot + n; rP2 = r.P2; rP3 = r.P3; rP4 = slot + 4 * n;
        }
        hipLaunchKernelGGL(k_adams_hermite, dim3(nb), dim3(256), 0, sq, r);
        HIPCHK(c, hipGetLastError());
      }
      while (isave < nsave && saveat_host[isave] <= t) {
        const float ts = saveat_host[isave++];
        if (ts != t) {
          AdamsEval e{};
          e.y0 = ry0; e.k1 = rk1; e.P2 = rP2; e.P3 = rP3; e.P4 = rP4; e.out = interp; e.th = (ts - tprev) / dt; e.ddt = dt; e.n = n;
          hipLaunchKernelGGL(k_adams_eval, dim3(nb), dim3(256), 0, sq, e);
          HIPCHK(c, hipGetLastError());
          if ((rc = push_save(ts, interp))) return rc;
        } else if ((rc = push_save(t, u))) return rc;
      }
      if (o->save_everystep && (rc = push_save(t, u))) return rc;
    } else {
      st->nreject++;
    }
  }
  // sol.u[end] into the caller's array when node_forward asked for it (the Tsit5 solve does this in its last launch)
  if (c->tail_copy_dst && c->tail_copy_slot >= 0 && c->tail_copy_slot == nsaved - 1)
    HIPCHK(c, hipMemcpyAsync(c->tail_copy_dst, u_saved + (size_t)c->tail_copy_slot * n, sizeof(float) * n, hipMemcpyDeviceToDevice, sq));
  c->tail_copy_dst = nullptr;
  HIPCHK(c, hipStreamSynchronize(sq));
  st->retcode = status; st->iters = iter; st->nsaved = nsaved; st->t_final = t; st->dt_final = dt;
  c->last_launches = launches;
  if (status != LRNDE_OK) return fail(c, status, "solve stopped with retcode %d at t=%g (iter %d)", status, (double)t, iter);
  return LRNDE_OK;
}
