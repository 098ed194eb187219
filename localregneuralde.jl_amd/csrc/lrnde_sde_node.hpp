// lrnde_sde_node.hpp — the NeuralDSDE layer as the reference runs it (src/layers/neural_sde.jl:50-123): ADAPTIVE solve of
// dudt / g from x over tspan, sol(t1) / `saveat` values, a fresh integrator at (sol(t1), t1) and one local Euler-Heun step
// for reg_val (:88-105, :109-123) — and its pullback (the reference tapes the solver's own arithmetic with TrackerAdjoint,
// :12; reg_val is differentiated w.r.t. the parameters only, :42).  Included by lrnde_kernels.hip after lrnde_sde_bwd.hpp.
//
// What is restated from un-vendored StochasticDiffEq (UPSTREAM-RECALL, not readable in this image):
//   * the adaptive loop: PI controller on the step's EEst (lrnde_sde_solve_adaptive), on the CALLER's Brownian path given on
//     a uniform grid — steps are whole grid intervals, a rejected step retries a shorter piece of the same path;
//   * saveat values between steps by the SDE solvers' linear interpolant  (1 - theta) uprev + theta u ;
//   * the automatic initial dt (sde_determine_initdt: the ODE heuristic with the diffusion entering as +-3 g).
// The forward keeps a dense record of the accepted steps (start index and length on the grid, end state); the backward is
// the reverse sweep over exactly those steps with their own dt and dW = W[i + m] - W[i] (discretise-then-differentiate:
// what a tape of the solve gives), cotangents of interpolated saveat values split (1 - theta, theta) onto the two step ends.

namespace {

struct SdeSeriesEntry { float t; int k; float theta; };  // value = (1-theta) * state_before(step k) + theta * rec_u[k]; k = -1: the start value
struct SdeNodeRecord {
  bool valid = false;
  unsigned long long gen = 0;
  int B = 0, nfine = 0, K = 0, mode = 0;
  float t0 = 0.f, t2 = 0.f, h = 0.f;
  lrnde_sde_adapt_opts o{};
  const float* W = nullptr;        // the caller's path (must stay alive until the backward call)
  float* x = nullptr;              // copy of the input state
  float* rec_u = nullptr; int2* rec_im_dev = nullptr; size_t rec_floats = 0; int rec_cap = 0;
  std::vector<int2> im;            // (i, m) of every accepted step
  std::vector<SdeSeriesEntry> series;  // the caller's view of the solution (after the _CorrectedDESolution filter)
  float* u1 = nullptr; float* dWloc = nullptr; float* tmp = nullptr; size_t n_alloc = 0;
  float t1 = 0.f, dt_loc = 0.f, ee_loc = 0.f;   // the local step: its time, dt and EEst (u_new stays in tmp)
  float* gdr = nullptr; float* gdf = nullptr; size_t pf = 0, pg = 0;
};

// out = (1 - theta) * a + theta * b   (StochasticDiffEq's linear sde_interpolant)
__global__ void k_sde_lerp(size_t n, const float* a, const float* b, float theta, float* out) {
  const float om = 1.0f - theta;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = om * a[i] + theta * b[i];
}
// acc += c * g
__global__ void k_sde_axpy(size_t n, float* acc, const float* g, float c) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc[i] = acc[i] + c * g[i];
}
// dW = sqrt(dt) * z
__global__ void k_sde_scale(size_t n, const float* z, float c, float* out) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = c * z[i];
}
// partial sums of sde_determine_initdt's three norms (fp64, one per block):
//  phase 0: d0 = ||u / sk||, d1 = ||max(|f0 + 3 g0|, |f0 - 3 g0|) / sk||      (sk = abstol + |u| reltol)
//  phase 1: d2 = ||max(|df + dg|, |df - dg|) / sk||, df = f1 - f0, dg = max(|3g0 - 3g1|, |3g0 + 3g1|)
__global__ __launch_bounds__(256) void k_sde_initdt(size_t n, const float* u, const float* f0, const float* g0, const float* f1,
                                                   const float* g1, float abstol, float reltol, int phase, double* part) {
  __shared__ double red[2][4];
  double a0 = 0.0, a1 = 0.0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float sk = abstol + __builtin_fabsf(u[i]) * reltol;
    const float G0 = 3.0f * g0[i];
    if (phase == 0) {
      const float r0 = u[i] / sk;
      const float r1 = fmaxf_(__builtin_fabsf(f0[i] + G0), __builtin_fabsf(f0[i] - G0)) / sk;
      a0 += (double)(r0 * r0); a1 += (double)(r1 * r1);
    } else {
      const float G1 = 3.0f * g1[i];
      const float dg = fmaxf_(__builtin_fabsf(G0 - G1), __builtin_fabsf(G0 + G1));
      const float df = f1[i] - f0[i];
      const float r2 = fmaxf_(__builtin_fabsf(df + dg), __builtin_fabsf(df - dg)) / sk;
      a0 += (double)(r2 * r2);
    }
  }
  a0 = wave_sum(a0); a1 = wave_sum(a1);
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = a0; red[1][threadIdx.x >> 6] = a1; }
  __syncthreads();
  if (threadIdx.x == 0) {
    part[blockIdx.x] = ((red[0][0] + red[0][1]) + red[0][2]) + red[0][3];
    part[64 + blockIdx.x] = ((red[1][0] + red[1][1]) + red[1][2]) + red[1][3];
  }
}

// StochasticDiffEq.sde_determine_initdt (UPSTREAM-RECALL): 2 drift + 2 diffusion evaluations.  order: the solver's strong
// order (1/2 for Euler-Heun).  ws: 5 state-sized device vectors.  Host synchronisations: two (init only).
int sde_init_dt(lrnde_sde* s, const float* u, int B, float t, float tend, float abstol, float reltol, float order, float* ws,
                float* dt_out) {
  lrnde_ctx* c = s->drift; lrnde_ctx* cg = s->diff;
  const size_t n = (size_t)B * c->desc.state_dim;
  float *f0 = ws, *g0 = ws + n, *u1 = ws + 2 * n, *f1 = ws + 3 * n, *g1 = ws + 4 * n;
  int rc;
  if (!s->idt_part) {
    HIPCHK(c, hipMalloc(&s->idt_part, sizeof(double) * 128));
    HIPCHK(c, hipHostMalloc(&s->idt_part_host, sizeof(double) * 128));
  }
  const float dtmax = tend - t;
  if ((rc = lrnde_rhs(c, u, t, B, f0))) return rc;
  if ((rc = lrnde_rhs(cg, u, t, B, g0))) return rc;
  hipLaunchKernelGGL(k_sde_initdt, dim3(64), dim3(256), 0, c->stream, n, u, (const float*)f0, (const float*)g0, (const float*)nullptr,
                     (const float*)nullptr, abstol, reltol, 0, s->idt_part);
  HIPCHK(c, hipMemcpyAsync(s->idt_part_host, s->idt_part, sizeof(double) * 128, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  double s0 = 0.0, s1 = 0.0;
  for (int i = 0; i < 64; ++i) { s0 += s->idt_part_host[i]; s1 += s->idt_part_host[64 + i]; }
  const float d0 = (float)sqrt(s0 / (double)n), d1 = (float)sqrt(s1 / (double)n);
  float dt0 = ((double)d0 < 1e-5 || (double)d1 < 1e-5) ? 1e-6f : (d0 / d1) / 100.0f;
  dt0 = fminf(dt0, dtmax);
  HIPCHK(c, hipMemcpyAsync(u1, u, sizeof(float) * n, hipMemcpyDeviceToDevice, c->stream));
  hipLaunchKernelGGL(k_sde_axpy, dim3(sde_nb(n)), dim3(256), 0, c->stream, n, u1, (const float*)f0, dt0);  // u1 = u + dt0 * f0
  if ((rc = lrnde_rhs(c, u1, t + dt0, B, f1))) return rc;
  if ((rc = lrnde_rhs(cg, u1, t + dt0, B, g1))) return rc;
  hipLaunchKernelGGL(k_sde_initdt, dim3(64), dim3(256), 0, c->stream, n, u, (const float*)f0, (const float*)g0, (const float*)f1,
                     (const float*)g1, abstol, reltol, 1, s->idt_part);
  HIPCHK(c, hipMemcpyAsync(s->idt_part_host, s->idt_part, sizeof(double) * 64, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  double s2 = 0.0;
  for (int i = 0; i < 64; ++i) s2 += s->idt_part_host[i];
  const float d2 = (float)sqrt(s2 / (double)n) / dt0;
  const float maxd = fmaxf(d1, d2);
  float dt1;
  if ((double)maxd <= 1e-15) dt1 = fmaxf(1e-6f, dt0 * 1e-3f);
  else {
    const float l10 = (float)log10((double)maxd);
    const float e = (-(2.0f + l10)) / (order + 0.5f);
    dt1 = (float)pow(10.0, (double)e);
  }
  *dt_out = fminf(fminf(100.0f * dt0, dt1), dtmax);
  return LRNDE_OK;
}

// sde_init_dt without the two host synchronisations and in three launches instead of nine, for the one-launch step's shape
// (no time input: the evaluations at t + dt0 do not need dt0 on the host).  The four evaluations and the three norms run on
// k_sde_eh_fast's tiles (its idt_phase 1 / 2: lrnde_sde_fast.hpp), the scalar tail in k_sde_initdt_fin — the expressions of
// the host code above, fp64 sums in the partial-vector order every reduction here uses — and the result stays on the
// device: scal[0] = dt0, scal[1] = d1, scal[2] = the initial dt.  The tail also starts what consumes the dt: the adaptive
// solve's control block (ctl) or the single step's (ctrl).
__global__ void k_sde_initdt_fin(const double* part2, int nwg, double n, float dtmax, float order, float* scal, SdeCtl* ctl, float h,
                                 int nfine, Ctrl* ctrl, float t0) {
  if (threadIdx.x >= 64 || blockIdx.x != 0) return;
  const Sum3 s = reduce_partials3(part2, nwg);
  if (threadIdx.x != 0) return;
  const float dt0 = scal[0], d1 = scal[1];
  const float d2 = (float)sqrt(s.a / n) / dt0;
  const float maxd = fmaxf_(d1, d2);
  float dt1;
  if ((double)maxd <= 1e-15) dt1 = fmaxf_(1e-6f, dt0 * 1e-3f);
  else {
    const float l10 = (float)log10((double)maxd);
    const float e = (-(2.0f + l10)) / (order + 0.5f);
    dt1 = (float)pow(10.0, (double)e);
  }
  const float dt = fminf_(fminf_(100.0f * dt0, dt1), dtmax);
  scal[2] = dt;
  if (ctl) {      // k_sde_ctl_init with the host's quantisation of dt to the path's grid
    int m0 = (int)(dt / h); if (m0 < 1) m0 = 1;
    if (m0 > nfine) m0 = nfine;
    SdeCtl c;
    c.status = ST_RUNNING; c.i = 0; c.m = m0; c.cur = 0; c.naccept = 0; c.nreject = 0; c.iters = 1; c.nf = 0;
    c.qold = 1e-4f; c.eest_last = 0.f; c.dtc = dt;
    *ctl = c;
  }
  if (ctrl) {     // k_ctrl_init
    Ctrl c;
    memset(&c, 0, sizeof(c));
    c.status = ST_RUNNING; c.first = 1;
    c.t = t0; c.dt = dt; c.qold = 1e-4f; c.q11 = 1.0f; c.dtpropose = dt;
    ctrl[0] = c;
    ctrl[1] = c;
  }
}
int sde_init_dt_dev(lrnde_sde* s, const float* u, int B, float t, float tend, float abstol, float reltol, float order, float* scal,
                    SdeCtl* ctl, float h, int nfine, Ctrl* ctrl) {
  lrnde_ctx* c = s->drift;
  const int nwg = (B + NB - 1) / NB;
  if (s->idt_pp_nwg < nwg) {
    if (s->idt_pp) HIPCHK(c, hipFree(s->idt_pp));
    s->idt_pp = nullptr; s->idt_pp_nwg = 0;
    HIPCHK(c, hipMalloc(&s->idt_pp, sizeof(double) * 2 * (size_t)nwg * PSTRIDE));
    s->idt_pp_nwg = nwg;
  }
  SdeFastArgs f{};
  sde_fast_args(s, f);
  f.u = u; f.B = B; f.abstol = abstol; f.reltol = reltol;
  f.n_norm = (double)((size_t)B * c->desc.state_dim);
  f.idt_part = s->idt_pp; f.idt_part2 = s->idt_pp + (size_t)nwg * PSTRIDE; f.idt_scal = scal; f.idt_dtmax = tend - t;
  f.dW = u; f.un = nullptr; f.dt = 1.0f;   // (not used by these phases; dW is loaded before the phase branch)
  f.idt_phase = 1;
  sde_fast_launch(f.D, c->desc.hidden_dim, nwg, c->stream, f);
  f.idt_phase = 2;
  sde_fast_launch(f.D, c->desc.hidden_dim, nwg, c->stream, f);
  hipLaunchKernelGGL(k_sde_initdt_fin, dim3(1), dim3(64), 0, c->stream, (const double*)f.idt_part2, nwg, f.n_norm, tend - t, order, scal, ctl, h,
                     nfine, ctrl, t);
  HIPCHK(c, hipGetLastError());
  return LRNDE_OK;
}

int sde_node_alloc(lrnde_sde* s, SdeNodeRecord& r, int B, int nfine) {
  lrnde_ctx* c = s->drift;
  const size_t n = (size_t)B * c->desc.state_dim;
  if (r.n_alloc != n) {
    for (float** p : {&r.x, &r.u1, &r.dWloc, &r.tmp}) { if (*p) hipFree(*p); *p = nullptr; }
    r.n_alloc = 0;
    HIPCHK(c, hipMalloc(&r.x, sizeof(float) * n));
    HIPCHK(c, hipMalloc(&r.u1, sizeof(float) * n));
    HIPCHK(c, hipMalloc(&r.dWloc, sizeof(float) * n));
    HIPCHK(c, hipMalloc(&r.tmp, sizeof(float) * 6 * n));
    r.n_alloc = n;
  }
  if (r.rec_floats < (size_t)nfine * n || r.rec_cap < nfine) {
    if (r.rec_u) hipFree(r.rec_u);
    if (r.rec_im_dev) hipFree(r.rec_im_dev);
    r.rec_u = nullptr; r.rec_im_dev = nullptr; r.rec_floats = 0; r.rec_cap = 0;
    HIPCHK(c, hipMalloc(&r.rec_u, sizeof(float) * (size_t)nfine * n));
    HIPCHK(c, hipMalloc(&r.rec_im_dev, sizeof(int2) * (size_t)nfine));
    r.rec_floats = (size_t)nfine * n; r.rec_cap = nfine;
  }
  const size_t Pf = lrnde_param_count(&c->desc), Pg = (size_t)c->desc.state_dim * c->desc.state_dim + (s->diff_bias ? c->desc.state_dim : 0);
  if (r.pf != Pf || r.pg != Pg) {
    if (r.gdr) hipFree(r.gdr);
    if (r.gdf) hipFree(r.gdf);
    r.gdr = r.gdf = nullptr;
    HIPCHK(c, hipMalloc(&r.gdr, sizeof(float) * Pf));
    HIPCHK(c, hipMalloc(&r.gdf, sizeof(float) * Pg));
    r.pf = Pf; r.pg = Pg;
  }
  return LRNDE_OK;
}

unsigned long long sde_node_generation(const lrnde_sde* s) { return (s->node && s->node->valid) ? s->node->gen : 0ull; }
void sde_node_release(lrnde_sde* s) {
  if (!s->node) return;
  SdeNodeRecord& r = *s->node;
  for (float** p : {&r.x, &r.u1, &r.dWloc, &r.tmp, &r.rec_u, &r.gdr, &r.gdf}) { if (*p) hipFree(*p); *p = nullptr; }
  if (r.rec_im_dev) hipFree(r.rec_im_dev);
  delete s->node;
  s->node = nullptr;
}

}  // namespace

extern "C" {

int lrnde_sde_node_forward_record(lrnde_sde* s, const float* x, const float* W, int32_t nfine, int32_t B, float t0, float t2,
                                  const lrnde_sde_adapt_opts* o, int32_t mode, float t1_or_rand, const float* z_local,
                                  int32_t save_start, const float* saveat_host, int32_t nsave, float* u_series,
                                  float* t_series_host, int32_t cap_series, int32_t* nseries_host, float* reg_val_host,
                                  int32_t* nfe_drift_host, int32_t* nfe_diffusion_host, lrnde_stats* st, float* t1_used_host) {
  if (!s) return LRNDE_BADARG;
  lrnde_ctx* c = s->drift;
  int rc = sde_check(s, x, W, u_series, B, 1.0f);
  if (rc) return rc;
  if (!o || !st || !t_series_host || !nseries_host || !reg_val_host || nfine < 1 || !(t2 > t0) || nsave < 0 || cap_series < 1)
    return fail(c, LRNDE_BADARG, "bad arguments (nfine >= 1, t2 > t0, non-null outputs)");
  if (mode < LRNDE_MODE_NONE || mode > LRNDE_MODE_BIASED) return fail(c, LRNDE_BADARG, "mode");
  if (mode != LRNDE_MODE_NONE && !z_local) return fail(c, LRNDE_BADARG, "z_local (the local step's standard-normal draw) is required when regularising");
  for (int i = 0; i < nsave; ++i)
    if (!(saveat_host[i] >= t0 && saveat_host[i] <= t2) || (i > 0 && saveat_host[i] < saveat_host[i - 1]))
      return fail(c, LRNDE_BADARG, "saveat must be ascending and inside tspan");
  if (!s->node) s->node = new SdeNodeRecord();
  SdeNodeRecord& r = *s->node;
  r.valid = false;
  if ((rc = sde_node_alloc(s, r, B, nfine))) return rc;
  const int D = c->desc.state_dim;
  const size_t n = (size_t)B * D;
  const float h = (t2 - t0) / (float)nfine;
  int nfe_f = 0, nfe_g = 0;
  HIPCHK(c, hipMemcpyAsync(r.x, x, sizeof(float) * n, hipMemcpyDeviceToDevice, c->stream));
  // the main solve (src/layers/neural_sde.jl:50-72): dt0 <= 0 -> automatic initial dt
  // (the one-launch step's shape keeps both automatic initial dts on the device: two host synchronisations per call — the
  //  end of the solve and the end of this function — instead of nine; LRNDE_SDE_HOST_INITDT=1: the host form)
  const bool devdt = sde_uses_fast(s) && !opt(OPT_SDE_HOST_LOOP) && !opt(OPT_SDE_HOST_INITDT);
  if (devdt && !s->idt_scal) {
    HIPCHK(c, hipMalloc(&s->idt_scal, sizeof(float) * 8));
    HIPCHK(c, hipHostMalloc(&s->idt_scal_host, sizeof(float) * 8));
  }
  lrnde_sde_adapt_opts oo = *o;
  const float* dt0_dev = nullptr;
  if (!(oo.dt0 > 0.f)) {
    if (devdt) {
      if ((rc = sde_adaptive_prepare(s, r.rec_cap))) return rc;   // (before the initial dt initialises the control block)
      if ((rc = sde_init_dt_dev(s, r.x, B, t0, t2, oo.abstol, oo.reltol, 0.5f, s->idt_scal, s->ad_ctl, h, nfine, nullptr))) return rc;
      dt0_dev = s->idt_scal + 2;
      oo.dt0 = t2 - t0;   // (placeholder for the argument checks; the control block is initialised from the device value)
    } else if ((rc = sde_init_dt(s, r.x, B, t0, t2, oo.abstol, oo.reltol, 0.5f, r.tmp, &oo.dt0))) return rc;
    nfe_f += 2; nfe_g += 2;
  }
  r.im.assign((size_t)nfine, make_int2(0, 0));
  // No regulariser, no saveat, no start value: the caller's series is the end state alone — the solve leaves it in u_series itself
  // (picked on the device) and its closing synchronisation is the call's only one.  Otherwise the end state is not asked for
  // (it is the record's last slot).
  const bool end_only = mode == LRNDE_MODE_NONE && nsave == 0 && save_start <= 0 && sde_uses_fast(s) && !opt(OPT_SDE_HOST_LOOP);
  float* u_end = end_only ? u_series : nullptr;
  rc = sde_solve_adaptive_impl(s, r.x, W, nfine, B, t0, t2, &oo, u_end, st, nullptr, 0, r.rec_u, r.rec_im_dev, r.im.data(), r.rec_cap, dt0_dev);
  if (rc) return rc;
  const int K = st->naccept;
  nfe_f += 3 * (st->naccept + st->nreject); nfe_g += 3 * (st->naccept + st->nreject);
  // sol.t / sol.u as StochasticDiffEq would hold them (UPSTREAM-RECALL): saveat values by linear interpolation inside the
  // accepted step that contains them; saveat = [] saves every step; save_start < 0: DiffEq's default rule
  auto tk = [&](int k) { return t0 + (float)r.im[k].x * h; };                       // start of accepted step k
  auto tk1 = [&](int k) { return (r.im[k].x + r.im[k].y >= nfine) ? t2 : t0 + (float)(r.im[k].x + r.im[k].y) * h; };
  auto entry_at = [&](float ts) {
    SdeSeriesEntry e; e.t = ts; e.k = -1; e.theta = 0.f;
    if (!(ts > t0)) return e;
    int k = 0;
    while (k < K - 1 && tk1(k) < ts) ++k;
    e.k = k;
    e.theta = (ts >= tk1(k)) ? 1.0f : (ts - tk(k)) / ((float)r.im[k].y * h);
    return e;
  };
  float t1 = t2;
  std::vector<float> sv;   // the solve's saveat
  bool needs_correction = false, everystep = false;
  if (mode == LRNDE_MODE_UNBIASED) {
    t1 = t1_or_rand;
    if (!(t1 >= t0 && t1 <= t2)) return fail(c, LRNDE_BADARG, "t1 outside tspan");
    if (nsave > 0) { sv.assign(saveat_host, saveat_host + nsave); sv.push_back(t1); std::stable_sort(sv.begin(), sv.end()); needs_correction = true; }
    else { sv = {t1, t2}; }
  } else if (nsave > 0) sv.assign(saveat_host, saveat_host + nsave);
  else if (mode == LRNDE_MODE_BIASED) everystep = true;
  else sv = {t2};
  bool with_start = save_start > 0;
  if (save_start < 0) with_start = everystep || (!sv.empty() && sv.front() == t0);   // DiffEq: save_everystep || isempty(saveat) || tspan[1] in saveat
  std::vector<SdeSeriesEntry> sol;
  if (with_start) { SdeSeriesEntry e; e.t = t0; e.k = -1; e.theta = 0.f; sol.push_back(e); }
  if (everystep) {
    for (int k = 0; k < K; ++k) { SdeSeriesEntry e; e.t = tk1(k); e.k = k; e.theta = 1.0f; sol.push_back(e); }
  } else {
    for (float ts : sv) { if (ts == t0 && with_start) continue; sol.push_back(entry_at(ts)); }
  }
  if (sol.empty()) return fail(c, LRNDE_BADARG, "the solve saves nothing");
  SdeSeriesEntry e1; e1.t = t2; e1.k = K - 1; e1.theta = 1.0f;
  if (mode == LRNDE_MODE_BIASED) {      // :114-115  t1 = rand(rng, sol.t[1:(end - 1)])
    const int m = (int)sol.size() - 1;
    if (m < 1) return fail(c, LRNDE_BADARG, ":biased needs at least two saved times");
    int idx = (int)(t1_or_rand * (float)m);
    if (idx >= m) idx = m - 1;
    if (idx < 0) idx = 0;
    e1 = sol[idx]; t1 = e1.t;
  } else if (mode == LRNDE_MODE_UNBIASED) {
    e1 = entry_at(t1);
  }
  auto value_of = [&](const SdeSeriesEntry& e, float* out) -> int {
    const int nb = sde_nb(n);
    if (e.k < 0) { HIPCHK(c, hipMemcpyAsync(out, r.x, sizeof(float) * n, hipMemcpyDeviceToDevice, c->stream)); return LRNDE_OK; }
    const float* b = r.rec_u + (size_t)e.k * n;
    if (e.theta == 1.0f) { HIPCHK(c, hipMemcpyAsync(out, b, sizeof(float) * n, hipMemcpyDeviceToDevice, c->stream)); return LRNDE_OK; }
    const float* a = e.k == 0 ? r.x : r.rec_u + (size_t)(e.k - 1) * n;
    hipLaunchKernelGGL(k_sde_lerp, dim3(nb), dim3(256), 0, c->stream, n, a, b, e.theta, out);
    HIPCHK(c, hipGetLastError());
    return LRNDE_OK;
  };
  // the local step (:94-98, :116-118): fresh integrator at (sol(t1), t1) on (t1, t2) -> its own initial dt; one Euler-Heun
  // step with a fresh increment sqrt(dt) z
  *reg_val_host = 0.f;
  r.t1 = t1; r.dt_loc = 0.f;
  bool local_pending = false;   // the local step's record is still in flight (device-side initial dt)
  if (mode != LRNDE_MODE_NONE) {
    if (!(t1 < t2)) return fail(c, LRNDE_BADARG, "t1 must lie before the end of tspan");
    if ((rc = value_of(e1, r.u1))) return rc;
    float dtl = o->dt0;
    if (!(dtl > 0.f) && devdt) {
      // dt, sqrt(dt) z and the step itself from the device value (sde_init_dt_dev clamps to t2 - t1 as the line below does);
      // EEst, EEst * dt and dt come back with this function's closing synchronisation
      float* scal = s->idt_scal + 4;
      if ((rc = sde_init_dt_dev(s, r.u1, B, t1, t2, o->abstol, o->reltol, 0.5f, scal, nullptr, 0.f, 0, nullptr))) return rc;
      nfe_f += 2; nfe_g += 2;
      // ONE launch: sqrt(dt) z formed and left in dWloc by the step itself, the step's footer (last workgroup) fills the record slot
      if ((rc = sde_step_enqueue(s, 0, r.u1, z_local, B, t1, t2 - t1, o->abstol, o->reltol, o->delta, r.tmp, nullptr, c->ctrl + 1, scal + 2, r.dWloc))) return rc;
      HIPCHK(c, hipMemcpyAsync(c->ctrl_host, c->ctrl + 1, sizeof(Ctrl), hipMemcpyDeviceToHost, c->stream));
      HIPCHK(c, hipMemcpyAsync(s->idt_scal_host, s->idt_scal, sizeof(float) * 8, hipMemcpyDeviceToHost, c->stream));
      local_pending = true;
    } else {
      if (!(dtl > 0.f)) {
        if ((rc = sde_init_dt(s, r.u1, B, t1, t2, o->abstol, o->reltol, 0.5f, r.tmp, &dtl))) return rc;
        nfe_f += 2; nfe_g += 2;
      }
      dtl = fminf(dtl, t2 - t1);
      hipLaunchKernelGGL(k_sde_scale, dim3(sde_nb(n)), dim3(256), 0, c->stream, n, z_local, sqrtf(dtl), r.dWloc);
      float ee = 0.f, rv = 0.f;
      if ((rc = sde_step_impl(s, 0, r.u1, r.dWloc, B, t1, dtl, o->abstol, o->reltol, o->delta, r.tmp, &ee, &rv))) return rc;
      *reg_val_host = rv;
      r.ee_loc = ee;
      r.dt_loc = dtl;
    }
    nfe_f += 3; nfe_g += 3;
  }
  // the caller's view: _CorrectedDESolution drops the entries at t1 (src/utils.jl:31-33: `sol.u[t1 .!= sol.t]`)
  r.series.clear();
  for (const SdeSeriesEntry& e : sol) if (!(needs_correction && e.t == t1)) r.series.push_back(e);
  const int ns = (int)r.series.size();
  *nseries_host = ns;
  if (ns > cap_series) return fail(c, LRNDE_CAPACITY, "series buffer too small (%d > %d)", ns, cap_series);
  const bool series_done = end_only && ns == 1 && r.series[0].k == K - 1 && r.series[0].theta == 1.0f;   // (u_series[0] holds it already)
  for (int i = 0; i < ns; ++i) {
    if (!series_done && (rc = value_of(r.series[i], u_series + (size_t)i * n))) return rc;
    t_series_host[i] = r.series[i].t;
  }
  if (!series_done) HIPCHK(c, hipStreamSynchronize(c->stream));
  if (local_pending) {
    *reg_val_host = c->ctrl_host[0].reg_error;   // EEst * dt (src/perform_step.jl:205)
    r.ee_loc = c->ctrl_host[0].eest_last;
    r.dt_loc = s->idt_scal_host[6];
  }
  if (nfe_drift_host) *nfe_drift_host = nfe_f;
  if (nfe_diffusion_host) *nfe_diffusion_host = nfe_g;
  if (t1_used_host) *t1_used_host = t1;
  r.valid = true; ++r.gen; r.B = B; r.nfine = nfine; r.K = K; r.mode = mode; r.t0 = t0; r.t2 = t2; r.h = h; r.o = *o; r.W = W;
  return LRNDE_OK;
}

}  // extern "C"
namespace {
// the reverse sweep over the recorded steps as one launch + the fixed-order sum of the workgroups' partials
// (lrnde_sde_bwd_fused.hpp); what remains for the caller is the regulariser's part
// a kernel's dynamic-LDS limit, raised only when it has to grow (the call is a few microseconds of host time: not per launch)
#define SDE_LDS_LIMIT(c, kern, bytes)                                                                                        \
  do {                                                                                                                       \
    static size_t lim_[64];   /* per device (zero = the 64-KiB default) */                                                 \
    size_t& l_ = lim_[(c)->device & 63];                                                                                     \
    if ((size_t)(bytes) > (l_ ? l_ : (size_t)64 * 1024)) {                                                                   \
      HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes))); \
      l_ = (size_t)(bytes);                                                                                                  \
    }                                                                                                                        \
  } while (0)
// reg (may be NULL): the layer's record when the regulariser's local step is to ride in the deferred form (its records behind
// the sweep's, one GEMM and one reduction for both); *reg_done tells the caller whether it did
int sde_sweep_fused_core(lrnde_sde* s, const SdeSweepSrc& r, int B, const float* du_series, float* dx, float* dp_drift, float* dp_diff,
                         bool sync_after, SdeNodeRecord* reg, float w_reg, bool* reg_done) {
  if (reg_done) *reg_done = false;
  const int nseries = r.nseries;
  lrnde_ctx* c = s->drift;
  const int D = c->desc.state_dim, H = c->desc.hidden_dim;
  const int Pf = (int)lrnde_param_count(&c->desc), Pg = D * D + (s->diff_bias ? D : 0), Ptot = Pf + D * D + D;
  const int nwg = (B + SBF_NS - 1) / SBF_NS;
  const size_t need = (size_t)nwg * Ptot;
  if (s->bwf_part_n < need) {
    if (s->bwf_part) HIPCHK(c, hipFree(s->bwf_part));
    s->bwf_part = nullptr; s->bwf_part_n = 0;
    HIPCHK(c, hipMalloc(&s->bwf_part, sizeof(float) * need));
    s->bwf_part_n = need;
  }
  const size_t meta = (size_t)2 * SBF_MAXSER + 2 * (size_t)(r.K > 0 ? r.K : 1);
  if (s->bwf_meta_n < meta) {
    if (s->bwf_meta) HIPCHK(c, hipFree(s->bwf_meta));
    s->bwf_meta = nullptr; s->bwf_meta_n = 0;
    HIPCHK(c, hipMalloc(&s->bwf_meta, sizeof(int) * meta));
    if (s->bwf_meta_pin) HIPCHK(c, hipHostFree(s->bwf_meta_pin));
    s->bwf_meta_pin = nullptr;
    HIPCHK(c, hipHostMalloc(&s->bwf_meta_pin, sizeof(int) * meta));
    s->bwf_meta_n = meta;
  }
  // [series k (MAXSER ints)][series theta (MAXSER floats)][(i, m) of the K steps], staged in pinned memory: the copy needs no
  // wait (every call that uses the buffer ends in a synchronisation of this stream before the next one fills it)
  int* hm = s->bwf_meta_pin;
  for (int j = 0; j < nseries; ++j) { hm[j] = r.ser_k[j]; hm[SBF_MAXSER + j] = __builtin_bit_cast(int, r.ser_theta[j]); }
  for (int k = 0; k < r.K; ++k) { hm[2 * SBF_MAXSER + 2 * k] = r.im[k].x; hm[2 * SBF_MAXSER + 2 * k + 1] = r.im[k].y; }
  HIPCHK(c, hipMemcpyAsync(s->bwf_meta, hm, sizeof(int) * ((size_t)2 * SBF_MAXSER + 2 * (size_t)r.K), hipMemcpyHostToDevice, c->stream));
  SdeBwdFusedArgs a{};
  a.pdr = s->pdr; a.Wg = s->p2 + (size_t)D * D + D; a.bg = a.Wg + (size_t)D * D;
  a.D = D; a.H = H; a.act = c->m.act; a.B = B; a.K = r.K;
  a.x = r.x; a.rec_u = r.rec_u; a.im = reinterpret_cast<const int2*>(s->bwf_meta + 2 * SBF_MAXSER); a.W = r.W; a.h = r.h; a.dw_direct = r.dw_direct;
  a.du_series = du_series; a.nseries = nseries; a.ser_k = s->bwf_meta; a.ser_theta = reinterpret_cast<const float*>(s->bwf_meta + SBF_MAXSER);
  a.dx = dx; a.part = s->bwf_part; a.Pf = Pf; a.Ptot = Ptot;
  int nwg_red = nwg;   // partial vectors k_sde_bwd_reduce adds
  // the MNIST-SDE shape class, deferred form: the sweep leaves a record per (step, sample, evaluation point) and the parameter
  // cotangent is formed from the records afterwards at full occupancy (LRNDE_SDE_BWD_NO_DEFER=1: accumulators in the sweep).
  // The history is 1.8 KB per (step, sample): beyond 4 GiB, or if the allocation fails, the in-sweep form runs instead.
  bool defer = D <= 32 && H <= 64 && !opt(OPT_SDE_BWD_LDSACC) && !opt(OPT_SDE_BWD_NO_DEFER) && r.K > 0;
  const size_t nrec_sweep = (size_t)r.K * B * 2, nrec = nrec_sweep + (reg ? (size_t)4 * B : 0), nh = nrec * SbfR<32, 64>::HREC;
  if (defer && (nh * sizeof(float) > ((size_t)4 << 30) || nrec > (size_t)0x7fffffff)) defer = false;
  if (defer && s->bwf_hist_n < nh) {
    if (s->bwf_hist) HIPCHK(c, hipFree(s->bwf_hist));
    s->bwf_hist = nullptr; s->bwf_hist_n = 0;
    if (hipMalloc(&s->bwf_hist, sizeof(float) * nh) == hipSuccess) s->bwf_hist_n = nh;
    else { (void)hipGetLastError(); s->bwf_hist = nullptr; defer = false; }
  }
  if (defer) {
    const int ngw = 512;   // (two resident workgroups per CU take turns waiting for their batches)
    if (s->bwf_part_n < (size_t)ngw * Ptot) {
      HIPCHK(c, hipFree(s->bwf_part));
      s->bwf_part = nullptr; s->bwf_part_n = 0;
      HIPCHK(c, hipMalloc(&s->bwf_part, sizeof(float) * (size_t)ngw * Ptot));
      s->bwf_part_n = (size_t)ngw * Ptot;
      a.part = s->bwf_part;
    }
    a.hist = s->bwf_hist; a.nrec = (int)nrec;
    const size_t smr = SbfR<32, 64>::smem_bytes(2, 1, 1);   // (no cotangent vector in this kernel)
    if (opt(OPT_SDE_BWD_NO_RESIDENT)) hipLaunchKernelGGL((k_sde_eh_bwd_fused_r<32, 64, true>), dim3(nwg), dim3(SBF_NT), smr, c->stream, a);
    else hipLaunchKernelGGL((k_sde_eh_bwd_sweep_res<32, 64>), dim3(nwg), dim3(SBF_NT), smr, c->stream, a);
    if (reg) {   // the regulariser's step: four records per sample behind the sweep's, seeded with w_reg
      SdeBwdFusedArgs g = a;
      g.u1 = reg->u1; g.dW1 = reg->dWloc; g.un1 = reg->tmp; g.dt1 = reg->dt_loc; g.eest = reg->ee_loc;
      g.abstol = reg->o.abstol; g.reltol = reg->o.reltol; g.delta = reg->o.delta;
      g.rec0 = (int)nrec_sweep; g.w_reg = w_reg;
      const size_t smg4 = SbfR<32, 64>::smem_bytes(4, 1, 1);
      SDE_LDS_LIMIT(c, (k_sde_eh_reg_fused_r<32, 64, true>), smg4);
      hipLaunchKernelGGL((k_sde_eh_reg_fused_r<32, 64, true>), dim3(nwg), dim3(SBF_NT), smg4, c->stream, g);
      if (reg_done) *reg_done = true;
    }
    const size_t smg = sizeof(float) * (size_t)SBF_GEMM_BATCH * SBF_GEMM_RS;
    SDE_LDS_LIMIT(c, (k_sde_bwd_hist_gemm<32, 64>), smg);
    hipLaunchKernelGGL((k_sde_bwd_hist_gemm<32, 64>), dim3(ngw), dim3(SBF_NT), smg, c->stream, a);
    nwg_red = ngw;
  } else if (D <= 32 && H <= 64 && !opt(OPT_SDE_BWD_LDSACC)) {
    // compile-time sizes, the parameter cotangent in registers, no barrier inside the sweep
    const size_t smr = SbfR<32, 64>::smem_bytes(2, D, H);
    SDE_LDS_LIMIT(c, (k_sde_eh_bwd_fused_r<32, 64, false>), smr);
    hipLaunchKernelGGL((k_sde_eh_bwd_fused_r<32, 64>), dim3(nwg), dim3(SBF_NT), smr, c->stream, a);
  } else {
    const size_t sm = sbf_smem_bytes(D, H, 2);
    // (the kernel also has 4 KB of static LDS: the limit asked for is what this launch needs, not the CU's 160 KB)
    SDE_LDS_LIMIT(c, k_sde_eh_bwd_fused, sm);
    hipLaunchKernelGGL(k_sde_eh_bwd_fused, dim3(nwg), dim3(SBF_NT), sm, c->stream, a);
  }
  hipLaunchKernelGGL(k_sde_bwd_reduce, dim3((Ptot + 31) / 32), dim3(256), 0, c->stream, (const float*)s->bwf_part, nwg_red, Ptot, Pf, Pg, dp_drift, dp_diff, 1.0f, 0);
  HIPCHK(c, hipGetLastError());
  if (sync_after) HIPCHK(c, hipStreamSynchronize(c->stream));
  return LRNDE_OK;
}
int sde_node_sweep_fused(lrnde_sde* s, SdeNodeRecord& r, int B, const float* du_series, int nseries, float* dx, float* dp_drift,
                         float* dp_diff, bool sync_after, bool reg, float w_reg, bool* reg_done) {
  std::vector<int> sk(nseries); std::vector<float> sth(nseries);
  for (int j = 0; j < nseries; ++j) { sk[j] = r.series[j].k; sth[j] = r.series[j].theta; }
  SdeSweepSrc src{r.K, r.im.data(), r.h, r.x, r.rec_u, r.W, 0, nseries, sk.data(), sth.data()};
  return sde_sweep_fused_core(s, src, B, du_series, dx, dp_drift, dp_diff, sync_after, reg ? &r : nullptr, w_reg, reg_done);
}
// the regulariser's parameter cotangent, w_reg * d(EEst*dt)/dp of the recorded local step, ADDED to dp_drift / dp_diff
int sde_node_reg_fused(lrnde_sde* s, SdeNodeRecord& r, int B, float w_reg, float* dp_drift, float* dp_diff) {
  lrnde_ctx* c = s->drift;
  const int D = c->desc.state_dim, H = c->desc.hidden_dim;
  const int Pf = (int)lrnde_param_count(&c->desc), Pg = D * D + (s->diff_bias ? D : 0), Ptot = Pf + D * D + D;
  const int nwg = (B + SBF_NS - 1) / SBF_NS;   // (bwf_part was sized by the sweep: same grid)
  SdeBwdFusedArgs a{};
  a.pdr = s->pdr; a.Wg = s->p2 + (size_t)D * D + D; a.bg = a.Wg + (size_t)D * D;
  a.D = D; a.H = H; a.act = c->m.act; a.B = B; a.part = s->bwf_part; a.Pf = Pf; a.Ptot = Ptot;
  a.u1 = r.u1; a.dW1 = r.dWloc; a.un1 = r.tmp; a.dt1 = r.dt_loc; a.eest = r.ee_loc;
  a.abstol = r.o.abstol; a.reltol = r.o.reltol; a.delta = r.o.delta;
  if (D <= 32 && H <= 64 && !opt(OPT_SDE_BWD_LDSACC)) {
    const size_t smr = SbfR<32, 64>::smem_bytes(3, D, H);
    SDE_LDS_LIMIT(c, (k_sde_eh_reg_fused_r<32, 64>), smr);
    hipLaunchKernelGGL((k_sde_eh_reg_fused_r<32, 64>), dim3(nwg), dim3(SBF_NT), smr, c->stream, a);
  } else {
    const size_t sm = sbf_smem_bytes(D, H, 3);
    SDE_LDS_LIMIT(c, k_sde_eh_reg_fused, sm);
    hipLaunchKernelGGL(k_sde_eh_reg_fused, dim3(nwg), dim3(SBF_NT), sm, c->stream, a);
  }
  hipLaunchKernelGGL(k_sde_bwd_reduce, dim3((Ptot + 31) / 32), dim3(256), 0, c->stream, (const float*)s->bwf_part, nwg, Ptot, Pf, Pg, dp_drift, dp_diff, w_reg, 1);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return LRNDE_OK;
}
bool sde_bwd_fused_ok(const lrnde_sde* s, int nseries) {
  const lrnde_ctx* c = s->drift;
  const int D = c->desc.state_dim, H = c->desc.hidden_dim;
  return !opt(OPT_NO_SDE_BWD_FUSED) && sde_uses_fast(s) && s->pdr && nseries <= SBF_MAXSER &&
         sbf_smem_bytes(D, H, 2) + 8 * SBF_MAXSER + 1024 <= 160 * 1024;   // (64 x 128 does not fit: weights 98 KB + cotangent 83 KB)
}
}  // namespace
extern "C" {

int lrnde_sde_node_backward_recorded(lrnde_sde* s, int32_t B, const float* du_series, int32_t nseries, float w_reg, float* dx,
                                     float* dp_drift, float* dp_diff) {
  if (!s) return LRNDE_BADARG;
  lrnde_ctx* c = s->drift; lrnde_ctx* cg = s->diff;
  int rc = check_ready(c, B);
  if (rc) return rc;
  if (!du_series || !dx || !dp_drift || !dp_diff) return fail(c, LRNDE_BADARG, "null pointer");
  if (!s->node) return fail(c, LRNDE_BADARG, "no forward record (call lrnde_sde_node_forward_record first)");
  SdeNodeRecord& r = *s->node;
  if (!r.valid || r.B != B) return fail(c, LRNDE_BADARG, "no forward record for this batch (call lrnde_sde_node_forward_record first)");
  if (nseries != (int)r.series.size()) return fail(c, LRNDE_BADARG, "%d cotangents for a series of %zu states", nseries, r.series.size());
  if (cg->stream != c->stream) return fail(c, LRNDE_BADARG, "drift and diffusion contexts must share a stream");
  const int D = c->desc.state_dim;
  const size_t n = (size_t)B * D, Pf = lrnde_param_count(&c->desc), Pg2 = lrnde_param_count(&cg->desc);
  const size_t Pg = (size_t)D * D + (s->diff_bias ? D : 0), goff = (size_t)D * D + D;
  float* v[13]; float* gpf[2]; float* gpg[2];
  if ((rc = sde_bwd_ws(s, n, Pf, Pg2, v, 13, gpf, gpg))) return rc;
  float *du1 = v[0], *L = v[1], *tmp = v[2], *fb2 = v[3], *gb2 = v[4], *dtf = v[5], *dtg = v[6], *du1b = v[7], *Lb = v[8],
        *up = v[9], *duf = v[10], *dug = v[11], *w = v[12];
  const int nb = sde_nb(n);
  const bool fused = sde_bwd_fused_ok(s, nseries);
  // (the regulariser's one-launch kernel follows on the stream and ends in the call's one synchronisation)
  const bool reg_fused = r.mode != LRNDE_MODE_NONE && w_reg != 0.0f && fused && sbf_smem_bytes(D, c->desc.hidden_dim, 3) + 1024 <= 160 * 1024;
  bool reg_done = false;
  if (fused) {
    if ((rc = sde_node_sweep_fused(s, r, B, du_series, nseries, dx, dp_drift, dp_diff, !reg_fused, reg_fused, w_reg, &reg_done))) return rc;
    if (reg_done) HIPCHK(c, hipStreamSynchronize(c->stream));   // (the sweep left the closing wait to the regulariser's part)
  }
  else {
  HIPCHK(c, hipMemsetAsync(dx, 0, sizeof(float) * n, c->stream));   // dx doubles as ub, the cotangent of the current step's end state
  HIPCHK(c, hipMemsetAsync(dp_drift, 0, sizeof(float) * Pf, c->stream));
  HIPCHK(c, hipMemsetAsync(dp_diff, 0, sizeof(float) * Pg, c->stream));
  for (int k = r.K - 1; k >= 0; --k) {
    // cotangents of the series values taken inside step k: theta of each onto the step's end state ...
    for (int j = 0; j < nseries; ++j)
      if (r.series[j].k == k && r.series[j].theta != 0.f)
        hipLaunchKernelGGL(k_sde_axpy, dim3(nb), dim3(256), 0, c->stream, n, dx, du_series + (size_t)j * n, r.series[j].theta);
    const int i0 = r.im[k].x, m = r.im[k].y;
    const float t = r.t0 + (float)i0 * r.h, dt = (float)m * r.h;
    const float* u = (k == 0) ? r.x : r.rec_u + (size_t)(k - 1) * n;
    hipLaunchKernelGGL(k_sde_dw, dim3(nb), dim3(256), 0, c->stream, n, r.W + (size_t)i0 * n, r.W + (size_t)(i0 + m) * n, w);
    if ((rc = lrnde_rhs(c, u, t, B, du1))) return rc;
    if ((rc = lrnde_rhs(cg, u, t, B, L))) return rc;
    hipLaunchKernelGGL(k_sdeb_seed, dim3(nb), dim3(256), 0, c->stream, n, u, (const float*)du1, (const float*)L, (const float*)w, (const float*)dx, dt, tmp, fb2, gb2);
    if ((rc = launch_vjp(c, tmp, nullptr, 0.f, 0.f, t + dt, fb2, B, dtf, gpf[0]))) return rc;
    if ((rc = launch_vjp(cg, tmp, nullptr, 0.f, 0.f, t + dt, gb2, B, dtg, gpg[0]))) return rc;
    hipLaunchKernelGGL(k_sdeb_mid, dim3(nb), dim3(256), 0, c->stream, n, (const float*)dtf, (const float*)dtg, (const float*)w, (const float*)dx, dt, du1b, Lb, up);
    if ((rc = launch_vjp(c, u, nullptr, 0.f, 0.f, t, du1b, B, duf, gpf[1]))) return rc;
    if ((rc = launch_vjp(cg, u, nullptr, 0.f, 0.f, t, Lb, B, dug, gpg[1]))) return rc;
    hipLaunchKernelGGL(k_sdeb_end, dim3(nb), dim3(256), 0, c->stream, n, (const float*)up, (const float*)duf, (const float*)dug, dx);
    hipLaunchKernelGGL(k_sdeb_acc, dim3(sde_nb(Pf)), dim3(256), 0, c->stream, Pf, dp_drift, (const float*)gpf[0], (const float*)gpf[1]);
    hipLaunchKernelGGL(k_sdeb_acc, dim3(sde_nb(Pg)), dim3(256), 0, c->stream, Pg, dp_diff, (const float*)(gpg[0] + goff), (const float*)(gpg[1] + goff));
    // ... and 1 - theta onto its start state
    for (int j = 0; j < nseries; ++j)
      if (r.series[j].k == k && r.series[j].theta != 1.0f)
        hipLaunchKernelGGL(k_sde_axpy, dim3(nb), dim3(256), 0, c->stream, n, dx, du_series + (size_t)j * n, 1.0f - r.series[j].theta);
    HIPCHK(c, hipGetLastError());
  }
  for (int j = 0; j < nseries; ++j)   // a saved start value is the input itself
    if (r.series[j].k < 0) hipLaunchKernelGGL(k_sde_axpy, dim3(nb), dim3(256), 0, c->stream, n, dx, du_series + (size_t)j * n, 1.0f);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipStreamSynchronize(c->stream));
  }
  // the regulariser (w.r.t. the parameters only: the local step's integrator is a constant of the tape, neural_sde.jl:42)
  if (reg_done) {   // (its records rode behind the sweep's: one GEMM, one reduction)
  } else if (reg_fused) {
    if ((rc = sde_node_reg_fused(s, r, B, w_reg, dp_drift, dp_diff))) return rc;
  } else if (r.mode != LRNDE_MODE_NONE && w_reg != 0.0f) {
    float rv = 0.f;
    if ((rc = lrnde_sde_euler_heun_reg_grad(s, r.u1, r.dWloc, B, r.t1, r.dt_loc, r.o.abstol, r.o.reltol, r.o.delta, r.gdr, r.gdf, &rv))) return rc;
    hipLaunchKernelGGL(k_sde_axpy, dim3(sde_nb(Pf)), dim3(256), 0, c->stream, Pf, dp_drift, (const float*)r.gdr, w_reg);
    hipLaunchKernelGGL(k_sde_axpy, dim3(sde_nb(Pg)), dim3(256), 0, c->stream, Pg, dp_diff, (const float*)r.gdf, w_reg);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(c->stream));
  }
  return LRNDE_OK;
}

}  // extern "C"

// =====================================================================================================================
// Reverse sweep of the four-stage SRI step (src/perform_step.jl:49-106, diagonal noise; the reference differentiates the
// solve with whatever n.solver is — SOSRI by default — through TrackerAdjoint, src/layers/neural_sde.jl:12).
// loss = <du_new, u'> + w_reg * EEst*dt  ->  dx (cotangent of uprev; NULL when uprev is a constant, as for the local
// step's regulariser), dp_drift / dp_diff ADDED to (the caller zeroes them).  Eight vector-Jacobian products.
// =====================================================================================================================
namespace {
struct SriBwd {
  const float *up, *dW, *chi1, *chi2, *chi3, *un, *du_new;
  const float* k[4]; const float* g[4];
  float* kb[4]; float* gb[4]; float* upb;
  float dt, sqdt, abstol, reltol, delta, eest, w_reg, nf;
};
// seeds: cotangents of k1..k4, g1..g4 and the direct part of uprev's from u' (:90-94) and from EEst*dt (:96-103)
__global__ void k_sri_bseed(size_t n, SriBwd a, lrnde_sri_tableau T) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float k1 = a.k[0][i], k2 = a.k[1][i], k3 = a.k[2][i], k4 = a.k[3][i];
    const float g1 = a.g[0][i], g2 = a.g[1][i], g3 = a.g[2][i], g4 = a.g[3][i];
    const float up = a.up[i], un = a.un[i];
    float unb = a.du_new ? a.du_new[i] : 0.f;
    float numb = 0.f;
    if (a.w_reg != 0.f && a.eest > 0.f) {
      const float s3 = ((T.beta31 * g1 + T.beta32 * g2) + T.beta33 * g3) + T.beta34 * g4;
      const float s4 = ((T.beta41 * g1 + T.beta42 * g2) + T.beta43 * g3) + T.beta44 * g4;
      const float E2 = a.chi2[i] * s3 + a.chi3[i] * s4;
      const float E1 = a.dt * (((k1 + k2) + k3) + k4);
      const float sc = a.abstol + fmaxf_(__builtin_fabsf(up), __builtin_fabsf(un)) * a.reltol;
      const float num = a.delta * E1 + E2;
      const float r = num / sc;
      const float rb = a.w_reg * a.dt * r / (a.nf * a.eest);     // reg = dt * sqrt(mean r^2)
      numb = rb / sc;
      if (__builtin_fabsf(un) > __builtin_fabsf(up)) unb += (-rb * num / (sc * sc)) * a.reltol * (un >= 0.f ? 1.f : -1.f);
    }
    const float e1b = a.delta * numb;            // cotangent of E1
    const float e2b = unb + numb;                // cotangent of E2 (u' contains E2)
    const float kc = a.dt * unb, ke = a.dt * e1b;
    a.kb[0][i] = T.alpha1 * kc + ke; a.kb[1][i] = T.alpha2 * kc + ke; a.kb[2][i] = T.alpha3 * kc + ke; a.kb[3][i] = T.alpha4 * kc + ke;
    const float w1 = unb * a.dW[i], w2 = unb * a.chi1[i], w3 = e2b * a.chi2[i], w4 = e2b * a.chi3[i];
    a.gb[0][i] = ((T.beta11 * w1 + T.beta21 * w2) + T.beta31 * w3) + T.beta41 * w4;
    a.gb[1][i] = ((T.beta12 * w1 + T.beta22 * w2) + T.beta32 * w3) + T.beta42 * w4;
    a.gb[2][i] = ((T.beta13 * w1 + T.beta23 * w2) + T.beta33 * w3) + T.beta43 * w4;
    a.gb[3][i] = ((T.beta14 * w1 + T.beta24 * w2) + T.beta34 * w3) + T.beta44 * w4;
    a.upb[i] = unb;
  }
}
// after the VJPs of stage st (k_{st+1} = f(H0_st), g_{st+1} = g(H1_st)): ha / hb = cotangents of H0_st / H1_st
__global__ void k_sri_bjoin(size_t n, SriBwd a, lrnde_sri_tableau T, int st, const float* ha, const float* hb) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float x = ha[i], y = hb[i];
    a.upb[i] = (a.upb[i] + x) + y;
    if (st == 0) continue;   // k1 = f(uprev), g1 = g(uprev)
    const float dx = a.dt * x, dy = a.dt * y, cx = a.chi2[i] * x, sy = a.sqdt * y;
    if (st == 1) {
      a.kb[0][i] += T.a021 * dx + T.a121 * dy; a.gb[0][i] += T.b021 * cx + T.b121 * sy;
    } else if (st == 2) {
      a.kb[0][i] += T.a031 * dx + T.a131 * dy; a.gb[0][i] += T.b031 * cx + T.b131 * sy;
      a.kb[1][i] += T.a032 * dx + T.a132 * dy; a.gb[1][i] += T.b032 * cx + T.b132 * sy;
    } else {
      a.kb[0][i] += T.a041 * dx + T.a141 * dy; a.gb[0][i] += T.b041 * cx + T.b141 * sy;
      a.kb[1][i] += T.a042 * dx + T.a142 * dy; a.gb[1][i] += T.b042 * cx + T.b142 * sy;
      a.kb[2][i] += T.a043 * dx + T.a143 * dy; a.gb[2][i] += T.b043 * cx + T.b143 * sy;
    }
  }
}
}  // namespace

extern "C" int lrnde_sde_sri_step_backward(lrnde_sde* s, const lrnde_sri_tableau* tab, const float* uprev, const float* dW, const float* dZ,
                                           int32_t B, float t, float dt, float abstol, float reltol, float delta, const float* du_new,
                                           float w_reg, float* dx, float* dp_drift, float* dp_diff, float* reg_val_host) {
  int rc = sde_check(s, uprev, dW, dp_drift, B, dt);
  if (rc) return rc;
  lrnde_ctx* c = s->drift; lrnde_ctx* cg = s->diff;
  if (!tab || !dZ || !dp_drift || !dp_diff) return fail(c, LRNDE_BADARG, "null pointer");
  if (cg->stream != c->stream) return fail(c, LRNDE_BADARG, "drift and diffusion contexts must share a stream");
  const int D = c->desc.state_dim;
  const size_t n = (size_t)B * D, Pf = lrnde_param_count(&c->desc), Pg2 = lrnde_param_count(&cg->desc);
  const size_t Pg = (size_t)D * D + (s->diff_bias ? D : 0), goff = (size_t)D * D + D;
  // the step itself (u', EEst) through the forward entry point, then its stages again keeping every H0 / H1
  if (!s->node) s->node = new SdeNodeRecord();
  SdeNodeRecord& r = *s->node;
  if ((rc = sde_node_alloc(s, r, B, r.rec_cap > 0 ? r.rec_cap : 1))) return rc;
  float* un = r.u1;   // (scratch of the layer record: a recorded adaptive forward and this sweep do not interleave)
  r.valid = false;
  float ee = 0.f, rv = 0.f;
  if ((rc = lrnde_sde_sri_step(s, tab, uprev, dW, dZ, B, t, dt, abstol, reltol, delta, un, &ee, &rv))) return rc;
  if (reg_val_host) *reg_val_host = rv;
  float* v[20]; float* gpf[2]; float* gpg[2];
  if ((rc = sde_bwd_ws(s, n, Pf, Pg2, v, 20, gpf, gpg))) return rc;
  // lrnde_sde_sri_step left k1..k4, g1..g4, chi1..3 in its workspace (13 n floats); H0 / H1 of the three stages are recomputed
  const float* w = s->sri_ws;
  SriBwd a{};
  a.up = uprev; a.dW = dW; a.un = un; a.du_new = du_new;
  for (int j = 0; j < 4; ++j) { a.k[j] = w + (size_t)j * n; a.g[j] = w + (size_t)(4 + j) * n; a.kb[j] = v[j]; a.gb[j] = v[4 + j]; }
  a.chi1 = w + 10 * n; a.chi2 = w + 11 * n; a.chi3 = w + 12 * n; a.upb = v[8];
  float* H0[3] = {v[9], v[10], v[11]}; float* H1[3] = {v[12], v[13], v[14]};
  float *ha = v[15], *hb = v[16];
  const float sqdt = sqrtf(fabsf(dt));
  a.dt = dt; a.sqdt = sqdt; a.abstol = abstol; a.reltol = reltol; a.delta = delta; a.eest = ee; a.w_reg = w_reg; a.nf = (float)n;
  const lrnde_sri_tableau& T = *tab;
  const int nb = sde_nb(n);
  SriPtrs p;
  p.uprev = uprev; p.dW = dW; p.dZ = dZ;
  for (int j = 0; j < 4; ++j) { p.k[j] = const_cast<float*>(a.k[j]); p.g[j] = const_cast<float*>(a.g[j]); }
  p.chi1 = const_cast<float*>(a.chi1); p.chi2 = const_cast<float*>(a.chi2); p.chi3 = const_cast<float*>(a.chi3);
  for (int st = 1; st <= 3; ++st) {   // the stage inputs, from the k's and g's the step left (same kernel, same arithmetic)
    p.H0 = H0[st - 1]; p.H1 = H1[st - 1];
    hipLaunchKernelGGL(k_sri_stage, dim3(nb), dim3(256), 0, c->stream, n, p, T, st, dt, sqdt);
  }
  hipLaunchKernelGGL(k_sri_bseed, dim3(nb), dim3(256), 0, c->stream, n, a, T);
  HIPCHK(c, hipGetLastError());
  const float cf[3] = {T.c02, T.c03, T.c04}, cgt[3] = {T.c12, T.c13, T.c14};
  for (int st = 3; st >= 0; --st) {
    const float* xf = st ? H0[st - 1] : uprev; const float* xg = st ? H1[st - 1] : uprev;
    const float tf = st ? t + cf[st - 1] * dt : t, tg = st ? t + cgt[st - 1] * dt : t + T.c11 * dt;
    if ((rc = launch_vjp(c, xf, nullptr, 0.f, 0.f, tf, a.kb[st], B, ha, gpf[0]))) return rc;
    if ((rc = launch_vjp(cg, xg, nullptr, 0.f, 0.f, tg, a.gb[st], B, hb, gpg[0]))) return rc;
    hipLaunchKernelGGL(k_sri_bjoin, dim3(nb), dim3(256), 0, c->stream, n, a, T, st, (const float*)ha, (const float*)hb);
    hipLaunchKernelGGL(k_sdeb_acc1, dim3(sde_nb(Pf)), dim3(256), 0, c->stream, Pf, dp_drift, (const float*)gpf[0]);
    hipLaunchKernelGGL(k_sdeb_acc1, dim3(sde_nb(Pg)), dim3(256), 0, c->stream, Pg, dp_diff, (const float*)(gpg[0] + goff));
    HIPCHK(c, hipGetLastError());
  }
  if (dx) HIPCHK(c, hipMemcpyAsync(dx, a.upb, sizeof(float) * n, hipMemcpyDeviceToDevice, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return LRNDE_OK;
}
