// lrnde_sde_fast.hpp — the Lamba Euler-Heun step (src/perform_step.jl:172-206) as ONE small-latency launch, for every
// NeuralDSDE whose drift is Chain(Dense(D => H, act), Dense(H => D)) without a time input and whose diffusion is
// Dense(D => D) (src/layers/neural_sde.jl:50-72; experiments/src/construct.jl:204-205 is D = 32, H = 64) with D <= 64 and
// H <= 128.  Included by lrnde_kernels.hip inside its anonymous namespace.
//
// k_sde_step (the generic kernel) runs the step's six field evaluations through feval_tile: per evaluation it restages
// the biases, reloads the resident weight fragments from L2, and passes every intermediate (du1, L, K, tmp, ...) through
// global memory between evaluations — 28 us for 15.7 MFLOP.  Here a workgroup is four waves on 16 columns and nothing
// leaves the CU between the first load of (u, dW) and the store of u_new:
//   * the three weight matrices (24 KB at 32/64) are MFMA A fragments in registers for the whole launch: wave w holds the
//     hidden tiles w, w + 4 of Dense-1; the 2 DT output-tile jobs of the second phase — DT tiles of Dense-2, DT tiles of
//     the diffusion — go round-robin to the waves (job j = wave + 4 r: Dense-2 tile j for j < DT, diffusion tile j - DT
//     otherwise; a wave owns at most one of each);
//   * a round = [drift Dense-1 + activation, and this wave's diffusion tile] barrier [drift Dense-2 of this wave's tile +
//     the elementwise algebra] barrier; the step is three rounds: (f,g)(u) -> (f,g)(tmp) -> f(K), g(utilde);
//   * the elementwise algebra of the step stays in the C-fragment registers of the wave that owns the Dense-2 tile
//     (4 rows x 1 column per lane).
// Arithmetic: the canonical dot products (k-ordered fma chains = v_mfma_f32_16x16x4_f32 chains over the k-groups in
// order, segments of 112 rows added left to right — H = 128 has two), the same activation polynomial and the same
// elementwise expressions as k_sde_step, so results are bit-identical to it and to the oracle
// (tests/test_gpu_parity.py::test_sde_*, tests/test_gpu_switches.py).  Rows beyond D / H are zero-padded fragments.
//
// The adaptive solve's controller (lrnde_sde_solve_adaptive, lrnde_sde_node_forward_record) runs in this kernel's footer
// (last workgroup to arrive): PI controller on EEst, position on the caller's Brownian grid, state buffer flip, and — for
// the layer's recorded forward — the dense record of the accepted steps (start index, length, end state) that the
// reverse sweep of lrnde_sde_node_backward_recorded walks.

constexpr int SF_NT = 256;
// -DLRNDE_SDE_STAMPS (tools/sde_persist_probe.hip): s_memtime of workgroup 0 at the phase boundaries of its first 32 steps
#ifdef LRNDE_SDE_STAMPS
__device__ unsigned long long g_sde_stamps[32][16];
#define SDE_STAMP(i) do { if (PERSIST && blockIdx.x == 0 && threadIdx.x == 0 && it < 32) g_sde_stamps[it][(i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define SDE_STAMP(i) do { } while (0)
#endif
constexpr int SF_MAXD = 64, SF_MAXH = 128;

struct SdeFastArgs {
  const f32x4* W1p; int KG1;         // drift Dense-1 fragments [MT1p][KG1][64]
  const f32x4* W2p; int KG2p;        // drift Dense-2 fragments [MT2][KG2p][64]
  const f32x4* Wgp; int KGgp;        // diffusion (second layer of the identity+Dense form) [MT2][KGgp][64]
  const float *b1, *b2, *bg;         // zero-padded to the tile sizes
  int act, D;
  const float* u; const float* dW; float* un;   // (B, D)
  int B;
  float dt, abstol, reltol, delta;
  const float* dt_dev;   // non-NULL: the single step's dt is read from here (a dt an earlier launch on the stream computed)
  float* dW_scaled;      // non-NULL: `dW` holds standard-normal draws z; the step uses sqrt(dt) * z and leaves it here (k_sde_scale's expression)
  // sde_determine_initdt on this kernel's tiles (idt_phase 1 / 2, lrnde_sde_node.hpp: sde_init_dt_dev): per-workgroup fp64
  // sums of the norms into idt_part (phase 1: d0, d1) / idt_part2 (phase 2: d2); phase 2 leaves {dt0, d1} in idt_scal
  int idt_phase; double* idt_part; double* idt_part2; float* idt_scal; float idt_dtmax;
  // fixed-grid solve as ONE launch (lrnde_sde_solve_fixed): march_n > 0 — the workgroup takes its columns through march_n steps,
  // step i with the increments dW + i * B * D, end state to un + i * B * D, and leaves its partial sum of step i in
  // march_part[(i * gridDim.x + workgroup) * PSTRIDE]: no step waits for another workgroup (nothing consumes EEst inside
  // the solve; k_sde_march_records forms every step's record afterwards, in the partial-vector order)
  int march_n; double* march_part;
  int dbg_stall;   // test hook (LRNDE_SDE_PERSIST_STALL): the persistent form waits for one workgroup more than there are — its barrier times out
  double* part;      // per-workgroup fp64 sums of the squared residual (PSTRIDE doubles each)
  int* arrive;       // fixed-grid solve: arrival counter of the step's footer, or NULL
  Ctrl* rec;         //   ... and the record slot the last workgroup fills (EEst, EEst*dt)
  double n_norm;     // elements of the norm (B * D)
  // adaptive solves: the controller lives on the device.  The launch reads the control block (grid position i, step
  // length m in grid intervals, which of the two state buffers is current), forms dW from the caller's path itself, and
  // its last workgroup runs the PI controller and writes the block for the next launch.
  struct SdeCtl* ctl; const float* Wpath; float* ua; float* ub; int nfine; float t0, h;
  float gamma, qmin, qmax, beta1, beta2; int maxiters;
  lrnde_trace_row* trace; int cap_trace;
  unsigned long long* prog;  // pinned host word: (launches whose footer ran) | status << 32
  int jlaunch;
  // the layer's recorded forward: end state of accepted step k -> rec_u[k] (slot = accepted steps so far; a rejected
  // attempt's slot is rewritten by the retry), its (i, m) -> rec_im[k]
  float* rec_u; int2* rec_im; int rec_cap;
  // persistent form (k_sde_eh_fast<DT, HT, true>, cooperative launch): the base of the TWO blocks of {partial sum, tag} slots the
  // steps alternate between (zero at launch: no slot carries a step's tag yet)
  double* part2;
};
// dtc: the controller's step proposal as a REAL number; the step taken is its floor on the path's grid (m intervals, at least
// one).  Growth accumulates in dtc — with qmax = 1.125 a proposal quantised after every step could never leave m = 1.
struct SdeCtl { int status, i, m, cur, naccept, nreject, iters, nf; float qold, eest_last, dtc; };

// canonical dot product over NKG k-groups of 16 rows: fma chains over segments of SEGK k-groups, partials left to right
template <int NKG>
__device__ __forceinline__ f32x4 sf_chain(const f32x4 (&frag)[NKG], const f32x4* xb, int lane) {
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int kg = 0; kg < (NKG < SEGK ? NKG : SEGK); ++kg) acc = mfma4(frag[kg], xb[kg * 64 + lane], acc);
  if constexpr (NKG > SEGK) {
    f32x4 acc2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kg = SEGK; kg < NKG; ++kg) acc2 = mfma4(frag[kg], xb[kg * 64 + lane], acc2);
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] = acc[r] + acc2[r];
  }
  return acc;
}

// The controller of lrnde_sde_solve_adaptive's host loop, expression for expression, on a copy of the control block: PI
// step factor from EEst, the proposal kept as a real number, position on the caller's grid, buffer flip.  `writer` (one
// thread of the whole grid) also leaves the trace row and the accepted step's (start, length).
// qpow = fastpow(c.qold, a.beta2): does not depend on this step's EEst — the cooperative form computes it while the partial
// sums are still in flight.
__device__ __forceinline__ void sde_ctl_update(SdeCtl& c, float eest, float dt, const SdeFastArgs& a, bool writer, float qpow) {
  const float qoldinit = 1e-4f;
  c.nf += 3; c.eest_last = eest;
  if (eest != eest) { c.status = LRNDE_DT_NAN; return; }
  float q;
  if (eest == 0.0f) q = 1.0f / a.qmax;
  else {
    const float q11 = fastpow(eest, a.beta1);
    q = q11 / qpow;
    q = fmaxf_(1.0f / a.qmax, fminf_(1.0f / a.qmin, q / a.gamma));
  }
  const int accepted = eest <= 1.0f;
  const int ntr = c.naccept + c.nreject;
  if (writer && a.trace && ntr < a.cap_trace) {
    lrnde_trace_row r; r.t = a.t0 + (float)c.i * a.h; r.dt = dt; r.eest = eest; r.accepted = accepted;
    a.trace[ntr] = r;
  }
  c.dtc = (accepted ? fmaxf_(c.dtc, dt) : dt) / q;
  int mnew = (int)(c.dtc / a.h);
  if (mnew < 1) mnew = 1;
  if (accepted) {
    if (a.rec_im) {
      if (c.naccept < a.rec_cap) { if (writer) a.rec_im[c.naccept] = make_int2(c.i, c.m); }
      else c.status = LRNDE_CAPACITY;
    }
    c.naccept++;
    c.qold = fmaxf_(eest, qoldinit);
    c.i += c.m;
    c.cur ^= 1;
    c.m = mnew;
    if (c.i >= a.nfine && c.status == ST_RUNNING) c.status = ST_DONE;
  } else {
    c.nreject++;
    if (c.m == 1) c.status = LRNDE_DT_LESS_THAN_MIN;  // the path's grid cannot be refined further
    else c.m = mnew < c.m ? mnew : c.m - 1;
  }
  if (c.status == ST_RUNNING) {
    if (c.m > a.nfine - c.i) c.m = a.nfine - c.i;
    if (++c.iters > a.maxiters) c.status = LRNDE_MAXITERS;
  }
}

// PERSIST: the whole adaptive solve in ONE cooperative launch.  The workgroups stay resident with their weight fragments and
// their columns' state in registers; a step ends in a grid barrier (tagged partial-sum slots + bounded polling) after which EVERY
// workgroup adds the partial sums in the same fixed order and runs the same controller on its own copy of the control
// block — the same decisions everywhere, no broadcast, and the arithmetic of a step is the code of the one-launch form.
template <int DT, int HT, bool PERSIST = false>
__global__ __launch_bounds__(SF_NT) void k_sde_eh_fast(SdeFastArgs a) {
  static_assert(DT >= 1 && DT <= 4 && HT >= 1 && HT <= 8, "D <= 64, H <= 128");
  constexpr int NJ = (HT + 3) / 4;   // hidden tiles per wave
  const bool adapt = a.ctl != nullptr;
  int ad_i = 0, ad_m = 0, ad_slot = 0;
  SdeCtl cc{};
  if (adapt) {
    cc = *a.ctl;   // written by the previous launch's last workgroup (kernel boundary in between) / by k_sde_ctl_init
    if (cc.status != ST_RUNNING) return;
    ad_i = cc.i; ad_m = cc.m; ad_slot = cc.naccept;
    a.dt = (float)ad_m * a.h;
    a.u = cc.cur ? a.ub : a.ua;
    a.un = cc.cur ? a.ua : a.ub;
  } else if (a.dt_dev) a.dt = *a.dt_dev;
  __shared__ SdeCtl sh_cc;
  // LDS: three x tiles in B-operand layout (16 DT rows x 16 columns each: [kg][64 lanes] float4), the h tile, the
  // diffusion results in C-fragment order, the reduction scratch
  __shared__ f32x4 xA[DT * 64], xB[DT * 64], xC[DT * 64], hl[HT * 64], gl[DT * 64];
  __shared__ double red[4];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int n = lane & 15, rq = lane >> 4;
  const int b0 = blockIdx.x * 16;
  const bool colok = b0 + n < a.B;
  const int D = a.D;
  // jobs of this wave: Dense-2 tile t (wave < DT), diffusion tile tg (job wave or wave + 4 in [DT, 2 DT))
  const bool has_d2 = wave < DT;
  const int t = has_d2 ? wave : 0;
  const int tg = (wave >= DT && wave < 2 * DT) ? wave - DT : ((wave + 4 >= DT && wave + 4 < 2 * DT) ? wave + 4 - DT : -1);
  // resident A fragments
  f32x4 w1[NJ][DT], w2[HT], wg[DT];
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int ht = wave + 4 * j;
#pragma unroll
    for (int kg = 0; kg < DT; ++kg) w1[j][kg] = ht < HT ? a.W1p[((size_t)ht * a.KG1 + kg) * 64 + lane] : zero4;
  }
#pragma unroll
  for (int kg = 0; kg < HT; ++kg) w2[kg] = has_d2 ? a.W2p[((size_t)t * a.KG2p + kg) * 64 + lane] : zero4;
#pragma unroll
  for (int kg = 0; kg < DT; ++kg) wg[kg] = tg >= 0 ? a.Wgp[((size_t)tg * a.KGgp + kg) * 64 + lane] : zero4;
  f32x4 b1v[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) b1v[j] = (wave + 4 * j < HT) ? *reinterpret_cast<const f32x4*>(a.b1 + (wave + 4 * j) * 16 + rq * 4) : zero4;
  const f32x4 b2v = has_d2 ? *reinterpret_cast<const f32x4*>(a.b2 + t * 16 + rq * 4) : zero4;
  const f32x4 bgv = tg >= 0 ? *reinterpret_cast<const f32x4*>(a.bg + tg * 16 + rq * 4) : zero4;
  // this lane's four rows (16 t + 4 rq + r) of column n: state and increments (the Dense-2 waves own the elementwise work)
  const int row0 = t * 16 + rq * 4;
  const bool vec = (D & 3) == 0;                 // rows in whole quads: one 16-byte access per lane
  const bool live = has_d2 && colok && row0 < D;
  const size_t g = (size_t)(b0 + n) * D + row0;
  const size_t nn = (size_t)a.B * D;
  auto ld4s = [&](const float* p) {
    f32x4 v = zero4;
    if (vec) v = *reinterpret_cast<const f32x4*>(p + g);
    else {
#pragma unroll
      for (int r = 0; r < 4; ++r) if (row0 + r < D) v[r] = p[g + r];
    }
    return v;
  };
  auto st4s = [&](float* p, const f32x4& v) {
    if (vec) *reinterpret_cast<f32x4*>(p + g) = v;
    else {
#pragma unroll
      for (int r = 0; r < 4; ++r) if (row0 + r < D) p[g + r] = v[r];
    }
  };
  f32x4 u4 = zero4, w4 = zero4;
  if (live) u4 = ld4s(a.u);
  for (int it = 0;; ++it) {   // (one trip unless PERSIST)
  SDE_STAMP(0);
  if (live) {
    if (adapt) {  // dW = W[i + m] - W[i], the path's own increment (the expression of k_sde_dw)
      const f32x4 lo = ld4s(a.Wpath + (size_t)ad_i * nn);
      const f32x4 hi = ld4s(a.Wpath + (size_t)(ad_i + ad_m) * nn);
#pragma unroll
      for (int r = 0; r < 4; ++r) w4[r] = hi[r] - lo[r];
    } else {
      w4 = ld4s(a.dW);
      if (a.dW_scaled) {
        const float cz = __builtin_sqrtf(a.dt);
#pragma unroll
        for (int r = 0; r < 4; ++r) w4[r] = cz * w4[r];
        st4s(a.dW_scaled, w4);
      }
    }
  }
  // B-operand image of rows 16 t + 4 rq + r, column n: float4 index t*64 + r*16 + n, component rq
  auto put = [&](f32x4* x, const f32x4& v) {
    float* p = reinterpret_cast<float*>(x) + ((t * 64 + n) << 2) + rq;
#pragma unroll
    for (int r = 0; r < 4; ++r) p[r * 64] = v[r];
  };
  if (has_d2) put(xA, u4);
  __syncthreads();
  const float dt = a.dt, hdt = dt / 2.0f, sqdt = __builtin_sqrtf(dt);

  // drift Dense-1 + activation of this wave's hidden tiles on the x tile xs -> hl
  auto dense1 = [&](const f32x4* xs) {
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int ht = wave + 4 * j;
      if (ht < HT) {
        const f32x4 acc = sf_chain<DT>(w1[j], xs, lane);
        float* p = reinterpret_cast<float*>(hl) + ((ht * 64 + n) << 2) + rq;
#pragma unroll
        for (int r = 0; r < 4; ++r) p[r * 64] = act_apply_sel(a.act, acc[r] + b1v[j][r]);   // (four independent elements: the select form interleaves them)
      }
    }
  };
  auto diffusion = [&](const f32x4* xs) {  // tile tg of g(xs) -> gl (C-fragment order)
    if (tg < 0) return;
    f32x4 acc = sf_chain<DT>(wg, xs, lane);
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] = acc[r] + bgv[r];
    gl[tg * 64 + lane] = acc;
  };
  auto dense2 = [&]() {  // tile t of f = W2 h + b2
    f32x4 acc = sf_chain<HT>(w2, hl, lane);
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] = acc[r] + b2v[r];
    return acc;
  };

  if constexpr (!PERSIST) {
    if (a.idt_phase) {
      // StochasticDiffEq.sde_determine_initdt's evaluations and norms (UPSTREAM-RECALL; the expressions of k_sde_initdt):
      // phase 1: f0 = f(u), g0 = g(u) -> sums of (u / sk)^2 and (max(|f0 + 3 g0|, |f0 - 3 g0|) / sk)^2;
      // phase 2: dt0 from phase 1's sums (every workgroup adds them in the same fixed order), u1 = u + dt0 f0, f1, g1 ->
      //          sum of (max(|df + dg|, |df - dg|) / sk)^2.  The launch ends here.
      __shared__ double red2[4];
      __shared__ float sh_dt0;
      dense1(xA);
      diffusion(xA);
      __syncthreads();
      f32x4 f0 = zero4, g0 = zero4;
      if (has_d2) { f0 = dense2(); g0 = gl[t * 64 + lane]; }
      double a0 = 0.0, a1 = 0.0;
      if (a.idt_phase == 1) {
        if (live) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            if (row0 + r >= D) continue;
            const float sk = a.abstol + __builtin_fabsf(u4[r]) * a.reltol;
            const float G0 = 3.0f * g0[r];
            const float r0 = u4[r] / sk;
            const float r1 = fmaxf_(__builtin_fabsf(f0[r] + G0), __builtin_fabsf(f0[r] - G0)) / sk;
            a0 += (double)(r0 * r0); a1 += (double)(r1 * r1);
          }
        }
      } else {
        if (threadIdx.x < 64) {
          const Sum3 s = reduce_partials3(a.idt_part, (int)gridDim.x);
          if (lane == 0) {
            const float d0 = (float)sqrt(s.a / a.n_norm), d1 = (float)sqrt(s.b / a.n_norm);
            float dt0 = ((double)d0 < 1e-5 || (double)d1 < 1e-5) ? 1e-6f : (d0 / d1) / 100.0f;
            dt0 = fminf_(dt0, a.idt_dtmax);
            sh_dt0 = dt0;
            if (blockIdx.x == 0) { a.idt_scal[0] = dt0; a.idt_scal[1] = d1; }
          }
        }
        __syncthreads();
        const float dt0 = sh_dt0;
        if (has_d2) {
          f32x4 u1;
#pragma unroll
          for (int r = 0; r < 4; ++r) u1[r] = u4[r] + dt0 * f0[r];
          put(xB, u1);
        }
        __syncthreads();
        dense1(xB);
        diffusion(xB);
        __syncthreads();
        if (has_d2) {
          const f32x4 f1 = dense2();
          const f32x4 g1 = gl[t * 64 + lane];
          if (live) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              if (row0 + r >= D) continue;
              const float sk = a.abstol + __builtin_fabsf(u4[r]) * a.reltol;
              const float G0 = 3.0f * g0[r], G1 = 3.0f * g1[r];
              const float dg = fmaxf_(__builtin_fabsf(G0 - G1), __builtin_fabsf(G0 + G1));
              const float df = f1[r] - f0[r];
              const float r2 = fmaxf_(__builtin_fabsf(df + dg), __builtin_fabsf(df - dg)) / sk;
              a0 += (double)(r2 * r2);
            }
          }
        }
      }
      if (has_d2) {
        a0 = wave_sum_dpp(a0); a1 = wave_sum_dpp(a1);
        if (lane == 0) { red[wave] = a0; red2[wave] = a1; }
      }
      __syncthreads();
      if (threadIdx.x == 0) {
        double t0s = red[0], t1s = red2[0];
#pragma unroll
        for (int w = 1; w < DT; ++w) { t0s += red[w]; t1s += red2[w]; }
        double* pp = (a.idt_phase == 1 ? a.idt_part : a.idt_part2) + (size_t)blockIdx.x * PSTRIDE;
        pp[0] = t0s; pp[1] = t1s; pp[2] = 0.0;
      }
      return;
    }
  }
  SDE_STAMP(1);
  // ---- round 1: du1 = f(u), L = g(u) (:174-176) ----
  dense1(xA);
  SDE_STAMP(8);
  diffusion(xA);
  SDE_STAMP(9);
  __syncthreads();
  SDE_STAMP(10);
  f32x4 du1 = zero4, L = zero4, Kv = zero4;
  if (has_d2) {
    du1 = dense2();
    SDE_STAMP(11);
    L = gl[t * 64 + lane];
    f32x4 tmp, ut;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      Kv[r] = u4[r] + dt * du1[r];          // :175
      tmp[r] = Kv[r] + L[r] * w4[r];        // :179,183
      ut[r] = u4[r] + L[r] * sqdt;          // :196
    }
    put(xB, tmp); put(xC, ut);
  }
  SDE_STAMP(12);
  __syncthreads();
  SDE_STAMP(2);
  // ---- round 2: g(tmp), f(tmp) at t + dt (:184, :191) ----
  dense1(xB);
  diffusion(xB);
  __syncthreads();
  f32x4 un = zero4;
  if (has_d2) {
    const f32x4 f2 = dense2();
    const f32x4 g2 = gl[t * 64 + lane];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float gtmp2 = 0.5f * (L[r] + g2[r]);
      const float noise2 = gtmp2 * w4[r];
      un[r] = (u4[r] + hdt * (du1[r] + f2[r])) + noise2;   // :191
    }
    if (live) {
      if (!PERSIST) st4s(a.un, un);   // (PERSIST: the state lives in registers until the solve ends)
      if (a.rec_u && ad_slot < a.rec_cap) st4s(a.rec_u + (size_t)ad_slot * nn, un);
    }
    put(xA, Kv);   // xA is free: every wave has read it (barrier above)
  }
  __syncthreads();
  SDE_STAMP(3);
  // ---- round 3: du2 = f(K, t + dt) (:193), g(utilde, t) (:197) ----
  dense1(xA);
  diffusion(xC);
  __syncthreads();
  double acc = 0.0;
  if (has_d2) {
    const f32x4 du2 = dense2();
    const f32x4 g3 = gl[t * 64 + lane];
    if (live) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (row0 + r >= D) continue;
        const float Ed = (dt * (du2[r] - du1[r])) / 2.0f;                        // :194
        const float ggp = (g3[r] - L[r]) / sqdt;                                 // :197
        const float En = (ggp * (w4[r] * w4[r])) / 2.0f;                         // :198
        const float sc = a.abstol + fmaxf_(__builtin_fabsf(u4[r]), __builtin_fabsf(un[r])) * a.reltol;
        const float rr = (a.delta * Ed + En) / sc;                               // :214-216
        const float sq = rr * rr;
        acc += (double)sq;
      }
    }
    acc = wave_sum_dpp(acc);
    if (lane == 0) red[wave] = acc;
  }
  __syncthreads();
  SDE_STAMP(4);
  if constexpr (PERSIST) {
    const int nwg = (int)gridDim.x + a.dbg_stall;
    double* blk = a.part2 + (size_t)(it & 1) * nwg * PSTRIDE;   // (two blocks: a fast workgroup's next step must not overwrite what a slow one still reads)
    if (threadIdx.x < 64) {
      double tot = red[0];
#pragma unroll
      for (int w = 1; w < DT; ++w) tot += red[w];
      // The step's grid barrier IS the exchange of the partial sums: a workgroup publishes {sum, tag} as ONE 16-byte agent-scope
      // store — tag = step number and a checksum of the sum's bits, so that a reader can tell a slot of this step from a stale
      // or (should a 16-byte access ever be split) a torn one — and every workgroup polls all slots with 16-byte loads until
      // each carries this step's tag.  Two trips past the L2 per step (publish, poll) instead of four (store, arrive, poll,
      // read).  Bounded: 50 ms on the 100-MHz clock, then the solve ends with an error instead of hanging the queue.
      typedef unsigned u32x4_ __attribute__((ext_vector_type(4)));
      const unsigned step_tag = (unsigned)(it + 1);
      if (lane == 0) {
        const unsigned long long vb = __builtin_bit_cast(unsigned long long, tot);
        const u32x4_ w = {(unsigned)vb, (unsigned)(vb >> 32), (unsigned)(vb >> 32) ^ (unsigned)vb, step_tag};
        double* p = blk + (size_t)blockIdx.x * PSTRIDE;
        asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(w) : "memory");
      }
      float qpow = fastpow(cc.qold, a.beta2);   // (ahead of the wait: see sde_ctl_update)
      asm volatile("" : "+v"(qpow));              // (pinned here: the value is only used after the polling loop)
      PartLoads L;
      const unsigned long long t0c = __builtin_amdgcn_s_memrealtime();
      bool ok = true;
      for (;;) {
        u32x4_ v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int i = lane + 64 * u;
          const double* p = blk + (size_t)(i < nwg ? i : 0) * PSTRIDE;
          asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v[u]) : "v"(p) : "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        bool good = true;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          asm volatile("" : "+v"(v[u]));
          const bool mine = lane + 64 * u < nwg;
          good = good && (!mine || (v[u].w == step_tag && v[u].z == (v[u].x ^ v[u].y)));
          L.va[u] = __builtin_bit_cast(double, (unsigned long long)v[u].x | ((unsigned long long)v[u].y << 32));
        }
        if (__ballot(good) == ~0ull) break;
        if (__builtin_amdgcn_s_memrealtime() - t0c > 5000000ull) { ok = false; break; }
        __builtin_amdgcn_s_sleep(1);
      }
      SDE_STAMP(5);
      SdeCtl c2 = cc;
      if (!ok) c2.status = LRNDE_HIP_ERROR;
      else {
        const Sum3 s3 = part_finish<false>(L, blk, nwg);   // (the fixed order of every reduction of these partials)
        const float eest = rms_from(s3.a, a.n_norm);
        sde_ctl_update(c2, eest, dt, a, blockIdx.x == 0 && lane == 0, qpow);
      }
      if (lane == 0) sh_cc = c2;
    }
    __syncthreads();
    SDE_STAMP(6);
    const int nacc0 = cc.naccept;
    cc = sh_cc;
    if (cc.naccept != nacc0) u4 = un;   // accepted: the end state is the next step's start state (rows of the Dense-2 waves)
    if (cc.status != ST_RUNNING) {
      if (live) st4s(cc.cur ? a.ub : a.ua, u4);
      if (blockIdx.x == 0 && threadIdx.x == 0) {
        *a.ctl = cc;
        __hip_atomic_store(a.prog, (unsigned long long)(unsigned)(it + 1) | ((unsigned long long)(unsigned)cc.status << 32),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
      return;
    }
    ad_i = cc.i; ad_m = cc.m; ad_slot = cc.naccept;
    a.dt = (float)ad_m * a.h;
    continue;
  } else {
  if (a.march_n > 0) {
    if (threadIdx.x == 0) {
      double tot = red[0];
#pragma unroll
      for (int w = 1; w < DT; ++w) tot += red[w];
      double* p = a.march_part + ((size_t)it * gridDim.x + blockIdx.x) * PSTRIDE;
      p[0] = tot; p[1] = 0.0; p[2] = 0.0;
    }
    if (it + 1 == a.march_n) return;
    u4 = un;                       // the end state is the next step's start state (rows of the Dense-2 waves)
    a.dW += nn; a.un += nn;
    __syncthreads();               // red and the tiles are rewritten by the next step
    continue;
  }
  if (threadIdx.x < 64) {
    double tot = red[0];
#pragma unroll
    for (int w = 1; w < DT; ++w) tot += red[w];
    if (!a.arrive) {
      if (lane == 0) { double* p = a.part + (size_t)blockIdx.x * PSTRIDE; p[0] = tot; p[1] = 0.0; p[2] = 0.0; }
      return;
    }
    // the step's own footer — the last workgroup to arrive reduces the partials and fills the record / runs the controller
    int last = 0;
    if (lane == 0) {
      double* p = a.part + (size_t)blockIdx.x * PSTRIDE;
      __hip_atomic_store(p + 0, tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(p + 1, 0.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(p + 2, 0.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      last = __hip_atomic_fetch_add(a.arrive, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == (int)gridDim.x - 1;
    }
    last = __shfl(last, 0, 64);
    if (!last) return;
    const Sum3 s = reduce_partials3(a.part, (int)gridDim.x);
    if (lane == 0) {
      const float eest = rms_from(s.a, a.n_norm);
      if (!adapt) {
        a.rec->eest_last = eest;
        a.rec->reg_error = eest * dt;
        a.rec->status = ST_DONE;
      } else {
        SdeCtl c = *a.ctl;
        sde_ctl_update(c, eest, dt, a, true, fastpow(c.qold, a.beta2));
        *a.ctl = c;
        __hip_atomic_store(a.prog, (unsigned long long)(unsigned)(a.jlaunch + 1) | ((unsigned long long)(unsigned)c.status << 32),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
      __hip_atomic_store(a.arrive, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  return;
  }
  }   // for (it)
}

// launch by shape: DT = ceil(D / 16) in 1..4, HT = ceil(H / 16) in 1..8
inline bool sde_fast_shape(int D, int H) { return D >= 1 && D <= SF_MAXD && H >= 1 && H <= SF_MAXH; }
template <int DT> inline void sde_fast_launch_h(int HT, int nwg, hipStream_t st, const SdeFastArgs& f) {
  switch (HT) {
    case 1: hipLaunchKernelGGL((k_sde_eh_fast<DT, 1>), dim3(nwg), dim3(SF_NT), 0, st, f); break;
    case 2: hipLaunchKernelGGL((k_sde_eh_fast<DT, 2>), dim3(nwg), dim3(SF_NT), 0, st, f); break;
    case 3: hipLaunchKernelGGL((k_sde_eh_fast<DT, 3>), dim3(nwg), dim3(SF_NT), 0, st, f); break;
    case 4: hipLaunchKernelGGL((k_sde_eh_fast<DT, 4>), dim3(nwg), dim3(SF_NT), 0, st, f); break;
    case 5: hipLaunchKernelGGL((k_sde_eh_fast<DT, 5>), dim3(nwg), dim3(SF_NT), 0, st, f); break;
    case 6: hipLaunchKernelGGL((k_sde_eh_fast<DT, 6>), dim3(nwg), dim3(SF_NT), 0, st, f); break;
    case 7: hipLaunchKernelGGL((k_sde_eh_fast<DT, 7>), dim3(nwg), dim3(SF_NT), 0, st, f); break;
    default: hipLaunchKernelGGL((k_sde_eh_fast<DT, 8>), dim3(nwg), dim3(SF_NT), 0, st, f); break;
  }
}
inline void sde_fast_launch(int D, int H, int nwg, hipStream_t st, const SdeFastArgs& f) {
  const int DT = (D + 15) / 16, HT = (H + 15) / 16;
  switch (DT) {
    case 1: sde_fast_launch_h<1>(HT, nwg, st, f); break;
    case 2: sde_fast_launch_h<2>(HT, nwg, st, f); break;
    case 3: sde_fast_launch_h<3>(HT, nwg, st, f); break;
    default: sde_fast_launch_h<4>(HT, nwg, st, f); break;
  }
}

// the persistent form.  coop: a cooperative launch (all workgroups resident, or the call fails and the caller takes the
// launch-per-step loop) — the API's guarantee costs 25-30 us per launch; !coop: a plain launch, for grids far smaller than the
// chip, whose barrier is bounded (a workgroup that waits 50 ms for the others ends the solve with an error and the caller runs
// the loop instead).  Returns the launch's hipError_t.
template <int DT, int HT> inline hipError_t sde_persist_launch_1(int nwg, hipStream_t st, SdeFastArgs& f, bool coop) {
  void* args[] = {&f};
  if (!coop) { hipLaunchKernelGGL((k_sde_eh_fast<DT, HT, true>), dim3(nwg), dim3(SF_NT), 0, st, f); return hipGetLastError(); }
  return hipLaunchCooperativeKernel(reinterpret_cast<const void*>(k_sde_eh_fast<DT, HT, true>), dim3(nwg), dim3(SF_NT), args, 0, st);
}
template <int DT> inline hipError_t sde_persist_launch_h(int HT, int nwg, hipStream_t st, SdeFastArgs& f, bool coop) {
  switch (HT) {
    case 1: return sde_persist_launch_1<DT, 1>(nwg, st, f, coop);
    case 2: return sde_persist_launch_1<DT, 2>(nwg, st, f, coop);
    case 3: return sde_persist_launch_1<DT, 3>(nwg, st, f, coop);
    case 4: return sde_persist_launch_1<DT, 4>(nwg, st, f, coop);
    case 5: return sde_persist_launch_1<DT, 5>(nwg, st, f, coop);
    case 6: return sde_persist_launch_1<DT, 6>(nwg, st, f, coop);
    case 7: return sde_persist_launch_1<DT, 7>(nwg, st, f, coop);
    default: return sde_persist_launch_1<DT, 8>(nwg, st, f, coop);
  }
}
inline hipError_t sde_persist_launch(int D, int H, int nwg, hipStream_t st, SdeFastArgs& f, bool coop) {
  const int DT = (D + 15) / 16, HT = (H + 15) / 16;
  switch (DT) {
    case 1: return sde_persist_launch_h<1>(HT, nwg, st, f, coop);
    case 2: return sde_persist_launch_h<2>(HT, nwg, st, f, coop);
    case 3: return sde_persist_launch_h<3>(HT, nwg, st, f, coop);
    default: return sde_persist_launch_h<4>(HT, nwg, st, f, coop);
  }
}
