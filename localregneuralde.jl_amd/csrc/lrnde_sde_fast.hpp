// lrnde_sde_fast.hpp — the Lamba Euler-Heun step (src/perform_step.jl:172-206) for the MNIST-SDE shape
// (experiments/src/construct.jl:204-205: state 32, drift Dense(32=>64,tanh)->Dense(64=>32), diagonal diffusion
// Dense(32=>32)) as ONE small-latency launch.  Included by lrnde_kernels.hip inside its anonymous namespace.
//
// k_sde_step (the generic kernel) runs the step's six field evaluations through feval_tile: per evaluation it restages
// the biases, reloads the resident weight fragments from L2, and passes every intermediate (du1, L, K, tmp, ...) through
// global memory between evaluations — 28 us for 15.7 MFLOP.  Here a workgroup is four waves on 16 columns and nothing
// leaves the CU between the first load of (u, dW) and the store of u_new:
//   * the three weight matrices (24 KB) are MFMA A fragments in registers for the whole launch (wave w: hidden tile w of
//     Dense-1; waves 0,1: output tile of Dense-2; waves 2,3: output tile of the diffusion);
//   * a round = [drift Dense-1 + tanh on all four waves] barrier [drift Dense-2 on waves 0,1 || diffusion on waves 2,3]
//     barrier; the step is three rounds: (f,g)(u) -> (f,g)(tmp) -> f(K), g(utilde);
//   * the elementwise algebra of the step stays in the C-fragment registers of waves 0,1 (4 rows x 1 column per lane).
// Arithmetic: the canonical dot products (k-ordered fma chains = v_mfma_f32_16x16x4_f32 chains over the k-groups in
// order), the same activation polynomial and the same elementwise expressions as k_sde_step, so results are bit-identical
// to it and to the oracle (tests/test_gpu_parity.py::test_sde_*).  Shape: D = 32, H = 64, no time dependence.

constexpr int SF_DT = 2;   // D / 16
constexpr int SF_HT = 4;   // H / 16
constexpr int SF_NT = 256;

struct SdeFastArgs {
  const f32x4* W1p; int KG1;         // drift Dense-1 fragments [MT1p][KG1][64]
  const f32x4* W2p; int KG2p;        // drift Dense-2 fragments [MT2][KG2p][64]
  const f32x4* Wgp; int KGgp;        // diffusion (second layer of the identity+Dense form) [MT2][KGgp][64]
  const float *b1, *b2, *bg;
  int act;
  const float* u; const float* dW; float* un;   // (B, 32)
  int B;
  float dt, abstol, reltol, delta;
  double* part;      // per-workgroup fp64 sums of the squared residual (PSTRIDE doubles each)
  int* arrive;       // fixed-grid solve: arrival counter of the step's footer, or NULL
  Ctrl* rec;         //   ... and the record slot the last workgroup fills (EEst, EEst*dt)
  double n_norm;     // elements of the norm (B * 32)
  // ADAPT (lrnde_sde_solve_adaptive on this shape): the controller lives on the device.  The launch reads the control
  // block (grid position i, step length m in grid intervals, which of the two state buffers is current), forms dW from the
  // caller's path itself, and its last workgroup runs the PI controller and writes the block for the next launch.
  struct SdeCtl* ctl; const float* Wpath; float* ua; float* ub; int nfine; float t0, h;
  float gamma, qmin, qmax, beta1, beta2; int maxiters;
  lrnde_trace_row* trace; int cap_trace;
  unsigned long long* prog;  // pinned host word: (launches whose footer ran) | status << 32
  int jlaunch;
};
struct SdeCtl { int status, i, m, cur, naccept, nreject, iters, nf; float qold, eest_last; };

__device__ __forceinline__ f32x4 sf_chain(const f32x4* frag, int nkg, const f32x4* xb, int lane) {
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int kg = 0; kg < nkg; ++kg) acc = mfma4(frag[kg], xb[kg * 64 + lane], acc);
  return acc;
}

template <bool ADAPT>
__global__ __launch_bounds__(SF_NT) void k_sde_eh_fast(SdeFastArgs a) {
  int ad_i = 0, ad_m = 0, ad_cur = 0;
  if (ADAPT) {
    const SdeCtl cc = *a.ctl;   // written by the previous launch's last workgroup (kernel boundary in between)
    if (cc.status != ST_RUNNING) return;
    ad_i = cc.i; ad_m = cc.m; ad_cur = cc.cur;
    a.dt = (float)ad_m * a.h;
    a.u = ad_cur ? a.ub : a.ua;
    a.un = ad_cur ? a.ua : a.ub;
  }
  // LDS: three x tiles in B-operand layout (32 rows x 16 columns each: [kg][64 lanes] float4), the h tile (64 rows), the
  // diffusion results of waves 2,3 in C-fragment order, the reduction scratch
  __shared__ f32x4 xA[SF_DT * 64], xB[SF_DT * 64], xC[SF_DT * 64], hl[SF_HT * 64], gl[2 * 64];
  __shared__ double red[2];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int n = lane & 15, rq = lane >> 4;
  const int b0 = blockIdx.x * 16;
  const bool colok = b0 + n < a.B;
  // resident A fragments
  f32x4 w1[SF_DT], w2[SF_HT], wg[SF_DT];
#pragma unroll
  for (int kg = 0; kg < SF_DT; ++kg) w1[kg] = a.W1p[((size_t)wave * a.KG1 + kg) * 64 + lane];
  const int t = wave & 1;  // output tile of Dense-2 (waves 0,1) / of the diffusion (waves 2,3)
#pragma unroll
  for (int kg = 0; kg < SF_HT; ++kg) w2[kg] = a.W2p[((size_t)t * a.KG2p + kg) * 64 + lane];
#pragma unroll
  for (int kg = 0; kg < SF_DT; ++kg) wg[kg] = a.Wgp[((size_t)t * a.KGgp + kg) * 64 + lane];
  const f32x4 b1v = *reinterpret_cast<const f32x4*>(a.b1 + wave * 16 + rq * 4);
  const f32x4 b2v = *reinterpret_cast<const f32x4*>(a.b2 + t * 16 + rq * 4);
  const f32x4 bgv = *reinterpret_cast<const f32x4*>(a.bg + t * 16 + rq * 4);
  // this lane's four rows (16 t + 4 rq + r) of column n: state and increments (waves 0,1 own the elementwise work)
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  f32x4 u4 = zero4, w4 = zero4;
  const size_t g = (size_t)(b0 + n) * 32 + t * 16 + rq * 4;
  if (wave < 2 && colok) {
    u4 = *reinterpret_cast<const f32x4*>(a.u + g);
    if (ADAPT) {  // dW = W[i + m] - W[i], the path's own increment (the expression of k_sde_dw)
      const size_t nn = (size_t)a.B * 32;
      const f32x4 lo = *reinterpret_cast<const f32x4*>(a.Wpath + (size_t)ad_i * nn + g);
      const f32x4 hi = *reinterpret_cast<const f32x4*>(a.Wpath + (size_t)(ad_i + ad_m) * nn + g);
#pragma unroll
      for (int r = 0; r < 4; ++r) w4[r] = hi[r] - lo[r];
    } else {
      w4 = *reinterpret_cast<const f32x4*>(a.dW + g);
    }
  }
  // B-operand image of rows 16 t + 4 rq + r, column n: float4 index t*64 + r*16 + n, component rq
  auto put = [&](f32x4* x, const f32x4& v) {
    float* p = reinterpret_cast<float*>(x) + ((t * 64 + n) << 2) + rq;
#pragma unroll
    for (int r = 0; r < 4; ++r) p[r * 64] = v[r];
  };
  if (wave < 2) put(xA, u4);
  __syncthreads();
  const float dt = a.dt, hdt = dt / 2.0f, sqdt = __builtin_sqrtf(dt);

  // drift Dense-1 + activation of hidden tile `wave` on the x tile xs -> hl
  auto dense1 = [&](const f32x4* xs) {
    const f32x4 acc = sf_chain(w1, SF_DT, xs, lane);
    float* p = reinterpret_cast<float*>(hl) + ((wave * 64 + n) << 2) + rq;
#pragma unroll
    for (int r = 0; r < 4; ++r) p[r * 64] = act_apply(a.act, acc[r] + b1v[r]);
  };
  auto diffusion = [&](const f32x4* xs) {  // waves 2,3: tile t of g(xs) -> gl (C-fragment order)
    f32x4 acc = sf_chain(wg, SF_DT, xs, lane);
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] = acc[r] + bgv[r];
    gl[t * 64 + lane] = acc;
  };
  auto dense2 = [&]() {  // waves 0,1: tile t of f = W2 h + b2
    f32x4 acc = sf_chain(w2, SF_HT, hl, lane);
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] = acc[r] + b2v[r];
    return acc;
  };

  // ---- round 1: du1 = f(u), L = g(u) (:174-176) ----
  dense1(xA);
  if (wave >= 2) diffusion(xA);
  __syncthreads();
  f32x4 du1 = zero4, L = zero4, Kv = zero4;
  if (wave < 2) {
    du1 = dense2();
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
  __syncthreads();
  // ---- round 2: g(tmp), f(tmp) at t + dt (:184, :191) ----
  dense1(xB);
  if (wave >= 2) diffusion(xB);
  __syncthreads();
  f32x4 un = zero4;
  if (wave < 2) {
    const f32x4 f2 = dense2();
    const f32x4 g2 = gl[t * 64 + lane];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float gtmp2 = 0.5f * (L[r] + g2[r]);
      const float noise2 = gtmp2 * w4[r];
      un[r] = (u4[r] + hdt * (du1[r] + f2[r])) + noise2;   // :191
    }
    if (colok) *reinterpret_cast<f32x4*>(a.un + g) = un;
    put(xA, Kv);   // xA is free: every wave has read it (barrier above)
  }
  __syncthreads();
  // ---- round 3: du2 = f(K, t + dt) (:193), g(utilde, t) (:197) ----
  dense1(xA);
  if (wave >= 2) diffusion(xC);
  __syncthreads();
  double acc = 0.0;
  if (wave < 2) {
    const f32x4 du2 = dense2();
    const f32x4 g3 = gl[t * 64 + lane];
    if (colok) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
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
  if (threadIdx.x < 64) {
    const double tot = red[0] + red[1];
    if (!a.arrive) {
      if (lane == 0) { double* p = a.part + (size_t)blockIdx.x * PSTRIDE; p[0] = tot; p[1] = 0.0; p[2] = 0.0; }
      return;
    }
    // fixed-grid solve: the step's own footer — the last workgroup to arrive reduces the partials and fills the record
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
      if (!ADAPT) {
        a.rec->eest_last = eest;
        a.rec->reg_error = eest * dt;
        a.rec->status = ST_DONE;
      } else {
        // the controller of lrnde_sde_solve_adaptive's host loop, expression for expression
        SdeCtl c = *a.ctl;
        const float qoldinit = 1e-4f;
        c.nf += 3; c.eest_last = eest;
        if (eest != eest) {
          c.status = LRNDE_DT_NAN;
        } else {
          float q;
          if (eest == 0.0f) q = 1.0f / a.qmax;
          else {
            const float q11 = fastpow(eest, a.beta1);
            q = q11 / fastpow(c.qold, a.beta2);
            q = fmaxf_(1.0f / a.qmax, fminf_(1.0f / a.qmin, q / a.gamma));
          }
          const int accepted = eest <= 1.0f;
          const int ntr = c.naccept + c.nreject;
          if (a.trace && ntr < a.cap_trace) {
            lrnde_trace_row r; r.t = a.t0 + (float)c.i * a.h; r.dt = dt; r.eest = eest; r.accepted = accepted;
            a.trace[ntr] = r;
          }
          int mnew = (int)((dt / q) / a.h);
          if (mnew < 1) mnew = 1;
          if (accepted) {
            c.naccept++;
            c.qold = fmaxf_(eest, qoldinit);
            c.i += c.m;
            c.cur ^= 1;
            c.m = mnew;
            if (c.i >= a.nfine) c.status = ST_DONE;
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
        *a.ctl = c;
        __hip_atomic_store(a.prog, (unsigned long long)(unsigned)(a.jlaunch + 1) | ((unsigned long long)(unsigned)c.status << 32),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
      __hip_atomic_store(a.arrive, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}
