// lrnde_backward.hpp — device kernels of the backward pass, included by
// lrnde_kernels.hip inside its anonymous namespace.
//
//   k_vjp      per 16-column batch tile: y (given, or the Tsit5 interpolant of the stored forward
//              step — SciMLSensitivity InterpolatingAdjoint), h = act(W1[y;t]+b1), then the
//              vector-Jacobian product  dy = W1^T ((W2^T lam) .* act'(pre))  with the two transposed
//              weight matrices pre-packed in the same MFMA fragment layouts as the forward ones.
//              Also leaves y, h, dpre in scratch for the parameter gradient.
//   k_pgrad    (df/dp)^T lam: the batch-reduction GEMMs  gW1 = dpre^T [y;t], gW2 = lam^T [h;t]
//              (K = batch, output-tiled: no cross-workgroup reduction) and the bias column sums.
//   elementwise helpers for the augmented state z = [lambda (B*D); mu (P)].
// Numerics: fp32 with fixed (deterministic) summation orders; parity with the oracle is by
// tolerance (the oracle sums in a different order), see tests/test_gpu_backward.py.

__device__ __forceinline__ float act_deriv_c(int act, float pre, float h) {
  if (act == 1) return 1.0f - h * h;
  if (act == 2) {
    const float two_lambda = 1.5957691216057308f;
    const float x2 = pre * pre;
    const float a = (two_lambda * pre) * fma_(x2, 0.044715f, 1.0f);
    const float sg = 1.0f / (1.0f + expf_c(-a));
    const float da = two_lambda * fma_(x2, 3.0f * 0.044715f, 1.0f);
    return sg + pre * sg * (1.0f - sg) * da;
  }
  return 1.0f;
}

struct VjpArgs {
  ModelDev m;
  const f32x4* V1p;  // W2^T in the Dense-1 layout  [MT1p][KG1][64][4]
  const f32x4* U2p;  // W1^T in the Dense-2 layout  [MT2][KG2p][64][4]
  int B;
  float t;
  const float* y;      // (B,D) or NULL -> interpolate from the dense record
  const float* dense;  // [uprev, k1..k7] of one forward step, 8 arrays of B*D
  float theta, dense_dt;
  const float* lam;    // (B,D)
  float* dy;           // (B,D)
  float* ysc;          // scratch (B,D): y
  float* hsc;          // scratch (B,Hp): h
  float* dpsc;         // scratch (B,Hp): dpre
};

// Dense-1-shaped GEMM phase: K split by 112-row segments over waves, partial sums into pl
__device__ __forceinline__ void gemm_ksplit(const ModelDev& m, __amdgpu_buffer_rsrc_t rs, const float* xl, float* pl) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int voff = lane * 16;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  const int nseg1 = (m.KG1 + SEGK - 1) / SEGK;
  const f32x4* xp = reinterpret_cast<const f32x4*>(xl) + lane;
  for (int seg = wave; seg < nseg1; seg += NW) {
    const int kg_lo = seg * SEGK, kg_hi = min(m.KG1, kg_lo + SEGK);
    for (int mt0 = 0; mt0 < m.MT1; mt0 += TG) {
      const int tbase = mt0 * m.KG1;
      f32x4 acc[TG], aX[TG], aY[TG], bX, bY;
#pragma unroll
      for (int i = 0; i < TG; ++i) acc[i] = zero4;
#define LRNDE_BL(a, b, kg)                                                                 \
  do {                                                                                     \
    b = xp[(kg) * 64];                                                                     \
    _Pragma("unroll") for (int i = 0; i < TG; ++i) a[i] = wload(rs, voff, (tbase + i * m.KG1 + (kg)) * 1024); \
    __builtin_amdgcn_sched_barrier(0);                                                     \
  } while (0)
#define LRNDE_BM(a, b)                                                                     \
  do {                                                                                     \
    _Pragma("unroll") for (int i = 0; i < TG; ++i) acc[i] = mfma4(a[i], b, acc[i]);        \
    __builtin_amdgcn_sched_barrier(0);                                                     \
  } while (0)
      LRNDE_BL(aX, bX, kg_lo);
      int kg = kg_lo;
#pragma unroll 1
      for (; kg + 2 < kg_hi; kg += 2) {
        LRNDE_BL(aY, bY, kg + 1);
        LRNDE_BM(aX, bX);
        LRNDE_BL(aX, bX, kg + 2);
        LRNDE_BM(aY, bY);
      }
      if (kg + 1 < kg_hi) { LRNDE_BL(aY, bY, kg + 1); LRNDE_BM(aX, bX); LRNDE_BM(aY, bY); }
      else { LRNDE_BM(aX, bX); }
#undef LRNDE_BL
#undef LRNDE_BM
      f32x4* pp = reinterpret_cast<f32x4*>(pl) + ((size_t)seg * m.MT1 + mt0) * 64 + lane;
#pragma unroll
      for (int i = 0; i < TG; ++i) if (mt0 + i < m.MT1) pp[i * 64] = acc[i];
    }
  }
}

// sum of the segment partials of C-fragment element e
__device__ __forceinline__ float seg_sum(const ModelDev& m, const float* pl, int e) {
  const int nseg1 = (m.KG1 + SEGK - 1) / SEGK, pstride = m.MT1 * 256;
  float v = pl[e];
  for (int s = 1; s < nseg1; ++s) v = v + pl[(size_t)s * pstride + e];
  return v;
}

// vector-Jacobian product on one 16-column tile
template <int W> __global__ __launch_bounds__(NT) void k_vjp(VjpArgs a) {
  const ModelDev& m = a.m;
  const Smem s = carve(m);
  float* dact = reinterpret_cast<float*>(reinterpret_cast<char*>(s.bc + 1) + 16);  // [Hp*16] act'(pre), C-fragment order (extra LDS)
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int n = lane & 15, rq = lane >> 4;
  const int b0 = blockIdx.x * NB, nvalid = min(NB, a.B - b0);
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  smem_init(m, s);
  const __amdgpu_buffer_rsrc_t rsW1 = __builtin_amdgcn_make_buffer_rsrc((void*)m.W1p, 0, (((m.MT1 + TG - 1) / TG) * TG) * m.KG1 * 1024, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsV1 = __builtin_amdgcn_make_buffer_rsrc((void*)a.V1p, 0, (((m.MT1 + TG - 1) / TG) * TG) * m.KG1 * 1024, 0x00020000);
  const int KG2p = ((m.KG2 + SEGK - 1) / SEGK) * SEGK;
  const __amdgpu_buffer_rsrc_t rsU2 = __builtin_amdgcn_make_buffer_rsrc((void*)a.U2p, 0, m.MT2 * KG2p * 1024, 0x00020000);
  // ---- phase 0: y tile (given or interpolated) -> LDS x tile and scratch ----
  float bw[7];
  if (!a.y) tsit5_bweights(a.theta, bw);
  __syncthreads();
  tile_foreach<W>(m, b0, nvalid, [&](int row, int nn, bool valid, size_t g) {
    Vec<W> x = vzero<W>();
    if (valid) {
      if (a.y) {
        x = vload<W>(a.y + g);
      } else {
        const size_t nst = (size_t)a.B * m.D;
        const Vec<W> y0 = vload<W>(a.dense + g), v1 = vload<W>(a.dense + nst + g), v2 = vload<W>(a.dense + 2 * nst + g),
                     v3 = vload<W>(a.dense + 3 * nst + g), v4 = vload<W>(a.dense + 4 * nst + g),
                     v5 = vload<W>(a.dense + 5 * nst + g), v6 = vload<W>(a.dense + 6 * nst + g),
                     v7 = vload<W>(a.dense + 7 * nst + g);
#pragma unroll
        for (int h = 0; h < W; ++h) {
          float sum = v1.v[h] * bw[0] + v2.v[h] * bw[1];
          sum = sum + v3.v[h] * bw[2];
          sum = sum + v4.v[h] * bw[3];
          sum = sum + v5.v[h] * bw[4];
          sum = sum + v6.v[h] * bw[5];
          sum = sum + v7.v[h] * bw[6];
          x.v[h] = y0.v[h] + a.dense_dt * sum;
        }
      }
      vstore<W>(a.ysc + g, x);
    }
    lds_put<W>(s.xl, row, nn, x);
  });
  __syncthreads();
  // ---- phase 1: pre = W1 [y;t] + b1 ; h, act' ----
  gemm_ksplit(m, rsW1, s.xl, s.pl);
  __syncthreads();
  {
    const float* w1t = s.bias; const float* b1 = w1t + m.Hp;
    for (int e = threadIdx.x; e < m.MT1 * 256; e += NT) {
      const int r = e & 3, l = (e >> 2) & 63, mt = e >> 8;
      const int o = mt * 16 + (l >> 4) * 4 + r, nn = l & 15;
      float pre = seg_sum(m, s.pl, e);
      pre = m.td ? fma_(w1t[o], a.t, pre) : pre;
      pre = pre + b1[o];
      const float h = act_apply(m.act, pre);
      dact[e] = act_deriv_c(m.act, pre, h);
      if (nn < nvalid) a.hsc[(size_t)(b0 + nn) * m.Hp + o] = h;
    }
  }
  __syncthreads();
  // ---- phase 2: dh = W2^T lam ; dpre = dh .* act' -> h tile image + scratch ----
  tile_foreach<W>(m, b0, nvalid, [&](int row, int nn, bool valid, size_t g) {
    const Vec<W> x = valid ? vload<W>(a.lam + g) : vzero<W>();
    lds_put<W>(s.xl, row, nn, x);
  });
  __syncthreads();
  gemm_ksplit(m, rsV1, s.xl, s.pl);
  __syncthreads();
  for (int e = threadIdx.x; e < m.MT1 * 256; e += NT) {
    const int r = e & 3, l = (e >> 2) & 63, mt = e >> 8;
    const int o = mt * 16 + (l >> 4) * 4 + r, nn = l & 15;
    const float dpre = seg_sum(m, s.pl, e) * dact[e];
    s.hl[((mt * 64 + r * 16 + nn) << 2) + (l >> 4)] = dpre;
    if (nn < nvalid) a.dpsc[(size_t)(b0 + nn) * m.Hp + o] = dpre;
  }
  __syncthreads();
  // ---- phase 3: dy = W1^T dpre ----
  {
    const f32x4* hp = reinterpret_cast<const f32x4*>(s.hl) + lane;
    const int voff = lane * 16;
    const int nseg2 = (m.KG2 + SEGK - 1) / SEGK;
    for (int mt = wave; mt < m.MT2; mt += NW) {
      f32x4 tot = zero4;
      for (int sg = 0; sg < nseg2; ++sg) {
        f32x4 acc = zero4;
        f32x4 av[SEGK], bv[SEGK];
#pragma unroll
        for (int j = 0; j < SEGK; ++j) { av[j] = wload(rsU2, voff, (mt * KG2p + sg * SEGK + j) * 1024); bv[j] = hp[(sg * SEGK + j) * 64]; }
#pragma unroll
        for (int j = 0; j < SEGK; ++j) acc = mfma4(av[j], bv[j], acc);
        if (sg == 0) tot = acc;
        else { tot.x = tot.x + acc.x; tot.y = tot.y + acc.y; tot.z = tot.z + acc.z; tot.w = tot.w + acc.w; }
      }
      const int row0 = mt * 16 + rq * 4;
      if (n < nvalid) {
        float* dst = a.dy + (size_t)(b0 + n) * m.D + row0;
        if constexpr (W == 4) { if (row0 < m.D) *reinterpret_cast<f32x4*>(dst) = tot; }
        else {
#pragma unroll
          for (int r = 0; r < 4; ++r) if (row0 + r < m.D) dst[r] = tot[r];
        }
      }
    }
  }
}

// (df/dp)^T lam in the flat Lux layout: gW1 (H x (D+td)), gb1, gW2 (D x (H+td)), gb2.
// One wave per 16x16 output tile, fp32 MFMA chain over the batch (K = B); bias / time columns by
// column sums.  out = scale_old*out + result is NOT done here: the kernel overwrites gp.
struct PgradArgs {
  int D, H, Hp, td, B;
  float t;
  const float* lam;   // (B,D)
  const float* y;     // (B,D)
  const float* h;     // (B,Hp)
  const float* dpre;  // (B,Hp)
  float* gp;          // flat (P)
  int ntile1, ntile2, nt1c, nt2c;  // tiles of gW1: ceil(H/16) x ceil(D/16); gW2: ceil(D/16) x ceil(H/16)
};

__global__ __launch_bounds__(256) void k_pgrad(PgradArgs a) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int tile = blockIdx.x * 4 + wave;
  const int li = lane & 15, lk = lane >> 4;
  const size_t ob1 = (size_t)a.H * (a.D + a.td), oW2 = ob1 + a.H, ob2 = oW2 + (size_t)a.D * (a.H + a.td);
  if (tile >= a.ntile1 + a.ntile2) return;
  // C[i][j] = sum_b A[b][i] * Bm[b][j]:  gW1: A = dpre (rows o), Bm = [y, t, 1];  gW2: A = lam (rows i), Bm = [h, t, 1].
  // The two virtual columns (value t and value 1 for every sample) give the time column of the
  // weight gradient and the bias gradient from the same MFMA chain.
  const bool first = tile < a.ntile1;
  const int tt = first ? tile : tile - a.ntile1;
  const int ncol = first ? a.nt1c : a.nt2c;
  const int ti = tt / ncol, tj = tt % ncol;
  const float* A = first ? a.dpre : a.lam;
  const float* Bm = first ? a.y : a.h;
  const int lda = first ? a.Hp : a.D, ldb = first ? a.D : a.Hp;
  const int M = first ? a.H : a.D, N = first ? a.D : a.H;
  const int row = ti * 16 + li, col = tj * 16 + li;
  const bool rok = row < M, cok = col < N;
  const float cconst = (col == N) ? a.t : ((col == N + 1) ? 1.0f : 0.0f);
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const float* Ap = A + (rok ? row : 0);
  const float* Bp = Bm + (cok ? col : 0);
  constexpr int UN = 8;  // 8 MFMA k-steps (32 samples) per batch of 16 independent loads
  int b0 = 0;
  for (; b0 + 4 * UN <= a.B; b0 += 4 * UN) {
    float av[UN], bv[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const size_t b = (size_t)(b0 + 4 * u + lk);
      av[u] = Ap[b * lda];
      bv[u] = Bp[b * ldb];
    }
#pragma unroll
    for (int u = 0; u < UN; ++u)
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(rok ? av[u] : 0.f, cok ? bv[u] : cconst, acc, 0, 0, 0);
  }
  for (; b0 < a.B; b0 += 4) {
    const int b = b0 + lk;
    const bool bok = b < a.B;
    const float av = (rok && bok) ? A[(size_t)b * lda + row] : 0.f;
    const float bv = bok ? (cok ? Bm[(size_t)b * ldb + col] : cconst) : 0.f;
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc, 0, 0, 0);
  }
  // D fragment: row = ti*16 + lk*4 + r, col = tj*16 + li ; flat weight index = row + M*col
  float* gW = a.gp + (first ? (size_t)0 : oW2);
  float* gb = a.gp + (first ? ob1 : ob2);
  const int c = tj * 16 + li;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int rr = ti * 16 + lk * 4 + r;
    if (rr >= M) continue;
    if (c < N) gW[(size_t)rr + (size_t)M * c] = acc[r];
    else if (c == N) { if (a.td) gW[(size_t)rr + (size_t)M * N] = acc[r]; }
    else if (c == N + 1) gb[rr] = acc[r];
  }
}

// W2^T / W1^T in the forward fragment layouts (see k_pack)
__global__ void k_pack_t(const float* p, int D, int H, int td, int Dp, int Hp, float* V1p, float* U2p) {
  const int KG1 = Dp / 16, KG2 = ((Hp / 16 + SEGK - 1) / SEGK) * SEGK;
  const int MT1p = ((Hp / 16 + TG - 1) / TG) * TG;
  const size_t n1 = (size_t)MT1p * KG1 * 256, n2 = (size_t)(Dp / 16) * KG2 * 256;
  const size_t base2 = (size_t)H * (D + td) + H;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n1 + n2; i += (size_t)gridDim.x * blockDim.x) {
    if (i < n1) {  // V1[o][k] = W2[k][o]
      const int q = i & 3, lane = (i >> 2) & 63;
      const size_t blk = i >> 8;
      const int kg = blk % KG1, mt = blk / KG1;
      const int o = mt * 16 + (lane & 15), k = kg * 16 + q * 4 + (lane >> 4);
      V1p[i] = (o < H && k < D) ? p[base2 + (size_t)k + (size_t)D * o] : 0.f;
    } else {  // U2[i][k] = W1[k][i]
      const size_t e = i - n1;
      const int q = e & 3, lane = (e >> 2) & 63;
      const size_t blk = e >> 8;
      const int kg = blk % KG2, mt = blk / KG2;
      const int o = mt * 16 + (lane & 15), k = kg * 16 + q * 4 + (lane >> 4);
      U2p[e] = (o < D && k < H) ? p[(size_t)k + (size_t)H * o] : 0.f;
    }
  }
}

// ---- elementwise kernels on flat vectors (augmented adjoint state, cotangents) ----
struct AxArgs { float* out; const float* base; const float* k[7]; float c[7]; int nk; float dt; size_t n; };
// out = base + dt * (c0*k0 + c1*k1 + ...)   (left to right)
__global__ void k_axpy(AxArgs a) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < a.n; i += (size_t)gridDim.x * blockDim.x) {
    float s = a.c[0] * a.k[0][i];
    for (int j = 1; j < a.nk; ++j) s = s + a.c[j] * a.k[j][i];
    a.out[i] = (a.base ? a.base[i] : 0.f) + a.dt * s;
  }
}
// sum over i of ((num_i [- num2_i]) / (abstol + max(|a_i|,|b_i|)*reltol))^2 -> per-block doubles
struct NormArgs { const float* num; const float* num2; const float* sa; const float* sb; float abstol, reltol; size_t n; double* part; };
__global__ __launch_bounds__(256) void k_norm(NormArgs a) {
  __shared__ double red[4];
  double acc = 0.0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < a.n; i += (size_t)gridDim.x * blockDim.x) {
    const float v = a.num2 ? a.num[i] - a.num2[i] : a.num[i];
    const float sa = __builtin_fabsf(a.sa[i]), sb = a.sb ? __builtin_fabsf(a.sb[i]) : sa;
    const float sc = a.abstol + fmaxf_(sa, sb) * a.reltol;
    const float r = v / sc;
    acc += (double)(r * r);
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) a.part[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// cotangent seeds of the regulariser's reverse sweep (src/perform_step.jl:34-47): kbar_2..7, ubar, g6bar
struct RegSeedArgs {
  size_t n;
  const float *uprev, *u, *g6;
  const float* k[7];
  float* kb[7];  // kb[1..6] <-> k2..k7
  float *ub, *g6b;
  float dt, abstol, reltol, eest, num, den;
  int reg_type;
};
__global__ void k_reg_seed(RegSeedArgs a) {
  const float nf = (float)a.n;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < a.n; i += (size_t)gridDim.x * blockDim.x) {
    if (a.reg_type == 0) {  // reg = EEst*dt, EEst = sqrt(mean(r^2)), r = utilde / sc
      float sum = (float)Tsit5::BT[0] * a.k[0][i] + (float)Tsit5::BT[1] * a.k[1][i];
#pragma unroll
      for (int j = 2; j < 7; ++j) sum = sum + (float)Tsit5::BT[j] * a.k[j][i];
      const float ut = a.dt * sum;
      const float up = a.uprev[i], un = a.u[i];
      const float sc = a.abstol + fmaxf_(__builtin_fabsf(up), __builtin_fabsf(un)) * a.reltol;
      const float r = ut / sc;
      const float rb = (a.eest > 0.f) ? a.dt * r / (nf * a.eest) : 0.f;
      const float utb = rb / sc;
      const float scb = -rb * ut / (sc * sc);
      if (__builtin_fabsf(un) > __builtin_fabsf(up)) a.ub[i] += scb * a.reltol * (un >= 0.f ? 1.f : -1.f);
#pragma unroll
      for (int j = 1; j < 7; ++j) a.kb[j][i] += a.dt * (float)Tsit5::BT[j] * utb;
    } else if (a.den != 0.f) {  // reg = |num/(den+eps)| / 3.5068
      const float eps = 1.1920929e-7f;
      const float qv = a.num / (a.den + eps);
      const float sgn = (qv >= 0.f ? 1.f : -1.f) / 3.5068f;
      const float numb = sgn / (a.den + eps), denb = -sgn * a.num / ((a.den + eps) * (a.den + eps));
      const float dk = a.k[6][i] - a.k[5][i], du = a.u[i] - a.g6[i];
      const float ca = (a.num > 0.f) ? numb * dk / (nf * a.num) : 0.f;
      const float cb = denb * du / (nf * a.den);
      a.kb[6][i] += ca; a.kb[5][i] -= ca;
      a.ub[i] += cb; a.g6b[i] -= cb;
    }
  }
}
