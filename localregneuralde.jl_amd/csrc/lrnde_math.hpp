// lrnde_math.hpp — canonical fp32 scalar math of the hot path (device + host).
//
// Every function is a fixed sequence of IEEE-754 fp32 operations (fma, mul,
// add, div, rint, integer bit ops), so the gfx950 kernels and any IEEE host
// produce the same bits.  Coefficients: oracle/gen_coeffs.py (the oracle holds
// its own copy of the same definition).  Compile with -ffp-contract=off: a
// fused multiply-add happens only where __builtin_fmaf is written.
//
// Activations stand in for NNlib.tanh / NNlib.gelu used by the reference's
// Dense layers (experiments/src/construct.jl:184, test/runtests.jl:10).
// fastlog2/fastpow2/fastpow restate DiffEqBase.fastpow used by the PI step
// controller (un-vendored upstream).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define LRNDE_HD __host__ __device__ __forceinline__
#else
#define LRNDE_HD inline
#endif

namespace lrnde {

LRNDE_HD uint32_t f2u(float x) { return __builtin_bit_cast(uint32_t, x); }
LRNDE_HD float u2f(uint32_t u) { return __builtin_bit_cast(float, u); }
LRNDE_HD float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

LRNDE_HD float expf_c(float x) {
  if (x > 87.0f) x = 87.0f;
  if (x < -87.0f) x = -87.0f;
  const float n = __builtin_rintf(x * 0x1.715476p+0f);
  float r = fma_(n, -0x1.63p-1f, x);
  r = fma_(n, 0x1.bd0106p-13f, r);
  float p = 0x1.a124f2p-13f;
  p = fma_(p, r, 0x1.6d4324p-10f);
  p = fma_(p, r, 0x1.1110e0p-7f);
  p = fma_(p, r, 0x1.5554eap-5f);
  p = fma_(p, r, 0x1.555556p-3f);
  p = fma_(p, r, 0x1.000000p-1f);
  const float r2 = r * r;
  float e = fma_(p, r2, r);
  e = e + 1.0f;
  if (e != e) return e;
  const int32_t ni = (int32_t)n;
  return u2f(f2u(e) + ((uint32_t)ni << 23));
}

LRNDE_HD float tanhf_c(float x) {
  const float ax = __builtin_fabsf(x);
  if (ax < 0.625f) {
    const float s = x * x;
    float q = -0x1.c4070cp-11f;
    q = fma_(q, s, 0x1.b159b6p-9f);
    q = fma_(q, s, -0x1.201022p-7f);
    q = fma_(q, s, 0x1.662708p-6f);
    q = fma_(q, s, -0x1.ba1a58p-5f);
    q = fma_(q, s, 0x1.111110p-3f);
    q = fma_(q, s, -0x1.555556p-2f);
    const float xs = x * s;
    return fma_(xs, q, x);
  }
  if (ax >= 9.0f) return __builtin_copysignf(1.0f, x);
  const float e = expf_c(2.0f * ax);
  const float r = 1.0f - 2.0f / (e + 1.0f);
  return __builtin_copysignf(r, x);
}

// tanhf_c without control flow: both pieces evaluated, the result selected — the same operations on the same inputs for the
// piece that counts, so the same bits.  For code that applies the activation to several independent elements per lane: the
// branches of tanhf_c keep the compiler from interleaving the elements' dependent chains (lrnde_sde_fast.hpp: 1.9k of a
// round's 3.5k cycles were four serial tanh).
LRNDE_HD float tanhf_sel(float x) {
  const float ax = __builtin_fabsf(x);
  const float s = x * x;
  float q = -0x1.c4070cp-11f;
  q = fma_(q, s, 0x1.b159b6p-9f);
  q = fma_(q, s, -0x1.201022p-7f);
  q = fma_(q, s, 0x1.662708p-6f);
  q = fma_(q, s, -0x1.ba1a58p-5f);
  q = fma_(q, s, 0x1.111110p-3f);
  q = fma_(q, s, -0x1.555556p-2f);
  const float xs = x * s;
  const float small = fma_(xs, q, x);
  // expf_c(2 ax), its clamps and its NaN exit as selects
  float y = 2.0f * ax;
  y = y > 87.0f ? 87.0f : y;
  const float n = __builtin_rintf(y * 0x1.715476p+0f);
  float r = fma_(n, -0x1.63p-1f, y);
  r = fma_(n, 0x1.bd0106p-13f, r);
  float p = 0x1.a124f2p-13f;
  p = fma_(p, r, 0x1.6d4324p-10f);
  p = fma_(p, r, 0x1.1110e0p-7f);
  p = fma_(p, r, 0x1.5554eap-5f);
  p = fma_(p, r, 0x1.555556p-3f);
  p = fma_(p, r, 0x1.000000p-1f);
  const float r2 = r * r;
  float e = fma_(p, r2, r);
  e = e + 1.0f;
  const int32_t ni = (int32_t)n;
  const float es = u2f(f2u(e) + ((uint32_t)ni << 23));
  const float ex = (e != e) ? e : es;
  const float big = __builtin_copysignf(1.0f - 2.0f / (ex + 1.0f), x);
  const float sat = __builtin_copysignf(1.0f, x);
  return ax < 0.625f ? small : (ax >= 9.0f ? sat : big);
}
LRNDE_HD float act_apply_sel(int act, float v);

LRNDE_HD float geluf_c(float x) {
  const float two_lambda = 1.5957691216057308f;
  const float x2 = x * x;
  const float inner = fma_(x2, 0.044715f, 1.0f);
  const float arg = (two_lambda * x) * inner;
  return x / (1.0f + expf_c(-arg));
}

LRNDE_HD float act_apply(int act, float v) {
  if (act == 1) return tanhf_c(v);
  if (act == 2) return geluf_c(v);
  return v;
}
LRNDE_HD float act_apply_sel(int act, float v) {
  if (act == 1) return tanhf_sel(v);
  if (act == 2) return geluf_c(v);
  return v;
}

LRNDE_HD float fastlog2(float x) {
  const float a = 0.338953f, b = 2.198599f, c = 1.523692f;
  const uint32_t ux1i = f2u(x);
  const int32_t ex = (int32_t)((ux1i & 0x7F800000u) >> 23);
  const uint32_t greater = ux1i & 0x00400000u;
  float signif, fexp;
  if (greater != 0u) {
    signif = u2f((ux1i & 0x007FFFFFu) | 0x3f000000u);
    fexp = (float)ex - 126.0f;
  } else {
    signif = u2f((ux1i & 0x007FFFFFu) | 0x3f800000u);
    fexp = (float)ex - 127.0f;
  }
  signif = signif - 1.0f;
  const float num = signif * (a * signif + b);
  return fexp + num / (signif + c);
}

LRNDE_HD float fastpow2(float x) {
  const float offset = (x < 0.0f) ? 1.0f : 0.0f;
  const float clipp = (x < -126.0f) ? -126.0f : x;
  const int32_t w = (int32_t)clipp;
  const float z = (clipp - (float)w) + offset;
  const float s = ((clipp + 121.2740575f) + 27.7280233f / (4.84252568f - z)) - 1.49012907f * z;
  const uint32_t v = (uint32_t)(8388608.0f * s);
  return u2f(v);
}

LRNDE_HD float fastpow(float x, float y) {
  if (x == 0.0f) return 0.0f;
  return fastpow2(y * fastlog2(x));
}

// Julia eps(::Float32) by bit manipulation
LRNDE_HD float eps_f(float x) {  // (selects, no branches: it sits in the step kernel's decision chain)
  const uint32_t b = f2u(x) & 0x7fffffffu;
  const uint32_t e = b >> 23;
  uint32_t r = (e - 23u) << 23;
  r = (e <= 23u) ? (1u << ((e - 1u) & 31u)) : r;
  r = (e == 0u) ? 1u : r;
  r = (e == 0xffu) ? 0x7fc00000u : r;
  return u2f(r);
}

LRNDE_HD float fminf_(float a, float b) { return __builtin_fminf(a, b); }
LRNDE_HD float fmaxf_(float a, float b) { return __builtin_fmaxf(a, b); }

// Tsit5 tableau (Float64 literals rounded to Float32 at use, src/perform_step.jl:6-8)
struct Tsit5 {
  static constexpr double C[6] = {0.161, 0.327, 0.9, 0.9800255409045097, 1.0, 1.0};
  static constexpr double A[21] = {
      0.161,
      -0.008480655492356989, 0.335480655492357,
      2.8971530571054935, -6.359448489975075, 4.3622954328695815,
      5.325864828439257, -11.748883564062828, 7.4955393428898365, -0.09249506636175525,
      5.86145544294642, -12.92096931784711, 8.159367898576159, -0.071584973281401,
      -0.028269050394068383,
      0.09646076681806523, 0.01, 0.4798896504144996, 1.379008574103742, -3.290069515436081,
      2.324710524099774};
  static constexpr double BT[7] = {-0.00178001105222577714, -0.0008164344596567469,
                                   0.007880878010261995,    -0.1447110071732629,
                                   0.5823571654525552,      -0.45808210592918697,
                                   0.015151515151515152};
  static constexpr double R[28] = {
      1.0, -2.763706197274826, 2.9132554618219126, -1.0530884977290216,
      0.0, 0.13169999999999998, -0.2234, 0.1017,
      0.0, 3.9302962368947516, -5.941033872131505, 2.490627285651253,
      0.0, -12.411077166933676, 30.33818863028232, -16.548102889244902,
      0.0, 37.50931341651104, -88.1789048947664, 47.37952196281928,
      0.0, -27.896526289197286, 65.09189467479366, -34.87065786149661,
      0.0, 1.5, -4.0, 2.5};
};

// dense-output weights b_i(theta) (OrdinaryDiffEq Tsit5 interpolant)
LRNDE_HD void tsit5_bweights(float th, float* b) {
  const float th2 = th * th;
  b[0] = th * fma_(th, fma_(th, fma_(th, (float)Tsit5::R[3], (float)Tsit5::R[2]), (float)Tsit5::R[1]),
                   (float)Tsit5::R[0]);
#pragma unroll
  for (int i = 1; i < 7; ++i)
    b[i] = th2 * fma_(th, fma_(th, (float)Tsit5::R[4 * i + 3], (float)Tsit5::R[4 * i + 2]),
                      (float)Tsit5::R[4 * i + 1]);
}

// The dense record of an accepted step in polynomial form: the interpolant's weights are quartics in theta
// without a constant term, only b_1 has a linear one (R[0] = 1) and sum_i b_i(theta) = theta, so
//   y(theta) = uprev + dt * (theta * k1 + theta^2 * (P2 + theta * (P3 + theta * P4))),  P_m = sum_{i=2..7} R[4(i-1)+m-1] * (k_i - k1):
// five arrays per step [uprev, k1, P2, P3, P4] instead of eight, formed from the DIFFERENCES k_i - k1 (the columns of R sum
// to zero with entries up to 88: formed from the k's themselves the sums would carry 1e-5 |k| of rounding).
constexpr int REC_ARRAYS = 5;  // arrays of B*D floats per recorded step
LRNDE_HD void tsit5_rec_poly(float k1, const float* k2to7, float* P) {
  float d[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) d[i] = k2to7[i] - k1;
#pragma unroll
  for (int m = 0; m < 3; ++m) {
    float s = (float)Tsit5::R[4 + m + 1] * d[0];
#pragma unroll
    for (int i = 1; i < 6; ++i) s = s + (float)Tsit5::R[4 * (i + 1) + m + 1] * d[i];
    P[m] = s;
  }
}
LRNDE_HD float tsit5_rec_eval(float y0, float k1, float P2, float P3, float P4, float th, float ddt) {
  float s = P4 * th;
  s = s + P3;
  s = s * th;
  s = s + P2;
  s = s * (th * th);
  s = s + th * k1;
  return y0 + ddt * s;
}

}  // namespace lrnde
