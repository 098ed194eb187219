/*
 * lrnde_oracle.c — CPU restatement of the LocalRegNeuralDE.jl adaptive Tsit5
 * neural-ODE path.  TEST INFRASTRUCTURE ONLY; see lrnde_oracle.h for the
 * file:line map into /root/reference and the "parity unpinned" statement.
 *
 * Build: gcc -O2 -march=x86-64-v3 -ffp-contract=off -fopenmp -shared -fPIC
 * (-ffp-contract=off: Julia never contracts a*b+c on its own, so every fused
 * multiply-add below is an explicit fmaf()).
 */
#include "lrnde_oracle.h"

#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------- */
/* canonical fp32 math                                                        */
/* ------------------------------------------------------------------------- */

static inline uint32_t f2u(float x) { uint32_t u; memcpy(&u, &x, 4); return u; }
static inline float u2f(uint32_t u) { float x; memcpy(&x, &u, 4); return x; }

/* coefficients from oracle/gen_coeffs.py */
static const float EXP_P[6] = {0x1.000000p-1f, 0x1.555556p-3f, 0x1.5554eap-5f,
                               0x1.1110e0p-7f, 0x1.6d4324p-10f, 0x1.a124f2p-13f};
static const float TANH_Q[7] = {-0x1.555556p-2f, 0x1.111110p-3f,  -0x1.ba1a58p-5f, 0x1.662708p-6f,
                                -0x1.201022p-7f, 0x1.b159b6p-9f, -0x1.c4070cp-11f};
#define LRO_LOG2E 0x1.715476p+0f
#define LRO_LN2_HI 0x1.63p-1f
#define LRO_LN2_LO -0x1.bd0106p-13f

/* exp(x) for x clamped to [-87, 87]:  2^n * (1 + r + r^2 P(r)),  n = rint(x log2 e) */
float lro_expf(float x) {
  if (x > 87.0f) x = 87.0f;
  if (x < -87.0f) x = -87.0f;
  float n = rintf(x * LRO_LOG2E);
  float r = fmaf(n, -LRO_LN2_HI, x);
  r = fmaf(n, -LRO_LN2_LO, r);
  float p = EXP_P[5];
  p = fmaf(p, r, EXP_P[4]);
  p = fmaf(p, r, EXP_P[3]);
  p = fmaf(p, r, EXP_P[2]);
  p = fmaf(p, r, EXP_P[1]);
  p = fmaf(p, r, EXP_P[0]);
  float r2 = r * r;
  float e = fmaf(p, r2, r);
  e = e + 1.0f;
  if (e != e) return e; /* NaN in, NaN out */
  int32_t ni = (int32_t)n;
  return u2f(f2u(e) + ((uint32_t)ni << 23));
}

/* tanh: odd polynomial for |x| < 0.625, 1 - 2/(exp(2|x|)+1) up to 9, then +-1 */
float lro_tanhf(float x) {
  float ax = fabsf(x);
  if (ax < 0.625f) {
    float s = x * x;
    float q = TANH_Q[6];
    q = fmaf(q, s, TANH_Q[5]);
    q = fmaf(q, s, TANH_Q[4]);
    q = fmaf(q, s, TANH_Q[3]);
    q = fmaf(q, s, TANH_Q[2]);
    q = fmaf(q, s, TANH_Q[1]);
    q = fmaf(q, s, TANH_Q[0]);
    float xs = x * s;
    return fmaf(xs, q, x);
  }
  if (ax >= 9.0f) return copysignf(1.0f, x);
  float e = lro_expf(2.0f * ax);
  float r = 1.0f - 2.0f / (e + 1.0f);
  return copysignf(r, x);
}

/* NNlib gelu (tanh form written as x*sigmoid(2*sqrt(2/pi)*(x + 0.044715 x^3))) */
float lro_geluf(float x) {
  const float two_lambda = 1.5957691216057308f; /* 2*sqrt(2/pi) */
  float x2 = x * x;
  float inner = fmaf(x2, 0.044715f, 1.0f);
  float arg = (two_lambda * x) * inner;
  return x / (1.0f + lro_expf(-arg));
}

static inline float act_apply(int act, float v) {
  switch (act) {
    case LRO_ACT_TANH: return lro_tanhf(v);
    case LRO_ACT_GELU: return lro_geluf(v);
    default: return v;
  }
}

/* DiffEqBase fastpow: fastpow2(y * fastlog2(x)), pure fp32/int ops */
float lro_fastlog2(float x) {
  const float a = 0.338953f, b = 2.198599f, c = 1.523692f;
  uint32_t ux1i = f2u(x);
  int32_t ex = (int32_t)((ux1i & 0x7F800000u) >> 23);
  uint32_t greater = ux1i & 0x00400000u;
  float signif, fexp;
  if (greater != 0u) {
    signif = u2f((ux1i & 0x007FFFFFu) | 0x3f000000u);
    fexp = (float)ex - 126.0f;
  } else {
    signif = u2f((ux1i & 0x007FFFFFu) | 0x3f800000u);
    fexp = (float)ex - 127.0f;
  }
  signif = signif - 1.0f;
  float num = signif * (a * signif + b);
  float lg2 = fexp + num / (signif + c);
  return lg2;
}

float lro_fastpow2(float x) {
  float offset = (x < 0.0f) ? 1.0f : 0.0f;
  float clipp = (x < -126.0f) ? -126.0f : x;
  int32_t w = (int32_t)clipp; /* trunc */
  float z = (clipp - (float)w) + offset;
  float s = ((clipp + 121.2740575f) + 27.7280233f / (4.84252568f - z)) - 1.49012907f * z;
  uint32_t v = (uint32_t)(8388608.0f * s);
  return u2f(v);
}

float lro_fastpow(float x, float y) {
  if (x == 0.0f) return 0.0f;
  return lro_fastpow2(y * lro_fastlog2(x));
}

/* Tsit5 tableau, Float64 literals converted to Float32 at use (src/perform_step.jl:6-8) */
static const double TS_C[6] = {0.161, 0.327, 0.9, 0.9800255409045097, 1.0, 1.0};
static const double TS_A[21] = {
    /* a21 */ 0.161,
    /* a31 a32 */ -0.008480655492356989, 0.335480655492357,
    /* a41.. */ 2.8971530571054935, -6.359448489975075, 4.3622954328695815,
    /* a51.. */ 5.325864828439257, -11.748883564062828, 7.4955393428898365, -0.09249506636175525,
    /* a61.. */ 5.86145544294642, -12.92096931784711, 8.159367898576159, -0.071584973281401,
    -0.028269050394068383,
    /* a71.. */ 0.09646076681806523, 0.01, 0.4798896504144996, 1.379008574103742,
    -3.290069515436081, 2.324710524099774};
static const double TS_BT[7] = {-0.00178001105222577714, -0.0008164344596567469,
                                0.007880878010261995,    -0.1447110071732629,
                                0.5823571654525552,      -0.45808210592918697,
                                0.015151515151515152};
/* dense output: b1(th) = th*(r11 + th*(r12 + th*(r13 + th*r14))), bi(th) = th^2*(ri2 + th*(ri3 + th*ri4)) */
static const double TS_R[28] = {
    1.0, -2.763706197274826, 2.9132554618219126, -1.0530884977290216,
    0.0, 0.13169999999999998, -0.2234, 0.1017,
    0.0, 3.9302962368947516, -5.941033872131505, 2.490627285651253,
    0.0, -12.411077166933676, 30.33818863028232, -16.548102889244902,
    0.0, 37.50931341651104, -88.1789048947664, 47.37952196281928,
    0.0, -27.896526289197286, 65.09189467479366, -34.87065786149661,
    0.0, 1.5, -4.0, 2.5};

int lro_tsit5_tableau(double* a, double* c, double* btilde, double* r) {
  if (a) memcpy(a, TS_A, sizeof(TS_A));
  if (c) memcpy(c, TS_C, sizeof(TS_C));
  if (btilde) memcpy(btilde, TS_BT, sizeof(TS_BT));
  if (r) memcpy(r, TS_R, sizeof(TS_R));
  return 0;
}

/* ------------------------------------------------------------------------- */
/* vector field: TDChain(Dense(D+td -> H, act), Dense(H+td -> D))             */
/* src/layers/common.jl:10-40; experiments/src/construct.jl:180-189           */
/* ------------------------------------------------------------------------- */

#define LRO_KSEG 112 /* rows per fma-chain segment of the canonical dot product */

int lro_mlp_param_count(int D, int H, int td) { return H * (D + td) + H + D * (H + td) + D; }

void lro_mlp_rhs(const lro_mlp* m, const float* u, float t, int B, float* du) {
  const int D = m->D, H = m->H, td = m->time_dep ? 1 : 0;
  const float* W1 = m->p;                    /* H x (D+td), column-major */
  const float* b1 = W1 + (size_t)H * (D + td);
  const float* W2 = b1 + H;                  /* D x (H+td), column-major */
  const float* b2 = W2 + (size_t)D * (H + td);
  int nth = m->nthreads > 0 ? m->nthreads : 1;
  (void)nth;
  enum { SB = 8 }; /* samples per weight pass: same per-element fma chain, better cache reuse */
  const int nblk = (B + SB - 1) / SB;
#pragma omp parallel num_threads(nth)
  {
    float* h = (float*)malloc(sizeof(float) * (size_t)SB * (size_t)(2 * H + D));
    float* hp = h + (size_t)SB * H;
    float* yp = hp + (size_t)SB * H;
#pragma omp for schedule(static)
    for (int blk = 0; blk < nblk; ++blk) {
      const int n0 = blk * SB;
      const int ns = (B - n0) < SB ? (B - n0) : SB;
      /* layer 1: canonical dot product = fma chains over consecutive segments of LRO_KSEG rows
       * (each from 0, increasing k), segment partials added left to right; then the t column
       * (fma), then + bias */
      for (int i = 0; i < ns * H; ++i) h[i] = 0.0f;
      for (int k0 = 0; k0 < D; k0 += LRO_KSEG) {
        const int k1 = (k0 + LRO_KSEG < D) ? k0 + LRO_KSEG : D;
        for (int i = 0; i < ns * H; ++i) hp[i] = 0.0f;
        for (int k = k0; k < k1; ++k) {
          const float* w = W1 + (size_t)k * H;
          for (int s = 0; s < ns; ++s) {
            const float xv = u[(size_t)(n0 + s) * D + k];
            float* hs = hp + (size_t)s * H;
            for (int o = 0; o < H; ++o) hs[o] = fmaf(w[o], xv, hs[o]);
          }
        }
        if (k0 == 0) for (int i = 0; i < ns * H; ++i) h[i] = hp[i];
        else for (int i = 0; i < ns * H; ++i) h[i] = h[i] + hp[i];
      }
      for (int s = 0; s < ns; ++s) {
        float* hs = h + (size_t)s * H;
        if (td) {
          const float* w = W1 + (size_t)D * H;
          for (int o = 0; o < H; ++o) hs[o] = fmaf(w[o], t, hs[o]);
        }
        for (int o = 0; o < H; ++o) hs[o] = act_apply(m->act, hs[o] + b1[o]);
      }
      /* layer 2 (same canonical order over the hidden index) */
      for (int k0 = 0; k0 < H; k0 += LRO_KSEG) {
        const int k1 = (k0 + LRO_KSEG < H) ? k0 + LRO_KSEG : H;
        for (int i = 0; i < ns * D; ++i) yp[i] = 0.0f;
        for (int k = k0; k < k1; ++k) {
          const float* w = W2 + (size_t)k * D;
          for (int s = 0; s < ns; ++s) {
            const float hv = h[(size_t)s * H + k];
            float* y = yp + (size_t)s * D;
            for (int o = 0; o < D; ++o) y[o] = fmaf(w[o], hv, y[o]);
          }
        }
        for (int s = 0; s < ns; ++s) {
          float* y = du + (size_t)(n0 + s) * D;
          const float* ys = yp + (size_t)s * D;
          if (k0 == 0) for (int o = 0; o < D; ++o) y[o] = ys[o];
          else for (int o = 0; o < D; ++o) y[o] = y[o] + ys[o];
        }
      }
      for (int s = 0; s < ns; ++s) {
        float* y = du + (size_t)(n0 + s) * D;
        if (td) {
          const float* w = W2 + (size_t)H * D;
          for (int o = 0; o < D; ++o) y[o] = fmaf(w[o], t, y[o]);
        }
        for (int o = 0; o < D; ++o) y[o] = y[o] + b2[o];
      }
    }
    free(h);
  }
}

static void mlp_field_tramp(void* ctx, const float* u, float t, int B, float* du) {
  lro_mlp_rhs((const lro_mlp*)ctx, u, t, B, du);
}
void lro_mlp_as_field(const lro_mlp* m, lro_field* out) {
  out->fn = mlp_field_tramp;
  out->ctx = (void*)m;
  out->D = m->D;
}

/* ------------------------------------------------------------------------- */
/* conv vector field (experiments/src/construct.jl:213-218)                   */
/* ------------------------------------------------------------------------- */
static float bf16_round(float x);
int lro_conv_param_count(int C, int Hc) {
  return 9 * (C + 1) * Hc + 2 * Hc + 9 * (Hc + 1) * Hc + 2 * Hc + 9 * (Hc + 1) * C;
}

/* out[b][co][y][x] = sum_{ky,kx,ci} w[kx,ky,ci,co] * in[b][ci][y+1-ky][x+1-kx]  (NNlib.conv: flipped
 * kernel, pad 1), channel ci == cin is the t plane (src/layers/common.jl:10-45).  in: (B, cin, H, W). */
static void conv3x3_t(const float* in, int B, int cin, int cout, int H, int W, const float* w, float t,
                      float* out, int nth) {
  const int cint = cin + 1;
  const long plane = (long)H * W;
#pragma omp parallel for collapse(2) schedule(static) num_threads(nth)
  for (int b = 0; b < B; ++b)
    for (int co = 0; co < cout; ++co) {
      float* o = out + ((long)b * cout + co) * plane;
      for (long i = 0; i < plane; ++i) o[i] = 0.0f;
      for (int ky = 0; ky < 3; ++ky)
        for (int kx = 0; kx < 3; ++kx)
          for (int ci = 0; ci < cint; ++ci) {
            const float wv = w[kx + 3 * (ky + 3 * (ci + (long)cint * co))];
            const int dy = 1 - ky, dx = 1 - kx;
            const int y0 = dy < 0 ? -dy : 0, y1 = dy > 0 ? H - dy : H;
            const int x0 = dx < 0 ? -dx : 0, x1 = dx > 0 ? W - dx : W;
            if (ci < cin) {
              const float* ip = in + ((long)b * cin + ci) * plane;
              for (int y = y0; y < y1; ++y) {
                float* orow = o + (long)y * W;
                const float* irow = ip + (long)(y + dy) * W + dx;
                for (int x = x0; x < x1; ++x) orow[x] = fmaf(wv, irow[x], orow[x]);
              }
            } else {
              for (int y = y0; y < y1; ++y) {
                float* orow = o + (long)y * W;
                for (int x = x0; x < x1; ++x) orow[x] = fmaf(wv, t, orow[x]);
              }
            }
          }
    }
}

/* Lux BatchNorm(ch, act) over (W,H,N) per channel, in place on (B, ch, H, W) */
static void batchnorm_act_ex(float* x, int B, int ch, long plane, const float* scale, const float* bias,
                             int train, const float* rmean, const float* rvar, float eps, int act, int nth, int round_raw,
                             float* run_mean, float* run_var) {
#pragma omp parallel for schedule(static) num_threads(nth)
  for (int c = 0; c < ch; ++c) {
    float mean, inv;
    if (train) {
      double s = 0.0;
      for (int b = 0; b < B; ++b) { const float* p = x + ((long)b * ch + c) * plane; for (long i = 0; i < plane; ++i) s += (double)p[i]; }
      const double mu = s / ((double)B * (double)plane);
      double v = 0.0;
      for (int b = 0; b < B; ++b) { const float* p = x + ((long)b * ch + c) * plane; for (long i = 0; i < plane; ++i) { const double d = (double)p[i] - mu; v += d * d; } }
      v /= ((double)B * (double)plane);
      mean = (float)mu;
      inv = (float)(1.0 / sqrt(v + (double)eps));
      if (run_mean) { /* Lux training-mode BatchNorm: running statistics advance on every call (UPSTREAM-RECALL) */
        const float momentum = 0.1f, cnt = (float)((double)B * (double)plane);
        const float mcorr = momentum * cnt / (cnt - 1.0f);
        run_mean[c] = (1.0f - momentum) * run_mean[c] + momentum * (float)mu;
        run_var[c] = (1.0f - momentum) * run_var[c] + mcorr * (float)v;
      }
    } else {
      mean = rmean ? rmean[c] : 0.0f;
      inv = (float)(1.0 / sqrt((double)(rvar ? rvar[c] : 1.0f) + (double)eps));
    }
    for (int b = 0; b < B; ++b) {
      float* p = x + ((long)b * ch + c) * plane;
      for (long i = 0; i < plane; ++i) {
        const float raw = round_raw ? bf16_round(p[i]) : p[i];
        const float xn = (raw - mean) * inv;
        const float y = xn * scale[c] + bias[c];
        p[i] = act_apply(act, y);
      }
    }
  }
}
static void batchnorm_act(float* x, int B, int ch, long plane, const float* scale, const float* bias,
                          int train, const float* rmean, const float* rvar, float eps, int act, int nth) {
  batchnorm_act_ex(x, B, ch, plane, scale, bias, train, rmean, rvar, eps, act, nth, 0, NULL, NULL);
}

static float bf16_round(float x) { /* round to nearest even on the top 16 bits */
  uint32_t u; memcpy(&u, &x, 4);
  u = (u + 0x7FFFu + ((u >> 16) & 1u)) & 0xFFFF0000u;
  float r; memcpy(&r, &u, 4);
  return r;
}
static void bf16_round_array(float* x, long n) { for (long i = 0; i < n; ++i) x[i] = bf16_round(x[i]); }
/* copy of a conv weight (3x3x(cin+1)xcout) with the real input channels rounded to bf16 */
static float* bf16_weights(const float* w, int cin, int cout) {
  const long n = 9L * (cin + 1) * cout;
  float* r = (float*)malloc(sizeof(float) * (size_t)n);
  for (long i = 0; i < n; ++i) { const int ci = (int)((i / 9) % (cin + 1)); r[i] = ci < cin ? bf16_round(w[i]) : w[i]; }
  return r;
}

void lro_conv_rhs(const lro_conv* m, const float* u, float t, int B, float* du) {
  const int C = m->C, Hc = m->Hc, H = m->H, W = m->W;
  const long plane = (long)H * W;
  const int nth = m->nthreads > 0 ? m->nthreads : 1;
  const float* w1 = m->p;
  const float* g1 = w1 + 9 * (C + 1) * Hc; const float* b1 = g1 + Hc;
  const float* w2 = b1 + Hc;
  const float* g2 = w2 + 9 * (Hc + 1) * Hc; const float* b2 = g2 + Hc;
  const float* w3 = b2 + Hc;
  float* y1 = (float*)malloc(sizeof(float) * (size_t)B * Hc * plane);
  float* y2 = (float*)malloc(sizeof(float) * (size_t)B * Hc * plane);
  const float* st = m->bn_state;
  const long ny = (long)B * Hc * plane;
  float* run = m->bn_train ? m->bn_run : NULL; /* running statistics advance on every training-mode call */
  if (m->bf16) {
    /* statistics come from the fp32 conv output, the normalised copy is the bf16-rounded one */
    float* w1r = bf16_weights(w1, C, Hc);
    float* w2r = bf16_weights(w2, Hc, Hc);
    float* w3r = bf16_weights(w3, Hc, C);
    float* ur = (float*)malloc(sizeof(float) * (size_t)B * C * plane);
    for (long i = 0; i < (long)B * C * plane; ++i) ur[i] = bf16_round(u[i]);
    conv3x3_t(ur, B, C, Hc, H, W, w1r, t, y1, nth);
    free(ur); free(w1r);
    batchnorm_act_ex(y1, B, Hc, plane, g1, b1, m->bn_train, st ? st : NULL, st ? st + Hc : NULL, m->eps, m->act, nth, 1, run, run ? run + Hc : NULL);
    bf16_round_array(y1, ny);
    conv3x3_t(y1, B, Hc, Hc, H, W, w2r, t, y2, nth);
    batchnorm_act_ex(y2, B, Hc, plane, g2, b2, m->bn_train, st ? st + 2 * Hc : NULL, st ? st + 3 * Hc : NULL, m->eps, m->act, nth, 1, run ? run + 2 * Hc : NULL, run ? run + 3 * Hc : NULL);
    bf16_round_array(y2, ny);
    conv3x3_t(y2, B, Hc, C, H, W, w3r, t, du, nth);
    free(w2r); free(w3r);
  } else {
    conv3x3_t(u, B, C, Hc, H, W, w1, t, y1, nth);
    batchnorm_act_ex(y1, B, Hc, plane, g1, b1, m->bn_train, st ? st : NULL, st ? st + Hc : NULL, m->eps, m->act, nth, 0, run, run ? run + Hc : NULL);
    conv3x3_t(y1, B, Hc, Hc, H, W, w2, t, y2, nth);
    batchnorm_act_ex(y2, B, Hc, plane, g2, b2, m->bn_train, st ? st + 2 * Hc : NULL, st ? st + 3 * Hc : NULL, m->eps, m->act, nth, 0, run ? run + 2 * Hc : NULL, run ? run + 3 * Hc : NULL);
    conv3x3_t(y2, B, Hc, C, H, W, w3, t, du, nth);
  }
  free(y1); free(y2);
}

static void conv_field_tramp(void* ctx, const float* u, float t, int B, float* du) {
  lro_conv_rhs((const lro_conv*)ctx, u, t, B, du);
}
void lro_conv_as_field(const lro_conv* m, lro_field* out) {
  out->fn = conv_field_tramp;
  out->ctx = (void*)m;
  out->D = m->W * m->H * m->C;
}

/* ------------------------------------------------------------------------- */
/* norms and residuals (src/perform_step.jl:208-212)                          */
/* ------------------------------------------------------------------------- */

/* sqrt(sum(x^2)/n): fp32 squares accumulated in fp64, final value rounded to fp32 */
static float rms_from_sumsq(double acc, long n) { return (float)sqrt(acc / (double)n); }

static double sumsq_resid(const float* ut, const float* u0, const float* u1, float abstol,
                          float reltol, long n) {
  double acc = 0.0;
  for (long i = 0; i < n; ++i) {
    float sc = abstol + fmaxf(fabsf(u0[i]), fabsf(u1[i])) * reltol;
    float r = ut[i] / sc;
    float sq = r * r;
    acc += (double)sq;
  }
  return acc;
}

static double sumsq_diff(const float* a, const float* b, long n) {
  double acc = 0.0;
  for (long i = 0; i < n; ++i) {
    float d = a[i] - b[i];
    float sq = d * d;
    acc += (double)sq;
  }
  return acc;
}

/* ------------------------------------------------------------------------- */
/* one Tsit5 step (src/perform_step.jl:3-47)                                  */
/* ------------------------------------------------------------------------- */

int lro_tsit5_step(const lro_field* f, const float* uprev, const float* k1, float t, float dt,
                   float abstol, float reltol, int B, float* u, float* k7, float* ks, float* g6o,
                   float* eest, float* reg_error, float* reg_stiff) {
  double sums[3];
  const long n = (long)f->D * B;
  int rc = lro_tsit5_step_sums(f, uprev, k1, t, dt, abstol, reltol, B, u, k7, ks, g6o, sums);
  /* EEst (upstream perform_step!) and :34-38 error_estimate = EEst*dt */
  float ee = rms_from_sumsq(sums[0], n);
  if (eest) *eest = ee;
  if (reg_error) *reg_error = ee * dt;
  /* :40-47 stiffness_estimate */
  if (reg_stiff) {
    float den = rms_from_sumsq(sums[2], n);
    if (den == 0.0f) {
      *reg_stiff = 0.0f;
    } else {
      float num = rms_from_sumsq(sums[1], n);
      *reg_stiff = fabsf(num / (den + 1.1920929e-7f)) / 3.5068f;
    }
  }
  return rc;
}

/* the step itself; sums = {sum r^2 (error residual), sum (k7-k6)^2, sum (u-g6)^2} over THIS
 * array only (fp64), so that a batch-sharded caller can add the sums of all shards */
int lro_tsit5_step_sums(const lro_field* f, const float* uprev, const float* k1, float t, float dt,
                        float abstol, float reltol, int B, float* u, float* k7, float* ks,
                        float* g6o, double* sums) {
  const long n = (long)f->D * B;
  const float c1 = (float)TS_C[0], c2 = (float)TS_C[1], c3 = (float)TS_C[2], c4 = (float)TS_C[3];
  float A[21], BT[7];
  for (int i = 0; i < 21; ++i) A[i] = (float)TS_A[i];
  for (int i = 0; i < 7; ++i) BT[i] = (float)TS_BT[i];
  float* own = NULL;
  float *k2, *k3, *k4, *k5, *k6;
  if (ks) {
    k2 = ks; k3 = ks + n; k4 = ks + 2 * n; k5 = ks + 3 * n; k6 = ks + 4 * n;
  } else {
    own = (float*)malloc(sizeof(float) * 5 * (size_t)n);
    k2 = own; k3 = own + n; k4 = own + 2 * n; k5 = own + 3 * n; k6 = own + 4 * n;
  }
  float* tmp = (float*)malloc(sizeof(float) * (size_t)n);
  float* g6 = g6o ? g6o : (float*)malloc(sizeof(float) * (size_t)n);
  float* utilde = (float*)malloc(sizeof(float) * (size_t)n);

  /* :11-12  a = dt*a21; k2 = f(uprev + a*k1, t + c1*dt) */
  const float a = dt * A[0];
  for (long i = 0; i < n; ++i) tmp[i] = uprev[i] + a * k1[i];
  f->fn(f->ctx, tmp, t + c1 * dt, B, k2);
  /* :13 */
  for (long i = 0; i < n; ++i) tmp[i] = uprev[i] + dt * (A[1] * k1[i] + A[2] * k2[i]);
  f->fn(f->ctx, tmp, t + c2 * dt, B, k3);
  /* :14 */
  for (long i = 0; i < n; ++i)
    tmp[i] = uprev[i] + dt * ((A[3] * k1[i] + A[4] * k2[i]) + A[5] * k3[i]);
  f->fn(f->ctx, tmp, t + c3 * dt, B, k4);
  /* :15 */
  for (long i = 0; i < n; ++i)
    tmp[i] = uprev[i] + dt * (((A[6] * k1[i] + A[7] * k2[i]) + A[8] * k3[i]) + A[9] * k4[i]);
  f->fn(f->ctx, tmp, t + c4 * dt, B, k5);
  /* :16-17 */
  for (long i = 0; i < n; ++i)
    g6[i] = uprev[i] +
            dt * ((((A[10] * k1[i] + A[11] * k2[i]) + A[12] * k3[i]) + A[13] * k4[i]) + A[14] * k5[i]);
  f->fn(f->ctx, g6, t + dt, B, k6);
  /* :18 */
  for (long i = 0; i < n; ++i)
    u[i] = uprev[i] + dt * (((((A[15] * k1[i] + A[16] * k2[i]) + A[17] * k3[i]) + A[18] * k4[i]) +
                             A[19] * k5[i]) +
                            A[20] * k6[i]);
  /* :19-20 */
  f->fn(f->ctx, u, t + dt, B, k7);
  /* :21-27 */
  for (long i = 0; i < n; ++i)
    utilde[i] = dt * ((((((BT[0] * k1[i] + BT[1] * k2[i]) + BT[2] * k3[i]) + BT[3] * k4[i]) +
                        BT[4] * k5[i]) +
                       BT[5] * k6[i]) +
                      BT[6] * k7[i]);
  sums[0] = sumsq_resid(utilde, uprev, u, abstol, reltol, n);
  sums[1] = sumsq_diff(k7, k6, n);
  sums[2] = sumsq_diff(u, g6, n);
  free(utilde);
  if (!g6o) free(g6);
  free(tmp);
  free(own);
  return LRO_OK;
}

/* ------------------------------------------------------------------------- */
/* initial dt (OrdinaryDiffEq ode_determine_initdt, out-of-place) */
/* ------------------------------------------------------------------------- */

static int init_dt_order(const lro_field* f, const float* u0, float t0, float tend, float abstol,
                         float reltol, int B, float* f0_out, float* dt_out, float order);
int lro_init_dt(const lro_field* f, const float* u0, float t0, float tend, float abstol,
                float reltol, int B, float* f0_out, float* dt_out) {
  return init_dt_order(f, u0, t0, tend, abstol, reltol, B, f0_out, dt_out, 5.0f);
}
/* (the exponent's divisor is get_current_alg_order(alg): 5 for Tsit5, 3 for the Adams methods below) */
static int init_dt_order(const lro_field* f, const float* u0, float t0, float tend, float abstol,
                         float reltol, int B, float* f0_out, float* dt_out, float order) {
  const long n = (long)f->D * B;
  const float dtmax = tend - t0;
  float* f0 = f0_out ? f0_out : (float*)malloc(sizeof(float) * (size_t)n);
  float* u1 = (float*)malloc(sizeof(float) * (size_t)n);
  float* f1 = (float*)malloc(sizeof(float) * (size_t)n);
  f->fn(f->ctx, u0, t0, B, f0);
  double a0 = 0.0, a1 = 0.0;
  for (long i = 0; i < n; ++i) {
    float sk = abstol + fabsf(u0[i]) * reltol;
    float r0 = u0[i] / sk;
    float r1 = f0[i] / sk;
    float s0 = r0 * r0, s1 = r1 * r1;
    a0 += (double)s0;
    a1 += (double)s1;
  }
  float d0 = rms_from_sumsq(a0, n), d1 = rms_from_sumsq(a1, n);
  float dt0;
  if ((double)d0 < 1e-5 || (double)d1 < 1e-5)
    dt0 = 1e-6f;
  else
    dt0 = (d0 / d1) / 100.0f;
  dt0 = fminf(dt0, dtmax);
  for (long i = 0; i < n; ++i) u1[i] = u0[i] + dt0 * f0[i];
  f->fn(f->ctx, u1, t0 + dt0, B, f1);
  double a2 = 0.0;
  for (long i = 0; i < n; ++i) {
    float sk = abstol + fabsf(u0[i]) * reltol;
    float r2 = (f1[i] - f0[i]) / sk;
    float s2 = r2 * r2;
    a2 += (double)s2;
  }
  float d2 = rms_from_sumsq(a2, n) / dt0;
  float maxd = fmaxf(d1, d2);
  float dt1;
  if ((double)maxd <= 1e-15) {
    dt1 = fmaxf(1e-6f, dt0 * 1e-3f);
  } else {
    float l10 = (float)log10((double)maxd);
    float e = (-(2.0f + l10)) / order;
    dt1 = (float)pow(10.0, (double)e);
  }
  *dt_out = fminf(fminf(100.0f * dt0, dt1), dtmax);
  free(f1);
  free(u1);
  if (!f0_out) free(f0);
  return LRO_OK;
}

/* ------------------------------------------------------------------------- */
/* Tsit5 dense output (OrdinaryDiffEq tsit5 interpolant)          */
/* ------------------------------------------------------------------------- */

static void tsit5_bweights(float th, float b[7]) {
  float R[28];
  for (int i = 0; i < 28; ++i) R[i] = (float)TS_R[i];
  float th2 = th * th;
  /* evalpoly (Horner with muladd) */
  b[0] = th * fmaf(th, fmaf(th, fmaf(th, R[3], R[2]), R[1]), R[0]);
  for (int i = 1; i < 7; ++i) {
    const float* r = R + 4 * i;
    b[i] = th2 * fmaf(th, fmaf(th, r[3], r[2]), r[1]);
  }
}

void lro_tsit5_interp(float theta, float dt, const float* y0, const float* const k[7], long n,
                      float* out) {
  float b[7];
  tsit5_bweights(theta, b);
  for (long i = 0; i < n; ++i) {
    float s = k[0][i] * b[0] + k[1][i] * b[1];
    s = s + k[2][i] * b[2];
    s = s + k[3][i] * b[3];
    s = s + k[4][i] * b[4];
    s = s + k[5][i] * b[5];
    s = s + k[6][i] * b[6];
    out[i] = y0[i] + dt * s;
  }
}

/* ------------------------------------------------------------------------- */
/* adaptive solve (OrdinaryDiffEq solve!/loopheader!/loopfooter!) */
/* ------------------------------------------------------------------------- */

static float eps_f(float x) { /* Julia eps(::Float32) via bit ops (same code on the GPU) */
  uint32_t b = f2u(x) & 0x7fffffffu;
  uint32_t e = b >> 23;
  if (e == 0xffu) return NAN;
  if (e == 0u) return u2f(1u);
  if (e <= 23u) return u2f(1u << (e - 1u));
  return u2f((e - 23u) << 23);
}

int lro_solve(const lro_field* f, const float* u0, int B, float t0, float t1, const lro_opts* o,
              const float* saveat, int nsave, float* u_saved, float* t_saved, int cap_saved,
              lro_stats* st, lro_trace_row* trace, int cap_trace) {
  return lro_solve_ex(f, u0, B, t0, t1, o, saveat, nsave, u_saved, t_saved, cap_saved, st, trace, cap_trace,
                      NULL, 0, NULL);
}

/* dense recorder: per accepted step t, dt and the Tsit5 interpolant (what InterpolatingAdjoint keeps as uprev, k1..k7) in
 * POLYNOMIAL form, five arrays [uprev, k1, P2, P3, P4]: the weights b_i(theta) are quartics without a constant term, only
 * b_1 has a linear one (r11 = 1) and sum_i b_i(theta) = theta, so
 *   y(theta) = uprev + dt*(theta*k1 + theta^2*(P2 + theta*(P3 + theta*P4))),  P_m = sum_{i=2..7} r_im * (k_i - k1).
 * Same interpolant, better conditioned (the columns of r sum to zero with entries up to 88), and the
 * operation order below is the product's (csrc/lrnde_math.hpp tsit5_rec_poly / tsit5_rec_eval), so that the adjoint's
 * y(t) - and with it every dt of the reversed solve - can be compared bit for bit. */
#define LRO_REC_ARRAYS 5
static int dense_push(lro_dense* d, float t, float dt, const float* uprev, const float* const k[7]) {
  if (d->nsteps >= d->cap) {
    int nc = d->cap ? 2 * d->cap : 64;
    d->t = (float*)realloc(d->t, sizeof(float) * nc);
    d->dt = (float*)realloc(d->dt, sizeof(float) * nc);
    d->data = (float*)realloc(d->data, sizeof(float) * (size_t)nc * LRO_REC_ARRAYS * d->n);
    if (!d->t || !d->dt || !d->data) return LRO_CAPACITY;
    d->cap = nc;
  }
  float* dst = d->data + (size_t)d->nsteps * LRO_REC_ARRAYS * d->n;
  memcpy(dst, uprev, sizeof(float) * d->n);
  memcpy(dst + d->n, k[0], sizeof(float) * d->n);
  float R[28];
  for (int i = 0; i < 28; ++i) R[i] = (float)TS_R[i];
  for (long e = 0; e < d->n; ++e) {
    float df[6];
    for (int i = 0; i < 6; ++i) df[i] = k[i + 1][e] - k[0][e];
    for (int m = 0; m < 3; ++m) {
      float sm = R[4 + m + 1] * df[0];
      for (int i = 1; i < 6; ++i) sm = sm + R[4 * (i + 1) + m + 1] * df[i];
      dst[(size_t)(2 + m) * d->n + e] = sm;
    }
  }
  d->t[d->nsteps] = t; d->dt[d->nsteps] = dt;
  d->nsteps++;
  return LRO_OK;
}
void lro_dense_free(lro_dense* d) { free(d->t); free(d->dt); free(d->data); memset(d, 0, sizeof(*d)); }

/* u(t) from the recorded steps (Tsit5 interpolant of the step containing t, Horner form of the record) */
void lro_dense_eval(const lro_dense* d, float t, float* out) {
  int lo = 0, hi = d->nsteps - 1;
  while (lo < hi) { int mid = (lo + hi + 1) / 2; if (d->t[mid] <= t) lo = mid; else hi = mid - 1; }
  const float* y0 = d->data + (size_t)lo * LRO_REC_ARRAYS * d->n;
  const float *k1 = y0 + d->n, *P2 = y0 + 2 * d->n, *P3 = y0 + 3 * d->n, *P4 = y0 + 4 * d->n;
  const float ddt = d->dt[lo];
  const float th = (t - d->t[lo]) / ddt;
  const float th2 = th * th;
  for (long e = 0; e < d->n; ++e) {
    float sm = P4[e] * th;
    sm = sm + P3[e];
    sm = sm * th;
    sm = sm + P2[e];
    sm = sm * th2;
    sm = sm + th * k1[e];
    out[e] = y0[e] + ddt * sm;
  }
}

/* one step's record given directly in polynomial form (the Adams methods' Hermite interpolant) */
static int dense_push_poly(lro_dense* d, float t, float dt, const float* const arr[LRO_REC_ARRAYS]) {
  if (d->nsteps >= d->cap) {
    int nc = d->cap ? 2 * d->cap : 64;
    d->t = (float*)realloc(d->t, sizeof(float) * nc);
    d->dt = (float*)realloc(d->dt, sizeof(float) * nc);
    d->data = (float*)realloc(d->data, sizeof(float) * (size_t)nc * LRO_REC_ARRAYS * d->n);
    if (!d->t || !d->dt || !d->data) return LRO_CAPACITY;
    d->cap = nc;
  }
  float* dst = d->data + (size_t)d->nsteps * LRO_REC_ARRAYS * d->n;
  for (int m = 0; m < LRO_REC_ARRAYS; ++m) memcpy(dst + (size_t)m * d->n, arr[m], sizeof(float) * d->n);
  d->t[d->nsteps] = t; d->dt[d->nsteps] = dt;
  d->nsteps++;
  return LRO_OK;
}
/* y(theta) of a record [y0, k1, P2, P3, P4] (the arithmetic of lro_dense_eval) */
static void rec_poly_eval(const float* y0, const float* k1, const float* P2, const float* P3, const float* P4, float th,
                          float ddt, long n, float* out) {
  const float th2 = th * th;
  for (long e = 0; e < n; ++e) {
    float sm = P4[e] * th;
    sm = sm + P3[e];
    sm = sm * th;
    sm = sm + P2[e];
    sm = sm * th2;
    sm = sm + th * k1[e];
    out[e] = y0[e] + ddt * sm;
  }
}

/* ------------------------------------------------------------------------- */
/* VCAB3 / VCABM3: the `solver` choices "vcab3" / "vcabm3" of experiments/src/construct.jl:154-164.
 * UPSTREAM-RECALL, parity unpinned: OrdinaryDiffEq's adams_bashforth_moulton_perform_step.jl / adams_utils.jl are not in
 * /root/reference.  Restated from the published algorithm they implement — Hairer, Norsett, Wanner, "Solving ODEs I",
 * III.5 (variable step size Adams methods in terms of phi_j(n), phi*_j(n), beta_j(n), g_j(n); recurrences (5.9)/(5.10)) —
 * with what is recalled of the package: order 3, the first two steps by Bogacki-Shampine 3(2) with its own error estimate,
 * VCAB3 = three-term predictor (PE), error dt*g_3*phi_3(n+1); VCABM3 = two-term predictor, corrector dt*g_2*phi_2(n+1),
 * second evaluation (PECE), error dt*(g_3 - g_2)*phi_3(n+1) (indices from 0 as in the book; the code below counts from 1);
 * a rejected step changes nothing but dt; PI controller with the order-3 exponents beta1 = 7/30, beta2 = 2/15; initial dt
 * by ode_determine_initdt with order 3; dense output / saveat = cubic Hermite on (u_n, f_n, u_{n+1}, f_{n+1}), kept in
 * the record's polynomial form: with D = (u_{n+1} - u_n)/dt, P2 = 3D - 2 f_n - f_{n+1}, P3 = f_n + f_{n+1} - 2D, P4 = 0.
 * The arithmetic (operation order) is the product's (csrc/lrnde_adams.hpp). */
/* ------------------------------------------------------------------------- */
static int adams_solve(const lro_field* f, const float* u0, int B, float t0, float t1, const lro_opts* o,
                       const float* saveat, int nsave, float* u_saved, float* t_saved, int cap_saved,
                       lro_stats* st, lro_trace_row* trace, int cap_trace, lro_dense* dense) {
  const long n = (long)f->D * B;
  const float abstol = o->abstol, reltol = o->reltol;
  const float gamma = 0.9f, qmin = 0.2f, qmax = 10.0f, qoldinit = 1e-4f;
  const float beta1 = (float)(7.0 / 30.0), beta2 = (float)(2.0 / 15.0);
  const int moulton = (o->alg == LRO_ALG_VCABM3);
  memset(st, 0, sizeof(*st));
  if (!(t1 > t0) || B <= 0) return (st->retcode = LRO_BADARG);
  for (int i = 1; i < nsave; ++i)
    if (!(saveat[i] >= saveat[i - 1])) return (st->retcode = LRO_BADARG);
  float* buf = (float*)malloc(sizeof(float) * (size_t)n * 16);
  float *uprev = buf, *u = buf + n, *k1 = buf + 2 * n, *kprev = buf + 3 * n, *kend = buf + 4 * n, *du = buf + 5 * n;
  float *sp2 = buf + 6 * n, *s2 = buf + 7 * n, *s3 = buf + 8 * n, *P2 = buf + 9 * n, *P3 = buf + 10 * n, *P4 = buf + 11 * n;
  float *tmp = buf + 12 * n, *kb2 = buf + 13 * n, *kb3 = buf + 14 * n, *interp = buf + 15 * n;
  for (long i = 0; i < n; ++i) P4[i] = 0.0f;
  int rc = LRO_OK;
  int nsaved = 0, isave = 0, ntrace = 0;
#define PUSH_SAVE(tt, uu)                                            \
  do {                                                               \
    if (nsaved >= cap_saved) { rc = LRO_CAPACITY; goto done; }       \
    memcpy(u_saved + (size_t)nsaved * n, (uu), sizeof(float) * n);   \
    if (t_saved) t_saved[nsaved] = (tt);                             \
    nsaved++;                                                        \
  } while (0)
  memcpy(uprev, u0, sizeof(float) * n);
  float t = t0;
  const float dtmax = t1 - t0;
  const float dtmin = fmaxf(eps_f(t1), eps_f(t0));
  float dt;
  init_dt_order(f, uprev, t0, t1, abstol, reltol, B, k1, &dt, 3.0f); /* k1 = f(u0,t0) = fsalfirst */
  st->nf = 3;
  st->dt_init = dt;
  float qold = qoldinit, q11 = 1.0f, dtpropose = dt;
  float h1 = 0.0f, h2 = 0.0f;   /* the last two accepted step sizes */
  int accept = 0, iter = 0;
  if (o->save_start) PUSH_SAVE(t0, u0);
  while (isave < nsave && saveat[isave] <= t0) isave++;
  if (dense) dense->n = n;
  const float a21 = 0.5f, a32 = 0.75f, a41 = (float)(2.0 / 9.0), a42 = (float)(1.0 / 3.0), a43 = (float)(4.0 / 9.0);
  const float bt1 = (float)(5.0 / 72.0), bt2 = (float)(-1.0 / 12.0), bt3 = (float)(-1.0 / 9.0), bt4 = 0.125f;
  const float c21 = 0.5f, c22 = (float)(1.0 / 6.0), c23 = (float)(1.0 / 12.0);   /* c_{1,q} = 1/(q(q+1)) */

  while (t < t1) {
    if (iter > 0) {
      if (accept) {
        float* sw = uprev; uprev = u; u = sw;
        sw = kprev; kprev = k1; k1 = kend; kend = sw;   /* phi*_0(n-1) <- f_n; fsalfirst <- fsallast */
        sw = sp2; sp2 = s2; s2 = sw;                    /* phi*_1(n-1) */
        dt = dtpropose;
      } else {
        dt = dt / fminf(1.0f / qmin, q11 / gamma);
      }
    }
    iter++;
    dt = fminf(dtmax, dt);
    dt = fmaxf(dt, dtmin);
    dt = fminf(fabsf(dt), fabsf(t1 - t));
    if (iter > o->maxiters) { rc = LRO_MAXITERS; break; }
    if (dt != dt) { rc = LRO_DT_NAN; break; }
    if (fabsf(dt) <= fabsf(dtmin)) { rc = LRO_DT_LESS_THAN_MIN; break; }
    double acc = 0.0;
    const int nacc = st->naccept;
    /* phi*_1(n), phi*_2(n) of this attempt (a start-up step keeps phi*_1 for the next one) */
    const float b2 = nacc >= 1 ? dt / h1 : 0.0f;
    const float b3 = nacc >= 2 ? b2 * ((dt + h1) / (h1 + h2)) : 0.0f;
    if (nacc < 2) {
      /* Bogacki-Shampine 3(2) */
      const float c2s = dt * a21;
      for (long i = 0; i < n; ++i) tmp[i] = uprev[i] + c2s * k1[i];
      f->fn(f->ctx, tmp, t + 0.5f * dt, B, kb2);
      const float c3s = dt * a32;
      for (long i = 0; i < n; ++i) tmp[i] = uprev[i] + c3s * kb2[i];
      f->fn(f->ctx, tmp, t + 0.75f * dt, B, kb3);
      for (long i = 0; i < n; ++i) {
        float sm = a41 * k1[i];
        sm = sm + a42 * kb2[i];
        sm = sm + a43 * kb3[i];
        u[i] = uprev[i] + dt * sm;
      }
      f->fn(f->ctx, u, t + dt, B, kend);
      st->nf += 3;
      for (long i = 0; i < n; ++i) {
        float sm = bt1 * k1[i];
        sm = sm + bt2 * kb2[i];
        sm = sm + bt3 * kb3[i];
        sm = sm + bt4 * kend[i];
        const float ut = 0.0f + dt * sm;
        const float sc = abstol + fmaxf(fabsf(uprev[i]), fabsf(u[i])) * reltol;
        const float r = ut / sc;
        acc += (double)(r * r);
      }
      if (nacc == 1)
        for (long i = 0; i < n; ++i) { const float p2 = k1[i] - kprev[i]; s2[i] = b2 * p2; }
    } else {
      const float r1 = dt / (dt + h1);
      const float g2 = c21;
      const float g3 = c21 - r1 * c22;
      const float c32 = c22 - r1 * c23;
      const float r2 = dt / ((dt + h1) + h2);
      const float g4 = g3 - r2 * c32;
      for (long i = 0; i < n; ++i) {
        const float p2 = k1[i] - kprev[i];
        const float v2 = b2 * p2;
        const float p3 = p2 - sp2[i];
        const float v3 = b3 * p3;
        s2[i] = v2; s3[i] = v3;
        float sm = k1[i] + g2 * v2;
        if (!moulton) sm = sm + g3 * v3;
        u[i] = uprev[i] + dt * sm;
      }
      f->fn(f->ctx, u, t + dt, B, du);
      st->nf += 1;
      const float cu = dt * g3, ce = moulton ? dt * (g4 - g3) : dt * g4;
      for (long i = 0; i < n; ++i) {
        const float q2 = du[i] - k1[i];
        const float q3 = q2 - s2[i];
        const float q4 = q3 - s3[i];
        if (moulton) u[i] = u[i] + cu * q3;
        const float ut = ce * q4;
        const float sc = abstol + fmaxf(fabsf(uprev[i]), fabsf(u[i])) * reltol;
        const float r = ut / sc;
        acc += (double)(r * r);
      }
      if (moulton) { f->fn(f->ctx, u, t + dt, B, kend); st->nf += 1; }
      else memcpy(kend, du, sizeof(float) * n);
    }
    const float eest = rms_from_sumsq(acc, n);
    if (eest != eest) { rc = LRO_DT_NAN; st->eest_last = eest; break; }
    float ttmp = t + dt;
    float q;
    if (eest == 0.0f) {
      q = 1.0f / qmax;
    } else {
      if (o->exact_pow) {
        q11 = (float)pow((double)eest, (double)beta1);
        q = q11 / (float)pow((double)qold, (double)beta2);
      } else {
        q11 = lro_fastpow(eest, beta1);
        q = q11 / lro_fastpow(qold, beta2);
      }
      q = fmaxf(1.0f / qmax, fminf(1.0f / qmin, q / gamma));
    }
    accept = (eest <= 1.0f);
    if (trace && ntrace < cap_trace) {
      trace[ntrace].t = t; trace[ntrace].dt = dt; trace[ntrace].eest = eest;
      trace[ntrace].accepted = accept; ntrace++;
    }
    st->eest_last = eest;
    if (accept) {
      st->naccept++;
      float dtnew = dt / q;
      qold = fmaxf(eest, qoldinit);
      float tprev = t;
      t = (fabsf(ttmp - t1) < 100.0f * eps_f(fmaxf(fabsf(t), fabsf(t1)))) ? t1 : ttmp;
      dtpropose = fmaxf(fminf(dtmax, dtnew), fmaxf(eps_f(t), dtmin));
      h2 = h1; h1 = dt;
      int need_poly = dense != NULL;
      for (int i = isave; i < nsave && saveat[i] <= t; ++i) if (saveat[i] != t) need_poly = 1;
      if (need_poly)
        for (long i = 0; i < n; ++i) {
          const float d = (u[i] - uprev[i]) / dt;
          P2[i] = (3.0f * d - 2.0f * k1[i]) - kend[i];
          P3[i] = (k1[i] + kend[i]) - 2.0f * d;
        }
      if (dense) {
        const float* arr[LRO_REC_ARRAYS] = {uprev, k1, P2, P3, P4};
        if ((rc = dense_push_poly(dense, tprev, dt, arr)) != LRO_OK) goto done;
      }
      while (isave < nsave && saveat[isave] <= t) {
        float ts = saveat[isave++];
        if (ts != t) {
          rec_poly_eval(uprev, k1, P2, P3, P4, (ts - tprev) / dt, dt, n, interp);
          PUSH_SAVE(ts, interp);
        } else {
          PUSH_SAVE(t, u);
        }
      }
      if (o->save_everystep) PUSH_SAVE(t, u);
    } else {
      st->nreject++;
    }
  }
done:
  st->retcode = rc;
  st->iters = iter;
  st->nsaved = nsaved;
  st->t_final = t;
  st->dt_final = dt;
  free(buf);
  return rc;
#undef PUSH_SAVE
}

/* tstops: ascending times strictly inside (t0,t1) the integrator must hit exactly
 * (the discrete cotangent times of the adjoint solve); dense: optional recorder */
int lro_solve_ex(const lro_field* f, const float* u0, int B, float t0, float t1, const lro_opts* o,
                 const float* saveat, int nsave, float* u_saved, float* t_saved, int cap_saved,
                 lro_stats* st, lro_trace_row* trace, int cap_trace, const float* tstops, int ntstops,
                 lro_dense* dense) {
  if (o->alg != LRO_ALG_TSIT5) {
    if (o->alg != LRO_ALG_VCAB3 && o->alg != LRO_ALG_VCABM3) { memset(st, 0, sizeof(*st)); return (st->retcode = LRO_BADARG); }
    if (ntstops > 0) { memset(st, 0, sizeof(*st)); return (st->retcode = LRO_BADARG); }
    return adams_solve(f, u0, B, t0, t1, o, saveat, nsave, u_saved, t_saved, cap_saved, st, trace, cap_trace, dense);
  }
  const long n = (long)f->D * B;
  const float abstol = o->abstol, reltol = o->reltol;
  const float gamma = 0.9f, qmin = 0.2f, qmax = 10.0f, qoldinit = 1e-4f;
  const float beta1 = (float)(7.0 / 50.0), beta2 = (float)(2.0 / 25.0);
  memset(st, 0, sizeof(*st));
  if (!(t1 > t0) || B <= 0) return (st->retcode = LRO_BADARG);
  for (int i = 1; i < nsave; ++i)
    if (!(saveat[i] >= saveat[i - 1])) return (st->retcode = LRO_BADARG);

  float* buf = (float*)malloc(sizeof(float) * (size_t)n * 10);
  float* uprev = buf;
  float* u = buf + n;
  float* k1 = buf + 2 * n;
  float* ks = buf + 3 * n; /* k2..k6 */
  float* k7 = buf + 8 * n;
  float* interp = buf + 9 * n;
  int rc = LRO_OK;
  int nsaved = 0, isave = 0, ntrace = 0;

#define PUSH_SAVE(tt, uu)                                            \
  do {                                                               \
    if (nsaved >= cap_saved) { rc = LRO_CAPACITY; goto done; }       \
    memcpy(u_saved + (size_t)nsaved * n, (uu), sizeof(float) * n);   \
    if (t_saved) t_saved[nsaved] = (tt);                             \
    nsaved++;                                                        \
  } while (0)

  memcpy(uprev, u0, sizeof(float) * n);
  float t = t0;
  const float dtmax = t1 - t0;
  const float dtmin = fmaxf(eps_f(t1), eps_f(t0));
  float dt;
  lro_init_dt(f, uprev, t0, t1, abstol, reltol, B, k1, &dt); /* k1 = f(u0,t0) = fsalfirst */
  st->nf = 3; /* initdt: 2, initialize!: 1 */
  st->dt_init = dt;
  float qold = qoldinit, q11 = 1.0f, dtpropose = dt;
  int accept = 0, iter = 0;
  if (o->save_start) PUSH_SAVE(t0, u0);
  while (isave < nsave && saveat[isave] <= t0) isave++; /* points at/before t0 are the start */
  int istop = 0;
  while (istop < ntstops && tstops[istop] <= t0) istop++;
  if (dense) { dense->n = n; }

  while (t < t1) {
    while (istop < ntstops && tstops[istop] <= t) istop++;      /* handle_tstop!: pop reached stops */
    const float tstop = (istop < ntstops && tstops[istop] < t1) ? tstops[istop] : t1;
    /* loopheader! */
    if (iter > 0) {
      if (accept) {
        float* sw = uprev; uprev = u; u = sw;   /* uprev <- u */
        sw = k1; k1 = k7; k7 = sw;              /* fsalfirst <- fsallast */
        dt = dtpropose;
      } else {
        dt = dt / fminf(1.0f / qmin, q11 / gamma);
      }
    }
    iter++;
    dt = fminf(dtmax, dt);
    dt = fmaxf(dt, dtmin);
    dt = fminf(fabsf(dt), fabsf(tstop - t));
    /* check_error! */
    if (iter > o->maxiters) { rc = LRO_MAXITERS; break; }
    if (dt != dt) { rc = LRO_DT_NAN; break; }
    if (fabsf(dt) <= fabsf(dtmin)) { rc = LRO_DT_LESS_THAN_MIN; break; }
    /* perform_step! */
    float eest;
    lro_tsit5_step(f, uprev, k1, t, dt, abstol, reltol, B, u, k7, ks, NULL, &eest, NULL, NULL);
    st->nf += 6;
    if (eest != eest) { rc = LRO_DT_NAN; st->eest_last = eest; break; } /* unstable_check */
    /* loopfooter!: PI controller */
    float ttmp = t + dt;
    float q;
    if (eest == 0.0f) {
      q = 1.0f / qmax;
    } else {
      if (o->exact_pow) {
        q11 = (float)pow((double)eest, (double)beta1);
        q = q11 / (float)pow((double)qold, (double)beta2);
      } else {
        q11 = lro_fastpow(eest, beta1);
        q = q11 / lro_fastpow(qold, beta2);
      }
      q = fmaxf(1.0f / qmax, fminf(1.0f / qmin, q / gamma));
    }
    accept = (eest <= 1.0f);
    if (trace && ntrace < cap_trace) {
      trace[ntrace].t = t; trace[ntrace].dt = dt; trace[ntrace].eest = eest;
      trace[ntrace].accepted = accept; ntrace++;
    }
    st->eest_last = eest;
    if (accept) {
      st->naccept++;
      float dtnew = dt / q;
      qold = fmaxf(eest, qoldinit);
      float tprev = t;
      /* (magnitudes: this loop also runs the adjoint in reversed time s = -t <= 0, where a signed max picks the time nearer
         zero and its eps; the reference's adjoint runs t from t2 down to t0 with positive times) */
      t = (fabsf(ttmp - tstop) < 100.0f * eps_f(fmaxf(fabsf(t), fabsf(tstop)))) ? tstop : ttmp;
      dtpropose = fmaxf(fminf(dtmax, dtnew), fmaxf(eps_f(t), dtmin));
      if (dense) {
        const float* kk7[7] = {k1, ks, ks + n, ks + 2 * n, ks + 3 * n, ks + 4 * n, k7};
        if ((rc = dense_push(dense, tprev, dt, uprev, kk7)) != LRO_OK) goto done;
      }
      /* savevalues! */
      while (isave < nsave && saveat[isave] <= t) {
        float ts = saveat[isave++];
        if (ts != t) {
          float theta = (ts - tprev) / dt;
          const float* kk[7] = {k1, ks, ks + n, ks + 2 * n, ks + 3 * n, ks + 4 * n, k7};
          lro_tsit5_interp(theta, dt, uprev, kk, n, interp);
          PUSH_SAVE(ts, interp);
        } else {
          PUSH_SAVE(t, u);
        }
      }
      if (o->save_everystep) PUSH_SAVE(t, u);
    } else {
      st->nreject++;
    }
  }
done:
  st->retcode = rc;
  st->iters = iter;
  st->nsaved = nsaved;
  st->t_final = t;
  st->dt_final = dt;
  free(buf);
  return rc;
#undef PUSH_SAVE
}

/* ------------------------------------------------------------------------- */
/* NeuralODE layer forward (src/layers/neural_ode.jl:56-100)                   */
/* ------------------------------------------------------------------------- */

int lro_node_forward(const lro_field* f, const float* x, int B, float t0, float t2,
                     const lro_opts* o, int mode, int reg_type, float t1_or_rand, float* u_end,
                     float* reg_val, int* nfe, lro_stats* st, float* t1_used) {
  const long n = (long)f->D * B;
  lro_opts oo = *o;
  int rc;
  *reg_val = 0.0f;
  if (t1_used) *t1_used = t2;
  if (mode == LRO_MODE_NONE) { /* _vanilla_node_fallback :56-60 */
    float sv[1] = {t2};
    float ts[2];
    oo.save_everystep = 0;
    float* us = (float*)malloc(sizeof(float) * (size_t)n * 2);
    rc = lro_solve(f, x, B, t0, t2, &oo, sv, 1, us, ts, 2, st, NULL, 0);
    if (st->nsaved > 0) memcpy(u_end, us + (size_t)(st->nsaved - 1) * n, sizeof(float) * n);
    *nfe = st->nf;
    free(us);
    return rc;
  }
  float t1;
  float* u1 = (float*)malloc(sizeof(float) * (size_t)n);
  if (mode == LRO_MODE_UNBIASED) { /* :68-84, saveat = [t1, t2] */
    t1 = t1_or_rand;
    float sv[2] = {t1, t2};
    float ts[3];
    oo.save_everystep = 0;
    float* us = (float*)malloc(sizeof(float) * (size_t)n * 3);
    rc = lro_solve(f, x, B, t0, t2, &oo, sv, 2, us, ts, 3, st, NULL, 0);
    if (rc != LRO_OK) { free(us); free(u1); return rc; }
    int i1 = oo.save_start ? 1 : 0;
    memcpy(u1, us + (size_t)i1 * n, sizeof(float) * n);     /* sol(t1): the saved knot */
    memcpy(u_end, us + (size_t)(st->nsaved - 1) * n, sizeof(float) * n);
    free(us);
  } else { /* :88-100 biased, saveat = [] => every accepted step */
    oo.save_everystep = 1;
    int cap = oo.maxiters + 2;
    if (cap > 4096) cap = 4096;
    float* us = (float*)malloc(sizeof(float) * (size_t)n * cap);
    float* ts = (float*)malloc(sizeof(float) * (size_t)cap);
    rc = lro_solve(f, x, B, t0, t2, &oo, NULL, 0, us, ts, cap, st, NULL, 0);
    if (rc != LRO_OK || st->nsaved < 2) {
      free(us); free(ts); free(u1);
      return rc != LRO_OK ? rc : LRO_BADARG;
    }
    int m = st->nsaved - 1;                 /* rand(rng, sol.t[1:end-1]) */
    int idx = (int)(t1_or_rand * (float)m);
    if (idx >= m) idx = m - 1;
    if (idx < 0) idx = 0;
    t1 = ts[idx];
    memcpy(u1, us + (size_t)idx * n, sizeof(float) * n);
    memcpy(u_end, us + (size_t)(st->nsaved - 1) * n, sizeof(float) * n);
    free(us); free(ts);
  }
  if (t1_used) *t1_used = t1;
  /* _get_ode_integrator :33-38 => init on (t1,t2): initdt (2 f) + fsalfirst (1 f) */
  float* k1 = (float*)malloc(sizeof(float) * (size_t)n);
  float* ub = (float*)malloc(sizeof(float) * (size_t)n);
  float* k7 = (float*)malloc(sizeof(float) * (size_t)n);
  float dtl, ee, re, rs;
  lro_init_dt(f, u1, t1, t2, oo.abstol, oo.reltol, B, k1, &dtl);
  /* _perform_step :77 */
  lro_tsit5_step(f, u1, k1, t1, dtl, oo.abstol, oo.reltol, B, ub, k7, NULL, NULL, &ee, &re, &rs);
  *reg_val = (reg_type == LRO_REG_STIFFNESS_ESTIMATE) ? rs : re;
  *nfe = st->nf + (6 + 3); /* :79 with src/perform_step.jl:31 */
  free(k7); free(ub); free(k1); free(u1);
  return LRO_OK;
}

/* ------------------------------------------------------------------------- */
/* adaptive Euler-Heun SDE step, diagonal noise (src/perform_step.jl:172-206)  */
/* ------------------------------------------------------------------------- */

int lro_euler_heun_step(const lro_field* fd, const lro_field* gd, const float* uprev,
                        const float* dW, float t, float dt, float abstol, float reltol, float delta,
                        int B, float* u, float* eest, float* reg_val) {
  const long n = (long)fd->D * B;
  float* w = (float*)malloc(sizeof(float) * (size_t)n * 8);
  float *du1 = w, *K = w + n, *L = w + 2 * n, *tmp = w + 3 * n, *g2 = w + 4 * n, *f2 = w + 5 * n,
        *du2 = w + 6 * n, *ut = w + 7 * n;
  const float sqdt = sqrtf(dt);
  fd->fn(fd->ctx, uprev, t, B, du1);                               /* :174 */
  for (long i = 0; i < n; ++i) K[i] = uprev[i] + dt * du1[i];      /* :175 */
  gd->fn(gd->ctx, uprev, t, B, L);                                 /* :176 */
  for (long i = 0; i < n; ++i) tmp[i] = K[i] + L[i] * dW[i];       /* :179,183 */
  gd->fn(gd->ctx, tmp, t + dt, B, g2);                             /* :184 */
  fd->fn(fd->ctx, tmp, t + dt, B, f2);                             /* :191 */
  const float hdt = dt / 2.0f;
  for (long i = 0; i < n; ++i) {
    float gtmp2 = 0.5f * (L[i] + g2[i]);
    float noise2 = gtmp2 * dW[i];
    u[i] = (uprev[i] + hdt * (du1[i] + f2[i])) + noise2;
  }
  fd->fn(fd->ctx, K, t + dt, B, du2);                              /* :193 */
  for (long i = 0; i < n; ++i) ut[i] = uprev[i] + L[i] * sqdt;     /* :196 */
  gd->fn(gd->ctx, ut, t, B, g2);                                   /* :197 */
  double acc = 0.0;
  for (long i = 0; i < n; ++i) {
    float Ed = (dt * (du2[i] - du1[i])) / 2.0f;                    /* :194 */
    float ggp = (g2[i] - L[i]) / sqdt;
    float En = (ggp * (dW[i] * dW[i])) / 2.0f;                     /* :198 */
    float sc = abstol + fmaxf(fabsf(uprev[i]), fabsf(u[i])) * reltol;
    float r = (delta * Ed + En) / sc;                              /* :214-216 */
    float sq = r * r;
    acc += (double)sq;
  }
  float ee = rms_from_sumsq(acc, n);
  if (eest) *eest = ee;
  if (reg_val) *reg_val = ee * dt;                                 /* :205 */
  free(w);
  return LRO_OK;
}

/* ------------------------------------------------------------------------- */
/* Milstein step, diagonal noise, Ito (src/perform_step.jl:108-170)            */
/* J = get_iterated_I (diagonal noise: dW.^2 ./ 2, UPSTREAM-RECALL) - |dt|/2.   */
/* du2 = f(K, t+dt) and En of the reference only feed a `tmp` that the next     */
/* line overwrites (:163-166): they do not reach u or EEst and are not          */
/* evaluated here.  EEst = rms((u - uprev) / (abstol + max(|uprev|,|u|) reltol)).*/
/* ------------------------------------------------------------------------- */
int lro_rkmil_step(const lro_field* fd, const lro_field* gd, const float* uprev, const float* dW, float t, float dt,
                   float abstol, float reltol, int B, float* u, float* eest, float* reg_val) {
  const long n = (long)fd->D * B;
  float* w = (float*)malloc(sizeof(float) * (size_t)n * 5);
  float *du1 = w, *L = w + n, *K = w + 2 * n, *tmp = w + 3 * n, *gt = w + 4 * n;
  const float sqdt = sqrtf(dt);
  fd->fn(fd->ctx, uprev, t, B, du1);                               /* :130 */
  gd->fn(gd->ctx, uprev, t, B, L);                                 /* :131 */
  for (long i = 0; i < n; ++i) { K[i] = uprev[i] + dt * du1[i]; tmp[i] = K[i] + sqdt * L[i]; } /* :133, :136-137 (Ito) */
  gd->fn(gd->ctx, tmp, t, B, gt);                                  /* :138 */
  const float hdt = 0.5f * fabsf(dt);
  double acc = 0.0;
  for (long i = 0; i < n; ++i) {
    float J = (0.5f * dW[i]) * dW[i] - hdt;                        /* :117, :122 */
    float Dgj = (gt[i] - L[i]) / sqdt;                             /* :139 */
    u[i] = (K[i] + L[i] * dW[i]) + Dgj * J;                        /* :141 */
    float sc = abstol + fmaxf(fabsf(uprev[i]), fabsf(u[i])) * reltol;
    float r = (u[i] - uprev[i]) / sc;                              /* :166, :218-220 */
    float sq = r * r;
    acc += (double)sq;
  }
  float ee = rms_from_sumsq(acc, n);
  if (eest) *eest = ee;
  if (reg_val) *reg_val = ee * dt;                                 /* :169 */
  free(w);
  return LRO_OK;
}

/* ------------------------------------------------------------------------- */
/* four-stage SRI step, diagonal noise (src/perform_step.jl:49-106).           */
/* Every line below is the reference's expression evaluated elementwise in its  */
/* own association order (Julia: left to right, scalar * scalar first in        */
/* `dt * a021 * k1`, `x^3` = (x*x)*x); the tableau is the caller's.             */
/* ------------------------------------------------------------------------- */
int lro_sri_step(const lro_field* fd, const lro_field* gd, const lro_sri_tableau* T, const float* uprev, const float* dW,
                 const float* dZ, float t, float dt, float abstol, float reltol, float delta, int B, float* u,
                 float* eest, float* reg_val) {
  const long n = (long)fd->D * B;
  float* w = (float*)malloc(sizeof(float) * (size_t)n * 13);
  float *k1 = w, *k2 = w + n, *k3 = w + 2 * n, *k4 = w + 3 * n, *g1 = w + 4 * n, *g2 = w + 5 * n, *g3 = w + 6 * n,
        *g4 = w + 7 * n, *H0 = w + 8 * n, *H1 = w + 9 * n, *chi1 = w + 10 * n, *chi2 = w + 11 * n, *chi3 = w + 12 * n;
  const float sqdt = sqrtf(fabsf(dt)), sqrt3 = sqrtf(3.0f);
  const float two_sqdt = 2.0f * sqdt, six_dt = 6.0f * dt;
  for (long i = 0; i < n; ++i) {                                                      /* :57-60 */
    chi1[i] = (dW[i] * dW[i] - fabsf(dt)) / two_sqdt;
    chi2[i] = (dW[i] + dZ[i] / sqrt3) / 2.0f;
    chi3[i] = ((dW[i] * dW[i]) * dW[i] - (3.0f * dW[i]) * dt) / six_dt;
  }
  fd->fn(fd->ctx, uprev, t, B, k1);                                                   /* :62 */
  gd->fn(gd->ctx, uprev, t + T->c11 * dt, B, g1);                                     /* :63 */
  { const float da = dt * T->a021, db = dt * T->a121, sb = sqdt * T->b121;
    for (long i = 0; i < n; ++i) {                                                    /* :65-66 */
      H0[i] = (uprev[i] + da * k1[i]) + (T->b021 * chi2[i]) * g1[i];
      H1[i] = (uprev[i] + db * k1[i]) + sb * g1[i];
    } }
  fd->fn(fd->ctx, H0, t + T->c02 * dt, B, k2);                                        /* :68 */
  gd->fn(gd->ctx, H1, t + T->c12 * dt, B, g2);                                        /* :69 */
  for (long i = 0; i < n; ++i) {                                                      /* :71-72 */
    H0[i] = (uprev[i] + dt * (T->a031 * k1[i] + T->a032 * k2[i])) + chi2[i] * (T->b031 * g1[i] + T->b032 * g2[i]);
    H1[i] = (uprev[i] + dt * (T->a131 * k1[i] + T->a132 * k2[i])) + sqdt * (T->b131 * g1[i] + T->b132 * g2[i]);
  }
  fd->fn(fd->ctx, H0, t + T->c03 * dt, B, k3);                                        /* :74 */
  gd->fn(gd->ctx, H1, t + T->c13 * dt, B, g3);                                        /* :75 */
  for (long i = 0; i < n; ++i) {                                                      /* :77-82 */
    H0[i] = (uprev[i] + dt * ((T->a041 * k1[i] + T->a042 * k2[i]) + T->a043 * k3[i])) +
            chi2[i] * ((T->b041 * g1[i] + T->b042 * g2[i]) + T->b043 * g3[i]);
    H1[i] = (uprev[i] + dt * ((T->a141 * k1[i] + T->a142 * k2[i]) + T->a143 * k3[i])) +
            sqdt * ((T->b141 * g1[i] + T->b142 * g2[i]) + T->b143 * g3[i]);
  }
  fd->fn(fd->ctx, H0, t + T->c04 * dt, B, k4);                                        /* :84 */
  gd->fn(gd->ctx, H1, t + T->c14 * dt, B, g4);                                        /* :85 */
  double acc = 0.0;
  for (long i = 0; i < n; ++i) {
    const float s3 = ((T->beta31 * g1[i] + T->beta32 * g2[i]) + T->beta33 * g3[i]) + T->beta34 * g4[i];
    const float s4 = ((T->beta41 * g1[i] + T->beta42 * g2[i]) + T->beta43 * g3[i]) + T->beta44 * g4[i];
    const float E2 = chi2[i] * s3 + chi3[i] * s4;                                     /* :87-88 */
    const float sa = ((T->alpha1 * k1[i] + T->alpha2 * k2[i]) + T->alpha3 * k3[i]) + T->alpha4 * k4[i];
    const float s1 = ((T->beta11 * g1[i] + T->beta12 * g2[i]) + T->beta13 * g3[i]) + T->beta14 * g4[i];
    const float s2 = ((T->beta21 * g1[i] + T->beta22 * g2[i]) + T->beta23 * g3[i]) + T->beta24 * g4[i];
    u[i] = (((uprev[i] + dt * sa) + E2) + dW[i] * s1) + chi1[i] * s2;                 /* :90-94 */
    const float E1 = dt * (((k1[i] + k2[i]) + k3[i]) + k4[i]);                        /* :98 */
    const float sc = abstol + fmaxf(fabsf(uprev[i]), fabsf(u[i])) * reltol;
    const float r = (delta * E1 + E2) / sc;                                           /* :100-103, :214-216 */
    const float sq = r * r;
    acc += (double)sq;
  }
  const float ee = rms_from_sumsq(acc, n);
  if (eest) *eest = ee;
  if (reg_val) *reg_val = ee * dt;                                                    /* :105 */
  free(w);
  return LRO_OK;
}

/* ========================================================================= */
/* Backward pass: continuous adjoint of the solve            */
/* (SciMLSensitivity InterpolatingAdjoint(autojacvec=ZygoteVJP()), un-vendored) */
/* and the reverse sweep of the local regularisation step                      */
/* (Zygote through src/perform_step.jl:3-47 with k1, dt, uprev constant —      */
/* src/layers/neural_ode.jl:40, src/utils.jl:60).                              */
/* ========================================================================= */

static float act_deriv(int act, float pre, float h) {
  if (act == LRO_ACT_TANH) return 1.0f - h * h;
  if (act == LRO_ACT_GELU) { /* d/dx [x * sigmoid(a)], a = 2*lambda*x*(1 + 0.044715 x^2) */
    const float two_lambda = 1.5957691216057308f;
    float x2 = pre * pre;
    float a = (two_lambda * pre) * fmaf(x2, 0.044715f, 1.0f);
    float sg = 1.0f / (1.0f + lro_expf(-a));
    float da = two_lambda * fmaf(x2, 3.0f * 0.044715f, 1.0f);
    return sg + pre * sg * (1.0f - sg) * da;
  }
  return 1.0f;
}

/* ------------------------------------------------------------------------- */
/* conv field: vector-Jacobian product (ZygoteVJP of the dudt closure)          */
/* ------------------------------------------------------------------------- */
/* din[b][ci][y][x] += sum_{co,ky,kx} w[kx,ky,ci,co] * g[b][co][y-1+ky][x-1+kx]   (real channels only);
 * dw[kx,ky,ci,co] += sum_{b,y',x'} g[b][co][y'][x'] * in[b][ci][y'+1-ky][x'+1-kx]  (t channel: in = t inside) */
static void conv3x3_t_bwd(const float* in, const float* g, int B, int cin, int cout, int H, int W, const float* w,
                          float t, float* din, double* dw, int nth) {
  const int cint = cin + 1;
  const long plane = (long)H * W;
  if (din) {
#pragma omp parallel for collapse(2) schedule(static) num_threads(nth)
    for (int b = 0; b < B; ++b)
      for (int ci = 0; ci < cin; ++ci) {
        float* d = din + ((long)b * cin + ci) * plane;
        for (long i = 0; i < plane; ++i) d[i] = 0.0f;
        for (int co = 0; co < cout; ++co) {
          const float* gp = g + ((long)b * cout + co) * plane;
          for (int ky = 0; ky < 3; ++ky)
            for (int kx = 0; kx < 3; ++kx) {
              const float wv = w[kx + 3 * (ky + 3 * (ci + (long)cint * co))];
              /* y' = y - 1 + ky in [0,H) */
              const int dy = ky - 1, dx = kx - 1;
              const int y0 = dy < 0 ? -dy : 0, y1 = dy > 0 ? H - dy : H;
              const int x0 = dx < 0 ? -dx : 0, x1 = dx > 0 ? W - dx : W;
              for (int y = y0; y < y1; ++y) {
                float* drow = d + (long)y * W;
                const float* grow = gp + (long)(y + dy) * W + dx;
                for (int x = x0; x < x1; ++x) drow[x] = fmaf(wv, grow[x], drow[x]);
              }
            }
        }
      }
  }
  if (dw) {
#pragma omp parallel for collapse(2) schedule(static) num_threads(nth)
    for (int co = 0; co < cout; ++co)
      for (int ci = 0; ci < cint; ++ci)
        for (int ky = 0; ky < 3; ++ky)
          for (int kx = 0; kx < 3; ++kx) {
            const int dy = 1 - ky, dx = 1 - kx; /* in[y'+dy][x'+dx] */
            const int y0 = dy < 0 ? -dy : 0, y1 = dy > 0 ? H - dy : H;
            const int x0 = dx < 0 ? -dx : 0, x1 = dx > 0 ? W - dx : W;
            double acc = 0.0;
            for (int b = 0; b < B; ++b) {
              const float* gp = g + ((long)b * cout + co) * plane;
              const float* ip = ci < cin ? in + ((long)b * cin + ci) * plane : NULL;
              for (int y = y0; y < y1; ++y)
                for (int x = x0; x < x1; ++x)
                  acc += (double)gp[(long)y * W + x] * (ip ? (double)ip[(long)(y + dy) * W + x + dx] : (double)t);
            }
            dw[kx + 3 * (ky + 3 * (ci + (long)cint * co))] += acc;
          }
  }
}

/* BatchNorm(ch, act) backward.  a: raw conv output (B,ch,plane), dh: cotangent of the activated output (in/out:
 * replaced by the cotangent of a).  dscale/dbias accumulate. */
static void batchnorm_act_bwd(const float* a, float* dh, int B, int ch, long plane, const float* scale, const float* bias,
                              int train, const float* rmean, const float* rvar, float eps, int act, double* dscale,
                              double* dbias, int nth) {
  const double N = (double)B * (double)plane;
#pragma omp parallel for schedule(static) num_threads(nth)
  for (int c = 0; c < ch; ++c) {
    float mean, inv;
    if (train) {
      double s = 0.0;
      for (int b = 0; b < B; ++b) { const float* p = a + ((long)b * ch + c) * plane; for (long i = 0; i < plane; ++i) s += (double)p[i]; }
      const double mu = s / N;
      double v = 0.0;
      for (int b = 0; b < B; ++b) { const float* p = a + ((long)b * ch + c) * plane; for (long i = 0; i < plane; ++i) { const double d = (double)p[i] - mu; v += d * d; } }
      v /= N;
      mean = (float)mu; inv = (float)(1.0 / sqrt(v + (double)eps));
    } else {
      mean = rmean ? rmean[c] : 0.0f;
      inv = (float)(1.0 / sqrt((double)(rvar ? rvar[c] : 1.0f) + (double)eps));
    }
    double s1 = 0.0, s2 = 0.0;
    for (int b = 0; b < B; ++b) {
      const float* p = a + ((long)b * ch + c) * plane;
      float* d = dh + ((long)b * ch + c) * plane;
      for (long i = 0; i < plane; ++i) {
        const float xn = (p[i] - mean) * inv;
        const float z = xn * scale[c] + bias[c];
        const float hh = act_apply(act, z);
        const float dz = d[i] * act_deriv(act, z, hh);
        d[i] = dz;
        s1 += (double)dz; s2 += (double)dz * (double)xn;
      }
    }
    if (dscale) dscale[c] += s2;
    if (dbias) dbias[c] += s1;
    const float g = scale[c];
    const float m1 = (float)(s1 / N), m2 = (float)(s2 / N);
    for (int b = 0; b < B; ++b) {
      const float* p = a + ((long)b * ch + c) * plane;
      float* d = dh + ((long)b * ch + c) * plane;
      for (long i = 0; i < plane; ++i) {
        if (train) { const float xn = (p[i] - mean) * inv; d[i] = (inv * g) * ((d[i] - m1) - xn * m2); }
        else d[i] = d[i] * (g * inv);
      }
    }
  }
}

void lro_conv_vjp(const lro_conv* m, const float* y, float t, const float* lam, int B, float* dy, float* gp) {
  const int C = m->C, Hc = m->Hc, H = m->H, W = m->W;
  const long plane = (long)H * W;
  const int nth = m->nthreads > 0 ? m->nthreads : 1;
  const int P = lro_conv_param_count(C, Hc);
  const long o_w1 = 0, o_g1 = o_w1 + 9L * (C + 1) * Hc, o_b1 = o_g1 + Hc, o_w2 = o_b1 + Hc,
             o_g2 = o_w2 + 9L * (Hc + 1) * Hc, o_b2 = o_g2 + Hc, o_w3 = o_b2 + Hc;
  const float* p = m->p;
  const float* st = m->bn_state;
  const size_t ny = (size_t)B * Hc * plane;
  float* a1 = (float*)malloc(sizeof(float) * ny); float* h1 = (float*)malloc(sizeof(float) * ny);
  float* a2 = (float*)malloc(sizeof(float) * ny); float* h2 = (float*)malloc(sizeof(float) * ny);
  float* d2 = (float*)malloc(sizeof(float) * ny); float* d1 = (float*)malloc(sizeof(float) * ny);
  double* g = gp ? (double*)calloc((size_t)P, sizeof(double)) : NULL;
  /* forward, keeping raw and activated hidden tensors */
  conv3x3_t(y, B, C, Hc, H, W, p + o_w1, t, a1, nth);
  memcpy(h1, a1, sizeof(float) * ny);
  batchnorm_act(h1, B, Hc, plane, p + o_g1, p + o_b1, m->bn_train, st ? st : NULL, st ? st + Hc : NULL, m->eps, m->act, nth);
  conv3x3_t(h1, B, Hc, Hc, H, W, p + o_w2, t, a2, nth);
  memcpy(h2, a2, sizeof(float) * ny);
  batchnorm_act(h2, B, Hc, plane, p + o_g2, p + o_b2, m->bn_train, st ? st + 2 * Hc : NULL, st ? st + 3 * Hc : NULL, m->eps, m->act, nth);
  /* backward */
  conv3x3_t_bwd(h2, lam, B, Hc, C, H, W, p + o_w3, t, d2, g ? g + o_w3 : NULL, nth);
  batchnorm_act_bwd(a2, d2, B, Hc, plane, p + o_g2, p + o_b2, m->bn_train, st ? st + 2 * Hc : NULL, st ? st + 3 * Hc : NULL,
                    m->eps, m->act, g ? g + o_g2 : NULL, g ? g + o_b2 : NULL, nth);
  conv3x3_t_bwd(h1, d2, B, Hc, Hc, H, W, p + o_w2, t, d1, g ? g + o_w2 : NULL, nth);
  batchnorm_act_bwd(a1, d1, B, Hc, plane, p + o_g1, p + o_b1, m->bn_train, st ? st : NULL, st ? st + Hc : NULL, m->eps,
                    m->act, g ? g + o_g1 : NULL, g ? g + o_b1 : NULL, nth);
  conv3x3_t_bwd(y, d1, B, C, Hc, H, W, p + o_w1, t, dy, g ? g + o_w1 : NULL, nth);
  if (gp) { for (int i = 0; i < P; ++i) gp[i] += (float)g[i]; free(g); } /* accumulates, like lro_mlp_vjp */
  free(a1); free(h1); free(a2); free(h2); free(d2); free(d1);
}

/* hidden pre-activation with the canonical dot product (same as lro_mlp_rhs layer 1) */
static void mlp_hidden_pre(const lro_mlp* m, const float* x, float t, float* pre, float* tmp) {
  const int D = m->D, H = m->H, td = m->time_dep ? 1 : 0;
  const float* W1 = m->p;
  const float* b1 = W1 + (size_t)H * (D + td);
  for (int o = 0; o < H; ++o) pre[o] = 0.0f;
  for (int k0 = 0; k0 < D; k0 += LRO_KSEG) {
    const int k1 = (k0 + LRO_KSEG < D) ? k0 + LRO_KSEG : D;
    for (int o = 0; o < H; ++o) tmp[o] = 0.0f;
    for (int k = k0; k < k1; ++k) {
      const float* w = W1 + (size_t)k * H;
      for (int o = 0; o < H; ++o) tmp[o] = fmaf(w[o], x[k], tmp[o]);
    }
    if (k0 == 0) for (int o = 0; o < H; ++o) pre[o] = tmp[o];
    else for (int o = 0; o < H; ++o) pre[o] = pre[o] + tmp[o];
  }
  if (td) { const float* w = W1 + (size_t)D * H; for (int o = 0; o < H; ++o) pre[o] = fmaf(w[o], t, pre[o]); }
  for (int o = 0; o < H; ++o) pre[o] = pre[o] + b1[o];
}

/* dy = (df/dy)^T lam  (B x D);  gp += (df/dp)^T lam  (flat Lux layout; gp may be NULL).
 * What Zygote.pullback(dudt, y, p, t) returns (ZygoteVJP of src/layers/neural_ode.jl:45-48).  The reference
 * leaves every summation order here to BLAS; this restatement fixes them to the product's (csrc/lrnde_backward.hpp), as
 * lro_mlp_rhs does for the forward, because at the experiments' tolerances the adjoint's embedded error estimate is
 * rounding noise of exactly these sums and its step sequence can only be compared bit for bit:
 *   pre, h          : the forward's canonical dot product (mlp_hidden_pre)
 *   dh = W2^T lam   : canonical dot product over the D rows (fma chains over segments of 112, partials left to right)
 *   dy = W1^T dpre  : the same over the H hidden units
 *   gW = sum_b ...  : the batch in blocks of 32 samples dealt round-robin to 4 chains (one per wave of the GEMM tile), each
 *                     an fma chain from 0 in increasing sample order, then ((c0 + c1) + c2) + c3; the time column and the
 *                     bias gradient ride the same chains as the virtual operand columns [t, 1].
 * The result does not depend on nthreads. */
static void pgrad_chain4(const float* A, long lda, const float* Bm, long ldb, float bconst, int B, float* out) {
  /* out = sum_b A[b*lda] * (Bm ? Bm[b*ldb] : bconst) in the order described above */
  float c[4] = {0.0f, 0.0f, 0.0f, 0.0f};
  const int nblk = (B + 31) / 32;
  for (int w = 0; w < 4; ++w)
    for (int blk = w; blk < nblk; blk += 4) {
      const int b1 = (blk * 32 + 32 < B) ? blk * 32 + 32 : B;
      for (int b = blk * 32; b < b1; ++b) c[w] = fmaf(A[(size_t)b * lda], Bm ? Bm[(size_t)b * ldb] : bconst, c[w]);
    }
  *out = ((c[0] + c[1]) + c[2]) + c[3];
}
void lro_mlp_vjp(const lro_mlp* m, const float* y, float t, const float* lam, int B, float* dy, float* gp) {
  const int D = m->D, H = m->H, td = m->time_dep ? 1 : 0;
  const float* W1 = m->p;
  const float* W2 = W1 + (size_t)H * (D + td) + H;
  const size_t oW1 = 0, ob1 = (size_t)H * (D + td), oW2 = ob1 + H, ob2 = oW2 + (size_t)D * (H + td);
  int nth = m->nthreads > 0 ? m->nthreads : 1;
  (void)nth;
  float* hs = gp ? (float*)malloc(sizeof(float) * 2 * (size_t)B * H) : NULL;   /* h, dpre of every sample (the GEMMs' operands) */
  float* dps = hs ? hs + (size_t)B * H : NULL;
#pragma omp parallel num_threads(nth)
  {
    float* pre = (float*)malloc(sizeof(float) * 4 * (size_t)H);
    float *tmp = pre + H, *h = pre + 2 * H, *dpre = pre + 3 * H;
#pragma omp for schedule(static)
    for (int n = 0; n < B; ++n) {
      const float* yy = y + (size_t)n * D;
      const float* ll = lam + (size_t)n * D;
      float* dd = dy + (size_t)n * D;
      mlp_hidden_pre(m, yy, t, pre, tmp);
      for (int o = 0; o < H; ++o) h[o] = act_apply(m->act, pre[o]);
      /* dh = W2[:, :H]^T lam ; dpre = dh .* act'(pre) */
      for (int o = 0; o < H; ++o) {
        const float* w = W2 + (size_t)o * D;
        float tot = 0.0f;
        for (int k0 = 0; k0 < D; k0 += LRO_KSEG) {
          const int k1 = (k0 + LRO_KSEG < D) ? k0 + LRO_KSEG : D;
          float acc = 0.0f;
          for (int i = k0; i < k1; ++i) acc = fmaf(w[i], ll[i], acc);
          tot = (k0 == 0) ? acc : tot + acc;
        }
        dpre[o] = tot * act_deriv(m->act, pre[o], h[o]);
      }
      /* dy = W1[:, :D]^T dpre */
      for (int k = 0; k < D; ++k) {
        const float* w = W1 + (size_t)k * H;
        float tot = 0.0f;
        for (int o0 = 0; o0 < H; o0 += LRO_KSEG) {
          const int o1 = (o0 + LRO_KSEG < H) ? o0 + LRO_KSEG : H;
          float acc = 0.0f;
          for (int o = o0; o < o1; ++o) acc = fmaf(w[o], dpre[o], acc);
          tot = (o0 == 0) ? acc : tot + acc;
        }
        dd[k] = tot;
      }
      if (hs) { memcpy(hs + (size_t)n * H, h, sizeof(float) * H); memcpy(dps + (size_t)n * H, dpre, sizeof(float) * H); }
    }
    free(pre);
    if (gp) {
      /* gW1 = dpre^T [y, t, 1] (H x (D+td), b1);  gW2 = lam^T [h, t, 1] (D x (H+td), b2): every entry its own chain set */
#pragma omp for schedule(static)
      for (int k = 0; k < D + 2; ++k)
        for (int o = 0; o < H; ++o) {
          float v;
          if (k < D) { pgrad_chain4(dps + o, H, y + k, D, 0.0f, B, &v); gp[oW1 + (size_t)k * H + o] += v; }
          else if (k == D) { if (td) { pgrad_chain4(dps + o, H, NULL, 0, t, B, &v); gp[oW1 + (size_t)D * H + o] += v; } }
          else { pgrad_chain4(dps + o, H, NULL, 0, 1.0f, B, &v); gp[ob1 + o] += v; }
        }
#pragma omp for schedule(static)
      for (int k = 0; k < H + 2; ++k)
        for (int i = 0; i < D; ++i) {
          float v;
          if (k < H) { pgrad_chain4(lam + i, D, hs + k, H, 0.0f, B, &v); gp[oW2 + (size_t)k * D + i] += v; }
          else if (k == H) { if (td) { pgrad_chain4(lam + i, D, NULL, 0, t, B, &v); gp[oW2 + (size_t)H * D + i] += v; } }
          else { pgrad_chain4(lam + i, D, NULL, 0, 1.0f, B, &v); gp[ob2 + i] += v; }
        }
    }
  }
  free(hs);
}

/* adjoint field in reversed time s = -t:  z = [lambda (n); mu (P)],  dz/ds = [J^T lambda; (df/dp)^T lambda] at y(t) */
/* a vector field with its vector-Jacobian product (the backward drivers are field-agnostic) */
typedef void (*lro_vjp_fn)(const void* ctx, const float* y, float t, const float* lam, int B, float* dy, float* gp);
typedef struct { lro_field field; lro_vjp_fn vjp; const void* ctx; int P; } lro_diff_field;
static void mlp_vjp_tramp(const void* ctx, const float* y, float t, const float* lam, int B, float* dy, float* gp) {
  lro_mlp_vjp((const lro_mlp*)ctx, y, t, lam, B, dy, gp);
}
static void conv_vjp_tramp(const void* ctx, const float* y, float t, const float* lam, int B, float* dy, float* gp) {
  lro_conv_vjp((const lro_conv*)ctx, y, t, lam, B, dy, gp);
}
static void diff_from_mlp(const lro_mlp* m, lro_diff_field* d) {
  lro_mlp_as_field(m, &d->field); d->vjp = mlp_vjp_tramp; d->ctx = m; d->P = lro_mlp_param_count(m->D, m->H, m->time_dep ? 1 : 0);
}
static void diff_from_conv(const lro_conv* m, lro_diff_field* d) {
  lro_conv_as_field(m, &d->field); d->vjp = conv_vjp_tramp; d->ctx = m; d->P = lro_conv_param_count(m->C, m->Hc);
}
typedef struct { const lro_diff_field* df; const lro_dense* dense; int B; long n; int P; float* y; } adj_ctx;
static void adjoint_field(void* vctx, const float* z, float s, int B1, float* dz) {
  (void)B1;
  adj_ctx* c = (adj_ctx*)vctx;
  const float t = -s;
  lro_dense_eval(c->dense, t, c->y);
  memset(dz + c->n, 0, sizeof(float) * (size_t)c->P);
  c->df->vjp(c->df->ctx, c->y, t, z, c->B, dz, dz + c->n);
}

/* gradient of the local regularisation value w.r.t. p (reverse sweep through one Tsit5 step) */
static int step_reg_grad_generic(const lro_diff_field* df, const float* uprev, const float* k1, float t, float dt,
                                 float abstol, float reltol, int B, int reg_type, float* gp, float* reg_val) {
  const int D = df->field.D;
  const long n = (long)D * B;
  const int P = df->P;
  lro_field f = df->field;
  float A[21], BT[7];
  for (int i = 0; i < 21; ++i) A[i] = (float)TS_A[i];
  for (int i = 0; i < 7; ++i) BT[i] = (float)TS_BT[i];
  const float cs[6] = {(float)TS_C[0], (float)TS_C[1], (float)TS_C[2], (float)TS_C[3], 1.0f, 1.0f};
  /* forward, keeping stage inputs x2..x7 (x7 = u) and k2..k7 */
  float* xs = (float*)malloc(sizeof(float) * 6 * (size_t)n);   /* x2..x7 */
  float* kk = (float*)malloc(sizeof(float) * 7 * (size_t)n);   /* k1..k7 */
  memcpy(kk, k1, sizeof(float) * n);
  for (int s = 2; s <= 7; ++s) {
    float* x = xs + (size_t)(s - 2) * n;
    const int off = (s - 2) * (s - 1) / 2;
    for (long i = 0; i < n; ++i) {
      float v;
      if (s == 2) { const float a = dt * A[0]; v = uprev[i] + a * kk[i]; }
      else { float sum = A[off] * kk[i] + A[off + 1] * kk[n + i];
             for (int j = 2; j < s - 1; ++j) sum = sum + A[off + j] * kk[(size_t)j * n + i];
             v = uprev[i] + dt * sum; }
      x[i] = v;
    }
    f.fn(f.ctx, x, t + cs[s - 2] * dt, B, kk + (size_t)(s - 1) * n);
  }
  const float* u = xs + 5 * (size_t)n; const float* g6 = xs + 4 * (size_t)n;
  const float* k6 = kk + 5 * (size_t)n; const float* k7 = kk + 6 * (size_t)n;
  /* cotangents */
  float* kb = (float*)calloc(7 * (size_t)n, sizeof(float));    /* kbar_1..7 (kbar_1 unused: constant) */
  float* ub = (float*)calloc((size_t)n, sizeof(float));
  float* g6b = (float*)calloc((size_t)n, sizeof(float));
  float rv;
  if (reg_type == LRO_REG_ERROR_ESTIMATE) {
    double acc = 0.0;
    float* ut = (float*)malloc(sizeof(float) * (size_t)n);
    for (long i = 0; i < n; ++i) {
      float sum = BT[0] * kk[i] + BT[1] * kk[n + i];
      for (int j = 2; j < 7; ++j) sum = sum + BT[j] * kk[(size_t)j * n + i];
      ut[i] = dt * sum;
      const float sc = abstol + fmaxf(fabsf(uprev[i]), fabsf(u[i])) * reltol;
      const float r = ut[i] / sc; acc += (double)(r * r);
    }
    const float ee = rms_from_sumsq(acc, n);
    rv = ee * dt;
    /* d(ee*dt)/dr_i = dt * r_i / (n * ee) */
    for (long i = 0; i < n; ++i) {
      const float sc = abstol + fmaxf(fabsf(uprev[i]), fabsf(u[i])) * reltol;
      const float r = ut[i] / sc;
      const float rb = (ee > 0.0f) ? dt * r / ((float)n * ee) : 0.0f;
      const float utb = rb / sc;
      const float scb = -rb * ut[i] / (sc * sc);
      if (fabsf(u[i]) > fabsf(uprev[i])) ub[i] += scb * reltol * (u[i] >= 0.0f ? 1.0f : -1.0f);
      for (int j = 1; j < 7; ++j) kb[(size_t)j * n + i] += BT[j] * (dt * utb);  /* Zygote's order: the pullback of `dt * (...)` first */
    }
    free(ut);
  } else {
    const double sd = sumsq_diff(u, g6, n), sn = sumsq_diff(k7, k6, n);
    const float den = rms_from_sumsq(sd, n), num = rms_from_sumsq(sn, n);
    if (den == 0.0f) { rv = 0.0f; }
    else {
      const float eps = 1.1920929e-7f;
      const float qv = num / (den + eps);
      rv = fabsf(qv) / 3.5068f;
      const float sgn = (qv >= 0.0f ? 1.0f : -1.0f) / 3.5068f;
      const float numb = sgn / (den + eps), denb = -sgn * num / ((den + eps) * (den + eps));
      for (long i = 0; i < n; ++i) {
        const float dk = k7[i] - k6[i], du_ = u[i] - g6[i];
        const float a = (num > 0.0f) ? numb * dk / ((float)n * num) : 0.0f;
        const float b = denb * du_ / ((float)n * den);
        kb[6 * (size_t)n + i] += a; kb[5 * (size_t)n + i] -= a;
        ub[i] += b; g6b[i] -= b;
      }
    }
  }
  memset(gp, 0, sizeof(float) * (size_t)P);
  float* xb = (float*)malloc(sizeof(float) * (size_t)n);
  /* reverse through stages 7..2: k_s = f(x_s); x_s = uprev + dt * sum_{j<s} a_sj k_j */
  for (int s = 7; s >= 2; --s) {
    const float* x = xs + (size_t)(s - 2) * n;
    df->vjp(df->ctx, x, t + cs[s - 2] * dt, kb + (size_t)(s - 1) * n, B, xb, gp);
    if (s == 7) for (long i = 0; i < n; ++i) xb[i] += ub[i];
    if (s == 6) for (long i = 0; i < n; ++i) xb[i] += g6b[i];
    const int off = (s - 2) * (s - 1) / 2;
    for (int j = 1; j < s - 1; ++j)  /* k_{j+1}, j = 0 is the constant k1 */
      for (long i = 0; i < n; ++i) kb[(size_t)j * n + i] += A[off + j] * (dt * xb[i]);
    if (s == 7) { /* u also feeds the next accumulation: xbar_7 already includes ub */ }
  }
  if (reg_val) *reg_val = rv;
  free(xb); free(g6b); free(ub); free(kb); free(kk); free(xs);
  return LRO_OK;
}

/* full backward of `loss = <du_end, sol.u[end]> + w_reg * reg_val` for the NeuralODE layer */
static int node_backward_generic(const lro_diff_field* df, const float* x, int B, float t0, float t2, const lro_opts* o, int mode,
                                 int reg_type, float t1_or_rand, const float* du_end, float w_reg, float* dx, float* dp,
                                 lro_stats* st_fwd, lro_stats* st_bwd, lro_trace_row* btrace, int cap_btrace) {
  const int D = df->field.D;
  const long n = (long)D * B;
  const int P = df->P;
  lro_field f = df->field;
  lro_opts oo = *o;
  lro_dense dense; memset(&dense, 0, sizeof(dense));
  int rc;
  /* forward re-solve with dense storage (what the adjoint interpolates) */
  float t1 = t2;
  float* u1 = (float*)malloc(sizeof(float) * (size_t)n);
  float* tst = NULL; int ntst = 0;
  if (mode == LRO_MODE_BIASED) {
    oo.save_everystep = 1;
    int cap = oo.maxiters + 2; if (cap > 4096) cap = 4096;
    float* us = (float*)malloc(sizeof(float) * (size_t)n * cap);
    float* ts = (float*)malloc(sizeof(float) * (size_t)cap);
    rc = lro_solve_ex(&f, x, B, t0, t2, &oo, NULL, 0, us, ts, cap, st_fwd, NULL, 0, NULL, 0, &dense);
    if (rc == LRO_OK && st_fwd->nsaved >= 2) {
      const int mm = st_fwd->nsaved - 1;
      int idx = (int)(t1_or_rand * (float)mm); if (idx >= mm) idx = mm - 1; if (idx < 0) idx = 0;
      t1 = ts[idx]; memcpy(u1, us + (size_t)idx * n, sizeof(float) * n);
      ntst = st_fwd->nsaved - 1;                         /* every saved time is a cotangent time */
      tst = (float*)malloc(sizeof(float) * (size_t)(ntst > 0 ? ntst : 1));
      for (int i = 0; i < ntst; ++i) tst[i] = ts[i];
    } else if (rc == LRO_OK) rc = LRO_BADARG;
    free(us); free(ts);
  } else {
    float sv[2]; int nsv;
    if (mode == LRO_MODE_UNBIASED) { t1 = t1_or_rand; sv[0] = t1; sv[1] = t2; nsv = 2; }
    else { sv[0] = t2; nsv = 1; }
    float ts[3]; float* us = (float*)malloc(sizeof(float) * (size_t)n * 3);
    oo.save_everystep = 0;
    rc = lro_solve_ex(&f, x, B, t0, t2, &oo, sv, nsv, us, ts, 3, st_fwd, NULL, 0, NULL, 0, &dense);
    if (rc == LRO_OK && mode == LRO_MODE_UNBIASED) {
      memcpy(u1, us + (size_t)(oo.save_start ? 1 : 0) * n, sizeof(float) * n);
      if (t1 > t0 && t1 < t2) { tst = (float*)malloc(sizeof(float)); tst[0] = t1; ntst = 1; }
    }
    free(us);
  }
  if (rc != LRO_OK) { free(u1); free(tst); lro_dense_free(&dense); return rc; }
  /* adjoint solve in s = -t from -t2 to -t0 on z = [lambda; mu] */
  const long N = n + P;
  float* z0 = (float*)calloc((size_t)N, sizeof(float));
  memcpy(z0, du_end, sizeof(float) * n);
  adj_ctx ac; ac.df = df; ac.dense = &dense; ac.B = B; ac.n = n; ac.P = P; ac.y = (float*)malloc(sizeof(float) * (size_t)n);
  lro_field af; af.fn = adjoint_field; af.ctx = &ac; af.D = (int)N;
  float* stops = (float*)malloc(sizeof(float) * (size_t)(ntst > 0 ? ntst : 1));
  for (int i = 0; i < ntst; ++i) stops[i] = -tst[ntst - 1 - i];   /* ascending in s */
  lro_opts ob = *o; ob.save_everystep = 0; ob.save_start = 0;
  ob.alg = LRO_ALG_TSIT5;   /* the reversed solve is Tsit5 whatever the forward's method was (the product's choice: csrc/lrnde_adams.hpp) */
  float sv[1] = {-t0}; float tsv[2];
  float* zs = (float*)malloc(sizeof(float) * (size_t)N * 2);
  rc = lro_solve_ex(&af, z0, 1, -t2, -t0, &ob, sv, 1, zs, tsv, 2, st_bwd, btrace, cap_btrace, stops, ntst, NULL);
  if (rc == LRO_OK) {
    const float* zf = zs + (size_t)(st_bwd->nsaved - 1) * N;
    memcpy(dx, zf, sizeof(float) * n);
    memcpy(dp, zf + n, sizeof(float) * (size_t)P);
    if (mode != LRO_MODE_NONE && w_reg != 0.0f) {
      float* k1 = (float*)malloc(sizeof(float) * (size_t)n);
      float* gr = (float*)malloc(sizeof(float) * (size_t)P);
      float dtl, rv;
      lro_init_dt(&f, u1, t1, t2, oo.abstol, oo.reltol, B, k1, &dtl);
      step_reg_grad_generic(df, u1, k1, t1, dtl, oo.abstol, oo.reltol, B, reg_type, gr, &rv);
      for (int i = 0; i < P; ++i) dp[i] += w_reg * gr[i];
      free(gr); free(k1);
    }
  }
  free(zs); free(stops); free(ac.y); free(z0); free(u1); free(tst);
  lro_dense_free(&dense);
  return rc;
}

/* ------------------------------------------------------------------------- */
/* classifier head + loss (experiments/src/construct.jl:199, utils.jl:88)      */
/* ------------------------------------------------------------------------- */
float lro_classifier_ce(const float* u, int B, int D, const float* pc, int K, const int* labels, float* logits,
                        float* du, float* dpc) {
  double total = 0.0;
  double* dl = (double*)malloc(sizeof(double) * (size_t)B * K);
  for (int b = 0; b < B; ++b) {
    double lg[64];
    double mx = -1e300;
    for (int c = 0; c < K; ++c) {
      double s = (double)pc[(size_t)K * D + c];
      for (int k = 0; k < D; ++k) s += (double)pc[(size_t)c + (size_t)K * k] * (double)u[(size_t)b * D + k];
      lg[c] = s;
      if (s > mx) mx = s;
      if (logits) logits[(size_t)b * K + c] = (float)s;
    }
    double se = 0.0;
    for (int c = 0; c < K; ++c) se += exp(lg[c] - mx);
    const double lse = mx + log(se);
    total += lse - lg[labels[b]];
    for (int c = 0; c < K; ++c) dl[(size_t)b * K + c] = (exp(lg[c] - lse) - (c == labels[b] ? 1.0 : 0.0)) / (double)B;
  }
  if (du)
    for (int b = 0; b < B; ++b)
      for (int k = 0; k < D; ++k) {
        double s = 0.0;
        for (int c = 0; c < K; ++c) s += dl[(size_t)b * K + c] * (double)pc[(size_t)c + (size_t)K * k];
        du[(size_t)b * D + k] = (float)s;
      }
  if (dpc)
    for (int k = 0; k <= D; ++k)
      for (int c = 0; c < K; ++c) {
        double s = 0.0;
        for (int b = 0; b < B; ++b) s += dl[(size_t)b * K + c] * (k < D ? (double)u[(size_t)b * D + k] : 1.0);
        dpc[(size_t)c + (size_t)K * k] = (float)s;
      }
  free(dl);
  return (float)(total / (double)B);
}

/* public wrappers of the field-agnostic backward drivers */
int lro_tsit5_step_reg_grad(const lro_mlp* m, const float* uprev, const float* k1, float t, float dt,
                            float abstol, float reltol, int B, int reg_type, float* gp, float* reg_val) {
  lro_diff_field d; diff_from_mlp(m, &d);
  return step_reg_grad_generic(&d, uprev, k1, t, dt, abstol, reltol, B, reg_type, gp, reg_val);
}
int lro_node_backward(const lro_mlp* m, const float* x, int B, float t0, float t2, const lro_opts* o, int mode,
                      int reg_type, float t1_or_rand, const float* du_end, float w_reg, float* dx, float* dp,
                      lro_stats* st_fwd, lro_stats* st_bwd) {
  lro_diff_field d; diff_from_mlp(m, &d);
  return node_backward_generic(&d, x, B, t0, t2, o, mode, reg_type, t1_or_rand, du_end, w_reg, dx, dp, st_fwd, st_bwd, NULL, 0);
}
/* the same with the adjoint solve's per-attempt trace (t, dt in reversed time s = -t; st_bwd->iters rows are filled) */
int lro_node_backward_traced(const lro_mlp* m, const float* x, int B, float t0, float t2, const lro_opts* o, int mode,
                             int reg_type, float t1_or_rand, const float* du_end, float w_reg, float* dx, float* dp,
                             lro_stats* st_fwd, lro_stats* st_bwd, lro_trace_row* btrace, int cap_btrace) {
  lro_diff_field d; diff_from_mlp(m, &d);
  return node_backward_generic(&d, x, B, t0, t2, o, mode, reg_type, t1_or_rand, du_end, w_reg, dx, dp, st_fwd, st_bwd, btrace,
                               cap_btrace);
}
int lro_conv_step_reg_grad(const lro_conv* m, const float* uprev, const float* k1, float t, float dt,
                           float abstol, float reltol, int B, int reg_type, float* gp, float* reg_val) {
  lro_diff_field d; diff_from_conv(m, &d);
  return step_reg_grad_generic(&d, uprev, k1, t, dt, abstol, reltol, B, reg_type, gp, reg_val);
}
int lro_conv_node_backward(const lro_conv* m, const float* x, int B, float t0, float t2, const lro_opts* o, int mode,
                           int reg_type, float t1_or_rand, const float* du_end, float w_reg, float* dx, float* dp,
                           lro_stats* st_fwd, lro_stats* st_bwd) {
  lro_diff_field d; diff_from_conv(m, &d);
  return node_backward_generic(&d, x, B, t0, t2, o, mode, reg_type, t1_or_rand, du_end, w_reg, dx, dp, st_fwd, st_bwd, NULL, 0);
}

/* ------------------------------------------------------------------------- */
/* CIFAR10 stem and head (experiments/src/construct.jl:224-227)                */
/* ------------------------------------------------------------------------- */
int lro_cifar_stem_param_count(void) { return 135 + 5 + 8 + 8; }
int lro_cifar_head_param_count(int H, int W, int K) { return 72 + 1 + K * H * W + K; }

/* raw stem activation a0 = cat(x, conv(x) + bias) (B,8,H,W); conv is NNlib.conv (flipped kernel), pad 1 */
static void stem_raw(const float* x, int B, int H, int W, const float* ps, float* a0) {
  const long plane = (long)H * W;
  const float* w = ps; const float* b = ps + 135;
  for (int n = 0; n < B; ++n) {
    for (int c = 0; c < 3; ++c) memcpy(a0 + ((long)n * 8 + c) * plane, x + ((long)n * 3 + c) * plane, sizeof(float) * plane);
    for (int co = 0; co < 5; ++co)
      for (int y = 0; y < H; ++y)
        for (int xx = 0; xx < W; ++xx) {
          double acc = (double)b[co];
          for (int ci = 0; ci < 3; ++ci)
            for (int ky = 0; ky < 3; ++ky)
              for (int kx = 0; kx < 3; ++kx) {
                const int yy = y + 1 - ky, xi = xx + 1 - kx;
                if (yy < 0 || yy >= H || xi < 0 || xi >= W) continue;
                acc += (double)w[kx + 3 * (ky + 3 * (ci + 3 * co))] * (double)x[((long)n * 3 + ci) * plane + (long)yy * W + xi];
              }
          a0[((long)n * 8 + 3 + co) * plane + (long)y * W + xx] = (float)acc;
        }
  }
}
static void stem_stats_run(const float* a0, int B, long plane, int c, int train, const float* st, float eps, float* mean, float* inv,
                           float* run /* [mean 8; var 8] to advance, or NULL */) {
  if (train) {
    const double N = (double)B * (double)plane;
    double s = 0.0;
    for (int n = 0; n < B; ++n) { const float* p = a0 + ((long)n * 8 + c) * plane; for (long i = 0; i < plane; ++i) s += (double)p[i]; }
    const double mu = s / N;
    double v = 0.0;
    for (int n = 0; n < B; ++n) { const float* p = a0 + ((long)n * 8 + c) * plane; for (long i = 0; i < plane; ++i) { const double d = (double)p[i] - mu; v += d * d; } }
    *mean = (float)mu; *inv = (float)(1.0 / sqrt(v / N + (double)eps));
    if (run) { /* Lux BatchNorm's training-mode update (momentum 0.1, n/(n-1) correction), as batchnorm_act_ex */
      const float m = 0.1f, bm = (float)mu, bv = (float)(v / N);
      const float mcorr = m * (float)N / ((float)N - 1.0f);
      run[c] = (1.0f - m) * run[c] + m * bm;
      run[8 + c] = (1.0f - m) * run[8 + c] + mcorr * bv;
    }
  } else {
    *mean = st ? st[c] : 0.0f; *inv = (float)(1.0 / sqrt((double)(st ? st[8 + c] : 1.0f) + (double)eps));
  }
}
static void stem_stats(const float* a0, int B, long plane, int c, int train, const float* st, float eps, float* mean, float* inv) {
  stem_stats_run(a0, B, plane, c, train, st, eps, mean, inv, NULL);
}
void lro_cifar_stem_forward(const float* x, int B, int H, int W, const float* ps, int bn_train, const float* bn_state, float eps,
                            float* u0, float* bn_state_out) {
  if (bn_state_out) for (int i = 0; i < 16; ++i) bn_state_out[i] = bn_state ? bn_state[i] : (i < 8 ? 0.0f : 1.0f);
  const long plane = (long)H * W;
  float* a0 = (float*)malloc(sizeof(float) * (size_t)B * 8 * plane);
  stem_raw(x, B, H, W, ps, a0);
  const float* g = ps + 140; const float* be = ps + 148;
  for (int c = 0; c < 8; ++c) {
    float mean, inv;
    stem_stats_run(a0, B, plane, c, bn_train, bn_state, eps, &mean, &inv, bn_state_out);
    for (int n = 0; n < B; ++n)
      for (long i = 0; i < plane; ++i) {
        const long o = ((long)n * 8 + c) * plane + i;
        const float xn = (a0[o] - mean) * inv;
        u0[o] = xn * g[c] + be[c];
      }
  }
  free(a0);
}
void lro_cifar_stem_backward(const float* x, int B, int H, int W, const float* ps, int bn_train, const float* bn_state, float eps,
                             const float* du0, float* dps) {
  const long plane = (long)H * W;
  const double N = (double)B * (double)plane;
  float* a0 = (float*)malloc(sizeof(float) * (size_t)B * 8 * plane);
  float* da = (float*)malloc(sizeof(float) * (size_t)B * 8 * plane);
  stem_raw(x, B, H, W, ps, a0);
  const float* g = ps + 140;
  for (int c = 0; c < 8; ++c) {
    float mean, inv;
    stem_stats(a0, B, plane, c, bn_train, bn_state, eps, &mean, &inv);
    double s1 = 0.0, s2 = 0.0;
    for (int n = 0; n < B; ++n)
      for (long i = 0; i < plane; ++i) {
        const long o = ((long)n * 8 + c) * plane + i;
        const float xn = (a0[o] - mean) * inv;
        s1 += (double)du0[o]; s2 += (double)du0[o] * (double)xn;
      }
    dps[140 + c] = (float)s2; dps[148 + c] = (float)s1;
    const float m1 = bn_train ? (float)(s1 / N) : 0.0f, m2 = bn_train ? (float)(s2 / N) : 0.0f;
    for (int n = 0; n < B; ++n)
      for (long i = 0; i < plane; ++i) {
        const long o = ((long)n * 8 + c) * plane + i;
        const float xn = (a0[o] - mean) * inv;
        da[o] = (inv * g[c]) * ((du0[o] - m1) - xn * m2);
      }
  }
  for (int co = 0; co < 5; ++co) {
    double sb = 0.0;
    for (int n = 0; n < B; ++n) for (long i = 0; i < plane; ++i) sb += (double)da[((long)n * 8 + 3 + co) * plane + i];
    dps[135 + co] = (float)sb;
    for (int ci = 0; ci < 3; ++ci)
      for (int ky = 0; ky < 3; ++ky)
        for (int kx = 0; kx < 3; ++kx) {
          double acc = 0.0;
          for (int n = 0; n < B; ++n)
            for (int y = 0; y < H; ++y)
              for (int xx = 0; xx < W; ++xx) {
                const int yy = y + 1 - ky, xi = xx + 1 - kx;
                if (yy < 0 || yy >= H || xi < 0 || xi >= W) continue;
                acc += (double)da[((long)n * 8 + 3 + co) * plane + (long)y * W + xx] * (double)x[((long)n * 3 + ci) * plane + (long)yy * W + xi];
              }
          dps[kx + 3 * (ky + 3 * (ci + 3 * co))] = (float)acc;
        }
  }
  free(a0); free(da);
}
float lro_cifar_head_ce(const float* u, int B, int H, int W, const float* ph, int K, const int* labels, float* logits, float* du,
                        float* dph) {
  const long plane = (long)H * W;
  const int D = (int)plane;
  const float* wc = ph; const float bc = ph[72]; const float* pd = ph + 73; /* [vec(Wd) K x D; bd] */
  float* z = (float*)malloc(sizeof(float) * (size_t)B * plane);
  float* v = (float*)malloc(sizeof(float) * (size_t)B * plane);
  for (int n = 0; n < B; ++n)
    for (int y = 0; y < H; ++y)
      for (int xx = 0; xx < W; ++xx) {
        double acc = (double)bc;
        for (int ci = 0; ci < 8; ++ci)
          for (int ky = 0; ky < 3; ++ky)
            for (int kx = 0; kx < 3; ++kx) {
              const int yy = y + 1 - ky, xi = xx + 1 - kx;
              if (yy < 0 || yy >= H || xi < 0 || xi >= W) continue;
              acc += (double)wc[kx + 3 * (ky + 3 * ci)] * (double)u[((long)n * 8 + ci) * plane + (long)yy * W + xi];
            }
        z[(long)n * plane + (long)y * W + xx] = (float)acc;
        v[(long)n * plane + (long)y * W + xx] = act_apply(LRO_ACT_GELU, (float)acc);
      }
  float* dv = (du || dph) ? (float*)malloc(sizeof(float) * (size_t)B * plane) : NULL;
  float* dpd = dph ? dph + 73 : NULL;
  const float loss = lro_classifier_ce(v, B, D, pd, K, labels, logits, dv, dpd);
  if (du || dph) {
    for (long i = 0; i < (long)B * plane; ++i) { const float hh = act_apply(LRO_ACT_GELU, z[i]); dv[i] = dv[i] * act_deriv(LRO_ACT_GELU, z[i], hh); }
    if (dph) {
      double sb = 0.0;
      for (long i = 0; i < (long)B * plane; ++i) sb += (double)dv[i];
      dph[72] = (float)sb;
      for (int ci = 0; ci < 8; ++ci)
        for (int ky = 0; ky < 3; ++ky)
          for (int kx = 0; kx < 3; ++kx) {
            double acc = 0.0;
            for (int n = 0; n < B; ++n)
              for (int y = 0; y < H; ++y)
                for (int xx = 0; xx < W; ++xx) {
                  const int yy = y + 1 - ky, xi = xx + 1 - kx;
                  if (yy < 0 || yy >= H || xi < 0 || xi >= W) continue;
                  acc += (double)dv[(long)n * plane + (long)y * W + xx] * (double)u[((long)n * 8 + ci) * plane + (long)yy * W + xi];
                }
            dph[kx + 3 * (ky + 3 * ci)] = (float)acc;
          }
    }
    if (du)
      for (int n = 0; n < B; ++n)
        for (int ci = 0; ci < 8; ++ci)
          for (int y = 0; y < H; ++y)
            for (int xx = 0; xx < W; ++xx) {
              double acc = 0.0;
              for (int ky = 0; ky < 3; ++ky)
                for (int kx = 0; kx < 3; ++kx) {
                  const int yo = y - 1 + ky, xo = xx - 1 + kx;
                  if (yo < 0 || yo >= H || xo < 0 || xo >= W) continue;
                  acc += (double)wc[kx + 3 * (ky + 3 * ci)] * (double)dv[(long)n * plane + (long)yo * W + xo];
                }
              du[((long)n * 8 + ci) * plane + (long)y * W + xx] = (float)acc;
            }
    free(dv);
  }
  free(z); free(v);
  return loss;
}
