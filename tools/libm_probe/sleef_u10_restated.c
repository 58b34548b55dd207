#include <math.h>
#include <stdint.h>
#include <string.h>
static inline float mla(float x, float y, float z) { return fmaf(x, y, z); }
static inline float u2f(uint32_t u){ float f; memcpy(&f,&u,4); return f; }
static inline uint32_t f2u(float f){ uint32_t u; memcpy(&u,&f,4); return u; }
static inline float pow2if(int q) { return u2f((uint32_t)(q + 0x7f) << 23); }
static inline float ldexp2f(float d, int e) { return d * pow2if(e >> 1) * pow2if(e - (e >> 1)); }
#define R_LN2f 1.442695040888963407359924681001892137426645954152985934135449406931f
#define L2Uf 0.693145751953125f
#define L2Lf 1.428606765330187045e-06f
float sl_expf(float d) {
  int q = (int)rintf(d * R_LN2f);
  float s, u;
  s = mla((float)q, -L2Uf, d);
  s = mla((float)q, -L2Lf, s);
  u = 0.000198527617612853646278381f;
  u = mla(u, s, 0.00139304355252534151077271f);
  u = mla(u, s, 0.00833336077630519866943359f);
  u = mla(u, s, 0.0416664853692054748535156f);
  u = mla(u, s, 0.166666671633720397949219f);
  u = mla(u, s, 0.5f);
  u = 1.0f + mla(s * s, u, s);
  u = ldexp2f(u, q);
  if (d < -104) u = 0;
  if (d > 100) u = INFINITY;
  return u;
}
typedef struct { float x, y; } f2;
static inline f2 dfadd2_f2_f(f2 x, float y) { f2 r; r.x = x.x + y; float v = r.x - x.x; r.y = (x.x - (r.x - v)) + (y - v); r.y += x.y; return r; }
static inline f2 dfadd2_f2_f2(f2 x, f2 y) { f2 r; r.x = x.x + y.x; float v = r.x - x.x; r.y = (x.x - (r.x - v)) + (y.x - v); r.y += x.y + y.y; return r; }
static inline f2 dfadd_f2_f2(f2 x, f2 y) { f2 r; r.x = x.x + y.x; r.y = x.x - r.x + y.x + x.y + y.y; return r; }
static inline f2 dfadd_f_f2(float x, f2 y) { f2 r; r.x = x + y.x; r.y = x - r.x + y.x + y.y; return r; }
static inline f2 dfmul_f2_f(f2 x, float y) { f2 r; r.x = x.x * y; r.y = fmaf(x.y, y, fmaf(x.x, y, -r.x)); return r; }
static inline f2 dfmul_f2_f2(f2 x, f2 y) { f2 r; r.x = x.x * y.x; r.y = fmaf(x.x, y.y, fmaf(x.y, y.x, fmaf(x.x, y.x, -r.x))); return r; }
static inline f2 dfsqu(f2 x) { f2 r; r.x = x.x * x.x; r.y = fmaf(x.x + x.x, x.y, fmaf(x.x, x.x, -r.x)); return r; }
static inline f2 dfrec(f2 d) { f2 s; s.x = 1.0f / d.x; s.y = s.x * fmaf(-d.y, s.x, fmaf(-d.x, s.x, 1.0f)); return s; }
static inline f2 dfdiv(f2 n, f2 d) { float t = 1.0f / d.x; float s = n.x * t; float u = fmaf(t, n.x, -s); float v = fmaf(-d.y, t, fmaf(-d.x, t, 1.0f)); f2 r; r.x = s; r.y = fmaf(s, v, fmaf(n.y, t, u)); return r; }
static inline f2 dfneg(f2 x) { f2 r = {-x.x, -x.y}; return r; }
static f2 expk2f(f2 d) {
  float u = (d.x + d.y) * R_LN2f;
  int q = (int)rintf(u);
  f2 s, t;
  s = dfadd2_f2_f(d, (float)q * -L2Uf);
  s = dfadd2_f2_f(s, (float)q * -L2Lf);
  u = +0.1980960224e-3f;
  u = mla(u, s.x, +0.1394256484e-2f);
  u = mla(u, s.x, +0.8333456703e-2f);
  u = mla(u, s.x, +0.4166637361e-1f);
  t = dfadd2_f2_f(dfmul_f2_f(s, u), +0.166666659414234244790680580464e+0f);
  t = dfadd2_f2_f(dfmul_f2_f2(s, t), 0.5f);
  t = dfadd2_f2_f2(s, dfmul_f2_f2(dfsqu(s), t));
  t = dfadd_f_f2(1.0f, t);
  t.x = ldexp2f(t.x, q);
  t.y = ldexp2f(t.y, q);
  if (d.x < -104) { t.x = 0; t.y = 0; }
  return t;
}
float sl_tanhf(float x) {
  float y = fabsf(x);
  f2 d0 = {y, 0};
  f2 d = expk2f(d0);
  f2 e = dfrec(d);
  d = dfdiv(dfadd_f2_f2(d, dfneg(e)), dfadd_f2_f2(d, e));
  y = d.x + d.y;
  if (fabsf(x) > 8.664339742f || y != y) y = 1.0f;
  y = copysignf(1.0f, x) * y;   /* vmulsign */
  if (x != x) y = NAN;
  return y;
}
void sl_exp_arr(const float* in, float* out, long n) { for (long i = 0; i < n; ++i) out[i] = sl_expf(in[i]); }
void sl_tanh_arr(const float* in, float* out, long n) { for (long i = 0; i < n; ++i) out[i] = sl_tanhf(in[i]); }
