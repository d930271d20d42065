/*
 * ORACLE — TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is imported, linked or executed by
 * the product path; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it.
 *
 * det_math.h — the numerics contract ("NUMERICS" section of DESIGN.md) restated in plain C.
 *
 * GLSL leaves the precision of sin/cos/log/exp/pow/normalize and the contraction of a*b+c to
 * the implementation, and the reference never pins them (no tests).  To make
 * "BVH hit indices bit-exact" a checkable statement, every floating-point operation on a path
 * that feeds an integer observable (hit primitive id, RNG stream, reprojected pixel) is defined
 * here as a fixed sequence of IEEE-754 binary32 operations (+,-,*,/,sqrt,fma: all correctly
 * rounded, identical on x86-64 SSE/FMA and on gfx950 VALU).  The HIP kernels implement the same
 * sequences independently (real_time_path_tracing_with_spatiotemporal_filtering_amd/csrc/
 * rtpt_math.hpp); tests compare the two bit-for-bit and this file against double-precision libm
 * (tests/test_oracle_math.py) so the contract itself is pinned to the real functions.
 *
 * Build flags that matter: -ffp-contract=off (no implicit fusion; every fma below is explicit),
 * -mfma (hardware fma), no -ffast-math.
 */
#ifndef ORACLE_DET_MATH_H
#define ORACLE_DET_MATH_H

#include <stdint.h>
#include <string.h>

typedef struct { float x, y, z; } vec3;

static inline float dm_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
static inline float dm_sqrt(float x) { return __builtin_sqrtf(x); }
static inline uint32_t dm_bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float dm_float(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

/* GLSL min/max as the spec words them (min(x,y) = y<x ? y : x; max(x,y) = x<y ? y : x):
 * NaN handling follows from the comparison, not from IEEE minNum. */
static inline float dm_min(float x, float y) { return (y < x) ? y : x; }
static inline float dm_max(float x, float y) { return (x < y) ? y : x; }

static inline vec3 v3(float x, float y, float z) { vec3 r = {x, y, z}; return r; }
static inline vec3 v3_add(vec3 a, vec3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline vec3 v3_sub(vec3 a, vec3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline vec3 v3_mul(vec3 a, vec3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline vec3 v3_scale(vec3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
static inline vec3 v3_neg(vec3 a) { return v3(-a.x, -a.y, -a.z); }

/* dot(a,b) := fma(az,bz, fma(ay,by, ax*bx)) */
static inline float v3_dot(vec3 a, vec3 b) { return dm_fma(a.z, b.z, dm_fma(a.y, b.y, a.x * b.x)); }
/* cross(a,b).x := fma(ay,bz, -(az*by)) and cyclic */
static inline vec3 v3_cross(vec3 a, vec3 b) {
  return v3(dm_fma(a.y, b.z, -(a.z * b.y)), dm_fma(a.z, b.x, -(a.x * b.z)), dm_fma(a.x, b.y, -(a.y * b.x)));
}
static inline float v3_length(vec3 a) { return dm_sqrt(v3_dot(a, a)); }
/* normalize(v) := v * (1 / sqrt(dot(v,v)))  — one IEEE division, three multiplies */
static inline vec3 v3_normalize(vec3 a) {
  float inv = 1.0f / dm_sqrt(v3_dot(a, a));
  return v3_scale(a, inv);
}

/* sin(2*pi*u), cos(2*pi*u) for u in [0,1]  (theta = 2*k_pi*rng at raytrace.comp.glsl:90,:256).
 * q = floor(4u + 0.5); r = u - q/4 (exact); phi = r * fl(2*pi) in [-pi/4, pi/4];
 * cephes sinf/cosf minimax polynomials, Horner with fma; quadrant select on q & 3. */
static inline void dm_sincos2pi(float u, float* s_out, float* c_out) {
  float qf = __builtin_floorf(dm_fma(4.0f, u, 0.5f));
  float r = dm_fma(qf, -0.25f, u);
  float phi = r * 6.28318548202514648f;
  float z = phi * phi;
  float sp = dm_fma(z, -1.9515295891e-4f, 8.3321608736e-3f);
  sp = dm_fma(sp, z, -1.6666654611e-1f);
  float s = dm_fma(sp * z, phi, phi);
  float cp = dm_fma(z, 2.443315711809948e-5f, -1.388731625493765e-3f);
  cp = dm_fma(cp, z, 4.166664568298827e-2f);
  float c = dm_fma(cp * z, z, dm_fma(z, -0.5f, 1.0f));
  int q = (int)qf & 3;
  float so = (q == 0) ? s : (q == 1) ? c : (q == 2) ? -s : -c;
  float co = (q == 0) ? c : (q == 1) ? -s : (q == 2) ? -c : s;
  *s_out = so;
  *c_out = co;
}

/* natural log for x in (0, +inf) finite (used on u1 in [1e-38, 1], raytrace.comp.glsl:87-89).
 * cephes logf: x = m*2^e, m in [sqrt(1/2), sqrt(2)); subnormal inputs pre-scaled by 2^24. */
static inline float dm_log(float x) {
  int e = 0;
  if (x < 1.17549435e-38f) { x = x * 16777216.0f; e = -24; }
  uint32_t ix = dm_bits(x);
  e += (int)(ix >> 23) - 126;
  float m = dm_float((ix & 0x007fffffu) | 0x3f000000u); /* [0.5,1) */
  if (m < 0.707106781186547524f) { e -= 1; m = m + m - 1.0f; } else { m = m - 1.0f; }
  float z = m * m;
  float p = dm_fma(7.0376836292e-2f, m, -1.1514610310e-1f);
  p = dm_fma(p, m, 1.1676998740e-1f);
  p = dm_fma(p, m, -1.2420140846e-1f);
  p = dm_fma(p, m, 1.4249322787e-1f);
  p = dm_fma(p, m, -1.6668057665e-1f);
  p = dm_fma(p, m, 2.0000714765e-1f);
  p = dm_fma(p, m, -2.4999993993e-1f);
  p = dm_fma(p, m, 3.3333331174e-1f);
  float fe = (float)e;
  float y = (m * z) * p;
  y = dm_fma(fe, -2.12194440e-4f, y);
  y = dm_fma(z, -0.5f, y);
  float r = m + y;
  r = dm_fma(fe, 0.693359375f, r);
  return r;
}

/* exp(x) for the edge-stopping weights (temporalFiltering.comp.glsl:68,:73), x <= 0 in use.
 * cephes expf; x < -87 := 0, x > 88 := +inf.  The HIP filter kernels use the hardware exp2
 * instead (float-only output, tolerance stated in tests/test_parity_gpu.py). */
static inline float dm_exp(float x) {
  if (x != x) return x;
  if (x < -87.0f) return 0.0f;
  if (x > 88.0f) return __builtin_inff();
  float n = __builtin_floorf(dm_fma(x, 1.44269504088896341f, 0.5f));
  float r = dm_fma(n, -0.693359375f, x);
  r = dm_fma(n, 2.12194440e-4f, r);
  float z = r * r;
  float p = dm_fma(1.9875691500e-4f, r, 1.3981999507e-3f);
  p = dm_fma(p, r, 8.3334519073e-3f);
  p = dm_fma(p, r, 4.1665795894e-2f);
  p = dm_fma(p, r, 1.6666665459e-1f);
  p = dm_fma(p, r, 5.0000001201e-1f);
  float y = dm_fma(p, z, r) + 1.0f;
  int ni = (int)n;
  return y * dm_float((uint32_t)(ni + 127) << 23);
}

/* pow(x, n) for integer n >= 1 by binary exponentiation (sigma_n = 128 -> seven squarings);
 * stands for pow(.,128) at temporalFiltering.comp.glsl:62 and temporalGradient.comp.glsl:92. */
static inline float dm_powi(float x, int n) {
  float r = 1.0f, b = x;
  int first = 1;
  while (n > 0) {
    if (n & 1) { r = first ? b : r * b; first = 0; }
    n >>= 1;
    if (n) b = b * b;
  }
  return r;
}

/* float -> int with truncation toward zero (GLSL ivec2(vec2), temporalFiltering.comp.glsl:238);
 * NaN := 0 and out-of-range saturates, matching gfx950 v_cvt_i32_f32. */
static inline int32_t dm_f2i(float x) {
  if (x != x) return 0;
  if (x >= 2147483648.0f) return 2147483647;
  if (x <= -2147483648.0f) return (int32_t)(-2147483647 - 1);
  return (int32_t)x;
}

#endif
