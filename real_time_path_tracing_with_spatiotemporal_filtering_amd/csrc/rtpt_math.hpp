// rtpt_math.hpp — device/host arithmetic of the product path (gfx950 HIP, also compiled for the
// host half of the library).
//
// Two families live here:
//   * exact:: fixed sequences of correctly-rounded binary32 operations (+ - * / sqrt fma).  Used
//     wherever a float feeds an integer observable of the reference — the primitive id returned by
//     the ray query (raytrace.comp.glsl:120), the RNG-driven path (raytrace.comp.glsl:256-261) and
//     the reprojected pixel (temporalFiltering.comp.glsl:238).  These sequences are this
//     project's NUMERICS contract; a CPU restatement of the same contract exists as
//     test infrastructure and the two are compared bit-for-bit through rtpt_selftest_math.
//   * fast::  single-instruction hardware approximations (v_exp_f32, v_sqrt_f32, v_rcp_f32) for
//     the HBM-bound filter kernels whose outputs are float planes only.
//
// The file must be compiled with -ffp-contract=off: every fused multiply-add is written out.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#define RT_HD __host__ __device__ __forceinline__

namespace rt {

struct f3 {
  float x, y, z;
};

RT_HD f3 make_f3(float x, float y, float z) { return f3{x, y, z}; }
RT_HD f3 operator+(f3 a, f3 b) { return f3{a.x + b.x, a.y + b.y, a.z + b.z}; }
RT_HD f3 operator-(f3 a, f3 b) { return f3{a.x - b.x, a.y - b.y, a.z - b.z}; }
RT_HD f3 operator*(f3 a, f3 b) { return f3{a.x * b.x, a.y * b.y, a.z * b.z}; }
RT_HD f3 operator*(f3 a, float s) { return f3{a.x * s, a.y * s, a.z * s}; }
RT_HD f3 operator-(f3 a) { return f3{-a.x, -a.y, -a.z}; }

RT_HD float fmaf_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

// GLSL-worded min/max: min(x,y) = y<x ? y : x ; max(x,y) = x<y ? y : x
RT_HD float glsl_min(float x, float y) { return (y < x) ? y : x; }
RT_HD float glsl_max(float x, float y) { return (x < y) ? y : x; }

RT_HD uint32_t f2u(float f) { return __builtin_bit_cast(uint32_t, f); }
RT_HD float u2f(uint32_t u) { return __builtin_bit_cast(float, u); }

namespace exact {

#if defined(__HIP_DEVICE_COMPILE__)
// Correctly rounded square root and reciprocal, cheaper than hipcc's IEEE expansions.  Every kernel of the frame is bound by
// VALU issue (profiles/r03_valu_issue_micro.txt) and the expansions are long because they serve every input: sqrt scales
// its argument into and out of the normal range and selects by class (16 VALU + v_sqrt_f32, 24.4 ns per wave64 operation
// and SIMD), 1/x is the full division (v_div_scale x2, v_rcp, 6 fma, v_div_fmas, v_div_fixup: 19.3 ns).  The sequences
// below give the IEEE result bit for bit on the inputs a frame produces and hand everything else to hipcc's sequence under
// a branch — which inputs are which was settled by running all 2^32 patterns (scripts/micro/exact_ops.hip), and
// rtpt_selftest_exhaustive re-checks the shipped functions against hipcc's over all 2^32 patterns on the GPU in use:
//   sqrt: +-0 and 2^-102 <= x < inf:  y = v_rsq_f32(max(x, 2^-102)); s = x*y; h = y/2; s + (x - s*s)*h      (7.9 ns)
//   1/x:  exponent field 1..252 (2^-126 <= |x| < 2^126):  r = v_rcp_f32(x); r + (1 - x*r)*r                  (7.4 ns)
// (NaN results keep hipcc's bit patterns because NaN inputs take hipcc's path.)
__device__ __forceinline__ float sqrt_(float x) {
  const float lo = 1.97215226305252951e-31f;  // 2^-102: below it the residual x - s*s leaves the normal range
  const float t = __builtin_fmaxf(x, lo);
  // t == x: lo <= x <= +inf (false for NaN, negatives, zeros, small x); the class test flips +inf out and the zeros in
  const bool direct = (t == x) != __builtin_amdgcn_classf(x, 0x260);  // 0x260 = +inf | +0 | -0
  const float y = __builtin_amdgcn_rsqf(t);
  const float s = x * y, h = 0.5f * y;
  float r = fmaf_(fmaf_(-s, s, x), h, s);
  if (__builtin_expect(!direct, 0)) r = __builtin_sqrtf(x);
  return r;
}
__device__ __forceinline__ float rcp_(float x) {
  const uint32_t b = f2u(x);
  const bool direct = (b + b) - 0x01000000u < 0xfc000000u;  // exponent field 1..252: x, 1/x and the residual are all normal
  float r = __builtin_amdgcn_rcpf(x);
  r = fmaf_(fmaf_(-x, r, 1.0f), r, r);
  if (__builtin_expect(!direct, 0)) r = 1.0f / x;
  return r;
}
// a / b, correctly rounded, for operands of ordinary magnitude: the reciprocal above (correctly rounded, hence Markstein's
// scheme applies), q = a*r, one residual correction — 5 VALU + v_rcp_f32 against hipcc's 10 + v_rcp_f32, and divisions that
// share a denominator share the reciprocal (3 VALU each).  There are too many operand pairs to run them all, but not too
// many SIGNIFICAND pairs: scripts/micro/exact_div.hip compared the sequence with hipcc's division on all 2^23 x 2^23 of them
// (0 mismatches; the variant without the Newton step fails 47,045 of them; profiles/r03_exact_div_exhaustive.txt) and
// checked that v_rcp_f32 commutes with scaling by two.  Every instruction of the sequence commutes with scaling the operands
// by powers of two as long as nothing leaves the normal range, so the result holds for every pair with 2^-62 <= |a|, |b| <
// 2^62 (quotient in 2^-125 .. 2^124, residual a multiple of 2^-170 >> 2^-149); zeros, infinities, NaNs and everything
// outside that window take hipcc's division.  rtpt_selftest_div re-runs any slice of the enumeration on the shipped function.
__device__ __forceinline__ float div_(float a, float b) {
  const uint32_t ua = f2u(a), ub = f2u(b);
  // exponent field 65..188 for both (the sign is shifted out)
  const bool direct = ((ua + ua) - 0x41000000u < 0x7c000000u) & ((ub + ub) - 0x41000000u < 0x7c000000u);
  float r = __builtin_amdgcn_rcpf(b);
  r = fmaf_(fmaf_(-b, r, 1.0f), r, r);
  float q = a * r;
  q = fmaf_(fmaf_(-b, q, a), r, q);
  if (__builtin_expect(!direct, 0)) q = a / b;
  return q;
}
#else
inline float sqrt_(float x) { return __builtin_sqrtf(x); }  // correctly rounded
inline float rcp_(float x) { return 1.0f / x; }             // correctly rounded division
inline float div_(float a, float b) { return a / b; }
#endif

RT_HD float dot(f3 a, f3 b) { return fmaf_(a.z, b.z, fmaf_(a.y, b.y, a.x * b.x)); }
RT_HD f3 cross(f3 a, f3 b) {
  return f3{fmaf_(a.y, b.z, -(a.z * b.y)), fmaf_(a.z, b.x, -(a.x * b.z)), fmaf_(a.x, b.y, -(a.y * b.x))};
}
RT_HD float length(f3 a) { return sqrt_(dot(a, a)); }
RT_HD f3 normalize(f3 a) {
  float inv = rcp_(sqrt_(dot(a, a)));
  return a * inv;
}

// sin/cos of 2*pi*u, u in [0,1]; quarter-turn reduction is exact, then degree-7/8 minimax kernels
RT_HD void sincos2pi(float u, float& s_out, float& c_out) {
  float qf = __builtin_floorf(fmaf_(4.0f, u, 0.5f));
  float r = fmaf_(qf, -0.25f, u);
  float phi = r * 6.28318548202514648f;
  float z = phi * phi;
  float sp = fmaf_(z, -1.9515295891e-4f, 8.3321608736e-3f);
  sp = fmaf_(sp, z, -1.6666654611e-1f);
  float s = fmaf_(sp * z, phi, phi);
  float cp = fmaf_(z, 2.443315711809948e-5f, -1.388731625493765e-3f);
  cp = fmaf_(cp, z, 4.166664568298827e-2f);
  float c = fmaf_(cp * z, z, fmaf_(z, -0.5f, 1.0f));
  int q = static_cast<int>(qf) & 3;
  bool swap = q & 1;
  float a = swap ? c : s;  // |sin| source
  float b = swap ? s : c;  // |cos| source
  s_out = (q & 2) ? -a : a;
  c_out = ((q + 1) & 2) ? -b : b;
}

// natural logarithm, finite x > 0 (subnormals pre-scaled by 2^24)
RT_HD float log_(float x) {
  int e = 0;
  if (x < 1.17549435e-38f) {
    x = x * 16777216.0f;
    e = -24;
  }
  uint32_t ix = f2u(x);
  e += static_cast<int>(ix >> 23) - 126;
  float m = u2f((ix & 0x007fffffu) | 0x3f000000u);
  if (m < 0.707106781186547524f) {
    e -= 1;
    m = m + m - 1.0f;
  } else {
    m = m - 1.0f;
  }
  float z = m * m;
  float p = fmaf_(7.0376836292e-2f, m, -1.1514610310e-1f);
  p = fmaf_(p, m, 1.1676998740e-1f);
  p = fmaf_(p, m, -1.2420140846e-1f);
  p = fmaf_(p, m, 1.4249322787e-1f);
  p = fmaf_(p, m, -1.6668057665e-1f);
  p = fmaf_(p, m, 2.0000714765e-1f);
  p = fmaf_(p, m, -2.4999993993e-1f);
  p = fmaf_(p, m, 3.3333331174e-1f);
  float fe = static_cast<float>(e);
  float y = (m * z) * p;
  y = fmaf_(fe, -2.12194440e-4f, y);
  y = fmaf_(z, -0.5f, y);
  float r = m + y;
  return fmaf_(fe, 0.693359375f, r);
}

// e^x by 2^n * P(r), r = x - n ln2 (two-constant reduction); x < -87 -> 0, x > 88 -> +inf.
// Only the strict-parity build of the filter uses it; the shipping filter uses fast::exp_.
RT_HD float exp_(float x) {
  if (x != x) return x;
  if (x < -87.0f) return 0.0f;
  if (x > 88.0f) return __builtin_inff();
  float n = __builtin_floorf(fmaf_(x, 1.44269504088896341f, 0.5f));
  float r = fmaf_(n, -0.693359375f, x);
  r = fmaf_(n, 2.12194440e-4f, r);
  float z = r * r;
  float p = fmaf_(1.9875691500e-4f, r, 1.3981999507e-3f);
  p = fmaf_(p, r, 8.3334519073e-3f);
  p = fmaf_(p, r, 4.1665795894e-2f);
  p = fmaf_(p, r, 1.6666665459e-1f);
  p = fmaf_(p, r, 5.0000001201e-1f);
  float y = fmaf_(p, z, r) + 1.0f;
  int ni = static_cast<int>(n);
  return y * u2f(static_cast<uint32_t>(ni + 127) << 23);
}

// x^n, integer n >= 1, square-and-multiply (n = 128: seven squarings)
RT_HD float powi(float x, int n) {
  float r = 1.0f, b = x;
  bool first = true;
  while (n > 0) {
    if (n & 1) {
      r = first ? b : r * b;
      first = false;
    }
    n >>= 1;
    if (n) b = b * b;
  }
  return r;
}

// float -> int, truncation toward zero, NaN -> 0, saturating (== v_cvt_i32_f32)
RT_HD int32_t f2i(float x) {
  if (x != x) return 0;
  if (x >= 2147483648.0f) return 2147483647;
  if (x <= -2147483648.0f) return static_cast<int32_t>(-2147483647 - 1);
  return static_cast<int32_t>(x);
}

// PCG RXS-M-XS 32 (raytrace.comp.glsl:71-78); the divisor 4294967295.0f is 2^32 in binary32
RT_HD float rng_next(uint32_t& s) {
  s = s * 747796405u + 1u;
  uint32_t w = ((s >> ((s >> 28) + 4u)) ^ s) * 277803737u;
  w = (w >> 22) ^ w;
  return static_cast<float>(w) * 2.3283064365386963e-10f;
}
RT_HD uint32_t rng_seed(uint32_t px, uint32_t py, uint32_t frame, uint32_t batch) {
  return (px * 3266489917u + py * 668265263u) ^ (frame * 374761393u) ^ (batch * 2654435761u);  // :297
}

// column-major mat4 helpers with a fixed accumulation order
RT_HD float mat_row_point(const float* M, int i, f3 p) {
  return fmaf_(M[8 + i], p.z, fmaf_(M[4 + i], p.y, M[i] * p.x)) + M[12 + i];
}
inline void mat_mul(const float* A, const float* B, float* C) {
  for (int c = 0; c < 4; c++)
    for (int r = 0; r < 4; r++) {
      float acc = A[r] * B[c * 4];
      acc = fmaf_(A[4 + r], B[c * 4 + 1], acc);
      acc = fmaf_(A[8 + r], B[c * 4 + 2], acc);
      acc = fmaf_(A[12 + r], B[c * 4 + 3], acc);
      C[c * 4 + r] = acc;
    }
}

}  // namespace exact

namespace fast {
__device__ __forceinline__ float exp_(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896341f); }
__device__ __forceinline__ float sqrt_(float x) { return __builtin_amdgcn_sqrtf(x); }
__device__ __forceinline__ float rcp_(float x) { return __builtin_amdgcn_rcpf(x); }
}  // namespace fast

}  // namespace rt
