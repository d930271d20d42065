// exact_ops.hip — exhaustive check (all 2^32 binary32 patterns) of cheaper correctly-rounded sqrt / reciprocal sequences
// against hipcc's IEEE expansions (-fhip-fp32-correctly-rounded-divide-sqrt), and what each costs.  Build:
//   hipcc -O3 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt --offload-arch=gfx950 exact_ops.hip -o exact_ops
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

__device__ __forceinline__ uint32_t f2u(float f) { return __builtin_bit_cast(uint32_t, f); }
__device__ __forceinline__ float u2f(uint32_t u) { return __builtin_bit_cast(float, u); }

// candidate 0: v_sqrt_f32 + one ulp either way by the sign of the exact residuals (hipcc's core without scaling / class select)
__device__ __forceinline__ float sqrt_c0(float x) {
  float s = __builtin_amdgcn_sqrtf(x);
  float sd = u2f(f2u(s) - 1u), su = u2f(f2u(s) + 1u);
  float rd = __builtin_fmaf(-sd, s, x), ru = __builtin_fmaf(-su, s, x);
  s = (rd <= 0.0f) ? sd : s;
  s = (ru > 0.0f) ? su : s;
  return s;
}
// candidate 1: rsq + coupled Newton step + residual correction
__device__ __forceinline__ float sqrt_c1(float x) {
  float y = __builtin_amdgcn_rsqf(x);
  float s = x * y, h = 0.5f * y;
  float e = __builtin_fmaf(-h, s, 0.5f);
  s = __builtin_fmaf(s, e, s);
  h = __builtin_fmaf(h, e, h);
  float d = __builtin_fmaf(-s, s, x);
  return __builtin_fmaf(d, h, s);
}
// candidate 2: v_sqrt_f32 + one residual correction with h = 0.5 * rcp? no: h from rsq is a second transcendental; use s-only form
__device__ __forceinline__ float sqrt_c2(float x) {
  float s = __builtin_amdgcn_sqrtf(x);
  float su = u2f(f2u(s) + 1u);
  // s is within 1 ulp: the correctly rounded value is s-, s or s+.  r = x - s*s exact; midpoints via s*(s+-u)
  float r0 = __builtin_fmaf(-s, s, x);          // sign tells which side
  float sd = u2f(f2u(s) - 1u);
  float rr = r0 > 0.0f ? __builtin_fmaf(-su, s, x) : __builtin_fmaf(-sd, s, x);
  float pick = r0 > 0.0f ? su : sd;
  bool take = r0 > 0.0f ? (rr > 0.0f) : (rr <= 0.0f);
  return take ? pick : s;
}
// candidate 3: rsq, one residual correction
__device__ __forceinline__ float sqrt_c3(float x) {
  float y = __builtin_amdgcn_rsqf(x);
  float s = x * y, h = 0.5f * y;
  float d = __builtin_fmaf(-s, s, x);
  return __builtin_fmaf(d, h, s);
}
// candidate 4: v_sqrt, residual correction with h = 0.5 * rsq-free estimate: h = 0.5 / s via v_rcp (two transcendentals)
__device__ __forceinline__ float sqrt_c4(float x) {
  float s = __builtin_amdgcn_sqrtf(x);
  float h = 0.5f * __builtin_amdgcn_rcpf(s);
  float d = __builtin_fmaf(-s, s, x);
  return __builtin_fmaf(d, h, s);
}
// candidate 5: rsq, one residual correction, then a second residual correction
__device__ __forceinline__ float sqrt_c5(float x) {
  float y = __builtin_amdgcn_rsqf(x);
  float s = x * y, h = 0.5f * y;
  float d = __builtin_fmaf(-s, s, x);
  s = __builtin_fmaf(d, h, s);
  d = __builtin_fmaf(-s, s, x);
  return __builtin_fmaf(d, h, s);
}
__device__ __forceinline__ float rcp_c0(float x) {
  float r = __builtin_amdgcn_rcpf(x);
  float e = __builtin_fmaf(-x, r, 1.0f);
  return __builtin_fmaf(e, r, r);
}
__device__ __forceinline__ float rcp_c1(float x) {
  float r = __builtin_amdgcn_rcpf(x);
  float e = __builtin_fmaf(-x, r, 1.0f);
  r = __builtin_fmaf(e, r, r);
  e = __builtin_fmaf(-x, r, 1.0f);
  return __builtin_fmaf(e, r, r);
}
// hipcc's own division core for 1/x without div_scale / div_fmas / div_fixup
__device__ __forceinline__ float rcp_c2(float x) {
  float r = __builtin_amdgcn_rcpf(x);
  float e = __builtin_fmaf(-x, r, 1.0f);
  r = __builtin_fmaf(e, r, r);
  float q = r;  // 1 * r
  float e2 = __builtin_fmaf(-x, q, 1.0f);
  q = __builtin_fmaf(e2, r, q);
  float e3 = __builtin_fmaf(-x, q, 1.0f);
  return __builtin_fmaf(e3, r, q);
}

struct Stat {
  unsigned int hist[512];       // failures by sign and exponent
  unsigned long long bad;
  uint32_t lo_abs, hi_abs;      // range of |x| bits among failures with finite normal-range inputs
  uint32_t first[4];
};

template <int C>
__global__ void check(Stat* st) {
  const uint64_t base = (uint64_t(blockIdx.x) * blockDim.x + threadIdx.x) * 64;
  for (int k = 0; k < 64; k++) {
    const uint32_t b = uint32_t(base + k);
    const float x = u2f(b);
    float want, got;
    if (C < 3 || C >= 6) {
      want = __builtin_sqrtf(x);
      got = C == 0 ? sqrt_c0(x) : C == 1 ? sqrt_c1(x) : C == 2 ? sqrt_c2(x) : C == 6 ? sqrt_c3(x) : C == 7 ? sqrt_c4(x) : sqrt_c5(x);
    } else {
      want = 1.0f / x;
      got = C == 3 ? rcp_c0(x) : C == 4 ? rcp_c1(x) : rcp_c2(x);
    }
    if (f2u(want) != f2u(got)) {
      const unsigned long long n = atomicAdd(&st->bad, 1ull);
      atomicAdd(&st->hist[b >> 23], 1u);
      if (n < 4) st->first[n] = b;
      const uint32_t a = b & 0x7fffffffu;
      const bool neg_sqrt = (C < 3 || C >= 6) && (b >> 31);
      if (!neg_sqrt) {
        atomicMin(&st->lo_abs, a);
        atomicMax(&st->hi_abs, a);
      }
    }
  }
}

template <int C>
void run(const char* name, Stat* d) {
  Stat h{};
  h.lo_abs = 0xffffffffu;
  hipMemcpy(d, &h, sizeof h, hipMemcpyHostToDevice);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL(check<C>, dim3((1u << 26) / 256), dim3(256), 0, 0, d);
  hipEventRecord(e1, 0);
  hipDeviceSynchronize();
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  hipMemcpy(&h, d, sizeof h, hipMemcpyDeviceToHost);
  printf("%-44s mismatches %llu of 2^32  |x| bits of failures [%08x, %08x]  first %08x %08x %08x %08x  (%.1f ms)\n", name, h.bad, h.lo_abs,
         h.hi_abs, h.first[0], h.first[1], h.first[2], h.first[3], ms);
  printf("    failures by sign/exponent field:");
  for (int e = 0; e < 512; e++)
    if (h.hist[e]) printf(" %s%d:%u", e >= 256 ? "-" : "+", e & 255, h.hist[e]);
  printf("\n");
}

// ---- cost: ns per operation and SIMD at 8 waves per SIMD (wall clock), 8 independent chains
template <int C>
__global__ void cost(float* out, int iters, float a) {
  float x[8];
  for (int j = 0; j < 8; j++) x[j] = threadIdx.x * 1e-3f + a + j;
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int j = 0; j < 8; j++) {
      float v = x[j];
      float r = C == 0 ? __builtin_sqrtf(v) : C == 1 ? sqrt_c0(v) : C == 2 ? sqrt_c1(v) : C == 3 ? 1.0f / v : C == 4 ? rcp_c0(v) : C == 5 ? rcp_c1(v) : C == 6 ? rcp_c2(v) : C == 7 ? sqrt_c3(v) : C == 8 ? sqrt_c4(v) : sqrt_c5(v);
      x[j] = r + a;
    }
  }
  float s = 0;
  for (int j = 0; j < 8; j++) s += x[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int C>
void time_it(const char* name, float* out) {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  const int iters = 2000, blocks = 2 * p.multiProcessorCount, threads = 1024;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  float ms = 0;
  for (int rep = 0; rep < 3; rep++) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(cost<C>, dim3(blocks), dim3(threads), 0, 0, out, iters, 1.5f);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    hipEventElapsedTime(&ms, e0, e1);
  }
  printf("%-44s %.2f ns per operation and SIMD (incl. one v_add per operation)\n", name, double(ms) * 1e6 / (double(iters) * 8 * 8));
}

int main() {
  Stat* d;
  hipMalloc(&d, sizeof(Stat));
  run<0>("sqrt c0: v_sqrt + ulp step by residual signs", d);
  run<1>("sqrt c1: rsq + Newton + residual", d);
  run<2>("sqrt c2: v_sqrt + one-sided residual", d);
  run<6>("sqrt c3: rsq + 1 residual", d);
  run<7>("sqrt c4: v_sqrt + residual * 0.5 rcp", d);
  run<8>("sqrt c5: rsq + 2 residuals", d);
  run<3>("rcp  c0: v_rcp + 1 Newton", d);
  run<4>("rcp  c1: v_rcp + 2 Newton", d);
  run<5>("rcp  c2: hipcc core without scale/fixup", d);
  float* out;
  hipMalloc(&out, 1 << 24);
  time_it<0>("sqrtf (hipcc IEEE)", out);
  time_it<1>("sqrt c0", out);
  time_it<2>("sqrt c1", out);
  time_it<3>("1/x (hipcc IEEE)", out);
  time_it<4>("rcp c0", out);
  time_it<5>("rcp c1", out);
  time_it<6>("rcp c2", out);
  time_it<7>("sqrt c3", out);
  time_it<8>("sqrt c4", out);
  time_it<9>("sqrt c5", out);
  return 0;
}
