// exact_div.hip — is a short division sequence correctly rounded?  Enumerates ALL 2^23 x 2^23 pairs of binary32 significands
// (a, b in [1, 2): every other pair of normal operands whose quotient and residuals stay in the normal range is one of these
// scaled by powers of two, which every instruction of the sequences commutes with — checked for v_rcp_f32 itself by the
// first kernel) and compares each candidate with hipcc's IEEE division.  ~1 minute on an MI355X.  Build:
//   hipcc -O3 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt --offload-arch=gfx950 exact_div.hip -o exact_div
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

__device__ __forceinline__ uint32_t f2u(float f) { return __builtin_bit_cast(uint32_t, f); }
__device__ __forceinline__ float u2f(uint32_t u) { return __builtin_bit_cast(float, u); }

// D0: v_rcp, q = a*r, one residual correction                                  (3 VALU + rcp)
__device__ __forceinline__ float div_d0(float a, float b) {
  float r = __builtin_amdgcn_rcpf(b);
  float q = a * r;
  return __builtin_fmaf(__builtin_fmaf(-b, q, a), r, q);
}
// D1: v_rcp + Newton step (the correctly rounded reciprocal, exact_ops.hip), q = a*r, one residual correction   (5 VALU + rcp)
__device__ __forceinline__ float div_d1(float a, float b) {
  float r = __builtin_amdgcn_rcpf(b);
  r = __builtin_fmaf(__builtin_fmaf(-b, r, 1.0f), r, r);
  float q = a * r;
  return __builtin_fmaf(__builtin_fmaf(-b, q, a), r, q);
}
// D2: hipcc's own core (two residual corrections) without v_div_scale / v_div_fmas / v_div_fixup               (7 VALU + rcp)
__device__ __forceinline__ float div_d2(float a, float b) {
  float r = __builtin_amdgcn_rcpf(b);
  r = __builtin_fmaf(__builtin_fmaf(-b, r, 1.0f), r, r);
  float q = a * r;
  q = __builtin_fmaf(__builtin_fmaf(-b, q, a), r, q);
  return __builtin_fmaf(__builtin_fmaf(-b, q, a), r, q);
}

struct Stat {
  unsigned long long bad[3];
  uint32_t first[3][2];
  unsigned long long rcp_scale_bad;
};

// v_rcp_f32 commutes with scaling by 2: rcp(2x) == rcp(x) / 2 for every x whose reciprocal and half-reciprocal are normal
__global__ void rcp_scale(Stat* st) {
  const uint64_t base = (uint64_t(blockIdx.x) * blockDim.x + threadIdx.x) * 64;
  for (int k = 0; k < 64; k++) {
    const uint32_t b = uint32_t(base + k);
    const uint32_t e = (b >> 23) & 0xffu;
    if (e < 2 || e > 250) continue;
    const float x = u2f(b);
    if (f2u(__builtin_amdgcn_rcpf(x + x)) != f2u(__builtin_amdgcn_rcpf(x) * 0.5f)) atomicAdd(&st->rcp_scale_bad, 1ull);
  }
}

// thread = one b; the launch covers 2^15 values of a starting at a0
__global__ void check(Stat* st, uint32_t a0) {
  const uint32_t mb = blockIdx.x * blockDim.x + threadIdx.x;  // 0 .. 2^23-1
  const float b = u2f(0x3f800000u | mb);
  unsigned bad0 = 0, bad1 = 0, bad2 = 0;
  uint32_t fa0 = 0, fa1 = 0, fa2 = 0;
  for (uint32_t i = 0; i < (1u << 15); i++) {
    const uint32_t ma = a0 + i;
    const float a = u2f(0x3f800000u | ma);
    const uint32_t want = f2u(a / b);
    if (f2u(div_d0(a, b)) != want) { bad0++; fa0 = ma; }
    if (f2u(div_d1(a, b)) != want) { bad1++; fa1 = ma; }
    if (f2u(div_d2(a, b)) != want) { bad2++; fa2 = ma; }
  }
  if (bad0) { if (atomicAdd(&st->bad[0], (unsigned long long)bad0) == 0) { st->first[0][0] = fa0; st->first[0][1] = mb; } }
  if (bad1) { if (atomicAdd(&st->bad[1], (unsigned long long)bad1) == 0) { st->first[1][0] = fa1; st->first[1][1] = mb; } }
  if (bad2) { if (atomicAdd(&st->bad[2], (unsigned long long)bad2) == 0) { st->first[2][0] = fa2; st->first[2][1] = mb; } }
}

int main() {
  Stat* d;
  if (hipMalloc(&d, sizeof(Stat)) != hipSuccess) return 1;
  (void)hipMemset(d, 0, sizeof(Stat));
  hipLaunchKernelGGL(rcp_scale, dim3((1u << 26) / 256), dim3(256), 0, 0, d);
  (void)hipDeviceSynchronize();
  Stat h;
  (void)hipMemcpy(&h, d, sizeof h, hipMemcpyDeviceToHost);
  printf("v_rcp_f32(2x) != v_rcp_f32(x)/2 for %llu inputs (exponent fields 2..250)\n", h.rcp_scale_bad);
  fflush(stdout);
  for (uint32_t pass = 0; pass < 256; pass++) {
    hipLaunchKernelGGL(check, dim3((1u << 23) / 256), dim3(256), 0, 0, d, pass << 15);
    if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
    if ((pass & 31) == 31) {
      (void)hipMemcpy(&h, d, sizeof h, hipMemcpyDeviceToHost);
      printf("a < 1 + %u/256: mismatches D0 %llu  D1 %llu  D2 %llu\n", pass + 1, h.bad[0], h.bad[1], h.bad[2]);
      fflush(stdout);
    }
  }
  (void)hipMemcpy(&h, d, sizeof h, hipMemcpyDeviceToHost);
  const char* names[3] = {"D0 rcp, mul, 1 residual (3 VALU + rcp)", "D1 rcp + Newton, mul, 1 residual (5 VALU + rcp)", "D2 hipcc core, 2 residuals (7 VALU + rcp)"};
  for (int c = 0; c < 3; c++)
    printf("%-52s mismatches %llu of 2^46 significand pairs  (one: a = 1+%u/2^23, b = 1+%u/2^23)\n", names[c], h.bad[c], h.first[c][0], h.first[c][1]);
  return 0;
}
