// Micro-benchmark: VALU issue cost per wave64 instruction on this GPU, by instruction kind and waves per SIMD.
// hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate && ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP 64
template <int KIND>
__global__ void k(float* out, long long* clk, int iters, float a, float b) {
  const long long t0 = clock64();
  float x0 = threadIdx.x * 1e-3f + a, x1 = x0 + 1.f, x2 = x0 + 2.f, x3 = x0 + 3.f, x4 = x0 + 4.f, x5 = x0 + 5.f, x6 = x0 + 6.f, x7 = x0 + 7.f;
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int r = 0; r < REP / 8; r++) {
      if (KIND == 0) {  // v_fma_f32, 8 independent chains
        asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                     "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9"
                     : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
      } else if (KIND == 1) {  // v_mul_f32
        asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n"
                     "v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8"
                     : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
      } else if (KIND == 2) {  // v_xor_b32
        asm volatile("v_xor_b32 %0, %0, %8\n v_xor_b32 %1, %1, %8\n v_xor_b32 %2, %2, %8\n v_xor_b32 %3, %3, %8\n"
                     "v_xor_b32 %4, %4, %8\n v_xor_b32 %5, %5, %8\n v_xor_b32 %6, %6, %8\n v_xor_b32 %7, %7, %8"
                     : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
      } else if (KIND == 3) {  // v_rcp_f32
        asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n"
                     "v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7"
                     : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
      } else if (KIND == 4) {  // v_cmp + v_cndmask pairs (4 pairs)
        asm volatile("v_cmp_lt_f32 vcc, %0, %8\n v_cndmask_b32 %1, %1, %9, vcc\n v_cmp_lt_f32 vcc, %2, %8\n v_cndmask_b32 %3, %3, %9, vcc\n"
                     "v_cmp_lt_f32 vcc, %4, %8\n v_cndmask_b32 %5, %5, %9, vcc\n v_cmp_lt_f32 vcc, %6, %8\n v_cndmask_b32 %7, %7, %9, vcc"
                     : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b) : "vcc");
      } else if (KIND == 5) {  // v_pk_fma_f32 on 4 register pairs
        typedef float v2 __attribute__((ext_vector_type(2)));
        v2 p0 = {x0, x1}, p1 = {x2, x3}, p2 = {x4, x5}, p3 = {x6, x7}, aa = {a, a}, bb = {b, b};
        asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                     "v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5"
                     : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(aa), "v"(bb));
        x0 = p0.x; x1 = p0.y; x2 = p1.x; x3 = p1.y; x4 = p2.x; x5 = p2.y; x6 = p3.x; x7 = p3.y;
      } else if (KIND == 7) {  // v_div_fixup_f32
        asm volatile("v_div_fixup_f32 %0, %0, %8, %9\n v_div_fixup_f32 %1, %1, %8, %9\n v_div_fixup_f32 %2, %2, %8, %9\n v_div_fixup_f32 %3, %3, %8, %9\n"
                     "v_div_fixup_f32 %4, %4, %8, %9\n v_div_fixup_f32 %5, %5, %8, %9\n v_div_fixup_f32 %6, %6, %8, %9\n v_div_fixup_f32 %7, %7, %8, %9"
                     : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
      } else if (KIND == 8) {  // v_div_scale_f32 (writes vcc)
        asm volatile("v_div_scale_f32 %0, vcc, %0, %8, %9\n v_div_scale_f32 %1, vcc, %1, %8, %9\n v_div_scale_f32 %2, vcc, %2, %8, %9\n v_div_scale_f32 %3, vcc, %3, %8, %9\n"
                     "v_div_scale_f32 %4, vcc, %4, %8, %9\n v_div_scale_f32 %5, vcc, %5, %8, %9\n v_div_scale_f32 %6, vcc, %6, %8, %9\n v_div_scale_f32 %7, vcc, %7, %8, %9"
                     : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b) : "vcc");
      } else if (KIND == 9) {  // v_div_fmas_f32 (reads vcc)
        asm volatile("v_div_fmas_f32 %0, %0, %8, %9\n v_div_fmas_f32 %1, %1, %8, %9\n v_div_fmas_f32 %2, %2, %8, %9\n v_div_fmas_f32 %3, %3, %8, %9\n"
                     "v_div_fmas_f32 %4, %4, %8, %9\n v_div_fmas_f32 %5, %5, %8, %9\n v_div_fmas_f32 %6, %6, %8, %9\n v_div_fmas_f32 %7, %7, %8, %9"
                     : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b) : "vcc");
      } else if (KIND == 10) {  // v_sqrt_f32
        asm volatile("v_sqrt_f32 %0, %0\n v_sqrt_f32 %1, %1\n v_sqrt_f32 %2, %2\n v_sqrt_f32 %3, %3\n"
                     "v_sqrt_f32 %4, %4\n v_sqrt_f32 %5, %5\n v_sqrt_f32 %6, %6\n v_sqrt_f32 %7, %7"
                     : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
      } else if (KIND == 11) {  // a / b as hipcc expands it (correctly rounded)
        x0 = x0 / a; x1 = x1 / a; x2 = x2 / a; x3 = x3 / a; x4 = x4 / b; x5 = x5 / b; x6 = x6 / b; x7 = x7 / b;
      } else if (KIND == 12) {  // sqrtf as hipcc expands it (correctly rounded)
        x0 = __builtin_sqrtf(x0 + a); x1 = __builtin_sqrtf(x1 + a); x2 = __builtin_sqrtf(x2 + a); x3 = __builtin_sqrtf(x3 + a);
        x4 = __builtin_sqrtf(x4 + a); x5 = __builtin_sqrtf(x5 + a); x6 = __builtin_sqrtf(x6 + a); x7 = __builtin_sqrtf(x7 + a);
      } else if (KIND == 6) {  // dependent v_fma chain (one accumulator)
        asm volatile("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                     "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2"
                     : "+v"(x0) : "v"(a), "v"(b));
      }
    }
  }
  const long long t1 = clock64();
  if ((threadIdx.x & 63) == 0) clk[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;  // shader cycles this wave spent in the loop
  out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}

template <int KIND>
void run(const char* name, float* out, long long* clk) {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount;
  const int iters = 2000;
  for (int wps : {1, 2, 4, 8}) {  // waves per SIMD: blocks of up to 1024 threads, two per CU for 8
    const int threads = wps == 8 ? 1024 : 64 * 4 * wps, blocks = wps == 8 ? 2 * cus : cus;
    const int waves = blocks * threads / 64;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float ms = 0.f;
    for (int rep = 0; rep < 3; rep++) {
      hipEventRecord(e0, 0);
      hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(threads), 0, 0, out, clk, iters, 1.0001f, 1e-9f);
      hipEventRecord(e1, 0);
      hipDeviceSynchronize();
      hipEventElapsedTime(&ms, e0, e1);
    }
    // wall-clock view: nanoseconds one SIMD spends per wave64 instruction (launch overhead included: ~2 % at these lengths)
    const double ns_per_simd = double(ms) * 1e6 / (double(iters) * REP * wps);
    std::vector<long long> h(waves);
    hipMemcpy(h.data(), clk, waves * sizeof(long long), hipMemcpyDeviceToHost);
    double sum = 0;
    for (long long v : h) sum += double(v);
    const double per_wave = sum / waves / (double(iters) * REP);  // shader cycles per instruction as one wave sees it
    printf("%-28s waves/SIMD %d  shader cycles per instruction: per wave %.2f, per SIMD %.2f | wall: %.3f ns per instruction and SIMD (kernel %.1f us)\n", name, wps, per_wave, per_wave / wps, ns_per_simd, ms * 1e3);
  }
}

int main() {
  float* out;
  long long* clk;
  hipMalloc(&out, 1 << 24);
  hipMalloc(&clk, 1 << 20);
  run<0>("v_fma_f32 x8 independent", out, clk);
  run<6>("v_fma_f32 dependent chain", out, clk);
  run<1>("v_mul_f32", out, clk);
  run<2>("v_xor_b32", out, clk);
  run<4>("v_cmp + v_cndmask", out, clk);
  run<3>("v_rcp_f32", out, clk);
  run<5>("v_pk_fma_f32", out, clk);
  run<10>("v_sqrt_f32", out, clk);
  run<8>("v_div_scale_f32", out, clk);
  run<9>("v_div_fmas_f32", out, clk);
  run<7>("v_div_fixup_f32", out, clk);
  run<11>("x / a (IEEE, hipcc), per division", out, clk);
  run<12>("sqrtf (IEEE, hipcc), per sqrt", out, clk);
  return 0;
}
