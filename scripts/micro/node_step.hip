// Micro-benchmark for the BVH traversal's two candidate limits on this GPU (round 4):
//  A. the node step's arithmetic alone (no memory): SIMD time per wave64 node step at 8 waves per SIMD, for the shipped
//     formulation (12 SDWA converts, 12 fma, min/max trees) and for candidates (near/far picked by a per-ray half-word
//     rotate instead of min/max; packed fma);
//  B. the node fetch alone: every lane chases its own chain of 32-byte nodes (two dwordx4 loads of one node per step)
//     through a buffer of a given size, with 64 / 32 / 16 lanes of each wave active: lane-loads per ns and CU, and the
//     time of a wave step — what the vector-memory path sustains when nothing else runs.
// hipcc --offload-arch=gfx950 -O3 -ffp-contract=off node_step.hip -o node_step && ./node_step
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <vector>

__device__ __forceinline__ float lo16(uint32_t w) { return static_cast<float>(w & 0xFFFFu); }
__device__ __forceinline__ float hi16(uint32_t w) { return static_cast<float>(w >> 16); }

// KIND 0: shipped slab arithmetic; 1: near/far by rotate (v_alignbit) + max3/min3; 2: shipped with packed fma
template <int KIND>
__global__ __launch_bounds__(256) void k_step(uint32_t* out, int iters, float ix, float iy, float iz, float ox, float oy, float oz, uint32_t seed) {
  uint32_t a0 = threadIdx.x * 2654435761u + seed, a1 = a0 * 3u, a2 = a0 * 5u, a3 = a0 * 7u, b0 = a0 * 11u, b1 = a0 * 13u;
  const float invx = ix + threadIdx.x * 1e-6f, invy = iy, invz = iz, oix = ox, oiy = oy, oiz = oz;
  const uint32_t rx = (threadIdx.x & 1) * 16u, ry = (threadIdx.x & 2) * 8u, rz = (threadIdx.x & 4) * 4u;  // per-ray rotate amounts
  float tb = 1e30f;
  uint32_t acc = 0;
  for (int i = 0; i < iters; i++) {
    float tl, tr;
    bool sl, sr;
    if (KIND == 1) {
      // each dword holds (min16 | max16 << 16) of one axis of one box; rotated by 16 for a negative direction the low half
      // is the near plane, the high half the far plane
      const uint32_t q0 = __builtin_amdgcn_alignbit(a0, a0, rx), q1 = __builtin_amdgcn_alignbit(a1, a1, ry), q2 = __builtin_amdgcn_alignbit(a2, a2, rz);
      const uint32_t q3 = __builtin_amdgcn_alignbit(a3, a3, rx), q4 = __builtin_amdgcn_alignbit(b0, b0, ry), q5 = __builtin_amdgcn_alignbit(b1, b1, rz);
      const float n0 = __builtin_fmaf(lo16(q0), invx, oix), f0 = __builtin_fmaf(hi16(q0), invx, oix);
      const float n1 = __builtin_fmaf(lo16(q1), invy, oiy), f1 = __builtin_fmaf(hi16(q1), invy, oiy);
      const float n2 = __builtin_fmaf(lo16(q2), invz, oiz), f2 = __builtin_fmaf(hi16(q2), invz, oiz);
      const float n3 = __builtin_fmaf(lo16(q3), invx, oix), f3 = __builtin_fmaf(hi16(q3), invx, oix);
      const float n4 = __builtin_fmaf(lo16(q4), invy, oiy), f4 = __builtin_fmaf(hi16(q4), invy, oiy);
      const float n5 = __builtin_fmaf(lo16(q5), invz, oiz), f5 = __builtin_fmaf(hi16(q5), invz, oiz);
      tl = __builtin_fmaxf(__builtin_fmaxf(n0, n1), __builtin_fmaxf(n2, 0.0f));
      tr = __builtin_fmaxf(__builtin_fmaxf(n3, n4), __builtin_fmaxf(n5, 0.0f));
      sl = tl <= __builtin_fminf(__builtin_fminf(f0, f1), __builtin_fminf(f2, tb));
      sr = tr <= __builtin_fminf(__builtin_fminf(f3, f4), __builtin_fminf(f5, tb));
    } else {
      float t0x, t1x, t0y, t1y, t0z, t1z, u0x, u1x, u0y, u1y, u0z, u1z;
      if (KIND == 2) {
        typedef float v2 __attribute__((ext_vector_type(2)));
        const v2 ivx = {invx, invx}, ivy = {invy, invy}, ivz = {invz, invz}, ovx = {oix, oix}, ovy = {oiy, oiy}, ovz = {oiz, oiz};
        const v2 px = __builtin_elementwise_fma(v2{lo16(a0), hi16(a0)}, ivx, ovx), py = __builtin_elementwise_fma(v2{lo16(a1), hi16(a1)}, ivy, ovy);
        const v2 pz = __builtin_elementwise_fma(v2{lo16(a2), hi16(a2)}, ivz, ovz), qx = __builtin_elementwise_fma(v2{lo16(a3), hi16(a3)}, ivx, ovx);
        const v2 qy = __builtin_elementwise_fma(v2{lo16(b0), hi16(b0)}, ivy, ovy), qz = __builtin_elementwise_fma(v2{lo16(b1), hi16(b1)}, ivz, ovz);
        t0x = px.x; t1x = px.y; t0y = py.x; t1y = py.y; t0z = pz.x; t1z = pz.y;
        u0x = qx.x; u1x = qx.y; u0y = qy.x; u1y = qy.y; u0z = qz.x; u1z = qz.y;
      } else {
        t0x = __builtin_fmaf(lo16(a0), invx, oix); t1x = __builtin_fmaf(hi16(a0), invx, oix);
        t0y = __builtin_fmaf(lo16(a1), invy, oiy); t1y = __builtin_fmaf(hi16(a1), invy, oiy);
        t0z = __builtin_fmaf(lo16(a2), invz, oiz); t1z = __builtin_fmaf(hi16(a2), invz, oiz);
        u0x = __builtin_fmaf(lo16(a3), invx, oix); u1x = __builtin_fmaf(hi16(a3), invx, oix);
        u0y = __builtin_fmaf(lo16(b0), invy, oiy); u1y = __builtin_fmaf(hi16(b0), invy, oiy);
        u0z = __builtin_fmaf(lo16(b1), invz, oiz); u1z = __builtin_fmaf(hi16(b1), invz, oiz);
      }
      tl = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(t0x, t1x), __builtin_fminf(t0y, t1y)), __builtin_fmaxf(__builtin_fminf(t0z, t1z), 0.0f));
      const float fl = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(t0x, t1x), __builtin_fmaxf(t0y, t1y)), __builtin_fminf(__builtin_fmaxf(t0z, t1z), tb));
      tr = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(u0x, u1x), __builtin_fminf(u0y, u1y)), __builtin_fmaxf(__builtin_fminf(u0z, u1z), 0.0f));
      const float fr = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(u0x, u1x), __builtin_fmaxf(u0y, u1y)), __builtin_fminf(__builtin_fmaxf(u0z, u1z), tb));
      sl = tl <= fl;
      sr = tr <= fr;
    }
    // consume the decision the way the traversal does (a select between the two child references) and make the next
    // "node" depend on it so nothing is hoisted
    const uint32_t nxt = (sl && sr) ? (tl <= tr ? a3 : b1) : (sl ? a3 : b0);
    acc += nxt;
    a0 = a0 * 1664525u + nxt; a1 ^= a0 >> 3; a2 += a1; a3 ^= a2 << 1; b0 += a3; b1 ^= b0 >> 5;
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

// B: every active lane chases a chain of nodes: next = hash(node contents) % n_nodes
__global__ __launch_bounds__(256) void k_chase(const uint4* nodes, uint32_t n_nodes, uint32_t* out, int iters, uint32_t active) {
  const uint32_t lane = threadIdx.x & 63u;
  uint32_t cur = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u % n_nodes;
  uint32_t acc = 0;
  if (lane < active) {
    for (int i = 0; i < iters; i++) {
      const uint4 a = nodes[2 * static_cast<size_t>(cur)], b = nodes[2 * static_cast<size_t>(cur) + 1];
      acc += a.y ^ b.z;
      cur = (a.x + b.w) % n_nodes;
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc + cur;
}

int main() {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount;
  uint32_t* out;
  hipMalloc(&out, 64 << 20);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  auto timed = [&](auto launch) {
    float best = 1e30f;
    for (int rep = 0; rep < 4; rep++) {
      hipEventRecord(e0, 0);
      launch();
      hipEventRecord(e1, 0);
      hipDeviceSynchronize();
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      if (rep && ms < best) best = ms;
    }
    return best;
  };
  const int blocks = cus * 8, iters = 4000;  // 8 workgroups of 4 waves per CU = 8 waves per SIMD
  const char* names[3] = {"shipped (cvt x12, fma x12, min/max)", "rotate + max3/min3", "shipped with packed fma"};
  for (int kind = 0; kind < 3; kind++) {
    const float ms = timed([&] {
      if (kind == 0) hipLaunchKernelGGL(k_step<0>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.1f, 0.9f, -1.2f, 3.f, 4.f, 5.f, 17u);
      if (kind == 1) hipLaunchKernelGGL(k_step<1>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.1f, 0.9f, -1.2f, 3.f, 4.f, 5.f, 17u);
      if (kind == 2) hipLaunchKernelGGL(k_step<2>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.1f, 0.9f, -1.2f, 3.f, 4.f, 5.f, 17u);
    });
    printf("A  %-40s %.1f ns of SIMD time per wave64 node step (8 waves per SIMD)\n", names[kind], double(ms) * 1e6 / (double(iters) * 8.0));
  }
  for (size_t mb : {8, 48, 512}) {
    const uint32_t n_nodes = static_cast<uint32_t>((mb << 20) / 32);
    std::vector<uint32_t> h(static_cast<size_t>(n_nodes) * 8);
    uint32_t s = 12345u;
    for (auto& v : h) v = (s = s * 1664525u + 1013904223u) >> 4;
    uint4* nodes;
    hipMalloc(&nodes, h.size() * 4);
    hipMemcpy(nodes, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    for (uint32_t active : {64u, 32u, 16u}) {
      const int it = 600;
      const float ms = timed([&] { hipLaunchKernelGGL(k_chase, dim3(blocks), dim3(256), 0, 0, nodes, n_nodes, out, it, active); });
      const double lane_loads = double(blocks) * 4 * active * it * 2;
      printf("B  %4zu MB of nodes, %2u lanes active: %.2f lane-loads (16 B) per ns and CU, %.0f ns per wave step, %.2f TB/s of node bytes\n", mb, active,
             lane_loads / (double(ms) * 1e6) / cus, double(ms) * 1e6 / it, lane_loads * 16 / (double(ms) * 1e9) / 1e3);
    }
    hipFree(nodes);
  }
  return 0;
}
