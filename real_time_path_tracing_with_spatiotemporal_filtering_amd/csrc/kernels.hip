// kernels.hip — hand-written gfx950 (CDNA4, wave64) kernels of the hot path.
//
//   k_scene_prepare / k_lut   per-triangle records, LUT (visibility.geom.glsl:44-59), normal table
//   k_gbuffer                 K0  visibility.{vert,geom,frag}.glsl as pixel-centre primary rays
//   k_gradient                K1  temporalGradient.comp.glsl:104-172
//   k_pathtrace               K2  raytrace.comp.glsl:273-344 (software closest-hit, no ray query)
//   k_atrous                  K3  temporalFiltering.comp.glsl:191-265, one iteration; FINAL fuses
//                                 reprojection + blend (:213-263)
//
// No MFMA anywhere: nothing on this path is a dense contraction.  Compile with -ffp-contract=off.
#include "device_common.hpp"

namespace rt {
namespace {

// ------------------------------------------------------------------------------------------
// closest hit.  D4: the winner is min over (t, id) of ONE ray-triangle routine, so the result
// does not depend on which structure enumerates the candidates.
// ------------------------------------------------------------------------------------------
struct HitRec {
  float t;       // best t so far (initialised to tmax)
  uint32_t id1;  // primitive id + 1, 0 = none
  float u, v, ad;  // scaled barycentrics of the best hit, u NEGATED: b1 = -u/ad, b2 = v/ad (tri_test)
};

// The divisions of the tracing kernels are hipcc's correctly rounded ones, not exact::div_ (rtpt_math.hpp): the tracing kernels
// sit at their 64-VGPR budget (8 waves per SIMD), the short sequence keeps the refined reciprocal live next to the operands
// of the long path, and what the fewer instructions bring (K2 at 4K 368.7 -> 364.9 us with both) the spills take back on the
// BVH kernel (3.34 -> 3.41 ms; profiles/r03_div_ab.csv).  Two names so that the two groups stay easy to find.
#define RTPT_DIV_TH(a, b) ((a) / (b))  // hit distances
#define RTPT_DIV_SH(a, b) ((a) / (b))  // shading (barycentrics, the light test, the pixel's ndc)

// Scalar-triple-product form of Moller-Trumbore, plane normal n = e1 x e2 precomputed per triangle,
// division deferred until a candidate passes the inside tests:
//   det = -d.n,  tt = (o-v0).n,  c = (o-v0) x d,  u = e2.c,  v = -e1.c        (21 flops instead of 27)
// record: r0 = (v0.xyz, e1.x)  r1 = (e1.yz, e2.xy)  r2 = (e2.z, n.xyz)
template <bool TIE_BREAK>
__device__ __forceinline__ void tri_test(f3 o, f3 d, float4 r0, float4 r1, float4 r2, uint32_t id1, HitRec& h) {
  f3 v0{r0.x, r0.y, r0.z}, e1{r0.w, r1.x, r1.y}, e2{r1.z, r1.w, r2.x}, n{r2.y, r2.z, r2.w};
  f3 tv = o - v0;
  // The signs of u, v, tt follow the sign of det = -d.n.  Instead of comparing det with 0 and selecting three negations
  // (a v_cmp whose result three VOP3 selects must wait two cycles for), the sign bit of d.n is XORed into e2.c, e1.c and
  // tt: that yields -u, +v and -tt of the oriented triangle (v = -e1.c takes the other sign), so two of the tests read
  // the other way round and HitRec keeps -u (HitRec::u: negated).  Same decisions bit for bit: x <= 0 iff -x >= 0,
  // x < 0 iff -x > 0, v - (-u) is u + v.
  const float dn = exact::dot(d, n);
  const uint32_t sm = f2u(dn) & 0x80000000u;
  const float tt = u2f(f2u(exact::dot(tv, n)) ^ sm);
  const f3 c = exact::cross(tv, d);
  const float u = u2f(f2u(exact::dot(e2, c)) ^ sm);
  const float v = u2f(f2u(exact::dot(e1, c)) ^ sm);
  const float ad = __builtin_fabsf(dn);
  // no "ad > 0" term: with ad == 0 only u == v == 0 passes, th is then +inf (tt > 0), and +inf never beats h.t
  const bool ok = (u <= 0.0f) && (v >= 0.0f) && (v - u <= ad) && (tt < 0.0f);
  if (ok) {
    const float th = RTPT_DIV_TH(-tt, ad);
    bool better = th < h.t;
    if (TIE_BREAK) better = better || (th == h.t && h.id1 != 0 && id1 < h.id1);
    if (better) {
      h.t = th;
      h.id1 = id1;
      h.u = u;
      h.v = v;
      h.ad = ad;
    }
  }
}

// Two triangles (a, b, c), (a, c, d) of one fan-triangulated face (main.cpp:416-428 via D5: every `f` line of the
// Cornell OBJ is a quad) share v0 and the edge c - a: tv = o - v0, c = tv x d and e2_A . c == e1_B . c are the SAME
// binary32 values in both tests, so the second test reuses them — 17 instead of 30 VALU, each triangle's own arithmetic
// and the ascending-id update order unchanged (bit-identical hits).  SceneView::paired says every pair (2q, 2q+1) of the
// scene is such a pair (checked bitwise on the uploaded vertices, rtpt_scene_upload).
__device__ __forceinline__ void tri_pair_test(f3 o, f3 d, float4 r0, float4 r1, float4 r2, float4 q1, float4 q2, uint32_t id1, HitRec& h) {
  const f3 v0{r0.x, r0.y, r0.z}, e1{r0.w, r1.x, r1.y}, e2{r1.z, r1.w, r2.x}, n{r2.y, r2.z, r2.w};
  const f3 e2b{q1.z, q1.w, q2.x}, nb{q2.y, q2.z, q2.w};  // B: e1_B == e2 (bitwise), v0_B == v0
  const f3 tv = o - v0;
  const f3 c = exact::cross(tv, d);
  const float e2c = exact::dot(e2, c);
  {
    const float dn = exact::dot(d, n);
    const uint32_t sm = f2u(dn) & 0x80000000u;
    const float tt = u2f(f2u(exact::dot(tv, n)) ^ sm);
    const float u = u2f(f2u(e2c) ^ sm);
    const float v = u2f(f2u(exact::dot(e1, c)) ^ sm);
    const float ad = __builtin_fabsf(dn);
    if ((u <= 0.0f) && (v >= 0.0f) && (v - u <= ad) && (tt < 0.0f)) {
      const float th = RTPT_DIV_TH(-tt, ad);
      if (th < h.t) {
        h.t = th;
        h.id1 = id1;
        h.u = u;
        h.v = v;
        h.ad = ad;
      }
    }
  }
  {
    const float dn = exact::dot(d, nb);
    const uint32_t sm = f2u(dn) & 0x80000000u;
    const float tt = u2f(f2u(exact::dot(tv, nb)) ^ sm);
    const float u = u2f(f2u(exact::dot(e2b, c)) ^ sm);
    const float v = u2f(f2u(e2c) ^ sm);  // e1_B . c
    const float ad = __builtin_fabsf(dn);
    if ((u <= 0.0f) && (v >= 0.0f) && (v - u <= ad) && (tt < 0.0f)) {
      const float th = RTPT_DIV_TH(-tt, ad);
      if (th < h.t) {
        h.t = th;
        h.id1 = id1 + 1;
        h.u = u;
        h.v = v;
        h.ad = ad;
      }
    }
  }
}

// the same for a BVH leaf: leaves are met in traversal order, so equal distances keep the lower id (tri_test<true>'s rule),
// and the two ids come from the leaf's id list
__device__ __forceinline__ void tri_pair_test_leaf(f3 o, f3 d, float4 r0, float4 r1, float4 r2, f3 e2b, f3 nb, uint32_t id1, uint32_t id1b,
                                                   HitRec& h) {
  const f3 v0{r0.x, r0.y, r0.z}, e1{r0.w, r1.x, r1.y}, e2{r1.z, r1.w, r2.x}, n{r2.y, r2.z, r2.w};  // B: e1_B == e2 (bitwise), v0_B == v0
  const f3 tv = o - v0;
  const f3 c = exact::cross(tv, d);
  const float e2c = exact::dot(e2, c);
  {
    const float dn = exact::dot(d, n);
    const uint32_t sm = f2u(dn) & 0x80000000u;
    const float tt = u2f(f2u(exact::dot(tv, n)) ^ sm);
    const float u = u2f(f2u(e2c) ^ sm);
    const float v = u2f(f2u(exact::dot(e1, c)) ^ sm);
    const float ad = __builtin_fabsf(dn);
    if ((u <= 0.0f) && (v >= 0.0f) && (v - u <= ad) && (tt < 0.0f)) {
      const float th = RTPT_DIV_TH(-tt, ad);
      bool better = th < h.t;
      better = better || (th == h.t && h.id1 != 0 && id1 < h.id1);
      if (better) {
        h.t = th;
        h.id1 = id1;
        h.u = u;
        h.v = v;
        h.ad = ad;
      }
    }
  }
  {
    const float dn = exact::dot(d, nb);
    const uint32_t sm = f2u(dn) & 0x80000000u;
    const float tt = u2f(f2u(exact::dot(tv, nb)) ^ sm);
    const float u = u2f(f2u(exact::dot(e2b, c)) ^ sm);
    const float v = u2f(f2u(e2c) ^ sm);  // e1_B . c
    const float ad = __builtin_fabsf(dn);
    if ((u <= 0.0f) && (v >= 0.0f) && (v - u <= ad) && (tt < 0.0f)) {
      const float th = RTPT_DIV_TH(-tt, ad);
      bool better = th < h.t;
      better = better || (th == h.t && h.id1 != 0 && id1b < h.id1);
      if (better) {
        h.t = th;
        h.id1 = id1b;
        h.u = u;
        h.v = v;
        h.ad = ad;
      }
    }
  }
}

// Small scenes (<= 64 triangles, the Cornell box has 32): every lane of the wave tests the same
// triangle at the same time, so the record address is wave-uniform and the loads are scalar
// (s_load_dwordx4 -> SGPR operands of the VALU ops).  No stack, no divergence, no memory latency.
__device__ __forceinline__ void closest_hit_brute(const SceneView& sc, f3 o, f3 d, HitRec& h) {
  // The records are read through the CONSTANT address space: they are never written while a kernel
  // that traces runs, and saying so keeps the loads scalar (s_load_dwordx4) even when the surrounding
  // loop also stores pixels — otherwise hipcc must assume the stores may clobber them and falls back
  // to one vector load + vmcnt(0) per triangle (measured: 2x slower).
  typedef float v4f __attribute__((ext_vector_type(4)));
  using cv4f = const __attribute__((address_space(4))) v4f;
  cv4f* rec = (cv4f*)sc.isect_id;
  const uint32_t n = sc.n_tris;
#ifndef RTPT_BRUTE_UNROLL
#define RTPT_BRUTE_UNROLL 8  // triangles whose records are fetched per batch of scalar loads; K2 at 4K: 2: 517, 4: 505, 8: 499, 16: 498 us
#endif
#ifndef RTPT_PAIR_UNROLL
#define RTPT_PAIR_UNROLL 8  // faces whose records are fetched per batch of scalar loads; K2 at 4K: 2: 384.7, 4: 378.1, 8: 376.3, 16: 376.0 us
#endif
  if (sc.paired) {  // wave-uniform
#pragma unroll RTPT_PAIR_UNROLL
    for (uint32_t i = 0; i < n; i += 2) {
      const v4f a0 = rec[3 * i], a1 = rec[3 * i + 1], a2 = rec[3 * i + 2], b1 = rec[3 * i + 4], b2 = rec[3 * i + 5];
      tri_pair_test(o, d, make_float4(a0.x, a0.y, a0.z, a0.w), make_float4(a1.x, a1.y, a1.z, a1.w), make_float4(a2.x, a2.y, a2.z, a2.w),
                    make_float4(b1.x, b1.y, b1.z, b1.w), make_float4(b2.x, b2.y, b2.z, b2.w), i + 1, h);
    }
    return;
  }
#pragma unroll RTPT_BRUTE_UNROLL
  for (uint32_t i = 0; i < n; i++) {
    const v4f a0 = rec[3 * i], a1 = rec[3 * i + 1], a2 = rec[3 * i + 2];
    tri_test<false>(o, d, make_float4(a0.x, a0.y, a0.z, a0.w), make_float4(a1.x, a1.y, a1.z, a1.w),
                    make_float4(a2.x, a2.y, a2.z, a2.w), i + 1, h);
  }
}

// Pixels of a 64 x 4 tile a wave starts on: not row w of the tile but the 16 x 4 block of columns [16 w, 16 w + 16).  Rays that
// start through neighbouring pixels walk the same BVH nodes, and a block's rays diverge later than a row's (node-loop lane
// utilisation of the primary rays 0.66 -> 0.77, profiles/r04_wave_block_ab.txt); on small scenes the block's padded
// screen rectangle meets fewer triangle bounds than a 64-pixel span's.  1: blocks, 0: rows (A/B builds).
#ifndef RTPT_WAVE_BLOCK
#define RTPT_WAVE_BLOCK 1
#endif
constexpr int kWaveW = RTPT_WAVE_BLOCK ? 16 : 64, kWaveH = RTPT_WAVE_BLOCK ? 4 : 1;  // a wave's footprint in its tile
__device__ __forceinline__ uint32_t tile_pixel(int wave, uint32_t lane) {  // index (row << 6 | column) within the tile
  if (RTPT_WAVE_BLOCK) return ((lane >> 4) << 6) | (static_cast<uint32_t>(wave) << 4) | (lane & 15u);
  return (static_cast<uint32_t>(wave) << 6) | lane;
}
// first column / row of wave `wave`'s footprint, relative to its tile
__device__ __forceinline__ int wave_x0(int wave) { return RTPT_WAVE_BLOCK ? 16 * wave : 0; }
__device__ __forceinline__ int wave_y0(int wave) { return RTPT_WAVE_BLOCK ? 0 : wave; }
// Primary rays of one wave (the kWaveW x kWaveH pixels from (x0, y0)) can only hit triangles whose padded
// screen bounds meet that rectangle.  Lane i classifies triangle i (n <= 64) and the ballot is the
// candidate set; must be called with the whole wave converged (before any early return).
__device__ __forceinline__ unsigned long long span_candidates(const TriBounds* bounds, uint32_t n, int x0, int y0) {
  const uint32_t lane = threadIdx.x & 63u;
  const TriBounds b = bounds[lane < n ? lane : 0u];
  const bool overlap = lane < n && !(b.x0 > x0 + kWaveW - 1 || b.x1 < x0 || b.y0 > y0 + kWaveH - 1 || b.y1 < y0);
  return __ballot(overlap);
}

// brute force over a wave-uniform candidate set; ascending ids, so equal-t ties keep the lower id
__device__ __forceinline__ void closest_hit_brute_set(const SceneView& sc, unsigned long long cand, f3 o, f3 d, HitRec& h) {
  typedef float v4f __attribute__((ext_vector_type(4)));
  using cv4f = const __attribute__((address_space(4))) v4f;
  cv4f* rec = (cv4f*)sc.isect_id;
  const bool paired = sc.paired != 0;
  while (cand) {
    const uint32_t i = static_cast<uint32_t>(__builtin_ctzll(cand));
    cand &= cand - 1;
    const v4f a0 = rec[3 * i], a1 = rec[3 * i + 1], a2 = rec[3 * i + 2];
    if (paired && !(i & 1u) && (cand & (1ull << (i + 1)))) {  // both triangles of a fan pair are candidates (wave-uniform)
      cand &= cand - 1;
      const v4f b1 = rec[3 * i + 4], b2 = rec[3 * i + 5];
      tri_pair_test(o, d, make_float4(a0.x, a0.y, a0.z, a0.w), make_float4(a1.x, a1.y, a1.z, a1.w), make_float4(a2.x, a2.y, a2.z, a2.w),
                    make_float4(b1.x, b1.y, b1.z, b1.w), make_float4(b2.x, b2.y, b2.z, b2.w), i + 1, h);
      continue;
    }
    tri_test<false>(o, d, make_float4(a0.x, a0.y, a0.z, a0.w), make_float4(a1.x, a1.y, a1.z, a1.w),
                    make_float4(a2.x, a2.y, a2.z, a2.w), i + 1, h);
  }
}

// General scenes: per-lane depth-first traversal of the child-pair BVH with the node stack in LDS
// (stack[level][thread]: consecutive lanes hit consecutive banks; the stack is sized to the depth of
// the tree that was built, PathtraceArgs/SceneView::stack_depth).
//
// "while-while" form: the inner loop only walks interior nodes; a leaf child that must be visited is
// pushed (or becomes `cur`) as a leaf reference and is tested in the outer loop.  In the first version
// every iteration fetched a node AND ran the triangle loops of whichever children were leaves under
// a divergent branch; PMC showed ~37 % lane utilisation on the 1.15M-triangle scene.  Here lanes that
// reach a leaf wait at the loop exit and the wave tests leaves together.
//   child reference: bit 31 clear = interior node index; bit 31 set = leaf, (first << 2) | (count - 1);
//   kBvhEmpty = absent child; kSentinel = empty stack.
// Nodes are 32 bytes (two dwordx4 per visit): boxes on the scene's 16-bit grid, rounded outward — they only
// order and cull; the triangle test is binary32 on the exact records.  A box corner q stands for origin + q * cell, so with
// inv = cell / d and oi = (origin - o) / d a slab distance is one convert + one fma: t = q * inv + oi.
constexpr uint32_t kLeafBit = 0x80000000u;
constexpr uint32_t kSentinel = 0xFFFFFFFEu;
#ifndef RTPT_GRAD_NT_STORE
#define RTPT_GRAD_NT_STORE 1
#endif
#ifndef RTPT_BVH_LEAF_RATIO
#define RTPT_BVH_LEAF_RATIO 2  // 0: 2981 us, 1: 2734, 2: 2714, 3: 2728 for K2 on the 1.15M-triangle frame (profiles/r04_bvh_ab.txt)
#endif
#ifndef RTPT_LEAF_BATCH
#define RTPT_LEAF_BATCH 2  // 1: 3.72 ms, 2: 3.65 ms, 4: 4.79 ms (registers) on the 1.15M-triangle trace
#endif

// Profiling builds (never the shipped library): -DRTPT_BVH_COUNT=1 counts the trips and active lanes of the traversal's loops
// (scripts/bvh_count.py), -DRTPT_TILE_TIMELINE=1 records when every K0 / K2 workgroup starts and ends (scripts/tile_timeline.py).
// Their definitions live in experiments/trace_instrumentation.inc; what remains here are the hooks.
#ifndef RTPT_BVH_COUNT
#define RTPT_BVH_COUNT 0
#endif
#ifndef RTPT_TILE_TIMELINE
#define RTPT_TILE_TIMELINE 0
#endif
#if RTPT_BVH_COUNT || RTPT_TILE_TIMELINE
#define RTPT_INSTR_DEVICE
#include "experiments/trace_instrumentation.inc"
#undef RTPT_INSTR_DEVICE
#endif
#if !RTPT_BVH_COUNT
#define RTPT_COUNT_TRIP(slot) do {} while (0)
#endif

template <bool PAIRS>  // the tree was built over fan pairs (bvh.hpp): a leaf is one or two of them (A, B, A, B)
__device__ __forceinline__ void closest_hit_bvh(const SceneView& sc, f3 o, f3 d, HitRec& h, uint32_t* stack,
                                               int tid, int nt = kThreads, int bucket = 0) {
  if (__builtin_isunordered(o.x, d.x) || __builtin_isunordered(o.y, d.y) || __builtin_isunordered(o.z, d.z)) return;  // see below
#if RTPT_BVH_COUNT
  RTPT_COUNT_BEGIN(bucket, d);
#endif
  // A ray with a NaN component cannot hit anything (every comparison of tri_test fails, D7) — but min/max drop
  // NaNs, so every box would "pass" and that one lane would walk all of the scene: on the 1.15M-triangle lattice
  // single frames took 38-47 ms instead of 6.3 because of a handful of such paths (the test at the top of the function).
  // A direction component that is exactly 0 (axis-parallel rays do occur: d = (0,0,1) was measured) would make
  // that axis' slab distances inf - inf = NaN, which min/max drop: the axis stops culling and the ray visits
  // every node ahead of it (34 755 node visits for one ray, 45 ms for the frame).  With |d| clamped to 1e-20 the
  // distances are +-huge with the right signs — the padded boxes keep every face of a box the ray can hit
  // further than the rounding error from the ray's coordinate — so the parallel axis culls correctly.
  auto nz = [](float v) { return __builtin_fabsf(v) < 1e-20f ? __builtin_copysignf(1e-20f, v) : v; };
  const f3 rd{fast::rcp_(nz(d.x)), fast::rcp_(nz(d.y)), fast::rcp_(nz(d.z))};
  // x = origin + q * cell  =>  t = (x - o) / d = q * (cell / d) + (origin - o) / d
  // the grid lives in device memory (a device-side refit rewrites it without the host knowing the numbers): six scalar
  // loads through the constant address space
  using cflt = const __attribute__((address_space(4))) float;
  cflt* gr = (cflt*)sc.bvh_grid;
  const f3 inv{gr[3] * rd.x, gr[4] * rd.y, gr[5] * rd.z};
  const f3 oi{(gr[0] - o.x) * rd.x, (gr[1] - o.y) * rd.y, (gr[2] - o.z) * rd.z};
  struct { uint32_t x, y, z; } const rot{rd.x < 0.0f ? 16u : 0u, rd.y < 0.0f ? 16u : 0u, rd.z < 0.0f ? 16u : 0u};
  int sp = 0;
  uint32_t cur = 0;  // root pair
  // entries [0, stack_lds) in LDS, the rest in global memory (SceneView::stack_spill)
  const int lds_levels = static_cast<int>(sc.stack_lds);
  const size_t spill_stride = static_cast<size_t>(gridDim.x) * gridDim.y * nt;
  uint32_t* const spill = sc.stack_spill + (static_cast<size_t>(blockIdx.y) * gridDim.x + blockIdx.x) * nt + tid;
  // the LDS half through an LDS-typed pointer: a generic one lets the compiler fold the two halves of a pop into one
  // flat_load_dword of a selected address
  using lds_u32 = __attribute__((address_space(3))) uint32_t;
  lds_u32* const stack_lds = (lds_u32*)stack;
  auto push = [&](uint32_t v) {
    if (sp < lds_levels)
      stack_lds[sp * nt + tid] = v;
    else
      spill[static_cast<size_t>(sp - lds_levels) * spill_stride] = v;
    sp++;
  };
  auto pop = [&]() -> uint32_t {
    if (sp > 0) {
      sp--;
      // always an LDS read (slot 0 when the entry is in the global half), then the rare global read under its own branch;
      // as one conditional expression over generic pointers the two became ONE flat_load_dword of a selected address,
      // and every pop went through the vector-memory pipe
      uint32_t v = stack_lds[(sp < lds_levels ? sp : 0) * nt + tid];
      if (__builtin_expect(sp >= lds_levels, 0)) v = spill[static_cast<size_t>(sp - lds_levels) * spill_stride];
      return v;
    }
    return kSentinel;
  };
  auto test_leaf = [&](uint32_t ref) {
    const uint32_t first = (ref & ~kLeafBit) >> 2, cnt = (ref & 3u) + 1u;
    if (PAIRS) {
      // (with kBvhMaxLeaf <= 2 a leaf is one pair: the loop is a single pass the compiler sees)
      for (uint32_t j = 0; j < (kBvhMaxLeaf <= 2 ? 1u : cnt); j += 2) {
#if RTPT_BVH_COUNT
        RTPT_COUNT_TRIP(2);
        my_leaves++;
#endif
        // one 80-byte pair record (SceneView::isect_leaf); base + 32-bit byte offset (rtpt_scene_upload bounds the scene)
        const float4* r = reinterpret_cast<const float4*>(reinterpret_cast<const unsigned char*>(sc.isect_leaf) + ((first + j) >> 1) * 80u);
        const float4 p0 = r[0], p1 = r[1], p2 = r[2], p3 = r[3], p4 = r[4];
        tri_pair_test_leaf(o, d, p0, p1, p2, f3{p3.x, p3.y, p3.z}, f3{p3.w, p4.x, p4.y}, f2u(p4.z) + 1, f2u(p4.w) + 1, h);
      }
      return;
    }
#if RTPT_LEAF_BATCH > 1
    // fetch the records of RTPT_LEAF_BATCH triangles before testing any of them: one memory round trip per
    // batch instead of one per triangle (indices past the leaf are clamped to its last triangle and skipped)
    for (uint32_t j0 = 0; j0 < cnt; j0 += RTPT_LEAF_BATCH) {
      float4 rec[RTPT_LEAF_BATCH][3];
      uint32_t id[RTPT_LEAF_BATCH];
#pragma unroll
      for (uint32_t u = 0; u < RTPT_LEAF_BATCH; u++) {
        const uint32_t j = j0 + u < cnt ? j0 + u : cnt - 1u;
        const float4* r = sc.isect_leaf + 3 * static_cast<size_t>(first + j);
        rec[u][0] = r[0];
        rec[u][1] = r[1];
        rec[u][2] = r[2];
        id[u] = sc.leaf_ids[first + j];
      }
#pragma unroll
      for (uint32_t u = 0; u < RTPT_LEAF_BATCH; u++)
        if (j0 + u < cnt) tri_test<true>(o, d, rec[u][0], rec[u][1], rec[u][2], id[u] + 1, h);
    }
#else
    for (uint32_t j = 0; j < cnt; j++) {
      const float4* r = sc.isect_leaf + 3 * static_cast<size_t>(first + j);
      tri_test<true>(o, d, r[0], r[1], r[2], sc.leaf_ids[first + j] + 1, h);
    }
#endif
  };
  auto node_step = [&]() {
#if RTPT_BVH_COUNT
    RTPT_COUNT_TRIP(0);
    my_nodes++;
#endif
    // base + 32-bit byte offset (a tree has < 2^27 nodes): the loads take the scalar base and a 32-bit vector offset instead
    // of a 64-bit address computed per step
    const uint4* np = reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned char*>(sc.nodes) + (cur << 5));
    const uint4 a = np[0], b = np[1];
    const uint32_t cl = b.z, cr = b.w;
    const float tb = h.t;
    float tl, tr;
    // Every dword of the six holds one axis of one box as (min16 | max16 << 16).  Rotated by 16 where the ray runs against
    // the axis (rot, per ray), its low half is the plane the ray meets first and its high half the one it leaves through:
    // the min / max pair per axis that sorted the two distances (12 per node) becomes one v_alignbit_b32 (6 per node).
    // Same values: fma(q, inv, oi) is monotonic in q, increasing for inv > 0 and decreasing for inv < 0 (|d| is clamped away
    // from 0 above, so inv is never 0 or NaN), so min(t(min), t(max)) IS t of the half selected here, bit for bit.
    auto near_far = [&](uint32_t w, uint32_t rot_, float inv_, float oi_, float& tn_, float& tf_) {
      const uint32_t q = __builtin_amdgcn_alignbit(w, w, rot_);
      tn_ = fmaf_(static_cast<float>(q & 0xFFFFu), inv_, oi_);
      tf_ = fmaf_(static_cast<float>(q >> 16), inv_, oi_);
    };
    float n0, n1, n2, f0, f1, f2;
    near_far(a.x, rot.x, inv.x, oi.x, n0, f0);
    near_far(a.y, rot.y, inv.y, oi.y, n1, f1);
    near_far(a.z, rot.z, inv.z, oi.z, n2, f2);
    tl = __builtin_fmaxf(__builtin_fmaxf(n0, n1), __builtin_fmaxf(n2, 0.0f));
    const bool sl = tl <= __builtin_fminf(__builtin_fminf(f0, f1), __builtin_fminf(f2, tb));
    near_far(a.w, rot.x, inv.x, oi.x, n0, f0);
    near_far(b.x, rot.y, inv.y, oi.y, n1, f1);
    near_far(b.y, rot.z, inv.z, oi.z, n2, f2);
    tr = __builtin_fmaxf(__builtin_fmaxf(n0, n1), __builtin_fmaxf(n2, 0.0f));
    const bool sr = tr <= __builtin_fminf(__builtin_fminf(f0, f1), __builtin_fminf(f2, tb));
    const bool hl = sl & (cl != kBvhEmpty), hr = sr & (cr != kBvhEmpty);
    if (hl && hr) {
      const bool left_first = tl <= tr;
      push(left_first ? cr : cl);
      cur = left_first ? cl : cr;
    } else if (hl) {
      cur = cl;
    } else if (hr) {
      cur = cr;
    } else {
      cur = pop();
    }
  };
#if RTPT_BVH_LEAF_RATIO
  // while-while with an early hand-over: the node loop stops not only when every lane holds a leaf (or is done) but as
  // soon as the lanes waiting with a leaf outnumber the lanes still walking RTPT_BVH_LEAF_RATIO to one — the few walkers
  // sit out one leaf test instead of the many waiting out the walkers' remaining steps
  while (cur != kSentinel) {
    while (true) {
      const bool walk = !(cur & kLeafBit);
      const unsigned long long mw = __ballot(walk);
      if (!mw) break;
      // lanes of this loop that are not walking hold a leaf (or have just run out of nodes): they wait
      const int nw = __builtin_popcountll(mw), na = __builtin_popcountll(__ballot(true));
      if (na - nw >= RTPT_BVH_LEAF_RATIO * nw) break;
      if (walk) node_step();
    }
    if ((cur & kLeafBit) && cur != kSentinel) {  // a leaf
      test_leaf(cur);
      cur = pop();
    }
  }
#else
  while (cur != kSentinel) {
    while (!(cur & kLeafBit)) node_step();  // interior (kSentinel has bit 31 set)
    if (cur != kSentinel) {  // a leaf
      test_leaf(cur);
      cur = pop();
    }
  }
#endif
}

template <int BVH>
__device__ __forceinline__ void closest_hit(const SceneView& sc, f3 o, f3 d, HitRec& h, uint32_t* stack, int tid,
                                            int nt = kThreads, int bucket = 0) {
  // BVH: 0 brute force, 1 BVH over triangles, 2 BVH over fan pairs (SceneView::leaf_pairs) — a kernel holds one leaf routine
  if (BVH)
    closest_hit_bvh<BVH == 2>(sc, o, d, h, stack, tid, nt, bucket);
  else
    closest_hit_brute(sc, o, d, h);
}

// v0*b0 + v1*b1 + v2*b2 (raytrace.comp.glsl:137) := fma(v2,b2, fma(v1,b1, v0*b0))
__device__ __forceinline__ f3 bary_point(f3 v0, f3 v1, f3 v2, float b0, float b1, float b2) {
  return f3{fmaf_(v2.x, b2, fmaf_(v1.x, b1, v0.x * b0)), fmaf_(v2.y, b2, fmaf_(v1.y, b1, v0.y * b0)),
            fmaf_(v2.z, b2, fmaf_(v1.z, b1, v0.z * b0))};
}

// ------------------------------------------------------------------------------------------
// scene records
// ------------------------------------------------------------------------------------------
__global__ void k_scene_prepare(ScenePrepArgs a) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.n_tris) return;
  {
    const float* t = a.tris + 9 * static_cast<size_t>(i);
    f3 v0 = ld3(t), v1 = ld3(t + 3), v2 = ld3(t + 6);
    f3 e1 = v1 - v0, e2 = v2 - v0;
    f3 nn = exact::cross(e1, e2);
    a.isect_id[3 * i] = make_float4(v0.x, v0.y, v0.z, e1.x);
    a.isect_id[3 * i + 1] = make_float4(e1.y, e1.z, e2.x, e2.y);
    a.isect_id[3 * i + 2] = make_float4(e2.z, nn.x, nn.y, nn.z);
    f3 n = exact::normalize(nn);  // raytrace.comp.glsl:150
    a.shade[3 * i] = make_float4(v0.x, v0.y, v0.z, n.x);
    a.shade[3 * i + 1] = make_float4(v1.x, v1.y, v1.z, n.y);
    a.shade[3 * i + 2] = make_float4(v2.x, v2.y, v2.z, n.z);
  }
  if (!a.leaf_pairs) {
    uint32_t id = a.leaf_order[i];  // leaf slot i holds triangle id (ids stay in leaf_order)
    const float* t = a.tris + 9 * static_cast<size_t>(id);
    f3 v0 = ld3(t), v1 = ld3(t + 3), v2 = ld3(t + 6);
    f3 e1 = v1 - v0, e2 = v2 - v0;
    f3 nn = exact::cross(e1, e2);
    a.isect_leaf[3 * i] = make_float4(v0.x, v0.y, v0.z, e1.x);
    a.isect_leaf[3 * i + 1] = make_float4(e1.y, e1.z, e2.x, e2.y);
    a.isect_leaf[3 * i + 2] = make_float4(e2.z, nn.x, nn.y, nn.z);
  } else if (!(i & 1u) && i + 1 < a.n_tris) {
    // leaf slots (i, i + 1) hold the two triangles A = (a, b, c), B = (a, c, d) of one fan pair: one 80-byte record, each
    // triangle's edges and plane normal computed exactly as for its own 48-byte record (SceneView::isect_leaf)
    const uint32_t ia = a.leaf_order[i], ib = a.leaf_order[i + 1];
    const float* ta = a.tris + 9 * static_cast<size_t>(ia);
    const float* tb = a.tris + 9 * static_cast<size_t>(ib);
    const f3 v0 = ld3(ta), e1 = ld3(ta + 3) - v0, e2 = ld3(ta + 6) - v0, nn = exact::cross(e1, e2);
    const f3 v0b = ld3(tb), e1b = ld3(tb + 3) - v0b, e2b = ld3(tb + 6) - v0b, nb = exact::cross(e1b, e2b);
    float4* r = a.isect_leaf + 5 * static_cast<size_t>(i >> 1);
    r[0] = make_float4(v0.x, v0.y, v0.z, e1.x);
    r[1] = make_float4(e1.y, e1.z, e2.x, e2.y);
    r[2] = make_float4(e2.z, nn.x, nn.y, nn.z);
    r[3] = make_float4(e2b.x, e2b.y, e2b.z, nb.x);
    r[4] = make_float4(nb.y, nb.z, u2f(ia), u2f(ib));
  }
}

// visibility.vert.glsl:24 + visibility.geom.glsl:57-59: LUT[t+1] = model * (v0,v1,v2), written for
// every triangle (the geometry stage precedes clipping).  Also the per-id normal table the filter
// gathers instead of re-deriving normalize(cross) from the LUT at each of its 10 reads per pixel
// (temporalFiltering.comp.glsl:80-91) — same values, computed once per triangle.
__global__ void k_lut(LutArgs a) {
  uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t == 0) {
    a.lut[0] = a.lut[1] = a.lut[2] = make_float4(0.f, 0.f, 0.f, 0.f);
    f3 n{0.f, 0.f, 1.f};  // temporalFiltering.comp.glsl:83
    a.normal_tab[0] = make_float4(n.x, n.y, n.z, exact::powi(glsl_max(0.0f, exact::dot(n, n)), a.sigma_n));
    a.area_tab[0] = make_float4(0.f, 0.f, 0.f, 0.f);  // never read: id 0 has no triangle (temporalGradient.comp.glsl:128-131)
  }
  if (t >= a.n_tris) return;
  f3 v[3];
  for (int k = 0; k < 3; k++) {
    f3 p = xyz(a.shade[3 * t + k]);
    v[k] = f3{exact::mat_row_point(a.model, 0, p), exact::mat_row_point(a.model, 1, p), exact::mat_row_point(a.model, 2, p)};
    a.lut[3 * (t + 1) + k] = make_float4(v[k].x, v[k].y, v[k].z, 0.f);
  }
  f3 n = exact::normalize(exact::cross(v[1] - v[0], v[2] - v[0]));  // :90
  a.normal_tab[t + 1] = make_float4(n.x, n.y, n.z, exact::powi(glsl_max(0.0f, exact::dot(n, n)), a.sigma_n));
  a.area_tab[t + 1] = make_float4(tri_area(v[0], v[1], v[2]), 0.f, 0.f, 0.f);  // temporalGradient.comp.glsl:60, once per triangle
}

// pow(max(0, dot(n_p, n_q)), sigma_n) for every id pair (temporalFiltering.comp.glsl:62), small scenes
__global__ void k_pair_weights(LutArgs a) {
  const uint32_t np = a.n_tris + 1;
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= np * np) return;
  const uint32_t p = i / np, q = i - p * np;
  const f3 n_p = xyz(a.normal_tab[p]), n_q = xyz(a.normal_tab[q]);
  a.pair_tab[i] = exact::powi(glsl_max(0.0f, exact::dot(n_p, n_q)), a.sigma_n);
}

// ------------------------------------------------------------------------------------------
// K1 temporal gradient
// ------------------------------------------------------------------------------------------
// temporalGradient.comp.glsl:71-101
__device__ __forceinline__ f3 phong(f3 p, f3 n, f3 cam, f3 lpos, f3 lcol) {
  f3 ldir = exact::normalize(lpos - p);
  f3 ambient = lcol * 0.1f;
  float diff = glsl_max(exact::dot(n, ldir), 0.0f);
  f3 diffuse = lcol * diff;
  f3 vdir = exact::normalize(cam - p);
  f3 I = -ldir;
  float two_ndi = 2.0f * exact::dot(n, I);
  f3 rdir{fmaf_(-two_ndi, n.x, I.x), fmaf_(-two_ndi, n.y, I.y), fmaf_(-two_ndi, n.z, I.z)};
  float spec = exact::powi(glsl_max(exact::dot(vdir, rdir), 0.0f), 128);
  f3 specular = lcol * (0.5f * spec);
  return ((ambient + diffuse) + specular) * 0.7f;
}

// temporalGradient.comp.glsl:128-167 for one pixel: relative change of the Phong shade of the visible surface point between
// the previous and the current light (and pose)
__device__ __forceinline__ float gradient_lambda(uint32_t id, f3 wp, const float4* lut, const float4* lut_prev, const float4* normal_tab,
                                                 const float4* area_tab, f3 cam, f3 light, f3 light_prev, f3 color, f3 color_prev) {
  if (id == 0) return 0.0f;  // :128-131
  f3 va = xyz(lut[3 * id]), vb = xyz(lut[3 * id + 1]), vc = xyz(lut[3 * id + 2]);
  // :142 normalize(cross(vb - va, vc - va)) — k_lut computed exactly that from exactly these vertices, once per triangle
  // (normal_tab[id]); per pixel it is 36 VALU of a VALU-bound kernel
  f3 nrm = xyz(normal_tab[id]);
  // :143; the whole triangle's area (:60) comes from k_lut's table: same vertices, same arithmetic, once per triangle
  f3 bc = bary_coords_at(wp, va, vb, vc, area_tab[id].x);
  f3 pa = xyz(lut_prev[3 * id]), pb = xyz(lut_prev[3 * id + 1]), pc = xyz(lut_prev[3 * id + 2]);
  f3 wpp = bary_mix(bc, pa, pb, pc);                          // :153
  f3 cur = phong(wp, nrm, cam, light, color);                 // :158
  f3 prv = phong(wpp, nrm, cam, light_prev, color_prev);      // :161 (current normal!)
  f3 tg = cur - prv;
  float delta = glsl_max(exact::length(cur), exact::length(prv));  // :166
  // (hipcc's division on purpose: with the light at rest the numerator is +0 for every pixel, which exact::div_ would hand
  // to the long path after paying for the short one)
  return glsl_min(1.0f, exact::length(tg) / delta);                // :167
}

__device__ __forceinline__ void store_gradient(float4* grad, size_t i, float lam) {
#if RTPT_GRAD_NT_STORE
  // nothing on the reference's path reads the gradient again (its consumer is commented out,
  // temporalFiltering.comp.glsl:247-248): keep the 133 MB out of the caches the filter passes need
  typedef float v4f_ __attribute__((ext_vector_type(4)));
  v4f_ g4 = {lam, lam, lam, 0.0f};
  __builtin_nontemporal_store(g4, reinterpret_cast<v4f_*>(grad + i));
#else
  grad[i] = make_float4(lam, lam, lam, 0.0f);
#endif
}

// ------------------------------------------------------------------------------------------
// K0 G-buffer
// ------------------------------------------------------------------------------------------
// dvx[x] = ((2 (x + .5) - W) / W) / P00 for every column, dvy[y] likewise for every frame row (see k_gbuffer)
__global__ void k_ray_tables(int W, int H, float p00, float p11, float* dvx, float* dvy) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < W) {
    const float fw = static_cast<float>(W);
    dvx[i] = (fmaf_(2.0f, static_cast<float>(i) + 0.5f, -fw) / fw) / p00;
  }
  if (i < H) {
    const float fh = static_cast<float>(H);
    dvy[i] = (fmaf_(2.0f, static_cast<float>(i) + 0.5f, -fh) / fh) / p11;
  }
}

// K0 (+ K1 when a.grad_on) for pixel (x, y), which must lie inside the frame and inside a's rows; `cand` = span_candidates of
// the pixel's wave (brute force with culling only).  alpha_image != NULL (the fused K0 + K1 + K2 launch): the depth also goes
// into the alpha of that image's pixel for rows [ay0, ay1) — the traced image is "rgbd" (atrous.hip) and the tracing
// workgroups of that launch store the colour's 12 bytes only.
template <int BVH>
__device__ __forceinline__ void gbuffer_pixel(const GbufferArgs& a, int x, int y, unsigned long long cand, uint32_t* stack, int tid, int nt,
                                              float4* alpha_image = nullptr, int ay0 = 0, int ay1 = 0) {
  // view-space direction of the pixel-centre ray: (ndc.x / P00, ndc.y / P11, -1) with ndc = (2 (x + .5) - W) / W.  Each
  // component is a function of the column or of the row alone: k_ray_tables evaluates the two divisions per column / row
  // once (same operands, same correctly-rounded operations), K0 reads them back — 4 divisions less per pixel
  f3 dv{a.dvx[x], a.dvy[y], -1.0f};
  f3 c0 = ld3(a.c0), c1 = ld3(a.c1), c2 = ld3(a.c2);
  f3 d = exact::normalize(f3{exact::dot(c0, dv), exact::dot(c1, dv), exact::dot(c2, dv)});
  f3 o = ld3(a.org);
  HitRec h{a.tmax, 0u, 0.f, 0.f, 1.f};
  if (!BVH && a.cull)
    closest_hit_brute_set(a.scene, cand, o, d, h);
  else
    closest_hit<BVH>(a.scene, o, d, h, stack, tid, nt);
  const size_t i = static_cast<size_t>(y - a.g.row_base) * a.g.W + x;
  a.vis[i] = h.id1;  // visibility.frag.glsl:23
  if (a.normals) a.normals[i] = a.normal_tab[h.id1];
  f3 wp{0.f, 0.f, 0.f};
  float dep = 1.0f;  // clear depth  main.cpp:1421
  if (h.id1) {
    float b1 = RTPT_DIV_SH(-h.u, h.ad), b2 = RTPT_DIV_SH(h.v, h.ad);
    float b0 = 1.0f - b1 - b2;
    const float4* s = a.scene.shade + 3 * static_cast<size_t>(h.id1 - 1);
    wp = bary_point(xyz(s[0]), xyz(s[1]), xyz(s[2]), b0, b1, b2);
    a.worldpos[i] = make_float4(wp.x, wp.y, wp.z, 1.0f);
    float cz = exact::mat_row_point(a.PV, 2, wp), cw = exact::mat_row_point(a.PV, 3, wp);
    dep = exact::div_(cz, cw);
  } else {
    a.worldpos[i] = make_float4(0.f, 0.f, 0.f, 1.0f);  // clear colour main.cpp:1420
  }
  a.depth[i] = dep;
  if (alpha_image && y >= ay0 && y < ay1) reinterpret_cast<float*>(alpha_image + i)[3] = dep;
  if (a.grad_on && y >= a.grad_y0 && y < a.grad_y1) {
    // K1 (temporalGradient.comp.glsl:104-172) on the values K0 just stored — the same bits it would load back (the world
    // position is computed ONCE: the stores above may alias the vertex records as far as the compiler knows, so a
    // second evaluation was a second evaluation)
    store_gradient(a.grad, i, gradient_lambda(h.id1, wp, a.lut, a.lut_prev, a.normal_tab, a.area_tab, ld3(a.g_cam), ld3(a.g_light), ld3(a.g_light_prev),
                                              ld3(a.g_color), ld3(a.g_color_prev)));
  }
}

// the 64 x 4-pixel tile (bx, by) of a's rows (k_gbuffer's workgroup, and the G-buffer workgroups of the fused launch)
template <int BVH>
__device__ __forceinline__ void gbuffer_tile(const GbufferArgs& a, uint32_t bx, uint32_t by, uint32_t* stack, float4* alpha_image = nullptr, int ay0 = 0,
                                             int ay1 = 0) {
  const int tid = threadIdx.y * kBlockX + threadIdx.x;
  const uint32_t tp = tile_pixel(static_cast<int>(threadIdx.y), threadIdx.x);
  const int x = static_cast<int>(bx) * kBlockX + static_cast<int>(tp & 63u);
  const int y = a.g.y0 + static_cast<int>(by) * kBlockY + static_cast<int>(tp >> 6);
  unsigned long long cand = 0;
  if (!BVH && a.cull)
    cand = span_candidates(a.bounds, a.scene.n_tris, static_cast<int>(bx) * kBlockX + wave_x0(static_cast<int>(threadIdx.y)),
                           a.g.y0 + static_cast<int>(by) * kBlockY + wave_y0(static_cast<int>(threadIdx.y)));
  if (x >= a.g.W || y >= a.g.y1) return;
  gbuffer_pixel<BVH>(a, x, y, cand, stack, tid, kThreads, alpha_image, ay0, ay1);
}

template <int BVH>
__global__ __launch_bounds__(kThreads) void k_gbuffer(GbufferArgs a) {
  extern __shared__ uint32_t stack[];  // BVH: stack_depth x 256 entries (dynamic); unused otherwise
#if RTPT_TILE_TIMELINE
  TimelineScope tl_(0, blockIdx.y * gridDim.x + blockIdx.x);
#endif
  gbuffer_tile<BVH>(a, blockIdx.x, blockIdx.y, stack);
}

__global__ __launch_bounds__(kThreads) void k_gradient(GradientArgs a) {
  const int x = blockIdx.x * kBlockX + threadIdx.x;
  const int y = a.g.y0 + blockIdx.y * kBlockY + threadIdx.y;
  if (x >= a.g.W || y >= a.g.y1) return;  // D10: bounds check first
  const size_t i = static_cast<size_t>(y - a.g.row_base) * a.g.W + x;
  const uint32_t id = a.vis[i];
  const f3 wp = id ? xyz(a.worldpos[i]) : f3{0.f, 0.f, 0.f};
  store_gradient(a.grad, i, gradient_lambda(id, wp, a.lut, a.lut_prev, a.normal_tab, a.area_tab, ld3(a.cam), ld3(a.light), ld3(a.light_prev), ld3(a.color),
                                            ld3(a.color_prev)));
}

// ------------------------------------------------------------------------------------------
// K2 path trace
// ------------------------------------------------------------------------------------------
// raytrace.comp.glsl:168-198 — only the boolean is consumed (:226), and the sphere is tested
// regardless of the triangle hit distance (the light is never occluded)
__device__ __forceinline__ bool ray_hits_light(f3 o, f3 d, f3 c, float r2) {
  f3 oc = o - c;
  float a = exact::dot(d, d);
  float b = 2.0f * exact::dot(oc, d);
  float cc = exact::dot(oc, oc) - r2;
  float disc = fmaf_(b, b, -((4.0f * a) * cc));
  if (disc < 0.0f) return false;
  float sq = exact::sqrt_(disc);
  // :190-197 "t1 > 0 || t2 > 0" with t1 = (-b - sq) / 2a, t2 = (-b + sq) / 2a.  sq >= 0 and 2a >= 0, and both the
  // addition and the correctly-rounded division are monotonic, so t1 <= t2 whenever neither is NaN: t1 > 0 implies
  // t2 > 0, and the disjunction IS "t2 > 0" (NaN operands make both comparisons false either way; 2a == 0 gives +-inf /
  // NaN with the same signs).  One division less per path segment.
  float t2 = RTPT_DIV_SH(-b + sq, 2.0f * a);
  return t2 > 0.0f;
}

__device__ __forceinline__ f3 sky_color(f3 d) {  // raytrace.comp.glsl:95-107
  if (d.y > 0.0f) {
    float t = d.y, it = 1.0f - t;
    return f3{fmaf_(0.25f, t, it), fmaf_(0.5f, t, it), fmaf_(1.0f, t, it)};
  }
  return f3{0.03f, 0.03f, 0.03f};
}

// the colour of a pixel of an "rgbd" image without touching its alpha (one global_store_dwordx3)
__device__ __forceinline__ void store_rgb(float4* px, f3 c) {
  typedef float v3f_ __attribute__((ext_vector_type(3)));
  *reinterpret_cast<v3f_*>(px) = v3f_{c.x, c.y, c.z};
}

// K2 tile: 64 x kPtRows pixels per workgroup.  More paths compacted together shrink the share of the half-empty
// last wave of every later segment, but every compaction is a workgroup barrier on the slowest wave.  Swept at
// 4K (Cornell 4 segments / 1.15M triangles 8 segments, k_pathtrace): 2 rows 862 us / 4.18 ms, 3 rows 569 / 4.04,
// 4 rows 517 / 3.99, 8 rows 540 / 4.23, 16 rows 728 / 5.80.
#ifndef RTPT_PT_ROWS
#define RTPT_PT_ROWS 4
#endif
constexpr int kPtRows = RTPT_PT_ROWS;
constexpr int kPtThreads = kBlockX * kPtRows;
// One path segment after its closest-hit query (raytrace.comp.glsl:226-267): the unoccluded light test, the sky, or a
// diffuse bounce.  Returns true when the path ended (its colour is then `acc`).
__device__ __forceinline__ bool shade_segment(const PathtraceArgs& a, const HitRec& h, uint32_t seg, f3 light_c, f3& o, f3& d,
                                              f3& acc, uint32_t& rng) {
  if (ray_hits_light(o, d, light_c, a.light_r2)) {  // :226
    acc = acc * (seg == 0 ? ld3(a.light_col_first) : ld3(a.light_col));  // :229,:233
    return true;
  }
  if (h.id1 == 0) {
    acc = acc * sky_color(d);  // :266
    return true;
  }
  const float4* s = a.scene.shade + 3 * static_cast<size_t>(h.id1 - 1);
  float4 s0 = s[0], s1 = s[1], s2 = s[2];
  float b1 = RTPT_DIV_SH(-h.u, h.ad), b2 = RTPT_DIV_SH(h.v, h.ad);
  float b0 = 1.0f - b1 - b2;                                       // :134
  f3 pos = bary_point(xyz(s0), xyz(s1), xyz(s2), b0, b1, b2);      // :137
  f3 n{s0.w, s1.w, s2.w};                                          // :150 (precomputed per triangle)
  f3 alb = (n.x > 0.99f) ? f3{1.f, 0.f, 0.f} : ((-n.x > 0.99f) ? f3{0.f, 1.f, 0.f} : f3{0.7f, 0.7f, 0.7f});  // :155-163
  if (a.scene.materials) {
    // SURVEY 8(f) rank 4, not reference behaviour: a material library replaces the normal-keyed colours; a surface
    // with emission ends the path like the analytic light does (:226-234), throughput *= Ke
    const float4* m = a.scene.materials + 2 * static_cast<size_t>((h.id1 - 1) % a.scene.n_base_tris);
    const float4 m0 = m[0], m1 = m[1];
    if (m1.w != 0.0f) {
      acc = acc * f3{m1.x, m1.y, m1.z};
      return true;
    }
    alb = f3{m0.x, m0.y, m0.z};
  }
  acc = acc * alb;                                                 // :244
  if (!(exact::dot(n, d) < 0.0f)) n = -n;                          // :247 faceforward
  o = f3{fmaf_(a.ray_offset, n.x, pos.x), fmaf_(a.ray_offset, n.y, pos.y), fmaf_(a.ray_offset, n.z, pos.z)};  // :250
  float st_, ct;
  exact::sincos2pi(exact::rng_next(rng), st_, ct);                 // :256
  float u = fmaf_(2.0f, exact::rng_next(rng), -1.0f);              // :257
  float r = exact::sqrt_(fmaf_(-u, u, 1.0f));                      // :258
  d = exact::normalize(f3{fmaf_(r, ct, n.x), fmaf_(r, st_, n.y), n.z + u});  // :259-261
  return seg + 1 >= a.max_segments;  // budget exhausted: the path returns its throughput (:270)
}

// Hand-over of the paths a launch did not finish (PathtraceArgs::seg_end < max_segments) to the next launch: a
// workgroup appends ALL of its survivors with one atomic (the per-wave counts are summed through LDS).  One atomic
// per wave on the same counter serialised badly (tile kernel 516 -> 814 us with a window of 2).  The queue can be
// split into kPathQueues regions with a counter each (a region then holds the survivors of at most
// ceil(blocks / kPathQueues) workgroups: PathtraceArgs::queue_region records); once the atomics are per workgroup
// one region is the fastest, so that is what ships.
__device__ __forceinline__ void enqueue_paths(const PathtraceArgs& a, uint32_t region, uint32_t* wave_cnt, uint32_t* bcast,
                                              bool alive, int wave, uint32_t lane, uint32_t pixg, uint32_t rng, f3 o, f3 d, f3 acc) {
  const unsigned long long m = __ballot(alive);
  if (lane == 0) wave_cnt[wave] = static_cast<uint32_t>(__builtin_popcountll(m));
  __syncthreads();
  uint32_t before = 0, total = 0;
#pragma unroll
  for (int w = 0; w < kPtRows; w++) {
    const uint32_t c = wave_cnt[w];
    if (w < wave) before += c;
    total += c;
  }
  if (wave == 0 && lane == 0) *bcast = total ? atomicAdd(a.q_out_count + region, total) : 0u;
  __syncthreads();
  if (alive) {
    const uint32_t slot = region * a.queue_region + *bcast + before +
                          __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(m >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(m), 0u));
    float4* q = reinterpret_cast<float4*>(a.q_out) + 3 * static_cast<size_t>(slot);
    q[0] = make_float4(__uint_as_float(pixg), __uint_as_float(rng), o.x, o.y);
    q[1] = make_float4(o.z, d.x, d.y, d.z);
    q[2] = make_float4(acc.x, acc.y, acc.z, 0.0f);
  }
  __syncthreads();  // wave_cnt / bcast may be reused right away
}

// K2 tile kernel with optional per-segment compaction (PathtraceArgs::compact).
//
// A 256-thread block owns a 64x4 pixel tile and advances all of its paths one segment at a time
// (the segment index is block-uniform).  The PRIMARY segment runs with the lane <-> pixel mapping
// intact, so a wave's rays are coherent and only the triangles whose screen bounds meet the wave's
// span are tested.  After every segment the surviving paths (light hit, sky and the segment budget
// end paths; at 4 segments the Cornell box averages 2.25 queries per pixel) are compacted to the
// front of the block through LDS, so secondary segments — incoherent anyway — run in full waves and
// emptied waves skip the segment entirely: 16 wave-segments per tile become ~10.  A path's RNG stream
// depends only on (x, y, frame, batch) (raytrace.comp.glsl:297) and every path runs the reference's
// loop body unchanged, so the image is independent of the schedule (bit-exact parity with the oracle).
// Measured in one process at 4K on the Cornell box (us, 4 / 8 / 32 segments): without compaction
// 503 / 985 / 3410, with 495 / 925 / 2810 — at 4 segments most waves retire early anyway (sky and the
// unoccluded light end whole waves), the gain grows with the path length.  (A per-wave path
// regeneration scheme — finished lanes pick up new pixels — was also built and measured: it mixes
// primary and secondary rays in every wave, loses the coherent, culled primary segment and ran
// 1.5x slower at 4 segments; it is not kept.)
struct PathState {  // SoA in LDS, one slot per thread
  uint32_t pix[kPtThreads];   // local pixel index (ty*64 + tx)
  uint32_t rng[kPtThreads];
  float ox[kPtThreads], oy[kPtThreads], oz[kPtThreads];
  float dx[kPtThreads], dy[kPtThreads], dz[kPtThreads];
  float ar[kPtThreads], ag[kPtThreads], ab[kPtThreads];
};

// The kernel needs its 8 waves per SIMD to hide the scalar-load latency of the brute-force loop; left alone the
// compiler keeps 106 SGPRs of kernel arguments live (7 waves).  Measured at 4K: 6 / 7 (compiler's choice) / 8 waves
// 506 / 505 / 479 us (a few dwords of scalar spills are cheaper than the lost wave).
#ifndef RTPT_PT_WAVES
#define RTPT_PT_WAVES 8
#endif
#ifndef RTPT_PT_CENTER_OUT
#define RTPT_PT_CENTER_OUT 1
#endif
// GB: the launch also holds the workgroups of K0 (+ K1) (k_gbuffer_pathtrace below), which put the G-buffer depth into the
// traced image's alpha themselves: a path that ends stores the 12 bytes of its colour only.  The launch's grid is then
// taller than the tile grid (a.tiles_y rows of tiles).
template <int BVH, bool COMPACT, bool GB>
__device__ __forceinline__ void pathtrace_tile(const PathtraceArgs& a) {
  // dynamic LDS, two tenants that are never live together: the BVH node stack (stack_depth x 256 entries, only
  // inside closest_hit) and the compaction exchange buffer (only between the barriers of the compaction step).
  // Sharing it takes the BVH kernel from 41 to 30 KB per block: 5 instead of 3 resident blocks per CU for a
  // kernel that PMC shows waiting on memory for 78 % of its wave-cycles.
  extern __shared__ __attribute__((aligned(16))) uint32_t stack[];
  PathState& st = *reinterpret_cast<PathState*>(stack);
  // per-pixel sample accumulators and RNG state between samples: behind the shared region, allocated (by
  // launch_pathtrace) only when spp > 1 — the reference runs 1 spp (raytrace.comp.glsl:306)
  float* const sum_r = reinterpret_cast<float*>(stack + a.multi_off);
  float* const sum_g = sum_r + kPtThreads;
  float* const sum_b = sum_g + kPtThreads;
  uint32_t* const rng_pix = reinterpret_cast<uint32_t*>(sum_b + kPtThreads);
  __shared__ uint32_t wave_cnt[kPtRows];
  __shared__ uint32_t q_base;
  __shared__ unsigned int block_rays;
  const int tid = threadIdx.y * kBlockX + threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.y));
  const uint32_t lane = threadIdx.x;
  if (tid == 0) block_rays = 0;
#if RTPT_PT_CENTER_OUT
  // workgroups are dispatched in the order of their linear index; tiles are taken column by column from the middle of
  // the frame outwards, so the last ones dispatched — the tail of the launch — are the outermost columns, where (camera
  // facing the scene) the paths are short.  A frame of a few thousand tiles is only ~2 generations of workgroups.
  const uint32_t tiles_y_ = GB ? a.tiles_y : gridDim.y;
  const uint32_t lin_ = blockIdx.y * gridDim.x + blockIdx.x, col_ = lin_ / tiles_y_;
  uint32_t bx_ = (col_ & 1u) ? (gridDim.x - 1u) / 2u + (col_ + 1u) / 2u : (gridDim.x - 1u) / 2u - col_ / 2u, by_ = lin_ % tiles_y_;
#if RTPT_TILE_TIMELINE
  if (g_tile_order_n == gridDim.x * gridDim.y) {
    const uint32_t t_ = g_tile_order[lin_];
    bx_ = t_ % gridDim.x;
    by_ = t_ / gridDim.x;
  }
#endif
#else
  const uint32_t bx_ = blockIdx.x, by_ = blockIdx.y;
#endif
#if RTPT_TILE_TIMELINE
  TimelineScope tl_(1, by_ * gridDim.x + bx_);  // indexed by tile
#endif
  const int tile_x0 = static_cast<int>(bx_) * kBlockX, tile_y0 = a.g.y0 + static_cast<int>(by_) * kPtRows;
  const float fw = static_cast<float>(a.g.W), fh = static_cast<float>(a.g.H);
  const f3 light_c = ld3(a.light_c);
  unsigned int rays = 0;
  unsigned long long cand = 0;  // candidate triangles of this wave's (jittered) primary rays
  if (!BVH && a.cull) cand = span_candidates(a.bounds, a.scene.n_tris, tile_x0 + wave_x0(wave), tile_y0 + wave_y0(wave));

  // per-thread path registers
  const uint32_t pix0 = tile_pixel(wave, lane);  // the pixel this thread starts on
  uint32_t pix = pix0, rng = 0;
  f3 o{0.f, 0.f, 0.f}, d{0.f, 0.f, -1.f}, acc{1.f, 1.f, 1.f};
  {
    const int x = tile_x0 + static_cast<int>(pix0 & 63u), y = tile_y0 + static_cast<int>(pix0 >> 6);
    rng = exact::rng_seed(static_cast<uint32_t>(x), static_cast<uint32_t>(y), a.frame, a.batch);  // :297
    if (a.spp > 1) sum_r[tid] = sum_g[tid] = sum_b[tid] = 0.0f;
  }
  for (uint32_t smp = 0; smp < a.spp; smp++) {
    // ---- primary rays: thread <-> its starting pixel pix0 (mapping restored at the start of every sample)
    if (smp > 0) {
      __syncthreads();
      pix = pix0;
      rng = rng_pix[pix0];  // the pixel's RNG state after its previous sample (stored at path end)
    }
    const int px0 = tile_x0 + static_cast<int>(pix0 & 63u), py0 = tile_y0 + static_cast<int>(pix0 >> 6);
    bool alive = (px0 < a.g.W) && (py0 < a.g.y1);
    if (alive) {
      float u1 = glsl_max(1e-38f, exact::rng_next(rng));  // :87 Box-Muller
      float u2 = exact::rng_next(rng);
      float rad = exact::sqrt_(-2.0f * exact::log_(u1));
      float sn, cs;
      exact::sincos2pi(u2, sn, cs);
      float cx = fmaf_(a.jitter, rad * cs, static_cast<float>(px0) + 0.5f);  // :314
      float cy = fmaf_(a.jitter, rad * sn, static_cast<float>(py0) + 0.5f);
      float ux = RTPT_DIV_SH(fmaf_(2.0f, cx, -fw), fh);     // :315
      float uy = -RTPT_DIV_SH(fmaf_(2.0f, cy, -fh), fh);  // :316
      d = exact::normalize(f3{a.slope * ux, a.slope * uy, -1.0f});  // :319-320
      o = ld3(a.cam);
      acc = f3{1.f, 1.f, 1.f};  // :201
    }
    for (uint32_t seg = 0; seg < a.seg_end; seg++) {  // :204 (block-uniform); seg_end < max_segments: the rest is handed over
      if (alive) {
        HitRec h{a.tmax, 0u, 0.f, 0.f, 1.f};
        if (!BVH && a.cull && seg == 0)
          closest_hit_brute_set(a.scene, cand, o, d, h);
        else
          closest_hit<BVH>(a.scene, o, d, h, stack, tid, kPtThreads, 1 + static_cast<int>(seg));  // :208-222
        const int x = tile_x0 + static_cast<int>(pix & 63u), y = tile_y0 + static_cast<int>(pix >> 6);
        if (y >= a.count_y0 && y < a.count_y1) rays++;
        const size_t gi = static_cast<size_t>(y - a.g.row_base) * a.g.W + x;
        if (seg == 0 && smp == 0 && a.hit_id) a.hit_id[gi] = h.id1;
        const bool done = shade_segment(a, h, seg, light_c, o, d, acc, rng);
        if (done) {
          alive = false;
          if (a.spp == 1) {
            // :328,:343 — alpha carries the G-buffer depth for the filter (rgbd)
            if (GB)
              store_rgb(a.image + gi, acc);  // alpha = the G-buffer depth, stored by the G-buffer workgroups of this launch
            else
              a.image[gi] = make_float4(acc.x, acc.y, acc.z, a.depth[gi]);
          } else {
            sum_r[pix] += acc.x;  // :325 (one path per pixel at a time: no race)
            sum_g[pix] += acc.y;
            sum_b[pix] += acc.z;
            rng_pix[pix] = rng;  // :307 the next sample of this pixel continues the same stream
          }
        }
      }
      if (seg + 1 >= a.seg_end) break;
      if (!COMPACT) {  // short paths: keep the lane <-> pixel mapping, a wave leaves when all its paths ended
        if (__ballot(alive) == 0ull) break;
        continue;
      }
      // ---- compact the survivors to the front of the block
      const unsigned long long live = __ballot(alive);
      if (lane == 0) wave_cnt[wave] = static_cast<uint32_t>(__builtin_popcountll(live));
      __syncthreads();
      uint32_t base = 0, total = 0;
#pragma unroll
      for (int w = 0; w < kPtRows; w++) {
        const uint32_t c = wave_cnt[w];
        if (w < wave) base += c;
        total += c;
      }
      if (total == 0) break;  // block-uniform
      if (alive) {
        const uint32_t slot = base + __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(live >> 32),
                                                              __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(live), 0u));
        st.pix[slot] = pix;
        st.rng[slot] = rng;
        st.ox[slot] = o.x; st.oy[slot] = o.y; st.oz[slot] = o.z;
        st.dx[slot] = d.x; st.dy[slot] = d.y; st.dz[slot] = d.z;
        st.ar[slot] = acc.x; st.ag[slot] = acc.y; st.ab[slot] = acc.z;
      }
      __syncthreads();
      alive = static_cast<uint32_t>(tid) < total;
      if (alive) {
        pix = st.pix[tid];
        rng = st.rng[tid];
        o = f3{st.ox[tid], st.oy[tid], st.oz[tid]};
        d = f3{st.dx[tid], st.dy[tid], st.dz[tid]};
        acc = f3{st.ar[tid], st.ag[tid], st.ab[tid]};
      }
      // BVH: the traversal's node stack aliases the exchange buffer, so everybody must have read its slot before any
      // wave walks on.  Brute force: nothing writes the buffer before the NEXT compaction's first barrier, which every
      // thread reaches only after these reads — a third barrier per segment would only hold the fast waves back.
      if (BVH) __syncthreads();
    }
    if (a.seg_end < a.max_segments) {  // only with spp == 1: the unfinished paths continue in a queue kernel
      const uint32_t pixg = (static_cast<uint32_t>(tile_y0 + static_cast<int>(pix >> 6)) << 16) |
                            static_cast<uint32_t>(tile_x0 + static_cast<int>(pix & 63u));
      const uint32_t blk = blockIdx.y * gridDim.x + blockIdx.x;
      enqueue_paths(a, blk % kPathQueues, wave_cnt, &q_base, alive, wave, lane, pixg, rng, o, d, acc);
    }
  }
  __syncthreads();
  if (a.spp > 1) {
    const int x = tile_x0 + static_cast<int>(lane), y = tile_y0 + wave;
    if (x < a.g.W && y < a.g.y1) {
      const float ns = static_cast<float>(a.spp);
      const size_t gi = static_cast<size_t>(y - a.g.row_base) * a.g.W + x;
      if (GB)
        store_rgb(a.image + gi, f3{sum_r[tid] / ns, sum_g[tid] / ns, sum_b[tid] / ns});
      else
        a.image[gi] = make_float4(sum_r[tid] / ns, sum_g[tid] / ns, sum_b[tid] / ns, a.depth[gi]);  // :328,:343
    }
  }
  // "ray" = one closest-hit query; one 64-bit atomic per block
  for (int off = 32; off > 0; off >>= 1) rays += __shfl_down(rays, off, 64);
  if ((tid & 63) == 0 && rays) atomicAdd(&block_rays, rays);
  __syncthreads();
  // spread over kRayCounters slots: 32 400 same-address atomics per 4K launch serialise in the memory system and kept
  // the workgroups' slots occupied until acknowledged (a 1-segment launch took 397 us, of which the arithmetic is ~110)
  if (tid == 0 && block_rays)
    atomicAdd(a.raycount + ((blockIdx.y * gridDim.x + blockIdx.x) & (kRayCounters - 1u)), static_cast<unsigned long long>(block_rays));
}

// the two kernels of the tile body: the wave-uniform brute-force one is pinned at 8 waves per SIMD (above), the BVH one
// is LDS-limited anyway and loses 4 % to the scalar spills the pin costs (1.15M-triangle trace 3.69 -> 3.85 ms)
// The BVH variant waits on memory for most of its wave-cycles and is that sensitive to occupancy (kernels.hpp
// SceneView::stack_lds); with the stack's LDS share cut to 16 entries the registers set the limit.  1.15M-triangle trace,
// 5 waves per SIMD (the compiler's 79 VGPRs and the old 30 KB stack) 3.69 ms; pinned at 6 / 7 / 8: 3.93 / 3.55 / 3.36 ms
// (at 8: 64 VGPRs and 8 dwords of scratch per lane).
#ifndef RTPT_PT_BVH_WAVES
#define RTPT_PT_BVH_WAVES 8
#endif
template <int BVH, bool COMPACT>
__global__ __launch_bounds__(kPtThreads)
#if RTPT_PT_BVH_WAVES
__attribute__((amdgpu_waves_per_eu(RTPT_PT_BVH_WAVES, RTPT_PT_BVH_WAVES)))
#endif
void k_pathtrace(PathtraceArgs a) {
  pathtrace_tile<BVH, COMPACT, false>(a);
}
template <bool COMPACT>
__global__ __launch_bounds__(kPtThreads)
#if RTPT_PT_WAVES
__attribute__((amdgpu_waves_per_eu(RTPT_PT_WAVES, RTPT_PT_WAVES)))
#endif
void k_pathtrace_small(PathtraceArgs a) {
  pathtrace_tile<false, COMPACT, false>(a);
}

// K0 + K1 + K2 in one launch (rtpt_gbuffer / rtpt_temporal_gradient recorded right before rtpt_raytrace: main.cpp:1105-1107
// is that order; compacting variants only, the default policy).  Workgroups are dispatched in the order of their linear
// index: rows [0, a.tiles_y) of the grid are the tracing tiles — long ones first, pathtrace_tile — and the rows behind them
// the G-buffer's 64 x 4 tiles, a few microseconds each, which therefore start while the last tracing tiles drain: the
// G-buffer's work fills the tail of the trace instead of having a launch, a ramp and a tail of its own.  Nothing in the
// trace reads what the G-buffer writes except the depth in the traced image's alpha, and that the G-buffer workgroups
// store themselves (gbuffer_pixel) while the tracing ones store the colour's 12 bytes — disjoint bytes, any order.
template <int BVH>
__global__ __launch_bounds__(kPtThreads)
#if RTPT_PT_BVH_WAVES
__attribute__((amdgpu_waves_per_eu(RTPT_PT_BVH_WAVES, RTPT_PT_BVH_WAVES)))
#endif
void k_gbuffer_pathtrace(PathtraceArgs a, GbufferArgs g) {
  if (blockIdx.y < a.tiles_y) {
    pathtrace_tile<BVH, true, true>(a);
  } else {
    extern __shared__ __attribute__((aligned(16))) uint32_t stack[];
#if RTPT_TILE_TIMELINE
    TimelineScope tl_(0, (blockIdx.y - a.tiles_y) * gridDim.x + blockIdx.x);
#endif
    gbuffer_tile<BVH>(g, blockIdx.x, blockIdx.y - a.tiles_y, stack, a.image, a.g.y0, a.g.y1);
  }
}
__global__ __launch_bounds__(kPtThreads)
#if RTPT_PT_WAVES
__attribute__((amdgpu_waves_per_eu(RTPT_PT_WAVES, RTPT_PT_WAVES)))
#endif
void k_gbuffer_pathtrace_small(PathtraceArgs a, GbufferArgs g) {
  if (blockIdx.y < a.tiles_y) {
    pathtrace_tile<0, true, true>(a);
  } else {
    extern __shared__ __attribute__((aligned(16))) uint32_t stack[];
#if RTPT_TILE_TIMELINE
    TimelineScope tl_(0, (blockIdx.y - a.tiles_y) * gridDim.x + blockIdx.x);
#endif
    gbuffer_tile<0>(g, blockIdx.x, blockIdx.y - a.tiles_y, stack, a.image, a.g.y0, a.g.y1);
  }
}

template <int BVH>
__global__ __launch_bounds__(kPtThreads) void k_pathtrace_queue(PathtraceArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint32_t stack[];
  PathState& st = *reinterpret_cast<PathState*>(stack);
  __shared__ uint32_t wave_cnt[kPtRows];
  __shared__ uint32_t q_base;
  __shared__ unsigned int block_rays;
  const int tid = threadIdx.y * kBlockX + threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.y));
  const uint32_t lane = threadIdx.x;
  if (tid == 0) block_rays = 0;
  const f3 light_c = ld3(a.light_c);
  // workgroup b serves region b mod kPathQueues, every (gridDim.x / kPathQueues)-th chunk of 256 records
  const uint32_t region = blockIdx.x % kPathQueues;
  const uint32_t n_in = a.q_in_count[region];
  const size_t region_base = static_cast<size_t>(region) * a.queue_region;
  unsigned int rays = 0;
  for (uint32_t base = (blockIdx.x / kPathQueues) * kPtThreads; base < n_in; base += (gridDim.x / kPathQueues) * kPtThreads) {  // block-uniform
    uint32_t pix = 0, rng = 0;
    f3 o{0.f, 0.f, 0.f}, d{0.f, 0.f, -1.f}, acc{1.f, 1.f, 1.f};
    bool alive = base + static_cast<uint32_t>(tid) < n_in;
    if (alive) {
      const float4* q = reinterpret_cast<const float4*>(a.q_in) + 3 * (region_base + base + tid);
      const float4 q0 = q[0], q1 = q[1], q2 = q[2];
      pix = __float_as_uint(q0.x);
      rng = __float_as_uint(q0.y);
      o = f3{q0.z, q0.w, q1.x};
      d = f3{q1.y, q1.z, q1.w};
      acc = f3{q2.x, q2.y, q2.z};
    }
    for (uint32_t seg = a.seg_begin; seg < a.seg_end; seg++) {
      if (alive) {
        HitRec h{a.tmax, 0u, 0.f, 0.f, 1.f};
        closest_hit<BVH>(a.scene, o, d, h, stack, tid, kPtThreads, 1 + static_cast<int>(seg));  // :208-222
        const int x = static_cast<int>(pix & 0xFFFFu), y = static_cast<int>(pix >> 16);
        if (y >= a.count_y0 && y < a.count_y1) rays++;
        if (shade_segment(a, h, seg, light_c, o, d, acc, rng)) {
          alive = false;
          const size_t gi = static_cast<size_t>(y - a.g.row_base) * a.g.W + x;
          a.image[gi] = make_float4(acc.x, acc.y, acc.z, a.depth[gi]);  // :328,:343 (+ depth in alpha)
        }
      }
      if (seg + 1 >= a.seg_end) break;
      // compact the survivors to the front of the block (as in k_pathtrace)
      const unsigned long long live = __ballot(alive);
      if (lane == 0) wave_cnt[wave] = static_cast<uint32_t>(__builtin_popcountll(live));
      __syncthreads();
      uint32_t cbase = 0, total = 0;
#pragma unroll
      for (int w = 0; w < kPtRows; w++) {
        const uint32_t c = wave_cnt[w];
        if (w < wave) cbase += c;
        total += c;
      }
      if (total == 0) break;  // block-uniform
      if (alive) {
        const uint32_t slot = cbase + __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(live >> 32),
                                                               __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(live), 0u));
        st.pix[slot] = pix;
        st.rng[slot] = rng;
        st.ox[slot] = o.x; st.oy[slot] = o.y; st.oz[slot] = o.z;
        st.dx[slot] = d.x; st.dy[slot] = d.y; st.dz[slot] = d.z;
        st.ar[slot] = acc.x; st.ag[slot] = acc.y; st.ab[slot] = acc.z;
      }
      __syncthreads();
      alive = static_cast<uint32_t>(tid) < total;
      if (alive) {
        pix = st.pix[tid];
        rng = st.rng[tid];
        o = f3{st.ox[tid], st.oy[tid], st.oz[tid]};
        d = f3{st.dx[tid], st.dy[tid], st.dz[tid]};
        acc = f3{st.ar[tid], st.ag[tid], st.ab[tid]};
      }
      if (BVH) __syncthreads();  // the node stack aliases the exchange buffer (see k_pathtrace)
    }
    if (a.seg_end < a.max_segments) enqueue_paths(a, region, wave_cnt, &q_base, alive, wave, lane, pix, rng, o, d, acc);
    __syncthreads();  // wave_cnt / st are reused by the next chunk
  }
  for (int off = 32; off > 0; off >>= 1) rays += __shfl_down(rays, off, 64);
  if ((tid & 63) == 0 && rays) atomicAdd(&block_rays, rays);
  __syncthreads();
  if (tid == 0 && block_rays) atomicAdd(a.raycount + (blockIdx.x & (kRayCounters - 1u)), static_cast<unsigned long long>(block_rays));
}

#ifndef RTPT_AB_VARIANTS
#define RTPT_AB_VARIANTS 0
#endif
#if RTPT_AB_VARIANTS
#include "experiments/pathtrace_pool.inc"
#endif

// ------------------------------------------------------------------------------------------
__global__ void k_selftest_math(int op, const float* in, float* out, size_t n) {
  size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float x = in[i], r = 0.f, s, c;
  switch (op) {
    case 0: r = exact::log_(x); break;
    case 1: exact::sincos2pi(x, s, c); r = s; break;
    case 2: exact::sincos2pi(x, s, c); r = c; break;
    case 3: r = exact::sqrt_(x); break;
    case 4: r = exact::rcp_(x); break;
    case 5: r = fast::exp_(x); break;
    case 6: { uint32_t st = f2u(x); r = exact::rng_next(st); break; }
    case 7: r = exact::exp_(x); break;
    default: break;
  }
  out[i] = r;
}

// every binary32 pattern through exact::sqrt_ (op 3) / exact::rcp_ (op 4) against hipcc's IEEE expansion of the same
// operation: out[0] = patterns whose results differ in any bit, out[1..4] = the first few of them
__global__ void k_selftest_exhaustive(int op, unsigned long long* out) {
  const uint64_t base = (static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x) * 64u;
  for (uint32_t k = 0; k < 64u; k++) {
    const uint32_t b = static_cast<uint32_t>(base + k);
    const float x = u2f(b);
    const float got = op == 3 ? exact::sqrt_(x) : exact::rcp_(x);
    const float want = op == 3 ? __builtin_sqrtf(x) : 1.0f / x;
    if (f2u(got) != f2u(want)) {
      const unsigned long long n = atomicAdd(out, 1ull);
      if (n < 4) out[1 + n] = b;
    }
  }
}

// exact::div_ against hipcc's division.  mode 0: slice `pass` of 256 of the enumeration of all 2^23 x 2^23 significand pairs
// (thread = one b, 2^15 values of a); mode 1: 2^33 operand pairs of arbitrary bits (every exponent, sign, zero, infinity,
// NaN and denormal class gets hit: the long path and the range test) from a counter-based generator
__global__ void k_selftest_div(int mode, uint32_t pass, unsigned long long* out) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;  // 0 .. 2^23-1
  unsigned bad = 0;
  uint32_t fa = 0, fb = 0;
  if (mode == 0) {
    const float b = u2f(0x3f800000u | t);
    for (uint32_t i = 0; i < (1u << 15); i++) {
      const uint32_t ma = (pass << 15) + i;
      const float a = u2f(0x3f800000u | ma);
      if (f2u(exact::div_(a, b)) != f2u(a / b)) { bad++; fa = f2u(a); fb = f2u(b); }
    }
  } else {
    uint32_t st = exact::rng_seed(t, pass, 0x9e3779b9u, 1u);
    for (uint32_t i = 0; i < (1u << 10); i++) {
      st = st * 747796405u + 2891336453u;
      uint32_t ua = ((st >> ((st >> 28) + 4u)) ^ st) * 277803737u;
      ua ^= ua >> 22;
      st = st * 747796405u + 2891336453u;
      uint32_t ub = ((st >> ((st >> 28) + 4u)) ^ st) * 277803737u;
      ub ^= ub >> 22;
      if (i & 1u) ub = (ub & 0x807fffffu) | (ua & 0x7f800000u);  // every other pair: same exponent (quotients near 1)
      const float a = u2f(ua), b = u2f(ub);
      const uint32_t got = f2u(exact::div_(a, b)), want = f2u(a / b);
      if (got != want && !((got & 0x7fffffffu) > 0x7f800000u && (want & 0x7fffffffu) > 0x7f800000u)) { bad++; fa = ua; fb = ub; }
    }
  }
  if (bad && atomicAdd(out, static_cast<unsigned long long>(bad)) == 0) {
    out[1] = fa;
    out[2] = fb;
  }
}

template <int BVH>
__global__ __launch_bounds__(kThreads) void k_selftest_trace(SceneView sc, const float* rays, size_t n, float tmax,
                                                             uint32_t* out_id, float* out_t) {
  extern __shared__ uint32_t stack[];  // BVH: stack_depth x 256 entries (dynamic); unused otherwise
  size_t i = static_cast<size_t>(blockIdx.x) * kThreads + threadIdx.x;
  if (i >= n) return;
  f3 o = ld3(rays + 6 * i), d = ld3(rays + 6 * i + 3);
  HitRec h{tmax, 0u, 0.f, 0.f, 1.f};
  closest_hit<BVH>(sc, o, d, h, stack, threadIdx.x);
  out_id[i] = h.id1;
  if (out_t) out_t[i] = h.id1 ? h.t : 0.0f;
}

}  // namespace

#if RTPT_BVH_COUNT || RTPT_TILE_TIMELINE
#define RTPT_INSTR_HOST
#include "experiments/trace_instrumentation.inc"
#undef RTPT_INSTR_HOST
#endif

void launch_scene_prepare(const ScenePrepArgs& a, hipStream_t s) {
  if (!a.n_tris) return;
  hipLaunchKernelGGL(k_scene_prepare, dim3((a.n_tris + 255) / 256), dim3(256), 0, s, a);
}
void launch_lut(const LutArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(k_lut, dim3((a.n_tris + 256) / 256), dim3(256), 0, s, a);
  if (a.pair_tab) {
    const uint32_t n = (a.n_tris + 1) * (a.n_tris + 1);
    hipLaunchKernelGGL(k_pair_weights, dim3((n + 255) / 256), dim3(256), 0, s, a);
  }
}
void launch_ray_tables(int W, int H, float p00, float p11, float* dvx, float* dvy, hipStream_t s) {
  const int n = W > H ? W : H;
  hipLaunchKernelGGL(k_ray_tables, dim3((n + 255) / 256), dim3(256), 0, s, W, H, p00, p11, dvx, dvy);
}
void launch_gbuffer(const GbufferArgs& a, hipStream_t s) {
  if (a.g.y1 <= a.g.y0) return;
  if (a.scene.use_bvh && a.scene.leaf_pairs)
    hipLaunchKernelGGL(k_gbuffer<2>, grid_for(a.g), dim3(kBlockX, kBlockY), a.scene.stack_lds * kThreads * 4, s, a);
  else if (a.scene.use_bvh)
    hipLaunchKernelGGL(k_gbuffer<1>, grid_for(a.g), dim3(kBlockX, kBlockY), a.scene.stack_lds * kThreads * 4, s, a);
  else
    hipLaunchKernelGGL(k_gbuffer<0>, grid_for(a.g), dim3(kBlockX, kBlockY), 0, s, a);
}
void launch_gradient(const GradientArgs& a, hipStream_t s) {
  if (a.g.y1 <= a.g.y0) return;
  hipLaunchKernelGGL(k_gradient, grid_for(a.g), dim3(kBlockX, kBlockY), 0, s, a);
}
bool pathtrace_fuses_gbuffer(const PathtraceArgs& a, const GbufferArgs& g) {
  // the G-buffer's rows must contain the traced ones (the depth in the traced image's alpha comes from them) and the two
  // workgroup shapes must agree (they share the launch)
  return a.compact && a.spp >= 1 && g.g.y0 <= a.g.y0 && g.g.y1 >= a.g.y1 && a.g.y1 > a.g.y0 && kPtRows == kBlockY;
}
// the path-pool form of the tile kernel (experiments/pathtrace_pool.inc, -DRTPT_AB_VARIANTS=1 builds only) serves scenes whose
// BVH is built over fan pairs, one sample per pixel
#if RTPT_AB_VARIANTS
bool pathtrace_uses_pool(const PathtraceArgs& a) {
  return a.pool_slab && a.scene.use_bvh && a.scene.leaf_pairs && a.compact && a.spp == 1;
}
size_t pathtrace_pool_bytes(int W, int rows) {  // slabs of the tracing workgroups of a launch over `rows` rows
  return static_cast<size_t>((W + kBlockX - 1) / kBlockX) * ((rows + kPoolRows - 1) / kPoolRows) * (7u * kPoolPaths * sizeof(float4));
}
#else
constexpr int kPoolRows = kPtRows;
bool pathtrace_uses_pool(const PathtraceArgs&) { return false; }
size_t pathtrace_pool_bytes(int, int) { return 0; }
#endif
uint32_t pathtrace_grid_blocks(const PathtraceArgs& a, const GbufferArgs* gb) {
  const uint32_t gx = (a.g.W + kBlockX - 1) / kBlockX;
  const int tr = pathtrace_uses_pool(a) ? kPoolRows : kPtRows;
  return gx * ((a.g.y1 - a.g.y0 + tr - 1) / tr + (gb ? (gb->g.y1 - gb->g.y0 + kBlockY - 1) / kBlockY : 0));
}
void launch_pathtrace(const PathtraceArgs& a, const GbufferArgs* gb, hipStream_t s) {
  if (a.g.y1 <= a.g.y0) return;
  dim3 block(kBlockX, kPtRows);
  const bool pool = pathtrace_uses_pool(a);
  const int tile_rows = pool ? kPoolRows : kPtRows;
  const uint32_t tiles_y = (a.g.y1 - a.g.y0 + tile_rows - 1) / tile_rows;
  const dim3 grid((a.g.W + kBlockX - 1) / kBlockX, tiles_y + (gb ? (gb->g.y1 - gb->g.y0 + kBlockY - 1) / kBlockY : 0), 1);
  const size_t stack_bytes = a.scene.use_bvh ? static_cast<size_t>(a.scene.stack_lds) * kPtThreads * 4 : 0;
  size_t dyn = stack_bytes > sizeof(PathState) ? stack_bytes : sizeof(PathState);  // shared by both tenants
  PathtraceArgs b = a;
  b.tiles_y = tiles_y;
  b.multi_off = static_cast<uint32_t>(dyn / 4);
  const size_t dyn_queue = dyn;
  if (a.spp > 1) dyn += 4 * kPtThreads * 4;  // sum_r, sum_g, sum_b, rng_pix
  const uint32_t phase0 = a.first_window ? a.first_window : pt_first_window(a.scene.use_bvh != 0);
  const bool split = a.spp == 1 && a.compact && a.queue[0] && a.queue_count && a.max_segments > phase0 &&
                     (a.max_segments <= 2 * phase0 || a.queue[1]);
  b.seg_begin = 0;
  b.seg_end = split ? phase0 : a.max_segments;
  b.q_in = nullptr;
  b.q_in_count = nullptr;
  b.q_out = split ? a.queue[0] : nullptr;
  b.q_out_count = split ? a.queue_count : nullptr;
  if (split) (void)hipMemsetAsync(a.queue_count, 0, 2 * kPathQueues * sizeof(uint32_t), s);
#if RTPT_AB_VARIANTS
  if (pool) {
    if (gb)
      hipLaunchKernelGGL((k_pathtrace_pool<true>), grid, block, dyn, s, b, *gb);
    else
      hipLaunchKernelGGL((k_pathtrace_pool<false>), grid, block, dyn, s, b, GbufferArgs{});
  } else
#endif
  if (gb) {
    if (a.scene.use_bvh && a.scene.leaf_pairs)
      hipLaunchKernelGGL((k_gbuffer_pathtrace<2>), grid, block, dyn, s, b, *gb);
    else if (a.scene.use_bvh)
      hipLaunchKernelGGL((k_gbuffer_pathtrace<1>), grid, block, dyn, s, b, *gb);
    else
      hipLaunchKernelGGL(k_gbuffer_pathtrace_small, grid, block, dyn, s, b, *gb);
  } else if (a.compact) {
    if (a.scene.use_bvh && a.scene.leaf_pairs)
      hipLaunchKernelGGL((k_pathtrace<2, true>), grid, block, dyn, s, b);
    else if (a.scene.use_bvh)
      hipLaunchKernelGGL((k_pathtrace<1, true>), grid, block, dyn, s, b);
    else
      hipLaunchKernelGGL((k_pathtrace_small<true>), grid, block, dyn, s, b);
  } else {
    if (a.scene.use_bvh && a.scene.leaf_pairs)
      hipLaunchKernelGGL((k_pathtrace<2, false>), grid, block, dyn, s, b);
    else if (a.scene.use_bvh)
      hipLaunchKernelGGL((k_pathtrace<1, false>), grid, block, dyn, s, b);
    else
      hipLaunchKernelGGL((k_pathtrace_small<false>), grid, block, dyn, s, b);
  }
  if (!split) return;
  const int n_cu = a.n_cu > 0 ? a.n_cu : 256;  // of the context's device (rtpt_create)
  const dim3 qgrid(static_cast<uint32_t>(n_cu) * 8u);
  int cur = 0;
  for (uint32_t begin = phase0, len = phase0; begin < a.max_segments; begin += len, len *= 2, cur ^= 1) {
    const uint32_t end = begin + len < a.max_segments ? begin + len : a.max_segments;
    PathtraceArgs c = b;
    c.seg_begin = begin;
    c.seg_end = end;
    c.q_in = a.queue[cur];
    c.q_in_count = a.queue_count + cur * kPathQueues;
    const bool more = end < a.max_segments;
    c.q_out = more ? a.queue[cur ^ 1] : nullptr;
    c.q_out_count = more ? a.queue_count + (cur ^ 1) * kPathQueues : nullptr;
    if (more) (void)hipMemsetAsync(a.queue_count + (cur ^ 1) * kPathQueues, 0, kPathQueues * sizeof(uint32_t), s);
    if (a.scene.use_bvh && a.scene.leaf_pairs)
      hipLaunchKernelGGL((k_pathtrace_queue<2>), qgrid, block, dyn_queue, s, c);
    else if (a.scene.use_bvh)
      hipLaunchKernelGGL((k_pathtrace_queue<1>), qgrid, block, dyn_queue, s, c);
    else
      hipLaunchKernelGGL((k_pathtrace_queue<0>), qgrid, block, dyn_queue, s, c);
    if (end >= a.max_segments) break;
  }
}
void launch_selftest_math(int op, const float* in, float* out, size_t n, hipStream_t s) {
  if (!n) return;
  hipLaunchKernelGGL(k_selftest_math, dim3((n + 255) / 256), dim3(256), 0, s, op, in, out, n);
}
void launch_selftest_exhaustive(int op, unsigned long long* out, hipStream_t s) {
  hipLaunchKernelGGL(k_selftest_exhaustive, dim3((1u << 26) / 256u), dim3(256), 0, s, op, out);  // 2^26 threads x 64 patterns
}
void launch_selftest_div(int mode, uint32_t pass, unsigned long long* out, hipStream_t s) {
  hipLaunchKernelGGL(k_selftest_div, dim3((1u << 23) / 256u), dim3(256), 0, s, mode, pass, out);
}
void launch_selftest_trace(const SceneView& scene, const float* rays, size_t n, float tmax, uint32_t* out_id,
                           float* out_t, hipStream_t s) {
  if (!n) return;
  dim3 grid((n + kThreads - 1) / kThreads), block(kThreads);
  if (scene.use_bvh && scene.leaf_pairs)
    hipLaunchKernelGGL(k_selftest_trace<2>, grid, block, scene.stack_lds * kThreads * 4, s, scene, rays, n, tmax, out_id, out_t);
  else if (scene.use_bvh)
    hipLaunchKernelGGL(k_selftest_trace<1>, grid, block, scene.stack_lds * kThreads * 4, s, scene, rays, n, tmax, out_id, out_t);
  else
    hipLaunchKernelGGL(k_selftest_trace<0>, grid, block, 0, s, scene, rays, n, tmax, out_id, out_t);
}

}  // namespace rt
