// bvh.hpp — host-side acceleration-structure builder replacing the driver's opaque BLAS/TLAS
// build behind buildAccelerationStructure (main.cpp:687-742).  CDNA4 has no ray-tracing unit and
// HIP has no ray query, so the product owns the structure: a binary BVH over the flattened
// world-space triangle soup, binned-SAH built, stored as 64-byte "child-pair" nodes so one node
// visit is one 64-byte fetch (4 x dwordx4) that yields both children's boxes.
#pragma once

#include <cstdint>
#include <vector>

namespace rt {

constexpr uint32_t kBvhEmpty = 0xFFFFFFFFu;
// Triangles per leaf: 2 (one fan pair, or two single triangles) since the end of round 4 — K0 + K1 + K2 on the 1.15 M-triangle
// frame 2 852 us with 4, 2 825 with 2 (profiles/r04_bvh_bins_ab.txt): a leaf is then ONE 80-byte record and its test has no loop.
#ifndef RTPT_BVH_MAX_LEAF
#define RTPT_BVH_MAX_LEAF 2
#endif
#ifndef RTPT_BVH_MIN_LEAF
#define RTPT_BVH_MIN_LEAF 1
#endif
#ifndef RTPT_BVH_NODE_COST
#define RTPT_BVH_NODE_COST 1.5f
#endif
constexpr int kBvhMinLeaf = RTPT_BVH_MIN_LEAF;  // groups this small are always leaves
constexpr float kNodeCost = RTPT_BVH_NODE_COST;  // cost of a node visit in triangle tests (SAH leaf decision)
constexpr int kBvhMaxLeaf = RTPT_BVH_MAX_LEAF;  // triangles per leaf (the leaf reference encodes count - 1 in 2 bits)
constexpr int kBvhMaxDepth = 48;  // traversal stack capacity (entries per lane)

// 64 bytes.  A child with cnt > 0 is a leaf: idx = offset of its first record in leaf order.
// cnt == 0: interior, idx = node index; idx == kBvhEmpty: absent child (root of a 1-leaf scene).
struct alignas(16) BvhNode {
  float lmin[3];
  uint32_t lidx;
  float lmax[3];
  uint32_t lcnt;
  float rmin[3];
  uint32_t ridx;
  float rmax[3];
  uint32_t rcnt;
};
static_assert(sizeof(BvhNode) == 64, "child-pair node is one 64-byte line");

struct Bvh {
  std::vector<BvhNode> nodes;       // node 0 is the root pair
  std::vector<uint32_t> leaf_order; // leaf_order[i] = triangle id stored at leaf slot i
  int max_depth = 0;
  float scene_min[3], scene_max[3];
};

// Device form of a node: 32 bytes.  Both child boxes on a 16-bit grid spanning the (padded) scene bounds, per
// axis: coordinate = origin + q * cell, min rounded down, max rounded up, so the quantised box contains the
// binary32 one; the two child references pre-encoded (bit 31 set: leaf, (first << 2) | (count - 1); clear: node
// index; kBvhEmpty: absent).  The traversal is bound by vector-memory instructions and L1 line accesses, not by
// arithmetic: half the bytes per node visit is half the loads (2 x dwordx4).  Boxes only cull and order; the
// triangle records stay binary32, so hits are unchanged.  (binary16 boxes were tried first: 11 bits of mantissa
// inflate the leaves of the 1.15M-triangle lattice by ~30 % and the trace ran 3x SLOWER; the grid is 32x finer.)
struct alignas(16) BvhNodeQ {
  // per child and axis one dword (min | max << 16): left x, y, z, right x, y, z.  The traversal rotates a dword by 16 where
  // the ray runs against the axis, so that its low half is always the plane the ray meets first (kernels.hip: node_step)
  uint16_t box[12];
  uint32_t lref, rref;
};
constexpr int bvh_box_lo(int side, int axis) { return side * 6 + 2 * axis; }
constexpr int bvh_box_hi(int side, int axis) { return side * 6 + 2 * axis + 1; }
static_assert(sizeof(BvhNodeQ) == 32, "quantised child-pair node is 32 bytes");

struct BvhGrid {
  float origin[3], cell[3];
};

// packs bvh.nodes into `out`; returns the grid
BvhGrid pack_quantised_nodes(const Bvh& bvh, std::vector<BvhNodeQ>& out);

// tris: n x 9 floats (v0,v1,v2 world space).  Boxes are padded by `pad_rel` x scene diagonal so
// that the conservative slab test can never reject a triangle the shared ray-triangle routine
// would accept (closest hit = min over (t, id) must not depend on the structure, D4).
// pairs: triangles (2q, 2q+1) are built as ONE primitive each (the two halves of a fan-triangulated quad, which the caller
// has checked): they land in the same leaf, next to each other and in this order, leaves hold 2 or 4 triangles, and the
// traversal tests them with the shared-edge pair test.  n must be even.
void build_bvh(const float* tris, uint32_t n, Bvh& out, float pad_rel = 1e-5f, bool pairs = false);

// Same topology and leaf order, new vertex positions (an animated `model` matrix, main.cpp:1469): recomputes every
// child box bottom-up from the moved triangles, the scene bounds and the padding.  The tree stays valid for any motion;
// its quality is that of the pose it was built for.
void refit_bvh(const float* tris, uint32_t n, Bvh& bvh, float pad_rel = 1e-5f);

}  // namespace rt
