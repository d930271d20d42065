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
constexpr int kBvhMaxLeaf = 4;    // triangles per leaf
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

// tris: n x 9 floats (v0,v1,v2 world space).  Boxes are padded by `pad_rel` x scene diagonal so
// that the conservative slab test can never reject a triangle the shared ray-triangle routine
// would accept (closest hit = min over (t, id) must not depend on the structure, D4).
void build_bvh(const float* tris, uint32_t n, Bvh& out, float pad_rel = 1e-5f);

}  // namespace rt
