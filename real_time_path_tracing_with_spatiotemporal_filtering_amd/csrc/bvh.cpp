// bvh.cpp — binned-SAH BVH2 builder (host).  See bvh.hpp.
#include "bvh.hpp"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstring>

namespace rt {
namespace {

struct Box {
  float mn[3], mx[3];
  void reset() {
    for (int a = 0; a < 3; a++) {
      mn[a] = FLT_MAX;
      mx[a] = -FLT_MAX;
    }
  }
  void grow(const float* p) {
    for (int a = 0; a < 3; a++) {
      mn[a] = std::min(mn[a], p[a]);
      mx[a] = std::max(mx[a], p[a]);
    }
  }
  void grow(const Box& b) {
    for (int a = 0; a < 3; a++) {
      mn[a] = std::min(mn[a], b.mn[a]);
      mx[a] = std::max(mx[a], b.mx[a]);
    }
  }
  float half_area() const {
    float dx = mx[0] - mn[0], dy = mx[1] - mn[1], dz = mx[2] - mn[2];
    if (dx < 0.f) return 0.f;
    return dx * dy + dy * dz + dz * dx;
  }
};

struct Prim {
  Box box;
  float c[3];
  uint32_t id;  // first triangle
  uint32_t w;   // triangles this primitive stands for: 1, or 2 (id, id + 1) when the scene is built over fan pairs
};

struct Child {
  Box box;
  uint32_t idx;
  uint32_t cnt;
};

// SAH bins per axis.  K0 + K1 + K2 on the 1.15 M-triangle frame (profiles/r04_bvh_bins_ab.txt): 16: 2 896 us, 32: 2 856,
// 64: 2 863, 128: 2 871; the build of that scene takes ~0.5 s with any of them.  SAH node cost 0.75 / 1.0 / 1.5: the same tree
// (2 896); 2.0: 3 003; 3.0: 3 033.
#ifndef RTPT_BVH_BINS
#define RTPT_BVH_BINS 32
#endif
constexpr int kBins = RTPT_BVH_BINS;

struct Builder {
  std::vector<Prim> prims;
  Bvh* out;
  float pad;

  Child make_leaf(uint32_t lo, uint32_t hi) {
    Child c;
    c.box.reset();
    c.idx = static_cast<uint32_t>(out->leaf_order.size());
    c.cnt = 0;
    for (uint32_t i = lo; i < hi; i++) {
      c.box.grow(prims[i].box);
      for (uint32_t k = 0; k < prims[i].w; k++) out->leaf_order.push_back(prims[i].id + k);
      c.cnt += prims[i].w;
    }
    return c;
  }

  // returns the descriptor of the subtree over prims[lo,hi)
  Child build(uint32_t lo, uint32_t hi, int depth) {
    out->max_depth = std::max(out->max_depth, depth);
    const uint32_t w = prims[lo].w;  // the same for every primitive of a build
    uint32_t n = (hi - lo) * w;      // triangles
    if (hi - lo == 1 || n <= static_cast<uint32_t>(kBvhMinLeaf)) return make_leaf(lo, hi);
    const bool may_leaf = n <= static_cast<uint32_t>(kBvhMaxLeaf);  // SAH decides below

    Box cb, bb;
    cb.reset();
    bb.reset();
    for (uint32_t i = lo; i < hi; i++) {
      cb.grow(prims[i].c);
      bb.grow(prims[i].box);
    }
    uint32_t mid = lo;
    // SAH over kBins bins on every axis; past a depth budget fall back to the object median so the
    // remaining depth is bounded by log2(n)
    bool median = depth >= kBvhMaxDepth - 26;
    if (median && may_leaf) return make_leaf(lo, hi);
    if (!median) {
      float best_cost = FLT_MAX;
      int best_axis = -1, best_split = -1;
      for (int a = 0; a < 3; a++) {
        float ext = cb.mx[a] - cb.mn[a];
        if (!(ext > 0.f)) continue;
        Box bins[kBins];
        uint32_t cnt[kBins];
        for (int b = 0; b < kBins; b++) {
          bins[b].reset();
          cnt[b] = 0;
        }
        float scale = kBins / ext;
        for (uint32_t i = lo; i < hi; i++) {
          int b = std::min(kBins - 1, static_cast<int>((prims[i].c[a] - cb.mn[a]) * scale));
          bins[b].grow(prims[i].box);
          cnt[b] += w;
        }
        float right_area[kBins];
        uint32_t right_cnt[kBins];
        Box acc;
        acc.reset();
        uint32_t c = 0;
        for (int b = kBins - 1; b > 0; b--) {
          acc.grow(bins[b]);
          c += cnt[b];
          right_area[b] = acc.half_area();
          right_cnt[b] = c;
        }
        acc.reset();
        c = 0;
        for (int b = 0; b < kBins - 1; b++) {
          acc.grow(bins[b]);
          c += cnt[b];
          if (c == 0 || right_cnt[b + 1] == 0) continue;
          float cost = acc.half_area() * c + right_area[b + 1] * right_cnt[b + 1];
          if (cost < best_cost) {
            best_cost = cost;
            best_axis = a;
            best_split = b;
          }
        }
      }
      // a group small enough to be a leaf is split only if the split is cheaper: one node visit (two box tests)
      // costs kNodeCost triangle tests (measured: ~55 vs ~35 VALU)
      if (may_leaf && (best_axis < 0 || kNodeCost * bb.half_area() + best_cost >= static_cast<float>(n) * bb.half_area()))
        return make_leaf(lo, hi);
      if (best_axis >= 0) {
        float ext = cb.mx[best_axis] - cb.mn[best_axis];
        float scale = kBins / ext;
        float mn = cb.mn[best_axis];
        int axis = best_axis, split = best_split;
        auto it = std::partition(prims.begin() + lo, prims.begin() + hi, [&](const Prim& p) {
          int b = std::min(kBins - 1, static_cast<int>((p.c[axis] - mn) * scale));
          return b <= split;
        });
        mid = static_cast<uint32_t>(it - prims.begin());
      }
    }
    if (mid == lo || mid == hi) {
      // object-median split on the widest centroid axis (also the degenerate-centroid fallback)
      int axis = 0;
      float best = -1.f;
      for (int a = 0; a < 3; a++) {
        float e = cb.mx[a] - cb.mn[a];
        if (e > best) {
          best = e;
          axis = a;
        }
      }
      mid = lo + (hi - lo) / 2;
      std::nth_element(prims.begin() + lo, prims.begin() + mid, prims.begin() + hi,
                       [axis](const Prim& x, const Prim& y) {
                         return x.c[axis] < y.c[axis] || (x.c[axis] == y.c[axis] && x.id < y.id);
                       });
    }

    uint32_t node_index = static_cast<uint32_t>(out->nodes.size());
    out->nodes.emplace_back();
    Child l = build(lo, mid, depth + 1);
    Child r = build(mid, hi, depth + 1);
    write_pair(node_index, l, r);
    Child me;
    me.box = l.box;
    me.box.grow(r.box);
    me.idx = node_index;
    me.cnt = 0;
    return me;
  }

  void write_pair(uint32_t node_index, const Child& l, const Child& r) {
    BvhNode& nd = out->nodes[node_index];
    for (int a = 0; a < 3; a++) {
      nd.lmin[a] = l.box.mn[a] - pad;
      nd.lmax[a] = l.box.mx[a] + pad;
      nd.rmin[a] = r.box.mn[a] - pad;
      nd.rmax[a] = r.box.mx[a] + pad;
    }
    nd.lidx = l.idx;
    nd.lcnt = l.cnt;
    nd.ridx = r.idx;
    nd.rcnt = r.cnt;
  }
};

}  // namespace

void build_bvh(const float* tris, uint32_t n, Bvh& out, float pad_rel, bool pairs) {
  out.nodes.clear();
  out.leaf_order.clear();
  out.max_depth = 0;
  Builder b;
  b.out = &out;
  const uint32_t w = (pairs && n >= 2 && n % 2 == 0) ? 2u : 1u;
  b.prims.resize(n / w);
  Box scene;
  scene.reset();
  for (uint32_t i = 0; i < n / w; i++) {
    Prim& p = b.prims[i];
    p.box.reset();
    for (uint32_t k = 0; k < 3 * w; k++) p.box.grow(tris + 9 * static_cast<size_t>(i) * w + 3 * k);
    for (int a = 0; a < 3; a++) p.c[a] = 0.5f * (p.box.mn[a] + p.box.mx[a]);
    p.id = i * w;
    p.w = w;
    scene.grow(p.box);
  }
  for (int a = 0; a < 3; a++) {
    out.scene_min[a] = n ? scene.mn[a] : 0.f;
    out.scene_max[a] = n ? scene.mx[a] : 0.f;
  }
  float diag = 0.f;
  if (n) {
    float dx = scene.mx[0] - scene.mn[0], dy = scene.mx[1] - scene.mn[1], dz = scene.mx[2] - scene.mn[2];
    diag = std::sqrt(dx * dx + dy * dy + dz * dz);
    float mag = 0.f;
    for (int a = 0; a < 3; a++) mag = std::max(mag, std::max(std::fabs(scene.mn[a]), std::fabs(scene.mx[a])));
    diag = std::max(diag, mag);
  }
  b.pad = pad_rel * diag;
  out.nodes.reserve(n / 2 + 2);
  out.leaf_order.reserve(n);

  if (n <= static_cast<uint32_t>(kBvhMaxLeaf)) {
    // single-leaf scene: root pair = {leaf, absent}
    out.nodes.emplace_back();
    Child l = b.make_leaf(0, n / w);
    Child r;
    r.box.reset();
    r.idx = kBvhEmpty;
    r.cnt = 0;
    if (n == 0) {
      l.idx = kBvhEmpty;
      l.cnt = 0;
    }
    b.write_pair(0, l, r);
    return;
  }
  Child root = b.build(0, n / w, 0);
  (void)root;  // n > kBvhMaxLeaf: the root is interior and is node 0 by construction
}


void refit_bvh(const float* tris, uint32_t n, Bvh& bvh, float pad_rel) {
  Box scene;
  scene.reset();
  for (uint32_t i = 0; i < n; i++)
    for (int k = 0; k < 3; k++) scene.grow(tris + 9 * static_cast<size_t>(i) + 3 * k);
  for (int a = 0; a < 3; a++) {
    bvh.scene_min[a] = n ? scene.mn[a] : 0.f;
    bvh.scene_max[a] = n ? scene.mx[a] : 0.f;
  }
  float diag = 0.f;
  if (n) {
    float dx = scene.mx[0] - scene.mn[0], dy = scene.mx[1] - scene.mn[1], dz = scene.mx[2] - scene.mn[2];
    diag = std::sqrt(dx * dx + dy * dy + dz * dz);
    float mag = 0.f;
    for (int a = 0; a < 3; a++) mag = std::max(mag, std::max(std::fabs(scene.mn[a]), std::fabs(scene.mx[a])));
    diag = std::max(diag, mag);
  }
  const float pad = pad_rel * diag;
  // nodes are numbered in pre-order (a parent before its subtrees), so walking the array backwards meets every child
  // before its parent; sub[i] = unpadded bounds of the subtree under node i
  std::vector<Box> sub(bvh.nodes.size());
  for (size_t ii = bvh.nodes.size(); ii-- > 0;) {
    BvhNode& nd = bvh.nodes[ii];
    Box me;
    me.reset();
    for (int side = 0; side < 2; side++) {
      const uint32_t idx = side ? nd.ridx : nd.lidx, cnt = side ? nd.rcnt : nd.lcnt;
      float* bmn = side ? nd.rmin : nd.lmin;
      float* bmx = side ? nd.rmax : nd.lmax;
      if (idx == kBvhEmpty) continue;
      Box cb;
      cb.reset();
      if (cnt) {
        for (uint32_t j = 0; j < cnt; j++) {
          const uint32_t id = bvh.leaf_order[idx + j];
          for (int k = 0; k < 3; k++) cb.grow(tris + 9 * static_cast<size_t>(id) + 3 * k);
        }
      } else {
        cb = sub[idx];
      }
      for (int a = 0; a < 3; a++) {
        bmn[a] = cb.mn[a] - pad;
        bmx[a] = cb.mx[a] + pad;
      }
      me.grow(cb);
    }
    sub[ii] = me;
  }
}

BvhGrid pack_quantised_nodes(const Bvh& bvh, std::vector<BvhNodeQ>& out) {
  double lo[3], hi[3];
  for (int a = 0; a < 3; a++) {
    lo[a] = bvh.scene_min[a];
    hi[a] = bvh.scene_max[a];
  }
  for (const BvhNode& n : bvh.nodes)  // padded boxes exceed the scene bounds slightly
    for (int a = 0; a < 3; a++) {
      if (n.lidx != kBvhEmpty) {
        lo[a] = std::min<double>(lo[a], n.lmin[a]);
        hi[a] = std::max<double>(hi[a], n.lmax[a]);
      }
      if (n.ridx != kBvhEmpty) {
        lo[a] = std::min<double>(lo[a], n.rmin[a]);
        hi[a] = std::max<double>(hi[a], n.rmax[a]);
      }
    }
  BvhGrid g;
  double cell[3];
  for (int a = 0; a < 3; a++) {
    const double ext = std::max(hi[a] - lo[a], 1e-20);
    // one grid step of slack at either end absorbs the binary32 rounding of origin and cell
    g.cell[a] = static_cast<float>(ext / 65533.0);
    g.origin[a] = static_cast<float>(lo[a] - static_cast<double>(g.cell[a]));
    cell[a] = g.cell[a];
  }
  // the decoded face is ONE binary32 fma, fma(q, cell, origin) — the same expression in the device refit (refit.hip) and in
  // the checkers (api_selftest.hip): step until THAT value is on the outer side.  The traversal computes
  // q * (cell / d) + (origin - o) / d instead; what lies between the two is a few ulp of the coordinate, two orders of
  // magnitude below the padding of the boxes, which is what absorbs it.
  auto decode = [&](double q, int a) { return std::fmaf(static_cast<float>(q), g.cell[a], g.origin[a]); };
  auto qdown = [&](float x, int a) -> uint16_t {
    double q = std::floor((static_cast<double>(x) - static_cast<double>(g.origin[a])) / cell[a]);
    q = std::min(65535.0, std::max(0.0, q));
    while (q > 0 && decode(q, a) > x) q -= 1;
    return static_cast<uint16_t>(q);
  };
  auto qup = [&](float x, int a) -> uint16_t {
    double q = std::ceil((static_cast<double>(x) - static_cast<double>(g.origin[a])) / cell[a]);
    q = std::min(65535.0, std::max(0.0, q));
    while (q < 65535 && decode(q, a) < x) q += 1;
    return static_cast<uint16_t>(q);
  };
  auto ref = [](uint32_t idx, uint32_t cnt) -> uint32_t {
    if (idx == kBvhEmpty) return kBvhEmpty;
    return cnt ? (0x80000000u | (idx << 2) | (cnt - 1u)) : idx;
  };
  out.resize(bvh.nodes.size());
  for (size_t i = 0; i < bvh.nodes.size(); i++) {
    const BvhNode& n = bvh.nodes[i];
    BvhNodeQ& h = out[i];
    const bool le = n.lidx == kBvhEmpty, re = n.ridx == kBvhEmpty;
    for (int a = 0; a < 3; a++) {
      h.box[bvh_box_lo(0, a)] = le ? 0 : qdown(n.lmin[a], a);
      h.box[bvh_box_hi(0, a)] = le ? 0 : qup(n.lmax[a], a);
      h.box[bvh_box_lo(1, a)] = re ? 0 : qdown(n.rmin[a], a);
      h.box[bvh_box_hi(1, a)] = re ? 0 : qup(n.rmax[a], a);
    }
    h.lref = ref(n.lidx, n.lcnt);
    h.rref = ref(n.ridx, n.rcnt);
  }
  return g;
}

}  // namespace rt
