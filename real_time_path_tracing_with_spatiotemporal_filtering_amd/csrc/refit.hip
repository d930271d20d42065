// refit.hip — device-side re-pose + BVH refit for an animated `ubo.model` (visibility.vert.glsl:24; the reference
// recomputes the matrix every frame, main.cpp:1469).  Round 2 did this on the host (re-pose, refit, re-upload, stream
// synchronise: tens of milliseconds for the 1.15M-triangle scene against a 4 ms frame); here everything stays on the
// device and on the context's stream, with no host synchronisation:
//   k_pose            posed triangle = model * uploaded triangle, the LUT's fixed-order fma arithmetic (exact::mat_row_point)
//   k_refit_level     child boxes of every node of one HEIGHT class, leaves first (the host sorts the nodes by height
//                     once per scene_upload: topology and leaf order never change in a refit)
//   k_refit_grid      scene bounds = the root's box; padding, 16-bit grid origin and cell (bvh.cpp: pack_quantised_nodes)
//   k_refit_quantise  every node's padded child boxes rounded OUTWARD onto the grid (min down, max up, checked against
//                     the binary32 decode the traversal's arithmetic implies)
// Boxes only cull and order candidates (closest hit = min over (t, id) of one shared triangle routine, D4), so the
// device refit need not — and does not try to — reproduce the host refit's boxes bit for bit; it keeps their guarantees.
#include "bvh.hpp"
#include "device_common.hpp"

namespace rt {
namespace {

struct FBox {  // unpadded binary32 boxes of a node's two children
  float lmn[3], lmx[3], rmn[3], rmx[3];
};

__global__ void k_pose(uint32_t n_verts, const float* __restrict__ src, float* __restrict__ dst, RefitModel m) {
  const uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= n_verts) return;
  const f3 p{src[3 * static_cast<size_t>(v)], src[3 * static_cast<size_t>(v) + 1], src[3 * static_cast<size_t>(v) + 2]};
  float* o = dst + 3 * static_cast<size_t>(v);
  if (m.identity) {
    o[0] = p.x; o[1] = p.y; o[2] = p.z;
  } else {
    o[0] = exact::mat_row_point(m.m, 0, p);
    o[1] = exact::mat_row_point(m.m, 1, p);
    o[2] = exact::mat_row_point(m.m, 2, p);
  }
}

__device__ __forceinline__ void box_reset(float* mn, float* mx) {
  for (int a = 0; a < 3; a++) {
    mn[a] = 3.402823466e+38f;
    mx[a] = -3.402823466e+38f;
  }
}

__global__ void k_refit_level(RefitArgs a, uint32_t first, uint32_t count) {
  const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= count) return;
  const uint32_t ni = a.order[first + k];
  const BvhNodeQ nd = a.nodes[ni];
  FBox* fb = reinterpret_cast<FBox*>(a.fbox);
  FBox out;
  for (int side = 0; side < 2; side++) {
    const uint32_t ref = side ? nd.rref : nd.lref;
    float* mn = side ? out.rmn : out.lmn;
    float* mx = side ? out.rmx : out.lmx;
    box_reset(mn, mx);
    if (ref == kBvhEmpty) continue;
    if (ref & 0x80000000u) {  // leaf: (first << 2) | (count - 1)
      const uint32_t f0 = (ref & 0x7FFFFFFFu) >> 2, cnt = (ref & 3u) + 1u;
      for (uint32_t j = 0; j < cnt; j++) {
        const float* t = a.tris + 9 * static_cast<size_t>(a.leaf_order[f0 + j]);
        for (int v = 0; v < 3; v++)
          for (int ax = 0; ax < 3; ax++) {
            mn[ax] = __builtin_fminf(mn[ax], t[3 * v + ax]);
            mx[ax] = __builtin_fmaxf(mx[ax], t[3 * v + ax]);
          }
      }
    } else {  // interior child: finished by an earlier launch (lower height)
      const FBox c = fb[ref];
      for (int ax = 0; ax < 3; ax++) {
        mn[ax] = __builtin_fminf(c.lmn[ax], c.rmn[ax]);
        mx[ax] = __builtin_fmaxf(c.lmx[ax], c.rmx[ax]);
      }
    }
  }
  fb[ni] = out;
}

// grid[0..2] origin, [3..5] cell, [6] pad  (bvh.cpp: refit_bvh's padding + pack_quantised_nodes' grid)
__global__ void k_refit_grid(RefitArgs a, float pad_rel) {
  if (blockIdx.x || threadIdx.x) return;
  const FBox r = reinterpret_cast<const FBox*>(a.fbox)[0];
  float smn[3], smx[3];
  for (int ax = 0; ax < 3; ax++) {
    smn[ax] = __builtin_fminf(r.lmn[ax], r.rmn[ax]);
    smx[ax] = __builtin_fmaxf(r.lmx[ax], r.rmx[ax]);
  }
  const float dx = smx[0] - smn[0], dy = smx[1] - smn[1], dz = smx[2] - smn[2];
  float diag = __builtin_sqrtf(dx * dx + dy * dy + dz * dz), mag = 0.f;
  for (int ax = 0; ax < 3; ax++) mag = __builtin_fmaxf(mag, __builtin_fmaxf(__builtin_fabsf(smn[ax]), __builtin_fabsf(smx[ax])));
  diag = __builtin_fmaxf(diag, mag);
  const float pad = pad_rel * diag;
  for (int ax = 0; ax < 3; ax++) {
    const double lo = static_cast<double>(smn[ax] - pad), hi = static_cast<double>(smx[ax] + pad);
    const double ext = (hi - lo) > 1e-20 ? (hi - lo) : 1e-20;
    const float cell = static_cast<float>(ext / 65533.0);  // one grid step of slack at either end
    a.grid[3 + ax] = cell;
    a.grid[ax] = static_cast<float>(lo - static_cast<double>(cell));
  }
  a.grid[6] = pad;
  a.grid[7] = 0.f;
}

__global__ void k_refit_quantise(RefitArgs a, uint32_t n_nodes) {
  const uint32_t ni = blockIdx.x * blockDim.x + threadIdx.x;
  if (ni >= n_nodes) return;
  const FBox b = reinterpret_cast<const FBox*>(a.fbox)[ni];
  BvhNodeQ nd = a.nodes[ni];
  const float pad = a.grid[6];
  for (int side = 0; side < 2; side++) {
    const bool empty = (side ? nd.rref : nd.lref) == kBvhEmpty;
    const float* mn = side ? b.rmn : b.lmn;
    const float* mx = side ? b.rmx : b.lmx;
    for (int ax = 0; ax < 3; ax++) {
      uint32_t qlo = 0, qhi = 0;
      if (!empty) {
        const float org = a.grid[ax], cell = a.grid[3 + ax];
        const float xlo = mn[ax] - pad, xhi = mx[ax] + pad;
        // the decoded face is formed as ONE binary32 fma, fma(q, cell, org) — written out so that the device, the host's
        // rtpt_debug_bvh_check and this loop agree whatever -ffp-contract says: step until THAT value is on the outer side.
        // The traversal never forms this value: it computes q * (cell / d) + (org - o) / d, which differs from
        // (fma(q, cell, org) - o) / d by a few ulps of the box coordinate; the 1e-5 x diagonal padding (pad, above) is what
        // absorbs that difference (round-3 advice: the check alone does not).
        double q = __builtin_floor((static_cast<double>(xlo) - static_cast<double>(org)) / static_cast<double>(cell));
        q = q < 0.0 ? 0.0 : (q > 65535.0 ? 65535.0 : q);
        while (q > 0.0 && __builtin_fmaf(static_cast<float>(q), cell, org) > xlo) q -= 1.0;
        qlo = static_cast<uint32_t>(q);
        q = __builtin_ceil((static_cast<double>(xhi) - static_cast<double>(org)) / static_cast<double>(cell));
        q = q < 0.0 ? 0.0 : (q > 65535.0 ? 65535.0 : q);
        while (q < 65535.0 && __builtin_fmaf(static_cast<float>(q), cell, org) < xhi) q += 1.0;
        qhi = static_cast<uint32_t>(q);
      }
      nd.box[bvh_box_lo(side, ax)] = static_cast<uint16_t>(qlo);
      nd.box[bvh_box_hi(side, ax)] = static_cast<uint16_t>(qhi);
    }
  }
  a.nodes[ni] = nd;
}

}  // namespace

void launch_pose(uint32_t n_verts, const float* src, float* dst, const RefitModel& m, hipStream_t s) {
  if (!n_verts) return;
  hipLaunchKernelGGL(k_pose, dim3((n_verts + 255) / 256), dim3(256), 0, s, n_verts, src, dst, m);
}

void launch_refit(const RefitArgs& a, const uint32_t* level_first, int n_levels, uint32_t n_nodes, float pad_rel, hipStream_t s) {
  for (int h = 0; h < n_levels; h++) {
    const uint32_t first = level_first[h], count = level_first[h + 1] - first;
    if (count) hipLaunchKernelGGL(k_refit_level, dim3((count + 127) / 128), dim3(128), 0, s, a, first, count);
  }
  hipLaunchKernelGGL(k_refit_grid, dim3(1), dim3(64), 0, s, a, pad_rel);
  hipLaunchKernelGGL(k_refit_quantise, dim3((n_nodes + 127) / 128), dim3(128), 0, s, a, n_nodes);
}

}  // namespace rt
