// device_common.hpp — helpers shared by the gfx950 kernel translation units.
#pragma once

#include "kernels.hpp"
#include "rtpt_math.hpp"

namespace rt {
namespace {

constexpr int kBlockX = 64;  // one wave = 64 consecutive pixels of a row: 1 KiB float4 stores
constexpr int kBlockY = 4;
constexpr int kThreads = kBlockX * kBlockY;

__device__ __forceinline__ f3 ld3(const float* p) { return f3{p[0], p[1], p[2]}; }
__device__ __forceinline__ f3 xyz(float4 v) { return f3{v.x, v.y, v.z}; }

// temporalGradient.comp.glsl:50-69 / temporalFiltering.comp.glsl:157-176 (area-ratio barycentrics)
__device__ __forceinline__ float tri_area(f3 a, f3 b, f3 c) { return exact::length(exact::cross(b - a, c - a)) * 0.5f; }
__device__ __forceinline__ f3 bary_coords(f3 p, f3 a, f3 b, f3 c) {
  float at = tri_area(a, b, c);
  return f3{exact::div_(tri_area(p, b, c), at), exact::div_(tri_area(a, p, c), at), exact::div_(tri_area(a, b, p), at)};
}
// the same with the triangle's own area supplied (at == tri_area(a, b, c) bit for bit: a per-triangle table)
__device__ __forceinline__ f3 bary_coords_at(f3 p, f3 a, f3 b, f3 c, float at) {
  return f3{exact::div_(tri_area(p, b, c), at), exact::div_(tri_area(a, p, c), at), exact::div_(tri_area(a, b, p), at)};
}
__device__ __forceinline__ f3 bary_mix(f3 bc, f3 a, f3 b, f3 c) {
  return f3{fmaf_(bc.z, c.x, fmaf_(bc.y, b.x, bc.x * a.x)), fmaf_(bc.z, c.y, fmaf_(bc.y, b.y, bc.x * a.y)),
            fmaf_(bc.z, c.z, fmaf_(bc.y, b.z, bc.x * a.z))};
}


// reprojection of a pixel into the previous frame (temporalFiltering.comp.glsl:213-239, worldToPixel :178-189), exact
// arithmetic: the truncated pixel coordinate is an integer observable
__device__ __forceinline__ void reproject_pixel(int W, int H, const float* PVprev, uint32_t idp, f3 wp, const float4* lut_prev,
                                                int x, int y, int& ppx, int& ppy) {
  ppx = x;
  ppy = y;
  if (idp < 1) return;
  const f3 va = xyz(lut_prev[3 * idp]), vb = xyz(lut_prev[3 * idp + 1]), vc = xyz(lut_prev[3 * idp + 2]);
  const f3 bc = bary_coords(wp, va, vb, vc);
  const f3 wpp = bary_mix(bc, va, vb, vc);
  const float clx = exact::mat_row_point(PVprev, 0, wpp), cly = exact::mat_row_point(PVprev, 1, wpp),
              clw = exact::mat_row_point(PVprev, 3, wpp);
  const float ndx = exact::div_(clx, clw), ndy = exact::div_(cly, clw);
  ppx = exact::f2i(fmaf_(ndx, 0.5f, 0.5f) * static_cast<float>(W));
  ppy = exact::f2i(fmaf_(ndy, 0.5f, 0.5f) * static_cast<float>(H));
}

inline dim3 grid_for(const FrameGeom& g) {
  return dim3((g.W + kBlockX - 1) / kBlockX, (g.y1 - g.y0 + kBlockY - 1) / kBlockY, 1);
}


}  // namespace
}  // namespace rt
