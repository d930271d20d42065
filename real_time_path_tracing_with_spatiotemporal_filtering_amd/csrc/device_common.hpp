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
  return f3{tri_area(p, b, c) / at, tri_area(a, p, c) / at, tri_area(a, b, p) / at};
}
__device__ __forceinline__ f3 bary_mix(f3 bc, f3 a, f3 b, f3 c) {
  return f3{fmaf_(bc.z, c.x, fmaf_(bc.y, b.x, bc.x * a.x)), fmaf_(bc.z, c.y, fmaf_(bc.y, b.y, bc.x * a.y)),
            fmaf_(bc.z, c.z, fmaf_(bc.y, b.z, bc.x * a.z))};
}


inline dim3 grid_for(const FrameGeom& g) {
  return dim3((g.W + kBlockX - 1) / kBlockX, (g.y1 - g.y0 + kBlockY - 1) / kBlockY, 1);
}


}  // namespace
}  // namespace rt
