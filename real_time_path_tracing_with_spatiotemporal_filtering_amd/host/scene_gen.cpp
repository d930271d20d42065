// scene_gen.cpp — see scene_gen.hpp.  Built with -ffp-contract=off: every float operation below is one binary32 rounding,
// in the order scenes.py evaluates it with numpy float32 arrays.
#include "scene_gen.hpp"

#include <algorithm>

namespace rtpt_host {

bool tessellate_quads(const std::vector<float>& xyz, const std::vector<uint32_t>& idx, int n, std::vector<float>& out_xyz,
                      std::vector<uint32_t>& out_idx) {
  const size_t n_tris = idx.size() / 3;
  if (n <= 1) {
    out_xyz = xyz;
    out_idx = idx;
    return true;
  }
  if (n_tris % 2) return false;
  for (size_t q = 0; q < n_tris / 2; q++)  // (a, b, c), (a, c, d)
    if (idx[6 * q] != idx[6 * q + 3] || idx[6 * q + 2] != idx[6 * q + 4]) return false;
  // numpy.linspace(0, 1, n + 1, dtype=float32): i * (1 / n) in double, rounded once; the last sample is the stop value
  std::vector<float> t(static_cast<size_t>(n) + 1);
  const double step = 1.0 / static_cast<double>(n);
  for (int i = 0; i < n; i++) t[static_cast<size_t>(i)] = static_cast<float>(static_cast<double>(i) * step);
  t[static_cast<size_t>(n)] = 1.0f;
  std::vector<float> vx;
  std::vector<uint32_t> ti;
  const size_t per_quad = static_cast<size_t>(n + 1) * static_cast<size_t>(n + 1);
  vx.reserve(n_tris / 2 * per_quad * 3);
  ti.reserve(n_tris / 2 * static_cast<size_t>(n) * n * 6);
  for (size_t q = 0; q < n_tris / 2; q++) {
    const float* A = &xyz[3 * static_cast<size_t>(idx[6 * q])];
    const float* B = &xyz[3 * static_cast<size_t>(idx[6 * q + 1])];
    const float* C = &xyz[3 * static_cast<size_t>(idx[6 * q + 2])];
    const float* D = &xyz[3 * static_cast<size_t>(idx[6 * q + 5])];
    const uint32_t base = static_cast<uint32_t>(q * per_quad);
    // bilinear grid P(s, t) = (1-s)(1-t) A + s (1-t) B + s t C + (1-s) t D
    for (int i = 0; i <= n; i++)
      for (int j = 0; j <= n; j++) {
        const float s = t[static_cast<size_t>(i)], u = t[static_cast<size_t>(j)];
        const float w0 = (1.0f - s) * (1.0f - u), w1 = s * (1.0f - u), w2 = s * u, w3 = (1.0f - s) * u;
        for (int a = 0; a < 3; a++) {
          const float p0 = w0 * A[a], p1 = w1 * B[a], p2 = w2 * C[a], p3 = w3 * D[a];
          const float s01 = p0 + p1;
          const float s012 = s01 + p2;
          vx.push_back(s012 + p3);
        }
      }
    for (int i = 0; i < n; i++)
      for (int j = 0; j < n; j++) {
        const uint32_t p00 = base + static_cast<uint32_t>(i * (n + 1) + j), p10 = base + static_cast<uint32_t>((i + 1) * (n + 1) + j);
        const uint32_t p11 = p10 + 1, p01 = p00 + 1;
        const uint32_t cell[6] = {p00, p10, p11, p00, p11, p01};
        ti.insert(ti.end(), cell, cell + 6);
      }
  }
  out_xyz.swap(vx);
  out_idx.swap(ti);
  return true;
}

std::vector<float> lattice_xforms(int nx, int ny, int nz, float pitch) {
  std::vector<float> out;
  out.reserve(static_cast<size_t>(nx) * ny * nz * 12);
  const double p = static_cast<double>(pitch);
  for (int iz = 0; iz < nz; iz++)
    for (int iy = 0; iy < ny; iy++)
      for (int ix = 0; ix < nx; ix++) {
        const float m[12] = {1, 0, 0, static_cast<float>((ix - (nx - 1) / 2.0) * p),  //
                             0, 1, 0, static_cast<float>(iy * p),                     //
                             0, 0, 1, static_cast<float>(-iz * p)};
        out.insert(out.end(), m, m + 12);
      }
  return out;
}

LatticeView lattice_view(int nx, int ny, int nz, float pitch) {
  const double p = static_cast<double>(pitch);
  const double height = (ny - 1) * p + 2.0, width = (nx - 1) * p + 2.0;
  const double dist = std::max(height, width * 9.0 / 16.0) / 2.0 / 0.2027 * 1.15;
  LatticeView v;
  v.camera[0] = -0.001f;
  v.camera[1] = static_cast<float>(height / 2.0);
  v.camera[2] = static_cast<float>(1.0 + dist);
  v.z_far = static_cast<float>(dist + nz * p + 10.0);
  v.light[0] = 1.0f;
  v.light[1] = v.camera[1];
  v.light[2] = static_cast<float>(static_cast<double>(v.camera[2]) - 8.0);
  return v;
}

void scene_bounds(const std::vector<float>& xyz, const std::vector<uint32_t>& idx, const float* xforms, uint32_t n_inst, double lo[3],
                  double hi[3]) {
  double blo[3] = {1e300, 1e300, 1e300}, bhi[3] = {-1e300, -1e300, -1e300};
  for (uint32_t i : idx)
    for (int a = 0; a < 3; a++) {
      const double v = xyz[3 * static_cast<size_t>(i) + a];
      blo[a] = std::min(blo[a], v);
      bhi[a] = std::max(bhi[a], v);
    }
  if (!xforms || !n_inst) {
    for (int a = 0; a < 3; a++) {
      lo[a] = blo[a];
      hi[a] = bhi[a];
    }
    return;
  }
  for (int a = 0; a < 3; a++) {
    lo[a] = 1e300;
    hi[a] = -1e300;
  }
  for (uint32_t n = 0; n < n_inst; n++) {
    const float* m = xforms + 12 * static_cast<size_t>(n);
    for (int c = 0; c < 8; c++) {
      const double p[3] = {(c & 4) ? bhi[0] : blo[0], (c & 2) ? bhi[1] : blo[1], (c & 1) ? bhi[2] : blo[2]};
      for (int a = 0; a < 3; a++) {
        const double w = static_cast<double>(m[4 * a]) * p[0] + static_cast<double>(m[4 * a + 1]) * p[1] + static_cast<double>(m[4 * a + 2]) * p[2] +
                         static_cast<double>(m[4 * a + 3]);
        lo[a] = std::min(lo[a], w);
        hi[a] = std::max(hi[a], w);
      }
    }
  }
}

}  // namespace rtpt_host
