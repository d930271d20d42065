// scene_gen.hpp — the synthetic scene of BASELINE.json configs[4] ("instanced Cornell box x1000, ~1M triangles") for the
// C++ host.  The reference builds ONE bottom-level structure and ONE identity instance (main.cpp:728-741); SURVEY.md 8(d)
// defines the stress scene on top of that: every quad of the OBJ tessellated n x n (the normal-keyed albedo,
// raytrace.comp.glsl:155-163, is tessellation-invariant) and the mesh instanced on a lattice of translations, handed to
// rtpt_scene_upload as 3x4 transforms.  Same arithmetic, operation for operation, as the Python mirror (scenes.py), so both
// hosts upload the same bits (tests/test_host_logic.py compares the two).
#pragma once

#include <cstdint>
#include <vector>

namespace rtpt_host {

// Splits each fan-triangulated quad (triangles 2q = (a,b,c), 2q+1 = (a,c,d)) into an n x n grid of cells, two triangles per
// cell, same winding.  n = 6 turns 32 triangles into 1,152.  false: the mesh is not such a quad list (outputs untouched).
bool tessellate_quads(const std::vector<float>& xyz, const std::vector<uint32_t>& idx, int n, std::vector<float>& out_xyz,
                      std::vector<uint32_t>& out_idx);

// nx*ny*nz translations (3x4 row-major, 12 floats each): centred on x, resting on y = 0, receding along -z
std::vector<float> lattice_xforms(int nx, int ny, int nz, float pitch);

struct LatticeView {
  float camera[3];  // frames the front layer of the lattice with a margin (K2 camera: vertical slope tan(0.2))
  float light[3];   // 8 units in front of the camera, at its height
  float z_far;      // far plane of the G-buffer projection, behind the last layer
};
LatticeView lattice_view(int nx, int ny, int nz, float pitch);

// world-space bounds of `n_inst` instances (3x4 transforms; none: the mesh itself) of the referenced vertices, in double
void scene_bounds(const std::vector<float>& xyz, const std::vector<uint32_t>& idx, const float* xforms, uint32_t n_inst, double lo[3],
                  double hi[3]);

}  // namespace rtpt_host
