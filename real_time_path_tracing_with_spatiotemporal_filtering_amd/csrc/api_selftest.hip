// api_selftest.hip — self tests of the device arithmetic and of the acceleration structure, and the host-side stand-ins
// for glm / tinyobjloader (rtpt_util_*): nothing a frame calls.
#include "api_internal.hpp"

extern "C" {

// ------------------------------------------------------------------------------------------ self tests
int rtpt_selftest_math(rtpt_ctx* c, int op, const float* in, float* out, size_t n) {
  if (!c || !in || !out) return fail(RTPT_E_INVALID, "NULL argument");
  if (n == 0) return RTPT_OK;
  HIP_TRY(hipSetDevice(c->device));
  float *din = nullptr, *dout = nullptr;
  HIP_TRY(hipMalloc(&din, n * 4));
  if (hipMalloc(&dout, n * 4) != hipSuccess) {
    (void)hipFree(din);
    return fail(RTPT_E_NOMEM, "hipMalloc");
  }
  hipError_t e = hipMemcpyAsync(din, in, n * 4, hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess) {
    rt::launch_selftest_math(op, din, dout, n, c->stream);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpyAsync(out, dout, n * 4, hipMemcpyDeviceToHost, c->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  (void)hipFree(din);
  (void)hipFree(dout);
  if (e != hipSuccess) return fail(RTPT_E_DEVICE, std::string("selftest_math: ") + hipGetErrorString(e));
  return RTPT_OK;
}

int rtpt_selftest_exhaustive(rtpt_ctx* c, int op, uint64_t* mismatches, uint32_t first_bad[4]) {
  if (!c || !mismatches) return fail(RTPT_E_INVALID, "NULL argument");
  if (op != 3 && op != 4) return fail(RTPT_E_INVALID, "rtpt_selftest_exhaustive: op must be 3 (sqrt) or 4 (1/x)");
  HIP_TRY(hipSetDevice(c->device));
  unsigned long long* d = nullptr;
  HIP_TRY(hipMalloc(&d, 5 * sizeof(unsigned long long)));
  unsigned long long h[5] = {0, 0, 0, 0, 0};
  hipError_t e = hipMemsetAsync(d, 0, sizeof h, c->stream);
  if (e == hipSuccess) {
    rt::launch_selftest_exhaustive(op, d, c->stream);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpyAsync(h, d, sizeof h, hipMemcpyDeviceToHost, c->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  (void)hipFree(d);
  if (e != hipSuccess) return fail(RTPT_E_DEVICE, std::string("selftest_exhaustive: ") + hipGetErrorString(e));
  *mismatches = h[0];
  if (first_bad)
    for (int i = 0; i < 4; i++) first_bad[i] = static_cast<uint32_t>(h[1 + i]);
  return RTPT_OK;
}

int rtpt_selftest_div(rtpt_ctx* c, int mode, uint32_t first_pass, uint32_t n_passes, uint64_t* mismatches, uint32_t first_bad[2]) {
  if (!c || !mismatches) return fail(RTPT_E_INVALID, "NULL argument");
  if (mode != 0 && mode != 1) return fail(RTPT_E_INVALID, "rtpt_selftest_div: mode must be 0 (significand pairs) or 1 (arbitrary bits)");
  if (mode == 0 && (first_pass >= 256u || n_passes > 256u - first_pass))
    return fail(RTPT_E_INVALID, "rtpt_selftest_div: the enumeration has 256 passes");
  HIP_TRY(hipSetDevice(c->device));
  unsigned long long* d = nullptr;
  HIP_TRY(hipMalloc(&d, 3 * sizeof(unsigned long long)));
  unsigned long long h[3] = {0, 0, 0};
  hipError_t e = hipMemsetAsync(d, 0, sizeof h, c->stream);
  for (uint32_t p = 0; e == hipSuccess && p < n_passes; p++) {
    rt::launch_selftest_div(mode, first_pass + p, d, c->stream);
    e = hipGetLastError();
    if (e == hipSuccess && (p & 7u) == 7u) e = hipStreamSynchronize(c->stream);  // ~0.15 s per pass: keep the queue short
  }
  if (e == hipSuccess) e = hipMemcpyAsync(h, d, sizeof h, hipMemcpyDeviceToHost, c->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  (void)hipFree(d);
  if (e != hipSuccess) return fail(RTPT_E_DEVICE, std::string("selftest_div: ") + hipGetErrorString(e));
  *mismatches = h[0];
  if (first_bad) {
    first_bad[0] = static_cast<uint32_t>(h[1]);
    first_bad[1] = static_cast<uint32_t>(h[2]);
  }
  return RTPT_OK;
}

int rtpt_selftest_trace(rtpt_ctx* c, const float* rays, size_t n, uint32_t* out_id, float* out_t) {
  if (!c || !rays || !out_id) return fail(RTPT_E_INVALID, "NULL argument");
  if (!c->n_tris) return fail(RTPT_E_NO_SCENE, "rtpt_scene_upload has not been called");
  if (n == 0) return RTPT_OK;
  HIP_TRY(hipSetDevice(c->device));
  FLUSH_FILTER(c);  // a recorded G-buffer call holds the scene view, and with it the stack's spill area, which may move below
  float *drays = nullptr, *dt = nullptr;
  uint32_t* did = nullptr;
  hipError_t e = hipMalloc(&drays, n * 24);
  if (e == hipSuccess) e = hipMalloc(&did, n * 4);
  if (e == hipSuccess) e = hipMalloc(&dt, n * 4);
  if (e == hipSuccess) e = hipMemcpyAsync(drays, rays, n * 24, hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess) {
    if (int rcs = ensure_stack_spill(c, std::max(frame_blocks(c), (n + 255) / 256))) {
      (void)hipFree(drays);
      (void)hipFree(did);
      (void)hipFree(dt);
      return rcs;
    }
    rt::launch_selftest_trace(scene_view(c), drays, n, c->cfg.ray_tmax, did, dt, c->stream);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpyAsync(out_id, did, n * 4, hipMemcpyDeviceToHost, c->stream);
  if (e == hipSuccess && out_t) e = hipMemcpyAsync(out_t, dt, n * 4, hipMemcpyDeviceToHost, c->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  (void)hipFree(drays);
  (void)hipFree(did);
  (void)hipFree(dt);
  if (e != hipSuccess) return fail(RTPT_E_DEVICE, std::string("selftest_trace: ") + hipGetErrorString(e));
  return RTPT_OK;
}

// ------------------------------------------------------------------------------------------ host helpers
void rtpt_util_look_at(const float eye[3], const float center[3], const float up[3], float m[16]) {
  // glm::lookAtRH (main.cpp:482, :1470)
  using namespace rt;
  f3 e{eye[0], eye[1], eye[2]};
  f3 f = exact::normalize(f3{center[0], center[1], center[2]} - e);
  f3 s = exact::normalize(exact::cross(f, f3{up[0], up[1], up[2]}));
  f3 u = exact::cross(s, f);
  std::memset(m, 0, 16 * sizeof(float));
  m[0] = s.x; m[4] = s.y; m[8] = s.z;
  m[1] = u.x; m[5] = u.y; m[9] = u.z;
  m[2] = -f.x; m[6] = -f.y; m[10] = -f.z;
  m[12] = -exact::dot(s, e);
  m[13] = -exact::dot(u, e);
  m[14] = exact::dot(f, e);
  m[15] = 1.0f;
}

void rtpt_util_perspective(float fovy, float aspect, float zn, float zf, float m[16]) {
  // glm::perspectiveRH_ZO (D6; main.cpp:483, :1471)
  const float t = static_cast<float>(std::tan(static_cast<double>(fovy) * 0.5));
  std::memset(m, 0, 16 * sizeof(float));
  m[0] = 1.0f / (aspect * t);
  m[5] = 1.0f / t;
  m[10] = zf / (zn - zf);
  m[11] = -1.0f;
  m[14] = -(zf * zn) / (zf - zn);
}

// The acceleration structure AS IT STANDS ON THE DEVICE (after rtpt_scene_upload, or after a model matrix re-posed and
// refit it inside rtpt_gbuffer): nodes, grid, posed triangles and leaf order are read back and checked on the host.
//   stats[0] nodes, [1] leaves, [2] deepest level, [3] largest leaf, [4] triangles not referenced exactly once,
//   [5] decoded (origin + q * cell, binary32) child boxes that do not contain every vertex below them,
//   [6] child boxes wider than the padded scene (a box that was never rewritten), [7] dangling references
int rtpt_debug_bvh_check(rtpt_ctx* c, uint64_t stats[8]) {
  if (!c || !stats) return fail(RTPT_E_INVALID, "NULL argument");
  if (!c->n_tris || !c->nodes.ptr) return fail(RTPT_E_NO_SCENE, "rtpt_scene_upload has not been called");
  HIP_TRY(hipSetDevice(c->device));
  FLUSH_FILTER(c);
  const uint32_t n = c->n_tris, nn = c->n_nodes;
  std::vector<rt::BvhNodeQ> q(nn);
  std::vector<float> tris(static_cast<size_t>(n) * 9);
  std::vector<uint32_t> leaf(n);
  float g[8];
  HIP_TRY(hipMemcpyAsync(q.data(), c->nodes.ptr, q.size() * sizeof(rt::BvhNodeQ), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipMemcpyAsync(tris.data(), c->tris.ptr, tris.size() * sizeof(float), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipMemcpyAsync(leaf.data(), c->leaf_order.ptr, leaf.size() * 4, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipMemcpyAsync(g, c->bvh_grid_dev.ptr, sizeof g, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  for (int i = 0; i < 8; i++) stats[i] = 0;
  stats[0] = nn;
  std::vector<uint32_t> seen(n, 0);
  for (uint32_t id : leaf) {
    if (id >= n)
      stats[4]++;
    else
      seen[id]++;
  }
  for (uint32_t i = 0; i < n; i++)
    if (seen[i] != 1) stats[4]++;
  struct Bounds { float mn[3], mx[3]; };
  std::vector<Bounds> sub(nn);
  std::vector<uint8_t> done(nn, 0);
  std::vector<std::pair<uint32_t, uint32_t>> st{{0u, 0u}};  // node, level
  while (!st.empty()) {
    const uint32_t ni = st.back().first, lvl = st.back().second;
    stats[2] = std::max<uint64_t>(stats[2], lvl);
    const rt::BvhNodeQ& nd = q[ni];
    bool ready = true;
    for (uint32_t ref : {nd.lref, nd.rref}) {
      if (ref == rt::kBvhEmpty || (ref & 0x80000000u)) continue;
      if (ref >= nn || ref <= ni) {  // pre-order: a child comes after its parent
        stats[7]++;
        continue;
      }
      if (!done[ref]) {
        st.push_back({ref, lvl + 1});
        ready = false;
      }
    }
    if (!ready) continue;
    st.pop_back();
    Bounds me{{FLT_MAX, FLT_MAX, FLT_MAX}, {-FLT_MAX, -FLT_MAX, -FLT_MAX}};
    for (int side = 0; side < 2; side++) {
      const uint32_t ref = side ? nd.rref : nd.lref;
      if (ref == rt::kBvhEmpty) continue;
      Bounds cb{{FLT_MAX, FLT_MAX, FLT_MAX}, {-FLT_MAX, -FLT_MAX, -FLT_MAX}};
      if (ref & 0x80000000u) {
        const uint32_t first = (ref & 0x7FFFFFFFu) >> 2, cnt = (ref & 3u) + 1u;
        stats[1]++;
        stats[3] = std::max<uint64_t>(stats[3], cnt);
        if (static_cast<uint64_t>(first) + cnt > n) {
          stats[7]++;
          continue;
        }
        for (uint32_t j = 0; j < cnt; j++)
          for (int v = 0; v < 3; v++)
            for (int a = 0; a < 3; a++) {
              const float x = tris[9 * static_cast<size_t>(leaf[first + j]) + 3 * v + a];
              cb.mn[a] = std::min(cb.mn[a], x);
              cb.mx[a] = std::max(cb.mx[a], x);
            }
      } else {
        if (ref >= nn || !done[ref]) continue;
        cb = sub[ref];
      }
      for (int a = 0; a < 3; a++) {
        const float qlo = std::fmaf(static_cast<float>(nd.box[rt::bvh_box_lo(side, a)]), g[3 + a], g[a]);  // the decode refit.hip checks against
        const float qhi = std::fmaf(static_cast<float>(nd.box[rt::bvh_box_hi(side, a)]), g[3 + a], g[a]);  // the decode refit.hip checks against
        if (!(qlo <= cb.mn[a] && qhi >= cb.mx[a])) stats[5]++;
        me.mn[a] = std::min(me.mn[a], cb.mn[a]);
        me.mx[a] = std::max(me.mx[a], cb.mx[a]);
      }
    }
    sub[ni] = me;
    done[ni] = 1;
  }
  // no child box may be wider than the root's: the grid spans the padded scene plus one cell at either end
  if (nn) {
    const Bounds& sc = sub[0];
    float diag = 0.f, mag = 0.f;
    for (int a = 0; a < 3; a++) {
      diag += (sc.mx[a] - sc.mn[a]) * (sc.mx[a] - sc.mn[a]);
      mag = std::max(mag, std::max(std::fabs(sc.mn[a]), std::fabs(sc.mx[a])));
    }
    const float pad = 1e-5f * std::max(std::sqrt(diag), mag);  // bvh.cpp / refit.hip: the padding of every box
    for (uint32_t ni = 0; ni < nn; ni++)
      for (int side = 0; side < 2; side++) {
        if ((side ? q[ni].rref : q[ni].lref) == rt::kBvhEmpty) continue;
        for (int a = 0; a < 3; a++) {
          const float slack = 4.0f * g[3 + a] + 2.0f * pad;
          const float qlo = std::fmaf(static_cast<float>(q[ni].box[rt::bvh_box_lo(side, a)]), g[3 + a], g[a]);  // the decode refit.hip checks against
          const float qhi = std::fmaf(static_cast<float>(q[ni].box[rt::bvh_box_hi(side, a)]), g[3 + a], g[a]);  // the decode refit.hip checks against
          if (qlo < sc.mn[a] - slack || qhi > sc.mx[a] + slack) stats[6]++;
        }
      }
  }
  return RTPT_OK;
}

static int bvh_check_impl(const float* build_tris, const float* tris, uint32_t n, uint64_t stats[8]) {
  if (!tris || !stats || n == 0) return fail(RTPT_E_INVALID, "NULL argument / empty scene");
  rt::Bvh bvh;
  rt::build_bvh(build_tris ? build_tris : tris, n, bvh);
  if (build_tris) rt::refit_bvh(tris, n, bvh);  // same topology, boxes recomputed for the moved triangles
  std::vector<rt::BvhNodeQ> q;
  const rt::BvhGrid g = rt::pack_quantised_nodes(bvh, q);
  for (int i = 0; i < 8; i++) stats[i] = 0;
  stats[0] = bvh.nodes.size();
  stats[2] = static_cast<uint64_t>(bvh.max_depth);
  std::vector<uint32_t> seen(n, 0);
  if (bvh.leaf_order.size() != n) stats[4] += 1;
  for (uint32_t id : bvh.leaf_order) {
    if (id >= n) {
      stats[4]++;
      continue;
    }
    seen[id]++;
  }
  for (uint32_t i = 0; i < n; i++)
    if (seen[i] != 1) stats[4]++;
  struct Item { uint32_t node; };
  // bounds of a subtree = union of the triangles below it: computed bottom-up by recursion with an explicit stack
  struct Bounds { float mn[3], mx[3]; };
  auto tri_bounds = [&](uint32_t first, uint32_t cnt) {
    Bounds b{{FLT_MAX, FLT_MAX, FLT_MAX}, {-FLT_MAX, -FLT_MAX, -FLT_MAX}};
    for (uint32_t j = 0; j < cnt; j++) {
      const uint32_t id = bvh.leaf_order[first + j];
      for (int v = 0; v < 3; v++)
        for (int a = 0; a < 3; a++) {
          const float x = tris[9 * static_cast<size_t>(id) + 3 * v + a];
          b.mn[a] = std::min(b.mn[a], x);
          b.mx[a] = std::max(b.mx[a], x);
        }
    }
    return b;
  };
  std::vector<Bounds> sub(bvh.nodes.size());
  std::vector<uint8_t> done(bvh.nodes.size(), 0);
  std::vector<uint32_t> st{0};
  while (!st.empty()) {
    const uint32_t ni = st.back();
    const rt::BvhNode& nd = bvh.nodes[ni];
    bool ready = true;
    for (int side = 0; side < 2; side++) {
      const uint32_t idx = side ? nd.ridx : nd.lidx, cnt = side ? nd.rcnt : nd.lcnt;
      if (idx == rt::kBvhEmpty || cnt) continue;
      if (idx >= bvh.nodes.size()) {
        stats[7]++;
        continue;
      }
      if (!done[idx]) {
        st.push_back(idx);
        ready = false;
      }
    }
    if (!ready) continue;
    st.pop_back();
    Bounds me{{FLT_MAX, FLT_MAX, FLT_MAX}, {-FLT_MAX, -FLT_MAX, -FLT_MAX}};
    for (int side = 0; side < 2; side++) {
      const uint32_t idx = side ? nd.ridx : nd.lidx, cnt = side ? nd.rcnt : nd.lcnt;
      const float* bmn = side ? nd.rmin : nd.lmin;
      const float* bmx = side ? nd.rmax : nd.lmax;
      if (idx == rt::kBvhEmpty) continue;
      Bounds cb;
      if (cnt) {
        stats[1]++;
        stats[3] = std::max<uint64_t>(stats[3], cnt);
        if (cnt > static_cast<uint32_t>(rt::kBvhMaxLeaf) || static_cast<uint64_t>(idx) + cnt > n) {
          stats[7]++;
          continue;
        }
        cb = tri_bounds(idx, cnt);
      } else {
        if (idx >= bvh.nodes.size()) continue;
        cb = sub[idx];
      }
      for (int a = 0; a < 3; a++) {
        if (!(bmn[a] <= cb.mn[a] && bmx[a] >= cb.mx[a])) stats[5]++;
        // the device box: origin + q * cell, evaluated as the traversal's arithmetic implies (binary32)
        const float qlo = std::fmaf(static_cast<float>(q[ni].box[rt::bvh_box_lo(side, a)]), g.cell[a], g.origin[a]);
        const float qhi = std::fmaf(static_cast<float>(q[ni].box[rt::bvh_box_hi(side, a)]), g.cell[a], g.origin[a]);
        if (!(qlo <= bmn[a] && qhi >= bmx[a])) stats[6]++;
        me.mn[a] = std::min(me.mn[a], cb.mn[a]);
        me.mx[a] = std::max(me.mx[a], cb.mx[a]);
      }
      const uint32_t want = cnt ? (0x80000000u | (idx << 2) | (cnt - 1u)) : idx;
      if ((side ? q[ni].rref : q[ni].lref) != want) stats[7]++;
    }
    sub[ni] = me;
    done[ni] = 1;
  }
  return RTPT_OK;
}

int rtpt_util_bvh_check(const float* tris, uint32_t n, uint64_t stats[8]) { return bvh_check_impl(nullptr, tris, n, stats); }
int rtpt_util_bvh_refit_check(const float* built_for, const float* moved, uint32_t n, uint64_t stats[8]) {
  if (!built_for) return fail(RTPT_E_INVALID, "NULL argument");
  return bvh_check_impl(built_for, moved, n, stats);
}

int rtpt_util_load_obj(const char* path, float* xyz, uint32_t* n_verts, uint32_t* idx, uint32_t* n_tris) {
  if (!path || !n_verts || !n_tris) return fail(RTPT_E_INVALID, "NULL argument");
  FILE* fp = std::fopen(path, "r");
  if (!fp) return fail(RTPT_E_INVALID, std::string("cannot open ") + path);
  std::vector<long> poly;
  uint32_t nv = 0, nt = 0;
  char line[2048];
  int rc = RTPT_OK;
  while (std::fgets(line, sizeof line, fp)) {
    const char* p = line;
    while (*p == ' ' || *p == '\t') p++;
    if (p[0] == 'v' && (p[1] == ' ' || p[1] == '\t')) {
      char* end = nullptr;
      float v[3];
      const char* q = p + 2;
      bool ok = true;
      for (int k = 0; k < 3; k++) {
        v[k] = std::strtof(q, &end);
        if (end == q) ok = false;
        q = end;
      }
      if (!ok) continue;
      if (xyz) std::memcpy(xyz + 3 * static_cast<size_t>(nv), v, sizeof v);
      nv++;
    } else if (p[0] == 'f' && (p[1] == ' ' || p[1] == '\t')) {
      poly.clear();
      const char* q = p + 2;
      while (*q) {
        while (*q == ' ' || *q == '\t') q++;
        if (*q == '\0' || *q == '\n' || *q == '\r') break;
        char* end = nullptr;
        long v = std::strtol(q, &end, 10);
        if (end == q) break;
        long resolved = v > 0 ? v - 1 : static_cast<long>(nv) + v;  // OBJ indices are 1-based; negative = relative
        if (resolved < 0 || resolved >= static_cast<long>(nv)) rc = fail(RTPT_E_INVALID, "OBJ face index out of range");
        poly.push_back(resolved);
        q = end;
        while (*q && *q != ' ' && *q != '\t' && *q != '\n' && *q != '\r') q++;  // skip "/vt/vn"
      }
      for (size_t k = 1; k + 1 < poly.size(); k++) {  // D5: fan triangulation in file order
        if (idx) {
          idx[3 * static_cast<size_t>(nt)] = static_cast<uint32_t>(poly[0]);
          idx[3 * static_cast<size_t>(nt) + 1] = static_cast<uint32_t>(poly[k]);
          idx[3 * static_cast<size_t>(nt) + 2] = static_cast<uint32_t>(poly[k + 1]);
        }
        nt++;
      }
    }
  }
  std::fclose(fp);
  *n_verts = nv;
  *n_tris = nt;
  return rc;
}


// Materials of an OBJ (tinyobjloader hands main.cpp:416-428 the same information, which the
// reference ignores — its colours are keyed on the normal, raytrace.comp.glsl:155-163, and the .mtl its OBJ names is
// missing upstream).  `mtllib` files are looked up next to the OBJ; `usemtl` selects the material of the faces that
// follow; a face fan-triangulates into poly - 2 triangles exactly like rtpt_util_load_obj (D5), so tri_material lines
// up with its index array.  Material 0 is the default (Kd 0.7, the reference's grey; Ke 0) for faces without a usable
// `usemtl`.  A missing library is not an error: *n_materials comes back 0 and the caller keeps the normal-keyed colours.
int rtpt_util_load_obj_materials(const char* path, uint32_t* tri_material, uint32_t* n_tris, rtpt_material* materials,
                                 uint32_t* n_materials) {
  if (!path || !n_tris || !n_materials) return fail(RTPT_E_INVALID, "NULL argument");
  FILE* fp = std::fopen(path, "r");
  if (!fp) return fail(RTPT_E_INVALID, std::string("cannot open ") + path);
  std::string dir(path);
  const size_t slash = dir.find_last_of('/');
  dir = slash == std::string::npos ? std::string() : dir.substr(0, slash + 1);
  std::vector<std::string> names{"<default>"};
  std::vector<rtpt_material> mats(1);
  mats[0] = rtpt_material{{0.7f, 0.7f, 0.7f}, {0.f, 0.f, 0.f}};
  bool any_library = false;
  auto word = [](const char* q, std::string& out) {
    while (*q == ' ' || *q == '\t') q++;
    out.clear();
    while (*q && *q != ' ' && *q != '\t' && *q != '\n' && *q != '\r') out.push_back(*q++);
  };
  auto load_mtl = [&](const std::string& file) {
    FILE* mf = std::fopen((dir + file).c_str(), "r");
    if (!mf) return;
    any_library = true;
    char ln[1024];
    int cur = -1;
    while (std::fgets(ln, sizeof ln, mf)) {
      const char* p = ln;
      while (*p == ' ' || *p == '\t') p++;
      if (!std::strncmp(p, "newmtl", 6) && (p[6] == ' ' || p[6] == '\t')) {
        std::string nm;
        word(p + 6, nm);
        names.push_back(nm);
        mats.push_back(rtpt_material{{0.7f, 0.7f, 0.7f}, {0.f, 0.f, 0.f}});
        cur = static_cast<int>(mats.size()) - 1;
      } else if (cur >= 0 && (p[0] == 'K') && (p[1] == 'd' || p[1] == 'e') && (p[2] == ' ' || p[2] == '\t')) {
        float v[3];
        if (std::sscanf(p + 2, "%f %f %f", &v[0], &v[1], &v[2]) == 3)
          std::memcpy(p[1] == 'd' ? mats[static_cast<size_t>(cur)].albedo : mats[static_cast<size_t>(cur)].emission, v, sizeof v);
      }
    }
    std::fclose(mf);
  };
  uint32_t nt = 0, cur_mat = 0;
  char line[2048];
  while (std::fgets(line, sizeof line, fp)) {
    const char* p = line;
    while (*p == ' ' || *p == '\t') p++;
    if (!std::strncmp(p, "mtllib", 6) && (p[6] == ' ' || p[6] == '\t')) {
      std::string file;
      word(p + 6, file);
      load_mtl(file);
    } else if (!std::strncmp(p, "usemtl", 6) && (p[6] == ' ' || p[6] == '\t')) {
      std::string nm;
      word(p + 6, nm);
      cur_mat = 0;
      for (size_t i = 1; i < names.size(); i++)
        if (names[i] == nm) cur_mat = static_cast<uint32_t>(i);
    } else if (p[0] == 'f' && (p[1] == ' ' || p[1] == '\t')) {
      size_t corners = 0;
      const char* q = p + 2;
      while (*q) {
        while (*q == ' ' || *q == '\t') q++;
        if (*q == '\0' || *q == '\n' || *q == '\r') break;
        char* end = nullptr;
        (void)std::strtol(q, &end, 10);
        if (end == q) break;
        corners++;
        q = end;
        while (*q && *q != ' ' && *q != '\t' && *q != '\n' && *q != '\r') q++;
      }
      for (size_t k = 1; k + 1 < corners; k++) {
        if (tri_material) tri_material[nt] = cur_mat;
        nt++;
      }
    }
  }
  std::fclose(fp);
  *n_tris = nt;
  if (!any_library) {
    *n_materials = 0;
    return RTPT_OK;
  }
  if (materials) {
    if (*n_materials < mats.size()) return fail(RTPT_E_INVALID, "materials array too small");
    std::memcpy(materials, mats.data(), mats.size() * sizeof(rtpt_material));
  }
  *n_materials = static_cast<uint32_t>(mats.size());
  return RTPT_OK;
}

}  // extern "C"
