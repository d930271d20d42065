// api_scene.hip — the scene side of the C ABI: rtpt_scene_upload (loadMesh + buildAccelerationStructure, main.cpp:409-462,
// :687-742), the posed scene of a changed ubo.model (device-side refit, refit.hip), materials.
#include "api_internal.hpp"

extern "C" {

// ------------------------------------------------------------------------------------------ scene

int rtpt_scene_upload(rtpt_ctx* c, const float* xyz, uint32_t n_verts, const uint32_t* idx, uint32_t n_tris,
                      const float* xf, uint32_t n_instances) {
  if (!c || !xyz || !idx) return fail(RTPT_E_INVALID, "NULL argument");
  if (n_tris == 0 || n_verts == 0) return fail(RTPT_E_INVALID, "empty mesh");
  for (uint32_t i = 0; i < 3 * n_tris; i++)
    if (idx[i] >= n_verts) return fail(RTPT_E_INVALID, "index out of range");
  HIP_TRY(hipSetDevice(c->device));
  FLUSH_FILTER(c);
  const uint32_t ni = (xf && n_instances) ? n_instances : 1;
  const uint64_t total64 = static_cast<uint64_t>(ni) * n_tris;
  if (total64 >= 0xFFFFFFF0ull) return fail(RTPT_E_INVALID, "too many triangles");
  const uint32_t total = static_cast<uint32_t>(total64);
  // flattened world-space triangle soup, id = instance * n_tris + t  (one identity instance in the
  // reference, main.cpp:728-741)
  std::vector<float> tris(static_cast<size_t>(total) * 9);
  for (uint32_t inst = 0; inst < ni; inst++)
    for (uint32_t t = 0; t < n_tris; t++)
      for (int k = 0; k < 3; k++) {
        const float* v = xyz + 3 * static_cast<size_t>(idx[3 * t + k]);
        float* o = tris.data() + 9 * (static_cast<size_t>(inst) * n_tris + t) + 3 * k;
        if (xf && n_instances) {
          const float* m = xf + 12 * static_cast<size_t>(inst);
          for (int r = 0; r < 3; r++)
            o[r] = rt::fmaf_(m[4 * r + 2], v[2], rt::fmaf_(m[4 * r + 1], v[1], m[4 * r] * v[0])) + m[4 * r + 3];
        } else {
          o[0] = v[0];
          o[1] = v[1];
          o[2] = v[2];
        }
      }
  // fan pairs (a, b, c), (a, c, d): the posed records are computed from these vertices with one arithmetic, so bitwise
  // equality here is bitwise equality of v0 and of e2_A / e1_B on the device, whatever the model matrix
  bool paired_all = total >= 2 && total % 2 == 0;
  for (uint32_t q = 0; paired_all && q < total / 2; q++) {
    const float* ta = tris.data() + 18 * static_cast<size_t>(q);
    const float* tb = ta + 9;
    paired_all = std::memcmp(ta, tb, 12) == 0 && std::memcmp(ta + 6, tb + 3, 12) == 0;
  }
  const bool leaf_pairs = paired_all && !c->no_pairing;
  rt::Bvh bvh;  // built aside: a failed upload leaves the context's scene (and the topology a later refit uses) untouched
  rt::build_bvh(tris.data(), total, bvh, 1e-5f, leaf_pairs);
  if (bvh.max_depth >= rt::kBvhMaxDepth) return fail(RTPT_E_INVALID, "BVH deeper than the traversal stack");
  // the traversal addresses leaf records and nodes as base + 32-bit byte offset (48 bytes per triangle at most, 32 per node)
  if (static_cast<uint64_t>(total) * 48u >= (1ull << 32) || bvh.nodes.size() >= (1ull << 27))
    return fail(RTPT_E_INVALID, "scene too large for the traversal's 32-bit record offsets (more than 89,478,485 triangles)");
  if (bvh.leaf_order.size() != total) return fail(RTPT_E_INVALID, "internal: BVH lost triangles");

  HIP_TRY(hipStreamSynchronize(c->stream));
  int rc;
  if ((rc = alloc_buf(c->tris, tris.size() * sizeof(float)))) return rc;
  if ((rc = alloc_buf(c->leaf_order, static_cast<size_t>(total) * 4))) return rc;
  if ((rc = alloc_buf(c->isect_id, static_cast<size_t>(total) * 48))) return rc;
  if ((rc = alloc_buf(c->isect_leaf, static_cast<size_t>(total) * 48))) return rc;
  if ((rc = alloc_buf(c->shade, static_cast<size_t>(total) * 48))) return rc;
  std::vector<rt::BvhNodeQ> nodes_h;
  c->bvh_grid = rt::pack_quantised_nodes(bvh, nodes_h);
  if ((rc = alloc_buf(c->nodes, nodes_h.size() * sizeof(rt::BvhNodeQ)))) return rc;
  if ((rc = alloc_buf(c->normal_tab, (static_cast<size_t>(total) + 1) * 32))) return rc;  // normals, then per-id areas
  if ((rc = alloc_buf(c->pair_tab, total + 1 <= 64 ? (static_cast<size_t>(total) + 1) * (total + 1) * 4 : 0))) return rc;
  for (int i = 0; i < 2; i++)
    if ((rc = alloc_buf(c->lut[i], (static_cast<size_t>(total) + 1) * sizeof(rtpt_visibility_data)))) return rc;
  HIP_TRY(hipMemcpyAsync(c->tris.ptr, tris.data(), tris.size() * sizeof(float), hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipMemcpyAsync(c->leaf_order.ptr, bvh.leaf_order.data(), static_cast<size_t>(total) * 4, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipMemcpyAsync(c->nodes.ptr, nodes_h.data(), nodes_h.size() * sizeof(rt::BvhNodeQ), hipMemcpyHostToDevice, c->stream));
  // the grid of the node boxes, read by the traversal from device memory (a device-side refit rewrites it)
  if ((rc = alloc_buf(c->bvh_grid_dev, 8 * sizeof(float)))) return rc;
  const float grid_h[8] = {c->bvh_grid.origin[0], c->bvh_grid.origin[1], c->bvh_grid.origin[2], c->bvh_grid.cell[0],
                           c->bvh_grid.cell[1], c->bvh_grid.cell[2], 0.f, 0.f};
  HIP_TRY(hipMemcpyAsync(c->bvh_grid_dev.ptr, grid_h, sizeof grid_h, hipMemcpyHostToDevice, c->stream));
  // device-side refit tables: nodes by HEIGHT (a node after both of its subtrees), the un-posed triangles, scratch boxes
  std::vector<uint32_t> order_h;
  c->refit_level_first.clear();
  c->n_nodes = static_cast<uint32_t>(nodes_h.size());
  {
    const size_t nn = nodes_h.size();
    std::vector<int> height(nn, 0);
    int maxh = 0;
    for (size_t ii = nn; ii-- > 0;) {  // pre-order numbering: children carry larger indices than their parent
      int hgt = 0;
      for (uint32_t ref : {nodes_h[ii].lref, nodes_h[ii].rref})
        if (ref != rt::kBvhEmpty && !(ref & 0x80000000u) && ref < nn) hgt = std::max(hgt, height[ref] + 1);
      height[ii] = hgt;
      maxh = std::max(maxh, hgt);
    }
    c->refit_level_first.assign(static_cast<size_t>(maxh) + 2, 0);
    for (size_t ii = 0; ii < nn; ii++) c->refit_level_first[static_cast<size_t>(height[ii]) + 1]++;
    for (size_t h = 1; h < c->refit_level_first.size(); h++) c->refit_level_first[h] += c->refit_level_first[h - 1];
    order_h.resize(nn);
    std::vector<uint32_t> fill(c->refit_level_first.begin(), c->refit_level_first.end() - 1);
    for (size_t ii = 0; ii < nn; ii++) order_h[fill[static_cast<size_t>(height[ii])]++] = static_cast<uint32_t>(ii);
  }
  if ((rc = alloc_buf(c->obj_tris_dev, tris.size() * sizeof(float)))) return rc;
  if ((rc = alloc_buf(c->refit_order, order_h.size() * 4))) return rc;
  if ((rc = alloc_buf(c->refit_fbox, order_h.size() * 12 * sizeof(float)))) return rc;
  HIP_TRY(hipMemcpyAsync(c->obj_tris_dev.ptr, tris.data(), tris.size() * sizeof(float), hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipMemcpyAsync(c->refit_order.ptr, order_h.data(), order_h.size() * 4, hipMemcpyHostToDevice, c->stream));
  for (int i = 0; i < 2; i++) HIP_TRY(hipMemsetAsync(c->lut[i].ptr, 0, c->lut[i].bytes, c->stream));
  rt::ScenePrepArgs sp;
  sp.n_tris = total;
  sp.tris = static_cast<const float*>(c->tris.ptr);
  sp.leaf_order = static_cast<const uint32_t*>(c->leaf_order.ptr);
  sp.isect_id = static_cast<float4*>(c->isect_id.ptr);
  sp.isect_leaf = static_cast<float4*>(c->isect_leaf.ptr);
  sp.shade = static_cast<float4*>(c->shade.ptr);
  sp.leaf_pairs = leaf_pairs ? 1u : 0u;
  rt::launch_scene_prepare(sp, c->stream);
  if ((rc = launch_check("scene_prepare"))) return rc;
  HIP_TRY(hipStreamSynchronize(c->stream));  // host staging vectors die at return
  c->n_tris = total;
  c->n_base_tris = n_tris;
  free_buf(c->materials);  // materials belong to the mesh that was replaced
  free_buf(c->stack_spill);  // sized by the depth of the tree that was replaced
  c->stack_spill_blocks = 0;
  if (total <= static_cast<uint32_t>(rt::kCullMaxTris))
    c->host_tris = tris;
  else
    c->host_tris.clear();
  c->tris_paired = paired_all && total <= static_cast<uint32_t>(rt::kCullMaxTris);  // the brute-force loops
  c->leaf_pairs = leaf_pairs;                                                        // the tree that was just built
  c->obj_tris.swap(tris);
  c->bvh_host = std::move(bvh);  // only now: the upload succeeded
  for (int i = 0; i < 16; i++) c->model[i] = (i % 5 == 0) ? 1.0f : 0.0f;
  c->model_version++;
  c->use_bvh = (total > 64) || (c->cfg.flags & RTPT_FLAG_FORCE_BVH);
  c->use_bvh = (total > 64) || (c->cfg.flags & RTPT_FLAG_FORCE_BVH);
  c->bvh_depth = bvh.max_depth;
  c->lut_prev_valid = false;
  c->lut_version[0] = c->lut_version[1] = ~0ull;
  c->tables_valid = false;
  c->normals_y0 = c->normals_y1 = 0;  // the per-pixel normal plane belongs to the previous scene
  return RTPT_OK;
}


// Pose the scene with a new model matrix (visibility.vert.glsl:24 `model * position`; the reference recomputes
// ubo.model every frame, main.cpp:1469, as the identity): world triangle = model * uploaded triangle, in the same
// fixed-order fma arithmetic the LUT uses (mat_row_point), the BVH keeps its topology and is REFIT to the moved
// triangles, the device records are rebuilt.  Every pass — K0, K2, the LUT — sees the posed geometry.
}  // extern "C"

// ubo.model changed (main.cpp:1469 recomputes it every frame; an animated scene passes another one): re-pose the scene
int rtpt_impl::apply_model(rtpt_ctx* c, const float* model) {
  const uint32_t total = c->n_tris;
  const bool ident = is_identity(model);
  if (c->use_bvh && !c->host_refit && c->obj_tris_dev.ptr && c->refit_order.ptr) {
    // everything on the device and on the context's stream: no upload, no synchronisation (refit.hip)
    rt::RefitModel rm;
    std::memcpy(rm.m, model, sizeof rm.m);
    rm.identity = ident ? 1 : 0;
    rt::launch_pose(total * 3, static_cast<const float*>(c->obj_tris_dev.ptr), static_cast<float*>(c->tris.ptr), rm, c->stream);
    rt::RefitArgs ra;
    ra.tris = static_cast<const float*>(c->tris.ptr);
    ra.leaf_order = static_cast<const uint32_t*>(c->leaf_order.ptr);
    ra.order = static_cast<const uint32_t*>(c->refit_order.ptr);
    ra.nodes = static_cast<rt::BvhNodeQ*>(c->nodes.ptr);
    ra.fbox = static_cast<float*>(c->refit_fbox.ptr);
    ra.grid = static_cast<float*>(c->bvh_grid_dev.ptr);
    rt::launch_refit(ra, c->refit_level_first.data(), static_cast<int>(c->refit_level_first.size()) - 1, c->n_nodes, 1e-5f, c->stream);
    rt::ScenePrepArgs sp;
    sp.n_tris = total;
    sp.tris = static_cast<const float*>(c->tris.ptr);
    sp.leaf_order = static_cast<const uint32_t*>(c->leaf_order.ptr);
    sp.isect_id = static_cast<float4*>(c->isect_id.ptr);
    sp.isect_leaf = static_cast<float4*>(c->isect_leaf.ptr);
    sp.shade = static_cast<float4*>(c->shade.ptr);
    sp.leaf_pairs = c->leaf_pairs ? 1u : 0u;
    rt::launch_scene_prepare(sp, c->stream);
    int rcd = launch_check("device refit");
    if (rcd) return rcd;
    if (total <= static_cast<uint32_t>(rt::kCullMaxTris)) {
      // a small scene forced onto the BVH path: the screen bounds (unused while it is) still follow the pose
      c->host_tris.resize(static_cast<size_t>(total) * 9);
      for (size_t v = 0; v < static_cast<size_t>(total) * 3; v++) {
        const float* p = c->obj_tris.data() + 3 * v;
        const rt::f3 q{p[0], p[1], p[2]};
        float* o = c->host_tris.data() + 3 * v;
        o[0] = ident ? p[0] : rt::exact::mat_row_point(model, 0, q);
        o[1] = ident ? p[1] : rt::exact::mat_row_point(model, 1, q);
        o[2] = ident ? p[2] : rt::exact::mat_row_point(model, 2, q);
      }
    }
    std::memcpy(c->model, model, sizeof c->model);
    c->model_version++;
    c->tables_valid = false;
    return RTPT_OK;
  }
  std::vector<float> tris(static_cast<size_t>(total) * 9);
  for (size_t v = 0; v < static_cast<size_t>(total) * 3; v++) {
    const float* p = c->obj_tris.data() + 3 * v;
    float* o = tris.data() + 3 * v;
    if (ident) {
      o[0] = p[0]; o[1] = p[1]; o[2] = p[2];
    } else {
      const rt::f3 q{p[0], p[1], p[2]};
      o[0] = rt::exact::mat_row_point(model, 0, q);
      o[1] = rt::exact::mat_row_point(model, 1, q);
      o[2] = rt::exact::mat_row_point(model, 2, q);
    }
  }
  rt::refit_bvh(tris.data(), total, c->bvh_host);
  std::vector<rt::BvhNodeQ> nodes_h;
  c->bvh_grid = rt::pack_quantised_nodes(c->bvh_host, nodes_h);
  if (nodes_h.size() * sizeof(rt::BvhNodeQ) != c->nodes.bytes) return fail(RTPT_E_INVALID, "internal: refit changed the node count");
  HIP_TRY(hipMemcpyAsync(c->tris.ptr, tris.data(), tris.size() * sizeof(float), hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipMemcpyAsync(c->nodes.ptr, nodes_h.data(), nodes_h.size() * sizeof(rt::BvhNodeQ), hipMemcpyHostToDevice, c->stream));
  const float grid_h[8] = {c->bvh_grid.origin[0], c->bvh_grid.origin[1], c->bvh_grid.origin[2], c->bvh_grid.cell[0],
                           c->bvh_grid.cell[1], c->bvh_grid.cell[2], 0.f, 0.f};
  HIP_TRY(hipMemcpyAsync(c->bvh_grid_dev.ptr, grid_h, sizeof grid_h, hipMemcpyHostToDevice, c->stream));
  rt::ScenePrepArgs sp;
  sp.n_tris = total;
  sp.tris = static_cast<const float*>(c->tris.ptr);
  sp.leaf_order = static_cast<const uint32_t*>(c->leaf_order.ptr);
  sp.isect_id = static_cast<float4*>(c->isect_id.ptr);
  sp.isect_leaf = static_cast<float4*>(c->isect_leaf.ptr);
  sp.shade = static_cast<float4*>(c->shade.ptr);
  sp.leaf_pairs = c->leaf_pairs ? 1u : 0u;
  rt::launch_scene_prepare(sp, c->stream);
  int rc = launch_check("scene_prepare");
  if (rc) return rc;
  HIP_TRY(hipStreamSynchronize(c->stream));  // host staging vectors die at return
  if (total <= static_cast<uint32_t>(rt::kCullMaxTris)) c->host_tris.swap(tris);
  std::memcpy(c->model, model, sizeof c->model);
  c->model_version++;
  c->tables_valid = false;  // per-id normals and pair weights follow the posed triangles
  return RTPT_OK;
}

extern "C" {

int rtpt_scene_set_materials(rtpt_ctx* c, const uint32_t* tri_material, uint32_t n_tris, const rtpt_material* materials,
                             uint32_t n_materials) {
  if (!c) return fail(RTPT_E_INVALID, "ctx is NULL");
  if (!c->n_tris) return fail(RTPT_E_NO_SCENE, "rtpt_scene_upload has not been called");
  HIP_TRY(hipSetDevice(c->device));
  FLUSH_FILTER(c);
  HIP_TRY(hipStreamSynchronize(c->stream));
  if (!tri_material || !materials || !n_materials) {  // back to the reference's normal-keyed colours
    free_buf(c->materials);
    return RTPT_OK;
  }
  if (n_tris != c->n_base_tris) return fail(RTPT_E_INVALID, "one material index per triangle of the uploaded mesh");
  std::vector<float> rec(static_cast<size_t>(n_tris) * 8);
  for (uint32_t t = 0; t < n_tris; t++) {
    if (tri_material[t] >= n_materials) return fail(RTPT_E_INVALID, "material index out of range");
    const rtpt_material& m = materials[tri_material[t]];
    float* r = rec.data() + 8 * static_cast<size_t>(t);
    r[0] = m.albedo[0]; r[1] = m.albedo[1]; r[2] = m.albedo[2]; r[3] = 0.0f;
    r[4] = m.emission[0]; r[5] = m.emission[1]; r[6] = m.emission[2];
    r[7] = (m.emission[0] != 0.0f || m.emission[1] != 0.0f || m.emission[2] != 0.0f) ? 1.0f : 0.0f;
  }
  int rc = alloc_buf(c->materials, rec.size() * sizeof(float));
  if (rc) return rc;
  HIP_TRY(hipMemcpyAsync(c->materials.ptr, rec.data(), rec.size() * sizeof(float), hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return RTPT_OK;
}

}  // extern "C"
