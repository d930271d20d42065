// api_passes.hip — one entry point per reference dispatch (K0 rtpt_gbuffer, K1 rtpt_temporal_gradient, K2 rtpt_raytrace,
// K3 rtpt_temporal_filter, K4 rtpt_end_frame, the swapchain blit rtpt_present) and what they record: K0 / K1 until K2 arrives,
// the filter iterations until the last one arrives (api_internal.hpp: FLUSH_FILTER).
#include "api_internal.hpp"

extern "C" {

// ------------------------------------------------------------------------------------------ K0
int rtpt_gbuffer(rtpt_ctx* c, const rtpt_ubo* ubo, uint32_t y0, uint32_t y1) {
  if (!c || !ubo) return fail(RTPT_E_INVALID, "NULL argument");
  if (!c->n_tris) return fail(RTPT_E_NO_SCENE, "rtpt_scene_upload has not been called");
  int rc = check_rows(c, y0, y1);
  if (rc) return rc;
  FLUSH_FILTER(c);
  {
    // an affine model only (last row 0 0 0 1): the posed vertex is the xyz of model * (v, 1), visibility.vert.glsl:24
    const float* m = ubo->model;
    if (!(m[3] == 0.0f && m[7] == 0.0f && m[11] == 0.0f && m[15] == 1.0f))
      return fail(RTPT_E_INVALID, "ubo.model must be affine (bottom row 0 0 0 1)");
    float det = m[0] * (m[5] * m[10] - m[9] * m[6]) - m[4] * (m[1] * m[10] - m[9] * m[2]) + m[8] * (m[1] * m[6] - m[5] * m[2]);
    if (!(det != 0.0f) || det != det) return fail(RTPT_E_INVALID, "ubo.model is singular");
  }
  HIP_TRY(hipSetDevice(c->device));
  if (std::memcmp(ubo->model, c->model, sizeof c->model) != 0) {
    int rcm = apply_model(c, ubo->model);
    if (rcm) return rcm;
  }
  // The LUT is a function of the posed scene: the geometry stage's per-frame rewrite (visibility.geom.glsl:57-59)
  // produces the same bytes every frame while the model rests, so only a buffer that does not hold the current
  // pose yet is rebuilt (after rtpt_scene_upload / a model change / rtpt_set_plane).  The device triangles are
  // already posed, so the kernel's own model is the identity.
  if (c->lut_version[c->lut_cur] != c->model_version || !c->tables_valid) {
    Timer tm(c, RTPT_K_LUT);
    rt::LutArgs la;
    la.n_tris = c->n_tris;
    la.shade = static_cast<const float4*>(c->shade.ptr);
    for (int i = 0; i < 16; i++) la.model[i] = (i % 5 == 0) ? 1.0f : 0.0f;
    la.lut = static_cast<float4*>(c->lut[c->lut_cur].ptr);
    la.normal_tab = static_cast<float4*>(c->normal_tab.ptr);
    la.area_tab = la.normal_tab + (c->n_tris + 1);
    la.pair_tab = static_cast<float*>(c->pair_tab.ptr);
    la.sigma_n = c->cfg.sigma_n;
    rt::launch_lut(la, c->stream);
    c->lut_version[c->lut_cur] = c->model_version;
    c->tables_valid = true;
  }
  if ((rc = launch_check("lut"))) return rc;
  if (!c->lut_prev_valid) {
    // D3: visibilityLUTprevious is read during frame 0 before anything wrote it; define it as LUT
    HIP_TRY(hipMemcpyAsync(c->lut[c->lut_cur ^ 1].ptr, c->lut[c->lut_cur].ptr, c->lut[c->lut_cur].bytes, hipMemcpyDeviceToDevice,
                           c->stream));
    c->lut_prev_valid = true;
    c->lut_version[c->lut_cur ^ 1] = c->model_version;
  }
  rt::GbufferArgs a;
  a.g = geom(c, y0, y1);
  if ((rc = ensure_stack_spill(c, frame_blocks(c)))) return rc;
  a.scene = scene_view(c);
  const float* V = ubo->view;
  rt::f3 tcol{V[12], V[13], V[14]};
  rt::f3 c0{V[0], V[1], V[2]}, c1{V[4], V[5], V[6]}, c2{V[8], V[9], V[10]};
  a.org[0] = -rt::exact::dot(c0, tcol);
  a.org[1] = -rt::exact::dot(c1, tcol);
  a.org[2] = -rt::exact::dot(c2, tcol);
  a.c0[0] = c0.x; a.c0[1] = c0.y; a.c0[2] = c0.z;
  a.c1[0] = c1.x; a.c1[1] = c1.y; a.c1[2] = c1.z;
  a.c2[0] = c2.x; a.c2[1] = c2.y; a.c2[2] = c2.z;
  a.p00 = ubo->proj[0];
  a.p11 = ubo->proj[5];
  // per-column / per-row view-space ray directions (kernels.hip k_ray_tables): rebuilt when the projection or the frame
  // size they were built for changes (the reference's projection is constant after start-up, main.cpp:1471)
  if (!c->ray_tab.ptr || c->ray_tab_p00 != a.p00 || c->ray_tab_p11 != a.p11 || c->ray_tab_w != c->cfg.width || c->ray_tab_h != c->cfg.height) {
    int rct = alloc_buf(c->ray_tab, (static_cast<size_t>(c->cfg.width) + c->cfg.height) * sizeof(float));
    if (rct) return rct;
    rt::launch_ray_tables(static_cast<int>(c->cfg.width), static_cast<int>(c->cfg.height), a.p00, a.p11, static_cast<float*>(c->ray_tab.ptr),
                          static_cast<float*>(c->ray_tab.ptr) + c->cfg.width, c->stream);
    c->ray_tab_p00 = a.p00;
    c->ray_tab_p11 = a.p11;
    c->ray_tab_w = c->cfg.width;
    c->ray_tab_h = c->cfg.height;
  }
  a.dvx = static_cast<const float*>(c->ray_tab.ptr);
  a.dvy = a.dvx + c->cfg.width;
  rt::exact::mat_mul(ubo->proj, ubo->view, a.PV);
  a.tmax = c->cfg.ray_tmax;
  {
    const double org[3] = {a.org[0], a.org[1], a.org[2]};
    const double d0[3] = {c0.x, c0.y, c0.z}, d1[3] = {c1.x, c1.y, c1.z}, d2[3] = {c2.x, c2.y, c2.z};
    // view-space axis i of a world vector r is dot(row i of R, r); the columns c0,c1,c2 of the view
    // matrix's rotation hold R^T's rows, i.e. x_view = (c0.x, c1.x, c2.x) . r
    const double rx[3] = {d0[0], d1[0], d2[0]}, ry[3] = {d0[1], d1[1], d2[1]}, rz[3] = {d0[2], d1[2], d2[2]};
    a.cull = (!c->use_bvh && c->width_fits_i16() && screen_bounds(c, org, rx, ry, rz, a.p00, a.p11, 0.0, a.bounds)) ? 1 : 0;
  }
  a.vis = static_cast<uint32_t*>(c->vis[c->vis_cur].ptr);
  a.worldpos = static_cast<float4*>(c->worldpos.ptr);
  a.depth = static_cast<float*>(c->depth.ptr);
  a.normals = nullptr;
  a.normal_tab = static_cast<const float4*>(c->normal_tab.ptr);
  a.area_tab = a.normal_tab + (c->n_tris + 1);
  if (!c->pair_tab.ptr) {  // more than 63 triangles: the filter stages per-pixel normals instead of ids
    if (!c->normals.ptr) {
      int rc2 = alloc_buf(c->normals, c->pixels() * 16);
      if (rc2) return rc2;
      c->normals_y0 = c->normals_y1 = 0;
    }
    a.normals = static_cast<float4*>(c->normals.ptr);
    // rows written so far this frame (strips call the pass once per range; a new frame starts a new range)
    if (c->normals_y1 == static_cast<int>(y0) && c->normals_frame == c->frames_ended)
      c->normals_y1 = static_cast<int>(y1);
    else {
      c->normals_y0 = static_cast<int>(y0);
      c->normals_y1 = static_cast<int>(y1);
    }
    c->normals_frame = c->frames_ended;
  }
  a.grad_on = 0;
  a.grad_y0 = a.grad_y1 = 0;
  a.lut = a.lut_prev = nullptr;
  a.grad = nullptr;
  for (int i = 0; i < 3; i++) a.g_cam[i] = a.g_light[i] = a.g_light_prev[i] = a.g_color[i] = a.g_color_prev[i] = 0.0f;
  if (!(c->cfg.flags & RTPT_FLAG_NO_FILTER_FUSION)) {
    // recorded: rtpt_temporal_gradient normally follows at once (main.cpp:1105-1106) and the two run as one launch;
    // any other entry point launches it first
    c->pending_gb = a;
    c->pending_gb_valid = true;
    return RTPT_OK;
  }
  {
    Timer tm(c, RTPT_K_GBUFFER);
    rt::launch_gbuffer(a, c->stream);
  }
  return launch_check("gbuffer");
}

}  // extern "C"

namespace rtpt_impl {
int gbuffer_flush(rtpt_ctx* c) {
  if (!c->pending_gb_valid) return RTPT_OK;
  c->pending_gb_valid = false;
  hipError_t e = hipSetDevice(c->device);
  if (e != hipSuccess) return fail(RTPT_E_DEVICE, std::string("hipSetDevice: ") + hipGetErrorString(e));
  {
    Timer tm(c, c->pending_gb.grad_on ? RTPT_K_GBUFFER_GRADIENT : RTPT_K_GBUFFER);
    rt::launch_gbuffer(c->pending_gb, c->stream);
  }
  return launch_check(c->pending_gb.grad_on ? "gbuffer + temporal_gradient" : "gbuffer");
}
}  // namespace rtpt_impl

extern "C" {

// ------------------------------------------------------------------------------------------ K1
int rtpt_temporal_gradient(rtpt_ctx* c, const rtpt_push_constants* pc, uint32_t y0, uint32_t y1) {
  if (!c || !pc) return fail(RTPT_E_INVALID, "NULL argument");
  if (!c->n_tris) return fail(RTPT_E_NO_SCENE, "rtpt_scene_upload has not been called");
  int rc = check_rows(c, y0, y1);
  if (rc) return rc;
  HIP_TRY(hipSetDevice(c->device));
  if (c->pending_gb_valid && static_cast<int32_t>(y0) >= c->pending_gb.g.y0 && static_cast<int32_t>(y1) <= c->pending_gb.g.y1) {
    // K0 + K1 in one launch: K1's inputs (id, world position) are K0's outputs for the same pixel
    rt::GbufferArgs& g = c->pending_gb;
    g.grad_on = 1;
    g.grad_y0 = static_cast<int32_t>(y0);
    g.grad_y1 = static_cast<int32_t>(y1);
    for (int i = 0; i < 3; i++) {
      g.g_cam[i] = pc->cameraPos[i];
      g.g_light[i] = pc->lightPos[i];
      g.g_light_prev[i] = pc->lightPosPrev[i];
      g.g_color[i] = pc->currentCameraColor[i];
      g.g_color_prev[i] = pc->previousCameraColor[i];
    }
    g.lut = static_cast<const float4*>(c->lut[c->lut_cur].ptr);
    g.lut_prev = static_cast<const float4*>(c->lut[c->lut_cur ^ 1].ptr);
    g.grad = static_cast<float4*>(c->gradient.ptr);
    int rcq = filter_flush(c, false);
    if (rcq) return rcq;
    // stays recorded: rtpt_raytrace normally follows at once (main.cpp:1107) and takes both passes into its launch; any other
    // entry point launches them first (FLUSH_FILTER)
    if (c->fuse_trace) return RTPT_OK;
    return gbuffer_flush(c);
  }
  FLUSH_FILTER(c);
  rt::GradientArgs a;
  a.g = geom(c, y0, y1);
  for (int i = 0; i < 3; i++) {
    a.cam[i] = pc->cameraPos[i];
    a.light[i] = pc->lightPos[i];
    a.light_prev[i] = pc->lightPosPrev[i];
    a.color[i] = pc->currentCameraColor[i];
    a.color_prev[i] = pc->previousCameraColor[i];
  }
  a.vis = static_cast<const uint32_t*>(c->vis[c->vis_cur].ptr);
  a.worldpos = static_cast<const float4*>(c->worldpos.ptr);
  a.lut = static_cast<const float4*>(c->lut[c->lut_cur].ptr);
  a.lut_prev = static_cast<const float4*>(c->lut[c->lut_cur ^ 1].ptr);
  a.normal_tab = static_cast<const float4*>(c->normal_tab.ptr);
  a.area_tab = a.normal_tab + (c->n_tris + 1);
  a.grad = static_cast<float4*>(c->gradient.ptr);
  {
    Timer tm(c, RTPT_K_GRADIENT);
    rt::launch_gradient(a, c->stream);
  }
  return launch_check("temporal_gradient");
}

// ------------------------------------------------------------------------------------------ K2
int rtpt_raytrace(rtpt_ctx* c, const rtpt_push_constants* pc, uint32_t y0, uint32_t y1) {
  if (!c || !pc) return fail(RTPT_E_INVALID, "NULL argument");
  if (!c->n_tris) return fail(RTPT_E_NO_SCENE, "rtpt_scene_upload has not been called");
  int rc = check_rows(c, y0, y1);
  if (rc) return rc;
  HIP_TRY(hipSetDevice(c->device));
  if ((rc = filter_flush(c, false))) return rc;  // a recorded K0 (+ K1) stays recorded: it may join this launch (below)
  rt::PathtraceArgs a;
  a.g = geom(c, y0, y1);
  a.scene = scene_view(c);
  a.frame = pc->frameNumber;
  a.batch = pc->sample_batch;
  a.max_segments = c->cfg.max_segments;
  a.spp = c->cfg.samples_per_pixel;
  for (int i = 0; i < 3; i++) {
    a.cam[i] = pc->cameraPos[i];
    a.light_c[i] = pc->lightPos[i];                                          // raytrace.comp.glsl:279
    a.light_col[i] = pc->currentCameraColor[i] * c->cfg.light_intensity;     // :281
    a.light_col_first[i] = a.light_col[i] / c->cfg.first_hit_light_divisor;  // :229
  }
  a.light_r2 = c->cfg.light_radius * c->cfg.light_radius;  // :173
  a.slope = c->cfg.fov_slope;
  a.jitter = c->cfg.pixel_jitter;
  a.ray_offset = c->cfg.ray_offset;
  a.tmax = c->cfg.ray_tmax;
  a.image = static_cast<float4*>(c->color[c->color_of_role[ROLE_IMAGE]].ptr);
  a.depth = static_cast<const float*>(c->depth.ptr);
  c->alpha_depth[c->color_of_role[ROLE_IMAGE]] = true;
  a.hit_id = (c->debug_mask & RTPT_DEBUG_HIT_ID) ? static_cast<uint32_t*>(c->hit_id.ptr) : nullptr;
  a.raycount = static_cast<unsigned long long*>(c->raycount.ptr);
  a.count_y0 = c->count_y0;
  a.count_y1 = c->count_y1;
  a.compact = (c->cfg.flags & RTPT_FLAG_NO_PATH_COMPACTION) ? 0 : 1;
  a.n_cu = c->n_cu;
  a.queue[0] = a.queue[1] = nullptr;
  a.queue_count = nullptr;
  a.queue_region = 0;
  a.pool_slab = nullptr;
  if (c->trace_pool && c->use_bvh && c->leaf_pairs && a.compact && a.spp == 1) {
    const size_t need = rt::pathtrace_pool_bytes(static_cast<int>(c->cfg.width), static_cast<int>(c->rows()));  // 0: not built in
    if (need && c->path_pool.bytes < need && (rc = alloc_buf(c->path_pool, need))) return rc;
    a.pool_slab = need ? c->path_pool.ptr : nullptr;
  }
  const uint32_t window = c->trace_window ? c->trace_window : rt::pt_first_window(c->use_bvh);
  a.first_window = window;
  if (a.compact && a.spp == 1 && a.max_segments > window && !(c->cfg.flags & RTPT_FLAG_SINGLE_LAUNCH_PATHS)) {
    // a region holds the survivors of ceil(workgroups / kPathQueues) workgroups of 256 paths (kernels.hip); the
    // second buffer is only needed when a third segment window exists
    const size_t blocks = ((static_cast<size_t>(c->cfg.width) + 63) / 64) * ((c->rows() + 3) / 4);
    const size_t region = ((blocks + rt::kPathQueues - 1) / rt::kPathQueues) * 256;
    const size_t cap = region * rt::kPathQueues;
    if (!c->path_queue_count.ptr && (rc = alloc_buf(c->path_queue_count, 2 * rt::kPathQueues * sizeof(uint32_t)))) return rc;
    if (!c->path_queue[0].ptr && (rc = alloc_buf(c->path_queue[0], cap * 48))) return rc;
    if (a.max_segments > 2u * window && !c->path_queue[1].ptr && (rc = alloc_buf(c->path_queue[1], cap * 48))) return rc;
    a.queue[0] = c->path_queue[0].ptr;
    a.queue[1] = c->path_queue[1].ptr;
    a.queue_count = static_cast<uint32_t*>(c->path_queue_count.ptr);
    a.queue_region = static_cast<uint32_t>(region);
  }
  a.cull = 0;
  if (!c->use_bvh && c->width_fits_i16()) {
    // K2 camera (raytrace.comp.glsl:314-320): at cameraPos, looking down -z, d = (slope*ux, slope*uy, -1) with
    // ux = (2cx - W)/H, uy = -(2cy - H)/H.  The Gaussian jitter is 0.375 * sqrt(-2 ln u1) <= 0.375 * 13.3 px
    // (u1 >= 1e-38, :87).
    const double org[3] = {pc->cameraPos[0], pc->cameraPos[1], pc->cameraPos[2]};
    const double ex[3] = {1, 0, 0}, ey[3] = {0, 1, 0}, ez[3] = {0, 0, 1};
    const double slope = c->cfg.fov_slope, W = c->cfg.width, H = c->cfg.height;
    if (slope > 0)
      a.cull = screen_bounds(c, org, ex, ey, ez, H / (W * slope), -1.0 / slope, std::fabs(c->cfg.pixel_jitter) * 13.3, a.bounds) ? 1 : 0;
  }
  c->final_swapped = false;
  c->image_alias = false;
  // K0 (+ K1) recorded right before this call run inside this launch, behind the tracing tiles (kernels.hip: k_gbuffer_pathtrace)
  const bool fused = c->pending_gb_valid && c->fuse_trace && rt::pathtrace_fuses_gbuffer(a, c->pending_gb);
  if (!fused && (rc = gbuffer_flush(c))) return rc;
  if ((rc = ensure_stack_spill(c, std::max<size_t>(frame_blocks(c), rt::pathtrace_grid_blocks(a, fused ? &c->pending_gb : nullptr))))) return rc;
  a.scene = scene_view(c);
  if (fused) {
    c->pending_gb.scene = a.scene;  // the spill area may have moved since the G-buffer call was recorded
    c->pending_gb_valid = false;
  }
  {
    Timer tm(c, fused ? RTPT_K_GBUFFER_PATHTRACE : RTPT_K_PATHTRACE);
    rt::launch_pathtrace(a, fused ? &c->pending_gb : nullptr, c->stream);
  }
  return launch_check(fused ? "gbuffer + temporal_gradient + raytrace" : "raytrace");
}

// ------------------------------------------------------------------------------------------ K3
// rtpt_temporal_filter keeps the reference's shape — one call per iteration of applyTemporalFiltering's loop
// (main.cpp:1259-1305) — but the calls of a frame are RECORDED and launched when the last iteration arrives, the way the
// reference records its dispatches into command buffers: consecutive iterations then run as one chained launch
// (atrous_chain.hip) whose intermediate image never leaves LDS.  Any call that observes or changes what an iteration
// reads or writes (readback, plane pointers, sync, another pass, ...) first runs the recorded iterations one by one, so
// between iterations every plane holds exactly what the separate dispatches would have left there.
}  // extern "C"

namespace {

int filter_validate(rtpt_ctx* c, const rtpt_push_constants* pc, const rtpt_ubo* ubo, uint32_t& y0, uint32_t& y1) {
  int rc = check_rows(c, y0, y1);
  if (rc) return rc;
  const int k = pc->waveletIteration, max_it = pc->maxWaveletIteration;
  if (k < 1 || max_it < 1 || k > max_it) return fail(RTPT_E_INVALID, "need 1 <= waveletIteration <= maxWaveletIteration");
  const uint32_t ext = c->cfg.flags & rt::kExtMask;
  if ((ext & rt::kExtPow2Stride) && k > 24) return fail(RTPT_E_INVALID, "RTPT_FLAG_EXT_POW2_STRIDE supports at most 24 iterations");
  const int stride = (ext & rt::kExtPow2Stride) ? (1 << (k - 1)) : k;
  const int64_t reach = static_cast<int64_t>(stride) * ((ext & rt::kExtGauss5) ? 2 : 1);
  // taps reach rows y +- reach (clamped to the frame, temporalFiltering.comp.glsl:135-136): they must be stored here
  const int64_t lo = std::max<int64_t>(0, static_cast<int64_t>(y0) - reach);
  const int64_t hi = std::min<int64_t>(c->cfg.height, static_cast<int64_t>(y1) + reach);
  if (y1 > y0 && (lo < c->cfg.row_begin || hi > c->cfg.row_end))
    return fail(RTPT_E_INVALID, "filter taps reaching " + std::to_string(reach) + " rows leave the stored rows (missing halo)");
  const bool final_pass = (k == max_it) && (k & 1);
  if (final_pass && !ubo) return fail(RTPT_E_INVALID, "the final pass needs the UBO (viewPrev/projPrev)");
  if ((ext & rt::kExtVariance) && !ubo)
    return fail(RTPT_E_INVALID, "RTPT_FLAG_EXT_VARIANCE needs the UBO (viewPrev/projPrev) on every iteration");
  return RTPT_OK;
}

// launch iteration f.pc.waveletIteration — or, with levels > 1, that iteration and the levels - 1 after it as one chain
int filter_launch(rtpt_ctx* c, const FilterCall& f, int levels) {
  const rtpt_push_constants* pc = &f.pc;
  const rtpt_ubo* ubo = f.has_ubo ? &f.ubo : nullptr;
  const uint32_t y0 = f.y0, y1 = f.y1;  // rows of the LAST iteration of the chain
  const int k = pc->waveletIteration, max_it = pc->maxWaveletIteration;
  const int k_last = k + levels - 1;
  const uint32_t ext = c->cfg.flags & rt::kExtMask;
  const int stride = (ext & rt::kExtPow2Stride) ? (1 << (k - 1)) : k;
  const int64_t reach = static_cast<int64_t>(stride) * ((ext & rt::kExtGauss5) ? 2 : 1);
  // main.cpp:1264-1281: odd k reads `image`, writes `filteredImageBuffer`; even k the reverse.
  // An even final pass blends into a buffer nothing reads (main.cpp:55 "must be an odd number"),
  // so only an odd final pass is a FINAL launch.
  const bool final_pass = (k_last == max_it) && (k_last & 1);
  HIP_TRY(hipSetDevice(c->device));
  int in_role = (k & 1) ? ROLE_IMAGE : ROLE_FILTERED;
  int out_role = (k & 1) ? ROLE_FILTERED : ROLE_IMAGE;
  if (levels == 1 && final_pass && c->final_swapped) std::swap(in_role, out_role);  // a second row range of the same final pass
  // a chain reads the first iteration's input and writes the OTHER buffer, whatever the parity of its length; the roles
  // are re-pointed below so that afterwards every role names the buffer the separate passes would have left it in
  const int in_buf = c->color_of_role[in_role], out_buf = c->color_of_role[out_role];
  rt::AtrousArgs a;
  std::memset(&a, 0, sizeof a);
  a.g = geom(c, y0, y1);
  a.k = k;
  a.stride = stride;
  a.ext = ext;
  a.exact = (c->cfg.flags & RTPT_FLAG_EXACT_FILTER) ? 1 : 0;
  a.direct = (c->cfg.flags & RTPT_FLAG_DIRECT_FILTER) ? 1 : 0;
  a.n_tris = c->n_tris;
  a.pair_tab = static_cast<const float*>(c->pair_tab.ptr);
  a.rows_stored = static_cast<int32_t>(c->rows());
  a.n_cu = c->n_cu;
  // the last iteration of an even N writes `image` and nothing filters it again: alpha 0 like the reference's
  // vec4(rgb, 0) (temporalFiltering.comp.glsl:152), so a device-side consumer of IMAGE never sees the depth
  a.alpha_zero = (k_last == max_it && !final_pass) ? 1 : 0;
  a.sigma_n = c->cfg.sigma_n;
  a.sigma_z = c->cfg.sigma_z;
  a.sigma_l = c->cfg.sigma_l;
  a.in = static_cast<const float4*>(c->color[in_buf].ptr);
  a.out = static_cast<float4*>(c->color[out_buf].ptr);
  a.vis = static_cast<const uint32_t*>(c->vis[c->vis_cur].ptr);
  a.normal_tab = static_cast<const float4*>(c->normal_tab.ptr);
  {
    const int64_t lo = std::max<int64_t>(0, static_cast<int64_t>(y0) - reach), hi = std::min<int64_t>(c->cfg.height, static_cast<int64_t>(y1) + reach);
    const bool covered = c->normals.ptr && c->normals_frame == c->frames_ended && c->normals_y0 <= lo && c->normals_y1 >= hi;
    a.normals = covered ? static_cast<const float4*>(c->normals.ptr) : nullptr;
  }
  if (!c->alpha_depth[in_buf]) {
    // the input plane was injected (rtpt_set_plane / rtpt_bind_plane): give it its depth channel
    rt::launch_stamp_depth(geom(c, c->cfg.row_begin, c->cfg.row_end), static_cast<float4*>(c->color[in_buf].ptr),
                           static_cast<const float*>(c->depth.ptr), c->stream);
    c->alpha_depth[in_buf] = true;
  }
  c->alpha_depth[out_buf] = !final_pass && !a.alpha_zero;
  if (final_pass) {
    a.frame = pc->frameNumber;
    a.alpha = c->cfg.alpha;
    a.worldpos = static_cast<const float4*>(c->worldpos.ptr);
    a.history = static_cast<const float4*>(c->color[c->color_of_role[ROLE_PREVIOUS]].ptr);
    a.lut_prev = static_cast<const float4*>(c->lut[c->lut_cur ^ 1].ptr);
    rt::exact::mat_mul(ubo->projPrev, ubo->viewPrev, a.PVprev);  // temporalFiltering.comp.glsl:180
    a.prev_pixel = (c->debug_mask & RTPT_DEBUG_PREV_PIXEL) ? static_cast<int2*>(c->prev_pixel.ptr) : nullptr;
    a.hist_row_base = static_cast<int32_t>(c->cfg.row_begin);
    a.hist_y0 = c->hist_y0;
    a.hist_y1 = c->hist_y1;
    a.gradient = static_cast<const float4*>(c->gradient.ptr);
    a.prev_vis = static_cast<const uint32_t*>(c->vis[c->vis_cur ^ 1].ptr);
    a.pvis_y0 = c->guides_y0;
    a.pvis_y1 = c->guides_y1;
    a.pvis_row_base = static_cast<int32_t>(c->cfg.row_begin);
    if (c->ext_prev_vis) {  // gathered across strips
      a.prev_vis = static_cast<const uint32_t*>(c->ext_prev_vis);
      a.pvis_y0 = c->ext_guides_y0;
      a.pvis_y1 = c->ext_guides_y1;
      a.pvis_row_base = c->ext_guides_y0;
    }
    if (c->ext_history) {  // all-gathered previous frame (multi-GPU strips)
      a.history = static_cast<const float4*>(c->ext_history);
      a.hist_row_base = c->ext_hist_y0;
      a.hist_y0 = c->ext_hist_y0;
      a.hist_y1 = c->ext_hist_y1;
    }
  }
  if (ext & rt::kExtVariance) {
    if (k == 1) {  // temporal accumulation of the luminance moments of the traced image (this iteration's input)
      rt::MomentsArgs m;
      std::memset(&m, 0, sizeof m);
      m.g = geom(c, c->cfg.row_begin, c->cfg.row_end);
      m.frame = pc->frameNumber;
      m.alpha = c->cfg.alpha;
      m.traced = a.in;
      m.vis = a.vis;
      m.worldpos = static_cast<const float4*>(c->worldpos.ptr);
      m.lut_prev = static_cast<const float4*>(c->lut[c->lut_cur ^ 1].ptr);
      rt::exact::mat_mul(ubo->projPrev, ubo->viewPrev, m.PVprev);
      m.prev_vis = static_cast<const uint32_t*>(c->vis[c->vis_cur ^ 1].ptr);
      m.moments_prev = static_cast<const float4*>(c->moments[c->moments_cur ^ 1].ptr);
      m.hist_row_base = static_cast<int32_t>(c->cfg.row_begin);
      m.hist_y0 = c->guides_y0;
      m.hist_y1 = c->guides_y1;
      if (c->ext_prev_vis && c->ext_moments) {  // gathered across strips (rtpt_set_external_guides)
        m.prev_vis = static_cast<const uint32_t*>(c->ext_prev_vis);
        m.moments_prev = static_cast<const float4*>(c->ext_moments);
        m.hist_row_base = m.hist_y0 = c->ext_guides_y0;
        m.hist_y1 = c->ext_guides_y1;
      }
      m.svgf = (ext & rt::kExtSvgfVariance) ? 1 : 0;
      m.rows_stored = static_cast<int32_t>(c->rows());
      m.moments_out = static_cast<float4*>(c->moments[c->moments_cur].ptr);
      m.var_out = static_cast<float*>(c->variance[0].ptr);
      rt::launch_moments(m, c->stream);
      c->variance_last = 0;
    }
    a.var_in = static_cast<const float*>(c->variance[c->variance_last].ptr);
    a.var_out = static_cast<float*>(c->variance[c->variance_last ^ 1].ptr);
    c->variance_last ^= 1;
    if ((ext & rt::kExtSvgfVariance) && c->var_scale.ptr) {  // SVGF's variance prefilter: the centre's scale only
      rt::launch_var_prefilter(geom(c, y0, y1), static_cast<int>(c->rows()), a.var_in, static_cast<float*>(c->var_scale.ptr), c->stream);
      a.var_scale = static_cast<const float*>(c->var_scale.ptr);
    }
  }
  if (final_pass) c->present_fused_dst = nullptr;  // a new frame's final pass: the previous frame's blit is history
  if (final_pass && levels == 1 && c->present_dst && static_cast<int>(y0) <= c->present_y0 && static_cast<int>(y1) >= c->present_y1 &&
      rt::atrous_final_fuses_present(a)) {
    a.present = static_cast<uint32_t*>(c->present_dst);
    a.present_y0 = c->present_y0;
    a.present_y1 = c->present_y1;
    c->present_fused_dst = c->present_dst;
    c->present_fused_y0 = c->present_y0;
    c->present_fused_y1 = c->present_y1;
  }
  {
    Timer tm(c, levels > 1 ? (final_pass ? RTPT_K_ATROUS_CHAIN_FINAL : RTPT_K_ATROUS_CHAIN) : (final_pass ? RTPT_K_ATROUS_FINAL : RTPT_K_ATROUS));
    if (levels > 1)
      rt::launch_atrous_chain(a, levels, final_pass, c->filter_policy, c->stream);
    else
      rt::launch_atrous(a, final_pass, c->stream);
  }
  int rc;
  if ((rc = launch_check("temporal_filter"))) return rc;
  if (levels > 1) {
    // point the roles at the buffers the separate passes would have left them in: the result sits in out_buf
    const int res_role = final_pass ? ROLE_IMAGE : ((k_last & 1) ? ROLE_FILTERED : ROLE_IMAGE);
    const int oth_role = res_role == ROLE_IMAGE ? ROLE_FILTERED : ROLE_IMAGE;
    c->color_of_role[res_role] = out_buf;
    c->color_of_role[oth_role] = in_buf;
    if (final_pass) {
      c->final_swapped = true;
      c->final_y0 = static_cast<int>(y0);
      c->final_y1 = static_cast<int>(y1);
    } else if (k_last == max_it) {
      c->final_y0 = static_cast<int>(y0);
      c->final_y1 = static_cast<int>(y1);
    }
    return RTPT_OK;
  }
  if (final_pass) {
    if (!c->final_swapped) {
      // D1: the blend went to a distinct buffer, which now becomes `image`
      std::swap(c->color_of_role[ROLE_IMAGE], c->color_of_role[ROLE_FILTERED]);
      c->final_swapped = true;
      c->final_y0 = static_cast<int>(y0);
      c->final_y1 = static_cast<int>(y1);
    } else {
      c->final_y0 = std::min(c->final_y0, static_cast<int>(y0));
      c->final_y1 = std::max(c->final_y1, static_cast<int>(y1));
    }
  } else if (k == max_it) {
    c->final_y0 = static_cast<int>(y0);
    c->final_y1 = static_cast<int>(y1);
  }
  return RTPT_OK;
}

}  // namespace

namespace rtpt_impl {
// run the recorded iterations.  fuse = false: one launch per iteration (an observer is about to look at the planes)
int filter_flush(rtpt_ctx* c, bool fuse) {
  if (c->pending.empty()) return RTPT_OK;
  std::vector<FilterCall> calls;
  calls.swap(c->pending);  // filter_launch may fail: the record is dropped either way
  const size_t n = calls.size();
  const int H = static_cast<int>(c->cfg.height);
  size_t i = 0;
  while (i < n) {
    int levels = 1;
    if (fuse && !(c->cfg.flags & (RTPT_FLAG_DIRECT_FILTER | RTPT_FLAG_NO_FILTER_FUSION)) && !(c->cfg.flags & rt::kExtMask) && c->pair_tab.ptr) {
      const int k0 = calls[i].pc.waveletIteration, max_it = calls[i].pc.maxWaveletIteration;
      // grow the chain while the next record is the next iteration, its rows are covered and the kernel has the LDS
      while (i + levels < n && levels < c->chain_max) {
        const FilterCall &cur = calls[i + levels - 1], &nxt = calls[i + levels];
        const int kn = nxt.pc.waveletIteration;
        if (kn != k0 + levels || nxt.pc.maxWaveletIteration != max_it) break;
        const bool nxt_final = (kn == max_it) && (kn & 1);
        if (nxt_final && !c->chain_final) break;
        if (nxt_final && c->final_swapped) break;
        const int need0 = std::max(0, static_cast<int>(nxt.y0) - kn), need1 = std::min(H, static_cast<int>(nxt.y1) + kn);
        if (nxt.y1 <= nxt.y0 || static_cast<int>(cur.y0) > need0 || static_cast<int>(cur.y1) < need1) break;
        if (static_cast<int64_t>(nxt.y1 - nxt.y0) * c->cfg.width < c->chain_min_pixels) break;
        if (!rt::atrous_chain_supported(k0, levels + 1, c->n_tris)) break;
        levels++;
        if (nxt_final) break;
      }
      // a chain must not end one short of a FINAL pass it could have included... nothing to do: greedy from the front
    }
    FilterCall f = calls[i];
    if (levels > 1) {
      const FilterCall& lastc = calls[i + levels - 1];
      f.y0 = lastc.y0;
      f.y1 = lastc.y1;
      f.has_ubo = lastc.has_ubo;
      f.ubo = lastc.ubo;
      f.pc.frameNumber = lastc.pc.frameNumber;
    }
    int rc = filter_launch(c, f, levels);
    if (rc) return rc;
    i += static_cast<size_t>(levels);
  }
  return RTPT_OK;
}

}  // namespace rtpt_impl

extern "C" {

int rtpt_temporal_filter(rtpt_ctx* c, const rtpt_push_constants* pc, const rtpt_ubo* ubo, uint32_t y0, uint32_t y1) {
  if (!c || !pc) return fail(RTPT_E_INVALID, "NULL argument");
  if (!c->n_tris) return fail(RTPT_E_NO_SCENE, "rtpt_scene_upload has not been called");
  int rc = filter_validate(c, pc, ubo, y0, y1);
  if (rc) return rc;
  FilterCall f;
  f.pc = *pc;
  f.has_ubo = ubo != nullptr;
  if (ubo) f.ubo = *ubo;
  f.y0 = y0;
  f.y1 = y1;
  const bool record = !(c->cfg.flags & (RTPT_FLAG_NO_FILTER_FUSION | RTPT_FLAG_DIRECT_FILTER)) && !(c->cfg.flags & rt::kExtMask) &&
                      c->pair_tab.ptr && c->chain_max > 1;
  if (!record) {
    FLUSH_FILTER(c);
    return filter_launch(c, f, 1);
  }
  // a recorded K0 (rtpt_gbuffer without rtpt_temporal_gradient behind it) goes out before the first filter record: the
  // filters read its id / depth planes and are launched from here on without looking at it again
  {
    int rcg = gbuffer_flush(c);
    if (rcg) return rcg;
  }
  // a record that does not continue the recorded run (same iteration twice, a restart) ends it
  if (!c->pending.empty() && (c->pending.back().pc.waveletIteration + 1 != pc->waveletIteration ||
                              c->pending.back().pc.maxWaveletIteration != pc->maxWaveletIteration))
    FLUSH_FILTER(c);
  c->pending.push_back(f);
  if (pc->waveletIteration == pc->maxWaveletIteration) return filter_flush(c, true);
  return RTPT_OK;
}

// ------------------------------------------------------------------------------------------ K4
int rtpt_end_frame(rtpt_ctx* c) {
  if (!c) return fail(RTPT_E_INVALID, "ctx is NULL");
  FLUSH_FILTER(c);
  // main.cpp:1364 image -> previousImage: rotate roles instead of blitting.  After the reference's
  // copy both images hold the same pixels; here IMAGE now names the old history buffer (about to be
  // overwritten by the next rtpt_raytrace), so until then rtpt_readback(IMAGE) is served from
  // PREVIOUS (image_alias).
  std::swap(c->color_of_role[ROLE_IMAGE], c->color_of_role[ROLE_PREVIOUS]);
  c->image_alias = true;
  c->hist_y0 = c->final_y0;
  c->hist_y1 = c->final_y1;
  // the id plane (and, with RTPT_FLAG_EXT_VARIANCE, the moment plane) of the frame just ended cover the stored rows
  c->guides_y0 = static_cast<int>(c->cfg.row_begin);
  c->guides_y1 = static_cast<int>(c->cfg.row_end);
  // main.cpp:1367 visibilityBuffer -> previousVisibilityBuffer; main.cpp:1372 LUT -> LUTprev
  c->vis_cur ^= 1;
  c->moments_cur ^= 1;
  c->lut_cur ^= 1;
  c->lut_prev_valid = c->n_tris != 0;
  c->final_swapped = false;
  c->frames_ended++;
  return RTPT_OK;
}

// the swapchain image rows the next final filter pass should also write (fused blit); NULL clears the registration
int rtpt_present_target(rtpt_ctx* c, void* dst_device, uint32_t y0, uint32_t y1) {
  if (!c) return fail(RTPT_E_INVALID, "ctx is NULL");
  if (!dst_device) {
    c->present_dst = nullptr;
    return RTPT_OK;
  }
  if (reinterpret_cast<uintptr_t>(dst_device) & 3u) return fail(RTPT_E_INVALID, "swapchain image must be 4-byte aligned");
  FLUSH_FILTER(c);  // recorded iterations were recorded without it: they go out as they are
  int rc = check_rows(c, y0, y1);
  if (rc) return rc;
  c->present_dst = dst_device;
  c->present_y0 = static_cast<int>(y0);
  c->present_y1 = static_cast<int>(y1);
  return RTPT_OK;
}

// main.cpp:1338-1361: the blit of `image` to the swapchain image
int rtpt_present(rtpt_ctx* c, void* dst_device, uint32_t y0, uint32_t y1) {
  if (!c || !dst_device) return fail(RTPT_E_INVALID, "NULL argument");
  if (reinterpret_cast<uintptr_t>(dst_device) & 3u) return fail(RTPT_E_INVALID, "swapchain image must be 4-byte aligned");
  FLUSH_FILTER(c);
  int rc = check_rows(c, y0, y1);
  if (rc) return rc;
  // already there: the frame's final pass wrote these rows of this image in swapchain format (rtpt_present_target)
  if (c->present_fused_dst && static_cast<int>(y0) >= c->present_fused_y0 && static_cast<int>(y1) <= c->present_fused_y1 &&
      static_cast<char*>(dst_device) == static_cast<char*>(c->present_fused_dst) + static_cast<size_t>(static_cast<int>(y0) - c->present_fused_y0) * c->cfg.width * 4)
    return RTPT_OK;
  // the finished frame: IMAGE until rtpt_end_frame, PREVIOUS after it (the reference blits before it copies, the pixels
  // are the same); only rows the last final pass wrote hold it
  Buf* b = plane_buf(c, c->image_alias ? RTPT_PLANE_PREVIOUS : RTPT_PLANE_IMAGE);
  if (!b || !b->ptr) return fail(RTPT_E_INVALID, "no image plane");
  const int f0 = c->image_alias ? c->hist_y0 : c->final_y0, f1 = c->image_alias ? c->hist_y1 : c->final_y1;
  if (static_cast<int>(y0) < f0 || static_cast<int>(y1) > f1)
    return fail(RTPT_E_INVALID, "rtpt_present: rows [" + std::to_string(y0) + "," + std::to_string(y1) + ") outside the rows of the finished frame [" +
                                    std::to_string(f0) + "," + std::to_string(f1) + ")");
  HIP_TRY(hipSetDevice(c->device));
  {
    Timer tm(c, RTPT_K_PRESENT);
    rt::launch_present(geom(c, y0, y1), static_cast<const float4*>(b->ptr), static_cast<uint32_t*>(dst_device), c->stream);
  }
  return launch_check("present");
}

}  // extern "C"
