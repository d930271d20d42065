// api_context.hip — the context of the C ABI (include/rtpt.h): creation, planes, streams, copies, counters, timing, and the
// helpers shared by the other api_*.hip units (api_internal.hpp).
#include "api_internal.hpp"

namespace rtpt_impl {

thread_local std::string g_err;

int fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}


// workgroups of the largest per-frame launch that traverses the BVH: the 64 x 4-pixel tiles of the stored rows
// (k_gbuffer, the tile kernel) or the persistent queue kernel's grid
size_t frame_blocks(const rtpt_ctx* c) {
  const size_t tiles = ((static_cast<size_t>(c->cfg.width) + 63) / 64) * ((static_cast<size_t>(c->cfg.row_end - c->cfg.row_begin) + 3) / 4);
  return std::max(tiles, static_cast<size_t>(c->n_cu > 0 ? c->n_cu : 256) * 8);
}
// the spill area of the BVH traversal stack holds (stack depth - LDS entries) x workgroups x 256 entries: grown (never
// shrunk) before a launch whose grid is larger than any before it
int ensure_stack_spill(rtpt_ctx* c, size_t blocks) {
  if (!c->use_bvh) return RTPT_OK;
  const size_t depth = static_cast<size_t>(c->bvh_depth + 2 < 8 ? 8 : c->bvh_depth + 2);
  const size_t lds = std::min<size_t>(depth, static_cast<size_t>(c->bvh_stack_lds));
  if (depth <= lds || blocks <= c->stack_spill_blocks) return RTPT_OK;
  if (c->stack_spill.ptr) {
    hipError_t e = hipStreamSynchronize(c->stream);  // launches that spill into the old area
    if (e != hipSuccess) return fail(RTPT_E_DEVICE, std::string("hipStreamSynchronize: ") + hipGetErrorString(e));
    free_buf(c->stack_spill);
  }
  c->stack_spill_blocks = 0;
  int rc = alloc_buf(c->stack_spill, (depth - lds) * blocks * 256 * sizeof(uint32_t));
  if (rc) return rc;
  c->stack_spill_blocks = blocks;
  return RTPT_OK;
}

int alloc_buf(Buf& b, size_t bytes) {
  if (b.owned && b.ptr) (void)hipFree(b.ptr);
  b = Buf{};
  if (bytes == 0) return RTPT_OK;
  void* p = nullptr;
  hipError_t e = hipMalloc(&p, bytes);
  if (e != hipSuccess) return fail(RTPT_E_NOMEM, std::string("hipMalloc: ") + hipGetErrorString(e));
  b.ptr = p;
  b.bytes = bytes;
  b.owned = true;
  return RTPT_OK;
}

void free_buf(Buf& b) {
  if (b.owned && b.ptr) (void)hipFree(b.ptr);
  b = Buf{};
}

Buf* plane_buf(rtpt_ctx* c, rtpt_plane which) {
  switch (which) {
    case RTPT_PLANE_IMAGE: return &c->color[c->color_of_role[ROLE_IMAGE]];
    case RTPT_PLANE_FILTERED: return &c->color[c->color_of_role[ROLE_FILTERED]];
    case RTPT_PLANE_PREVIOUS: return &c->color[c->color_of_role[ROLE_PREVIOUS]];
    case RTPT_PLANE_WORLDPOS: return &c->worldpos;
    case RTPT_PLANE_GRADIENT: return &c->gradient;
    case RTPT_PLANE_DEPTH: return &c->depth;
    case RTPT_PLANE_VIS_ID: return &c->vis[c->vis_cur];
    case RTPT_PLANE_PREV_VIS_ID: return &c->vis[c->vis_cur ^ 1];
    case RTPT_PLANE_LUT: return &c->lut[c->lut_cur];
    case RTPT_PLANE_LUT_PREV: return &c->lut[c->lut_cur ^ 1];
    case RTPT_PLANE_PREV_PIXEL: return &c->prev_pixel;
    case RTPT_PLANE_RAYCOUNT: return &c->raycount;
    case RTPT_PLANE_HIT_ID: return &c->hit_id;
    case RTPT_PLANE_MOMENTS: return &c->moments[c->moments_cur];
    case RTPT_PLANE_MOMENTS_PREV: return &c->moments[c->moments_cur ^ 1];
    case RTPT_PLANE_VARIANCE: return &c->variance[c->variance_last];
    default: return nullptr;
  }
}

size_t plane_size(const rtpt_ctx* c, rtpt_plane which) {
  const size_t px = c->pixels();
  switch (which) {
    case RTPT_PLANE_IMAGE:
    case RTPT_PLANE_FILTERED:
    case RTPT_PLANE_PREVIOUS:
    case RTPT_PLANE_WORLDPOS:
    case RTPT_PLANE_GRADIENT: return px * 16;
    case RTPT_PLANE_DEPTH:
    case RTPT_PLANE_VIS_ID:
    case RTPT_PLANE_PREV_VIS_ID:
    case RTPT_PLANE_HIT_ID:
    case RTPT_PLANE_VARIANCE: return px * 4;
    case RTPT_PLANE_MOMENTS:
    case RTPT_PLANE_MOMENTS_PREV: return px * 16;
    case RTPT_PLANE_PREV_PIXEL: return px * 8;
    case RTPT_PLANE_LUT:
    case RTPT_PLANE_LUT_PREV: return (static_cast<size_t>(c->n_tris) + 1) * sizeof(rtpt_visibility_data);
    case RTPT_PLANE_RAYCOUNT: return 8;
    default: return 0;
  }
}


int check_rows(const rtpt_ctx* c, uint32_t& y0, uint32_t& y1) {
  if (y0 == 0 && y1 == 0) {
    y0 = c->cfg.row_begin;
    y1 = c->cfg.row_end;
  }
  if (y0 > y1 || y0 < c->cfg.row_begin || y1 > c->cfg.row_end)
    return fail(RTPT_E_INVALID, "row range [" + std::to_string(y0) + "," + std::to_string(y1) + ") outside stored rows [" +
                                    std::to_string(c->cfg.row_begin) + "," + std::to_string(c->cfg.row_end) + ")");
  return RTPT_OK;
}

rt::FrameGeom geom(const rtpt_ctx* c, uint32_t y0, uint32_t y1) {
  rt::FrameGeom g;
  g.W = static_cast<int32_t>(c->cfg.width);
  g.H = static_cast<int32_t>(c->cfg.height);
  g.row_base = static_cast<int32_t>(c->cfg.row_begin);
  g.y0 = static_cast<int32_t>(y0);
  g.y1 = static_cast<int32_t>(y1);
  return g;
}

rt::SceneView scene_view(const rtpt_ctx* c) {
  rt::SceneView s;
  s.isect_id = static_cast<const float4*>(c->isect_id.ptr);
  s.isect_leaf = static_cast<const float4*>(c->isect_leaf.ptr);
  s.leaf_ids = static_cast<const uint32_t*>(c->leaf_order.ptr);
  s.shade = static_cast<const float4*>(c->shade.ptr);
  s.nodes = static_cast<const rt::BvhNodeQ*>(c->nodes.ptr);
  s.bvh_grid = static_cast<const float*>(c->bvh_grid_dev.ptr);
  s.n_tris = c->n_tris;
  s.use_bvh = c->use_bvh ? 1u : 0u;
  s.paired = (c->tris_paired && !c->no_pairing) ? 1u : 0u;
  s.leaf_pairs = c->leaf_pairs ? 1u : 0u;
  s.stack_depth = static_cast<uint32_t>(c->bvh_depth + 2 < 8 ? 8 : c->bvh_depth + 2);
  s.stack_lds = std::min<uint32_t>(s.stack_depth, static_cast<uint32_t>(c->bvh_stack_lds));
  s.stack_spill = static_cast<uint32_t*>(c->stack_spill.ptr);
  s.materials = static_cast<const float4*>(c->materials.ptr);
  s.n_base_tris = c->n_base_tris ? c->n_base_tris : 1u;
  return s;
}

// Conservative screen bounds of the triangles of a small scene for a pinhole camera at `org` whose
// view-space axes are the columns c0,c1,c2 and whose pixel (x,y) looks along
// (nx/p00, ny/p11, -1), nx = (2(x+.5)-W)/W, ny = (2(y+.5)-H)/H  (the K0 ray; the K2 camera is the
// special case c = identity, p00 = H/(W*slope), p11 = -1/slope).  `jitter_px` widens the bounds by
// the largest possible sub-pixel offset of a primary ray.  A vertex at or behind the camera plane
// makes the projection unbounded: such a triangle is never culled.
bool screen_bounds(const rtpt_ctx* c, const double org[3], const double c0[3], const double c1[3], const double c2[3],
                   double p00, double p11, double jitter_px, rt::TriBounds* out) {
  if (c->host_tris.empty() || c->n_tris > static_cast<uint32_t>(rt::kCullMaxTris)) return false;
  const double W = c->cfg.width, H = c->cfg.height;
  for (uint32_t t = 0; t < c->n_tris; t++) {
    double xmin = 1e30, xmax = -1e30, ymin = 1e30, ymax = -1e30;
    bool unbounded = false;
    for (int k = 0; k < 3; k++) {
      const float* P = c->host_tris.data() + 9 * static_cast<size_t>(t) + 3 * k;
      const double r[3] = {P[0] - org[0], P[1] - org[1], P[2] - org[2]};
      const double xv = c0[0] * r[0] + c0[1] * r[1] + c0[2] * r[2];
      const double yv = c1[0] * r[0] + c1[1] * r[1] + c1[2] * r[2];
      const double zv = c2[0] * r[0] + c2[1] * r[1] + c2[2] * r[2];
      const double len = std::sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
      if (!(-zv > 1e-4 * (len + 1.0))) {
        unbounded = true;
        break;
      }
      const double cx = (p00 * xv / (-zv) + 1.0) * 0.5 * W, cy = (p11 * yv / (-zv) + 1.0) * 0.5 * H;
      xmin = std::min(xmin, cx); xmax = std::max(xmax, cx);
      ymin = std::min(ymin, cy); ymax = std::max(ymax, cy);
    }
    auto clamp16 = [](double v) { return static_cast<int16_t>(std::max(-32000.0, std::min(32000.0, v))); };
    if (unbounded || !(xmax >= xmin)) {
      out[t] = rt::TriBounds{-32000, -32000, 32000, 32000};
    } else {
      const double pad = jitter_px + 1.5;  // pixel index = continuous coordinate - 0.5, +1 px of slack
      out[t] = rt::TriBounds{clamp16(std::floor(xmin - pad)), clamp16(std::floor(ymin - pad)), clamp16(std::ceil(xmax + pad)),
                             clamp16(std::ceil(ymax + pad))};
    }
  }
  return true;
}

bool is_identity(const float* m) {
  for (int i = 0; i < 16; i++)
    if (m[i] != ((i % 5 == 0) ? 1.0f : 0.0f)) return false;
  return true;
}

int launch_check(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(RTPT_E_DEVICE, std::string(what) + ": " + hipGetErrorString(e));
  return RTPT_OK;
}

}  // namespace rtpt_impl

extern "C" {


const char* rtpt_last_error(const rtpt_ctx*) { return g_err.c_str(); }

int rtpt_config_default(rtpt_config* cfg, uint32_t width, uint32_t height) {
  if (!cfg) return fail(RTPT_E_INVALID, "cfg is NULL");
  std::memset(cfg, 0, sizeof(*cfg));
  cfg->struct_size = sizeof(rtpt_config);
  cfg->width = width;
  cfg->height = height;
  cfg->row_begin = 0;
  cfg->row_end = height;
  cfg->max_segments = 32;              // raytrace.comp.glsl:204
  cfg->samples_per_pixel = 1;          // raytrace.comp.glsl:306
  cfg->sigma_n = 128;                  // temporalFiltering.comp.glsl:203
  cfg->sigma_z = 1.0f;                 // :204
  cfg->sigma_l = 4.0f;                 // :205
  cfg->alpha = 0.3f;                   // :243
  cfg->light_radius = 0.20f;           // raytrace.comp.glsl:280
  cfg->light_intensity = 30.0f;        // :281
  cfg->first_hit_light_divisor = 5.0f; // :229
  cfg->fov_slope = 0.20271003f;        // tan(FOV = 0.20), common.h:16 / raytrace.comp.glsl:300
  cfg->pixel_jitter = 0.375f;          // :314
  cfg->ray_offset = 0.0001f;           // :250
  cfg->ray_tmax = 10000.0f;            // :216
  cfg->flags = 0;
  cfg->device = -1;
  return RTPT_OK;
}

// (re)allocate every per-pixel plane for c->cfg's frame and row range and reset the frame state; the scene
// (triangles, BVH, LUTs, tables) is untouched.  Caller-bound planes (rtpt_bind_plane) are dropped.
static int alloc_planes(rtpt_ctx* c) {
  const size_t px = c->pixels();
  int rc = RTPT_OK;
  for (int i = 0; i < 3 && rc == RTPT_OK; i++) rc = alloc_buf(c->color[i], px * 16);
  for (int i = 0; i < 2 && rc == RTPT_OK; i++) rc = alloc_buf(c->vis[i], px * 4);
  if (rc == RTPT_OK) rc = alloc_buf(c->worldpos, px * 16);
  if (rc == RTPT_OK) rc = alloc_buf(c->gradient, px * 16);
  if (rc == RTPT_OK) rc = alloc_buf(c->depth, px * 4);
  if (rc == RTPT_OK && !c->raycount.ptr) rc = alloc_buf(c->raycount, 8 * rt::kRayCounters);
  for (auto& b : c->path_queue) free_buf(b);  // sized per frame: re-created by the next rtpt_raytrace
  free_buf(c->normals);  // sized per frame: re-created by the next rtpt_gbuffer
  c->normals_y0 = c->normals_y1 = 0;
  if (c->cfg.flags & RTPT_FLAG_EXT_VARIANCE) {
    for (int i = 0; i < 2 && rc == RTPT_OK; i++) rc = alloc_buf(c->moments[i], px * 16);
    for (int i = 0; i < 2 && rc == RTPT_OK; i++) rc = alloc_buf(c->variance[i], px * 4);
    for (int i = 0; i < 2 && rc == RTPT_OK; i++) {
      (void)hipMemsetAsync(c->moments[i].ptr, 0, px * 16, c->stream);
      (void)hipMemsetAsync(c->variance[i].ptr, 0, px * 4, c->stream);
    }
    c->moments_cur = 0;
    c->variance_last = 0;
    if (rc == RTPT_OK && (c->cfg.flags & RTPT_FLAG_EXT_SVGF_VARIANCE)) rc = alloc_buf(c->var_scale, px * 4);
  }
  if (rc == RTPT_OK && (c->debug_mask & RTPT_DEBUG_HIT_ID)) rc = alloc_buf(c->hit_id, px * 4);
  if (rc == RTPT_OK && (c->debug_mask & RTPT_DEBUG_PREV_PIXEL)) rc = alloc_buf(c->prev_pixel, px * 8);
  if (rc != RTPT_OK) return rc;
  // Vulkan images start undefined; zero them so readback before the first frame is defined
  for (int i = 0; i < 3; i++) (void)hipMemsetAsync(c->color[i].ptr, 0, px * 16, c->stream);
  for (int i = 0; i < 2; i++) (void)hipMemsetAsync(c->vis[i].ptr, 0, px * 4, c->stream);
  (void)hipMemsetAsync(c->worldpos.ptr, 0, px * 16, c->stream);
  (void)hipMemsetAsync(c->gradient.ptr, 0, px * 16, c->stream);
  (void)hipMemsetAsync(c->depth.ptr, 0, px * 4, c->stream);
  (void)hipMemsetAsync(c->raycount.ptr, 0, 8 * rt::kRayCounters, c->stream);
  if (c->hit_id.ptr) (void)hipMemsetAsync(c->hit_id.ptr, 0, px * 4, c->stream);
  if (c->prev_pixel.ptr) (void)hipMemsetAsync(c->prev_pixel.ptr, 0, px * 8, c->stream);
  hipError_t e = hipStreamSynchronize(c->stream);
  if (e != hipSuccess) return fail(RTPT_E_DEVICE, std::string("initial clear: ") + hipGetErrorString(e));
  for (int i = 0; i < 3; i++) {
    c->color_of_role[i] = i;
    c->alpha_depth[i] = false;
  }
  c->vis_cur = 0;
  c->final_swapped = false;
  c->image_alias = false;
  c->hist_y0 = c->hist_y1 = 0;
  c->final_y0 = c->final_y1 = 0;
  c->guides_y0 = c->guides_y1 = 0;
  c->ext_history = nullptr;
  c->ext_prev_vis = c->ext_moments = nullptr;
  c->count_y0 = static_cast<int>(c->cfg.row_begin);
  c->count_y1 = static_cast<int>(c->cfg.row_end);
  return RTPT_OK;
}

int rtpt_create(const rtpt_config* cfg, rtpt_ctx** out) {
  if (!cfg || !out) return fail(RTPT_E_INVALID, "NULL argument");
  *out = nullptr;
  if (cfg->struct_size != sizeof(rtpt_config)) return fail(RTPT_E_INVALID, "rtpt_config.struct_size mismatch (ABI)");
  if (cfg->width == 0 || cfg->height == 0 || cfg->row_begin >= cfg->row_end || cfg->row_end > cfg->height)
    return fail(RTPT_E_INVALID, "bad frame / row range");
  if (cfg->max_segments == 0 || cfg->samples_per_pixel == 0 || cfg->sigma_n < 1)
    return fail(RTPT_E_INVALID, "max_segments, samples_per_pixel and sigma_n must be >= 1");
  if ((cfg->flags & RTPT_FLAG_EXT_SVGF_VARIANCE) && !(cfg->flags & RTPT_FLAG_EXT_VARIANCE))
    return fail(RTPT_E_INVALID, "RTPT_FLAG_EXT_SVGF_VARIANCE completes RTPT_FLAG_EXT_VARIANCE: set both");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
    (void)hipGetLastError();
    return fail(RTPT_E_NO_GPU, "no HIP device visible; this library has no CPU fallback");
  }
  int dev = cfg->device;
  if (dev < 0) {
    HIP_TRY(hipGetDevice(&dev));
  } else {
    if (dev >= ndev) return fail(RTPT_E_INVALID, "device ordinal out of range");
    HIP_TRY(hipSetDevice(dev));
  }
  rtpt_ctx* c = new (std::nothrow) rtpt_ctx();
  if (!c) return fail(RTPT_E_NOMEM, "host allocation failed");
  c->cfg = *cfg;
  c->device = dev;
  for (int i = 0; i < 3; i++) c->color_of_role[i] = i;
  hipError_t e = hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking);
  if (e != hipSuccess) {
    delete c;
    return fail(RTPT_E_DEVICE, std::string("hipStreamCreate: ") + hipGetErrorString(e));
  }
  c->stream = c->own_stream;
  {
    // per-DEVICE launch state (contexts on different GPUs of one process are independent, rtpt.h): CU count for the
    // persistent grids and the >64 KiB dynamic-LDS attribute of the staged filter kernels
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) c->n_cu = prop.multiProcessorCount;
    e = rt::prepare_device_atrous();
    if (e == hipSuccess) e = rt::prepare_device_atrous_chain();
    if (e != hipSuccess) {
      rtpt_destroy(c);
      return fail(RTPT_E_DEVICE, std::string("hipFuncSetAttribute: ") + hipGetErrorString(e));
    }
  }
  c->count_y0 = static_cast<int>(cfg->row_begin);
  c->count_y1 = static_cast<int>(cfg->row_end);
  // tuning knobs for A/B runs on the box (never needed for correctness: every setting computes the same pixels)
  if (const char* v = std::getenv("RTPT_NO_TRI_PAIRS")) c->no_pairing = std::atoi(v) != 0;
  if (const char* v = std::getenv("RTPT_HOST_REFIT")) c->host_refit = std::atoi(v) != 0;
  if (const char* v = std::getenv("RTPT_NO_TRACE_FUSION")) c->fuse_trace = std::atoi(v) == 0;
  if (const char* v = std::getenv("RTPT_TRACE_POOL")) c->trace_pool = std::atoi(v) != 0;
  if (const char* v = std::getenv("RTPT_PT_WINDOW")) c->trace_window = static_cast<uint32_t>(std::max(0, std::atoi(v)));
  if (const char* v = std::getenv("RTPT_CHAIN_G1")) c->filter_policy.chain_g_pin = std::atoi(v);
  if (const char* v = std::getenv("RTPT_CHAIN_GENERIC")) c->filter_policy.chain_generic = std::atoi(v);
  if (const char* v = std::getenv("RTPT_CHAIN_WG_PER_CU")) c->filter_policy.chain_wg_per_cu = std::atoi(v);
  if (const char* v = std::getenv("RTPT_CHAIN_SKEW")) {
    c->filter_policy.chain_skew = c->filter_policy.chain_skew2 = std::atoi(v);
    if (const char* comma = std::strchr(v, ',')) c->filter_policy.chain_skew2 = std::atoi(comma + 1);
  }
  if (const char* v = std::getenv("RTPT_CHAIN_BW")) c->filter_policy.chain_bw = std::atoi(v);
  if (const char* v = std::getenv("RTPT_CHAIN_SW")) c->filter_policy.chain_sw = std::atoi(v);
  if (const char* v = std::getenv("RTPT_CHAIN_SW_G1")) c->filter_policy.chain_sw_g1 = std::atoi(v);
  if (const char* v = std::getenv("RTPT_CHAIN_SW_G3")) c->filter_policy.chain_sw_g3 = std::atoi(v);
  if (const char* v = std::getenv("RTPT_CHAIN_MAX")) c->chain_max = std::max(1, std::min(3, std::atoi(v)));
  if (const char* v = std::getenv("RTPT_CHAIN_FINAL")) c->chain_final = std::atoi(v) != 0;
  if (const char* v = std::getenv("RTPT_BVH_STACK_LDS")) c->bvh_stack_lds = std::max(1, std::atoi(v));
  if (const char* v = std::getenv("RTPT_CHAIN_MIN_PIXELS")) c->chain_min_pixels = std::atoll(v);
  int rc = alloc_planes(c);
  if (rc != RTPT_OK) {
    rtpt_destroy(c);
    return rc;
  }
  *out = c;
  return RTPT_OK;
}

int rtpt_destroy(rtpt_ctx* c) {
  if (!c) return RTPT_OK;
  (void)hipSetDevice(c->device);
  if (c->own_stream) (void)hipStreamSynchronize(c->own_stream);
  for (auto& t : c->timed) {
    (void)hipEventDestroy(t.start);
    (void)hipEventDestroy(t.stop);
  }
  for (auto& e : c->event_pool) (void)hipEventDestroy(e);
  if (c->handoff_event) (void)hipEventDestroy(c->handoff_event);
  for (auto& b : c->color) free_buf(b);
  for (auto& b : c->vis) free_buf(b);
  free_buf(c->normals);
  free_buf(c->path_queue_count);
  free_buf(c->path_pool);
  for (auto& b : c->path_queue) free_buf(b);
  for (auto& b : c->moments) free_buf(b);
  for (auto& b : c->variance) free_buf(b);
  free_buf(c->var_scale);
  for (auto& b : c->lut) free_buf(b);
  for (Buf* b : {&c->worldpos, &c->gradient, &c->depth, &c->prev_pixel, &c->hit_id, &c->raycount, &c->normal_tab, &c->pair_tab, &c->tris,
                 &c->leaf_order, &c->isect_id, &c->isect_leaf, &c->shade, &c->nodes, &c->materials, &c->obj_tris_dev, &c->refit_order,
                 &c->refit_fbox, &c->bvh_grid_dev, &c->ray_tab})
    free_buf(*b);
  if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
  delete c;
  return RTPT_OK;
}

int rtpt_resize(rtpt_ctx* c, uint32_t width, uint32_t height, uint32_t row_begin, uint32_t row_end) {
  if (!c) return fail(RTPT_E_INVALID, "ctx is NULL");
  if (row_begin == 0 && row_end == 0) row_end = height;
  if (width == 0 || height == 0 || row_begin >= row_end || row_end > height) return fail(RTPT_E_INVALID, "bad frame / row range");
  HIP_TRY(hipSetDevice(c->device));
  FLUSH_FILTER(c);
  HIP_TRY(hipStreamSynchronize(c->stream));
  rtpt_config old = c->cfg;
  c->cfg.width = width;
  c->cfg.height = height;
  c->cfg.row_begin = row_begin;
  c->cfg.row_end = row_end;
  c->present_dst = c->present_fused_dst = nullptr;  // a swapchain image registered for the old size is not this size's
  int rc = alloc_planes(c);
  if (rc != RTPT_OK) {  // leave a usable context behind if the old size still fits
    c->cfg = old;
    (void)alloc_planes(c);
    return rc;
  }
  return RTPT_OK;
}

int rtpt_set_stream(rtpt_ctx* c, void* hip_stream) {
  if (!c) return fail(RTPT_E_INVALID, "ctx is NULL");
  FLUSH_FILTER(c);
  c->stream = hip_stream ? static_cast<hipStream_t>(hip_stream) : c->own_stream;
  return RTPT_OK;
}

int rtpt_plane_bytes(const rtpt_ctx* c, rtpt_plane which, size_t* bytes) {
  if (!c || !bytes) return fail(RTPT_E_INVALID, "NULL argument");
  if (which < 0 || which >= RTPT_PLANE_COUNT) return fail(RTPT_E_INVALID, "unknown plane");
  *bytes = plane_size(c, which);
  return RTPT_OK;
}

int rtpt_plane_ptr(rtpt_ctx* c, rtpt_plane which, void** device_ptr) {
  if (!c || !device_ptr) return fail(RTPT_E_INVALID, "NULL argument");
  FLUSH_FILTER(c);  // roles rotate when recorded iterations run
  Buf* b = plane_buf(c, which);
  if (!b) return fail(RTPT_E_INVALID, "unknown plane");
  *device_ptr = b->ptr;
  return RTPT_OK;
}

int rtpt_bind_plane(rtpt_ctx* c, rtpt_plane which, void* device_ptr, size_t bytes) {
  if (!c) return fail(RTPT_E_INVALID, "ctx is NULL");
  FLUSH_FILTER(c);
  Buf* b = plane_buf(c, which);
  if (!b || which == RTPT_PLANE_RAYCOUNT || which == RTPT_PLANE_LUT || which == RTPT_PLANE_LUT_PREV)
    return fail(RTPT_E_INVALID, "plane cannot be bound");
  const size_t need = plane_size(c, which);
  if (device_ptr == nullptr) {
    if (b->owned) return RTPT_OK;
    return alloc_buf(*b, need);
  }
  if (bytes < need) return fail(RTPT_E_INVALID, "bound buffer too small");
  if ((reinterpret_cast<uintptr_t>(device_ptr) & 15u) != 0) return fail(RTPT_E_INVALID, "bound buffer must be 16-byte aligned");
  HIP_TRY(hipStreamSynchronize(c->stream));
  free_buf(*b);
  b->ptr = device_ptr;
  b->bytes = bytes;
  b->owned = false;
  for (int i = 0; i < 3; i++)
    if (b == &c->color[i]) c->alpha_depth[i] = false;
  return RTPT_OK;
}

int rtpt_set_external_history(rtpt_ctx* c, const void* device_ptr, uint32_t row_begin, uint32_t row_end) {
  if (!c) return fail(RTPT_E_INVALID, "ctx is NULL");
  if (device_ptr && (row_begin >= row_end || row_end > c->cfg.height)) return fail(RTPT_E_INVALID, "bad history row range");
  if (device_ptr && (reinterpret_cast<uintptr_t>(device_ptr) & 15u)) return fail(RTPT_E_INVALID, "history buffer must be 16-byte aligned");
  c->ext_history = device_ptr;
  c->ext_hist_y0 = static_cast<int>(row_begin);
  c->ext_hist_y1 = static_cast<int>(row_end);
  return RTPT_OK;
}

int rtpt_set_external_guides(rtpt_ctx* c, const void* prev_vis, const void* moments_prev, uint32_t row_begin, uint32_t row_end) {
  if (!c) return fail(RTPT_E_INVALID, "ctx is NULL");
  if (!prev_vis && !moments_prev) {
    c->ext_prev_vis = c->ext_moments = nullptr;
    return RTPT_OK;
  }
  if (!prev_vis) return fail(RTPT_E_INVALID, "the previous id plane is needed whenever guides are registered");
  if (row_begin >= row_end || row_end > c->cfg.height) return fail(RTPT_E_INVALID, "bad guide row range");
  if ((reinterpret_cast<uintptr_t>(prev_vis) & 3u) || (reinterpret_cast<uintptr_t>(moments_prev) & 15u))
    return fail(RTPT_E_INVALID, "guide buffers must be 4- / 16-byte aligned");
  c->ext_prev_vis = prev_vis;
  c->ext_moments = moments_prev;
  c->ext_guides_y0 = static_cast<int>(row_begin);
  c->ext_guides_y1 = static_cast<int>(row_end);
  return RTPT_OK;
}

int rtpt_stream_wait(rtpt_ctx* c, rtpt_ctx* other) {
  if (!c || !other) return fail(RTPT_E_INVALID, "NULL argument");
  // "everything submitted to `other` so far" includes its recorded iterations; this context's own recorded iterations
  // do not depend on `other` and go out ahead of the wait.  Neither call looks at a plane, so both may run chained.
  int rcw = gbuffer_flush(other);
  if (rcw == RTPT_OK) rcw = gbuffer_flush(c);
  if (rcw == RTPT_OK) rcw = filter_flush(other, true);
  if (rcw == RTPT_OK) rcw = filter_flush(c, true);
  if (rcw) return rcw;
  if (c == other || c->stream == other->stream) return RTPT_OK;  // one stream is already in order
  if (c->device != other->device) return fail(RTPT_E_INVALID, "rtpt_stream_wait: the contexts are on different devices");
  HIP_TRY(hipSetDevice(c->device));

  if (!other->handoff_event) HIP_TRY(hipEventCreateWithFlags(&other->handoff_event, hipEventDisableTiming));
  HIP_TRY(hipEventRecord(other->handoff_event, other->stream));
  HIP_TRY(hipStreamWaitEvent(c->stream, other->handoff_event, 0));
  return RTPT_OK;
}

int rtpt_enable_debug(rtpt_ctx* c, uint32_t mask) {
  if (!c) return fail(RTPT_E_INVALID, "ctx is NULL");
  HIP_TRY(hipSetDevice(c->device));
  if ((mask & RTPT_DEBUG_HIT_ID) && !c->hit_id.ptr) {
    int rc = alloc_buf(c->hit_id, c->pixels() * 4);
    if (rc) return rc;
    HIP_TRY(hipMemsetAsync(c->hit_id.ptr, 0, c->pixels() * 4, c->stream));
  }
  if ((mask & RTPT_DEBUG_PREV_PIXEL) && !c->prev_pixel.ptr) {
    int rc = alloc_buf(c->prev_pixel, c->pixels() * 8);
    if (rc) return rc;
    HIP_TRY(hipMemsetAsync(c->prev_pixel.ptr, 0, c->pixels() * 8, c->stream));
  }
  c->debug_mask = mask;
  return RTPT_OK;
}

// ------------------------------------------------------------------------------------------ sync / copies
int rtpt_sync(rtpt_ctx* c) {
  if (!c) return fail(RTPT_E_INVALID, "ctx is NULL");
  HIP_TRY(hipSetDevice(c->device));
  FLUSH_FILTER(c);
  HIP_TRY(hipStreamSynchronize(c->stream));
  return RTPT_OK;
}

int rtpt_readback(rtpt_ctx* c, rtpt_plane which, void* dst, size_t bytes) {
  if (!c || !dst) return fail(RTPT_E_INVALID, "NULL argument");
  FLUSH_FILTER(c);
  Buf* b = plane_buf(c, (which == RTPT_PLANE_IMAGE && c->image_alias) ? RTPT_PLANE_PREVIOUS : which);
  if (!b) return fail(RTPT_E_INVALID, "unknown plane");
  if (!b->ptr) return fail(RTPT_E_INVALID, "plane not allocated (scene not uploaded / debug plane not enabled)");
  const size_t need = plane_size(c, which);
  if (bytes < need) return fail(RTPT_E_INVALID, "destination too small");
  HIP_TRY(hipSetDevice(c->device));
  if (which == RTPT_PLANE_RAYCOUNT) {  // kept as partial sums on the device
    unsigned long long part[rt::kRayCounters];
    HIP_TRY(hipMemcpyAsync(part, b->ptr, sizeof part, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    unsigned long long total = 0;
    for (unsigned long long v : part) total += v;
    std::memcpy(dst, &total, sizeof total);
    return RTPT_OK;
  }
  HIP_TRY(hipMemcpyAsync(dst, b->ptr, need, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  for (int i = 0; i < 3; i++)
    if (b == &c->color[i] && c->alpha_depth[i]) {
      // the reference's colour images have alpha 0; internally alpha carries depth between passes
      float* f = static_cast<float*>(dst);
      for (size_t px = 0, n = need / 16; px < n; px++) f[4 * px + 3] = 0.0f;
    }
  return RTPT_OK;
}

int rtpt_set_plane(rtpt_ctx* c, rtpt_plane which, const void* src, size_t bytes) {
  if (!c || !src) return fail(RTPT_E_INVALID, "NULL argument");
  FLUSH_FILTER(c);
  Buf* b = plane_buf(c, which);
  if (!b) return fail(RTPT_E_INVALID, "unknown plane");
  if (!b->ptr) return fail(RTPT_E_INVALID, "plane not allocated (scene not uploaded / debug plane not enabled)");
  const size_t need = plane_size(c, which);
  if (bytes < need) return fail(RTPT_E_INVALID, "source too small");
  HIP_TRY(hipSetDevice(c->device));
  HIP_TRY(hipMemcpyAsync(b->ptr, src, need, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  for (int i = 0; i < 3; i++)
    if (b == &c->color[i]) c->alpha_depth[i] = false;
  if (which == RTPT_PLANE_PREVIOUS) {
    c->hist_y0 = static_cast<int>(c->cfg.row_begin);
    c->hist_y1 = static_cast<int>(c->cfg.row_end);
  }
  if (which == RTPT_PLANE_PREV_VIS_ID || which == RTPT_PLANE_MOMENTS_PREV) {
    c->guides_y0 = static_cast<int>(c->cfg.row_begin);
    c->guides_y1 = static_cast<int>(c->cfg.row_end);
  }
  if (which == RTPT_PLANE_LUT_PREV) {
    c->lut_prev_valid = true;
    c->lut_version[c->lut_cur ^ 1] = ~0ull;  // injected content: rebuild when it becomes current
  }
  if (which == RTPT_PLANE_LUT) c->lut_version[c->lut_cur] = ~0ull;
  if (which == RTPT_PLANE_VIS_ID) c->normals_y0 = c->normals_y1 = 0;  // the normal plane no longer matches the ids
  return RTPT_OK;
}

int rtpt_reset_counters(rtpt_ctx* c) {
  if (!c) return fail(RTPT_E_INVALID, "ctx is NULL");
  HIP_TRY(hipSetDevice(c->device));
  HIP_TRY(hipMemsetAsync(c->raycount.ptr, 0, 8 * rt::kRayCounters, c->stream));
  return RTPT_OK;
}

int rtpt_set_count_rows(rtpt_ctx* c, uint32_t y0, uint32_t y1) {
  if (!c) return fail(RTPT_E_INVALID, "ctx is NULL");
  if (y0 > y1) return fail(RTPT_E_INVALID, "y0 > y1");
  c->count_y0 = static_cast<int>(y0);
  c->count_y1 = static_cast<int>(y1);
  return RTPT_OK;
}

// ------------------------------------------------------------------------------------------ timing
int rtpt_timing_enable(rtpt_ctx* c, int enable) {
  if (!c) return fail(RTPT_E_INVALID, "ctx is NULL");
  c->timing_period = enable > 0 ? enable : 0;
  return RTPT_OK;
}

int rtpt_timing_collect(rtpt_ctx* c, double ms_sum[RTPT_K_COUNT], uint32_t launches[RTPT_K_COUNT]) {
  if (!c || !ms_sum || !launches) return fail(RTPT_E_INVALID, "NULL argument");
  HIP_TRY(hipSetDevice(c->device));
  FLUSH_FILTER(c);
  HIP_TRY(hipStreamSynchronize(c->stream));
  for (int i = 0; i < RTPT_K_COUNT; i++) {
    ms_sum[i] = 0.0;
    launches[i] = 0;
  }
  for (auto& t : c->timed) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, t.start, t.stop) == hipSuccess) {
      ms_sum[t.kernel] += ms;
      launches[t.kernel]++;
    }
    c->event_pool.push_back(t.start);
    c->event_pool.push_back(t.stop);
  }
  c->timed.clear();
  return RTPT_OK;
}

const char* rtpt_kernel_name(rtpt_kernel_id k) {
  switch (k) {
    case RTPT_K_GBUFFER: return "k_gbuffer";
    case RTPT_K_LUT: return "k_lut";
    case RTPT_K_GRADIENT: return "k_gradient";
    case RTPT_K_PATHTRACE: return "k_pathtrace";
    case RTPT_K_ATROUS: return "k_atrous";
    case RTPT_K_ATROUS_FINAL: return "k_atrous_final";
    case RTPT_K_ATROUS_CHAIN: return "k_atrous_chain";
    case RTPT_K_ATROUS_CHAIN_FINAL: return "k_atrous_chain_final";
    case RTPT_K_GBUFFER_GRADIENT: return "k_gbuffer_gradient";
    case RTPT_K_GBUFFER_PATHTRACE: return "k_gbuffer_pathtrace";
    case RTPT_K_PRESENT: return "k_present";
    default: return "?";
  }
}

}  // extern "C"
