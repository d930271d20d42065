// api.hip — the C ABI of include/rtpt.h: context, plane roles, scene upload, one entry point per
// reference dispatch.  Host code only (compiled by hipcc together with kernels.hip).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cfloat>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../include/rtpt.h"
#include "bvh.hpp"
#include "kernels.hpp"
#include "rtpt_math.hpp"

namespace {

thread_local std::string g_err;

int fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}

#define HIP_TRY(expr)                                                                                 \
  do {                                                                                                \
    hipError_t e_ = (expr);                                                                           \
    if (e_ != hipSuccess)                                                                             \
      return fail(RTPT_E_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_));                  \
  } while (0)

struct Buf {
  void* ptr = nullptr;
  size_t bytes = 0;
  bool owned = false;
};

enum ColorRole { ROLE_IMAGE = 0, ROLE_FILTERED = 1, ROLE_PREVIOUS = 2 };

struct FilterCall {  // one recorded rtpt_temporal_filter call
  rtpt_push_constants pc;
  rtpt_ubo ubo;
  bool has_ubo;
  uint32_t y0, y1;
};

struct TimedLaunch {
  int kernel;
  hipEvent_t start, stop;
};

}  // namespace

struct rtpt_ctx {
  rtpt_config cfg;
  int device = 0;
  int n_cu = 256;  // compute units of `device` (persistent-grid sizes); per context, not per process
  hipStream_t own_stream = nullptr;
  hipStream_t stream = nullptr;

  Buf color[3];          // physical RGBA32F buffers
  int color_of_role[3];  // role -> physical index
  bool alpha_depth[3] = {false, false, false};  // physical buffer carries depth in alpha ("rgbd")
  Buf vis[2];
  int vis_cur = 0;  // vis[vis_cur] = VIS_ID, the other PREV_VIS_ID
  Buf lut[2];
  int lut_cur = 0;
  Buf worldpos, gradient, depth, prev_pixel, hit_id, raycount, normal_tab, pair_tab;
  Buf moments[2], variance[2];  // RTPT_FLAG_EXT_VARIANCE
  Buf var_scale;                // RTPT_FLAG_EXT_SVGF_VARIANCE: the prefiltered variance of the iteration being launched
  Buf path_queue[2], path_queue_count;  // long paths: survivors handed from one k_pathtrace launch to the next
  Buf normals;                  // per-pixel normal plane for the LDS-staged filter of scenes without an id-pair table
  int normals_y0 = 0, normals_y1 = 0;  // rows for which it matches VIS_ID
  uint64_t normals_frame = ~0ull;      // frame (frames_ended) those rows belong to
  int moments_cur = 0;          // moments[moments_cur] is written this frame, the other one is the history
  int variance_last = 0;        // variance[] buffer holding the newest values

  // scene
  uint32_t n_tris = 0;
  Buf tris, leaf_order, isect_id, isect_leaf, shade, nodes;
  // device-side re-pose + refit (refit.hip): the uploaded (un-posed) triangles, the nodes sorted by height, the scratch
  // boxes and the grid the traversal reads.  BVH scenes only; small brute-force scenes keep host_tris for the screen bounds
  Buf obj_tris_dev, refit_order, refit_fbox, bvh_grid_dev;
  // rtpt_present_target: the swapchain image rows the NEXT final pass also writes (fused blit); present_fused_* describe
  // what the last final pass actually wrote, so that rtpt_present can skip its own launch
  void* present_dst = nullptr;
  int present_y0 = 0, present_y1 = 0;
  void* present_fused_dst = nullptr;
  int present_fused_y0 = 0, present_fused_y1 = 0;
  Buf ray_tab;  // K0: view-space ray direction per column / per row, for the projection and size below
  float ray_tab_p00 = 0.f, ray_tab_p11 = 0.f;
  uint32_t ray_tab_w = 0, ray_tab_h = 0;
  std::vector<uint32_t> refit_level_first;  // slice of refit_order per height (levels + 1 entries)
  uint32_t n_nodes = 0;
  bool host_refit = false;  // RTPT_HOST_REFIT=1: round 2's host path for every scene (A/B)
  bool use_bvh = false;
  rt::BvhGrid bvh_grid{};
  int bvh_depth = 0;
  bool tris_paired = false;  // every (2q, 2q+1) is a fan pair: same v0, v2_A == v1_B bitwise (kernels.hip tri_pair_test)
  bool leaf_pairs = false;   // ... and the BVH was built over those pairs (bvh.hpp build_bvh(pairs))
  bool no_pairing = false;   // RTPT_NO_TRI_PAIRS=1: A/B switch
  std::vector<float> host_tris;  // flattened world-space triangles, kept for small scenes (screen bounds)
  // animated model matrix (main.cpp:1469 recomputes ubo.model every frame; it is the identity there): the scene as
  // uploaded (object space = instance transforms applied, model not), its BVH topology, and the model it is posed with
  std::vector<float> obj_tris;
  rt::Bvh bvh_host;
  float model[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  uint64_t model_version = 0;          // bumped whenever the posed geometry changes
  uint64_t lut_version[2] = {~0ull, ~0ull};  // model_version each LUT buffer was built for
  Buf materials;                       // optional per-base-triangle (Kd, Ke) records, rtpt_scene_set_materials
  uint32_t n_base_tris = 0;

  // frame state
  bool lut_prev_valid = false;   // D3
  bool tables_valid = false;     // normal / id-pair tables match the scene
  bool final_swapped = false;    // the final filter pass already rotated IMAGE <-> FILTERED this frame
  bool image_alias = false;      // between rtpt_end_frame and the next rtpt_raytrace IMAGE reads as PREVIOUS
  int hist_y0 = 0, hist_y1 = 0;  // rows of PREVIOUS holding a valid previous frame
  int final_y0 = 0, final_y1 = 0;
  uint32_t debug_mask = 0;
  const void* ext_history = nullptr;  // rtpt_set_external_history
  const void* ext_prev_vis = nullptr; // rtpt_set_external_guides: previous frame's ids / moments gathered across strips
  const void* ext_moments = nullptr;
  int ext_guides_y0 = 0, ext_guides_y1 = 0;
  int guides_y0 = 0, guides_y1 = 0;   // rows of the context's own previous id / moment planes that hold a previous frame
  hipEvent_t handoff_event = nullptr; // rtpt_stream_wait(x, this): recorded on this context's stream
  int ext_hist_y0 = 0, ext_hist_y1 = 0;
  int count_y0 = 0, count_y1 = 0;  // rows counted into RAYCOUNT

  // K0 recorded by rtpt_gbuffer: launched together with K1 when rtpt_temporal_gradient follows at once, alone otherwise
  rt::GbufferArgs pending_gb{};
  bool pending_gb_valid = false;
  // recorded K0 (+ K1): rtpt_raytrace right behind them launches all three as one grid (kernels.hip: k_gbuffer_pathtrace);
  // RTPT_NO_TRACE_FUSION=1 (read at rtpt_create) keeps K0 + K1 a launch of their own for A/B runs
  bool fuse_trace = true;
  rt::FilterPolicy filter_policy;  // RTPT_CHAIN_* (read once, here: rtpt_create)
  // K3 iterations recorded by rtpt_temporal_filter and not launched yet (see filter_flush)
  std::vector<FilterCall> pending;
  int chain_max = 2;        // iterations per chained launch (1 = never chain)
  bool chain_final = false; // may a chain end in the FINAL pass
  // A chain slides down column strips in row segments and pays sum(s) + lag rows of pipeline fill per segment: with
  // fewer pixels than this per launch the segments that fill the GPU are too short for that to pay, so smaller launches
  // run one kernel per iteration (measured, pair vs 2 separate: 4K 98 vs 128 us, 1080p 38.2 vs 40.3, a 300-row strip of
  // 3840 columns 27.1 vs 28.9 — the kernel itself no longer wins there, the launch it saves does: frame 0.171 vs 0.173 ms)
  int64_t chain_min_pixels = 1000000;

  // timing
  // BVH traversal: stack entries per lane kept in LDS (kernels.hpp SceneView::stack_lds) and the global-memory home of
  // the deeper ones, sized for the largest grid that traverses (ensure_stack_spill)
  int bvh_stack_lds = 16;
  Buf stack_spill;
  size_t stack_spill_blocks = 0;

  int timing_period = 0;          // 0 off, n: kernels of every n-th frame are bracketed by events
  uint64_t frames_ended = 0;
  bool timing_now() const { return timing_period > 0 && (frames_ended % static_cast<uint64_t>(timing_period)) == 0; }
  std::vector<TimedLaunch> timed;
  std::vector<hipEvent_t> event_pool;

  uint32_t rows() const { return cfg.row_end - cfg.row_begin; }
  bool width_fits_i16() const { return cfg.width < 30000 && cfg.height < 30000; }
  size_t pixels() const { return static_cast<size_t>(rows()) * cfg.width; }
};

namespace {

// K3 iterations recorded by rtpt_temporal_filter are launched before anything else looks at or changes the planes
int filter_flush(rtpt_ctx* c, bool fuse);
int gbuffer_flush(rtpt_ctx* c);
#define FLUSH_FILTER(c)                              \
  do {                                               \
    int rcf_ = gbuffer_flush(c);                     \
    if (rcf_ == RTPT_OK) rcf_ = filter_flush((c), false); \
    if (rcf_) return rcf_;                           \
  } while (0)

int alloc_buf(Buf& b, size_t bytes);
void free_buf(Buf& b);
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

struct Timer {
  rtpt_ctx* c;
  bool on;
  TimedLaunch t;
  Timer(rtpt_ctx* ctx, int kernel) : c(ctx), on(ctx->timing_now()) {
    if (!on) return;
    t.kernel = kernel;
    for (hipEvent_t* e : {&t.start, &t.stop}) {
      if (!c->event_pool.empty()) {
        *e = c->event_pool.back();
        c->event_pool.pop_back();
      } else if (hipEventCreate(e) != hipSuccess) {
        on = false;
        return;
      }
    }
    (void)hipEventRecord(t.start, c->stream);
  }
  ~Timer() {
    if (!on) return;
    (void)hipEventRecord(t.stop, c->stream);
    c->timed.push_back(t);
  }
};

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

}  // namespace

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
  if (const char* v = std::getenv("RTPT_CHAIN_G1")) c->filter_policy.chain_g_pin = std::atoi(v);
  if (const char* v = std::getenv("RTPT_CHAIN_GENERIC")) c->filter_policy.chain_generic = std::atoi(v);
  if (const char* v = std::getenv("RTPT_CHAIN_WG_PER_CU")) c->filter_policy.chain_wg_per_cu = std::atoi(v);
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
static int apply_model(rtpt_ctx* c, const float* model) {
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

namespace {
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
}  // namespace

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
  if (a.compact && a.spp == 1 && a.max_segments > rt::pt_first_window(c->use_bvh) && !(c->cfg.flags & RTPT_FLAG_SINGLE_LAUNCH_PATHS)) {
    // a region holds the survivors of ceil(workgroups / kPathQueues) workgroups of 256 paths (kernels.hip); the
    // second buffer is only needed when a third segment window exists
    const size_t blocks = ((static_cast<size_t>(c->cfg.width) + 63) / 64) * ((c->rows() + 3) / 4);
    const size_t region = ((blocks + rt::kPathQueues - 1) / rt::kPathQueues) * 256;
    const size_t cap = region * rt::kPathQueues;
    if (!c->path_queue_count.ptr && (rc = alloc_buf(c->path_queue_count, 2 * rt::kPathQueues * sizeof(uint32_t)))) return rc;
    if (!c->path_queue[0].ptr && (rc = alloc_buf(c->path_queue[0], cap * 48))) return rc;
    if (a.max_segments > 2u * rt::pt_first_window(c->use_bvh) && !c->path_queue[1].ptr && (rc = alloc_buf(c->path_queue[1], cap * 48))) return rc;
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

}  // namespace

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
        const float qlo = g[a] + static_cast<float>(nd.box[rt::bvh_box_lo(side, a)]) * g[3 + a];
        const float qhi = g[a] + static_cast<float>(nd.box[rt::bvh_box_hi(side, a)]) * g[3 + a];
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
          const float qlo = g[a] + static_cast<float>(q[ni].box[rt::bvh_box_lo(side, a)]) * g[3 + a];
          const float qhi = g[a] + static_cast<float>(q[ni].box[rt::bvh_box_hi(side, a)]) * g[3 + a];
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
        const float qlo = g.origin[a] + static_cast<float>(q[ni].box[rt::bvh_box_lo(side, a)]) * g.cell[a];
        const float qhi = g.origin[a] + static_cast<float>(q[ni].box[rt::bvh_box_hi(side, a)]) * g.cell[a];
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
