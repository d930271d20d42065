// api_internal.hpp — what the translation units of the C ABI (api_*.hip) share: the context, the recording state and the
// helpers every entry point uses.  Nothing here is part of the ABI (include/rtpt.h is).
#pragma once
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


namespace rtpt_impl {


extern thread_local std::string g_err;  // rtpt_last_error (api_context.hip)
int fail(int code, const std::string& msg);

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


}  // namespace rtpt_impl
using namespace rtpt_impl;

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
  // K2 of scenes whose BVH is built over fan pairs as the path-pool kernel (kernels.hip: k_pathtrace_pool): RTPT_TRACE_POOL
  // (read at rtpt_create); path_pool = the workgroups' slabs, allocated on first use
  bool trace_pool = false;
  // segments of a path the tile kernel runs before the survivors go through the queue kernels (kernels.hpp: pt_first_window is
  // the default); RTPT_PT_WINDOW at rtpt_create, 0 = the default
  uint32_t trace_window = 0;
  Buf path_pool;
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

namespace rtpt_impl {

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

size_t frame_blocks(const rtpt_ctx* c);
int ensure_stack_spill(rtpt_ctx* c, size_t blocks);
Buf* plane_buf(rtpt_ctx* c, rtpt_plane which);
size_t plane_size(const rtpt_ctx* c, rtpt_plane which);
int check_rows(const rtpt_ctx* c, uint32_t& y0, uint32_t& y1);
rt::FrameGeom geom(const rtpt_ctx* c, uint32_t y0, uint32_t y1);
rt::SceneView scene_view(const rtpt_ctx* c);
bool screen_bounds(const rtpt_ctx* c, const double org[3], const double c0[3], const double c1[3], const double c2[3], double p00, double p11,
                   double jitter_px, rt::TriBounds* out);
bool is_identity(const float* m);
int launch_check(const char* what);
int apply_model(rtpt_ctx* c, const float* model);  // api_scene.hip

// HIP events around a launch on the launch stream, every rtpt_timing_enable(period)-th frame (rtpt_timing_collect)
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
      } else if (hipEventCreateWithFlags(e, hipEventDisableSystemFence) != hipSuccess) {
        // (timing only: without the system-scope fence a record does not write back and invalidate the caches, which is
        // what made a bracketed launch cost ~5 us and a fully bracketed frame 6 %)
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

}  // namespace rtpt_impl
