// kernels.hpp — launch interface between the C-ABI layer (api_*.hip) and the gfx950 kernels
// (kernels.hip).  Plain structs passed by value as kernel arguments.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "bvh.hpp"

namespace rt {

// Device view of the scene.  Records are float4 triples:
//   isect record  r0 = (v0.x, v0.y, v0.z, e1.x)  r1 = (e1.y, e1.z, e2.x, e2.y)  r2 = (e2.z, n.x, n.y, n.z)   n = e1 x e2
//   shade record  s0 = (v0.xyz, n.x)  s1 = (v1.xyz, n.y)  s2 = (v2.xyz, n.z)     n = unit geometric normal
struct SceneView {
  const float4* isect_id;    // id order  (small-scene wave-uniform brute force)
  // BVH leaf order.  leaf_pairs == 0: one 48-byte isect record per leaf slot.  leaf_pairs != 0: one 80-byte record per PAIR
  // of slots (2q, 2q + 1) — the two fan triangles share v0 and the edge e2_A == e1_B, and their ids ride in the record:
  //   p0 = (v0.xyz, e1.x)  p1 = (e1.yz, e2.xy)  p2 = (e2.z, n.xyz)  p3 = (e2_B.xyz, n_B.x)  p4 = (n_B.yz, id_A, id_B as bits)
  // five 16-byte loads per pair test instead of five + two id loads (the traversal is bound by its vector-memory requests)
  const float4* isect_leaf;
  const uint32_t* leaf_ids;  // triangle id of each leaf slot
  const float4* shade;       // id order
  const BvhNodeQ* nodes;     // 32-byte child-pair nodes, boxes on a 16-bit grid (bvh.hpp)
  const float* bvh_grid;  // device: origin xyz, cell xyz of the nodes' 16-bit grid (rewritten by a device-side refit)
  uint32_t n_tris;
  uint32_t use_bvh;  // 0: brute force over isect_id, 1: BVH traversal
  uint32_t paired;   // brute force only: triangles (2q, 2q+1) share v0 and the edge v2_A == v1_B bitwise (fan-triangulated faces)
  uint32_t leaf_pairs;  // BVH: built over such pairs (bvh.hpp): every leaf holds whole pairs, A right before B
  uint32_t stack_depth;  // BVH traversal stack entries per lane (tree depth + 2)
  // The first stack_lds entries of a lane's stack live in LDS (what sets the workgroups per CU: the 1.15M-triangle
  // tree is 28 deep, 30 KB per workgroup = 5 per CU, and its trace is that sensitive to occupancy: 2 / 3 / 4 / 5
  // workgroups per CU 5.75 / 4.74 / 4.07 / 3.62 ms); deeper entries, which few rays ever reach, go to stack_spill in
  // global memory: [entry - stack_lds][workgroup of the launch][thread]
  uint32_t stack_lds;
  uint32_t* stack_spill;
  // optional per-triangle materials of the BASE mesh (.mtl Kd / Ke; instance i, triangle t reads record t):
  //   m0 = (Kd.rgb, 0)  m1 = (Ke.rgb, 1 if emissive else 0).  NULL: the reference's normal-keyed colours
  const float4* materials;
  uint32_t n_base_tris;
};

// Screen-space bounds of every triangle of a small scene (<= 64), computed on the host per call and
// passed in the kernel-argument segment (scalar loads).  A wave covers 64 pixels of one row; a
// triangle whose padded bounds miss that span cannot be hit by any of the wave's PRIMARY rays, so the
// wave skips it with a scalar branch.  Conservative by construction => results are unchanged.
struct TriBounds {
  int16_t x0, y0, x1, y1;  // inclusive pixel bounds, already padded; x0 > x1 marks "never"
};
constexpr int kCullMaxTris = 64;

struct FrameGeom {
  int32_t W, H;      // full frame
  int32_t row_base;  // first frame row stored in the planes of this context
  int32_t y0, y1;    // rows to compute (frame coordinates)
};

struct GbufferArgs {
  FrameGeom g;
  SceneView scene;
  float org[3];          // camera origin = -R^T t
  float c0[3], c1[3], c2[3];  // columns of the view rotation
  float p00, p11;        // proj[0][0], proj[1][1]
  const float* dvx;      // [W] view-space ray direction x per column, [H] y per frame row (launch_ray_tables)
  const float* dvy;
  float PV[16];          // proj * view (host product, fixed order)
  float tmax;
  uint32_t* vis;
  float4* worldpos;
  float* depth;
  float4* normals;            // nullable: per-pixel normal_tab[id] for the LDS-staged filter of large scenes
  const float4* normal_tab;   // per-id (n.xyz, self weight), built by k_lut
  const float4* area_tab;     // per-id (area of LUT[id], 0, 0, 0), built by k_lut (K1 in the same launch)
  int32_t cull;                      // 1: bounds[] is valid
  // K1 in the same launch (rtpt_temporal_gradient arrived right behind rtpt_gbuffer): the pixel's id and world position
  // are still in registers, so the gradient costs its 16 B/px store and none of its 20 B/px of loads
  int32_t grad_on;
  int32_t grad_y0, grad_y1;          // rows the gradient was asked for
  float g_cam[3], g_light[3], g_light_prev[3], g_color[3], g_color_prev[3];
  const float4* lut;
  const float4* lut_prev;
  float4* grad;
  TriBounds bounds[kCullMaxTris];
};

struct LutArgs {
  uint32_t n_tris;
  const float4* shade;  // world-space vertices
  float model[16];
  float4* lut;          // (n+1) x 3 float4 (stride 48 B)
  float4* normal_tab;   // (n+1) float4: xyz = unit normal of LUT[id] (id 0: (0,0,1)), w = self weight
  float4* area_tab;     // (n+1) float4: x = tri_area(LUT[id]) — the denominator of the area-ratio barycentrics
                        // (temporalGradient.comp.glsl:60), a per-triangle value K1 used to recompute per pixel
  float* pair_tab;      // (n+1)^2 pow(max(0,dot(n_p,n_q)),sigma_n), NULL when n+1 > 64
  int32_t sigma_n;
};

struct GradientArgs {
  FrameGeom g;
  float cam[3], light[3], light_prev[3], color[3], color_prev[3];
  const uint32_t* vis;
  const float4* worldpos;
  const float4* lut;
  const float4* lut_prev;
  const float4* normal_tab;  // per-id normals of the current LUT (k_lut)
  const float4* area_tab;    // per-id triangle areas of the current LUT (k_lut)
  float4* grad;
};

struct PathtraceArgs {
  FrameGeom g;
  SceneView scene;
  uint32_t frame, batch;
  uint32_t max_segments, spp;
  float cam[3];
  float light_c[3];
  float light_col[3];        // currentCameraColor * intensity
  float light_col_first[3];  // light_col / first_hit_light_divisor
  float light_r2;            // radius * radius
  float slope, jitter, ray_offset, tmax;
  float4* image;        // written as (rgb, depth): see atrous.hip "rgbd"
  const float* depth;   // G-buffer depth of the same pixels
  uint32_t* hit_id;  // nullable
  unsigned long long* raycount;
  int32_t count_y0, count_y1;  // rows whose queries are counted
  int32_t compact;             // 1: compact surviving paths to the front of the block after every segment
  int32_t cull;                // 1: bounds[] is valid for the primary segment
  int32_t n_cu;                // compute units of the context's device (persistent queue-kernel grid)
  uint32_t multi_off;          // dword offset of the spp > 1 accumulators in dynamic LDS (set by launch_pathtrace)
  uint32_t tiles_y;            // rows of tiles of the traced rows (set by launch_pathtrace; the fused launch's grid is taller)
  void* pool_slab;             // path-pool form of the tile kernel (kernels.hip: k_pathtrace_pool): pathtrace_pool_bytes() of
                               // scratch owned by the context, or NULL
  // long paths (spp == 1, max_segments > 4): a launch covers the segment window [seg_begin, seg_end) and hands the
  // unfinished paths to the next one through a queue of 48-byte records (set by launch_pathtrace)
  uint32_t seg_begin, seg_end;
  uint32_t first_window;       // segments the tile kernel runs before it hands the survivors to the queue kernels (0: pt_first_window())
  void* q_out;                 // records written by this launch
  uint32_t* q_out_count;
  const void* q_in;            // records read by k_pathtrace_queue
  const uint32_t* q_in_count;
  // queue storage owned by the context: two buffers of `queue_capacity` records and two counters, or NULL
  void* queue[2];
  uint32_t* queue_count;       // [2][kPathQueues]
  uint32_t queue_region;       // records per region (a queue buffer holds kPathQueues regions)
  TriBounds bounds[kCullMaxTris];
};

struct AtrousArgs {
  FrameGeom g;
  int32_t k;             // tap stride = waveletIteration
  int32_t exact;         // 1: contract arithmetic for the weights (RTPT_FLAG_EXACT_FILTER)
  int32_t direct;        // 1: force the direct-load kernel (no LDS staging)
  int32_t rows_stored;   // rows held by the planes (row_base .. row_base+rows_stored-1)
  int32_t cwp;           // comb kernel: staged row length in cells (set by launch_atrous)
  int32_t n_cu;          // compute units of the context's device (persistent grid size)
  int32_t alpha_zero;    // 1: a k < N launch writes alpha 0 instead of the depth (last iteration of an even N)
  int32_t n_strips, n_segs, seg_rows;  // chained iterations (atrous_chain.hip): column strips x row segments (set by launch_atrous_chain)
  int32_t strip_w;       // chained iterations: columns a strip stores (set by launch_atrous_chain; at most 128 - 2 * sum of the later strides)
  const float* pair_tab; // (n_tris+1)^2 id-pair normal weights, NULL when the scene is too large
  const float4* normals; // per-pixel (n.xyz, self weight) plane written by k_gbuffer for such scenes, NULL otherwise
  uint32_t n_tris;       // normal_tab has n_tris + 1 entries
  int32_t sigma_n;
  float cz, cl;          // fast path: -log2(e)/sigma_z, -log2(e)/sigma_l (set by launch_atrous)
  int32_t tiles_x, tiles_y;  // 64x4 tiles covering [y0,y1) (set by launch_atrous)
  float sigma_z, sigma_l;
  const float4* in;
  float4* out;
  const uint32_t* vis;
  const float4* normal_tab;
  // final pass only
  uint32_t frame;
  float alpha;
  const float4* worldpos;
  const float4* history;
  const float4* lut_prev;
  float PVprev[16];
  int2* prev_pixel;      // nullable
  int32_t hist_row_base;          // first frame row stored by the history plane
  int32_t hist_y0, hist_y1;       // frame rows of the history plane that hold a valid previous frame
  // extension modes (RTPT_FLAG_EXT_*; 0 = the reference's filter).  Any bit routes to k_atrous_ext.
  uint32_t ext;
  int32_t stride;                 // tap stride (k, or 2^(k-1) with RTPT_FLAG_EXT_POW2_STRIDE)
  const float4* gradient;         // K1 output (adaptive alpha)
  const uint32_t* prev_vis;       // previous frame's id plane, rows [pvis_y0,pvis_y1) valid, stored from pvis_row_base
  int32_t pvis_y0, pvis_y1, pvis_row_base;
  // final pass only, optional: the swapchain blit (main.cpp:1338-1361) fused into the pass — rows [present_y0, present_y1)
  // of the blended frame are also written as B8G8R8A8_UNORM to present (first byte = pixel (0, present_y0)); rtpt_present_target
  uint32_t* present;
  int32_t present_y0, present_y1;
  const float* var_scale;         // RTPT_FLAG_EXT_SVGF_VARIANCE: the 3x3-prefiltered variance that scales the luminance weight of
                                  // the centre pixel (k_var_prefilter of var_in); NULL: var_in's own value
  const float* var_in;            // RTPT_FLAG_EXT_VARIANCE: per-pixel luminance variance read by this iteration
  float* var_out;                 //                         ... and the filtered variance it writes
};

// RTPT_FLAG_EXT_VARIANCE: temporal accumulation of the luminance moments before the first filter iteration
struct MomentsArgs {
  FrameGeom g;
  uint32_t frame;
  float alpha;
  const float4* traced;
  const uint32_t* vis;
  const float4* worldpos;
  const float4* lut_prev;
  float PVprev[16];
  const uint32_t* prev_vis;     // previous frame's ids and moments: frame rows [hist_y0, hist_y1), stored from hist_row_base
  const float4* moments_prev;
  int32_t hist_row_base, hist_y0, hist_y1;
  int32_t svgf;         // RTPT_FLAG_EXT_SVGF_VARIANCE: spatial estimate for histories shorter than 4 frames
  int32_t rows_stored;  // rows the planes hold from g.row_base on: the 7x7 estimate never reads beyond them (strips trace 3
                        // rows more than their filter needs, so every variance that is consumed saw its whole window)
  float4* moments_out;  // (m1, m2, n, var)
  float* var_out;
};
void launch_moments(const MomentsArgs& a, hipStream_t s);
// RTPT_FLAG_EXT_SVGF_VARIANCE: out = 3x3 Gaussian (1 2 1 / 2 4 2 / 1 2 1) / 16 of var, frame-clamped, rows [g.y0, g.y1)
void launch_var_prefilter(const FrameGeom& g, int rows_stored, const float* var, float* out, hipStream_t s);

// Long paths: segments handled by the tile kernel before the survivors are queued; every later window is twice as
// long.  Swept at 4K on the Cornell box (k_pathtrace, 8 / 16 / 32 segments; single launch 969 / 1627 / 2843 us):
// 2: 1028 / 1376 / 1622, 3: 893 / 1246 / 1489, 4: 836 / 1186 / 1421, 6: 891 / 1238 / 1494.  BVH scenes use twice the
// window: the queue order is the order of arrival, not of the image, and the lost ray coherence costs the traversal
// more than the denser waves win (1.15M triangles, 8 segments: 3.65 -> 3.80 ms with a window of 4).  Windows shorter
// than the BASELINE configs' 4 segments lose everywhere, also on the 270-row strip of an 8-rank job (k_pathtrace
// 89 us in one launch, 108 / 115 us with windows of 3 / 2): the hand-over costs more than the drain tail it removes.
#ifndef RTPT_PT_PHASE0
#define RTPT_PT_PHASE0 4
#endif
#ifndef RTPT_PATH_QUEUES
#define RTPT_PATH_QUEUES 1  // 1 / 8 / 16 / 64 regions: reference frame 352 / 403 / 421 / 418 us, 4K 32 segments 1421 / 1434 / 1441 / 1429
#endif
constexpr uint32_t kPathQueues = RTPT_PATH_QUEUES;  // regions (and counters) per queue buffer
#ifndef RTPT_PT_BVH_MULT
#define RTPT_PT_BVH_MULT 2
#endif
inline uint32_t pt_first_window(bool use_bvh) { return use_bvh ? RTPT_PT_BVH_MULT * RTPT_PT_PHASE0 : RTPT_PT_PHASE0; }

constexpr uint32_t kRayCounters = 256;  // RAYCOUNT is kept as this many partial sums (power of two), added up at readback

constexpr uint32_t kExtAdaptiveAlpha = 0x10u, kExtGauss5 = 0x20u, kExtPow2Stride = 0x40u, kExtDisocclusion = 0x80u;
constexpr uint32_t kExtVariance = 0x100u;
constexpr uint32_t kExtSvgfVariance = 0x800u;  // with kExtVariance: spatial estimate for short histories + 3x3 prefilter
constexpr uint32_t kExtMask = 0x9F0u;

struct ScenePrepArgs {
  uint32_t n_tris;
  const float* tris;  // n x 9 world-space floats
  const uint32_t* leaf_order;
  float4* isect_id;
  float4* isect_leaf;
  float4* shade;
  uint32_t leaf_pairs;  // the BVH was built over fan pairs: isect_leaf holds one 80-byte record per pair (SceneView::isect_leaf)
};

void launch_scene_prepare(const ScenePrepArgs& a, hipStream_t s);

// device-side re-pose + BVH refit (refit.hip)
struct RefitModel {
  float m[16];       // column-major model matrix (rtpt_ubo::model)
  int32_t identity;  // 1: copy the vertices
};
struct RefitArgs {
  const float* tris;           // posed triangles, n x 9
  const uint32_t* leaf_order;  // leaf slot -> triangle id
  const uint32_t* order;       // node indices sorted by height (children before parents)
  BvhNodeQ* nodes;             // device nodes: references are read, boxes rewritten
  float* fbox;                 // n_nodes x 12 floats: unpadded binary32 child boxes (scratch)
  float* grid;                 // 8 floats: origin xyz, cell xyz, pad, 0
};
void launch_pose(uint32_t n_verts, const float* src, float* dst, const RefitModel& m, hipStream_t s);
// level_first[h] .. level_first[h + 1]: the slice of `order` holding the nodes of height h (n_levels + 1 entries)
void launch_refit(const RefitArgs& a, const uint32_t* level_first, int n_levels, uint32_t n_nodes, float pad_rel, hipStream_t s);
// gather isect records into class order: out[t] = isect_id[ids[t]] (3 float4 each), n entries
void launch_lut(const LutArgs& a, hipStream_t s);
void launch_gbuffer(const GbufferArgs& a, hipStream_t s);
void launch_ray_tables(int W, int H, float p00, float p11, float* dvx, float* dvy, hipStream_t s);
void launch_gradient(const GradientArgs& a, hipStream_t s);
// gb != NULL: K0 (+ K1) of gb's rows run inside the tile kernel's launch, behind the tracing tiles (pathtrace_fuses_gbuffer says
// whether they can; pathtrace_grid_blocks = the workgroups of that launch, what the BVH stack's spill area is sized for)
void launch_pathtrace(const PathtraceArgs& a, const GbufferArgs* gb, hipStream_t s);
bool pathtrace_fuses_gbuffer(const PathtraceArgs& a, const GbufferArgs& g);
uint32_t pathtrace_grid_blocks(const PathtraceArgs& a, const GbufferArgs* gb);
bool pathtrace_uses_pool(const PathtraceArgs& a);
size_t pathtrace_pool_bytes(int W, int rows);
void launch_atrous(const AtrousArgs& a, bool final_pass, hipStream_t s);
bool atrous_final_fuses_present(const AtrousArgs& a);
// `levels` consecutive iterations k, k+1, .. in one launch, intermediates in LDS (atrous_chain.hip): a.k = the first
// stride, a.in / a.out = input of the first and output of the last iteration (distinct buffers), final_pass = the last
// level is the frame's FINAL pass (reprojection + blend)
// launch policy switches of the filter kernels, read from the environment ONCE per context (rtpt_create) — A/B runs and the
// tests that pin a variant set them before creating the context; 0 = the shipped choice
struct FilterPolicy {
  int chain_g_pin = 0;      // RTPT_CHAIN_G1: rows per level and step of a chained pair (2, 3, 4)
  int chain_generic = 0;    // RTPT_CHAIN_GENERIC: the instantiation that reads its strides instead of having them compiled in
  int chain_wg_per_cu = 0;  // RTPT_CHAIN_WG_PER_CU: workgroups per CU the row segments are sized for
  int chain_sw = 0;         // RTPT_CHAIN_SW: the sliding-window form of the pair (round 3, slower)
  int chain_sw_g1 = 0, chain_sw_g3 = 0;  // RTPT_CHAIN_SW_G1 / _G3: its rows per step
  // RTPT_CHAIN_SKEW: percent by which the row segments of a chained launch differ with the age of their workgroups on a CU
  // (atrous_chain.hip: chain_segments), "a" or "a,b": a for four workgroups per CU, b for two or three; -1 = the built-in values
  int chain_skew = -1, chain_skew2 = -1;
  int chain_bw = 0;         // RTPT_CHAIN_BW: columns a strip of the pair (1,2) stores (A/B; 0 = the widest, 124)
};
void launch_atrous_chain(const AtrousArgs& a, int levels, bool final_pass, const FilterPolicy& pol, hipStream_t s);
bool atrous_chain_supported(int k0, int levels, uint32_t n_tris);
hipError_t prepare_device_atrous_chain();
hipError_t prepare_device_atrous();  // per device, from rtpt_create: raises the dynamic-LDS limit of the staged filter kernels
void launch_stamp_depth(const FrameGeom& g, float4* color, const float* depth, hipStream_t s);
// swapchain blit (main.cpp:1338-1361): rows [g.y0,g.y1) of `image` -> B8G8R8A8_UNORM at dst (first byte = pixel (0, g.y0))
void launch_present(const FrameGeom& g, const float4* image, uint32_t* dst, hipStream_t s);
void launch_selftest_math(int op, const float* in, float* out, size_t n, hipStream_t s);
void launch_selftest_exhaustive(int op, unsigned long long* out, hipStream_t s);
void launch_selftest_div(int mode, uint32_t pass, unsigned long long* out, hipStream_t s);
void launch_selftest_trace(const SceneView& scene, const float* rays, size_t n, float tmax, uint32_t* out_id,
                           float* out_t, hipStream_t s);

}  // namespace rt
