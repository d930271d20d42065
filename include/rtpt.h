/*
 * rtpt.h — C ABI of the MI355X-native hot path (G-buffer -> temporal gradient -> 1-spp path
 * trace -> N edge-stopping a-trous passes with reprojection + temporal blend).
 *
 * The reference (OnurBasci/Real_Time_Path_Tracing_With_SpatioTemporal_Filtering) has no FFI:
 * its seam is the Vulkan compute dispatch (pipeline + descriptor set + 112-byte push
 * constants + grid).  Every entry point below replaces one such dispatch site (or the
 * resource/scene call that feeds it) and cites it as `file:line` relative to the reference
 * tree.  Plain pointers and sizes only; no C++/torch types cross this boundary.
 *
 * Conventions
 *   - every function returns 0 (RTPT_OK) or a negative RTPT_E_* code; the message is
 *     available from rtpt_last_error().  No exception crosses the ABI.
 *     (reference: NVVK_CHECK aborts / std::runtime_error never caught, main.cpp:99-111,:1532)
 *   - all work is enqueued on ONE HIP stream per context in program order, which gives the
 *     same observable ordering as the reference's vkQueueWaitIdle after each dispatch
 *     (main.cpp:110-111) without the waits.  rtpt_sync / rtpt_readback are the blocking calls.
 *   - a context is not thread-safe; distinct contexts (one per GPU) are independent.
 *   - images are linear row-major, element (x,y) of a plane at index (y-row_begin)*width+x.
 */
#ifndef RTPT_H
#define RTPT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RTPT_ABI_VERSION 4

/* ---- status codes -------------------------------------------------------------------- */
#define RTPT_OK 0
#define RTPT_E_INVALID (-1)  /* bad argument / bad call order */
#define RTPT_E_NOMEM (-2)    /* host or device allocation failed */
#define RTPT_E_DEVICE (-3)   /* HIP runtime error (message in rtpt_last_error) */
#define RTPT_E_NO_SCENE (-4) /* a pass needs rtpt_scene_upload first */
#define RTPT_E_NO_GPU (-5)   /* no HIP device visible: the product path never falls back to CPU */

/* ---- structs shared with the reference host -------------------------------------------- */

/* PushConstants — main.cpp:35-49, raytrace.comp.glsl:9-23 (identical copies in
 * temporalGradient.comp.glsl:11-25 and temporalFiltering.comp.glsl:11-25).
 * Offsets 0,4,16,32,48,64,80,92,96, sizeof 112 (verified against the three .spv). */
typedef struct rtpt_push_constants {
  uint32_t sample_batch;        /* @0  */
  uint32_t frameNumber;         /* @4  */
  uint32_t _pad0[2];
  float cameraPos[3];           /* @16 */
  float _pad1;
  float lightPos[3];            /* @32 */
  float _pad2;
  float lightPosPrev[3];        /* @48 */
  float _pad3;
  float currentCameraColor[3];  /* @64 */
  float _pad4;
  float previousCameraColor[3]; /* @80 */
  int32_t waveletIteration;     /* @92 */
  int32_t maxWaveletIteration;  /* @96 */
  uint32_t _pad5[3];
} rtpt_push_constants;

/* UniformBufferObject — main.cpp:82-90, temporalFiltering.comp.glsl:43-51,
 * visibility.vert.glsl:3-11.  Six column-major mat4 (m[col][row]) @0,64,...,320. */
typedef struct rtpt_ubo {
  float model[16];
  float view[16];
  float proj[16];
  float modelPrev[16];
  float viewPrev[16];
  float projPrev[16];
} rtpt_ubo;

/* VisibilityData — temporalGradient.comp.glsl:5-9 (std430: vec3 padded to 16, stride 48).
 * LUT[t+1] holds the world-space vertices of triangle t; LUT[0] is the background slot
 * (visibility.geom.glsl:56-59). */
typedef struct rtpt_visibility_data {
  float v1[3];
  float _p1;
  float v2[3];
  float _p2;
  float v3[3];
  float _p3;
} rtpt_visibility_data;

/* ---- configuration: the reference's compile-time constants made explicit ---------------- */
#define RTPT_FLAG_EXACT_FILTER 0x1u /* strict-parity filter: contract exp/sqrt/div instead of the
                                       hardware v_exp/v_sqrt/v_rcp (slower; bit-identical to the oracle) */
#define RTPT_FLAG_FORCE_BVH 0x2u    /* traverse the BVH even for scenes small enough for the
                                       wave-uniform brute-force path (<= 64 triangles) */

#define RTPT_FLAG_DIRECT_FILTER 0x4u /* a-trous taps by direct global loads instead of the LDS-staged
                                       tile kernel (the fallback for strides whose halo exceeds LDS) */

#define RTPT_FLAG_NO_PATH_COMPACTION 0x8u /* path tracer: keep one pixel per lane for the whole path instead of
                                            compacting the surviving paths of a tile after every segment */

#define RTPT_FLAG_SINGLE_LAUNCH_PATHS 0x200u /* path tracer: run all segments of a path in the tile kernel instead of
                                              handing the paths that survive 4 / 8 / 16 segments to follow-up launches
                                              through a queue (A/B switch; the image is the same) */

#define RTPT_FLAG_NO_FILTER_FUSION 0x400u /* launch every pass when it is called, one kernel per call, instead of
                                             recording rtpt_temporal_filter's calls of a frame (consecutive iterations
                                             then run chained in one launch) and rtpt_gbuffer (which runs in one launch
                                             with the rtpt_temporal_gradient that follows it) (A/B switch; same pixels) */

/* Extension modes — NOT reference behaviour, default off.  They switch on the pieces of the textbook
 * A-SVGF that the reference declares but leaves unused; any of them routes K3 to a
 * generic direct-load kernel.  They cannot be parity-checked against the reference; tests/ check them
 * against the oracle's restatement of the same definitions. */
#define RTPT_FLAG_EXT_ADAPTIVE_ALPHA 0x10u /* alpha = (1-g)*alpha + g, g = temporalGradient.r
                                              (temporalFiltering.comp.glsl:247-248, commented out there) */
#define RTPT_FLAG_EXT_GAUSS5 0x20u         /* 5x5 taps weighted by gaussianKernel2D/273 (:93-99, unused there) */
#define RTPT_FLAG_EXT_POW2_STRIDE 0x40u    /* tap stride 2^(k-1) instead of k (:135) */
#define RTPT_FLAG_EXT_DISOCCLUSION 0x80u   /* blend history only where the reprojected pixel of the previous
                                              frame's id plane (previousVisibilityBuffer, main.cpp:1367: copied,
                                              never read) shows the same primitive */
#define RTPT_FLAG_EXT_VARIANCE 0x100u      /* SVGF-style variance guidance: before the first filter iteration the first
                                              and second moments of the traced luminance are accumulated along the
                                              reprojected pixel (history only where the previous id plane agrees;
                                              a = max(alpha, 1/(n+1)); short histories n < 4 scale the variance by
                                              4/n); the colour term of the tap weight becomes
                                              exp(-|lum_p - lum_q| / (sigma_l * sqrt(var_p) + 1e-4)) and the variance
                                              is filtered along with weights (h w)^2.  RTPT_PLANE_MOMENTS / _VARIANCE.
                                              On strip contexts every stored row must have been traced (redundant
                                              halo rows), and under camera motion the host gathers the previous
                                              frame's id and moment rows the strip can reach from the other strips
                                              (rtpt_set_external_guides) like it does for the history image. */

#define RTPT_FLAG_EXT_SVGF_VARIANCE 0x800u /* with RTPT_FLAG_EXT_VARIANCE (required), the two pieces of SVGF's variance handling
                                              (Schied et al. 2017) that flag leaves out: a pixel whose moment history is shorter
                                              than 4 frames takes its variance from the 7x7 neighbourhood of the current frame's
                                              luminance (taps on the same primitive only) instead of the temporal estimate, still
                                              scaled by 4/n; and the variance that scales an iteration's luminance weight is the 3x3
                                              Gaussian (1 2 1 / 2 4 2 / 1 2 1) / 16 of the variance plane around the pixel.
                                              On strip contexts the traced rows must reach 3 rows beyond every row whose variance
                                              an iteration reads (the hosts' strip plans do that: strips.py / host/strips.cpp). */

typedef struct rtpt_config {
  uint32_t struct_size;          /* = sizeof(rtpt_config), ABI guard */
  uint32_t width, height;        /* full frame; main.cpp:52-53 (1000x800) */
  uint32_t row_begin, row_end;   /* rows stored by this context (0,height on one GPU);
                                    multi-GPU strips allocate strip +- halo rows */
  uint32_t max_segments;         /* raytrace.comp.glsl:204 (32) */
  uint32_t samples_per_pixel;    /* raytrace.comp.glsl:306 (1) */
  int32_t sigma_n;               /* temporalFiltering.comp.glsl:203 (128; integer exponent) */
  float sigma_z;                 /* :204 (1.0) */
  float sigma_l;                 /* :205 (4.0) */
  float alpha;                   /* :243 (0.3) */
  float light_radius;            /* raytrace.comp.glsl:280 (0.20) */
  float light_intensity;         /* :281 (30) */
  float first_hit_light_divisor; /* :229 (5.0) */
  float fov_slope;               /* tan(FOV), common.h:16, raytrace.comp.glsl:300 (0.20271003) */
  float pixel_jitter;            /* :314 (0.375) */
  float ray_offset;              /* :250 (1e-4) */
  float ray_tmax;                /* :216 (10000) */
  uint32_t flags;
  int32_t device;                /* HIP device ordinal, -1 = current device */
} rtpt_config;

/* planes = the reference's images/buffers (createBuffers main.cpp:357-407) */
typedef enum rtpt_plane {
  RTPT_PLANE_IMAGE = 0,     /* `image`              RGBA32F  main.cpp:359 */
  RTPT_PLANE_FILTERED = 1,  /* `filteredImageBuffer` RGBA32F  main.cpp:400 */
  RTPT_PLANE_PREVIOUS = 2,  /* `previousImage`      RGBA32F  main.cpp:365 */
  RTPT_PLANE_WORLDPOS = 3,  /* `positionBuffer`     RGBA32F  main.cpp:383 */
  RTPT_PLANE_GRADIENT = 4,  /* `temporalGradientBuffer` RGBA32F main.cpp:397 */
  RTPT_PLANE_DEPTH = 5,     /* `depthImage`         f32 (D32F read as .r, D8) main.cpp:378 */
  RTPT_PLANE_VIS_ID = 6,    /* `visibilityBuffer`   u32 (reference: R16F, D9) main.cpp:371 */
  RTPT_PLANE_PREV_VIS_ID = 7, /* `previousVisibilityBuffer` u32 main.cpp:375 */
  RTPT_PLANE_LUT = 8,       /* `visibilityLUT`  (T+1) x rtpt_visibility_data main.cpp:390 */
  RTPT_PLANE_LUT_PREV = 9,  /* `visibilityLUTprevious` main.cpp:395 */
  RTPT_PLANE_PREV_PIXEL = 10, /* build-only observable: reprojected pixel (int32 x,y) written
                                 by the final filter pass; temporalFiltering.comp.glsl:238 */
  RTPT_PLANE_RAYCOUNT = 11, /* build-only: u64[1] closest-hit queries issued by rtpt_raytrace
                               since rtpt_reset_counters (SURVEY 8d "ray").  On the device it is kept as 256
                               partial sums (one counter serialised 32 400 atomics per 4K launch); rtpt_readback
                               adds them up, rtpt_plane_ptr returns the u64[256] array */
  RTPT_PLANE_HIT_ID = 12,   /* build-only observable (debug): u32 first-hit primitive id+1 of the
                               jittered primary ray of rtpt_raytrace, only if enabled */
  RTPT_PLANE_MOMENTS = 13,  /* extension RTPT_FLAG_EXT_VARIANCE: (m1, m2, history length, variance) float4 */
  RTPT_PLANE_VARIANCE = 14, /* extension: f32 variance written by the last filter iteration (or by the moments pass) */
  RTPT_PLANE_MOMENTS_PREV = 15, /* extension: the previous frame's moments (what this frame's accumulation reads) */
  RTPT_PLANE_COUNT = 16
} rtpt_plane;

typedef struct rtpt_ctx rtpt_ctx;

/* ---- lifetime ------------------------------------------------------------------------- */

/* fills the reference's constants (citations on rtpt_config) for a width x height frame */
int rtpt_config_default(rtpt_config* cfg, uint32_t width, uint32_t height);

/* createBuffers (main.cpp:357-407) + createCommandPool/Context init: allocates every plane on
 * the device and one stream.  Returns RTPT_E_NO_GPU when no HIP device is present. */
int rtpt_create(const rtpt_config* cfg, rtpt_ctx** out);
/* freeRessources (main.cpp:1477-1528) */
int rtpt_destroy(rtpt_ctx* ctx);
/* The reference re-creates its size-dependent resources when the framebuffer changes
 * (framebufferResizeCallback main.cpp:275-278, swapChain.acquireAutoResize :1310).  Re-allocates every
 * per-pixel plane for a width x height frame storing rows [row_begin,row_end) (0,0 = the whole frame),
 * zeroes them and restarts the history (the next final pass is a frame-0 pass for the blend unless the
 * caller injects PREVIOUS); the uploaded scene, LUTs and configuration constants are kept.  Planes bound
 * with rtpt_bind_plane are dropped and must be bound again.  Blocks until the stream is idle. */
int rtpt_resize(rtpt_ctx* ctx, uint32_t width, uint32_t height, uint32_t row_begin, uint32_t row_end);
/* last error message of this thread's most recent failing call (ctx may be NULL) */
const char* rtpt_last_error(const rtpt_ctx* ctx);

/* run subsequent passes on a caller-owned hipStream_t (e.g. the stream RCCL halo exchanges are
 * ordered on).  NULL restores the context's own stream. */
int rtpt_set_stream(rtpt_ctx* ctx, void* hip_stream);

/* Bind caller-owned device memory as the storage of one colour/guide plane, like the reference
 * app owning every VkImage bound into the descriptor sets (createAndBindDescriptorSet,
 * main.cpp:744-908).  bytes must be >= rtpt_plane_bytes.  NULL returns to context-owned memory. */
int rtpt_bind_plane(rtpt_ctx* ctx, rtpt_plane which, void* device_ptr, size_t bytes);
/* current device pointer playing the role `which`.  Roles ROTATE among the three colour buffers — at the final filter
 * pass, when iterations run chained, and at rtpt_end_frame — so re-query after those calls; a buffer bound as IMAGE
 * does not stay IMAGE.  Alpha convention on the device: IMAGE after the last iteration of a frame and PREVIOUS after
 * rtpt_end_frame have alpha 0 like the reference's vec4(rgb, 0); between rtpt_raytrace and the last iteration the
 * colour planes carry the G-buffer depth in alpha ("rgbd"), and FILTERED is scratch whose alpha may hold it at any
 * time.  rtpt_readback always returns alpha 0. */
int rtpt_plane_ptr(rtpt_ctx* ctx, rtpt_plane which, void** device_ptr);
int rtpt_plane_bytes(const rtpt_ctx* ctx, rtpt_plane which, size_t* bytes);

/* Multi-GPU strips: the final filter pass fetches previousFrameImage at the REPROJECTED pixel
 * (temporalFiltering.comp.glsl:253), which under camera motion can lie in another rank's strip.
 * The host all-gathers the strips of the previous frame into one buffer covering frame rows
 * [row_begin,row_end) and registers it here; the final pass then reads history from it instead of
 * the context's own PREVIOUS plane.  NULL returns to the PREVIOUS plane. */
int rtpt_set_external_history(rtpt_ctx* ctx, const void* device_ptr, uint32_t row_begin, uint32_t row_end);

/* The same for the other two per-pixel planes of the previous frame that a strip reads at reprojected pixels: the id
 * plane (RTPT_FLAG_EXT_DISOCCLUSION, RTPT_FLAG_EXT_VARIANCE) and the moment plane (RTPT_FLAG_EXT_VARIANCE; may be NULL
 * otherwise).  Both buffers cover frame rows [row_begin,row_end), u32 and float4 per pixel.  NULL, NULL returns to the
 * context's own planes. */
int rtpt_set_external_guides(rtpt_ctx* ctx, const void* prev_vis, const void* moments_prev, uint32_t row_begin, uint32_t row_end);

/* Two frames in flight.  The reference serialises everything with vkQueueWaitIdle (main.cpp:110-111); the only
 * dependency between consecutive frames of this path is the final pass's history fetch, so a host may render
 * even frames in one context and odd frames in another (each on its own stream) and hand the finished frame
 * across: rtpt_stream_wait(ctx, other) makes everything submitted to `ctx` from now on start only after
 * everything submitted to `other` so far has finished (an event, no host block); the history itself is passed
 * with rtpt_set_external_history(ctx, <other's PREVIOUS plane>, ...).  Same device only. */
int rtpt_stream_wait(rtpt_ctx* ctx, rtpt_ctx* other);

/* ---- scene ---------------------------------------------------------------------------- */

/* loadMesh's RT arrays (main.cpp:416-428: objVertices tightly packed xyz, objIndices u32) +
 * buildAccelerationStructure (main.cpp:687-742: one BLAS, instances with 3x4 row-major
 * transforms; NULL/0 = the reference's single identity instance).  Builds the flattened
 * world-space triangle set, the BVH and sizes the LUTs.  Triangle id = instance*n_tris + t. */
int rtpt_scene_upload(rtpt_ctx* ctx, const float* xyz, uint32_t n_verts, const uint32_t* idx,
                      uint32_t n_tris, const float* instance_xforms, uint32_t n_instances);

/* Per-triangle materials (SURVEY.md 8(f) rank 4 — NOT reference behaviour: the reference keys its colours on the
 * normal, raytrace.comp.glsl:155-163, and ships no material library).  tri_material[t] indexes `materials` for
 * triangle t of the mesh given to rtpt_scene_upload (instances share them).  A hit then takes Kd as its albedo
 * instead of the normal-keyed colour (:244), and a surface with Ke != 0 ends the path like the analytic light does
 * (:226-234): throughput *= Ke.  NULL / 0 returns to the reference's colours; rtpt_scene_upload drops them. */
typedef struct rtpt_material {
  float albedo[3];   /* .mtl Kd */
  float emission[3]; /* .mtl Ke */
} rtpt_material;
int rtpt_scene_set_materials(rtpt_ctx* ctx, const uint32_t* tri_material, uint32_t n_tris, const rtpt_material* materials,
                             uint32_t n_materials);

/* ---- per-frame passes, one call per reference dispatch ----------------------------------- */

/* drawVisbilityBuffer (main.cpp:1187-1199; visibility.{vert,geom,frag}.glsl): id, world
 * position, NDC depth planes + LUT for all triangles.  Rows [y0,y1) of the frame (clamped to
 * the stored rows); y0=y1=0 means all stored rows.
 * ubo->model poses the scene (visibility.vert.glsl:24; recomputed per frame at main.cpp:1469, the identity there):
 * when it differs from the last call's, every triangle is re-posed (model * v, the LUT's arithmetic), the BVH is
 * refit and the tables are rebuilt before the pass runs, and rtpt_raytrace traces the posed scene too.  It must be
 * affine and invertible.  LUT_PREV keeps the previous frame's pose, which is what K1 and the reprojection read.
 * Cost of a changed model: BVH scenes (more than 64 triangles, or RTPT_FLAG_FORCE_BVH) are re-posed and refit ON THE
 * DEVICE, on the context's stream, without a host synchronisation (csrc/refit.hip: +0.27 ms for 1,152,000 triangles);
 * small brute-force scenes are re-posed on the host (microseconds) and the call then waits for the stream once. */
int rtpt_gbuffer(rtpt_ctx* ctx, const rtpt_ubo* ubo, uint32_t y0, uint32_t y1);
/* computeTemporalGradient (main.cpp:1201-1220; temporalGradient.comp.glsl:104-172).  Called right behind rtpt_gbuffer
 * (the reference's order, main.cpp:1105-1106) for rows that call covered, the two run as ONE launch: rtpt_gbuffer records
 * its dispatch, and every entry point other than this one launches it first, alone.  Since ABI version 4 the pair stays
 * recorded until rtpt_raytrace (below). */
int rtpt_temporal_gradient(rtpt_ctx* ctx, const rtpt_push_constants* pc, uint32_t y0, uint32_t y1);
/* drawSceneToImage (main.cpp:1222-1253; raytrace.comp.glsl:273-344).  Called right behind rtpt_gbuffer +
 * rtpt_temporal_gradient (main.cpp:1105-1107) whose rows contain [y0, y1), the three run as ONE launch (the G-buffer's
 * workgroups are dispatched behind the tracing ones and fill the trace's tail): any other entry point in between launches the
 * recorded passes first, so the planes always hold what the separate dispatches leave there.  RTPT_FLAG_NO_FILTER_FUSION (or
 * RTPT_NO_TRACE_FUSION=1 in rtpt_create's environment) keeps the launches apart. */
int rtpt_raytrace(rtpt_ctx* ctx, const rtpt_push_constants* pc, uint32_t y0, uint32_t y1);
/* one iteration of applyTemporalFiltering's loop body (main.cpp:1259-1305;
 * temporalFiltering.comp.glsl:191-265).  The host loops k = 1..maxWaveletIteration exactly
 * like main.cpp:1259.  Odd k reads IMAGE and writes FILTERED, even k the reverse
 * (main.cpp:1264-1281).  On k == max (odd) the fused reprojection + blend result becomes
 * IMAGE (D1: taps read the pre-pass snapshot).  ubo supplies viewPrev/projPrev and may be
 * NULL when k < max.
 * The calls of a frame are RECORDED, like the reference records its dispatches into command buffers
 * (main.cpp:1284-1303), and launched when the call with k == max arrives: consecutive iterations then run as one
 * chained kernel whose intermediate image stays in LDS instead of travelling through `filteredImageBuffer` / `image`
 * (same arithmetic per pixel, same bits).  Every other entry point that reads or changes a plane, a stream or the
 * frame state first launches what is recorded, one kernel per iteration, so between iterations each plane holds what
 * the separate dispatches leave there.  After the LAST iteration IMAGE holds the frame; FILTERED is scratch (the
 * reference's own last store to it, temporalFiltering.comp.glsl:152, is dead).  RTPT_FLAG_NO_FILTER_FUSION turns the
 * recording off.  An error of a recorded launch is reported by the call that triggers it. */
int rtpt_temporal_filter(rtpt_ctx* ctx, const rtpt_push_constants* pc, const rtpt_ubo* ubo,
                         uint32_t y0, uint32_t y1);
/* history hand-over of copyImageToSwapChainsCurrentImage (main.cpp:1364-1372):
 * previousImage <- image, previousVisibilityBuffer <- visibilityBuffer, LUTprev <- LUT,
 * done by rotating plane roles (no copy kernels). */
int rtpt_end_frame(rtpt_ctx* ctx);

/* the swapchain blit of copyImageToSwapChainsCurrentImage (main.cpp:1338-1361: `image`, RGBA32F, blitted to the acquired
 * swapchain image, VK_FORMAT_B8G8R8A8_UNORM): frame rows [y0,y1) of the finished frame (call after rtpt_end_frame, or
 * after the last rtpt_temporal_filter) are converted — clamp to [0,1], x*255 + 0.5 truncated, NaN -> 0; bytes B,G,R,A in
 * memory — and written to `dst_device`, a device buffer whose first byte is pixel (0, y0): 4*W bytes per row.  The
 * buffer is the caller's "swapchain image"; with several ranks each rank converts its own rows and the presenting rank
 * gathers them (4 B/px on the wire instead of 16).  Runs on the context's stream. */
int rtpt_present(rtpt_ctx* ctx, void* dst_device, uint32_t y0, uint32_t y1);
/* Optional, before the frame's rtpt_temporal_filter calls: name the swapchain rows of this frame in advance.  The final
 * filter pass then writes them in swapchain format as it stores the frame (one launch and one 16 B/px read less), and the
 * later rtpt_present of the same rows and image returns at once; where the final pass runs in a kernel that cannot fuse
 * the store (extension modes, direct-load variants), rtpt_present does the work as before — the calling sequence is the
 * same either way.  The registration stays until changed; dst_device == NULL clears it. */
int rtpt_present_target(rtpt_ctx* ctx, void* dst_device, uint32_t y0, uint32_t y1);

/* ---- synchronisation / data movement ---------------------------------------------------- */
int rtpt_sync(rtpt_ctx* ctx);
/* blocking copy of a whole plane (stored rows) to host memory */
int rtpt_readback(rtpt_ctx* ctx, rtpt_plane which, void* dst, size_t bytes);
/* blocking upload of a whole plane from host memory (inject fixtures / history; SURVEY 5
 * "checkpoint/resume": the only cross-frame state is PREVIOUS, PREV_VIS_ID, LUT_PREV) */
int rtpt_set_plane(rtpt_ctx* ctx, rtpt_plane which, const void* src, size_t bytes);
int rtpt_reset_counters(rtpt_ctx* ctx);
/* only closest-hit queries of pixels in frame rows [y0,y1) are added to RAYCOUNT (default: all
 * stored rows).  Strip ranks that trace halo rows redundantly set this to their owned rows so the
 * sum over ranks equals the single-GPU count. */
int rtpt_set_count_rows(rtpt_ctx* ctx, uint32_t y0, uint32_t y1);
/* enable build-only observables (off by default: they cost extra stores per pixel) */
#define RTPT_DEBUG_HIT_ID 0x1u     /* RTPT_PLANE_HIT_ID written by rtpt_raytrace */
#define RTPT_DEBUG_PREV_PIXEL 0x2u /* RTPT_PLANE_PREV_PIXEL written by the final filter pass */
int rtpt_enable_debug(rtpt_ctx* ctx, uint32_t mask);

/* per-kernel timing hooks for bench.py: HIP events recorded on the stream the kernel runs on.
 * rtpt_timing_enable(n), n >= 1, makes the passes of every n-th frame (frames are counted by
 * rtpt_end_frame) record a start/stop event pair — the pairs cost ~6 % of a 1 ms frame when every
 * launch is bracketed, so long runs sample; 0 turns it off.  rtpt_timing_collect blocks, sums the
 * durations per kernel since the last collect and returns them. */
typedef enum rtpt_kernel_id {
  RTPT_K_GBUFFER = 0,
  RTPT_K_LUT = 1,
  RTPT_K_GRADIENT = 2,
  RTPT_K_PATHTRACE = 3,
  RTPT_K_ATROUS = 4,
  RTPT_K_ATROUS_FINAL = 5,
  RTPT_K_ATROUS_CHAIN = 6,       /* several consecutive iterations k < N in one launch (intermediates in LDS) */
  RTPT_K_ATROUS_CHAIN_FINAL = 7, /* ... ending in the final pass */
  RTPT_K_GBUFFER_GRADIENT = 8,   /* K0 and K1 in one launch (rtpt_temporal_gradient right behind rtpt_gbuffer) */
  RTPT_K_PRESENT = 9,            /* rtpt_present: RGBA32F -> B8G8R8A8_UNORM */
  RTPT_K_GBUFFER_PATHTRACE = 10, /* K0, K1 and K2 in one launch (rtpt_raytrace right behind rtpt_gbuffer + rtpt_temporal_gradient,
                                    the reference's own order, main.cpp:1105-1107): ABI version 4 */
  RTPT_K_COUNT = 11
} rtpt_kernel_id;
int rtpt_timing_enable(rtpt_ctx* ctx, int enable);
int rtpt_timing_collect(rtpt_ctx* ctx, double ms_sum[RTPT_K_COUNT], uint32_t launches[RTPT_K_COUNT]);
const char* rtpt_kernel_name(rtpt_kernel_id k);

/* device-side evaluation of the deterministic math used on bit-exact paths, for parity tests
 * against the oracle: op 0 log, 1 sin(2*pi*u), 2 cos(2*pi*u), 3 sqrt, 4 1/x, 5 exp (filter
 * fast path), 6 pcg step float.  in/out are host arrays of n floats (u32 bits for op 6). */
int rtpt_selftest_math(rtpt_ctx* ctx, int op, const float* in, float* out, size_t n);
/* the product's correctly-rounded sqrt (op 3) and reciprocal (op 4) are shorter instruction sequences than the compiler's
 * IEEE expansions (csrc/rtpt_math.hpp): this runs ALL 2^32 binary32 patterns through both on the device and returns how
 * many results differ in any bit (the contract is 0) and the first few offending patterns.  ~0.1 s. */
int rtpt_selftest_exhaustive(rtpt_ctx* ctx, int op, uint64_t* mismatches, uint32_t first_bad[4]);
/* the product's division (csrc/rtpt_math.hpp, exact::div_) against the compiler's IEEE division, on the device.  mode 0:
 * passes [first_pass, first_pass + n_passes) of the 256-pass enumeration of ALL 2^23 x 2^23 pairs of binary32 significands
 * (~0.15 s per pass; all 256 is the proof that the short sequence is correctly rounded for operands of ordinary magnitude);
 * mode 1: n_passes x 2^33 operand pairs of arbitrary bits, pass numbers seeding the generator (range test + long path).
 * *mismatches = results that differ in any bit (two NaNs count as equal); first_bad = the bits of one offending (a, b). */
int rtpt_selftest_div(rtpt_ctx* ctx, int mode, uint32_t first_pass, uint32_t n_passes, uint64_t* mismatches, uint32_t first_bad[2]);
/* closest-hit of arbitrary rays through the product's traversal (parity vs the oracle's brute
 * force): rays = n x {ox,oy,oz,dx,dy,dz}; out_id[n] = primitive id+1 or 0; out_t[n] may be NULL */
int rtpt_selftest_trace(rtpt_ctx* ctx, const float* rays, size_t n, uint32_t* out_id, float* out_t);

/* ---- host-side helpers shared by the C++ and Python hosts -------------------------------- */
/* glm::lookAt / glm::perspective as used at main.cpp:482-484,:1470-1472 (right-handed,
 * zero-to-one depth — D6; the caller applies proj[1][1] *= -1 like the reference does). */
void rtpt_util_look_at(const float eye[3], const float center[3], const float up[3], float out[16]);
void rtpt_util_perspective(float fovy, float aspect, float z_near, float z_far, float out[16]);
/* minimal OBJ reader standing in for tinyobjloader at main.cpp:416-428: `v` and `f` records,
 * polygons fan-triangulated (0,1,2),(0,2,3) in file order (D5).  Two-call pattern: pass NULL
 * arrays to obtain counts. */
int rtpt_util_load_obj(const char* path, float* xyz, uint32_t* n_verts, uint32_t* idx, uint32_t* n_tris);
/* the material side of the same file: `mtllib` (looked up next to the OBJ), `usemtl`, and Kd / Ke of every `newmtl`.
 * tri_material lines up with rtpt_util_load_obj's triangles; material 0 is the default (Kd 0.7, Ke 0).  Two-call
 * pattern (NULL arrays: counts; *n_materials in = capacity).  A missing library is not an error — the reference's own
 * OBJ names one that does not exist (scenes/CornellBox-Original-Merged.obj:3) — *n_materials comes back 0. */
int rtpt_util_load_obj_materials(const char* path, uint32_t* tri_material, uint32_t* n_tris, rtpt_material* materials,
                                 uint32_t* n_materials);
/* Host-only self check of the acceleration-structure builder that stands in for the driver's BLAS/TLAS build
 * (buildAccelerationStructure, main.cpp:687-742): builds the BVH over `n_tris` world-space triangles (9 floats
 * each), packs the device nodes and verifies the invariants the traversal relies on.  Needs no GPU.
 *   stats[0] nodes, [1] leaves, [2] max depth, [3] largest leaf,
 *   [4] triangles not referenced exactly once, [5] boxes that do not contain their subtree,
 *   [6] device (16-bit grid) boxes that do not contain the binary32 box, [7] dangling child references */
int rtpt_util_bvh_check(const float* tris, uint32_t n_tris, uint64_t stats[8]);
/* The structure as it stands ON THE DEVICE of a context — after rtpt_scene_upload, or after a changed ubo->model re-posed
 * the scene and refit the tree inside rtpt_gbuffer (on the device, on the context's stream, without a host
 * synchronisation: refit.hip) — read back and checked on the host (blocks).
 *   stats[0] nodes, [1] leaves, [2] deepest level, [3] largest leaf, [4] triangles not referenced exactly once,
 *   [5] decoded device boxes that do not contain every vertex below them, [6] boxes reaching beyond the padded scene,
 *   [7] dangling child references */
int rtpt_debug_bvh_check(rtpt_ctx* ctx, uint64_t stats[8]);
/* the same invariants after a REFIT: the tree is built over `built_for` and refit to `moved` (the same n_tris
 * triangles after an animated model matrix, rtpt_gbuffer) — topology and leaf order kept, boxes recomputed */
int rtpt_util_bvh_refit_check(const float* built_for, const float* moved, uint32_t n_tris, uint64_t stats[8]);

#ifdef __cplusplus
}
static_assert(sizeof(rtpt_push_constants) == 112, "PushConstants is 112 bytes (main.cpp:35-49)");
static_assert(offsetof(rtpt_push_constants, cameraPos) == 16, "cameraPos@16");
static_assert(offsetof(rtpt_push_constants, lightPos) == 32, "lightPos@32");
static_assert(offsetof(rtpt_push_constants, lightPosPrev) == 48, "lightPosPrev@48");
static_assert(offsetof(rtpt_push_constants, currentCameraColor) == 64, "currentCameraColor@64");
static_assert(offsetof(rtpt_push_constants, previousCameraColor) == 80, "previousCameraColor@80");
static_assert(offsetof(rtpt_push_constants, waveletIteration) == 92, "waveletIteration@92");
static_assert(offsetof(rtpt_push_constants, maxWaveletIteration) == 96, "maxWaveletIteration@96");
static_assert(sizeof(rtpt_ubo) == 384, "UniformBufferObject is 384 bytes (main.cpp:82-90)");
static_assert(sizeof(rtpt_visibility_data) == 48, "VisibilityData stride 48 (std430)");
#endif

#endif /* RTPT_H */
