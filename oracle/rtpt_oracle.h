/*
 * ORACLE — TEST INFRASTRUCTURE ONLY (see det_math.h).  CPU scalar restatement of the
 * reference's per-frame compute chain, one function per shader, each citing the GLSL it
 * follows.  PARITY UNPINNED by reference fixtures: the reference has no tests, golden images or
 * runnable build in this environment; this oracle is pinned by the
 * known-answer vectors derivable from the reference sources (tests/test_oracle_kat.py) and by
 * line-by-line citation.
 */
#ifndef RTPT_ORACLE_H
#define RTPT_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct oracle_config {
  uint32_t width, height;        /* main.cpp:52-53 */
  uint32_t max_segments;         /* raytrace.comp.glsl:204 */
  uint32_t samples_per_pixel;    /* raytrace.comp.glsl:306 */
  int32_t sigma_n;               /* temporalFiltering.comp.glsl:203 */
  float sigma_z, sigma_l;        /* :204-205 */
  float alpha;                   /* :243 */
  float light_radius;            /* raytrace.comp.glsl:280 */
  float light_intensity;         /* :281 */
  float first_hit_light_divisor; /* :229 */
  float fov_slope;               /* :300 tan(FOV) */
  float pixel_jitter;            /* :314 */
  float ray_offset;              /* :250 */
  float ray_tmax;                /* :216 */
  uint32_t ext_flags;            /* opt-in extension modes (not reference behaviour), see ORACLE_EXT_* */
} oracle_config;

/* Extension modes = the pieces of the textbook A-SVGF that sit unused or commented out in the reference's
 * own sources.  Default off; they cannot be parity-checked against the reference. */
#define ORACLE_EXT_ADAPTIVE_ALPHA 0x10u /* temporalFiltering.comp.glsl:247-248 (commented out): alpha = (1-g)*alpha + g */
#define ORACLE_EXT_GAUSS5 0x20u         /* gaussianKernel2D, temporalFiltering.comp.glsl:93-99 (declared, unused): 5x5 taps */
#define ORACLE_EXT_POW2_STRIDE 0x40u    /* tap stride 2^(k-1) instead of k (:135) */
#define ORACLE_EXT_VARIANCE 0x100u      /* SVGF-style luminance moments: temporally accumulated first and second moments
                                           of the traced luminance give a per-pixel variance that scales the colour
                                           edge-stopping term and is filtered along (see oracle_moments) */
#define ORACLE_EXT_SVGF_VARIANCE 0x800u /* with ORACLE_EXT_VARIANCE, the two pieces of SVGF's variance handling (Schied et al. 2017)
                                          the flag above leaves out: (i) a pixel whose moment history is shorter than 4 frames
                                          takes its variance from the 7x7 neighbourhood of the CURRENT frame's luminance (taps on
                                          the same primitive only) instead of the temporal one, still scaled by 4/n; (ii) the
                                          variance that scales an iteration's luminance weight is the 3x3 Gaussian
                                          (1 2 1 / 2 4 2 / 1 2 1) / 16 of the variance plane around the pixel, not its own value */
#define ORACLE_EXT_DISOCCLUSION 0x80u   /* previousVisibilityBuffer (main.cpp:375,:1367: copied every frame, never read):
                                           history is used only where the reprojected pixel showed the same primitive */

/* PushConstants, main.cpp:35-49 (112 bytes, same layout as the shaders) */
typedef struct oracle_push_constants {
  uint32_t sample_batch, frameNumber, _pad0[2];
  float cameraPos[3], _pad1;
  float lightPos[3], _pad2;
  float lightPosPrev[3], _pad3;
  float currentCameraColor[3], _pad4;
  float previousCameraColor[3];
  int32_t waveletIteration;
  int32_t maxWaveletIteration;
  uint32_t _pad5[3];
} oracle_push_constants;

/* UniformBufferObject, main.cpp:82-90: six column-major mat4 */
typedef struct oracle_ubo {
  float model[16], view[16], proj[16], modelPrev[16], viewPrev[16], projPrev[16];
} oracle_ubo;

void oracle_config_default(oracle_config* cfg, uint32_t width, uint32_t height);
/* number of worker threads for the row loops (plain pthreads, created and joined per call); 1 = scalar */
void oracle_set_threads(int n);
int oracle_get_threads(void);
/* restrict the per-pixel passes to columns [x0, x1) (x1 <= x0: every column again); timing samples only */
void oracle_set_columns(int x0, int x1);
/* test processes only: print the faulting NATIVE thread's frames on SIGSEGV/SIGBUS/SIGABRT, then chain */
void oracle_install_crash_trace(void);

/* --- numerics contract, exported for bit-exact comparison with the device ---------------- */
float oracle_log(float x);
float oracle_sin2pi(float u);
float oracle_cos2pi(float u);
float oracle_exp(float x);
float oracle_sqrt(float x);
float oracle_rcp(float x);
float oracle_powi(float x, int n);
void oracle_math_array(int op, const float* in, float* out, uint64_t n);

/* --- RNG (raytrace.comp.glsl:71-78, :297) -------------------------------------------------- */
uint32_t oracle_rng_seed(uint32_t px, uint32_t py, uint32_t frame, uint32_t batch);
/* steps the state, returns the output word; *f receives the float in [0,1] */
uint32_t oracle_rng_step(uint32_t* state, float* f);

/* --- host helpers (glm::lookAt / glm::perspective, zero-to-one depth, D6) ------------------ */
void oracle_look_at(const float eye[3], const float center[3], const float up[3], float out[16]);
void oracle_perspective(float fovy, float aspect, float z_near, float z_far, float out[16]);

/* --- scene --------------------------------------------------------------------------------- */
/* OBJ reader: `v` / `f` records, fan triangulation (0,1,2),(0,2,3) in file order (D5).
 * Pass NULL arrays to get the counts. */
int oracle_load_obj(const char* path, float* xyz, uint32_t* n_verts, uint32_t* idx, uint32_t* n_tris);
/* world-space triangle soup: tris[9*(inst*n_tris+t)] = xform_inst * (v0,v1,v2); xforms 3x4
 * row-major, NULL/0 = one identity instance (main.cpp:728-741) */
void oracle_flatten(const float* xyz, const uint32_t* idx, uint32_t n_tris, const float* xforms,
                    uint32_t n_inst, float* tris);
/* closest hit (D4: min over (t, id)) by brute force over all triangles: returns id+1 or 0 */
uint32_t oracle_closest_hit(const float* tris, uint32_t n, const float o[3], const float d[3],
                            float tmax, float* t_out, float* b1_out, float* b2_out);
void oracle_trace_rays(const float* tris, uint32_t n, const float* rays, uint64_t n_rays, float tmax,
                       uint32_t* out_id, float* out_t);

/* --- passes.  All image arrays are full-frame, index y*width+x; rows [y0,y1) are computed --- */
/* visibility.geom.glsl:44-59: lut has (n+1)*12 floats (stride 48 B); slot 0 is zeroed */
void oracle_lut(const float* tris, uint32_t n, const float model[16], float* lut);
/* K0: visibility.{vert,geom,frag}.glsl as pixel-centre primary rays */
void oracle_gbuffer(const oracle_config* cfg, const float* tris, uint32_t n, const oracle_ubo* ubo,
                    uint32_t y0, uint32_t y1, uint32_t* vis, float* worldpos, float* depth);
/* K1: temporalGradient.comp.glsl:104-172 */
void oracle_temporal_gradient(const oracle_config* cfg, const oracle_push_constants* pc,
                              const uint32_t* vis, const float* worldpos, const float* lut,
                              const float* lut_prev, uint32_t y0, uint32_t y1, float* grad);
/* K2: raytrace.comp.glsl:273-344.  raycount accumulates closest-hit queries; hit_id (nullable)
 * receives the primary ray's primitive id+1 */
void oracle_raytrace(const oracle_config* cfg, const oracle_push_constants* pc, const float* tris,
                     uint32_t n, uint32_t y0, uint32_t y1, float* image, uint64_t* raycount,
                     uint32_t* hit_id);
void oracle_raytrace_mat(const oracle_config* cfg, const oracle_push_constants* pc, const float* tris,
                         uint32_t n, const float* tri_mat, uint32_t n_base, uint32_t y0, uint32_t y1, float* image,
                         uint64_t* raycount, uint32_t* hit_id);
/* K3: temporalFiltering.comp.glsl:191-265, one iteration.  `in` is the colorImage snapshot (D1),
 * `out` receives the filtered colour (k < max) or the blend (k == max).  prev_pixel (nullable,
 * 2 ints per pixel) receives previousPixelPos when k == max. */
void oracle_atrous(const oracle_config* cfg, const oracle_push_constants* pc, const oracle_ubo* ubo,
                   const float* in, const float* depth, const uint32_t* vis, const float* lut,
                   const float* lut_prev, const float* worldpos, const float* history,
                   uint32_t y0, uint32_t y1, float* out, int32_t* prev_pixel);
/* same with the extension inputs: gradient plane (K1 output) and the previous frame's id plane (nullable
 * unless the corresponding ORACLE_EXT_* flag is set) */
void oracle_atrous_ext(const oracle_config* cfg, const oracle_push_constants* pc, const oracle_ubo* ubo,
                       const float* in, const float* depth, const uint32_t* vis, const float* lut,
                       const float* lut_prev, const float* worldpos, const float* history,
                       const float* gradient, const uint32_t* prev_vis,
                       uint32_t y0, uint32_t y1, float* out, int32_t* prev_pixel);

/* Extension ORACLE_EXT_VARIANCE, not reference behaviour.  Before the first filter iteration: per pixel
 *   lum = 0.2126 r + 0.7152 g + 0.0722 b of the traced colour; q = the reprojected pixel (same arithmetic as the final
 *   pass); valid = frame > 0, q inside the image and prev_vis[q] == id;
 *   a = valid ? max(alpha, 1/(n_prev[q]+1)) : 1;  m1 = mix(m1_prev[q], lum, a);  m2 = mix(m2_prev[q], lum^2, a);
 *   n = valid ? min(n_prev[q]+1, 255) : 1;  var = max(0, m2 - m1^2) * (n < 4 ? 4/n : 1).
 * moments planes are 4 floats per pixel (m1, m2, n, var); var_out 1 float per pixel. */
void oracle_moments(const oracle_config* cfg, const oracle_push_constants* pc, const oracle_ubo* ubo,
                    const float* traced, const uint32_t* vis, const float* worldpos, const float* lut_prev,
                    const uint32_t* prev_vis, const float* moments_prev, uint32_t y0, uint32_t y1,
                    float* moments_out, float* var_out);
/* oracle_atrous_ext with the variance planes: with ORACLE_EXT_VARIANCE the colour term of the weight becomes
 *   exp(-|lum_p - lum_q| / (sigma_l * sqrt(var_in[p]) + 1e-4))
 * and var_out[p] = sum((h w)^2 var_in[q]) / (sum(h w))^2 is filtered along.  var_in / var_out may be NULL otherwise. */
/* ORACLE_EXT_SVGF_VARIANCE (ii): out[p] = sum_{dy,dx in -1..1} g[dy][dx] var[clamp(p + (dx,dy))] / 16, dy outer, dx inner, fma
 * accumulation from 0, then * 0.0625 */
void oracle_var_prefilter(const oracle_config* cfg, const float* var, uint32_t y0, uint32_t y1, float* out);
void oracle_atrous_var(const oracle_config* cfg, const oracle_push_constants* pc, const oracle_ubo* ubo,
                       const float* in, const float* depth, const uint32_t* vis, const float* lut,
                       const float* lut_prev, const float* worldpos, const float* history,
                       const float* gradient, const uint32_t* prev_vis, const float* var_in,
                       uint32_t y0, uint32_t y1, float* out, int32_t* prev_pixel, float* var_out);

#ifdef __cplusplus
}
#endif
#endif
