/*
 * ORACLE — TEST INFRASTRUCTURE ONLY.  See rtpt_oracle.h / det_math.h for the contract.
 * Plain C scalar restatement of the reference's compute chain; citations are file:line
 * relative to the reference tree (shaders/ prefix omitted for *.glsl).
 */
#define _GNU_SOURCE /* gettid, backtrace */
#include "rtpt_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "det_math.h"

#include <execinfo.h>
#include <pthread.h>
#include <signal.h>
#include <sys/prctl.h>
#include <unistd.h>

/* Row-parallel driver.  Plain pthreads, created and joined inside every call: no OpenMP runtime, no pool that outlives
 * a call or is resized between calls, nothing shared with the process the checker is loaded into (pytest/bench.py
 * processes also hold torch's own libgomp image and the HIP runtime's threads; DESIGN.md 2, "The round-2 abort").
 * Work is handed out in chunks from one atomic counter; every chunk runs the SAME serial row function the one-thread
 * path runs, so the thread count cannot change a result. */
static int g_threads = 1;
void oracle_set_threads(int n) { g_threads = n < 1 ? 1 : (n > 256 ? 256 : n); }
/* Column window of the per-pixel passes (G-buffer, gradient, trace, moments, a-trous): pixels outside [x0, x1) are left
 * untouched.  Default: every column.  bench.py's cpu_baseline uses it to time a bounded pixel sample of BASELINE configs[4],
 * where one closest-hit query is a brute force over 1,152,000 triangles and a whole row would take minutes. */
static int g_col0 = 0, g_col1 = 0x7fffffff;
void oracle_set_columns(int x0, int x1) {
  g_col0 = x0 < 0 ? 0 : x0;
  g_col1 = x1 <= g_col0 ? 0x7fffffff : x1;
}
static inline int col_lo(void) { return g_col0; }
static inline int col_hi(int w) { return g_col1 < w ? g_col1 : w; }
int oracle_get_threads(void) { return g_threads; }

/* Crash attribution for the test processes (tests/conftest.py installs it for -m gpu runs): Python's faulthandler
 * prints Python frames only, and round 2's abort was in a thread that had none.  On SIGSEGV / SIGBUS /
 * SIGABRT this prints the faulting thread's name, the faulting address and its NATIVE frames (module + offset) to
 * stderr, then hands the signal to whatever handler was installed before (faulthandler's). */
static struct sigaction g_prev_sa[3];
static const int g_crash_sigs[3] = {SIGSEGV, SIGBUS, SIGABRT};

static void crash_trace_handler(int sig, siginfo_t* si, void* uc) {
  char name[32] = "?";
  prctl(PR_GET_NAME, name, 0, 0, 0);
  char msg[160];
  int n = snprintf(msg, sizeof msg, "\n[oracle crash trace] signal %d at address %p in native thread '%s' (tid %ld):\n", sig,
                   si ? si->si_addr : NULL, name, (long)gettid());
  if (n > 0) (void)!write(2, msg, (size_t)n);
  void* frames[64];
  int nf = backtrace(frames, 64);
  backtrace_symbols_fd(frames, nf, 2);
  for (int i = 0; i < 3; i++)
    if (g_crash_sigs[i] == sig) {
      sigaction(sig, &g_prev_sa[i], NULL); /* restore and re-deliver to the previous handler */
      raise(sig);
      return;
    }
}

void oracle_install_crash_trace(void) {
  void* warm[4];
  (void)backtrace(warm, 4); /* loads libgcc's unwinder now, not inside the handler */
  struct sigaction sa;
  memset(&sa, 0, sizeof sa);
  sa.sa_sigaction = crash_trace_handler;
  sa.sa_flags = SA_SIGINFO | SA_ONSTACK | SA_NODEFER;
  sigemptyset(&sa.sa_mask);
  for (int i = 0; i < 3; i++) sigaction(g_crash_sigs[i], &sa, &g_prev_sa[i]);
}

typedef void (*range_fn)(void* ctx, int64_t a, int64_t b);
typedef struct {
  range_fn fn;
  void* ctx;
  int64_t end, chunk;
  int64_t next; /* atomic */
} range_job;

static void* range_worker(void* p) {
  range_job* j = (range_job*)p;
  for (;;) {
    int64_t a = __atomic_fetch_add(&j->next, j->chunk, __ATOMIC_RELAXED);
    if (a >= j->end) break;
    int64_t b = a + j->chunk < j->end ? a + j->chunk : j->end;
    j->fn(j->ctx, a, b);
  }
  return NULL;
}

static void run_ranges(int64_t begin, int64_t end, int64_t chunk, range_fn fn, void* ctx) {
  if (end <= begin) return;
  if ((end - begin + chunk - 1) / chunk < g_threads) chunk = 1; /* few rows: one per thread (chunking never changes a result) */
  int64_t n_chunks = (end - begin + chunk - 1) / chunk;
  int nt = g_threads < n_chunks ? g_threads : (int)n_chunks;
  if (nt <= 1) {
    fn(ctx, begin, end);
    return;
  }
  range_job job = {fn, ctx, end, chunk, begin};
  pthread_t th[256];
  pthread_attr_t at;
  pthread_attr_init(&at);
  pthread_attr_setstacksize(&at, 1 << 20); /* the row functions keep a few hundred bytes of locals */
  int started = 0;
  for (int i = 0; i < nt - 1; i++)
    if (pthread_create(&th[started], &at, range_worker, &job) == 0) started++; /* a failed create only lowers the width */
  pthread_attr_destroy(&at);
  range_worker(&job); /* the calling thread works too */
  for (int i = 0; i < started; i++) pthread_join(th[i], NULL);
}

void oracle_config_default(oracle_config* c, uint32_t w, uint32_t h) {
  memset(c, 0, sizeof(*c));
  c->width = w;
  c->height = h;
  c->max_segments = 32;              /* raytrace.comp.glsl:204 */
  c->samples_per_pixel = 1;          /* raytrace.comp.glsl:306 */
  c->sigma_n = 128;                  /* temporalFiltering.comp.glsl:203 */
  c->sigma_z = 1.0f;                 /* :204 */
  c->sigma_l = 4.0f;                 /* :205 */
  c->alpha = 0.3f;                   /* :243 */
  c->light_radius = 0.20f;           /* raytrace.comp.glsl:280 */
  c->light_intensity = 30.0f;        /* :281 */
  c->first_hit_light_divisor = 5.0f; /* :229 */
  c->fov_slope = 0.20271003f;        /* tan(FOV=0.20), common.h:16; constant folded in the .spv */
  c->pixel_jitter = 0.375f;          /* :314 */
  c->ray_offset = 0.0001f;           /* :250 */
  c->ray_tmax = 10000.0f;            /* :216 */
  c->ext_flags = 0;
}

/* ---------------------------------------------------------------- numerics exports */
float oracle_log(float x) { return dm_log(x); }
float oracle_sin2pi(float u) { float s, c; dm_sincos2pi(u, &s, &c); return s; }
float oracle_cos2pi(float u) { float s, c; dm_sincos2pi(u, &s, &c); return c; }
float oracle_exp(float x) { return dm_exp(x); }
float oracle_sqrt(float x) { return dm_sqrt(x); }
float oracle_rcp(float x) { return 1.0f / x; }
float oracle_powi(float x, int n) { return dm_powi(x, n); }

uint32_t oracle_rng_seed(uint32_t px, uint32_t py, uint32_t frame, uint32_t batch) {
  /* raytrace.comp.glsl:297 — uint32 wrap-around arithmetic */
  return (px * 3266489917u + py * 668265263u) ^ (frame * 374761393u) ^ (batch * 2654435761u);
}

uint32_t oracle_rng_step(uint32_t* state, float* f) {
  /* raytrace.comp.glsl:71-78 (pcg_output_rxs_m_xs_32_32) */
  uint32_t s = *state * 747796405u + 1u;
  *state = s;
  uint32_t word = ((s >> ((s >> 28) + 4u)) ^ s) * 277803737u;
  word = (word >> 22) ^ word;
  /* :77 — 4294967295.0f rounds to 2^32 in binary32, so this is float(word) * 2^-32 exactly */
  if (f) *f = (float)word / 4294967295.0f;
  return word;
}

static inline float rng_float(uint32_t* state) {
  float f;
  oracle_rng_step(state, &f);
  return f;
}

void oracle_math_array(int op, const float* in, float* out, uint64_t n) {
  for (uint64_t i = 0; i < n; i++) {
    float x = in[i], r;
    switch (op) {
      case 0: r = dm_log(x); break;
      case 1: r = oracle_sin2pi(x); break;
      case 2: r = oracle_cos2pi(x); break;
      case 3: r = dm_sqrt(x); break;
      case 4: r = 1.0f / x; break;
      case 5: r = dm_exp(x); break;
      case 6: { uint32_t s = dm_bits(x); r = rng_float(&s); break; }
      default: r = 0.0f;
    }
    out[i] = r;
  }
}

/* ---------------------------------------------------------------- host helpers */
void oracle_look_at(const float eye[3], const float center[3], const float up[3], float m[16]) {
  /* glm::lookAtRH as used at main.cpp:482,:1470 (host arithmetic, plain float ops) */
  vec3 e = v3(eye[0], eye[1], eye[2]);
  vec3 f = v3_normalize(v3_sub(v3(center[0], center[1], center[2]), e));
  vec3 s = v3_normalize(v3_cross(f, v3(up[0], up[1], up[2])));
  vec3 u = v3_cross(s, f);
  memset(m, 0, 16 * sizeof(float));
  m[0] = s.x; m[4] = s.y; m[8] = s.z;
  m[1] = u.x; m[5] = u.y; m[9] = u.z;
  m[2] = -f.x; m[6] = -f.y; m[10] = -f.z;
  m[12] = -v3_dot(s, e);
  m[13] = -v3_dot(u, e);
  m[14] = v3_dot(f, e);
  m[15] = 1.0f;
}

void oracle_perspective(float fovy, float aspect, float zn, float zf, float m[16]) {
  /* glm::perspectiveRH_ZO (D6), main.cpp:483,:1471; the caller flips m[5] (main.cpp:484) */
  float t = (float)tan((double)fovy * 0.5);
  memset(m, 0, 16 * sizeof(float));
  m[0] = 1.0f / (aspect * t);
  m[5] = 1.0f / t;
  m[10] = zf / (zn - zf);
  m[11] = -1.0f;
  m[14] = -(zf * zn) / (zf - zn);
}

/* C = A * B, column-major; C[c][r] = fma(a3,b3, fma(a2,b2, fma(a1,b1, a0*b0))) */
static void mat4_mul(const float* A, const float* B, float* C) {
  for (int c = 0; c < 4; c++)
    for (int r = 0; r < 4; r++) {
      float acc = A[0 * 4 + r] * B[c * 4 + 0];
      acc = dm_fma(A[1 * 4 + r], B[c * 4 + 1], acc);
      acc = dm_fma(A[2 * 4 + r], B[c * 4 + 2], acc);
      acc = dm_fma(A[3 * 4 + r], B[c * 4 + 3], acc);
      C[c * 4 + r] = acc;
    }
}

/* (M * vec4(p,1))[i] = fma(m2i,z, fma(m1i,y, m0i*x)) + m3i */
static inline float mat4_row_point(const float* M, int i, vec3 p) {
  return dm_fma(M[8 + i], p.z, dm_fma(M[4 + i], p.y, M[i] * p.x)) + M[12 + i];
}

/* ---------------------------------------------------------------- scene */
int oracle_load_obj(const char* path, float* xyz, uint32_t* n_verts, uint32_t* idx, uint32_t* n_tris) {
  FILE* fp = fopen(path, "r");
  if (!fp) return -1;
  char line[1024];
  uint32_t nv = 0, nt = 0;
  while (fgets(line, sizeof line, fp)) {
    if (line[0] == 'v' && (line[1] == ' ' || line[1] == '\t')) {
      float x, y, z;
      if (sscanf(line + 2, "%f %f %f", &x, &y, &z) == 3) {
        if (xyz) { xyz[3 * nv] = x; xyz[3 * nv + 1] = y; xyz[3 * nv + 2] = z; }
        nv++;
      }
    } else if (line[0] == 'f' && (line[1] == ' ' || line[1] == '\t')) {
      long poly[64];
      int np = 0;
      char* p = line + 2;
      while (*p && np < 64) {
        while (*p == ' ' || *p == '\t') p++;
        if (*p == '\0' || *p == '\n' || *p == '\r') break;
        char* end;
        long v = strtol(p, &end, 10);
        if (end == p) break;
        poly[np++] = v > 0 ? v - 1 : (long)nv + v; /* OBJ: 1-based, negative = relative */
        p = end;
        while (*p && *p != ' ' && *p != '\t' && *p != '\n' && *p != '\r') p++; /* skip /vt/vn */
      }
      for (int k = 1; k + 1 < np; k++) { /* D5: fan (0,k,k+1) in file order */
        if (idx) { idx[3 * nt] = (uint32_t)poly[0]; idx[3 * nt + 1] = (uint32_t)poly[k]; idx[3 * nt + 2] = (uint32_t)poly[k + 1]; }
        nt++;
      }
    }
  }
  fclose(fp);
  *n_verts = nv;
  *n_tris = nt;
  return 0;
}

void oracle_flatten(const float* xyz, const uint32_t* idx, uint32_t n_tris, const float* xf,
                    uint32_t n_inst, float* tris) {
  uint32_t ni = (xf && n_inst) ? n_inst : 1;
  for (uint32_t inst = 0; inst < ni; inst++)
    for (uint32_t t = 0; t < n_tris; t++)
      for (int k = 0; k < 3; k++) {
        const float* v = xyz + 3 * idx[3 * t + k];
        float* o = tris + 9 * ((uint64_t)inst * n_tris + t) + 3 * k;
        if (xf && n_inst) {
          const float* m = xf + 12 * inst; /* 3x4 row-major */
          for (int r = 0; r < 3; r++)
            o[r] = dm_fma(m[4 * r + 2], v[2], dm_fma(m[4 * r + 1], v[1], m[4 * r] * v[0])) + m[4 * r + 3];
        } else {
          o[0] = v[0]; o[1] = v[1]; o[2] = v[2];
        }
      }
}

/* One ray-triangle routine shared by every closest-hit query (D4).  Scalar-triple-product form of
 * Moller-Trumbore with the plane normal n = e1 x e2 precomputed per triangle and the division
 * deferred:  det = -d.n,  tt = (o-v0).n,  c = (o-v0) x d,  u = e2.c,  v = -e1.c
 * (e1.(d x e2) = -d.n,  tv.(d x e2) = e2.(tv x d),  d.(tv x e1) = -e1.(tv x d),  e2.(tv x e1) = tv.n).
 * Returns 1 and (t, b1, b2) when the ray o + t d, 0 < t < tmax, meets the triangle (both faces;
 * raytrace.comp.glsl:209-216: opaque, no culling, tmin 0, tmax 1e4). */
static inline int tri_hit(vec3 o, vec3 d, const float* tri, float tmax, float* t, float* b1, float* b2) {
  vec3 v0 = v3(tri[0], tri[1], tri[2]);
  vec3 e1 = v3(tri[3] - tri[0], tri[4] - tri[1], tri[5] - tri[2]);
  vec3 e2 = v3(tri[6] - tri[0], tri[7] - tri[1], tri[8] - tri[2]);
  vec3 n = v3_cross(e1, e2);
  vec3 tv = v3_sub(o, v0);
  float det = -v3_dot(d, n);
  float tt = v3_dot(tv, n);
  vec3 c = v3_cross(tv, d);
  float u = v3_dot(e2, c);
  float v = -v3_dot(e1, c);
  float ad = fabsf(det);
  if (!(ad > 0.0f)) return 0; /* det == 0 or NaN */
  if (det < 0.0f) { u = -u; v = -v; tt = -tt; }
  if (!(u >= 0.0f) || !(v >= 0.0f) || !(u + v <= ad) || !(tt > 0.0f)) return 0;
  float th = tt / ad;
  if (!(th < tmax)) return 0;
  *t = th;
  *b1 = u / ad;
  *b2 = v / ad;
  return 1;
}

uint32_t oracle_closest_hit(const float* tris, uint32_t n, const float o_[3], const float d_[3],
                            float tmax, float* t_out, float* b1_out, float* b2_out) {
  vec3 o = v3(o_[0], o_[1], o_[2]), d = v3(d_[0], d_[1], d_[2]);
  uint32_t best = 0;
  float bt = 0.f, bb1 = 0.f, bb2 = 0.f;
  for (uint32_t i = 0; i < n; i++) {
    float t, b1, b2;
    if (tri_hit(o, d, tris + 9 * (uint64_t)i, tmax, &t, &b1, &b2)) {
      if (best == 0 || t < bt) { best = i + 1; bt = t; bb1 = b1; bb2 = b2; } /* ascending id: ties keep lower id */
    }
  }
  if (best) {
    if (t_out) *t_out = bt;
    if (b1_out) *b1_out = bb1;
    if (b2_out) *b2_out = bb2;
  }
  return best;
}

typedef struct {
  const float* tris;
  uint32_t n;
  const float* rays;
  float tmax;
  uint32_t* out_id;
  float* out_t;
} trace_rays_args;

static void trace_rays_range(void* ctx, int64_t i0, int64_t i1) {
  const trace_rays_args* A = (const trace_rays_args*)ctx;
  const float *tris = A->tris, *rays = A->rays;
  const uint32_t n = A->n;
  const float tmax = A->tmax;
  uint32_t* out_id = A->out_id;
  float* out_t = A->out_t;
  for (int64_t i = i0; i < i1; i++) {
    float t = 0.f;
    out_id[i] = oracle_closest_hit(tris, n, rays + 6 * i, rays + 6 * i + 3, tmax, &t, NULL, NULL);
    if (out_t) out_t[i] = out_id[i] ? t : 0.0f;
  }
}

void oracle_trace_rays(const float* tris, uint32_t n, const float* rays, uint64_t n_rays, float tmax,
                       uint32_t* out_id, float* out_t) {
  trace_rays_args A = {tris, n, rays, tmax, out_id, out_t};
  run_ranges(0, (int64_t)n_rays, 64, trace_rays_range, &A);
}

/* v0*b.x + v1*b.y + v2*b.z (raytrace.comp.glsl:137) := fma(v2,b2, fma(v1,b1, v0*b0)) */
static inline vec3 bary_point(const float* tri, float b0, float b1, float b2) {
  return v3(dm_fma(tri[6], b2, dm_fma(tri[3], b1, tri[0] * b0)),
            dm_fma(tri[7], b2, dm_fma(tri[4], b1, tri[1] * b0)),
            dm_fma(tri[8], b2, dm_fma(tri[5], b1, tri[2] * b0)));
}

/* ---------------------------------------------------------------- K0 */
void oracle_lut(const float* tris, uint32_t n, const float model[16], float* lut) {
  /* visibility.vert.glsl:24 worldPos = (model * vec4(p,1)).rgb; visibility.geom.glsl:57-59 */
  memset(lut, 0, 12 * sizeof(float));
  for (uint32_t t = 0; t < n; t++)
    for (int k = 0; k < 3; k++) {
      vec3 p = v3(tris[9 * (uint64_t)t + 3 * k], tris[9 * (uint64_t)t + 3 * k + 1], tris[9 * (uint64_t)t + 3 * k + 2]);
      float* o = lut + 12 * ((uint64_t)t + 1) + 4 * k;
      o[0] = mat4_row_point(model, 0, p);
      o[1] = mat4_row_point(model, 1, p);
      o[2] = mat4_row_point(model, 2, p);
      o[3] = 0.0f;
    }
}

static void gbuffer_rows(const oracle_config* cfg, const float* tris, uint32_t n, const oracle_ubo* ubo,
                    uint32_t y0, uint32_t y1, uint32_t* vis, float* worldpos, float* depth) {
  const int W = (int)cfg->width, H = (int)cfg->height;
  const float* V = ubo->view;
  /* camera origin = -R^T t, ray basis = columns of R (view = [R t]) */
  vec3 tcol = v3(V[12], V[13], V[14]);
  vec3 c0 = v3(V[0], V[1], V[2]), c1 = v3(V[4], V[5], V[6]), c2 = v3(V[8], V[9], V[10]);
  vec3 org = v3(-v3_dot(c0, tcol), -v3_dot(c1, tcol), -v3_dot(c2, tcol));
  float PV[16];
  mat4_mul(ubo->proj, ubo->view, PV); /* visibility.vert.glsl:22 proj * view (* model = I) */
  const float p00 = ubo->proj[0], p11 = ubo->proj[5];
  const float fw = (float)W, fh = (float)H;
  for (int y = (int)y0; y < (int)y1; y++)
    for (int x = col_lo(); x < col_hi(W); x++) {
      /* pixel-centre sample: ndc = (2(x+.5) - W)/W ; view-space direction (ndc.x/P00, ndc.y/P11, -1) */
      float nx = dm_fma(2.0f, (float)x + 0.5f, -fw) / fw;
      float ny = dm_fma(2.0f, (float)y + 0.5f, -fh) / fh;
      vec3 dv = v3(nx / p00, ny / p11, -1.0f);
      vec3 d = v3_normalize(v3(v3_dot(c0, dv), v3_dot(c1, dv), v3_dot(c2, dv)));
      float t, b1, b2;
      float oo[3] = {org.x, org.y, org.z}, dd[3] = {d.x, d.y, d.z};
      uint32_t id = oracle_closest_hit(tris, n, oo, dd, cfg->ray_tmax, &t, &b1, &b2);
      uint64_t i = (uint64_t)y * W + x;
      vis[i] = id; /* visibility.frag.glsl:23 primitiveID+1, clear 0 (main.cpp:1419) */
      if (id) {
        float b0 = 1.0f - b1 - b2;
        vec3 wp = bary_point(tris + 9 * (uint64_t)(id - 1), b0, b1, b2);
        worldpos[4 * i] = wp.x; worldpos[4 * i + 1] = wp.y; worldpos[4 * i + 2] = wp.z; worldpos[4 * i + 3] = 1.0f;
        float cz = mat4_row_point(PV, 2, wp), cw = mat4_row_point(PV, 3, wp);
        depth[i] = cz / cw;
      } else {
        worldpos[4 * i] = 0.f; worldpos[4 * i + 1] = 0.f; worldpos[4 * i + 2] = 0.f; worldpos[4 * i + 3] = 1.0f; /* main.cpp:1420 */
        depth[i] = 1.0f; /* main.cpp:1421 */
      }
    }
}

typedef struct {
  const oracle_config* cfg;
  const float* tris;
  uint32_t n;
  const oracle_ubo* ubo;
  uint32_t* vis;
  float* worldpos;
  float* depth;
} gbuffer_args;

static void gbuffer_range(void* ctx, int64_t a, int64_t b) {
  const gbuffer_args* A = (const gbuffer_args*)ctx;
  gbuffer_rows(A->cfg, A->tris, A->n, A->ubo, (uint32_t)a, (uint32_t)b, A->vis, A->worldpos, A->depth);
}

void oracle_gbuffer(const oracle_config* cfg, const float* tris, uint32_t n, const oracle_ubo* ubo, uint32_t y0, uint32_t y1, uint32_t* vis, float* worldpos, float* depth) {
  gbuffer_args A = {cfg, tris, n, ubo, vis, worldpos, depth};
  run_ranges((int64_t)y0, (int64_t)y1, 4, gbuffer_range, &A);
}

/* ---------------------------------------------------------------- K1 */
static inline vec3 lut_v(const float* lut, uint32_t id, int k) {
  const float* p = lut + 12 * (uint64_t)id + 4 * k;
  return v3(p[0], p[1], p[2]);
}

/* temporalGradient.comp.glsl:50-55 / temporalFiltering.comp.glsl:157-162 */
static inline float tri_area(vec3 a, vec3 b, vec3 c) {
  return v3_length(v3_cross(v3_sub(b, a), v3_sub(c, a))) * 0.5f;
}
/* temporalGradient.comp.glsl:57-69 / temporalFiltering.comp.glsl:164-176 */
static inline vec3 bary_coords(vec3 p, vec3 a, vec3 b, vec3 c) {
  float at = tri_area(a, b, c);
  return v3(tri_area(p, b, c) / at, tri_area(a, p, c) / at, tri_area(a, b, p) / at);
}
static inline vec3 bary_mix(vec3 bc, vec3 a, vec3 b, vec3 c) {
  /* barCoord.x * v1p + barCoord.y * v2p + barCoord.z * v3p */
  return v3(dm_fma(bc.z, c.x, dm_fma(bc.y, b.x, bc.x * a.x)), dm_fma(bc.z, c.y, dm_fma(bc.y, b.y, bc.x * a.y)),
            dm_fma(bc.z, c.z, dm_fma(bc.y, b.z, bc.x * a.z)));
}

/* temporalGradient.comp.glsl:71-101 */
static vec3 phong(vec3 p, vec3 n, vec3 cam, vec3 lpos, vec3 lcol, int shininess) {
  vec3 ldir = v3_normalize(v3_sub(lpos, p));
  vec3 ambient = v3_scale(lcol, 0.1f);
  float diff = dm_max(v3_dot(n, ldir), 0.0f);
  vec3 diffuse = v3_scale(lcol, diff);
  vec3 vdir = v3_normalize(v3_sub(cam, p));
  vec3 I = v3_neg(ldir);
  float two_ndi = 2.0f * v3_dot(n, I); /* reflect(I,N) = I - 2 dot(N,I) N */
  vec3 rdir = v3(dm_fma(-two_ndi, n.x, I.x), dm_fma(-two_ndi, n.y, I.y), dm_fma(-two_ndi, n.z, I.z));
  float spec = dm_powi(dm_max(v3_dot(vdir, rdir), 0.0f), shininess);
  float ss = 0.5f * spec;
  vec3 specular = v3_scale(lcol, ss);
  vec3 sum = v3_add(v3_add(ambient, diffuse), specular);
  return v3_scale(sum, 0.7f); /* attenuation 1.0 (exact), objectColor 0.7 */
}

static void temporal_gradient_rows(const oracle_config* cfg, const oracle_push_constants* pc,
                              const uint32_t* vis, const float* worldpos, const float* lut,
                              const float* lut_prev, uint32_t y0, uint32_t y1, float* grad) {
  const int W = (int)cfg->width;
  vec3 cam = v3(pc->cameraPos[0], pc->cameraPos[1], pc->cameraPos[2]);
  vec3 lp = v3(pc->lightPos[0], pc->lightPos[1], pc->lightPos[2]);
  vec3 lpp = v3(pc->lightPosPrev[0], pc->lightPosPrev[1], pc->lightPosPrev[2]);
  vec3 lc = v3(pc->currentCameraColor[0], pc->currentCameraColor[1], pc->currentCameraColor[2]);
  vec3 lcp = v3(pc->previousCameraColor[0], pc->previousCameraColor[1], pc->previousCameraColor[2]);
  for (int y = (int)y0; y < (int)y1; y++)
    for (int x = col_lo(); x < col_hi(W); x++) {
      uint64_t i = (uint64_t)y * W + x;
      float* g = grad + 4 * i;
      g[0] = g[1] = g[2] = g[3] = 0.0f; /* :119 */
      uint32_t id = vis[i];
      if (id == 0) continue; /* :131 */
      vec3 wp = v3(worldpos[4 * i], worldpos[4 * i + 1], worldpos[4 * i + 2]);
      vec3 a = lut_v(lut, id, 0), b = lut_v(lut, id, 1), c = lut_v(lut, id, 2);
      vec3 nrm = v3_normalize(v3_cross(v3_sub(b, a), v3_sub(c, a))); /* :142 */
      vec3 bc = bary_coords(wp, a, b, c);                            /* :143 */
      vec3 ap = lut_v(lut_prev, id, 0), bp = lut_v(lut_prev, id, 1), cp = lut_v(lut_prev, id, 2);
      vec3 wpp = bary_mix(bc, ap, bp, cp);                           /* :153 */
      vec3 cur = phong(wp, nrm, cam, lp, lc, 128);                   /* :158 */
      vec3 prv = phong(wpp, nrm, cam, lpp, lcp, 128);                /* :161 (current normal!) */
      vec3 tg = v3_sub(cur, prv);
      float delta = dm_max(v3_length(cur), v3_length(prv));          /* :166 */
      float lam = dm_min(1.0f, v3_length(tg) / delta);               /* :167 */
      g[0] = g[1] = g[2] = lam;
      g[3] = 0.0f; /* :170 */
    }
}

typedef struct {
  const oracle_config* cfg;
  const oracle_push_constants* pc;
  const uint32_t* vis;
  const float* worldpos;
  const float* lut;
  const float* lut_prev;
  float* grad;
} temporal_gradient_args;

static void temporal_gradient_range(void* ctx, int64_t a, int64_t b) {
  const temporal_gradient_args* A = (const temporal_gradient_args*)ctx;
  temporal_gradient_rows(A->cfg, A->pc, A->vis, A->worldpos, A->lut, A->lut_prev, (uint32_t)a, (uint32_t)b, A->grad);
}

void oracle_temporal_gradient(const oracle_config* cfg, const oracle_push_constants* pc, const uint32_t* vis, const float* worldpos, const float* lut, const float* lut_prev, uint32_t y0, uint32_t y1, float* grad) {
  temporal_gradient_args A = {cfg, pc, vis, worldpos, lut, lut_prev, grad};
  run_ranges((int64_t)y0, (int64_t)y1, 4, temporal_gradient_range, &A);
}

/* ---------------------------------------------------------------- K2 */
/* raytrace.comp.glsl:84-92 */
static inline void random_gaussian(uint32_t* rng, float* gx, float* gy) {
  float u1 = dm_max(1e-38f, rng_float(rng));
  float u2 = rng_float(rng);
  float r = dm_sqrt(-2.0f * dm_log(u1));
  float s, c;
  dm_sincos2pi(u2, &s, &c); /* theta = 2*k_pi*u2 */
  *gx = r * c;
  *gy = r * s;
}

/* raytrace.comp.glsl:95-107; mix(x,y,a) = x*(1-a) + y*a := fma(y,a, x*(1-a)) */
static inline vec3 sky_color(vec3 d) {
  if (d.y > 0.0f) {
    float a = d.y, ia = 1.0f - a;
    return v3(dm_fma(0.25f, a, 1.0f * ia), dm_fma(0.5f, a, 1.0f * ia), dm_fma(1.0f, a, 1.0f * ia));
  }
  return v3(0.03f, 0.03f, 0.03f);
}

/* raytrace.comp.glsl:168-198 — only the boolean is consumed (:226) */
static inline int ray_hits_light(vec3 o, vec3 d, vec3 center, float radius) {
  vec3 oc = v3_sub(o, center);
  float a = v3_dot(d, d);
  float b = 2.0f * v3_dot(oc, d);
  float c = v3_dot(oc, oc) - radius * radius;
  float disc = dm_fma(b, b, -((4.0f * a) * c));
  if (disc < 0.0f) return 0;
  float sq = dm_sqrt(disc);
  float t1 = (-b - sq) / (2.0f * a);
  float t2 = (-b + sq) / (2.0f * a);
  if (t1 > 0.0f) return 1;
  if (t2 > 0.0f) return 1;
  return 0;
}

void oracle_raytrace(const oracle_config* cfg, const oracle_push_constants* pc, const float* tris,
                     uint32_t n, uint32_t y0, uint32_t y1, float* image, uint64_t* raycount,
                     uint32_t* hit_id) {
  oracle_raytrace_mat(cfg, pc, tris, n, NULL, 0, y0, y1, image, raycount, hit_id);
}

/* materials (not reference behaviour): tri_mat = n_base x 8 floats (Kd.rgb, 0, Ke.rgb, emissive
 * flag), triangle id reads record id % n_base (instances share the base mesh's materials); NULL = the reference's
 * normal-keyed colours (raytrace.comp.glsl:155-163) */
static void raytrace_mat_rows(const oracle_config* cfg, const oracle_push_constants* pc, const float* tris,
                         uint32_t n, const float* tri_mat, uint32_t n_base, uint32_t y0, uint32_t y1, float* image,
                         uint64_t* raycount, uint32_t* hit_id) {
  const int W = (int)cfg->width, H = (int)cfg->height;
  vec3 light_c = v3(pc->lightPos[0], pc->lightPos[1], pc->lightPos[2]); /* :279 */
  vec3 light_col = v3(pc->currentCameraColor[0] * cfg->light_intensity, pc->currentCameraColor[1] * cfg->light_intensity,
                      pc->currentCameraColor[2] * cfg->light_intensity); /* :281 */
  vec3 cam = v3(pc->cameraPos[0], pc->cameraPos[1], pc->cameraPos[2]);
  const float slope = cfg->fov_slope;
  const float fw = (float)W, fh = (float)H;
  uint64_t rays_total = 0;
  for (int y = (int)y0; y < (int)y1; y++)
    for (int x = col_lo(); x < col_hi(W); x++) {
      uint32_t rng = oracle_rng_seed((uint32_t)x, (uint32_t)y, pc->frameNumber, pc->sample_batch); /* :297 */
      vec3 sum = v3(0.f, 0.f, 0.f);
      uint32_t first_id = 0;
      for (uint32_t smp = 0; smp < cfg->samples_per_pixel; smp++) { /* :307 */
        vec3 o = cam;
        float gx, gy;
        random_gaussian(&rng, &gx, &gy);
        float cx = dm_fma(cfg->pixel_jitter, gx, (float)x + 0.5f); /* :314 */
        float cy = dm_fma(cfg->pixel_jitter, gy, (float)y + 0.5f);
        float ux = dm_fma(2.0f, cx, -fw) / fh;    /* :315 */
        float uy = -(dm_fma(2.0f, cy, -fh) / fh); /* :316 */
        vec3 d = v3_normalize(v3(slope * ux, slope * uy, -1.0f)); /* :319-320 */
        vec3 acc = v3(1.f, 1.f, 1.f);                             /* :201 */
        for (uint32_t seg = 0; seg < cfg->max_segments; seg++) {  /* :204 */
          float t, b1, b2;
          float oo[3] = {o.x, o.y, o.z}, dd[3] = {d.x, d.y, d.z};
          uint32_t id = oracle_closest_hit(tris, n, oo, dd, cfg->ray_tmax, &t, &b1, &b2); /* :208-222 */
          rays_total++;
          if (seg == 0 && smp == 0) first_id = id;
          if (ray_hits_light(o, d, light_c, cfg->light_radius)) { /* :226 — not compared with t */
            if (seg == 0) {
              acc = v3_mul(acc, v3(light_col.x / cfg->first_hit_light_divisor, light_col.y / cfg->first_hit_light_divisor,
                                   light_col.z / cfg->first_hit_light_divisor)); /* :229 */
              break;
            }
            acc = v3_mul(acc, light_col); /* :233 */
            break;
          }
          if (id) { /* :238 */
            const float* tri = tris + 9 * (uint64_t)(id - 1);
            float b0 = 1.0f - b1 - b2;           /* :134 */
            vec3 pos = bary_point(tri, b0, b1, b2); /* :137 */
            vec3 v0 = v3(tri[0], tri[1], tri[2]);
            vec3 nrm = v3_normalize(v3_cross(v3_sub(v3(tri[3], tri[4], tri[5]), v0), v3_sub(v3(tri[6], tri[7], tri[8]), v0))); /* :150 */
            vec3 alb;
            if (nrm.x > 0.99f) alb = v3(1.f, 0.f, 0.f);        /* :155 dot(n,(1,0,0)) == n.x exactly */
            else if (-nrm.x > 0.99f) alb = v3(0.f, 1.f, 0.f);  /* :158 */
            else alb = v3(0.7f, 0.7f, 0.7f);                   /* :162 */
            if (tri_mat) {
              const float* m = tri_mat + 8 * (uint64_t)((id - 1) % n_base);
              if (m[7] != 0.0f) { /* an emissive surface ends the path like the analytic light (:226-234) */
                acc = v3_mul(acc, v3(m[4], m[5], m[6]));
                break;
              }
              alb = v3(m[0], m[1], m[2]);
            }
            acc = v3_mul(acc, alb);                            /* :244 */
            if (!(v3_dot(nrm, d) < 0.0f)) nrm = v3_neg(nrm);   /* :247 faceforward(N,I,Nref) */
            o = v3(dm_fma(cfg->ray_offset, nrm.x, pos.x), dm_fma(cfg->ray_offset, nrm.y, pos.y), dm_fma(cfg->ray_offset, nrm.z, pos.z)); /* :250 */
            float s, c;
            dm_sincos2pi(rng_float(&rng), &s, &c);        /* :256 */
            float u = dm_fma(2.0f, rng_float(&rng), -1.0f); /* :257 */
            float r = dm_sqrt(dm_fma(-u, u, 1.0f));       /* :258 */
            d = v3_normalize(v3(dm_fma(r, c, nrm.x), dm_fma(r, s, nrm.y), nrm.z + u)); /* :259-261 */
          } else {
            acc = v3_mul(acc, sky_color(d)); /* :266 */
            break;
          }
        }
        sum = v3_add(sum, acc); /* :325 */
      }
      float ns = (float)cfg->samples_per_pixel;
      uint64_t i = (uint64_t)y * W + x;
      image[4 * i] = sum.x / ns; /* :328 */
      image[4 * i + 1] = sum.y / ns;
      image[4 * i + 2] = sum.z / ns;
      image[4 * i + 3] = 0.0f; /* :343 */
      if (hit_id) hit_id[i] = first_id;
    }
  if (raycount) __atomic_fetch_add(raycount, rays_total, __ATOMIC_RELAXED); /* chunks of one call share it */
}

typedef struct {
  const oracle_config* cfg;
  const oracle_push_constants* pc;
  const float* tris;
  uint32_t n;
  const float* tri_mat;
  uint32_t n_base;
  float* image;
  uint64_t* raycount;
  uint32_t* hit_id;
} raytrace_mat_args;

static void raytrace_mat_range(void* ctx, int64_t a, int64_t b) {
  const raytrace_mat_args* A = (const raytrace_mat_args*)ctx;
  raytrace_mat_rows(A->cfg, A->pc, A->tris, A->n, A->tri_mat, A->n_base, (uint32_t)a, (uint32_t)b, A->image, A->raycount, A->hit_id);
}

void oracle_raytrace_mat(const oracle_config* cfg, const oracle_push_constants* pc, const float* tris, uint32_t n, const float* tri_mat, uint32_t n_base, uint32_t y0, uint32_t y1, float* image, uint64_t* raycount, uint32_t* hit_id) {
  raytrace_mat_args A = {cfg, pc, tris, n, tri_mat, n_base, image, raycount, hit_id};
  run_ranges((int64_t)y0, (int64_t)y1, 4, raytrace_mat_range, &A);
}

/* ---------------------------------------------------------------- K3 */
/* temporalFiltering.comp.glsl:80-91 */
static inline vec3 normal_from_id(const float* lut, uint32_t id) {
  if (id == 0) return v3(0.f, 0.f, 1.f);
  vec3 a = lut_v(lut, id, 0), b = lut_v(lut, id, 1), c = lut_v(lut, id, 2);
  return v3_normalize(v3_cross(v3_sub(b, a), v3_sub(c, a)));
}

/* temporalFiltering.comp.glsl:213-239 + worldToPixel :178-189 */
static void reproject(int W, int H, const float* PVp, uint32_t id, const float* worldpos4, const float* lut_prev, int x,
                      int y, int* ppx, int* ppy) {
  *ppx = x;
  *ppy = y;
  if (id < 1) return; /* :216 */
  vec3 wp = v3(worldpos4[0], worldpos4[1], worldpos4[2]);
  vec3 a = lut_v(lut_prev, id, 0), b = lut_v(lut_prev, id, 1), c = lut_v(lut_prev, id, 2); /* :223-233 */
  vec3 bc = bary_coords(wp, a, b, c);
  vec3 wpp = bary_mix(bc, a, b, c); /* :236 */
  float clx = mat4_row_point(PVp, 0, wpp), cly = mat4_row_point(PVp, 1, wpp), clw = mat4_row_point(PVp, 3, wpp);
  float ndx = clx / clw, ndy = cly / clw;                          /* :183 */
  float sx = dm_fma(ndx, 0.5f, 0.5f) * (float)W;                   /* :186 */
  float sy = dm_fma(ndy, 0.5f, 0.5f) * (float)H;
  *ppx = dm_f2i(sx); /* :238 ivec2() truncation */
  *ppy = dm_f2i(sy);
}

static inline float luminance(vec3 c) { return dm_fma(0.0722f, c.z, dm_fma(0.7152f, c.y, 0.2126f * c.x)); }

static void moments_rows(const oracle_config* cfg, const oracle_push_constants* pc, const oracle_ubo* ubo,
                    const float* traced, const uint32_t* vis, const float* worldpos, const float* lut_prev,
                    const uint32_t* prev_vis, const float* moments_prev, uint32_t y0, uint32_t y1,
                    float* moments_out, float* var_out) {
  const int W = (int)cfg->width, H = (int)cfg->height;
  float PVp[16];
  mat4_mul(ubo->projPrev, ubo->viewPrev, PVp);
  for (int y = (int)y0; y < (int)y1; y++)
    for (int x = col_lo(); x < col_hi(W); x++) {
      uint64_t ip = (uint64_t)y * W + x;
      uint32_t id = vis[ip];
      float lum = luminance(v3(traced[4 * ip], traced[4 * ip + 1], traced[4 * ip + 2]));
      int ppx, ppy;
      reproject(W, H, PVp, id, worldpos + 4 * ip, lut_prev, x, y, &ppx, &ppy);
      int valid = pc->frameNumber > 0 && ppx >= 0 && ppx < W && ppy >= 0 && ppy < H;
      uint64_t iq = valid ? (uint64_t)ppy * W + ppx : 0;
      if (valid) valid = prev_vis[iq] == id;
      float m1 = lum, m2 = lum * lum, n = 1.0f;
      if (valid) {
        const float* mp = moments_prev + 4 * iq;
        float a = dm_max(cfg->alpha, 1.0f / (mp[2] + 1.0f)), oma = 1.0f - a;
        m1 = dm_fma(lum, a, mp[0] * oma);
        m2 = dm_fma(lum * lum, a, mp[1] * oma);
        n = dm_min(mp[2] + 1.0f, 255.0f);
      }
      float var = dm_max(0.0f, dm_fma(-m1, m1, m2));
      if (n < 4.0f) {
        if (cfg->ext_flags & ORACLE_EXT_SVGF_VARIANCE) {
          /* SVGF: too short a history says nothing about the variance yet — estimate it spatially, from the current
           * frame's luminance over the 7x7 neighbourhood, taps on the same primitive only (the centre always counts) */
          float s1 = 0.0f, s2 = 0.0f, cnt = 0.0f;
          for (int dy = -3; dy <= 3; dy++)
            for (int dx = -3; dx <= 3; dx++) {
              int qx = x + dx, qy = y + dy;
              qx = qx < 0 ? 0 : (qx > W - 1 ? W - 1 : qx);
              qy = qy < 0 ? 0 : (qy > H - 1 ? H - 1 : qy);
              uint64_t iqn = (uint64_t)qy * W + qx;
              if (vis[iqn] != id) continue;
              float l = luminance(v3(traced[4 * iqn], traced[4 * iqn + 1], traced[4 * iqn + 2]));
              s1 = s1 + l;
              s2 = dm_fma(l, l, s2);
              cnt = cnt + 1.0f;
            }
          float m1s = s1 / cnt, m2s = s2 / cnt;
          var = dm_max(0.0f, dm_fma(-m1s, m1s, m2s));
        }
        var = var * (4.0f / n);
      }
      float* mo = moments_out + 4 * ip;
      mo[0] = m1; mo[1] = m2; mo[2] = n; mo[3] = var;
      var_out[ip] = var;
    }
}

typedef struct {
  const oracle_config* cfg;
  const oracle_push_constants* pc;
  const oracle_ubo* ubo;
  const float* traced;
  const uint32_t* vis;
  const float* worldpos;
  const float* lut_prev;
  const uint32_t* prev_vis;
  const float* moments_prev;
  float* moments_out;
  float* var_out;
} moments_args;

static void moments_range(void* ctx, int64_t a, int64_t b) {
  const moments_args* A = (const moments_args*)ctx;
  moments_rows(A->cfg, A->pc, A->ubo, A->traced, A->vis, A->worldpos, A->lut_prev, A->prev_vis, A->moments_prev, (uint32_t)a, (uint32_t)b, A->moments_out, A->var_out);
}

void oracle_moments(const oracle_config* cfg, const oracle_push_constants* pc, const oracle_ubo* ubo, const float* traced, const uint32_t* vis, const float* worldpos, const float* lut_prev, const uint32_t* prev_vis, const float* moments_prev, uint32_t y0, uint32_t y1, float* moments_out, float* var_out) {
  moments_args A = {cfg, pc, ubo, traced, vis, worldpos, lut_prev, prev_vis, moments_prev, moments_out, var_out};
  run_ranges((int64_t)y0, (int64_t)y1, 4, moments_range, &A);
}

/* ORACLE_EXT_SVGF_VARIANCE (ii): 3x3 Gaussian of the variance plane around (x, y), frame-clamped */
static inline float var_prefilter_at(const float* var, int W, int H, int x, int y) {
  static const float g[3][3] = {{1.f, 2.f, 1.f}, {2.f, 4.f, 2.f}, {1.f, 2.f, 1.f}};
  float acc = 0.0f;
  for (int dy = -1; dy <= 1; dy++)
    for (int dx = -1; dx <= 1; dx++) {
      int qx = x + dx, qy = y + dy;
      qx = qx < 0 ? 0 : (qx > W - 1 ? W - 1 : qx);
      qy = qy < 0 ? 0 : (qy > H - 1 ? H - 1 : qy);
      acc = dm_fma(g[dy + 1][dx + 1], var[(uint64_t)qy * W + qx], acc);
    }
  return acc * 0.0625f;
}
void oracle_var_prefilter(const oracle_config* cfg, const float* var, uint32_t y0, uint32_t y1, float* out) {
  const int W = (int)cfg->width, H = (int)cfg->height;
  for (int y = (int)y0; y < (int)y1; y++)
    for (int x = 0; x < W; x++) out[(uint64_t)y * W + x] = var_prefilter_at(var, W, H, x, y);
}

/* gaussianKernel2D, temporalFiltering.comp.glsl:93-99 (sum 273) */
static const float k_gauss5[5][5] = {{1, 4, 7, 4, 1}, {4, 16, 26, 16, 4}, {7, 26, 41, 26, 7}, {4, 16, 26, 16, 4}, {1, 4, 7, 4, 1}};

void oracle_atrous(const oracle_config* cfg, const oracle_push_constants* pc, const oracle_ubo* ubo,
                   const float* in, const float* depth, const uint32_t* vis, const float* lut,
                   const float* lut_prev, const float* worldpos, const float* history,
                   uint32_t y0, uint32_t y1, float* out, int32_t* prev_pixel) {
  oracle_atrous_ext(cfg, pc, ubo, in, depth, vis, lut, lut_prev, worldpos, history, NULL, NULL, y0, y1, out, prev_pixel);
}

void oracle_atrous_ext(const oracle_config* cfg, const oracle_push_constants* pc, const oracle_ubo* ubo,
                       const float* in, const float* depth, const uint32_t* vis, const float* lut,
                       const float* lut_prev, const float* worldpos, const float* history,
                       const float* gradient, const uint32_t* prev_vis,
                       uint32_t y0, uint32_t y1, float* out, int32_t* prev_pixel) {
  oracle_atrous_var(cfg, pc, ubo, in, depth, vis, lut, lut_prev, worldpos, history, gradient, prev_vis, NULL, y0, y1, out,
                    prev_pixel, NULL);
}

static void atrous_var_rows(const oracle_config* cfg, const oracle_push_constants* pc, const oracle_ubo* ubo,
                       const float* in, const float* depth, const uint32_t* vis, const float* lut,
                       const float* lut_prev, const float* worldpos, const float* history,
                       const float* gradient, const uint32_t* prev_vis, const float* var_in,
                       uint32_t y0, uint32_t y1, float* out, int32_t* prev_pixel, float* var_out) {
  const uint32_t ext = cfg->ext_flags;
  const int use_var = (ext & ORACLE_EXT_VARIANCE) && var_in;
  const int R = (ext & ORACLE_EXT_GAUSS5) ? 2 : 1;
  const int W = (int)cfg->width, H = (int)cfg->height;
  const int kk = pc->waveletIteration, max_it = pc->maxWaveletIteration; /* :208-209 */
  const int k = (ext & ORACLE_EXT_POW2_STRIDE) ? (1 << (kk - 1)) : kk;   /* tap stride */
  /* main.cpp:1264-1281: on even k colorImage is filteredImageBuffer, so the blend of an even
   * final pass lands in a buffer nothing reads ("must be an odd number", main.cpp:55) and
   * `image` keeps the plain filtered colour: only an odd final pass blends observably. */
  const int final_pass = (kk == max_it) && (kk & 1);
  float PVp[16];
  if (final_pass) mat4_mul(ubo->projPrev, ubo->viewPrev, PVp); /* :180 projMatrix * viewMatrix */
  const float h = 1.0f / 9.0f;                    /* :145 */
  const float one_minus_alpha = 1.0f - cfg->alpha; /* :254 */
  for (int y = (int)y0; y < (int)y1; y++)
    for (int x = col_lo(); x < col_hi(W); x++) {
      uint64_t ip = (uint64_t)y * W + x;
      vec3 cp = v3(in[4 * ip], in[4 * ip + 1], in[4 * ip + 2]); /* :122 */
      float dp = depth[ip];                                    /* :123 */
      vec3 np = normal_from_id(lut, vis[ip]);                  /* :125-127 */
      vec3 num = v3(0.f, 0.f, 0.f);
      float den = 0.f, vsum = 0.f;
      const float lum_p = luminance(cp);
      const float var_c = !use_var ? 0.0f : ((ext & ORACLE_EXT_SVGF_VARIANCE) ? var_prefilter_at(var_in, W, H, x, y) : var_in[ip]);
      const float lum_scale = use_var ? dm_fma(cfg->sigma_l, dm_sqrt(dm_max(var_c, 0.0f)), 1e-4f) : 1.0f;
      for (int i = -R; i <= R; i++)     /* :132 (-1..1; -2..2 with the 5x5 table) */
        for (int j = -R; j <= R; j++) { /* :133 */
          int qx = x + i * k, qy = y + j * k; /* :135 */
          qx = qx < 0 ? 0 : (qx > W - 1 ? W - 1 : qx); /* :136 */
          qy = qy < 0 ? 0 : (qy > H - 1 ? H - 1 : qy);
          uint64_t iq = (uint64_t)qy * W + qx;
          vec3 cq = v3(in[4 * iq], in[4 * iq + 1], in[4 * iq + 2]);
          float dq = depth[iq];
          vec3 nq = normal_from_id(lut, vis[iq]);
          float wn = dm_powi(dm_max(0.0f, v3_dot(np, nq)), cfg->sigma_n);    /* :62 */
          float wd = dm_exp(-fabsf(dp - dq) / cfg->sigma_z);                 /* :67-68 */
          float wl = use_var ? dm_exp(-fabsf(lum_p - luminance(cq)) / lum_scale)
                             : dm_exp(-v3_length(v3_sub(cp, cq)) / cfg->sigma_l); /* :73 */
          float w = (wn * wd) * wl;                                          /* :77 */
          float hw = ((ext & ORACLE_EXT_GAUSS5) ? k_gauss5[i + 2][j + 2] * (1.0f / 273.0f) : h) * w;
          num = v3(dm_fma(hw, cq.x, num.x), dm_fma(hw, cq.y, num.y), dm_fma(hw, cq.z, num.z)); /* :146 */
          den = den + hw;                                                     /* :147 */
          if (use_var) vsum = dm_fma(hw * hw, var_in[iq], vsum);
        }
      vec3 filtered = v3(num.x / den, num.y / den, num.z / den); /* :150 */
      if (use_var && var_out) var_out[ip] = vsum / (den * den);
      float* o = out + 4 * ip;
      if (!final_pass) {
        o[0] = filtered.x; o[1] = filtered.y; o[2] = filtered.z; o[3] = 0.0f; /* :152 */
        continue;
      }
      /* :213-239 reprojection (evaluated by the reference on every iteration, consumed only here) */
      uint32_t id = vis[ip];
      int ppx, ppy;
      reproject(W, H, PVp, id, worldpos + 4 * ip, lut_prev, x, y, &ppx, &ppy);
      if (prev_pixel) { prev_pixel[2 * ip] = ppx; prev_pixel[2 * ip + 1] = ppy; }
      vec3 blend;
      int use_history = pc->frameNumber > 0; /* :251 */
      const int inside = (ppx >= 0 && ppx < W && ppy >= 0 && ppy < H);
      if (use_history && (ext & ORACLE_EXT_DISOCCLUSION)) /* extension: same primitive at the reprojected pixel */
        use_history = inside && prev_vis[(uint64_t)ppy * W + ppx] == id;
      if (use_history) {
        vec3 hc = v3(0.f, 0.f, 0.f); /* D2: out-of-image history reads as 0 */
        if (inside) {
          uint64_t ih = (uint64_t)ppy * W + ppx;
          hc = v3(history[4 * ih], history[4 * ih + 1], history[4 * ih + 2]);
        }
        float alpha = cfg->alpha, oma = one_minus_alpha;
        if (ext & ORACLE_EXT_ADAPTIVE_ALPHA) { /* :247-248, commented out in the reference */
          float g = gradient[4 * ip];
          alpha = dm_fma(1.0f - g, alpha, g);
          oma = 1.0f - alpha;
        }
        blend = v3(dm_fma(filtered.x, alpha, hc.x * oma), dm_fma(filtered.y, alpha, hc.y * oma),
                   dm_fma(filtered.z, alpha, hc.z * oma)); /* :254 */
      } else {
        blend = filtered; /* :258 */
      }
      o[0] = blend.x; o[1] = blend.y; o[2] = blend.z; o[3] = 0.0f; /* :263 */
    }
}

typedef struct {
  const oracle_config* cfg;
  const oracle_push_constants* pc;
  const oracle_ubo* ubo;
  const float* in;
  const float* depth;
  const uint32_t* vis;
  const float* lut;
  const float* lut_prev;
  const float* worldpos;
  const float* history;
  const float* gradient;
  const uint32_t* prev_vis;
  const float* var_in;
  float* out;
  int32_t* prev_pixel;
  float* var_out;
} atrous_var_args;

static void atrous_var_range(void* ctx, int64_t a, int64_t b) {
  const atrous_var_args* A = (const atrous_var_args*)ctx;
  atrous_var_rows(A->cfg, A->pc, A->ubo, A->in, A->depth, A->vis, A->lut, A->lut_prev, A->worldpos, A->history, A->gradient, A->prev_vis, A->var_in, (uint32_t)a, (uint32_t)b, A->out, A->prev_pixel, A->var_out);
}

void oracle_atrous_var(const oracle_config* cfg, const oracle_push_constants* pc, const oracle_ubo* ubo, const float* in, const float* depth, const uint32_t* vis, const float* lut, const float* lut_prev, const float* worldpos, const float* history, const float* gradient, const uint32_t* prev_vis, const float* var_in, uint32_t y0, uint32_t y1, float* out, int32_t* prev_pixel, float* var_out) {
  atrous_var_args A = {cfg, pc, ubo, in, depth, vis, lut, lut_prev, worldpos, history, gradient, prev_vis, var_in, out, prev_pixel, var_out};
  run_ranges((int64_t)y0, (int64_t)y1, 4, atrous_var_range, &A);
}
