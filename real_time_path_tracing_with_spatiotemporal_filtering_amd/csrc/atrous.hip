// atrous.hip — K3: one iteration of the edge-stopping a-trous filter
// (temporalFiltering.comp.glsl:191-265): 3x3 taps at stride k, h = 1/9, weights on normal / depth /
// colour; the FINAL variants fuse reprojection + temporal blend (:213-263).
//
//   k_atrous_comb   the shipping kernel: wave-private LDS staging by LDS-DMA, comb row assignment
//   k_atrous        direct global-load kernel: scenes whose id-pair weight table does not fit LDS
//                   (> 63 triangles; a comb variant gathering per-id normals from an 18 MB table was
//                   measured SLOWER on the 1.15M-triangle scene: 197 vs 149 us), strides k > 16, and
//                   RTPT_FLAG_DIRECT_FILTER
//
// Measured on MI355X at 3840x2160 (profiles/): the direct kernel is bound by the vector-memory pipe
// (27 load instructions per pixel: 107 us even with every tap an L1 hit); tile kernels with a
// workgroup barrier run the chip in load/compute lockstep.  The comb kernel has neither problem.
//
// Colour planes are "rgbd" while a frame is being filtered: the alpha channel, which the reference
// always writes as 0 (raytrace.comp.glsl:343, temporalFiltering.comp.glsl:152), carries the pixel's
// G-buffer depth from rtpt_raytrace through every non-final pass, so a tap is ONE 16-byte cell
// (colour + depth) plus the 4-byte id instead of three separate planes — 36 instead of 40 bytes of
// HBM traffic per pixel and a third fewer load instructions.  The final pass writes alpha 0 again;
// the C-ABI layer hides the convention (rtpt_readback masks alpha, rtpt_set_plane re-stamps depth).
#include "device_common.hpp"
#include "lds_dma.hpp"

#ifndef RTPT_TILE_TIMELINE
#define RTPT_TILE_TIMELINE 0  // timeline build (scripts/tile_timeline.py): when the workgroups of the single-pass launches lived
#endif

namespace rt {
namespace {

#if RTPT_TILE_TIMELINE
#define RTPT_SPAN_SLOTS 2  // k_atrous_comb_sh: an iteration k < N, the final pass
#define RTPT_SPAN_READER rtpt_debug_comb_span
#define RTPT_SPAN_DEVICE
#include "experiments/span_instrumentation.inc"
#undef RTPT_SPAN_DEVICE
#endif

// main.cpp:1338-1361 vkCmdBlitImage image (RGBA32F) -> swapchain image (B8G8R8A8_UNORM): the float -> UNORM
// conversion clamps to [0,1] and quantises; defined here as trunc(x*255 + 0.5) with separate multiply and add (the
// file is compiled -ffp-contract=off), NaN -> 0 (max(NaN, 0) = 0), which is what output.to_unorm8 / the oracle compute.
__device__ __forceinline__ uint32_t unorm8(float x) {
  const float c = fminf(fmaxf(x, 0.0f), 1.0f);
  return static_cast<uint32_t>(c * 255.0f + 0.5f);
}


// x^n for the normal weight (temporalFiltering.comp.glsl:62).  The reference's exponent is 128: seven squarings,
// written straight-line — exact::powi's square-and-multiply LOOP yields the same products in the same order but
// runs its control flow on the CU's single scalar unit, which made per-tap use of it SALU-bound.
__device__ __forceinline__ float pow_sigma(float x, int n) {
  if (n == 128) {
    const float x2 = x * x, x4 = x2 * x2, x8 = x4 * x4, x16 = x8 * x8, x32 = x16 * x16, x64 = x32 * x32;
    return x64 * x64;
  }
  return exact::powi(x, n);
}

// XCD-aware tile mapping.  Workgroups are dealt round-robin over the 8 XCDs (block b and b+8 share an
// XCD and its private 4 MiB L2).  The stencil re-reads every input row at y-k, y and y+k, so the three
// uses must meet in ONE L2 while the rows in between are streamed through it.  Tiles are therefore
// enumerated strip-major — vertical strips kStripTiles tiles (512 px) wide, row-major inside a strip,
// so the reuse window of 2k+few rows is a few hundred KB — and each XCD gets one contiguous chunk of
// that enumeration (bijective for any block count).  Placement is a speed assumption only; any
// dispatch order computes the same pixels.
constexpr int kStripTiles = 1 << 20;  // one strip = the whole width: row-major bands (strips measured slower)
__device__ __forceinline__ void xcd_strip_tile(int tiles_x, int tiles_y, int& bx, int& by) {
  const uint32_t tx = static_cast<uint32_t>(tiles_x), ty = static_cast<uint32_t>(tiles_y);
  const uint32_t nb = tx * ty;
  const uint32_t b = blockIdx.x;
  const uint32_t q = nb >> 3, r = nb & 7u, xcd = b & 7u, j = b >> 3;
  const uint32_t lb = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
  const uint32_t full = tx / kStripTiles;           // full-width strips
  const uint32_t per_strip = kStripTiles * ty;
  uint32_t strip, rem, sw;
  if (lb < full * per_strip) {
    strip = lb / per_strip;
    rem = lb - strip * per_strip;
    sw = kStripTiles;
  } else {
    strip = full;
    rem = lb - full * per_strip;
    sw = tx - full * kStripTiles;
  }
  const uint32_t row = rem / sw;
  by = static_cast<int>(row);
  bx = static_cast<int>(strip * kStripTiles + (rem - row * sw));
}

template <bool FINAL, bool EXACT>
__global__ __launch_bounds__(kThreads) void k_atrous(AtrousArgs a) {
  int bx, by;
  xcd_strip_tile(a.tiles_x, a.tiles_y, bx, by);
  // blockDim.x == 64: a wave is one row segment, so the row index is wave-uniform (SGPR) and every
  // tap row resolves to a scalar base address + a 32-bit per-lane column offset
  const int ty = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.y));
  const int x = bx * kBlockX + static_cast<int>(threadIdx.x);
  const int y = a.g.y0 + by * kBlockY + ty;
  if (x >= a.g.W || y >= a.g.y1) return;
  const int W = a.g.W, H = a.g.H, k = a.k;
  const size_t rowp = static_cast<size_t>(y - a.g.row_base) * W;
  const float4 cp4 = a.in[rowp + x];
  const f3 cp = xyz(cp4);
  const float dp = cp4.w;  // rgbd
  const uint32_t idp = a.vis[rowp + x];
  const float4 np4 = a.normal_tab[idp];
  const f3 np = xyz(np4);
  f3 num{0.f, 0.f, 0.f};
  float den = 0.f;
  const float h = 1.0f / 9.0f;  // temporalFiltering.comp.glsl:145
  int qxs[3], qys[3];
#pragma unroll
  for (int i = 0; i < 3; i++) {
    int qx = x + (i - 1) * k, qy = y + (i - 1) * k;  // :135
    qxs[i] = qx < 0 ? 0 : (qx > W - 1 ? W - 1 : qx);  // :136
    qys[i] = qy < 0 ? 0 : (qy > H - 1 ? H - 1 : qy);
  }
#pragma unroll
  for (int i = 0; i < 3; i++) {  // :132 (x offset outer, as in the reference's accumulation order)
#pragma unroll
    for (int j = 0; j < 3; j++) {  // :133
      float w;
      f3 cq;
      if (i == 1 && j == 1 && k > 0) {
        // centre tap: q == p, so both exponentials are exactly 1 and w = pow(max(0,dot(np,np)),sigma_n)
        cq = cp;
        w = np4.w;
      } else {
        const size_t rowq = static_cast<size_t>(qys[j] - a.g.row_base) * W;
        const int qx = qxs[i];
        const float4 cq4 = a.in[rowq + qx];
        cq = xyz(cq4);
        const float dq = cq4.w;  // rgbd
        const uint32_t idq = a.vis[rowq + qx];
        float wn;
        if (idq == idp) {
          wn = np4.w;  // same primitive: the per-id self weight (same bits as recomputing it)
        } else {
          const f3 nq = xyz(a.normal_tab[idq]);
          wn = pow_sigma(glsl_max(0.0f, exact::dot(np, nq)), a.sigma_n);  // :62
        }
        const f3 dc = cp - cq;
        if (EXACT) {
          const float wd = exact::exp_(-__builtin_fabsf(dp - dq) / a.sigma_z);  // :67-68
          const float wl = exact::exp_(-exact::length(dc) / a.sigma_l);         // :73
          w = (wn * wd) * wl;                                                   // :77
        } else {
          // exp(-|dz|/sz) * exp(-|dc|/sl) = exp2(|dz| * cz + |dc| * cl), cz/cl = -log2(e)/sigma
          const float e = fmaf_(__builtin_fabsf(dp - dq), a.cz, fast::sqrt_(exact::dot(dc, dc)) * a.cl);
          w = wn * __builtin_amdgcn_exp2f(e);
        }
      }
      if (EXACT) {
        const float hw = h * w;
        num = f3{fmaf_(hw, cq.x, num.x), fmaf_(hw, cq.y, num.y), fmaf_(hw, cq.z, num.z)};  // :146
        den = den + hw;                                                                    // :147
      } else {
        // h = 1/9 scales numerator and denominator alike; the fast path drops it
        num = f3{fmaf_(w, cq.x, num.x), fmaf_(w, cq.y, num.y), fmaf_(w, cq.z, num.z)};
        den = den + w;
      }
    }
  }
  f3 filtered;
  if (EXACT) {
    filtered = f3{num.x / den, num.y / den, num.z / den};  // :150
  } else {
    const float rd = fast::rcp_(den);
    filtered = num * rd;
  }
  if (!FINAL) {
    a.out[rowp + x] = make_float4(filtered.x, filtered.y, filtered.z, a.alpha_zero ? 0.0f : dp);  // :152 (+ depth in alpha)
    return;
  }
  // :213-239 reprojection — exact arithmetic: the truncated pixel coordinate is an integer observable
  int ppx = x, ppy = y;
  if (!(idp < 1)) {
    const f3 wp = xyz(a.worldpos[rowp + x]);
    const f3 va = xyz(a.lut_prev[3 * idp]), vb = xyz(a.lut_prev[3 * idp + 1]), vc = xyz(a.lut_prev[3 * idp + 2]);  // :223-233
    const f3 bc = bary_coords(wp, va, vb, vc);
    const f3 wpp = bary_mix(bc, va, vb, vc);  // :236
    const float clx = exact::mat_row_point(a.PVprev, 0, wpp), cly = exact::mat_row_point(a.PVprev, 1, wpp),
                clw = exact::mat_row_point(a.PVprev, 3, wpp);
    const float ndx = clx / clw, ndy = cly / clw;                      // :183
    ppx = exact::f2i(fmaf_(ndx, 0.5f, 0.5f) * static_cast<float>(W));  // :186,:238
    ppy = exact::f2i(fmaf_(ndy, 0.5f, 0.5f) * static_cast<float>(H));
  }
  if (a.prev_pixel) a.prev_pixel[rowp + x] = make_int2(ppx, ppy);
  f3 blend = filtered;  // :258
  if (a.frame > 0) {    // :251
    f3 hc{0.f, 0.f, 0.f};  // D2: out-of-image history fetch returns 0
    if (ppx >= 0 && ppx < W && ppy >= a.hist_y0 && ppy < a.hist_y1)
      hc = xyz(a.history[static_cast<size_t>(ppy - a.hist_row_base) * W + ppx]);
    const float oma = 1.0f - a.alpha;
    blend = f3{fmaf_(filtered.x, a.alpha, hc.x * oma), fmaf_(filtered.y, a.alpha, hc.y * oma),
               fmaf_(filtered.z, a.alpha, hc.z * oma)};  // :254
  }
  a.out[rowp + x] = make_float4(blend.x, blend.y, blend.z, 0.0f);  // :263 (D1: distinct buffer)
}


// Extension modes (RTPT_FLAG_EXT_*): the pieces of the textbook A-SVGF the reference declares but does not
// use — the 5x5 gaussianKernel2D table (temporalFiltering.comp.glsl:93-99), 2^(k-1) tap stride, the
// gradient-adaptive alpha (:247-248, commented out) and a disocclusion test on previousVisibilityBuffer
// (main.cpp:1367: copied every frame, never read).  Opt-in and outside the reference's behaviour, so this is
// one generic direct-load kernel (same tap arithmetic as k_atrous) rather than a tuned one.
__constant__ float kGauss5[5][5] = {{1, 4, 7, 4, 1}, {4, 16, 26, 16, 4}, {7, 26, 41, 26, 7}, {4, 16, 26, 16, 4}, {1, 4, 7, 4, 1}};

__device__ __forceinline__ float luminance(f3 c) { return fmaf_(0.0722f, c.z, fmaf_(0.7152f, c.y, 0.2126f * c.x)); }

// RTPT_FLAG_EXT_VARIANCE (extension, see include/rtpt.h): first and second luminance moments accumulated along the
// reprojected pixel, history length, and the variance the filter iterations are guided by.
__global__ __launch_bounds__(kThreads) void k_moments(MomentsArgs a) {
  const int x = blockIdx.x * kBlockX + threadIdx.x;
  const int y = a.g.y0 + blockIdx.y * kBlockY + threadIdx.y;
  if (x >= a.g.W || y >= a.g.y1) return;
  const int W = a.g.W, H = a.g.H;
  const size_t ip = static_cast<size_t>(y - a.g.row_base) * W + x;
  const uint32_t id = a.vis[ip];
  const float lum = luminance(xyz(a.traced[ip]));
  int ppx, ppy;
  reproject_pixel(W, H, a.PVprev, id, xyz(a.worldpos[ip]), a.lut_prev, x, y, ppx, ppy);
  // the previous frame's id and moment planes hold frame rows [hist_y0, hist_y1) from hist_row_base on: the context's
  // own rows, or the bands gathered from the other strips (rtpt_set_external_guides); inside the frame but outside
  // those rows cannot happen when the host registered the rows the strip's pixels can reach
  bool valid = a.frame > 0 && ppx >= 0 && ppx < W && ppy >= 0 && ppy < H && ppy >= a.hist_y0 && ppy < a.hist_y1;
  const size_t iq = valid ? static_cast<size_t>(ppy - a.hist_row_base) * W + ppx : 0;
  if (valid) valid = a.prev_vis[iq] == id;
  float m1 = lum, m2 = lum * lum, n = 1.0f;
  if (valid) {
    const float4 mp = a.moments_prev[iq];
    const float al = glsl_max(a.alpha, 1.0f / (mp.z + 1.0f)), oma = 1.0f - al;
    m1 = fmaf_(lum, al, mp.x * oma);
    m2 = fmaf_(lum * lum, al, mp.y * oma);
    n = glsl_min(mp.z + 1.0f, 255.0f);
  }
  float var = glsl_max(0.0f, fmaf_(-m1, m1, m2));
  if (n < 4.0f) {
    if (a.svgf) {
      // SVGF: a history this short says nothing about the variance yet — estimate it from the current frame's luminance
      // over the 7x7 neighbourhood, taps on the same primitive only (the centre always counts), clamped at the frame border.
      // On a strip the window of the outermost rows would leave the stored rows: those reads are clamped to what is stored
      // (memory safety only — a strip stores 3 traced rows beyond the rows whose variance anything consumes, strips.py)
      const int r_lo = a.g.row_base, r_hi = a.g.row_base + a.rows_stored - 1;
      float s1 = 0.0f, s2 = 0.0f, cnt = 0.0f;
      for (int dy = -3; dy <= 3; dy++)
        for (int dx = -3; dx <= 3; dx++) {
          int qx = x + dx, qy = y + dy;
          qx = qx < 0 ? 0 : (qx > W - 1 ? W - 1 : qx);
          qy = qy < 0 ? 0 : (qy > H - 1 ? H - 1 : qy);
          qy = qy < r_lo ? r_lo : (qy > r_hi ? r_hi : qy);
          const size_t iqn = static_cast<size_t>(qy - a.g.row_base) * W + qx;
          if (a.vis[iqn] != id) continue;
          const float l = luminance(xyz(a.traced[iqn]));
          s1 = s1 + l;
          s2 = fmaf_(l, l, s2);
          cnt = cnt + 1.0f;
        }
      const float m1s = s1 / cnt, m2s = s2 / cnt;
      var = glsl_max(0.0f, fmaf_(-m1s, m1s, m2s));
    }
    var = var * (4.0f / n);
  }
  a.moments_out[ip] = make_float4(m1, m2, n, var);
  a.var_out[ip] = var;
}

// RTPT_FLAG_EXT_SVGF_VARIANCE: the variance that scales an iteration's luminance weight is the 3x3 Gaussian of the variance
// plane around the pixel (SVGF's variance prefilter), unit spacing whatever the iteration's stride: a pass of its own (8 B/px)
// because the staged filter kernels hold rows k apart
__global__ __launch_bounds__(kThreads) void k_var_prefilter(FrameGeom g, int rows_stored, const float* __restrict__ var, float* __restrict__ out) {
  const int x = blockIdx.x * kBlockX + threadIdx.x;
  const int y = g.y0 + blockIdx.y * kBlockY + threadIdx.y;
  if (x >= g.W || y >= g.y1) return;
  float acc = 0.0f;
#pragma unroll
  for (int dy = -1; dy <= 1; dy++)
#pragma unroll
    for (int dx = -1; dx <= 1; dx++) {
      int qx = x + dx, qy = y + dy;
      qx = qx < 0 ? 0 : (qx > g.W - 1 ? g.W - 1 : qx);
      qy = qy < 0 ? 0 : (qy > g.H - 1 ? g.H - 1 : qy);
      qy = qy < g.row_base ? g.row_base : (qy > g.row_base + rows_stored - 1 ? g.row_base + rows_stored - 1 : qy);  // memory safety on strips
      const float gw = (dx == 0 ? 2.0f : 1.0f) * (dy == 0 ? 2.0f : 1.0f);
      acc = fmaf_(gw, var[static_cast<size_t>(qy - g.row_base) * g.W + qx], acc);
    }
  out[static_cast<size_t>(y - g.row_base) * g.W + x] = acc * 0.0625f;
}

template <bool FINAL, bool EXACT>
__global__ __launch_bounds__(kThreads) void k_atrous_ext(AtrousArgs a) {
  int bx, by;
  xcd_strip_tile(a.tiles_x, a.tiles_y, bx, by);
  const int ty = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.y));
  const int x = bx * kBlockX + static_cast<int>(threadIdx.x);
  const int y = a.g.y0 + by * kBlockY + ty;
  if (x >= a.g.W || y >= a.g.y1) return;
  const int W = a.g.W, H = a.g.H, k = a.stride;
  const bool gauss = (a.ext & kExtGauss5) != 0;
  const int R = gauss ? 2 : 1;
  const size_t rowp = static_cast<size_t>(y - a.g.row_base) * W;
  const float4 cp4 = a.in[rowp + x];
  const f3 cp = xyz(cp4);
  const float dp = cp4.w;  // rgbd
  const uint32_t idp = a.vis[rowp + x];
  const float4 np4 = a.normal_tab[idp];
  const f3 np = xyz(np4);
  f3 num{0.f, 0.f, 0.f};
  float den = 0.f, vsum = 0.f;
  // variance guidance (RTPT_FLAG_EXT_VARIANCE): the colour term compares luminances, scaled by the pixel's own
  // standard deviation
  const bool use_var = (a.ext & kExtVariance) && a.var_in;
  const float lum_p = luminance(cp);
  const float lum_scale = use_var ? fmaf_(a.sigma_l, exact::sqrt_(glsl_max((a.var_scale ? a.var_scale : a.var_in)[rowp + x], 0.0f)), 1e-4f) : 1.0f;
  const float cl_var = -1.44269504088896341f * fast::rcp_(lum_scale);
  for (int i = -R; i <= R; i++) {    // :132
    for (int j = -R; j <= R; j++) {  // :133
      int qx = x + i * k, qy = y + j * k;  // :135
      qx = qx < 0 ? 0 : (qx > W - 1 ? W - 1 : qx);  // :136
      qy = qy < 0 ? 0 : (qy > H - 1 ? H - 1 : qy);
      const size_t rowq = static_cast<size_t>(qy - a.g.row_base) * W;
      const float4 cq4 = a.in[rowq + qx];
      const f3 cq = xyz(cq4);
      const float dq = cq4.w;
      const uint32_t idq = a.vis[rowq + qx];
      const f3 nq = xyz(a.normal_tab[idq]);
      const float wn = pow_sigma(glsl_max(0.0f, exact::dot(np, nq)), a.sigma_n);  // :62
      const f3 dc = cp - cq;
      float w;
      if (EXACT) {
        const float wd = exact::exp_(-__builtin_fabsf(dp - dq) / a.sigma_z);  // :67-68
        const float wl = use_var ? exact::exp_(-__builtin_fabsf(lum_p - luminance(cq)) / lum_scale)
                                 : exact::exp_(-exact::length(dc) / a.sigma_l);  // :73
        w = (wn * wd) * wl;                                                   // :77
      } else {
        const float dl = use_var ? __builtin_fabsf(lum_p - luminance(cq)) * cl_var : fast::sqrt_(exact::dot(dc, dc)) * a.cl;
        const float e = fmaf_(__builtin_fabsf(dp - dq), a.cz, dl);
        w = wn * __builtin_amdgcn_exp2f(e);
      }
      const float h = gauss ? kGauss5[i + 2][j + 2] * (1.0f / 273.0f) : 1.0f / 9.0f;  // :145
      const float hw = h * w;
      num = f3{fmaf_(hw, cq.x, num.x), fmaf_(hw, cq.y, num.y), fmaf_(hw, cq.z, num.z)};  // :146
      den = den + hw;                                                                    // :147
      if (use_var) vsum = fmaf_(hw * hw, a.var_in[rowq + qx], vsum);
    }
  }
  const f3 filtered = f3{num.x / den, num.y / den, num.z / den};  // :150
  if (use_var && a.var_out) a.var_out[rowp + x] = vsum / (den * den);
  if (!FINAL) {
    a.out[rowp + x] = make_float4(filtered.x, filtered.y, filtered.z, a.alpha_zero ? 0.0f : dp);
    return;
  }
  int ppx, ppy;
  reproject_pixel(W, H, a.PVprev, idp, xyz(a.worldpos[rowp + x]), a.lut_prev, x, y, ppx, ppy);
  if (a.prev_pixel) a.prev_pixel[rowp + x] = make_int2(ppx, ppy);
  bool use_history = a.frame > 0;  // :251
  const bool inside = ppx >= 0 && ppx < W && ppy >= 0 && ppy < H;
  if (use_history && (a.ext & kExtDisocclusion)) {
    // same primitive at the reprojected pixel; rows this context does not hold count as disoccluded
    use_history = inside && ppy >= a.pvis_y0 && ppy < a.pvis_y1 &&
                  a.prev_vis[static_cast<size_t>(ppy - a.pvis_row_base) * W + ppx] == idp;
  }
  f3 blend = filtered;  // :258
  if (use_history) {
    f3 hc{0.f, 0.f, 0.f};  // D2
    if (ppx >= 0 && ppx < W && ppy >= a.hist_y0 && ppy < a.hist_y1)
      hc = xyz(a.history[static_cast<size_t>(ppy - a.hist_row_base) * W + ppx]);
    float alpha = a.alpha, oma = 1.0f - a.alpha;
    if (a.ext & kExtAdaptiveAlpha) {  // :247-248
      const float g = a.gradient[rowp + x].x;
      alpha = fmaf_(1.0f - g, alpha, g);
      oma = 1.0f - alpha;
    }
    blend = f3{fmaf_(filtered.x, alpha, hc.x * oma), fmaf_(filtered.y, alpha, hc.y * oma), fmaf_(filtered.z, alpha, hc.z * oma)};  // :254
  }
  a.out[rowp + x] = make_float4(blend.x, blend.y, blend.z, 0.0f);
}



// "Comb" kernel.
//
// A wave owns a 64-px-wide column segment and kCombM output rows spaced k apart: y_m = yc + m*k + r
// (a comb of residue r).  Their tap rows y_m - k, y_m, y_m + k are again rows of the comb, so each
// staged row serves up to three outputs.  Rows (64 + 2k px: colour+depth cell, id) are staged into
// LDS with LDS-DMA; the horizontal taps x-k / x / x+k are the same staged row read at three lane
// offsets: 2 vector-memory instructions per staged row instead of 27 per pixel.
//
// Normal weights pow(max(0, dot(n_p, n_q)), sigma_n) (:62) depend only on the id pair, so k_lut
// tabulates them once per frame ((T+1)^2 floats, same arithmetic) and the block copies the table to
// LDS: one ds_read_b32 per tap replaces the compare/branch/gather/pow sequence.
#ifndef RTPT_COMB_M
#define RTPT_COMB_M 2
#endif
constexpr int kCombM = RTPT_COMB_M;  // output rows per wave
constexpr int kPairMax = 64;    // ids (T+1) for which the pair table is kept in LDS

// The kernel.  The kShWaves waves of a block filter kShWaves CONSECUTIVE chunks of one comb: 4M outputs
// per column whose tap rows are the 4M+2 rows yg + (j-1)k.  Each distinct row is staged by exactly
// one wave into a block-shared LDS region and a workgroup barrier publishes it.  (A first version
// staged M+2 rows per wave privately — no barrier at all — but then a block fetches 4(M+2) rows of
// which only 4M+2 are distinct, and the duplicates are in flight at the same time so L2 serves few
// of them: PMC showed 1.74x the algorithmic read bytes reaching the fabric, and at 5.6 TB/s the
// kernel was fabric-bound at 72-76 us.  Sharing brought it to 66-69 us.  Sweep at 4K, k = 5:
// (M, waves, 64-px segments per wave row) = (2,4,1) 67 us, (3,4,1) 74, (2,8,1) 68, (1,8,1) 71,
// (2,4,2) 66-77, (2,8,2) 72-88.)
#ifndef RTPT_COMB_WAVES
#define RTPT_COMB_WAVES 4
#endif
#ifndef RTPT_COMB_PRIO
#define RTPT_COMB_PRIO 1  // waves run at priority 3 while they issue an item's DMAs: 4K 67.0 -> 65.5 us (in-process A/B)
#endif
#ifndef RTPT_COMB_NT_STORE
#define RTPT_COMB_NT_STORE 1  // non-temporal stores for the k < N passes (the pass does not re-read its output): 65 -> 62.5 us
                              // in a frame; a pass repeated on the SAME buffers drops to 54 us (77 %) because its input then stays
                              // in the 256 MB memory-side cache — in a frame the input was just written by the previous pass.
                              // Tried without gain: nt on the final pass's store, nt on the staging loads (66-69 us),
                              // alternating the walk direction per pass so a pass starts on the rows written last
#endif
#ifndef RTPT_COMB_HALVES
#define RTPT_COMB_HALVES 1
#endif
constexpr int kShWaves = RTPT_COMB_WAVES;    // waves (consecutive chunks) per block
constexpr int kShHalves = RTPT_COMB_HALVES;  // 64-px segments per wave row: the 2k-column halo is paid once per 64*kShHalves px
constexpr int kShThreads = 64 * kShWaves;

// NRM = true is the variant for scenes whose id-pair table does not fit LDS (more than 63 triangles, i.e. every
// scene but the Cornell box): instead of the 4-byte id, each staged cell carries the pixel's normal — a second
// 16-byte plane written by k_gbuffer (normal_tab[id]: n.xyz and the self weight) — and the normal weight is computed
// per tap from the two staged normals with the same arithmetic as the direct kernel.  32 instead of 20 staged bytes
// per pixel, but 3 vector-memory instructions per staged row instead of the direct kernel's 27 per pixel.
// R / EXTA: the extension modes that only change the TAPS (RTPT_FLAG_EXT_GAUSS5: radius R = 2 with the gaussianKernel2D
// weights the reference declares and never uses, temporalFiltering.comp.glsl:93-99; RTPT_FLAG_EXT_POW2_STRIDE: stride
// 2^(k-1), passed as a.stride) run in this kernel too — the same comb staging with 2R halo rows per workgroup and 2R*s
// halo columns per segment, and the arithmetic of k_atrous_ext (h kept per tap, correctly-rounded final division), so
// the two are bit-identical.  25 taps are 25 LDS reads here instead of 75 global loads per pixel.
#ifndef RTPT_FINAL_WAVES
#define RTPT_FINAL_WAVES 1  // id-pair final pass: waves per SIMD to squeeze the registers for (A/B: 4 by itself)
#endif
#ifndef RTPT_COMB_MIN_WAVES
#define RTPT_COMB_MIN_WAVES 0
#endif
template <int CWp, bool FINAL, bool EXACT, bool NRM = false, int R = 1, bool EXTA = false, bool VAR = false>
__global__ __launch_bounds__(kShThreads)
#if RTPT_COMB_MIN_WAVES
__attribute__((amdgpu_waves_per_eu(RTPT_COMB_MIN_WAVES)))
#else
// the per-pixel-normal final pass sits at the edge of six waves per SIMD (79-81 VGPRs as the surrounding code changes;
// 142-147 us with six waves, 169-176 us with five at 4K): pin it.  Every other instantiation keeps what it gets.
__attribute__((amdgpu_waves_per_eu(FINAL && NRM && R == 1 && !EXTA && !VAR && !EXACT ? 6 : (FINAL && !NRM && R == 1 && !EXTA && !VAR && !EXACT ? RTPT_FINAL_WAVES : 1))))
#endif
void k_atrous_comb_sh(AtrousArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
#if RTPT_TILE_TIMELINE
  SpanScope span_(FINAL ? 1u : 0u, false);
#endif
  const int W = a.g.W, H = a.g.H, k = a.stride;  // a.stride == a.k (main.cpp:1259-1260, :135) unless POW2_STRIDE
  constexpr int rows = kShWaves * kCombM + 2 * R, cells = rows * CWp;  // block-shared rows
  const int NP = NRM ? 0 : static_cast<int>(a.n_tris) + 1;
  const int lane = static_cast<int>(threadIdx.x);
  const int wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.y));
  float* pairw = reinterpret_cast<float*>(lds_raw);  // [NP][NP]
  const int pair_bytes = (NP * NP * 4 + 15) & ~15;
  unsigned char* mine = lds_raw + pair_bytes;  // one region per block
  const float4* col = reinterpret_cast<const float4*>(mine);                // (r, g, b, depth)
  const uint32_t* ids = reinterpret_cast<const uint32_t*>(mine + 16 * cells);
  const float4* nrm = reinterpret_cast<const float4*>(mine + 16 * cells);   // NRM: (n.xyz, self weight) instead of ids
  const uint32_t lds0 = static_cast<uint32_t>(reinterpret_cast<size_t>((__attribute__((address_space(3))) unsigned char*)mine));
  const uint32_t lds_col = lds0, lds_ids = lds0 + 16u * static_cast<uint32_t>(cells);
  // EXTA with RTPT_FLAG_EXT_VARIANCE: the variance plane this iteration reads is staged as a third plane (4 B per cell)
  constexpr bool use_var = EXTA && !NRM && VAR;  // the launch picks VAR iff (a.ext & kExtVariance) && a.var_in
  const float* varp = reinterpret_cast<const float*>(mine + 20 * cells);
  const uint32_t lds_var = lds0 + 20u * static_cast<uint32_t>(cells);

  // id-pair weight table -> LDS (plain loads; no DMA is in flight yet)
  for (int i = wave * 64 + lane; i < NP * NP; i += kShThreads) pairw[i] = a.pair_tab[i];
  __syncthreads();

  // Work list.  A logical block = four CONSECUTIVE chunks (one per wave) of one residue and one
  // column.  Each XCD (physical blocks b, b+8, ...) owns a contiguous eighth of the list; see
  // the work list below for the order inside it.  The grid is persistent: the pair table is loaded once
  // per block, not per work item.  Speed only, never correctness: any mapping filters every pixel exactly
  // once.  (Tried on top of order 1 and dropped: an L2 prefetch of the block's next item, one dword per
  // 128-byte line — 68 -> 82 us; the memory system is saturated by requests, not starved of them.)
  const uint32_t per_res = static_cast<uint32_t>(a.tiles_x) * static_cast<uint32_t>(a.tiles_y);  // tiles_y = chunk groups
  const uint32_t nlb = per_res * static_cast<uint32_t>(k);
  const uint32_t xcd = blockIdx.x & 7u, jx = blockIdx.x >> 3, per_xcd = gridDim.x >> 3;
  const uint32_t x_lo = static_cast<uint32_t>((static_cast<uint64_t>(nlb) * xcd) >> 3);
  const uint32_t x_hi = static_cast<uint32_t>((static_cast<uint64_t>(nlb) * (xcd + 1)) >> 3);
  const uint32_t lb_lo = x_lo + static_cast<uint32_t>((static_cast<uint64_t>(x_hi - x_lo) * jx) / per_xcd);
  const uint32_t lb_hi = x_lo + static_cast<uint32_t>((static_cast<uint64_t>(x_hi - x_lo) * (jx + 1)) / per_xcd);
  const int row_lo = a.g.row_base, row_hi = a.g.row_base + a.rows_stored - 1;
  const bool tail_lane = lane < 2 * R * k;  // columns 64 .. 64+2Rk-1
  const float h = 1.0f / 9.0f;  // :145
  // (residue, column, chunk group) of lb_lo, then advanced incrementally (scalar adds, no divisions)
  int r = static_cast<int>(lb_lo / per_res);
  int bx, cg;
  {
    const uint32_t rem = lb_lo - static_cast<uint32_t>(r) * per_res;
    bx = static_cast<int>(rem / static_cast<uint32_t>(a.tiles_y));
    cg = static_cast<int>(rem - static_cast<uint32_t>(bx) * static_cast<uint32_t>(a.tiles_y));
  }
  r = __builtin_amdgcn_readfirstlane(r);
  bx = __builtin_amdgcn_readfirstlane(bx);
  cg = __builtin_amdgcn_readfirstlane(cg);

  // row-major list (residue, chunk group, column), dealt to the XCD's blocks item by item: the blocks
  // resident on an XCD work on a few consecutive row bands at any time, so the column halos of x-neighbours
  // and the rows shared by consecutive bands meet in that XCD's L2 (PMC: 245 -> 200 MB fetched per 4K
  // launch against round 2's order, each block walking down a column; 4K 66-72 -> 64-68 us, 1080p 20.5 -> 18.5 us)
  (void)lb_lo; (void)lb_hi; (void)r; (void)bx; (void)cg;
#pragma unroll 1
  for (uint32_t lb = x_lo + jx; lb < x_hi; lb += per_xcd) {
  const uint32_t r_u = lb / per_res, rem_u = lb - r_u * per_res;
  const uint32_t cg_u = rem_u / static_cast<uint32_t>(a.tiles_x);
  const int r_now = __builtin_amdgcn_readfirstlane(static_cast<int>(r_u));
  const int cg_now = __builtin_amdgcn_readfirstlane(static_cast<int>(cg_u));
  const int bx_now = __builtin_amdgcn_readfirstlane(static_cast<int>(rem_u - cg_u * static_cast<uint32_t>(a.tiles_x)));
  const int yg = a.g.y0 + cg_now * (kShWaves * kCombM * k) + r_now;  // first output row of the group
  if (yg >= a.g.y1) continue;                                      // block-uniform
  const int x0 = bx_now * (kBlockX * kShHalves);
  uint32_t o16[kShHalves + 1], o4[kShHalves + 1];  // per-lane source offsets of the full chunks and the tail chunk
#pragma unroll
  for (int hf = 0; hf <= kShHalves; hf++) {
    int gx = x0 - R * k + hf * 64 + lane;
    gx = gx < 0 ? 0 : (gx > W - 1 ? W - 1 : gx);  // :136
    o16[hf] = static_cast<uint32_t>(gx) * 16u;
    o4[hf] = static_cast<uint32_t>(gx) * 4u;
  }
  // every wave is done reading the previous group's rows before anybody overwrites them
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
#if RTPT_COMB_PRIO
  __builtin_amdgcn_s_setprio(3);  // issue this item's DMAs ahead of the other waves' arithmetic
#endif
  // wave w stages rows w*M+1 .. w*M+M; wave 0 also row 0, the last wave also row 4M+1
  const int j_lo = wave * kCombM + (wave == 0 ? 0 : R);
  const int j_hi = wave * kCombM + kCombM + R - 1 + (wave == kShWaves - 1 ? R : 0);
  for (int j = j_lo; j <= j_hi; j++) {
    int gy = yg + (j - R) * k;
    gy = gy < 0 ? 0 : (gy > H - 1 ? H - 1 : gy);              // :136
    gy = gy < row_lo ? row_lo : (gy > row_hi ? row_hi : gy);  // rows only masked outputs could reach
    const size_t grow = static_cast<size_t>(gy - a.g.row_base) * W;  // wave-uniform
    const float4* rin = a.in + grow;
    const uint32_t* rvis = a.vis + grow;
    const float4* rnrm = NRM ? a.normals + grow : nullptr;
    const float* rvar = use_var ? a.var_in + grow : nullptr;
    const uint32_t cj = static_cast<uint32_t>(j * CWp);
#pragma unroll
    for (int hf = 0; hf < kShHalves; hf++) {
      dma_b128(rin, o16[hf], lds_col + (cj + 64u * hf) * 16u);
      if (NRM)
        dma_b128(rnrm, o16[hf], lds_ids + (cj + 64u * hf) * 16u);
      else
        dma_b32(rvis, o4[hf], lds_ids + (cj + 64u * hf) * 4u);
      if (use_var) dma_b32(rvar, o4[hf], lds_var + (cj + 64u * hf) * 4u);
    }
    if (tail_lane) {
      dma_b128(rin, o16[kShHalves], lds_col + (cj + 64u * kShHalves) * 16u);
      if (NRM)
        dma_b128(rnrm, o16[kShHalves], lds_ids + (cj + 64u * kShHalves) * 16u);
      else
        dma_b32(rvis, o4[kShHalves], lds_ids + (cj + 64u * kShHalves) * 4u);
      if (use_var) dma_b32(rvar, o4[kShHalves], lds_var + (cj + 64u * kShHalves) * 4u);
    }
  }
#if RTPT_COMB_PRIO
  __builtin_amdgcn_s_setprio(0);
#endif
  // final pass: world positions fetched with the staging DMA and the history ahead of the taps.  Measured at 4K: it pays
  // in the per-pixel-normal variant (1.15M triangles: 155.8 -> 144.1 us; its taps compute x^128 per tap and leave time to
  // hide the fetch) and costs in the id-pair variant (110.5 -> 125.6 us: the 2 KiB of world positions per wave lengthen the
  // wait in front of the workgroup barrier, which the other workgroups' taps no longer cover)
#ifndef RTPT_FINAL_EARLY
#define RTPT_FINAL_EARLY NRM
#endif
  // FINAL: the world positions of this wave's output pixels depend on nothing staged — fetch them now, in flight together
  // with the DMA and waited for by the same vmcnt(0)
  f3 wp_pre[kCombM * kShHalves];
  if (FINAL && RTPT_FINAL_EARLY) {
#pragma unroll
    for (int mh = 0; mh < kCombM * kShHalves; mh++) {
      const int m = mh / kShHalves, hf = mh % kShHalves;
      const int x = x0 + hf * 64 + lane;
      const int y = yg + (wave * kCombM + m) * k;
      wp_pre[mh] = f3{0.f, 0.f, 0.f};
      if (x < W && y < a.g.y1) wp_pre[mh] = xyz(a.worldpos[static_cast<size_t>(y - a.g.row_base) * W + x]);
    }
  }
  // own DMA landed, then the barrier publishes every wave's rows to the block
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

#pragma unroll
  for (int mh = 0; mh < kCombM * kShHalves; mh++) {
    const int m = mh / kShHalves, hf = mh % kShHalves;
    const int x = x0 + hf * 64 + lane;
    const int y = yg + (wave * kCombM + m) * k;
    if (x >= W || y >= a.g.y1) continue;
    const int cc = (wave * kCombM + m + R) * CWp + hf * 64 + lane + R * k;
    const float4 cp4 = col[cc];
    const f3 cp = xyz(cp4);
    const float dp = cp4.w;
    const size_t ip = static_cast<size_t>(y - a.g.row_base) * W + x;
    const float4 np4 = NRM ? nrm[cc] : make_float4(0.f, 0.f, 0.f, 0.f);
    const f3 np = xyz(np4);
    const uint32_t idp = NRM ? (FINAL ? a.vis[ip] : 0u) : ids[cc];  // NRM: the id is only needed for the reprojection
    const float* prow = pairw + idp * NP;
    const float wself = NRM ? np4.w : prow[idp];
    // FINAL: reproject first (:213-239 — exact arithmetic: the truncated pixel coordinate is an integer observable) so that
    // the history fetch is in flight under the taps' arithmetic
    int ppx = x, ppy = y;
    f3 hc{0.f, 0.f, 0.f};  // D2: out-of-image history fetch returns 0
    if (FINAL && RTPT_FINAL_EARLY) {
      if (!(idp < 1)) {
        const f3 wp = wp_pre[mh];
        const f3 va = xyz(a.lut_prev[3 * idp]), vb = xyz(a.lut_prev[3 * idp + 1]), vc = xyz(a.lut_prev[3 * idp + 2]);  // :223-233
        const f3 bc = bary_coords(wp, va, vb, vc);
        const f3 wpp = bary_mix(bc, va, vb, vc);  // :236
        const float clx = exact::mat_row_point(a.PVprev, 0, wpp), cly = exact::mat_row_point(a.PVprev, 1, wpp),
                    clw = exact::mat_row_point(a.PVprev, 3, wpp);
        const float ndx = clx / clw, ndy = cly / clw;                      // :183
        ppx = exact::f2i(fmaf_(ndx, 0.5f, 0.5f) * static_cast<float>(W));  // :186,:238
        ppy = exact::f2i(fmaf_(ndy, 0.5f, 0.5f) * static_cast<float>(H));
      }
      if (a.frame > 0 && ppx >= 0 && ppx < W && ppy >= a.hist_y0 && ppy < a.hist_y1)  // :251,:253
        hc = xyz(a.history[static_cast<size_t>(ppy - a.hist_row_base) * W + ppx]);
    }
    f3 num{0.f, 0.f, 0.f};
    float den = 0.f, vsum = 0.f;
    // variance guidance (k_atrous_ext's arithmetic): the colour term compares luminances, scaled by the pixel's own deviation
    const float lum_p = use_var ? luminance(cp) : 0.0f;
    const float lum_scale = use_var ? fmaf_(a.sigma_l, exact::sqrt_(glsl_max(a.var_scale ? a.var_scale[ip] : varp[cc], 0.0f)), 1e-4f) : 1.0f;
    const float cl_var = -1.44269504088896341f * fast::rcp_(lum_scale);
#pragma unroll
    for (int i = -R; i <= R; i++) {  // :132 (x offset outer: the reference's accumulation order)
#pragma unroll
      for (int jj = -R; jj <= R; jj++) {  // :133
        float w;
        f3 cq;
        if (!EXTA && i == 0 && jj == 0) {
          cq = cp;
          w = wself;  // centre tap: q == p, both exponentials are exactly 1
        } else {
          const int qi = cc + jj * CWp + i * k;
          const float4 cq4 = col[qi];
          cq = xyz(cq4);
          const float dq = cq4.w;
          float wn;
          if (NRM) {
            // :62 (pow_sigma: with exact::powi's loop this variant was SALU-bound at 166 us)
            wn = pow_sigma(glsl_max(0.0f, exact::dot(np, xyz(nrm[qi]))), a.sigma_n);
          } else
            wn = prow[ids[qi]];  // :62 via the id-pair table
          const f3 dc = cp - cq;
          if (EXACT) {
            const float wd = exact::exp_(-__builtin_fabsf(dp - dq) / a.sigma_z);  // :67-68
            const float wl = (EXTA && use_var) ? exact::exp_(-__builtin_fabsf(lum_p - luminance(cq)) / lum_scale)
                                               : exact::exp_(-exact::length(dc) / a.sigma_l);         // :73
            w = (wn * wd) * wl;                                                   // :77
          } else {
            // exp(-|dz|/sz) * exp(-|dc|/sl) = exp2(|dz| * cz + |dc| * cl), cz/cl = -log2(e)/sigma
            const float dl = (EXTA && use_var) ? __builtin_fabsf(lum_p - luminance(cq)) * cl_var : fast::sqrt_(exact::dot(dc, dc)) * a.cl;
            const float e = fmaf_(__builtin_fabsf(dp - dq), a.cz, dl);
            w = wn * __builtin_amdgcn_exp2f(e);
          }
        }
        if (EXTA) {
          // k_atrous_ext's accumulation: the tap's own h (gaussianKernel2D / 273, :93-99, or 1/9, :145) stays in
          constexpr float kG5[5] = {1.f, 4.f, 7.f, 4.f, 1.f}, kG5m[5] = {4.f, 16.f, 26.f, 16.f, 4.f}, kG5c[5] = {7.f, 26.f, 41.f, 26.f, 7.f};
          const float g = R == 2 ? ((i == 0) ? kG5c[jj + 2] : ((i == -1 || i == 1) ? kG5m[jj + 2] : kG5[jj + 2])) : 9.0f;
          const float hh = R == 2 ? g * (1.0f / 273.0f) : 1.0f / 9.0f;
          const float hw = hh * w;
          num = f3{fmaf_(hw, cq.x, num.x), fmaf_(hw, cq.y, num.y), fmaf_(hw, cq.z, num.z)};  // :146
          den = den + hw;                                                                    // :147
          if (use_var) vsum = fmaf_(hw * hw, varp[cc + jj * CWp + i * k], vsum);
        } else if (EXACT) {
          const float hw = h * w;
          num = f3{fmaf_(hw, cq.x, num.x), fmaf_(hw, cq.y, num.y), fmaf_(hw, cq.z, num.z)};  // :146
          den = den + hw;                                                                    // :147
        } else {
          // h = 1/9 scales numerator and denominator alike; the fast path drops it
          num = f3{fmaf_(w, cq.x, num.x), fmaf_(w, cq.y, num.y), fmaf_(w, cq.z, num.z)};
          den = den + w;
        }
      }
    }
    f3 filtered;
    if (EXACT || EXTA) {
      filtered = f3{num.x / den, num.y / den, num.z / den};  // :150
    } else {
      filtered = num * fast::rcp_(den);
    }
    if (EXTA && use_var && a.var_out) a.var_out[ip] = vsum / (den * den);
    if (!FINAL) {
#if RTPT_COMB_NT_STORE
      {
        typedef float v4f_ __attribute__((ext_vector_type(4)));
        v4f_ o4 = {filtered.x, filtered.y, filtered.z, a.alpha_zero ? 0.0f : dp};
        __builtin_nontemporal_store(o4, reinterpret_cast<v4f_*>(a.out + ip));  // :152; not re-read by this pass
      }
#else
      a.out[ip] = make_float4(filtered.x, filtered.y, filtered.z, a.alpha_zero ? 0.0f : dp);  // :152 (+ depth in alpha)
#endif
      continue;
    }
    if (!RTPT_FINAL_EARLY) {
      // :213-239 reprojection — exact arithmetic: the truncated pixel coordinate is an integer observable
      if (!(idp < 1)) {
        const f3 wp = xyz(a.worldpos[ip]);
        const f3 va = xyz(a.lut_prev[3 * idp]), vb = xyz(a.lut_prev[3 * idp + 1]), vc = xyz(a.lut_prev[3 * idp + 2]);  // :223-233
        const f3 bc = bary_coords(wp, va, vb, vc);
        const f3 wpp = bary_mix(bc, va, vb, vc);  // :236
        const float clx = exact::mat_row_point(a.PVprev, 0, wpp), cly = exact::mat_row_point(a.PVprev, 1, wpp),
                    clw = exact::mat_row_point(a.PVprev, 3, wpp);
        const float ndx = clx / clw, ndy = cly / clw;                      // :183
        ppx = exact::f2i(fmaf_(ndx, 0.5f, 0.5f) * static_cast<float>(W));  // :186,:238
        ppy = exact::f2i(fmaf_(ndy, 0.5f, 0.5f) * static_cast<float>(H));
      }
      if (a.frame > 0 && ppx >= 0 && ppx < W && ppy >= a.hist_y0 && ppy < a.hist_y1)
        hc = xyz(a.history[static_cast<size_t>(ppy - a.hist_row_base) * W + ppx]);
    }
    if (a.prev_pixel) a.prev_pixel[ip] = make_int2(ppx, ppy);
    f3 blend = filtered;  // :258
    bool use_history = a.frame > 0;  // :251
    float alpha = a.alpha;
    if (EXTA) {  // the final pass of the extension modes (k_atrous_ext's epilogue)
      if (use_history && (a.ext & kExtDisocclusion)) {
        // same primitive at the reprojected pixel; rows this context does not hold count as disoccluded
        const bool inside = ppx >= 0 && ppx < W && ppy >= 0 && ppy < H;
        use_history = inside && ppy >= a.pvis_y0 && ppy < a.pvis_y1 &&
                      a.prev_vis[static_cast<size_t>(ppy - a.pvis_row_base) * W + ppx] == idp;
      }
      if (a.ext & kExtAdaptiveAlpha) {  // :247-248
        const float g = a.gradient[ip].x;
        alpha = fmaf_(1.0f - g, alpha, g);
      }
    }
    if (use_history) {
      const float oma = 1.0f - alpha;
      blend = f3{fmaf_(filtered.x, alpha, hc.x * oma), fmaf_(filtered.y, alpha, hc.y * oma),
                 fmaf_(filtered.z, alpha, hc.z * oma)};  // :254
    }
#if RTPT_COMB_NT_STORE > 1
    {
      typedef float v4f_ __attribute__((ext_vector_type(4)));
      v4f_ o4 = {blend.x, blend.y, blend.z, 0.0f};
      __builtin_nontemporal_store(o4, reinterpret_cast<v4f_*>(a.out + ip));  // :263 (D1: distinct buffer)
    }
#else
    a.out[ip] = make_float4(blend.x, blend.y, blend.z, 0.0f);  // :263 (D1: distinct buffer)
#endif
    // main.cpp:1338-1361, fused: the blit reads exactly the value stored above (alpha 0), k_present's conversion
    // (not in the per-pixel-normal variant: its final pass sits at 79 VGPRs = 6 waves per SIMD, and the store's operands
    // cost it a wave — 142 -> 176 us at 4K; k_present serves those scenes)
    if (!NRM && a.present && y >= a.present_y0 && y < a.present_y1)  // index = ip minus a wave-uniform row offset: no new per-lane address
      (a.present - static_cast<ptrdiff_t>(a.present_y0 - a.g.row_base) * W)[ip] = unorm8(blend.z) | (unorm8(blend.y) << 8) | (unorm8(blend.x) << 16);
  }
  }  // work list
}

// copy the G-buffer depth into the alpha channel of a colour plane (used when a plane was injected
// through rtpt_set_plane / rtpt_bind_plane and does not carry it yet)
__global__ __launch_bounds__(kThreads) void k_stamp_depth(FrameGeom g, float4* color, const float* depth) {
  const int x = blockIdx.x * kBlockX + threadIdx.x;
  const int y = g.y0 + blockIdx.y * kBlockY + threadIdx.y;
  if (x >= g.W || y >= g.y1) return;
  const size_t i = static_cast<size_t>(y - g.row_base) * g.W + x;
  reinterpret_cast<float*>(color + i)[3] = depth[i];
}

__global__ __launch_bounds__(kThreads) void k_present(FrameGeom g, const float4* __restrict__ image, uint32_t* __restrict__ dst) {
  const int x = blockIdx.x * kBlockX + threadIdx.x;
  const int y = g.y0 + blockIdx.y * kBlockY + threadIdx.y;
  if (x >= g.W || y >= g.y1) return;
  const float4 c = image[static_cast<size_t>(y - g.row_base) * g.W + x];
  dst[static_cast<size_t>(y - g.y0) * g.W + x] = unorm8(c.z) | (unorm8(c.y) << 8) | (unorm8(c.x) << 16) | (unorm8(c.w) << 24);
}

}  // namespace

#if RTPT_TILE_TIMELINE
#define RTPT_SPAN_HOST
#include "experiments/span_instrumentation.inc"
#undef RTPT_SPAN_HOST
#endif

void launch_present(const FrameGeom& g, const float4* image, uint32_t* dst, hipStream_t s) {
  if (g.y1 <= g.y0) return;
  hipLaunchKernelGGL(k_present, grid_for(g), dim3(kBlockX, kBlockY), 0, s, g, image, dst);
}

void launch_var_prefilter(const FrameGeom& g, int rows_stored, const float* var, float* out, hipStream_t s) {
  if (g.y1 <= g.y0) return;
  hipLaunchKernelGGL(k_var_prefilter, grid_for(g), dim3(kBlockX, kBlockY), 0, s, g, rows_stored, var, out);
}

void launch_moments(const MomentsArgs& a, hipStream_t s) {
  if (a.g.y1 <= a.g.y0) return;
  hipLaunchKernelGGL(k_moments, grid_for(a.g), dim3(kBlockX, kBlockY), 0, s, a);
}

void launch_stamp_depth(const FrameGeom& g, float4* color, const float* depth, hipStream_t s) {
  if (g.y1 <= g.y0) return;
  hipLaunchKernelGGL(k_stamp_depth, grid_for(g), dim3(kBlockX, kBlockY), 0, s, g, color, depth);
}

// Kernels that ask for more than 64 KiB of dynamic LDS need the attribute raised once per DEVICE (it is a
// property of the function on the current device, not of the process): rtpt_create calls this after
// hipSetDevice, so contexts on different GPUs of one process are independent.
template <int CW, bool NRM>
static hipError_t comb_attrs() {
  constexpr int kMax = 160 * 1024;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_atrous_comb_sh<CW, true, true, NRM>), hipFuncAttributeMaxDynamicSharedMemorySize, kMax);
  if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_atrous_comb_sh<CW, false, true, NRM>), hipFuncAttributeMaxDynamicSharedMemorySize, kMax);
  if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_atrous_comb_sh<CW, true, false, NRM>), hipFuncAttributeMaxDynamicSharedMemorySize, kMax);
  if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_atrous_comb_sh<CW, false, false, NRM>), hipFuncAttributeMaxDynamicSharedMemorySize, kMax);
  return e;
}
template <int CW, int RR, bool VV>
static hipError_t comb_ext_attrs() {
  constexpr int kMax = 160 * 1024;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_atrous_comb_sh<CW, false, true, false, RR, true, VV>), hipFuncAttributeMaxDynamicSharedMemorySize, kMax);
  if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_atrous_comb_sh<CW, false, false, false, RR, true, VV>), hipFuncAttributeMaxDynamicSharedMemorySize, kMax);
  if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_atrous_comb_sh<CW, true, true, false, RR, true, VV>), hipFuncAttributeMaxDynamicSharedMemorySize, kMax);
  if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_atrous_comb_sh<CW, true, false, false, RR, true, VV>), hipFuncAttributeMaxDynamicSharedMemorySize, kMax);
  return e;
}
hipError_t prepare_device_atrous() {
  {
    hipError_t e = comb_ext_attrs<72, 1, false>();
    if (e == hipSuccess) e = comb_ext_attrs<96, 1, false>();
    if (e == hipSuccess) e = comb_ext_attrs<72, 2, false>();
    if (e == hipSuccess) e = comb_ext_attrs<96, 2, false>();
    if (e == hipSuccess) e = comb_ext_attrs<128, 2, false>();
    if (e == hipSuccess) e = comb_ext_attrs<72, 1, true>();
    if (e == hipSuccess) e = comb_ext_attrs<96, 1, true>();
    if (e == hipSuccess) e = comb_ext_attrs<72, 2, true>();
    if (e == hipSuccess) e = comb_ext_attrs<96, 2, true>();
    if (e == hipSuccess) e = comb_ext_attrs<128, 2, true>();
    if (e != hipSuccess) return e;
  }
  constexpr int b = kBlockX * kShHalves;
  hipError_t e = comb_attrs<b + 8, false>();
  if (e == hipSuccess) e = comb_attrs<b + 16, false>();
  if (e == hipSuccess) e = comb_attrs<b + 32, false>();
  if (e == hipSuccess) e = comb_attrs<b + 8, true>();
  if (e == hipSuccess) e = comb_attrs<b + 16, true>();
  if (e == hipSuccess) e = comb_attrs<b + 32, true>();
  return e;
}

// true when a FINAL launch of `a` runs in the LDS-staged kernel, whose epilogue can also write the swapchain format
// (AtrousArgs::present); every other final variant leaves the blit to k_present
// the extension modes run LDS-staged (k_atrous_comb_sh<..., R, EXTA>) when the scene has an id-pair table and the halo
// 2 R s fits the widest staged row: every flag combination, final pass included (round 3; before: tap shapes of k < N only)
static bool ext_staged(const AtrousArgs& a) {
  const int np = static_cast<int>(a.n_tris) + 1;
  const int R = (a.ext & kExtGauss5) ? 2 : 1;
  // staged row strides instantiated: 72 / 96 for 3x3 taps (strides <= 16), 72 / 96 / 128 for 5x5 taps
  return a.ext && !a.direct && a.pair_tab && np <= kPairMax && a.stride >= 1 && kBlockX + 2 * R * a.stride <= (R == 1 ? 96 : 128);
}
bool atrous_final_fuses_present(const AtrousArgs& a) {
  const int np = static_cast<int>(a.n_tris) + 1;
  const bool pair_mode = a.pair_tab && np <= kPairMax;
  const bool nrm_mode = !pair_mode && a.normals != nullptr;
  if (a.ext) return ext_staged(a);
  (void)nrm_mode;  // the per-pixel-normal final pass does not carry the store (register budget, see the kernel)
  return !a.direct && pair_mode && a.k >= 1 && a.k <= 16;
}

// Workgroups per XCD of the persistent comb kernel: every resident slot, but no more workgroups than work items (an XCD takes an
// eighth of the list and deals it to its workgroups item by item).  Giving every workgroup the same number of items instead
// (1 050 workgroups x 2 items rather than 2 048 slots for the 2 100 items of a strip's final pass) was measured in round 4
// and changes nothing (strip final 22.0 -> 23.1 us, 4K 110.8 -> 106.8, within the spread): the launch is bound by bytes in
// flight, not by its last items.
static uint32_t comb_blocks_per_xcd(uint32_t nlb, uint32_t slots_per_xcd) {
  const uint32_t items_xcd = (nlb + 7u) / 8u;
  return slots_per_xcd > items_xcd ? (items_xcd ? items_xcd : 1u) : (slots_per_xcd ? slots_per_xcd : 1u);
}

void launch_atrous(const AtrousArgs& a0, bool final_pass, hipStream_t s) {
  if (a0.g.y1 <= a0.g.y0) return;
  AtrousArgs a = a0;
  a.cz = -1.44269504088896341f / a.sigma_z;
  a.cl = -1.44269504088896341f / a.sigma_l;
  dim3 block(kBlockX, kBlockY);
  const int np = static_cast<int>(a.n_tris) + 1;
  {
    // the extension modes run LDS-staged when the scene has an id-pair table and the halo 2*R*s fits a staged row the
    // kernel is instantiated for — tap shapes, variance guidance and the final pass's adaptive alpha / disocclusion test
    // alike (round 3); what does not fit runs in the generic direct-load kernel k_atrous_ext (same arithmetic)
    const int R = (a.ext & kExtGauss5) ? 2 : 1, sk = a.stride;
    if (ext_staged(a)) {
      const int seg_w = kBlockX;
      a.tiles_x = (a.g.W + seg_w - 1) / seg_w;
      const int nrows = a.g.y1 - a.g.y0;
      const int chunks = (nrows + kCombM * sk - 1) / (kCombM * sk);
      a.tiles_y = (chunks + kShWaves - 1) / kShWaves;
      const int n_cu = a.n_cu > 0 ? a.n_cu : 256;
      const uint32_t nlb = static_cast<uint32_t>(a.tiles_x) * static_cast<uint32_t>(a.tiles_y) * static_cast<uint32_t>(sk);
      const int need = seg_w + 2 * R * sk;
      const int cw = need <= 72 ? 72 : (need <= 96 ? 96 : 128);  // staged row strides the kernel is instantiated for
      const bool use_var = (a.ext & kExtVariance) && a.var_in;
      const size_t lds = static_cast<size_t>((np * np * 4 + 15) & ~15) + static_cast<size_t>(kShWaves * kCombM + 2 * R) * cw * (use_var ? 24 : 20);
      uint32_t per_cu = static_cast<uint32_t>((160u * 1024u) / lds);
      if (per_cu > 32u / kShWaves) per_cu = 32u / kShWaves;
      if (per_cu < 1u) per_cu = 1u;
      const uint32_t per_xcd = comb_blocks_per_xcd(nlb, static_cast<uint32_t>((n_cu + 7) / 8) * per_cu);
      const dim3 grid(per_xcd * 8u), sblock(kBlockX, kShWaves);
#define RTPT_LAUNCH_EXT(CW, RR, VV)                                                                                 \
  do {                                                                                                              \
    if (final_pass) {                                                                                               \
      if (a.exact)                                                                                                  \
        hipLaunchKernelGGL((k_atrous_comb_sh<CW, true, true, false, RR, true, VV>), grid, sblock, lds, s, a);      \
      else                                                                                                          \
        hipLaunchKernelGGL((k_atrous_comb_sh<CW, true, false, false, RR, true, VV>), grid, sblock, lds, s, a);     \
    } else if (a.exact)                                                                                             \
      hipLaunchKernelGGL((k_atrous_comb_sh<CW, false, true, false, RR, true, VV>), grid, sblock, lds, s, a);       \
    else                                                                                                            \
      hipLaunchKernelGGL((k_atrous_comb_sh<CW, false, false, false, RR, true, VV>), grid, sblock, lds, s, a);      \
  } while (0)
#define RTPT_LAUNCH_EXT_CW(RR, VV)                   \
  do {                                               \
    if (cw == 72) RTPT_LAUNCH_EXT(72, RR, VV);       \
    else if (cw == 96) RTPT_LAUNCH_EXT(96, RR, VV);  \
    else RTPT_LAUNCH_EXT(128, RR, VV);               \
  } while (0)
      if (R == 2) {
        if (use_var) RTPT_LAUNCH_EXT_CW(2, true); else RTPT_LAUNCH_EXT_CW(2, false);
      } else if (cw <= 96) {  // R == 1: 64 + 2 s <= 96 for every stride the staged kernel takes (s <= 16)
        if (use_var) {
          if (cw == 72) RTPT_LAUNCH_EXT(72, 1, true); else RTPT_LAUNCH_EXT(96, 1, true);
        } else {
          if (cw == 72) RTPT_LAUNCH_EXT(72, 1, false); else RTPT_LAUNCH_EXT(96, 1, false);
        }
      }
#undef RTPT_LAUNCH_EXT_CW
#undef RTPT_LAUNCH_EXT
      return;
    }
  }
  if (a.ext) {
    const dim3 g2 = grid_for(a.g);
    a.tiles_x = static_cast<int32_t>(g2.x);
    a.tiles_y = static_cast<int32_t>(g2.y);
    dim3 grid(g2.x * g2.y);
    if (a.exact) {
      if (final_pass)
        hipLaunchKernelGGL((k_atrous_ext<true, true>), grid, block, 0, s, a);
      else
        hipLaunchKernelGGL((k_atrous_ext<false, true>), grid, block, 0, s, a);
    } else {
      if (final_pass)
        hipLaunchKernelGGL((k_atrous_ext<true, false>), grid, block, 0, s, a);
      else
        hipLaunchKernelGGL((k_atrous_ext<false, false>), grid, block, 0, s, a);
    }
    return;
  }
  const bool pair_mode = a.pair_tab && np <= kPairMax;
  const bool nrm_mode = !pair_mode && a.normals != nullptr;
  if (!a.direct && (pair_mode || nrm_mode) && a.k >= 1 && a.k <= 16) {
    const int seg_w = kBlockX * kShHalves;
    a.tiles_x = (a.g.W + seg_w - 1) / seg_w;
    const int nrows = a.g.y1 - a.g.y0;
    const int chunks = (nrows + kCombM * a.k - 1) / (kCombM * a.k);
    a.tiles_y = (chunks + kShWaves - 1) / kShWaves;  // chunk groups (kShWaves consecutive chunks per block)
    const int n_cu = a.n_cu > 0 ? a.n_cu : 256;  // of the context's device (rtpt_create)
    const uint32_t nlb = static_cast<uint32_t>(a.tiles_x) * static_cast<uint32_t>(a.tiles_y) * static_cast<uint32_t>(a.k);
    // staged row stride (cells): segment + 2k, rounded to the template instances
    const int need = seg_w + 2 * a.k;
    const int base_w = seg_w;
    const int cw = need <= base_w + 8 ? base_w + 8 : (need <= base_w + 16 ? base_w + 16 : base_w + 32);
    const size_t lds = nrm_mode ? static_cast<size_t>(kShWaves * kCombM + 2) * cw * 32
                                : static_cast<size_t>((np * np * 4 + 15) & ~15) + static_cast<size_t>(kShWaves * kCombM + 2) * cw * 20;
    // persistent grid: as many blocks per CU as 160 KiB of LDS and 32 waves admit
    uint32_t per_cu = static_cast<uint32_t>((160u * 1024u) / lds);
    if (per_cu > 32u / kShWaves) per_cu = 32u / kShWaves;
    if (per_cu < 1u) per_cu = 1u;
    const uint32_t per_xcd = comb_blocks_per_xcd(nlb, static_cast<uint32_t>((n_cu + 7) / 8) * per_cu);
    dim3 grid(per_xcd * 8u), sblock(kBlockX, kShWaves);
#define RTPT_LAUNCH_COMB(CW, NRM)                                                                          \
  do {                                                                                                \
    if (a.exact) {                                                                                    \
      if (final_pass)                                                                                 \
        hipLaunchKernelGGL((k_atrous_comb_sh<CW, true, true, NRM>), grid, sblock, lds, s, a);              \
      else                                                                                            \
        hipLaunchKernelGGL((k_atrous_comb_sh<CW, false, true, NRM>), grid, sblock, lds, s, a);             \
    } else {                                                                                          \
      if (final_pass)                                                                                 \
        hipLaunchKernelGGL((k_atrous_comb_sh<CW, true, false, NRM>), grid, sblock, lds, s, a);             \
      else                                                                                            \
        hipLaunchKernelGGL((k_atrous_comb_sh<CW, false, false, NRM>), grid, sblock, lds, s, a);            \
    }                                                                                                 \
  } while (0)
    if (nrm_mode) {
      if (cw == base_w + 8)
        RTPT_LAUNCH_COMB(kBlockX * kShHalves + 8, true);
      else if (cw == base_w + 16)
        RTPT_LAUNCH_COMB(kBlockX * kShHalves + 16, true);
      else
        RTPT_LAUNCH_COMB(kBlockX * kShHalves + 32, true);
    } else {
      if (cw == base_w + 8)
        RTPT_LAUNCH_COMB(kBlockX * kShHalves + 8, false);
      else if (cw == base_w + 16)
        RTPT_LAUNCH_COMB(kBlockX * kShHalves + 16, false);
      else
        RTPT_LAUNCH_COMB(kBlockX * kShHalves + 32, false);
    }
#undef RTPT_LAUNCH_COMB
    return;
  }
  const dim3 g2 = grid_for(a.g);
  a.tiles_x = static_cast<int32_t>(g2.x);
  a.tiles_y = static_cast<int32_t>(g2.y);
  dim3 grid(g2.x * g2.y);
  if (a.exact) {
    if (final_pass)
      hipLaunchKernelGGL((k_atrous<true, true>), grid, block, 0, s, a);
    else
      hipLaunchKernelGGL((k_atrous<false, true>), grid, block, 0, s, a);
  } else {
    if (final_pass)
      hipLaunchKernelGGL((k_atrous<true, false>), grid, block, 0, s, a);
    else
      hipLaunchKernelGGL((k_atrous<false, false>), grid, block, 0, s, a);
  }
}

}  // namespace rt
