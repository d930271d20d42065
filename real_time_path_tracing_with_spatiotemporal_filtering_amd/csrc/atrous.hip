// atrous.hip — K3: one iteration of the edge-stopping a-trous filter
// (temporalFiltering.comp.glsl:191-265): 3x3 taps at stride k, h = 1/9, weights on normal / depth /
// colour; the FINAL variants fuse reprojection + temporal blend (:213-263).
//
//   k_atrous_comb   the shipping kernel: wave-private LDS staging by LDS-DMA, comb row assignment
//   k_atrous        direct global-load kernel: fallback for scenes whose id-pair weight table does
//                   not fit LDS (> 63 triangles) and for RTPT_FLAG_DIRECT_FILTER
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

namespace rt {
namespace {

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
          wn = exact::powi(glsl_max(0.0f, exact::dot(np, nq)), a.sigma_n);  // :62
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
    a.out[rowp + x] = make_float4(filtered.x, filtered.y, filtered.z, dp);  // :152 (+ depth in alpha)
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


// LDS-DMA issued from inline asm.  hipcc models `__builtin_amdgcn_global_load_lds` as an LDS store and
// puts `s_waitcnt vmcnt(0)` in front of every later ds_read that may alias it — with a ring buffer
// that is every read, which drains the prefetch each step.  Hidden in asm, the DMA is invisible to
// that pass and is ordered by hand: counted vmcnt + s_barrier before the reads (below).  M0 carries
// the wave-uniform LDS byte address; lane i lands at M0 + i*size.  One wait state is required between
// the SALU write of M0 and the LDS-DMA that reads it.
// `base` is a wave-uniform pointer (SGPR pair), `voff` the per-lane byte offset: no 64-bit VALU math.
__device__ __forceinline__ void dma_b128(const void* base, uint32_t voff, uint32_t lds_addr) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(base), "s"(lds_addr) : "memory");
}
__device__ __forceinline__ void dma_b32(const void* base, uint32_t voff, uint32_t lds_addr) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1" ::"v"(voff), "s"(base), "s"(lds_addr) : "memory");
}


// Wave-private "comb" kernel.
//
// A wave owns a 64-px-wide column segment and kCombM (= 3; 2/3/4 measured 78/72/72 us at 4K) output rows spaced k apart: y_m = yc + m*k + r.
// Their tap rows y_m - k, y_m, y_m + k are the kCombM + 2 rows yc + (j-1)*k + r, j = 0..M+1, so each
// staged row serves up to three outputs (2 rows fetched per output at M = 2 instead of 3).  The wave
// stages those rows (64 + 2k px: colour, depth, id) into its OWN slice of LDS with LDS-DMA, waits for
// its own vmcnt, and filters — no workgroup barrier anywhere after the prologue, so waves drift apart
// and loads, LDS reads and VALU work of different waves overlap.  The horizontal taps x-k / x / x+k
// are the same staged row read at three lane offsets: 3 vector-memory instructions per staged row
// instead of 27 per pixel.
//
// Normal weights pow(max(0, dot(n_p, n_q)), sigma_n) (:62) depend only on the id pair, so k_lut
// tabulates them once per frame ((T+1)^2 floats, same arithmetic) and the block copies the table to
// LDS: one ds_read_b32 per tap replaces the compare/branch/gather/pow sequence.
#ifndef RTPT_COMB_M
#define RTPT_COMB_M 3
#endif
constexpr int kCombM = RTPT_COMB_M;  // output rows per wave
constexpr int kPairMax = 64;    // ids (T+1) for which the pair table is kept in LDS

template <int CWp, bool FINAL, bool EXACT>  // CWp: staged row stride in cells, >= 64 + 2k (compile time:
__global__ __launch_bounds__(kThreads) void k_atrous_comb(AtrousArgs a) {  // tap rows become ds_read immediates)
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  const int W = a.g.W, H = a.g.H, k = a.k;
  constexpr int rows = kCombM + 2, cells = rows * CWp;
  const int NP = static_cast<int>(a.n_tris) + 1;
  const int lane = static_cast<int>(threadIdx.x);
  const int wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.y));
  float* pairw = reinterpret_cast<float*>(lds_raw);  // [NP][NP]
  const int pair_bytes = (NP * NP * 4 + 15) & ~15;
  unsigned char* mine = lds_raw + pair_bytes + wave * (cells * 20);
  const float4* col = reinterpret_cast<const float4*>(mine);                // (r, g, b, depth)
  const uint32_t* ids = reinterpret_cast<const uint32_t*>(mine + 16 * cells);
  const uint32_t lds0 = static_cast<uint32_t>(reinterpret_cast<size_t>((__attribute__((address_space(3))) unsigned char*)mine));
  const uint32_t lds_col = lds0, lds_ids = lds0 + 16u * static_cast<uint32_t>(cells);

  // id-pair weight table -> LDS (plain loads; no DMA is in flight yet)
  for (int i = wave * 64 + lane; i < NP * NP; i += kThreads) pairw[i] = a.pair_tab[i];
  __syncthreads();

  // Work list.  A logical block = four CONSECUTIVE chunks (one per wave) of one residue and one
  // column: chunk c and c+1 share two of their four staged rows, so the second fetch is an L1/L2 hit.
  // Logical blocks are ordered (residue, column, chunk group); each XCD (physical blocks b, b+8, ...)
  // owns a contiguous eighth of that order and each of its resident blocks a contiguous run of it, so
  // consecutive iterations of a block walk down one column and re-use the rows they share.  The grid
  // is persistent (<= 5 blocks per CU): the pair table is loaded once per block, not per comb.
  // Speed only, never correctness: any mapping filters every pixel exactly once.
  const uint32_t per_res = static_cast<uint32_t>(a.tiles_x) * static_cast<uint32_t>(a.tiles_y);  // tiles_y = chunk groups
  const uint32_t nlb = per_res * static_cast<uint32_t>(k);
  const uint32_t xcd = blockIdx.x & 7u, jx = blockIdx.x >> 3, per_xcd = gridDim.x >> 3;
  const uint32_t x_lo = static_cast<uint32_t>((static_cast<uint64_t>(nlb) * xcd) >> 3);
  const uint32_t x_hi = static_cast<uint32_t>((static_cast<uint64_t>(nlb) * (xcd + 1)) >> 3);
  const uint32_t lb_lo = x_lo + static_cast<uint32_t>((static_cast<uint64_t>(x_hi - x_lo) * jx) / per_xcd);
  const uint32_t lb_hi = x_lo + static_cast<uint32_t>((static_cast<uint64_t>(x_hi - x_lo) * (jx + 1)) / per_xcd);
  const int row_lo = a.g.row_base, row_hi = a.g.row_base + a.rows_stored - 1;
  const bool tail_lane = lane < 2 * k;  // columns 64 .. 64+2k-1
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

#pragma unroll 1
  for (uint32_t lb = lb_lo; lb < lb_hi; lb++) {
  const int r_now = r, bx_now = bx, cg_now = cg;
  if (++cg == a.tiles_y) {
    cg = 0;
    if (++bx == a.tiles_x) {
      bx = 0;
      ++r;
    }
  }
  const int chunk = cg_now * kBlockY + wave;
  const int yc = a.g.y0 + chunk * (kCombM * k) + r_now;  // first output row of the comb
  if (yc >= a.g.y1) continue;
  const int x0 = bx_now * kBlockX;
  int gx0 = x0 - k + lane, gx1 = x0 - k + 64 + lane;
  gx0 = gx0 < 0 ? 0 : (gx0 > W - 1 ? W - 1 : gx0);  // :136
  gx1 = gx1 < 0 ? 0 : (gx1 > W - 1 ? W - 1 : gx1);
  const uint32_t o16a = static_cast<uint32_t>(gx0) * 16u, o4a = static_cast<uint32_t>(gx0) * 4u;
  const uint32_t o16b = static_cast<uint32_t>(gx1) * 16u, o4b = static_cast<uint32_t>(gx1) * 4u;
  // the previous comb's ds_reads have returned (their values were consumed); make that explicit
  // before the DMA overwrites the cells
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
  for (int j = 0; j < rows; j++) {
    int gy = yc + (j - 1) * k;
    gy = gy < 0 ? 0 : (gy > H - 1 ? H - 1 : gy);              // :136
    gy = gy < row_lo ? row_lo : (gy > row_hi ? row_hi : gy);  // rows only masked outputs could reach
    const size_t grow = static_cast<size_t>(gy - a.g.row_base) * W;  // wave-uniform
    const float4* rin = a.in + grow;
    const uint32_t* rvis = a.vis + grow;
    const uint32_t cj = static_cast<uint32_t>(j * CWp);
    dma_b128(rin, o16a, lds_col + cj * 16u);
    dma_b32(rvis, o4a, lds_ids + cj * 4u);
    if (tail_lane) {
      dma_b128(rin, o16b, lds_col + (cj + 64u) * 16u);
      dma_b32(rvis, o4b, lds_ids + (cj + 64u) * 4u);
    }
  }
  // only this wave reads these cells: its own vmcnt orders the DMA before the ds_reads below
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  const int x = x0 + lane;
#pragma unroll
  for (int m = 0; m < kCombM; m++) {
    const int y = yc + m * k;
    if (x >= W || y >= a.g.y1) continue;
    const int cc = (m + 1) * CWp + lane + k;
    const float4 cp4 = col[cc];
    const f3 cp = xyz(cp4);
    const float dp = cp4.w;
    const uint32_t idp = ids[cc];
    const float* prow = pairw + idp * NP;
    const float wself = prow[idp];
    f3 num{0.f, 0.f, 0.f};
    float den = 0.f;
#pragma unroll
    for (int i = -1; i < 2; i++) {  // :132 (x offset outer: the reference's accumulation order)
#pragma unroll
      for (int jj = -1; jj < 2; jj++) {  // :133
        float w;
        f3 cq;
        if (i == 0 && jj == 0) {
          cq = cp;
          w = wself;  // centre tap: q == p, both exponentials are exactly 1
        } else {
          const int qi = cc + jj * CWp + i * k;
          const float4 cq4 = col[qi];
          cq = xyz(cq4);
          const float dq = cq4.w;
          const float wn = prow[ids[qi]];  // :62 via the id-pair table
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
      filtered = num * fast::rcp_(den);
    }
    const size_t ip = static_cast<size_t>(y - a.g.row_base) * W + x;
    if (!FINAL) {
      a.out[ip] = make_float4(filtered.x, filtered.y, filtered.z, dp);  // :152 (+ depth in alpha)
      continue;
    }
    // :213-239 reprojection — exact arithmetic: the truncated pixel coordinate is an integer observable
    int ppx = x, ppy = y;
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
    if (a.prev_pixel) a.prev_pixel[ip] = make_int2(ppx, ppy);
    f3 blend = filtered;  // :258
    if (a.frame > 0) {    // :251
      f3 hc{0.f, 0.f, 0.f};  // D2: out-of-image history fetch returns 0
      if (ppx >= 0 && ppx < W && ppy >= a.hist_y0 && ppy < a.hist_y1)
        hc = xyz(a.history[static_cast<size_t>(ppy - a.hist_row_base) * W + ppx]);
      const float oma = 1.0f - a.alpha;
      blend = f3{fmaf_(filtered.x, a.alpha, hc.x * oma), fmaf_(filtered.y, a.alpha, hc.y * oma),
                 fmaf_(filtered.z, a.alpha, hc.z * oma)};  // :254
    }
    a.out[ip] = make_float4(blend.x, blend.y, blend.z, 0.0f);  // :263 (D1: distinct buffer)
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

}  // namespace

void launch_stamp_depth(const FrameGeom& g, float4* color, const float* depth, hipStream_t s) {
  if (g.y1 <= g.y0) return;
  hipLaunchKernelGGL(k_stamp_depth, grid_for(g), dim3(kBlockX, kBlockY), 0, s, g, color, depth);
}

void launch_atrous(const AtrousArgs& a0, bool final_pass, hipStream_t s) {
  if (a0.g.y1 <= a0.g.y0) return;
  AtrousArgs a = a0;
  a.cz = -1.44269504088896341f / a.sigma_z;
  a.cl = -1.44269504088896341f / a.sigma_l;
  dim3 block(kBlockX, kBlockY);
  const int np = static_cast<int>(a.n_tris) + 1;
  if (!a.direct && a.pair_tab && np <= kPairMax && a.k >= 1 && a.k <= 16) {
    a.tiles_x = (a.g.W + kBlockX - 1) / kBlockX;
    const int nrows = a.g.y1 - a.g.y0;
    const int chunks = (nrows + kCombM * a.k - 1) / (kCombM * a.k);
    a.tiles_y = (chunks + kBlockY - 1) / kBlockY;         // chunk groups (4 consecutive chunks per block)
    static int n_cu = 0;
    if (!n_cu) {
      hipDeviceProp_t prop;
      int dev = 0;
      (void)hipGetDevice(&dev);
      n_cu = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
    }
    const uint32_t nlb = static_cast<uint32_t>(a.tiles_x) * static_cast<uint32_t>(a.tiles_y) * static_cast<uint32_t>(a.k);
    // persistent grid: as many blocks per CU as 160 KiB of LDS admits (5 at M = 2, k <= 8)
    const size_t lds_block = static_cast<size_t>((np * np * 4 + 15) & ~15) +
                             static_cast<size_t>(kBlockY) * (kCombM + 2) * (a.k <= 4 ? 72 : (a.k <= 8 ? 80 : 96)) * 20;
    uint32_t per_cu = static_cast<uint32_t>((160u * 1024u) / lds_block);
    if (per_cu > 8u) per_cu = 8u;
    if (per_cu < 1u) per_cu = 1u;
    uint32_t per_xcd = static_cast<uint32_t>((n_cu + 7) / 8) * per_cu;
    if (per_xcd > (nlb + 7) / 8) per_xcd = (nlb + 7) / 8;
    dim3 grid(per_xcd * 8u);
    // staged row stride: 72 cells for k <= 4, 80 for k <= 8, 96 for k <= 16
#define RTPT_LAUNCH_COMB(CW)                                                                         \
  do {                                                                                               \
    const size_t lds = static_cast<size_t>((np * np * 4 + 15) & ~15) +                               \
                       static_cast<size_t>(kBlockY) * (kCombM + 2) * (CW) * 20;                      \
    if (a.exact) {                                                                                   \
      if (final_pass)                                                                                \
        hipLaunchKernelGGL((k_atrous_comb<CW, true, true>), grid, block, lds, s, a);                 \
      else                                                                                           \
        hipLaunchKernelGGL((k_atrous_comb<CW, false, true>), grid, block, lds, s, a);                \
    } else {                                                                                         \
      if (final_pass)                                                                                \
        hipLaunchKernelGGL((k_atrous_comb<CW, true, false>), grid, block, lds, s, a);                \
      else                                                                                           \
        hipLaunchKernelGGL((k_atrous_comb<CW, false, false>), grid, block, lds, s, a);               \
    }                                                                                                \
  } while (0)
    if (a.k <= 4)
      RTPT_LAUNCH_COMB(72);
    else if (a.k <= 8)
      RTPT_LAUNCH_COMB(80);
    else
      RTPT_LAUNCH_COMB(96);
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
