// atrous_chain.hip — K3, several CONSECUTIVE iterations of applyTemporalFiltering's loop (main.cpp:1259-1305) in one
// launch: iteration k reads what iteration k-1 wrote, so run back to back every pass costs a 16 B/px store and a
// 20 B/px load that exist only to carry the intermediate image through HBM.  Here the intermediates live in LDS.
//
// Per-pixel arithmetic is the single-pass kernel's (temporalFiltering.comp.glsl:118-155: 3x3 taps at stride k,
// x offset outer / y offset inner, same weights, same accumulation order), and an intermediate is the same binary32
// value whether it rests in HBM or in LDS — so the chain is bit-identical to the separate passes, in the exact and in
// the fast weight arithmetic (tests compare the two bit for bit at full size).
//
// Shape.  A workgroup owns a column strip (bw output pixels wide) of a row segment and slides down it.  Level l
// (iteration k0 + l, stride s_l) produces G rows per step from a ring of rows of level l-1 kept in LDS; the levels run
// as a software pipeline — in one step every level works on the rows the level before it finished in the previous
// step — so there is ONE workgroup barrier per step and no vertical halo inside a segment.  Level l is computed
// E_l = sum_{j>l} s_j columns beyond the strip on either side (what the later levels' taps reach), the input is staged
// E_0 + s_0 beyond it: for the pair (3,4) that is 128 computed and 134 staged columns per 120 stored — 1.07x / 1.12x
// instead of the 9 (or 81) global taps per pixel of the separate passes.
//   rows of level l at step t:  y = in_start + t*G + g - lag_l,   lag_0 = s_0 + P*G,  lag_l = lag_{l-1} + s_l + G
//   ring of level l's input:    2*s_l + 2*G rows (what level l still reads + what its producer writes this step);
//                               the staged input ring has (P-1)*G more
// The input ring is filled by LDS-DMA issued at the top of a step for the rows consumed P steps later, so the loads
// are in flight under P steps of arithmetic (the wait at the barrier is a counted vmcnt: only the rows issued a step
// ago must have landed); only the level-0 waves issue DMA and only the last level's waves store to global memory, so
// that wait never includes a store.
// Frame borders follow the reference's clamp (:136): rows by clamping the tap row before the ring slot is formed
// (scalar), columns by letting a lane that stands for a column outside the frame compute the clamped column's value
// (one v_med3 per task), so every staged cell of every level holds exactly what the clamped fetch would return.
#include <cstdlib>

#include "device_common.hpp"
#include "lds_dma.hpp"

#include <type_traits>

#ifndef RTPT_TILE_TIMELINE
#define RTPT_TILE_TIMELINE 0  // timeline build (scripts/tile_timeline.py): when the workgroups of a launch lived, one workgroup's steps
#endif

namespace rt {
namespace {

#if RTPT_TILE_TIMELINE
#define RTPT_SPAN_SLOTS 4  // pair (1,2), pair (3,4) / any other first stride, the same two as the FINAL pass of a frame
#define RTPT_SPAN_READER rtpt_debug_chain_span
#define RTPT_SPAN_DEVICE
#include "experiments/span_instrumentation.inc"
#undef RTPT_SPAN_DEVICE
#endif

#ifndef RTPT_CHAIN_G
#define RTPT_CHAIN_G 3
#endif
constexpr int kChG = RTPT_CHAIN_G;  // rows per level per step = waves per level / 2 (the default; chain_g() picks per pair).
                                    // 4K pair launches, both pairs averaged: 1: 107.9 us, 2: 100.4, 3: 97.7, 4: 103.1 (more waves
                                    // per workgroup hide more latency until the rings cost a workgroup per CU).  Per pair
                                    // (profiles/r03_chain_g_ab.csv): (1,2): 2: 90.1, 3: 95.2, 4: 90.4; (3,4): 2: 107.2, 3: 97.5, 4: 111.0
#ifndef RTPT_CHAIN_P
#define RTPT_CHAIN_P 1
#endif
constexpr int kChP = RTPT_CHAIN_P;  // steps between staging an input row and its first use.  Measured at 4K (pair launches
                                    // averaged): 1: 100.7 us, 2: 103.6, 3: 104.1 — with two workgroups per CU the other one's
                                    // arithmetic already covers the DMA flight, and every extra step costs G ring rows of LDS
constexpr int kChCols = 128;        // columns a level computes at most: two waves per row
constexpr int kChMaxSegs = 127;     // row segments a launch can give individual heights (ChainSegs; more: equal heights)

// first row (relative to the launch's first row) of every row segment; start[n_segs] = the row count.  n = 0: equal segments
// of AtrousArgs::seg_rows rows.
struct ChainSegs {
  uint16_t n;
  uint16_t start[kChMaxSegs + 1];
};

__device__ __forceinline__ int posmod(int n, int r) {
  int m = n % r;
  return m < 0 ? m + r : m;
}

// K0: the first stride as a compile-time constant (0: read a.k).  With the strides known, a tap's LDS address is its row
// base plus an immediate: the generic form spends 40 of its 157 VALU per wave and step on tap addresses, and the kernel is
// bound by VALU issue.  The shipping pairs (1,2) and (3,4) are instantiated with K0 = 1 and 3.
template <int L, bool FINAL, bool EXACT, int G, int K0>
__global__ __launch_bounds__(128 * L * G) void k_atrous_chain(AtrousArgs a, ChainSegs sg) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  const int W = a.g.W, H = a.g.H;
  const int lane = static_cast<int>(threadIdx.x);
  const int wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.y));
  const int NP = static_cast<int>(a.n_tris) + 1;

  // ---- geometry of the chain (all wave-uniform)
  int s[L], E[L], R[L], lag[L];
  for (int l = 0; l < L; l++) s[l] = (K0 ? K0 : a.k) + l;  // main.cpp:1259-1260: waveletIteration = k, tap offset i*k (:135)
  E[L - 1] = 0;
  for (int l = L - 2; l >= 0; l--) E[l] = E[l + 1] + s[l + 1];
  for (int l = 0; l < L; l++) R[l] = 2 * s[l] + 2 * G;  // ring l = input of level l
  R[0] += (kChP - 1) * G;
  lag[0] = s[0] + kChP * G;
  for (int l = 1; l < L; l++) lag[l] = lag[l - 1] + s[l] + G;
  const int bw = a.strip_w;  // <= kChCols - 2 * E[0]
  const int in_stride = kChCols + 2 * s[0];  // staged input columns
  // LDS: [id-pair weights][ring 0: colour cells, ids][ring 1: colour, ids]...
  uint32_t ring_col[L], ring_ids[L];
  {
    uint32_t off = static_cast<uint32_t>((NP * NP * 4 + 15) & ~15);
    for (int l = 0; l < L; l++) {
      const uint32_t cells = static_cast<uint32_t>(R[l] * (l == 0 ? in_stride : kChCols));
      ring_col[l] = off;
      ring_ids[l] = off + 16u * cells;
      off += 20u * cells;
    }
  }
  float* pairw = reinterpret_cast<float*>(lds_raw);
  for (int i = wave * 64 + lane; i < NP * NP; i += 128 * L * G) pairw[i] = a.pair_tab[i];

  // ---- this workgroup's strip and row segment.  XCD-aware: physical blocks b, b+8, ... share an L2; each XCD gets a
  // contiguous run of the (segment-major, strip-minor) list so the column halos shared by neighbouring strips meet there.
  const uint32_t nb = static_cast<uint32_t>(a.n_strips) * static_cast<uint32_t>(a.n_segs);
  const uint32_t per_xcd = (nb + 7u) >> 3;
  const uint32_t lb = (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
  if (lb >= nb) return;  // block-uniform, before any barrier
#if RTPT_TILE_TIMELINE
  SpanScope span_((a.k == 1 ? 0u : 1u) + (FINAL ? 2u : 0u), lb == nb / 2);
#endif
  const int seg = static_cast<int>(lb / static_cast<uint32_t>(a.n_strips));
  const int strip = static_cast<int>(lb - static_cast<uint32_t>(seg) * static_cast<uint32_t>(a.n_strips));
  const int x0 = strip * bw;
  const int ya = a.g.y0 + (sg.n ? static_cast<int>(sg.start[seg]) : seg * a.seg_rows);
  const int yb_ = sg.n ? a.g.y0 + static_cast<int>(sg.start[seg + 1]) : ya + a.seg_rows;
  const int yb = yb_ < a.g.y1 ? yb_ : a.g.y1;
  if (ya >= yb) return;
  // rows every level must produce for this segment (frame clamp of the taps, :136), and the input rows to stage
  int lo[L], hi[L];
  lo[L - 1] = ya;
  hi[L - 1] = yb - 1;
  for (int l = L - 2; l >= 0; l--) {
    lo[l] = lo[l + 1] - s[l + 1] < 0 ? 0 : lo[l + 1] - s[l + 1];
    hi[l] = hi[l + 1] + s[l + 1] > H - 1 ? H - 1 : hi[l + 1] + s[l + 1];
  }
  const int ilo = lo[0] - s[0] < 0 ? 0 : lo[0] - s[0];
  const int ihi = hi[0] + s[0] > H - 1 ? H - 1 : hi[0] + s[0];
  int in_start = ya;
  for (int l = 0; l < L; l++) in_start -= s[l];
  const int T = (yb - 1 - in_start + lag[L - 1]) / G + 1;  // steps until the last output row has been produced

  // ---- this wave's role: level lw, row gw of the step, half hw of the row
  const int lw = wave / (2 * G), gw = (wave >> 1) % G, hw = wave & 1;
  int sl = s[0], El = E[0], Rs = R[0], Rd = 1, lagw = lag[0], low = lo[0], hiw = hi[0];
  uint32_t src_col = ring_col[0], src_ids = ring_ids[0], dst_col = 0, dst_ids = 0;
#pragma unroll
  for (int l = 1; l < L; l++)
    if (lw == l) {
      sl = s[l]; El = E[l]; Rs = R[l]; lagw = lag[l]; low = lo[l]; hiw = hi[l];
      src_col = ring_col[l]; src_ids = ring_ids[l];
    }
#pragma unroll
  for (int l = 0; l + 1 < L; l++)
    if (lw == l) {
      Rd = R[l + 1];
      dst_col = ring_col[l + 1];
      dst_ids = ring_ids[l + 1];
    }
  const bool last = lw == L - 1;
  const int src_stride = lw == 0 ? in_stride : kChCols;
  const int colv = hw * 64 + lane;                   // column within this level's extent
  const bool lane_on = colv < bw + 2 * El;
  const int xv = x0 - El + colv;                     // frame column this lane stands for
  const int xc = xv < 0 ? 0 : (xv > W - 1 ? W - 1 : xv);  // :136
  const int csrc = xc - x0 + El + sl;                // its column in the source ring (origin x0 - El - sl)
  const float4* scol = reinterpret_cast<const float4*>(lds_raw + src_col);
  const uint32_t* sids = reinterpret_cast<const uint32_t*>(lds_raw + src_ids);
  float4* dcol = reinterpret_cast<float4*>(lds_raw + dst_col);
  uint32_t* dids = reinterpret_cast<uint32_t*>(lds_raw + dst_ids);
  int slot_s = posmod(gw - lagw, Rs), slot_d = posmod(gw - lagw, Rd);

  // ---- staging duty (level-0 waves): wave w < G stages columns [0,64) and the tail [128, in_stride) of row w of the
  // step, wave G <= w < 2G columns [64,128) of row w - G
  const bool stager = wave < 2 * G;
  const int srow = wave < G ? wave : wave - G;
  const int schunk = wave < G ? 0 : 1;
  uint32_t o16 = 0, o16t = 0;
  {
    const int origin = x0 - E[0] - s[0];
    int gx = origin + schunk * 64 + lane;
    gx = gx < 0 ? 0 : (gx > W - 1 ? W - 1 : gx);  // :136
    o16 = static_cast<uint32_t>(gx) * 16u;
    int gt = origin + 128 + lane;
    gt = gt < 0 ? 0 : (gt > W - 1 ? W - 1 : gt);
    o16t = static_cast<uint32_t>(gt) * 16u;
  }
  const bool tail_lane = lane < 2 * s[0];
  const uint32_t lds0 = static_cast<uint32_t>(reinterpret_cast<size_t>((__attribute__((address_space(3))) unsigned char*)lds_raw));
  int slot_in = srow % R[0];  // ring slot of input row in_start + t*G + srow

  __syncthreads();  // pair table visible
#if RTPT_TILE_TIMELINE
  span_.mark(1);
#endif

  const float h9 = 1.0f / 9.0f;  // :145
#pragma unroll 1
  for (int t = 0; t < T; t++) {
    // 1. stage the input rows of this step (consumed kChP steps from now)
    bool issued = false;
    if (stager) {
      const int iy = in_start + t * G + srow;
      if (iy >= ilo && iy <= ihi) {
        const size_t grow = static_cast<size_t>(iy - a.g.row_base) * W;
        const float4* rin = a.in + grow;
        const uint32_t* rvis = a.vis + grow;
        const uint32_t cell = static_cast<uint32_t>(slot_in * in_stride + schunk * 64);
        __builtin_amdgcn_s_setprio(3);
        dma_b128(rin, o16, lds0 + ring_col[0] + cell * 16u);
        dma_b32(rvis, o16 >> 2, lds0 + ring_ids[0] + cell * 4u);
        if (schunk == 0 && tail_lane) {
          const uint32_t cellt = static_cast<uint32_t>(slot_in * in_stride + 128);
          dma_b128(rin, o16t, lds0 + ring_col[0] + cellt * 16u);
          dma_b32(rvis, o16t >> 2, lds0 + ring_ids[0] + cellt * 4u);
        }
        __builtin_amdgcn_s_setprio(0);
        issued = true;
      }
      slot_in += G;
      if (slot_in >= R[0]) slot_in -= R[0];
    }
    // 2. this wave's row of its level
    const int y = in_start + t * G + gw - lagw;
    if (y >= low && y <= hiw && lane_on) {
      const int ym = y - sl < 0 ? 0 : y - sl, yp = y + sl > H - 1 ? H - 1 : y + sl;  // :136
      int sm = slot_s + (ym - y), sp = slot_s + (yp - y);
      if (sm < 0) sm += Rs;
      if (sp >= Rs) sp -= Rs;
      // tap (i, jj) of this row: cell rowb[jj + 1] + (i + 1) * stride (rowb already stands one stride left of the pixel)
      const int rowb[3] = {sm * src_stride + csrc - sl, slot_s * src_stride + csrc - sl, sp * src_stride + csrc - sl};
      const int cc = rowb[1] + sl;
      const float4 cp4 = scol[cc];
      const f3 cp = xyz(cp4);
      const float dp = cp4.w;
      const uint32_t idp = sids[cc];
      const float* prow = pairw + idp * NP;
      f3 num{0.f, 0.f, 0.f};
      float den = 0.f;
      auto taps = [&](auto stride_tag) {
        constexpr int SC = decltype(stride_tag)::value;  // this level's stride if the kernel knows it, else 0
        const int st = SC ? SC : sl;
#pragma unroll
        for (int i = -1; i < 2; i++) {  // :132 (x offset outer: the reference's accumulation order)
#pragma unroll
          for (int jj = -1; jj < 2; jj++) {  // :133
            float w;
            f3 cq;
            if (i == 0 && jj == 0) {
              cq = cp;
              w = prow[idp];  // centre tap: q == p, both exponentials are exactly 1
            } else {
              const int qi = rowb[jj + 1] + (i + 1) * st;
              const float4 cq4 = scol[qi];
              cq = xyz(cq4);
              const float dq = cq4.w;
              const float wn = prow[sids[qi]];  // :62 via the id-pair table
              const f3 dc = cp - cq;
              if (EXACT) {
                const float wd = exact::exp_(-__builtin_fabsf(dp - dq) / a.sigma_z);  // :67-68
                const float wl = exact::exp_(-exact::length(dc) / a.sigma_l);         // :73
                w = (wn * wd) * wl;                                                   // :77
              } else {
                const float e = fmaf_(__builtin_fabsf(dp - dq), a.cz, fast::sqrt_(exact::dot(dc, dc)) * a.cl);
                w = wn * __builtin_amdgcn_exp2f(e);
              }
            }
            if (EXACT) {
              const float hw_ = h9 * w;
              num = f3{fmaf_(hw_, cq.x, num.x), fmaf_(hw_, cq.y, num.y), fmaf_(hw_, cq.z, num.z)};  // :146
              den = den + hw_;                                                                       // :147
            } else {
              num = f3{fmaf_(w, cq.x, num.x), fmaf_(w, cq.y, num.y), fmaf_(w, cq.z, num.z)};
              den = den + w;
            }
          }
        }
        // the two (three) instantiations differ only in constants; left alone, the optimiser sinks them back into ONE body
        // with the stride in a register.  A distinct statement at the end of each keeps them apart.
        if (SC) asm volatile("; taps of the level with stride %0" ::"n"(SC));
      };
      if (K0 == 0)
        taps(std::integral_constant<int, 0>{});
      else if (lw == 0)
        taps(std::integral_constant<int, K0>{});
      else if (L < 3 || lw == 1)
        taps(std::integral_constant<int, K0 ? K0 + 1 : 0>{});
      else
        taps(std::integral_constant<int, K0 ? K0 + 2 : 0>{});
      f3 filtered;
      if (EXACT)
        filtered = f3{num.x / den, num.y / den, num.z / den};  // :150
      else
        filtered = num * fast::rcp_(den);
      if (!last) {
        // :152 into the next level's ring instead of filteredImageBuffer / image
        const int di = slot_d * kChCols + colv;
        dcol[di] = make_float4(filtered.x, filtered.y, filtered.z, dp);
        dids[di] = idp;
      } else if (colv < bw && xv < W) {
        const size_t ip = static_cast<size_t>(y - a.g.row_base) * W + xv;
        if (!FINAL) {
          typedef float v4f_ __attribute__((ext_vector_type(4)));
          v4f_ o4 = {filtered.x, filtered.y, filtered.z, a.alpha_zero ? 0.0f : dp};
          __builtin_nontemporal_store(o4, reinterpret_cast<v4f_*>(a.out + ip));  // :152; nothing in this launch re-reads it
        } else {
          // :213-263 reprojection + temporal blend, exact arithmetic (the truncated pixel is an integer observable)
          int ppx, ppy;
          reproject_pixel(W, H, a.PVprev, idp, xyz(a.worldpos[ip]), a.lut_prev, xv, y, ppx, ppy);
          if (a.prev_pixel) a.prev_pixel[ip] = make_int2(ppx, ppy);
          f3 blend = filtered;  // :258
          if (a.frame > 0) {    // :251
            f3 hc{0.f, 0.f, 0.f};  // D2
            if (ppx >= 0 && ppx < W && ppy >= a.hist_y0 && ppy < a.hist_y1)
              hc = xyz(a.history[static_cast<size_t>(ppy - a.hist_row_base) * W + ppx]);
            const float oma = 1.0f - a.alpha;
            blend = f3{fmaf_(filtered.x, a.alpha, hc.x * oma), fmaf_(filtered.y, a.alpha, hc.y * oma),
                       fmaf_(filtered.z, a.alpha, hc.z * oma)};  // :254
          }
          a.out[ip] = make_float4(blend.x, blend.y, blend.z, 0.0f);  // :263 (D1: distinct buffer)
        }
      }
    }
    slot_s += G;
    if (slot_s >= Rs) slot_s -= Rs;
    slot_d += G;
    if (slot_d >= Rd) slot_d -= Rd;
    // 3. staged rows landed, ring writes done; the barrier publishes both to the next step
    if (stager) {
      // vector-memory operations return in issue order: allowing this step's DMAs (4 per row for the wave that also
      // stages the tail columns, 2 otherwise) to stay in flight waits exactly for everything older
      if (kChP < 2 || !issued)
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      else if (schunk == 0)
        asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
      else
        asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");
    } else
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#if RTPT_TILE_TIMELINE
    span_.step();
#endif
  }
}


#ifndef RTPT_AB_VARIANTS
#define RTPT_AB_VARIANTS 0
#endif
#if RTPT_AB_VARIANTS
#include "experiments/chain_sliding_window.inc"
#endif

}  // namespace

#if RTPT_TILE_TIMELINE
#define RTPT_SPAN_HOST
#include "experiments/span_instrumentation.inc"
#undef RTPT_SPAN_HOST
#endif

// rows per level per step of a launch (G; a workgroup is 4 G waves of a pair).  A row segment pays 2 (s0 + s1) + 2 G rows of
// pipeline fill and re-staging whatever its length, so what decides is how long the segments of a launch are:
//   * long segments (a 4K frame: 135 rows): three rows per step; the short-stride pair (1,2) two, its rings then admit four
//     workgroups per CU (4K: 90.4 us against 95.6);
//   * short segments (an 8-rank strip of the 4K frame: 19 rows at G = 3; 1080p: 34): FOUR rows per step — 16 waves per
//     workgroup, still two workgroups per CU, a quarter fewer steps for the same fill.  Round 4, profiles/r04_chain_g_ab.txt:
//     the 300-row strip 25.0 -> 21.9 us per pair, 1080p 32.9 -> 31.9, where the 4K frame loses (103 against 97.7).
// RTPT_CHAIN_G1 (read by rtpt_create, FilterPolicy::chain_g_pin) pins 2, 3 or 4 for A/B runs and tests.
static int chain_g(int k0, int levels, int n_strips, int rows, int n_cu, int pin) {
  if (levels == 3) return 2;  // two waves per row and level: three levels fit a workgroup of 1 024 threads with two rows per step only
  if (levels != 2 || kChG != 3) return kChG;
  if (pin >= 12 && pin <= 14) {  // 12 / 13 / 14: pin the pair (1,2) alone to 2 / 3 / 4 (A/B)
    if (k0 == 1) return pin - 10;
    pin = 0;
  }
  if (pin == 4 || ((pin == 2 || pin == 3) && k0 == 1)) return pin;
  const int strips = n_strips > 0 ? n_strips : 1;
  const int segs3 = (n_cu * 2) / strips;  // G = 3: two workgroups of 12 waves per CU
  if (!pin && (segs3 < 1 || rows / segs3 < 48)) return 4;
  if (k0 != 1) return kChG;
  const int n_segs = (n_cu * 4) / strips;
  return n_segs >= 1 && rows / n_segs >= 40 ? 2 : 3;
}

// bytes of dynamic LDS of a chain of `levels` iterations starting at stride k0 (host mirror of the kernel's layout)
static size_t chain_lds(int k0, int levels, uint32_t n_tris, int G) {
  const int np = static_cast<int>(n_tris) + 1;
  size_t off = static_cast<size_t>((np * np * 4 + 15) & ~15);
  for (int l = 0; l < levels; l++) {
    const int s = k0 + l;
    const size_t cells = static_cast<size_t>(2 * s + 2 * G + (l == 0 ? (kChP - 1) * G : 0)) * (l == 0 ? kChCols + 2 * k0 : kChCols);
    off += 20 * cells;
  }
  return off;
}

#if RTPT_AB_VARIANTS
// rows per step of the sliding-window kernel: swept per pair (RTPT_CHAIN_SW_G1 / _G3, FilterPolicy, override for A/B)
static int chain_sw_g(int pin) {
  const int g = pin ? pin : kSwG;
  return g == 2 || g == 4 || g == 6 ? g : 3;
}
static size_t atrous_chain_sw_lds(int k0, uint32_t n_tris, int g) {
  const int np = static_cast<int>(n_tris) + 1;
  const size_t tab = static_cast<size_t>((np * np * 4 + 15) & ~15);
  return tab + 20u * static_cast<size_t>(2 * g) * (static_cast<size_t>(kChCols + 2 * k0) + kChCols);
}
// the sliding-window kernel serves pairs whose 2 (s0 + s1) waves fit a workgroup
static bool chain_sw_supported(int k0) { return 64 * 2 * (2 * k0 + 1) <= 896; }
// FilterPolicy::chain_sw (RTPT_CHAIN_SW=1 at rtpt_create) selects the sliding-window kernel for A/B runs.  It is NOT the
// default: it issues 44 % fewer LDS instructions and 12 % fewer VALU instructions per launch than k_atrous_chain and is slower
// (4K pairs 113-116 us against 95-97; profiles/r03_chain_sw_ab.csv, r03_chain_pmc_*.json): the launch is bound by VALU issue
// inside barrier-phased steps, not by the LDS array, and a residue class per wave leaves the waves of a step unevenly loaded.
// marks a library built with the A/B variants (tests skip the variant checks without it)
extern "C" __attribute__((visibility("default"))) int rtpt_debug_ab_variants(void) { return 1; }
#endif

int atrous_chain_strip_width(int k0, int levels) {
  int e0 = 0;
  for (int l = 1; l < levels; l++) e0 += k0 + l;
  return kChCols - 2 * e0;
}

template <int L, int G, int K0>
static hipError_t chain_attrs() {
  constexpr int kMax = 160 * 1024;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_atrous_chain<L, false, false, G, K0>), hipFuncAttributeMaxDynamicSharedMemorySize, kMax);
  if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_atrous_chain<L, false, true, G, K0>), hipFuncAttributeMaxDynamicSharedMemorySize, kMax);
  if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_atrous_chain<L, true, false, G, K0>), hipFuncAttributeMaxDynamicSharedMemorySize, kMax);
  if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_atrous_chain<L, true, true, G, K0>), hipFuncAttributeMaxDynamicSharedMemorySize, kMax);
  return e;
}
hipError_t prepare_device_atrous_chain() {
  hipError_t e = chain_attrs<2, kChG, 0>();
  if (e == hipSuccess) e = (chain_attrs<2, kChG, 3>());  // the pair (3,4)
  if (e == hipSuccess) e = (chain_attrs<2, kChG, 1>());
  if (e == hipSuccess && kChG == 3) e = (chain_attrs<2, 2, 1>());  // the pair (1,2) on tall frames
  if (e == hipSuccess && kChG == 3) e = (chain_attrs<2, 2, 0>());
  if (e == hipSuccess && kChG == 3) e = (chain_attrs<2, 4, 0>());  // short row segments (strips, small frames): chain_g()
  if (e == hipSuccess && kChG == 3) e = (chain_attrs<2, 4, 1>());
  if (e == hipSuccess && kChG == 3) e = (chain_attrs<2, 4, 3>());
#if RTPT_AB_VARIANTS
  constexpr int kMaxLds = 160 * 1024;
#define RTPT_SW_ATTR(GG)                                                                                                              \
  if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_atrous_chain_sw<false, GG>), hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLds); \
  if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_atrous_chain_sw<true, GG>), hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLds);
  RTPT_SW_ATTR(2) RTPT_SW_ATTR(3) RTPT_SW_ATTR(4) RTPT_SW_ATTR(6)
#undef RTPT_SW_ATTR
#endif
  if (e == hipSuccess) e = (chain_attrs<3, 2, 0>());
  if (e == hipSuccess) e = (chain_attrs<3, 2, 1>());                 // (1,2,3)
  if (e == hipSuccess) e = (chain_attrs<2, kChG, 4>());              // (4,5), the pair that may end a frame of N = 5 in its FINAL pass
  if (e == hipSuccess && kChG == 3) e = (chain_attrs<2, 4, 4>());
  return e;
}

// true when `levels` consecutive iterations from stride k0 can run as one chain on this scene
bool atrous_chain_supported(int k0, int levels, uint32_t n_tris) {
  if (levels < 2 || levels > 3 || k0 < 1) return false;
  const int g_max = levels == 3 ? 2 : (levels == 2 && kChG == 3 ? 4 : kChG);  // the most rows per step chain_g() picks
  if (128 * levels * (levels == 3 ? 2 : kChG) > 1024) return false;  // two waves per row and level: the workgroup must fit 1024 threads
  if (n_tris + 1 > 64) return false;  // id-pair table in LDS (the per-pixel-normal variant is not chained)
  if (atrous_chain_strip_width(k0, levels) < 64) return false;
  return chain_lds(k0, levels, n_tris, g_max) <= 160 * 1024;
}

// Heights of the row segments of a launch.  The workgroups of a chained launch all start within ~1.5 us and a CU hosts
// per_cu of them, but they do not share it evenly: the issue arbiter serves the oldest wave first, so of the four
// workgroups of a (1,2) launch on a CU the one dispatched first ends after 48 us, the others after 63, 79 and 92
// (profiles/r04_frame_timeline_4k.txt, equal segments of 66 rows) — and what is left to the last one alone (8 waves on a
// CU) runs at 1.0 us per step where four resident workgroups take 0.5 us per workgroup-step between them.  Equal work ends
// unequally; unequal work can end together.  The dispatch order is known: physical block b runs on XCD b & 7 as that XCD's
// (b >> 3)-th workgroup, the first n_cu / 8 of an XCD are the oldest on their CUs, the next n_cu / 8 the second oldest, ...;
// with the (segment-major, strip-minor) list cut into one contiguous run per XCD that age is a property of the row
// SEGMENT (up to the few workgroups where a segment straddles two ages).  A segment whose workgroups have mean age r (0 the
// oldest) gets the weight 1 + skew (1 - 2 r / (per_cu - 1)) of the rows.
// The count of segments is rounded down to a multiple of 8 so that every XCD's run holds whole segments (a segment split between
// two XCDs is old on one and young on the other: 33 segments at 4K made the (1,2) launch 20 % slower instead of faster).
static void chain_segments(ChainSegs& sg, AtrousArgs& a, int rows, int n_cu, int per_cu, int skew_pct, int fill_rows) {
  sg.n = 0;
  // built-in (profiles/r04_chain_segment_skew_ab.json): 35 % with two or three workgroups per CU — the (3,4) launch of a 4K frame
  // 109.8 -> 103.7 us with every workgroup ending within 0.3 us of the others, 4K frame 0.737 -> 0.730 ms, a 270-row strip
  // 0.1281 -> 0.1267; none with four (the (1,2) launch at 4K: 20-80 % all cost 1-5 us — eight segments fewer for the alignment
  // leave 32 CUs with three workgroups, and its oldest workgroups still end 18 us before its youngest at 45 %)
  if (skew_pct < 0) skew_pct = per_cu >= 4 ? 0 : 35;
  if (per_cu < 2 || skew_pct == 0 || a.n_segs < 8 || rows > 65535) return;
  const int n_segs = a.n_segs & ~7;
  if (n_segs > kChMaxSegs) return;
  const uint32_t nb = static_cast<uint32_t>(a.n_strips) * static_cast<uint32_t>(n_segs);
  const uint32_t per_xcd = nb >> 3, cu_xcd = static_cast<uint32_t>((n_cu + 7) / 8);
  if (per_xcd <= cu_xcd) return;  // a single workgroup per CU: nothing to share
  const double skew = skew_pct / 100.0;
  double w[kChMaxSegs], total = 0.0;
  for (int sgi = 0; sgi < n_segs; sgi++) {
    double age = 0.0;
    for (int st = 0; st < a.n_strips; st++) {
      const uint32_t lb = static_cast<uint32_t>(sgi) * static_cast<uint32_t>(a.n_strips) + static_cast<uint32_t>(st);
      const uint32_t r = (lb % per_xcd) / cu_xcd;
      age += r < static_cast<uint32_t>(per_cu - 1) ? r : static_cast<uint32_t>(per_cu - 1);
    }
    age /= a.n_strips;
    w[sgi] = 1.0 + skew * (1.0 - 2.0 * age / (per_cu - 1));
    if (w[sgi] < 0.2) w[sgi] = 0.2;
    total += w[sgi];
  }
  // cumulative rounding; a segment shorter than the pipeline's fill would cost more than it balances: then equal heights
  double acc = 0.0;
  sg.start[0] = 0;
  for (int sgi = 0; sgi < n_segs; sgi++) {
    acc += w[sgi];
    const int end = sgi + 1 == n_segs ? rows : static_cast<int>(acc / total * rows + 0.5);
    if (end - static_cast<int>(sg.start[sgi]) < fill_rows / 2 + 1) return;
    sg.start[sgi + 1] = static_cast<uint16_t>(end);
  }
  sg.n = static_cast<uint16_t>(n_segs);
  a.n_segs = n_segs;
}

void launch_atrous_chain(const AtrousArgs& a0, int levels, bool final_pass, const FilterPolicy& pol, hipStream_t s) {
  if (a0.g.y1 <= a0.g.y0) return;
  AtrousArgs a = a0;
  a.cz = -1.44269504088896341f / a.sigma_z;
  a.cl = -1.44269504088896341f / a.sigma_l;
#if RTPT_AB_VARIANTS
  const bool sw = levels == 2 && !final_pass && pol.chain_sw != 0 && chain_sw_supported(a.k);
#else
  const bool sw = false;
#endif
  int bw = atrous_chain_strip_width(a.k, levels);
  if (!sw && a.k == 1 && levels == 2 && pol.chain_bw >= 64 && pol.chain_bw < bw) bw = pol.chain_bw;  // A/B
  a.strip_w = bw;
  a.n_strips = (a.g.W + bw - 1) / bw;
  const int g = chain_g(a.k, levels, a.n_strips, a.g.y1 - a.g.y0, a.n_cu > 0 ? a.n_cu : 256, pol.chain_g_pin);
#if RTPT_AB_VARIANTS
  const int sw_g = chain_sw_g(a.k == 1 ? pol.chain_sw_g1 : pol.chain_sw_g3);
  const size_t lds = sw ? atrous_chain_sw_lds(a.k, a.n_tris, sw_g) : chain_lds(a.k, levels, a.n_tris, g);
#else
  const size_t lds = chain_lds(a.k, levels, a.n_tris, g);
#endif
  // one resident generation of workgroups: as many per CU as the LDS admits, row segments sized to fill them
  const int n_cu = a.n_cu > 0 ? a.n_cu : 256;
  const int waves = sw ? 2 * (2 * a.k + 1) : 2 * levels * g;
  int per_cu = static_cast<int>((160u * 1024u) / lds);
  if (per_cu > 32 / waves) per_cu = 32 / waves;
  if (per_cu < 1) per_cu = 1;
  const int rows = a.g.y1 - a.g.y0;
  if (pol.chain_wg_per_cu >= 1 && pol.chain_wg_per_cu < per_cu) per_cu = pol.chain_wg_per_cu;  // A/B: workgroups per CU the segments are sized for
  int n_segs = (n_cu * per_cu) / a.n_strips;
  if (n_segs < 1) n_segs = 1;
  int seg_rows = (rows + n_segs - 1) / n_segs;
  // a segment re-stages 2*sum(s) rows and pays the pipeline fill once: keep it at least this long.  (16 k rows were the
  // first choice; at 1080p that left the pair (3,4) 23 segments x 16 strips = 368 workgroups for 512 slots: 38.0 us per
  // pair, 34.3 with 8 k rows and 480 workgroups; 4K and a 300-row strip do not change)
  const int min_rows = 8 * a.k;
  if (seg_rows < min_rows) seg_rows = min_rows;
  // (not rounded up to whole steps of g rows: a partly used last step costs less than the workgroup slots the longer
  // segments leave empty — 1080p pair (1,2): 30 segments of 36 rows 34.5 us, 32 of 34 rows 33.4)
  a.seg_rows = seg_rows;
  a.n_segs = (rows + seg_rows - 1) / seg_rows;
  ChainSegs sg;
  chain_segments(sg, a, rows, n_cu, per_cu, sw ? 0 : (per_cu >= 4 ? pol.chain_skew : pol.chain_skew2), 2 * (levels * a.k + levels * (levels - 1) / 2) + 2 * g);
  const uint32_t nb = static_cast<uint32_t>(a.n_strips) * static_cast<uint32_t>(a.n_segs);
  const dim3 grid(((nb + 7u) / 8u) * 8u), block(64, waves);
#if RTPT_AB_VARIANTS
  if (sw) {
#define RTPT_SW_LAUNCH(GG)                                                              \
  case GG:                                                                              \
    if (a.exact)                                                                        \
      hipLaunchKernelGGL((k_atrous_chain_sw<true, GG>), grid, block, lds, s, a);        \
    else                                                                                \
      hipLaunchKernelGGL((k_atrous_chain_sw<false, GG>), grid, block, lds, s, a);       \
    break;
    switch (sw_g) {
      RTPT_SW_LAUNCH(2) RTPT_SW_LAUNCH(4) RTPT_SW_LAUNCH(6)
      default: RTPT_SW_LAUNCH(3)
    }
#undef RTPT_SW_LAUNCH
    return;
  }
#endif
#define RTPT_LAUNCH_CHAIN(LV, GG, KK)                                                              \
  do {                                                                                             \
    if (a.exact) {                                                                                 \
      if (final_pass)                                                                              \
        hipLaunchKernelGGL((k_atrous_chain<LV, true, true, GG, KK>), grid, block, lds, s, a, sg);      \
      else                                                                                         \
        hipLaunchKernelGGL((k_atrous_chain<LV, false, true, GG, KK>), grid, block, lds, s, a, sg);     \
    } else {                                                                                       \
      if (final_pass)                                                                              \
        hipLaunchKernelGGL((k_atrous_chain<LV, true, false, GG, KK>), grid, block, lds, s, a, sg);     \
      else                                                                                         \
        hipLaunchKernelGGL((k_atrous_chain<LV, false, false, GG, KK>), grid, block, lds, s, a, sg);    \
    }                                                                                              \
  } while (0)
  // strides as compile-time constants for the pairs a default frame runs (N = 5: (1,2) and (3,4)); RTPT_CHAIN_GENERIC=1
  // runs the generic instantiation for A/B and for the test that the two agree
  const bool generic = pol.chain_generic != 0;
  if (levels == 2 && g == 4 && kChG == 3) {
    if (a.k == 1 && !generic)
      RTPT_LAUNCH_CHAIN(2, 4, 1);
    else if (a.k == 3 && !generic)
      RTPT_LAUNCH_CHAIN(2, 4, 3);
    else if (a.k == 4 && !generic)
      RTPT_LAUNCH_CHAIN(2, 4, 4);
    else
      RTPT_LAUNCH_CHAIN(2, 4, 0);
  } else if (levels == 2 && a.k == 4 && !generic && g == kChG) {
    RTPT_LAUNCH_CHAIN(2, kChG, 4);
  } else if (levels == 2 && a.k == 1 && !generic) {
    if (g == 2 && kChG == 3)
      RTPT_LAUNCH_CHAIN(2, 2, 1);
    else
      RTPT_LAUNCH_CHAIN(2, kChG, 1);
  } else if (levels == 2 && a.k == 3 && !generic)
    RTPT_LAUNCH_CHAIN(2, kChG, 3);
  else if (levels == 2 && g == 2 && kChG == 3)
    RTPT_LAUNCH_CHAIN(2, 2, 0);
  else if (levels == 2)
    RTPT_LAUNCH_CHAIN(2, kChG, 0);
  else if (a.k == 1 && !generic)
    RTPT_LAUNCH_CHAIN(3, 2, 1);
  else
    RTPT_LAUNCH_CHAIN(3, 2, 0);
#undef RTPT_LAUNCH_CHAIN
}

}  // namespace rt
