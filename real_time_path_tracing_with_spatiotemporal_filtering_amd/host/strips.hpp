// strips.hpp — row-strip sharding of one frame across ranks for the C++ host.  New work: the
// reference is single-device (its only multi-device hook, useDeviceGroups, is dead: context.hpp:153,
// context.cpp:406-412).  Same plan as the Python mirror (strips.py), which the CPU tests exercise with gloo ranks.
//
// Rank r of R owns frame rows [r*H/R, (r+1)*H/R).  K0/K1/K2 are per-pixel independent and the RNG seed depends only on
// absolute (x, y, frame, batch) (raytrace.comp.glsl:297), so any partition reproduces the single-GPU image bit for
// bit.  K3 iteration k reads rows y-k, y, y+k (temporalFiltering.comp.glsl:135) with a GLOBAL border clamp (:136):
//   exchange   k rows per neighbour per iteration travel (the one real exchange step of the path)
//   redundant  nothing travels: every rank traces and filters the sum_{j>k} j extra rows it will need
// The final pass fetches history at the REPROJECTED pixel (:253), which under camera motion lies in other strips:
// reprojection_rows() bounds the rows a strip can reach from the camera matrices and the scene bounds, and the ranks
// swap exactly those bands of the previous frame.
#pragma once

#include <cstdint>
#include <string>
#include <utility>
#include <vector>

#include "../../include/rtpt.h"

namespace rtpt_host {

using Rows = std::pair<int, int>;  // [first, second)

struct StripPlan {
  int height = 0, world = 1, rank = 0, iterations = 0;
  bool exchange = false;  // halo mode: exchange | redundant
  uint32_t ext_flags = 0; // RTPT_FLAG_EXT_*: the tap-shape modes change how far iteration k reaches
  std::vector<int> splits;  // empty = equal strips; else world + 1 ascending rows 0 .. height, rank r owns [splits[r], splits[r + 1])
                            // (strips.py StripPlan.splits; rtpt_app --splits; balanced_splits below moves them)

  static Rows bounds(int height, int world, int rank, const std::vector<int>& splits = {}) {
    if (!splits.empty()) return {splits[static_cast<size_t>(rank)], splits[static_cast<size_t>(rank) + 1]};
    return {static_cast<int>(static_cast<int64_t>(rank) * height / world), static_cast<int>(static_cast<int64_t>(rank + 1) * height / world)};
  }
  Rows rows_of(int r) const { return bounds(height, world, r, splits); }
  Rows own() const { return rows_of(rank); }
  // throws std::runtime_error unless splits is empty or world + 1 ascending rows from 0 to height
  void validate() const;
  // rows above/below a pixel that iteration k reads: k for the reference's 3x3 linear-stride taps
  // (temporalFiltering.comp.glsl:135); radius 2 with EXT_GAUSS5, stride 2^(k-1) with EXT_POW2_STRIDE (strips.py: reach)
  int reach(int k) const {
    const int stride = (ext_flags & RTPT_FLAG_EXT_POW2_STRIDE) ? (1 << (k - 1)) : k;
    return stride * ((ext_flags & RTPT_FLAG_EXT_GAUSS5) ? 2 : 1);
  }
  // RTPT_FLAG_EXT_SVGF_VARIANCE (with _EXT_VARIANCE): the 7x7 spatial variance estimate needs 3 more traced rows (strips.py)
  int svgf_pad() const { return (ext_flags & 0x900u) == 0x900u ? 3 : 0; }
  int halo() const;                     // rows stored beyond the owned strip on each side
  Rows stored() const;
  Rows grow(int rows) const;
  Rows gbuffer_rows() const { return stored(); }
  Rows gradient_rows() const { return own(); }
  Rows raytrace_rows() const;
  Rows filter_rows(int k) const;
  struct Exchange {
    int peer;
    Rows send, recv;
  };
  // what travels before iteration k in exchange mode; throws std::runtime_error when a strip is shorter than k rows
  std::vector<Exchange> exchange_rows(int k) const;
};

// previous-frame rows the final pass of frame rows `rows` can fetch history from; see strips.py:reprojection_rows for
// the argument (extremes of a function monotone along x, y and view depth are at the 8 corners of the box)
Rows reprojection_rows(const rtpt_ubo& ubo, int width, int height, Rows rows, const double bounds_min[3], const double bounds_max[3],
                       float z_near, int pad = 2);

struct HistoryOp {
  int peer;
  bool send;  // false: receive
  Rows rows;
};
// per rank: what it sends of its own strip / receives of the peers' strips so that it holds needs[rank]
std::vector<std::vector<HistoryOp>> history_exchange_plan(int height, int world, const std::vector<Rows>& needs, const std::vector<int>& splits = {});

// new strip boundaries from the ranks' measured frame times (strips.py balanced_splits: the same operations in the same
// order, so both hosts arrive at the same rows): cost[r] = what rank r spent on its rows, spread evenly over them; the
// boundaries cut the cumulative cost into equal parts; every strip keeps at least min_rows rows.  Apply again to the
// times measured with the new boundaries — equal times are the fixed point.
std::vector<int> balanced_splits(const std::vector<int>& splits, const std::vector<double>& cost, int min_rows = 1);

// ---- transports: how rows move between ranks -----------------------------------------------------------------------
// One message = `bytes` bytes from a device pointer of rank `src` to a device pointer of rank `dst`.  All messages of
// one exchange step are posted between begin() and end(); they are ordered on the stream passed to begin().
class Transport {
 public:
  virtual ~Transport() = default;
  virtual void begin(void* hip_stream) = 0;
  virtual void send(int src_rank, const void* src, int dst_rank, size_t bytes) = 0;  // called by the process owning src_rank
  virtual void recv(int dst_rank, void* dst, int src_rank, size_t bytes) = 0;        // called by the process owning dst_rank
  virtual void end() = 0;
  virtual uint64_t bytes_sent() const = 0;
};

// every rank lives in this process (strip contexts on one GPU, one shared stream): a message is a device-to-device copy
Transport* make_local_transport();
// one rank per process, one GPU per rank: ncclSend / ncclRecv in a group on the context's stream (RCCL over xGMI).
// Rendezvous without MPI: rank 0 removes whatever `id_file` holds, then publishes {magic, nonce, ncclUniqueId} there
// (written to a temporary name and renamed); the others poll until the file carries THEIR nonce — a file left behind by
// an earlier launch (another nonce) is ignored, not trusted.  `nonce` must be the same on every rank of one launch and
// should differ between launches (rtpt_app --rccl-nonce; default: the launcher's pid).  ncclCommInitRank runs under a
// watchdog: if the communicator is not up after `timeout_s` seconds the process prints why and exits with status 3
// instead of hanging (mismatched ids and missing peers both show up as exactly that hang).
Transport* make_rccl_transport(int world, int rank, const std::string& id_file, uint64_t nonce, int timeout_s);

// HIP runtime calls the host needs besides the C ABI (kept out of app.cpp, which sees rtpt.h only)
int host_device_count();
void host_set_device(int dev);
void* host_stream_create();
void host_stream_destroy(void* stream);
void* host_device_alloc(size_t bytes);
void host_device_free(void* p);
void host_device_copy(void* dst, const void* src, size_t bytes, void* stream);
void host_device_to_host(void* dst, const void* src, size_t bytes, void* stream);  // blocking
// stream-to-stream ordering without blocking the host: `waiter` continues once everything submitted to `on` so far is done
void host_stream_wait_stream(void* waiter, void* on);
void host_stream_sync(void* stream);
// events (timing disabled): record on a stream now, make another stream wait for that point later
void* host_event_create();
void host_event_destroy(void* ev);
void host_event_record(void* ev, void* stream);
void host_stream_wait_event(void* stream, void* ev);

}  // namespace rtpt_host
