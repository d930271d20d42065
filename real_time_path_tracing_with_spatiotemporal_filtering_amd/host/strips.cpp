// strips.cpp — see strips.hpp.
#include "strips.hpp"

#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <map>
#include <stdexcept>
#include <thread>

namespace rtpt_host {

// ---------------------------------------------------------------------------------------------------- plan
void StripPlan::validate() const {
  if (splits.empty()) return;
  bool ok = static_cast<int>(splits.size()) == world + 1 && splits.front() == 0 && splits.back() == height;
  for (size_t i = 0; ok && i + 1 < splits.size(); i++) ok = splits[i + 1] > splits[i];
  if (!ok) throw std::runtime_error("splits must be " + std::to_string(world + 1) + " ascending rows from 0 to " + std::to_string(height));
}
std::vector<int> balanced_splits(const std::vector<int>& sp, const std::vector<double>& cost, int min_rows) {
  const int world = static_cast<int>(sp.size()) - 1;
  bool ok = world >= 1 && static_cast<int>(cost.size()) == world;
  for (int i = 0; ok && i < world; i++) ok = sp[static_cast<size_t>(i) + 1] > sp[static_cast<size_t>(i)] && cost[static_cast<size_t>(i)] > 0.0;
  if (!ok) throw std::runtime_error("balanced_splits: world + 1 ascending rows and one positive cost per rank");
  if (static_cast<int64_t>(world) * min_rows > sp.back() - sp.front()) throw std::runtime_error("balanced_splits: min_rows does not fit the frame");
  double total = 0.0;
  for (double c : cost) total += c;
  std::vector<int> out{sp.front()};
  int r = 0;
  double acc = 0.0;  // cost of the rows above sp[r]
  for (int j = 1; j < world; j++) {
    const double target = total * j / world;
    while (r < world - 1 && acc + cost[static_cast<size_t>(r)] < target) {
      acc += cost[static_cast<size_t>(r)];
      r++;
    }
    const double y = sp[static_cast<size_t>(r)] + (target - acc) * (sp[static_cast<size_t>(r) + 1] - sp[static_cast<size_t>(r)]) / cost[static_cast<size_t>(r)];
    out.push_back(static_cast<int>(y + 0.5));
  }
  out.push_back(sp.back());
  for (int j = 1; j < world; j++) out[static_cast<size_t>(j)] = std::max(out[static_cast<size_t>(j)], out[static_cast<size_t>(j) - 1] + min_rows);
  for (int j = world - 1; j >= 1; j--) out[static_cast<size_t>(j)] = std::min(out[static_cast<size_t>(j)], out[static_cast<size_t>(j) + 1] - min_rows);
  return out;
}
int StripPlan::halo() const {
  if (world == 1) return 0;
  int h = 0;
  for (int k = 1; k <= iterations; k++) h = exchange ? std::max(h, reach(k)) : h + reach(k);
  return exchange ? std::max(h, reach(1) + svgf_pad()) : h + svgf_pad();
}
Rows StripPlan::grow(int rows) const {
  const Rows o = own();
  return {std::max(0, o.first - rows), std::min(height, o.second + rows)};
}
Rows StripPlan::stored() const { return grow(halo()); }
Rows StripPlan::raytrace_rows() const {
  if (world == 1 || exchange) return own();
  return grow(halo());
}
Rows StripPlan::filter_rows(int k) const {
  if (world == 1 || exchange) return own();
  int remaining = 0;
  for (int j = k + 1; j <= iterations; j++) remaining += reach(j);
  return grow(remaining);
}
std::vector<StripPlan::Exchange> StripPlan::exchange_rows(int k) const {
  std::vector<Exchange> out;
  if (world == 1 || !exchange) return out;
  const Rows o = own();
  const int r = reach(k) + (k == 1 ? svgf_pad() : 0);  // iteration 1's variance taps look 3 traced rows further (strips.py)
  auto check = [&](int peer) {
    const Rows p = rows_of(peer);
    if (p.second - p.first < r || o.second - o.first < r)
      throw std::runtime_error("strip shorter than the " + std::to_string(r) + "-row halo of iteration " + std::to_string(k));
  };
  if (rank > 0) {
    check(rank - 1);
    out.push_back({rank - 1, {o.first, o.first + r}, {o.first - r, o.first}});
  }
  if (rank + 1 < world) {
    check(rank + 1);
    out.push_back({rank + 1, {o.second - r, o.second}, {o.second, o.second + r}});
  }
  return out;
}

// ---------------------------------------------------------------------------------------------------- reprojection reach
namespace {
struct M4 {
  double m[4][4];  // row-major
};
M4 from_cm(const float* a) {  // column-major float[16] (rtpt_ubo) -> row-major
  M4 r;
  for (int c = 0; c < 4; c++)
    for (int i = 0; i < 4; i++) r.m[i][c] = a[c * 4 + i];
  return r;
}
M4 mul(const M4& a, const M4& b) {
  M4 r;
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 4; j++) {
      double s = 0;
      for (int k = 0; k < 4; k++) s += a.m[i][k] * b.m[k][j];
      r.m[i][j] = s;
    }
  return r;
}
}  // namespace

Rows reprojection_rows(const rtpt_ubo& ubo, int width, int height, Rows rows, const double bmin[3], const double bmax[3], float z_near,
                       int pad) {
  // a model matrix that changed since the previous frame: the shader's previous position is not M_prev M^-1 p (it takes
  // the current point's area ratios against the previous triangle, temporalFiltering.comp.glsl:223-233) and nothing
  // bounds it: the whole frame (strips.py:reprojection_rows has the argument)
  if (std::memcmp(ubo.model, ubo.modelPrev, sizeof ubo.model) != 0) return {0, height};
  const M4 V = from_cm(ubo.view), P = from_cm(ubo.proj);
  const M4 PVp = mul(from_cm(ubo.projPrev), from_cm(ubo.viewPrev));
  // camera origin = -R^T t, world direction of a view-space direction = R^T v
  double org[3];
  for (int i = 0; i < 3; i++) org[i] = -(V.m[0][i] * V.m[0][3] + V.m[1][i] * V.m[1][3] + V.m[2][i] * V.m[2][3]);
  double zmin = 1e300, zmax = -1e300;
  for (int c = 0; c < 8; c++) {
    const double p[3] = {(c & 1) ? bmax[0] : bmin[0], (c & 2) ? bmax[1] : bmin[1], (c & 4) ? bmax[2] : bmin[2]};
    const double depth = -(V.m[2][0] * p[0] + V.m[2][1] * p[1] + V.m[2][2] * p[2] + V.m[2][3]);
    zmin = std::min(zmin, depth);
    zmax = std::max(zmax, depth);
  }
  const double z_lo = std::max<double>(z_near, zmin), z_hi = std::max(z_lo, zmax);
  const int y0 = rows.first, y1 = rows.second;
  const int ya = std::min(y0, height - 1), yb = std::max(y0, std::min(y1, height) - 1);
  double lo = 1e300, hi = -1e300;
  for (int xi = 0; xi < 2; xi++)
    for (int yi = 0; yi < 2; yi++) {
      const double x = xi ? width - 1 : 0, y = yi ? yb : ya;
      const double nx = (2.0 * (x + 0.5) - width) / width, ny = (2.0 * (y + 0.5) - height) / height;
      const double v[3] = {nx / P.m[0][0], ny / P.m[1][1], -1.0};
      double d[3];
      for (int i = 0; i < 3; i++) d[i] = V.m[0][i] * v[0] + V.m[1][i] * v[1] + V.m[2][i] * v[2];
      for (int zi = 0; zi < 2; zi++) {
        const double z = zi ? z_hi : z_lo;
        const double p[4] = {org[0] + z * d[0], org[1] + z * d[1], org[2] + z * d[2], 1.0};
        double cy = 0, cw = 0;
        for (int k = 0; k < 4; k++) {
          cy += PVp.m[1][k] * p[k];
          cw += PVp.m[3][k] * p[k];
        }
        if (!(cw > 1e-6)) return {0, height};
        const double ppy = (cy / cw * 0.5 + 0.5) * height;
        lo = std::min(lo, ppy);
        hi = std::max(hi, ppy);
      }
    }
  int n0 = static_cast<int>(std::floor(lo)) - pad, n1 = static_cast<int>(std::floor(hi)) + 1 + pad;
  n0 = std::min(n0, y0);  // background pixels fetch their own pixel (temporalFiltering.comp.glsl:215-217)
  n1 = std::max(n1, y1);
  return {std::max(0, std::min(n0, height)), std::max(0, std::min(n1, height))};
}

std::vector<std::vector<HistoryOp>> history_exchange_plan(int height, int world, const std::vector<Rows>& needs, const std::vector<int>& splits) {
  std::vector<std::vector<HistoryOp>> table(static_cast<size_t>(world));
  for (int r = 0; r < world; r++) {
    const Rows own_r = StripPlan::bounds(height, world, r, splits);
    for (int q = 0; q < world; q++) {
      if (q == r) continue;
      const Rows own_q = StripPlan::bounds(height, world, q, splits);
      const int s0 = std::max(own_r.first, needs[q].first), s1 = std::min(own_r.second, needs[q].second);
      if (s1 > s0) table[r].push_back({q, true, {s0, s1}});
      const int g0 = std::max(own_q.first, needs[r].first), g1 = std::min(own_q.second, needs[r].second);
      if (g1 > g0) table[r].push_back({q, false, {g0, g1}});
    }
  }
  return table;
}

// ---------------------------------------------------------------------------------------------------- transports
namespace {

#define HIP_OK(expr)                                                                                       \
  do {                                                                                                     \
    hipError_t e_ = (expr);                                                                                \
    if (e_ != hipSuccess) throw std::runtime_error(std::string(#expr) + ": " + hipGetErrorString(e_));     \
  } while (0)
#define NCCL_OK(expr)                                                                                      \
  do {                                                                                                     \
    ncclResult_t r_ = (expr);                                                                              \
    if (r_ != ncclSuccess) throw std::runtime_error(std::string(#expr) + ": " + ncclGetErrorString(r_));   \
  } while (0)

class LocalTransport : public Transport {
 public:
  void begin(void* stream) override {
    stream_ = static_cast<hipStream_t>(stream);
    sends_.clear();
    recvs_.clear();
  }
  void send(int src_rank, const void* src, int dst_rank, size_t bytes) override { sends_[{src_rank, dst_rank}].push_back({src, bytes}); }
  void recv(int dst_rank, void* dst, int src_rank, size_t bytes) override { recvs_[{src_rank, dst_rank}].push_back({dst, bytes}); }
  void end() override {
    // the k-th send of (src, dst) meets the k-th receive of (src, dst), like point-to-point messages do
    for (auto& kv : sends_) {
      auto it = recvs_.find(kv.first);
      if (it == recvs_.end() || it->second.size() != kv.second.size()) throw std::runtime_error("unmatched send/recv between local ranks");
      for (size_t i = 0; i < kv.second.size(); i++) {
        if (kv.second[i].second != it->second[i].second) throw std::runtime_error("send/recv size mismatch between local ranks");
        HIP_OK(hipMemcpyAsync(const_cast<void*>(it->second[i].first), kv.second[i].first, kv.second[i].second, hipMemcpyDeviceToDevice, stream_));
        sent_ += kv.second[i].second;
      }
    }
    for (auto& kv : recvs_)
      if (!sends_.count(kv.first)) throw std::runtime_error("receive without a send between local ranks");
  }
  uint64_t bytes_sent() const override { return sent_; }

 private:
  using Msg = std::pair<const void*, size_t>;
  hipStream_t stream_ = nullptr;
  std::map<std::pair<int, int>, std::vector<Msg>> sends_, recvs_;
  uint64_t sent_ = 0;
};

class RcclTransport : public Transport {
 public:
  RcclTransport(int world, int rank, const std::string& id_file, uint64_t nonce, int timeout_s) : rank_(rank) {
    struct Published {
      uint64_t magic, nonce;
      ncclUniqueId id;
    } pub;
    constexpr uint64_t kMagic = 0x3144494C43435452ull;  // "RTCCLID1"
    const auto deadline = std::chrono::steady_clock::now() + std::chrono::seconds(timeout_s);
    if (rank == 0) {
      (void)::unlink(id_file.c_str());  // a file of an earlier launch must not be readable while ours is being made
      pub.magic = kMagic;
      pub.nonce = nonce;
      NCCL_OK(ncclGetUniqueId(&pub.id));
      const std::string tmp = id_file + ".tmp." + std::to_string(::getpid());
      {
        std::ofstream f(tmp, std::ios::binary | std::ios::trunc);
        f.write(reinterpret_cast<const char*>(&pub), sizeof pub);
        f.flush();
        if (!f) throw std::runtime_error("cannot write the RCCL id to " + tmp);
      }
      if (std::rename(tmp.c_str(), id_file.c_str()) != 0) throw std::runtime_error("cannot publish the RCCL id at " + id_file);
    } else {
      for (;;) {
        std::ifstream f(id_file, std::ios::binary);
        if (f && f.read(reinterpret_cast<char*>(&pub), sizeof pub) && pub.magic == kMagic && pub.nonce == nonce) break;
        if (std::chrono::steady_clock::now() > deadline)
          throw std::runtime_error("timed out waiting for an RCCL id with this launch's nonce at " + id_file +
                                   " (is rank 0 running, with the same --rccl-nonce?)");
        std::this_thread::sleep_for(std::chrono::milliseconds(50));
      }
    }
    // ncclCommInitRank blocks until all `world` ranks have joined: a peer that never comes, or one holding another id,
    // is an endless wait.  The watchdog turns that into an error exit.
    std::atomic<bool> up{false};
    std::thread watchdog([&up, deadline, rank, world] {
      while (!up.load()) {
        if (std::chrono::steady_clock::now() > deadline) {
          std::fprintf(stderr, "rtpt_app: rank %d: the RCCL communicator of %d ranks did not come up in time (a rank is missing or "
                               "holds the id of another launch); giving up\n", rank, world);
          std::fflush(stderr);
          ::_exit(3);
        }
        std::this_thread::sleep_for(std::chrono::milliseconds(100));
      }
    });
    const ncclResult_t rc = ncclCommInitRank(&comm_, world, pub.id, rank);
    up.store(true);
    watchdog.join();
    NCCL_OK(rc);
  }
  ~RcclTransport() override {
    if (comm_) ncclCommDestroy(comm_);
  }
  void begin(void* stream) override {
    stream_ = static_cast<hipStream_t>(stream);
    NCCL_OK(ncclGroupStart());
  }
  void send(int src_rank, const void* src, int dst_rank, size_t bytes) override {
    if (src_rank != rank_) throw std::runtime_error("RCCL transport: send from a rank of another process");
    NCCL_OK(ncclSend(src, bytes, ncclChar, dst_rank, comm_, stream_));
    sent_ += bytes;
  }
  void recv(int dst_rank, void* dst, int src_rank, size_t bytes) override {
    if (dst_rank != rank_) throw std::runtime_error("RCCL transport: receive into a rank of another process");
    NCCL_OK(ncclRecv(dst, bytes, ncclChar, src_rank, comm_, stream_));
  }
  void end() override { NCCL_OK(ncclGroupEnd()); }
  uint64_t bytes_sent() const override { return sent_; }

 private:
  int rank_;
  ncclComm_t comm_ = nullptr;
  hipStream_t stream_ = nullptr;
  uint64_t sent_ = 0;
};

}  // namespace

int host_device_count() {
  int n = 0;
  return hipGetDeviceCount(&n) == hipSuccess ? n : 0;
}
void host_set_device(int dev) { HIP_OK(hipSetDevice(dev)); }
void* host_stream_create() {
  hipStream_t s = nullptr;
  HIP_OK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  return s;
}
void host_stream_destroy(void* stream) { (void)hipStreamDestroy(static_cast<hipStream_t>(stream)); }
void* host_device_alloc(size_t bytes) {
  void* p = nullptr;
  HIP_OK(hipMalloc(&p, bytes));
  return p;
}
void host_device_free(void* p) { (void)hipFree(p); }
void host_device_copy(void* dst, const void* src, size_t bytes, void* stream) {
  HIP_OK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, static_cast<hipStream_t>(stream)));
}

void host_device_to_host(void* dst, const void* src, size_t bytes, void* stream) {
  HIP_OK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, static_cast<hipStream_t>(stream)));
  HIP_OK(hipStreamSynchronize(static_cast<hipStream_t>(stream)));
}
void host_stream_wait_stream(void* waiter, void* on) {
  hipEvent_t e = nullptr;
  HIP_OK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  HIP_OK(hipEventRecord(e, static_cast<hipStream_t>(on)));
  HIP_OK(hipStreamWaitEvent(static_cast<hipStream_t>(waiter), e, 0));
  HIP_OK(hipEventDestroy(e));  // the recorded wait keeps what it needs; the handle can go
}
void host_stream_sync(void* stream) { HIP_OK(hipStreamSynchronize(static_cast<hipStream_t>(stream))); }
void* host_event_create() {
  hipEvent_t e = nullptr;
  HIP_OK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  return e;
}
void host_event_destroy(void* ev) { (void)hipEventDestroy(static_cast<hipEvent_t>(ev)); }
void host_event_record(void* ev, void* stream) { HIP_OK(hipEventRecord(static_cast<hipEvent_t>(ev), static_cast<hipStream_t>(stream))); }
void host_stream_wait_event(void* stream, void* ev) { HIP_OK(hipStreamWaitEvent(static_cast<hipStream_t>(stream), static_cast<hipEvent_t>(ev), 0)); }

Transport* make_local_transport() { return new LocalTransport(); }
Transport* make_rccl_transport(int world, int rank, const std::string& id_file, uint64_t nonce, int timeout_s) {
  return new RcclTransport(world, rank, id_file, nonce, timeout_s);
}

}  // namespace rtpt_host
