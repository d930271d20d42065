// app.cpp — see app.hpp.  Every method cites the reference lines it stands for.
#include "app.hpp"
#include "scene_gen.hpp"

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <stdexcept>

namespace rtpt_host {

namespace {
constexpr float kFov = 0.20f;    // common.h:16
constexpr float kSpeed = 0.1f;   // main.cpp:68

void identity(float m[16]) {
  std::memset(m, 0, 16 * sizeof(float));
  m[0] = m[5] = m[10] = m[15] = 1.0f;
}
}  // namespace

PathTracingApplication::PathTracingApplication(const Options& opt) : opt_(opt) {}

PathTracingApplication::~PathTracingApplication() { freeRessources(); }

void PathTracingApplication::check(int rc, const char* what) {
  // the reference aborts through NVVK_CHECK / throws std::runtime_error (main.cpp:99-111, :125)
  if (rc != RTPT_OK) throw std::runtime_error(std::string(what) + ": " + rtpt_last_error(ctx_));
}

void PathTracingApplication::run(int frames, const std::vector<std::string>& script) {
  initVulkan();
  for (int f = 0; f < frames; f++)  // mainLoop, main.cpp:301-307
    drawScene(static_cast<size_t>(f) < script.size() ? script[static_cast<size_t>(f)] : std::string());
}

void PathTracingApplication::initVulkan() {
  loadMesh();
  createBuffers();
  // uploadBuffers, main.cpp:481-489: first view looks at (0,1,0); prev matrices = current
  identity(ubo.model);
  const float center[3] = {0.0f, 1.0f, 0.0f}, up[3] = {0.0f, 1.0f, 0.0f};
  rtpt_util_look_at(cameraOrigin, center, up, ubo.view);
  rtpt_util_perspective(kFov * 2, static_cast<float>(opt_.width) / static_cast<float>(opt_.height), 0.1f, zFar_, ubo.proj);
  ubo.proj[5] *= -1;
  std::memcpy(ubo.modelPrev, ubo.model, sizeof ubo.model);
  std::memcpy(ubo.viewPrev, ubo.view, sizeof ubo.view);
  std::memcpy(ubo.projPrev, ubo.proj, sizeof ubo.proj);
  buildAccelerationStructure();
  initializeSceneConstants();
}

void PathTracingApplication::loadMesh() {
  uint32_t nv = 0, nt = 0;
  check(rtpt_util_load_obj(opt_.scene.c_str(), nullptr, &nv, nullptr, &nt), "loadMesh");
  objVertices.resize(static_cast<size_t>(nv) * 3);
  objIndices.resize(static_cast<size_t>(nt) * 3);
  check(rtpt_util_load_obj(opt_.scene.c_str(), objVertices.data(), &nv, objIndices.data(), &nt), "loadMesh");
  // the material side of the file (tinyobjloader returns it too, main.cpp:416-421; the reference ignores it and its
  // OBJ's library is missing upstream): used when the OBJ names a readable .mtl, otherwise the normal-keyed colours stay
  uint32_t nm = 0, ntm = 0;
  check(rtpt_util_load_obj_materials(opt_.scene.c_str(), nullptr, &ntm, nullptr, &nm), "loadMesh");
  objMaterials.clear();
  triMaterial.clear();
  if (nm > 0 && ntm == nt) {
    objMaterials.resize(nm);
    triMaterial.resize(ntm);
    check(rtpt_util_load_obj_materials(opt_.scene.c_str(), triMaterial.data(), &ntm, objMaterials.data(), &nm), "loadMesh");
  }
  // configs[4]: tessellated quads on a lattice of instances (scene_gen.hpp); camera, light and far plane frame the lattice
  if (opt_.tessellate > 1) {
    if (!objMaterials.empty()) throw std::runtime_error("--tessellate does not carry a material library over");
    if (!tessellate_quads(objVertices, objIndices, opt_.tessellate, objVertices, objIndices))
      throw std::runtime_error("--tessellate needs a mesh of fan-triangulated quads");
  }
  instanceXforms_.clear();
  if (!opt_.instances.empty()) {
    if (opt_.lattice[0] > 0) throw std::runtime_error("--instances and --lattice exclude each other");
    std::ifstream f(opt_.instances);
    if (!f) throw std::runtime_error("cannot read " + opt_.instances);
    float v;
    while (f >> v) instanceXforms_.push_back(v);
    if (instanceXforms_.empty() || instanceXforms_.size() % 12) throw std::runtime_error(opt_.instances + ": 12 floats per instance expected");
  }
  if (opt_.lattice[0] > 0) {
    instanceXforms_ = lattice_xforms(opt_.lattice[0], opt_.lattice[1], opt_.lattice[2], opt_.pitch);
    const LatticeView v = lattice_view(opt_.lattice[0], opt_.lattice[1], opt_.lattice[2], opt_.pitch);
    std::memcpy(cameraOrigin, v.camera, sizeof cameraOrigin);
    std::memcpy(lightPos, v.light, sizeof lightPos);
    zFar_ = v.z_far;
  }
}

void PathTracingApplication::sceneBounds() {
  scene_bounds(objVertices, objIndices, instanceXforms_.empty() ? nullptr : instanceXforms_.data(),
               static_cast<uint32_t>(instanceXforms_.size() / 12), sceneMin_, sceneMax_);
}

void PathTracingApplication::dumpScene(const std::string& path) {
  std::ofstream f(path, std::ios::binary);
  const uint32_t head[3] = {static_cast<uint32_t>(objVertices.size() / 3), static_cast<uint32_t>(objIndices.size() / 3),
                            static_cast<uint32_t>(instanceXforms_.size() / 12)};
  f.write(reinterpret_cast<const char*>(head), sizeof head);
  f.write(reinterpret_cast<const char*>(objVertices.data()), static_cast<std::streamsize>(objVertices.size() * 4));
  f.write(reinterpret_cast<const char*>(objIndices.data()), static_cast<std::streamsize>(objIndices.size() * 4));
  f.write(reinterpret_cast<const char*>(instanceXforms_.data()), static_cast<std::streamsize>(instanceXforms_.size() * 4));
  f.write(reinterpret_cast<const char*>(cameraOrigin), sizeof cameraOrigin);
  f.write(reinterpret_cast<const char*>(lightPos), sizeof lightPos);
  f.write(reinterpret_cast<const char*>(&zFar_), sizeof zFar_);
  if (!f) throw std::runtime_error("cannot write " + path);
}

void PathTracingApplication::createBuffers() {
  rtpt_config cfg;
  check(rtpt_config_default(&cfg, opt_.width, opt_.height), "rtpt_config_default");
  cfg.max_segments = opt_.max_segments;
  cfg.flags = opt_.flags;
  if (multi()) {
    // one strip context per rank this process runs; all on one stream, which is also the stream the transport's
    // messages are ordered on (RCCL point-to-point calls or device-to-device copies)
    // two frames in flight on strips: every rank builds even frames in one context and odd frames in another, the ranks'
    // contexts of one parity share a stream; the finished strip — and, with the guided extension modes, its id / moment
    // planes — are handed across like on one GPU, or as bands of the ranks' other contexts when the camera moved
    // (prepareHistory, prepareGuides)
    if (opt_.frames_in_flight != 1 && opt_.frames_in_flight != 2) throw std::runtime_error("frames_in_flight must be 1 or 2");
    const bool local = opt_.rank < 0;
    if (!local && opt_.rank >= opt_.ranks) throw std::runtime_error("rank out of range");
    int dev = opt_.device;
    if (dev < 0 && !local) {
      const int n = host_device_count();
      dev = n > 0 ? opt_.rank % n : 0;
    }
    if (dev >= 0) host_set_device(dev);
    cfg.device = dev;
    for (int i = 0; i < opt_.frames_in_flight; i++) streams_[i] = host_stream_create();
    stream_ = streams_[0];
    transport_ = local ? make_local_transport()
                       : make_rccl_transport(opt_.ranks, opt_.rank, opt_.rccl_id_file, opt_.rccl_nonce, opt_.rccl_timeout_s);
    if (opt_.present) {
      presentStream_ = host_stream_create();
      presentTransport_ = local ? make_local_transport()
                                : make_rccl_transport(opt_.ranks, opt_.rank, opt_.rccl_id_file + ".present", opt_.rccl_nonce, opt_.rccl_timeout_s);
    }
    for (int r = local ? 0 : opt_.rank; r < (local ? opt_.ranks : opt_.rank + 1); r++) {
      RankState rs;
      rs.plan.height = static_cast<int>(opt_.height);
      rs.plan.world = opt_.ranks;
      rs.plan.rank = r;
      rs.plan.iterations = opt_.maxWaveletIteration;
      rs.plan.exchange = opt_.exchange_halo;
      rs.plan.ext_flags = opt_.flags & 0x9F0u;
      rs.plan.splits = opt_.splits;
      rs.plan.validate();
      const Rows st = rs.plan.stored(), own = rs.plan.own();
      cfg.row_begin = static_cast<uint32_t>(st.first);
      cfg.row_end = static_cast<uint32_t>(st.second);
      for (int i = 0; i < opt_.frames_in_flight; i++) {
        check(rtpt_create(&cfg, &rs.ctxs[i]), "createBuffers");
        check(rtpt_set_stream(rs.ctxs[i], streams_[i]), "rtpt_set_stream");
        check(rtpt_set_count_rows(rs.ctxs[i], static_cast<uint32_t>(own.first), static_cast<uint32_t>(own.second)), "rtpt_set_count_rows");
      }
      rs.ctx = rs.last = rs.ctxs[0];
      ranks_.push_back(rs);
    }
    ctx_ = last_ = ranks_[0].ctx;
    return;
  }
  if (opt_.frames_in_flight != 1 && opt_.frames_in_flight != 2) throw std::runtime_error("frames_in_flight must be 1 or 2");
  for (int i = 0; i < opt_.frames_in_flight; i++) check(rtpt_create(&cfg, &ctxs_[i]), "createBuffers");
  ctx_ = last_ = ctxs_[0];
}

void PathTracingApplication::buildAccelerationStructure() {
  // world-space bounds of the scene: strips bound the reprojection reach with them (strips.hpp)
  sceneBounds();
  auto materials = [&](rtpt_ctx* ctx) {
    if (!objMaterials.empty())
      check(rtpt_scene_set_materials(ctx, triMaterial.data(), static_cast<uint32_t>(triMaterial.size()), objMaterials.data(),
                                     static_cast<uint32_t>(objMaterials.size())),
            "rtpt_scene_set_materials");
  };
  // main.cpp:728-741: the instance list (one identity transform in the reference; the lattice of configs[4] here)
  const float* xf = instanceXforms_.empty() ? nullptr : instanceXforms_.data();
  const uint32_t n_inst = static_cast<uint32_t>(instanceXforms_.size() / 12);
  if (multi()) {
    for (auto& rs : ranks_)
      for (int i = 0; i < opt_.frames_in_flight; i++) {
        check(rtpt_scene_upload(rs.ctxs[i], objVertices.data(), static_cast<uint32_t>(objVertices.size() / 3), objIndices.data(),
                                static_cast<uint32_t>(objIndices.size() / 3), xf, n_inst),
              "buildAccelerationStructure");
        materials(rs.ctxs[i]);
      }
    return;
  }
  for (int i = 0; i < opt_.frames_in_flight; i++) {
    check(rtpt_scene_upload(ctxs_[i], objVertices.data(), static_cast<uint32_t>(objVertices.size() / 3), objIndices.data(),
                            static_cast<uint32_t>(objIndices.size() / 3), xf, n_inst),
          "buildAccelerationStructure");
    materials(ctxs_[i]);
  }
}

void PathTracingApplication::initializeSceneConstants() {
  std::memcpy(pushConstants.currentCameraColor, lightColor, sizeof lightColor);
  std::memcpy(pushConstants.lightPos, lightPos, sizeof lightPos);
  // lightPosPrev is a zero-initialised global at this point (main.cpp:71, :665)
  pushConstants.lightPosPrev[0] = pushConstants.lightPosPrev[1] = pushConstants.lightPosPrev[2] = 0.0f;
}

void PathTracingApplication::updateUBO() {
  std::memcpy(ubo.modelPrev, ubo.model, sizeof ubo.model);
  std::memcpy(ubo.viewPrev, ubo.view, sizeof ubo.view);
  std::memcpy(ubo.projPrev, ubo.proj, sizeof ubo.proj);
  identity(ubo.model);
  const float center[3] = {cameraOrigin[0], cameraOrigin[1], cameraOrigin[2] - 6.0f}, up[3] = {0.0f, 1.0f, 0.0f};
  rtpt_util_look_at(cameraOrigin, center, up, ubo.view);
  rtpt_util_perspective(kFov * 2, static_cast<float>(opt_.width) / static_cast<float>(opt_.height), 0.1f, zFar_, ubo.proj);
  ubo.proj[5] *= -1;
}

void PathTracingApplication::updateScene(const std::string& keys) {
  auto held = [&](char k) { return keys.find(k) != std::string::npos; };
  if (held('S')) { cameraOrigin[2] += kSpeed; cameraMoved = true; }
  if (held('W')) { cameraOrigin[2] -= kSpeed; cameraMoved = true; }
  if (held('A')) { cameraOrigin[0] -= kSpeed; cameraMoved = true; }
  if (held('D')) { cameraOrigin[0] += kSpeed; cameraMoved = true; }
  if (held('E')) { cameraOrigin[1] += kSpeed; cameraMoved = true; }
  if (held('Q')) { cameraOrigin[1] -= kSpeed; cameraMoved = true; }
  if (held('I')) lightPos[2] -= kSpeed;
  if (held('K')) lightPos[2] += kSpeed;
  if (held('L')) { lightPos[0] += kSpeed; if (lightPos[0] > 2) lightPos[0] = -20; }
  if (held('J')) { lightPos[0] -= kSpeed; if (lightPos[0] < -20) lightPos[0] = 2; }
  if (held('O')) lightPos[1] += kSpeed;
  if (held('U')) lightPos[1] -= kSpeed;
  pushConstants.frameNumber = frameCount;                                                                  // :1171
  std::memcpy(pushConstants.previousCameraColor, pushConstants.currentCameraColor, 3 * sizeof(float));     // :1173
  std::memcpy(pushConstants.currentCameraColor, lightColor, sizeof lightColor);                            // :1175
  std::memcpy(pushConstants.lightPosPrev, pushConstants.lightPos, 3 * sizeof(float));                      // :1177
  std::memcpy(pushConstants.lightPos, lightPos, sizeof lightPos);                                          // :1178
  updateUBO();                                                                                             // :1180
  if (cameraMoved || frameCount == 0) {                                                                    // :1181
    std::memcpy(pushConstants.cameraPos, cameraOrigin, sizeof cameraOrigin);
    cameraMoved = false;
  }
}

void PathTracingApplication::drawVisbilityBuffer() {
  if (multi()) {
    for (auto& rs : ranks_) {
      const Rows r = rs.plan.gbuffer_rows();
      check(rtpt_gbuffer(rs.ctx, &ubo, static_cast<uint32_t>(r.first), static_cast<uint32_t>(r.second)), "drawVisbilityBuffer");
    }
    return;
  }
  check(rtpt_gbuffer(ctx_, &ubo, 0, 0), "drawVisbilityBuffer");
}

void PathTracingApplication::computeTemporalGradient() {
  if (multi()) {
    for (auto& rs : ranks_) {
      const Rows r = rs.plan.gradient_rows();
      check(rtpt_temporal_gradient(rs.ctx, &pushConstants, static_cast<uint32_t>(r.first), static_cast<uint32_t>(r.second)),
            "computeTemporalGradient");
    }
    return;
  }
  check(rtpt_temporal_gradient(ctx_, &pushConstants, 0, 0), "computeTemporalGradient");
}

void PathTracingApplication::drawSceneToImage() {
  pushConstants.sample_batch = 0;  // NUM_SAMPLE_BATCHES = 1, main.cpp:1223,:1237
  if (multi()) {
    for (auto& rs : ranks_) {
      const Rows r = rs.plan.raytrace_rows();
      check(rtpt_raytrace(rs.ctx, &pushConstants, static_cast<uint32_t>(r.first), static_cast<uint32_t>(r.second)), "drawSceneToImage");
    }
    return;
  }
  check(rtpt_raytrace(ctx_, &pushConstants, 0, 0), "drawSceneToImage");
}

bool PathTracingApplication::cameraStatic() const {
  // "static" = every pixel reprojects onto itself: view, proj AND model unchanged (a moving model matrix sends pixels
  // to other rows just like a moving camera does; this host's model is the identity of main.cpp:1469 every frame)
  return std::memcmp(ubo.view, ubo.viewPrev, sizeof ubo.view) == 0 && std::memcmp(ubo.proj, ubo.projPrev, sizeof ubo.proj) == 0 &&
         std::memcmp(ubo.model, ubo.modelPrev, sizeof ubo.model) == 0;
}

// exchange mode: before iteration k every rank sends its k boundary rows of the iteration's INPUT plane (rgbd cells, so
// the depth travels with the colour) to each neighbour — the one real exchange step of the path
void PathTracingApplication::exchangeHalo(int k) {
  exchangeHaloPlane(k, (k & 1) ? RTPT_PLANE_IMAGE : RTPT_PLANE_FILTERED, 16);
  // RTPT_FLAG_EXT_VARIANCE: the variance iteration k-1 filtered travels with the colour it guides; iteration 1's comes from
  // the moment accumulation, which every rank also runs on the halo rows it has just received the colour of
  if (k > 1 && (opt_.flags & RTPT_FLAG_EXT_VARIANCE)) exchangeHaloPlane(k, RTPT_PLANE_VARIANCE, 4);
}

void PathTracingApplication::exchangeHaloPlane(int k, rtpt_plane in_plane, size_t px_bytes) {
  const size_t row_bytes = static_cast<size_t>(opt_.width) * px_bytes;
  transport_->begin(stream_);
  for (auto& rs : ranks_) {
    void* base = nullptr;
    check(rtpt_plane_ptr(rs.ctx, in_plane, &base), "rtpt_plane_ptr");  // also launches what rtpt_temporal_filter recorded
    const int row0 = rs.plan.stored().first;
    auto rows_ptr = [&](int y) { return static_cast<char*>(base) + static_cast<size_t>(y - row0) * row_bytes; };
    for (const auto& e : rs.plan.exchange_rows(k)) {
      transport_->send(rs.plan.rank, rows_ptr(e.send.first), e.peer, static_cast<size_t>(e.send.second - e.send.first) * row_bytes);
      transport_->recv(rs.plan.rank, rows_ptr(e.recv.first), e.peer, static_cast<size_t>(e.recv.second - e.recv.first) * row_bytes);
    }
  }
  transport_->end();
}

// The final pass fetches previousFrameImage at the reprojected pixel (temporalFiltering.comp.glsl:253).  While the
// camera rests that is the pixel itself and the strip-local PREVIOUS plane serves it.  When it moved, every rank bounds
// the previous-frame rows its pixels can reach (reprojection_rows: the same numbers on every rank, no negotiation) and
// the ranks swap exactly those bands of their finished strips.
void PathTracingApplication::prepareHistory() {
  const bool handed = opt_.frames_in_flight == 2 && frameCount > 0;  // the previous frame rests in the rank's OTHER context
  if (handed)  // ... which finished (or is finishing) it on the other stream
    for (auto& rs : ranks_) check(rtpt_stream_wait(rs.ctx, rs.last), "rtpt_stream_wait");
  if (frameCount == 0 || cameraStatic()) {
    for (auto& rs : ranks_) {
      void* prev = nullptr;
      const Rows st = rs.plan.stored();
      if (handed) check(rtpt_plane_ptr(rs.last, RTPT_PLANE_PREVIOUS, &prev), "rtpt_plane_ptr");
      check(rtpt_set_external_history(rs.ctx, prev, handed ? static_cast<uint32_t>(st.first) : 0u, handed ? static_cast<uint32_t>(st.second) : 0u),
            "rtpt_set_external_history");
    }
    return;
  }
  const int H = static_cast<int>(opt_.height);
  const size_t row_bytes = static_cast<size_t>(opt_.width) * 16;
  std::vector<Rows> needs;
  for (int r = 0; r < opt_.ranks; r++)
    needs.push_back(reprojection_rows(ubo, static_cast<int>(opt_.width), H, StripPlan::bounds(H, opt_.ranks, r, opt_.splits), sceneMin_, sceneMax_, 0.1f));
  const auto table = history_exchange_plan(H, opt_.ranks, needs, opt_.splits);
  transport_->begin(stream_);
  for (auto& rs : ranks_) {
    if (!rs.history) rs.history = host_device_alloc(static_cast<size_t>(H) * row_bytes);
    void* prev = nullptr;
    check(rtpt_plane_ptr(rs.last, RTPT_PLANE_PREVIOUS, &prev), "rtpt_plane_ptr");  // rs.last == rs.ctx with one frame in flight
    const int row0 = rs.plan.stored().first;
    auto prev_rows = [&](int y) { return static_cast<char*>(prev) + static_cast<size_t>(y - row0) * row_bytes; };
    auto hist_rows = [&](int y) { return static_cast<char*>(rs.history) + static_cast<size_t>(y) * row_bytes; };
    const Rows own = rs.plan.own(), need = needs[static_cast<size_t>(rs.plan.rank)];
    const int a = std::max(own.first, need.first), b = std::min(own.second, need.second);
    if (b > a) host_device_copy(hist_rows(a), prev_rows(a), static_cast<size_t>(b - a) * row_bytes, stream_);
    for (const auto& op : table[static_cast<size_t>(rs.plan.rank)]) {
      const size_t bytes = static_cast<size_t>(op.rows.second - op.rows.first) * row_bytes;
      if (op.send)
        transport_->send(rs.plan.rank, prev_rows(op.rows.first), op.peer, bytes);
      else
        transport_->recv(rs.plan.rank, hist_rows(op.rows.first), op.peer, bytes);
    }
  }
  transport_->end();
  for (auto& rs : ranks_) {
    const Rows need = needs[static_cast<size_t>(rs.plan.rank)];
    check(rtpt_set_external_history(rs.ctx, static_cast<char*>(rs.history) + static_cast<size_t>(need.first) * row_bytes,
                                    static_cast<uint32_t>(need.first), static_cast<uint32_t>(need.second)),
          "rtpt_set_external_history");
  }
}

// One plane of the previous frame, gathered where the ranks' pixels can reach it: rank r holds rows needs[r] of `plane`
// in its full-frame buffer RankState::*dst afterwards (its own rows by a device copy, the others' through the transport)
void PathTracingApplication::exchangeBands(const std::vector<Rows>& needs, rtpt_plane plane, size_t px_bytes, void* RankState::*dst) {
  const int H = static_cast<int>(opt_.height);
  const size_t row_bytes = static_cast<size_t>(opt_.width) * px_bytes;
  const auto table = history_exchange_plan(H, opt_.ranks, needs, opt_.splits);
  transport_->begin(stream_);
  for (auto& rs : ranks_) {
    if (!(rs.*dst)) rs.*dst = host_device_alloc(static_cast<size_t>(H) * row_bytes);
    void* src = nullptr;
    check(rtpt_plane_ptr(rs.last, plane, &src), "rtpt_plane_ptr");  // rs.last == rs.ctx with one frame in flight
    const int row0 = rs.plan.stored().first;
    auto src_rows = [&](int y) { return static_cast<char*>(src) + static_cast<size_t>(y - row0) * row_bytes; };
    auto dst_rows = [&](int y) { return static_cast<char*>(rs.*dst) + static_cast<size_t>(y) * row_bytes; };
    // what this rank owns of what it needs; everything else comes from its owner (strips.py: exchange_history)
    const Rows own = rs.plan.own(), need = needs[static_cast<size_t>(rs.plan.rank)];
    const int a = std::max(own.first, need.first), b = std::min(own.second, need.second);
    if (b > a) host_device_copy(dst_rows(a), src_rows(a), static_cast<size_t>(b - a) * row_bytes, stream_);
    for (const auto& op : table[static_cast<size_t>(rs.plan.rank)]) {
      const size_t bytes = static_cast<size_t>(op.rows.second - op.rows.first) * row_bytes;
      if (op.send)
        transport_->send(rs.plan.rank, src_rows(op.rows.first), op.peer, bytes);
      else
        transport_->recv(rs.plan.rank, dst_rows(op.rows.first), op.peer, bytes);
    }
  }
  transport_->end();
}

// RTPT_FLAG_EXT_VARIANCE / _DISOCCLUSION on strips: the moment accumulation (before iteration 1) and the disocclusion test
// (final pass) read the PREVIOUS frame's id and moment planes at reprojected pixels.  Every rank holds them for its stored
// rows; when the camera (or the model) moved, the rows a rank's stored pixels can reach beyond that are gathered from their
// owners — the history image's bound and plan — into buffers registered with rtpt_set_external_guides.
void PathTracingApplication::prepareGuides() {
  const bool handed = opt_.frames_in_flight == 2 && frameCount > 0;  // the previous frame's planes rest in the rank's OTHER context
  const bool variance = (opt_.flags & RTPT_FLAG_EXT_VARIANCE) != 0;
  if (handed)
    for (auto& rs : ranks_) check(rtpt_stream_wait(rs.ctx, rs.last), "rtpt_stream_wait");
  if (frameCount == 0 || cameraStatic()) {
    for (auto& rs : ranks_) {
      void *ids = nullptr, *mom = nullptr;
      const Rows st = rs.plan.stored();
      if (handed) {
        check(rtpt_plane_ptr(rs.last, RTPT_PLANE_PREV_VIS_ID, &ids), "rtpt_plane_ptr");
        if (variance) check(rtpt_plane_ptr(rs.last, RTPT_PLANE_MOMENTS_PREV, &mom), "rtpt_plane_ptr");
      }
      check(rtpt_set_external_guides(rs.ctx, ids, mom, handed ? static_cast<uint32_t>(st.first) : 0u, handed ? static_cast<uint32_t>(st.second) : 0u),
            "rtpt_set_external_guides");
    }
    return;
  }
  const int H = static_cast<int>(opt_.height);
  std::vector<Rows> needs;
  for (int r = 0; r < opt_.ranks; r++) {
    StripPlan p = ranks_[0].plan;
    p.rank = r;
    needs.push_back(reprojection_rows(ubo, static_cast<int>(opt_.width), H, p.stored(), sceneMin_, sceneMax_, 0.1f));
  }
  exchangeBands(needs, RTPT_PLANE_PREV_VIS_ID, 4, &RankState::guide_ids);
  if (variance) exchangeBands(needs, RTPT_PLANE_MOMENTS_PREV, 16, &RankState::guide_moments);
  for (auto& rs : ranks_) {
    const Rows need = needs[static_cast<size_t>(rs.plan.rank)];
    const size_t W = opt_.width;
    check(rtpt_set_external_guides(rs.ctx, static_cast<char*>(rs.guide_ids) + static_cast<size_t>(need.first) * W * 4,
                                   variance ? static_cast<char*>(rs.guide_moments) + static_cast<size_t>(need.first) * W * 16 : nullptr,
                                   static_cast<uint32_t>(need.first), static_cast<uint32_t>(need.second)),
          "rtpt_set_external_guides");
  }
}

void* PathTracingApplication::presentImage(RankState* rs, int idx) {
  const size_t bytes = static_cast<size_t>(opt_.width) * opt_.height * (opt_.present == 1 ? 4 : 16);
  void*& p = rs ? rs->swap[idx] : swapSingle_[idx];
  if (!p) p = host_device_alloc(bytes);
  return p;
}

// vkAcquireNextImageKHR's role (main.cpp:1310-1316), at the top of the frame: what the gather of two frames ago read may be
// written again only once that gather is done.  For rgba8 that is swapchain image idx (the fused final pass stores into
// it); for f32 it is the colour buffer itself — the PREVIOUS plane a rank sent from at frame f is IMAGE again at frame
// f + 2 and rtpt_raytrace overwrites it, so the wait has to stand before the trace, not before the blit (round-3 advice:
// with redundant halo rows and a camera at rest nothing else couples the ranks, and a rank could run frames ahead of the
// presenting rank's receive).  app.py::_acquire is the same rule.
void PathTracingApplication::acquirePresent() {
  if (!opt_.present || !multi()) return;
  const int idx = static_cast<int>(frameCount & 1);
  if (presentDone_[idx]) host_stream_wait_event(stream_, presentDone_[idx]);
}

// the rows of swapchain image idx are named to the final filter pass, which stores them in swapchain format itself
void PathTracingApplication::armPresent() {
  if (opt_.present != 1) return;
  const int idx = static_cast<int>(frameCount & 1);
  if (!multi()) {
    check(rtpt_present_target(ctx_, presentImage(nullptr, idx), 0, opt_.height), "rtpt_present_target");
    return;
  }
  const size_t row_bytes = static_cast<size_t>(opt_.width) * 4;
  for (auto& rs : ranks_) {
    const Rows own = rs.plan.own();
    char* img = static_cast<char*>(presentImage(&rs, idx));
    check(rtpt_present_target(rs.ctx, img + static_cast<size_t>(own.first) * row_bytes, static_cast<uint32_t>(own.first),
                              static_cast<uint32_t>(own.second)),
          "rtpt_present_target");
  }
}

void PathTracingApplication::applyTemporalFiltering() {
  pushConstants.maxWaveletIteration = opt_.maxWaveletIteration;   // :1258
  armPresent();
  if (multi()) {
    if (opt_.flags & (RTPT_FLAG_EXT_VARIANCE | RTPT_FLAG_EXT_DISOCCLUSION)) prepareGuides();
    for (int k = 1; k <= opt_.maxWaveletIteration; k++) {         // :1259
      pushConstants.waveletIteration = k;                         // :1260
      if (opt_.exchange_halo) exchangeHalo(k);
      if (k == opt_.maxWaveletIteration && (k & 1)) prepareHistory();
      for (auto& rs : ranks_) {
        const Rows r = rs.plan.filter_rows(k);
        check(rtpt_temporal_filter(rs.ctx, &pushConstants, &ubo, static_cast<uint32_t>(r.first), static_cast<uint32_t>(r.second)),
              "applyTemporalFiltering");
      }
    }
    return;
  }
  for (int k = 1; k <= opt_.maxWaveletIteration; k++) {           // :1259
    pushConstants.waveletIteration = k;                           // :1260
    if (opt_.frames_in_flight == 2 && k == 1 && frameCount > 0 && (opt_.flags & (RTPT_FLAG_EXT_VARIANCE | RTPT_FLAG_EXT_DISOCCLUSION))) {
      // the moment accumulation (this iteration) and the disocclusion test (the final one) read the previous frame's id and
      // moment planes, which the other context holds: handed across like the history image (app.py: PipelinedBackend)
      void *ids = nullptr, *mom = nullptr;
      check(rtpt_stream_wait(ctx_, last_), "rtpt_stream_wait");
      check(rtpt_plane_ptr(last_, RTPT_PLANE_PREV_VIS_ID, &ids), "rtpt_plane_ptr");
      if (opt_.flags & RTPT_FLAG_EXT_VARIANCE) check(rtpt_plane_ptr(last_, RTPT_PLANE_MOMENTS_PREV, &mom), "rtpt_plane_ptr");
      check(rtpt_set_external_guides(ctx_, ids, mom, 0, opt_.height), "rtpt_set_external_guides");
    }
    if (opt_.frames_in_flight == 2 && k == opt_.maxWaveletIteration && (k & 1) && frameCount > 0) {
      // the blend reads the previous frame, which the other context finished (or is finishing) on its own stream
      void* prev = nullptr;
      check(rtpt_stream_wait(ctx_, last_), "rtpt_stream_wait");
      check(rtpt_plane_ptr(last_, RTPT_PLANE_PREVIOUS, &prev), "rtpt_plane_ptr");
      check(rtpt_set_external_history(ctx_, prev, 0, opt_.height), "rtpt_set_external_history");
    }
    // the descriptor swap of :1264-1281 is the ping-pong rule inside rtpt_temporal_filter
    check(rtpt_temporal_filter(ctx_, &pushConstants, &ubo, 0, 0), "applyTemporalFiltering");
  }
}

void PathTracingApplication::copyImageToSwapChainsCurrentImage() {
  if (multi()) {
    for (auto& rs : ranks_) check(rtpt_end_frame(rs.ctx), "copyImageToSwapChainsCurrentImage");
    if (opt_.present) presentFrame();
    const int next = static_cast<int>((frameCount + 1) % static_cast<uint32_t>(opt_.frames_in_flight));
    for (auto& rs : ranks_) {
      rs.last = rs.ctx;
      rs.ctx = rs.ctxs[next];
    }
    stream_ = streams_[next];
    return;
  }
  check(rtpt_end_frame(ctx_), "copyImageToSwapChainsCurrentImage");  // history hand-over, :1364-1372
  last_ = ctx_;
  ctx_ = ctxs_[(frameCount + 1) % static_cast<uint32_t>(opt_.frames_in_flight)];
  if (opt_.present == 1) presentFrame();  // f32 on one context: the frame is already whole where it is
}

// main.cpp:1338-1361: `image` is blitted to the acquired swapchain image.  Here: rtpt_present converts the rows a
// context owns into its swapchain image (two images, used alternately like a two-image swapchain); with several ranks
// the presenting rank (0) then receives every other strip — in swapchain format, or as float rows.  The messages run on
// the present stream behind this frame's kernels, so the next frame's passes do not wait for the wire; before an image
// (or, for f32, the strip buffer that was sent) is written again two frames later, the frame stream waits for that
// gather — vkAcquireNextImageKHR's role (main.cpp:1310-1316).
void PathTracingApplication::presentFrame() {
  const size_t W = opt_.width, H = opt_.height;
  const int idx = static_cast<int>(frameCount & 1);
  if (!multi()) {
    // returns at once when the final pass already wrote the image (armPresent)
    check(rtpt_present(last_, presentImage(nullptr, idx), 0, opt_.height), "rtpt_present");
    return;
  }
  const bool rgba8 = opt_.present == 1;
  const size_t px_bytes = rgba8 ? 4 : 16, row_bytes = W * px_bytes;
  // (the image and, for f32, the strip buffers the others send from were acquired at the top of the frame: acquirePresent)
  std::vector<const char*> mine(ranks_.size());
  for (size_t i = 0; i < ranks_.size(); i++) {
    RankState& rs = ranks_[i];
    const Rows own = rs.plan.own();
    const bool root = rs.plan.rank == 0;
    if (rgba8) {
      char* dst = static_cast<char*>(presentImage(&rs, idx)) + static_cast<size_t>(own.first) * row_bytes;
      check(rtpt_present(rs.ctx, dst, static_cast<uint32_t>(own.first), static_cast<uint32_t>(own.second)), "rtpt_present");
      mine[i] = dst;
    } else {
      void* prev = nullptr;
      check(rtpt_plane_ptr(rs.ctx, RTPT_PLANE_PREVIOUS, &prev), "rtpt_plane_ptr");
      mine[i] = static_cast<const char*>(prev) + static_cast<size_t>(own.first - rs.plan.stored().first) * row_bytes;
      if (root)
        host_device_copy(static_cast<char*>(presentImage(&rs, idx)) + static_cast<size_t>(own.first) * row_bytes, mine[i],
                         static_cast<size_t>(own.second - own.first) * row_bytes, stream_);
    }
  }
  host_stream_wait_stream(presentStream_, stream_);
  presentTransport_->begin(presentStream_);
  for (size_t i = 0; i < ranks_.size(); i++) {
    RankState& rs = ranks_[i];
    const Rows own = rs.plan.own();
    if (rs.plan.rank == 0) {
      for (int r = 1; r < opt_.ranks; r++) {
        const Rows o = StripPlan::bounds(static_cast<int>(H), opt_.ranks, r, opt_.splits);
        presentTransport_->recv(0, static_cast<char*>(rs.swap[idx]) + static_cast<size_t>(o.first) * row_bytes, r,
                                static_cast<size_t>(o.second - o.first) * row_bytes);
      }
    } else {
      presentTransport_->send(rs.plan.rank, mine[i], 0, static_cast<size_t>(own.second - own.first) * row_bytes);
    }
  }
  presentTransport_->end();
  if (!presentDone_[idx]) presentDone_[idx] = host_event_create();
  host_event_record(presentDone_[idx], presentStream_);
}

std::vector<unsigned char> PathTracingApplication::readPresented() {
  std::vector<unsigned char> out;
  if (!opt_.present || frameCount == 0) return out;
  const int idx = static_cast<int>((frameCount - 1) & 1);
  const size_t W = opt_.width, H = opt_.height;
  if (!multi()) {
    if (opt_.present != 1 || !swapSingle_[idx]) return out;
    out.resize(W * H * 4);
    check(rtpt_sync(last_), "rtpt_sync");
    host_device_to_host(out.data(), swapSingle_[idx], out.size(), nullptr);
    return out;
  }
  for (auto& rs : ranks_)
    if (rs.plan.rank == 0 && rs.swap[idx]) {
      out.resize(W * H * (opt_.present == 1 ? 4 : 16));
      for (void* st : streams_)
        if (st) host_stream_sync(st);
      host_stream_sync(presentStream_);
      host_device_to_host(out.data(), rs.swap[idx], out.size(), presentStream_);
    }
  return out;
}

void PathTracingApplication::drawScene(const std::string& keys) {
  acquirePresent();
  updateScene(keys);
  drawVisbilityBuffer();
  computeTemporalGradient();
  drawSceneToImage();
  applyTemporalFiltering();
  copyImageToSwapChainsCurrentImage();
  frameCount++;  // :1111
}

void PathTracingApplication::freeRessources() {
  for (auto& rs : ranks_) {
    for (rtpt_ctx*& c : rs.ctxs) {
      if (c) rtpt_destroy(c);
      c = nullptr;
    }
    rs.ctx = rs.last = nullptr;
    if (rs.history) host_device_free(rs.history);
    if (rs.guide_ids) host_device_free(rs.guide_ids);
    if (rs.guide_moments) host_device_free(rs.guide_moments);
    for (void*& p : rs.swap) {
      if (p) host_device_free(p);
      p = nullptr;
    }
  }
  for (void*& p : swapSingle_) {
    if (p) host_device_free(p);
    p = nullptr;
  }
  for (void*& e : presentDone_) {
    if (e) host_event_destroy(e);
    e = nullptr;
  }
  if (!ranks_.empty()) {
    ranks_.clear();
    ctxs_[0] = ctxs_[1] = nullptr;
    delete transport_;
    transport_ = nullptr;
    delete presentTransport_;
    presentTransport_ = nullptr;
    for (void*& st : streams_) {
      if (st) host_stream_destroy(st);
      st = nullptr;
    }
    if (presentStream_) host_stream_destroy(presentStream_);
    stream_ = presentStream_ = nullptr;
    ctx_ = last_ = nullptr;
    return;
  }
  for (auto& c : ctxs_) {
    if (c) rtpt_destroy(c);
    c = nullptr;
  }
  ctx_ = last_ = nullptr;
}

std::string PathTracingApplication::planJson(int frames, const std::vector<std::string>& script) {
  loadMesh();
  sceneBounds();
  // uploadBuffers (main.cpp:481-489) + initializeSceneConstants, as in initVulkan, without any device work
  identity(ubo.model);
  const float center[3] = {0.0f, 1.0f, 0.0f}, up[3] = {0.0f, 1.0f, 0.0f};
  rtpt_util_look_at(cameraOrigin, center, up, ubo.view);
  rtpt_util_perspective(kFov * 2, static_cast<float>(opt_.width) / static_cast<float>(opt_.height), 0.1f, zFar_, ubo.proj);
  ubo.proj[5] *= -1;
  std::memcpy(ubo.modelPrev, ubo.model, sizeof ubo.model);
  std::memcpy(ubo.viewPrev, ubo.view, sizeof ubo.view);
  std::memcpy(ubo.projPrev, ubo.proj, sizeof ubo.proj);
  initializeSceneConstants();
  const int H = static_cast<int>(opt_.height), R = opt_.ranks, N = opt_.maxWaveletIteration;
  std::string out = "{\"ranks\": [";
  auto rows = [](Rows r) { return "[" + std::to_string(r.first) + ", " + std::to_string(r.second) + "]"; };
  for (int r = 0; r < R; r++) {
    StripPlan p;
    p.height = H; p.world = R; p.rank = r; p.iterations = N; p.exchange = opt_.exchange_halo; p.ext_flags = opt_.flags & 0x9F0u; p.splits = opt_.splits;
    p.validate();
    out += std::string(r ? ", " : "") + "{\"own\": " + rows(p.own()) + ", \"stored\": " + rows(p.stored()) + ", \"raytrace\": " +
           rows(p.raytrace_rows()) + ", \"filter\": [";
    for (int k = 1; k <= N; k++) out += std::string(k > 1 ? ", " : "") + rows(p.filter_rows(k));
    out += "]}";
  }
  out += "], \"splits\": [";
  for (int r = 0; r <= R; r++) out += std::string(r ? ", " : "") + std::to_string(r < R ? StripPlan::bounds(H, R, r, opt_.splits).first : H);
  out += "]";
  if (!opt_.balance_cost.empty()) {
    std::vector<int> cur;
    for (int r = 0; r <= R; r++) cur.push_back(r < R ? StripPlan::bounds(H, R, r, opt_.splits).first : H);
    StripPlan p0;
    p0.height = H; p0.world = R; p0.iterations = N; p0.exchange = opt_.exchange_halo; p0.ext_flags = opt_.flags & 0x9F0u;
    const std::vector<int> b = balanced_splits(cur, opt_.balance_cost, std::max(1, p0.halo()));
    out += ", \"balanced\": [";
    for (size_t i = 0; i < b.size(); i++) out += std::string(i ? ", " : "") + std::to_string(b[i]);
    out += "]";
  }
  out += ", \"frames\": [";
  for (int f = 0; f < frames; f++) {
    updateScene(static_cast<size_t>(f) < script.size() ? script[static_cast<size_t>(f)] : std::string());
    out += std::string(f ? ", " : "") + "{\"moved\": " + (frameCount > 0 && !cameraStatic() ? "true" : "false") + ", \"needs\": [";
    for (int r = 0; r < R; r++)
      out += std::string(r ? ", " : "") + rows(reprojection_rows(ubo, static_cast<int>(opt_.width), H, StripPlan::bounds(H, R, r, opt_.splits), sceneMin_, sceneMax_, 0.1f));
    out += "]}";
    frameCount++;
  }
  out += "]}";
  return out;
}

uint64_t PathTracingApplication::bytesSent() const {
  return (transport_ ? transport_->bytes_sent() : 0) + (presentTransport_ ? presentTransport_->bytes_sent() : 0);
}

void PathTracingApplication::sync() {
  if (multi()) {
    for (auto& rs : ranks_)
      for (rtpt_ctx* c : rs.ctxs)
        if (c) check(rtpt_sync(c), "rtpt_sync");
    if (presentStream_) host_stream_sync(presentStream_);
    return;
  }
  for (int i = 0; i < opt_.frames_in_flight; i++) check(rtpt_sync(ctxs_[i]), "rtpt_sync");
}

std::vector<float> PathTracingApplication::readImage() {
  std::vector<float> img(static_cast<size_t>(opt_.width) * opt_.height * 4);
  if (multi()) {
    // the rows this process owns (all of them in local mode; one strip per process with RCCL, the rest stays 0)
    const size_t row_floats = static_cast<size_t>(opt_.width) * 4;
    for (auto& rs : ranks_) {
      const Rows st = rs.plan.stored(), own = rs.plan.own();
      std::vector<float> strip(static_cast<size_t>(st.second - st.first) * row_floats);
      check(rtpt_readback(rs.last, RTPT_PLANE_PREVIOUS, strip.data(), strip.size() * sizeof(float)), "rtpt_readback");
      std::memcpy(img.data() + static_cast<size_t>(own.first) * row_floats, strip.data() + static_cast<size_t>(own.first - st.first) * row_floats,
                  static_cast<size_t>(own.second - own.first) * row_floats * sizeof(float));
    }
    return img;
  }
  // after rtpt_end_frame IMAGE and PREVIOUS hold the same pixels (main.cpp:1364)
  check(rtpt_readback(last_, RTPT_PLANE_PREVIOUS, img.data(), img.size() * sizeof(float)), "rtpt_readback");
  return img;
}

uint64_t PathTracingApplication::rayCount() {
  uint64_t total = 0;
  if (multi()) {
    for (auto& rs : ranks_)
      for (rtpt_ctx* c : rs.ctxs)
        if (c) {
          uint64_t n = 0;
          check(rtpt_readback(c, RTPT_PLANE_RAYCOUNT, &n, sizeof n), "rtpt_readback");
          total += n;
        }
    return total;
  }
  for (int i = 0; i < opt_.frames_in_flight; i++) {
    uint64_t n = 0;
    check(rtpt_readback(ctxs_[i], RTPT_PLANE_RAYCOUNT, &n, sizeof n), "rtpt_readback");
    total += n;
  }
  return total;
}

void PathTracingApplication::writePFM(const std::string& path) {
  std::vector<float> img = readImage();
  std::ofstream f(path, std::ios::binary);
  if (!f) throw std::runtime_error("cannot write " + path);
  f << "PF\n" << opt_.width << " " << opt_.height << "\n-1.0\n";
  std::vector<float> row(static_cast<size_t>(opt_.width) * 3);
  for (uint32_t y = 0; y < opt_.height; y++) {
    const float* src = img.data() + static_cast<size_t>(opt_.height - 1 - y) * opt_.width * 4;
    for (uint32_t x = 0; x < opt_.width; x++) {
      row[3 * x] = src[4 * x];
      row[3 * x + 1] = src[4 * x + 1];
      row[3 * x + 2] = src[4 * x + 2];
    }
    f.write(reinterpret_cast<const char*>(row.data()), static_cast<std::streamsize>(row.size() * sizeof(float)));
  }
}

}  // namespace rtpt_host
