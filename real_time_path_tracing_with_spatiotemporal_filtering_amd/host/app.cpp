// app.cpp — see app.hpp.  Every method cites the reference lines it stands for.
#include "app.hpp"

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
  rtpt_util_perspective(kFov * 2, static_cast<float>(opt_.width) / static_cast<float>(opt_.height), 0.1f, 10.0f, ubo.proj);
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
}

void PathTracingApplication::createBuffers() {
  rtpt_config cfg;
  check(rtpt_config_default(&cfg, opt_.width, opt_.height), "rtpt_config_default");
  cfg.max_segments = opt_.max_segments;
  cfg.flags = opt_.flags;
  if (opt_.frames_in_flight != 1 && opt_.frames_in_flight != 2) throw std::runtime_error("frames_in_flight must be 1 or 2");
  for (int i = 0; i < opt_.frames_in_flight; i++) check(rtpt_create(&cfg, &ctxs_[i]), "createBuffers");
  ctx_ = last_ = ctxs_[0];
}

void PathTracingApplication::buildAccelerationStructure() {
  for (int i = 0; i < opt_.frames_in_flight; i++)
    check(rtpt_scene_upload(ctxs_[i], objVertices.data(), static_cast<uint32_t>(objVertices.size() / 3), objIndices.data(),
                            static_cast<uint32_t>(objIndices.size() / 3), nullptr, 0),
          "buildAccelerationStructure");
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
  rtpt_util_perspective(kFov * 2, static_cast<float>(opt_.width) / static_cast<float>(opt_.height), 0.1f, 10.0f, ubo.proj);
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

void PathTracingApplication::drawVisbilityBuffer() { check(rtpt_gbuffer(ctx_, &ubo, 0, 0), "drawVisbilityBuffer"); }

void PathTracingApplication::computeTemporalGradient() {
  check(rtpt_temporal_gradient(ctx_, &pushConstants, 0, 0), "computeTemporalGradient");
}

void PathTracingApplication::drawSceneToImage() {
  pushConstants.sample_batch = 0;  // NUM_SAMPLE_BATCHES = 1, main.cpp:1223,:1237
  check(rtpt_raytrace(ctx_, &pushConstants, 0, 0), "drawSceneToImage");
}

void PathTracingApplication::applyTemporalFiltering() {
  pushConstants.maxWaveletIteration = opt_.maxWaveletIteration;   // :1258
  for (int k = 1; k <= opt_.maxWaveletIteration; k++) {           // :1259
    pushConstants.waveletIteration = k;                           // :1260
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
  check(rtpt_end_frame(ctx_), "copyImageToSwapChainsCurrentImage");  // history hand-over, :1364-1372
  last_ = ctx_;
  ctx_ = ctxs_[(frameCount + 1) % static_cast<uint32_t>(opt_.frames_in_flight)];
}

void PathTracingApplication::drawScene(const std::string& keys) {
  updateScene(keys);
  drawVisbilityBuffer();
  computeTemporalGradient();
  drawSceneToImage();
  applyTemporalFiltering();
  copyImageToSwapChainsCurrentImage();
  frameCount++;  // :1111
}

void PathTracingApplication::freeRessources() {
  for (auto& c : ctxs_) {
    if (c) rtpt_destroy(c);
    c = nullptr;
  }
  ctx_ = last_ = nullptr;
}

void PathTracingApplication::sync() {
  for (int i = 0; i < opt_.frames_in_flight; i++) check(rtpt_sync(ctxs_[i]), "rtpt_sync");
}

std::vector<float> PathTracingApplication::readImage() {
  std::vector<float> img(static_cast<size_t>(opt_.width) * opt_.height * 4);
  // after rtpt_end_frame IMAGE and PREVIOUS hold the same pixels (main.cpp:1364)
  check(rtpt_readback(last_, RTPT_PLANE_PREVIOUS, img.data(), img.size() * sizeof(float)), "rtpt_readback");
  return img;
}

uint64_t PathTracingApplication::rayCount() {
  uint64_t total = 0;
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
