// app.hpp — headless C++ host: the reference's PathTracingApplication (main.cpp:179-1529) with the
// same method names and per-frame order, driving the HIP hot path through the C ABI of rtpt.h.
// What the reference did with Vulkan objects (images, descriptor sets, pipelines, command buffers)
// is one rtpt_* call per dispatch here; window, swapchain and keyboard are replaced by a scripted
// key list and an optional PFM dump (out of scope as code, in scope as a scripted path).
#pragma once

#include <cstdint>
#include <string>
#include <vector>

#include "../../include/rtpt.h"
#include "strips.hpp"

namespace rtpt_host {

struct Options {
  uint32_t width = 1000, height = 800;  // main.cpp:52-53
  int maxWaveletIteration = 9;          // main.cpp:55
  uint32_t max_segments = 32;           // raytrace.comp.glsl:204
  uint32_t flags = 0;
  int frames_in_flight = 1;             // 2: even/odd frames in two contexts on two streams (rtpt_stream_wait)
  std::string scene;                    // scenes/CornellBox-Original-Merged.obj (main.cpp:417)
  // BASELINE.json configs[4] (scene_gen.hpp): every quad of the OBJ tessellated n x n, the mesh instanced on
  // a lattice of translations (the reference's own instance list is one identity transform, main.cpp:728-741); the camera,
  // the light and the far plane then frame the lattice.  --lattice 10x10x10 --tessellate 6 = 1,152,000 triangles
  int tessellate = 1;
  int lattice[3] = {0, 0, 0};           // 0: no instancing (one identity instance)
  float pitch = 2.5f;
  std::string instances;                // text file, 12 floats per instance (3x4 row-major transform): a general instance list
                                        // in place of main.cpp:728-741's identity (camera and light stay the reference's)
  // row strips across GPUs (new work, the reference is single-device)
  int ranks = 1;                        // strips the frame is split into
  int rank = -1;                        // >= 0: this process is that rank (one GPU per process, RCCL between them);
                                        // -1 with ranks > 1: every strip is a context of THIS process on one GPU
                                        // (rehearsal / tests: a message is a device-to-device copy)
  bool exchange_halo = false;           // true: k rows per neighbour travel before iteration k; false: redundant rows
  std::vector<int> splits;              // unequal strips: ranks + 1 ascending rows 0 .. height (empty: equal; strips.hpp)
  std::vector<double> balance_cost;     // --plan-only: the ranks' frame times; the plan also prints balanced_splits of them
  std::string rccl_id_file;             // rendezvous file for the ncclUniqueId (rank >= 0)
  uint64_t rccl_nonce = 0;              // same on every rank of one launch, different between launches (strips.hpp)
  int rccl_timeout_s = 120;             // rendezvous + communicator bring-up watchdog
  // the swapchain blit of main.cpp:1338-1361 inside every frame: 0 none (the frame stays where the final pass left it,
  // strips stay on their ranks), 1 rgba8 (rtpt_present: B8G8R8A8_UNORM; strips gathered on rank 0 in that format),
  // 2 f32 (the float strips gathered on rank 0 as they are)
  int present = 0;
  int device = -1;                      // HIP device of this process (-1: rank % device count, or the current device)
};

class PathTracingApplication {
 public:
  explicit PathTracingApplication(const Options& opt);
  ~PathTracingApplication();
  PathTracingApplication(const PathTracingApplication&) = delete;
  PathTracingApplication& operator=(const PathTracingApplication&) = delete;

  // run(): initWindow (nothing to do headless), initVulkan, mainLoop — main.cpp:185-189
  void run(int frames, const std::vector<std::string>& script);

  void initVulkan();                           // main.cpp:280-300 (device + resources + scene)
  void loadMesh();                             // main.cpp:409-462
  void createBuffers();                        // main.cpp:357-407  -> rtpt_create
  void buildAccelerationStructure();           // main.cpp:687-742  -> rtpt_scene_upload
  void initializeSceneConstants();             // main.cpp:661-666
  void updateScene(const std::string& keys);   // main.cpp:1115-1185, keys = GLFW keys held this frame
  void updateUBO();                            // main.cpp:1463-1475
  void drawVisbilityBuffer();                  // main.cpp:1187-1199 -> rtpt_gbuffer
  void computeTemporalGradient();              // main.cpp:1201-1220 -> rtpt_temporal_gradient
  void drawSceneToImage();                     // main.cpp:1222-1253 -> rtpt_raytrace
  void applyTemporalFiltering();               // main.cpp:1255-1306 -> rtpt_temporal_filter x N
  void copyImageToSwapChainsCurrentImage();    // main.cpp:1308-1406 -> rtpt_end_frame (+ optional dump)
  void drawScene(const std::string& keys);     // main.cpp:1090-1113
  void freeRessources();                       // main.cpp:1477-1528

  // read-back helpers (the reference's only output is the swapchain image)
  std::vector<float> readImage();              // RGBA32F, width*height*4
  // the frame as presented by the last drawScene (Options::present != 0): the swapchain image of rank 0 —
  // width*height*4 bytes B,G,R,A (rgba8) or width*height*16 bytes of floats (f32).  Empty on the other ranks' processes.
  std::vector<unsigned char> readPresented();
  uint64_t rayCount();
  uint64_t bytesSent() const;                  // strips: bytes this process sent (halo rows + history bands)
  // host-only (no GPU, no context): the strip plan of every rank and, per scripted frame, the previous-frame rows each
  // rank's final pass can reach — as one JSON object; what the CPU tests compare with the Python mirror
  std::string planJson(int frames, const std::vector<std::string>& script);
  void writePFM(const std::string& path);      // linear RGB, bottom-up rows as PFM prescribes
  // host-only: the mesh and the instance transforms as rtpt_scene_upload receives them, raw little-endian:
  // u32 n_verts, n_tris, n_instances; float xyz[3 n_verts]; u32 idx[3 n_tris]; float xforms[12 n_instances]; float camera[3],
  // light[3], z_far
  void dumpScene(const std::string& path);
  void sync();

  uint32_t frameCount = 0;                     // main.cpp:259
  rtpt_push_constants pushConstants{};         // main.cpp:51
  rtpt_ubo ubo{};                              // main.cpp:248

 private:
  void check(int rc, const char* what);
  Options opt_;
  // frames_in_flight contexts; frame f is built in ctxs_[f % n].  ctx_ = the context of the frame being built
  // (between frames: of the next frame); last_ = the context holding the last finished frame.
  rtpt_ctx* ctxs_[2] = {nullptr, nullptr};
  rtpt_ctx* ctx_ = nullptr;
  rtpt_ctx* last_ = nullptr;
  // strips: the ranks this process runs (all of them in local mode, one with RCCL), sharing stream_
  struct RankState {
    StripPlan plan;
    rtpt_ctx* ctx = nullptr;                    // the context of the frame being built
    rtpt_ctx* ctxs[2] = {nullptr, nullptr};     // frames_in_flight 2: even / odd frames (ctx alternates between them)
    rtpt_ctx* last = nullptr;                   // the context that finished the previous frame (== ctx with one frame in flight)
    void* history = nullptr;  // full-frame device buffer the previous frame's bands are gathered into (lazily allocated)
    void* guide_ids = nullptr;      // extension modes: the previous frame's id plane ([H, W] u32) ...
    void* guide_moments = nullptr;  // ... and moment plane ([H, W] float4), bands gathered like the history
    void* swap[2] = {nullptr, nullptr};  // swapchain images (whole-frame size; a non-presenting rank fills its own rows)
  };
  void* swapSingle_[2] = {nullptr, nullptr};  // the swapchain images of the one-context host
  Transport* presentTransport_ = nullptr;     // its own communicator + stream: the gather runs behind the frame, beside the next
  void* presentStream_ = nullptr;
  void* presentDone_[2] = {nullptr, nullptr}; // recorded behind the gather that read swapchain image i
  void presentFrame();                        // main.cpp:1338-1361
  void acquirePresent();
  void armPresent();                          // acquire this frame's image, name its rows to the final pass (rtpt_present_target)
  void* presentImage(RankState* rs, int idx); // allocate on first use
  std::vector<RankState> ranks_;
  Transport* transport_ = nullptr;
  void* stream_ = nullptr;                    // the stream of the frame being built
  void* streams_[2] = {nullptr, nullptr};     // strips with two frames in flight: one stream per frame parity, shared by the ranks
  double sceneMin_[3] = {0, 0, 0}, sceneMax_[3] = {0, 0, 0};  // world-space bounds of the POSED, INSTANCED scene
  std::vector<float> instanceXforms_;          // 3x4 row-major per instance; empty: one identity instance (main.cpp:728-741)
  float zFar_ = 10.0f;                         // main.cpp:483
  void sceneBounds();
  bool multi() const { return opt_.ranks > 1; }
  bool cameraStatic() const;
  void exchangeHalo(int k);
  void exchangeHaloPlane(int k, rtpt_plane plane, size_t px_bytes);
  void prepareHistory();
  void prepareGuides();   // RTPT_FLAG_EXT_VARIANCE / _DISOCCLUSION on strips (app.py: _prepare_guides)
  void exchangeBands(const std::vector<Rows>& needs, rtpt_plane plane, size_t px_bytes, void* RankState::*dst);
  std::vector<float> objVertices;              // main.cpp:255
  std::vector<uint32_t> objIndices;            // main.cpp:256
  std::vector<rtpt_material> objMaterials;     // tinyobj's `materials` (main.cpp:419), used when the OBJ has a library
  std::vector<uint32_t> triMaterial;           // material index per triangle
  float cameraOrigin[3] = {-0.001f, 1.0f, 6.0f};  // main.cpp:65
  float lightPos[3] = {1.0f, 1.0f, -0.4f};        // main.cpp:70
  float lightColor[3] = {0.5f, 0.5f, 0.5f};       // main.cpp:72
  bool cameraMoved = false;
};

}  // namespace rtpt_host
