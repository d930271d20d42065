// main.cpp — CLI of the headless host (the reference's main(), main.cpp:1532-1537, plus the knobs the
// reference keeps as compile-time constants).
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include <unistd.h>

#include "app.hpp"

static void usage(const char* argv0) {
  std::printf(
      "usage: %s [--width W] [--height H] [--frames N] [--segments S] [--iterations K]\n"
      "          [--scene file.obj] [--script \"keys0,keys1,...\"] [--dump out.pfm] [--exact-filter]\n"
      "          [--tessellate n] [--lattice NXxNYxNZ [--pitch P] | --instances file] [--dump-scene out.bin]\n"
      "          [--frames-in-flight 1|2]\n"
      "          [--ranks R [--rank r --rccl-id-file F [--rccl-nonce N] [--rccl-timeout S]] [--halo redundant|exchange] [--splits 0,a,b,..,H] [--device D]]\n"
      "          [--present none|rgba8|f32 [--dump-present out.raw]]\n"
      "          [--plan-only   (print the strip plan and the history bands of the scripted frames as JSON; needs no GPU)]\n"
      "  --ranks R splits the frame into R row strips: with --rank r this process is rank r on its own GPU and talks RCCL\n"
      "  (start R processes; rank 0 publishes the ncclUniqueId in F); without --rank all R strips run in this process\n"
      "          [--flags N   (RTPT_FLAG_* bits of include/rtpt.h, e.g. 0xF0 = all extension modes)]\n"
      "  --rccl-nonce N: any number, the same on every rank of one launch and different between launches (default: the\n"
      "  launcher's pid); an id file of another launch is ignored; the communicator bring-up gives up after S seconds (120)\n"
      "  --present: the swapchain blit of every frame (main.cpp:1338-1361): rgba8 converts to B8G8R8A8_UNORM, and with --ranks the\n"
      "  strips are gathered on rank 0 (rgba8: in that format, f32: as float rows); --dump-present writes rank 0's last image raw\n"
      "  --tessellate n splits every quad of the OBJ into n x n cells; --lattice instances the mesh on a lattice of translations and\n"
      "  frames it (camera, light, far plane): --lattice 10x10x10 --tessellate 6 is BASELINE.json configs[4], 1,152,000 triangles;\n"
      "  --instances file: 12 floats per instance (3x4 row-major transforms) instead of the one identity instance;\n"
      "  --dump-scene writes what rtpt_scene_upload receives (with --plan-only: no GPU needed)\n"
      "  keys per frame are the reference's GLFW keys: WASDQE move the camera, IJKLUO the light\n"
      "  defaults are the reference's constants: 1000x800, 32 segments, 9 iterations (main.cpp:52-55)\n",
      argv0);
}

int main(int argc, char** argv) {
  rtpt_host::Options opt;
  int frames = 3;
  bool plan_only = false;
  std::string dump, dump_present, dump_scene, script_arg;
  opt.rccl_nonce = static_cast<uint64_t>(::getppid());
  // scene path relative to this binary: <pkg>/scenes/...
  std::string self(argv[0]);
  size_t slash = self.find_last_of('/');
  opt.scene = (slash == std::string::npos ? std::string(".") : self.substr(0, slash)) + "/scenes/CornellBox-Original-Merged.obj";
  for (int i = 1; i < argc; i++) {
    auto need = [&](const char* name) -> const char* {
      if (i + 1 >= argc) {
        std::fprintf(stderr, "%s needs a value\n", name);
        std::exit(2);
      }
      return argv[++i];
    };
    if (!std::strcmp(argv[i], "--width")) opt.width = static_cast<uint32_t>(std::atoi(need("--width")));
    else if (!std::strcmp(argv[i], "--height")) opt.height = static_cast<uint32_t>(std::atoi(need("--height")));
    else if (!std::strcmp(argv[i], "--frames")) frames = std::atoi(need("--frames"));
    else if (!std::strcmp(argv[i], "--segments")) opt.max_segments = static_cast<uint32_t>(std::atoi(need("--segments")));
    else if (!std::strcmp(argv[i], "--iterations")) opt.maxWaveletIteration = std::atoi(need("--iterations"));
    else if (!std::strcmp(argv[i], "--scene")) opt.scene = need("--scene");
    else if (!std::strcmp(argv[i], "--script")) script_arg = need("--script");
    else if (!std::strcmp(argv[i], "--tessellate")) opt.tessellate = std::atoi(need("--tessellate"));
    else if (!std::strcmp(argv[i], "--pitch")) opt.pitch = static_cast<float>(std::atof(need("--pitch")));
    else if (!std::strcmp(argv[i], "--dump-scene")) dump_scene = need("--dump-scene");
    else if (!std::strcmp(argv[i], "--instances")) opt.instances = need("--instances");
    else if (!std::strcmp(argv[i], "--lattice")) {
      if (std::sscanf(need("--lattice"), "%dx%dx%d", &opt.lattice[0], &opt.lattice[1], &opt.lattice[2]) != 3 || opt.lattice[0] < 1 ||
          opt.lattice[1] < 1 || opt.lattice[2] < 1) {
        std::fprintf(stderr, "--lattice takes NXxNYxNZ, e.g. 10x10x10\n");
        return 2;
      }
    }
    else if (!std::strcmp(argv[i], "--dump")) dump = need("--dump");
    else if (!std::strcmp(argv[i], "--exact-filter")) opt.flags |= RTPT_FLAG_EXACT_FILTER;
    else if (!std::strcmp(argv[i], "--frames-in-flight")) opt.frames_in_flight = std::atoi(need("--frames-in-flight"));
    else if (!std::strcmp(argv[i], "--ranks")) opt.ranks = std::atoi(need("--ranks"));
    else if (!std::strcmp(argv[i], "--rank")) opt.rank = std::atoi(need("--rank"));
    else if (!std::strcmp(argv[i], "--halo")) opt.exchange_halo = !std::strcmp(need("--halo"), "exchange");
    else if (!std::strcmp(argv[i], "--splits") || !std::strcmp(argv[i], "--balance")) {
      // --splits 0,300,...,H: unequal strips; --balance t0,t1,...: (with --plan-only) the ranks' frame times, the plan then
      // also prints the boundaries balanced_splits derives from them
      const bool sp = !std::strcmp(argv[i], "--splits");
      const char* v = need(argv[i]);
      for (const char* q = v; *q;) {
        char* e = nullptr;
        const double x = std::strtod(q, &e);
        if (e == q) { std::fprintf(stderr, "%s takes comma-separated numbers\n", sp ? "--splits" : "--balance"); return 2; }
        if (sp) opt.splits.push_back(static_cast<int>(x)); else opt.balance_cost.push_back(x);
        q = *e == ',' ? e + 1 : e;
      }
    }
    else if (!std::strcmp(argv[i], "--rccl-id-file")) opt.rccl_id_file = need("--rccl-id-file");
    else if (!std::strcmp(argv[i], "--device")) opt.device = std::atoi(need("--device"));
    else if (!std::strcmp(argv[i], "--rccl-nonce")) opt.rccl_nonce = std::strtoull(need("--rccl-nonce"), nullptr, 0);
    else if (!std::strcmp(argv[i], "--rccl-timeout")) opt.rccl_timeout_s = std::atoi(need("--rccl-timeout"));
    else if (!std::strcmp(argv[i], "--dump-present")) dump_present = need("--dump-present");
    else if (!std::strcmp(argv[i], "--present")) {
      const char* v = need("--present");
      opt.present = !std::strcmp(v, "rgba8") ? 1 : !std::strcmp(v, "f32") ? 2 : !std::strcmp(v, "none") ? 0 : -1;
      if (opt.present < 0) { std::fprintf(stderr, "--present takes none, rgba8 or f32\n"); return 2; }
    }
    else if (!std::strcmp(argv[i], "--plan-only")) plan_only = true;
    else if (!std::strcmp(argv[i], "--flags") && i + 1 < argc) opt.flags |= static_cast<uint32_t>(std::strtoul(argv[++i], nullptr, 0));
    else if (!std::strcmp(argv[i], "--help") || !std::strcmp(argv[i], "-h")) { usage(argv[0]); return 0; }
    else { std::fprintf(stderr, "unknown option %s\n", argv[i]); usage(argv[0]); return 2; }
  }
  std::vector<std::string> script;
  {
    std::stringstream ss(script_arg);
    std::string item;
    while (std::getline(ss, item, ',')) script.push_back(item);
  }
  try {
    rtpt_host::PathTracingApplication app(opt);
    if (plan_only) {  // host-only: no GPU is touched
      if (!dump_scene.empty()) {  // the scene as uploaded, before any scripted key moves the camera
        app.loadMesh();
        app.dumpScene(dump_scene);
      }
      std::printf("%s\n", app.planJson(frames, script).c_str());
      return 0;
    }
    app.initVulkan();
    if (!dump_scene.empty()) app.dumpScene(dump_scene);
    app.sync();
    auto t0 = std::chrono::steady_clock::now();
    for (int f = 0; f < frames; f++) app.drawScene(static_cast<size_t>(f) < script.size() ? script[static_cast<size_t>(f)] : "");
    app.sync();
    double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    uint64_t rays = app.rayCount();
    std::printf("{\"frames\": %d, \"width\": %u, \"height\": %u, \"ranks\": %d, \"rank\": %d, \"ms_per_frame\": %.4f, \"rays\": %llu, "
                "\"mray_per_s\": %.1f, \"bytes_sent\": %llu}\n",
                frames, opt.width, opt.height, opt.ranks, opt.rank, ms / frames, static_cast<unsigned long long>(rays),
                rays / (ms * 1e-3) / 1e6, static_cast<unsigned long long>(app.bytesSent()));
    if (!dump.empty()) app.writePFM(dump);
    if (!dump_present.empty()) {
      const std::vector<unsigned char> img = app.readPresented();
      if (!img.empty()) {  // rank 0's process (or the only one)
        FILE* f = std::fopen(dump_present.c_str(), "wb");
        if (!f || std::fwrite(img.data(), 1, img.size(), f) != img.size()) throw std::runtime_error("cannot write " + dump_present);
        std::fclose(f);
      }
    }
  } catch (const std::exception& e) {
    std::fprintf(stderr, "rtpt_app: %s\n", e.what());
    return 1;
  }
  return 0;
}
