// rgbid_track_sequence -- chunk-sharded tracking of ONE recorded sequence on 1..N GPUs (BASELINE config 4): the C++ host
// of the batched path.  What the reference's eval loop does for one tracker on one GPU (tools/RGBID_SLAMapp.cpp:360-433 playback + CLI,
// tools/evaluation.cpp:380-439 trajectory file), done for a sequence cut into chunks: dataset reader (host/tum_io.cpp) ->
// rgbid_dist_track_sequence (csrc/dist.cpp: partition -> one engine lane per chunk -> uploads behind the previous step -> lock-step engine
// steps -> device-side record pack -> ONE all-gather over RCCL -> composition) -> "<stamp> tx ty tz qx qy qz qw" per frame.
//
//   rgbid_track_sequence -eval <dataset_folder/> [-match_file <f>] [-chunks 8] [-out trajectory.txt] [-max_frames N] [-rows 480] [-cols 640]
//                        [-K fx fy cx cy] [-gpu <id>] [-fast 0|1] [-fused 0|1] [-report <json file>]
//                        [-world W -rank R -master_addr A -master_port P] [-exchange rccl|tcp]
//   one process per GPU; -world / -rank / -gpu / -master_* default to WORLD_SIZE / RANK / LOCAL_RANK / MASTER_ADDR / MASTER_PORT + 1 (a launcher's
//   own store listens on MASTER_PORT), e.g.  python -m torch.distributed.run --no-python --nproc-per-node 8 ... rgbid_track_sequence -eval ...
//   -inject <file> [-frames F]: take the per-chunk records from <file> ([chunks][chunk_len] x 392 bytes) instead of running the engine --
//   partition / exchange (tcp) / composition / file output without a GPU (tests/test_dist_cpu.py).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <string>
#include <vector>

#include "../../include/rgbid/evaluation.h"
#include "../../include/rgbid_dist.h"
#include "../../include/rgbid_host.h"

using namespace RGBID_SLAM;

static bool arg_value(int argc, char** argv, const char* key, std::string& val) {
  for (int i = 1; i + 1 < argc; ++i) if (!std::strcmp(argv[i], key)) { val = argv[i + 1]; return true; }
  return false;
}
static int arg_int(int argc, char** argv, const char* key, int dflt) { std::string s; return arg_value(argc, argv, key, s) ? std::atoi(s.c_str()) : dflt; }
static int env_int(const char* key, int dflt) { const char* e = std::getenv(key); return e ? std::atoi(e) : dflt; }

int main(int argc, char* argv[]) {
  // before the HIP runtime comes up: the host driver only supports dmabuf IPC (RCCL across processes)
  setenv("HSA_ENABLE_IPC_MODE_LEGACY", "0", 0);
  if (rgbid_engine_config_size() != sizeof(rgbid_engine_config)) {   // this binary and librgbid_hip.so come from different revisions of rgbid_engine.h
    fprintf(stderr, "rgbid_track_sequence: built against a %zu-byte rgbid_engine_config, the library has %zu: rebuild\n", sizeof(rgbid_engine_config), rgbid_engine_config_size());
    return 2;
  }
  std::string folder, match_file, out = "trajectory.txt", inject_file, report_file, exchange = "rccl", addr, s;
  const bool have_eval = arg_value(argc, argv, "-eval", folder);
  arg_value(argc, argv, "-match_file", match_file);
  arg_value(argc, argv, "-out", out);
  arg_value(argc, argv, "-inject", inject_file);
  arg_value(argc, argv, "-report", report_file);
  arg_value(argc, argv, "-exchange", exchange);
  const int chunks = arg_int(argc, argv, "-chunks", 8), max_frames = arg_int(argc, argv, "-max_frames", -1);
  const int rows = arg_int(argc, argv, "-rows", 480), cols = arg_int(argc, argv, "-cols", 640);
  const int world = arg_int(argc, argv, "-world", env_int("WORLD_SIZE", 1)), rank = arg_int(argc, argv, "-rank", env_int("RANK", 0));
  const int gpu = arg_int(argc, argv, "-gpu", env_int("LOCAL_RANK", 0));
  if (!arg_value(argc, argv, "-master_addr", addr)) { const char* e = std::getenv("MASTER_ADDR"); addr = e ? e : "127.0.0.1"; }
  int port = arg_int(argc, argv, "-master_port", 0);
  if (port <= 0) port = env_int("MASTER_PORT", 29540) + 1;
  float K[4] = {Evaluation::fx, Evaluation::fy, Evaluation::cx, Evaluation::cy};   // tools/evaluation.cpp:64-67
  for (int i = 1; i + 4 < argc; ++i) if (!std::strcmp(argv[i], "-K")) for (int k = 0; k < 4; ++k) K[k] = (float)std::atof(argv[i + 1 + k]);
  if ((!have_eval && inject_file.empty()) || chunks < 1 || world < 1 || rank < 0 || rank >= world || (exchange != "rccl" && exchange != "tcp")) {
    std::cout << "usage: rgbid_track_sequence -eval <dataset_folder/> [-match_file f] [-chunks N] [-out file] [-max_frames n] [-rows r -cols c] [-K fx fy cx cy]\n"
                 "                            [-gpu id] [-fast 0|1] [-fused 0|1] [-report json] [-world W -rank R -master_addr A -master_port P] [-exchange rccl|tcp]\n"
                 "                            [-inject records.bin [-frames F]]\n";
    return 2;
  }

  // ---- frames (the product's dataset reader: association files, 16-bit PNG depth x 0.2 -> mm, BGR -> RGB) into pinned host memory
  std::vector<double> stamps;
  uint16_t* depth = nullptr;
  uint8_t* rgb = nullptr;
  int T = 0;
  const size_t fd = (size_t)rows * cols * 2, fc = (size_t)rows * cols * 3;
  if (have_eval && inject_file.empty()) {
    Evaluation::Ptr ev;
    try { ev.reset(new Evaluation(folder, match_file)); }
    catch (const std::exception& e) { std::cerr << "cannot open dataset: " << e.what() << std::endl; return 1; }
    const int n = max_frames < 0 ? (int)ev->size() : std::min((int)ev->size(), max_frames);
    void *pd = nullptr, *pc = nullptr;
    if (rgbid_malloc_host(&pd, fd * n) != 0 || rgbid_malloc_host(&pc, fc * n) != 0) { std::cerr << "cannot allocate " << (fd + fc) * n << " bytes of pinned host memory\n"; return 1; }
    depth = (uint16_t*)pd; rgb = (uint8_t*)pc;
    ImageWrapper<unsigned short> dw; ImageWrapper<PixelRGB> cw;
    for (int k = 0; k < n; ++k) {
      bool ok = false;
      try { ok = ev->grab(k, dw, cw); } catch (const std::exception& e) { std::cerr << e.what() << std::endl; return 1; }
      if (!ok) continue;                       // unreadable pair: the reference's playback skips it too
      if (dw.rows != rows || dw.cols != cols || cw.rows != rows || cw.cols != cols) { std::cerr << "frame " << k << " is " << dw.cols << "x" << dw.rows << ", expected " << cols << "x" << rows << std::endl; return 1; }
      std::memcpy((char*)depth + fd * T, dw.data, fd);
      std::memcpy(rgb + fc * T, cw.data, fc);
      stamps.push_back(ev->stamp(k));
      ++T;
    }
  } else {
    T = arg_int(argc, argv, "-frames", 0);
    for (int k = 0; k < T; ++k) stamps.push_back(k / 30.0);
  }
  if (T < chunks + 1) { std::cerr << "need at least chunks + 1 readable frames (" << T << " frames, " << chunks << " chunks)\n"; return 1; }

  rgbid_seq_config cfg;
  std::memset(&cfg, 0, sizeof(cfg));
  rgbid_engine_default_config(&cfg.engine);
  cfg.engine.rows = rows; cfg.engine.cols = cols;
  cfg.engine.fx = K[0]; cfg.engine.fy = K[1]; cfg.engine.cx = K[2]; cfg.engine.cy = K[3];
  cfg.engine.fast_numerics = arg_int(argc, argv, "-fast", cfg.engine.fast_numerics);
  cfg.engine.fused_gn = arg_int(argc, argv, "-fused", cfg.engine.fused_gn);
  cfg.n_chunks = chunks; cfg.world = world; cfg.rank = rank;
  cfg.exchange = exchange == "tcp" ? RGBID_EXCHANGE_TCP : RGBID_EXCHANGE_RCCL;
  cfg.master_addr = addr.c_str(); cfg.master_port = port;

  std::vector<rgbid_gather_record> inject;
  int inject_chunk_len = 0;
  if (!inject_file.empty()) {
    std::ifstream f(inject_file.c_str(), std::ios::binary);
    if (!f) { std::cerr << "cannot open " << inject_file << std::endl; return 1; }
    f.seekg(0, std::ios::end); const size_t bytes = (size_t)f.tellg(); f.seekg(0);
    if (bytes == 0 || bytes % (sizeof(rgbid_gather_record) * chunks) != 0) { std::cerr << inject_file << ": not [chunks][chunk_len] records\n"; return 1; }
    inject.resize(bytes / sizeof(rgbid_gather_record));
    f.read((char*)inject.data(), (std::streamsize)bytes);
    inject_chunk_len = (int)(inject.size() / (size_t)chunks);   // the driver refuses it unless it is the chunk length -frames / -chunks imply
  }
  cfg.inject_chunk_len = inject_chunk_len;
  cfg.warmup_frames = arg_int(argc, argv, "-warmup", 0);   // frames every chunk but the first tracks before its own first frame (velocity prior, settled keyframe)
  rgbid_ctx* ctx = nullptr;
  if (inject.empty()) {
    int e = rgbid_ctx_create(&ctx, gpu, nullptr);
    if (e) { std::cerr << "no usable HIP device " << gpu << ": " << rgbid_error_string(e) << " (there is no CPU path)\n"; return 1; }
  }
  std::vector<double> R((size_t)T * 9), t((size_t)T * 3);
  std::vector<int> status(T);
  rgbid_seq_report rep;
  std::memset(&rep, 0, sizeof(rep));
  const int e = rgbid_dist_track_sequence(ctx, &cfg, depth, rgb, T, inject.empty() ? nullptr : inject.data(), R.data(), t.data(), status.data(), nullptr, &rep);
  if (ctx) rgbid_ctx_destroy(ctx);
  if (depth) rgbid_free_host(depth);
  if (rgb) rgbid_free_host(rgb);
  if (e) {
    std::cerr << "rank " << rank << ": rgbid_dist_track_sequence failed with " << e << (e <= RGBID_E_RCCL ? " (RCCL)" : e == RGBID_E_NET ? " (rendezvous)" : "") << std::endl;
    return 1;
  }
  if (rank == 0) {
    std::ofstream f(out.c_str());
    for (int k = 0; k < T; ++k) f << format_pose_line(stamps[k], &R[(size_t)k * 9], &t[(size_t)k * 3]) << "\n";   // tools/evaluation.cpp:380-439
    f.close();
    char line[1024];
    std::snprintf(line, sizeof(line),
                  "{\"frames\": %d, \"chunks\": %d, \"world\": %d, \"rccl_ranks\": %d, \"lanes_per_gpu\": %d, \"chunk_len\": %d, \"setup_ms\": %.3f, \"track_ms\": %.3f, "
                  "\"gather_ms\": %.3f, \"compose_ms\": %.3f, \"total_ms\": %.3f, \"frames_per_s\": %.1f, \"staged_bytes\": %llu, \"engine_bytes\": %llu, \"out\": \"%s\"}",
                  T, chunks, world, rep.rccl_ranks, rep.lanes, rep.chunk_len, rep.setup_ms, rep.track_ms, rep.gather_ms, rep.compose_ms, rep.total_ms,
                  rep.total_ms > 0 ? 1e3 * T / rep.total_ms : 0.0, rep.staged_bytes, rep.engine_bytes, out.c_str());
    std::cout << line << std::endl;
    if (!report_file.empty()) { std::ofstream rf(report_file.c_str()); rf << line << "\n"; }
  }
  return 0;
}
