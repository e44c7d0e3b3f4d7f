// C++-level check of the drop-in surface: the reference's container semantics (ThirdParty/pcl_gpu_containers) and a few bridge
// calls exactly as src/visodo.cpp writes them, compiled with plain g++ against include/rgbid/*.h (no HIP header).
//   g++ -std=c++17 -Iinclude tests/cpp/test_bridge.cpp -Lrgbid-slam_amd/lib -lrgbid_host -lrgbid_hip -o test_bridge
// Prints "ok <name>" per check, exits non-zero on the first failure.  A final deliberately failing call checks the error
// convention (print `Error: ...\t<file>:<line>` and exit(0), error.cpp:42-46) when run with the argument "error".
#include <array>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

#include "rgbid/internal.h"
#include "rgbid_host.h"

using namespace RGBID_SLAM;
using namespace RGBID_SLAM::device;
using pcl::gpu::DeviceArray;
using pcl::gpu::DeviceArray2D;

DeviceProp RGBID_SLAM::device::dev_prop;
int RGBID_SLAM::device::dev_id = 0;

#define CHECK(cond, name) do { if (!(cond)) { std::printf("FAILED %s (%s:%d)\n", name, __FILE__, __LINE__); return 1; } std::printf("ok %s\n", name); } while (0)

int main(int argc, char** argv) {
  pcl::gpu::setDevice(0);
  if (rgbid_get_device_prop(0, &dev_prop) != 0) { std::printf("no HIP device\n"); return 2; }
  CHECK(dev_prop.multiProcessorCount == 256, "dev_prop.multiProcessorCount (MI355X: 256 CUs)");

  const int rows = 48, cols = 64;
  std::vector<float> h((size_t)rows * cols);
  for (int y = 0; y < rows; ++y) for (int x = 0; x < cols; ++x) h[(size_t)y * cols + x] = 0.5f * x + 0.25f * y;  // a plane: Sobel/8 = (0.5, 0.25)

  // --- DeviceArray2D: create / upload / download, pitched rows, create() is a no-op for an unchanged shape
  DepthMapf a;
  a.create(rows, cols);
  const void* p0 = a.ptr();
  a.create(rows, cols);
  CHECK(a.ptr() == p0, "create() with unchanged shape keeps the allocation");
  CHECK(a.step() >= (size_t)cols * 4 && a.step() % 256 == 0, "rows are 256-byte aligned");
  a.upload(h.data(), (size_t)cols * 4, rows, cols);
  std::vector<float> back((size_t)rows * cols, -1.f);
  a.download(back.data(), (size_t)cols * 4);
  CHECK(back == h, "upload / download round trip");
  // --- copy shares the buffer (ref-count), copyTo makes a deep copy
  DepthMapf shared = a, deep;
  CHECK(shared.ptr() == a.ptr(), "copy construction shares the buffer");
  a.copyTo(deep);
  CHECK(deep.ptr() != a.ptr() && deep.rows() == rows && deep.cols() == cols, "copyTo allocates a new buffer");
  a.release();
  CHECK(a.empty() && !shared.empty(), "release() drops one reference only");
  shared.download(back.data(), (size_t)cols * 4);
  CHECK(back == h, "the shared buffer survives the release of its sibling");

  // --- bridge calls with the reference's prototypes
  GradientMap gx, gy;
  gx.create(rows, cols); gy.create(rows, cols);
  float ms = computeGradientDepth(shared, gx, gy);
  CHECK(ms >= 0.f, "bridge functions return elapsed milliseconds");
  std::vector<float> hgx((size_t)rows * cols), hgy((size_t)rows * cols);
  gx.download(hgx.data(), (size_t)cols * 4); gy.download(hgy.data(), (size_t)cols * 4);
  CHECK(std::fabs(hgx[(size_t)10 * cols + 20] - 0.5f) < 1e-6f && std::fabs(hgy[(size_t)10 * cols + 20] - 0.25f) < 1e-6f, "Sobel/8 of a plane");
  DepthMapf half;
  pyrDownDepth(shared, half);  // allocates dst (pyrdown.cu:199)
  CHECK(half.rows() == rows / 2 && half.cols() == cols / 2, "pyrDown creates the half-size destination");
  DeviceArray<float> err;
  computeErrorGridStride(deep, shared, err, 100);
  CHECK(err.size() > 0, "computeErrorGridStride creates the residual array");
  std::vector<float> herr; err.download(herr);
  bool zero = true; for (float v : herr) zero = zero && v == 0.f;
  CHECK(zero, "residual lattice of identical maps is exactly zero");
  Intr K(525.f / 10, 525.f / 10, 31.5f, 23.5f);
  MapArr vmap; createVMap(K, shared, vmap);
  CHECK(vmap.rows() == 3 * rows && vmap.cols() == cols, "createVMap creates the planar 3*rows map");
  Mat33 R = device_cast<Mat33>(std::array<float, 9>{1, 0, 0, 0, 1, 0, 0, 0, 1});
  CHECK(R.data[1].y == 1.f && R.data[2].x == 0.f, "device_cast<Mat33> of a row-major float[9]");
  sync();

  if (argc > 1 && !std::strcmp(argv[1], "error")) {
    DepthMapf small; small.create(8, 8);
    computeGradientDepth(shared, small, gy);  // size mismatch: must print "Error: ..." and exit(0) like pcl::gpu::error
    std::printf("FAILED error convention: the call returned\n");
    return 1;
  }
  std::printf("all ok\n");
  return 0;
}
