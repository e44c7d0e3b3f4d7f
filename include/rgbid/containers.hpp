// containers.hpp -- HIP-native equivalents of the reference's device containers
// (ThirdParty/pcl_gpu_containers: include/device_memory.h, device_array.h, kernel_containers.h,
// impl/safe_call.hpp, initialization.h; src/device_memory.cpp:107-321, error.cpp:42-46).
//
// Same namespace, class names, member functions and semantics so the reference's host code compiles against
// them unchanged: ref-counted buffers (copying shares the allocation), `create()` is a no-op when the shape is
// unchanged, upload/download are synchronous, `step` is in bytes, errors print "Error: <what>\t<file>:<line>"
// and exit(0).  Storage comes from the C-ABI (rgbid_malloc / rgbid_malloc_pitch), i.e. hipMalloc with
// 256-byte-aligned rows, so no HIP header is needed to compile host code that uses these containers.
#pragma once
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <atomic>

#include "../rgbid.h"

namespace pcl {
namespace gpu {

// error.cpp:42-46
inline void error(const char* error_string, const char* file, const int line, const char* func = "") {
  std::printf("Error: %s\t%s:%d\n", error_string, file, line);
  (void)func;
  std::exit(0);
}
inline void ___rgbidSafeCall(int err, const char* file, const int line, const char* func = "") {
  if (err != RGBID_OK) error(rgbid_error_string(err), file, line, func);
}
#define rgbidSafeCall(expr) pcl::gpu::___rgbidSafeCall(expr, __FILE__, __LINE__)
#ifndef cudaSafeCall
#define cudaSafeCall(expr) rgbidSafeCall(expr)  /* source compatibility with the reference's call sites */
#endif
static inline int divUp(int total, int grain) { return (total + grain - 1) / grain; }

// device selected with setDevice() (the reference: pcl::gpu::setDevice before any allocation)
inline std::atomic<int>& current_device() { static std::atomic<int> d{0}; return d; }

// the per-thread context every container copy / bridge call runs on (the reference: per-thread default stream)
inline rgbid_ctx* default_ctx() {
  struct Holder {
    rgbid_ctx* c = nullptr;
    ~Holder() { if (c) rgbid_ctx_destroy(c); }
  };
  static thread_local Holder h;
  if (!h.c) rgbidSafeCall(rgbid_ctx_create(&h.c, current_device().load(), nullptr));
  return h.c;
}

// The reference's bridge contract is synchronous: every device function returns the elapsed milliseconds of its kernel (cudaTimer,
// device.hpp:83-106) after a stream synchronise.  That is the default here too.  A caller that ignores those return values -- the tracker's
// own frame loop does -- can open a ScopedAsyncBridge on its thread: until it closes, the bridge functions are enqueued without timing
// events or synchronisation (they return 0.f); uploads / downloads and every function that hands results to host memory (normal equations,
// sigma / nu, visibility ratios) still complete before returning, on the same per-thread stream, so program order is preserved.
inline bool& bridge_untimed() { static thread_local bool u = false; return u; }
struct ScopedAsyncBridge {
  bool engaged;
  explicit ScopedAsyncBridge(bool on = true) : engaged(on && !bridge_untimed()) {
    if (engaged) { bridge_untimed() = true; rgbidSafeCall(rgbid_ctx_set_async(default_ctx(), 1)); }
  }
  ~ScopedAsyncBridge() {
    if (engaged) { rgbid_ctx_sync(default_ctx()); rgbid_ctx_set_async(default_ctx(), 0); bridge_untimed() = false; }
  }
  ScopedAsyncBridge(const ScopedAsyncBridge&) = delete;
  ScopedAsyncBridge& operator=(const ScopedAsyncBridge&) = delete;
};
inline float* ms_arg(float& ms) { ms = 0.f; return bridge_untimed() ? nullptr : &ms; }

// kernel_containers.h:54-106
template <typename T> struct DevPtr {
  typedef T elem_type;
  const static size_t elem_size = sizeof(elem_type);
  T* data;
  DevPtr() : data(0) {}
  DevPtr(T* data_arg) : data(data_arg) {}
  size_t elemSize() const { return elem_size; }
  operator T*() { return data; }
  operator const T*() const { return data; }
};
template <typename T> struct PtrSz : public DevPtr<T> {
  PtrSz() : size(0) {}
  PtrSz(T* data_arg, size_t size_arg) : DevPtr<T>(data_arg), size(size_arg) {}
  size_t size;
};
template <typename T> struct PtrStep : public DevPtr<T> {
  PtrStep() : step(0) {}
  PtrStep(T* data_arg, size_t step_arg) : DevPtr<T>(data_arg), step(step_arg) {}
  size_t step;  // bytes, always
  T* ptr(int y = 0) { return (T*)((char*)DevPtr<T>::data + y * step); }
  const T* ptr(int y = 0) const { return (const T*)((const char*)DevPtr<T>::data + y * step); }
};
template <typename T> struct PtrStepSz : public PtrStep<T> {
  PtrStepSz() : cols(0), rows(0) {}
  PtrStepSz(int rows_arg, int cols_arg, T* data_arg, size_t step_arg) : PtrStep<T>(data_arg, step_arg), cols(cols_arg), rows(rows_arg) {}
  int cols;
  int rows;
};

// device_memory.cpp:107-207
class DeviceMemory {
 public:
  DeviceMemory() : data_(0), sizeBytes_(0), refcount_(0) {}
  ~DeviceMemory() { release(); }
  DeviceMemory(size_t sizeBytes_arg) : data_(0), sizeBytes_(0), refcount_(0) { create(sizeBytes_arg); }
  DeviceMemory(void* ptr_arg, size_t sizeBytes_arg) : data_(ptr_arg), sizeBytes_(sizeBytes_arg), refcount_(0) {}
  DeviceMemory(const DeviceMemory& o) : data_(o.data_), sizeBytes_(o.sizeBytes_), refcount_(o.refcount_) { if (refcount_) refcount_->fetch_add(1); }
  DeviceMemory& operator=(const DeviceMemory& o) {
    if (this != &o) {
      if (o.refcount_) o.refcount_->fetch_add(1);
      release();
      data_ = o.data_; sizeBytes_ = o.sizeBytes_; refcount_ = o.refcount_;
    }
    return *this;
  }
  void create(size_t sizeBytes_arg) {
    if (sizeBytes_arg == sizeBytes_) return;  // device_memory.cpp:139-140
    if (sizeBytes_arg > 0) {
      if (data_) release();
      sizeBytes_ = sizeBytes_arg;
      rgbidSafeCall(rgbid_malloc(&data_, sizeBytes_));
      refcount_ = new std::atomic<int>(1);
    }
  }
  void release() {
    if (refcount_ && refcount_->fetch_sub(1) == 1) {
      delete refcount_;
      rgbidSafeCall(rgbid_free(data_));
    }
    data_ = 0; sizeBytes_ = 0; refcount_ = 0;
  }
  void copyTo(DeviceMemory& other) const {
    if (empty()) other.release();
    else {
      other.create(sizeBytes_);
      rgbidSafeCall(rgbid_memcpy_d2d(default_ctx(), other.data_, data_, sizeBytes_));
    }
  }
  void upload(const void* host_ptr_arg, size_t sizeBytes_arg) {
    create(sizeBytes_arg);
    rgbidSafeCall(rgbid_memcpy_h2d(default_ctx(), data_, host_ptr_arg, sizeBytes_));
  }
  void download(void* host_ptr_arg) const { rgbidSafeCall(rgbid_memcpy_d2h(default_ctx(), host_ptr_arg, data_, sizeBytes_)); }
  void swap(DeviceMemory& o) { std::swap(data_, o.data_); std::swap(sizeBytes_, o.sizeBytes_); std::swap(refcount_, o.refcount_); }
  template <class T> T* ptr() { return (T*)data_; }
  template <class T> const T* ptr() const { return (const T*)data_; }
  template <class U> operator PtrSz<U>() const { PtrSz<U> r; r.data = (U*)ptr<U>(); r.size = sizeBytes_ / sizeof(U); return r; }
  bool empty() const { return !data_; }
  size_t sizeBytes() const { return sizeBytes_; }

 private:
  void* data_;
  size_t sizeBytes_;
  std::atomic<int>* refcount_;
};

// device_memory.cpp:211-321
class DeviceMemory2D {
 public:
  DeviceMemory2D() : data_(0), step_(0), colsBytes_(0), rows_(0), refcount_(0) {}
  ~DeviceMemory2D() { release(); }
  DeviceMemory2D(int rows_arg, int colsBytes_arg) : data_(0), step_(0), colsBytes_(0), rows_(0), refcount_(0) { create(rows_arg, colsBytes_arg); }
  DeviceMemory2D(int rows_arg, int colsBytes_arg, void* data_arg, size_t step_arg)
      : data_(data_arg), step_(step_arg), colsBytes_(colsBytes_arg), rows_(rows_arg), refcount_(0) {}
  DeviceMemory2D(const DeviceMemory2D& o) : data_(o.data_), step_(o.step_), colsBytes_(o.colsBytes_), rows_(o.rows_), refcount_(o.refcount_) {
    if (refcount_) refcount_->fetch_add(1);
  }
  DeviceMemory2D& operator=(const DeviceMemory2D& o) {
    if (this != &o) {
      if (o.refcount_) o.refcount_->fetch_add(1);
      release();
      colsBytes_ = o.colsBytes_; rows_ = o.rows_; data_ = o.data_; step_ = o.step_; refcount_ = o.refcount_;
    }
    return *this;
  }
  void create(int rows_arg, int colsBytes_arg) {
    if (colsBytes_ == colsBytes_arg && rows_ == rows_arg) return;  // device_memory.cpp:249-250
    if (rows_arg > 0 && colsBytes_arg > 0) {
      if (data_) release();
      colsBytes_ = colsBytes_arg; rows_ = rows_arg;
      rgbidSafeCall(rgbid_malloc_pitch(&data_, &step_, (size_t)colsBytes_, (size_t)rows_));
      refcount_ = new std::atomic<int>(1);
    }
  }
  void release() {
    if (refcount_ && refcount_->fetch_sub(1) == 1) {
      delete refcount_;
      rgbidSafeCall(rgbid_free(data_));
    }
    colsBytes_ = 0; rows_ = 0; data_ = 0; step_ = 0; refcount_ = 0;
  }
  void copyTo(DeviceMemory2D& other) const {
    if (empty()) other.release();
    else {
      other.create(rows_, colsBytes_);
      rgbidSafeCall(rgbid_memcpy2d_d2d(default_ctx(), other.data_, other.step_, data_, step_, (size_t)colsBytes_, (size_t)rows_));
    }
  }
  void upload(const void* host_ptr_arg, size_t host_step_arg, int rows_arg, int colsBytes_arg) {
    create(rows_arg, colsBytes_arg);
    rgbidSafeCall(rgbid_memcpy2d_h2d(default_ctx(), data_, step_, host_ptr_arg, host_step_arg, (size_t)colsBytes_, (size_t)rows_));
  }
  void download(void* host_ptr_arg, size_t host_step_arg) const {
    rgbidSafeCall(rgbid_memcpy2d_d2h(default_ctx(), host_ptr_arg, host_step_arg, data_, step_, (size_t)colsBytes_, (size_t)rows_));
  }
  void swap(DeviceMemory2D& o) {
    std::swap(data_, o.data_); std::swap(step_, o.step_); std::swap(colsBytes_, o.colsBytes_); std::swap(rows_, o.rows_); std::swap(refcount_, o.refcount_);
  }
  template <class T> T* ptr(int y_arg = 0) { return (T*)((char*)data_ + y_arg * step_); }
  template <class T> const T* ptr(int y_arg = 0) const { return (const T*)((const char*)data_ + y_arg * step_); }
  template <class U> operator PtrStep<U>() const { PtrStep<U> r; r.data = (U*)ptr<U>(); r.step = step_; return r; }
  template <class U> operator PtrStepSz<U>() const {
    PtrStepSz<U> r; r.data = (U*)ptr<U>(); r.step = step_; r.cols = colsBytes_ / sizeof(U); r.rows = rows_; return r;
  }
  bool empty() const { return !data_; }
  int colsBytes() const { return colsBytes_; }
  int rows() const { return rows_; }
  size_t step() const { return step_; }

 private:
  void* data_;
  size_t step_;
  int colsBytes_;
  int rows_;
  std::atomic<int>* refcount_;
};

// device_array.h / impl/device_array.hpp
template <class T> class DeviceArray : public DeviceMemory {
 public:
  typedef T type;
  enum { elem_size = sizeof(T) };
  DeviceArray() {}
  DeviceArray(size_t size) : DeviceMemory(size * elem_size) {}
  DeviceArray(T* ptr, size_t size) : DeviceMemory(ptr, size * elem_size) {}
  DeviceArray(const DeviceArray& other) : DeviceMemory(other) {}
  DeviceArray& operator=(const DeviceArray& other) { DeviceMemory::operator=(other); return *this; }
  void create(size_t size) { DeviceMemory::create(size * elem_size); }
  void release() { DeviceMemory::release(); }
  void copyTo(DeviceArray& other) const { DeviceMemory::copyTo(other); }
  void upload(const T* host_ptr, size_t size) { DeviceMemory::upload(host_ptr, size * elem_size); }
  void download(T* host_ptr) const { DeviceMemory::download(host_ptr); }
  template <class A> void upload(const std::vector<T, A>& data) { upload(&data[0], data.size()); }
  template <typename A> void download(std::vector<T, A>& data) const { data.resize(size()); if (!data.empty()) download(&data[0]); }
  void swap(DeviceArray& other_arg) { DeviceMemory::swap(other_arg); }
  T* ptr() { return DeviceMemory::ptr<T>(); }
  const T* ptr() const { return DeviceMemory::ptr<T>(); }
  operator T*() { return ptr(); }
  operator const T*() const { return ptr(); }
  size_t size() const { return sizeBytes() / elem_size; }
};

template <class T> class DeviceArray2D : public DeviceMemory2D {
 public:
  typedef T type;
  enum { elem_size = sizeof(T) };
  DeviceArray2D() {}
  DeviceArray2D(int rows, int cols) : DeviceMemory2D(rows, cols * elem_size) {}
  DeviceArray2D(int rows, int cols, void* data, size_t stepBytes) : DeviceMemory2D(rows, cols * elem_size, data, stepBytes) {}
  DeviceArray2D(const DeviceArray2D& other) : DeviceMemory2D(other) {}
  DeviceArray2D& operator=(const DeviceArray2D& other) { DeviceMemory2D::operator=(other); return *this; }
  void create(int rows, int cols) { DeviceMemory2D::create(rows, cols * elem_size); }
  void release() { DeviceMemory2D::release(); }
  void copyTo(DeviceArray2D& other) const { DeviceMemory2D::copyTo(other); }
  void upload(const void* host_ptr, size_t host_step, int rows, int cols) { DeviceMemory2D::upload(host_ptr, host_step, rows, cols * elem_size); }
  void download(void* host_ptr, size_t host_step) const { DeviceMemory2D::download(host_ptr, host_step); }
  void swap(DeviceArray2D& other_arg) { DeviceMemory2D::swap(other_arg); }
  template <class A> void upload(const std::vector<T, A>& data, int cols) { upload(&data[0], cols * elem_size, (int)(data.size() / cols), cols); }
  template <class A> void download(std::vector<T, A>& data, int& elem_step) const {
    elem_step = cols();
    data.resize(cols() * rows());
    if (!data.empty()) download(&data[0], cols() * elem_size);
  }
  T* ptr(int y = 0) { return DeviceMemory2D::ptr<T>(y); }
  const T* ptr(int y = 0) const { return DeviceMemory2D::ptr<T>(y); }
  operator T*() { return ptr(); }
  operator const T*() const { return ptr(); }
  int cols() const { return DeviceMemory2D::colsBytes() / elem_size; }
  int rows() const { return DeviceMemory2D::rows(); }
  size_t elem_step() const { return DeviceMemory2D::step() / elem_size; }
  // rgbid_img view for the C-ABI
  rgbid_img img() const { rgbid_img i; i.data = (void*)ptr(); i.step = step(); i.rows = rows(); i.cols = cols(); return i; }
};

// initialization.h:48-72
inline int getCudaEnabledDeviceCount() { int n = 0; rgbid_device_count(&n); return n; }
inline void setDevice(int device) { rgbidSafeCall(rgbid_set_device(device)); current_device().store(device); }
inline void printShortCudaDeviceInfo(int device) {
  rgbid_device_prop p;
  rgbidSafeCall(rgbid_get_device_prop(device, &p));
  std::printf("[%s %s] Device %d: \"%s\"  %.0fMb, %d CUs, wavefront %d\n", "rgbid", p.gcnArchName, device, p.name,
              (double)p.totalGlobalMem / 1048576.0, p.multiProcessorCount, p.warpSize);
}

}  // namespace gpu
namespace device {
using pcl::gpu::DeviceArray;
using pcl::gpu::DeviceArray2D;
using pcl::gpu::DeviceMemory;
using pcl::gpu::DeviceMemory2D;
using pcl::gpu::divUp;
using pcl::gpu::PtrStep;
using pcl::gpu::PtrStepSz;
using pcl::gpu::PtrSz;
}  // namespace device
}  // namespace pcl
