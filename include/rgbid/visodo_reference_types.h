// visodo_reference_types.h -- OPT-IN adapters between the types on the tracker surface of this repository (include/rgbid/visodo.h: std::mutex,
// plain row-major matrices, TrackerSink) and the types the reference exposes there (include/visodo.h:95-147: boost::mutex,
// boost::condition_variable, Eigen::Affine3f / Affine3d / Matrix3d / Vector3d, KeyframeManagerPtr), for a caller that has Boost and / or Eigen
// and wants to keep its call sites as they are.
//
//   * always:                 RGBID_SLAM::compat::mutex -- a std::mutex with Boost's nested lock types (scoped_lock, scoped_try_lock), so that
//                             `boost::mutex::scoped_try_lock lock(visodo_->mutex_)` (tools/RGBID_SLAMapp.cpp:146,173) becomes
//                             `RGBID_SLAM::compat::mutex::scoped_try_lock lock(visodo_->mutex_)` -- a type name, not a rewrite of the locking code;
//   * -DRGBID_WITH_EIGEN:     toEigen() / fromEigen() between Matrix3ft / Vector3ft / Affine3d / Matrix6d and their Eigen counterparts, and
//                             getCameraPoseEigen() with the reference's return type (Eigen::Affine3f, include/visodo.h:95);
//   * -DRGBID_WITH_BOOST:     a TrackerSink that forwards to any object with the reference KeyframeManager's members (poses_, constraints_,
//                             buffer_keyframes_, mutex_odometry_: include/keyframe_manager.h:77-104) is out of this header's reach without the
//                             reference's own class definitions; INTEGRATION.md section 2 shows the ten-line adapter instead.
// NOT TESTED IN THIS REPOSITORY beyond the default configuration: the image the repository is built and tested in has neither Eigen nor Boost
// (tests/test_cpu_surface.py compiles the default branch; the Eigen branch is written against Eigen 3's documented API and is untested).
#pragma once
#include "visodo.h"

#include <mutex>

namespace RGBID_SLAM {
namespace compat {

// std::mutex under Boost's spelling of its lock types (boost/thread/mutex.hpp: mutex::scoped_lock, mutex::scoped_try_lock)
class mutex : public std::mutex {
 public:
  typedef std::unique_lock<std::mutex> scoped_lock;
  class scoped_try_lock {
   public:
    explicit scoped_try_lock(std::mutex& m) : l_(m, std::try_to_lock) {}
    explicit operator bool() const { return l_.owns_lock(); }
    bool owns_lock() const { return l_.owns_lock(); }
    void unlock() { l_.unlock(); }
    std::unique_lock<std::mutex>& native() { return l_; }   // what std::condition_variable::wait takes
   private:
    std::unique_lock<std::mutex> l_;
  };
};

}  // namespace compat
}  // namespace RGBID_SLAM

#ifdef RGBID_WITH_EIGEN
#include <Eigen/Core>
#include <Eigen/Geometry>

namespace RGBID_SLAM {
namespace compat {

typedef Eigen::Matrix<double, 3, 3, Eigen::RowMajor> EigenMatrix3ft;   // include/types.h:490-496 of the reference
inline EigenMatrix3ft toEigen(const Matrix3ft& m) { return Eigen::Map<const EigenMatrix3ft>(m.data()); }
inline Eigen::Vector3d toEigen(const Vector3ft& v) { return Eigen::Map<const Eigen::Vector3d>(v.data()); }
inline Matrix3ft fromEigen(const EigenMatrix3ft& m) { Matrix3ft r; Eigen::Map<EigenMatrix3ft>(r.data()) = m; return r; }
inline Vector3ft fromEigen(const Eigen::Vector3d& v) { Vector3ft r; Eigen::Map<Eigen::Vector3d>(r.data()) = v; return r; }
inline Eigen::Affine3d toEigen(const Affine3d& a) {
  Eigen::Affine3d e = Eigen::Affine3d::Identity();
  e.linear() = toEigen(a.linear());
  e.translation() = toEigen(a.translation());
  return e;
}
inline Affine3d fromEigen(const Eigen::Affine3d& e) {
  Affine3d a;
  a.R = fromEigen(EigenMatrix3ft(e.linear()));
  a.t = fromEigen(Eigen::Vector3d(e.translation()));
  return a;
}
inline Eigen::Matrix<double, 6, 6> toEigen(const Matrix6d& c) { return Eigen::Map<const Eigen::Matrix<double, 6, 6, Eigen::RowMajor> >(c.data()); }
// VisodoTracker::getCameraPose with the reference's return type (include/visodo.h:95)
inline Eigen::Affine3f getCameraPoseEigen(const VisodoTracker& t, int time = -1) { return toEigen(t.getCameraPose(time)).cast<float>(); }
inline void setSharedCameraPoseEigen(VisodoTracker& t, const Eigen::Affine3d& pose) { t.setSharedCameraPose(fromEigen(pose)); }

}  // namespace compat
}  // namespace RGBID_SLAM
#endif  // RGBID_WITH_EIGEN
