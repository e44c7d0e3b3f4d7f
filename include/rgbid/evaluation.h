// evaluation.h -- TUM RGB-D / ICL-NUIM dataset playback and result writers, the step before and after the tracking path
// (reference tools/evaluation.h:70-129, tools/evaluation.cpp:122-351,380-500).  OpenCV-free: PNGs are decoded by a small
// zlib-based reader (8/16-bit, grey / RGB / palette / alpha, non-interlaced -- everything the TUM, ICL and ETH3D sets use).
#pragma once
#include <memory>
#include <string>
#include <utility>
#include <vector>

#include "visodo.h"

namespace RGBID_SLAM {

template <typename T>
struct ImageWrapper {  // include/types.h ImageWrapper: non-owning view
  const T* data = nullptr;
  int rows = 0, cols = 0;
  size_t step = 0;
  size_t elemSize() const { return sizeof(T); }
};

// Decoded PNG: samples are host-endian, interleaved, `channels` per pixel, 8 or 16 bits each.
struct PngImage {
  int rows = 0, cols = 0, channels = 0, bit_depth = 0;
  std::vector<unsigned char> bytes;   // rows*cols*channels*(bit_depth/8)
};
// Throws std::runtime_error (loud) on a malformed / unsupported file.  read_png keeps the file's channel layout
// (palette expanded to RGB, sub-byte grey scaled to 8 bit as cv::imread does).
PngImage read_png(const std::string& path);
void write_png(const std::string& path, const void* data, int rows, int cols, int channels, int bit_depth);

class Evaluation {
 public:
  typedef std::shared_ptr<Evaluation> Ptr;
  typedef PixelRGB RGB;
  // folder holds depth_associated.txt + rgb_associated.txt (3 header lines each), or pass a 4-column match file
  // "t_depth depth_name t_rgb rgb_name" (evaluation.cpp:122-147,190-207)
  Evaluation(const std::string& folder, const std::string& match_file);
  void setMatchFile(const std::string& file);
  void associate_depth_rgb(const std::string& file_depth, const std::string& file_rgb);

  bool grab(double stamp, ImageWrapper<RGB>& rgb24);                 // index, not a stamp (evaluation.cpp:238)
  bool grab(double stamp, ImageWrapper<unsigned short>& depth);      // PNG x 0.2 -> millimetres (evaluation.cpp:293)
  bool grab(int stamp, ImageWrapper<unsigned short>& depth, ImageWrapper<RGB>& rgb24);

  static const float fx, fy, cx, cy;                                 // 525, 525, 319.5, 239.5 (evaluation.cpp:64-67)
  size_t size() const { return accociations_.empty() ? depth_stamps_and_filenames_.size() : accociations_.size(); }
  double stamp(size_t i) const { return accociations_.empty() ? depth_stamps_and_filenames_[i].first : accociations_[i].time1; }

  // "<stamp> tx ty tz qx qy qz qw", fixed notation, 6 decimals; misc file = timing summary + "<stamp>  <ms>" per frame
  void saveAllPoses(const VisodoTracker& visodo, int frame_number = -1, const std::string& poses_logfile = "visodo_poses.txt",
                    const std::string& chi_tests_logfile = "visodo_chi_tests.txt") const;
  void saveAllPoses(std::vector<Pose>& poses, const VisodoTracker& visodo, int frame_number = -1,
                    const std::string& poses_logfile = "visodo_poses.txt", const std::string& chi_tests_logfile = "visodo_chi_tests.txt") const;
  // back-end columns (segmentation, BoW, loop detection, pose graph) belong to the KeyframeManager, which is out of scope
  // here: they are written as the values passed in `backend_times` (rows of 5) or 0.
  void saveTimeLogFiles(const VisodoTracker& visodo, const std::vector<float>& backend_times, const std::string& kftimes_logfile) const;

 private:
  struct Association { double time1 = 0, time2 = 0; std::string name1, name2; };
  std::string folder_;
  std::vector<std::pair<double, std::string> > rgb_stamps_and_filenames_, depth_stamps_and_filenames_;
  std::vector<Association> accociations_;
  std::vector<unsigned short> depth_buffer_;
  std::vector<RGB> rgb_buffer_;
  void readFile(const std::string& file, std::vector<std::pair<double, std::string> >& output);
  bool load_depth(const std::string& file, ImageWrapper<unsigned short>& depth);
  bool load_rgb(const std::string& file, ImageWrapper<RGB>& rgb24);
};

// Eigen::Quaternionf(Matrix3f) restated (the writer's rotation -> quaternion step, evaluation.cpp:421-422); q = x,y,z,w
void rotation_to_quaternion_f(const double R[9], float q[4]);
// one trajectory line exactly as the reference formats it
std::string format_pose_line(double stamp, const double R[9], const double t[3]);

}  // namespace RGBID_SLAM
