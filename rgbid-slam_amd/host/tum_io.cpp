// tum_io.cpp -- see include/rgbid/evaluation.h.  Follows tools/evaluation.cpp of the reference for file formats and
// playback semantics; the PNG codec replaces cv::imread (OpenCV is not a dependency of this build).
#include <zlib.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>
#include <sstream>
#include <stdexcept>

#include "../../include/rgbid/evaluation.h"
#include "../../include/rgbid/se3.h"

namespace RGBID_SLAM {

const float Evaluation::fx = 525.0f;
const float Evaluation::fy = 525.0f;
const float Evaluation::cx = 319.5f;
const float Evaluation::cy = 239.5f;

// ------------------------------------------------------------------------------------------------ PNG
namespace {
uint32_t be32(const unsigned char* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }
[[noreturn]] void png_fail(const std::string& path, const char* why) { throw std::runtime_error("png: " + path + ": " + why); }
inline int paeth(int a, int b, int c) {
  int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
  return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}
}  // namespace

PngImage read_png(const std::string& path) {
  std::ifstream f(path.c_str(), std::ios::binary);
  if (!f) png_fail(path, "cannot open");
  std::vector<unsigned char> file((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
  static const unsigned char sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
  if (file.size() < 8 + 25 || std::memcmp(file.data(), sig, 8)) png_fail(path, "not a PNG");
  size_t pos = 8;
  int w = 0, h = 0, depth = 0, ctype = -1, interlace = 0;
  std::vector<unsigned char> idat, plte;
  bool end = false;
  while (!end && pos + 12 <= file.size()) {
    uint32_t len = be32(&file[pos]);
    const unsigned char* type = &file[pos + 4];
    if (pos + 12 + (size_t)len > file.size()) png_fail(path, "truncated chunk");
    const unsigned char* data = &file[pos + 8];
    uint32_t crc = (uint32_t)crc32(crc32(0, Z_NULL, 0), type, len + 4);
    if (crc != be32(data + len)) png_fail(path, "chunk CRC mismatch");
    if (!std::memcmp(type, "IHDR", 4)) {
      if (len != 13) png_fail(path, "bad IHDR");
      w = (int)be32(data); h = (int)be32(data + 4); depth = data[8]; ctype = data[9]; interlace = data[12];
      if (data[10] != 0 || data[11] != 0) png_fail(path, "unknown compression/filter method");
    } else if (!std::memcmp(type, "PLTE", 4)) plte.assign(data, data + len);
    else if (!std::memcmp(type, "IDAT", 4)) idat.insert(idat.end(), data, data + len);
    else if (!std::memcmp(type, "IEND", 4)) end = true;
    pos += 12 + (size_t)len;
  }
  if (ctype < 0 || w <= 0 || h <= 0) png_fail(path, "missing IHDR");
  if (interlace) png_fail(path, "Adam7 interlacing not supported");
  int ch = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 3 ? 1 : ctype == 4 ? 2 : ctype == 6 ? 4 : 0;
  if (!ch) png_fail(path, "bad colour type");
  bool depth_ok = (ctype == 0) ? (depth == 1 || depth == 2 || depth == 4 || depth == 8 || depth == 16)
                  : (ctype == 3) ? (depth == 1 || depth == 2 || depth == 4 || depth == 8) : (depth == 8 || depth == 16);
  if (!depth_ok) png_fail(path, "bad bit depth");
  size_t bpp_bits = (size_t)ch * depth, stride = ((size_t)w * bpp_bits + 7) / 8, bpp = (bpp_bits + 7) / 8;
  std::vector<unsigned char> raw((stride + 1) * (size_t)h);
  uLongf raw_len = (uLongf)raw.size();
  int zr = uncompress(raw.data(), &raw_len, idat.data(), (uLong)idat.size());
  if (zr != Z_OK || raw_len != raw.size()) png_fail(path, "inflate failed / wrong size");
  // unfilter in place (rows keep their leading filter byte)
  std::vector<unsigned char> zero(stride, 0);
  for (int y = 0; y < h; ++y) {
    unsigned char* cur = &raw[(stride + 1) * (size_t)y + 1];
    const unsigned char* up = y ? &raw[(stride + 1) * (size_t)(y - 1) + 1] : zero.data();
    int ft = cur[-1];
    switch (ft) {
      case 0: break;
      case 1: for (size_t i = bpp; i < stride; ++i) cur[i] = (unsigned char)(cur[i] + cur[i - bpp]); break;
      case 2: for (size_t i = 0; i < stride; ++i) cur[i] = (unsigned char)(cur[i] + up[i]); break;
      case 3: for (size_t i = 0; i < stride; ++i) cur[i] = (unsigned char)(cur[i] + (((i >= bpp ? cur[i - bpp] : 0) + up[i]) >> 1)); break;
      case 4: for (size_t i = 0; i < stride; ++i)
                cur[i] = (unsigned char)(cur[i] + paeth(i >= bpp ? cur[i - bpp] : 0, up[i], i >= bpp ? up[i - bpp] : 0));
              break;
      default: png_fail(path, "bad filter type");
    }
  }
  PngImage img;
  img.rows = h; img.cols = w;
  if (ctype == 3) {  // palette -> RGB8
    img.channels = 3; img.bit_depth = 8;
    img.bytes.resize((size_t)w * h * 3);
    for (int y = 0; y < h; ++y) {
      const unsigned char* row = &raw[(stride + 1) * (size_t)y + 1];
      for (int x = 0; x < w; ++x) {
        size_t bit = (size_t)x * depth;
        unsigned idx = (row[bit >> 3] >> (8 - depth - (bit & 7))) & ((1u << depth) - 1);
        if ((size_t)idx * 3 + 2 >= plte.size()) png_fail(path, "palette index out of range");
        std::memcpy(&img.bytes[((size_t)y * w + x) * 3], &plte[(size_t)idx * 3], 3);
      }
    }
    return img;
  }
  img.channels = ch;
  if (depth < 8) {  // sub-byte grey -> 8 bit, scaled to full range
    img.bit_depth = 8;
    img.bytes.resize((size_t)w * h);
    unsigned maxv = (1u << depth) - 1;
    for (int y = 0; y < h; ++y) {
      const unsigned char* row = &raw[(stride + 1) * (size_t)y + 1];
      for (int x = 0; x < w; ++x) {
        size_t bit = (size_t)x * depth;
        unsigned v = (row[bit >> 3] >> (8 - depth - (bit & 7))) & maxv;
        img.bytes[(size_t)y * w + x] = (unsigned char)(v * 255u / maxv);
      }
    }
    return img;
  }
  img.bit_depth = depth;
  img.bytes.resize((size_t)h * stride);
  for (int y = 0; y < h; ++y) {
    const unsigned char* row = &raw[(stride + 1) * (size_t)y + 1];
    unsigned char* dst = &img.bytes[(size_t)y * stride];
    if (depth == 8) std::memcpy(dst, row, stride);
    else {  // big-endian samples -> host uint16
      uint16_t* d16 = reinterpret_cast<uint16_t*>(dst);
      for (size_t i = 0; i < stride / 2; ++i) d16[i] = (uint16_t)((row[2 * i] << 8) | row[2 * i + 1]);
    }
  }
  return img;
}

void write_png(const std::string& path, const void* data, int rows, int cols, int channels, int bit_depth) {
  int ctype = channels == 1 ? 0 : channels == 2 ? 4 : channels == 3 ? 2 : channels == 4 ? 6 : -1;
  if (ctype < 0 || (bit_depth != 8 && bit_depth != 16) || rows <= 0 || cols <= 0) png_fail(path, "unsupported layout for writing");
  size_t stride = (size_t)cols * channels * (bit_depth / 8);
  std::vector<unsigned char> raw((stride + 1) * (size_t)rows);
  const unsigned char* src = static_cast<const unsigned char*>(data);
  for (int y = 0; y < rows; ++y) {
    unsigned char* row = &raw[(stride + 1) * (size_t)y];
    row[0] = 0;
    if (bit_depth == 8) std::memcpy(row + 1, src + (size_t)y * stride, stride);
    else {
      const uint16_t* s16 = reinterpret_cast<const uint16_t*>(src + (size_t)y * stride);
      for (size_t i = 0; i < stride / 2; ++i) { row[1 + 2 * i] = (unsigned char)(s16[i] >> 8); row[2 + 2 * i] = (unsigned char)(s16[i] & 255); }
    }
  }
  uLongf zlen = compressBound((uLong)raw.size());
  std::vector<unsigned char> z(zlen);
  if (compress2(z.data(), &zlen, raw.data(), (uLong)raw.size(), 6) != Z_OK) png_fail(path, "deflate failed");
  std::ofstream f(path.c_str(), std::ios::binary);
  if (!f) png_fail(path, "cannot create");
  auto put_chunk = [&](const char* type, const unsigned char* d, uint32_t len) {
    unsigned char hdr[8] = {(unsigned char)(len >> 24), (unsigned char)(len >> 16), (unsigned char)(len >> 8), (unsigned char)len,
                            (unsigned char)type[0], (unsigned char)type[1], (unsigned char)type[2], (unsigned char)type[3]};
    uint32_t crc = (uint32_t)crc32(crc32(0, Z_NULL, 0), hdr + 4, 4);
    if (len) crc = (uint32_t)crc32(crc, d, len);
    unsigned char c[4] = {(unsigned char)(crc >> 24), (unsigned char)(crc >> 16), (unsigned char)(crc >> 8), (unsigned char)crc};
    f.write((const char*)hdr, 8);
    if (len) f.write((const char*)d, len);
    f.write((const char*)c, 4);
  };
  static const unsigned char sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
  f.write((const char*)sig, 8);
  unsigned char ihdr[13] = {(unsigned char)(cols >> 24), (unsigned char)(cols >> 16), (unsigned char)(cols >> 8), (unsigned char)cols,
                            (unsigned char)(rows >> 24), (unsigned char)(rows >> 16), (unsigned char)(rows >> 8), (unsigned char)rows,
                            (unsigned char)bit_depth, (unsigned char)ctype, 0, 0, 0};
  put_chunk("IHDR", ihdr, 13);
  put_chunk("IDAT", z.data(), (uint32_t)zlen);
  put_chunk("IEND", nullptr, 0);
}

// ------------------------------------------------------------------------------------------------ dataset index
Evaluation::Evaluation(const std::string& folder, const std::string& match_file) : folder_(folder) {
  if (folder_.empty() || (folder_[folder_.size() - 1] != '\\' && folder_[folder_.size() - 1] != '/')) folder_.push_back('/');
  if (!match_file.empty()) setMatchFile(match_file);
  else {
    std::string depth_file = folder_ + "depth_associated.txt", rgb_file = folder_ + "rgb_associated.txt";
    readFile(depth_file, depth_stamps_and_filenames_);
    readFile(rgb_file, rgb_stamps_and_filenames_);
    associate_depth_rgb(depth_file, rgb_file);
  }
}

namespace {
void skip_lines(std::istream& s, int n) { std::string l; for (int i = 0; i < n; ++i) std::getline(s, l); }
[[noreturn]] void io_fail(const std::string& what) { std::cout << what << std::endl; throw std::runtime_error(what); }
}  // namespace

// The reference loops `while(!eof) { s >> time >> name; push_back }` (evaluation.cpp:176-181,196-201,222-228), which
// appends one empty entry after a trailing newline; that entry can never be grabbed (empty file name -> imread fails), so
// playback ends there.  Here a failed extraction simply ends the list -- same frames, no phantom entry.
void Evaluation::associate_depth_rgb(const std::string& file_depth, const std::string& file_rgb) {
  std::ifstream d(file_depth.c_str()), c(file_rgb.c_str());
  if (!d || !c) io_fail("Can't read rgbd" + file_depth);
  skip_lines(d, 3); skip_lines(c, 3);
  accociations_.clear();
  for (;;) {
    Association a;
    bool okd = (bool)(d >> a.time1 >> a.name1), okc = (bool)(c >> a.time2 >> a.name2);
    if (!okd || !okc) break;
    accociations_.push_back(a);
  }
}

void Evaluation::setMatchFile(const std::string& file) {
  std::string full = folder_ + file;
  std::ifstream iff(full.c_str());
  if (!iff) io_fail("Can't read " + file);
  accociations_.clear();
  Association a;
  while (iff >> a.time1 >> a.name1 >> a.time2 >> a.name2) accociations_.push_back(a);
}

void Evaluation::readFile(const std::string& file, std::vector<std::pair<double, std::string> >& output) {
  std::ifstream iff(file.c_str());
  if (!iff) io_fail("Can't read" + file);
  skip_lines(iff, 3);
  std::vector<std::pair<double, std::string> > tmp;
  double time; std::string name;
  while (iff >> time >> name) tmp.push_back(std::make_pair(time, name));
  tmp.swap(output);
}

// ------------------------------------------------------------------------------------------------ grabbing
bool Evaluation::load_rgb(const std::string& file, ImageWrapper<RGB>& rgb24) {
  PngImage img;
  try { img = read_png(file); } catch (const std::exception&) { return false; }   // cv::imread(...).empty() -> false
  // cv::imread default flag: always 3 x 8-bit (16-bit >> 8, grey replicated, alpha dropped); cvtColor BGR2RGB -> r,g,b
  size_t n = (size_t)img.rows * img.cols;
  rgb_buffer_.resize(n);
  int ch = img.channels, bs = img.bit_depth / 8;
  for (size_t i = 0; i < n; ++i) {
    unsigned v[4] = {0, 0, 0, 0};
    for (int k = 0; k < ch; ++k) {
      if (bs == 1) v[k] = img.bytes[i * ch + k];
      else v[k] = reinterpret_cast<const uint16_t*>(img.bytes.data())[i * ch + k] >> 8;
    }
    if (ch <= 2) { rgb_buffer_[i].r = rgb_buffer_[i].g = rgb_buffer_[i].b = (unsigned char)v[0]; }
    else { rgb_buffer_[i].r = (unsigned char)v[0]; rgb_buffer_[i].g = (unsigned char)v[1]; rgb_buffer_[i].b = (unsigned char)v[2]; }
  }
  rgb24.data = rgb_buffer_.data(); rgb24.cols = img.cols; rgb24.rows = img.rows; rgb24.step = (size_t)img.cols * sizeof(RGB);
  return true;
}

bool Evaluation::load_depth(const std::string& file, ImageWrapper<unsigned short>& depth) {
  PngImage img;
  try { img = read_png(file); } catch (const std::exception&) { return false; }
  if (img.bit_depth != 16 || img.channels != 1) {  // evaluation.cpp:286-290
    std::cout << "Image was not opend in 16-bit format. Please use OpenCV 2.3.1 or higher" << std::endl;
    throw std::runtime_error("depth image is not 16-bit single channel: " + file);
  }
  // Datasets store depth*5000; convertTo(type, 0.2) = saturate_cast<ushort>(cvRound(v*0.2f)) -> millimetres (evaluation.cpp:296)
  size_t n = (size_t)img.rows * img.cols;
  depth_buffer_.resize(n);
  const uint16_t* s = reinterpret_cast<const uint16_t*>(img.bytes.data());
  for (size_t i = 0; i < n; ++i) depth_buffer_[i] = (unsigned short)std::nearbyintf((float)s[i] * 0.2f);
  depth.data = depth_buffer_.data(); depth.cols = img.cols; depth.rows = img.rows; depth.step = (size_t)img.cols * 2;
  return true;
}

bool Evaluation::grab(double stamp, ImageWrapper<RGB>& rgb24) {
  size_t i = static_cast<size_t>(stamp);
  size_t total = accociations_.empty() ? rgb_stamps_and_filenames_.size() : accociations_.size();
  if (i >= total) return false;
  return load_rgb(folder_ + (accociations_.empty() ? rgb_stamps_and_filenames_[i].second : accociations_[i].name2), rgb24);
}

bool Evaluation::grab(double stamp, ImageWrapper<unsigned short>& depth) {
  size_t i = static_cast<size_t>(stamp);
  size_t total = accociations_.empty() ? depth_stamps_and_filenames_.size() : accociations_.size();
  if (i >= total) return false;
  return load_depth(folder_ + (accociations_.empty() ? depth_stamps_and_filenames_[i].second : accociations_[i].name1), depth);
}

bool Evaluation::grab(int stamp, ImageWrapper<unsigned short>& depth, ImageWrapper<RGB>& rgb24) {
  if (accociations_.empty()) io_fail("Please set match file");
  size_t i = static_cast<size_t>(stamp);
  if (stamp < 0 || i >= accociations_.size()) return false;
  if (!load_depth(folder_ + accociations_[i].name1, depth)) return false;
  return load_rgb(folder_ + accociations_[i].name2, rgb24);
}

// ------------------------------------------------------------------------------------------------ writers
void rotation_to_quaternion_f(const double Rd[9], float q[4]) {
  // Affine3f::rotation() (polar factor of the linear part) then Eigen::Quaternionf(Matrix3f): Eigen/src/Geometry/Quaternion.h
  // quaternionbase_assign_impl<Matrix3f>: trace branch / largest-diagonal branch, all in float.
  double Ro[9];
  rgbid::se3::force_orthogonal(Rd, Ro);
  float m[3][3];
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) m[i][j] = (float)Ro[i * 3 + j];
  float t = m[0][0] + m[1][1] + m[2][2];
  if (t > 0.f) {
    t = std::sqrt(t + 1.0f);
    q[3] = 0.5f * t;
    t = 0.5f / t;
    q[0] = (m[2][1] - m[1][2]) * t;
    q[1] = (m[0][2] - m[2][0]) * t;
    q[2] = (m[1][0] - m[0][1]) * t;
  } else {
    int i = 0;
    if (m[1][1] > m[0][0]) i = 1;
    if (m[2][2] > m[i][i]) i = 2;
    int j = (i + 1) % 3, k = (j + 1) % 3;
    t = std::sqrt(m[i][i] - m[j][j] - m[k][k] + 1.0f);
    q[i] = 0.5f * t;
    t = 0.5f / t;
    q[3] = (m[k][j] - m[j][k]) * t;
    q[j] = (m[j][i] + m[i][j]) * t;
    q[k] = (m[k][i] + m[i][k]) * t;
  }
}

std::string format_pose_line(double stamp, const double R[9], const double t[3]) {
  float q[4];
  rotation_to_quaternion_f(R, q);
  std::ostringstream s;
  s.setf(std::ios::fixed, std::ios::floatfield);
  s << stamp << " " << (float)t[0] << " " << (float)t[1] << " " << (float)t[2] << " " << q[0] << " " << q[1] << " " << q[2] << " " << q[3];
  return s.str();
}

namespace {
void write_time_summary(std::ostream& misc, const VisodoTracker& visodo, int frame_number) {
  // evaluation.cpp:398-424 (mean / population std / max of the per-frame tracking time, float accumulation)
  float mean = 0.f, sd = 0.f, mx = 0.f;
  for (int i = 0; i < frame_number; ++i) {
    mean += visodo.getVisOdoTime(i) / frame_number;
    if (visodo.getVisOdoTime(i) > mx) mx = visodo.getVisOdoTime(i);
  }
  for (int i = 0; i < frame_number; ++i) sd += (visodo.getVisOdoTime(i) - mean) * (visodo.getVisOdoTime(i) - mean) / frame_number;
  sd = std::sqrt(sd);
  misc << "Mean time per frame: " << mean << std::endl << "Std time per frame: " << sd << std::endl << "Max time per frame: " << mx << std::endl;
  std::cout << "Mean time per frame: " << mean << std::endl << "Std time per frame: " << sd << std::endl << "Max time per frame: " << mx << std::endl;
}
}  // namespace

void Evaluation::saveAllPoses(const VisodoTracker& visodo, int frame_number, const std::string& poses_logfile, const std::string& chi_tests_logfile) const {
  if (frame_number < 0) frame_number = (int)size();
  frame_number = std::min(frame_number, (int)visodo.getNumberOfPoses());
  std::cout << "Writing " << frame_number << " poses to " << poses_logfile << std::endl;
  std::ofstream poses(poses_logfile.c_str()), misc(chi_tests_logfile.c_str());
  misc.setf(std::ios::fixed, std::ios::floatfield);
  write_time_summary(misc, visodo, frame_number);
  for (int i = 0; i < frame_number; ++i) {
    Affine3d pose = visodo.getCameraPose(i);
    poses << format_pose_line(stamp(i), pose.R.m, pose.t.v) << std::endl;
    misc << stamp(i) << " " << " " << visodo.getVisOdoTime(i) << std::endl;
  }
}

void Evaluation::saveAllPoses(std::vector<Pose>& poses_in, const VisodoTracker& visodo, int frame_number, const std::string& poses_logfile,
                              const std::string& chi_tests_logfile) const {
  if (frame_number < 0) frame_number = (int)size();
  frame_number = std::min(frame_number, (int)visodo.getNumberOfPoses());
  std::cout << "Writing " << frame_number << " poses to " << poses_logfile << std::endl;
  std::ofstream poses(poses_logfile.c_str()), misc(chi_tests_logfile.c_str());
  misc.setf(std::ios::fixed, std::ios::floatfield);
  write_time_summary(misc, visodo, frame_number);
  for (size_t i = 0; i < poses_in.size() && i < size(); ++i) {
    poses << format_pose_line(stamp(i), poses_in[i].rotation_.m, poses_in[i].translation_.v) << std::endl;
    misc << stamp(i) << " " << " " << visodo.getVisOdoTime((int)i) << std::endl;
  }
}

void Evaluation::saveTimeLogFiles(const VisodoTracker& visodo, const std::vector<float>& backend_times, const std::string& kftimes_logfile) const {
  // evaluation.cpp:353-377
  std::ofstream f(kftimes_logfile.c_str());
  f.setf(std::ios::fixed, std::ios::floatfield);
  f << "ObtainKeyframe " << "ProcessKeyframeTotal " << "Segmentation " << "DescriptionBoW " << "LoopDetection " << "PoseGraphOptim" << std::endl;
  for (size_t i = 0; i < visodo.kf_times_.size(); ++i) {
    f << visodo.kf_times_[i];
    for (int k = 0; k < 5; ++k) f << " " << (i * 5 + k < backend_times.size() ? backend_times[i * 5 + k] : 0.f);
    f << std::endl;
  }
}

}  // namespace RGBID_SLAM
