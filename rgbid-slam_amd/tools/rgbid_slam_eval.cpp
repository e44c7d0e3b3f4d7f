// rgbid_slam_eval -- offline evaluation harness for the tracking front-end: the `-eval` mode of the reference application
// (tools/RGBID_SLAMapp.cpp:171-222 playback loop, :360-500 CLI and result files) without the back-end / viewer / ROS.
//   rgbid_slam_eval -eval <dataset_folder/> [-match_file <f>] [-config <ini>] [-calib <ini>] [-gpu <id>] [-out <dir>] [-max_frames N]
//                   [-threaded [-sleep_ms 30]]
// -threaded reproduces the reference's two-thread playback: the tracker runs on its own thread (VisodoTracker::start, operator()),
// the grabber try-locks visodo.mutex_, uploads depth_ / rgb24_, notifies new_frame_cond_ and sleeps 30 ms (simulateLoopCallback).
// writes <dataset>_poses.txt ("stamp tx ty tz qx qy qz qw"), <dataset>_misc.txt, <dataset>_kf_times.txt.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <chrono>
#include <mutex>
#include <string>
#include <thread>

#include "../../include/rgbid/evaluation.h"
#include "../../include/rgbid/settings.h"
#include "../../include/rgbid/visodo.h"
#include "../../include/rgbid_host.h"

using namespace RGBID_SLAM;

static int print_cli_help() {
  std::cout << "\nVisodo RGBD parameters:\n"
            << "    --help, -h                          : print this message\n"
            << "    -config  <filename>  : Load configuration file  \n"
            << "    -calib <filename>  : Load calibration file  \n"
            << "    -eval <dataset_folder>  : Evaluation mode for dataset in TUM format \n"
            << "    -match_file <filename> : Provide file with matches between RGB and depth frames  \n"
            << "    -gpu <id> : Specify gpu id (in case there is more than one, id=0 by default)  \n"
            << "    -out <dir> : directory for the result files (default: current directory)\n"
            << "    -max_frames <n> : stop after n frames\n"
            << "    -threaded : tracker on its own thread, grabber notifies it (the reference's playback); -sleep_ms <n> between frames (30)\n\n";
  return 0;
}

static bool arg_value(int argc, char** argv, const char* key, std::string& val) {
  for (int i = 1; i + 1 < argc; ++i) if (!std::strcmp(argv[i], key)) { val = argv[i + 1]; return true; }
  return false;
}
static bool arg_switch(int argc, char** argv, const char* key) {
  for (int i = 1; i < argc; ++i) if (!std::strcmp(argv[i], key)) return true;
  return false;
}

int main(int argc, char* argv[]) {
  if (arg_switch(argc, argv, "--help") || arg_switch(argc, argv, "-h")) return print_cli_help();
  std::string s, config_file, calib_file, eval_folder, match_file, out_dir;
  int max_frames = -1, sleep_ms = 30;
  const bool threaded = arg_switch(argc, argv, "-threaded");
  if (arg_value(argc, argv, "-sleep_ms", s)) sleep_ms = std::atoi(s.c_str());
  device::dev_id = 0;
  if (arg_value(argc, argv, "-gpu", s)) device::dev_id = std::atoi(s.c_str());
  arg_value(argc, argv, "-config", config_file);
  arg_value(argc, argv, "-calib", calib_file);
  arg_value(argc, argv, "-match_file", match_file);
  arg_value(argc, argv, "-out", out_dir);
  if (arg_value(argc, argv, "-max_frames", s)) max_frames = std::atoi(s.c_str());
  if (!arg_value(argc, argv, "-eval", eval_folder)) { std::cout << "-eval <dataset_folder> is required (live capture is not part of this build)\n"; return 2; }
  if (!out_dir.empty() && out_dir[out_dir.size() - 1] != '/') out_dir.push_back('/');

  pcl::gpu::setDevice(device::dev_id);
  if (rgbid_get_device_prop(device::dev_id, &device::dev_prop) != 0) { std::cout << "no HIP device " << device::dev_id << std::endl; return 1; }
  pcl::gpu::printShortCudaDeviceInfo(device::dev_id);

  // log names (RGBID_SLAMapp.cpp:412-433): <dataset>_poses.txt ...; dataset = last directory component of the folder
  std::size_t found_last = eval_folder.find_last_of("/\\");
  std::string eval_folder2 = eval_folder.substr(0, found_last);
  std::size_t found_prelast = eval_folder2.find_last_of("/\\");
  std::string dataset_name = eval_folder2.substr(found_prelast + 1);
  std::string poses_logfile = out_dir + dataset_name + "_poses.txt", misc_logfile = out_dir + dataset_name + "_misc.txt",
              kf_times_logfile = out_dir + dataset_name + "_kf_times.txt";

  VisodoTracker visodo;
  Evaluation::Ptr evaluation;
  try { evaluation.reset(new Evaluation(eval_folder, match_file)); }
  catch (const std::exception& e) { return 1; }
  visodo.setRGBIntrinsics(Evaluation::fx, Evaluation::fy, Evaluation::cx, Evaluation::cy);   // toggleEvaluationMode :146
  if (!config_file.empty()) {
    std::ifstream fs(config_file.c_str());
    if (!fs.is_open()) std::cout << "Could not open configuration file " << config_file << std::endl;
    else { Settings settings(fs); visodo.loadSettings(settings); }
  }
  if (!calib_file.empty()) visodo.loadCalibration(calib_file);
  // simulateLoopCallback (:171-222): frame index walks the association list; ten consecutive unreadable pairs end playback
  visodo.compute_deltat_flag_ = false;
  ImageWrapper<unsigned short> depth; ImageWrapper<PixelRGB> rgb24;
  int currentIndex = 0, num_failures = 0, tracked = 0;
  auto size_ok = [&]() {
    if (depth.rows == visodo.rows() && depth.cols == visodo.cols() && rgb24.rows == visodo.rows() && rgb24.cols == visodo.cols()) return true;
    std::cout << "frame " << currentIndex << " is " << depth.cols << "x" << depth.rows << ", tracker expects " << visodo.cols() << "x" << visodo.rows() << std::endl;
    return false;
  };
  if (threaded) {
    visodo.start();  // the tracker thread now waits on new_frame_cond_ (it holds mutex_ whenever it is tracking)
    while (max_frames < 0 || tracked < max_frames) {
      bool grab_success = false;
      {
        std::unique_lock<std::mutex> lock(visodo.mutex_, std::try_to_lock);
        if (!lock) { std::this_thread::yield(); continue; }  // tracker busy with the previous frame
        try { grab_success = evaluation->grab(currentIndex, depth, rgb24); }
        catch (const std::exception&) { std::cout << "Exception grabbing" << std::endl; break; }
        if (grab_success) {
          num_failures = 0;
          if (!size_ok()) return 1;
          visodo.depth_.upload(depth.data, depth.step, depth.rows, depth.cols);
          visodo.rgb24_.upload(rgb24.data, rgb24.step, rgb24.rows, rgb24.cols);
          visodo.new_frame_cond_.notify_one();
          ++tracked;
        } else num_failures += 1;
        currentIndex += 1;
        if (num_failures == 10) break;
      }
      if (grab_success) std::this_thread::sleep_for(std::chrono::milliseconds(sleep_ms));
    }
    // the tracker gives mutex_ back only when it waits for the next frame: taking it means the last frame is done
    std::this_thread::sleep_for(std::chrono::milliseconds(sleep_ms));
    std::unique_lock<std::mutex> lock(visodo.mutex_);
  } else {
    while (max_frames < 0 || tracked < max_frames) {
      bool grab_success = false;
      try { grab_success = evaluation->grab(currentIndex, depth, rgb24); }
      catch (const std::exception&) { std::cout << "Exception grabbing" << std::endl; break; }
      if (grab_success) {
        num_failures = 0;
        if (!size_ok()) return 1;
        visodo.depth_.upload(depth.data, depth.step, depth.rows, depth.cols);
        visodo.rgb24_.upload(rgb24.data, rgb24.step, rgb24.rows, rgb24.cols);
        visodo.trackNewFrame();
        ++tracked;
      } else num_failures += 1;
      currentIndex += 1;
      if (num_failures == 10) break;
    }
  }
  evaluation->saveAllPoses(visodo, -1, poses_logfile, misc_logfile);
  evaluation->saveTimeLogFiles(visodo, std::vector<float>(), kf_times_logfile);
  std::cout << "visodo exiting...\n";
  return 0;
}
