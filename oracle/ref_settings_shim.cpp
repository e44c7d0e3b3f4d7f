// C-ABI shim over the REFERENCE's own RGBID_SLAM::Settings parser (include/settings.h,
// src/settings.cpp, compiled from /root/reference where they lie; see oracle/Makefile).
// Test infrastructure only: lets tests/ compare the product's INI loader with the real one.
#include "settings.h"
#include <cstring>
#include <fstream>
#include <string>

extern "C" int ref_settings_get(const char* path, const char* section, const char* key, char* out, int cap) {
  std::ifstream f(path);
  if (!f.is_open()) return -1;
  RGBID_SLAM::Settings s(f);
  RGBID_SLAM::Section sec;
  if (!s.getSection(section, sec)) return -2;
  RGBID_SLAM::Entry e;
  if (!sec.getEntry(key, e)) return -3;
  std::string v = e.getValue();
  if ((int)v.size() + 1 > cap) return -4;
  std::memcpy(out, v.c_str(), v.size() + 1);
  return (int)v.size();
}
