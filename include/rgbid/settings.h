// settings.h -- INI settings (include/settings.h + src/settings.cpp of the reference): same classes, same
// parsing rules (';' and '#' comment lines, "[SECTION]", "name = value", continuation lines appended with '\n',
// duplicate entries keep the first value).  Implemented in rgbid-slam_amd/host/settings.cpp; tests compare it
// with the reference's own parser compiled into oracle/_ref.
#pragma once
#include <fstream>
#include <map>
#include <string>

namespace RGBID_SLAM {

std::string trim(std::string src, char const* delims = " \t\r\n");

class Entry {
 public:
  Entry(std::string name = "", std::string value = "") : name_(name), value_(value) {}
  std::string getName() const { return name_; }
  std::string getValue() const { return value_; }
  void setValue(std::string new_value) { value_ = new_value; }

 private:
  std::string name_, value_;
};

class Section {
 public:
  Section(std::string name = "") : name_(name) {}
  void addEntry(Entry& new_entry);
  bool getEntry(const std::string& entry_name, Entry& entry) const;
  std::string getName() const { return name_; }
  std::map<std::string, Entry> entries_;

 private:
  std::string name_;
};

class Settings {
 public:
  Settings(std::ifstream& filestream, bool verbose = true);
  void load(std::ifstream& filestream);
  void addSection(Section& new_section);
  bool getSection(const std::string& section_name, Section& section) const;

 private:
  std::map<std::string, Section> sections_;
  bool verbose_;
};

}  // namespace RGBID_SLAM
