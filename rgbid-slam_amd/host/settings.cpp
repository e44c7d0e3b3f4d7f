// settings.cpp -- INI parser with the reference's semantics (src/settings.cpp:25-170): see include/rgbid/settings.h.
#include "../../include/rgbid/settings.h"

#include <iostream>

namespace RGBID_SLAM {

std::string trim(std::string src, char const* delims) {
  // include/settings.h:32-46
  std::string res(src);
  std::string::size_type index = res.find_last_not_of(delims);
  if (index != std::string::npos) res.erase(++index);
  index = res.find_first_not_of(delims);
  if (index != std::string::npos) res.erase(0, index);
  else res.erase();
  return res;
}

void Section::addEntry(Entry& new_entry) {
  if (entries_.find(new_entry.getName()) != entries_.end()) {
    std::cout << "Warning: entry " << new_entry.getName() << " is already loaded." << std::endl;
    return;  // the first value wins
  }
  entries_[new_entry.getName()] = new_entry;
}

bool Section::getEntry(const std::string& entry_name, Entry& entry) const {
  std::map<std::string, Entry>::const_iterator it = entries_.find(entry_name);
  if (it == entries_.end()) return false;
  entry = it->second;
  return true;
}

Settings::Settings(std::ifstream& filestream, bool verbose) : verbose_(verbose) { load(filestream); }

void Settings::load(std::ifstream& filestream) {
  std::string entry_name, entry_value, section_name, line;
  while (std::getline(filestream, line)) {
    line = trim(line);
    if (!line.length()) continue;
    if (line[0] == '#' || line[0] == ';') continue;
    if (line[0] == '[') {
      section_name = trim(line.substr(1, line.find(']') - 1));
      Section new_section(section_name);
      addSection(new_section);
      continue;
    }
    std::string::size_type pos_equal = line.find('=');
    if (pos_equal != std::string::npos) {
      entry_name = trim(line.substr(0, pos_equal));
      entry_value = trim(line.substr(pos_equal + 1));
      if (!entry_name.empty()) {
        Entry new_entry(entry_name, entry_value);
        sections_[section_name].addEntry(new_entry);
      }
    } else if (!entry_name.empty()) {
      // continuation line: appended to the running value and stored (src/settings.cpp:117-124)
      entry_value += '\n';
      entry_value += trim(line);
      sections_[section_name].entries_[entry_name].setValue(entry_value);
    }
  }
  if (verbose_)
    for (std::map<std::string, Section>::iterator it = sections_.begin(); it != sections_.end(); ++it) {
      std::cout << it->second.getName() << std::endl;
      for (std::map<std::string, Entry>::iterator e = it->second.entries_.begin(); e != it->second.entries_.end(); ++e)
        std::cout << "    " << e->second.getName() << ": " << e->second.getValue() << std::endl;
    }
}

void Settings::addSection(Section& new_section) {
  if (sections_.find(new_section.getName()) != sections_.end()) {
    std::cout << "Warning: section " << new_section.getName() << " is already loaded." << std::endl;
    return;
  }
  sections_[new_section.getName()] = new_section;
}

bool Settings::getSection(const std::string& section_name, Section& section) const {
  std::map<std::string, Section>::const_iterator it = sections_.find(section_name);
  if (it == sections_.end()) return false;
  section = it->second;
  return true;
}

}  // namespace RGBID_SLAM
