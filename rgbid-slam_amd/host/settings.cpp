// settings.cpp -- loader for the tracker's INI files (the class interface of include/rgbid/settings.h is what
// VisodoTracker::loadSettings / loadCalibration consume; the reference's counterpart is include/settings.h + src/settings.cpp).
//
// Written from the file GRAMMAR, as a line tokenizer feeding a small state machine:
//
//   file     := { line '\n' }
//   line     := ws* ( blank | comment | header | assign | cont ) ws*
//   comment  := ('#' | ';') any*
//   header   := '[' name [ ']' any* ]              -- a missing ']' takes the rest of the line as the name
//   assign   := key '=' value                      -- split at the FIRST '='; key and value are stripped of blanks
//   cont     := any+ without '='                   -- continues the value of the last `assign` line
//
// Semantics the shipped configuration files rely on, and that tests/test_cpu_host.py pins against the reference's own parser built into
// oracle/_ref: keys assigned before any header live in the section named ""; a repeated header re-opens the existing section; the FIRST
// assignment of a key wins; an assignment with an empty key is dropped and also closes the open value (continuation lines after it go
// nowhere); a continuation line extends the value of the last assignment with '\n' + the stripped line -- and stores it under that key even
// when the assignment itself lost against an earlier one (the running value of the LAST assignment replaces the stored one).
#include "../../include/rgbid/settings.h"

#include <iostream>

namespace RGBID_SLAM {

std::string trim(std::string src, char const* delims) {
  const std::string::size_type first = src.find_first_not_of(delims);
  if (first == std::string::npos) return std::string();
  const std::string::size_type last = src.find_last_not_of(delims);
  return src.substr(first, last - first + 1);
}

namespace {

enum class LineKind { Blank, Comment, Header, Assignment, Continuation };

struct LineToken {
  LineKind kind = LineKind::Blank;
  std::string head;  // Header: section name; Assignment: key; Continuation: the stripped text
  std::string tail;  // Assignment: value
};

// one physical line -> one token; no state
LineToken tokenize(const std::string& physical_line) {
  LineToken tok;
  const std::string text = trim(physical_line);
  if (text.empty()) return tok;
  const char lead = text.front();
  if (lead == '#' || lead == ';') {
    tok.kind = LineKind::Comment;
    return tok;
  }
  if (lead == '[') {
    tok.kind = LineKind::Header;
    std::string::size_type close = 1;
    while (close < text.size() && text[close] != ']') ++close;
    tok.head = trim(text.substr(1, close - 1));
    return tok;
  }
  std::string::size_type eq = 0;
  while (eq < text.size() && text[eq] != '=') ++eq;
  if (eq == text.size()) {
    tok.kind = LineKind::Continuation;
    tok.head = text;
    return tok;
  }
  tok.kind = LineKind::Assignment;
  tok.head = trim(text.substr(0, eq));
  tok.tail = trim(text.substr(eq + 1));
  return tok;
}

}  // namespace

void Section::addEntry(Entry& new_entry) {
  const bool inserted = entries_.insert(std::make_pair(new_entry.getName(), new_entry)).second;
  if (!inserted) std::cout << "[settings] key '" << new_entry.getName() << "' assigned twice in [" << name_ << "]: the first value is kept" << std::endl;
}

bool Section::getEntry(const std::string& entry_name, Entry& entry) const {
  const auto hit = entries_.find(entry_name);
  if (hit != entries_.end()) entry = hit->second;
  return hit != entries_.end();
}

Settings::Settings(std::ifstream& filestream, bool verbose) : verbose_(verbose) { load(filestream); }

void Settings::load(std::ifstream& filestream) {
  // parser state: the section assignments go to, and the value a continuation line would extend (key empty = nothing open)
  std::string open_section, open_key, open_value;
  for (std::string physical_line; std::getline(filestream, physical_line);) {
    const LineToken tok = tokenize(physical_line);
    switch (tok.kind) {
      case LineKind::Blank:
      case LineKind::Comment:
        break;
      case LineKind::Header: {
        open_section = tok.head;
        Section opened(open_section);
        addSection(opened);
        break;
      }
      case LineKind::Assignment: {
        open_key = tok.head;
        open_value = tok.tail;
        if (open_key.empty()) break;   // "= value": dropped, and nothing stays open
        Entry assigned(open_key, open_value);
        sections_[open_section].addEntry(assigned);
        break;
      }
      case LineKind::Continuation: {
        if (open_key.empty()) break;
        open_value.push_back('\n');
        open_value.append(tok.head);
        sections_[open_section].entries_[open_key].setValue(open_value);
        break;
      }
    }
  }
  if (!verbose_) return;
  for (const auto& sec : sections_) {
    std::cout << sec.second.getName() << std::endl;
    for (const auto& ent : sec.second.entries_) std::cout << "    " << ent.second.getName() << ": " << ent.second.getValue() << std::endl;
  }
}

void Settings::addSection(Section& new_section) {
  const bool inserted = sections_.insert(std::make_pair(new_section.getName(), new_section)).second;
  if (!inserted) std::cout << "[settings] section [" << new_section.getName() << "] opened again: later keys join the existing section" << std::endl;
}

bool Settings::getSection(const std::string& section_name, Section& section) const {
  const auto hit = sections_.find(section_name);
  if (hit != sections_.end()) section = hit->second;
  return hit != sections_.end();
}

}  // namespace RGBID_SLAM
