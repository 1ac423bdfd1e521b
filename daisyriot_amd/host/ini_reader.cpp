#include "ini_reader.h"

#include <algorithm>
#include <cctype>
#include <cstdlib>
#include <fstream>
#include <sstream>

namespace daisy {
namespace {

std::string lower(std::string s) {
    std::transform(s.begin(), s.end(), s.begin(), [](unsigned char c) { return (char)std::tolower(c); });
    return s;
}
std::string rstrip(std::string s) {
    while (!s.empty() && std::isspace((unsigned char)s.back())) s.pop_back();
    return s;
}
std::string lstrip(const std::string& s) {
    size_t i = 0;
    while (i < s.size() && std::isspace((unsigned char)s[i])) i++;
    return s.substr(i);
}
// an inline comment starts at a ';' that follows whitespace
std::string cut_inline_comment(const std::string& s) {
    bool prev_space = false;
    for (size_t i = 0; i < s.size(); i++) {
        if (s[i] == ';' && prev_space) return s.substr(0, i);
        prev_space = std::isspace((unsigned char)s[i]) != 0;
    }
    return s;
}

}  // namespace

INIReader::INIReader(const std::string& filename) : error_(0) {
    std::ifstream f(filename.c_str());
    if (!f.is_open()) { error_ = -1; return; }
    parse(f);
}

INIReader INIReader::FromString(const std::string& text) {
    INIReader r;
    r.error_ = 0;
    std::istringstream in(text);
    r.parse(in);
    return r;
}

void INIReader::parse(std::istream& in) {
    std::string line, section, prev_name;
    int lineno = 0;
    bool first = true;
    while (std::getline(in, line)) {
        lineno++;
        if (first) {        // UTF-8 byte order mark
            first = false;
            if (line.size() >= 3 && (unsigned char)line[0] == 0xEF && (unsigned char)line[1] == 0xBB && (unsigned char)line[2] == 0xBF)
                line = line.substr(3);
        }
        const bool led_by_space = !line.empty() && std::isspace((unsigned char)line[0]);
        std::string s = rstrip(lstrip(line));
        if (s.empty() || s[0] == ';' || s[0] == '#') continue;
        if (led_by_space && !prev_name.empty()) {       // continuation of the previous value
            std::string& v = values_[key(section, prev_name)];
            v += "\n";
            v += rstrip(cut_inline_comment(s));
            continue;
        }
        if (s[0] == '[') {
            size_t end = s.find(']');
            if (end == std::string::npos) { if (!error_) error_ = lineno; continue; }
            section = s.substr(1, end - 1);
            sections_.insert(section);
            prev_name.clear();
            continue;
        }
        size_t sep = s.find_first_of("=:");
        if (sep == std::string::npos) { if (!error_) error_ = lineno; continue; }
        std::string name = rstrip(s.substr(0, sep));
        std::string value = rstrip(lstrip(cut_inline_comment(s.substr(sep + 1))));
        std::string& slot = values_[key(section, name)];
        if (!slot.empty()) slot += "\n";                // repeated keys accumulate
        slot += value;
        sections_.insert(section);
        prev_name = name;
    }
}

std::string INIReader::key(const std::string& section, const std::string& name) { return lower(section + "=" + name); }

std::string INIReader::Get(const std::string& section, const std::string& name, const std::string& def) const {
    auto it = values_.find(key(section, name));
    return it == values_.end() ? def : it->second;
}
long INIReader::GetInteger(const std::string& section, const std::string& name, long def) const {
    std::string v = Get(section, name, "");
    char* end = nullptr;
    long n = std::strtol(v.c_str(), &end, 0);       // decimal and 0x.. hex
    return end > v.c_str() ? n : def;
}
double INIReader::GetReal(const std::string& section, const std::string& name, double def) const {
    std::string v = Get(section, name, "");
    char* end = nullptr;
    double n = std::strtod(v.c_str(), &end);
    return end > v.c_str() ? n : def;
}
bool INIReader::GetBoolean(const std::string& section, const std::string& name, bool def) const {
    std::string v = lower(Get(section, name, ""));
    if (v == "true" || v == "yes" || v == "on" || v == "1") return true;
    if (v == "false" || v == "no" || v == "off" || v == "0") return false;
    return def;
}

}  // namespace daisy
