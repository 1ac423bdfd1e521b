#include "ini_reader.h"

#include <algorithm>
#include <cctype>
#include <cstdlib>
#include <cstring>
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
// position of the first character of `stops` in s[from..], or of an inline comment -- a ';' that follows a
// whitespace character seen since `from` -- whichever comes first; s.size() if neither (the scan the reference's
// parser applies to section lines, names and values alike)
size_t scan(const std::string& s, size_t from, const char* stops) {
    bool was_space = false;
    for (size_t i = from; i < s.size(); i++) {
        if (stops && std::strchr(stops, s[i])) return i;
        if (was_space && s[i] == ';') return i;
        was_space = std::isspace((unsigned char)s[i]) != 0;
    }
    return s.size();
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

// Same line grammar as the parser inside the reference's INIReader.h (its compile-time options: multi-line values,
// BOM, inline ';' comments after whitespace, no stop at the first error; names and sections kept to 49 characters where
// that parser copies them into fixed buffers): full-line comments (';' or '#') first, then continuation lines (leading
// whitespace while a name is current), then [section], then name[=:]value.  Not mirrored: its 199-character line buffer.
void INIReader::parse(std::istream& in) {
    std::string line, section, prev_name;
    int lineno = 0;
    auto store = [&](const std::string& name, const std::string& value) {
        std::string& slot = values_[key(section, name)];
        if (!slot.empty()) slot += "\n";                // repeated keys and continuation lines accumulate
        slot += value;
        sections_.insert(section);
    };
    while (std::getline(in, line)) {
        lineno++;
        if (lineno == 1 && line.size() >= 3 && (unsigned char)line[0] == 0xEF && (unsigned char)line[1] == 0xBB &&
            (unsigned char)line[2] == 0xBF)
            line = line.substr(3);                       // UTF-8 byte order mark
        const bool led_by_space = !line.empty() && std::isspace((unsigned char)line[0]);
        std::string s = lstrip(rstrip(line));
        if (s.empty()) continue;
        if (s[0] == ';' || s[0] == '#') continue;
        if (led_by_space && !prev_name.empty()) {       // continuation of the previous name's value
            store(prev_name, rstrip(s.substr(0, scan(s, 0, nullptr))));
            continue;
        }
        if (s[0] == '[') {
            const size_t end = scan(s, 1, "]");
            if (end < s.size() && s[end] == ']') {
                section = s.substr(1, end - 1).substr(0, 49);
                prev_name.clear();
            } else if (!error_) error_ = lineno;
            continue;
        }
        const size_t sep = scan(s, 0, "=:");
        if (sep < s.size() && (s[sep] == '=' || s[sep] == ':')) {
            const std::string name = rstrip(s.substr(0, sep));
            std::string value = lstrip(s.substr(sep + 1));
            value = rstrip(value.substr(0, scan(value, 0, nullptr)));
            prev_name = name.substr(0, 49);
            store(name, value);
        } else if (!error_) error_ = lineno;
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
