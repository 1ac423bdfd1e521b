// ini_reader.h -- config.ini reader with the interface of the reference's INIReader
// (visual studio/INIReader.h:305-343): Get / GetInteger / GetReal / GetBoolean / ParseError,
// case-insensitive section and key names, ';' and '#' comment lines, inline " ;" comments,
// "name = value" and "name: value", whitespace-led continuation lines.  The keys the reference
// reads are listed at main.cpp:68-79; the same file works here unchanged.
#pragma once
#include <map>
#include <set>
#include <string>

namespace daisy {

class INIReader {
public:
    INIReader() : error_(-1) {}
    explicit INIReader(const std::string& filename);
    // 0 = ok, line number of the first malformed line, -1 = file could not be opened
    int ParseError() const { return error_; }
    const std::set<std::string>& Sections() const { return sections_; }
    std::string Get(const std::string& section, const std::string& name, const std::string& default_value) const;
    long GetInteger(const std::string& section, const std::string& name, long default_value) const;
    double GetReal(const std::string& section, const std::string& name, double default_value) const;
    bool GetBoolean(const std::string& section, const std::string& name, bool default_value) const;
    // parse from memory (tests)
    static INIReader FromString(const std::string& text);

private:
    void parse(std::istream& in);
    static std::string key(const std::string& section, const std::string& name);
    int error_;
    std::map<std::string, std::string> values_;
    std::set<std::string> sections_;
};

}  // namespace daisy
