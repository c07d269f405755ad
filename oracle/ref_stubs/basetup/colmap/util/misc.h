#pragma once
#include <cstring>
#include <set>
#include <string>
#include <vector>
namespace colmap {
template <typename T> bool VectorContainsDuplicateValues(const std::vector<T>& v) { return std::set<T>(v.begin(), v.end()).size() != v.size(); }
inline void PrintHeading2(const std::string&) {}
}  // namespace colmap
