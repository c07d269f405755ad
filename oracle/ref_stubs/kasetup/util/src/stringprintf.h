#pragma once
#include <string>
inline std::string StringPrintf(const char*, ...) { return std::string(); }
