// Shadows the reference's util/src/simple_logger.h: silent STDLOG and progress bar.
#pragma once
#include <cstddef>
#include <ostream>
namespace pixsfm {
enum headless { CERR = 2, COUT };
enum typelog { DEBUG = 0, INFO, WARN, ERROR };
class STDLOG {
 public:
  STDLOG() {}
  explicit STDLOG(typelog) {}
  explicit STDLOG(headless) {}
  template <typename T>
  STDLOG& operator<<(const T&) { return *this; }
  STDLOG& operator<<(std::ostream& (*)(std::ostream&)) { return *this; }
};
class LogProgressbar {
 public:
  LogProgressbar() {}
  explicit LogProgressbar(size_t, bool = true, bool = true) {}
  void update(size_t = 1) {}
};
}  // namespace pixsfm
