// Stub for the reference's util/src/simple_logger.h (which needs third-party/progressbar.h): base/src/graph.cc only
// streams progress messages into STDLOG(INFO); this one swallows them.
#pragma once
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
}  // namespace pixsfm
