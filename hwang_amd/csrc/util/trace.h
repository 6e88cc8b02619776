// Development aid: HWANG_TRACE=1 prints a time-stamped line per pipeline event of the drop-in path (configure, feed
// of a sync sample, a picture handed to the executor, batch launches / completions, a fetched frame, the copy-out
// barrier) to stderr - microseconds since the first event, thread id - so that one interval's latency can be read
// off as a timeline.  Off: one predictable branch per call site.
#pragma once

#include <chrono>
#include <cstdio>
#include <cstdlib>

namespace hwang {
namespace trace {

inline bool on() {
  static const bool v = getenv("HWANG_TRACE") != nullptr;
  return v;
}
inline double now_us() {
  static const auto t0 = std::chrono::steady_clock::now();
  return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
}
inline void event(const char* what, long a = 0, long b = 0) {
  if (!on()) return;
  fprintf(stderr, "[trace %10.1f us] %-28s %ld %ld\n", now_us(), what, a, b);
}

}  // namespace trace
}  // namespace hwang
