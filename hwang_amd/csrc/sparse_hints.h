// Optional side interface a backend may implement next to
// VideoDecoderInterface.  Not in the reference: it lets DecoderAutomata /
// the parallel retriever tell a backend which display-order ordinals of the
// coming interval will actually be fetched, so it can (a) skip unwanted
// non-reference pictures and (b) colour-convert wanted frames only
// (SURVEY.md §7.3).  A backend that ignores it stays correct.
#pragma once

#include "common.h"

#include <cstddef>
#include <cstdint>
#include <vector>

namespace hwang {

class SparseDecodeHints {
 public:
  virtual ~SparseDecodeHints() {}
  // `ordinals` are 0-based output positions counted from the interval's
  // first sample, sorted ascending.  Valid until the next flush().
  virtual void set_wanted_frames(const std::vector<uint64_t>& ordinals) = 0;
  // Optional companion: which samples of the interval (same numbering) are
  // sync samples.  An interval may span several GOPs (the planner merges
  // byte-adjacent ones); with this a backend can stop each GOP behind its last
  // wanted frame instead of only the interval.  Valid until the next flush().
  virtual void set_interval_keyframes(const std::vector<uint64_t>& /*ordinals*/) {}
};

}  // namespace hwang

namespace hwang {

// Optional: the caller says whether this decoder is one of many that it will keep fed from a long queue of work
// items (ParallelRetriever with many items per lane).  A backend may then trade a picture's latency for wider GPU
// launches (GpuExecutor: engines that linger); without it - the reference's callers never say it - every picture goes
// the shortest way.
class BatchingHints {
 public:
  virtual ~BatchingHints() {}
  virtual void set_long_queue(bool long_queue) = 0;
  // True while the next feed() would start by waiting for the GPU (the staging memory its next picture needs is still
  // being read by an earlier picture's kernels).  A thread that drives several decoders in turn feeds another one first.
  // Only a hint: feed() is correct whatever this says.
  virtual bool feed_would_wait() { return false; }
};

// Optional: planar output of the normative decoder result (cropped I420:
// Y w*h, U then V ceil(w/2)*ceil(h/2)), consumed like get_frame().  The
// bit-exact parity tests compare this against the oracle; RGB is derived
// from it by the colour-conversion kernel.
class PlanarOutput {
 public:
  virtual ~PlanarOutput() {}
  virtual Result get_frame_yuv(uint8_t* decoded_buffer, size_t decoded_size) = 0;
};

// Optional: per-instance work counters.
struct BackendCounters {
  int64_t samples_fed = 0;
  int64_t pictures_reconstructed = 0;
  int64_t pictures_skipped = 0;
  int64_t bytes_uploaded = 0;
  int64_t entropy_ns = 0;   // host parse time, GPU stalls excluded
  int64_t stall_ns = 0;     // time the parser was blocked on the GPU (staging ring full)
};
class CountedBackend {
 public:
  virtual ~CountedBackend() {}
  virtual BackendCounters counters() = 0;
  virtual void reset_counters() = 0;
};

}  // namespace hwang
