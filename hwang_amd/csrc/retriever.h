// ParallelRetriever: the whole of python/hwang/decoder.py:30-69
// (Decoder.retrieve) in one native call, with the plan's keyframe-anchored
// intervals fanned out over independent lanes.
//
// A lane = one host thread + two backend instances (for the AMD backend: a
// DPB of surfaces in HBM and a pinned staging ring each) used alternately, so
// that the thread is already parsing the next interval - whose keyframe takes
// the long intra-only path on the GPU - while the previous interval's last
// pictures are still being reconstructed.  Intervals are independent
// (closed GOPs) so lanes never exchange data; each writes its
// frames straight into the caller's output at the row's position.  Compared
// with the reference's per-interval loop this
//   * configures a backend once per call, not once per interval
//     (decoder.py:65 re-initializes every time);
//   * stops feeding an interval once its last wanted frame is out;
//   * passes the wanted ordinals down (SparseDecodeHints) so unwanted
//     non-reference pictures are never reconstructed.
#pragma once

#include "common.h"
#include "sparse_hints.h"
#include "video_decoder_factory.h"
#include "video_index.h"

#include <atomic>
#include <memory>
#include <string>
#include <vector>

namespace hwang {

class ParallelRetriever {
 public:
  struct Stats {
    int64_t intervals = 0;
    int64_t samples_fed = 0;
    int64_t pictures_reconstructed = 0;
    int64_t pictures_skipped = 0;
    int64_t frames_returned = 0;
    int64_t bytes_uploaded = 0;
    int64_t entropy_ns = 0;
    int64_t stall_ns = 0;
    int64_t wall_ns = 0;
  };

  // num_lanes <= 0 picks 3 lanes per CPU the process may use (effective_cpus), at most 128.
  static int32_t effective_cpus();
  static ParallelRetriever* make(DeviceHandle device, VideoDecoderType type,
                                 int32_t num_lanes);
  ~ParallelRetriever();

  // rows sorted ascending; out holds rows.size() frames (RGB24 w*h*3 each, or
  // cropped I420 when yuv).  `file` is the whole MP4 in host memory.
  Result retrieve(const VideoIndex& index, const uint8_t* file,
                  size_t file_size, const std::vector<uint64_t>& rows,
                  uint8_t* out, size_t out_capacity, bool yuv);

  Stats stats() const { return stats_; }
  int32_t num_lanes() const { return static_cast<int32_t>(lanes_.size()); }

 private:
  struct WorkItem {
    uint64_t start = 0, end = 0;            // sample range of the interval
    const std::vector<uint64_t>* valid = nullptr;
    size_t out_slot = 0;                    // position of valid[0] in rows
    uint64_t cost = 0;
  };
  static constexpr int kBackendsPerLane = 2;
  struct Lane {
    std::unique_ptr<VideoDecoderInterface> backend[kBackendsPerLane];
    Result result;
    int64_t samples_fed = 0;
    int64_t frames_returned = 0;
  };

  ParallelRetriever() {}
  Result run_item(Lane& lane, VideoDecoderInterface* dec, const WorkItem& item, const VideoIndex& index,
                  const uint8_t* file, size_t file_size, uint8_t* out,
                  size_t frame_bytes, bool yuv);

  std::vector<Lane> lanes_;
  bool auto_lanes_ = false;  // lane count chosen here, not by the caller: retrieve() may use fewer for large pictures
  Stats stats_;
};

}  // namespace hwang
