// VideoIndex value class + seek planner.
//
// API mirrors the reference's hwang/video_index.h:22-86.  serialize() /
// deserialize() speak the proto3 wire format of
// hwang/hwang_descriptors.proto:5-15 through a hand-rolled varint codec
// (no libprotobuf in this image), so indexes are interchangeable with files
// written by the reference.
#pragma once

#include <cstdint>
#include <string>
#include <tuple>
#include <vector>

namespace hwang {

class VideoIndex {
 public:
  VideoIndex() {}

  VideoIndex(uint32_t timescale, uint64_t duration, uint32_t width,
             uint32_t height, const std::string& format,
             const std::vector<uint64_t>& sample_offsets,
             const std::vector<uint64_t>& sample_sizes,
             const std::vector<uint64_t>& keyframe_indices,
             const std::vector<uint8_t>& metadata)
      : timescale_(timescale),
        duration_(duration),
        frame_width_(width),
        frame_height_(height),
        format_(format),
        num_frames_(sample_sizes.size()),
        sample_offsets_(sample_offsets),
        sample_sizes_(sample_sizes),
        keyframe_indices_(keyframe_indices),
        metadata_bytes_(metadata) {
    check();
  }

  // Invariants every consumer below relies on (the reference used .at() and
  // assertions; here they are checked once): offsets and sizes have one entry
  // per sample, keyframe indices are strictly increasing and < frames(), and
  // an index with samples has at least one keyframe.  deserialize() and
  // hwang_video_index_create() refuse unequal table lengths; an index whose
  // keyframes are inconsistent can still be held and re-serialized (the
  // reference's own wire golden is such an index), but the planner, the
  // automata glue and ParallelRetriever refuse to decode from it.
  bool consistent() const { return problem_.empty(); }
  const std::string& problem() const { return problem_; }

  // Returns false (and leaves *out untouched) on malformed input instead of
  // the reference's silent partial parse.
  static bool deserialize(const uint8_t* data, size_t size, VideoIndex* out);
  static VideoIndex deserialize(const std::vector<uint8_t>& data);

  std::vector<uint8_t> serialize() const;

  const std::vector<uint64_t>& sample_sizes() const { return sample_sizes_; }
  const std::vector<uint64_t>& sample_offsets() const {
    return sample_offsets_;
  }
  const std::vector<uint64_t>& keyframe_indices() const {
    return keyframe_indices_;
  }
  const std::vector<uint8_t>& metadata_bytes() const { return metadata_bytes_; }

  uint32_t timescale() const { return timescale_; }
  uint64_t duration() const { return duration_; }
  double fps() const { return num_frames_ / (duration_ / (double)timescale_); }
  uint32_t frame_width() const { return frame_width_; }
  uint32_t frame_height() const { return frame_height_; }
  const std::string& format() const { return format_; }
  uint64_t frames() const { return num_frames_; }

 private:
  uint32_t timescale_ = 0;
  uint64_t duration_ = 0;
  uint32_t frame_width_ = 0;
  uint32_t frame_height_ = 0;
  std::string format_;
  uint64_t num_frames_ = 0;
  std::vector<uint64_t> sample_offsets_;
  std::vector<uint64_t> sample_sizes_;
  std::vector<uint64_t> keyframe_indices_;
  std::vector<uint8_t> metadata_bytes_;
  std::string problem_;
  void check();
};

struct VideoIntervals {
  std::vector<std::tuple<size_t, size_t>> sample_index_intervals;
  std::vector<std::vector<uint64_t>> valid_frames;
};

// Sorted rows -> keyframe-anchored [start_sample, end_sample) intervals plus
// the wanted rows inside each (reference: video_index.cpp:62-109).  The
// reference asserts on a row >= frames(); here `ok` (if given) is cleared and
// an empty plan is returned.
VideoIntervals slice_into_video_intervals(const VideoIndex& index,
                                          const std::vector<uint64_t>& rows,
                                          bool* ok = nullptr);

}  // namespace hwang
