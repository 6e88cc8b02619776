// Streaming ISO-BMFF indexer: pull-protocol parser that turns an MP4 file
// into a VideoIndex without ever holding more than one top-level box.
//
// Public surface and protocol mirror the reference's
// hwang/mp4_index_creator.h:23-45 / mp4_index_creator.cpp:30-755:
//   * first chunk handed in is file offset 0 (callers use 1024 bytes);
//   * feed() returns true and (next_offset, next_size) while it needs more,
//     false once done or failed;
//   * ftyp / moov / moof that do not fit in the chunk are re-requested whole,
//     every other top-level box is skipped with a 1024-byte probe at the next
//     box boundary (clamped at EOF).
// The parser itself is a new byte-cursor design (the reference walks a
// bit-at-a-time reader through nested lambdas) and never aborts: malformed
// input sets is_error().
#pragma once

#include "video_index.h"

#include <cstdint>
#include <string>
#include <vector>

namespace hwang {

class MP4IndexCreator {
 public:
  explicit MP4IndexCreator(uint64_t file_size);

  bool feed(const uint8_t* data, size_t size, uint64_t& next_offset,
            uint64_t& next_size);

  VideoIndex get_video_index();

  bool is_done() {
    return done_ || (parsed_ftyp_ && parsed_moov_ && !fragments_present_);
  }
  bool is_error() { return error_; }
  const std::string& error_message() { return error_message_; }
  // Latches an error from outside the parser (the C ABI uses it when an
  // allocation inside feed() throws).
  void abort(const std::string& msg) { fail(msg); }

 private:
  struct TrackDefaults {  // 'trex'
    uint32_t track_id = 0;
    uint32_t sample_size = 0;
    uint32_t sample_flags = 0;
  };

  bool fail(const std::string& msg);
  // Request helpers; both return the value feed() must return.
  bool want_exact(uint64_t offset, uint64_t size, uint64_t& next_offset,
                  uint64_t& next_size);
  bool want_probe(uint64_t offset, uint64_t& next_offset, uint64_t& next_size);

  bool parse_ftyp(const uint8_t* box, uint64_t box_size, uint64_t header);
  bool parse_moov(const uint8_t* box, uint64_t box_size, uint64_t header);
  bool parse_stbl(const uint8_t* body, uint64_t body_size);
  bool parse_moof(const uint8_t* box, uint64_t box_size, uint64_t header,
                  uint64_t moof_file_offset);

  const uint64_t file_size_;
  bool done_ = false;
  bool error_ = false;
  std::string error_message_;

  uint64_t pos_ = 0;  // file offset of the next top-level box

  bool parsed_ftyp_ = false;
  bool parsed_moov_ = false;
  bool fragments_present_ = false;
  std::vector<TrackDefaults> trex_;

  uint32_t timescale_ = 0;
  uint64_t duration_ = 0;
  uint32_t width_ = 0;
  uint32_t height_ = 0;
  std::string format_;
  std::vector<uint64_t> sample_offsets_;
  std::vector<uint64_t> sample_sizes_;
  std::vector<uint64_t> keyframe_indices_;
  std::vector<uint8_t> extradata_;
};

}  // namespace hwang
