// The backend boundary DecoderAutomata drives.  Signature-identical to the
// reference's hwang/video_decoder_interface.h:26-55 (seven pure virtuals +
// FrameInfo) so the MI355X backend is selectable behind the same vtable.
//
// Contract (taken from decoder_automata.cpp:80-404 and
// software_video_decoder.cpp:103-347):
//  * configure(): may be called repeatedly; extradata = raw avcC payload.
//  * feed(): one MP4 sample (AVCC length-prefixed NALs) per call;
//    (nullptr, 0, false) marks the end of an interval and is followed by
//    flush().  Exactly one output frame per non-empty feed, display order.
//  * get_frame(): copies the oldest buffered frame as packed RGB24
//    (stride width*3) into caller memory; fails if decoded_size is too small;
//    OK + no-op when nothing is buffered.
//  * feed/flush run on the feeder thread concurrently with
//    get_frame/discard_frame/decoded_frames_buffered on the caller thread.
#pragma once

#include "common.h"

#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

namespace hwang {

class VideoDecoderInterface {
 public:
  virtual ~VideoDecoderInterface(){};

  struct FrameInfo {
    uint32_t width;
    uint32_t height;
    std::string format;
  };
  virtual Result configure(const FrameInfo& metadata,
                           const std::vector<uint8_t>& extradata) = 0;

  virtual Result feed(const uint8_t* encoded_buffer, size_t encoded_size,
                      bool keyframe) = 0;

  virtual Result flush() = 0;

  virtual Result discard_frame() = 0;

  virtual Result get_frame(uint8_t* decoded_buffer, size_t decoded_size) = 0;

  virtual int decoded_frames_buffered() = 0;

  virtual Result wait_until_frames_copied() = 0;
};

}  // namespace hwang
