// The HEVC half of the MI355X backend (hevc_video_decoder.cpp): what AMDVideoDecoder forwards to once configure()
// has seen one of the format names the reference reads as HEVC (impls/software/software_video_decoder.cpp:128-132:
// "h265", "hev1", "hevc"; "hvc1" is the other sample entry type that carries it).
#pragma once

#include "../sparse_hints.h"
#include "../video_decoder_interface.h"

#include <cstdint>
#include <vector>

namespace hwang {

class HevcDelegate {
 public:
  virtual ~HevcDelegate() {}
  virtual Result configure(const VideoDecoderInterface::FrameInfo& info, const std::vector<uint8_t>& extradata) = 0;
  virtual Result feed(const uint8_t* data, size_t size, bool keyframe) = 0;
  virtual Result flush() = 0;
  virtual Result discard_frame() = 0;
  virtual Result get_frame(uint8_t* dst, size_t size) = 0;
  virtual Result get_frame_yuv(uint8_t* dst, size_t size) = 0;
  virtual int decoded_frames_buffered() = 0;
  virtual Result wait_until_frames_copied() = 0;
  virtual void set_wanted_frames(const std::vector<uint64_t>& ordinals) = 0;
  virtual void set_interval_keyframes(const std::vector<uint64_t>& ordinals) = 0;
  virtual BackendCounters counters() = 0;
  virtual void reset_counters() = 0;
  virtual bool feed_would_wait() = 0;   // BatchingHints
};

// Never null; the device is opened at the first configure().
HevcDelegate* make_hevc_delegate(int device_id);

}  // namespace hwang
