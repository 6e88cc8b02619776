// TEST FIXTURE, not a decoder: emits one "frame" per fed sample whose first
// four bytes are the sample's first four payload bytes (rest zero).  Lets the
// DecoderAutomata / retriever protocol be pinned on machines without a GPU,
// the same way SURVEY.md Appendix B-3 drove the reference automata with a
// counting fake backend.  Selected only by VideoDecoderType::STAMP.
#pragma once

#include "sparse_hints.h"
#include "util/queue.h"
#include "video_decoder_interface.h"

#include <mutex>

namespace hwang {

class StampVideoDecoder : public VideoDecoderInterface, public SparseDecodeHints {
 public:
  struct Trace {
    int configures = 0, feeds = 0, keyframe_feeds = 0, empty_feeds = 0,
        flushes = 0, gets = 0, discards = 0, hints = 0;
  };

  Result configure(const FrameInfo& metadata,
                   const std::vector<uint8_t>& extradata) override;
  Result feed(const uint8_t* encoded_buffer, size_t encoded_size,
              bool keyframe) override;
  Result flush() override;
  Result discard_frame() override;
  Result get_frame(uint8_t* decoded_buffer, size_t decoded_size) override;
  int decoded_frames_buffered() override;
  Result wait_until_frames_copied() override { return Result(); }
  void set_wanted_frames(const std::vector<uint64_t>& ordinals) override;

  Trace trace();

 private:
  std::mutex m_;
  FrameInfo info_{};
  Queue<uint32_t> stamps_{1 << 20};  // one entry per fed sample, consumed by get_frame / discard_frame
  Trace trace_;
};

}  // namespace hwang
