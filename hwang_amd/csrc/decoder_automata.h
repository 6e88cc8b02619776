// DecoderAutomata: feeds the MP4 samples of keyframe-anchored intervals into
// a VideoDecoderInterface backend on a feeder thread and hands the wanted
// frames to the caller.
//
// Public API mirrors the reference's hwang/decoder_automata.h:38-70
// (make_instance / EncodedData / initialize / get_frames).  The state
// machine is a new design: one mutex + condition variable instead of the
// reference's atomics-and-yield loops (decoder_automata.cpp:120-404), and two
// behavioural improvements that callers cannot observe in the returned
// frames:
//   * an interval stops being fed as soon as its last wanted frame has been
//     retrieved (the reference feeds every sample of every interval but the
//     last, decoder_automata.cpp:287,310);
//   * no per-packet debug print (decoder_automata.cpp:366).
// Kept: one configure() per initialize(); one feed() per sample with
// keyframe=true on the interval's keyframes; (nullptr,0,false) + flush() at
// the end of every interval; back-pressure at 8 buffered frames; feeder
// errors are latched and returned from get_frames().
#pragma once

#include "util/avc_sample.h"

#include "video_decoder_factory.h"
#include "video_decoder_interface.h"

#include <condition_variable>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace hwang {

class DecoderAutomata {
  DecoderAutomata() = delete;
  DecoderAutomata(const DecoderAutomata&) = delete;

 public:
  static DecoderAutomata* make_instance(DeviceHandle device_handle,
                                        int32_t num_devices,
                                        VideoDecoderType decoder_type);
  // Adopts `decoder` (takes ownership).  Seam used by tests and by callers
  // that construct a backend themselves.
  static DecoderAutomata* make_with_decoder(VideoDecoderInterface* decoder);
  ~DecoderAutomata();

  struct EncodedData {
    bool operator==(const EncodedData& o) const {
      return encoded_video == o.encoded_video && width == o.width &&
             height == o.height && start_keyframe == o.start_keyframe &&
             end_keyframe == o.end_keyframe && format == o.format &&
             sample_offsets == o.sample_offsets &&
             sample_sizes == o.sample_sizes && keyframes == o.keyframes &&
             valid_frames == o.valid_frames;
    }

    std::vector<uint8_t> encoded_video;
    uint32_t width = 0;
    uint32_t height = 0;
    uint64_t start_keyframe = 0;
    uint64_t end_keyframe = 0;
    std::string format;
    std::vector<uint64_t> sample_offsets;
    std::vector<uint64_t> sample_sizes;
    std::vector<uint64_t> keyframes;
    std::vector<uint64_t> valid_frames;
  };

  Result initialize(const std::vector<EncodedData>& encoded_data,
                    const std::vector<uint8_t>& extradata);
  // The same, taking the intervals over instead of copying them (not in the reference).
  Result initialize(std::vector<EncodedData>&& encoded_data,
                    const std::vector<uint8_t>& extradata);

  Result get_frames(uint8_t* buffer, int32_t num_frames);

  // Counters (the reference's removed profiler intervals frames_fed /
  // frames_decoded / frames_used, decoder_automata.cpp:245-249).
  struct Stats {
    int64_t samples_fed = 0;
    int64_t frames_discarded = 0;
    int64_t frames_returned = 0;
    int64_t flushes = 0;
  };
  Stats stats();

 private:
  explicit DecoderAutomata(VideoDecoderInterface* decoder);
  void feeder();
  void drain_locked(std::unique_lock<std::mutex>& lk);

  static constexpr int32_t MAX_BUFFERED_FRAMES = 8;

  std::unique_ptr<VideoDecoderInterface> decoder_;
  std::thread feeder_thread_;

  std::mutex m_;
  std::condition_variable cv_;
  bool quit_ = false;
  bool active_ = false;        // a get_frames() call is in progress
  bool feeder_busy_ = false;   // feeder is inside a backend call
  bool dirty_ = false;         // something was fed since the last flush
  std::vector<EncodedData> segments_;
  size_t frame_size_ = 0;
  SampleFraming framing_;      // from the avcC / hvcC handed to initialize(): to tell sync samples that start a closed GOP from recovery points

  // feeder cursor
  size_t feed_seg_ = 0;
  size_t feed_sample_ = 0;
  size_t next_kf_idx_ = 0;
  int64_t flushed_segs_ = 0;   // number of segments fully fed + flushed
  // retriever cursor
  size_t ret_seg_ = 0;
  size_t ret_valid_ = 0;
  uint64_t ret_frame_ = 0;
  std::vector<char> seg_complete_;  // retriever got every wanted frame

  bool failed_ = false;
  Result failure_;
  Stats stats_;
};

}  // namespace hwang
