#include "decoder_automata.h"

#include "sparse_hints.h"
#include "hevc/hvcc.h"
#include "util/avc_sample.h"
#include "util/trace.h"

#include <chrono>

namespace hwang {

namespace {
constexpr auto kPoll = std::chrono::microseconds(100);
// A short timed wait.  libstdc++'s wait_for goes through pthread_cond_clockwait, which GCC 11's
// ThreadSanitizer does not intercept (it then sees the mutex "locked twice" and invents races on everything
// the lock protects); under TSAN the wait is phrased against the system clock = pthread_cond_timedwait.
inline void wait_poll(std::condition_variable& cv, std::unique_lock<std::mutex>& lk) {
#if defined(__SANITIZE_THREAD__)
  cv.wait_until(lk, std::chrono::system_clock::now() + kPoll);
#else
  cv.wait_for(lk, kPoll);
#endif
}
}

DecoderAutomata::DecoderAutomata(VideoDecoderInterface* decoder)
    : decoder_(decoder) {
  feeder_thread_ = std::thread(&DecoderAutomata::feeder, this);
}

DecoderAutomata* DecoderAutomata::make_instance(DeviceHandle device_handle,
                                                int32_t num_devices,
                                                VideoDecoderType decoder_type) {
  VideoDecoderInterface* decoder = VideoDecoderFactory::make_from_config(
      device_handle, static_cast<uint32_t>(num_devices), decoder_type);
  if (decoder == nullptr) return nullptr;
  return new DecoderAutomata(decoder);
}

DecoderAutomata* DecoderAutomata::make_with_decoder(
    VideoDecoderInterface* decoder) {
  if (decoder == nullptr) return nullptr;
  return new DecoderAutomata(decoder);
}

DecoderAutomata::~DecoderAutomata() {
  {
    std::unique_lock<std::mutex> lk(m_);
    quit_ = true;
  }
  cv_.notify_all();
  feeder_thread_.join();
  if (dirty_) decoder_->flush();
  while (decoder_->decoded_frames_buffered() > 0) {
    if (!decoder_->discard_frame().ok) break;
  }
}

DecoderAutomata::Stats DecoderAutomata::stats() {
  std::unique_lock<std::mutex> lk(m_);
  return stats_;
}

// Drops every frame the backend currently buffers.  Called with the lock
// held; releases it around the backend calls.
void DecoderAutomata::drain_locked(std::unique_lock<std::mutex>& lk) {
  lk.unlock();
  int64_t dropped = 0;
  while (decoder_->decoded_frames_buffered() > 0) {
    if (!decoder_->discard_frame().ok) break;
    ++dropped;
  }
  lk.lock();
  stats_.frames_discarded += dropped;
}

Result DecoderAutomata::initialize(const std::vector<EncodedData>& encoded_data,
                                   const std::vector<uint8_t>& extradata) {
  return initialize(std::vector<EncodedData>(encoded_data), extradata);
}

Result DecoderAutomata::initialize(std::vector<EncodedData>&& encoded_data,
                                   const std::vector<uint8_t>& extradata) {
  if (encoded_data.empty()) {
    return Result(false, "DecoderAutomata::initialize: no encoded data");
  }
  for (const EncodedData& d : encoded_data) {
    if (d.sample_offsets.size() != d.sample_sizes.size()) {
      return Result(false,
                    "DecoderAutomata::initialize: sample_offsets and "
                    "sample_sizes differ in length");
    }
  }
  trace::event("initialize", static_cast<long>(encoded_data.size()));
  std::unique_lock<std::mutex> lk(m_);
  // get_frames() leaves the feeder parked; wait out a backend call that may
  // still be in flight.
  cv_.wait(lk, [&] { return !feeder_busy_; });

  if (dirty_) {
    // An interval that was cut off (an error, or fewer frames taken than planned) is closed here.  What flush() has
    // to say about it - a backend reports a picture that failed late at the flush - concerns the request that is
    // over; the one that starts now configures the backend afresh.
    lk.unlock();
    (void)decoder_->flush();
    lk.lock();
    dirty_ = false;
  }
  drain_locked(lk);

  segments_ = std::move(encoded_data);
  frame_size_ = static_cast<size_t>(segments_[0].width) *
                segments_[0].height * 3;
  seg_complete_.assign(segments_.size(), 0);
  feed_seg_ = 0;
  feed_sample_ = 0;
  next_kf_idx_ = 0;
  flushed_segs_ = 0;
  ret_seg_ = 0;
  ret_valid_ = 0;
  ret_frame_ = segments_[0].keyframes.empty() ? segments_[0].start_keyframe
                                              : segments_[0].keyframes[0];
  failed_ = false;
  failure_ = Result();

  framing_ = sample_framing(hevc::is_hevc_format(segments_[0].format), extradata);
  VideoDecoderInterface::FrameInfo info;
  info.width = segments_[0].width;
  info.height = segments_[0].height;
  info.format = segments_[0].format;
  lk.unlock();
  return decoder_->configure(info, extradata);
}

Result DecoderAutomata::get_frames(uint8_t* buffer, int32_t num_frames) {
  trace::event("get_frames", num_frames);
  std::unique_lock<std::mutex> lk(m_);
  if (segments_.empty()) {
    return Result(false, "DecoderAutomata::get_frames before initialize");
  }
  active_ = true;
  cv_.notify_all();

  // Every way out parks the feeder first.  On an error the frames fetched so far may still be on their way into
  // `buffer` (a device backend copies asynchronously): they land - or are given up - before the caller gets its
  // buffer back, whatever wait_until_frames_copied() itself has to report.
  auto finish = [&](const Result& r) {
    active_ = false;
    cv_.notify_all();
    cv_.wait(lk, [&] { return !feeder_busy_; });
    if (!r.ok) {
      lk.unlock();
      (void)decoder_->wait_until_frames_copied();
      lk.lock();
    }
    return r;
  };

  int32_t got = 0;
  while (got < num_frames) {
    if (failed_) return finish(failure_);
    if (ret_seg_ >= segments_.size()) {
      return finish(Result(
          false, "DecoderAutomata::get_frames: more frames requested than "
                 "the initialized intervals hold"));
    }
    const EncodedData& seg = segments_[ret_seg_];

    if (ret_valid_ >= seg.valid_frames.size()) {
      // Interval finished on our side: let the feeder cut it short, wait for
      // its end-of-interval flush, drop what that flush pushed out, then
      // move both cursors to the next interval.
      seg_complete_[ret_seg_] = 1;
      cv_.notify_all();
      while (flushed_segs_ <= static_cast<int64_t>(ret_seg_) && !failed_) {
        drain_locked(lk);  // keep back-pressure from stalling the feeder
        if (flushed_segs_ > static_cast<int64_t>(ret_seg_)) break;
        wait_poll(cv_, lk);
      }
      if (failed_) return finish(failure_);
      drain_locked(lk);
      ++ret_seg_;
      ret_valid_ = 0;
      if (ret_seg_ < segments_.size()) {
        const EncodedData& next = segments_[ret_seg_];
        ret_frame_ = next.keyframes.empty() ? next.start_keyframe
                                            : next.keyframes[0];
      }
      cv_.notify_all();
      continue;
    }

    if (decoder_->decoded_frames_buffered() > 0) {
      const bool wanted = ret_frame_ == seg.valid_frames[ret_valid_];
      uint8_t* dst = buffer + static_cast<size_t>(got) * frame_size_;
      lk.unlock();
      Result r = wanted ? decoder_->get_frame(dst, frame_size_)
                        : decoder_->discard_frame();
      lk.lock();
      if (!r.ok) return finish(r);
      if (wanted) {
        ++got;
        ++ret_valid_;
        ++stats_.frames_returned;
      } else {
        ++stats_.frames_discarded;
      }
      ++ret_frame_;
      cv_.notify_all();
    } else {
      if (flushed_segs_ > static_cast<int64_t>(ret_seg_)) {
        // Whole interval was fed and flushed, nothing is buffered, and a
        // wanted frame is still outstanding: rows outside the interval,
        // unsorted rows, or a backend that dropped a frame.
        return finish(Result(
            false, "DecoderAutomata::get_frames: interval exhausted before "
                   "frame " + std::to_string(seg.valid_frames[ret_valid_]) +
                   " was produced"));
      }
      wait_poll(cv_, lk);
    }
  }
  Result done = finish(Result());
  lk.unlock();
  HWANG_RETURN_ON_ERROR(decoder_->wait_until_frames_copied());
  trace::event("get_frames returns");
  return done;
}

void DecoderAutomata::feeder() {
  std::unique_lock<std::mutex> lk(m_);
  for (;;) {
    // Work on interval k only once the retriever has moved on to k, so the
    // frames of two intervals never mix in the backend's queue.
    cv_.wait(lk, [&] {
      return quit_ || (active_ && !failed_ && feed_seg_ < segments_.size() &&
                       feed_seg_ <= ret_seg_);
    });
    if (quit_) return;

    const EncodedData& seg = segments_[feed_seg_];
    const size_t nsamples = seg.sample_offsets.size();
    auto latch = [&](const Result& r) {
      failed_ = true;
      failure_ = r;
    };

    if (seg_complete_[feed_seg_] || feed_sample_ >= nsamples) {
      feeder_busy_ = true;
      lk.unlock();
      Result r = decoder_->feed(nullptr, 0, false);
      if (r.ok) r = decoder_->flush();
      lk.lock();
      feeder_busy_ = false;
      dirty_ = false;
      ++stats_.flushes;
      if (!r.ok) latch(r);
      ++feed_seg_;
      feed_sample_ = 0;
      next_kf_idx_ = 0;
      ++flushed_segs_;
      cv_.notify_all();
      continue;
    }

    if (decoder_->decoded_frames_buffered() > MAX_BUFFERED_FRAMES) {
      wait_poll(cv_, lk);
      continue;
    }

    const uint64_t first_frame =
        seg.keyframes.empty() ? seg.start_keyframe : seg.keyframes[0];
    const uint64_t frame = first_frame + feed_sample_;
    const uint64_t off = seg.sample_offsets[feed_sample_];
    const uint64_t size = seg.sample_sizes[feed_sample_];
    if (off > seg.encoded_video.size() ||
        size > seg.encoded_video.size() - off) {
      latch(Result(false, "DecoderAutomata: sample " + std::to_string(frame) +
                              " lies outside encoded_video"));
      cv_.notify_all();
      continue;
    }
    bool keyframe = false;
    if (next_kf_idx_ < seg.keyframes.size() &&
        seg.keyframes[next_kf_idx_] == frame) {
      keyframe = true;
      ++next_kf_idx_;
    }
    const bool first_of_interval = feed_sample_ == 0;
    const uint8_t* ptr = seg.encoded_video.data() + off;

    feeder_busy_ = true;
    lk.unlock();
    if (first_of_interval) {
      if (auto* hints = dynamic_cast<SparseDecodeHints*>(decoder_.get())) {
        std::vector<uint64_t> ordinals;
        ordinals.reserve(seg.valid_frames.size());
        for (uint64_t f : seg.valid_frames)
          if (f >= first_frame) ordinals.push_back(f - first_frame);
        hints->set_wanted_frames(ordinals);
        // seg.keyframes: the interval's sync samples, then its end.  Only those that are IDR pictures bound a closed
        // GOP: behind a recovery-point I picture (open GOP) leading pictures still refer to what came before, so the
        // backend may not cut the decoding short in front of one (the reference decodes an interval continuously).
        std::vector<uint64_t> syncs;
        for (uint64_t k : seg.keyframes) {
          if (k < first_frame || k >= seg.end_keyframe) continue;
          const size_t si = static_cast<size_t>(k - first_frame);
          if (si >= nsamples) continue;
          const uint64_t so = seg.sample_offsets[si], ss = seg.sample_sizes[si];
          if (so > seg.encoded_video.size() || ss > seg.encoded_video.size() - so) continue;
          if (si == 0 || framing_.codec == 0 ||  // (no avcC / hvcC - another codec, a test backend: trust the table)
              sample_starts_closed_gop(seg.encoded_video.data() + so, static_cast<size_t>(ss), framing_))
            syncs.push_back(k - first_frame);
        }
        hints->set_interval_keyframes(syncs);
      }
    }
    Result r = decoder_->feed(ptr, static_cast<size_t>(size), keyframe);
    lk.lock();
    feeder_busy_ = false;
    dirty_ = true;
    ++stats_.samples_fed;
    ++feed_sample_;
    if (!r.ok) latch(r);
    cv_.notify_all();
  }
}

}  // namespace hwang
