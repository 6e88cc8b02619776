#include "stamp_video_decoder.h"

#include <cstring>

namespace hwang {

Result StampVideoDecoder::configure(const FrameInfo& metadata,
                                    const std::vector<uint8_t>&) {
  std::lock_guard<std::mutex> lk(m_);
  info_ = metadata;
  ++trace_.configures;
  return Result();
}

Result StampVideoDecoder::feed(const uint8_t* encoded_buffer,
                               size_t encoded_size, bool keyframe) {
  std::lock_guard<std::mutex> lk(m_);
  if (encoded_size == 0) {
    ++trace_.empty_feeds;
    return Result();
  }
  ++trace_.feeds;
  if (keyframe) ++trace_.keyframe_feeds;
  uint32_t stamp = 0;
  std::memcpy(&stamp, encoded_buffer, encoded_size < 4 ? encoded_size : 4);
  stamps_.push(stamp);
  return Result();
}

Result StampVideoDecoder::flush() {
  std::lock_guard<std::mutex> lk(m_);
  ++trace_.flushes;
  return Result();
}

Result StampVideoDecoder::discard_frame() {
  std::lock_guard<std::mutex> lk(m_);
  uint32_t stamp;
  if (stamps_.try_pop(stamp)) ++trace_.discards;
  return Result();
}

Result StampVideoDecoder::get_frame(uint8_t* decoded_buffer,
                                    size_t decoded_size) {
  std::lock_guard<std::mutex> lk(m_);
  uint32_t stamp;
  if (!stamps_.try_peek(stamp)) return Result();
  const size_t need = static_cast<size_t>(info_.width) * info_.height * 3;
  if (decoded_size < need || need < 4) {
    return Result(false, "StampVideoDecoder::get_frame: buffer too small");
  }
  std::memset(decoded_buffer, 0, need);
  std::memcpy(decoded_buffer, &stamp, 4);
  stamps_.try_pop(stamp);
  ++trace_.gets;
  return Result();
}

int StampVideoDecoder::decoded_frames_buffered() {
  return static_cast<int>(stamps_.size());
}

void StampVideoDecoder::set_wanted_frames(const std::vector<uint64_t>&) {
  std::lock_guard<std::mutex> lk(m_);
  ++trace_.hints;
}

StampVideoDecoder::Trace StampVideoDecoder::trace() {
  std::lock_guard<std::mutex> lk(m_);
  return trace_;
}

}  // namespace hwang
