// Bounded blocking queue between the producing side of a backend (the thread
// that feeds samples: pictures leave the reorder buffer in display order) and
// the consuming side (get_frame / discard_frame).  Same role and call names as
// the reference's hwang/util/queue.h:22-58 / queue.inl:20-123, where it carries
// decoded AVFrames between SoftwareVideoDecoder::feed and get_frame
// (software_video_decoder.h: frame_pool_ / decoded_frame_queue_); here it
// carries surface slots (AMDVideoDecoder) and stamps (the STAMP fixture).
// One mutex, two condition variables; try_peek lets a single consumer inspect
// the head before deciding to take it.
#pragma once

#include <condition_variable>
#include <cstddef>
#include <deque>
#include <mutex>

namespace hwang {

template <typename T>
class Queue {
 public:
  explicit Queue(size_t max_size = 1024) : max_size_(max_size) {}

  size_t size() {
    std::lock_guard<std::mutex> lk(mutex_);
    return data_.size();
  }

  // Blocks while the queue is full.
  void push(T item) {
    std::unique_lock<std::mutex> lk(mutex_);
    not_full_.wait(lk, [&] { return data_.size() < max_size_; });
    data_.push_back(std::move(item));
    lk.unlock();
    not_empty_.notify_one();
  }

  bool try_push(T item) {
    std::unique_lock<std::mutex> lk(mutex_);
    if (data_.size() >= max_size_) return false;
    data_.push_back(std::move(item));
    lk.unlock();
    not_empty_.notify_one();
    return true;
  }

  bool try_pop(T& item) {
    std::unique_lock<std::mutex> lk(mutex_);
    if (data_.empty()) return false;
    item = std::move(data_.front());
    data_.pop_front();
    lk.unlock();
    not_full_.notify_one();
    return true;
  }

  // Blocks while the queue is empty.
  void pop(T& item) {
    std::unique_lock<std::mutex> lk(mutex_);
    not_empty_.wait(lk, [&] { return !data_.empty(); });
    item = std::move(data_.front());
    data_.pop_front();
    lk.unlock();
    not_full_.notify_one();
  }

  // The head without removing it (the reference's peek blocks; this one reports emptiness).
  bool try_peek(T& item) {
    std::lock_guard<std::mutex> lk(mutex_);
    if (data_.empty()) return false;
    item = data_.front();
    return true;
  }

  void clear() {
    std::unique_lock<std::mutex> lk(mutex_);
    data_.clear();
    lk.unlock();
    not_full_.notify_all();
  }

 private:
  const size_t max_size_;
  std::mutex mutex_;
  std::condition_variable not_empty_;
  std::condition_variable not_full_;
  std::deque<T> data_;
};

}  // namespace hwang
