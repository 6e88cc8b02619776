// Bounded blocking MPMC queue (host plumbing between the entropy/launch side
// of a backend and the frame consumer).  Same role and call names as the
// reference's hwang/util/queue.h:22-58 / queue.inl:20-123 (push / try_pop /
// pop / peek-free size / clear), written around one mutex and two condition
// variables.
#pragma once

#include <condition_variable>
#include <deque>
#include <mutex>

namespace hwang {

template <typename T>
class Queue {
 public:
  explicit Queue(size_t max_size = 4) : max_size_(max_size) {}

  size_t size() {
    std::lock_guard<std::mutex> lk(mutex_);
    return data_.size();
  }

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

  void pop(T& item) {
    std::unique_lock<std::mutex> lk(mutex_);
    not_empty_.wait(lk, [&] { return !data_.empty(); });
    item = std::move(data_.front());
    data_.pop_front();
    lk.unlock();
    not_full_.notify_one();
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
