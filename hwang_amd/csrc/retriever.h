// ParallelRetriever: the whole of python/hwang/decoder.py:30-69
// (Decoder.retrieve) in one native call, with the plan's keyframe-anchored
// intervals fanned out over independent lanes - of ONE GPU or of several.
//
// Fan-out (BASELINE.json north_star): every GOP of a request
// that holds a wanted row is one work item; the items of ALL requests of a call
// (retrieve_batch: e.g. BASELINE C3's 128 clips) go into one shared queue,
// longest first; lanes are bound round-robin to the devices the retriever was
// made for and pull from that queue, so the GPUs of a node balance themselves.
// No data ever moves between GPUs or ranks: a lane writes its frames into the
// request's output at the row's position.  With one process per GPU
// (torch.distributed launch) set_shard(rank, world) makes this process take
// only its share of every call's items, chosen by a deterministic
// longest-first split that every rank computes identically.
//
// A lane = one host thread + two backend instances (for the AMD backend: a
// DPB of surfaces in HBM and a pinned staging ring each) used alternately, so
// that the thread is already parsing the next interval - whose keyframe takes
// the long intra-only path on the GPU - while the previous interval's last
// pictures are still being reconstructed.  Intervals are independent
// (closed GOPs) so lanes never exchange data; each writes its
// frames straight into the caller's output at the row's position.  Compared
// with the reference's per-interval loop this
//   * configures a backend once per call, not once per interval
//     (decoder.py:65 re-initializes every time);
//   * stops feeding an interval once its last wanted frame is out;
//   * passes the wanted ordinals down (SparseDecodeHints) so unwanted
//     non-reference pictures are never reconstructed.
#pragma once

#include "common.h"
#include "sparse_hints.h"
#include "video_decoder_factory.h"
#include "video_index.h"

#include <atomic>
#include <memory>
#include <string>
#include <vector>

namespace hwang {

class ParallelRetriever {
 public:
  struct Stats {
    int64_t intervals = 0;
    int64_t samples_fed = 0;
    int64_t pictures_reconstructed = 0;
    int64_t pictures_skipped = 0;
    int64_t frames_returned = 0;
    int64_t bytes_uploaded = 0;
    int64_t entropy_ns = 0;
    int64_t stall_ns = 0;
    int64_t wall_ns = 0;
    int64_t lane_threads = 0;  // threads that drove the lanes of the last call
    // the lane threads' own CPU time, and (HWANG_HOST_COUNTERS=1, where perf_event_open is allowed) what they executed
    int64_t lane_cpu_ns = 0, lane_instructions = 0, lane_cycles = 0, lane_branch_misses = 0;
  };

  // What each device of the retriever did in the last call.
  struct DeviceStats {
    int32_t lanes = 0;
    int64_t items = 0;
    int64_t frames_returned = 0;
    int64_t samples_fed = 0;
  };
  // One clip's worth of a batched call.
  struct Request {
    const VideoIndex* index = nullptr;
    const uint8_t* file = nullptr;  // the whole MP4 in host memory
    size_t file_size = 0;
    const uint64_t* rows = nullptr;  // sorted ascending
    size_t num_rows = 0;
    uint8_t* out = nullptr;          // num_rows frames; host memory, or HBM of any device of this process
    size_t out_capacity = 0;
  };

  // num_lanes <= 0 picks 3 lanes per CPU the process may use (effective_cpus), at most 128
  // (never fewer than one per device).  Lane i decodes on devices[i % devices.size()].
  static int32_t effective_cpus();
  static ParallelRetriever* make(const std::vector<DeviceHandle>& devices, VideoDecoderType type,
                                 int32_t num_lanes);
  static ParallelRetriever* make(DeviceHandle device, VideoDecoderType type,
                                 int32_t num_lanes) {
    return make(std::vector<DeviceHandle>{device}, type, num_lanes);
  }
  ~ParallelRetriever();

  // One process per GPU: this process handles share `rank` of `world` of every
  // following call's work items (rows of other shares are left untouched in
  // the output).  (0, 1) = everything, the default.
  Result set_shard(int32_t rank, int32_t world);

  // rows sorted ascending; out holds rows.size() frames (RGB24 w*h*3 each, or
  // cropped I420 when yuv).  `file` is the whole MP4 in host memory.
  Result retrieve(const VideoIndex& index, const uint8_t* file,
                  size_t file_size, const std::vector<uint64_t>& rows,
                  uint8_t* out, size_t out_capacity, bool yuv);
  // Several clips in one call (BASELINE C3): all their GOP items share one queue.
  Result retrieve_batch(const std::vector<Request>& requests, bool yuv);

  Stats stats() const { return stats_; }
  int32_t num_lanes() const { return static_cast<int32_t>(lanes_.size()); }
  int32_t num_devices() const { return static_cast<int32_t>(devices_.size()); }
  DeviceStats device_stats(int32_t slot) const {
    return slot >= 0 && slot < static_cast<int32_t>(device_stats_.size()) ? device_stats_[slot] : DeviceStats();
  }

 private:
  struct WorkItem {
    uint32_t request = 0;                   // index into the call's requests
    uint64_t start = 0, end = 0;            // sample range of the interval
    const std::vector<uint64_t>* valid = nullptr;
    size_t out_slot = 0;                    // position of valid[0] in the request's rows
    uint64_t cost = 0;
  };
  static constexpr int kBackendsPerLane = 2;
  struct Lane {
    std::unique_ptr<VideoDecoderInterface> backend[kBackendsPerLane];
    // what each backend is configured for (width, height, format, avcC): a lane reconfigures only when
    // the next item's clip differs in any of them
    std::vector<uint8_t> config_key[kBackendsPerLane];
    int32_t device_slot = 0;
    Result result;
    int64_t samples_fed = 0;
    int64_t frames_returned = 0;
    int64_t items = 0;
    int64_t cpu_ns = 0, instructions = 0, cycles = 0, branch_misses = 0;  // of the lane's thread in the last call
  };

  ParallelRetriever() {}
  struct ItemRun;  // a work item on its way through a backend, a sample at a time (retriever.cpp)

  std::vector<DeviceHandle> devices_;
  std::vector<DeviceStats> device_stats_;
  int32_t shard_rank_ = 0, shard_world_ = 1;
  std::vector<Lane> lanes_;
  bool auto_lanes_ = false;  // lane count chosen here, not by the caller: retrieve() may use fewer for large pictures
  Stats stats_;
};

}  // namespace hwang
