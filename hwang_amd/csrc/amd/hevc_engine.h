// The submission engine of the HEVC path: ONE thread per GPU puts the work of every HEVC decoder on that GPU into a few
// HIP streams, a batch of pictures per launch (k_hevc.hip: launch_batch).
//
// Why not a stream per decoder (the first version, profiles/r06_hevc_*): HIP folds a process's streams onto
// GPU_MAX_HW_QUEUES hardware queues - 4 by default, 16 with this library loaded first (gpu_executor.cpp asks for them) -
// and a queue is served in order: a picture of decoder A waits behind the copies and the 5 ms intra wavefront of the
// decoder B that happens to share its queue.  Measured with 32 decoders on 16 queues: the GPU ran 1.8 kernels at a time on
// average, was idle 29 % of the time, every kernel's longest run was an intra picture's 11 - 12 ms
// (profiles/r06_hevc_kernel_trace_analysis.txt; tools/experiments/src/stream_concurrency.hip: 4 / 8 / 16 queues run 4 / 8 /
// 15 spinning kernels at once, 32 queues 5.6).  The engine uses five queues on purpose: copies, two for "light" batches
// (pictures with scattered intra blocks; fetched frames' conversion and copy-out), two for "heavy" batches (intra pictures:
// a long, thin wavefront that should not hold anything up).  How many of each no longer matters: while the intra kernels
// handed over with agent-scope fences (whole-L2 write-backs and invalidations) more streams only made every kernel slower;
// since they do not, one to four heavy streams measure the same (profiles/r06_hevc_fence_free_sweeps.txt).
//
// Order: a decoder's jobs are launched in the order it submitted them.  Inside a stream that is stream order; when a
// job's predecessor went to another stream the job waits (in the engine's list, not in the stream: nothing queues up
// behind it) until the predecessor's batch is through.  Batches are formed from whatever is eligible whenever a stream has
// fewer than two batches in flight - while they run, the next one fills up.
#pragma once

#include "hevc_frame_job.h"

#include <hip/hip_runtime.h>

#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <deque>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace hwang {
namespace amd {
namespace hevc {

struct EngineJob {
  enum Kind { kPicture, kRgb, kYuv };
  Kind kind = kPicture;
  // kPicture
  PicParams pp;                       // host copy: the grids
  const PicParams* pp_dev = nullptr;
  struct Copy {
    void* dst;
    const void* src;
    size_t bytes;
  } copy[3] = {{nullptr, nullptr, 0}, {nullptr, nullptr, 0}, {nullptr, nullptr, 0}};  // host -> device, before the kernels
  int16_t* clear_resid = nullptr;     // residual planes to zero first (after a faulted picture)
  size_t clear_resid_bytes = 0;
  // kRgb / kYuv: a fetched frame.  y/u/v: the cropped picture in its surface; `out`: device, page-locked or bounce memory
  const uint8_t* y = nullptr;
  const uint8_t* u = nullptr;
  const uint8_t* v = nullptr;
  int y_pitch = 0, width = 0, height = 0;
  uint8_t* rgb_dev = nullptr;         // kRgb with a host destination: converted here, copied to `out`
  uint8_t* out = nullptr;
  bool out_on_device = false;
  // completion
  hipEvent_t done = nullptr;          // owned by whoever owns the job; recorded behind the job's work
  std::atomic<int> launched{0};       // 1: `done` has been recorded (or the job failed: `failed`)
  bool failed = false;
};

class Engine {
 public:
  // The engine of `device` (created with its first user, gone with its last).
  static std::shared_ptr<Engine> for_device(int device);
  ~Engine();

  int add_client();                       // a decoder: its jobs keep their order
  void remove_client(int client);         // (after drain)
  void submit(int client, EngineJob* job);
  // Blocks until `job` / every job of `client` has been put into a stream; false: the launch failed (error()).
  bool wait_launched(EngineJob* job);
  bool drain(int client);
  // The client has waited for everything it submitted (drain + its jobs' events) and may destroy those events.
  void settled(int client);
  std::string error();

 private:
  explicit Engine(int device);
  void run();
  struct Client {
    bool used = false;
    std::deque<EngineJob*> queue;
    int last_stream = -1;                 // where its most recent job went
    uint64_t last_seq = 0;                // ... and which batch of that stream it was in
    bool taking = false;                  // jobs of it are between the list and a stream
  };
  // streams 0 .. kLightStreams - 1 take light batches, the rest heavy ones
  static constexpr int kLightStreams = 2, kHeavyStreams = 4, kStreams = kLightStreams + kHeavyStreams;  // (at most; two heavy ones are used)
  const int device_;
  std::mutex mu_;
  std::condition_variable cv_work_, cv_done_;
  std::vector<Client> clients_;
  bool stop_ = false;
  std::string error_;
  hipStream_t copy_stream_ = nullptr, stream_[kStreams] = {};
  std::vector<hipEvent_t> event_pool_;
  std::deque<hipEvent_t> inflight_[kStreams];    // a batch event each (the engine thread's own)
  uint64_t launched_seq_[kStreams] = {}, completed_seq_[kStreams] = {};  // batches put into / through each stream
  std::deque<hipEvent_t> copy_inflight_;  // the copy stream's events, kept until they have fired
  std::thread thread_;
};

}  // namespace hevc
}  // namespace amd
}  // namespace hwang
