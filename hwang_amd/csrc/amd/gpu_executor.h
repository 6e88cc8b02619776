// Per-device batching executor.
//
// Sparse decoding is parallel across keyframe-anchored intervals ("lanes":
// one host thread + one DPB each) far more than inside a picture, whose
// vector-prediction, intra and deblocking stages are raster-order pipelines
// that keep a CU per four macroblock rows busy.  To fill 256 CUs the GPU
// therefore has to work on many lanes' pictures at the same time.  Doing that with one HIP stream per lane runs into the
// hardware-queue limit and into per-call runtime overhead, so lanes only
// *describe* work here and a single submitter thread per GPU turns whatever
// is pending into a few large launches:
//
//     H2D of the new pictures' records            (copy engine)
//     k_mv       grid (pictures x bands)          row pipeline, 4 rows per workgroup
//     k_inter    grid (max_mbs, pictures)         every inter MB of every picture
//     k_intra    grid (pictures x bands)          row pipeline
//     k_bs       grid (max_mbs/8, pictures)
//     k_deblock  grid (pictures x bands)          row pipeline
//     convert    grid (.., .., frames)            I420 -> RGB24 / plane copies
//     D2H of frames that go to host memory
//
// A batch takes at most one picture per lane (a lane's pictures depend on
// each other) plus the conversions queued behind it.
//
// Several engines run side by side, each with its own submitter thread,
// compute stream and ring of in-flight batches:
//   light      P/B pictures; batches are formed late (when the previous one has
//              completed) so that they are wide
//   heavy      intra-only pictures: every macroblock goes through the intra
//              and deblocking row pipelines, ~2 ms per 1080p picture.  A
//              heavy batch is latency bound and small (17 CUs a picture); it
//              launches eagerly beside the light batches (up to kHeavyEngines
//              such engines exist, lane % H picks one; one is the default);
//              a lane waits for its keyframe before its first P picture can
//              go anywhere, which makes this latency the critical path of
//              every interval.
// A kernel launch lasts as long as its slowest workgroup, so one I picture in
// a batch of P pictures would hold the whole batch (and every lane in it) for
// its own duration; with two engines the P/B batches keep their short period.
// Ordering across engines: a lane's next item is only taken by an engine when
// everything the lane launched before has either completed or was launched
// on that same engine (stream order).  Uploads (k_table, k_upload) are the first
// kernels of a batch on the same stream; an optional copy stream is described
// in GpuExecutor::start.
#pragma once

#include "frame_job.h"

#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <deque>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace hwang {
namespace amd {

struct PicSubmit {
  JobDesc desc;  // device pointers
  struct Copy {
    void* dst;
    const void* src;
    size_t bytes;
  } copies[kJobUploads];
  int ncopies = 0;
  int num_mbs = 0;
  int heavy = 0;  // engine: 1 for intra-only pictures
  int num_intra_mbs = 0;
  // surfaces the picture predicts from (chain launches: a later picture of the same launch may not reuse them)
  int16_t ref_slots[32];
  int num_ref_slots = 0;
  uint64_t bytes_inter = 0, bytes_intra = 0, bytes_deblock = 0, bytes_mv = 0;  // algorithmic, for the roofline
};

struct ConvSubmit {
  ConvertDesc desc;
  void* d2h_dst = nullptr;  // pinned host target, or null when desc.dst is the final destination
  size_t d2h_bytes = 0;
  uint64_t bytes = 0;
};

class GpuExecutor {
 public:
  // One executor per device for the life of the process.
  static GpuExecutor* get(int device, std::string* error);

  // light_group: which of the 1 + kExtraLightEngines light engines runs the lane's P / B pictures (taken modulo).
  // chain: the lane's pictures - each predicted from the one before - travel SEVERAL to a batch and run as one chain
  // launch (k_chain: picture n + 1 trails picture n by a few macroblock rows), intra-only pictures included.  For a
  // decoder that is served one interval at a time (the reference's DecoderAutomata): latency is all it has.  Lanes
  // of a caller that keeps dozens of decoders busy (ParallelRetriever) leave it off: one picture per lane and batch,
  // in launches as wide as the lanes make them.
  int register_lane(int light_group = 0, bool chain = false);
  void unregister_lane(int lane);
  // The lane's owner will keep it fed from a long queue (BatchingHints): the engines may linger for wider launches
  // while every busy lane is such a lane.
  void set_lane_patient(int lane, bool patient);
  // Both return the lane-local sequence number of the submission (1, 2, ...).
  uint64_t submit_picture(int lane, const PicSubmit& job);
  uint64_t submit_convert(int lane, const ConvSubmit& job);
  // True once the executor has been shut down (process exit): decoders that are
  // destroyed later must not touch the HIP runtime any more.
  bool stopped();
  // How many of the lane's submissions have finished (non-blocking).
  uint64_t completed(int lane);
  void completed(const int* lanes, int n, uint64_t* out);  // several lanes under one lock
  // Blocks until the lane's submissions up to `seq` have finished on the GPU.
  bool wait(int lane, uint64_t seq, std::string* error);

  struct Stats {
    uint64_t batches = 0, pictures = 0, conversions = 0;
    uint64_t chain_retries = 0;  // chain launches that timed out and were run again through the wide kernels
  };
  Stats stats();

  // ---- Replay: the GPU with the host out of the way (bench.py's `replay` block, profiles/rNN_replay.json).
  // keep_recent(n) makes the executor remember the last n pictures submitted (their descriptors: the records and
  // surfaces they name stay where they are as long as their decoders live and sit idle).  replay() then launches the
  // sample-stage kernels - and the records-only ones - for `width` of those pictures per launch, `launches` times, back
  // to back on one stream, every launch with fresh epochs: what the kernels deliver at a launch width the host's
  // entropy decoders never produce.  The pictures are decoded in place again and again, so what ends up in the surfaces
  // is whatever the last launch left (pictures that predict from each other are in the same launch): a measurement,
  // not a decode - call it when nothing else is submitted and before the decoders are used again for real only after
  // they have been re-configured (a new interval starts from a keyframe anyway).
  void keep_recent(int pictures);
  // Decoders report frames that are decoded (or on their way) and not yet fetched or discarded: replay() decodes in
  // place over the decoders' surfaces and refuses to while anybody still has a frame to collect from them.
  void note_undelivered(int delta) { undelivered_.fetch_add(delta, std::memory_order_acq_rel); }
  struct ReplayResult {
    int pictures_available = 0;  // distinct pictures the executor still holds descriptors for (P / B pictures)
    int width = 0;               // pictures per launch actually used
    int launches = 0;
    double seconds = 0;          // wall time of the launches (HIP events)
  };
  bool replay(int width, int launches, bool chain, ReplayResult* out, std::string* error);

 private:
  explicit GpuExecutor(int device) : device_(device) {}
  bool start(std::string* error);
  void run(int engine);
  void run_pre();
  void stop();

  friend void shutdown_executors();

  struct Item {
    bool is_picture;
    int engine = 0;  // pictures: the engine that must run it
    int pre = 2;     // pictures: the records-only stage: 0 waiting, 1 in flight, 2 done (or not used)
    std::chrono::steady_clock::time_point t_submit;  // pictures: when the lane handed it over (batch formation: linger)
    PicSubmit pic;
    ConvSubmit conv;
  };
  struct Lane {
    bool active = false;
    std::deque<Item> fifo;
    uint64_t submitted = 0, launched = 0, completed = 0;
    std::chrono::steady_clock::time_point t_last_submit{};  // the lane's latest picture (run(): which lanes are at work)
    int last_engine = 0;
    int light_engine = 0;  // engine of the lane's P / B pictures
    bool chain = false;    // see register_lane
    bool patient = false;  // see set_lane_patient
    int waiters = 0;  // threads blocked in wait()
    // A completed batch wakes exactly the lanes it contained (one shared
    // condition variable woke every blocked lane thread for every batch: 10 %
    // of the process's CPU time went into futex traffic).
    std::condition_variable cv;
  };
  struct Batch {
    std::vector<std::pair<int, int>> lane_counts;  // (lane, items taken)
    bool in_flight = false;
    GpuExecutor* owner = nullptr;
    int engine = 0;
    uint32_t ticket = 0;              // value k_done stores into *flag when the batch is through
    volatile uint32_t* flag = nullptr;  // pinned host memory
    volatile uint32_t* fault = nullptr; // pinned host memory: FAULT_* code stored by a row kernel whose poll expired
    std::vector<Item*> pre_items;       // a batch of the records-only stage: the items it covers (they stay in their
                                        // lanes' queues - a deque keeps references valid - until the sample stage took them)
    // A chain launch whose polls expired (FAULT_CHAIN_*: nothing in the launch moved for the polls' bound - a device
    // shared with something that starves it, or a lost hand-off) is not the end of the executor: its pictures'
    // records are intact and the wide kernels need no hand-off between pictures, so the engine runs the batch once
    // more through them, picture by picture (complete() arms it, run() does it; a fault in THAT run is final).
    bool chain = false;                 // launched as a chain
    bool retried = false;               // ... and already run a second time
    bool retry_pending = false;         // (mu_) complete() asks the engine thread for the second run
    int slot = 0;                       // the engine's table set the batch was built in
    std::vector<int> retry_mbs;         // per picture of the table: macroblocks
    std::vector<ConvSubmit> retry_convs;  // the conversions that followed the pictures
  };

  // End of a batch: credits the lanes and wakes waiters.  Detected by the
  // poller thread, which watches the tickets that k_done stores into pinned
  // host memory (default), or by a stream callback (hipLaunchHostFunc,
  // HWANG_COMPLETION=hostfunc) on a HIP runtime thread.  The poller costs a
  // fraction of one core; the runtime's callback machinery cost ~8 % of the
  // process's CPU time, which the entropy decoders need.
  static void on_batch_done(void* batch);
  void complete(Batch* b, hipError_t status);
  static const char* fault_name(uint32_t code);
  void poll_loop();

  static constexpr int kInflight = 3;
  static constexpr int kMaxJobs = 512;
  static constexpr int kHeavyEngines = 3;
  // Further light engines (streams of their own, on the light share of the CUs) for lanes that ask for one at
  // registration.  A batch runs the row pipelines of its pictures one after the other (vectors, intra, deblocking:
  // three latency-bound kernels) and occupies 17 CUs per picture while it does; a decoder with entropy workers has
  // some eight lanes with work at a time, so one batch leaves most of the GPU idle for ~3 ms.  Spread over several
  // engines its lanes' batches overlap.  The retriever's lanes all use engine 0 (48 lanes: one wide batch is best).
  static constexpr int kExtraLightEngines = 3;
  // One more engine runs the stage of a picture that needs no samples (GpuExecutor::run_pre): upload of the records,
  // k_mv, k_bs, k_residual - for every picture that has been submitted, whatever its place in its lane's queue.  The
  // engines above then run k_inter -> k_intra -> k_deblock only, the chain that really ties a picture to the one
  // before it: 1.5 ms per dependent picture instead of 2.4 ms.  HWANG_PRESTAGE=0 puts all kernels back into one batch.
  static constexpr int kPreEngine = 1 + kHeavyEngines + kExtraLightEngines;
  static constexpr int kEngines = kPreEngine + 1;
  static bool is_light(int engine) { return engine == 0 || (engine > kHeavyEngines && engine < kPreEngine); }

  struct Engine {
    std::thread thread;
    int inflight = 0;  // batches launched and not yet completed (mu_)
    hipStream_t stream = nullptr, copy_stream = nullptr;
    Batch batches[kInflight];
    hipEvent_t copied[kInflight] = {nullptr, nullptr, nullptr};
    JobDesc* job_table_host[kInflight] = {nullptr, nullptr, nullptr};
    JobDesc* job_table_dev[kInflight] = {nullptr, nullptr, nullptr};
    ConvertDesc* conv_table_host[kInflight] = {nullptr, nullptr, nullptr};
    ConvertDesc* conv_table_dev[kInflight] = {nullptr, nullptr, nullptr};
    // chain launches: [0 .. 3] the two ticket counters (zero), then the row units, then the inter units (ChainArgs)
    uint32_t* unit_table_host[kInflight] = {nullptr, nullptr, nullptr};
    uint32_t* unit_table_dev[kInflight] = {nullptr, nullptr, nullptr};
  };
  static constexpr size_t kUnitWords = size_t(1) << 18;  // per table: 1 MB, ~250 1080p pictures' worth of units
  void rerun_wide(Engine& en, Batch& batch);

  // May engine `e` take the head of this lane's queue?  (mu_ held)
  bool eligible(const Lane& l, int e) const {
    if (l.fifo.empty()) return false;
    if (l.launched != l.completed && l.last_engine != e) return false;
    const Item& it = l.fifo.front();
    if (it.is_picture) return it.engine == e && it.pre == 2;
    return l.launched == l.completed ? e == l.last_engine : true;
  }

  const int device_;
  std::mutex mu_;
  std::condition_variable cv_work_[kEngines];  // one per engine thread
  bool stop_ = false;
  std::string error_;
  std::deque<Lane> lanes_;  // stable addresses: lanes hold a condition variable
  Engine engines_[kEngines];
  std::thread poller_;
  std::condition_variable cv_poll_;
  uint32_t* done_flags_ = nullptr;  // pinned: [kEngines][kInflight]
  uint32_t* fault_flags_ = nullptr; // pinned: [kEngines][kInflight]
  // How much runs at once (interleaved A/B runs on the C1 probe, 16 CPUs):
  //  * one heavy engine instead of three: same rate and CPU time, but the
  //    intra-only pictures travel in half as many, twice as wide launches that
  //    each finish sooner (k_residual + k_intra span 2.3 -> 1.6 ms);
  //  * every engine forms its next batch only when the previous one has
  //    completed (1 in flight instead of 2-3): the widest possible launches,
  //    the fewest of them, nothing of the same engine competing with itself -
  //    +4 % (heavy) and +3 % (light) returned frames/s at lower CPU time; the
  //    4K configurations do not change.
  // HWANG_HEAVY_ENGINES / HWANG_HEAVY_INFLIGHT / HWANG_LIGHT_INFLIGHT override.
  bool no_heavy_ = false;
  // A light engine that serves many lanes forms its next batch when `linger_fill_` per cent of its active lanes have a
  // picture ready - or the oldest of them has waited `linger_us_` (0: as soon as the previous batch is through).
  // (round 6, after the share was taken of the lanes at work instead of all registered ones - the test could not be met
  // before: `value` is flat from 0 to 12 ms of age limit, 984 - 1 000 frames/s, because the 16 CPUs' parsers bound it;
  // what changes is the width of the launches: 587 launches of 777 us -> 440 of 813 us per step, the deblocking kernel's
  // achieved bandwidth 0.019 -> 0.024 of peak.  24 ms costs 20 % of `value` with 48 lanes.  profiles/r06_linger_sweep*.txt)
  int linger_us_ = 12000, linger_fill_ = 100;
  int linger_idle_us_ = 1500;  // ... and no engine lingers once nothing has joined its batch for this long
  // ... a heavy engine when `heavy_linger_pictures_` keyframes are ready or the oldest has waited `heavy_linger_us_`, the
  // records-only stage when `pre_linger_pictures_` pictures wait or the oldest has waited `pre_linger_us_` (keyframes
  // never wait there)
  int heavy_linger_us_ = 6000, heavy_linger_pictures_ = 8;
  int pre_linger_us_ = 2500, pre_linger_pictures_ = 32;
  int heavy_engines_ = 1;
  int heavy_inflight_ = 1;
  int light_inflight_ = 1;
  int light_engines_ = 4;  // light engines that lanes are dealt to (HWANG_LIGHT_ENGINES, up to 1 + kExtraLightEngines)
  uint32_t next_ticket_ = 0;
  int total_inflight_ = 0;
  bool use_hostfunc_ = false;
  bool prestage_ = true;
  std::atomic<int64_t> undelivered_{0};
  int keep_recent_ = 0;                 // (mu_)
  std::deque<PicSubmit> recent_;        // (mu_) the last keep_recent_ pictures submitted
  int chain_max_lanes_ = 2;  // chain launches only while at most this many lanes of the executor have work (HWANG_CHAIN_MAX_LANES)
  Stats stats_;
};

}  // namespace amd
}  // namespace hwang
