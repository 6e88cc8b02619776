#include "gpu_executor.h"

#include "../util/trace.h"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <ctime>
#include <pthread.h>
#include <map>
#include <memory>

namespace hwang {

hipError_t launch_convert_batch(const amd::ConvertDesc* reqs_dev, int n, int max_width, int max_height,
                                uint64_t bytes, hipStream_t stream);

namespace amd {

hipError_t launch_upload(const JobDesc* jobs, int njobs, hipStream_t stream);
hipError_t launch_table_copy(void* dst, const void* src, size_t bytes, hipStream_t stream);
hipError_t launch_done(uint32_t* flag, uint32_t ticket, hipStream_t stream);
hipError_t launch_batch(const JobDesc* jobs, int njobs, int max_mbs, int max_rows, bool any_inter, bool any_intra,
                        bool any_deblock, bool any_mv, const uint64_t bytes[4], hipStream_t stream);
hipError_t launch_pre(const JobDesc* jobs, int njobs, int max_mbs, int max_rows, bool any_mv, bool any_bs, bool any_residual,
                      const uint64_t bytes[3], hipStream_t stream);
hipError_t launch_main(const JobDesc* jobs, int njobs, int max_mbs, int max_rows, bool any_inter, bool any_intra,
                       bool any_deblock, const uint64_t bytes[3], hipStream_t stream);
hipError_t launch_chain(const ChainArgs& args, int workgroups, hipStream_t stream);

namespace {
// The HIP runtime multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues
// (default 4) and kernels of streams that share a queue serialise.  The
// executor needs its engines to really run side by side, so ask for more
// before the runtime initialises (no effect if the process already did, or if
// the user set the variable).
struct HwQueueEnv {
  HwQueueEnv() { setenv("GPU_MAX_HW_QUEUES", "16", 0); }
} g_hw_queue_env;

std::mutex g_registry_mu;
std::map<int, GpuExecutor*> g_registry;
bool g_atexit_registered = false;
}  // namespace

void shutdown_executors() {
  std::lock_guard<std::mutex> lk(g_registry_mu);
  for (auto& kv : g_registry) kv.second->stop();
}

GpuExecutor* GpuExecutor::get(int device, std::string* error) {
  std::lock_guard<std::mutex> lk(g_registry_mu);
  auto it = g_registry.find(device);
  if (it != g_registry.end()) return it->second;
  std::unique_ptr<GpuExecutor> ex(new GpuExecutor(device));
  if (!ex->start(error)) return nullptr;
  if (!g_atexit_registered) {
    // registered after the HIP runtime came up, so it runs before the
    // runtime's own teardown
    std::atexit(shutdown_executors);
    g_atexit_registered = true;
  }
  GpuExecutor* raw = ex.release();  // lives for the rest of the process
  g_registry[device] = raw;
  return raw;
}

bool GpuExecutor::start(std::string* error) {
  auto fail = [&](const char* what) {
    *error = std::string("GpuExecutor: ") + what;
    return false;
  };
  if (hipSetDevice(device_) != hipSuccess) return fail("cannot select the device");
  // CU partition.  An intra-only picture is a wavefront of row pipelines
  // whose speed is the latency of a dependent chain; sharing its CUs with
  // k_inter waves of the light engine more than doubles that latency (measured
  // with the original one-workgroup kernels: 7 ms -> 17 ms per 1080p I
  // picture), and every interval waits for its keyframe.  So every 8th CU (one per XCD position, 32 of 256) is reserved
  // for the heavy engines and the light engine's streams get the rest.
  // HWANG_CU_PARTITION=0 disables the masks.
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device_) != hipSuccess) return fail("cannot query the device");
  const int num_cus = prop.multiProcessorCount;
  const char* part_env = getenv("HWANG_CU_PARTITION");
  const int stride = part_env ? atoi(part_env) : 8;
  std::vector<uint32_t> heavy_mask((num_cus + 31) / 32, 0u), light_mask((num_cus + 31) / 32, 0u);
  for (int cu = 0; cu < num_cus; ++cu) {
    const bool heavy = stride > 1 && (cu % stride) == stride - 1;
    (heavy ? heavy_mask : light_mask)[cu / 32] |= 1u << (cu % 32);
  }
  auto make_stream = [&](hipStream_t* st, const std::vector<uint32_t>& mask) {
    if (stride > 1 && hipExtStreamCreateWithCUMask(st, static_cast<uint32_t>(mask.size()), mask.data()) == hipSuccess)
      return true;
    return hipStreamCreateWithFlags(st, hipStreamNonBlocking) == hipSuccess;
  };
  for (int ei = 0; ei < kEngines; ++ei) {
    Engine& en = engines_[ei];
    if (!make_stream(&en.stream, (is_light(ei) || ei == kPreEngine) ? light_mask : heavy_mask)) return fail("cannot create a stream");
    // Uploads go on the engine's compute stream.  A separate copy stream for the
    // light engine (HWANG_COPY_STREAM=1) lets batch N+1's records travel while
    // batch N's kernels run, but the hipEventRecord / hipStreamWaitEvent pair
    // that orders the two streams keeps one of the runtime's threads spinning in
    // ioctl() for as long as work is in flight - a whole CPU out of the 16 the
    // entropy decoders are short of (measured: 750-770 -> 817 frames/s without).
    const char* cs_env = getenv("HWANG_COPY_STREAM");
    if (ei == 0 && cs_env && atoi(cs_env) != 0) {
      if (!make_stream(&en.copy_stream, light_mask)) return fail("cannot create a stream");
    } else {
      en.copy_stream = en.stream;
    }
    for (int i = 0; i < kInflight; ++i) {
      if (hipHostMalloc(reinterpret_cast<void**>(&en.job_table_host[i]), sizeof(JobDesc) * kMaxJobs, hipHostMallocDefault) != hipSuccess ||
          hipMalloc(reinterpret_cast<void**>(&en.job_table_dev[i]), sizeof(JobDesc) * kMaxJobs) != hipSuccess ||
          hipHostMalloc(reinterpret_cast<void**>(&en.conv_table_host[i]), sizeof(ConvertDesc) * kMaxJobs, hipHostMallocDefault) != hipSuccess ||
          hipMalloc(reinterpret_cast<void**>(&en.conv_table_dev[i]), sizeof(ConvertDesc) * kMaxJobs) != hipSuccess ||
          (is_light(ei) && (hipHostMalloc(reinterpret_cast<void**>(&en.unit_table_host[i]), sizeof(uint32_t) * kUnitWords, hipHostMallocDefault) != hipSuccess ||
                            hipMalloc(reinterpret_cast<void**>(&en.unit_table_dev[i]), sizeof(uint32_t) * kUnitWords) != hipSuccess)) ||
          hipEventCreateWithFlags(&en.copied[i], hipEventDisableTiming) != hipSuccess)
        return fail("out of memory for job tables");
    }
  }
  // experiment knobs: how many of the heavy engines take work, how many batches each keeps in flight
  if (const char* e = getenv("HWANG_HEAVY_ENGINES")) heavy_engines_ = std::max(1, std::min(kHeavyEngines, atoi(e)));
  if (const char* e = getenv("HWANG_LINGER_US")) linger_us_ = std::max(0, std::min(atoi(e), 50000));
  if (const char* e = getenv("HWANG_LINGER_IDLE_US")) linger_idle_us_ = std::max(50, std::min(atoi(e), 50000));
  if (const char* e = getenv("HWANG_LINGER_FILL")) linger_fill_ = std::max(1, std::min(atoi(e), 100));
  if (const char* e = getenv("HWANG_HEAVY_LINGER_US")) heavy_linger_us_ = std::max(0, std::min(atoi(e), 50000));
  if (const char* e = getenv("HWANG_HEAVY_LINGER_PICTURES")) heavy_linger_pictures_ = std::max(1, atoi(e));
  if (const char* e = getenv("HWANG_PRE_LINGER_US")) pre_linger_us_ = std::max(0, std::min(atoi(e), 50000));
  if (const char* e = getenv("HWANG_PRE_LINGER_PICTURES")) pre_linger_pictures_ = std::max(1, atoi(e));
  if (const char* e = getenv("HWANG_NO_HEAVY")) no_heavy_ = atoi(e) != 0;  // experiment: intra-only pictures travel in the light batches
  if (const char* e = getenv("HWANG_HEAVY_INFLIGHT")) heavy_inflight_ = std::max(1, std::min(kInflight, atoi(e)));
  if (const char* e = getenv("HWANG_LIGHT_INFLIGHT")) light_inflight_ = std::max(1, std::min(kInflight, atoi(e)));
  if (const char* e = getenv("HWANG_LIGHT_ENGINES")) light_engines_ = std::max(1, std::min(1 + kExtraLightEngines, atoi(e)));
  if (const char* e = getenv("HWANG_PRESTAGE")) prestage_ = atoi(e) != 0;
  if (const char* e = getenv("HWANG_CHAIN_MAX_LANES")) chain_max_lanes_ = std::max(1, atoi(e));
  const char* mode = getenv("HWANG_COMPLETION");
  use_hostfunc_ = mode && std::string(mode) == "hostfunc";
  if (hipHostMalloc(reinterpret_cast<void**>(&done_flags_), sizeof(uint32_t) * kEngines * kInflight, hipHostMallocDefault) != hipSuccess)
    return fail("out of memory for completion flags");
  if (hipHostMalloc(reinterpret_cast<void**>(&fault_flags_), sizeof(uint32_t) * kEngines * kInflight, hipHostMallocDefault) != hipSuccess)
    return fail("out of memory for completion flags");
  for (int i = 0; i < kEngines * kInflight; ++i) done_flags_[i] = fault_flags_[i] = 0;
  for (int e = 0; e < kEngines; ++e) engines_[e].thread = std::thread(&GpuExecutor::run, this, e);
  if (!use_hostfunc_) poller_ = std::thread(&GpuExecutor::poll_loop, this);
  return true;
}

void GpuExecutor::stop() {
  {
    std::lock_guard<std::mutex> lk(mu_);
    if (stop_) return;
    stop_ = true;
  }
  const bool dbg = getenv("HWANG_EXEC_DEBUG") != nullptr;
  if (dbg) fprintf(stderr, "[executor] stop: joining engine threads\n");
  for (auto& cv : cv_work_) cv.notify_all();
  {
    std::lock_guard<std::mutex> lk(mu_);
    for (Lane& l : lanes_) l.cv.notify_all();
  }
  for (Engine& en : engines_)
    if (en.thread.joinable()) en.thread.join();
  cv_poll_.notify_all();
  if (poller_.joinable()) poller_.join();
  // With every decoder gone, leave nothing of ours for the runtime's own
  // teardown to trip over (rocprofv3 crashed in __cxa_finalize otherwise).  If
  // decoders are still alive (objects destroyed after the interpreter's atexit
  // handlers), their buffers are still registered with the runtime and tearing
  // the streams down under them blocks: leave everything to process exit.
  bool lanes_alive = false;
  {
    std::lock_guard<std::mutex> lk(mu_);
    for (const Lane& l : lanes_) lanes_alive |= l.active;
  }
  if (dbg) fprintf(stderr, "[executor] stop: threads joined%s\n", lanes_alive ? " (decoders still alive: streams kept)" : ", releasing streams");
  if (lanes_alive) return;
  (void)hipSetDevice(device_);
  for (Engine& en : engines_) {
    if (dbg) fprintf(stderr, "[executor] stop: engine %d sync\n", (int)(&en - engines_));
    if (en.stream) (void)hipStreamSynchronize(en.stream);
    if (dbg) fprintf(stderr, "[executor] stop: engine %d destroy\n", (int)(&en - engines_));
    if (en.copy_stream && en.copy_stream != en.stream) {
      (void)hipStreamSynchronize(en.copy_stream);
      (void)hipStreamDestroy(en.copy_stream);
    }
    if (en.stream) (void)hipStreamDestroy(en.stream);
    en.stream = en.copy_stream = nullptr;
    if (dbg) fprintf(stderr, "[executor] stop: engine %d free\n", (int)(&en - engines_));
    for (int i = 0; i < kInflight; ++i) {
      if (en.copied[i]) (void)hipEventDestroy(en.copied[i]);
      if (en.job_table_host[i]) (void)hipHostFree(en.job_table_host[i]);
      if (en.job_table_dev[i]) (void)hipFree(en.job_table_dev[i]);
      if (en.conv_table_host[i]) (void)hipHostFree(en.conv_table_host[i]);
      if (en.conv_table_dev[i]) (void)hipFree(en.conv_table_dev[i]);
      if (en.unit_table_host[i]) (void)hipHostFree(en.unit_table_host[i]);
      if (en.unit_table_dev[i]) (void)hipFree(en.unit_table_dev[i]);
      en.unit_table_host[i] = en.unit_table_dev[i] = nullptr;
      en.copied[i] = nullptr;
      en.job_table_host[i] = en.job_table_dev[i] = nullptr;
      en.conv_table_host[i] = en.conv_table_dev[i] = nullptr;
    }
  }
  if (dbg) fprintf(stderr, "[executor] stop: done\n");
}

int GpuExecutor::register_lane(int light_group, bool chain) {
  if (const char* e = getenv("HWANG_CHAIN")) chain = atoi(e) != 0;  // 1: every lane (the test suite runs the corpus through it), 0: none
  const int g = light_group % light_engines_;
  const int light_engine = g == 0 ? 0 : kHeavyEngines + g;
  std::lock_guard<std::mutex> lk(mu_);
  for (size_t i = 0; i < lanes_.size(); ++i)
    if (!lanes_[i].active && lanes_[i].fifo.empty() && lanes_[i].completed == lanes_[i].submitted) {
      Lane& l = lanes_[i];
      l.submitted = l.launched = l.completed = 0;
      l.last_engine = light_engine;
      l.light_engine = light_engine;
      l.chain = chain;
      l.patient = false;
      l.waiters = 0;
      l.active = true;
      return static_cast<int>(i);
    }
  lanes_.emplace_back();
  lanes_.back().active = true;
  lanes_.back().light_engine = lanes_.back().last_engine = light_engine;
  lanes_.back().chain = chain;
  return static_cast<int>(lanes_.size() - 1);
}

void GpuExecutor::unregister_lane(int lane) {
  std::unique_lock<std::mutex> lk(mu_);
  lanes_[lane].cv.wait(lk, [&] { return lanes_[lane].completed == lanes_[lane].submitted || !error_.empty() || stop_; });
  lanes_[lane].active = false;
}

void GpuExecutor::set_lane_patient(int lane, bool patient) {
  std::lock_guard<std::mutex> lk(mu_);
  lanes_[lane].patient = patient;
}

uint64_t GpuExecutor::submit_picture(int lane, const PicSubmit& job) {
  uint64_t seq;
  int target = 0;
  {
    std::lock_guard<std::mutex> lk(mu_);
    Item it;
    it.is_picture = true;
    it.pic = job;
    // A chain lane's intra-only pictures stay with its other pictures (the P picture behind a keyframe then follows it
    // row by row inside one launch instead of waiting for another engine's batch to complete) - as long as the
    // executor is not busy with many lanes: then batches are wide, chains are off (run()), and a keyframe inside a
    // batch of P pictures would hold every lane of that batch for its own duration, which is what the heavy engines
    // are there to avoid.
    bool with_its_chain = false;
    if (job.heavy && lanes_[lane].chain) {
      int busy = 0;
      for (const Lane& l : lanes_) busy += l.active && (!l.fifo.empty() || l.launched != l.completed);
      with_its_chain = busy <= chain_max_lanes_;
    }
    it.engine = (job.heavy && !with_its_chain && !no_heavy_) ? 1 + lane % heavy_engines_ : lanes_[lane].light_engine;
    it.pre = prestage_ ? 0 : 2;
    it.t_submit = std::chrono::steady_clock::now();
    lanes_[lane].t_last_submit = it.t_submit;
    lanes_[lane].fifo.push_back(it);
    seq = ++lanes_[lane].submitted;
    target = prestage_ ? kPreEngine : it.engine;
    if (keep_recent_ > 0) {
      recent_.push_back(job);
      while (static_cast<int>(recent_.size()) > keep_recent_) recent_.pop_front();
    }
  }
  cv_work_[target].notify_one();
  return seq;
}

uint64_t GpuExecutor::submit_convert(int lane, const ConvSubmit& job) {
  uint64_t seq;
  int target = 0;
  {
    std::lock_guard<std::mutex> lk(mu_);
    Item it;
    it.is_picture = false;
    it.conv = job;
    lanes_[lane].fifo.push_back(it);
    seq = ++lanes_[lane].submitted;
    target = lanes_[lane].last_engine;  // conversions follow the lane's last picture
  }
  cv_work_[target].notify_one();
  return seq;
}

bool GpuExecutor::stopped() {
  std::lock_guard<std::mutex> lk(mu_);
  return stop_;
}

uint64_t GpuExecutor::completed(int lane) {
  std::lock_guard<std::mutex> lk(mu_);
  return lanes_[lane].completed;
}

void GpuExecutor::completed(const int* lanes, int n, uint64_t* out) {
  std::lock_guard<std::mutex> lk(mu_);
  for (int i = 0; i < n; ++i) out[i] = lanes_[lanes[i]].completed;
}

bool GpuExecutor::wait(int lane, uint64_t seq, std::string* error) {
  std::unique_lock<std::mutex> lk(mu_);
  ++lanes_[lane].waiters;
  lanes_[lane].cv.wait(lk, [&] { return lanes_[lane].completed >= seq || !error_.empty() || stop_; });
  --lanes_[lane].waiters;
  if (!error_.empty()) {
    if (error) *error = error_;
    return false;
  }
  if (lanes_[lane].completed < seq) {
    if (error) *error = "GpuExecutor: shut down";
    return false;
  }
  return true;
}

void GpuExecutor::keep_recent(int pictures) {
  std::lock_guard<std::mutex> lk(mu_);
  keep_recent_ = pictures > 0 ? pictures : 0;
  while (static_cast<int>(recent_.size()) > keep_recent_) recent_.pop_front();
}

bool GpuExecutor::replay(int width, int launches, bool chain, ReplayResult* out, std::string* error) {
  // The pictures whose records and surfaces are still what they were: per staging set (desc.mbs) the LAST picture
  // that used it; inter pictures only (a keyframe's cost is the heavy engine's, measured by the bench's spans).
  std::vector<PicSubmit> pics;
  {
    std::lock_guard<std::mutex> lk(mu_);
    for (const Lane& l : lanes_)
      if (!l.fifo.empty() || l.launched != l.completed) {
        *error = "GpuExecutor::replay: the executor is busy";
        return false;
      }
    for (auto it = recent_.rbegin(); it != recent_.rend(); ++it) {
      if (!it->desc.do_inter) continue;
      bool dup = false;
      for (const PicSubmit& p : pics) dup |= p.desc.mbs == it->desc.mbs || p.desc.pp.surfaces[p.desc.pp.cur_slot].y == it->desc.pp.surfaces[it->desc.pp.cur_slot].y;
      if (!dup) pics.push_back(*it);
    }
  }
  out->pictures_available = static_cast<int>(pics.size());
  if (pics.empty()) {
    *error = "GpuExecutor::replay: no pictures kept (keep_recent before the decode)";
    return false;
  }
  if (undelivered_.load(std::memory_order_acquire) > 0) {
    // the pictures are decoded in place again: a frame somebody has yet to fetch from one of these surfaces may come
    // out of a picture whose references have been reused since
    *error = "GpuExecutor::replay: a decoder of this device still holds decoded frames that were not fetched or discarded";
    return false;
  }
  width = std::max(1, std::min<int>({width, static_cast<int>(pics.size()), kMaxJobs}));
  if (chain) {
    // a chain launch's unit table holds kUnitWords words: fewer pictures per launch rather than a truncated list
    size_t words = 4;
    int fit = 0;
    for (; fit < width; ++fit) {
      const PicSubmit& p = pics[fit % pics.size()];
      const size_t bands = static_cast<size_t>((p.desc.pp.mb_height + 3) / 4);
      const size_t runs = static_cast<size_t>(p.desc.pp.mb_height) * ((p.desc.pp.mb_width + kChainRun - 1) / kChainRun);
      const size_t units = (p.desc.do_inter ? runs : 0) + (p.desc.do_intra ? bands : 0) + (p.desc.do_deblock ? bands : 0);
      if (words + units > kUnitWords) break;
      words += units;
    }
    // (the launches rotate through the kept pictures: size for the largest of them)
    size_t worst = 0;
    for (const PicSubmit& p : pics) {
      const size_t bands = static_cast<size_t>((p.desc.pp.mb_height + 3) / 4);
      const size_t runs = static_cast<size_t>(p.desc.pp.mb_height) * ((p.desc.pp.mb_width + kChainRun - 1) / kChainRun);
      worst = std::max(worst, runs + 2 * bands);
    }
    fit = std::min<int>(fit, static_cast<int>((kUnitWords - 4) / std::max<size_t>(worst, 1)));
    if (fit < 1) {
      *error = "GpuExecutor::replay: one picture's units do not fit a chain launch's unit table";
      return false;
    }
    width = fit;
  }
  if (hipSetDevice(device_) != hipSuccess) {
    *error = "GpuExecutor::replay: cannot select the device";
    return false;
  }
  Engine& en = engines_[0];
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) {
    *error = "GpuExecutor::replay: no events";
    return false;
  }
  static uint32_t replay_epoch = 0x40000000u;
  __atomic_store_n(fault_flags_, 0u, __ATOMIC_RELEASE);
  hipError_t e = hipEventRecord(e0, en.stream);
  for (int it = 0; it < launches && e == hipSuccess; ++it) {
    int max_mbs = 0, max_rows = 0;
    bool any_inter = false, any_intra = false, any_deblock = false, any_mv = false;
    uint64_t bytes[4] = {0, 0, 0, 0}, pre_bytes[3] = {0, 0, 0};
    for (int i = 0; i < width; ++i) {
      const PicSubmit& p = pics[(static_cast<size_t>(it) * width + i) % pics.size()];
      JobDesc& jd = en.job_table_host[0][i];
      jd = p.desc;
      jd.num_uploads = 0;  // the records are where the decode left them
      jd.ticket_mv = jd.ticket_db = jd.ticket_in = 0;
      jd.fault = fault_flags_;  // (engine 0's first word: the executor is idle, nobody else writes it)
      jd.chain_ctl = chain ? en.unit_table_dev[0] : nullptr;
      jd.epoch = ++replay_epoch ? replay_epoch : ++replay_epoch;
      for (int k = 0; k < 256; ++k) jd.slot_job[k] = -1;
      max_mbs = std::max(max_mbs, p.num_mbs);
      max_rows = std::max(max_rows, p.desc.pp.mb_height);
      any_inter |= p.desc.do_inter != 0;
      any_intra |= p.desc.do_intra != 0;
      any_deblock |= p.desc.do_deblock != 0;
      any_mv |= p.desc.do_mv != 0;
      bytes[0] += p.bytes_inter;
      bytes[1] += p.bytes_intra;
      bytes[2] += p.bytes_deblock;
      pre_bytes[0] += p.bytes_mv;
      pre_bytes[1] += p.desc.do_deblock ? static_cast<uint64_t>(p.num_mbs) * (176 + 96) : 0;
      pre_bytes[2] += p.desc.do_intra ? static_cast<uint64_t>(p.num_intra_mbs) * 768 : 0;
    }
    // (the same picture twice in a launch would share its hand-off words: width <= pictures available sees to it)
    e = launch_table_copy(en.job_table_dev[0], en.job_table_host[0], sizeof(JobDesc) * width, en.stream);
    // the host table is rewritten for the next launch: the copy has to have happened (costs a round trip per launch,
    // outside the kernels' spans; the wall time below includes it and says so)
    if (e == hipSuccess) e = hipStreamSynchronize(en.stream);
    if (e == hipSuccess) e = launch_pre(en.job_table_dev[0], width, max_mbs, max_rows, any_mv, any_deblock, any_intra, pre_bytes, en.stream);
    if (e != hipSuccess) break;
    if (chain) {
      uint32_t* table = en.unit_table_host[0];
      table[0] = table[1] = table[2] = table[3] = 0;
      uint32_t* rows = table + 4;
      uint32_t nrows = 0, ninter = 0;
      for (int stage = 0; stage < 2; ++stage)
        for (int i = 0; i < width; ++i) {
          const JobDesc& jd = en.job_table_host[0][i];
          if (!(stage ? jd.do_deblock : jd.do_intra)) continue;
          for (int b = 0; b < (jd.pp.mb_height + 3) / 4; ++b) rows[nrows++] = static_cast<uint32_t>(i) << 8 | static_cast<uint32_t>(stage) << 7 | static_cast<uint32_t>(b);
        }
      uint32_t* inter = rows + nrows;
      for (int i = 0; i < width; ++i) {
        const JobDesc& jd = en.job_table_host[0][i];
        const int runs = jd.pp.mb_height * ((jd.pp.mb_width + kChainRun - 1) / kChainRun);
        if (!jd.do_inter) continue;
        for (int r = 0; r < runs; ++r) inter[ninter++] = static_cast<uint32_t>(i) << 20 | static_cast<uint32_t>(r);  // (width was sized for it)
      }
      ChainArgs args;
      args.jobs = en.job_table_dev[0];
      args.tickets = en.unit_table_dev[0];
      args.row_units = en.unit_table_dev[0] + 4;
      args.inter_units = en.unit_table_dev[0] + 4 + nrows;
      args.num_row_units = nrows;
      args.num_inter_units = ninter;
      args.bytes = bytes[0] + bytes[1] + bytes[2];
      // (chain replays need the run masks zeroed, which is k_upload's job in a decode)
      e = launch_upload(en.job_table_dev[0], width, en.stream);
      if (e == hipSuccess) e = launch_table_copy(en.unit_table_dev[0], en.unit_table_host[0], sizeof(uint32_t) * (4 + nrows + ninter), en.stream);
      if (e == hipSuccess) e = hipStreamSynchronize(en.stream);
      if (e == hipSuccess) e = launch_chain(args, static_cast<int>(std::min<uint32_t>(std::max<uint32_t>(std::max<uint32_t>(nrows, (ninter + 3) / 4), 1u), 256u)), en.stream);
    } else {
      const uint64_t b3[3] = {bytes[0], bytes[1], bytes[2]};
      e = launch_main(en.job_table_dev[0], width, max_mbs, max_rows, any_inter, any_intra, any_deblock, b3, en.stream);
    }
  }
  if (e == hipSuccess) e = hipEventRecord(e1, en.stream);
  if (e == hipSuccess) e = hipEventSynchronize(e1);
  float ms = 0.f;
  if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  if (e != hipSuccess) {
    *error = std::string("GpuExecutor::replay: ") + hipGetErrorString(e);
    return false;
  }
  if (const uint32_t code = __atomic_load_n(fault_flags_, __ATOMIC_ACQUIRE)) {
    *error = std::string("GpuExecutor::replay: a hand-off timed out (") + fault_name(code) + "); the timing is void";
    return false;
  }
  out->width = width;
  out->launches = launches;
  out->seconds = ms / 1000.0;
  return true;
}

GpuExecutor::Stats GpuExecutor::stats() {
  std::lock_guard<std::mutex> lk(mu_);
  return stats_;
}

void GpuExecutor::on_batch_done(void* p) {
  Batch* b = static_cast<Batch*>(p);
  b->owner->complete(b, hipSuccess);
}

const char* GpuExecutor::fault_name(uint32_t code) {
  switch (code) {
    case FAULT_MV_IMPORT: return "k_mv: the motion-vector line of the band above never arrived";
    case FAULT_INTRA_LINE: return "k_intra: the neighbour line of the band above never arrived";
    case FAULT_INTRA_PROGRESS: return "k_intra: the band above never reported progress";
    case FAULT_DEBLOCK_PROGRESS: return "k_deblock: the band above never reported progress";
    case FAULT_CHAIN_REFERENCE: return "k_chain: a reference picture of the same launch never got as far as a macroblock reads";
    case FAULT_CHAIN_INTER: return "k_chain: inter macroblocks that a row waits for never landed";
    case FAULT_CHAIN_INTRA: return "k_chain: the intra stage of a row never got as far as the deblocking needs";
    default: return "unknown fault code";
  }
}

void GpuExecutor::complete(Batch* b, hipError_t status) {
  trace::event("batch complete", b->engine);
  std::lock_guard<std::mutex> lk(mu_);
  if (status != hipSuccess && error_.empty())
    error_ = std::string("GpuExecutor: launch failed: ") + hipGetErrorString(status);
  // A row kernel gave up waiting for another band: the pictures of this batch
  // (and everything predicted from them) hold wrong samples.  Sticky, like a
  // launch failure: every lane's wait() fails from here on.
  if (b->fault) {
    const uint32_t code = __atomic_load_n(const_cast<uint32_t*>(b->fault), __ATOMIC_ACQUIRE);
    const bool chain_fault = code == FAULT_CHAIN_REFERENCE || code == FAULT_CHAIN_INTER || code == FAULT_CHAIN_INTRA;
    // (only with this batch alone on its engine: a later batch already queued on the stream behind the faulted launch
    // predicts from and converts the surfaces the chain left unfinished, and its completion would credit the lanes
    // past this batch's sequence numbers - with HWANG_*_INFLIGHT > 1 the fault stays what it was, a sticky error)
    if (chain_fault && b->chain && !b->retried && status == hipSuccess && error_.empty() && !stop_ &&
        engines_[b->engine].inflight == 1) {
      // A chain launch gave up: nothing in it moved for its polls' bound.  The pictures' records are intact and the
      // wide kernels order pictures by kernel boundaries, not by hand-offs: the engine thread runs the batch again
      // through them (rerun_wide); the batch stays in flight under a new ticket until that run is through.
      fprintf(stderr, "hwang_amd: a chain launch timed out (%s); running its %zu picture(s) again through the wide kernels\n",
              fault_name(code), b->retry_mbs.size());
      b->retried = true;
      b->retry_pending = true;
      b->ticket = ++next_ticket_;
      __atomic_store_n(const_cast<uint32_t*>(b->fault), 0u, __ATOMIC_RELEASE);
      ++stats_.chain_retries;
      cv_work_[b->engine].notify_one();
      return;
    }
    if (code && error_.empty())
      error_ = std::string("GpuExecutor: row-pipeline hand-off timed out (") + fault_name(code) +
               "); the batch's pictures are invalid";
  }
  const bool failed = !error_.empty();
  for (Item* it : b->pre_items) it->pre = 2;  // (a batch of the records-only stage credits no lane: the pictures go on)
  b->pre_items.clear();
  for (auto& lc : b->lane_counts) {
    Lane& l = lanes_[lc.first];
    l.completed += lc.second;
    if (l.waiters) l.cv.notify_all();
  }
  if (failed)
    for (Lane& l : lanes_) l.cv.notify_all();  // everyone must see the error
  b->lane_counts.clear();
  b->in_flight = false;
  --engines_[b->engine].inflight;
  --total_inflight_;
  // a batch slot is free here, and the other engines may now take over these lanes
  for (auto& cv : cv_work_) cv.notify_one();
}

void GpuExecutor::poll_loop() {
  pthread_setname_np(pthread_self(), "hw-poller");
  std::vector<Batch*> done;
  std::unique_lock<std::mutex> lk(mu_);
  for (;;) {
    if (stop_ && total_inflight_ == 0) break;
    if (total_inflight_ == 0) {
      cv_poll_.wait(lk);
      continue;
    }
    done.clear();
    for (Engine& en : engines_)
      for (Batch& b : en.batches)
        if (b.in_flight && b.flag && __atomic_load_n(const_cast<uint32_t*>(b.flag), __ATOMIC_ACQUIRE) == b.ticket) done.push_back(&b);
    lk.unlock();
    if (done.empty()) {
      struct timespec ts = {0, 40000};  // 40 us: far below a batch's duration
      nanosleep(&ts, nullptr);
    } else {
      for (Batch* b : done) complete(b, hipSuccess);
    }
    lk.lock();
  }
}

// Engine thread: sleeps until one of its lanes has something eligible AND a
// batch slot is free, forms the batch, launches it and appends a host
// function to the stream that reports completion.  Nothing polls and nothing
// blocks inside the HIP runtime, so the threads cost no CPU while the GPU
// works (the host's CPUs belong to the entropy decoders).
// The records-only stage.  Eager, one batch in flight: while it runs, whatever the lanes submit accumulates into the
// next one, so the launches are as wide as the submission rate makes them.
void GpuExecutor::run_pre() {
  pthread_setname_np(pthread_self(), "hw-pre");
  (void)hipSetDevice(device_);
  Engine& en = engines_[kPreEngine];
  const bool debug = getenv("HWANG_EXEC_DEBUG") != nullptr;
  uint64_t n_batches = 0, n_pics = 0;
  bool last_was_keyframes_only = false;
  std::vector<Item*> items;
  for (;;) {
    items.clear();
    int slot = -1;
    {
      std::unique_lock<std::mutex> lk(mu_);
      auto have_work = [&] {
        for (const Lane& l : lanes_)
          for (const Item& it : l.fifo)
            if (it.is_picture && it.pre == 0) return true;
        return false;
      };
      int seen_waiting = 0;
      auto last_growth = std::chrono::steady_clock::now();
      for (;;) {
        cv_work_[kPreEngine].wait(lk, [&] { return stop_ || (en.inflight < 1 && have_work()); });
        if (stop_ || !pre_linger_us_) break;
        // Linger (see run()): k_mv is a row pipeline of ~0.3 ms whatever the width; launched for whatever has
        // arrived since the last launch it ran two thousand times a step for four pictures each.  Keyframes do not
        // wait (their stage is an upload and k_residual, and their whole GOP waits for them), nor does anything
        // while few lanes are busy.
        int busy = 0, waiting = 0;
        bool keyframe = false, latency_lanes = false;
        const auto now = std::chrono::steady_clock::now();
        auto oldest = now;
        for (const Lane& l : lanes_) {
          if (!l.active) continue;
          const bool b = !l.fifo.empty() || l.launched != l.completed;
          busy += b;
          latency_lanes |= b && (l.chain || !l.patient);
          for (const Item& it : l.fifo)
            if (it.is_picture && it.pre == 0) {
              ++waiting;
              keyframe |= it.pic.heavy != 0;
              if (it.t_submit < oldest) oldest = it.t_submit;
            }
        }
        if (busy <= chain_max_lanes_ || latency_lanes || (keyframe && !last_was_keyframes_only) || waiting >= pre_linger_pictures_) break;
        if (waiting > seen_waiting) {
          seen_waiting = waiting;
          last_growth = now;
        }
        const auto waited = std::chrono::duration_cast<std::chrono::microseconds>(now - oldest).count();
        const auto idle = std::chrono::duration_cast<std::chrono::microseconds>(now - last_growth).count();
        if (waited >= pre_linger_us_ || idle >= linger_idle_us_) break;
        cv_work_[kPreEngine].wait_for(lk, std::chrono::microseconds(std::max<long>(std::min<long>(pre_linger_us_ - waited, linger_idle_us_ - idle), 50)));
      }
      if (stop_) break;
      for (int i = 0; i < kInflight; ++i)
        if (!en.batches[i].in_flight) slot = i;
      Batch& batch = en.batches[slot];
      // Intra-only pictures first, in a batch of their own: their records-only stage is an upload and k_residual
      // (tens of microseconds), and every picture of their GOP waits for them - behind a P picture's k_mv (0.7 ms)
      // a keyframe reached its sample stage 0.7 ms late, once per interval of the reference's caller.
      // (never twice in a row: with many GOPs in flight there is always another keyframe, and the P pictures must
      // not wait for ever)
      bool keyframes_waiting = false;
      if (!last_was_keyframes_only)
        for (const Lane& l : lanes_)
          for (const Item& it : l.fifo) keyframes_waiting |= it.is_picture && it.pre == 0 && it.pic.heavy;
      last_was_keyframes_only = keyframes_waiting;
      for (Lane& l : lanes_)
        for (Item& it : l.fifo)
          if (it.is_picture && it.pre == 0 && items.size() < static_cast<size_t>(kMaxJobs) && (!keyframes_waiting || it.pic.heavy)) {
            it.pre = 1;
            items.push_back(&it);
          }
      batch.pre_items = items;
      batch.in_flight = true;
      batch.owner = this;
      batch.engine = kPreEngine;
      batch.ticket = ++next_ticket_;
      batch.flag = done_flags_ + kPreEngine * kInflight + slot;
      batch.fault = fault_flags_ + kPreEngine * kInflight + slot;
      *batch.fault = 0;
      ++en.inflight;
      ++total_inflight_;
    }
    Batch& batch = en.batches[slot];
    trace::event("records-only batch formed", static_cast<long>(items.size()));
    ++n_batches;
    n_pics += items.size();
    int max_mbs = 0, max_rows = 0;
    bool any_mv = false, any_bs = false, any_residual = false;
    uint64_t bytes[3] = {0, 0, 0};
    for (size_t i = 0; i < items.size(); ++i) {
      const PicSubmit& p = items[i]->pic;  // (stable: the item cannot leave its queue before this stage is through)
      JobDesc& jd = en.job_table_host[slot][i];
      jd = p.desc;
      jd.num_uploads = p.ncopies;
      jd.ticket_mv = jd.ticket_db = jd.ticket_in = 0;
      jd.fault = const_cast<uint32_t*>(batch.fault);
      for (int c = 0; c < p.ncopies; ++c) {
        jd.uploads[c].dst = p.copies[c].dst;
        jd.uploads[c].src = p.copies[c].src;
        jd.uploads[c].bytes = p.copies[c].bytes;
      }
      max_mbs = p.num_mbs > max_mbs ? p.num_mbs : max_mbs;
      max_rows = p.desc.pp.mb_height > max_rows ? p.desc.pp.mb_height : max_rows;
      any_mv |= p.desc.do_mv != 0;
      any_bs |= p.desc.do_deblock != 0;
      any_residual |= p.desc.do_intra != 0;
      bytes[0] += p.bytes_mv;
      bytes[1] += p.desc.do_deblock ? static_cast<uint64_t>(p.num_mbs) * (176 + 96) : 0;
      bytes[2] += p.desc.do_intra ? static_cast<uint64_t>(p.num_intra_mbs) * 768 : 0;  // (inter macroblocks return at once)
    }
    hipError_t e = launch_table_copy(en.job_table_dev[slot], en.job_table_host[slot], sizeof(JobDesc) * items.size(), en.stream);
    if (e == hipSuccess) e = launch_upload(en.job_table_dev[slot], static_cast<int>(items.size()), en.stream);
    if (e == hipSuccess)
      e = launch_pre(en.job_table_dev[slot], static_cast<int>(items.size()), max_mbs, max_rows, any_mv, any_bs, any_residual,
                     bytes, en.stream);
    if (e == hipSuccess) {
      if (use_hostfunc_) {
        e = hipLaunchHostFunc(en.stream, &GpuExecutor::on_batch_done, &batch);
      } else {
        e = launch_done(const_cast<uint32_t*>(batch.flag), batch.ticket, en.stream);
        cv_poll_.notify_one();
      }
    }
    if (e != hipSuccess) {
      (void)hipStreamSynchronize(en.stream);
      complete(&batch, e);
    }
  }
  if (debug) fprintf(stderr, "[executor pre] batches %llu pics %llu\n", (unsigned long long)n_batches, (unsigned long long)n_pics);
  (void)hipStreamSynchronize(en.stream);
}

// The second run of a chain batch whose polls expired (complete()): every picture of the table through the wide
// kernels, one launch_main per picture in table order - a lane's pictures stand in the table in decoding order and
// pictures of different lanes do not refer to each other, so stream order is all the ordering there is - then the
// batch's conversions and its completion mark again.  The records-only stage's results (vectors, boundary strengths,
// intra residuals) and the records are where they were; every sample of every picture is written again before it is
// filtered, so what the abandoned launch left in the surfaces does not matter.  Fresh epochs: the abandoned launch
// may have left hand-off words of these pictures behind.
void GpuExecutor::rerun_wide(Engine& en, Batch& batch) {
  const int slot = batch.slot;
  const int n = static_cast<int>(batch.retry_mbs.size());
  trace::event("chain batch rerun", n);
  for (int i = 0; i < n; ++i) {
    JobDesc& jd = en.job_table_host[slot][i];
    jd.chain_ctl = nullptr;
    jd.num_uploads = 0;
    jd.ticket_mv = jd.ticket_db = jd.ticket_in = 0;
    jd.epoch ^= 0x80000000u;  // (a decoder counts its epochs up from 1: the upper half is free)
    if (jd.epoch == 0) jd.epoch = 0x80000000u;
    if (jd.inject_fault == 2) jd.inject_fault = 0;  // (the test hook of the chain's hand-off)
    for (int k = 0; k < 256; ++k) jd.slot_job[k] = -1;
  }
  hipError_t e = launch_table_copy(en.job_table_dev[slot], en.job_table_host[slot], sizeof(JobDesc) * n, en.stream);
  const uint64_t no_bytes[3] = {0, 0, 0};  // (the spans of the first attempt carry the batch's bytes)
  for (int i = 0; i < n && e == hipSuccess; ++i) {
    const JobDesc& jd = en.job_table_host[slot][i];
    e = launch_main(en.job_table_dev[slot] + i, 1, batch.retry_mbs[i], jd.pp.mb_height, jd.do_inter != 0, jd.do_intra != 0,
                    jd.do_deblock != 0, no_bytes, en.stream);
  }
  const std::vector<ConvSubmit>& convs = batch.retry_convs;
  if (!convs.empty() && e == hipSuccess) {
    int mw = 0, mh = 0;
    for (const ConvSubmit& c : convs) {
      mw = std::max(mw, c.desc.width);
      mh = std::max(mh, c.desc.height);
    }
    // (the conversion table of the first attempt is still on the device)
    e = launch_convert_batch(en.conv_table_dev[slot], static_cast<int>(convs.size()), mw, mh, 0, en.stream);
    for (size_t i = 0; i < convs.size() && e == hipSuccess; ++i)
      if (convs[i].d2h_dst)
        e = hipMemcpyAsync(convs[i].d2h_dst, convs[i].desc.dst, convs[i].d2h_bytes, hipMemcpyDeviceToHost, en.stream);
  }
  if (e == hipSuccess) {
    if (use_hostfunc_) {
      e = hipLaunchHostFunc(en.stream, &GpuExecutor::on_batch_done, &batch);
    } else {
      e = launch_done(const_cast<uint32_t*>(batch.flag), batch.ticket, en.stream);
      cv_poll_.notify_one();
    }
  }
  if (e != hipSuccess) {
    (void)hipStreamSynchronize(en.stream);
    complete(&batch, e);
  }
}

void GpuExecutor::run(int engine) {
  if (engine == kPreEngine) return run_pre();
  {
    char name[16];
    snprintf(name, sizeof(name), is_light(engine) ? "hw-light%d" : "hw-heavy%d", engine);
    pthread_setname_np(pthread_self(), name);
  }
  (void)hipSetDevice(device_);
  Engine& en = engines_[engine];
  // Batches are formed late, not eagerly: while one batch runs, every lane
  // keeps parsing, so by the time it finishes most lanes have their next
  // picture ready and the next batch is wide.  (Launching whatever is pending
  // right away produced batches of ~5 pictures that each held the GPU for a
  // full pipeline latency.)  See gpu_executor.h for the measured settings.
  const int max_inflight = is_light(engine) ? light_inflight_ : heavy_inflight_;

  std::vector<PicSubmit> pics;
  std::vector<ConvSubmit> convs;
  std::vector<int> pic_lane, pic_pos, used_slots;  // per picture of the batch: its lane, its place in the lane's chain
  // words a picture adds to a chain launch's unit table
  auto chain_units = [](const PicSubmit& p) {
    const size_t bands = static_cast<size_t>((p.desc.pp.mb_height + 3) / 4);
    const size_t runs = static_cast<size_t>(p.desc.pp.mb_height) * ((p.desc.pp.mb_width + kChainRun - 1) / kChainRun);
    return (p.desc.do_inter ? runs : 0) + (p.desc.do_intra ? bands : 0) + (p.desc.do_deblock ? bands : 0);
  };
  uint64_t n_chain_batches = 0, n_chain_pics = 0, max_chain = 0;
  const bool debug = getenv("HWANG_EXEC_DEBUG") != nullptr;
  uint64_t hist[9] = {0}, n_batches = 0, n_pics = 0, n_convs = 0, lane_state[5] = {0, 0, 0, 0, 0};
  for (;;) {
    pics.clear();
    convs.clear();
    pic_lane.clear();
    pic_pos.clear();
    bool any_chain = false;
    size_t chain_words = 0;
    int slot = -1;
    {
      std::unique_lock<std::mutex> lk(mu_);
      auto have_work = [&] {
        for (const Lane& l : lanes_)
          if (eligible(l, engine)) return true;
        return false;
      };
      Batch* again = nullptr;
      auto retry_wanted = [&] {
        for (Batch& b : en.batches)
          if (b.in_flight && b.retry_pending) again = &b;
        return again != nullptr;
      };
      int seen_ready = 0;
      auto last_growth = std::chrono::steady_clock::now();
      for (;;) {
        cv_work_[engine].wait(lk, [&] { return stop_ || retry_wanted() || (en.inflight < max_inflight && have_work()); });
        const int linger_us = is_light(engine) ? linger_us_ : heavy_linger_us_;
        if (stop_ || again || !linger_us) break;
        // Linger: a batch holds at most one picture per lane and lasts as long as its slowest row pipeline whatever its
        // width.  With a thread per CPU the lanes deliver a steady stream (no bursts between quota freezes), and a
        // batch formed the moment the last one ended would carry a third of the lanes: wait until most of this
        // engine's active lanes have a picture ready, or the oldest one has waited its share.  Not with few lanes busy
        // (the reference's caller: latency is all it has).
        int busy = 0, mine = 0, ready = 0, coming = 0;
        bool latency_lanes = false;  // a decoder with entropy workers (one automata, intervals in turn) is served at once
        const auto now = std::chrono::steady_clock::now();
        auto oldest = now;
        for (const Lane& l : lanes_) {
          if (!l.active) continue;
          const bool b = !l.fifo.empty() || l.launched != l.completed;
          busy += b;
          latency_lanes |= b && (l.chain || !l.patient);
          if (is_light(engine) && l.light_engine != engine) continue;
          // A lane counts towards the engine's share while it is at work: something queued or in flight, or a picture
          // submitted within the last linger period (its thread is parsing the next one).  Not every REGISTERED lane:
          // the retriever registers two backends per lane and decodes with one at a time, and another decoder's idle
          // lanes on the device are registered too - with those in `mine` the fill test could never be met and every
          // batch waited for the age limit instead.
          const bool at_work = b || std::chrono::duration_cast<std::chrono::microseconds>(now - l.t_last_submit).count() < linger_us;
          if (!at_work) continue;
          ++mine;
          if (eligible(l, engine)) {
            ++ready;
            const Item& it = l.fifo.front();
            if (it.is_picture && it.t_submit < oldest) oldest = it.t_submit;
          } else if (!l.fifo.empty() && l.fifo.front().is_picture && l.fifo.front().engine == engine && l.fifo.front().pre != 2) {
            ++coming;  // in (or waiting for) the records-only stage: eligible within a launch of that
          }
        }
        // (a heavy engine cannot know how many keyframes are on their way: a count, not a share of the lanes)
        const bool full = is_light(engine) ? ready * 100 >= mine * linger_fill_ : ready >= heavy_linger_pictures_;
        if (busy <= chain_max_lanes_ || latency_lanes || full) break;
        // ... and only while pictures keep arriving: at the tail of a request, or in a small one, a batch that nothing
        // has joined for `linger_idle_us_` goes as it is
        if (ready + coming > seen_ready) {
          seen_ready = ready + coming;
          last_growth = now;
        }
        const auto waited = std::chrono::duration_cast<std::chrono::microseconds>(now - oldest).count();
        const auto idle = std::chrono::duration_cast<std::chrono::microseconds>(now - last_growth).count();
        if (waited >= linger_us || (coming == 0 && idle >= linger_idle_us_)) break;
        long remaining = linger_us - waited;
        if (coming == 0) remaining = std::min<long>(remaining, linger_idle_us_ - idle);
        cv_work_[engine].wait_for(lk, std::chrono::microseconds(std::max<long>(remaining, 50)));
      }
      if (!again) retry_wanted();  // (stop_ may have ended the wait before the predicate looked)
      if (again) {  // (also while stopping: the batch is in flight until its second run has completed)
        again->retry_pending = false;
        lk.unlock();
        rerun_wide(en, *again);
        continue;
      }
      if (stop_) break;
      for (int i = 0; i < kInflight; ++i)
        if (!en.batches[i].in_flight) slot = i;
      Batch& batch = en.batches[slot];
      // Chains are for a decoder that has ONE thing going - an interval of the reference's caller: a keyframe and the
      // pictures behind it, nothing else on the device, latency is all there is.  With several lanes busy the GPU is
      // fed by width instead (one picture per lane and batch through the wide kernels, whose throughput per picture is
      // several times k_chain's: 8 160 inter waves at a time against 900 - and a chain launch's persistent workgroups
      // hold the light CUs for milliseconds, in the way of the other engines' launches), and a decoder whose caller
      // hands it many intervals at once (hwang.Decoder's grouped caller, 16 lanes over four engines) is host-bound
      // either way - measured: 680 frames/s through the wide kernels, 500 with chain launches wherever an engine had
      // two lanes or fewer ready.  So: the whole executor's busy lanes, not this engine's.
      int busy_lanes = 0;
      for (const Lane& l : lanes_) busy_lanes += l.active && (!l.fifo.empty() || l.launched != l.completed);
      const bool chains_allowed = is_light(engine) && busy_lanes <= chain_max_lanes_;  // (a chain lane's keyframe may have been routed to a heavy engine)
      // per lane: one picture and the conversions queued behind it, or the
      // leading conversions when the head of the queue is a conversion
      for (size_t li = 0; li < lanes_.size(); ++li) {
        Lane& l = lanes_[li];
        if (debug && l.active) {
          const int cls = eligible(l, engine) ? 0
                          : !l.fifo.empty() ? 1                       // queued for / held by another engine
                          : l.launched != l.completed ? (l.waiters ? 2 : 3)  // in flight: thread blocked / parsing ahead
                          : 4;                                        // nothing in flight: thread parsing or idle
          ++lane_state[cls];
        }
        int taken = 0, pictures = 0;
        used_slots.clear();
        while (eligible(l, engine)) {
          Item& it = l.fifo.front();
          if (it.is_picture) {
            if (pics.size() >= kMaxJobs) break;
            if (!l.chain || !chains_allowed) {
              if (taken > 0) break;  // one picture per lane and batch, then the conversions behind it
            } else {
              // A chain: the lane's next pictures as long as nothing but pictures has been taken, each predicted from
              // the ones before inside the launch.  It ends in front of a picture that would overwrite a surface an
              // earlier one of the chain writes or reads (in a chain "earlier in the lane" no longer means "finished"),
              // behind a picture without deblocking (what follows it has no row progress to go by), and where the
              // unit table is full.
              if (taken > pictures) break;
              const PicSubmit& p = it.pic;
              if (pictures > 0) {
                if (!pics.back().desc.do_deblock) break;
                bool clash = false;
                for (int sl : used_slots) clash |= sl == p.desc.pp.cur_slot;
                if (clash) break;
              }
              const size_t units = chain_units(p);
              if (chain_words + units > kUnitWords - 8) break;
              chain_words += units;
              used_slots.push_back(p.desc.pp.cur_slot);
              for (int i = 0; i < p.num_ref_slots; ++i) used_slots.push_back(p.ref_slots[i]);
              any_chain = true;
            }
            pics.push_back(it.pic);
            pic_lane.push_back(static_cast<int>(li));
            pic_pos.push_back(pictures);
            ++pictures;
          } else {
            if (convs.size() >= kMaxJobs) break;
            convs.push_back(it.conv);
          }
          l.fifo.pop_front();
          ++l.launched;
          l.last_engine = engine;
          ++taken;
        }
        if (taken) batch.lane_counts.emplace_back(static_cast<int>(li), taken);
      }
      batch.in_flight = true;
      batch.chain = any_chain;
      batch.retried = batch.retry_pending = false;
      batch.slot = slot;
      batch.retry_mbs.clear();
      batch.retry_convs.clear();
      if (any_chain) {
        for (const PicSubmit& p : pics) batch.retry_mbs.push_back(p.num_mbs);
        batch.retry_convs = convs;
      }
      batch.owner = this;
      batch.engine = engine;
      batch.ticket = ++next_ticket_;
      batch.flag = done_flags_ + engine * kInflight + slot;
      batch.fault = fault_flags_ + engine * kInflight + slot;
      *batch.fault = 0;
      ++en.inflight;
      ++total_inflight_;
      ++stats_.batches;
      stats_.pictures += pics.size();
      stats_.conversions += convs.size();
    }
    Batch& batch = en.batches[slot];
    trace::event(any_chain ? "chain batch formed" : "batch formed", static_cast<long>(pics.size()), static_cast<long>(convs.size()));
    ++n_batches;
    n_pics += pics.size();
    n_convs += convs.size();
    {
      size_t n = pics.size();
      int b = n == 0 ? 0 : n <= 1 ? 1 : n <= 2 ? 2 : n <= 4 ? 3 : n <= 8 ? 4 : n <= 16 ? 5 : n <= 32 ? 6 : n <= 64 ? 7 : 8;
      hist[b]++;
    }
    hipError_t e = hipSuccess;
    if (!pics.empty()) {
      int max_mbs = 0, max_rows = 0;
      bool any_inter = false, any_intra = false, any_deblock = false, any_mv = false;
      uint64_t bytes[4] = {0, 0, 0, 0};
      for (size_t i = 0; i < pics.size(); ++i) {
        const PicSubmit& p = pics[i];
        JobDesc& jd = en.job_table_host[slot][i];
        jd = p.desc;
        jd.num_uploads = prestage_ ? 0 : p.ncopies;
        jd.ticket_mv = jd.ticket_db = jd.ticket_in = 0;
        jd.fault = const_cast<uint32_t*>(batch.fault);
        jd.chain_ctl = any_chain ? en.unit_table_dev[slot] : nullptr;
        for (int c = 0; c < p.ncopies; ++c) {
          jd.uploads[c].dst = p.copies[c].dst;
          jd.uploads[c].src = p.copies[c].src;
          jd.uploads[c].bytes = p.copies[c].bytes;
        }
        max_mbs = p.num_mbs > max_mbs ? p.num_mbs : max_mbs;
        max_rows = p.desc.pp.mb_height > max_rows ? p.desc.pp.mb_height : max_rows;
        any_inter |= p.desc.do_inter != 0;
        any_intra |= p.desc.do_intra != 0;
        any_deblock |= p.desc.do_deblock != 0;
        any_mv |= p.desc.do_mv != 0;
        bytes[0] += p.bytes_inter;
        bytes[1] += p.bytes_intra;
        bytes[2] += p.bytes_deblock;
        bytes[3] += p.bytes_mv;
      }
      if (any_chain) {
        // ---- chain launch: which job of the table writes which surface, and the two unit lists in dependency order
        // (position by position of the lanes' chains; a picture's inter runs / intra bands / deblocking bands in turn)
        int max_pos = 0;
        for (size_t i = 0; i < pics.size(); ++i) {
          JobDesc& jd = en.job_table_host[slot][i];
          for (int k = 0; k < 256; ++k) jd.slot_job[k] = -1;
          for (size_t j = 0; j < i; ++j)
            if (pic_lane[j] == pic_lane[i]) jd.slot_job[pics[j].desc.pp.cur_slot] = static_cast<int16_t>(j);
          max_pos = pic_pos[i] > max_pos ? pic_pos[i] : max_pos;
        }
        uint32_t* table = en.unit_table_host[slot];
        table[0] = table[1] = table[2] = table[3] = 0;  // the tickets
        uint32_t* rows = table + 4;
        uint32_t nrows = 0, ninter = 0;
        for (int pos = 0; pos <= max_pos; ++pos) {
          for (int stage = 0; stage < 2; ++stage)
            for (size_t i = 0; i < pics.size(); ++i) {
              if (pic_pos[i] != pos || !(stage ? pics[i].desc.do_deblock : pics[i].desc.do_intra)) continue;
              const int bands = (pics[i].desc.pp.mb_height + 3) / 4;
              for (int b = 0; b < bands; ++b) rows[nrows++] = static_cast<uint32_t>(i) << 8 | static_cast<uint32_t>(stage) << 7 | static_cast<uint32_t>(b);
            }
        }
        uint32_t* inter = rows + nrows;
        for (int pos = 0; pos <= max_pos; ++pos)
          for (size_t i = 0; i < pics.size(); ++i) {
            if (pic_pos[i] != pos || !pics[i].desc.do_inter) continue;
            const int runs = pics[i].desc.pp.mb_height * ((pics[i].desc.pp.mb_width + kChainRun - 1) / kChainRun);  // per row
            for (int r = 0; r < runs; ++r) inter[ninter++] = static_cast<uint32_t>(i) << 20 | static_cast<uint32_t>(r);
          }
        ChainArgs args;
        args.jobs = en.job_table_dev[slot];
        args.tickets = en.unit_table_dev[slot];
        args.row_units = en.unit_table_dev[slot] + 4;
        args.inter_units = en.unit_table_dev[slot] + 4 + nrows;
        args.num_row_units = nrows;
        args.num_inter_units = ninter;
        args.bytes = bytes[0] + bytes[1] + bytes[2];
        e = launch_table_copy(en.job_table_dev[slot], en.job_table_host[slot], sizeof(JobDesc) * pics.size(), en.stream);
        if (e == hipSuccess)
          e = launch_table_copy(en.unit_table_dev[slot], en.unit_table_host[slot], sizeof(uint32_t) * (4 + nrows + ninter), en.stream);
        if (!prestage_) {
          const uint64_t b3[3] = {bytes[3], 0, 0};
          if (e == hipSuccess) e = launch_upload(en.job_table_dev[slot], static_cast<int>(pics.size()), en.stream);
          if (e == hipSuccess)
            e = launch_pre(en.job_table_dev[slot], static_cast<int>(pics.size()), max_mbs, max_rows, any_mv, any_deblock, any_intra, b3,
                           en.stream);
        }
        // a workgroup is four row waves + four inter waves; more of them than the device holds at once only queue
        const uint32_t want = std::max<uint32_t>(nrows, (ninter + 3) / 4);
        if (e == hipSuccess) e = launch_chain(args, static_cast<int>(std::min<uint32_t>(std::max<uint32_t>(want, 1u), 256u)), en.stream);
        ++n_chain_batches;
        n_chain_pics += pics.size();
        max_chain = std::max<uint64_t>(max_chain, static_cast<uint64_t>(max_pos + 1));
      } else {
      e = launch_table_copy(en.job_table_dev[slot], en.job_table_host[slot], sizeof(JobDesc) * pics.size(), en.copy_stream);
      if (prestage_) {
        // records uploaded, vectors / boundary strengths / intra residuals computed by the records-only stage
        const uint64_t b3[3] = {bytes[0], bytes[1], bytes[2]};
        if (e == hipSuccess)
          e = launch_main(en.job_table_dev[slot], static_cast<int>(pics.size()), max_mbs, max_rows, any_inter, any_intra,
                          any_deblock, b3, en.stream);
      } else {
        if (e == hipSuccess) e = launch_upload(en.job_table_dev[slot], static_cast<int>(pics.size()), en.copy_stream);
        if (en.copy_stream != en.stream) {
          if (e == hipSuccess) e = hipEventRecord(en.copied[slot], en.copy_stream);
          if (e == hipSuccess) e = hipStreamWaitEvent(en.stream, en.copied[slot], 0);
        }
        if (e == hipSuccess)
          e = launch_batch(en.job_table_dev[slot], static_cast<int>(pics.size()), max_mbs, max_rows, any_inter, any_intra,
                           any_deblock, any_mv, bytes, en.stream);
      }
      }
    }
    if (!convs.empty() && e == hipSuccess) {
      int mw = 0, mh = 0;
      uint64_t bytes = 0;
      for (size_t i = 0; i < convs.size(); ++i) {
        en.conv_table_host[slot][i] = convs[i].desc;
        mw = convs[i].desc.width > mw ? convs[i].desc.width : mw;
        mh = convs[i].desc.height > mh ? convs[i].desc.height : mh;
        bytes += convs[i].bytes;
      }
      e = launch_table_copy(en.conv_table_dev[slot], en.conv_table_host[slot], sizeof(ConvertDesc) * convs.size(), en.stream);
      if (e == hipSuccess)
        e = launch_convert_batch(en.conv_table_dev[slot], static_cast<int>(convs.size()), mw, mh, bytes, en.stream);
      for (size_t i = 0; i < convs.size() && e == hipSuccess; ++i)
        if (convs[i].d2h_dst)
          e = hipMemcpyAsync(convs[i].d2h_dst, convs[i].desc.dst, convs[i].d2h_bytes, hipMemcpyDeviceToHost, en.stream);
    }
    if (e == hipSuccess) {
      if (use_hostfunc_) {
        e = hipLaunchHostFunc(en.stream, &GpuExecutor::on_batch_done, &batch);
      } else {
        e = launch_done(const_cast<uint32_t*>(batch.flag), batch.ticket, en.stream);
        cv_poll_.notify_one();
      }
    }
    if (e != hipSuccess) {
      // whatever was enqueued must drain before the lanes may reuse their buffers
      (void)hipStreamSynchronize(en.stream);
      complete(&batch, e);
    }
  }
  if (debug) {
    fprintf(stderr, "[executor %s%d] batches %llu pics %llu convs %llu | batch size hist 0:%llu 1:%llu 2:%llu <=4:%llu <=8:%llu <=16:%llu <=32:%llu <=64:%llu >64:%llu\n",
            is_light(engine) ? "light" : "heavy", engine, (unsigned long long)n_batches, (unsigned long long)n_pics, (unsigned long long)n_convs,
            (unsigned long long)hist[0], (unsigned long long)hist[1], (unsigned long long)hist[2],
            (unsigned long long)hist[3], (unsigned long long)hist[4], (unsigned long long)hist[5], (unsigned long long)hist[6],
            (unsigned long long)hist[7], (unsigned long long)hist[8]);
    if (n_chain_batches)
      fprintf(stderr, "[executor %s%d] chain launches %llu with %llu pictures, longest chain %llu\n", is_light(engine) ? "light" : "heavy", engine,
              (unsigned long long)n_chain_batches, (unsigned long long)n_chain_pics, (unsigned long long)max_chain);
    if (n_batches)
      fprintf(stderr, "[executor %s%d] lanes at batch formation (avg): taken %.1f, other-engine %.1f, in-flight+blocked %.1f, in-flight+parsing %.1f, idle/parsing %.1f\n",
              is_light(engine) ? "light" : "heavy", engine, lane_state[0] / (double)n_batches, lane_state[1] / (double)n_batches,
              lane_state[2] / (double)n_batches, lane_state[3] / (double)n_batches, lane_state[4] / (double)n_batches);
  }
  (void)hipStreamSynchronize(en.stream);  // outstanding completions run before the thread goes away
}

}  // namespace amd
}  // namespace hwang

extern "C" void hwang_shutdown(void) { hwang::amd::shutdown_executors(); }

// Replay hooks (include/hwang_amd.h): see GpuExecutor::keep_recent / replay.
extern "C" int hwang_replay_keep(int device, int pictures) {
  std::string error;
  hwang::amd::GpuExecutor* ex = hwang::amd::GpuExecutor::get(device, &error);
  if (!ex) return 1;
  ex->keep_recent(pictures);
  return 0;
}
extern "C" int hwang_replay_run(int device, int width, int launches, int chain, double out[4], char* error, size_t error_capacity) {
  std::string err;
  hwang::amd::GpuExecutor* ex = hwang::amd::GpuExecutor::get(device, &err);
  hwang::amd::GpuExecutor::ReplayResult r;
  if (!ex || !ex->replay(width, launches, chain != 0, &r, &err)) {
    if (error && error_capacity) snprintf(error, error_capacity, "%s", err.c_str());
    return 1;
  }
  out[0] = r.pictures_available;
  out[1] = r.width;
  out[2] = r.launches;
  out[3] = r.seconds;
  return 0;
}
