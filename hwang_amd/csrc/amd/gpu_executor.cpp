#include "gpu_executor.h"

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <memory>

namespace hwang {

hipError_t launch_convert_batch(const amd::ConvertDesc* reqs_dev, int n, int max_width, int max_height,
                                uint64_t bytes, hipStream_t stream);

namespace amd {

hipError_t launch_upload(const JobDesc* jobs, int njobs, hipStream_t stream);
hipError_t launch_batch(const JobDesc* jobs, int njobs, int max_mbs, bool any_inter, bool any_intra,
                        bool any_deblock, const uint64_t bytes[3], hipStream_t stream);

namespace {
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
  for (Engine& en : engines_) {
    if (hipStreamCreateWithFlags(&en.stream, hipStreamNonBlocking) != hipSuccess ||
        hipStreamCreateWithFlags(&en.copy_stream, hipStreamNonBlocking) != hipSuccess)
      return fail("cannot create a stream");
    for (int i = 0; i < kInflight; ++i) {
      if (hipHostMalloc(reinterpret_cast<void**>(&en.job_table_host[i]), sizeof(JobDesc) * kMaxJobs, hipHostMallocDefault) != hipSuccess ||
          hipMalloc(reinterpret_cast<void**>(&en.job_table_dev[i]), sizeof(JobDesc) * kMaxJobs) != hipSuccess ||
          hipHostMalloc(reinterpret_cast<void**>(&en.conv_table_host[i]), sizeof(ConvertDesc) * kMaxJobs, hipHostMallocDefault) != hipSuccess ||
          hipMalloc(reinterpret_cast<void**>(&en.conv_table_dev[i]), sizeof(ConvertDesc) * kMaxJobs) != hipSuccess ||
          hipEventCreateWithFlags(&en.batches[i].done, hipEventDisableTiming) != hipSuccess ||
          hipEventCreateWithFlags(&en.copied[i], hipEventDisableTiming) != hipSuccess)
        return fail("out of memory for job tables");
    }
  }
  for (int e = 0; e < kEngines; ++e) engines_[e].thread = std::thread(&GpuExecutor::run, this, e);
  return true;
}

void GpuExecutor::stop() {
  {
    std::lock_guard<std::mutex> lk(mu_);
    if (stop_) return;
    stop_ = true;
  }
  cv_work_.notify_all();
  for (Engine& en : engines_)
    if (en.thread.joinable()) en.thread.join();
}

int GpuExecutor::register_lane() {
  std::lock_guard<std::mutex> lk(mu_);
  for (size_t i = 0; i < lanes_.size(); ++i)
    if (!lanes_[i].active && lanes_[i].fifo.empty() && lanes_[i].completed == lanes_[i].submitted) {
      lanes_[i] = Lane();
      lanes_[i].active = true;
      return static_cast<int>(i);
    }
  lanes_.emplace_back();
  lanes_.back().active = true;
  return static_cast<int>(lanes_.size() - 1);
}

void GpuExecutor::unregister_lane(int lane) {
  std::unique_lock<std::mutex> lk(mu_);
  cv_done_.wait(lk, [&] { return lanes_[lane].completed == lanes_[lane].submitted || !error_.empty() || stop_; });
  lanes_[lane].active = false;
}

uint64_t GpuExecutor::submit_picture(int lane, const PicSubmit& job) {
  uint64_t seq;
  {
    std::lock_guard<std::mutex> lk(mu_);
    Item it;
    it.is_picture = true;
    it.pic = job;
    lanes_[lane].fifo.push_back(it);
    seq = ++lanes_[lane].submitted;
  }
  cv_work_.notify_all();
  return seq;
}

uint64_t GpuExecutor::submit_convert(int lane, const ConvSubmit& job) {
  uint64_t seq;
  {
    std::lock_guard<std::mutex> lk(mu_);
    Item it;
    it.is_picture = false;
    it.conv = job;
    lanes_[lane].fifo.push_back(it);
    seq = ++lanes_[lane].submitted;
  }
  cv_work_.notify_all();
  return seq;
}

bool GpuExecutor::wait(int lane, uint64_t seq, std::string* error) {
  std::unique_lock<std::mutex> lk(mu_);
  cv_done_.wait(lk, [&] { return lanes_[lane].completed >= seq || !error_.empty() || stop_; });
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

GpuExecutor::Stats GpuExecutor::stats() {
  std::lock_guard<std::mutex> lk(mu_);
  return stats_;
}

void GpuExecutor::run(int engine) {
  (void)hipSetDevice(device_);
  Engine& en = engines_[engine];
  uint64_t batch_index = 0;
  auto retire = [&](Batch& b) {
    if (!b.in_flight) return;
    hipError_t e = hipEventSynchronize(b.done);
    std::lock_guard<std::mutex> lk(mu_);
    if (e != hipSuccess && error_.empty()) error_ = std::string("GpuExecutor: batch failed: ") + hipGetErrorString(e);
    for (auto& lc : b.lane_counts) lanes_[lc.first].completed += lc.second;
    b.lane_counts.clear();
    b.in_flight = false;
    cv_done_.notify_all();
    cv_work_.notify_all();  // the other engine may now take over some of these lanes
  };

  std::vector<PicSubmit> pics;
  std::vector<ConvSubmit> convs;
  const bool debug = getenv("HWANG_EXEC_DEBUG") != nullptr;
  auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  double t_retire = 0, t_idle = 0, t_launch = 0;
  uint64_t hist[9] = {0}, n_batches = 0, n_pics = 0, n_convs = 0;
  for (;;) {
    double t0 = now();
    const int slot = static_cast<int>(batch_index % kInflight);
    Batch& batch = en.batches[slot];
    retire(batch);  // the slot's previous occupant (kInflight batches ago)
    // Form the next batch late, not eagerly: while two batches are already
    // queued on the GPU, every lane keeps parsing, so by the time the older
    // one finishes most lanes have their next picture ready and the batch is
    // wide.  (Launching whatever is pending right away produced batches of
    // ~5 pictures that each held the GPU for a full pipeline latency.)
    if (en.batches[(slot + kInflight - 2) % kInflight].in_flight && en.batches[(slot + kInflight - 1) % kInflight].in_flight)
      retire(en.batches[(slot + kInflight - 2) % kInflight]);
    t_retire += now() - t0;
    t0 = now();

    pics.clear();
    convs.clear();
    {
      std::unique_lock<std::mutex> lk(mu_);
      auto have_work = [&] {
        for (const Lane& l : lanes_)
          if (eligible(l, engine)) return true;
        return false;
      };
      while (!stop_ && !have_work()) {
        // nothing to launch: finish what is in flight so waiters wake up
        bool any = false;
        for (int i = 0; i < kInflight; ++i) any |= en.batches[i].in_flight;
        if (any) {
          lk.unlock();
          for (int i = 1; i <= kInflight; ++i) retire(en.batches[(slot + i) % kInflight]);
          lk.lock();
          continue;
        }
        cv_work_.wait(lk);
      }
      if (stop_) break;
      // per lane: one picture and the conversions queued behind it, or the
      // leading conversions when the head of the queue is a conversion
      for (size_t li = 0; li < lanes_.size(); ++li) {
        Lane& l = lanes_[li];
        int taken = 0;
        bool took_picture = false;
        while (eligible(l, engine)) {
          Item& it = l.fifo.front();
          if (it.is_picture) {
            if (took_picture || taken > 0 || pics.size() >= kMaxJobs) break;
            pics.push_back(it.pic);
            took_picture = true;
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
      ++stats_.batches;
      stats_.pictures += pics.size();
      stats_.conversions += convs.size();
    }
    ++n_batches;
    n_pics += pics.size();
    n_convs += convs.size();

    t_idle += now() - t0;
    t0 = now();
    {
      size_t n = pics.size();
      int b = n == 0 ? 0 : n <= 1 ? 1 : n <= 2 ? 2 : n <= 4 ? 3 : n <= 8 ? 4 : n <= 16 ? 5 : n <= 32 ? 6 : n <= 64 ? 7 : 8;
      hist[b]++;
    }
    hipError_t e = hipSuccess;
    if (!pics.empty()) {
      int max_mbs = 0;
      bool any_inter = false, any_intra = false, any_deblock = false;
      uint64_t bytes[3] = {0, 0, 0};
      for (size_t i = 0; i < pics.size() && e == hipSuccess; ++i) {
        const PicSubmit& p = pics[i];
        JobDesc& jd = en.job_table_host[slot][i];
        jd = p.desc;
        jd.num_uploads = p.ncopies;
        for (int c = 0; c < p.ncopies; ++c) {
          jd.uploads[c].dst = p.copies[c].dst;
          jd.uploads[c].src = p.copies[c].src;
          jd.uploads[c].bytes = p.copies[c].bytes;
        }
        max_mbs = p.num_mbs > max_mbs ? p.num_mbs : max_mbs;
        any_inter |= p.desc.do_inter != 0;
        any_intra |= p.desc.do_intra != 0;
        any_deblock |= p.desc.do_deblock != 0;
        bytes[0] += p.bytes_inter;
        bytes[1] += p.bytes_intra;
        bytes[2] += p.bytes_deblock;
      }
      if (e == hipSuccess)
        e = hipMemcpyAsync(en.job_table_dev[slot], en.job_table_host[slot], sizeof(JobDesc) * pics.size(),
                           hipMemcpyHostToDevice, en.copy_stream);
      if (e == hipSuccess) e = launch_upload(en.job_table_dev[slot], static_cast<int>(pics.size()), en.copy_stream);
      if (e == hipSuccess) e = hipEventRecord(en.copied[slot], en.copy_stream);
      if (e == hipSuccess) e = hipStreamWaitEvent(en.stream, en.copied[slot], 0);
      if (e == hipSuccess)
        e = launch_batch(en.job_table_dev[slot], static_cast<int>(pics.size()), max_mbs, any_inter, any_intra,
                         any_deblock, bytes, en.stream);
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
      e = hipMemcpyAsync(en.conv_table_dev[slot], en.conv_table_host[slot], sizeof(ConvertDesc) * convs.size(),
                         hipMemcpyHostToDevice, en.stream);
      if (e == hipSuccess)
        e = launch_convert_batch(en.conv_table_dev[slot], static_cast<int>(convs.size()), mw, mh, bytes, en.stream);
      for (size_t i = 0; i < convs.size() && e == hipSuccess; ++i)
        if (convs[i].d2h_dst)
          e = hipMemcpyAsync(convs[i].d2h_dst, convs[i].desc.dst, convs[i].d2h_bytes, hipMemcpyDeviceToHost, en.stream);
    }
    if (e == hipSuccess) e = hipEventRecord(batch.done, en.stream);
    if (e != hipSuccess) {
      std::lock_guard<std::mutex> lk(mu_);
      if (error_.empty()) error_ = std::string("GpuExecutor: launch failed: ") + hipGetErrorString(e);
      for (auto& lc : batch.lane_counts) lanes_[lc.first].completed += lc.second;
      batch.lane_counts.clear();
      cv_done_.notify_all();
      cv_work_.notify_all();
    } else {
      batch.in_flight = true;
    }
    ++batch_index;
    t_launch += now() - t0;
  }
  if (debug)
    fprintf(stderr, "[executor %s] batches %llu pics %llu convs %llu | retire %.3fs idle %.3fs launch %.3fs | batch size hist 0:%llu 1:%llu 2:%llu <=4:%llu <=8:%llu <=16:%llu <=32:%llu <=64:%llu >64:%llu\n",
            engine ? "heavy" : "light", (unsigned long long)n_batches, (unsigned long long)n_pics, (unsigned long long)n_convs,
            t_retire, t_idle, t_launch, (unsigned long long)hist[0], (unsigned long long)hist[1], (unsigned long long)hist[2],
            (unsigned long long)hist[3], (unsigned long long)hist[4], (unsigned long long)hist[5], (unsigned long long)hist[6],
            (unsigned long long)hist[7], (unsigned long long)hist[8]);
  for (int i = 0; i < kInflight; ++i) retire(en.batches[i]);
  (void)hipStreamSynchronize(en.stream);
}

}  // namespace amd
}  // namespace hwang
