#include "hevc_engine.h"

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <map>

#include <pthread.h>

namespace hwang {

hipError_t launch_yuv420_to_rgb24(const uint8_t* y, const uint8_t* u, const uint8_t* v, int y_stride, int c_stride, int width, int height,
                                  uint8_t* rgb, hipStream_t stream);

namespace amd {
namespace hevc {

hipError_t launch_batch(const PicParams* const* pp, const Batch& batch, int n, hipStream_t stream);

namespace {
// batches in flight per stream: one running, one queued behind it; the third is being collected
constexpr size_t kBatchesInFlightDefault = 2;
std::mutex g_engines_mu;
std::map<int, std::weak_ptr<Engine>> g_engines;
}  // namespace

std::shared_ptr<Engine> Engine::for_device(int device) {
  std::lock_guard<std::mutex> lk(g_engines_mu);
  std::shared_ptr<Engine> e = g_engines[device].lock();
  if (!e) {
    e.reset(new Engine(device));
    g_engines[device] = e;
  }
  return e;
}

Engine::Engine(int device) : device_(device) { thread_ = std::thread([this] { run(); }); }

Engine::~Engine() {
  {
    std::lock_guard<std::mutex> lk(mu_);
    stop_ = true;
  }
  cv_work_.notify_all();
  if (thread_.joinable()) thread_.join();
}

int Engine::add_client() {
  std::lock_guard<std::mutex> lk(mu_);
  for (size_t i = 0; i < clients_.size(); ++i)
    if (!clients_[i].used) {
      clients_[i] = Client();
      clients_[i].used = true;
      return static_cast<int>(i);
    }
  clients_.emplace_back();
  clients_.back().used = true;
  return static_cast<int>(clients_.size()) - 1;
}

void Engine::remove_client(int client) {
  std::lock_guard<std::mutex> lk(mu_);
  if (client >= 0 && client < static_cast<int>(clients_.size())) clients_[client].used = false;
}

void Engine::submit(int client, EngineJob* job) {
  {
    std::lock_guard<std::mutex> lk(mu_);
    job->launched.store(0, std::memory_order_relaxed);
    job->failed = false;
    clients_[client].queue.push_back(job);
  }
  cv_work_.notify_one();
}

bool Engine::wait_launched(EngineJob* job) {
  if (!job->launched.load(std::memory_order_acquire)) {
    std::unique_lock<std::mutex> lk(mu_);
    cv_done_.wait(lk, [&] { return job->launched.load(std::memory_order_acquire) != 0; });
  }
  return !job->failed;
}

bool Engine::drain(int client) {
  std::unique_lock<std::mutex> lk(mu_);
  cv_done_.wait(lk, [&] { return clients_[client].queue.empty() && !clients_[client].taking; });
  return error_.empty();
}

void Engine::settled(int client) {
  std::lock_guard<std::mutex> lk(mu_);
  clients_[client].last_stream = -1;
}

std::string Engine::error() {
  std::lock_guard<std::mutex> lk(mu_);
  return error_;
}

void Engine::run() {
  pthread_setname_np(pthread_self(), "hw-hevc-engine");
  (void)hipSetDevice(device_);
  bool ok = hipStreamCreateWithFlags(&copy_stream_, hipStreamNonBlocking) == hipSuccess;
  for (hipStream_t& s : stream_) ok = ok && hipStreamCreateWithFlags(&s, hipStreamNonBlocking) == hipSuccess;
  auto new_event = [&]() {
    hipEvent_t e = nullptr;
    if (!event_pool_.empty()) {
      e = event_pool_.back();
      event_pool_.pop_back();
    } else if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) {
      e = nullptr;
    }
    return e;
  };
  // (experiment knobs: how many of the light / heavy streams are used, how many batches each keeps in flight)
  auto knob = [](const char* name, int dflt, int lo, int hi) {
    const char* e = std::getenv(name);
    const int v = e ? std::atoi(e) : dflt;
    return v < lo ? lo : v > hi ? hi : v;
  };
  const int light_streams = knob("HWANG_HEVC_LIGHT_STREAMS", 2, 1, kLightStreams), heavy_streams = knob("HWANG_HEVC_HEAVY_STREAMS", 2, 1, kHeavyStreams);
  const size_t light_room = static_cast<size_t>(knob("HWANG_HEVC_LIGHT_INFLIGHT", static_cast<int>(kBatchesInFlightDefault), 1, 4));
  const size_t heavy_room = static_cast<size_t>(knob("HWANG_HEVC_HEAVY_INFLIGHT", static_cast<int>(kBatchesInFlightDefault), 1, 4));
  std::vector<EngineJob*> pictures[kStreams], outputs[kStreams];
  std::vector<int> touched;
  // HWANG_HEVC_ENGINE_STATS=1: what the batches looked like, on stderr when the engine goes
  const bool stats = std::getenv("HWANG_HEVC_ENGINE_STATS") != nullptr;
  const bool copy_on_own_stream = !std::getenv("HWANG_HEVC_COPY_STREAM") || std::atoi(std::getenv("HWANG_HEVC_COPY_STREAM")) != 0;
  int64_t n_batches[2] = {0, 0}, n_pictures[2] = {0, 0}, n_outputs = 0, issue_ns = 0, copy_ns = 0, issue_max_ns = 0, idle_rounds = 0;
  auto now_ns = [] { return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  std::unique_lock<std::mutex> lk(mu_);
  if (!ok) error_ = "the HEVC engine could not create its streams";
  for (;;) {
    // which batches are through (the engine's own lists: no lock - a lane that hands over a job must not wait for a
    // HIP call; with the queries under the lock the lanes spent 0.5 ms per picture in submit())
    lk.unlock();
    for (int s = 0; s < kStreams; ++s)
      while (!inflight_[s].empty() && hipEventQuery(inflight_[s].front()) != hipErrorNotReady) {
        event_pool_.push_back(inflight_[s].front());
        inflight_[s].pop_front();
        ++completed_seq_[s];
      }
    while (!copy_inflight_.empty() && hipEventQuery(copy_inflight_.front()) != hipErrorNotReady) {
      event_pool_.push_back(copy_inflight_.front());
      copy_inflight_.pop_front();
    }
    lk.lock();
    // ---- collect what may go now
    for (int s = 0; s < kStreams; ++s) {
      pictures[s].clear();
      outputs[s].clear();
    }
    touched.clear();
    bool waiting = false, queued = false, any = false;
    for (size_t ci = 0; ci < clients_.size(); ++ci) {
      Client& c = clients_[ci];
      bool took_picture = false, took_output = false, took = false;
      while (!c.queue.empty()) {
        queued = true;
        EngineJob* j = c.queue.front();
        if (!error_.empty()) {  // nothing runs any more: fail what is queued
          j->failed = true;
          j->launched.store(1, std::memory_order_release);
          c.queue.pop_front();
          continue;
        }
        const bool heavy = j->kind == EngineJob::kPicture && j->pp.intra_rows;
        const int first = heavy ? kLightStreams : 0, last = heavy ? kLightStreams + heavy_streams : light_streams;
        // Which stream of its kind: the one the decoder's last job went to if it is of that kind and has room (stream
        // order then, no waiting) - else the emptiest with room, which the job may enter once its predecessor's batch in
        // the other stream is through.
        int s = -1;
        const size_t room = heavy ? heavy_room : light_room;
        if (c.last_stream >= first && c.last_stream < last && inflight_[c.last_stream].size() < room) s = c.last_stream;
        if (s < 0)
          for (int k = first; k < last; ++k)
            if (inflight_[k].size() < room && (s < 0 || inflight_[k].size() + pictures[k].size() < inflight_[s].size() + pictures[s].size())) s = k;
        if (s < 0) break;  // every stream of the kind is full
        // one picture of a decoder per round, and none behind a fetched frame of it (a batch's pictures run before its
        // conversions, and the picture may be going into the very surface the frame is read from)
        if (j->kind == EngineJob::kPicture && (took_picture || took_output || pictures[s].size() >= static_cast<size_t>(kMaxBatch))) break;
        if (c.last_stream >= 0 && c.last_stream != s && completed_seq_[c.last_stream] < c.last_seq) {
          // its predecessor went to another stream: not before that one's batch has finished
          waiting = true;
          break;
        }
        c.queue.pop_front();
        (j->kind == EngineJob::kPicture ? pictures[s] : outputs[s]).push_back(j);
        took_picture = took_picture || j->kind == EngineJob::kPicture;
        took_output = took_output || j->kind != EngineJob::kPicture;
        took = any = true;
        c.last_stream = s;
        c.last_seq = launched_seq_[s] + 1;  // (the batch this round makes for the stream)
      }
      if (took) {
        c.taking = true;  // (drain: popped, not yet in a stream)
        touched.push_back(static_cast<int>(ci));
      }
    }
    if (!error_.empty()) cv_done_.notify_all();
    if (!any) {
      if (stop_ && !queued) break;
      ++idle_rounds;
      bool busy = false;
      for (int s = 0; s < kStreams; ++s) busy = busy || !inflight_[s].empty();
      if (waiting || queued || busy)
        cv_work_.wait_for(lk, std::chrono::microseconds(50));
      else
        cv_work_.wait(lk);
      continue;
    }
    lk.unlock();
    // ---- into the streams
    const int64_t t_issue = stats ? now_ns() : 0;
    hipError_t e = hipSuccess;
    bool copied = false;
    for (int s = 0; s < kStreams; ++s)
      for (EngineJob* j : pictures[s])
        for (const EngineJob::Copy& cp : j->copy)
          if (cp.bytes && e == hipSuccess) {
            e = hipMemcpyAsync(cp.dst, cp.src, cp.bytes, hipMemcpyHostToDevice, copy_on_own_stream ? copy_stream_ : stream_[s]);
            copied = copy_on_own_stream;
          }
    if (stats) copy_ns += now_ns() - t_issue;
    hipEvent_t copies_done = nullptr;
    if (copied && e == hipSuccess) {
      copies_done = new_event();
      e = copies_done ? hipEventRecord(copies_done, copy_stream_) : hipErrorOutOfMemory;
    }
    hipEvent_t batch_event[kStreams] = {};
    for (int s = 0; s < kStreams && e == hipSuccess; ++s) {
      if (pictures[s].empty() && outputs[s].empty()) continue;
      if (!pictures[s].empty()) {
        if (copies_done) e = hipStreamWaitEvent(stream_[s], copies_done, 0);
        Batch batch;
        const PicParams* pps[kMaxBatch];
        int n = 0;
        for (EngineJob* j : pictures[s]) {
          if (j->clear_resid && e == hipSuccess) e = hipMemsetAsync(j->clear_resid, 0, j->clear_resid_bytes, stream_[s]);
          pps[n] = &j->pp;
          batch.pic[n++] = j->pp_dev;
        }
        if (e == hipSuccess) e = launch_batch(pps, batch, n, stream_[s]);
        for (EngineJob* j : pictures[s])
          if (e == hipSuccess) e = hipEventRecord(j->done, stream_[s]);
      }
      for (EngineJob* j : outputs[s]) {
        if (e != hipSuccess) break;
        const size_t w = j->width, h = j->height, cw = (w + 1) / 2, ch = (h + 1) / 2;
        if (j->kind == EngineJob::kRgb) {
          uint8_t* rgb = j->out_on_device ? j->out : j->rgb_dev;
          e = launch_yuv420_to_rgb24(j->y, j->u, j->v, j->y_pitch, j->y_pitch / 2, j->width, j->height, rgb, stream_[s]);
          if (e == hipSuccess && !j->out_on_device) e = hipMemcpyAsync(j->out, rgb, w * h * 3, hipMemcpyDeviceToHost, stream_[s]);
        } else {
          const hipMemcpyKind kind = j->out_on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
          e = hipMemcpy2DAsync(j->out, w, j->y, static_cast<size_t>(j->y_pitch), w, h, kind, stream_[s]);
          if (e == hipSuccess) e = hipMemcpy2DAsync(j->out + w * h, cw, j->u, static_cast<size_t>(j->y_pitch / 2), cw, ch, kind, stream_[s]);
          if (e == hipSuccess) e = hipMemcpy2DAsync(j->out + w * h + cw * ch, cw, j->v, static_cast<size_t>(j->y_pitch / 2), cw, ch, kind, stream_[s]);
        }
        if (e == hipSuccess) e = hipEventRecord(j->done, stream_[s]);
      }
      if (e == hipSuccess) {
        batch_event[s] = new_event();
        e = batch_event[s] ? hipEventRecord(batch_event[s], stream_[s]) : hipErrorOutOfMemory;
      }
    }
    if (stats) {
      const int64_t d = now_ns() - t_issue;
      issue_ns += d;
      issue_max_ns = d > issue_max_ns ? d : issue_max_ns;
      for (int s = 0; s < kStreams; ++s) {
        if (!pictures[s].empty()) {
          ++n_batches[s >= kLightStreams];
          n_pictures[s >= kLightStreams] += static_cast<int64_t>(pictures[s].size());
        }
        n_outputs += static_cast<int64_t>(outputs[s].size());
      }
    }
    lk.lock();
    if (copies_done) copy_inflight_.push_back(copies_done);
    for (int s = 0; s < kStreams; ++s) {
      if (pictures[s].empty() && outputs[s].empty()) continue;
      ++launched_seq_[s];
      if (batch_event[s]) inflight_[s].push_back(batch_event[s]);
      else ++completed_seq_[s];  // (the launch failed: nothing will ever fire; error_ ends the engine's work)
    }
    if (e != hipSuccess && error_.empty()) error_ = std::string("the HEVC engine failed to launch a batch: ") + hipGetErrorString(e);
    for (int s = 0; s < kStreams; ++s)
      for (auto* list : {&pictures[s], &outputs[s]})
        for (EngineJob* j : *list) {
          j->failed = e != hipSuccess;
          j->launched.store(1, std::memory_order_release);
        }
    for (int ci : touched) clients_[ci].taking = false;
    cv_done_.notify_all();
  }
  lk.unlock();
  if (stats)
    std::fprintf(stderr, "[hevc engine %d] light: %lld pictures in %lld batches; heavy: %lld in %lld; %lld fetched frames; issuing took %.1f ms (copies %.1f ms), "
                 "longest round %.2f ms; %lld empty rounds\n", device_, (long long)n_pictures[0], (long long)n_batches[0], (long long)n_pictures[1], (long long)n_batches[1],
                 (long long)n_outputs, issue_ns / 1e6, copy_ns / 1e6, issue_max_ns / 1e6, (long long)idle_rounds);
  for (int k = -1; k < kStreams; ++k) {
    hipStream_t s = k < 0 ? copy_stream_ : stream_[k];
    if (s) {
      (void)hipStreamSynchronize(s);
      (void)hipStreamDestroy(s);
    }
  }
  for (auto& fl : inflight_)
    for (hipEvent_t e : fl) (void)hipEventDestroy(e);
  for (hipEvent_t e : copy_inflight_) (void)hipEventDestroy(e);
  for (hipEvent_t e : event_pool_) (void)hipEventDestroy(e);
}

}  // namespace hevc
}  // namespace amd
}  // namespace hwang
