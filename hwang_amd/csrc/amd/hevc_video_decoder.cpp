// The HEVC half of the MI355X backend behind VideoDecoderInterface (the reference's
// software backend takes HEVC tracks too, impls/software/software_video_decoder.cpp:128-132).  AMDVideoDecoder
// (amd_video_decoder.cpp) hands every call over to an object of this class once configure() has seen an HEVC format
// name.  A decoder owns its surfaces, its staging sets (pinned host memory + device copies of a picture's records,
// written by the host parser hevc/hevc_parser.cpp) and its scratch planes; what it does NOT own is a stream: every picture
// and every fetched frame is a job handed to the GPU's submission engine (hevc_engine.h), which batches the jobs of all
// decoders into a few streams.
#include "hevc_video_decoder.h"

#include "../hevc/hevc_parser.h"
#include "../sparse_hints.h"
#include "../../../include/hwang_amd.h"
#include "hevc_engine.h"
#include "hevc_frame_job.h"
#include "hip_check.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <mutex>
#include <thread>
#include <vector>

namespace hwang {

namespace amd {
namespace hevc {
hipError_t launch_batch(const PicParams* const* pp, const Batch& batch, int n, hipStream_t stream);
}
}  // namespace amd

namespace {

namespace gh = amd::hevc;
// Staging sets per decoder: how many pictures the host parser may be ahead of the GPU.  An intra picture takes the GPU
// 5 - 10 ms from submission to its end and the next pictures parse in ~3 ms each: with three sets the thread stalled on
// the intra picture's set once per interval (1080p bench: 440 -> 460 frames/s with five).  A set is sized for the worst
// picture (1080p: 20 MB pinned + 20 MB of HBM), so large pictures keep three.
constexpr int kSets = 5;
constexpr size_t kSmallSetBytes = size_t(32) << 20;
constexpr int kOutputs = 6;  // fetched frames on their way (take): conversion, copy-out, for pageable caller memory a bounce buffer

// Waits for an event without holding a CPU: hipEventSynchronize spins for the whole wait (measured: the lanes' CPU time
// per picture was parse + stall to the millisecond, with hipEventBlockingSync or without), and the GPU box gives the
// process 16 CPUs that the parsers need.  A look every 100 us costs a picture's set at most that much latency.
std::atomic<long long> g_event_waits{0}, g_event_sleeps{0};  // (HWANG_HEVC_ENGINE_STATS: how often a wait found its event unfinished)
hipError_t sleepy_event_wait(hipEvent_t e) {
  g_event_waits.fetch_add(1, std::memory_order_relaxed);
  for (;;) {
    const hipError_t q = hipEventQuery(e);
    if (q != hipErrorNotReady) return q;
    g_event_sleeps.fetch_add(1, std::memory_order_relaxed);
    std::this_thread::sleep_for(std::chrono::microseconds(100));
  }
}

class HEVCVideoDecoder : public HevcDelegate, public hevc::PictureSink {
 public:
  explicit HEVCVideoDecoder(int device) : device_(device), engine_(gh::Engine::for_device(device)), client_(engine_->add_client()) {
    surf_.reserve(gh::kMaxSurfaces);  // (grows under mu_ on the feeding thread; never moves under the fetching one)
  }
  ~HEVCVideoDecoder() override {
    (void)hipSetDevice(device_);
    if (std::getenv("HWANG_HEVC_ENGINE_STATS") && parser_ && g_event_waits.exchange(0))
      std::fprintf(stderr, "[hevc decoders] event waits (all decoders, since the last line): %lld sleeps of 100 us\n", g_event_sleeps.exchange(0));
    if (std::getenv("HWANG_HEVC_ENGINE_STATS") && parser_ && (dbg_stalls_[0] || dbg_stalls_[1]))
      std::fprintf(stderr, "[hevc decoder %p] waited for a set %lld times behind an inter picture (submitted %.2f ms before, on average), %lld times behind an intra picture (%.2f ms)\n",
                   static_cast<void*>(this), (long long)dbg_stalls_[0], dbg_stalls_[0] ? dbg_age_ns_[0] / 1e6 / dbg_stalls_[0] : 0.0, (long long)dbg_stalls_[1],
                   dbg_stalls_[1] ? dbg_age_ns_[1] / 1e6 / dbg_stalls_[1] : 0.0);
    if (std::getenv("HWANG_HEVC_ENGINE_STATS") && parser_)
      std::fprintf(stderr, "[hevc decoder %p] pictures %lld: parse %.1f ms, waiting for a set %.1f ms (%.1f ms of it until the engine had launched the set's picture), submitting %.1f ms (%.1f ms of it handing the job to the engine), copying frames out of bounce buffers %.1f ms\n",
                   static_cast<void*>(this), (long long)(folded_decoded_ + parser_->pictures_decoded()), counters_.entropy_ns / 1e6, counters_.stall_ns / 1e6, launch_wait_ns_ / 1e6,
                   submit_ns_ / 1e6, engine_submit_ns_ / 1e6, copy_out_ns_ / 1e6);
    parser_.reset();
    (void)quiesce();
    free_all();
    if (fill_stream_) (void)hipStreamDestroy(fill_stream_);
    for (Output& o : outputs_) {
      if (o.host) (void)hipHostFree(o.host);
      if (o.job.done) (void)hipEventDestroy(o.job.done);
    }
    engine_->remove_client(client_);
  }

  Result configure(const VideoDecoderInterface::FrameInfo& info, const std::vector<uint8_t>& extradata) override {
    HWANG_HIP_RETURN(hipSetDevice(device_));
    if (parser_) {
      folded_decoded_ += parser_->pictures_decoded();
      folded_skipped_ += parser_->pictures_skipped();
    }
    parser_.reset();  // (releases its surfaces)
    HWANG_RETURN_ON_ERROR(quiesce());
    {
      std::lock_guard<std::mutex> lk(mu_);
      for (int slot : outq_)
        if (slot >= 0) surf_[slot].held &= ~kOutput;
      outq_.clear();
      buffered_.store(0);
    }
    parser_.reset(new hevc::Parser(this));
    if (!parser_->configure(extradata.data(), extradata.size())) return Result(false, "AMDVideoDecoder::configure (HEVC): " + parser_->error());
    if (parser_->width() != static_cast<int>(info.width) || parser_->height() != static_cast<int>(info.height))
      return Result(false, "AMDVideoDecoder::configure (HEVC): stream is " + std::to_string(parser_->width()) + "x" + std::to_string(parser_->height()) +
                               " but the container says " + std::to_string(info.width) + "x" + std::to_string(info.height));
    info_ = info;
    if (!pending_wanted_.empty()) parser_->set_wanted(pending_wanted_);
    if (!pending_keyframes_.empty()) parser_->set_keyframes(pending_keyframes_);
    return Result();
  }

  Result feed(const uint8_t* data, size_t size, bool /*keyframe*/) override {
    if (!size) return Result();
    if (!parser_) return Result(false, "AMDVideoDecoder::feed before configure");
    HWANG_HIP_RETURN(hipSetDevice(device_));
    {
      // (frames fetched earlier whose copies have landed go out of their bounce buffers now - unless the fetching thread
      // is at the ring this very moment: DecoderAutomata feeds and fetches from two threads)
      std::unique_lock<std::mutex> ol(out_mu_, std::try_to_lock);
      if (ol.owns_lock()) HWANG_RETURN_ON_ERROR(deliver_outputs(false));
    }
    ++counters_.samples_fed;
    const auto t0 = std::chrono::steady_clock::now();
    const int64_t stall0 = counters_.stall_ns, submit0 = submit_ns_;
    const bool ok = parser_->decode(data, size);
    // (entropy_ns: the host parser alone - waits for a staging set and the submission of the picture taken out)
    counters_.entropy_ns += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count() -
                            (counters_.stall_ns - stall0) - (submit_ns_ - submit0);
    if (!ok) {
      const std::string why = parser_->error();
      return Result(false, "AMDVideoDecoder::feed (HEVC): " + why);
    }
    return take_error();
  }

  Result flush() override {
    if (!parser_) return Result();
    HWANG_HIP_RETURN(hipSetDevice(device_));
    parser_->flush();
    pending_wanted_.clear();
    pending_keyframes_.clear();
    return take_error();
  }

  Result discard_frame() override {
    std::lock_guard<std::mutex> lk(mu_);
    if (outq_.empty()) return Result();
    const int slot = outq_.front();
    outq_.pop_front();
    buffered_.fetch_sub(1, std::memory_order_acq_rel);
    if (slot >= 0) surf_[slot].held &= ~kOutput;
    return Result();
  }

  Result get_frame(uint8_t* dst, size_t size) override { return take(dst, size, false); }
  Result get_frame_yuv(uint8_t* dst, size_t size) override { return take(dst, size, true); }
  int decoded_frames_buffered() override { return buffered_.load(std::memory_order_acquire); }

  Result wait_until_frames_copied() override {
    HWANG_HIP_RETURN(hipSetDevice(device_));
    HWANG_RETURN_ON_ERROR(quiesce());
    if (fault_host_ && *fault_host_) {
      *fault_host_ = 0;
      resid_dirty_ = true;
      return Result(false, "AMDVideoDecoder (HEVC): the intra wavefront timed out waiting for the row above; the picture is invalid");
    }
    return take_error();
  }

  void set_wanted_frames(const std::vector<uint64_t>& ordinals) override {
    pending_wanted_ = ordinals;
    if (parser_) parser_->set_wanted(ordinals);
  }
  void set_interval_keyframes(const std::vector<uint64_t>& ordinals) override {
    pending_keyframes_ = ordinals;
    if (parser_) parser_->set_keyframes(ordinals);
  }
  bool feed_would_wait() override {
    if (!engine_) return false;
    Set& set = sets_[next_set_];
    if (!set.busy) return false;
    if (!set.job.launched.load(std::memory_order_acquire)) return true;
    return !set.job.failed && hipEventQuery(set.done) == hipErrorNotReady;
  }
  BackendCounters counters() override {
    BackendCounters c = counters_;
    c.bytes_uploaded = bytes_uploaded_;
    c.pictures_reconstructed = folded_decoded_ + (parser_ ? parser_->pictures_decoded() : 0) - decoded_base_;
    c.pictures_skipped = folded_skipped_ + (parser_ ? parser_->pictures_skipped() : 0) - skipped_base_;
    return c;
  }
  void reset_counters() override {
    counters_ = BackendCounters();
    bytes_uploaded_ = 0;
    decoded_base_ = folded_decoded_ + (parser_ ? parser_->pictures_decoded() : 0);
    skipped_base_ = folded_skipped_ + (parser_ ? parser_->pictures_skipped() : 0);
  }

  // ----------------------------------------------------------------------------------------- PictureSink
  bool begin_picture(const hevc::PicSetup& s, hevc::PicBuffers* out, std::string* error) override {
    if (!ensure_geometry(s, error)) return false;
    Set& set = sets_[next_set_];
    if (set.busy) {
      const auto t0 = std::chrono::steady_clock::now();
      if (!engine_->wait_launched(&set.job)) {
        *error = engine_->error();
        return false;
      }
      launch_wait_ns_ += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
      if (hipEventQuery(set.done) == hipErrorNotReady) {  // (debug statistics: which pictures are waited for, how old they are)
        const int k = set.job.pp.intra_rows ? 1 : 0;
        ++dbg_stalls_[k];
        dbg_age_ns_[k] += std::chrono::duration_cast<std::chrono::nanoseconds>(t0 - set.submitted).count();
      }
      if (sleepy_event_wait(set.done) != hipSuccess) {
        *error = "waiting for a staging set's event failed";
        return false;
      }
      counters_.stall_ns += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
      set.busy = false;
    }
    int slot = -1;
    {
      std::lock_guard<std::mutex> lk(mu_);
      for (size_t i = 0; i < surf_.size() && slot < 0; ++i)
        if (!surf_[i].held) slot = static_cast<int>(i);
      if (slot < 0 && surf_.size() < static_cast<size_t>(gh::kMaxSurfaces)) {
        surf_.push_back(SurfaceSlot());
        slot = static_cast<int>(surf_.size()) - 1;
      }
      if (slot >= 0) surf_[slot].held = kParser;
    }
    if (slot < 0) {
      *error = "no free surface: decoded frames are not being fetched or discarded";
      return false;
    }
    if (!surf_[slot].dev && hipMalloc(reinterpret_cast<void**>(&surf_[slot].dev), surface_bytes_) != hipSuccess) {
      *error = "out of HBM for an HEVC surface";
      std::lock_guard<std::mutex> lk(mu_);
      surf_[slot].held = 0;
      return false;
    }
    uint8_t* h = set.host;
    out->slot = slot;
    out->cookie = &set;
    out->res = reinterpret_cast<gh::ResTu*>(h + off_.res);
    out->cap_res = cap_.res;
    out->coefs = reinterpret_cast<uint32_t*>(h + off_.coefs);
    out->cap_coefs = cap_.coefs;
    out->pus = reinterpret_cast<gh::InterPu*>(h + off_.pus);
    out->cap_pus = cap_.pus;
    out->weights = reinterpret_cast<gh::WeightEntry*>(h + off_.weights);
    out->cap_weights = cap_.weights;
    for (int c = 0; c < 3; ++c) {
      out->intra[c] = reinterpret_cast<gh::IntraTu*>(h + off_.intra[c]);
      out->cap_intra[c] = cap_.intra[c];
      out->intra_ctb_first[c] = reinterpret_cast<uint32_t*>(h + off_.first[c]);
    }
    out->edges = reinterpret_cast<gh::EdgeInfo*>(h + off_.edges);
    out->sao = reinterpret_cast<gh::SaoCtb*>(h + off_.sao);
    out->slices = reinterpret_cast<gh::SliceLoop*>(h + off_.slices);
    out->scaling = h + off_.scaling;
    return true;
  }

  bool end_picture(const hevc::PicDone& d, std::string* error) override {
    struct Timer {
      int64_t* acc;
      std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
      ~Timer() { *acc += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count(); }
    } timer{&submit_ns_};
    Set& set = *static_cast<Set*>(d.cookie);
    gh::PicParams& pp = set.job.pp;
    std::memset(&pp, 0, sizeof(pp));
    pp.width = cw_;
    pp.height = ch_;
    pp.y_pitch = y_pitch_;
    pp.c_pitch = y_pitch_ / 2;
    pp.log2_ctb = log2_ctb_;
    pp.width_ctb = wctb_;
    pp.height_ctb = hctb_;
    pp.cb_qp_offset = d.cb_qp_offset;
    pp.cr_qp_offset = d.cr_qp_offset;
    pp.pcm_bits_luma = d.pcm_bits_luma;
    pp.pcm_bits_chroma = d.pcm_bits_chroma;
    pp.num_res = static_cast<int32_t>(d.num_res);
    pp.num_pu = static_cast<int32_t>(d.num_pus);
    for (int c = 0; c < 3; ++c) pp.num_intra[c] = static_cast<int32_t>(d.num_intra[c]);
    pp.sao_enabled = d.sao;
    pp.deblock_enabled = d.deblock;
    pp.dst = surface_of(surf_[d.slot].dev);
    pp.recon = d.sao ? surface_of(work_) : pp.dst;
    for (int i = 0; i < d.num_ref_slots; ++i) pp.refs[d.ref_slots[i]] = surface_of(surf_[d.ref_slots[i]].dev);
    uint8_t* dv = set.dev;
    const size_t ctbs = static_cast<size_t>(wctb_) * hctb_;
    // mostly intra (an I picture, a scene cut): the row wavefront; scattered intra blocks: the dataflow kernel
    pp.intra_rows = d.num_intra[0] * 64 > static_cast<size_t>(cw_ / 4) * (ch_ / 4) * 3 ? 1 : 0;
    // Three copies: part A with the transform block list behind it (where the parser wrote them), the level stream
    // (where it is), and the small lists - prediction blocks in place, the ones after them moved down to follow.  (One
    // copy of everything packed downwards moved a megabyte a picture in host memory: 0.46 ms of the lane's CPU.)
    pp.res = reinterpret_cast<const gh::ResTu*>(dv + off_.res);
    pp.coefs = reinterpret_cast<const uint32_t*>(dv + off_.coefs);
    size_t packed = off_.pus;
    auto pack = [&](size_t from, size_t bytes) {
      const size_t at = packed;
      if (bytes && from != at) std::memmove(set.host + at, set.host + from, bytes);
      packed = (at + bytes + 15) & ~static_cast<size_t>(15);
      return at;
    };
    pp.pus = reinterpret_cast<const gh::InterPu*>(dv + pack(off_.pus, d.num_pus * sizeof(gh::InterPu)));
    pp.weights = reinterpret_cast<const gh::WeightEntry*>(dv + pack(off_.weights, d.num_weights * sizeof(gh::WeightEntry)));
    for (int c = 0; c < 3; ++c) pp.intra[c] = reinterpret_cast<const gh::IntraTu*>(dv + pack(off_.intra[c], d.num_intra[c] * sizeof(gh::IntraTu)));
    for (int c = 0; c < 3; ++c) pp.intra_ctb_first[c] = reinterpret_cast<const uint32_t*>(dv + pack(off_.first[c], d.num_intra[c] ? (ctbs + 1) * sizeof(uint32_t) : 0));
    pp.scaling = d.scaling ? dv + pack(off_.scaling, 4064) : nullptr;
    pp.edges = reinterpret_cast<const gh::EdgeInfo*>(dv + off_.edges);
    pp.sao = reinterpret_cast<const gh::SaoCtb*>(dv + off_.sao);
    pp.slices = reinterpret_cast<const gh::SliceLoop*>(dv + off_.slices);
    pp.resid[0] = resid_;
    pp.resid[1] = resid_ + static_cast<size_t>(y_pitch_) * ch_;
    pp.resid[2] = pp.resid[1] + static_cast<size_t>(y_pitch_ / 2) * (ch_ / 2);
    pp.intra_ticket = progress_;
    pp.progress = progress_ + 4;
    for (int c = 0; c < 3; ++c) pp.intra_done[c] = reinterpret_cast<uint8_t*>(progress_ + 4 + 3 * hctb_) + static_cast<size_t>(c) * (cw_ / 4) * (ch_ / 4);
    pp.fault = fault_dev_;
    pp.pu_map = pu_map_;
    // what the picture's intra kernel counts in, zeroed by the batch's first kernel: the rows' progress words or everything
    pp.clear_words = static_cast<int32_t>((pp.intra_rows ? 16 + sizeof(uint32_t) * 3 * static_cast<size_t>(hctb_) : progress_bytes_) / 4);
    std::memcpy(set.host + off_.params, &pp, sizeof(pp));
    gh::EngineJob& job = set.job;
    job.kind = gh::EngineJob::kPicture;
    job.pp_dev = reinterpret_cast<const gh::PicParams*>(dv + off_.params);
    job.copy[0] = {dv, set.host, off_.res + d.num_res * sizeof(gh::ResTu)};
    job.copy[1] = {dv + off_.coefs, set.host + off_.coefs, d.num_coefs * sizeof(uint32_t)};
    job.copy[2] = {dv + off_.pus, set.host + off_.pus, packed - off_.pus};
    bytes_uploaded_ += static_cast<int64_t>(job.copy[0].bytes + job.copy[1].bytes + job.copy[2].bytes);
    // the residual planes are zero between pictures (the kernels that read a residual clear it); only a picture that
    // did not run to its end leaves them in doubt
    job.clear_resid = resid_dirty_ ? resid_ : nullptr;
    job.clear_resid_bytes = resid_bytes_;
    resid_dirty_ = false;
    job.done = set.done;
    {
      const auto t0 = std::chrono::steady_clock::now();
      engine_->submit(client_, &job);
      engine_submit_ns_ += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
      set.submitted = t0;
    }
    (void)error;
    set.busy = true;
    next_set_ = (next_set_ + 1) % num_sets_;
    return true;
  }

  void abandon_picture(const hevc::PicDone& /*d*/) override {}

  void output_picture(int slot) override {
    std::lock_guard<std::mutex> lk(mu_);
    if (outq_.size() >= 1024) {
      sink_error_ = "too many undelivered pictures (decoded frames are not being fetched or discarded)";
      return;
    }
    if (slot >= 0) surf_[slot].held |= kOutput;
    outq_.push_back(slot);
    buffered_.fetch_add(1, std::memory_order_acq_rel);
  }

  void release_surface(int slot) override {
    std::lock_guard<std::mutex> lk(mu_);
    if (slot >= 0 && slot < static_cast<int>(surf_.size())) surf_[slot].held &= ~kParser;
  }

  // A reference picture that was never in the stream (decoding entered at a CRA picture; hevc_parser.h): a surface of
  // 128s.  Rare - once or twice per such entry - and outside the engine's order, so it waits: a free surface may still be
  // read by pictures of this decoder that are on the GPU (a surface is free from the moment the PARSER lets go of it;
  // a picture that takes it over queues up behind those, this fill does not).
  int missing_surface(std::string* error) override {
    for (int i = 0; i < num_sets_; ++i) {
      Set& set = sets_[i];
      if (!set.busy) continue;
      if (!engine_->wait_launched(&set.job)) {
        *error = engine_->error();
        return -1;
      }
      if (sleepy_event_wait(set.done) != hipSuccess) {
        *error = "waiting for a staging set's event failed";
        return -1;
      }
      set.busy = false;
    }
    int slot = -1;
    {
      std::lock_guard<std::mutex> lk(mu_);
      for (size_t i = 0; i < surf_.size() && slot < 0; ++i)
        if (!surf_[i].held) slot = static_cast<int>(i);
      if (slot < 0 && surf_.size() < static_cast<size_t>(gh::kMaxSurfaces)) {
        surf_.push_back(SurfaceSlot());
        slot = static_cast<int>(surf_.size()) - 1;
      }
      if (slot >= 0) surf_[slot].held = kParser;
    }
    if (slot < 0) {
      *error = "no free surface: decoded frames are not being fetched or discarded";
      return -1;
    }
    bool ok = surf_[slot].dev || hipMalloc(reinterpret_cast<void**>(&surf_[slot].dev), surface_bytes_) == hipSuccess;
    if (ok && !fill_stream_) ok = hipStreamCreateWithFlags(&fill_stream_, hipStreamNonBlocking) == hipSuccess;
    ok = ok && hipMemsetAsync(surf_[slot].dev, 128, surface_bytes_, fill_stream_) == hipSuccess && hipStreamSynchronize(fill_stream_) == hipSuccess;
    if (!ok) {
      *error = "could not make a grey surface for a reference picture that is not in the stream";
      std::lock_guard<std::mutex> lk(mu_);
      surf_[slot].held = 0;
      return -1;
    }
    return slot;
  }

 private:
  enum { kParser = 1, kOutput = 2 };
  struct SurfaceSlot {
    uint8_t* dev = nullptr;
    int held = 0;
  };
  struct Set {
    uint8_t* host = nullptr;
    uint8_t* dev = nullptr;
    hipEvent_t done = nullptr;
    bool busy = false;
    gh::EngineJob job;
    std::chrono::steady_clock::time_point submitted;
  };
  struct Offsets {
    size_t params, res, coefs, pus, weights, intra[3], first[3], edges, sao, slices, scaling, part_b, total;
  };
  struct Caps {
    size_t res, coefs, pus, weights, intra[3];
  };

  gh::Surface surface_of(uint8_t* base) const {
    gh::Surface s;
    s.y = base;
    s.u = base + static_cast<size_t>(y_pitch_) * ch_;
    s.v = s.u + static_cast<size_t>(y_pitch_ / 2) * (ch_ / 2);
    return s;
  }

  Result take_error() {
    std::lock_guard<std::mutex> lk(mu_);
    if (sink_error_.empty()) return Result();
    Result r(false, "AMDVideoDecoder (HEVC): " + sink_error_);
    sink_error_.clear();
    return r;
  }

  void free_all() {
    for (SurfaceSlot& s : surf_)
      if (s.dev) (void)hipFree(s.dev);
    surf_.clear();  // (keeps its capacity)
    for (Set& s : sets_) {
      if (s.host) (void)hipHostFree(s.host);
      if (s.dev) (void)hipFree(s.dev);
      if (s.done) (void)hipEventDestroy(s.done);
      s.host = s.dev = nullptr;
      s.done = nullptr;
      s.busy = false;
    }
    if (work_) (void)hipFree(work_);
    if (resid_) (void)hipFree(resid_);
    if (progress_) (void)hipFree(progress_);
    if (pu_map_) (void)hipFree(pu_map_);
    pu_map_ = nullptr;
    if (rgb_dev_) (void)hipFree(rgb_dev_);
    if (fault_host_) (void)hipHostFree(fault_host_);
    work_ = nullptr;
    resid_ = nullptr;
    progress_ = nullptr;
    rgb_dev_ = nullptr;
    fault_host_ = nullptr;
    fault_dev_ = nullptr;
    cw_ = ch_ = 0;
  }

  bool ensure_geometry(const hevc::PicSetup& s, std::string* error) {
    if (s.width == cw_ && s.height == ch_ && s.log2_ctb == log2_ctb_) {
      crop_l_ = s.crop_left;
      crop_t_ = s.crop_top;
      return true;
    }
    (void)quiesce();
    {
      std::lock_guard<std::mutex> lk(mu_);
      for (const SurfaceSlot& sl : surf_)
        if (sl.held) {
          *error = "the picture size changed while frames of the old size are still held";
          return false;
        }
    }
    free_all();
    cw_ = s.width;
    ch_ = s.height;
    log2_ctb_ = s.log2_ctb;
    crop_l_ = s.crop_left;
    crop_t_ = s.crop_top;
    wctb_ = (cw_ + (1 << log2_ctb_) - 1) >> log2_ctb_;
    hctb_ = (ch_ + (1 << log2_ctb_) - 1) >> log2_ctb_;
    y_pitch_ = (cw_ + 255) & ~255;
    surface_bytes_ = static_cast<size_t>(y_pitch_) * ch_ * 3 / 2;
    const size_t w4h4 = static_cast<size_t>(cw_ / 4) * (ch_ / 4), ctbs = static_cast<size_t>(wctb_) * hctb_;
    cap_.res = w4h4 * 3 / 2 + 64;
    cap_.coefs = static_cast<size_t>(cw_) * ch_ * 3 / 2 + 64;
    cap_.pus = w4h4 / 2 + 64;
    cap_.weights = 2 * 16 * static_cast<size_t>(gh::kMaxSlices);
    cap_.intra[0] = w4h4 + 64;
    cap_.intra[1] = cap_.intra[2] = w4h4 / 4 + 64;
    size_t o = 0;
    auto take = [&](size_t bytes) {
      const size_t at = o;
      o = (o + bytes + 255) & ~static_cast<size_t>(255);
      return at;
    };
    // part A, copied as one piece with every picture: parameters, slices, SAO parameters, the per-4x4 edge records
    off_.params = take(sizeof(gh::PicParams));
    off_.slices = take(gh::kMaxSlices * sizeof(gh::SliceLoop));
    off_.sao = take(ctbs * sizeof(gh::SaoCtb));
    off_.edges = take(w4h4 * sizeof(gh::EdgeInfo));
    off_.part_b = o;
    // part B: the lists, written by the parser at these worst-case offsets, packed downwards before their one copy
    off_.res = take(cap_.res * sizeof(gh::ResTu));
    off_.coefs = take(cap_.coefs * sizeof(uint32_t));
    off_.pus = take(cap_.pus * sizeof(gh::InterPu));
    off_.weights = take(cap_.weights * sizeof(gh::WeightEntry));
    for (int c = 0; c < 3; ++c) off_.intra[c] = take(cap_.intra[c] * sizeof(gh::IntraTu));
    for (int c = 0; c < 3; ++c) off_.first[c] = take((ctbs + 1) * sizeof(uint32_t));
    off_.scaling = take(4064);
    off_.total = o;
    bool ok = true;
    num_sets_ = off_.total <= kSmallSetBytes ? kSets : 3;
    for (int k = 0; k < num_sets_; ++k) {
      Set& set = sets_[k];
      ok = ok && hipHostMalloc(reinterpret_cast<void**>(&set.host), off_.total, hipHostMallocDefault) == hipSuccess;
      ok = ok && hipMalloc(reinterpret_cast<void**>(&set.dev), off_.total) == hipSuccess;
      ok = ok && hipEventCreateWithFlags(&set.done, hipEventDisableTiming | hipEventBlockingSync) == hipSuccess;
    }
    resid_bytes_ = surface_bytes_ * sizeof(int16_t);
    ok = ok && hipMalloc(reinterpret_cast<void**>(&work_), surface_bytes_) == hipSuccess;
    ok = ok && hipMalloc(reinterpret_cast<void**>(&resid_), resid_bytes_) == hipSuccess;
    progress_bytes_ = (16 + sizeof(uint32_t) * 3 * static_cast<size_t>(hctb_) + 3 * w4h4 + 3) & ~static_cast<size_t>(3);  // tickets, row progress words, a done map per plane
    ok = ok && hipMalloc(reinterpret_cast<void**>(&progress_), progress_bytes_) == hipSuccess;
    ok = ok && hipMalloc(reinterpret_cast<void**>(&pu_map_), sizeof(uint32_t) * w4h4) == hipSuccess;
    ok = ok && hipMalloc(reinterpret_cast<void**>(&rgb_dev_), static_cast<size_t>(cw_) * ch_ * 3) == hipSuccess;
    ok = ok && hipHostMalloc(reinterpret_cast<void**>(&fault_host_), sizeof(uint32_t), hipHostMallocMapped) == hipSuccess;
    if (ok) {
      *fault_host_ = 0;
      ok = hipHostGetDevicePointer(reinterpret_cast<void**>(&fault_dev_), fault_host_, 0) == hipSuccess;
    }
    if (!ok) {
      *error = "out of memory for the HEVC decoder's buffers (" + std::to_string(cw_) + "x" + std::to_string(ch_) + ")";
      free_all();
      return false;
    }
    next_set_ = 0;
    resid_dirty_ = true;
    return true;
  }

  Result take(uint8_t* dst, size_t dst_size, bool yuv) {
    HWANG_HIP_RETURN(hipSetDevice(device_));
    const size_t w = info_.width, h = info_.height, cw = (w + 1) / 2, chh = (h + 1) / 2;
    const size_t need = yuv ? w * h + 2 * cw * chh : w * h * 3;
    int slot;
    uint8_t* surface_dev = nullptr;
    {
      std::lock_guard<std::mutex> lk(mu_);
      if (outq_.empty()) return Result();
      if (dst_size < need) return Result(false, "AMDVideoDecoder::get_frame: buffer too small");
      slot = outq_.front();
      if (slot < 0)
        return Result(false, "AMDVideoDecoder::get_frame: this output position was excluded by the sparse hint (set_wanted) or could not be decoded");
      outq_.pop_front();
      buffered_.fetch_sub(1, std::memory_order_acq_rel);
      surface_dev = surf_[slot].dev;
    }
    const gh::Surface s = surface_of(surface_dev);
    const uint8_t* y = s.y + static_cast<size_t>(crop_t_) * y_pitch_ + crop_l_;
    const uint8_t* u = s.u + static_cast<size_t>(crop_t_ / 2) * (y_pitch_ / 2) + crop_l_ / 2;
    const uint8_t* v = s.v + static_cast<size_t>(crop_t_ / 2) * (y_pitch_ / 2) + crop_l_ / 2;
    // Where the frame goes: HBM (the conversion writes it), page-locked host memory (the D2H copy writes it), or pageable
    // host memory - a copy into that would block its caller until every picture in flight is done; it lands in a
    // page-locked bounce buffer instead and is copied out when a later call finds it there (wait_until_frames_copied at
    // the latest: the interface's own contract for when fetched frames are valid).
    hipPointerAttribute_t attr;
    const bool known = hipPointerGetAttributes(&attr, dst) == hipSuccess;
    if (!known) (void)hipGetLastError();
    const bool device_dst = known && attr.type == hipMemoryTypeDevice;
    const bool direct = device_dst || (known && attr.type == hipMemoryTypeHost);
    // The ring of fetched frames belongs to one thread at a time: feed() - another thread under DecoderAutomata - delivers
    // from it too, and an entry that is armed here (active, destination set) before its job has been handed to the engine
    // looked to that thread like the entry's PREVIOUS frame, landed: it copied the old bounce buffer into the new
    // destination and closed the entry (a fetched frame that was an earlier one of the same request, 1 - 3 % of the
    // requests through the reference's API: tools/experiments/src/hevc_serial_race.py).
    std::lock_guard<std::mutex> ol(out_mu_);
    Output& o = outputs_[output_next_];
    output_next_ = (output_next_ + 1) % kOutputs;
    HWANG_RETURN_ON_ERROR(deliver(o, true));
    if (!o.job.done && hipEventCreateWithFlags(&o.job.done, hipEventDisableTiming | hipEventBlockingSync) != hipSuccess)
      return Result(false, "AMDVideoDecoder::get_frame (HEVC): hipEventCreate failed");
    if (!direct && o.cap < need) {
      if (o.host) (void)hipHostFree(o.host);
      o.host = nullptr;
      o.cap = 0;
      if (hipHostMalloc(reinterpret_cast<void**>(&o.host), need, hipHostMallocDefault) != hipSuccess)
        return Result(false, "AMDVideoDecoder::get_frame (HEVC): no page-locked memory for the output bounce buffer");
      o.cap = need;
    }
    gh::EngineJob& job = o.job;
    job.kind = yuv ? gh::EngineJob::kYuv : gh::EngineJob::kRgb;
    job.y = y;
    job.u = u;
    job.v = v;
    job.y_pitch = y_pitch_;
    job.width = static_cast<int>(w);
    job.height = static_cast<int>(h);
    job.rgb_dev = rgb_dev_;
    job.out = direct ? dst : o.host;
    job.out_on_device = device_dst;
    o.dst = direct ? nullptr : dst;
    o.bytes = need;
    o.active = true;
    engine_->submit(client_, &job);
    {
      std::lock_guard<std::mutex> lk(mu_);
      surf_[slot].held &= ~kOutput;  // (whatever takes the surface next is a later job of this decoder: behind this one)
    }
    return Result();
  }

  struct Output {
    gh::EngineJob job;
    uint8_t* host = nullptr;  // bounce buffer
    size_t cap = 0;
    uint8_t* dst = nullptr;   // pageable caller memory the frame is owed to (null: the job writes the destination itself)
    size_t bytes = 0;
    bool active = false;
  };
  // Finishes a fetched frame: its job is through (and, from a bounce buffer, copied out).  `wait`: waits for that, else
  // leaves it if it is not there yet.
  Result deliver(Output& o, bool wait) {
    if (!o.active) return Result();
    if (!wait && !o.job.launched.load(std::memory_order_acquire)) return Result();
    if (!engine_->wait_launched(&o.job)) {
      o.active = false;
      return Result(false, "AMDVideoDecoder::get_frame (HEVC): " + engine_->error());
    }
    if (wait) {
      HWANG_HIP_RETURN(sleepy_event_wait(o.job.done));
    } else {
      const hipError_t q = hipEventQuery(o.job.done);
      if (q == hipErrorNotReady) return Result();
      HWANG_HIP_RETURN(q);
    }
    if (o.dst) {
      const auto t0 = std::chrono::steady_clock::now();
      std::memcpy(o.dst, o.host, o.bytes);
      copy_out_ns_ += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
    }
    o.dst = nullptr;
    o.active = false;
    return Result();
  }
  Result deliver_outputs(bool wait) {
    for (int k = 0; k < kOutputs; ++k) HWANG_RETURN_ON_ERROR(deliver(outputs_[(output_next_ + k) % kOutputs], wait));  // oldest first
    return Result();
  }
  // Everything this decoder has submitted is done: its pictures, its fetched frames (delivered).
  Result quiesce() {
    const bool ok = engine_->drain(client_);
    for (Set& set : sets_)
      if (set.busy) {
        (void)engine_->wait_launched(&set.job);
        if (set.done) (void)sleepy_event_wait(set.done);
        set.busy = false;
      }
    Result r;
    {
      std::lock_guard<std::mutex> ol(out_mu_);
      r = deliver_outputs(true);
    }
    engine_->settled(client_);
    if (!ok && r.ok) r = Result(false, "AMDVideoDecoder (HEVC): " + engine_->error());
    return r;
  }

  const int device_;
  hipStream_t fill_stream_ = nullptr;   // missing_surface()
  std::shared_ptr<gh::Engine> engine_;
  const int client_;
  std::mutex out_mu_;   // outputs_, output_next_
  Output outputs_[kOutputs];
  int output_next_ = 0;
  std::unique_ptr<hevc::Parser> parser_;
  VideoDecoderInterface::FrameInfo info_;
  std::vector<uint64_t> pending_wanted_, pending_keyframes_;
  std::mutex mu_;
  std::vector<SurfaceSlot> surf_;
  std::deque<int> outq_;
  std::atomic<int> buffered_{0};
  std::string sink_error_;
  Set sets_[kSets];
  int num_sets_ = kSets;  // (of sets_: the ones with buffers)
  int next_set_ = 0;
  Offsets off_ = {};
  Caps cap_ = {};
  int cw_ = 0, ch_ = 0, log2_ctb_ = 0, wctb_ = 0, hctb_ = 0, y_pitch_ = 0, crop_l_ = 0, crop_t_ = 0;
  size_t surface_bytes_ = 0, resid_bytes_ = 0, progress_bytes_ = 0;
  bool resid_dirty_ = true;  // (freshly allocated, or a picture faulted)
  uint8_t* work_ = nullptr;
  int16_t* resid_ = nullptr;
  uint32_t* progress_ = nullptr;
  uint32_t* pu_map_ = nullptr;  // per 4x4 luma block: its prediction block (k_hevc_inter -> k_hevc_deblock)
  uint8_t* rgb_dev_ = nullptr;
  uint32_t* fault_host_ = nullptr;
  uint32_t* fault_dev_ = nullptr;
  BackendCounters counters_;
  int64_t submit_ns_ = 0, copy_out_ns_ = 0, engine_submit_ns_ = 0, launch_wait_ns_ = 0;
  int64_t dbg_stalls_[2] = {0, 0}, dbg_age_ns_[2] = {0, 0};
  int64_t bytes_uploaded_ = 0, folded_decoded_ = 0, folded_skipped_ = 0, decoded_base_ = 0, skipped_base_ = 0;
};

}  // namespace

HevcDelegate* make_hevc_delegate(int device_id) { return new HEVCVideoDecoder(device_id); }

}  // namespace hwang

// C ABI test hook (include/hwang_amd.h: hwang_hevc_test_picture): one picture's worth of hand-made records through
// k_hevc.hip, so that the kernels can be compared with the numpy statements of tests/hevc_numpy.py block by block
// (tests/test_gpu_hevc_kernels.py) - without a stream, a parser or the oracle in between.
extern "C" int hwang_hevc_run_test_picture(const hwang_hevc_test_picture* t, char* error, size_t error_capacity) {
  using namespace hwang;
  namespace gh = amd::hevc;
  auto fail = [&](const char* what) {
    if (error && error_capacity) snprintf(error, error_capacity, "%s", what);
    return 1;
  };
  if (!t || t->width <= 0 || t->height <= 0 || (t->width & 7) || (t->height & 7) || t->num_refs < 0 || t->num_refs >= gh::kMaxSurfaces - 2) return fail("bad geometry");
  const int W = t->width, H = t->height, pitch = (W + 255) & ~255;
  const size_t surf_bytes = static_cast<size_t>(pitch) * H * 3 / 2;
  const int wctb = (W + (1 << t->log2_ctb) - 1) >> t->log2_ctb, hctb = (H + (1 << t->log2_ctb) - 1) >> t->log2_ctb;
  const size_t ctbs = static_cast<size_t>(wctb) * hctb, w4h4 = static_cast<size_t>(W / 4) * (H / 4);
  std::vector<void*> owned;
  auto dev = [&](size_t bytes, const void* src) -> uint8_t* {
    void* p = nullptr;
    if (hipMalloc(&p, bytes ? bytes : 16) != hipSuccess) return nullptr;
    owned.push_back(p);
    if (src && bytes && hipMemcpy(p, src, bytes, hipMemcpyHostToDevice) != hipSuccess) return nullptr;
    if (!src && hipMemset(p, 0, bytes ? bytes : 16) != hipSuccess) return nullptr;
    return static_cast<uint8_t*>(p);
  };
  auto cleanup = [&] {
    for (void* p : owned) (void)hipFree(p);
  };
  auto upload_surface = [&](const uint8_t* i420) -> uint8_t* {
    uint8_t* d = dev(surf_bytes, nullptr);
    if (!d || !i420) return d;
    bool ok = hipMemcpy2D(d, static_cast<size_t>(pitch), i420, static_cast<size_t>(W), static_cast<size_t>(W), static_cast<size_t>(H), hipMemcpyHostToDevice) == hipSuccess;
    const uint8_t* u = i420 + static_cast<size_t>(W) * H;
    uint8_t* du = d + static_cast<size_t>(pitch) * H;
    ok = ok && hipMemcpy2D(du, static_cast<size_t>(pitch / 2), u, static_cast<size_t>(W / 2), static_cast<size_t>(W / 2), static_cast<size_t>(H / 2), hipMemcpyHostToDevice) == hipSuccess;
    ok = ok && hipMemcpy2D(du + static_cast<size_t>(pitch / 2) * (H / 2), static_cast<size_t>(pitch / 2), u + static_cast<size_t>(W / 2) * (H / 2), static_cast<size_t>(W / 2),
                           static_cast<size_t>(W / 2), static_cast<size_t>(H / 2), hipMemcpyHostToDevice) == hipSuccess;
    return ok ? d : nullptr;
  };
  auto surface_of = [&](uint8_t* base) {
    gh::Surface s;
    s.y = base;
    s.u = base + static_cast<size_t>(pitch) * H;
    s.v = s.u + static_cast<size_t>(pitch / 2) * (H / 2);
    return s;
  };
  gh::PicParams pp;
  std::memset(&pp, 0, sizeof(pp));
  pp.width = W; pp.height = H; pp.y_pitch = pitch; pp.c_pitch = pitch / 2;
  pp.log2_ctb = t->log2_ctb; pp.width_ctb = wctb; pp.height_ctb = hctb;
  pp.cb_qp_offset = t->cb_qp_offset; pp.cr_qp_offset = t->cr_qp_offset;
  pp.pcm_bits_luma = pp.pcm_bits_chroma = 8;
  pp.num_res = t->num_res; pp.num_pu = t->num_pus;
  for (int c = 0; c < 3; ++c) pp.num_intra[c] = t->num_intra[c];
  pp.sao_enabled = t->sao_enabled; pp.deblock_enabled = t->deblock_enabled;
  uint8_t* recon = upload_surface(t->recon_in);
  uint8_t* dst = t->sao_enabled ? dev(surf_bytes, nullptr) : recon;
  bool ok = recon && dst;
  pp.recon = surface_of(recon);
  pp.dst = surface_of(dst);
  for (int r = 0; r < t->num_refs && ok; ++r) {
    uint8_t* s = upload_surface(t->refs + static_cast<size_t>(r) * W * H * 3 / 2);
    ok = s != nullptr;
    pp.refs[r] = surface_of(s);
  }
  pp.res = reinterpret_cast<const gh::ResTu*>(dev(sizeof(gh::ResTu) * static_cast<size_t>(t->num_res), t->res));
  pp.coefs = reinterpret_cast<const uint32_t*>(dev(sizeof(uint32_t) * static_cast<size_t>(t->num_coefs), t->coefs));
  pp.pus = reinterpret_cast<const gh::InterPu*>(dev(sizeof(gh::InterPu) * static_cast<size_t>(t->num_pus), t->pus));
  pp.weights = reinterpret_cast<const gh::WeightEntry*>(dev(sizeof(gh::WeightEntry) * static_cast<size_t>(t->num_weights), t->weights));
  for (int c = 0; c < 3; ++c) {
    pp.intra[c] = reinterpret_cast<const gh::IntraTu*>(dev(sizeof(gh::IntraTu) * static_cast<size_t>(t->num_intra[c]), t->intra[c]));
  }
  pp.edges = reinterpret_cast<const gh::EdgeInfo*>(dev(sizeof(gh::EdgeInfo) * w4h4, t->edges));
  pp.sao = reinterpret_cast<const gh::SaoCtb*>(dev(sizeof(gh::SaoCtb) * ctbs, t->sao));
  pp.slices = reinterpret_cast<const gh::SliceLoop*>(dev(sizeof(gh::SliceLoop) * static_cast<size_t>(std::max(t->num_slices, 1)), t->slices));
  pp.scaling = t->scaling ? dev(4064, t->scaling) : nullptr;
  int16_t* resid = reinterpret_cast<int16_t*>(dev(surf_bytes * 2, nullptr));
  pp.resid[0] = resid;
  pp.resid[1] = resid + static_cast<size_t>(pitch) * H;
  pp.resid[2] = pp.resid[1] + static_cast<size_t>(pitch / 2) * (H / 2);
  // (handed over full of ones: k_hevc_clear, the batch's first kernel, is what zeroes it)
  const size_t progress_bytes = (16 + sizeof(uint32_t) * 3 * static_cast<size_t>(hctb) + 3 * w4h4 + 3) & ~static_cast<size_t>(3);
  const std::vector<uint8_t> ones(progress_bytes, 0xff);
  pp.intra_ticket = reinterpret_cast<uint32_t*>(dev(progress_bytes, ones.data()));
  pp.clear_words = static_cast<int32_t>(progress_bytes / 4);
  pp.progress = pp.intra_ticket ? pp.intra_ticket + 4 : nullptr;
  for (int c = 0; c < 3 && pp.intra_ticket; ++c) pp.intra_done[c] = reinterpret_cast<uint8_t*>(pp.intra_ticket + 4 + 3 * hctb) + static_cast<size_t>(c) * w4h4;
  pp.intra_rows = t->intra_rows;
  {  // the intra kernels' table: the lists must be grouped by coding tree block, in raster order
    for (int c = 0; c < 3 && ok; ++c) {
      std::vector<uint32_t> first(ctbs + 1, 0);
      const gh::IntraTu* it = static_cast<const gh::IntraTu*>(t->intra[c]);
      long prev = -1;
      for (int k = 0; k < t->num_intra[c]; ++k) {
        const long ctb = static_cast<long>(((it[k].y << (c ? 1 : 0)) >> t->log2_ctb) * wctb + ((it[k].x << (c ? 1 : 0)) >> t->log2_ctb));
        if (ctb < prev) {
          cleanup();
          return fail("the intra records are not grouped by coding tree block");
        }
        prev = ctb;
        ++first[static_cast<size_t>(ctb) + 1];
      }
      for (size_t k = 1; k <= ctbs; ++k) first[k] += first[k - 1];
      pp.intra_ctb_first[c] = reinterpret_cast<const uint32_t*>(dev(sizeof(uint32_t) * (ctbs + 1), first.data()));
      ok = pp.intra_ctb_first[c] != nullptr;
    }
  }
  pp.fault = reinterpret_cast<uint32_t*>(dev(sizeof(uint32_t), nullptr));
  pp.pu_map = reinterpret_cast<uint32_t*>(dev(sizeof(uint32_t) * w4h4, nullptr));
  const gh::PicParams* pp_dev = reinterpret_cast<const gh::PicParams*>(dev(sizeof(pp), &pp));
  ok = ok && pp.res && pp.coefs && pp.pus && pp.weights && pp.edges && pp.sao && pp.slices && resid && pp.intra_ticket && pp.fault && pp.pu_map && pp_dev;
  if (ok) {
    gh::Batch batch;
    batch.pic[0] = pp_dev;
    const gh::PicParams* pps[1] = {&pp};
    ok = gh::launch_batch(pps, batch, 1, nullptr) == hipSuccess && hipDeviceSynchronize() == hipSuccess;
  }
  uint32_t fault = 0;
  if (ok) ok = hipMemcpy(&fault, pp.fault, sizeof(fault), hipMemcpyDeviceToHost) == hipSuccess;
  if (ok && t->out) {
    const gh::Surface s = pp.dst;
    ok = hipMemcpy2D(t->out, static_cast<size_t>(W), s.y, static_cast<size_t>(pitch), static_cast<size_t>(W), static_cast<size_t>(H), hipMemcpyDeviceToHost) == hipSuccess;
    uint8_t* ou = t->out + static_cast<size_t>(W) * H;
    ok = ok && hipMemcpy2D(ou, static_cast<size_t>(W / 2), s.u, static_cast<size_t>(pitch / 2), static_cast<size_t>(W / 2), static_cast<size_t>(H / 2), hipMemcpyDeviceToHost) == hipSuccess;
    ok = ok && hipMemcpy2D(ou + static_cast<size_t>(W / 2) * (H / 2), static_cast<size_t>(W / 2), s.v, static_cast<size_t>(pitch / 2), static_cast<size_t>(W / 2), static_cast<size_t>(H / 2),
                           hipMemcpyDeviceToHost) == hipSuccess;
  }
  if (ok && t->resid_out) {  // the three residual planes, w x h and (w/2) x (h/2) x 2, int16
    int16_t* o = t->resid_out;
    ok = hipMemcpy2D(o, static_cast<size_t>(W) * 2, pp.resid[0], static_cast<size_t>(pitch) * 2, static_cast<size_t>(W) * 2, static_cast<size_t>(H), hipMemcpyDeviceToHost) == hipSuccess;
    o += static_cast<size_t>(W) * H;
    for (int c = 1; c < 3 && ok; ++c, o += static_cast<size_t>(W / 2) * (H / 2))
      ok = hipMemcpy2D(o, static_cast<size_t>(W / 2) * 2, pp.resid[c], static_cast<size_t>(pitch / 2) * 2, static_cast<size_t>(W / 2) * 2, static_cast<size_t>(H / 2), hipMemcpyDeviceToHost) == hipSuccess;
  }
  cleanup();
  if (!ok) return fail("a HIP call failed (out of memory, or no GPU)");
  if (fault) return fail("the intra wavefront timed out");
  return 0;
}

namespace hwang {

}  // namespace hwang
