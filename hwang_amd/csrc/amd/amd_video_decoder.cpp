// MI355X backend behind VideoDecoderInterface.
//
// Shape of a device backend in the reference: impls/nvidia/
// nvidia_video_decoder.cpp:227-470 (configure / feed -> parser callbacks ->
// ring of decoded surfaces / get_frame -> colour kernel -> copy).  Here the
// "parser" is this repo's host-side H.264 front end (h264/h264_parser) and
// the "decoder" is the set of HIP kernels in k_recon.hip / k_rgb.hip.  An
// instance owns surfaces and staging buffers but no stream: it describes its
// work to the device's GpuExecutor, which batches the pictures of all
// instances into a few wide launches (gpu_executor.h):
//
//   feed(sample)  host: entropy decode + motion derivation -> pinned staging,
//                 submit_picture: k_upload -> k_inter -> k_residual/k_intra
//                 -> k_bs/k_deblock in the executor's next batch
//   get_frame()   submit_convert: convert_batch (surface -> caller memory if it
//                 is in HBM, else -> pinned bounce buffer -> caller memory)
//
// Nothing here waits for the GPU except when a staging slot or a surface has
// to be reused, and wait_until_frames_copied().
#include "amd_video_decoder.h"

#include "hevc_video_decoder.h"

#include "../h264/h264_parser.h"
#include "../hevc/hvcc.h"
#include "../sparse_hints.h"
#include "../util/cpus.h"
#include "../util/numa.h"
#include "../util/queue.h"
#include "../util/trace.h"
#include "frame_job.h"
#include "gpu_executor.h"
#include "hip_check.h"
#include "prof.h"

#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <deque>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace hwang {

namespace amd {
size_t deblock_scratch_bytes(int num_mbs);
size_t band_ctl_bytes(int mb_height);
}  // namespace amd

namespace {

using amd::MbInfo;
using amd::MbMotion;
using amd::PMotion;
using amd::SliceParams;

constexpr int kJobs = 4;        // staging ring depth (serial entropy decoding; with workers: pictures in flight + 3)
constexpr int kPipeLanes = 16;  // executor lanes a decoder with entropy workers rotates through, one per interval
constexpr size_t kPipeStagingBytes = size_t(2) << 30;  // ... within this much pinned host memory
constexpr int kPipeDepth = 160;  // ... and how many pictures may be queued between its workers and the GPU
constexpr int kMaxSlices = 256;
constexpr int kBounce = 4;      // pinned output bounce buffers (serial entropy decoding)
constexpr int kPipeBounce = 24; // ... with entropy workers: frames are fetched before their pictures reach the GPU
constexpr int kCopiers = 4;     // ... and this many threads copy them out of the bounce buffers (copier_loop)

// The entropy workers of every decoder that was created with num_devices = 1 ("the backend decides"): ONE pool for
// the process.  The reference's callers make one such decoder per thread or per pipeline instance
// (python/hwang/decoder.py:28; Scanner: one DecoderAutomata per pipeline instance); with a pool each, N of them would
// run 16 N threads on the same CPUs.  Sized by the first decoder that asks; never destroyed (its threads sleep on a
// condition variable and touch nothing but the jobs they are handed; parsers drain their own pictures before they go).
h264::WorkerPool* shared_entropy_pool(int threads) {
  static h264::WorkerPool* pool = new h264::WorkerPool(threads);
  return pool;
}

class AMDVideoDecoder : public VideoDecoderInterface,
                        public SparseDecodeHints,
                        public BatchingHints,
                        public PlanarOutput,
                        public CountedBackend,
                        public h264::PictureSink {
 public:
  // num_workers > 1: the reference's `num_devices` (the software backend's decoder thread count,
  // software_video_decoder.cpp:148) - that many entropy workers decode consecutive pictures' slice data at once.
  // pool_member: one of many decoders that a caller drives from threads of its own (ParallelRetriever's lanes):
  // no workers, and its pictures are meant for the executor's wide batches rather than for the shortest way through.
  AMDVideoDecoder(int device_id, int num_workers, bool pool_member, bool auto_workers)
      : device_(device_id), num_workers_(num_workers > 1 ? num_workers : 0), pool_member_(pool_member), auto_workers_(auto_workers) {}
  ~AMDVideoDecoder() override { teardown(); }

  bool open(std::string* error) {
    if (hipSetDevice(device_) != hipSuccess) {
      *error = "cannot select GPU " + std::to_string(device_);
      return false;
    }
    exec_ = amd::GpuExecutor::get(device_, error);
    if (!exec_) return false;
    // One executor lane orders a decoder's pictures.  With entropy workers the host runs far ahead of the GPU, whose
    // row kernels need ~2 ms per dependent picture: successive intervals (independent by construction) go to
    // different lanes so that their pictures share batches.
    const int nlanes = num_workers_ ? kPipeLanes : 1;
    // (with workers: the lanes are dealt to the executor's light engines in turn, so that the batches of GOPs in
    // flight overlap on the GPU instead of queueing behind each other)
    // (a decoder that is not one of a pool's lanes is served one interval at a time: its pictures travel as chains)
    for (int i = 0; i < nlanes; ++i) lanes_.push_back(exec_->register_lane(num_workers_ ? i : 0, !pool_member_));
    last_seq_.reset(new std::atomic<uint64_t>[nlanes]);
    for (int i = 0; i < nlanes; ++i) last_seq_[i].store(0);
    if (num_workers_) {
      // At most this many pictures are between the parser thread and the hand-over to the GPU; each holds a staging
      // set and a surface.  Pictures are handed over in decoding order, so everything behind a slow one (an IDR
      // picture takes five to ten times as long as a P picture) waits with the workers' results in hand: the window
      // has to span the GOPs that go by while one IDR picture is parsed, or the parser thread stalls on it.
      // (a decoder that chose its workers itself - num_devices = 1, the reference's own caller - serves one interval
      // at a time: a window of two pictures per worker is all it can fill, and it should not pin 1.6 GB for it)
      max_in_flight_ = auto_workers_ ? 2 * num_workers_ : std::min(std::max(4 * num_workers_, 64), 128);
      if (auto_workers_) {
        shared_pool_ = shared_entropy_pool(num_workers_);
      } else {
        pool_.reset(new h264::WorkerPool(num_workers_));
      }
    }
    // A staging set is tied up from begin_picture until the GPU is through with the picture.  The GPU needs ~2 ms per
    // dependent picture and interval, the workers deliver a picture every ~0.15 ms: only with the pictures of many
    // intervals queued do the lanes' batches fill (HWANG_PIPE_DEPTH overrides the queue depth).
    int depth = auto_workers_ ? 32 : kPipeDepth;
    if (const char* e = getenv("HWANG_PIPE_DEPTH")) depth = std::max(4, std::min(atoi(e), 1024));
    pipe_depth_ = depth;
    // (with workers: sized with the picture geometry, ensure_geometry; a serial decoder outside a pool may have a
    // chain's worth of pictures on the GPU at a time)
    jobs_.resize(pool_member_ ? kJobs : 3 * kJobs);
    bounce_.resize(num_workers_ ? kPipeBounce : kBounce);
    if (num_workers_)
      for (int i = 0; i < kCopiers; ++i) copiers_.emplace_back([this] { copier_loop(); });
    return true;
  }

  // ------------------------------------------------------ VideoDecoderInterface
  Result configure(const FrameInfo& metadata, const std::vector<uint8_t>& extradata) override {
    trace::event("configure");
    HWANG_HIP_RETURN(hipSetDevice(device_));
    // The format switch of the reference's software backend (software_video_decoder.cpp:124-135): h264 / avc1, the
    // HEVC names, anything else.  An HEVC track goes to the HEVC half of this backend (hevc_video_decoder.cpp: host
    // parser + k_hevc.hip); every later call is forwarded to it until a configure() names H.264 again.
    if (hevc::is_hevc_format(metadata.format)) {
      if (!hevc_) hevc_.reset(make_hevc_delegate(device_));
      hevc_active_ = true;
      if (!pending_wanted_.empty()) hevc_->set_wanted_frames(pending_wanted_);
      if (!pending_keyframes_.empty()) hevc_->set_interval_keyframes(pending_keyframes_);
      return hevc_->configure(metadata, extradata);
    }
    hevc_active_ = false;
    if (metadata.format != "h264" && metadata.format != "avc1") {
      return Result(false, "Unsupported video codec: " + metadata.format + ". Supported codecs are: h264 (avc1) and hevc (h265, hev1, hvc1)");
    }
    if (parser_) {
      folded_decoded_ += parser_->pictures_decoded();
      folded_skipped_ += parser_->pictures_skipped();
      parser_.reset();  // (with entropy workers: waits until every picture in flight has been handed over)
    }
    HWANG_RETURN_ON_ERROR(drain_and_reset());
    info_ = metadata;
    parser_.reset(new h264::Parser(this, shared_pool_ ? shared_pool_ : pool_.get(), max_in_flight_));
    if (!parser_->configure(extradata.data(), extradata.size()))
      return Result(false, "AMDVideoDecoder::configure: " + parser_->error());
    if (parser_->width() != static_cast<int>(metadata.width) ||
        parser_->height() != static_cast<int>(metadata.height)) {
      return Result(false, "AMDVideoDecoder::configure: stream is " + std::to_string(parser_->width()) + "x" +
                               std::to_string(parser_->height()) + " but the container says " +
                               std::to_string(metadata.width) + "x" + std::to_string(metadata.height));
    }
    if (!pending_wanted_.empty()) {
      parser_->set_wanted(pending_wanted_);
      if (!pending_keyframes_.empty()) parser_->set_keyframes(pending_keyframes_);
    }
    return Result();
  }

  Result feed(const uint8_t* encoded_buffer, size_t encoded_size, bool keyframe) override {
    if (hevc_active_) return hevc_->feed(encoded_buffer, encoded_size, keyframe);
    if (encoded_size == 0) return Result();  // end-of-interval marker; flush() follows
    if (!parser_) return Result(false, "AMDVideoDecoder::feed before configure");
    HWANG_HIP_RETURN(hipSetDevice(device_));
    // A sync sample starts a closed GOP: nothing behind it refers to anything before it, so its pictures may go to
    // another executor lane and run beside the previous GOP's (an interval is often several byte-adjacent GOPs: on
    // one lane that is a chain of 40-60 dependent pictures at ~2 ms each).
    if (keyframe || rotate_always_) cur_lane_ = (cur_lane_ + 1) % static_cast<int>(lanes_.size());
    if (keyframe) trace::event("feed sync sample", static_cast<long>(encoded_size));
    const auto t0 = std::chrono::steady_clock::now();
    const int64_t stall0 = counters_.stall_ns;
    const bool ok = parser_->decode(encoded_buffer, encoded_size);
    counters_.entropy_ns +=
        std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count() -
        (counters_.stall_ns - stall0);
    ++counters_.samples_fed;
    if (!ok) return Result(false, "AMDVideoDecoder::feed: " + parser_->error());
    return take_sink_error("AMDVideoDecoder::feed: ");
  }

  Result flush() override {
    if (hevc_active_) {
      pending_wanted_.clear();
      pending_keyframes_.clear();
      return hevc_->flush();
    }
    if (!parser_) return Result();
    HWANG_HIP_RETURN(hipSetDevice(device_));
    if (!parser_->flush()) return Result(false, "AMDVideoDecoder::flush: " + parser_->error());
    HWANG_RETURN_ON_ERROR(take_sink_error("AMDVideoDecoder::flush: "));
    pending_wanted_.clear();
    pending_keyframes_.clear();
    return Result();
  }

  Result discard_frame() override {
    if (hevc_active_) return hevc_->discard_frame();
    int slot;
    if (!outq_.try_pop(slot)) return Result();
    buffered_.fetch_sub(1, std::memory_order_acq_rel);
    if (slot >= 0) note_undelivered(-1);
    std::lock_guard<std::mutex> lk(out_mu_);
    if (slot >= 0) unref_surface_locked(slot, kHeldByOutput);
    return Result();
  }

  Result get_frame(uint8_t* decoded_buffer, size_t decoded_size) override {
    if (hevc_active_) return hevc_->get_frame(decoded_buffer, decoded_size);
    return take_frame(decoded_buffer, decoded_size, false);
  }
  Result get_frame_yuv(uint8_t* decoded_buffer, size_t decoded_size) override {
    if (hevc_active_) return hevc_->get_frame_yuv(decoded_buffer, decoded_size);
    return take_frame(decoded_buffer, decoded_size, true);
  }

  int decoded_frames_buffered() override {
    if (hevc_active_) return hevc_->decoded_frames_buffered();
    return buffered_.load(std::memory_order_acquire);
  }

  Result wait_until_frames_copied() override {
    if (hevc_active_) return hevc_->wait_until_frames_copied();
    trace::event("wait_until_frames_copied");
    struct Done {
      ~Done() { trace::event("frames copied"); }
    } done;
    HWANG_HIP_RETURN(hipSetDevice(device_));
    // entropy workers: frames may have been fetched before their pictures were decoded; those pictures first
    std::string worker_error;
    const bool workers_ok = !parser_ || parser_->wait_idle(&worker_error);
    // Whatever wait_all() reports, no copier may still be writing into the caller's buffer when this returns: the
    // caller frees it on an error (DecoderAutomata::get_frames -> the pybind wrapper).  A copier whose conversion had
    // succeeded can be in the middle of its memcpy when a later batch trips the executor's sticky error.
    const Result waited = wait_all();
    {
      std::unique_lock<std::mutex> lk(out_mu_);
      drain_bounces(lk);
      if (!waited.ok) {
        copy_error_.clear();
        return waited;
      }
      if (!copy_error_.empty()) {
        Result r(false, "AMDVideoDecoder: " + copy_error_);
        copy_error_.clear();
        return r;
      }
    }
    if (!workers_ok) return Result(false, "AMDVideoDecoder: " + worker_error);
    return take_sink_error("AMDVideoDecoder: ");
  }

  // ------------------------------------------------------------ extensions
  void set_long_queue(bool long_queue) override {
    if (exec_)
      for (int lane : lanes_) exec_->set_lane_patient(lane, long_queue);
  }
  bool feed_would_wait() override { return hevc_active_ && hevc_->feed_would_wait(); }
  void set_wanted_frames(const std::vector<uint64_t>& ordinals) override {
    pending_wanted_ = ordinals;
    if (hevc_active_) {
      hevc_->set_wanted_frames(ordinals);
      return;
    }
    if (parser_) parser_->set_wanted(ordinals);
  }
  void set_interval_keyframes(const std::vector<uint64_t>& ordinals) override {
    pending_keyframes_ = ordinals;
    if (hevc_active_) {
      hevc_->set_interval_keyframes(ordinals);
      return;
    }
    if (parser_) parser_->set_keyframes(ordinals);
  }
  BackendCounters counters() override {
    if (hevc_active_) return hevc_->counters();
    BackendCounters c = counters_;
    c.bytes_uploaded = bytes_uploaded_.load();
    c.pictures_reconstructed = folded_decoded_ + (parser_ ? parser_->pictures_decoded() : 0) - decoded_base_;
    c.pictures_skipped = folded_skipped_ + (parser_ ? parser_->pictures_skipped() : 0) - skipped_base_;
    return c;
  }
  void reset_counters() override {
    if (hevc_) hevc_->reset_counters();
    counters_ = BackendCounters();
    bytes_uploaded_.store(0);
    decoded_base_ = folded_decoded_ + (parser_ ? parser_->pictures_decoded() : 0);
    skipped_base_ = folded_skipped_ + (parser_ ? parser_->pictures_skipped() : 0);
  }

  // ------------------------------------------------------------ PictureSink
  bool begin_picture(const h264::PicSetup& s, h264::PicBuffers* out, std::string* error) override {
    if (!ensure_geometry(s, error)) return false;
    // A free surface.  Its pinned motion buffer is the parser's output for
    // this picture, so it must not still be the source of an earlier picture's
    // upload, and - when intervals rotate through several lanes - nothing that
    // another lane submitted earlier may still read or write it: prefer a free
    // surface that is through with both, then a new one (up to a soft limit),
    // and only then wait for the oldest.  (Taking the lowest free index re-used
    // the surface released most recently and made every picture wait for the
    // one submitted just before it.)
    int slot = -1;
    uint64_t done[kPipeLanes];  // how far each of this decoder's lanes has got (a moment ago: good enough to tell "free")
    exec_->completed(lanes_.data(), static_cast<int>(lanes_.size()), done);
    {
      auto settled = [&](const Stamp& st) { return !st.seq || done[st.lane] >= st.seq; };
      std::lock_guard<std::mutex> lk(out_mu_);
      int fresh = -1, oldest = -1, allocated = 0;
      for (int i = 0; i < amd::kMaxSurfaces; ++i) {
        if (surf_[i].dev) ++allocated;
        if (surf_[i].held) continue;
        if (!surf_[i].dev) {
          // (a decoder with workers takes new surfaces from the top: its tests then cover surface numbers >= 128)
          if (fresh < 0 || num_workers_) fresh = i;
        } else if (settled(surf_[i].motion[next_set(surf_[i])]) && busy_elsewhere(surf_[i], cur_lane_, done) < 0) {
          slot = i;
          break;
        } else if (oldest < 0 || surf_[i].order < surf_[oldest].order) {
          oldest = i;
        }
      }
      if (slot < 0) {
        slot = (fresh >= 0 && (allocated < soft_surfaces() || oldest < 0)) ? fresh : oldest;
        if (slot >= 0 && slot == oldest) {  // (debug statistics: what the surfaces were doing when none was ready)
          ++dbg_none_ready_;
          for (size_t i = 0; i < lanes_.size(); ++i) {
            const uint64_t pend = last_seq_[i].load() - done[i];
            dbg_pending_total_ += pend;
            if (pend) ++dbg_pending_lanes_;
          }
          for (int i = 0; i < amd::kMaxSurfaces; ++i) {
            if (surf_[i].dev && !surf_[i].held) {
              if (!settled(surf_[i].motion[next_set(surf_[i])])) ++dbg_unsettled_motion_;
              if (busy_elsewhere(surf_[i], -1, done) >= 0) ++dbg_unsettled_busy_;
            }
            if (!surf_[i].dev) continue;
            if (surf_[i].held & kHeldByParser) ++dbg_held_parser_;
            else if (surf_[i].held & kHeldByOutput) ++dbg_held_output_;
            else ++dbg_free_busy_;
          }
        }
      }
      if (slot >= 0) surf_[slot].held = kHeldByParser;
    }
    if (slot < 0) {
      *error = "AMDVideoDecoder: out of decode surfaces (output frames are not being consumed)";
      out->no_surface = true;
      return false;
    }
    {
      const bool fresh_surface = !surf_[slot].dev;
      const auto a0 = std::chrono::steady_clock::now();
      if (!alloc_surface(slot, error)) return false;
      if (fresh_surface) {
        dbg_alloc_ns_ += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - a0).count();
        ++dbg_allocs_;
      }
    }
    struct StallTimer {
      BackendCounters& c;
      std::chrono::steady_clock::time_point t0;
      ~StallTimer() {
        c.stall_ns += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
      }
    };
    Stamp motion;
    uint64_t busy[kPipeLanes];
    {
      std::lock_guard<std::mutex> lk(out_mu_);
      surf_[slot].flip = next_set(surf_[slot]);
      motion = surf_[slot].motion[surf_[slot].flip];
      std::memcpy(busy, surf_[slot].busy_seq, sizeof(busy));
    }
    // the motion buffer of this surface may still be the source of an upload
    if (motion.seq) {
      StallTimer st{counters_, std::chrono::steady_clock::now()};
      const auto w0 = std::chrono::steady_clock::now();
      if (!wait_stamp(motion, error)) return false;
      if (std::chrono::steady_clock::now() - w0 > std::chrono::microseconds(50)) ++dbg_motion_waits_;
    }
    // work of other lanes that still touches the surface (this lane's own is ordered by the lane): every one of them -
    // a reader on one lane and a conversion on another may both be pending
    for (int li = 0; li < static_cast<int>(lanes_.size()); ++li) {
      if (li == cur_lane_ || !busy[li] || done[li] >= busy[li]) continue;
      StallTimer st{counters_, std::chrono::steady_clock::now()};
      const auto w0 = std::chrono::steady_clock::now();
      if (!wait_stamp(Stamp{li, busy[li]}, error)) return false;
      if (std::chrono::steady_clock::now() - w0 > std::chrono::microseconds(50)) ++dbg_busy_waits_;
    }
    {
      std::lock_guard<std::mutex> lk(out_mu_);
      surf_[slot].motion[surf_[slot].flip] = Stamp();
      surf_[slot].submitted = surf_[slot].abandoned = false;
    }
    // A staging set: any one the GPU is through with.  (Taking them round robin made the parser wait for the picture
    // submitted jobs_.size() pictures ago in particular - with intervals spread over lanes that is a picture in the
    // middle of some lane's chain, and the parser then advanced one picture per batch, in step with that one lane.)
    int ji = -1;
    {
      int oldest = -1;
      const int nj = jobs_ready_;
      for (int k = 0; k < nj && ji < 0; ++k) {
        const int i = (job_cursor_ + k) % nj;
        const Job& cand = jobs_[i];
        if (cand.in_use.load(std::memory_order_acquire)) continue;
        if (!cand.stamp.seq || done[cand.stamp.lane] >= cand.stamp.seq) ji = i;
        else if (oldest < 0 || cand.order < jobs_[oldest].order) oldest = i;
      }
      // (a decoder that sized its own window - num_devices = 1 - makes its staging sets as the intervals turn out to
      // need them: one more rather than a wait)
      if (ji < 0 && jobs_ready_ < static_cast<int>(jobs_.size()) && !jobs_alloc_failed_) {
        std::string alloc_error;
        if (alloc_job(jobs_[jobs_ready_], &alloc_error)) ji = jobs_ready_++;
        else {
          // memory is short: no further attempts (each costs a ms-scale hipHostMalloc on the feed thread), the decoder
          // waits for its oldest set from here on
          jobs_alloc_failed_ = true;
          if (oldest < 0) {
            *error = alloc_error;
            return false;
          }
        }
      }
      if (ji < 0) ji = oldest;
    }
    if (ji < 0) {
      *error = "AMDVideoDecoder: every staging set is taken by a picture that is still being parsed";
      return false;
    }
    Job& j = jobs_[ji];
    if (j.stamp.seq) {
      StallTimer st{counters_, std::chrono::steady_clock::now()};
      const auto w0 = std::chrono::steady_clock::now();
      if (!wait_stamp(j.stamp, error)) return false;
      if (std::chrono::steady_clock::now() - w0 > std::chrono::microseconds(50)) ++dbg_job_waits_;
      j.stamp = Stamp();
    }
    j.in_use.store(true, std::memory_order_release);
    job_cursor_ = (ji + 1) % std::max(1, jobs_ready_);
    j.lane = cur_lane_;
    out->cookie = ji;
    out->mb = j.mb_host;
    const int set = surf_[slot].flip;
    j.motion_set = set;
    out->motion = surf_[slot].motion_host[set];
    out->pmo = surf_[slot].pmo_host[set];
    out->pmx = surf_[slot].pmx_host[set];
    out->pmx_capacity = pmx_words_;
    out->coefs = j.coef_host;
    out->coef_capacity = coef_capacity_;
    out->slices = j.slice_host;
    out->max_slices = kMaxSlices;
    out->slot = slot;
    // k_mv derives the P macroblocks' motion vectors (its LDS line buffer covers kLineMbs macroblocks per row)
    out->device_mv = mbw_ <= amd::kLineMbs && !host_mv_;
    return true;
  }

  // (with entropy workers: called by a worker thread, one call at a time, in decoding order)
  bool end_picture(const h264::PicDone& d, std::string* /*error*/) override {
    Job& j = jobs_[d.cookie];
    const int lane_index = j.lane;
    const int lane = lanes_[lane_index];
    const size_t nmb = static_cast<size_t>(mbw_) * mbh_;
    amd::PicSubmit sub;
    std::memset(&sub.desc, 0, sizeof(sub.desc));
    auto add_copy = [&](void* dst, const void* src, size_t bytes) {
      static_assert(amd::kJobUploads >= 6, "MbInfo, coefficients, SliceParams, MbMotion, PMotion, extras");
      sub.copies[sub.ncopies].dst = dst;
      sub.copies[sub.ncopies].src = src;
      sub.copies[sub.ncopies].bytes = bytes;
      ++sub.ncopies;
      bytes_uploaded_.fetch_add(static_cast<int64_t>(bytes), std::memory_order_relaxed);
    };
    add_copy(j.mb_dev, j.mb_host, nmb * sizeof(MbInfo));
    if (d.coef_count) add_copy(j.coef_dev, j.coef_host, d.coef_count * sizeof(uint32_t));
    add_copy(j.slice_dev, j.slice_host, d.num_slices * sizeof(SliceParams));
    // motion: full 144-byte records only for macroblocks the host derived itself (B slices, pictures too wide
    // for k_mv); everything k_mv derives travels as 12-byte records + the extra differences
    const int set = j.motion_set;
    if (d.has_inter && d.full_motion) add_copy(j.motion_dev, surf_[d.slot].motion_host[set], nmb * sizeof(MbMotion));
    if (d.compact_motion && d.has_inter) {  // (an intra-only picture's records are read by nobody)
      add_copy(j.pmo_dev, surf_[d.slot].pmo_host[set], nmb * sizeof(PMotion));
      add_copy(j.pmx_dev, surf_[d.slot].pmx_host[set], d.pmx_count * sizeof(uint32_t));
    }
    amd::PictureParams& pp = sub.desc.pp;
    pp.mb_width = mbw_;
    pp.mb_height = mbh_;
    pp.y_pitch = y_pitch_;
    pp.c_pitch = c_pitch_;
    pp.cur_slot = d.slot;
    pp.num_slices = d.num_slices;
    std::memcpy(pp.scaling4, d.scaling4, sizeof(pp.scaling4));
    std::memcpy(pp.scaling8, d.scaling8, sizeof(pp.scaling8));
    {
      std::lock_guard<std::mutex> lk(out_mu_);  // (begin_picture may be allocating a surface on the parser thread)
      for (int i = 0; i < amd::kMaxSurfaces; ++i) {
        pp.surfaces[i].y = surf_[i].dev;
        pp.surfaces[i].u = surf_[i].dev ? surf_[i].dev + u_offset_ : nullptr;
        pp.surfaces[i].v = surf_[i].dev ? surf_[i].dev + v_offset_ : nullptr;
      }
    }
    sub.desc.mbs = j.mb_dev;
    sub.desc.motions = j.motion_dev;
    sub.desc.coefs = j.coef_dev;
    sub.desc.slices = j.slice_dev;
    sub.desc.db = j.db_dev;
    sub.desc.resid = j.resid_dev;
    sub.desc.band_ctl = j.band_ctl_dev;
    sub.desc.pmo = j.pmo_dev;
    sub.desc.pmx = j.pmx_dev;
    if (++epoch_ == 0) ++epoch_;
    sub.desc.epoch = epoch_;
    sub.desc.do_inter = d.has_inter;
    sub.desc.do_intra = d.has_intra;
    sub.desc.do_deblock = d.any_deblock;
    sub.desc.do_mv = d.mv_deferred;
    // tests/test_gpu_decode.py, test_gpu_chain.py: the fault paths (1: k_deblock withholds a band hand-off word; 2: a
    // chain launch's inter waves withhold one run's bit)
    static const int inject = getenv("HWANG_INJECT_FAULT") ? std::max(1, atoi(getenv("HWANG_INJECT_FAULT"))) : 0;
    sub.desc.inject_fault = static_cast<uint32_t>(inject);
    sub.num_mbs = static_cast<int>(nmb);
    sub.heavy = (d.has_intra && !d.has_inter) ? 1 : 0;
    sub.num_intra_mbs = d.num_intra_mbs;
    sub.num_ref_slots = 0;
    for (int i = 0; i < d.num_ref_slots && i < 32; ++i) sub.ref_slots[sub.num_ref_slots++] = static_cast<int16_t>(d.ref_slots[i]);
    const uint64_t S = nmb * 384;  // coded I420 bytes
    sub.bytes_inter = d.has_inter ? S * (1 + ((d.slice_type_mask & 2) ? 2 : 1)) : 0;
    sub.bytes_intra = d.has_intra ? S : 0;
    sub.bytes_deblock = d.any_deblock ? 2 * S : 0;
    sub.bytes_mv = d.mv_deferred ? nmb * (sizeof(MbInfo) + 80 + 64) : 0;  // header + differences/refs in, vectors out
    if (lanes_.size() > 1) {
      // The lanes rotate at sync samples and the executor orders work only within a lane.  A sync sample that is not
      // an IDR picture (open GOP) may be followed by pictures that refer to the GOP before it - on another lane:
      // whatever that lane still has to do with such a reference is waited for here.  (Closed GOPs: never.)
      uint64_t foreign[kPipeLanes] = {0};  // per lane: the latest submission that touches one of the references
      bool any_foreign = false;
      {
        std::lock_guard<std::mutex> lk(out_mu_);
        for (int i = 0; i < d.num_ref_slots; ++i) {
          const SurfaceSlot& r = surf_[d.ref_slots[i]];
          if (!r.dev || r.lane == lane_index) continue;
          for (int li = 0; li < static_cast<int>(lanes_.size()); ++li)
            if (li != lane_index && r.busy_seq[li] > foreign[li]) {
              foreign[li] = r.busy_seq[li];
              any_foreign = true;
            }
        }
      }
      std::string ignored;
      if (any_foreign)
        for (int li = 0; li < static_cast<int>(lanes_.size()); ++li)
          if (foreign[li]) (void)wait_stamp(Stamp{li, foreign[li]}, &ignored);  // (a failed lane fails every later wait too)
    }
    trace::event(sub.heavy ? "end_picture (intra-only)" : "end_picture", d.slot, static_cast<long>(d.coef_count));
    j.stamp = Stamp{lane_index, exec_->submit_picture(lane, sub)};
    note_seq(j.stamp);

    {
      std::lock_guard<std::mutex> lk(out_mu_);
      SurfaceSlot& cs = surf_[d.slot];
      if (d.has_inter || d.compact_motion) cs.motion[set] = j.stamp;  // the surface's pinned buffers are upload sources
      mark_busy(cs, j.stamp);   // written by this submission ...
      cs.lane = lane_index;
      cs.order = ++use_order_;
      j.order = use_order_;
      for (int i = 0; i < d.num_ref_slots; ++i) {  // ... which reads these
        mark_busy(surf_[d.ref_slots[i]], j.stamp);
        surf_[d.ref_slots[i]].order = use_order_;
      }
    }
    // conversions that were asked for before the picture got here (entropy workers: output_picture precedes
    // end_picture, and the caller fetches frames as soon as they are announced) go right behind it
    for (;;) {
      Deferred def;
      {
        std::lock_guard<std::mutex> lk(out_mu_);
        SurfaceSlot& cs = surf_[d.slot];
        if (cs.deferred.empty()) {
          cs.submitted = true;
          break;
        }
        def = cs.deferred.front();
        cs.deferred.pop_front();
      }
      submit_conversion(d.slot, lane_index, def.sub, def.bounce);
    }
    out_cv_.notify_all();
    j.in_use.store(false, std::memory_order_release);
    return true;
  }

  // (entropy workers) the picture will never arrive: what was asked of its frame cannot be served
  void abandon_picture(const h264::PicDone& d) override {
    if (d.cookie >= 0 && d.cookie < static_cast<int>(jobs_.size())) jobs_[d.cookie].in_use.store(false, std::memory_order_release);
    {
      std::lock_guard<std::mutex> lk(out_mu_);
      SurfaceSlot& cs = surf_[d.slot];
      cs.abandoned = true;
      for (Deferred& def : cs.deferred) {
        if (def.bounce >= 0) bounce_[def.bounce].dst = nullptr;
        sink_error_ = "a frame that was fetched could not be decoded (its picture or an earlier one of the interval failed)";
        unref_surface_locked(d.slot, kHeldByOutput);
      }
      cs.deferred.clear();
    }
    out_cv_.notify_all();
  }

  // Submits one conversion of `slot` on the lane its picture went to, and lets go of the frame.
  void submit_conversion(int slot, int lane_index, const amd::ConvSubmit& sub, int bounce) {
    const Stamp st{lane_index, exec_->submit_convert(lanes_[lane_index], sub)};
    note_seq(st);
    std::lock_guard<std::mutex> lk(out_mu_);
    mark_busy(surf_[slot], st);
    if (bounce >= 0) {
      bounce_[bounce].stamp = st;
      bounce_[bounce].queued = true;
      out_cv_.notify_all();  // (a copier may take it from here)
    }
    unref_surface_locked(slot, kHeldByOutput);
  }

  void output_picture(int slot) override {
    {
      std::lock_guard<std::mutex> lk(out_mu_);
      if (slot >= 0) surf_[slot].held |= kHeldByOutput;
    }
    // display order; the consumer side is get_frame / discard_frame.  Never blocks: the thread that feeds may also be
    // the only consumer (ParallelRetriever::run_item), so a full queue is an error of the caller's protocol, not a
    // wait.  Counted after it is in the queue: a consumer acts on the count (the automata polls it), so the count must
    // never run ahead of what get_frame can find (a consumer that fetches between the two steps sees the count dip to
    // -1 for a moment, which nobody acts on).
    if (!outq_.try_push(slot)) {
      std::lock_guard<std::mutex> lk(out_mu_);
      if (slot >= 0) unref_surface_locked(slot, kHeldByOutput);
      sink_error_ = "too many undelivered pictures (decoded frames are not being fetched or discarded)";
      return;
    }
    buffered_.fetch_add(1, std::memory_order_acq_rel);
    if (slot >= 0) note_undelivered(1);
  }

  void release_surface(int slot) override {
    std::lock_guard<std::mutex> lk(out_mu_);
    unref_surface_locked(slot, kHeldByParser);
  }

 private:
  std::unique_ptr<HevcDelegate> hevc_;  // the HEVC half (hevc_video_decoder.cpp), made at the first HEVC configure()
  bool hevc_active_ = false;            // ... and in charge since the last configure()
  enum { kHeldByParser = 1, kHeldByOutput = 2 };
  // beyond this many surfaces, wait for an upload rather than grow (with entropy workers: the pictures in flight and
  // the intervals still on the GPU hold surfaces too)
  int soft_surfaces() const { return num_workers_ ? amd::kMaxSurfaces : (pool_member_ ? 12 : 24); }

  // A submission: which of this decoder's lanes, and the lane-local sequence number.  seq 0 = none.
  struct Stamp {
    int lane = 0;
    uint64_t seq = 0;
  };
  bool wait_stamp(const Stamp& st, std::string* error) { return !st.seq || exec_->wait(lanes_[st.lane], st.seq, error); }

  struct Deferred {
    amd::ConvSubmit sub;
    int bounce = -1;  // index into bounce_, -1: straight into caller memory in HBM
  };
  struct SurfaceSlot {
    uint8_t* dev = nullptr;
    // Motion records of the picture in the surface: written by the parser, uploaded from here, and read again by the
    // parser for co-located lookups as long as the picture is a reference.  Two sets with entropy workers, used in
    // turn: the set the last picture in this surface wrote may still be waiting for its upload when the surface is
    // taken again within its GOP (same lane, no wait for the GPU otherwise); the one before that hardly ever is.
    MbMotion* motion_host[2] = {nullptr, nullptr};  // pinned; also the co-located source
    PMotion* pmo_host[2] = {nullptr, nullptr};      // pinned: compact motion records ...
    uint32_t* pmx_host[2] = {nullptr, nullptr};     // ... and their extra stream (row table + difference words)
    Stamp motion[2];                  // submission that still uploads from the set
    int flip = 0;                     // set of the picture in the surface now
    uint64_t busy_seq[kPipeLanes] = {0};  // per lane of this decoder: the latest submission that writes or reads the
                                      // surface (its picture, pictures predicted from it, conversions of it); one
                                      // stamp for all lanes lost the reader on lane B when a conversion on lane A
                                      // followed it (open GOPs, HWANG_TEST_ROTATE_LANES)
    int lane = 0;                     // lane index of the picture that was reconstructed into it
    uint64_t order = 0;               // when it was last used (for "the oldest")
    int held = 0;
    bool submitted = true;            // its picture has been handed to the executor (end_picture)
    bool abandoned = false;           // ... or never will be
    std::deque<Deferred> deferred;    // conversions waiting for that
  };
  // which motion record set the next picture in the surface takes
  int next_set(const SurfaceSlot& s) const { return num_workers_ ? s.flip ^ 1 : 0; }
  struct Job {
    MbInfo* mb_host = nullptr;
    uint32_t* coef_host = nullptr;
    SliceParams* slice_host = nullptr;
    MbInfo* mb_dev = nullptr;
    MbMotion* motion_dev = nullptr;
    PMotion* pmo_dev = nullptr;
    uint32_t* pmx_dev = nullptr;
    uint32_t* coef_dev = nullptr;
    SliceParams* slice_dev = nullptr;
    void* db_dev = nullptr;  // k_bs -> k_deblock records
    int16_t* resid_dev = nullptr;  // k_residual -> k_intra
    void* band_ctl_dev = nullptr;    // k_mv, k_deblock: hand-off between a picture's workgroups
    Stamp stamp;       // submission that still reads the staging buffers
    uint64_t order = 0;
    int lane = 0;      // lane index the picture in these buffers goes to (begin_picture -> end_picture)
    int motion_set = 0;  // which of the surface's motion record sets the picture uses
    std::atomic<bool> in_use{false};  // between begin_picture and end_picture / abandon_picture
    Job() = default;
    Job(const Job& o) { *this = o; }
    Job& operator=(const Job& o) {
      mb_host = o.mb_host; coef_host = o.coef_host; slice_host = o.slice_host; mb_dev = o.mb_dev; motion_dev = o.motion_dev;
      pmo_dev = o.pmo_dev; pmx_dev = o.pmx_dev; coef_dev = o.coef_dev; slice_dev = o.slice_dev; db_dev = o.db_dev;
      resid_dev = o.resid_dev; band_ctl_dev = o.band_ctl_dev; stamp = o.stamp; order = o.order; lane = o.lane;
      motion_set = o.motion_set;
      in_use.store(o.in_use.load());
      return *this;
    }
  };
  struct Bounce {
    uint8_t* host = nullptr;
    uint8_t* dev = nullptr;
    size_t capacity = 0;
    uint8_t* dst = nullptr;  // pending copy-out target
    bool direct = false;     // ... is page-locked: the D2H copy writes it, nothing is left to copy when it has landed
    size_t size = 0;
    Stamp stamp;             // the conversion that fills it
    bool queued = false;     // ... has been submitted (entropy workers: it may wait for its picture)
    bool copying = false;    // a thread is waiting for that conversion / copying the frame out (out_mu_ released)
  };

  static void mark_busy(SurfaceSlot& s, const Stamp& st) {
    if (st.seq > s.busy_seq[st.lane]) s.busy_seq[st.lane] = st.seq;
  }
  // A lane other than `own_lane` (-1: any lane) that still has work pending on the surface, given how far each lane
  // had got (`done`); -1 if none.
  int busy_elsewhere(const SurfaceSlot& s, int own_lane, const uint64_t* done) const {
    for (int li = 0; li < static_cast<int>(lanes_.size()); ++li)
      if (li != own_lane && s.busy_seq[li] && done[li] < s.busy_seq[li]) return li;
    return -1;
  }

  void unref_surface_locked(int slot, int who) {
    if (slot >= 0 && slot < amd::kMaxSurfaces) surf_[slot].held &= ~who;
  }

  Result wait_result(const Stamp& st) {
    std::string error;
    if (!wait_stamp(st, &error)) return Result(false, "AMDVideoDecoder: " + error);
    return Result();
  }
  // everything this decoder has submitted, on every lane
  Result wait_all() {
    for (size_t i = 0; i < lanes_.size(); ++i)
      HWANG_RETURN_ON_ERROR(wait_result(Stamp{static_cast<int>(i), last_seq_[i].load()}));
    return Result();
  }
  Result take_sink_error(const char* where) {
    std::lock_guard<std::mutex> lk(out_mu_);
    if (sink_error_.empty()) return Result();
    Result r(false, where + sink_error_);
    sink_error_.clear();
    return r;
  }
  // ---- copy-out of fetched frames: pinned bounce buffer -> the caller's (pageable) memory.
  // That is a 6 MB memcpy per 1080p frame into pages the caller has typically never touched: ~2 ms.  Done by the one
  // consumer thread between its get_frame calls (and, as it was, with out_mu_ held) it capped a decoder at ~500
  // frames/s whatever the entropy workers and the GPU delivered.  The copies happen outside the lock now, and a
  // decoder with entropy workers has kCopiers threads for them; the consumer thread copies only when it needs a
  // buffer that nobody has started on.
  // (out_mu_ held) a frame whose conversion has been submitted and that nobody is copying yet, oldest first
  int claimable_bounce_locked() const {
    const int n = static_cast<int>(bounce_.size());
    for (int k = 0; k < n; ++k) {
      const int i = (bounce_cursor_ + k) % n;
      if (bounce_[i].dst && bounce_[i].queued && !bounce_[i].copying) return i;
    }
    return -1;
  }
  // Waits for bounce i's conversion and copies the frame out; out_mu_ is released meanwhile.
  void copy_out(std::unique_lock<std::mutex>& lk, int i) {
    Bounce& b = bounce_[i];
    b.copying = true;
    const Stamp st = b.stamp;
    uint8_t* const dst = b.dst;
    const size_t size = b.size;
    const bool direct = b.direct;
    lk.unlock();
    std::string error;
    const bool ok = wait_stamp(st, &error);
    trace::event("copy-out starts", static_cast<long>(size));
    if (ok && !direct) std::memcpy(dst, b.host, size);
    trace::event("copy-out done");
    lk.lock();
    if (!ok && copy_error_.empty()) copy_error_ = error;
    b.dst = nullptr;
    b.copying = false;
    out_cv_.notify_all();
  }
  void copier_loop() {
    std::unique_lock<std::mutex> lk(out_mu_);
    for (;;) {
      int i = -1;
      out_cv_.wait(lk, [&] { return copiers_stop_ || (i = claimable_bounce_locked()) >= 0; });
      if (copiers_stop_) return;
      copy_out(lk, i);
    }
  }
  void stop_copiers() {
    {
      std::lock_guard<std::mutex> lk(out_mu_);
      copiers_stop_ = true;
    }
    out_cv_.notify_all();
    for (std::thread& t : copiers_) t.join();
    copiers_.clear();
  }
  // Every fetched frame whose conversion has been submitted reaches its destination (this thread helps).  A frame
  // whose conversion never was submitted (its picture did not arrive) is given up.
  void drain_bounces(std::unique_lock<std::mutex>& lk) {
    for (;;) {
      const int i = claimable_bounce_locked();
      if (i >= 0) {
        copy_out(lk, i);
        continue;
      }
      bool copying = false;
      for (Bounce& b : bounce_) copying |= b.copying;
      if (!copying) break;
      out_cv_.wait(lk);
    }
    for (Bounce& b : bounce_) b.dst = nullptr;
  }

  Result drain_and_reset() {
    const Result waited = wait_all();
    std::unique_lock<std::mutex> lk(out_mu_);
    // frames the caller has already taken with get_frame() are his: their copy-outs
    // complete (a re-configure between two clips of a batched call used to drop them) - also when the executor
    // has failed: a copier that is in the middle of a frame finishes before anything is reset or reported
    drain_bounces(lk);
    if (!waited.ok) return waited;
    // what the request that is over still had to say (a fetched frame whose picture was given up after the call had
    // already returned with another error) does not belong to the one that starts now
    sink_error_.clear();
    copy_error_.clear();
    outq_.clear();
    buffered_.store(0);
    note_undelivered(-undelivered_.load(std::memory_order_acquire));
    for (SurfaceSlot& s : surf_) s.held = 0;
    return Result();
  }

  bool ensure_geometry(const h264::PicSetup& s, std::string* error) {
    // The output size was fixed by configure() (get_frame's contract is w*h*3 of
    // the container's size): an in-band SPS may not change what a frame is.
    const int cw = s.mb_width * 16 - s.crop_left - s.crop_right, ch = s.mb_height * 16 - s.crop_top - s.crop_bottom;
    if (cw != static_cast<int>(info_.width) || ch != static_cast<int>(info_.height)) {
      *error = "AMDVideoDecoder: the stream switches to " + std::to_string(cw) + "x" + std::to_string(ch) +
               " pictures after configure() announced " + std::to_string(info_.width) + "x" +
               std::to_string(info_.height);
      return false;
    }
    if (s.mb_width == mbw_ && s.mb_height == mbh_) {
      if ((s.crop_left != crop_left_ || s.crop_top != crop_top_) && frames_pending_output()) {
        *error = "AMDVideoDecoder: the cropping offsets change while decoded frames are still queued";
        return false;
      }
      crop_left_ = s.crop_left;
      crop_top_ = s.crop_top;
      return true;
    }
    if (s.mb_height > amd::kMaxMbRows || s.mb_width > amd::kMaxMbRows) {
      *error = "AMDVideoDecoder: pictures beyond 8192 samples a side are not supported";
      return false;
    }
    // New coded geometry: every surface and staging buffer is re-made.  Frames
    // that were emitted but not fetched yet still name the old surfaces, so
    // this is only legal when nothing is queued (configure() drains; a stream
    // that changes its coded size at an IDR with output pending is refused).
    if (frames_pending_output()) {
      *error = "AMDVideoDecoder: the coded picture size changes while decoded frames are still queued";
      return false;
    }
    (void)wait_all();
    {
      std::unique_lock<std::mutex> lk(out_mu_);
      drain_bounces(lk);
    }
    free_buffers();  // also zeroes mbw_/mbh_: they are set again only when every allocation has succeeded
    const int new_mbw = s.mb_width, new_mbh = s.mb_height;
    struct Rollback {
      AMDVideoDecoder* d;
      bool armed = true;
      ~Rollback() {
        if (armed) d->free_buffers();
      }
    } rollback{this};
    mbw_ = new_mbw;
    mbh_ = new_mbh;
    crop_left_ = s.crop_left;
    crop_top_ = s.crop_top;
    y_pitch_ = ((mbw_ * 16 + 255) / 256) * 256;
    c_pitch_ = y_pitch_ / 2;
    u_offset_ = static_cast<size_t>(y_pitch_) * mbh_ * 16;
    v_offset_ = u_offset_ + static_cast<size_t>(c_pitch_) * mbh_ * 8;
    surface_bytes_ = v_offset_ + static_cast<size_t>(c_pitch_) * mbh_ * 8;
    const size_t nmb = static_cast<size_t>(mbw_) * mbh_;
    // worst case per macroblock: kMaxSparseCoefs entries, or 26 dense groups (fewer words)
    coef_capacity_ = nmb * amd::kMaxSparseCoefs;
    pmx_words_ = static_cast<size_t>(mbh_) + nmb * 15;  // row table + up to 15 further partitions per macroblock
    if (num_workers_) {
      // Staging sets: the pictures in the workers' hands plus the queue towards the GPU - as deep as kPipeStagingBytes
      // of pinned memory allow (1080p: 7.3 MB a set, the full depth; 4K: 29 MB a set, the queue shrinks to a few).
      const size_t per_job = nmb * sizeof(MbInfo) + coef_capacity_ * sizeof(uint32_t) + kMaxSlices * sizeof(SliceParams);
      const int affordable = static_cast<int>(std::min<size_t>(kPipeStagingBytes / per_job, 4096));
      jobs_.clear();
      jobs_.resize(std::max(max_in_flight_ + 8, std::min(max_in_flight_ + pipe_depth_, affordable)));
    }
    // (with workers chosen by the backend itself only a floor of sets is made here, the rest on demand: begin_picture)
    const int eager = auto_workers_ ? std::min<int>(static_cast<int>(jobs_.size()), 8) : static_cast<int>(jobs_.size());
    jobs_ready_ = 0;
    jobs_alloc_failed_ = false;
    for (int i = 0; i < eager; ++i) {
      if (!alloc_job(jobs_[i], error)) return false;
      jobs_ready_ = i + 1;
    }
    job_cursor_ = 0;
    rollback.armed = false;
    return true;
  }

  // One staging set: pinned host buffers for the records of a picture, their device copies and the per-picture scratch.
  bool alloc_job(Job& j, std::string* error) {
    const size_t nmb = static_cast<size_t>(mbw_) * mbh_;
    if (hipHostMalloc(reinterpret_cast<void**>(&j.mb_host), nmb * sizeof(MbInfo), hipHostMallocDefault) != hipSuccess ||
        hipHostMalloc(reinterpret_cast<void**>(&j.coef_host), coef_capacity_ * sizeof(uint32_t), hipHostMallocDefault) != hipSuccess ||
        hipHostMalloc(reinterpret_cast<void**>(&j.slice_host), kMaxSlices * sizeof(SliceParams), hipHostMallocDefault) != hipSuccess ||
        hipMalloc(reinterpret_cast<void**>(&j.mb_dev), nmb * sizeof(MbInfo)) != hipSuccess ||
        hipMalloc(reinterpret_cast<void**>(&j.motion_dev), nmb * sizeof(MbMotion)) != hipSuccess ||
        hipMalloc(reinterpret_cast<void**>(&j.pmo_dev), nmb * sizeof(PMotion)) != hipSuccess ||
        hipMalloc(reinterpret_cast<void**>(&j.pmx_dev), (pmx_words_ + 64) * sizeof(uint32_t)) != hipSuccess ||
        hipMalloc(reinterpret_cast<void**>(&j.coef_dev), coef_capacity_ * sizeof(uint32_t)) != hipSuccess ||
        hipMalloc(reinterpret_cast<void**>(&j.slice_dev), kMaxSlices * sizeof(SliceParams)) != hipSuccess ||
        hipMalloc(&j.db_dev, amd::deblock_scratch_bytes(static_cast<int>(nmb))) != hipSuccess ||
        hipMalloc(reinterpret_cast<void**>(&j.resid_dev), nmb * 384 * sizeof(int16_t)) != hipSuccess ||
        hipMalloc(&j.band_ctl_dev, amd::band_ctl_bytes(mbh_)) != hipSuccess ||
        hipMemset(j.band_ctl_dev, 0, amd::band_ctl_bytes(mbh_)) != hipSuccess) {
      (void)hipGetLastError();
      free_job(j);  // (what was allocated before the call that failed)
      *error = "AMDVideoDecoder: out of memory for staging buffers";
      return false;
    }
    if (!gpu_visible(j.mb_host) || !gpu_visible(j.coef_host) || !gpu_visible(j.slice_host)) {
      free_job(j);
      *error = "AMDVideoDecoder: pinned staging memory is not mapped at its host address on this device";
      return false;
    }
    j.stamp = Stamp();
    return true;
  }

  bool frames_pending_output() { return outq_.size() != 0; }

  bool alloc_surface(int slot, std::string* error) {
    if (surf_[slot].dev) return true;
    SurfaceSlot s;
    const size_t nmb = static_cast<size_t>(mbw_) * mbh_;
    auto give_up = [&](const char* msg) {
      if (s.dev) (void)hipFree(s.dev);
      for (int k = 0; k < 2; ++k) {
        if (s.motion_host[k]) (void)hipHostFree(s.motion_host[k]);
        if (s.pmo_host[k]) (void)hipHostFree(s.pmo_host[k]);
        if (s.pmx_host[k]) (void)hipHostFree(s.pmx_host[k]);
      }
      std::lock_guard<std::mutex> lk(out_mu_);
      surf_[slot].held = 0;
      *error = msg;
      return false;
    };
    // + 64: k_inter fetches unaligned dwords and may read 3 bytes past a plane's last sample
    if (hipMalloc(reinterpret_cast<void**>(&s.dev), surface_bytes_ + 64) != hipSuccess)
      return give_up("AMDVideoDecoder: out of memory for a decode surface");
    for (int k = 0; k < (num_workers_ ? 2 : 1); ++k) {
      if (hipHostMalloc(reinterpret_cast<void**>(&s.motion_host[k]), nmb * sizeof(MbMotion), hipHostMallocDefault) != hipSuccess ||
          hipHostMalloc(reinterpret_cast<void**>(&s.pmo_host[k]), nmb * sizeof(PMotion), hipHostMallocDefault) != hipSuccess ||
          hipHostMalloc(reinterpret_cast<void**>(&s.pmx_host[k]), pmx_words_ * sizeof(uint32_t), hipHostMallocDefault) != hipSuccess)
        return give_up("AMDVideoDecoder: out of memory for a decode surface");
      if (!gpu_visible(s.motion_host[k]) || !gpu_visible(s.pmo_host[k]) || !gpu_visible(s.pmx_host[k]))
        return give_up("AMDVideoDecoder: pinned staging memory is not mapped at its host address on this device");
    }
    std::lock_guard<std::mutex> lk(out_mu_);  // (a worker may be copying the surface table in end_picture)
    s.held = surf_[slot].held;
    surf_[slot] = s;
    return true;
  }

  // k_upload reads the staging buffers through their host addresses.
  static bool gpu_visible(void* host) {
    void* dev = nullptr;
    return hipHostGetDevicePointer(&dev, host, 0) == hipSuccess && dev == host;
  }

  static void free_job(Job& j) {
    if (j.mb_host) (void)hipHostFree(j.mb_host);
    if (j.coef_host) (void)hipHostFree(j.coef_host);
    if (j.slice_host) (void)hipHostFree(j.slice_host);
    if (j.mb_dev) (void)hipFree(j.mb_dev);
    if (j.motion_dev) (void)hipFree(j.motion_dev);
    if (j.pmo_dev) (void)hipFree(j.pmo_dev);
    if (j.pmx_dev) (void)hipFree(j.pmx_dev);
    if (j.coef_dev) (void)hipFree(j.coef_dev);
    if (j.slice_dev) (void)hipFree(j.slice_dev);
    if (j.db_dev) (void)hipFree(j.db_dev);
    if (j.resid_dev) (void)hipFree(j.resid_dev);
    if (j.band_ctl_dev) (void)hipFree(j.band_ctl_dev);
    j = Job();
  }

  void free_buffers() {
    for (Job& j : jobs_) free_job(j);
    for (SurfaceSlot& s : surf_) {
      if (s.dev) (void)hipFree(s.dev);
      for (int k = 0; k < 2; ++k) {
        if (s.motion_host[k]) (void)hipHostFree(s.motion_host[k]);
        if (s.pmo_host[k]) (void)hipHostFree(s.pmo_host[k]);
        if (s.pmx_host[k]) (void)hipHostFree(s.pmx_host[k]);
      }
      s = SurfaceSlot();
    }
    for (Bounce& b : bounce_) {
      if (b.host) (void)hipHostFree(b.host);
      if (b.dev) (void)hipFree(b.dev);
      b = Bounce();
    }
    mbw_ = mbh_ = 0;
  }

  void teardown() {
    note_undelivered(-undelivered_.load(std::memory_order_acquire));  // (frames nobody will fetch any more)
    if (getenv("HWANG_EXEC_DEBUG"))
      fprintf(stderr, "[decoder lane %d] samples fed %lld in %.3f s of feed() (%.3f s of it stalled on the GPU: motion waits %lld, "
                      "job-ring waits %lld, surface waits %lld), workers %d\n", lanes_.empty() ? -1 : lanes_[0],
              (long long)counters_.samples_fed, (counters_.entropy_ns + counters_.stall_ns) / 1e9, counters_.stall_ns / 1e9,
              (long long)dbg_motion_waits_, (long long)dbg_job_waits_, (long long)dbg_busy_waits_, num_workers_);
    if (getenv("HWANG_EXEC_DEBUG"))
      fprintf(stderr, "[decoder lane %d] %lld surfaces allocated in %.3f s; the consumer waited %.3f s for a bounce buffer\n",
              lanes_.empty() ? -1 : lanes_[0], (long long)dbg_allocs_, dbg_alloc_ns_ / 1e9, dbg_bounce_wait_ns_ / 1e9);
    if (getenv("HWANG_EXEC_DEBUG") && num_workers_) {
      int64_t ps[6];
      h264::pipe_debug_stats(ps);
      fprintf(stderr, "[decoder lane %d] process-wide: parser waited %.3f s for room in the window, %.3f s for a surface; workers "
                      "%.3f s on slice data of %lld pictures, %.3f s in the hand-over\n", lanes_.empty() ? -1 : lanes_[0], ps[0] / 1e9,
              ps[1] / 1e9, ps[2] / 1e9, (long long)ps[4], ps[3] / 1e9);
    }
    if (getenv("HWANG_EXEC_DEBUG") && dbg_none_ready_)
      fprintf(stderr, "[decoder lane %d] no surface ready %lld times; then on average %.1f held by the parser, %.1f by the "
                      "output queue, %.1f free but still in use on the GPU\n", lanes_.empty() ? -1 : lanes_[0],
              (long long)dbg_none_ready_, 1.0 * dbg_held_parser_ / dbg_none_ready_, 1.0 * dbg_held_output_ / dbg_none_ready_,
              1.0 * dbg_free_busy_ / dbg_none_ready_);
    if (getenv("HWANG_EXEC_DEBUG") && dbg_none_ready_)
      fprintf(stderr, "[decoder lane %d] ... of the free ones %.1f with a pending upload, %.1f with pending GPU work; %.1f lanes "
                      "with %.1f submissions pending in all\n", lanes_.empty() ? -1 : lanes_[0],
              1.0 * dbg_unsettled_motion_ / dbg_none_ready_, 1.0 * dbg_unsettled_busy_ / dbg_none_ready_,
              1.0 * dbg_pending_lanes_ / dbg_none_ready_, 1.0 * dbg_pending_total_ / dbg_none_ready_);
    stop_copiers();
    if (exec_ && exec_->stopped()) {
      // Destroyed after hwang_shutdown() (interpreter exit destroys module-level
      // objects after its atexit handlers): the streams are gone and the HIP
      // runtime may be tearing down; leave the device memory to process exit.
      parser_.reset();
      pool_.reset();
      return;
    }
    if (hipSetDevice(device_) != hipSuccess) return;
    parser_.reset();  // (entropy workers: every picture in flight is handed over first)
    pool_.reset();
    if (exec_) {
      (void)wait_all();
      for (int lane : lanes_) exec_->unregister_lane(lane);
    }
    free_buffers();
  }

  // Is `p` in HBM?  If it is another GPU's HBM (one process driving several
  // devices: a request's output lives on one of them), the conversion kernel
  // writes it over xGMI - point-to-point stores, no copy, no collective - for
  // which peer access is switched on the first time the pair is seen.
  // ... or page-locked host memory (hipHostMalloc / hipHostRegister: `*pinned_host`)?  Then the copy engine can write
  // the frame where the caller wants it and the bounce buffer's host half - and a 6 MB memcpy on a CPU - is not needed.
  bool is_device_pointer(const void* p, Result* peer_error, bool* pinned_host) {
    *pinned_host = false;
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, p) != hipSuccess) {
      (void)hipGetLastError();  // unregistered host memory reports an error
      return false;
    }
    if (attr.type == hipMemoryTypeHost) *pinned_host = true;
    if (attr.type != hipMemoryTypeDevice) return false;
    if (attr.device != device_ && attr.device != peer_enabled_) {
      const hipError_t e = hipDeviceEnablePeerAccess(attr.device, 0);
      if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) {
        *peer_error = Result(false, "AMDVideoDecoder: output buffer lives on GPU " + std::to_string(attr.device) +
                                        " and GPU " + std::to_string(device_) + " cannot reach it (" +
                                        hipGetErrorString(e) + ")");
        return true;
      }
      (void)hipGetLastError();
      peer_enabled_ = attr.device;
    }
    return true;
  }

  Result take_frame(uint8_t* dst, size_t dst_size, bool yuv) {
    HWANG_HIP_RETURN(hipSetDevice(device_));
    const size_t w = info_.width, h = info_.height;
    const size_t cw = (w + 1) / 2, ch = (h + 1) / 2;
    const size_t need = yuv ? w * h + 2 * cw * ch : w * h * 3;
    int slot;
    // (one consumer thread per instance - get_frame / discard_frame - so peek-then-pop is safe)
    if (!outq_.try_peek(slot)) return Result();  // reference behaviour: nothing to do
    if (dst_size < need) return Result(false, "AMDVideoDecoder::get_frame: buffer too small");
    if (slot < 0)
      return Result(false,
                    "AMDVideoDecoder::get_frame: this output position was excluded by the sparse hint "
                    "(set_wanted) and was not decoded");
    outq_.try_pop(slot);
    buffered_.fetch_sub(1, std::memory_order_acq_rel);
    note_undelivered(-1);
    trace::event("get_frame", slot);
    const SurfaceSlot& s = surf_[slot];
    amd::ConvSubmit sub;
    std::memset(&sub.desc, 0, sizeof(sub.desc));
    sub.desc.y = s.dev + static_cast<size_t>(crop_top_) * y_pitch_ + crop_left_;
    sub.desc.u = s.dev + u_offset_ + static_cast<size_t>(crop_top_ / 2) * c_pitch_ + crop_left_ / 2;
    sub.desc.v = s.dev + v_offset_ + static_cast<size_t>(crop_top_ / 2) * c_pitch_ + crop_left_ / 2;
    sub.desc.y_pitch = y_pitch_;
    sub.desc.c_pitch = c_pitch_;
    sub.desc.width = static_cast<int>(w);
    sub.desc.height = static_cast<int>(h);
    sub.desc.yuv = yuv ? 1 : 0;
    sub.bytes = need + (yuv ? need : w * h + 2 * cw * ch);
    Result r;
    int bounce = -1;
    bool pinned_host = false;
    if (is_device_pointer(dst, &r, &pinned_host)) {
      if (!r.ok) return finish_take(slot, r);
      sub.desc.dst = dst;  // the kernel writes caller memory in HBM directly
    } else {
      r = acquire_bounce(need, &bounce);
      if (!r.ok) return finish_take(slot, r);
      Bounce& b = bounce_[bounce];
      sub.desc.dst = b.dev;
      sub.d2h_dst = pinned_host ? dst : b.host;  // (page-locked caller memory: the D2H copy lands in it)
      sub.d2h_bytes = need;
      std::lock_guard<std::mutex> lk(out_mu_);
      b.dst = dst;
      b.direct = pinned_host;
      b.size = need;
      b.queued = false;
      b.stamp = Stamp();
    }
    // behind the picture it shows, on that picture's lane - which, with entropy workers, may not have been handed to
    // the executor yet: then the conversion waits with the surface and end_picture submits it
    int lane_index;
    {
      std::lock_guard<std::mutex> lk(out_mu_);
      SurfaceSlot& cs = surf_[slot];
      if (cs.abandoned) {
        if (bounce >= 0) bounce_[bounce].dst = nullptr;
        unref_surface_locked(slot, kHeldByOutput);
        return Result(false, "AMDVideoDecoder::get_frame: this frame could not be decoded (its picture or an earlier one "
                             "of the interval failed)");
      }
      if (!cs.submitted) {
        Deferred def;
        def.sub = sub;
        def.bounce = bounce;
        cs.deferred.push_back(def);
        return Result();
      }
      lane_index = cs.lane;
    }
    submit_conversion(slot, lane_index, sub, bounce);
    return Result();
  }

  Result finish_take(int slot, const Result& r) {
    std::lock_guard<std::mutex> lk(out_mu_);
    unref_surface_locked(slot, kHeldByOutput);
    return r;
  }

  // A bounce buffer whose previous copy-out has been delivered.
  Result acquire_bounce(size_t need, int* out) {
    std::unique_lock<std::mutex> lk(out_mu_);
    const int bi = bounce_cursor_;
    Bounce& b = bounce_[bi];
    bounce_cursor_ = (bounce_cursor_ + 1) % static_cast<int>(bounce_.size());
    if (b.dst) {
      // the ring wrapped: this buffer's conversion has to be submitted (it may wait for its picture), then to land,
      // then the frame is delivered - by a copier thread, or here when none has picked it up
      const auto w0 = std::chrono::steady_clock::now();
      out_cv_.wait(lk, [&] { return b.queued || !b.dst; });
      if (b.dst && !b.copying) copy_out(lk, bi);
      out_cv_.wait(lk, [&] { return !b.dst; });
      dbg_bounce_wait_ns_ += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - w0).count();
    }
    if (!copy_error_.empty()) {
      Result r(false, "AMDVideoDecoder: " + copy_error_);
      copy_error_.clear();
      return r;
    }
    lk.unlock();
    if (b.capacity < need) {
      if (b.host) (void)hipHostFree(b.host);
      if (b.dev) (void)hipFree(b.dev);
      b.host = b.dev = nullptr;
      b.capacity = 0;
      HWANG_HIP_RETURN(hipHostMalloc(reinterpret_cast<void**>(&b.host), need, hipHostMallocDefault));
      HWANG_HIP_RETURN(hipMalloc(reinterpret_cast<void**>(&b.dev), need));
      b.capacity = need;
    }
    *out = bi;
    return Result();
  }

  const int device_;
  int peer_enabled_ = -1;  // last peer device whose HBM was made reachable from device_
  amd::GpuExecutor* exec_ = nullptr;
  const int num_workers_;       // entropy workers (0: slice data is decoded inside feed())
  const bool pool_member_;      // created with num_devices = 0: a lane of a caller that brings the parallelism
  const bool auto_workers_;     // created with num_devices = 1: the worker count is this backend's own choice
  int max_in_flight_ = 0;       // pictures whose slice data may be with the workers at a time
  int64_t dbg_alloc_ns_ = 0, dbg_allocs_ = 0, dbg_bounce_wait_ns_ = 0;  // HWANG_EXEC_DEBUG
  // test hook (tests/test_gpu_workers.py): the lanes rotate at EVERY sample, so every picture's references were
  // written on another lane - what a stream with open GOPs does at its sync samples, all the time
  const bool rotate_always_ = getenv("HWANG_TEST_ROTATE_LANES") != nullptr;
  const bool host_mv_ = getenv("HWANG_HOST_MV") != nullptr;  // experiment: P vectors derived on the host
  int pipe_depth_ = 0;          // ... and how many more may be queued towards the GPU (fewer when the staging budget says so)
  std::unique_ptr<h264::WorkerPool> pool_;       // num_devices > 1: this decoder's own threads
  h264::WorkerPool* shared_pool_ = nullptr;      // num_devices = 1: the process-wide pool (shared_entropy_pool)
  std::vector<std::thread> copiers_;
  bool copiers_stop_ = false;   // (out_mu_)
  std::string copy_error_;      // (out_mu_) a conversion that failed under a copier; reported by the next get_frame
  std::vector<int> lanes_;      // executor lanes; one without workers, kPipeLanes with
  int cur_lane_ = 0;            // index into lanes_: where the current interval's pictures go
  uint64_t use_order_ = 0;      // (out_mu_)
  std::atomic<int64_t> bytes_uploaded_{0};
  // Highest sequence number this decoder has submitted, per lane.  Written by the feeder
  // thread or an entropy worker (end_picture) and by the caller thread (take_frame) under
  // DecoderAutomata, so it only ever moves forward (note_seq): a stale store
  // from one thread must not hide the other's later submission from
  // wait_until_frames_copied().
  std::unique_ptr<std::atomic<uint64_t>[]> last_seq_;
  void note_seq(const Stamp& st) {
    std::atomic<uint64_t>& a = last_seq_[st.lane];
    uint64_t cur = a.load(std::memory_order_relaxed);
    while (cur < st.seq && !a.compare_exchange_weak(cur, st.seq, std::memory_order_release)) {
    }
  }
  uint32_t epoch_ = 0;  // per submitted picture, see JobDesc::epoch
  FrameInfo info_{};
  std::unique_ptr<h264::Parser> parser_;
  std::string sink_error_;
  std::vector<uint64_t> pending_wanted_, pending_keyframes_;

  int mbw_ = 0, mbh_ = 0, y_pitch_ = 0, c_pitch_ = 0, crop_left_ = 0, crop_top_ = 0;
  size_t u_offset_ = 0, v_offset_ = 0, surface_bytes_ = 0, coef_capacity_ = 0, pmx_words_ = 0;
  int64_t dbg_motion_waits_ = 0, dbg_job_waits_ = 0, dbg_busy_waits_ = 0;
  int64_t dbg_none_ready_ = 0, dbg_held_parser_ = 0, dbg_held_output_ = 0, dbg_free_busy_ = 0;
  int64_t dbg_unsettled_motion_ = 0, dbg_unsettled_busy_ = 0, dbg_pending_lanes_ = 0, dbg_pending_total_ = 0;
  SurfaceSlot surf_[amd::kMaxSurfaces];
  std::vector<Job> jobs_;
  int job_cursor_ = 0;
  int jobs_ready_ = 0;  // jobs_[0 .. jobs_ready_) have their buffers (the rest are made on demand, alloc_job)
  bool jobs_alloc_failed_ = false;  // an on-demand set could not be allocated: stop trying until the buffers are rebuilt
  std::vector<Bounce> bounce_;
  int bounce_cursor_ = 0;

  std::mutex out_mu_;
  std::condition_variable out_cv_;  // a deferred conversion was submitted or cancelled
  Queue<int> outq_{4096};  // surface slot per frame in display order, -1 = skipped (hwang/util/queue.h's role)
  std::atomic<int> buffered_{0};
  // real frames among them (skipped output positions hold no surface): what GpuExecutor::replay must not run over
  std::atomic<int> undelivered_{0};
  void note_undelivered(int delta) {
    if (!delta) return;
    undelivered_.fetch_add(delta, std::memory_order_acq_rel);
    if (exec_) exec_->note_undelivered(delta);
  }

  BackendCounters counters_;
  int64_t decoded_base_ = 0, skipped_base_ = 0, folded_decoded_ = 0, folded_skipped_ = 0;
};

}  // namespace

int amd_device_numa_node(int32_t device_id) {
  char bus_id[64] = {0};
  if (hipDeviceGetPCIBusId(bus_id, sizeof(bus_id), device_id) != hipSuccess) {
    (void)hipGetLastError();
    return -1;
  }
  return numa::node_of_pci_device(bus_id);
}

// `num_devices` (video_decoder_factory.cpp:90-91 hands it to the backend; the reference's software backend reads it as
// libavcodec's thread_count, software_video_decoder.cpp:148, its NVIDIA backend ignores it):
//   n > 1  n threads entropy decode consecutive pictures of this ONE decoder at the same time;
//   1      what the reference's own Python passes (python/hwang/decoder.py:28), i.e. "one decoder, nothing said about
//          threads": the backend picks the worker count itself - the usable CPUs, at most 16; HWANG_ENTROPY_WORKERS
//          overrides (0 or 1: the thread that calls feed() decodes alone, as in rounds 1-3);
//   0      the caller is the parallelism (ParallelRetriever: one decoder per lane thread): no workers.
VideoDecoderInterface* make_amd_video_decoder(int32_t device_id, uint32_t num_devices) {
  int count = 0;
  const hipError_t ce = hipGetDeviceCount(&count);
  if (ce != hipSuccess || device_id < 0 || device_id >= count) {
    fprintf(stderr, "hwang_amd: no GPU %d (%d visible%s%s)\n", device_id, count, ce != hipSuccess ? ": " : "",
            ce != hipSuccess ? hipGetErrorString(ce) : "");
    return nullptr;
  }
  int workers = static_cast<int>(std::min<uint32_t>(num_devices, 64));
  const bool auto_workers = num_devices == 1;
  if (auto_workers) {
    const char* e = getenv("HWANG_ENTROPY_WORKERS");
    workers = e && *e ? std::max(0, std::min(atoi(e), 64)) : std::max(2, std::min<int>(usable_cpus(), 16));
  }
  auto* d = new AMDVideoDecoder(device_id, workers, num_devices == 0, auto_workers && workers > 1);
  std::string error;
  if (!d->open(&error)) {
    fprintf(stderr, "hwang_amd: cannot open the MI355X backend on GPU %d: %s\n", device_id, error.c_str());
    delete d;
    return nullptr;
  }
  return d;
}

}  // namespace hwang
