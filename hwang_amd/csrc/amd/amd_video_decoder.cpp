// MI355X backend behind VideoDecoderInterface.
//
// Shape of a device backend in the reference: impls/nvidia/
// nvidia_video_decoder.cpp:227-470 (configure / feed -> parser callbacks ->
// ring of decoded surfaces / get_frame -> colour kernel -> copy).  Here the
// "parser" is this repo's host-side H.264 front end (h264/h264_parser) and
// the "decoder" is the set of HIP kernels in k_recon.hip / k_rgb.hip, all on
// one HIP stream per instance:
//
//   feed(sample)  host: entropy decode + motion derivation -> pinned staging
//                 stream: H2D records -> k_inter -> k_intra -> k_deblock
//   get_frame()   stream: k_rgb (surface -> caller memory if it is in HBM,
//                 else -> pinned bounce buffer -> caller memory)
//
// Nothing here waits for the GPU except when a staging slot or a surface has
// to be reused, and wait_until_frames_copied().
#include "amd_video_decoder.h"

#include "../h264/h264_parser.h"
#include "../sparse_hints.h"
#include "frame_job.h"
#include "hip_check.h"
#include "prof.h"

#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <cstring>
#include <deque>
#include <memory>
#include <mutex>
#include <vector>

namespace hwang {

namespace amd {
hipError_t launch_inter(const PictureParams& pp, const MbInfo* mbs, const MbMotion* motions, const int16_t* coefs,
                        const SliceParams* slices, hipStream_t stream);
hipError_t launch_intra(const PictureParams& pp, const MbInfo* mbs, const int16_t* coefs, unsigned* sync,
                        hipStream_t stream);
hipError_t launch_deblock(const PictureParams& pp, const MbInfo* mbs, const MbMotion* motions,
                          const SliceParams* slices, unsigned* sync, hipStream_t stream);
}  // namespace amd
hipError_t launch_yuv420_to_rgb24(const uint8_t* y, const uint8_t* u, const uint8_t* v, int y_stride, int c_stride,
                                  int width, int height, uint8_t* rgb, hipStream_t stream);

namespace {

using amd::MbInfo;
using amd::MbMotion;
using amd::SliceParams;

constexpr int kJobs = 4;        // staging ring depth
constexpr int kMaxSlices = 256;
constexpr int kBounce = 4;      // pinned output bounce buffers

class AMDVideoDecoder : public VideoDecoderInterface,
                        public SparseDecodeHints,
                        public PlanarOutput,
                        public CountedBackend,
                        public h264::PictureSink {
 public:
  explicit AMDVideoDecoder(int device_id) : device_(device_id) {}
  ~AMDVideoDecoder() override { teardown(); }

  bool open(std::string* error) {
    if (hipSetDevice(device_) != hipSuccess) {
      *error = "cannot select GPU " + std::to_string(device_);
      return false;
    }
    if (hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking) != hipSuccess) {
      *error = "cannot create a HIP stream";
      return false;
    }
    return true;
  }

  // ------------------------------------------------------ VideoDecoderInterface
  Result configure(const FrameInfo& metadata, const std::vector<uint8_t>& extradata) override {
    HWANG_HIP_RETURN(hipSetDevice(device_));
    if (metadata.format != "h264" && metadata.format != "avc1") {
      return Result(false, "AMDVideoDecoder: unsupported format '" + metadata.format +
                               "' (H.264 in avc1/h264 sample entries only)");
    }
    HWANG_RETURN_ON_ERROR(drain_and_reset());
    info_ = metadata;
    parser_.reset(new h264::Parser(this));
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
    }
    return Result();
  }

  Result feed(const uint8_t* encoded_buffer, size_t encoded_size, bool /*keyframe*/) override {
    if (encoded_size == 0) return Result();  // end-of-interval marker; flush() follows
    if (!parser_) return Result(false, "AMDVideoDecoder::feed before configure");
    HWANG_HIP_RETURN(hipSetDevice(device_));
    const auto t0 = std::chrono::steady_clock::now();
    sink_error_.clear();
    const bool ok = parser_->decode(encoded_buffer, encoded_size);
    counters_.entropy_ns +=
        std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
    ++counters_.samples_fed;
    if (!ok) return Result(false, "AMDVideoDecoder::feed: " + parser_->error());
    return Result();
  }

  Result flush() override {
    if (!parser_) return Result();
    HWANG_HIP_RETURN(hipSetDevice(device_));
    if (!parser_->flush()) return Result(false, "AMDVideoDecoder::flush: " + parser_->error());
    pending_wanted_.clear();
    return Result();
  }

  Result discard_frame() override {
    std::lock_guard<std::mutex> lk(out_mu_);
    if (outq_.empty()) return Result();
    const int slot = outq_.front();
    outq_.pop_front();
    buffered_.store(static_cast<int>(outq_.size()));
    if (slot >= 0) unref_surface_locked(slot, kHeldByOutput);
    return Result();
  }

  Result get_frame(uint8_t* decoded_buffer, size_t decoded_size) override {
    return take_frame(decoded_buffer, decoded_size, false);
  }
  Result get_frame_yuv(uint8_t* decoded_buffer, size_t decoded_size) override {
    return take_frame(decoded_buffer, decoded_size, true);
  }

  int decoded_frames_buffered() override { return buffered_.load(std::memory_order_acquire); }

  Result wait_until_frames_copied() override {
    HWANG_HIP_RETURN(hipSetDevice(device_));
    HWANG_HIP_RETURN(hipStreamSynchronize(stream_));
    std::lock_guard<std::mutex> lk(out_mu_);
    for (Bounce& b : bounce_) {
      if (b.dst) {
        std::memcpy(b.dst, b.host, b.size);
        b.dst = nullptr;
      }
    }
    return check_gpu_flags();
  }

  // ------------------------------------------------------------ extensions
  void set_wanted_frames(const std::vector<uint64_t>& ordinals) override {
    pending_wanted_ = ordinals;
    if (parser_) parser_->set_wanted(ordinals);
  }
  BackendCounters counters() override {
    BackendCounters c = counters_;
    if (parser_) {
      c.pictures_reconstructed = parser_->pictures_decoded() - decoded_base_;
      c.pictures_skipped = parser_->pictures_skipped() - skipped_base_;
    }
    return c;
  }
  void reset_counters() override {
    counters_ = BackendCounters();
    decoded_base_ = parser_ ? parser_->pictures_decoded() : 0;
    skipped_base_ = parser_ ? parser_->pictures_skipped() : 0;
  }

  // ------------------------------------------------------------ PictureSink
  bool begin_picture(const h264::PicSetup& s, h264::PicBuffers* out, std::string* error) override {
    if (!ensure_geometry(s, error)) return false;
    // a free surface
    int slot = -1;
    {
      std::lock_guard<std::mutex> lk(out_mu_);
      for (int i = 0; i < amd::kMaxSurfaces; ++i)
        if (!surf_[i].held) {
          slot = i;
          break;
        }
      if (slot >= 0) surf_[slot].held = kHeldByParser;
    }
    if (slot < 0) {
      *error = "AMDVideoDecoder: out of decode surfaces (output frames are not being consumed)";
      return false;
    }
    if (!alloc_surface(slot, error)) return false;
    // the motion buffer of this surface may still be the source of an upload
    if (surf_[slot].motion_busy) {
      if (hipEventSynchronize(surf_[slot].motion_done) != hipSuccess) {
        *error = "AMDVideoDecoder: waiting for a motion upload failed";
        return false;
      }
      surf_[slot].motion_busy = false;
    }
    // a staging job
    Job& j = jobs_[job_cursor_];
    if (j.in_flight) {
      if (hipEventSynchronize(j.done) != hipSuccess) {
        *error = "AMDVideoDecoder: waiting for a staging slot failed";
        return false;
      }
      j.in_flight = false;
      if (*j.timeout_host) {
        *error = "AMDVideoDecoder: GPU row-wavefront timed out";
        return false;
      }
    }
    cur_job_ = job_cursor_;
    job_cursor_ = (job_cursor_ + 1) % kJobs;
    out->mb = j.mb_host;
    out->motion = surf_[slot].motion_host;
    out->coefs = j.coef_host;
    out->coef_capacity = coef_capacity_;
    out->slices = j.slice_host;
    out->max_slices = kMaxSlices;
    out->slot = slot;
    return true;
  }

  bool end_picture(const h264::PicDone& d, std::string* error) override {
    Job& j = jobs_[cur_job_];
    const size_t nmb = static_cast<size_t>(mbw_) * mbh_;
    auto fail = [&](const char* what, hipError_t e) {
      *error = std::string("AMDVideoDecoder: ") + what + ": " + hipGetErrorString(e);
      return false;
    };
    hipError_t e;
    if ((e = hipMemcpyAsync(j.mb_dev, j.mb_host, nmb * sizeof(MbInfo), hipMemcpyHostToDevice, stream_)) != hipSuccess)
      return fail("upload of macroblock records", e);
    if (d.coef_count &&
        (e = hipMemcpyAsync(j.coef_dev, j.coef_host, d.coef_count * 2, hipMemcpyHostToDevice, stream_)) != hipSuccess)
      return fail("upload of coefficients", e);
    if ((e = hipMemcpyAsync(j.slice_dev, j.slice_host, d.num_slices * sizeof(SliceParams), hipMemcpyHostToDevice,
                            stream_)) != hipSuccess)
      return fail("upload of slice parameters", e);
    counters_.bytes_uploaded += nmb * sizeof(MbInfo) + d.coef_count * 2 + d.num_slices * sizeof(SliceParams);
    if (d.has_inter) {
      if ((e = hipMemcpyAsync(j.motion_dev, surf_[d.slot].motion_host, nmb * sizeof(MbMotion),
                              hipMemcpyHostToDevice, stream_)) != hipSuccess)
        return fail("upload of motion records", e);
      (void)hipEventRecord(surf_[d.slot].motion_done, stream_);
      surf_[d.slot].motion_busy = true;
      counters_.bytes_uploaded += nmb * sizeof(MbMotion);
    }
    amd::PictureParams pp;
    std::memset(&pp, 0, sizeof(pp));
    pp.mb_width = mbw_;
    pp.mb_height = mbh_;
    pp.y_pitch = y_pitch_;
    pp.c_pitch = c_pitch_;
    pp.cur_slot = d.slot;
    pp.num_slices = d.num_slices;
    std::memcpy(pp.scaling4, d.scaling4, sizeof(pp.scaling4));
    std::memcpy(pp.scaling8, d.scaling8, sizeof(pp.scaling8));
    for (int i = 0; i < amd::kMaxSurfaces; ++i) {
      pp.surfaces[i].y = surf_[i].dev;
      pp.surfaces[i].u = surf_[i].dev ? surf_[i].dev + u_offset_ : nullptr;
      pp.surfaces[i].v = surf_[i].dev ? surf_[i].dev + v_offset_ : nullptr;
    }
    if (d.has_inter &&
        (e = amd::launch_inter(pp, j.mb_dev, j.motion_dev, j.coef_dev, j.slice_dev, stream_)) != hipSuccess)
      return fail("k_inter launch", e);
    if (d.has_intra && (e = amd::launch_intra(pp, j.mb_dev, j.coef_dev, j.sync_dev, stream_)) != hipSuccess)
      return fail("k_intra launch", e);
    if ((e = amd::launch_deblock(pp, j.mb_dev, j.motion_dev, j.slice_dev, j.sync_dev + sync_words_, stream_)) !=
        hipSuccess)
      return fail("k_deblock launch", e);
    // bring the wavefront timeout words back with the job
    (void)hipMemcpyAsync(j.timeout_host, j.sync_dev + 1, sizeof(unsigned), hipMemcpyDeviceToHost, stream_);
    (void)hipMemcpyAsync(j.timeout_host + 1, j.sync_dev + sync_words_ + 1, sizeof(unsigned), hipMemcpyDeviceToHost,
                         stream_);
    (void)hipEventRecord(j.done, stream_);
    j.in_flight = true;
    return true;
  }

  void output_picture(int slot) override {
    std::lock_guard<std::mutex> lk(out_mu_);
    if (slot >= 0) surf_[slot].held |= kHeldByOutput;
    outq_.push_back(slot);
    buffered_.store(static_cast<int>(outq_.size()), std::memory_order_release);
  }

  void release_surface(int slot) override {
    std::lock_guard<std::mutex> lk(out_mu_);
    unref_surface_locked(slot, kHeldByParser);
  }

 private:
  enum { kHeldByParser = 1, kHeldByOutput = 2 };

  struct SurfaceSlot {
    uint8_t* dev = nullptr;
    MbMotion* motion_host = nullptr;  // pinned; also the co-located source
    hipEvent_t motion_done = nullptr;
    bool motion_busy = false;
    int held = 0;
  };
  struct Job {
    MbInfo* mb_host = nullptr;
    int16_t* coef_host = nullptr;
    SliceParams* slice_host = nullptr;
    unsigned* timeout_host = nullptr;  // [2]
    MbInfo* mb_dev = nullptr;
    MbMotion* motion_dev = nullptr;
    int16_t* coef_dev = nullptr;
    SliceParams* slice_dev = nullptr;
    unsigned* sync_dev = nullptr;  // two blocks of sync_words_
    hipEvent_t done = nullptr;
    bool in_flight = false;
  };
  struct Bounce {
    uint8_t* host = nullptr;
    uint8_t* dev = nullptr;
    size_t capacity = 0;
    uint8_t* dst = nullptr;  // pending copy-out target
    size_t size = 0;
  };

  void unref_surface_locked(int slot, int who) {
    if (slot >= 0 && slot < amd::kMaxSurfaces) surf_[slot].held &= ~who;
  }

  Result check_gpu_flags() {
    for (Job& j : jobs_)
      if (j.timeout_host && (j.timeout_host[0] || j.timeout_host[1])) {
        j.timeout_host[0] = j.timeout_host[1] = 0;
        return Result(false, "AMDVideoDecoder: GPU row-wavefront timed out");
      }
    return Result();
  }

  Result drain_and_reset() {
    if (stream_) HWANG_HIP_RETURN(hipStreamSynchronize(stream_));
    std::lock_guard<std::mutex> lk(out_mu_);
    outq_.clear();
    buffered_.store(0);
    for (SurfaceSlot& s : surf_) s.held = 0;
    for (Bounce& b : bounce_) b.dst = nullptr;
    return Result();
  }

  bool ensure_geometry(const h264::PicSetup& s, std::string* error) {
    if (s.mb_width == mbw_ && s.mb_height == mbh_) {
      crop_left_ = s.crop_left;
      crop_top_ = s.crop_top;
      return true;
    }
    // new geometry: everything in flight belongs to the old one
    (void)hipStreamSynchronize(stream_);
    free_buffers();
    mbw_ = s.mb_width;
    mbh_ = s.mb_height;
    crop_left_ = s.crop_left;
    crop_top_ = s.crop_top;
    y_pitch_ = ((mbw_ * 16 + 255) / 256) * 256;
    c_pitch_ = y_pitch_ / 2;
    u_offset_ = static_cast<size_t>(y_pitch_) * mbh_ * 16;
    v_offset_ = u_offset_ + static_cast<size_t>(c_pitch_) * mbh_ * 8;
    surface_bytes_ = v_offset_ + static_cast<size_t>(c_pitch_) * mbh_ * 8;
    const size_t nmb = static_cast<size_t>(mbw_) * mbh_;
    // worst case: every one of 27 groups of every macroblock coded
    coef_capacity_ = nmb * 27 * 16;
    sync_words_ = ((2 + mbh_ + 3) / 4) * 4;
    for (Job& j : jobs_) {
      if (hipHostMalloc(reinterpret_cast<void**>(&j.mb_host), nmb * sizeof(MbInfo), hipHostMallocDefault) != hipSuccess ||
          hipHostMalloc(reinterpret_cast<void**>(&j.coef_host), coef_capacity_ * 2, hipHostMallocDefault) != hipSuccess ||
          hipHostMalloc(reinterpret_cast<void**>(&j.slice_host), kMaxSlices * sizeof(SliceParams), hipHostMallocDefault) != hipSuccess ||
          hipHostMalloc(reinterpret_cast<void**>(&j.timeout_host), 2 * sizeof(unsigned), hipHostMallocDefault) != hipSuccess ||
          hipMalloc(reinterpret_cast<void**>(&j.mb_dev), nmb * sizeof(MbInfo)) != hipSuccess ||
          hipMalloc(reinterpret_cast<void**>(&j.motion_dev), nmb * sizeof(MbMotion)) != hipSuccess ||
          hipMalloc(reinterpret_cast<void**>(&j.coef_dev), coef_capacity_ * 2) != hipSuccess ||
          hipMalloc(reinterpret_cast<void**>(&j.slice_dev), kMaxSlices * sizeof(SliceParams)) != hipSuccess ||
          hipMalloc(reinterpret_cast<void**>(&j.sync_dev), 2 * sync_words_ * sizeof(unsigned)) != hipSuccess ||
          hipEventCreateWithFlags(&j.done, hipEventDisableTiming) != hipSuccess) {
        *error = "AMDVideoDecoder: out of memory for staging buffers";
        return false;
      }
      j.timeout_host[0] = j.timeout_host[1] = 0;
      j.in_flight = false;
    }
    job_cursor_ = 0;
    return true;
  }

  bool alloc_surface(int slot, std::string* error) {
    SurfaceSlot& s = surf_[slot];
    if (s.dev) return true;
    const size_t nmb = static_cast<size_t>(mbw_) * mbh_;
    if (hipMalloc(reinterpret_cast<void**>(&s.dev), surface_bytes_) != hipSuccess ||
        hipHostMalloc(reinterpret_cast<void**>(&s.motion_host), nmb * sizeof(MbMotion), hipHostMallocDefault) != hipSuccess ||
        hipEventCreateWithFlags(&s.motion_done, hipEventDisableTiming) != hipSuccess) {
      *error = "AMDVideoDecoder: out of memory for a decode surface";
      return false;
    }
    return true;
  }

  void free_buffers() {
    for (Job& j : jobs_) {
      if (j.mb_host) (void)hipHostFree(j.mb_host);
      if (j.coef_host) (void)hipHostFree(j.coef_host);
      if (j.slice_host) (void)hipHostFree(j.slice_host);
      if (j.timeout_host) (void)hipHostFree(j.timeout_host);
      if (j.mb_dev) (void)hipFree(j.mb_dev);
      if (j.motion_dev) (void)hipFree(j.motion_dev);
      if (j.coef_dev) (void)hipFree(j.coef_dev);
      if (j.slice_dev) (void)hipFree(j.slice_dev);
      if (j.sync_dev) (void)hipFree(j.sync_dev);
      if (j.done) (void)hipEventDestroy(j.done);
      j = Job();
    }
    for (SurfaceSlot& s : surf_) {
      if (s.dev) (void)hipFree(s.dev);
      if (s.motion_host) (void)hipHostFree(s.motion_host);
      if (s.motion_done) (void)hipEventDestroy(s.motion_done);
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
    if (hipSetDevice(device_) != hipSuccess) return;
    if (stream_) (void)hipStreamSynchronize(stream_);
    parser_.reset();
    free_buffers();
    if (stream_) (void)hipStreamDestroy(stream_);
    stream_ = nullptr;
  }

  static bool is_device_pointer(const void* p) {
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, p) != hipSuccess) {
      (void)hipGetLastError();  // unregistered host memory reports an error
      return false;
    }
    return attr.type == hipMemoryTypeDevice;
  }

  Result take_frame(uint8_t* dst, size_t dst_size, bool yuv) {
    HWANG_HIP_RETURN(hipSetDevice(device_));
    const size_t w = info_.width, h = info_.height;
    const size_t cw = (w + 1) / 2, ch = (h + 1) / 2;
    const size_t need = yuv ? w * h + 2 * cw * ch : w * h * 3;
    int slot;
    {
      std::lock_guard<std::mutex> lk(out_mu_);
      if (outq_.empty()) return Result();  // reference behaviour: nothing to do
      if (dst_size < need) return Result(false, "AMDVideoDecoder::get_frame: buffer too small");
      slot = outq_.front();
      if (slot < 0)
        return Result(false,
                      "AMDVideoDecoder::get_frame: this output position was excluded by the sparse hint "
                      "(set_wanted) and was not decoded");
      outq_.pop_front();
      buffered_.store(static_cast<int>(outq_.size()));
    }
    const SurfaceSlot& s = surf_[slot];
    const uint8_t* y = s.dev + static_cast<size_t>(crop_top_) * y_pitch_ + crop_left_;
    const uint8_t* u = s.dev + u_offset_ + static_cast<size_t>(crop_top_ / 2) * c_pitch_ + crop_left_ / 2;
    const uint8_t* v = s.dev + v_offset_ + static_cast<size_t>(crop_top_ / 2) * c_pitch_ + crop_left_ / 2;
    const bool to_device = is_device_pointer(dst);
    Result r;
    uint8_t* target = dst;
    Bounce* bounce = nullptr;
    if (!to_device || !yuv) {
      // RGB to the device writes straight into dst; everything else goes
      // through a bounce buffer (device for the kernel output, pinned host
      // for the copy-out)
      if (!(to_device && !yuv)) {
        r = acquire_bounce(need, &bounce);
        if (!r.ok) return finish_take(slot, r);
        target = bounce->dev;
      }
    }
    hipError_t e = hipSuccess;
    if (yuv) {
      uint8_t* t = to_device ? dst : target;
      e = hipMemcpy2DAsync(t, w, y, y_pitch_, w, h, hipMemcpyDeviceToDevice, stream_);
      if (e == hipSuccess) e = hipMemcpy2DAsync(t + w * h, cw, u, c_pitch_, cw, ch, hipMemcpyDeviceToDevice, stream_);
      if (e == hipSuccess)
        e = hipMemcpy2DAsync(t + w * h + cw * ch, cw, v, c_pitch_, cw, ch, hipMemcpyDeviceToDevice, stream_);
    } else {
      e = launch_yuv420_to_rgb24(y, u, v, y_pitch_, c_pitch_, static_cast<int>(w), static_cast<int>(h), target, stream_);
    }
    if (e == hipSuccess && !to_device) {
      e = hipMemcpyAsync(bounce->host, bounce->dev, need, hipMemcpyDeviceToHost, stream_);
      bounce->dst = dst;
      bounce->size = need;
    }
    if (e != hipSuccess) r = Result(false, std::string("AMDVideoDecoder::get_frame: ") + hipGetErrorString(e));
    return finish_take(slot, r);
  }

  Result finish_take(int slot, const Result& r) {
    std::lock_guard<std::mutex> lk(out_mu_);
    unref_surface_locked(slot, kHeldByOutput);
    return r;
  }

  // A bounce buffer whose previous copy-out has been delivered.
  Result acquire_bounce(size_t need, Bounce** out) {
    Bounce& b = bounce_[bounce_cursor_];
    bounce_cursor_ = (bounce_cursor_ + 1) % kBounce;
    if (b.dst) {
      // the ring wrapped: deliver the oldest pending frame now
      HWANG_HIP_RETURN(hipStreamSynchronize(stream_));
      for (Bounce& p : bounce_)
        if (p.dst) {
          std::memcpy(p.dst, p.host, p.size);
          p.dst = nullptr;
        }
    }
    if (b.capacity < need) {
      if (b.host) (void)hipHostFree(b.host);
      if (b.dev) (void)hipFree(b.dev);
      b.host = b.dev = nullptr;
      HWANG_HIP_RETURN(hipHostMalloc(reinterpret_cast<void**>(&b.host), need, hipHostMallocDefault));
      HWANG_HIP_RETURN(hipMalloc(reinterpret_cast<void**>(&b.dev), need));
      b.capacity = need;
    }
    *out = &b;
    return Result();
  }

  const int device_;
  hipStream_t stream_ = nullptr;
  FrameInfo info_{};
  std::unique_ptr<h264::Parser> parser_;
  std::string sink_error_;
  std::vector<uint64_t> pending_wanted_;

  int mbw_ = 0, mbh_ = 0, y_pitch_ = 0, c_pitch_ = 0, crop_left_ = 0, crop_top_ = 0;
  size_t u_offset_ = 0, v_offset_ = 0, surface_bytes_ = 0, coef_capacity_ = 0;
  int sync_words_ = 0;
  SurfaceSlot surf_[amd::kMaxSurfaces];
  Job jobs_[kJobs];
  int job_cursor_ = 0, cur_job_ = 0;
  Bounce bounce_[kBounce];
  int bounce_cursor_ = 0;

  std::mutex out_mu_;
  std::deque<int> outq_;  // surface slot per frame in display order, -1 = skipped
  std::atomic<int> buffered_{0};

  BackendCounters counters_;
  int64_t decoded_base_ = 0, skipped_base_ = 0;
};

}  // namespace

VideoDecoderInterface* make_amd_video_decoder(int32_t device_id, uint32_t /*num_devices*/) {
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || device_id < 0 || device_id >= count) return nullptr;
  auto* d = new AMDVideoDecoder(device_id);
  std::string error;
  if (!d->open(&error)) {
    delete d;
    return nullptr;
  }
  return d;
}

}  // namespace hwang
