// Host side of the MI355X H.264 backend: everything that is serial per slice
// or context dependent.  NAL / parameter-set / slice-header parsing, CAVLC and
// CABAC entropy decoding, intra-mode and motion-vector derivation (including
// spatial / temporal direct), reference lists, marking, POC and display-order
// output.  It reconstructs no samples: per picture it fills the frame_job.h
// records and hands them to a PictureSink, which runs the HIP kernels.
//
// This is what the reference obtains from libavcodec's h264 decoder up to,
// but not including, hl_decode_mb (software_video_decoder.cpp:353-357 ->
// avcodec_send_packet).  Supported: Baseline/Main/High 4:2:0 8-bit, frame
// (progressive) coding, CAVLC + CABAC, I/P/B, multiple references and slices,
// 8x8 transform, scaling matrices, explicit/implicit weighted prediction.
// Rejected with an error (never silently mis-decoded): interlaced / MBAFF,
// FMO/ASO, SP/SI, 4:2:2 / 4:4:4, > 8 bit.
#pragma once

#include "../amd/frame_job.h"

#include <cstddef>
#include <cstdint>
#include <functional>
#include <string>
#include <vector>

namespace hwang {
namespace h264 {

struct PicBuffers {
  amd::MbInfo* mb = nullptr;         // [mb_width * mb_height]
  amd::MbMotion* motion = nullptr;   // [mb_width * mb_height], lives with the surface
  uint32_t* coefs = nullptr;         // MbInfo::coef_offset / ncoef
  size_t coef_capacity = 0;          // in 32-bit words, >= 216 per macroblock
  amd::SliceParams* slices = nullptr;
  int max_slices = 0;
  int slot = -1;                     // surface the picture is reconstructed into
  // The sink derives the motion vectors of P macroblocks itself (k_mv) from
  // the differences; if false the parser does it before end_picture.
  bool device_mv = false;
  // With device_mv: macroblocks of P and I slices are handed over as PMotion
  // records plus the extra stream (frame_job.h) instead of MbMotion.  Both live
  // with the surface, like `motion` (a later B picture may need them for its
  // co-located lookups).  pmx[0 .. mb_height) is the row table.
  amd::PMotion* pmo = nullptr;       // [mb_width * mb_height]
  uint32_t* pmx = nullptr;
  size_t pmx_capacity = 0;           // in words, >= mb_height + 15 * macroblocks
  // Handed back in PicDone::cookie (with entropy workers several pictures are
  // between begin_picture and end_picture at a time: which staging buffers).
  int cookie = 0;
  // begin_picture failed because every surface is taken: with pictures still
  // at the workers the parser waits for one to finish and asks again.
  bool no_surface = false;
};

struct PicSetup {
  int mb_width, mb_height;
  int crop_left, crop_right, crop_top, crop_bottom;  // luma samples
};

struct PicDone {
  int slot;
  int cookie;            // PicBuffers::cookie of this picture's begin_picture
  int num_slices;
  size_t coef_count;     // 32-bit words used
  bool has_inter;        // motion[] must be uploaded
  bool has_intra;
  int num_intra_mbs;     // macroblocks that go through k_residual / k_intra
  bool any_deblock;      // some slice has the in-loop filter enabled
  bool mv_deferred;      // MbInfo::mv_flags & MVF_DEFERRED on some macroblocks: run k_mv first
  bool compact_motion;   // some macroblocks are MVF_COMPACT: pmo[] and pmx[0 .. pmx_count) must be uploaded
  bool full_motion;      // some inter macroblocks are not: motion[] must be uploaded
  size_t pmx_count;      // words of pmx in use (row table included)
  int slice_type_mask;   // bit0 P, bit1 B, bit2 I
  const uint8_t (*scaling4)[16];
  const uint8_t (*scaling8)[64];
  int ref_slots[32];     // surfaces read by this picture (for lifetime tracking)
  int num_ref_slots;
};

// Threads: with a serial Parser every call comes from the thread that calls
// Parser::decode / flush, and a picture's output_picture follows its
// end_picture.  With entropy workers begin_picture and output_picture (display
// order) still come from that thread, but output_picture may come BEFORE the
// picture's end_picture: the DPB releases a picture on the strength of its
// headers while its slice data is still with a worker.  end_picture - or, for a
// picture that failed or follows one that did, abandon_picture - comes from a
// worker thread, one call at a time, in decoding order; release_surface from
// any thread at any time.
class PictureSink {
 public:
  virtual ~PictureSink() {}
  // A new picture starts; provide a free surface and record buffers.
  virtual bool begin_picture(const PicSetup& setup, PicBuffers* out, std::string* error) = 0;
  // All macroblocks of the picture are in the buffers: reconstruct it.
  virtual bool end_picture(const PicDone& done, std::string* error) = 0;
  // Display-order emission.  slot < 0: a picture that was skipped on purpose
  // (unwanted non-reference picture); it still occupies one output position.
  virtual void output_picture(int slot) = 0;
  // The parser no longer needs the surface as a reference nor for output.
  virtual void release_surface(int slot) = 0;
  // The picture begun with done.slot / done.cookie (nothing else of `done` is
  // meaningful) will not get an end_picture: its slice data, or with entropy
  // workers an earlier picture's, could not be decoded.  Whatever was asked of
  // its output position cannot be served; its record buffers are free again.
  virtual void abandon_picture(const PicDone& /*done*/) {}
};

// Threads that decode slice data for one or more Parsers (the reference's
// software backend hands `num_devices` to libavcodec as its thread count,
// software_video_decoder.cpp:148; this is that knob).  Jobs start in the order
// they were submitted.
class WorkerPool {
 public:
  explicit WorkerPool(int threads);
  ~WorkerPool();
  WorkerPool(const WorkerPool&) = delete;
  WorkerPool& operator=(const WorkerPool&) = delete;
  int size() const;
  void submit(std::function<void()> job);

 private:
  struct State;
  State* state_;
};

// Development aid (HWANG_EXEC_DEBUG): process-wide nanoseconds the parser threads waited for room in the window
// [0] / for a surface [1], the workers spent on slice data [2] and inside the in-order hand-over [3]; [4] pictures.
void pipe_debug_stats(int64_t out[6]);

class Parser {
 public:
  // Serial: slice data is decoded inside decode().  With `workers` the calling
  // thread only parses headers and runs the DPB; the slice data of up to
  // `max_in_flight` (default: two per worker) consecutive pictures is decoded by
  // the pool at the same time, pictures reach the sink in decoding order, and an
  // error in a picture's slice data is reported by a later decode() / flush().
  // The pool must outlive the Parser.
  explicit Parser(PictureSink* sink, WorkerPool* workers = nullptr, int max_in_flight = 0);
  ~Parser();

  // avcC payload.
  bool configure(const uint8_t* avcc, size_t size);
  // One MP4 sample = one access unit (AVCC framing).
  bool decode(const uint8_t* sample, size_t size);
  // End of interval: emit everything pending, drop all references.  With
  // entropy workers this does not wait for them.
  bool flush();
  // Entropy workers: blocks until every picture handed to them has reached the
  // sink (end_picture or abandon_picture); false + message if one failed and the
  // caller has not been told yet.  May be called from another thread than
  // decode() / flush() as long as those are not running.  Serial: true at once.
  bool wait_idle(std::string* error);

  // Sparse decoding: output positions, counted from the first
  // picture after the last flush/configure, that will be fetched.  Unwanted
  // non-reference pictures are not entropy decoded at all.  Empty = all.
  void set_wanted(const std::vector<uint64_t>& ordinals);
  // With set_wanted: which samples of the interval (0-based, decoding order) are sync samples, i.e. start closed
  // GOPs.  Lets an interval of several GOPs end each GOP behind its last wanted position instead of only the last.
  void set_keyframes(const std::vector<uint64_t>& ordinals);

  const std::string& error() const { return error_; }
  int width() const;    // cropped
  int height() const;
  int64_t pictures_decoded() const { return pictures_decoded_; }
  int64_t pictures_skipped() const { return pictures_skipped_; }
  // Slices parsed / slices whose data ended exactly on rbsp_trailing_bits
  // (an entropy-decoder desync check that needs no second decoder).
  int64_t slices() const { return slices_; }
  int64_t slices_clean_end() const { return slices_clean_end_; }

  struct Impl;

 private:
  Impl* impl_;
  std::string error_;
  int64_t pictures_decoded_ = 0;
  int64_t pictures_skipped_ = 0;
  int64_t slices_ = 0, slices_clean_end_ = 0;
};

}  // namespace h264
}  // namespace hwang
