// Host side of the HEVC path (the reference hands HEVC tracks to libavcodec:
// impls/software/software_video_decoder.cpp:128-132, 349-457): from the samples of an `hev1` / `hvc1` track to the
// per-picture records of amd/hevc_frame_job.h.  Parameter sets and slice segment headers (H.265 7.3.2, 7.3.6),
// picture order count, reference picture sets and lists (8.3), output order (C.5.2), CABAC (9.3) and the slice data
// syntax (7.3.8); and everything that is serial or context-dependent between syntax and samples: intra mode
// derivation (8.4.2), merge / AMVP / temporal motion candidates (8.5.3.2), the QP predictor (8.6.1), neighbour
// availability for intra prediction (6.4.1, constrained_intra_pred), boundary strengths (8.7.2.4).  Main profile
// (8 bit, 4:2:0); no tiles, no dependent slice segments.  Host only, no GPU calls; written independently of
// oracle/hevc_*.c (which is test infrastructure and never linked into the product).
#pragma once

#include "../amd/hevc_frame_job.h"

#include <cstddef>
#include <cstdint>
#include <memory>
#include <string>
#include <vector>

namespace hwang {
namespace hevc {

struct PicSetup {
  int width = 0, height = 0;  // coded size
  int crop_left = 0, crop_right = 0, crop_top = 0, crop_bottom = 0;
  int log2_ctb = 0;
};

// Where the parser writes a picture's records (the backend's pinned staging memory) and how much fits.
struct PicBuffers {
  int slot = -1;            // surface the picture is reconstructed into
  void* cookie = nullptr;
  amd::hevc::ResTu* res = nullptr;
  size_t cap_res = 0;
  uint32_t* coefs = nullptr;
  size_t cap_coefs = 0;
  amd::hevc::InterPu* pus = nullptr;
  size_t cap_pus = 0;
  amd::hevc::WeightEntry* weights = nullptr;
  size_t cap_weights = 0;
  amd::hevc::IntraTu* intra[3] = {nullptr, nullptr, nullptr};
  size_t cap_intra[3] = {0, 0, 0};
  uint32_t* intra_ctb_first[3] = {nullptr, nullptr, nullptr};  // per plane: coding tree blocks + 1 offsets (filled at the end of the picture)
  amd::hevc::EdgeInfo* edges = nullptr;                        // w4 * h4
  amd::hevc::SaoCtb* sao = nullptr;                            // ctbs
  amd::hevc::SliceLoop* slices = nullptr;                      // kMaxSlices
  uint8_t* scaling = nullptr;                                  // 4064 bytes
};

struct PicDone {
  int slot = -1;
  void* cookie = nullptr;
  size_t num_res = 0, num_coefs = 0, num_pus = 0, num_weights = 0, num_intra[3] = {0, 0, 0};
  int num_slices = 0;
  bool sao = false;       // some slice has SAO on: the reconstruction goes through the scratch surface
  bool deblock = false;   // some edge has a strength
  bool scaling = false;   // scaling lists in use (the table is in PicBuffers::scaling)
  int cb_qp_offset = 0, cr_qp_offset = 0, pcm_bits_luma = 8, pcm_bits_chroma = 8;
  int ref_slots[amd::hevc::kMaxSurfaces];
  int num_ref_slots = 0;  // surfaces the picture predicts from
};

class PictureSink {
 public:
  virtual ~PictureSink() {}
  virtual bool begin_picture(const PicSetup& setup, PicBuffers* out, std::string* error) = 0;
  virtual bool end_picture(const PicDone& done, std::string* error) = 0;
  // display order; slot < 0: an output position whose picture was skipped on purpose (sparse hint)
  virtual void output_picture(int slot) = 0;
  virtual void release_surface(int slot) = 0;
  // 8.3.3.2: a surface of the current picture's size whose samples are all 128, for a reference picture that was never
  // decoded (a stream entered at a CRA picture: its leading pictures predict from the GOP before).  Asked for when a
  // prediction block first refers to such a picture, released through release_surface like any other.  < 0: none.
  virtual int missing_surface(std::string* error) {
    *error = "this sink has no surfaces for made-up reference pictures";
    return -1;
  }
  virtual void abandon_picture(const PicDone& /*done*/) {}
};

class Parser {
 public:
  explicit Parser(PictureSink* sink);
  ~Parser();
  bool configure(const uint8_t* hvcc, size_t size);
  bool decode(const uint8_t* sample, size_t size);
  bool flush();
  // display-order positions (from the start of the interval) that will be fetched: unwanted pictures that nothing
  // refers to (sub-layer non-reference pictures of the highest temporal layer) are not decoded
  void set_wanted(const std::vector<uint64_t>& ordinals);
  // ... and, optionally, which samples of the interval (decoding order, same origin) start a closed GOP: what lies
  // between a GOP's last wanted position (+ the reorder depth) and the next of these is not decoded at all, and neither
  // is anything behind the interval's last wanted position.  Every such sample is still one (void) output position.
  void set_keyframes(const std::vector<uint64_t>& ordinals);
  int width() const;   // cropped
  int height() const;
  const std::string& error() const;
  int64_t pictures_decoded() const;
  int64_t pictures_skipped() const;
  std::string describe() const;

 private:
  struct Impl;
  std::unique_ptr<Impl> impl_;
};

}  // namespace hevc
}  // namespace hwang
