// The host -> GPU seam of the HEVC path: what the host-side parser
// (hevc/hevc_parser.cpp: CABAC, merge / AMVP / temporal candidates, intra mode and QP derivation, boundary
// strengths) hands to the HIP kernels of k_hevc.hip for one picture.  As on the H.264 path everything serial and
// context-dependent is resolved on the host; what crosses PCIe is sample-parallel work, already in the units the
// kernels are launched over:
//   ResTu      one transform block with levels: k_hevc_residual (scaling 8.6.3, transforms 8.6.4) -> residual planes
//   InterPu    one prediction block: k_hevc_inter (8.5.3.3 interpolation + weighting, + residual) -> reconstruction
//   IntraTu    one intra-predicted transform block per colour plane, in decoding order (8.4.4.2): k_hevc_intra_rows
//              walks them as a wavefront of coding tree block rows (intra pictures), k_hevc_intra_flow takes them by
//              ticket and lets each wait for exactly the neighbouring blocks it reads (scattered intra blocks)
//   edge maps  per 4x4 luma block: which of its sides are block edges that may be filtered, levels / intra, QpY, "leave alone" flag:
//              k_hevc_deblock (8.7.2), k_hevc_sao (8.7.3, with SaoCtb per coding tree block)
// Surfaces are planar I420 of the coded size; luma pitch a multiple of 256 bytes, chroma pitch half of it.
#pragma once

#include <cstdint>

namespace hwang {
namespace amd {
namespace hevc {

constexpr int kMaxSurfaces = 32;   // DPB (16 + current) and pictures waiting to be fetched
constexpr int kMaxSlices = 255;

struct ResTu {
  uint16_t x, y;        // position in samples of its plane
  uint8_t log2;         // 2 .. 5
  uint8_t c_idx;
  uint8_t qp;           // qP of 8.6.2 for this plane (chroma: after the table)
  uint8_t flags;        // RES_*
  uint8_t matrix;       // scaling list: 0xff = flat, else matrixId (sizeId follows from log2)
  uint8_t pad[3];
  uint32_t coef_offset; // first entry in the picture's level stream
  uint32_t ncoef;       // entries: position << 16 | (level & 0xffff), position = y * n + x
};
static_assert(sizeof(ResTu) == 20, "ResTu");
enum : uint8_t { RES_DST = 1, RES_TSKIP = 2, RES_BYPASS = 4 };

struct InterPu {
  uint16_t x, y;        // luma samples
  uint8_t w, h;         // luma samples (4 .. 64)
  uint8_t pred_flag;    // bit 0 list 0, bit 1 list 1
  uint8_t weighted;     // explicit weights follow in `wp`
  uint8_t ref_slot[2];  // surface of the reference picture per list
  int16_t mv[2][2];     // quarter luma samples
  uint16_t wp[2];       // index into the picture's weight table per list (8.5.3.3.4.3)
  uint16_t pad;
};
static_assert(sizeof(InterPu) == 24, "InterPu");

struct WeightEntry {     // one reference picture of one list of one slice
  int16_t w[3], o[3];    // luma, Cb, Cr
  uint8_t log2wd[2];     // luma / chroma denominators (before the + 6 of 14-bit samples)
  uint8_t pad[2];
};

struct IntraTu {
  uint16_t x, y;        // position in samples of its plane
  uint8_t log2;         // 2 .. 5
  uint8_t mode;         // 0 .. 34, INTRA_PCM: the residual plane holds the samples
  uint8_t flags;        // INTRA_*
  uint8_t pad;
  uint32_t avail_left;  // bit k: unit k of the left column exists (units of 4 luma samples, from the top, 2n samples)
  uint32_t avail_top;   // bit k: unit k of the row above (from the left, 2n samples)
  uint32_t wait_left;   // ... and is written by an intra block of this picture (an earlier record of the same list):
  uint32_t wait_top;    //     k_hevc_intra waits for its "done" mark; everything else is final before the kernel starts
};
static_assert(sizeof(IntraTu) == 24, "IntraTu");
enum : uint8_t { INTRA_CORNER = 1, INTRA_STRONG = 2 /* strong_intra_smoothing_enabled_flag */, INTRA_WAIT_CORNER = 4 };
constexpr uint8_t kIntraPcm = 0xff;

struct SaoCtb {
  uint8_t type[3], band_pos[3], eo_class[3];
  int8_t offset[3][4];
  uint8_t pad[3];
};
static_assert(sizeof(SaoCtb) == 24, "SaoCtb");

struct SliceLoop {       // per slice: what the in-loop filters need
  int8_t beta_offset, tc_offset;  // already doubled
  uint8_t across;                 // slice_loop_filter_across_slices_enabled_flag
  uint8_t sao_luma, sao_chroma;
  uint8_t pad[3];
  uint32_t first_ctb;             // SliceAddrRs
};

// per 4x4 luma block (EdgeInfo[w4 * h4]): what the in-loop filters need to know about the block.  The boundary
// strengths (8.7.2.4) are derived by k_hevc_deblock from the two blocks either side of an 8x8-grid edge: 2 if either is
// intra, 1 at a transform block edge with levels on either side, else by the motion of the two prediction blocks.
struct EdgeInfo {
  uint8_t edge;    // EDGE_*
  int8_t qp;       // QpY
  uint8_t flags;   // bit 0: pcm with the loop filter off / transquant bypass: the in-loop filters leave the samples alone
  uint8_t slice;   // index into SliceLoop
};
enum : uint8_t {
  EDGE_TU_LEFT = 1, EDGE_PU_LEFT = 2,   // the block's left side is a transform / prediction block edge
  EDGE_TU_TOP = 4, EDGE_PU_TOP = 8,
  EDGE_ON_LEFT = 16, EDGE_ON_TOP = 32,  // ... and may be filtered (deblocking on in the slice; not a picture edge, not a slice edge the flags exclude)
  EDGE_LEVELS = 64,                     // the block's transform block has levels
  EDGE_INTRA = 128
};

struct Surface {
  uint8_t* y;
  uint8_t* u;
  uint8_t* v;
};

struct PicParams {
  int32_t width, height;       // coded size (multiples of 8)
  int32_t y_pitch, c_pitch;
  int32_t log2_ctb, width_ctb, height_ctb;
  int32_t cb_qp_offset, cr_qp_offset;  // pps: for the chroma deblocking (cQpPicOffset)
  int32_t pcm_bits_luma, pcm_bits_chroma;
  int32_t num_res, num_pu, num_intra[3];
  int32_t sao_enabled, deblock_enabled;
  Surface recon;     // where prediction + residual go and deblocking happens: the scratch surface when SAO follows, else dst
  Surface dst;       // the picture's DPB surface
  Surface refs[kMaxSurfaces];
  // device pointers of the picture's records
  const ResTu* res;
  const uint32_t* coefs;
  const InterPu* pus;
  const WeightEntry* weights;
  const IntraTu* intra[3];             // per plane, decoding order
  const uint32_t* intra_ctb_first[3];  // per plane: width_ctb * height_ctb + 1 offsets into intra[c] (k_hevc_intra_rows)
  int32_t intra_rows;                  // 1: the picture is mostly intra, k_hevc_intra_rows; 0: k_hevc_intra_flow
  int32_t clear_words;                 // 32-bit words from `intra_ticket` on that the picture's intra kernel expects zeroed (k_hevc_clear)
  const EdgeInfo* edges;
  const SaoCtb* sao;
  const SliceLoop* slices;
  const uint8_t* scaling;      // ScalingFactor tables: f4[6][16] f8[6][64] f16[6][256] f32[2][1024], or null
  int16_t* resid[3];           // residual planes, pitches as the surfaces'
  uint32_t* progress;          // k_hevc_intra_rows: [plane][block row] -> blocks completed (zeroed per picture)
  uint32_t* intra_ticket;      // k_hevc_intra_flow: [3] next record of each plane's list (zeroed per picture)
  uint8_t* intra_done[3];      // per plane, per 4x4 luma block: an intra block covering it is final (zeroed per picture)
  uint32_t* pu_map;            // per 4x4 luma block: which InterPu covers it (written by k_hevc_inter, read by k_hevc_deblock)
  uint32_t* fault;             // set by a kernel whose bounded wait expired
};

// What one launch works on: pictures of different decoders (nothing in common but the GPU), each with its own grid
// coordinate; kernels whose grid is sized by the largest picture let the blocks past a smaller one's end return.
constexpr int kMaxBatch = 32;
struct Batch {
  const PicParams* pic[kMaxBatch];  // device pointers
  int32_t spin_sleep = 1;           // how long a wavefront that waits for a neighbour sleeps between two looks (x 64 clocks x 8)
  int32_t pad = 0;
};

}  // namespace hevc
}  // namespace amd
}  // namespace hwang
