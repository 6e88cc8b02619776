// The host -> GPU seam of the MI355X backend: what the host-side H.264
// parser (entropy decode, motion-vector derivation, DPB) hands to the HIP
// reconstruct kernels for one picture.  Everything serial and
// context-dependent has been resolved on the host; what is left is
// sample-parallel.
//
// Layout in HBM per picture (one "job", uploaded with one H2D copy per
// array from a pinned staging slot):
//   MbInfo   mb[num_mbs]        32 B each   macroblock header
//   MbMotion motion[num_mbs]    144 B each  only for P/B pictures
//   int16_t  coefs[...]         dense 16-coefficient groups, only for coded
//                               transform blocks, addressed by
//                               MbInfo::coef_offset + popcount(coded_mask)
//   SliceParams slices[n]       weights / POCs / deblock offsets per slice
// Surfaces (decoded pictures) are planar I420 of the CODED size, luma pitch a
// multiple of 256 B so every macroblock row starts line-aligned.
#pragma once

#include <cstdint>

namespace hwang {
namespace amd {

enum MbKind : uint8_t {
  MBK_I4x4 = 0,
  MBK_I8x8 = 1,
  MBK_I16x16 = 2,
  MBK_IPCM = 3,
  MBK_INTER = 4,
};

// MbInfo::flags
enum : uint8_t {
  MBF_T8X8 = 1,        // transform_size_8x8_flag
  MBF_INTRA = 2,
  // intra-prediction availability of the neighbouring macroblocks (slice
  // membership and constrained_intra_pred already applied)
  MBF_AVAIL_A = 4,     // left
  MBF_AVAIL_B = 8,     // top
  MBF_AVAIL_C = 16,    // top-right
  MBF_AVAIL_D = 32,    // top-left
  // deblocking: which of this macroblock's outer edges are filtered
  MBF_DB_LEFT = 64,
  MBF_DB_TOP = 128,
};

// coded_mask bits: which 16-coefficient groups are present at coef_offset,
// in this order.
//   0..15  luma 4x4 block at raster position k (y*4+x); with MBF_T8X8 the
//          8x8 block b8 sets bits 4*b8 .. 4*b8+3 and its 64 coefficients
//          (raster 8x8) occupy those four groups
//   16     Intra16x16 luma DC (16 values, raster = block position)
//   17     chroma DC (Cb[4], Cr[4], 8 zeros)
//   18..21 Cb AC blocks 0..3, 22..25 Cr AC blocks 0..3 (DC slot zero)
// An I_PCM macroblock stores its 384 samples as 96 words (four bytes each,
// little endian) and sets coded_mask = 0.
constexpr int kMaskLumaDc = 16;
constexpr int kMaskChromaDc = 17;
constexpr int kMaskChromaAc = 18;

struct MbInfo {
  uint8_t kind;
  uint8_t flags;
  uint8_t qp;          // QP_Y (0 for I_PCM, as the deblocking filter wants)
  uint8_t qpc[2];      // QP_C for Cb, Cr
  uint8_t pred_modes;  // low nibble: Intra16x16 mode, high nibble: chroma mode
  uint8_t slice;       // index into SliceParams
  uint8_t deblock_internal;  // 1: internal edges filtered (idc != 1)
  uint32_t coded_mask;
  uint32_t coef_offset;      // in 32-bit words
  uint8_t i4_modes[8];       // 16 x 4 bit, raster 4x4 order
  uint16_t nz_mask;          // 4x4 luma blocks (raster) holding non-zero
                             // coefficients, as the bS derivation sees them
  // Motion-vector derivation left to k_mv (P macroblocks): MbMotion::mv[0]
  // then holds the DIFFERENCES as parsed, one per partition, replicated over
  // the partition's 4x4 blocks.
  uint8_t mv_flags;          // MVF_* below
  uint8_t mv_shape;          // 0 16x16, 1 16x8, 2 8x16, 3 8x8
  uint8_t mv_sub;            // 8x8 only: 2 bits per quadrant: 0 8x8, 1 8x4, 2 4x8, 3 4x4
  // Coefficients at coef_offset: `ncoef` sparse entries, one word per NON-ZERO level:
  //   (group * 16 + position) << 16 | level (int16)   - group / position as in coded_mask above
  // (a coded 4x4 block of a P picture holds one or two non-zero levels on average: 4-8 bytes instead of
  // the 32 of a dense group).  kDenseCoefs: more than kMaxSparseCoefs levels - the groups follow densely,
  // 16 int16 per set bit of coded_mask in bit order, which bounds a macroblock at 864 bytes either way.
  uint8_t pad;
  uint16_t ncoef;
};
constexpr uint16_t kDenseCoefs = 0xffff;
constexpr int kMaxSparseCoefs = 216;
enum : uint8_t {
  MVF_DEFERRED = 1,          // mv[0] holds differences; derive (8.4.1.3)
  MVF_PSKIP = 2,             // P_Skip inference (8.4.1.1)
  MVF_COMPACT = 4,           // the macroblock's motion travels as PMotion (+ extra differences), not as MbMotion
  // neighbouring macroblocks usable for mv prediction (same slice, inside the picture)
  MVF_AVAIL_A = 16,
  MVF_AVAIL_B = 32,
  MVF_AVAIL_C = 64,
  MVF_AVAIL_D = 128,
};
static_assert(sizeof(MbInfo) == 32, "MbInfo must stay 32 bytes");

struct MbMotion {
  int16_t mv[2][16][2];   // [list][raster 4x4][x, y] quarter samples
  int8_t ref_slot[2][4];  // DPB surface per 8x8 as a byte (0 .. kMaxSurfaces - 1), -1 / 0xff = list unused
  int8_t ref_idx[2][4];   // index into the slice's list (for weights)
};
static_assert(sizeof(MbMotion) == 144, "MbMotion must stay 144 bytes");

// Compact motion of a macroblock in a P (or I) slice whose vectors k_mv derives:
// what the entropy decoder knows, nothing replicated.  P_Skip and 16x16 - 88 %
// of the inter macroblocks of the bench corpus - are complete with this record;
// other shapes keep the differences of their 2nd .. 16th partition (decode
// order) in the picture's extra stream, whose words the macroblocks of a row
// consume back to back from row_start[mby] on.  k_mv expands every such
// macroblock to its full MbMotion in HBM (k_inter and k_bs read that), so the
// 144-byte records never cross PCIe: 12 B instead of 144 B per macroblock.
struct PMotion {
  int8_t ref_idx[4];   // list 0, per 8x8; -1 = intra macroblock
  int8_t ref_slot[4];  // DPB surface of that reference
  int16_t mvd0[2];     // difference of the first (or only) partition; 0 for P_Skip / intra
};
static_assert(sizeof(PMotion) == 12, "PMotion must stay 12 bytes");

struct SliceParams {
  int8_t alpha_offset, beta_offset;  // already multiplied by 2
  uint8_t weight_mode;               // 0 default, 1 explicit, 2 implicit
  uint8_t luma_log2_denom, chroma_log2_denom;
  uint8_t pad[3];
  int32_t cur_poc;
  int16_t luma_weight[2][32], luma_offset[2][32];
  int16_t chroma_weight[2][32][2], chroma_offset[2][32][2];
  int32_t ref_poc[2][32];
  uint8_t ref_long[2][32];
};

struct Surface {
  uint8_t* y;
  uint8_t* u;
  uint8_t* v;
};

// (MbMotion::ref_slot is a byte, 0xff = "list not used": surfaces 0 .. 254.  A serial decoder uses a dozen; one with
// entropy workers keeps the pictures of many GOPs in flight between the host and the GPU, each in a surface of
// its own - its parser thread is through with a GOP's headers before the first of its pictures has been decoded.)
constexpr int kMaxSurfaces = 250;

struct PictureParams {
  int32_t mb_width, mb_height;
  int32_t y_pitch, c_pitch;
  int32_t cur_slot;
  int32_t num_slices;
  uint8_t scaling4[6][16];  // weightScale, raster order
  uint8_t scaling8[2][64];
  Surface surfaces[kMaxSurfaces];
};

// One picture's worth of work as the batched kernels see it.  The executor
// uploads an array of these per batch; workgroup -> job is blockIdx.y (k_inter)
// or blockIdx.x (row pipelines, k_bs uses blockIdx.y).
// One host -> HBM transfer of k_upload: `src` is pinned host memory mapped
// into the GPU's address space, read over PCIe by the kernel itself.
struct UploadDesc {
  void* dst;
  const void* src;
  uint64_t bytes;
};
// MbInfo, coefficients, SliceParams, then MbMotion (slices whose vectors the host derived) and / or PMotion + extra
// differences (slices left to k_mv): a picture with B and P slices brings all six
constexpr int kJobUploads = 6;

// Width (macroblocks) that the LDS line buffers of k_mv and of k_intra's
// LDS-neighbour path cover: 4K (240) fits; wider pictures take the host path
// for motion vectors and the surface path for intra neighbours.
constexpr int kLineMbs = 256;
// Largest picture side in macroblocks (8192 samples): sizes the row pipelines'
// band tables (k_recon.hip: kMaxRows) - the decoder refuses anything larger.
constexpr int kMaxMbRows = 512;

struct alignas(16) JobDesc {
  PictureParams pp;
  const MbInfo* mbs;
  const MbMotion* motions;
  const uint32_t* coefs;   // sparse level entries / dense groups / I_PCM samples, see MbInfo::ncoef
  const SliceParams* slices;
  void* db;            // k_bs -> k_deblock records
  int16_t* resid;      // k_residual -> k_intra: 384 int16 per macroblock (256 luma raster, 64 Cb, 64 Cr)
  int32_t do_inter, do_intra, do_deblock;
  int32_t do_mv;       // some macroblocks carry MVF_DEFERRED
  int32_t num_uploads;
  uint32_t ticket_mv;  // jobs[0] only: band ticket counters of k_mv / k_deblock / k_intra (zeroed by the host with every table)
  void* band_ctl;      // hand-off between the workgroups of one picture (amd::band_ctl_bytes)
  uint32_t epoch;      // tags this picture's words in band_ctl (nothing is cleared between pictures); never 0
  uint32_t ticket_db;
  uint32_t ticket_in;
  uint32_t inject_fault;  // test hook (HWANG_INJECT_FAULT): k_deblock's exporting rows withhold their progress word
  // Pinned host word of the batch this picture travels in.  A row kernel whose
  // bounded poll of another band expires (a broken hand-off: impossible by
  // construction, but round 1 shipped one) stores a FAULT_* code here; the
  // executor reads it when the batch completes and fails every waiter instead
  // of returning wrong samples.
  uint32_t* fault;
  // compact motion (macroblocks flagged MVF_COMPACT): records, and the extra stream
  // = row_start[mb_height] followed by the difference words
  const PMotion* pmo;
  const uint32_t* pmx;
  // Chain launches: the launch's control words (ChainArgs::tickets; k_recon.hip: kChainFaultWord, kChainBeatWord);
  // nullptr in every other launch
  const uint32_t* chain_ctl;
  UploadDesc uploads[kJobUploads];
  // Chain launches (k_chain): the job - index into the launch's table - that reconstructs surface s in this same
  // launch, -1 if the surface was final before the launch.  A picture that predicts from such a surface follows the
  // progress its producer publishes (BandCtl::fin) instead of waiting for a kernel boundary.
  int16_t slot_job[256];
};
static_assert(kMaxSurfaces <= 256, "JobDesc::slot_job");

enum : uint32_t {
  FAULT_MV_IMPORT = 1,       // k_mv: line of the band above never arrived
  FAULT_INTRA_LINE = 2,      // k_intra (LDS neighbours): likewise
  FAULT_INTRA_PROGRESS = 3,  // k_intra (surface neighbours): progress word of the band above
  FAULT_DEBLOCK_PROGRESS = 4,
  FAULT_CHAIN_REFERENCE = 5,  // k_chain: a reference picture of the same launch never got as far as a macroblock needs
  FAULT_CHAIN_INTER = 6,      // k_chain: a row's inter macroblocks never all arrived
  FAULT_CHAIN_INTRA = 7,      // k_chain: a row's intra prediction never finished
};

// A chain launch (k_chain): every stage of every picture of the batch as ONE launch of persistent workgroups that take
// units from two lists by ticket - row units (a band of a picture's intra prediction or of its deblocking) and inter
// units (a run of macroblocks) - both lists in an order in which a unit only ever depends on units before it (in
// either list: gpu_executor.cpp builds them position by position of the lanes' chains).
struct ChainArgs {
  const JobDesc* jobs;
  const uint32_t* row_units;    // job << 8 | stage << 7 | band     (stage 0 intra, 1 deblock)
  const uint32_t* inter_units;  // job << 20 | run                  (run = row * ceil(mb_width / kChainRun) + j: macroblocks 8 j .. of the row)
  uint32_t* tickets;            // [0] rows, [1] inter, [2] the launch's first fault, [3] its heartbeat; zeroed with the table
  uint32_t num_row_units, num_inter_units;
  uint64_t bytes;               // algorithmic, for the span
};
constexpr int kChainRun = 8;    // macroblocks per inter unit (64 runs cover the widest row: BandCtl::inter_mask)
static_assert(kMaxMbRows <= 64 * kChainRun, "BandCtl::inter_mask is one 64-bit word per row");

// One colour-conversion / plane-copy request.
struct ConvertDesc {
  const uint8_t* y;
  const uint8_t* u;
  const uint8_t* v;
  uint8_t* dst;        // device memory: RGB24 (w*h*3) or I420 (w*h*3/2)
  int32_t y_pitch, c_pitch, width, height;
  int32_t yuv;         // 1: copy cropped planes instead of converting
  int32_t pad[3];
};

// tables of these travel to the device in 16-byte chunks (launch_table_copy)
static_assert(sizeof(JobDesc) % 16 == 0, "JobDesc must be a multiple of 16 bytes");
static_assert(sizeof(ConvertDesc) % 16 == 0, "ConvertDesc must be a multiple of 16 bytes");

}  // namespace amd
}  // namespace hwang
