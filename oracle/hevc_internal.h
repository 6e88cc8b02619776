/* TEST INFRASTRUCTURE - internal structures of the HEVC oracle (see hevc_oracle.h for what it is and is not).
 * Shared by oracle/hevc_headers.c (parameter sets, slice headers, reference picture sets, DPB, the public entry
 * points), oracle/hevc_slice.c (CABAC engine - decoder AND encoder - and the slice data syntax 7.3.8, written once and
 * walked in either direction), oracle/hevc_recon.c (8.4 intra, 8.5 inter, 8.6 scaling + transforms, 8.7 in-loop
 * filters) and corpus/hevc_encoder.c (which reconstructs through the same functions, as corpus/h264_encoder.c does
 * with the H.264 oracle). */
#ifndef ORACLE_HEVC_INTERNAL_H_
#define ORACLE_HEVC_INTERNAL_H_

#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "hevc_oracle.h"
#include "hevc_tables.h"

#define HV_MAX_DPB 17
#define HV_MAX_REFS 16
#define HV_MAX_SLICES 600

static inline int hv_clip3(int lo, int hi, int v) { return v < lo ? lo : v > hi ? hi : v; }
static inline int hv_clip8(int v) { return v < 0 ? 0 : v > 255 ? 255 : v; }
static inline int hv_min(int a, int b) { return a < b ? a : b; }
static inline int hv_max(int a, int b) { return a > b ? a : b; }
static inline int hv_abs(int a) { return a < 0 ? -a : a; }
static inline int hv_sign(int a) { return a > 0 ? 1 : a < 0 ? -1 : 0; }
static inline int hv_ceil_log2(unsigned v) {
  int n = 0;
  while ((1u << n) < v) ++n;
  return n;
}

/* ---- bit reader over an RBSP (emulation prevention already removed) ---- */
typedef struct {
  const uint8_t* p;
  size_t size; /* bytes */
  size_t pos;  /* bits */
  int overrun;
} HvBits;
static inline unsigned hvb_u(HvBits* b, int n) {
  unsigned v = 0;
  for (int i = 0; i < n; ++i) {
    unsigned bit = 0;
    if ((b->pos >> 3) < b->size)
      bit = (b->p[b->pos >> 3] >> (7 - (b->pos & 7))) & 1;
    else
      b->overrun = 1;
    ++b->pos;
    v = v << 1 | bit;
  }
  return v;
}
static inline unsigned hvb_ue(HvBits* b) {
  int z = 0;
  while (z < 32 && !hvb_u(b, 1) && !b->overrun) ++z;
  return z ? ((1u << z) - 1 + hvb_u(b, z)) : 0;
}
static inline int hvb_se(HvBits* b) {
  unsigned k = hvb_ue(b);
  return (k & 1) ? (int)((k + 1) >> 1) : -(int)(k >> 1);
}

/* ---- parameter sets ---- */
typedef struct {
  int num_negative, num_positive;
  int delta_poc_s0[16], delta_poc_s1[16]; /* signed deltas (s0 negative) */
  uint8_t used_s0[16], used_s1[16];
} HvRps;

typedef struct {
  /* ScalingFactor as 8.6.4.2 uses it, [sizeId][matrixId][y * n + x]: 4x4, 8x8, 16x16, 32x32 */
  uint8_t f4[6][16], f8[6][64], f16[6][256], f32[2][1024];
} HvScaling;

typedef struct {
  int valid;
  int id, vps_id, max_sub_layers, chroma_format_idc;
  int width, height; /* pic_width / height_in_luma_samples */
  int conf_left, conf_right, conf_top, conf_bottom; /* luma samples */
  int bit_depth_luma, bit_depth_chroma, log2_max_poc_lsb;
  int max_dec_pic_buffering, max_num_reorder, max_latency_increase_plus1; /* of the highest sub-layer */
  int log2_min_cb, log2_ctb, log2_min_tb, log2_max_tb;
  int max_th_depth_inter, max_th_depth_intra;
  int scaling_list_enabled, scaling_list_present;
  HvScaling scaling;
  int amp_enabled, sao_enabled, pcm_enabled;
  int pcm_bits_luma, pcm_bits_chroma, log2_min_pcm, log2_max_pcm, pcm_loop_filter_disabled;
  int num_st_rps;
  HvRps st_rps[65];
  int long_term_present, num_lt_sps, lt_poc_lsb_sps[32], lt_used_sps[32];
  int temporal_mvp_enabled, strong_intra_smoothing;
  /* derived */
  int ctb_size, width_ctb, height_ctb, width_min_cb, height_min_cb;
} HvSps;

typedef struct {
  int valid;
  int id, sps_id;
  int dependent_slices_enabled, output_flag_present, num_extra_slice_header_bits;
  int sign_data_hiding, cabac_init_present, num_ref_idx_default[2], init_qp;
  int constrained_intra_pred, transform_skip_enabled, cu_qp_delta_enabled, diff_cu_qp_delta_depth;
  int cb_qp_offset, cr_qp_offset, slice_chroma_qp_offsets_present;
  int weighted_pred, weighted_bipred, transquant_bypass_enabled;
  int tiles_enabled, entropy_coding_sync;
  int loop_filter_across_slices, deblocking_control_present, deblocking_override_enabled, deblocking_disabled;
  int beta_offset_div2, tc_offset_div2;
  int scaling_list_present;
  HvScaling scaling;
  int lists_modification_present, log2_parallel_merge_level, slice_header_extension_present;
} HvPps;

enum { HV_SLICE_B = 0, HV_SLICE_P = 1, HV_SLICE_I = 2 };

typedef struct {
  int first_slice_segment_in_pic, no_output_of_prior_pics, pps_id, dependent, segment_address;
  int slice_type, pic_output_flag, poc_lsb;
  int st_rps_sps_flag, st_rps_idx;
  HvRps st_rps_local;
  const HvRps* rps;
  int num_lt, lt_poc_lsb[32], lt_used[32], lt_msb_present[32], lt_delta_msb_cycle[32];
  int temporal_mvp, sao_luma, sao_chroma;
  int num_ref_idx[2];
  int rplm_flag[2], list_entry[2][16];
  int mvd_l1_zero, cabac_init_flag, collocated_from_l0, collocated_ref_idx;
  int luma_log2_wd, chroma_log2_wd;
  int16_t luma_weight[2][16], luma_offset[2][16], chroma_weight[2][16][2], chroma_offset[2][16][2];
  int max_num_merge_cand;
  int qp_delta, cb_qp_offset, cr_qp_offset;
  int deblocking_disabled, beta_offset_div2, tc_offset_div2, loop_filter_across_slices;
  int num_entry_points;
  uint32_t entry_point[256]; /* entry_point_offset_minus1 + 1 of the first 256: checked against the rows the walk finds */
  /* derived */
  int slice_qp, slice_addr_rs; /* SliceAddrRs: address of the first CTB of the (independent) slice */
  size_t data_byte_offset;     /* of slice_segment_data() within the RBSP */
} HvSliceHeader;

/* ---- pictures ---- */
typedef struct {
  int16_t mv[2][2];
  int8_t ref_idx[2];
  uint8_t pred_flag;   /* bit 0: list 0, bit 1: list 1; 0: intra (or not coded) */
  uint8_t ref_long[2]; /* the reference was a long-term picture when this one was decoded */
  int32_t ref_poc[2];
} HvMv;

typedef struct {
  uint8_t* plane[3]; /* w x h, (w/2) x (h/2) x 2: exactly the coded size, no padding */
  HvMv* mvf;         /* per 4x4 luma block */
  int poc, ref_state /* 0 unused, 1 short-term, 2 long-term */, needed_for_output, in_use;
  int latency_count;
  uint64_t decode_order;
  int missing; /* generated in place of a reference that was never decoded (8.3.3) */
} HvPic;

/* ---- what the slice data says about one coding unit (filled by the parser, or by an encoder's analysis) ---- */
enum { HV_PART_2Nx2N = 0, HV_PART_2NxN, HV_PART_Nx2N, HV_PART_NxN, HV_PART_2NxnU, HV_PART_2NxnD, HV_PART_nLx2N, HV_PART_nRx2N };

typedef struct {
  uint8_t merge_flag, merge_idx, inter_pred_idc /* 0 L0, 1 L1, 2 BI */, mvp_flag[2];
  int8_t ref_idx[2];
  int16_t mvd[2][2];
  /* derived (8.5.3.2) */
  HvMv mv;
  int x, y, w, h;
} HvPu;

typedef struct {
  uint16_t x, y;       /* luma position in the picture */
  uint8_t log2;        /* luma transform block size */
  uint8_t cbf[3];
  uint8_t tskip[3];
  uint8_t has_chroma;  /* chroma blocks are reconstructed with this unit: log2 > 2, or the fourth 4x4 of an 8x8 */
  uint16_t cx, cy;     /* chroma block position (chroma samples) */
  uint8_t clog2;
  uint8_t scan_idx[3]; /* as used when parsed (informative) */
  int32_t coef[3];     /* offsets into the picture's coefficient pool: n x n levels, raster, of the blocks with cbf */
} HvTu;

typedef struct {
  uint16_t x, y;
  uint8_t log2, depth;
  uint8_t tq_bypass, skip, intra /* CuPredMode == MODE_INTRA */, part_mode, pcm;
  int8_t qp_y;
  int8_t qp_delta;      /* CuQpDeltaVal coded in this unit (0 when none) */
  uint8_t qp_delta_coded;
  /* intra */
  uint8_t prev_intra_luma_pred_flag[4], mpm_idx[4], rem_intra_luma_pred_mode[4], intra_chroma_pred_mode;
  uint8_t luma_mode[4], chroma_mode; /* derived */
  /* inter */
  uint8_t num_pu;
  HvPu pu[4];
  uint8_t rqt_root_cbf;
  /* transform tree: nodes in pre-order */
  int32_t first_node, num_nodes;
  int32_t first_tu, num_tu;
  int32_t pcm_samples; /* pool offset: luma n*n then Cb, Cr */
} HvCu;

typedef struct {
  uint8_t split, cbf_cb, cbf_cr, cbf_luma;
} HvNode;

typedef struct {
  uint8_t type[3];       /* SaoTypeIdx: 0 off, 1 band, 2 edge */
  uint8_t band_pos[3], eo_class[3];
  int8_t offset[3][4];   /* signed SaoOffsetVal[1..4] */
  uint8_t merge_left, merge_up;
} HvSao;

/* per-slice facts the in-loop filters need after the slice header is gone */
typedef struct {
  int slice_addr_rs;
  int deblocking_disabled, beta_offset_div2, tc_offset_div2, loop_filter_across_slices;
  int cb_qp_offset, cr_qp_offset;
  int sao_luma, sao_chroma;
} HvSliceFacts;

/* ---- CABAC ---- */
typedef struct {
  const uint8_t* p;
  size_t size, pos; /* bits */
  unsigned range, offset;
  int overrun;
} HvCabacDec;
typedef struct {
  uint8_t* buf;
  size_t cap, nbits;
  unsigned low, range;
  int outstanding, first;
} HvCabacEnc;

struct HevcOracle {
  char error[256];
  /* configuration */
  int nal_length_size;
  HvSps sps[16];
  HvPps pps[64];
  const HvSps* cur_sps;
  const HvPps* cur_pps;
  /* DPB */
  HvPic dpb[HV_MAX_DPB];
  HvPic* cur;
  uint64_t decode_counter;
  int prev_tid0_poc;
  int first_picture; /* the next picture starts a coded video sequence for this decoder (after open / flush) */
  int no_rasl_output;
  int poc;
  /* reference picture set of the current picture and the lists of the current slice */
  HvPic* rps_st_before[16];
  HvPic* rps_st_after[16];
  HvPic* rps_lt[16];
  int n_st_before, n_st_after, n_lt;
  HvPic* ref_list[2][HV_MAX_REFS];
  /* output queue (display order) */
  struct {
    uint8_t* i420;
    int w, h, poc;
  } outq[64];
  int out_head, out_count;
  /* ---- current picture: maps per 4x4 luma block (w4 x h4) ---- */
  int w4, h4;
  uint8_t* m_pred;   /* 0 not decoded yet, 1 inter, 2 intra */
  uint8_t* m_skip;
  uint8_t* m_depth;  /* CtDepth */
  uint8_t* m_imode;  /* IntraPredModeY */
  uint8_t* m_flags;  /* bit 0: pcm, bit 1: cu_transquant_bypass, bit 2: pcm+loop filter disabled or bypass (in-loop filters leave it alone) */
  int8_t* m_qp;      /* QpY */
  uint8_t* m_nz;     /* the luma transform block that covers it has non-zero levels */
  uint8_t* m_edge;   /* bit 0 / 1: transform / prediction block edge on its left; bit 2 / 3: on its top; bit 4 / 5: left / top edge is to be filtered (filterEdgeFlag) */
  uint16_t* m_slice; /* per CTB: index into slice_facts */
  HvSao* sao;        /* per CTB */
  HvSliceFacts slice_facts[HV_MAX_SLICES];
  int num_slices;
  uint8_t* bs[2];    /* [0] vertical edges, [1] horizontal: per 4x4 block, the strength of the edge on its left / top */
  /* the picture's pools */
  HvCu* cus;
  int num_cu, cap_cu;
  HvTu* tus;
  int num_tu, cap_tu;
  HvNode* nodes;
  int num_nodes, cap_nodes;
  int16_t* coefs;
  int num_coefs, cap_coefs;
  uint8_t* deblocked[3]; /* scratch: the picture after deblocking, input of SAO */
  /* ---- current slice ---- */
  HvSliceHeader sh;
  int slice_idx;
  int enc; /* the slice data walk writes instead of reading */
  HvCabacDec cd;
  HvCabacEnc ce;
  uint8_t ctx[HC_CTX_COUNT];       /* pStateIdx << 1 | valMps */
  uint8_t ctx_wpp[HC_CTX_COUNT];   /* saved after the second CTB of a row (9.3.2.2 with entropy_coding_sync) */
  int is_cu_qp_delta_coded, cu_qp_delta_val, qp_y_prev, qg_x, qg_y;
  int last_cu_qp;  /* QpY of the last coding unit in decoding order (qPY_PREV of the next quantisation group) */
  int ctb_addr_rs;
  int recon_next;  /* units [recon_next, num_cu) have been parsed but not reconstructed (a slice at a time) */
  int walk_cu;     /* enc: next unit of the plan (d->cus) the walk writes */
  int enc_end_ctb; /* enc: address of the coding tree block behind the slice's last one */
  int enc_nsub;    /* enc, wavefront rows: the substreams begun after the slice's first ... */
  size_t enc_sub_start[256]; /* ... and the byte of the slice data each begins at (the encoder's entry points) */
  int greater1_state; /* c1 of 9.3.4.2.6 carried across sub-blocks */
  /* expected picture hash (SEI), -1 = none */
  int have_md5;
  uint8_t md5[3][16];
  int md5_checked, md5_mismatch;
  int64_t counters[12];
  int parse_only;
};

/* hevc_headers.c */
void hv_fail(HevcOracle* d, const char* fmt, ...);
size_t hv_unescape(const uint8_t* nal, size_t n, uint8_t* out);
int hv_parse_sps(HevcOracle* d, HvBits* b);
int hv_parse_pps(HevcOracle* d, HvBits* b);
int hv_parse_slice_header(HevcOracle* d, HvBits* b, int nal_type, HvSliceHeader* sh);
int hv_parse_st_rps(HvBits* b, HvRps* out, int idx, int num_sps_sets, const HvRps* sets);
void hv_scaling_default(HvScaling* s);
int hv_parse_scaling_list(HvBits* b, HvScaling* s);
HvPic* hv_alloc_pic(HevcOracle* d);
int hv_start_picture(HevcOracle* d, int nal_type, int temporal_id);
int hv_build_ref_lists(HevcOracle* d);
void hv_finish_picture(HevcOracle* d);
int hv_begin_picture_state(HevcOracle* d);
int hv_complete_picture(HevcOracle* d);
void hevco_set_parse_only(HevcOracle* d, int on);
void hv_md5(const uint8_t* data, size_t n, uint8_t out[16]);
void hv_picture_md5(const HevcOracle* d, const HvPic* p, uint8_t out[3][16]);

/* hevc_slice.c */
void hv_cabac_init_contexts(HevcOracle* d);
void hv_cabac_dec_start(HvCabacDec* c, const uint8_t* p, size_t size, size_t byte_pos);
void hv_cabac_enc_start(HvCabacEnc* e);
void hv_cabac_enc_finish_slice(HvCabacEnc* e); /* after end_of_slice_segment_flag = 1: flush incl. the stop bit, aligned */
int hv_slice_data(HevcOracle* d); /* walks slice_segment_data(): parses (d->enc == 0) or writes (d->enc == 1) */
/* derivations shared with an encoder's analysis */
void hv_mpm_candidates(const HevcOracle* d, int x, int y, int cand[3]);
int hv_zscan_available(const HevcOracle* d, int xc, int yc, int xn, int yn);
void hv_merge_candidates(HevcOracle* d, const HvCu* cu, int xp, int yp, int w, int h, int part_idx, HvMv out[5], int* count);
void hv_amvp_candidates(HevcOracle* d, const HvCu* cu, int xp, int yp, int w, int h, int part_idx, int list, int ref_idx,
                        int16_t out[2][2]);
void hv_pu_geometry(const HvCu* cu, int part_idx, int* x, int* y, int* w, int* h);
int hv_num_parts(int part_mode);
void hv_derive_pu_motion(HevcOracle* d, HvCu* cu, int part_idx);
void hv_store_pu_motion(HevcOracle* d, const HvPu* pu);
int hv_qp_y_pred(HevcOracle* d, int x_qg, int y_qg);
void hv_set_cu_maps(HevcOracle* d, const HvCu* cu);
HvCu* hv_new_cu(HevcOracle* d);
HvTu* hv_new_tu(HevcOracle* d);
int hv_new_nodes(HevcOracle* d, int n);
int32_t hv_new_coefs(HevcOracle* d, int n);
int hv_chroma_qp(int qp_y, int offset);
int hv_chroma_mode_from_syntax(int intra_chroma_pred_mode, int luma_mode);
void hv_sdh_fix(int16_t* levels, int log2, int scan_idx);
void hv_derive_intra_modes(HevcOracle* d, HvCu* cu);
/* predSamplesLX of 8.5.3.3.3 (14-bit intermediate) for one block of one plane */
void hv_interp_block(const uint8_t* ref, int rw, int rh, int chroma, int x, int y, int w, int h, int mvx, int mvy, int16_t* out);

/* hevc_recon.c */
void hv_intra_predict(HevcOracle* d, int c_idx, int x, int y, int log2, int mode);
void hv_inter_predict_pu(HevcOracle* d, const HvPu* pu);
void hv_residual_add(HevcOracle* d, const HvCu* cu, const HvTu* tu, int c_idx);
void hv_inverse_transform(const HevcOracle* d, const int16_t* levels, int log2, int c_idx, int qp, int intra, int tskip,
                          int bypass, int16_t* res);
void hv_recon_cu(HevcOracle* d, const HvCu* cu);
void hv_mark_edges(HevcOracle* d, const HvCu* cu); /* (hv_recon_cu does it; an encoder that reconstructs piecewise calls it itself) */
void hv_deblock_picture(HevcOracle* d);
void hv_sao_picture(HevcOracle* d);
const uint8_t* hv_diag_scan(int log2_blocks); /* up-right diagonal scan of a (1 << log2) square: x | y << 4 */
const uint8_t* hv_scan(int log2_blocks, int scan_idx);

#endif
