/* TEST INFRASTRUCTURE -- internal structures of the CPU oracle decoder. */
#ifndef ORACLE_H264_INTERNAL_H_
#define ORACLE_H264_INTERNAL_H_

#include "h264_oracle.h"
#include "h264_tables.h"

#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define MAX_SPS 32
#define MAX_PPS 256
#define MAX_DPB 36      /* picture pool */
#define MAX_REFS 32

static inline int iclip(int lo, int hi, int v) { return v < lo ? lo : (v > hi ? hi : v); }
static inline int clip8(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }
static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int imax(int a, int b) { return a > b ? a : b; }
static inline int iabs(int a) { return a < 0 ? -a : a; }
static inline int median3(int a, int b, int c) {
  return a > b ? (b > c ? b : (a > c ? c : a)) : (a > c ? a : (b > c ? c : b));
}

/* ---- bit reader over an RBSP (emulation prevention already removed) ------ */
typedef struct {
  const uint8_t* buf;
  int size_bits;
  int pos;
  int overrun;
} Bits;

static inline int bits_left(const Bits* b) { return b->size_bits - b->pos; }
static inline unsigned get_bit(Bits* b) {
  if (b->pos >= b->size_bits) {
    b->overrun = 1;
    return 0;
  }
  unsigned v = (b->buf[b->pos >> 3] >> (7 - (b->pos & 7))) & 1;
  b->pos++;
  return v;
}
static inline unsigned get_bits(Bits* b, int n) {
  unsigned v = 0;
  while (n-- > 0) v = (v << 1) | get_bit(b);
  return v;
}
static inline unsigned show_bits(const Bits* b, int n) {
  Bits t = *b;
  return get_bits(&t, n);
}
static inline unsigned get_ue(Bits* b) {
  int zeros = 0;
  while (!get_bit(b)) {
    if (++zeros > 32 || b->overrun) return 0;
  }
  return zeros ? ((1u << zeros) - 1 + get_bits(b, zeros)) : 0;
}
static inline int get_se(Bits* b) {
  unsigned k = get_ue(b);
  return (k & 1) ? (int)((k + 1) >> 1) : -(int)(k >> 1);
}
static inline int get_te(Bits* b, int max) {
  if (max > 1) return (int)get_ue(b);
  return !get_bit(b);
}

/* ---- parameter sets ------------------------------------------------------- */
typedef struct {
  int valid;
  int profile_idc, level_idc, constraint_flags;
  int chroma_format_idc;
  int log2_max_frame_num;
  int poc_type;
  int log2_max_poc_lsb;
  int delta_pic_order_always_zero;
  int offset_for_non_ref_pic, offset_for_top_to_bottom;
  int num_ref_frames_in_poc_cycle;
  int offset_for_ref_frame[256];
  int max_num_ref_frames;
  int gaps_allowed;
  int mb_width, mb_height;
  int frame_mbs_only;
  int direct_8x8_inference;
  int crop_left, crop_right, crop_top, crop_bottom; /* in luma samples */
  int bitstream_restriction, num_reorder_frames, max_dec_frame_buffering;
  int scaling_matrix_present;
  uint8_t scaling4[6][16]; /* raster order */
  uint8_t scaling8[2][64];
} SPS;

typedef struct {
  int valid;
  int sps_id;
  int cabac;
  int bottom_field_pic_order_present;
  int num_ref_idx_default[2];
  int weighted_pred, weighted_bipred_idc;
  int init_qp;
  int chroma_qp_offset[2];
  int deblocking_control_present;
  int constrained_intra_pred;
  int redundant_pic_cnt_present;
  int transform_8x8_mode;
  uint8_t scaling4[6][16];
  uint8_t scaling8[2][64];
} PPS;

/* ---- pictures ------------------------------------------------------------- */
typedef struct Picture {
  uint8_t* plane[3];
  int stride[3];
  int in_pool_use;       /* owns live data */
  int uid;               /* unique per decoded picture */
  int poc;
  int frame_num;
  int frame_num_wrap;
  int pic_num;
  int ref;               /* 0 unused, 1 short-term, 2 long-term */
  int long_term_idx;
  int needed_for_output;
  int is_idr;
  /* motion data kept for co-located lookups (temporal/spatial direct) and
   * for deblocking while this is the current picture */
  int16_t (*mv[2])[2];   /* [mb*16 + raster4x4][x,y] */
  int8_t* ref_idx[2];    /* [mb*4 + blk8] ; -1 = list unused */
  int32_t* ref_uid[2];   /* uid of the picture referenced, -1 unused */
  int32_t* ref_poc[2];   /* POC of that picture */
  uint8_t* ref_long[2];  /* that picture was long-term when referenced */
  uint8_t* mb_intra;     /* per MB */
} Picture;

typedef struct {
  uint8_t *y, *u, *v;
  int w, h, poc;
} OutFrame;

/* ---- slice ---------------------------------------------------------------- */
typedef struct {
  int first_mb;
  int type; /* 0 P, 1 B, 2 I */
  int pps_id;
  int frame_num;
  int idr_pic_id;
  int poc_lsb, delta_poc_bottom, delta_poc[2];
  int direct_spatial;
  int num_ref_idx[2];
  int cabac_init_idc;
  int qp;
  int disable_deblock;
  int alpha_offset, beta_offset; /* already *2 */
  /* weighted prediction */
  int luma_log2_denom, chroma_log2_denom;
  int luma_weight[2][MAX_REFS], luma_offset[2][MAX_REFS];
  int chroma_weight[2][MAX_REFS][2], chroma_offset[2][MAX_REFS][2];
  int use_weight; /* 0 default, 1 explicit, 2 implicit */
  int implicit_weight[MAX_REFS][MAX_REFS];
  /* ref list modification + marking ops are applied while parsing */
  int nal_ref_idc, is_idr;
  int no_output_of_prior_pics, long_term_reference_flag;
  int adaptive_marking;
  int mmco_count;
  struct { int op, a, b; } mmco[66];
  Picture* ref_list[2][MAX_REFS + 1];
  int ref_count[2];
} Slice;

/* per-slice deblock params are needed per MB (filter uses the slice of q) */
typedef struct {
  int8_t disable, alpha, beta;
} MbDeblock;

/* ---- CABAC engine ----------------------------------------------------------- */
typedef struct {
  Bits* bits;
  unsigned range, offset;
  uint8_t state[CABAC_CTX_TOTAL]; /* (pStateIdx << 1) | valMPS */
} Cabac;

/* ---- bit writer + CABAC encoder (corpus generator only) ---------------------- */
typedef struct {
  uint8_t* buf;
  size_t cap;
  size_t pos; /* bits */
} BitWriter;

typedef struct {
  BitWriter* w;
  unsigned low, range;
  int outstanding, first;
} CabacEnc;

/* ---- current macroblock ------------------------------------------------------ */
enum { MB_NONE = 0, MB_I4x4, MB_I8x8, MB_I16x16, MB_IPCM, MB_INTER };

typedef struct {
  int mbx, mby, addr;
  int kind;
  int skip, direct16; /* B_Skip / B_Direct_16x16 */
  int t8x8;
  int i16_mode, chroma_mode;
  int8_t i4_mode[16];   /* by raster4x4 */
  int cbp;              /* luma | chroma << 4 */
  int qp, qpc[2];
  int qp_delta;
  /* partitions (inter) */
  int mb_type_raw;
  int sub_type[4];
  int16_t mv[2][16][2];  /* raster4x4 */
  int8_t ref[2][4];      /* per 8x8 */
  /* residual levels at raster positions */
  int16_t luma_dc[16];
  int16_t luma[16][16];  /* [raster4x4 block][raster coef] */
  int16_t luma8[4][64];
  int16_t chroma_dc[2][4];
  int16_t chroma_ac[2][4][16];
  uint8_t pcm[384];
  uint8_t nnz[24];       /* total_coeff per block: luma raster 0..15, cb 16..19, cr 20..23 */
  uint8_t nnz444[32];    /* 4:4:4 (parse-only mode): Cb raster 0..15, Cr 16..31 - coded like luma */
  int cbf_dc;            /* bit0 luma DC, bit1 cb DC, bit2 cr DC */
} MB;

struct H264Oracle {
  char error[256];
  int nal_length_size;
  SPS sps[MAX_SPS];
  PPS pps[MAX_PPS];
  const SPS* asps; /* active */
  const PPS* apps;
  int mbw, mbh;

  /* RBSP scratch */
  uint8_t* rbsp;
  size_t rbsp_cap;

  /* picture pool / DPB */
  Picture pool[MAX_DPB];
  int pool_w, pool_h;
  int next_uid;
  Picture* cur;
  int pic_in_progress;

  /* POC / frame_num state */
  int prev_poc_msb, prev_poc_lsb;
  int prev_frame_num, prev_frame_num_offset;
  int prev_ref_has_mmco5;
  int max_long_term_frame_idx; /* -1 = none */

  /* output */
  OutFrame* outq;
  int outq_len, outq_cap;

  /* current slice + per-picture MB state */
  Slice sl;
  int slice_index;        /* increments per slice within the picture */
  int16_t* mb_slice;      /* -1 = not decoded yet */
  uint8_t* mb_kind;
  uint8_t* mb_skip;
  uint8_t* mb_direct;
  uint8_t* direct8;       /* mb*4 + blk8: sub-macroblock predicted in direct mode */
  uint8_t* mb_t8x8;
  uint8_t* mb_cbp;
  uint8_t* mb_cbf_dc;
  int8_t* mb_qp;
  uint8_t* mb_chroma_mode;
  int8_t* i4mode;         /* mb*16 + raster4x4 */
  uint8_t* nnz;           /* mb*24 */
  uint8_t* nnz444;        /* mb*32: Cb, Cr blocks of 4:4:4 pictures (parse-only mode) */
  /* Parse-only mode (h264o_set_parse_only): syntax, entropy decoding, reference handling and output order run, no
   * sample is reconstructed.  In this mode - and only in it - High 4:4:4 Predictive streams (chroma_format_idc 3,
   * one colour plane arrangement, 8 bit) are accepted: Cb and Cr residuals are parsed like luma. */
  int parse_only;
  int chroma444;          /* the active SPS has chroma_format_idc 3 */
  int x264_build;         /* "x264 - core N" from the stream's version SEI, -1 if none (set once, kept) */
  int no_quirks;          /* h264o_set_parse_only(d, 2): the version SEI is ignored */
  uint8_t (*mvd)[2][16][2]; /* |mvd| clipped, per MB/list/raster4x4/comp */
  MbDeblock* mb_deblock;
  int8_t* mb_qpc[2];
  /* per-MB reference picture identity for deblocking is in cur->ref_uid */

  Cabac cabac;
  CabacEnc cenc;
  int last_qp_delta_nonzero;
  int qp; /* running QP_Y */
  MB mb;

  int64_t counters[10];
  int64_t* cov;        /* coverage histogram (h264_oracle.h), NULL = not collected */
  int cov_ctx_table;   /* which H264O_COV_CTX table the current slice's bins count in */
  int pred_luma_only; /* corpus encoder: motion search evaluates luma only */
};

#define COV(d, cell) do { if ((d)->cov) (d)->cov[(cell)]++; } while (0)

/* h264_headers.c */
int h264i_fail(H264Oracle* d, const char* msg);
int h264i_parse_sps(H264Oracle* d, Bits* b);
int h264i_parse_pps(H264Oracle* d, Bits* b);
int h264i_parse_slice_header(H264Oracle* d, Bits* b, int nal_ref_idc, int nal_type);
int h264i_start_picture(H264Oracle* d);
int h264i_finish_picture(H264Oracle* d);
int h264i_more_rbsp_data(const Bits* b);
void h264i_output_all(H264Oracle* d);

/* h264_slice.c */
int h264i_decode_slice_data(H264Oracle* d, Bits* b);
/* encoder side of the same syntax engine (used by corpus/h264_encoder.c) */
void bw_put(BitWriter* w, unsigned v, int n);
void bw_ue(BitWriter* w, unsigned v);
void bw_se(BitWriter* w, int v);
typedef struct {
  int is_skip; /* P_Skip / B_Skip */
  MB mb;       /* target values: kind, mb_type_raw, modes, sub types, refs, mvs, cbp, t8x8, qp, levels */
} MbPlan;
void h264i_enc_begin_slice(H264Oracle* d, BitWriter* w);
/* Codes macroblock `addr` as `plan` says, reconstructs it and updates all
 * neighbour state exactly as the decoder would.  Returns 0 / -1. */
int h264i_enc_mb(H264Oracle* d, BitWriter* w, int addr, const MbPlan* plan, int last_in_slice);
void h264i_enc_end_slice(H264Oracle* d, BitWriter* w);
/* Neighbour-state queries the mode decision needs (valid after
 * h264i_begin_mb). */
void h264i_begin_mb(H264Oracle* d, int addr);
void h264i_pskip_mv(H264Oracle* d, int mv[2]);
void h264i_pred_mv16(H264Oracle* d, int list, int ref, int mv[2]);
int h264i_pred_direct_mb(H264Oracle* d);
void h264i_pred_mv_part(H264Oracle* d, int shape, int part, int ref, int mv[2]);
void h264i_set_part(H264Oracle* d, int shape, int part, int ref, int mvx, int mvy);
void h264i_pred_mv_part_l(H264Oracle* d, int list, int shape, int part, int ref, int mv[2]);
void h264i_set_part_l(H264Oracle* d, int list, int shape, int part, int ref, int mvx, int mvy);
void h264i_pred_mv_blk(H264Oracle* d, int bx, int by, int bw, int ref, int mv[2]);
void h264i_set_blk(H264Oracle* d, int bx, int by, int bw, int bh, int mvx, int mvy);
void h264i_set_ref8(H264Oracle* d, int b8, int ref);

/* h264_recon.c */
void h264i_recon_mb(H264Oracle* d);
void h264i_inter_pred_mb(H264Oracle* d);
void h264i_deblock_picture(H264Oracle* d);
/* prediction primitives, also used by the corpus encoder's mode decision */
void h264i_pred4x4(uint8_t* dst, int stride, int mode, int has_left, int has_top, int has_tl, int has_tr);
void h264i_pred8x8l(uint8_t* dst, int stride, int mode, int has_left, int has_top, int has_tl, int has_tr);
void h264i_pred16x16(uint8_t* dst, int stride, int mode, int has_left, int has_top);
void h264i_pred_chroma8x8(uint8_t* dst, int stride, int mode, int has_left, int has_top);
int h264i_intra_mb_ok(const H264Oracle* d, int mbx, int mby);
int h264i_intra_blk_ok(const H264Oracle* d, int rx, int ry, int cur_zz);
/* w x h block of plane `plane` predicted from `ref` at integer position
 * (x, y) + motion vector (quarter-sample luma / eighth-sample chroma units) */
void h264i_mc_block(const Picture* ref, int plane, int pic_w, int pic_h, int x, int y, int mvx,
                    int mvy, int w, int h, uint8_t* dst, int dst_stride);
int h264i_level_scale4(const uint8_t* weight, int qp, int pos);
int h264i_level_scale8(const uint8_t* weight, int qp, int pos);

#endif
