/* TEST INFRASTRUCTURE -- oracle: NAL layer, parameter sets, slice header,
 * POC (8.2.1), reference lists (8.2.4), marking (8.2.5), DPB output
 * (Annex C bumping) and the public API.  Restates what the reference gets
 * from FFmpeg's h264dec.c / h264_ps.c / h264_slice.c / h264_refs.c /
 * h264_picture.c behind software_video_decoder.cpp:349-457. */
#include "h264_internal.h"

#include <stdio.h>

int h264i_fail(H264Oracle* d, const char* msg) {
  if (!d->error[0]) snprintf(d->error, sizeof(d->error), "%s", msg);
  return -1;
}

/* ------------------------------------------------------------------------- */
static int unescape_nal(H264Oracle* d, const uint8_t* src, size_t n) {
  if (d->rbsp_cap < n + 8) {
    free(d->rbsp);
    d->rbsp_cap = n + 1024;
    d->rbsp = (uint8_t*)malloc(d->rbsp_cap);
    if (!d->rbsp) return -1;
  }
  size_t o = 0;
  int zeros = 0;
  for (size_t i = 0; i < n; ++i) {
    uint8_t c = src[i];
    if (zeros >= 2 && c == 3) {
      zeros = 0;
      continue; /* emulation_prevention_three_byte */
    }
    d->rbsp[o++] = c;
    zeros = c == 0 ? zeros + 1 : 0;
  }
  return (int)o;
}

/* true while there is payload before rbsp_trailing_bits */
int h264i_more_rbsp_data(const Bits* b) {
  int last = b->size_bits - 1;
  while (last >= 0 && !((b->buf[last >> 3] >> (7 - (last & 7))) & 1)) --last;
  return b->pos < last;
}

/* ------------------------------------------------------------------------- */
static void scaling_list(Bits* b, uint8_t* out_raster, int size, const uint8_t* fallback_raster,
                         const uint8_t* default_zz) {
  const uint8_t* scan = size == 16 ? zigzag4x4 : zigzag8x8;
  if (!get_bit(b)) { /* list not present: fall-back rule */
    memcpy(out_raster, fallback_raster, size);
    return;
  }
  int last = 8, next = 8;
  for (int j = 0; j < size; ++j) {
    if (next != 0) {
      int delta = get_se(b);
      next = (last + delta + 256) & 255;
      if (j == 0 && next == 0) { /* useDefaultScalingMatrixFlag */
        for (int k = 0; k < size; ++k) out_raster[scan[k]] = default_zz[k];
        return;
      }
    }
    last = next == 0 ? last : next;
    out_raster[scan[j]] = (uint8_t)last;
  }
}

static void parse_scaling_matrices(Bits* b, int want8x8, uint8_t s4[6][16], uint8_t s8[2][64],
                                   const uint8_t fb4[6][16], const uint8_t fb8[2][64],
                                   int use_sps_fallback) {
  uint8_t d4i[16], d4p[16], d8i[64], d8p[64];
  for (int k = 0; k < 16; ++k) {
    d4i[zigzag4x4[k]] = default_4x4_intra[k];
    d4p[zigzag4x4[k]] = default_4x4_inter[k];
  }
  for (int k = 0; k < 64; ++k) {
    d8i[zigzag8x8[k]] = default_8x8_intra[k];
    d8p[zigzag8x8[k]] = default_8x8_inter[k];
  }
  /* fall-back rule A (SPS): defaults; rule B (PPS): the SPS lists */
  const uint8_t* f0 = use_sps_fallback ? fb4[0] : d4i;
  const uint8_t* f3 = use_sps_fallback ? fb4[3] : d4p;
  const uint8_t* f6 = use_sps_fallback ? fb8[0] : d8i;
  const uint8_t* f7 = use_sps_fallback ? fb8[1] : d8p;
  scaling_list(b, s4[0], 16, f0, default_4x4_intra);
  scaling_list(b, s4[1], 16, s4[0], default_4x4_intra);
  scaling_list(b, s4[2], 16, s4[1], default_4x4_intra);
  scaling_list(b, s4[3], 16, f3, default_4x4_inter);
  scaling_list(b, s4[4], 16, s4[3], default_4x4_inter);
  scaling_list(b, s4[5], 16, s4[4], default_4x4_inter);
  if (want8x8) {
    scaling_list(b, s8[0], 64, f6, default_8x8_intra);
    scaling_list(b, s8[1], 64, f7, default_8x8_inter);
  } else {
    memcpy(s8[0], f6, 64);
    memcpy(s8[1], f7, 64);
  }
}

static int max_dpb_frames(const SPS* s) {
  static const int lv[][2] = {{10, 396},   {11, 900},   {12, 2376},  {13, 2376},   {20, 2376},
                              {21, 4752},  {22, 8100},  {30, 8100},  {31, 18000},  {32, 20480},
                              {40, 32768}, {41, 32768}, {42, 34816}, {50, 110400}, {51, 184320},
                              {52, 184320}};
  int mbs = 184320;
  for (size_t i = 0; i < sizeof(lv) / sizeof(lv[0]); ++i)
    if (lv[i][0] == s->level_idc) mbs = lv[i][1];
  if (s->level_idc == 11 && (s->constraint_flags & 0x10)) mbs = 396; /* level 1b */
  int n = mbs / (s->mb_width * s->mb_height);
  return iclip(1, 16, n);
}

int h264i_parse_sps(H264Oracle* d, Bits* b) {
  SPS s;
  memset(&s, 0, sizeof(s));
  s.profile_idc = get_bits(b, 8);
  s.constraint_flags = get_bits(b, 8);
  s.level_idc = get_bits(b, 8);
  unsigned id = get_ue(b);
  if (id >= MAX_SPS) return h264i_fail(d, "sps_id out of range");
  s.chroma_format_idc = 1;
  memset(s.scaling4, 16, sizeof(s.scaling4));
  memset(s.scaling8, 16, sizeof(s.scaling8));
  int p = s.profile_idc;
  if (p == 100 || p == 110 || p == 122 || p == 244 || p == 44 || p == 83 || p == 86 || p == 118 ||
      p == 128 || p == 138 || p == 139 || p == 134 || p == 135) {
    s.chroma_format_idc = get_ue(b);
    int separate = 0;
    if (s.chroma_format_idc == 3) separate = get_bit(b); /* separate_colour_plane_flag */
    int bdl = get_ue(b), bdc = get_ue(b);
    int bypass = get_bit(b);
    /* 4:4:4 (one plane arrangement): in parse-only mode only - its slice data can be parsed, not reconstructed */
    const int ok444 = d->parse_only && s.chroma_format_idc == 3 && !separate;
    if ((s.chroma_format_idc != 1 && !ok444) || bdl != 0 || bdc != 0 || bypass)
      return h264i_fail(d, "unsupported: only 8-bit 4:2:0 without transform bypass");
    s.scaling_matrix_present = get_bit(b);
    if (s.scaling_matrix_present) {
      uint8_t flat4[6][16], flat8[2][64];
      memset(flat4, 16, sizeof(flat4));
      memset(flat8, 16, sizeof(flat8));
      parse_scaling_matrices(b, 1, s.scaling4, s.scaling8, flat4, flat8, 0);
      if (s.chroma_format_idc == 3) { /* the 8x8 lists of Cb and Cr (7.3.2.1.1: twelve lists): read and dropped */
        uint8_t scratch[64];
        for (int k = 0; k < 4; ++k) scaling_list(b, scratch, 64, flat8[0], default_8x8_intra);
      }
    }
  }
  s.log2_max_frame_num = get_ue(b) + 4;
  s.poc_type = get_ue(b);
  if (s.poc_type == 0) {
    s.log2_max_poc_lsb = get_ue(b) + 4;
  } else if (s.poc_type == 1) {
    s.delta_pic_order_always_zero = get_bit(b);
    s.offset_for_non_ref_pic = get_se(b);
    s.offset_for_top_to_bottom = get_se(b);
    s.num_ref_frames_in_poc_cycle = get_ue(b);
    if (s.num_ref_frames_in_poc_cycle > 255) return h264i_fail(d, "bad poc cycle");
    for (int i = 0; i < s.num_ref_frames_in_poc_cycle; ++i) s.offset_for_ref_frame[i] = get_se(b);
  } else if (s.poc_type != 2) {
    return h264i_fail(d, "bad pic_order_cnt_type");
  }
  s.max_num_ref_frames = get_ue(b);
  s.gaps_allowed = get_bit(b);
  s.mb_width = get_ue(b) + 1;
  s.mb_height = get_ue(b) + 1;
  s.frame_mbs_only = get_bit(b);
  if (!s.frame_mbs_only) return h264i_fail(d, "unsupported: interlaced (frame_mbs_only_flag=0)");
  s.direct_8x8_inference = get_bit(b);
  if (get_bit(b)) { /* frame_cropping_flag; 4:2:0 frame units are 2 luma samples (4:4:4: one) */
    const int unit = s.chroma_format_idc == 3 ? 1 : 2;
    s.crop_left = unit * get_ue(b);
    s.crop_right = unit * get_ue(b);
    s.crop_top = unit * get_ue(b);
    s.crop_bottom = unit * get_ue(b);
  }
  s.num_reorder_frames = -1;
  if (get_bit(b)) { /* vui_parameters_present_flag */
    if (get_bit(b)) { /* aspect_ratio_info_present */
      if (get_bits(b, 8) == 255) get_bits(b, 32);
    }
    if (get_bit(b)) get_bit(b); /* overscan */
    if (get_bit(b)) {           /* video_signal_type */
      get_bits(b, 4);
      if (get_bit(b)) get_bits(b, 24);
    }
    if (get_bit(b)) { /* chroma_loc */
      get_ue(b);
      get_ue(b);
    }
    if (get_bit(b)) { /* timing */
      get_bits(b, 32);
      get_bits(b, 32);
      get_bit(b);
    }
    int hrd_present = 0;
    for (int pass = 0; pass < 2; ++pass) { /* nal_hrd then vcl_hrd */
      if (!get_bit(b)) continue;
      hrd_present = 1;
      int cpb = get_ue(b) + 1;
      get_bits(b, 8); /* bit_rate_scale, cpb_size_scale */
      for (int i = 0; i < cpb && !b->overrun; ++i) {
        get_ue(b);
        get_ue(b);
        get_bit(b);
      }
      get_bits(b, 20); /* four 5-bit length fields */
    }
    if (hrd_present) get_bit(b); /* low_delay_hrd_flag */
    get_bit(b); /* pic_struct_present_flag */
    s.bitstream_restriction = get_bit(b);
    if (s.bitstream_restriction) {
      get_bit(b);
      get_ue(b);
      get_ue(b);
      get_ue(b);
      get_ue(b);
      s.num_reorder_frames = get_ue(b);
      s.max_dec_frame_buffering = get_ue(b);
    }
  }
  if (b->overrun) return h264i_fail(d, "truncated SPS");
  if (s.mb_width > 512 || s.mb_height > 512) return h264i_fail(d, "picture too large");
  s.valid = 1;
  d->sps[id] = s;
  return 0;
}

int h264i_parse_pps(H264Oracle* d, Bits* b) {
  PPS p;
  memset(&p, 0, sizeof(p));
  unsigned id = get_ue(b);
  if (id >= MAX_PPS) return h264i_fail(d, "pps_id out of range");
  p.sps_id = get_ue(b);
  if (p.sps_id >= MAX_SPS || !d->sps[p.sps_id].valid) return h264i_fail(d, "PPS refers to missing SPS");
  const SPS* s = &d->sps[p.sps_id];
  p.cabac = get_bit(b);
  p.bottom_field_pic_order_present = get_bit(b);
  if (get_ue(b) != 0) return h264i_fail(d, "unsupported: slice groups (FMO)");
  p.num_ref_idx_default[0] = get_ue(b) + 1;
  p.num_ref_idx_default[1] = get_ue(b) + 1;
  p.weighted_pred = get_bit(b);
  p.weighted_bipred_idc = get_bits(b, 2);
  p.init_qp = 26 + get_se(b);
  get_se(b); /* pic_init_qs */
  p.chroma_qp_offset[0] = get_se(b);
  p.deblocking_control_present = get_bit(b);
  p.constrained_intra_pred = get_bit(b);
  p.redundant_pic_cnt_present = get_bit(b);
  p.chroma_qp_offset[1] = p.chroma_qp_offset[0];
  memcpy(p.scaling4, s->scaling4, sizeof(p.scaling4));
  memcpy(p.scaling8, s->scaling8, sizeof(p.scaling8));
  if (h264i_more_rbsp_data(b)) {
    p.transform_8x8_mode = get_bit(b);
    if (get_bit(b)) /* pic_scaling_matrix_present_flag */
    {
      parse_scaling_matrices(b, p.transform_8x8_mode, p.scaling4, p.scaling8, s->scaling4,
                             s->scaling8, s->scaling_matrix_present);
      if (s->chroma_format_idc == 3 && p.transform_8x8_mode) { /* Cb / Cr 8x8 lists: read and dropped */
        uint8_t scratch[64];
        for (int k = 0; k < 4; ++k) scaling_list(b, scratch, 64, s->scaling8[0], default_8x8_intra);
      }
    }
    p.chroma_qp_offset[1] = get_se(b);
  }
  if (b->overrun) return h264i_fail(d, "truncated PPS");
  p.valid = 1;
  d->pps[id] = p;
  return 0;
}

/* ------------------------------------------------------------------------- */
/* picture pool */
static void free_picture(Picture* p) {
  for (int i = 0; i < 3; ++i) free(p->plane[i]);
  for (int l = 0; l < 2; ++l) {
    free(p->mv[l]);
    free(p->ref_idx[l]);
    free(p->ref_uid[l]);
    free(p->ref_poc[l]);
    free(p->ref_long[l]);
  }
  free(p->mb_intra);
  memset(p, 0, sizeof(*p));
}

static int alloc_pool(H264Oracle* d, int mbw, int mbh) {
  if (d->pool_w == mbw && d->pool_h == mbh) return 0;
  for (int i = 0; i < MAX_DPB; ++i) free_picture(&d->pool[i]);
  int nmb = mbw * mbh;
  for (int i = 0; i < MAX_DPB; ++i) {
    Picture* p = &d->pool[i];
    p->stride[0] = mbw * 16;
    p->stride[1] = p->stride[2] = mbw * 8;
    p->plane[0] = (uint8_t*)calloc((size_t)p->stride[0] * mbh * 16, 1);
    p->plane[1] = (uint8_t*)calloc((size_t)p->stride[1] * mbh * 8, 1);
    p->plane[2] = (uint8_t*)calloc((size_t)p->stride[2] * mbh * 8, 1);
    for (int l = 0; l < 2; ++l) {
      p->mv[l] = (int16_t(*)[2])calloc((size_t)nmb * 16, sizeof(int16_t[2]));
      p->ref_idx[l] = (int8_t*)calloc((size_t)nmb * 4, 1);
      p->ref_uid[l] = (int32_t*)calloc((size_t)nmb * 4, sizeof(int32_t));
      p->ref_poc[l] = (int32_t*)calloc((size_t)nmb * 4, sizeof(int32_t));
      p->ref_long[l] = (uint8_t*)calloc((size_t)nmb * 4, 1);
    }
    p->mb_intra = (uint8_t*)calloc(nmb, 1);
    if (!p->plane[0] || !p->plane[1] || !p->plane[2] || !p->mb_intra) return -1;
  }
#define REALLOC(ptr, type, count)                      \
  do {                                                 \
    free(ptr);                                         \
    ptr = (type*)calloc((size_t)(count), sizeof(type)); \
    if (!ptr) return -1;                               \
  } while (0)
  REALLOC(d->mb_slice, int16_t, nmb);
  REALLOC(d->mb_kind, uint8_t, nmb);
  REALLOC(d->mb_skip, uint8_t, nmb);
  REALLOC(d->mb_direct, uint8_t, nmb);
  REALLOC(d->direct8, uint8_t, nmb * 4);
  REALLOC(d->mb_t8x8, uint8_t, nmb);
  REALLOC(d->mb_cbp, uint8_t, nmb);
  REALLOC(d->mb_cbf_dc, uint8_t, nmb);
  REALLOC(d->mb_qp, int8_t, nmb);
  REALLOC(d->mb_chroma_mode, uint8_t, nmb);
  REALLOC(d->i4mode, int8_t, nmb * 16);
  REALLOC(d->nnz, uint8_t, nmb * 24);
  REALLOC(d->nnz444, uint8_t, nmb * 32);
  REALLOC(d->mb_deblock, MbDeblock, nmb);
  REALLOC(d->mb_qpc[0], int8_t, nmb);
  REALLOC(d->mb_qpc[1], int8_t, nmb);
  free(d->mvd);
  d->mvd = (uint8_t(*)[2][16][2])calloc((size_t)nmb, sizeof(*d->mvd));
  if (!d->mvd) return -1;
  d->pool_w = mbw;
  d->pool_h = mbh;
  return 0;
}

static Picture* grab_picture(H264Oracle* d) {
  for (int i = 0; i < MAX_DPB; ++i) {
    Picture* p = &d->pool[i];
    if (!p->in_pool_use) {
      p->in_pool_use = 1;
      p->uid = ++d->next_uid;
      p->ref = 0;
      p->needed_for_output = 0;
      p->long_term_idx = -1;
      return p;
    }
  }
  return NULL;
}

static void release_unused(H264Oracle* d) {
  for (int i = 0; i < MAX_DPB; ++i) {
    Picture* p = &d->pool[i];
    if (p->in_pool_use && p != d->cur && !p->ref && !p->needed_for_output) p->in_pool_use = 0;
  }
}

/* ------------------------------------------------------------------------- */
/* output */
static void emit_picture(H264Oracle* d, Picture* p) {
  const SPS* s = d->asps;
  int w = d->mbw * 16 - s->crop_left - s->crop_right;
  int h = d->mbh * 16 - s->crop_top - s->crop_bottom;
  int cw = (w + 1) / 2, ch = (h + 1) / 2;
  if (d->outq_len == d->outq_cap) {
    d->outq_cap = d->outq_cap ? d->outq_cap * 2 : 16;
    d->outq = (OutFrame*)realloc(d->outq, sizeof(OutFrame) * d->outq_cap);
  }
  OutFrame* o = &d->outq[d->outq_len++];
  o->w = w;
  o->h = h;
  o->poc = p->poc;
  o->y = (uint8_t*)malloc((size_t)w * h + 2 * (size_t)cw * ch);
  o->u = o->y + (size_t)w * h;
  o->v = o->u + (size_t)cw * ch;
  for (int y = 0; y < h; ++y)
    memcpy(o->y + (size_t)y * w, p->plane[0] + (size_t)(y + s->crop_top) * p->stride[0] + s->crop_left, w);
  for (int y = 0; y < ch; ++y) {
    memcpy(o->u + (size_t)y * cw, p->plane[1] + (size_t)(y + s->crop_top / 2) * p->stride[1] + s->crop_left / 2, cw);
    memcpy(o->v + (size_t)y * cw, p->plane[2] + (size_t)(y + s->crop_top / 2) * p->stride[2] + s->crop_left / 2, cw);
  }
  p->needed_for_output = 0;
}

static Picture* lowest_poc_pending(H264Oracle* d, int* count) {
  Picture* best = NULL;
  int n = 0;
  for (int i = 0; i < MAX_DPB; ++i) {
    Picture* p = &d->pool[i];
    if (!p->in_pool_use || !p->needed_for_output || p == d->cur) continue;
    ++n;
    if (!best || p->poc < best->poc) best = p;
  }
  if (count) *count = n;
  return best;
}

void h264i_output_all(H264Oracle* d) {
  Picture* p;
  while ((p = lowest_poc_pending(d, NULL)) != NULL) emit_picture(d, p);
  release_unused(d);
}

static int reorder_depth(const H264Oracle* d) {
  const SPS* s = d->asps;
  if (s->poc_type == 2) return 0; /* output order == decoding order */
  if (s->bitstream_restriction && s->num_reorder_frames >= 0) return s->num_reorder_frames;
  return max_dpb_frames(s);
}

/* ------------------------------------------------------------------------- */
/* POC (8.2.1) */
static void compute_poc(H264Oracle* d, Slice* sl, Picture* cur) {
  const SPS* s = d->asps;
  int max_frame_num = 1 << s->log2_max_frame_num;
  int frame_num_offset;
  if (sl->is_idr)
    frame_num_offset = 0;
  else if (d->prev_frame_num > sl->frame_num)
    frame_num_offset = d->prev_frame_num_offset + max_frame_num;
  else
    frame_num_offset = d->prev_frame_num_offset;

  int top, bottom;
  if (s->poc_type == 0) {
    int max_lsb = 1 << s->log2_max_poc_lsb;
    if (sl->is_idr) {
      d->prev_poc_msb = 0;
      d->prev_poc_lsb = 0;
    }
    int msb;
    if (sl->poc_lsb < d->prev_poc_lsb && d->prev_poc_lsb - sl->poc_lsb >= max_lsb / 2)
      msb = d->prev_poc_msb + max_lsb;
    else if (sl->poc_lsb > d->prev_poc_lsb && sl->poc_lsb - d->prev_poc_lsb > max_lsb / 2)
      msb = d->prev_poc_msb - max_lsb;
    else
      msb = d->prev_poc_msb;
    top = msb + sl->poc_lsb;
    bottom = top + sl->delta_poc_bottom;
    if (sl->nal_ref_idc) {
      d->prev_poc_msb = msb;
      d->prev_poc_lsb = sl->poc_lsb;
    }
  } else if (s->poc_type == 1) {
    int abs_frame_num = s->num_ref_frames_in_poc_cycle ? frame_num_offset + sl->frame_num : 0;
    if (!sl->nal_ref_idc && abs_frame_num > 0) abs_frame_num--;
    int expected_delta_per_cycle = 0;
    for (int i = 0; i < s->num_ref_frames_in_poc_cycle; ++i)
      expected_delta_per_cycle += s->offset_for_ref_frame[i];
    int expected = 0;
    if (abs_frame_num > 0) {
      int cycle = (abs_frame_num - 1) / s->num_ref_frames_in_poc_cycle;
      int in_cycle = (abs_frame_num - 1) % s->num_ref_frames_in_poc_cycle;
      expected = cycle * expected_delta_per_cycle;
      for (int i = 0; i <= in_cycle; ++i) expected += s->offset_for_ref_frame[i];
    }
    if (!sl->nal_ref_idc) expected += s->offset_for_non_ref_pic;
    top = expected + sl->delta_poc[0];
    bottom = top + s->offset_for_top_to_bottom + sl->delta_poc[1];
  } else {
    int t = sl->is_idr ? 0 : 2 * (frame_num_offset + sl->frame_num) - (sl->nal_ref_idc ? 0 : 1);
    top = bottom = t;
  }
  cur->poc = imin(top, bottom);
  d->prev_frame_num_offset = frame_num_offset;
  d->prev_frame_num = sl->frame_num;
}

/* ------------------------------------------------------------------------- */
/* reference lists (8.2.4) */
static int cmp_desc_picnum(const void* a, const void* b) {
  return (*(Picture* const*)b)->pic_num - (*(Picture* const*)a)->pic_num;
}
static int cmp_asc_longidx(const void* a, const void* b) {
  return (*(Picture* const*)a)->long_term_idx - (*(Picture* const*)b)->long_term_idx;
}
static int cmp_desc_poc(const void* a, const void* b) {
  return (*(Picture* const*)b)->poc - (*(Picture* const*)a)->poc;
}
static int cmp_asc_poc(const void* a, const void* b) {
  return (*(Picture* const*)a)->poc - (*(Picture* const*)b)->poc;
}

static void init_ref_lists(H264Oracle* d, Slice* sl) {
  const SPS* s = d->asps;
  int max_frame_num = 1 << s->log2_max_frame_num;
  Picture* st[MAX_DPB];
  Picture* lt[MAX_DPB];
  int nst = 0, nlt = 0;
  for (int i = 0; i < MAX_DPB; ++i) {
    Picture* p = &d->pool[i];
    if (!p->in_pool_use || p == d->cur) continue;
    if (p->ref == 1) {
      p->frame_num_wrap = p->frame_num > sl->frame_num ? p->frame_num - max_frame_num : p->frame_num;
      p->pic_num = p->frame_num_wrap;
      st[nst++] = p;
    } else if (p->ref == 2) {
      p->pic_num = p->long_term_idx; /* LongTermPicNum */
      lt[nlt++] = p;
    }
  }
  qsort(lt, nlt, sizeof(lt[0]), cmp_asc_longidx);
  for (int l = 0; l < 2; ++l) {
    memset(sl->ref_list[l], 0, sizeof(sl->ref_list[l]));
    sl->ref_count[l] = 0;
  }
  if (sl->type == 0) {
    qsort(st, nst, sizeof(st[0]), cmp_desc_picnum);
    int n = 0;
    for (int i = 0; i < nst && n < MAX_REFS; ++i) sl->ref_list[0][n++] = st[i];
    for (int i = 0; i < nlt && n < MAX_REFS; ++i) sl->ref_list[0][n++] = lt[i];
    sl->ref_count[0] = n;
  } else if (sl->type == 1) {
    Picture* before[MAX_DPB];
    Picture* after[MAX_DPB];
    int nb = 0, na = 0;
    for (int i = 0; i < nst; ++i) {
      if (st[i]->poc < d->cur->poc)
        before[nb++] = st[i];
      else
        after[na++] = st[i];
    }
    qsort(before, nb, sizeof(before[0]), cmp_desc_poc);
    qsort(after, na, sizeof(after[0]), cmp_asc_poc);
    int n0 = 0, n1 = 0;
    for (int i = 0; i < nb; ++i) sl->ref_list[0][n0++] = before[i];
    for (int i = 0; i < na; ++i) sl->ref_list[0][n0++] = after[i];
    for (int i = 0; i < nlt; ++i) sl->ref_list[0][n0++] = lt[i];
    for (int i = 0; i < na; ++i) sl->ref_list[1][n1++] = after[i];
    for (int i = 0; i < nb; ++i) sl->ref_list[1][n1++] = before[i];
    for (int i = 0; i < nlt; ++i) sl->ref_list[1][n1++] = lt[i];
    if (n1 > 1 && n0 == n1 && !memcmp(sl->ref_list[0], sl->ref_list[1], n1 * sizeof(Picture*))) {
      Picture* t = sl->ref_list[1][0];
      sl->ref_list[1][0] = sl->ref_list[1][1];
      sl->ref_list[1][1] = t;
    }
    sl->ref_count[0] = n0;
    sl->ref_count[1] = n1;
  }
}

static int modify_ref_list(H264Oracle* d, Bits* b, Slice* sl, int list) {
  const SPS* s = d->asps;
  int max_pic_num = 1 << s->log2_max_frame_num;
  int pred = sl->frame_num; /* CurrPicNum */
  int idx = 0;
  int n = sl->num_ref_idx[list];
  Picture** L = sl->ref_list[list];
  for (;;) {
    unsigned op = get_ue(b);
    if (op == 3) break;
    if (op > 3 || b->overrun || idx >= n) return h264i_fail(d, "bad ref_pic_list_modification");
    Picture* pic = NULL;
    if (op < 2) {
      int diff = get_ue(b) + 1;
      if (op == 0) {
        pred -= diff;
        if (pred < 0) pred += max_pic_num;
      } else {
        pred += diff;
        if (pred >= max_pic_num) pred -= max_pic_num;
      }
      int pic_num = pred > sl->frame_num ? pred - max_pic_num : pred;
      for (int i = 0; i < MAX_DPB; ++i) {
        Picture* p = &d->pool[i];
        if (p->in_pool_use && p != d->cur && p->ref == 1 && p->pic_num == pic_num) pic = p;
      }
    } else {
      int lt_num = get_ue(b);
      for (int i = 0; i < MAX_DPB; ++i) {
        Picture* p = &d->pool[i];
        if (p->in_pool_use && p != d->cur && p->ref == 2 && p->long_term_idx == lt_num) pic = p;
      }
    }
    if (!pic) return h264i_fail(d, "ref_pic_list_modification names a missing picture");
    for (int c = n; c > idx; --c) L[c] = L[c - 1];
    L[idx++] = pic;
    int w = idx;
    for (int c = idx; c <= n; ++c)
      if (L[c] != pic) L[w++] = L[c];
  }
  return 0;
}

static void implicit_weights(H264Oracle* d, Slice* sl) {
  for (int i = 0; i < sl->num_ref_idx[0]; ++i) {
    for (int j = 0; j < sl->num_ref_idx[1]; ++j) {
      Picture* p0 = sl->ref_list[0][i];
      Picture* p1 = sl->ref_list[1][j];
      int w = 32;
      if (p0 && p1 && p0->ref != 2 && p1->ref != 2) {
        int tb = iclip(-128, 127, d->cur->poc - p0->poc);
        int td = iclip(-128, 127, p1->poc - p0->poc);
        if (td != 0) {
          int tx = (16384 + iabs(td / 2)) / td;
          int dsf = iclip(-1024, 1023, (tb * tx + 32) >> 6);
          if ((dsf >> 2) >= -64 && (dsf >> 2) <= 128) w = dsf >> 2;
        }
      }
      sl->implicit_weight[i][j] = w; /* w1; w0 = 64 - w1 */
    }
  }
}

/* ------------------------------------------------------------------------- */
int h264i_parse_slice_header(H264Oracle* d, Bits* b, int nal_ref_idc, int nal_type) {
  Slice* sl = &d->sl;
  memset(sl, 0, sizeof(*sl));
  sl->nal_ref_idc = nal_ref_idc;
  sl->is_idr = nal_type == 5;
  sl->first_mb = get_ue(b);
  unsigned st = get_ue(b);
  if (st > 9) return h264i_fail(d, "bad slice_type");
  st %= 5;
  if (st > 2) return h264i_fail(d, "unsupported: SP/SI slices");
  sl->type = st;
  sl->pps_id = get_ue(b);
  if (sl->pps_id >= MAX_PPS || !d->pps[sl->pps_id].valid) return h264i_fail(d, "slice refers to missing PPS");
  const PPS* pps = &d->pps[sl->pps_id];
  const SPS* sps = &d->sps[pps->sps_id];
  sl->frame_num = get_bits(b, sps->log2_max_frame_num);
  if (sl->is_idr) sl->idr_pic_id = get_ue(b);
  if (sps->poc_type == 0) {
    sl->poc_lsb = get_bits(b, sps->log2_max_poc_lsb);
    if (pps->bottom_field_pic_order_present) sl->delta_poc_bottom = get_se(b);
  } else if (sps->poc_type == 1 && !sps->delta_pic_order_always_zero) {
    sl->delta_poc[0] = get_se(b);
    if (pps->bottom_field_pic_order_present) sl->delta_poc[1] = get_se(b);
  }
  if (pps->redundant_pic_cnt_present) {
    if (get_ue(b) != 0) return 1; /* redundant picture: skip */
  }
  if (sl->type == 1) sl->direct_spatial = get_bit(b);
  sl->num_ref_idx[0] = pps->num_ref_idx_default[0];
  sl->num_ref_idx[1] = pps->num_ref_idx_default[1];
  if (sl->type != 2) {
    if (get_bit(b)) {
      sl->num_ref_idx[0] = get_ue(b) + 1;
      if (sl->type == 1) sl->num_ref_idx[1] = get_ue(b) + 1;
    }
    if (sl->num_ref_idx[0] > MAX_REFS || sl->num_ref_idx[1] > MAX_REFS)
      return h264i_fail(d, "num_ref_idx_active out of range");
  }
  if (sl->type != 1) sl->num_ref_idx[1] = 0;
  if (sl->type == 2) sl->num_ref_idx[0] = 0;

  /* A new picture starts at first_mb_in_slice == 0 (one access unit per
   * sample, no ASO). */
  if (sl->first_mb == 0 || !d->pic_in_progress) {
    if (d->pic_in_progress) {
      if (h264i_finish_picture(d) < 0) return -1;
    }
    d->apps = pps;
    d->asps = sps;
    d->chroma444 = sps->chroma_format_idc == 3;
    if (h264i_start_picture(d) < 0) return -1;
  } else if (pps->sps_id != d->apps->sps_id) {
    return h264i_fail(d, "SPS changed inside a picture");
  } else {
    d->apps = pps;
  }

  if (sl->type != 2) {
    init_ref_lists(d, sl);
    if (get_bit(b)) {
      if (modify_ref_list(d, b, sl, 0) < 0) return -1;
    }
    if (sl->type == 1 && get_bit(b)) {
      if (modify_ref_list(d, b, sl, 1) < 0) return -1;
    }
    for (int l = 0; l < 2; ++l)
      for (int i = 0; i < sl->num_ref_idx[l]; ++i)
        if (!sl->ref_list[l][i]) return h264i_fail(d, "reference list has a missing picture");
  }

  sl->use_weight = 0;
  if ((pps->weighted_pred && sl->type == 0) || (pps->weighted_bipred_idc == 1 && sl->type == 1)) {
    sl->use_weight = 1;
    sl->luma_log2_denom = get_ue(b);
    sl->chroma_log2_denom = get_ue(b);
    for (int l = 0; l < (sl->type == 1 ? 2 : 1); ++l) {
      for (int i = 0; i < sl->num_ref_idx[l]; ++i) {
        sl->luma_weight[l][i] = 1 << sl->luma_log2_denom;
        sl->luma_offset[l][i] = 0;
        if (get_bit(b)) {
          sl->luma_weight[l][i] = get_se(b);
          sl->luma_offset[l][i] = get_se(b);
        }
        for (int c = 0; c < 2; ++c) {
          sl->chroma_weight[l][i][c] = 1 << sl->chroma_log2_denom;
          sl->chroma_offset[l][i][c] = 0;
        }
        if (get_bit(b)) {
          for (int c = 0; c < 2; ++c) {
            sl->chroma_weight[l][i][c] = get_se(b);
            sl->chroma_offset[l][i][c] = get_se(b);
          }
        }
      }
    }
  } else if (pps->weighted_bipred_idc == 2 && sl->type == 1) {
    sl->use_weight = 2;
    implicit_weights(d, sl);
  }

  if (nal_ref_idc) {
    if (sl->is_idr) {
      sl->no_output_of_prior_pics = get_bit(b);
      sl->long_term_reference_flag = get_bit(b);
    } else {
      sl->adaptive_marking = get_bit(b);
      if (sl->adaptive_marking) {
        for (;;) {
          unsigned op = get_ue(b);
          if (op == 0) break;
          if (op > 6 || sl->mmco_count >= 64 || b->overrun) return h264i_fail(d, "bad MMCO");
          int a = 0, c = 0;
          if (op == 1 || op == 3) a = get_ue(b);
          if (op == 2) a = get_ue(b);
          if (op == 3 || op == 6) c = get_ue(b);
          if (op == 4) a = get_ue(b);
          sl->mmco[sl->mmco_count].op = op;
          sl->mmco[sl->mmco_count].a = a;
          sl->mmco[sl->mmco_count].b = c;
          sl->mmco_count++;
        }
      }
    }
  }
  if (pps->cabac && sl->type != 2) {
    sl->cabac_init_idc = get_ue(b);
    if (sl->cabac_init_idc > 2) return h264i_fail(d, "bad cabac_init_idc");
  }
  sl->qp = pps->init_qp + get_se(b);
  if (sl->qp < 0 || sl->qp > 51) return h264i_fail(d, "slice QP out of range");
  sl->disable_deblock = 0;
  if (pps->deblocking_control_present) {
    sl->disable_deblock = get_ue(b);
    if (sl->disable_deblock > 2) return h264i_fail(d, "bad disable_deblocking_filter_idc");
    if (sl->disable_deblock != 1) {
      sl->alpha_offset = get_se(b) * 2;
      sl->beta_offset = get_se(b) * 2;
    }
  }
  if (b->overrun) return h264i_fail(d, "truncated slice header");
  d->slice_index++;
  d->counters[4 + (sl->type == 2 ? 0 : sl->type == 0 ? 1 : 2)]++;
  return 0;
}

/* ------------------------------------------------------------------------- */
int h264i_start_picture(H264Oracle* d) {
  const SPS* s = d->asps;
  Slice* sl = &d->sl;
  d->mbw = s->mb_width;
  d->mbh = s->mb_height;
  if (alloc_pool(d, d->mbw, d->mbh) < 0) return h264i_fail(d, "out of memory");

  if (sl->is_idr) {
    /* C.4.4: an IDR empties the DPB; everything pending goes out first
     * (no_output_of_prior_pics_flag would drop them; one-frame-per-sample
     * callers need them, so they are always output). */
    d->cur = NULL;
    h264i_output_all(d);
    for (int i = 0; i < MAX_DPB; ++i) d->pool[i].ref = 0;
    release_unused(d);
    d->prev_frame_num = 0;
    d->prev_frame_num_offset = 0;
    d->max_long_term_frame_idx = -1;
  } else if (d->prev_ref_has_mmco5) {
    d->prev_frame_num = 0;
    d->prev_frame_num_offset = 0;
  }
  d->prev_ref_has_mmco5 = 0;

  release_unused(d);
  Picture* cur = grab_picture(d);
  if (!cur) {
    /* DPB overflow in a broken stream: bump one picture and retry. */
    Picture* p = lowest_poc_pending(d, NULL);
    if (p) emit_picture(d, p);
    release_unused(d);
    cur = grab_picture(d);
    if (!cur) return h264i_fail(d, "decoded picture buffer exhausted");
  }
  d->cur = cur;
  cur->frame_num = sl->frame_num;
  cur->is_idr = sl->is_idr;
  compute_poc(d, sl, cur);
  int nmb = d->mbw * d->mbh;
  for (int i = 0; i < nmb; ++i) d->mb_slice[i] = -1;
  memset(d->mb_kind, 0, nmb);
  memset(cur->mb_intra, 0, nmb);
  d->slice_index = 0;
  d->pic_in_progress = 1;
  d->counters[3]++;
  return 0;
}

static void unmark_long_idx(H264Oracle* d, int idx, Picture* except) {
  for (int i = 0; i < MAX_DPB; ++i) {
    Picture* p = &d->pool[i];
    if (p->in_pool_use && p != except && p->ref == 2 && p->long_term_idx == idx) p->ref = 0;
  }
}

static void mark_references(H264Oracle* d) {
  Slice* sl = &d->sl;
  Picture* cur = d->cur;
  const SPS* s = d->asps;
  if (!sl->nal_ref_idc) return;
  if (sl->is_idr) {
    if (sl->long_term_reference_flag) {
      cur->ref = 2;
      cur->long_term_idx = 0;
      d->max_long_term_frame_idx = 0;
    } else {
      cur->ref = 1;
      d->max_long_term_frame_idx = -1;
    }
    return;
  }
  int cur_is_long = 0;
  if (sl->adaptive_marking) {
    for (int k = 0; k < sl->mmco_count; ++k) {
      int op = sl->mmco[k].op, a = sl->mmco[k].a, bb = sl->mmco[k].b;
      if (op == 1 || op == 3) {
        int pic_num = sl->frame_num - (a + 1);
        for (int i = 0; i < MAX_DPB; ++i) {
          Picture* p = &d->pool[i];
          if (!p->in_pool_use || p == cur || p->ref != 1 || p->pic_num != pic_num) continue;
          if (op == 1) {
            p->ref = 0;
          } else {
            unmark_long_idx(d, bb, p);
            p->ref = 2;
            p->long_term_idx = bb;
          }
        }
      } else if (op == 2) {
        for (int i = 0; i < MAX_DPB; ++i) {
          Picture* p = &d->pool[i];
          if (p->in_pool_use && p != cur && p->ref == 2 && p->long_term_idx == a) p->ref = 0;
        }
      } else if (op == 4) {
        d->max_long_term_frame_idx = a - 1;
        for (int i = 0; i < MAX_DPB; ++i) {
          Picture* p = &d->pool[i];
          if (p->in_pool_use && p != cur && p->ref == 2 && p->long_term_idx > a - 1) p->ref = 0;
        }
      } else if (op == 5) {
        for (int i = 0; i < MAX_DPB; ++i)
          if (&d->pool[i] != cur) d->pool[i].ref = 0;
        d->max_long_term_frame_idx = -1;
        d->prev_ref_has_mmco5 = 1;
        /* 8.2.1: after MMCO5 the picture is treated as frame_num 0 and its
         * POC is rebased to 0 for everything that follows. */
        d->prev_poc_msb = 0;
        d->prev_poc_lsb = 0; /* frame: tempPicOrderCnt -> TopFieldOrderCnt = 0 */
        cur->frame_num = 0;
      } else if (op == 6) {
        unmark_long_idx(d, a, cur);
        cur->ref = 2;
        cur->long_term_idx = a;
        cur_is_long = 1;
      }
    }
  } else {
    /* sliding window */
    int n = 0;
    Picture* oldest = NULL;
    for (int i = 0; i < MAX_DPB; ++i) {
      Picture* p = &d->pool[i];
      if (!p->in_pool_use || p == cur || !p->ref) continue;
      ++n;
      if (p->ref == 1 && (!oldest || p->frame_num_wrap < oldest->frame_num_wrap)) oldest = p;
    }
    if (n >= imax(s->max_num_ref_frames, 1) && oldest) oldest->ref = 0;
  }
  if (!cur_is_long) cur->ref = 1;
}

int h264i_finish_picture(H264Oracle* d) {
  if (!d->pic_in_progress) return 0;
  d->pic_in_progress = 0;
  Picture* cur = d->cur;
  /* every macroblock must have been decoded */
  int nmb = d->mbw * d->mbh;
  for (int i = 0; i < nmb; ++i)
    if (d->mb_slice[i] < 0) return h264i_fail(d, "picture ended with undecoded macroblocks");

  if (!d->parse_only) h264i_deblock_picture(d);

  /* frame_num_wrap of the current refs is needed by the sliding window */
  {
    int max_frame_num = 1 << d->asps->log2_max_frame_num;
    for (int i = 0; i < MAX_DPB; ++i) {
      Picture* p = &d->pool[i];
      if (p->in_pool_use && p != cur && p->ref == 1) {
        p->frame_num_wrap = p->frame_num > d->sl.frame_num ? p->frame_num - max_frame_num : p->frame_num;
        p->pic_num = p->frame_num_wrap;
      }
    }
  }
  mark_references(d);
  cur->needed_for_output = 1;
  if (d->prev_ref_has_mmco5) {
    /* MMCO5 acts like an IDR for output: flush everything before it, then
     * rebase its POC. */
    d->cur = NULL;
    cur->needed_for_output = 0;
    h264i_output_all(d);
    cur->needed_for_output = 1;
    cur->poc = 0;
  }
  d->cur = NULL;

  /* bumping: keep at most reorder_depth pictures waiting */
  int depth = reorder_depth(d);
  for (;;) {
    int n = 0;
    Picture* p = lowest_poc_pending(d, &n);
    if (!p || n <= depth) break;
    emit_picture(d, p);
  }
  release_unused(d);
  return 0;
}

/* ------------------------------------------------------------------------- */
/* public API */
H264Oracle* h264o_open(void) {
  H264Oracle* d = (H264Oracle*)calloc(1, sizeof(H264Oracle));
  if (d) {
    d->nal_length_size = 4;
    d->max_long_term_frame_idx = -1;
    d->x264_build = -1;
  }
  return d;
}

void h264o_close(H264Oracle* d) {
  if (!d) return;
  for (int i = 0; i < MAX_DPB; ++i) free_picture(&d->pool[i]);
  for (int i = 0; i < d->outq_len; ++i) free(d->outq[i].y);
  free(d->outq);
  free(d->rbsp);
  free(d->mb_slice);
  free(d->mb_kind);
  free(d->mb_skip);
  free(d->mb_direct);
  free(d->direct8);
  free(d->mb_t8x8);
  free(d->mb_cbp);
  free(d->mb_cbf_dc);
  free(d->mb_qp);
  free(d->mb_chroma_mode);
  free(d->i4mode);
  free(d->nnz);
  free(d->nnz444);
  free(d->mvd);
  free(d->mb_deblock);
  free(d->mb_qpc[0]);
  free(d->mb_qpc[1]);
  free(d->cov);
  free(d);
}

const char* h264o_error(H264Oracle* d) { return d->error; }

static int decode_nal(H264Oracle* d, const uint8_t* nal, size_t n) {
  if (n < 1) return 0;
  int ref_idc = (nal[0] >> 5) & 3;
  int type = nal[0] & 31;
  if (type == 6 && d->parse_only) {
    /* x264's version SEI (user_data_unregistered: a UUID and "x264 - core N ..."): 4:4:4 streams of builds before
     * 151 code the coded_block_flag of 8x8 blocks with contexts of their own making, see h264_slice.c */
    static const char tag[] = "x264 - core ";
    for (size_t i = 1; i + sizeof(tag) + 2 < n; ++i)
      if (!memcmp(nal + i, tag, sizeof(tag) - 1)) {
        int v = 0;
        for (size_t k = i + sizeof(tag) - 1; k < n && nal[k] >= '0' && nal[k] <= '9'; ++k) v = v * 10 + (nal[k] - '0');
        if (!d->no_quirks) d->x264_build = v;
        break;
      }
    return 0;
  }
  if (type != 1 && type != 5 && type != 7 && type != 8) return 0; /* SEI, AUD, ... */
  int rn = unescape_nal(d, nal + 1, n - 1);
  if (rn < 0) return h264i_fail(d, "out of memory");
  Bits b = {d->rbsp, rn * 8, 0, 0};
  if (type == 7) return h264i_parse_sps(d, &b);
  if (type == 8) return h264i_parse_pps(d, &b);
  int r = h264i_parse_slice_header(d, &b, ref_idc, type);
  if (r != 0) return r < 0 ? r : 0;
  return h264i_decode_slice_data(d, &b);
}

int h264o_configure(H264Oracle* d, const uint8_t* avcc, size_t size) {
  d->error[0] = 0;
  if (size < 7 || avcc[0] != 1) return h264i_fail(d, "extradata is not an avcC record");
  d->nal_length_size = (avcc[4] & 3) + 1;
  size_t pos = 5;
  for (int pass = 0; pass < 2; ++pass) {
    if (pos >= size) return h264i_fail(d, "truncated avcC");
    int count = pass == 0 ? (avcc[pos] & 31) : avcc[pos];
    pos++;
    for (int i = 0; i < count; ++i) {
      if (pos + 2 > size) return h264i_fail(d, "truncated avcC");
      size_t len = (avcc[pos] << 8) | avcc[pos + 1];
      pos += 2;
      if (pos + len > size) return h264i_fail(d, "truncated avcC");
      if (decode_nal(d, avcc + pos, len) < 0) return -1;
      pos += len;
    }
  }
  return 0;
}

int h264o_decode(H264Oracle* d, const uint8_t* sample, size_t size) {
  d->error[0] = 0;
  size_t pos = 0;
  const int ls = d->nal_length_size;
  while (pos + ls <= size) {
    size_t len = 0;
    for (int i = 0; i < ls; ++i) len = (len << 8) | sample[pos + i];
    pos += ls;
    if (len > size - pos) return h264i_fail(d, "NAL length exceeds sample");
    if (decode_nal(d, sample + pos, len) < 0) {
      d->pic_in_progress = 0;
      if (d->cur) {
        d->cur->in_pool_use = 0;
        d->cur = NULL;
      }
      return -1;
    }
    pos += len;
  }
  /* one sample == one access unit: the picture is complete */
  return h264i_finish_picture(d);
}

int h264o_flush(H264Oracle* d) {
  if (d->pic_in_progress && h264i_finish_picture(d) < 0) return -1;
  if (!d->asps) return 0;
  d->cur = NULL;
  h264i_output_all(d);
  for (int i = 0; i < MAX_DPB; ++i) d->pool[i].ref = 0;
  release_unused(d);
  return 0;
}

int h264o_frames_ready(H264Oracle* d) { return d->outq_len; }

int h264o_frame_info(H264Oracle* d, int* width, int* height, int* poc) {
  if (!d->outq_len) return -1;
  if (width) *width = d->outq[0].w;
  if (height) *height = d->outq[0].h;
  if (poc) *poc = d->outq[0].poc;
  return 0;
}

int h264o_pop_frame(H264Oracle* d, uint8_t* dst, size_t capacity) {
  if (!d->outq_len) return -1;
  OutFrame* o = &d->outq[0];
  size_t need = (size_t)o->w * o->h + 2 * (size_t)((o->w + 1) / 2) * ((o->h + 1) / 2);
  int rc = 0;
  if (dst) {
    if (capacity < need)
      rc = -1;
    else
      memcpy(dst, o->y, need);
  }
  free(o->y);
  memmove(d->outq, d->outq + 1, sizeof(OutFrame) * (d->outq_len - 1));
  d->outq_len--;
  return rc;
}

void h264o_counters(H264Oracle* d, int64_t out[10]) { memcpy(out, d->counters, sizeof(d->counters)); }

/* on: 1 = parse only; 2 = parse only and take every stream at its word (no emulation of old x264 builds' 4:4:4
 * contexts, h264_slice.c cbf8_inc): the negative control of the desync test */
void h264o_set_parse_only(H264Oracle* d, int on) {
  d->parse_only = on != 0;
  d->no_quirks = on == 2;
}

void h264o_coverage_enable(H264Oracle* d) {
  if (!d->cov) d->cov = (int64_t*)calloc(H264O_COV_CELLS, sizeof(int64_t));
}
int h264o_coverage(H264Oracle* d, int64_t* out, int n) {
  if (n > H264O_COV_CELLS) n = H264O_COV_CELLS;
  if (d->cov) memcpy(out, d->cov, (size_t)n * sizeof(int64_t));
  else memset(out, 0, (size_t)n * sizeof(int64_t));
  return H264O_COV_CELLS;
}

void h264o_yuv420_to_rgb24(const uint8_t* i420, int width, int height, uint8_t* rgb) {
  const int cw = (width + 1) / 2, ch = (height + 1) / 2;
  const uint8_t* yp = i420;
  const uint8_t* up = yp + (size_t)width * height;
  const uint8_t* vp = up + (size_t)cw * ch;
  for (int y = 0; y < height; ++y) {
    for (int x = 0; x < width; ++x) {
      int Y = yp[(size_t)y * width + x];
      int U = up[(size_t)(y >> 1) * cw + (x >> 1)];
      int V = vp[(size_t)(y >> 1) * cw + (x >> 1)];
      int l = (((Y << 3) - 128) * 9539) >> 16;
      int us = (U << 3) - 1024, vs = (V << 3) - 1024;
      int r = l + ((vs * 13075) >> 16);
      int g = l + ((us * -3209) >> 16) + ((vs * -6660) >> 16);
      int bl = l + ((us * 16525) >> 16);
      uint8_t* o = rgb + ((size_t)y * width + x) * 3;
      o[0] = (uint8_t)clip8(r);
      o[1] = (uint8_t)clip8(g);
      o[2] = (uint8_t)clip8(bl);
    }
  }
}
