/* TEST INFRASTRUCTURE -- oracle: slice_data() / macroblock_layer() parsing
 * with both entropy coders (9.2 CAVLC, 9.3 CABAC), intra mode and motion
 * vector derivation (8.3.1.1, 8.4.1).  Restates FFmpeg's h264_cabac.c /
 * h264_cavlc.c / h264_mvpred.h / h264_direct.c as used behind
 * software_video_decoder.cpp:353-357 (avcodec_send_packet). */
#include "h264_internal.h"

#include <stdio.h>
#include <stdlib.h>

/* ======================================================================== */
/* CABAC engine (9.3.1.2, 9.3.3.2)                                          */
/* ======================================================================== */
static void cabac_start(Cabac* c, Bits* b) {
  c->bits = b;
  c->range = 510;
  c->offset = get_bits(b, 9);
}

static void cabac_init_contexts(Cabac* c, int slice_type, int idc, int qp) {
  const int t = slice_type == 2 ? 0 : 1 + idc;
  qp = iclip(0, 51, qp);
  for (int i = 0; i < CABAC_CTX_COUNT; ++i) {
    int m = cabac_init_mn[t][i][0], n = cabac_init_mn[t][i][1];
    int pre = iclip(1, 126, ((m * qp) >> 4) + n);
    if (pre <= 63)
      c->state[i] = (uint8_t)((63 - pre) << 1);
    else
      c->state[i] = (uint8_t)(((pre - 64) << 1) | 1);
  }
  /* ctxIdx 460 .. 1023 (4:4:4 only; frame macroblocks).  Tables 9-12 .. 9-33 give the contexts of Cb (ctxBlockCat
   * 6 .. 9) and Cr (10 .. 13) the same (m, n) as the luma contexts of the same role (ctxBlockCat 0 .. 2, 5), and the
   * coded_block_flag contexts of 8x8 blocks (1012 .. 1023, one set per colour component) those of ctxBlockCat 2 -
   * typed here as that rule, not as 564 more pairs; the field-macroblock ranges stay unused.  Whether the rule is
   * right is what the desync check on a real 4:4:4 stream decides (tests/test_oracle_realstream.py). */
  static const struct { int dst, src, n; } copies[] = {
      {460, 85, 12},  {472, 85, 12},                       /* coded_block_flag: Cb, Cr cat 6-8 / 10-12 <- luma cat 0-2 */
      {484, 105, 44}, {528, 105, 44},                      /* significant_coeff_flag (frame) */
      {572, 166, 44}, {616, 166, 44},                      /* last_significant_coeff_flag (frame) */
      {660, 402, 15}, {690, 417, 9},  {708, 426, 10},      /* Cb 8x8 (cat 9): sig, last, abs <- luma 8x8 (cat 5) */
      {718, 402, 15}, {748, 417, 9},  {766, 426, 10},      /* Cr 8x8 (cat 13) */
      {952, 227, 30}, {982, 227, 30},                      /* coeff_abs_level_minus1 */
      {1012, 93, 4},  {1016, 93, 4},  {1020, 93, 4}};      /* coded_block_flag of 8x8 blocks: Y, Cb, Cr */
  for (size_t k = 0; k < sizeof(copies) / sizeof(copies[0]); ++k)
    memcpy(c->state + copies[k].dst, c->state + copies[k].src, (size_t)copies[k].n);
}

static int cabac_bin(Cabac* c, int ctx) {
  int p = c->state[ctx] >> 1, mps = c->state[ctx] & 1;
  unsigned lps = cabac_range_lps[p][(c->range >> 6) & 3];
  int bin;
  c->range -= lps;
  if (c->offset >= c->range) {
    bin = !mps;
    c->offset -= c->range;
    c->range = lps;
    if (p == 0) mps = !mps;
    p = cabac_trans_lps[p];
  } else {
    bin = mps;
    if (p < 62) ++p;
  }
  c->state[ctx] = (uint8_t)((p << 1) | mps);
  while (c->range < 256) {
    c->range <<= 1;
    c->offset = (c->offset << 1) | get_bit(c->bits);
  }
  return bin;
}

static int cabac_bypass(Cabac* c) {
  c->offset = (c->offset << 1) | get_bit(c->bits);
  if (c->offset >= c->range) {
    c->offset -= c->range;
    return 1;
  }
  return 0;
}

static int cabac_terminate(Cabac* c) {
  c->range -= 2;
  if (c->offset >= c->range) return 1;
  while (c->range < 256) {
    c->range <<= 1;
    c->offset = (c->offset << 1) | get_bit(c->bits);
  }
  return 0;
}

/* ======================================================================== */
/* bit writer + CABAC encoder (9.3.4.2) -- corpus generation only           */
/* ======================================================================== */
void bw_put(BitWriter* w, unsigned v, int n) {
  for (int i = n - 1; i >= 0; --i) {
    if ((w->pos >> 3) >= w->cap) {
      w->cap = w->cap ? w->cap * 2 : 1 << 16;
      w->buf = (uint8_t*)realloc(w->buf, w->cap);
    }
    if ((w->pos & 7) == 0) w->buf[w->pos >> 3] = 0;
    if ((v >> i) & 1) w->buf[w->pos >> 3] |= (uint8_t)(0x80 >> (w->pos & 7));
    w->pos++;
  }
}
void bw_ue(BitWriter* w, unsigned v) {
  unsigned x = v + 1;
  int len = 0;
  while ((x >> len) > 1) ++len;
  bw_put(w, 0, len);
  bw_put(w, x, len + 1);
}
void bw_se(BitWriter* w, int v) { bw_ue(w, v > 0 ? 2u * v - 1 : 2u * (unsigned)(-v)); }

static void cenc_start(CabacEnc* e, BitWriter* w) {
  e->w = w;
  e->low = 0;
  e->range = 510;
  e->outstanding = 0;
  e->first = 1;
}
static void cenc_put(CabacEnc* e, int b) {
  if (e->first)
    e->first = 0;
  else
    bw_put(e->w, b, 1);
  while (e->outstanding > 0) {
    bw_put(e->w, !b, 1);
    e->outstanding--;
  }
}
static void cenc_renorm(CabacEnc* e) {
  while (e->range < 256) {
    if (e->low < 256) {
      cenc_put(e, 0);
    } else if (e->low >= 512) {
      e->low -= 512;
      cenc_put(e, 1);
    } else {
      e->low -= 256;
      e->outstanding++;
    }
    e->range <<= 1;
    e->low <<= 1;
  }
}
static int cenc_bin(CabacEnc* e, uint8_t* state, int bin) {
  int p = *state >> 1, mps = *state & 1;
  unsigned lps = cabac_range_lps[p][(e->range >> 6) & 3];
  e->range -= lps;
  if (bin != mps) {
    e->low += e->range;
    e->range = lps;
    if (p == 0) mps = !mps;
    p = cabac_trans_lps[p];
  } else if (p < 62) {
    ++p;
  }
  *state = (uint8_t)((p << 1) | mps);
  cenc_renorm(e);
  return bin;
}
static int cenc_bypass(CabacEnc* e, int bin) {
  e->low <<= 1;
  if (bin) e->low += e->range;
  if (e->low >= 1024) {
    cenc_put(e, 1);
    e->low -= 1024;
  } else if (e->low < 512) {
    cenc_put(e, 0);
  } else {
    e->low -= 512;
    e->outstanding++;
  }
  return bin;
}
static int cenc_terminate(CabacEnc* e, int bin) {
  e->range -= 2;
  if (bin) {
    e->low += e->range;
    e->range = 2;
    cenc_renorm(e);
    cenc_put(e, (e->low >> 9) & 1);
    bw_put(e->w, ((e->low >> 7) & 3) | 1, 2);
  } else {
    cenc_renorm(e);
  }
  return bin;
}

/* ======================================================================== */
/* neighbour helpers                                                        */
/* ======================================================================== */
static inline int mb_avail(const H264Oracle* d, int mbx, int mby) {
  if (mbx < 0 || mby < 0 || mbx >= d->mbw || mby >= d->mbh) return 0;
  return d->mb_slice[mby * d->mbw + mbx] == d->slice_index;
}

typedef struct {
  int ref; /* -2 partition not available, -1 list not used / intra, >= 0 refIdx */
  int mv[2];
} NB;

/* (rx, ry): 4x4-block coordinates relative to the current macroblock,
 * -1..4.  cur_zz: luma4x4BlkIdx of the predicting partition's first block
 * (blocks of the current MB later than that in decoding order are "not yet
 * decoded", 6.4.11.7). */
static NB fetch_nb(const H264Oracle* d, int list, int rx, int ry, int cur_zz) {
  NB nb = {-2, {0, 0}};
  if (rx > 3 && ry >= 0) return nb;
  const int ox = rx < 0 ? -1 : (rx > 3 ? 1 : 0), oy = ry < 0 ? -1 : 0;
  const int nx = d->mb.mbx + ox, ny = d->mb.mby + oy;
  const int bx = rx & 3, by = ry & 3;
  if (ox == 0 && oy == 0) {
    if (raster_to_blk4[by * 4 + bx] >= cur_zz) return nb;
  } else if (!mb_avail(d, nx, ny)) {
    return nb;
  }
  const int addr = ny * d->mbw + nx;
  const Picture* cur = d->cur;
  nb.ref = -1;
  if (cur->mb_intra[addr]) return nb;
  int r = cur->ref_idx[list][addr * 4 + (by >> 1) * 2 + (bx >> 1)];
  if (r < 0) return nb;
  nb.ref = r;
  nb.mv[0] = cur->mv[list][addr * 16 + by * 4 + bx][0];
  nb.mv[1] = cur->mv[list][addr * 16 + by * 4 + bx][1];
  return nb;
}

enum { PRED_MEDIAN = 0, PRED_16x8_TOP, PRED_16x8_BOTTOM, PRED_8x16_LEFT, PRED_8x16_RIGHT };

/* 8.4.1.3 */
static void pred_mv(const H264Oracle* d, int list, int bx, int by, int bw, int ref, int dir,
                    int out[2]) {
  const int zz = raster_to_blk4[by * 4 + bx];
  NB A = fetch_nb(d, list, bx - 1, by, zz);
  NB B = fetch_nb(d, list, bx, by - 1, zz);
  NB C = fetch_nb(d, list, bx + bw, by - 1, zz);
  if (C.ref == -2) C = fetch_nb(d, list, bx - 1, by - 1, zz);
  const NB* pick = NULL;
  if (dir == PRED_16x8_TOP && B.ref == ref) pick = &B;
  if (dir == PRED_16x8_BOTTOM && A.ref == ref) pick = &A;
  if (dir == PRED_8x16_LEFT && A.ref == ref) pick = &A;
  if (dir == PRED_8x16_RIGHT && C.ref == ref) pick = &C;
  if (!pick) {
    int match = (A.ref == ref) + (B.ref == ref) + (C.ref == ref);
    if (match == 1) {
      pick = A.ref == ref ? &A : (B.ref == ref ? &B : &C);
    } else if (match == 0 && B.ref == -2 && C.ref == -2 && A.ref != -2) {
      pick = &A;
    }
  }
  if (pick) {
    out[0] = pick->mv[0];
    out[1] = pick->mv[1];
  } else {
    out[0] = median3(A.mv[0], B.mv[0], C.mv[0]);
    out[1] = median3(A.mv[1], B.mv[1], C.mv[1]);
  }
}

static void set_mv_block(H264Oracle* d, int list, int bx, int by, int bw, int bh, int mvx, int mvy) {
  const int base = d->mb.addr * 16;
  for (int y = by; y < by + bh; ++y)
    for (int x = bx; x < bx + bw; ++x) {
      d->cur->mv[list][base + y * 4 + x][0] = (int16_t)mvx;
      d->cur->mv[list][base + y * 4 + x][1] = (int16_t)mvy;
      d->mb.mv[list][y * 4 + x][0] = (int16_t)mvx;
      d->mb.mv[list][y * 4 + x][1] = (int16_t)mvy;
    }
}

static void set_mvd_block(H264Oracle* d, int list, int bx, int by, int bw, int bh, int dx, int dy) {
  uint8_t ax = (uint8_t)imin(iabs(dx), 255), ay = (uint8_t)imin(iabs(dy), 255);
  for (int y = by; y < by + bh; ++y)
    for (int x = bx; x < bx + bw; ++x) {
      d->mvd[d->mb.addr][list][y * 4 + x][0] = ax;
      d->mvd[d->mb.addr][list][y * 4 + x][1] = ay;
    }
}

static void set_ref8(H264Oracle* d, int list, int b8, int ref) {
  Picture* cur = d->cur;
  const int i = d->mb.addr * 4 + b8;
  d->mb.ref[list][b8] = (int8_t)ref;
  cur->ref_idx[list][i] = (int8_t)ref;
  if (ref >= 0 && ref < d->sl.num_ref_idx[list] && d->sl.ref_list[list][ref]) {
    const Picture* r = d->sl.ref_list[list][ref];
    cur->ref_uid[list][i] = r->uid;
    cur->ref_poc[list][i] = r->poc;
    cur->ref_long[list][i] = r->ref == 2;
  } else {
    cur->ref_uid[list][i] = -1;
    cur->ref_poc[list][i] = 0;
    cur->ref_long[list][i] = 0;
  }
}

/* ======================================================================== */
/* direct prediction (8.4.1.2)                                              */
/* ======================================================================== */
/* Fills mv/ref of 8x8 block b8 (or all four when b8 < 0). */
static int pred_direct(H264Oracle* d, int only_b8) {
  Slice* sl = &d->sl;
  MB* m = &d->mb;
  const Picture* col = sl->ref_list[1][0];
  if (!col) return h264i_fail(d, "direct prediction without RefPicList1[0]");
  const int inf8 = d->asps->direct_8x8_inference;
  const int addr = m->addr;
  COV(d, H264O_COV_DIRECT + ((sl->direct_spatial ? 0 : 1) * 2 + (only_b8 >= 0)) * 2 + (inf8 != 0));

  if (sl->direct_spatial) {
    int ref[2], mvp[2][2] = {{0, 0}, {0, 0}};
    for (int l = 0; l < 2; ++l) {
      NB A = fetch_nb(d, l, -1, 0, 0);
      NB B = fetch_nb(d, l, 0, -1, 0);
      NB C = fetch_nb(d, l, 4, -1, 0);
      if (C.ref == -2) C = fetch_nb(d, l, -1, -1, 0);
      int r = -1;
      if (A.ref >= 0) r = A.ref;
      if (B.ref >= 0 && (r < 0 || B.ref < r)) r = B.ref;
      if (C.ref >= 0 && (r < 0 || C.ref < r)) r = C.ref;
      ref[l] = r;
      if (r >= 0) pred_mv(d, l, 0, 0, 4, r, PRED_MEDIAN, mvp[l]);
    }
    if (ref[0] < 0 && ref[1] < 0) ref[0] = ref[1] = 0; /* directZeroPrediction */
    const int col_short = col->ref == 1;
    for (int b8 = 0; b8 < 4; ++b8) {
      if (only_b8 >= 0 && b8 != only_b8) continue;
      for (int l = 0; l < 2; ++l) set_ref8(d, l, b8, ref[l]);
      for (int s = 0; s < 4; ++s) {
        const int bx = (b8 & 1) * 2 + (s & 1), by = (b8 >> 1) * 2 + (s >> 1);
        /* co-located 4x4: the corner block under direct_8x8_inference */
        const int cx = inf8 ? (b8 & 1) * 3 : bx, cy = inf8 ? (b8 >> 1) * 3 : by;
        int col_zero = 0;
        if (col_short && !col->mb_intra[addr]) {
          const int c8 = addr * 4 + (cy >> 1) * 2 + (cx >> 1);
          int cl = col->ref_idx[0][c8] >= 0 ? 0 : 1;
          if (col->ref_idx[cl][c8] == 0) {
            const int16_t* cm = col->mv[cl][addr * 16 + cy * 4 + cx];
            col_zero = cm[0] >= -1 && cm[0] <= 1 && cm[1] >= -1 && cm[1] <= 1;
          }
        }
        for (int l = 0; l < 2; ++l) {
          int zero = ref[l] < 0 || (ref[l] == 0 && col_zero);
          set_mv_block(d, l, bx, by, 1, 1, zero ? 0 : mvp[l][0], zero ? 0 : mvp[l][1]);
        }
      }
    }
    return 0;
  }

  /* temporal direct (8.4.1.2.3) */
  for (int b8 = 0; b8 < 4; ++b8) {
    if (only_b8 >= 0 && b8 != only_b8) continue;
    for (int s = 0; s < 4; ++s) {
      const int bx = (b8 & 1) * 2 + (s & 1), by = (b8 >> 1) * 2 + (s >> 1);
      const int cx = inf8 ? (b8 & 1) * 3 : bx, cy = inf8 ? (b8 >> 1) * 3 : by;
      int ref0 = 0, mvc[2] = {0, 0};
      if (!col->mb_intra[addr]) {
        const int c8 = addr * 4 + (cy >> 1) * 2 + (cx >> 1);
        int cl = col->ref_idx[0][c8] >= 0 ? 0 : 1;
        if (col->ref_idx[cl][c8] >= 0) {
          mvc[0] = col->mv[cl][addr * 16 + cy * 4 + cx][0];
          mvc[1] = col->mv[cl][addr * 16 + cy * 4 + cx][1];
          int uid = col->ref_uid[cl][c8];
          ref0 = -1;
          for (int i = 0; i < sl->num_ref_idx[0]; ++i)
            if (sl->ref_list[0][i] && sl->ref_list[0][i]->uid == uid) {
              ref0 = i;
              break;
            }
          if (ref0 < 0) return h264i_fail(d, "temporal direct: co-located reference not in RefPicList0");
        }
      }
      const Picture* p0 = sl->ref_list[0][ref0];
      int mv0[2], mv1[2];
      int td = iclip(-128, 127, col->poc - p0->poc);
      if (p0->ref == 2 || td == 0) {
        mv0[0] = mvc[0];
        mv0[1] = mvc[1];
        mv1[0] = mv1[1] = 0;
      } else {
        int tb = iclip(-128, 127, d->cur->poc - p0->poc);
        int tx = (16384 + iabs(td / 2)) / td;
        int dsf = iclip(-1024, 1023, (tb * tx + 32) >> 6);
        mv0[0] = (dsf * mvc[0] + 128) >> 8;
        mv0[1] = (dsf * mvc[1] + 128) >> 8;
        mv1[0] = mv0[0] - mvc[0];
        mv1[1] = mv0[1] - mvc[1];
      }
      if (s == 0 || !inf8) {
        /* refs are per 8x8; without 8x8 inference every 4x4 of the 8x8 maps
         * to the same picture in a frame-coded co-located MB partition */
      }
      if (s == 0) {
        set_ref8(d, 0, b8, ref0);
        set_ref8(d, 1, b8, 0);
      }
      set_mv_block(d, 0, bx, by, 1, 1, mv0[0], mv0[1]);
      set_mv_block(d, 1, bx, by, 1, 1, mv1[0], mv1[1]);
    }
  }
  return 0;
}

/* ======================================================================== */
/* syntax element I/O: one code path for parsing (oracle decoder) and for    */
/* writing (corpus encoder).  Every sx_* takes the value to write; when     */
/* parsing the argument is ignored and the parsed value is returned.        */
/* ======================================================================== */
typedef struct {
  H264Oracle* d;
  Bits* b;        /* parse side */
  int cabac;
  int enc;        /* 1 = writing */
  BitWriter* w;   /* write side */
  const MbPlan* plan;
} Rd;

static inline int sx_bin(Rd* r, int ctx, int v) {
  const int bin = r->enc ? cenc_bin(&r->d->cenc, &r->d->cabac.state[ctx], v) : cabac_bin(&r->d->cabac, ctx);
  if (ctx < CABAC_CTX_COUNT) COV(r->d, H264O_COV_CTX + (r->d->cov_ctx_table * CABAC_CTX_COUNT + ctx) * 2 + (bin != 0));
  return bin;
}
static inline int sx_bypass(Rd* r, int v) {
  return r->enc ? cenc_bypass(&r->d->cenc, v) : cabac_bypass(&r->d->cabac);
}
static inline int sx_terminate(Rd* r, int v) {
  return r->enc ? cenc_terminate(&r->d->cenc, v) : cabac_terminate(&r->d->cabac);
}
static inline unsigned sx_u(Rd* r, int n, unsigned v) {
  if (!r->enc) return get_bits(r->b, n);
  bw_put(r->w, v, n);
  return v;
}
static inline unsigned sx_ue(Rd* r, unsigned v) {
  if (!r->enc) return get_ue(r->b);
  bw_ue(r->w, v);
  return v;
}
static inline int sx_se(Rd* r, int v) {
  if (!r->enc) return get_se(r->b);
  bw_se(r->w, v);
  return v;
}

/* ======================================================================== */
/* residual: CAVLC (9.2)                                                    */
/* ======================================================================== */
static int read_vlc(Bits* b, const uint8_t* len, const uint8_t* bits, int n) {
  /* tables are small: linear search on (len, code) */
  for (int i = 0; i < n; ++i) {
    if (!len[i]) continue;
    if (bits_left(b) >= len[i] && show_bits(b, len[i]) == bits[i]) {
      b->pos += len[i];
      return i;
    }
  }
  b->overrun = 1;
  return -1;
}
static int sx_vlc(Rd* r, const uint8_t* len, const uint8_t* bits, int n, int v) {
  if (!r->enc) return read_vlc(r->b, len, bits, n);
  bw_put(r->w, bits[v], len[v]);
  return v;
}

/* nC from the total_coeff of the left/top neighbours of block (rx, ry) in
 * plane-local 4x4 units; plane: 0 luma (4 blocks per MB row), 1/2 chroma. */
static int cavlc_nc(const H264Oracle* d, int plane, int rx, int ry) {
  const int n = plane == 0 ? 4 : 2;
  const int base = plane == 0 ? 0 : (plane == 1 ? 16 : 20);
  int na = -1, nb = -1;
  if (rx > 0) {
    na = d->mb.nnz[base + ry * n + rx - 1];
  } else if (mb_avail(d, d->mb.mbx - 1, d->mb.mby)) {
    na = d->nnz[(d->mb.addr - 1) * 24 + base + ry * n + n - 1];
  }
  if (ry > 0) {
    nb = d->mb.nnz[base + (ry - 1) * n + rx];
  } else if (mb_avail(d, d->mb.mbx, d->mb.mby - 1)) {
    nb = d->nnz[(d->mb.addr - d->mbw) * 24 + base + (n - 1) * n + rx];
  }
  if (na >= 0 && nb >= 0) return (na + nb + 1) >> 1;
  if (na >= 0) return na;
  if (nb >= 0) return nb;
  return 0;
}

/* One residual block; coefficients live at io[scan[start + k]].  Parsing
 * fills them, writing emits them.  Returns total_coeff or -1.  nc == -1
 * selects the chroma DC tables. */
static int cavlc_block(Rd* r, int nc, int max_coeff, const uint8_t* scan, int start, int16_t* io) {
  H264Oracle* d = r->d;
  Bits* b = r->b;
  int level[16], run[16];
  int total = 0, t1 = 0, zeros_total = 0;
  if (r->enc) {
    /* gather levels from the highest scan position down */
    int last = -1;
    for (int k = max_coeff - 1; k >= 0; --k) {
      int v = io[scan[start + k]];
      if (!v) continue;
      if (last < 0) {
        last = k;
      } else {
        run[total - 1] = last - k - 1;
        last = k;
      }
      level[total++] = v;
    }
    if (total) {
      run[total - 1] = last; /* zeros before the lowest coefficient */
      zeros_total = 0;
      for (int i = 0; i < total; ++i) zeros_total += run[i];
      while (t1 < 3 && t1 < total && (level[t1] == 1 || level[t1] == -1)) ++t1;
    }
  }
  int token = 4 * total + t1;
  if (nc == -1) {
    token = sx_vlc(r, chroma_dc_coeff_token_len, chroma_dc_coeff_token_bits, 20, token);
  } else {
    int t = nc < 2 ? 0 : (nc < 4 ? 1 : (nc < 8 ? 2 : 3));
    token = sx_vlc(r, coeff_token_len[t], coeff_token_bits[t], 68, token);
  }
  if (token < 0) return h264i_fail(d, "CAVLC: bad coeff_token");
  COV(d, H264O_COV_COEFF_TOKEN + (nc == -1 ? 4 : (nc < 2 ? 0 : (nc < 4 ? 1 : (nc < 8 ? 2 : 3)))) * 68 + token);
  total = token >> 2;
  t1 = token & 3;
  if (total == 0) return 0;
  if (total > max_coeff) return h264i_fail(d, "CAVLC: total_coeff exceeds block size");
  for (int i = 0; i < t1; ++i) {
    int neg = sx_u(r, 1, r->enc ? level[i] < 0 : 0);
    level[i] = neg ? -1 : 1;
  }
  int suffix_len = (total > 10 && t1 < 3) ? 1 : 0;
  for (int i = t1; i < total; ++i) {
    int prefix, code;
    if (r->enc) {
      /* invert 9.2.2.1: find (level_prefix, level_suffix) for this level */
      code = level[i] > 0 ? 2 * level[i] - 2 : -2 * level[i] - 1;
      if (i == t1 && t1 < 3) code -= 2;
      int size;
      if (suffix_len == 0) {
        if (code < 14) { prefix = code; size = 0; }
        else if (code < 30) { prefix = 14; size = 4; }
        else { prefix = 15; size = 12; }
      } else {
        if ((code >> suffix_len) < 15) { prefix = code >> suffix_len; size = suffix_len; }
        else { prefix = 15; size = 12; }
      }
      int base = imin(15, prefix) << suffix_len;
      if (prefix >= 15 && suffix_len == 0) base += 15;
      int suffix = code - base;
      if (prefix == 15 && suffix >= (1 << 12)) {
        /* escape beyond 12 bits: longer prefixes (level_prefix >= 16) */
        prefix = 16;
        for (;;) {
          int b16 = (15 << suffix_len) + (suffix_len == 0 ? 15 : 0) + (1 << (prefix - 3)) - 4096;
          if (code - b16 < (1 << (prefix - 3))) { suffix = code - b16; size = prefix - 3; break; }
          ++prefix;
        }
      }
      if (prefix >= 14 && !(prefix == 14 && suffix_len)) COV(d, H264O_COV_ESCAPE + (prefix == 14 ? 0 : (prefix == 15 ? 1 : 2)));
      bw_put(r->w, 0, prefix);
      bw_put(r->w, 1, 1);
      if (size > 0) bw_put(r->w, suffix, size);
    } else {
      prefix = 0;
      while (!get_bit(b)) {
        if (++prefix > 31 || b->overrun) return h264i_fail(d, "CAVLC: bad level_prefix");
      }
      if (prefix >= 14 && !(prefix == 14 && suffix_len)) COV(d, H264O_COV_ESCAPE + (prefix == 14 ? 0 : (prefix == 15 ? 1 : 2)));
      code = imin(15, prefix) << suffix_len;
      int size = suffix_len;
      if (prefix == 14 && suffix_len == 0) size = 4;
      if (prefix >= 15) size = prefix - 3;
      if (size > 0) code += get_bits(b, size);
      if (prefix >= 15 && suffix_len == 0) code += 15;
      if (prefix >= 16) code += (1 << (prefix - 3)) - 4096;
      if (i == t1 && t1 < 3) code += 2;
      level[i] = (code & 1) ? (-code - 1) >> 1 : (code + 2) >> 1;
    }
    if (suffix_len == 0) suffix_len = 1;
    if (iabs(level[i]) > (3 << (suffix_len - 1)) && suffix_len < 6) ++suffix_len;
  }
  int zeros_left = 0;
  if (total < max_coeff) {
    int tz;
    if (nc == -1)
      tz = sx_vlc(r, chroma_dc_total_zeros_len[total - 1], chroma_dc_total_zeros_bits[total - 1], 4, zeros_total);
    else
      tz = sx_vlc(r, total_zeros_len[total - 1], total_zeros_bits[total - 1], total_zeros_count[total - 1], zeros_total);
    if (tz < 0) return h264i_fail(d, "CAVLC: bad total_zeros");
    if (nc == -1) COV(d, H264O_COV_TOTAL_ZEROS_DC + (total - 1) * 4 + tz);
    else COV(d, H264O_COV_TOTAL_ZEROS + (total - 1) * 16 + tz);
    zeros_left = tz;
  }
  if (zeros_left + total > max_coeff) return h264i_fail(d, "CAVLC: total_zeros exceeds block size");
  int pos = zeros_left + total - 1; /* scan index of the highest coefficient */
  for (int i = 0; i < total; ++i) {
    if (!r->enc) io[scan[start + pos]] = (int16_t)level[i];
    if (i == total - 1) break;
    int rb = 0;
    if (zeros_left > 0) {
      int t = imin(zeros_left, 7) - 1;
      rb = sx_vlc(r, run_before_len[t], run_before_bits[t], run_before_count[t], r->enc ? run[i] : 0);
      if (rb < 0 || rb > zeros_left) return h264i_fail(d, "CAVLC: bad run_before");
      COV(d, H264O_COV_RUN_BEFORE + t * 15 + rb);
      zeros_left -= rb;
    }
    pos -= rb + 1;
  }
  return total;
}

/* ======================================================================== */
/* residual: CABAC (9.3.2.x / 9.3.3.1.3)                                    */
/* ======================================================================== */
static const int cbf_cat_offset[6] = {0, 4, 8, 12, 16, 0};
static const int sig_cat_offset[6] = {0, 15, 29, 44, 47, 0};
static const int abs_cat_offset[6] = {0, 10, 20, 30, 39, 0};

/* cat: 0 I16 DC, 1 I16 AC, 2 luma 4x4, 3 chroma DC, 4 chroma AC, 5 luma 8x8.
 * cbf_inc < 0: no coded_block_flag (cat 5 in 4:2:0).  Returns nonzero count. */
static int cabac_block_comp(Rd* r, int comp, int cat, int cbf_inc, int max_coeff, const uint8_t* scan, int start,
                            int16_t* io);
static int cabac_block(Rd* r, int cat, int cbf_inc, int max_coeff, const uint8_t* scan, int start,
                       int16_t* io) {
  return cabac_block_comp(r, 0, cat, cbf_inc, max_coeff, scan, start, io);
}
/* comp: colour component (0; 1 Cb / 2 Cr of a 4:4:4 picture, coded like luma with ctxBlockCat cat + 6 / cat + 10 resp.
 * 9 / 13 for cat 5: Table 9-34's ctxIdxOffsets for those categories) */
static int cabac_block_comp(Rd* r, int comp, int cat, int cbf_inc, int max_coeff, const uint8_t* scan, int start,
                            int16_t* io) {
  H264Oracle* d = r->d;
  static const int cbf_base[3] = {85, 460, 472}, cbf8_base[3] = {1012, 1016, 1020};
  static const int sig_base_c[3] = {105, 484, 528}, sig8_base[3] = {402, 660, 718};
  static const int last_base_c[3] = {166, 572, 616}, last8_base[3] = {417, 690, 748};
  static const int abs_base_c[3] = {227, 952, 982}, abs8_base[3] = {426, 708, 766};
  int want[64];
  int nz = 0, last_nz = -1;
  if (r->enc) {
    for (int k = 0; k < max_coeff; ++k) {
      want[k] = io[scan[start + k]];
      if (want[k]) {
        ++nz;
        last_nz = k;
      }
    }
  }
  if (cbf_inc >= 0) {
    if (!sx_bin(r, (cat == 5 ? cbf8_base[comp] : cbf_base[comp] + cbf_cat_offset[cat]) + cbf_inc, nz != 0)) return 0;
  } else if (r->enc && !nz) {
    return h264i_fail(d, "encoder: 8x8 block flagged in CBP has no coefficients");
  }
  const int sig_base = cat == 5 ? sig8_base[comp] : sig_base_c[comp] + sig_cat_offset[cat];
  const int last_base = cat == 5 ? last8_base[comp] : last_base_c[comp] + sig_cat_offset[cat];
  const int abs_base = cat == 5 ? abs8_base[comp] : abs_base_c[comp] + abs_cat_offset[cat];
  int idx[64], n = 0;
  int i;
  for (i = 0; i < max_coeff - 1; ++i) {
    int sinc, linc;
    if (cat == 5) {
      sinc = sig_coeff_offset_8x8[i];
      linc = last_coeff_offset_8x8[i];
    } else if (cat == 3) {
      sinc = linc = imin(i, 2);
    } else {
      sinc = linc = i;
    }
    if (sx_bin(r, sig_base + sinc, r->enc ? want[i] != 0 : 0)) {
      idx[n++] = i;
      if (sx_bin(r, last_base + linc, i == last_nz)) break;
    }
  }
  if (i == max_coeff - 1) idx[n++] = max_coeff - 1; /* inferred significant */
  int gt1 = 0, eq1 = 0;
  for (int k = n - 1; k >= 0; --k) {
    const int target = r->enc ? iabs(want[idx[k]]) : 0;
    int inc0 = gt1 ? 0 : imin(4, 1 + eq1);
    int val;
    if (!sx_bin(r, abs_base + inc0, target > 1)) {
      val = 1;
      ++eq1;
    } else {
      int inc = 5 + imin(4 - (cat == 3 ? 1 : 0), gt1);
      val = 2;
      while (val < 15 && sx_bin(r, abs_base + inc, target > val)) ++val;
      if (val == 15) { /* prefix saturated: Exp-Golomb k=0 suffix */
        COV(d, H264O_COV_ESCAPE + 3);
        int k0 = 0, rem = target - 15;
        while (sx_bypass(r, rem >= (1 << k0))) {
          val += 1 << k0;
          rem -= 1 << k0;
          if (++k0 > 24) return h264i_fail(d, "CABAC: coeff_abs_level overflow");
        }
        while (k0--) val += sx_bypass(r, (rem >> k0) & 1) << k0;
      }
      ++gt1;
    }
    if (sx_bypass(r, r->enc ? want[idx[k]] < 0 : 0)) val = -val;
    if (!r->enc) io[scan[start + idx[k]]] = (int16_t)val;
  }
  return n;
}

/* coded_block_flag ctxIdxInc from the two neighbouring transform blocks.
 * kind: 0 luma DC, 1 luma 4x4 block (rx, ry), 2 chroma DC plane p, 3 chroma
 * AC block (rx, ry) of plane p. */
static int cbf_neighbour(const H264Oracle* d, int kind, int p, int rx, int ry, int left) {
  const MB* m = &d->mb;
  const int cur_intra = m->kind != MB_INTER;
  int nx = m->mbx, ny = m->mby, inside = 0;
  if (kind == 1 || kind == 3) {
    if (left) {
      if (rx > 0) inside = 1; else nx--;
    } else {
      if (ry > 0) inside = 1; else ny--;
    }
  } else {
    if (left) nx--; else ny--;
  }
  if (inside) {
    if (kind == 1) return m->nnz[(left ? ry * 4 + rx - 1 : (ry - 1) * 4 + rx)] != 0;
    const int base = 16 + 4 * p;
    return m->nnz[base + (left ? ry * 2 + rx - 1 : (ry - 1) * 2 + rx)] != 0;
  }
  if (!mb_avail(d, nx, ny)) return cur_intra;
  const int addr = ny * d->mbw + nx;
  if (d->mb_kind[addr] == MB_IPCM) return 1;
  switch (kind) {
    case 0: return d->mb_kind[addr] == MB_I16x16 && (d->mb_cbf_dc[addr] & 1);
    case 1: return d->nnz[addr * 24 + (left ? ry * 4 + 3 : 12 + rx)] != 0;
    case 2: return (d->mb_cbf_dc[addr] >> (1 + p)) & 1;
    default: return d->nnz[addr * 24 + 16 + 4 * p + (left ? ry * 2 + 1 : 2 + rx)] != 0;
  }
}
static int cbf_inc(const H264Oracle* d, int kind, int p, int rx, int ry) {
  return cbf_neighbour(d, kind, p, rx, ry, 1) + 2 * cbf_neighbour(d, kind, p, rx, ry, 0);
}

/* ======================================================================== */
/* macroblock syntax elements                                               */
/* ======================================================================== */
static int sx_intra_mb_type(Rd* r, int ctx_base, int intra_slice, int v) {
  /* v: 0 I_NxN, 1..24 I_16x16 family, 25 I_PCM */
  const int t = v - 1;
  const int luma = v >= 13, chroma = (t >> 2) % 3, mode = t & 3;
  if (intra_slice) {
    const H264Oracle* d = r->d;
    int inc = 0;
    if (mb_avail(d, d->mb.mbx - 1, d->mb.mby)) {
      int k = d->mb_kind[d->mb.addr - 1];
      inc += k != MB_I4x4 && k != MB_I8x8;
    }
    if (mb_avail(d, d->mb.mbx, d->mb.mby - 1)) {
      int k = d->mb_kind[d->mb.addr - d->mbw];
      inc += k != MB_I4x4 && k != MB_I8x8;
    }
    if (!sx_bin(r, ctx_base + inc, v != 0)) return 0;
    ctx_base += 2;
  } else {
    if (!sx_bin(r, ctx_base, v != 0)) return 0;
  }
  if (sx_terminate(r, v == 25)) return 25;
  int out = 1;
  out += 12 * sx_bin(r, ctx_base + 1, luma);
  if (sx_bin(r, ctx_base + 2, chroma != 0)) out += 4 + 4 * sx_bin(r, ctx_base + 2 + intra_slice, chroma == 2);
  out += 2 * sx_bin(r, ctx_base + 3 + intra_slice, mode >> 1);
  out += sx_bin(r, ctx_base + 3 + 2 * intra_slice, mode & 1);
  return out;
}

/* slice-type specific mb_type number */
static int sx_mb_type(Rd* r, int v) {
  H264Oracle* d = r->d;
  const int st = d->sl.type;
  if (!r->cabac) return (int)sx_ue(r, v);
  if (st == 2) return sx_intra_mb_type(r, 3, 1, v);
  if (st == 0) {
    if (!sx_bin(r, 14, v >= 5)) {
      if (!sx_bin(r, 15, v == 1 || v == 2)) return 3 * sx_bin(r, 16, v == 3);
      return 2 - sx_bin(r, 17, v == 1);
    }
    return 5 + sx_intra_mb_type(r, 17, 0, v - 5);
  }
  int inc = 0;
  if (mb_avail(d, d->mb.mbx - 1, d->mb.mby) && !d->mb_direct[d->mb.addr - 1]) inc++;
  if (mb_avail(d, d->mb.mbx, d->mb.mby - 1) && !d->mb_direct[d->mb.addr - d->mbw]) inc++;
  if (!sx_bin(r, 27 + inc, v != 0)) return 0;
  if (!sx_bin(r, 27 + 3, v > 2)) return 1 + sx_bin(r, 27 + 5, v == 2);
  /* 4-bit code (5 for the 12..21 group) */
  int code, extra = -1;
  if (v >= 23) code = 13;
  else if (v == 11) code = 14;
  else if (v == 22) code = 15;
  else if (v <= 10) code = v - 3;
  else { code = (v + 4) >> 1; extra = (v + 4) & 1; }
  int bits = sx_bin(r, 27 + 4, (code >> 3) & 1) << 3;
  bits |= sx_bin(r, 27 + 5, (code >> 2) & 1) << 2;
  bits |= sx_bin(r, 27 + 5, (code >> 1) & 1) << 1;
  bits |= sx_bin(r, 27 + 5, code & 1);
  if (bits < 8) return bits + 3;
  if (bits == 13) return 23 + sx_intra_mb_type(r, 32, 0, v - 23);
  if (bits == 14) return 11;
  if (bits == 15) return 22;
  bits = (bits << 1) | sx_bin(r, 27 + 5, extra);
  return bits - 4;
}

static int sx_sub_mb_type(Rd* r, int v) {
  H264Oracle* d = r->d;
  if (!r->cabac) return (int)sx_ue(r, v);
  if (d->sl.type == 0) {
    if (sx_bin(r, 21, v == 0)) return 0;
    if (!sx_bin(r, 22, v != 1)) return 1;
    return sx_bin(r, 23, v == 2) ? 2 : 3;
  }
  if (!sx_bin(r, 36, v != 0)) return 0;
  if (!sx_bin(r, 37, v > 2)) return 1 + sx_bin(r, 39, v == 2);
  int t = 3;
  if (sx_bin(r, 38, v >= 7)) {
    if (sx_bin(r, 39, v >= 11)) return 11 + sx_bin(r, 39, v == 12);
    t += 4;
  }
  const int rem = v - t; /* 0..3 */
  t += 2 * sx_bin(r, 39, (rem >> 1) & 1);
  t += sx_bin(r, 39, rem & 1);
  return t;
}

static int sx_ref_idx(Rd* r, int list, int b8, int v) {
  H264Oracle* d = r->d;
  const int n = d->sl.num_ref_idx[list];
  if (n <= 1) return 0;
  if (!r->cabac) {
    if (n == 2) return !sx_u(r, 1, !v);
    return (int)sx_ue(r, v);
  }
  const int bx = (b8 & 1) * 2, by = (b8 >> 1) * 2;
  int inc = 0;
  for (int k = 0; k < 2; ++k) {
    int rx = k == 0 ? bx - 1 : bx, ry = k == 0 ? by : by - 1;
    int nx = d->mb.mbx + (rx < 0 ? -1 : 0), ny = d->mb.mby + (ry < 0 ? -1 : 0);
    int inside = rx >= 0 && ry >= 0;
    if (!inside && !mb_avail(d, nx, ny)) continue;
    int addr = ny * d->mbw + nx;
    int q = ((ry & 3) >> 1) * 2 + ((rx & 3) >> 1);
    if (d->cur->mb_intra[addr]) continue;
    if (d->sl.type == 1 && d->direct8[addr * 4 + q]) continue;
    if (d->cur->ref_idx[list][addr * 4 + q] > 0) inc += k == 0 ? 1 : 2;
  }
  int ref = 0;
  while (sx_bin(r, 54 + inc, v > ref)) {
    ++ref;
    inc = (inc >> 2) + 4;
    if (ref >= 32) return -1;
  }
  return ref;
}

static int sx_mvd(Rd* r, int list, int bx, int by, int comp, int v) {
  H264Oracle* d = r->d;
  if (!r->cabac) return sx_se(r, v);
  int sum = 0;
  for (int k = 0; k < 2; ++k) {
    int rx = k == 0 ? bx - 1 : bx, ry = k == 0 ? by : by - 1;
    int nx = d->mb.mbx + (rx < 0 ? -1 : 0), ny = d->mb.mby + (ry < 0 ? -1 : 0);
    int inside = rx >= 0 && ry >= 0;
    if (!inside && !mb_avail(d, nx, ny)) continue;
    sum += d->mvd[ny * d->mbw + nx][list][(ry & 3) * 4 + (rx & 3)][comp];
  }
  const int base = comp == 0 ? 40 : 47;
  const int a = iabs(v);
  int inc = sum < 3 ? 0 : (sum > 32 ? 2 : 1);
  if (!sx_bin(r, base + inc, a != 0)) return 0;
  int val = 1, ctx = base + 3;
  while (val < 9 && sx_bin(r, ctx, a > val)) {
    if (val < 4) ++ctx;
    ++val;
  }
  if (val >= 9) {
    COV(d, H264O_COV_ESCAPE + 4);
    int k = 3, rem = a - 9;
    while (sx_bypass(r, rem >= (1 << k))) {
      val += 1 << k;
      rem -= 1 << k;
      if (++k > 24) return 0;
    }
    while (k--) val += sx_bypass(r, (rem >> k) & 1) << k;
  }
  return sx_bypass(r, v < 0) ? -val : val;
}

static int sx_cbp(Rd* r, int intra, int v) {
  H264Oracle* d = r->d;
  MB* m = &d->mb;
  if (!r->cabac) {
    unsigned code = 0;
    if (r->enc) {
      const uint8_t* tab = intra ? cbp_intra_from_code : cbp_inter_from_code;
      while (tab[code] != v) ++code;
    }
    code = sx_ue(r, code);
    if (code > 47) return -1;
    return intra ? cbp_intra_from_code[code] : cbp_inter_from_code[code];
  }
  const int la = mb_avail(d, m->mbx - 1, m->mby), ta = mb_avail(d, m->mbx, m->mby - 1);
  /* unavailable neighbours behave like "all luma bits set" */
  const int cbp_a = la ? d->mb_cbp[m->addr - 1] : 0x0f;
  const int cbp_b = ta ? d->mb_cbp[m->addr - d->mbw] : 0x0f;
  int cbp = 0;
  for (int b8 = 0; b8 < 4; ++b8) {
    int bit_a = (b8 & 1) ? (cbp >> (b8 - 1)) & 1 : (cbp_a >> (b8 + 1)) & 1;
    int bit_b = (b8 & 2) ? (cbp >> (b8 - 2)) & 1 : (cbp_b >> (b8 + 2)) & 1;
    cbp |= sx_bin(r, 73 + !bit_a + 2 * !bit_b, (v >> b8) & 1) << b8;
  }
  if (d->chroma444) return cbp; /* ChromaArrayType 3: the bin string is the prefix alone (9.3.2.6) */
  const int ca = la ? d->mb_cbp[m->addr - 1] >> 4 : 0;
  const int cb = ta ? d->mb_cbp[m->addr - d->mbw] >> 4 : 0;
  if (sx_bin(r, 77 + (ca != 0) + 2 * (cb != 0), (v >> 4) != 0)) {
    int c2 = 1 + sx_bin(r, 77 + 4 + (ca == 2) + 2 * (cb == 2), (v >> 4) == 2);
    cbp |= c2 << 4;
  }
  return cbp;
}

static int sx_qp_delta(Rd* r, int v) {
  H264Oracle* d = r->d;
  if (!r->cabac) return sx_se(r, v);
  int ctx = d->last_qp_delta_nonzero ? 1 : 0;
  const int target = v > 0 ? 2 * v - 1 : -2 * v;
  int n = 0;
  while (sx_bin(r, 60 + ctx, target > n)) {
    ctx = 2 + (ctx >> 1);
    if (++n > 104) return 1000;
  }
  return (n & 1) ? (n + 1) >> 1 : -((n + 1) >> 1);
}

static int sx_chroma_pred_mode(Rd* r, int v) {
  H264Oracle* d = r->d;
  const MB* m = &d->mb;
  if (!r->cabac) return (int)sx_ue(r, v);
  int inc = 0;
  if (mb_avail(d, m->mbx - 1, m->mby) && d->mb_chroma_mode[m->addr - 1]) inc++;
  if (mb_avail(d, m->mbx, m->mby - 1) && d->mb_chroma_mode[m->addr - d->mbw]) inc++;
  if (!sx_bin(r, 64 + inc, v != 0)) return 0;
  if (!sx_bin(r, 64 + 3, v != 1)) return 1;
  return 2 + sx_bin(r, 64 + 3, v == 3);
}

static int sx_t8x8_flag(Rd* r, int v) {
  H264Oracle* d = r->d;
  const MB* m = &d->mb;
  if (!r->cabac) return (int)sx_u(r, 1, v);
  int inc = 0;
  if (mb_avail(d, m->mbx - 1, m->mby) && d->mb_t8x8[m->addr - 1]) inc++;
  if (mb_avail(d, m->mbx, m->mby - 1) && d->mb_t8x8[m->addr - d->mbw]) inc++;
  return sx_bin(r, 399 + inc, v);
}

/* prev_intra4x4_pred_mode_flag / rem_intra4x4_pred_mode; -1 = "use
 * predicted", else rem 0..7 */
static int sx_intra_pred_mode(Rd* r, int v) {
  if (!r->cabac) {
    if (sx_u(r, 1, v < 0)) return -1;
    return (int)sx_u(r, 3, v);
  }
  if (sx_bin(r, 68, v < 0)) return -1;
  int out = sx_bin(r, 69, v & 1);
  out |= sx_bin(r, 69, (v >> 1) & 1) << 1;
  out |= sx_bin(r, 69, (v >> 2) & 1) << 2;
  return out;
}

/* predicted Intra4x4/8x8 mode for the block whose top-left 4x4 is (bx,by) */
static int predicted_intra_mode(const H264Oracle* d, int bx, int by) {
  const MB* m = &d->mb;
  int mode[2];
  for (int k = 0; k < 2; ++k) {
    int rx = k == 0 ? bx - 1 : bx, ry = k == 0 ? by : by - 1;
    if (rx >= 0 && ry >= 0) {
      mode[k] = m->i4_mode[ry * 4 + rx];
      continue;
    }
    int nx = m->mbx + (rx < 0 ? -1 : 0), ny = m->mby + (ry < 0 ? -1 : 0);
    if (!mb_avail(d, nx, ny)) return 2;
    int addr = ny * d->mbw + nx;
    int kind = d->mb_kind[addr];
    if (kind == MB_INTER && d->apps->constrained_intra_pred) return 2;
    if (kind != MB_I4x4 && kind != MB_I8x8)
      mode[k] = 2;
    else
      mode[k] = d->i4mode[addr * 16 + (ry & 3) * 4 + (rx & 3)];
  }
  return imin(mode[0], mode[1]);
}

/* ======================================================================== */
/* residual()                                                               */
/* ======================================================================== */
static const uint8_t identity_scan16[16] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15};

/* coded_block_flag ctxIdxInc of a Cb / Cr block of a 4:4:4 picture: as for luma, from the same colour component's
 * blocks to the left and above (kind 0: the Intra16x16 DC block, 1: 4x4 block (rx, ry) - for an 8x8 block its
 * top-left 4x4 block, every 4x4 block of a coded 8x8 block carrying the 8x8 block's count). */
static int cbf444_neighbour(const H264Oracle* d, int comp, int kind, int rx, int ry, int left) {
  const MB* m = &d->mb;
  const int cur_intra = m->kind != MB_INTER;
  int nx = m->mbx, ny = m->mby, inside = 0;
  if (kind == 1) {
    if (left) {
      if (rx > 0) inside = 1; else nx--;
    } else {
      if (ry > 0) inside = 1; else ny--;
    }
  } else {
    if (left) nx--; else ny--;
  }
  const int base = (comp - 1) * 16;
  if (inside) return m->nnz444[base + (left ? ry * 4 + rx - 1 : (ry - 1) * 4 + rx)] != 0;
  if (!mb_avail(d, nx, ny)) return cur_intra;
  const int addr = ny * d->mbw + nx;
  if (d->mb_kind[addr] == MB_IPCM) return 1;
  if (kind == 0) return d->mb_kind[addr] == MB_I16x16 && ((d->mb_cbf_dc[addr] >> comp) & 1);
  return d->nnz444[addr * 32 + base + (left ? ry * 4 + 3 : 12 + rx)] != 0;
}
static int cbf444_inc(const H264Oracle* d, int comp, int kind, int rx, int ry) {
  return cbf444_neighbour(d, comp, kind, rx, ry, 1) + 2 * cbf444_neighbour(d, comp, kind, rx, ry, 0);
}
/* coded_block_flag ctxIdxInc of an 8x8 block (ctxBlockCat 5 / 9 / 13, in the stream only in 4:4:4) whose top-left 4x4
 * block is (rx, ry).  As libavcodec does it - the reference's decoder (h264_cabac.c: the blocks' coefficient counts
 * in one cache, a 4x4-transformed neighbour counting with its adjacent 4x4 block) - including its emulation of x264
 * before build 151, which coded these flags as if an available neighbouring macroblock WITHOUT the 8x8 transform
 * were not available (1 for an intra macroblock, 0 for an inter one): real 4:4:4 files of that age - imageio's
 * cockatoo.mp4 is "x264 - core 142" - only parse that way. */
static int cbf8_inc(const H264Oracle* d, int comp, int rx, int ry) {
  const MB* m = &d->mb;
  int inc = 0;
  for (int left = 1; left >= 0; --left) {
    int v;
    const int outside = left ? rx == 0 : ry == 0;
    const int nx = m->mbx - (left ? 1 : 0), ny = m->mby - (left ? 0 : 1);
    if (outside && d->x264_build >= 0 && d->x264_build < 151 && mb_avail(d, nx, ny) && !d->mb_t8x8[ny * d->mbw + nx])
      v = m->kind != MB_INTER;
    else
      v = comp == 0 ? cbf_neighbour(d, 1, 0, rx, ry, left) : cbf444_neighbour(d, comp, 1, rx, ry, left);
    inc += v << (left ? 0 : 1);
  }
  return inc;
}

/* residual() of a 4:4:4 macroblock (7.3.5.3 with ChromaArrayType 3): residual_luma three times - Y, Cb, Cr - all under
 * the luma bits of coded_block_pattern; the coded_block_flag of 8x8 blocks is in the stream.  Parse-only: the levels
 * go nowhere. */
static int code_residual444(Rd* r) {
  H264Oracle* d = r->d;
  MB* m = &d->mb;
  const int i16 = m->kind == MB_I16x16;
  if (!r->cabac || r->enc) return h264i_fail(d, "4:4:4: CABAC streams only, parse only");
  int16_t scratch[64];
  for (int comp = 0; comp < 3; ++comp) {
    uint8_t* nz = comp == 0 ? m->nnz : m->nnz444 + (comp - 1) * 16;
    if (i16) {
      memset(scratch, 0, sizeof(scratch));
      const int inc = comp == 0 ? cbf_inc(d, 0, 0, 0, 0) : cbf444_inc(d, comp, 0, 0, 0);
      int n = cabac_block_comp(r, comp, 0, inc, 16, zigzag4x4, 0, scratch);
      if (n < 0) return -1;
      if (n) m->cbf_dc |= 1 << comp;
    }
    for (int b8 = 0; b8 < 4; ++b8) {
      if (!((m->cbp >> b8) & 1)) continue;
      const int bx0 = (b8 & 1) * 2, by0 = (b8 >> 1) * 2;
      if (m->t8x8) {
        memset(scratch, 0, sizeof(scratch));
        const int inc = cbf8_inc(d, comp, bx0, by0);
        int n = cabac_block_comp(r, comp, 5, inc, 64, zigzag8x8, 0, scratch);
        if (n < 0) return -1;
        for (int s = 0; s < 4; ++s) nz[(by0 + (s >> 1)) * 4 + bx0 + (s & 1)] = (uint8_t)n;
        continue;
      }
      for (int s = 0; s < 4; ++s) {
        const int bx = bx0 + (s & 1), by = by0 + (s >> 1);
        memset(scratch, 0, sizeof(scratch));
        const int inc = comp == 0 ? cbf_inc(d, 1, 0, bx, by) : cbf444_inc(d, comp, 1, bx, by);
        int n = i16 ? cabac_block_comp(r, comp, 1, inc, 15, zigzag4x4, 1, scratch)
                    : cabac_block_comp(r, comp, 2, inc, 16, zigzag4x4, 0, scratch);
        if (n < 0) return -1;
        nz[by * 4 + bx] = (uint8_t)n;
      }
    }
  }
  if (r->b->overrun) return h264i_fail(d, "residual ran past the end of the slice");
  return 0;
}

static int code_residual(Rd* r) {
  H264Oracle* d = r->d;
  MB* m = &d->mb;
  const int i16 = m->kind == MB_I16x16;
  if (d->chroma444) return code_residual444(r);

  if (i16) {
    /* Intra16x16DCLevel: zig-zag over the 4x4 matrix of DCs whose
     * (row, col) is the 4x4 block's (y, x) */
    int n = r->cabac ? cabac_block(r, 0, cbf_inc(d, 0, 0, 0, 0), 16, zigzag4x4, 0, m->luma_dc)
                     : cavlc_block(r, cavlc_nc(d, 0, 0, 0), 16, zigzag4x4, 0, m->luma_dc);
    if (n < 0) return -1;
    if (n) m->cbf_dc |= 1;
  }
  for (int b8 = 0; b8 < 4; ++b8) {
    if (!((m->cbp >> b8) & 1)) continue;
    if (m->t8x8 && r->cabac) {
      int n = cabac_block(r, 5, -1, 64, zigzag8x8, 0, m->luma8[b8]);
      if (n < 0) return -1;
      for (int s = 0; s < 4; ++s)
        m->nnz[((b8 >> 1) * 2 + (s >> 1)) * 4 + (b8 & 1) * 2 + (s & 1)] = (uint8_t)n;
      continue;
    }
    for (int s = 0; s < 4; ++s) {
      const int bx = (b8 & 1) * 2 + (s & 1), by = (b8 >> 1) * 2 + (s >> 1);
      const int ri = by * 4 + bx;
      int n;
      if (m->t8x8) {
        /* CAVLC 8x8: four interleaved 4x4 passes, coefficient 4k+s */
        int16_t tmp[16];
        for (int k = 0; k < 16; ++k) tmp[k] = r->enc ? m->luma8[b8][zigzag8x8[4 * k + s]] : 0;
        n = cavlc_block(r, cavlc_nc(d, 0, bx, by), 16, identity_scan16, 0, tmp);
        if (n < 0) return -1;
        for (int k = 0; k < 16; ++k)
          if (tmp[k]) m->luma8[b8][zigzag8x8[4 * k + s]] = tmp[k];
      } else if (i16) {
        n = r->cabac ? cabac_block(r, 1, cbf_inc(d, 1, 0, bx, by), 15, zigzag4x4, 1, m->luma[ri])
                     : cavlc_block(r, cavlc_nc(d, 0, bx, by), 15, zigzag4x4, 1, m->luma[ri]);
      } else {
        n = r->cabac ? cabac_block(r, 2, cbf_inc(d, 1, 0, bx, by), 16, zigzag4x4, 0, m->luma[ri])
                     : cavlc_block(r, cavlc_nc(d, 0, bx, by), 16, zigzag4x4, 0, m->luma[ri]);
      }
      if (n < 0) return -1;
      m->nnz[ri] = (uint8_t)n;
    }
  }
  const int cc = m->cbp >> 4;
  if (cc) {
    static const uint8_t dc_scan[4] = {0, 1, 2, 3};
    for (int p = 0; p < 2; ++p) {
      int n = r->cabac ? cabac_block(r, 3, cbf_inc(d, 2, p, 0, 0), 4, dc_scan, 0, m->chroma_dc[p])
                       : cavlc_block(r, -1, 4, dc_scan, 0, m->chroma_dc[p]);
      if (n < 0) return -1;
      if (n) m->cbf_dc |= 2 << p;
    }
  }
  if (cc == 2) {
    for (int p = 0; p < 2; ++p)
      for (int s = 0; s < 4; ++s) {
        const int bx = s & 1, by = s >> 1;
        int n = r->cabac
                    ? cabac_block(r, 4, cbf_inc(d, 3, p, bx, by), 15, zigzag4x4, 1, m->chroma_ac[p][s])
                    : cavlc_block(r, cavlc_nc(d, 1 + p, bx, by), 15, zigzag4x4, 1, m->chroma_ac[p][s]);
        if (n < 0) return -1;
        m->nnz[16 + 4 * p + s] = (uint8_t)n;
      }
  }
  if (!r->enc && r->b->overrun) return h264i_fail(d, "residual ran past the end of the slice");
  return 0;
}

/* ======================================================================== */
/* inter prediction syntax                                                  */
/* ======================================================================== */
/* B macroblock partition table: shape 0=16x16 1=16x8 2=8x16 3=8x8; pred per
 * part 0=L0 1=L1 2=Bi */
static const uint8_t b_mb_info[23][3] = {
    {0, 0, 0}, {0, 0, 0}, {0, 1, 1}, {0, 2, 2}, {1, 0, 0}, {2, 0, 0}, {1, 1, 1}, {2, 1, 1},
    {1, 0, 1}, {2, 0, 1}, {1, 1, 0}, {2, 1, 0}, {1, 0, 2}, {2, 0, 2}, {1, 1, 2}, {2, 1, 2},
    {1, 2, 0}, {2, 2, 0}, {1, 2, 1}, {2, 2, 1}, {1, 2, 2}, {2, 2, 2}, {3, 0, 0}};
/* B sub-macroblock: shape 0=8x8 1=8x4 2=4x8 3=4x4; pred 0=L0 1=L1 2=Bi 3=direct */
static const uint8_t b_sub_info[13][2] = {{0, 3}, {0, 0}, {0, 1}, {0, 2}, {1, 0}, {2, 0}, {1, 1},
                                          {2, 1}, {1, 2}, {2, 2}, {3, 0}, {3, 1}, {3, 2}};

static inline int uses_list(int pred, int list) { return pred == 2 || pred == list; }

static int code_inter(Rd* r, int mb_type) {
  H264Oracle* d = r->d;
  MB* m = &d->mb;
  const MB* P = r->enc ? &r->plan->mb : NULL;
  const int is_b = d->sl.type == 1;
  int shape, pred[4], sub_shape[4] = {0, 0, 0, 0};
  int nparts;

  if (!is_b) {
    shape = mb_type == 0 ? 0 : (mb_type == 1 ? 1 : (mb_type == 2 ? 2 : 3));
    pred[0] = pred[1] = pred[2] = pred[3] = 0;
  } else {
    shape = b_mb_info[mb_type][0];
    pred[0] = b_mb_info[mb_type][1];
    pred[1] = b_mb_info[mb_type][2];
  }
  nparts = shape == 0 ? 1 : (shape == 3 ? 4 : 2);

  if (shape == 3) {
    for (int i = 0; i < 4; ++i) {
      int t = sx_sub_mb_type(r, P ? P->sub_type[i] : 0);
      if (t < 0 || t > (is_b ? 12 : 3)) return h264i_fail(d, "bad sub_mb_type");
      m->sub_type[i] = t;
      COV(d, (is_b ? H264O_COV_SUB_B : H264O_COV_SUB_P) + t);
      if (is_b) {
        sub_shape[i] = b_sub_info[t][0];
        pred[i] = b_sub_info[t][1];
      } else {
        sub_shape[i] = t;
        pred[i] = 0;
      }
    }
    if (is_b) {
      for (int i = 0; i < 4; ++i)
        if (pred[i] == 3) {
          d->direct8[m->addr * 4 + i] = 1;
          if (pred_direct(d, i) < 0) return -1;
        }
    }
  }

  /* ref_idx_l0 for every partition, then ref_idx_l1 */
  for (int l = 0; l < (is_b ? 2 : 1); ++l) {
    for (int i = 0; i < nparts; ++i) {
      if (pred[i] == 3) continue;
      int ref = -1;
      const int b8 = shape == 0 ? 0 : (shape == 1 ? i * 2 : i);
      if (uses_list(pred[i], l)) {
        if (!is_b && mb_type == 4) {
          ref = 0; /* P_8x8ref0 */
        } else {
          ref = sx_ref_idx(r, l, b8, P ? P->ref[l][b8] : 0);
          if (ref < 0 || ref >= d->sl.num_ref_idx[l]) return h264i_fail(d, "ref_idx out of range");
        }
      }
      if (shape == 0) {
        for (int q = 0; q < 4; ++q) set_ref8(d, l, q, ref);
      } else if (shape == 1) {
        set_ref8(d, l, i * 2, ref);
        set_ref8(d, l, i * 2 + 1, ref);
      } else if (shape == 2) {
        set_ref8(d, l, i, ref);
        set_ref8(d, l, i + 2, ref);
      } else {
        set_ref8(d, l, i, ref);
      }
    }
  }
  /* mvd_l0 for every (sub-)partition, then mvd_l1 */
  for (int l = 0; l < (is_b ? 2 : 1); ++l) {
    for (int i = 0; i < nparts; ++i) {
      if (pred[i] == 3 || !uses_list(pred[i], l)) continue;
      int px, py, pw, ph;
      if (shape == 0) { px = 0; py = 0; pw = 4; ph = 4; }
      else if (shape == 1) { px = 0; py = i * 2; pw = 4; ph = 2; }
      else if (shape == 2) { px = i * 2; py = 0; pw = 2; ph = 4; }
      else { px = (i & 1) * 2; py = (i >> 1) * 2; pw = 2; ph = 2; }
      int sw = pw, sh = ph;
      if (shape == 3) {
        const int ss = sub_shape[i];
        if (ss == 2 || ss == 3) sw = 1;
        if (ss == 1 || ss == 3) sh = 1;
      }
      const int ref = m->ref[l][(py >> 1) * 2 + (px >> 1)];
      for (int sy = 0; sy < ph; sy += sh)
        for (int sx = 0; sx < pw; sx += sw) {
          int bx = px + sx, by = py + sy;
          int dir = PRED_MEDIAN;
          if (shape == 1) dir = i == 0 ? PRED_16x8_TOP : PRED_16x8_BOTTOM;
          if (shape == 2) dir = i == 0 ? PRED_8x16_LEFT : PRED_8x16_RIGHT;
          int mvp[2];
          pred_mv(d, l, bx, by, sw, ref, dir, mvp);
          int tx = P ? P->mv[l][by * 4 + bx][0] - mvp[0] : 0;
          int ty = P ? P->mv[l][by * 4 + bx][1] - mvp[1] : 0;
          int dx = sx_mvd(r, l, bx, by, 0, tx);
          int dy = sx_mvd(r, l, bx, by, 1, ty);
          set_mvd_block(d, l, bx, by, sw, sh, dx, dy);
          set_mv_block(d, l, bx, by, sw, sh, mvp[0] + dx, mvp[1] + dy);
        }
    }
  }
  return 0;
}

/* ======================================================================== */
/* macroblock_layer()                                                       */
/* ======================================================================== */
void h264i_begin_mb(H264Oracle* d, int addr) {
  MB* m = &d->mb;
  memset(m, 0, sizeof(*m));
  m->addr = addr;
  m->mbx = addr % d->mbw;
  m->mby = addr / d->mbw;
  memset(m->ref, -1, sizeof(m->ref));
  memset(m->i4_mode, 2, sizeof(m->i4_mode));
  Picture* cur = d->cur;
  for (int l = 0; l < 2; ++l) {
    memset(cur->mv[l] + addr * 16, 0, 16 * sizeof(int16_t[2]));
    for (int q = 0; q < 4; ++q) {
      cur->ref_idx[l][addr * 4 + q] = -1;
      cur->ref_uid[l][addr * 4 + q] = -1;
    }
  }
  memset(d->mvd[addr], 0, sizeof(d->mvd[addr]));
  memset(d->direct8 + addr * 4, 0, 4);
  cur->mb_intra[addr] = 0;
  d->mb_slice[addr] = (int16_t)d->slice_index;
  d->mb_kind[addr] = MB_NONE;
  m->qp = d->qp;
}

static void commit_mb(H264Oracle* d) {
  MB* m = &d->mb;
  const int a = m->addr;
  const PPS* pps = d->apps;
  m->qpc[0] = chroma_qp_table[iclip(0, 51, m->qp + pps->chroma_qp_offset[0])];
  m->qpc[1] = chroma_qp_table[iclip(0, 51, m->qp + pps->chroma_qp_offset[1])];
  d->mb_kind[a] = (uint8_t)m->kind;
  d->mb_skip[a] = (uint8_t)m->skip;
  d->mb_direct[a] = (uint8_t)((m->skip || m->direct16) && d->sl.type == 1);
  d->mb_t8x8[a] = (uint8_t)m->t8x8;
  d->mb_cbp[a] = (uint8_t)(m->kind == MB_IPCM ? 0x2f : m->cbp);
  d->mb_cbf_dc[a] = (uint8_t)m->cbf_dc;
  d->mb_qp[a] = (int8_t)(m->kind == MB_IPCM ? 0 : m->qp);
  d->mb_qpc[0][a] = (int8_t)(m->kind == MB_IPCM ? chroma_qp_table[iclip(0, 51, pps->chroma_qp_offset[0])] : m->qpc[0]);
  d->mb_qpc[1][a] = (int8_t)(m->kind == MB_IPCM ? chroma_qp_table[iclip(0, 51, pps->chroma_qp_offset[1])] : m->qpc[1]);
  d->mb_chroma_mode[a] = (uint8_t)((m->kind != MB_INTER && m->kind != MB_IPCM) ? m->chroma_mode : 0);
  memcpy(d->i4mode + a * 16, m->i4_mode, 16);
  memcpy(d->nnz + a * 24, m->nnz, 24);
  memcpy(d->nnz444 + a * 32, m->nnz444, 32);
  d->mb_deblock[a].disable = (int8_t)d->sl.disable_deblock;
  d->mb_deblock[a].alpha = (int8_t)d->sl.alpha_offset;
  d->mb_deblock[a].beta = (int8_t)d->sl.beta_offset;
  d->cur->mb_intra[a] = m->kind != MB_INTER;
  d->counters[2]++;
  if (m->kind != MB_INTER) d->counters[7]++;
  if (m->skip) d->counters[8]++;
  if (m->t8x8) d->counters[9]++;
}

void h264i_pskip_mv(H264Oracle* d, int mv[2]) {
  NB A = fetch_nb(d, 0, -1, 0, 0), B = fetch_nb(d, 0, 0, -1, 0);
  mv[0] = mv[1] = 0;
  if (!(A.ref == -2 || B.ref == -2 || (A.ref == 0 && A.mv[0] == 0 && A.mv[1] == 0) ||
        (B.ref == 0 && B.mv[0] == 0 && B.mv[1] == 0)))
    pred_mv(d, 0, 0, 0, 4, 0, PRED_MEDIAN, mv);
}

void h264i_pred_mv16(H264Oracle* d, int list, int ref, int mv[2]) {
  pred_mv(d, list, 0, 0, 4, ref, PRED_MEDIAN, mv);
}

int h264i_pred_direct_mb(H264Oracle* d) { return pred_direct(d, -1); }

/* (corpus generator, --global-mv 7) the prediction for one partition of the current macroblock - shape 1: 16x8, 2: 8x16,
 * part 0 / 1 - and a way to put a partition's reference and vector where the next partition's prediction looks for it */
void h264i_pred_mv_part(H264Oracle* d, int shape, int part, int ref, int mv[2]) {
  if (shape == 1) pred_mv(d, 0, 0, part * 2, 4, ref, part == 0 ? PRED_16x8_TOP : PRED_16x8_BOTTOM, mv);
  else pred_mv(d, 0, part * 2, 0, 2, ref, part == 0 ? PRED_8x16_LEFT : PRED_8x16_RIGHT, mv);
}
/* ... per list, for the partitions of a B macroblock (shape 0: 16x16) */
void h264i_pred_mv_part_l(H264Oracle* d, int list, int shape, int part, int ref, int mv[2]) {
  if (shape == 0) pred_mv(d, list, 0, 0, 4, ref, PRED_MEDIAN, mv);
  else if (shape == 1) pred_mv(d, list, 0, part * 2, 4, ref, part == 0 ? PRED_16x8_TOP : PRED_16x8_BOTTOM, mv);
  else pred_mv(d, list, part * 2, 0, 2, ref, part == 0 ? PRED_8x16_LEFT : PRED_8x16_RIGHT, mv);
}
void h264i_set_part_l(H264Oracle* d, int list, int shape, int part, int ref, int mvx, int mvy) {
  for (int q = 0; q < 4; ++q)
    if (shape == 0 || (shape == 1 ? (q >> 1) : (q & 1)) == part) set_ref8(d, list, q, ref);
  if (shape == 0) set_mv_block(d, list, 0, 0, 4, 4, ref >= 0 ? mvx : 0, ref >= 0 ? mvy : 0);
  else if (shape == 1) set_mv_block(d, list, 0, part * 2, 4, 2, ref >= 0 ? mvx : 0, ref >= 0 ? mvy : 0);
  else set_mv_block(d, list, part * 2, 0, 2, 4, ref >= 0 ? mvx : 0, ref >= 0 ? mvy : 0);
}
/* ... and the same for one sub-macroblock partition (bx, by, bw, bh in 4x4 blocks inside the macroblock) */
void h264i_pred_mv_blk(H264Oracle* d, int bx, int by, int bw, int ref, int mv[2]) { pred_mv(d, 0, bx, by, bw, ref, PRED_MEDIAN, mv); }
void h264i_set_blk(H264Oracle* d, int bx, int by, int bw, int bh, int mvx, int mvy) { set_mv_block(d, 0, bx, by, bw, bh, mvx, mvy); }
void h264i_set_ref8(H264Oracle* d, int b8, int ref) { set_ref8(d, 0, b8, ref); }
void h264i_set_part(H264Oracle* d, int shape, int part, int ref, int mvx, int mvy) {
  for (int q = 0; q < 4; ++q)
    if ((shape == 1 ? (q >> 1) : (q & 1)) == part) set_ref8(d, 0, q, ref);
  if (shape == 1) set_mv_block(d, 0, 0, part * 2, 4, 2, mvx, mvy);
  else set_mv_block(d, 0, part * 2, 0, 2, 4, mvx, mvy);
}

static int code_skip_mb(H264Oracle* d, int addr) {
  h264i_begin_mb(d, addr);
  MB* m = &d->mb;
  m->kind = MB_INTER;
  m->skip = 1;
  d->last_qp_delta_nonzero = 0;
  COV(d, d->sl.type == 0 ? H264O_COV_MBTYPE_P + 31 : H264O_COV_MBTYPE_B + 49);
  if (d->sl.type == 0) {
    for (int q = 0; q < 4; ++q) set_ref8(d, 0, q, 0);
    int mv[2];
    h264i_pskip_mv(d, mv);
    set_mv_block(d, 0, 0, 0, 4, 4, mv[0], mv[1]);
  } else {
    m->direct16 = 1;
    memset(d->direct8 + addr * 4, 1, 4);
    if (pred_direct(d, -1) < 0) return -1;
  }
  commit_mb(d);
  if (!d->parse_only) h264i_recon_mb(d);
  return 0;
}

static int code_mb(Rd* r, int addr) {
  H264Oracle* d = r->d;
  const PPS* pps = d->apps;
  const MB* P = r->enc ? &r->plan->mb : NULL;
  h264i_begin_mb(d, addr);
  MB* m = &d->mb;
  const int st = d->sl.type;
  int t = sx_mb_type(r, P ? P->mb_type_raw : 0);
  int intra_t = -1;
  if (st == 2) {
    intra_t = t;
  } else if (st == 0) {
    if (t > 30 || t < 0) return h264i_fail(d, "bad mb_type (P)");
    if (t >= 5) intra_t = t - 5;
  } else {
    if (t > 48 || t < 0) return h264i_fail(d, "bad mb_type (B)");
    if (t >= 23) intra_t = t - 23;
  }
  if (intra_t > 25) return h264i_fail(d, "bad mb_type (I)");
  m->mb_type_raw = t;
  COV(d, (st == 2 ? H264O_COV_MBTYPE_I : (st == 0 ? H264O_COV_MBTYPE_P : H264O_COV_MBTYPE_B)) + t);

  if (intra_t == 25) { /* I_PCM */
    m->kind = MB_IPCM;
    if (r->enc) {
      /* CABAC: the terminating bin flushed the arithmetic coder; its final
       * '1' is followed by pcm_alignment_zero_bits */
      while (r->w->pos & 7) bw_put(r->w, 0, 1);
      for (int i = 0; i < 384; ++i) {
        m->pcm[i] = P->pcm[i];
        bw_put(r->w, m->pcm[i], 8);
      }
      if (r->cabac) cenc_start(&d->cenc, r->w);
    } else {
      Bits* b = r->b;
      while (b->pos & 7) get_bit(b);
      for (int i = 0; i < 384; ++i) m->pcm[i] = (uint8_t)get_bits(b, 8);
      if (d->chroma444) /* 2 x 256 chroma samples instead of 2 x 64 (parse only: dropped) */
        for (int i = 0; i < 384; ++i) (void)get_bits(b, 8);
      if (b->overrun) return h264i_fail(d, "truncated I_PCM");
      if (r->cabac) cabac_start(&d->cabac, b);
    }
    memset(m->nnz, 16, sizeof(m->nnz));
    memset(m->nnz444, 16, sizeof(m->nnz444));
    m->cbf_dc = 7;
    d->last_qp_delta_nonzero = 0;
    commit_mb(d);
    if (!d->parse_only) h264i_recon_mb(d);
    return 0;
  }

  if (intra_t >= 0) {
    if (intra_t == 0) {
      m->kind = MB_I4x4;
      if (pps->transform_8x8_mode && sx_t8x8_flag(r, P ? P->t8x8 : 0)) {
        m->kind = MB_I8x8;
        m->t8x8 = 1;
      }
      const int nblk = m->kind == MB_I4x4 ? 16 : 4;
      for (int k = 0; k < nblk; ++k) {
        const int bx = nblk == 16 ? blk4_x[k] : (k & 1) * 2, by = nblk == 16 ? blk4_y[k] : (k >> 1) * 2;
        const int pm = predicted_intra_mode(d, bx, by);
        int want = -1;
        if (P) {
          int target = P->i4_mode[by * 4 + bx];
          want = target == pm ? -1 : (target < pm ? target : target - 1);
        }
        int rem = sx_intra_pred_mode(r, want);
        int mode = rem < 0 ? pm : (rem < pm ? rem : rem + 1);
        if (nblk == 16) {
          m->i4_mode[by * 4 + bx] = (int8_t)mode;
        } else {
          for (int s = 0; s < 4; ++s) m->i4_mode[(by + (s >> 1)) * 4 + bx + (s & 1)] = (int8_t)mode;
        }
      }
    } else {
      m->kind = MB_I16x16;
      m->i16_mode = (intra_t - 1) & 3;
      m->cbp = (((intra_t - 1) >> 2) % 3) << 4 | ((intra_t - 1) >= 12 ? 15 : 0);
    }
    /* intra_chroma_pred_mode: with ChromaArrayType 1 or 2 only (7.3.5.1) */
    m->chroma_mode = d->chroma444 ? 0 : sx_chroma_pred_mode(r, P ? P->chroma_mode : 0);
    if (m->chroma_mode > 3 || m->chroma_mode < 0) return h264i_fail(d, "bad intra_chroma_pred_mode");
    d->cur->mb_intra[addr] = 1;
  } else {
    m->kind = MB_INTER;
    if (st == 1 && t == 0) {
      m->direct16 = 1;
      memset(d->direct8 + addr * 4, 1, 4);
      if (pred_direct(d, -1) < 0) return -1;
    } else {
      if (code_inter(r, t) < 0) return -1;
    }
  }

  if (m->kind != MB_I16x16) {
    int cbp = sx_cbp(r, m->kind != MB_INTER, P ? P->cbp : 0);
    if (cbp < 0) return h264i_fail(d, "bad coded_block_pattern");
    m->cbp = cbp;
    if ((cbp & 15) && pps->transform_8x8_mode && m->kind == MB_INTER) {
      int ok = 1;
      if (st == 1 && t == 0) ok = d->asps->direct_8x8_inference;
      if ((st == 0 && (t == 3 || t == 4)) || (st == 1 && t == 22)) {
        for (int i = 0; i < 4; ++i) {
          if (st == 1 && m->sub_type[i] == 0) {
            if (!d->asps->direct_8x8_inference) ok = 0;
          } else if (st == 1 ? b_sub_info[m->sub_type[i]][0] != 0 : m->sub_type[i] != 0) {
            ok = 0;
          }
        }
      }
      if (ok) m->t8x8 = sx_t8x8_flag(r, P ? P->t8x8 : 0);
    }
  }
  if (m->cbp || m->kind == MB_I16x16) {
    int want = 0;
    if (P) {
      want = P->qp - d->qp;
      if (want > 25) want -= 52;
      if (want < -26) want += 52;
    }
    int dq = sx_qp_delta(r, want);
    if (dq < -26 || dq > 25) return h264i_fail(d, "mb_qp_delta out of range");
    d->last_qp_delta_nonzero = dq != 0;
    d->qp = (d->qp + dq + 52) % 52;
    m->qp = d->qp;
    if (P) { /* the levels to write */
      memcpy(m->luma_dc, P->luma_dc, sizeof(m->luma_dc));
      memcpy(m->luma, P->luma, sizeof(m->luma));
      memcpy(m->luma8, P->luma8, sizeof(m->luma8));
      memcpy(m->chroma_dc, P->chroma_dc, sizeof(m->chroma_dc));
      memcpy(m->chroma_ac, P->chroma_ac, sizeof(m->chroma_ac));
    }
    if (code_residual(r) < 0) return -1;
  } else {
    d->last_qp_delta_nonzero = 0;
  }
  commit_mb(d);
  if (!d->parse_only) h264i_recon_mb(d);
  return 0;
}

/* ======================================================================== */
/* slice_data(): parse                                                      */
/* ======================================================================== */
static int skip_flag_ctx(H264Oracle* d, int addr) {
  d->mb.addr = addr;
  d->mb.mbx = addr % d->mbw;
  d->mb.mby = addr / d->mbw;
  int inc = 0;
  if (mb_avail(d, d->mb.mbx - 1, d->mb.mby) && !d->mb_skip[addr - 1]) inc++;
  if (mb_avail(d, d->mb.mbx, d->mb.mby - 1) && !d->mb_skip[addr - d->mbw]) inc++;
  return (d->sl.type == 0 ? 11 : 24) + inc;
}

int h264i_decode_slice_data(H264Oracle* d, Bits* b) {
  Slice* sl = &d->sl;
  const PPS* pps = d->apps;
  const int nmb = d->mbw * d->mbh;
  Rd r = {d, b, pps->cabac, 0, NULL, NULL};
  int addr = sl->first_mb;
  if (addr >= nmb) return h264i_fail(d, "first_mb_in_slice out of range");
  d->qp = sl->qp;
  d->last_qp_delta_nonzero = 0;
  d->counters[0]++;

  if (pps->cabac) {
    while (b->pos & 7) {
      if (!get_bit(b)) return h264i_fail(d, "cabac_alignment_one_bit is zero");
    }
    cabac_init_contexts(&d->cabac, sl->type, sl->cabac_init_idc, sl->qp);
    d->cov_ctx_table = sl->type == 2 ? 0 : (sl->type == 0 ? 1 : 4) + sl->cabac_init_idc;
    cabac_start(&d->cabac, b);
    for (;;) {
      if (addr >= nmb) return h264i_fail(d, "slice runs past the end of the picture");
      int skipped = 0;
      if (sl->type != 2) {
        const int sctx = skip_flag_ctx(d, addr);
        skipped = cabac_bin(&d->cabac, sctx);
        COV(d, H264O_COV_CTX + (d->cov_ctx_table * CABAC_CTX_COUNT + sctx) * 2 + (skipped != 0));
      }
      if (skipped) {
        if (code_skip_mb(d, addr) < 0) return -1;
      } else {
        if (code_mb(&r, addr) < 0) return -1;
      }
      if (b->overrun) return h264i_fail(d, "slice data ran past the end of the NAL unit");
      ++addr;
      if (cabac_terminate(&d->cabac)) break;
    }
    /* desync check.  The arithmetic decoder runs 9 bits ahead of the
     * encoder's committed output and the encoder never writes its first
     * (dummy) bit, so after the terminating bin the LAST bit consumed is the
     * rbsp_stop_one_bit the encoder's flush ends with; only alignment zeros
     * (and cabac_zero_words) may follow. */
    {
      Bits t = *b;
      int ok = t.pos > 0 && ((t.buf[(t.pos - 1) >> 3] >> (7 - ((t.pos - 1) & 7))) & 1);
      while (ok && t.pos < t.size_bits)
        if (get_bit(&t)) ok = 0;
      /* Parse-only mode (foreign encoders): x264 ends a slice with a flush of its own - the arithmetic codeword is
       * padded to the byte boundary and a pseudo-random bit follows (cabac.c: x264_cabac_encode_flush), so up to a
       * byte of unread bits may lie between the decoder's position and the NAL unit's last set bit.  A decoder that
       * had lost sync would not end on the picture's last macroblock within eight bits of the end of the NAL unit. */
      if (!ok && d->parse_only && !b->overrun && b->size_bits - b->pos <= 8) ok = 1;
      if (ok && !b->overrun) d->counters[1]++;
    }
    return 0;
  }

  /* CAVLC */
  for (;;) {
    if (sl->type != 2) {
      unsigned run = get_ue(b);
      if (run > (unsigned)(nmb - addr)) return h264i_fail(d, "mb_skip_run out of range");
      while (run--) {
        if (code_skip_mb(d, addr) < 0) return -1;
        ++addr;
      }
      if (!h264i_more_rbsp_data(b)) break;
    }
    if (addr >= nmb) return h264i_fail(d, "slice runs past the end of the picture");
    if (code_mb(&r, addr) < 0) return -1;
    if (b->overrun) return h264i_fail(d, "slice data ran past the end of the NAL unit");
    ++addr;
    if (!h264i_more_rbsp_data(b)) break;
  }
  {
    Bits t = *b;
    int ok = get_bit(&t) == 1;
    while (ok && t.pos < t.size_bits)
      if (get_bit(&t)) ok = 0;
    if (ok && !t.overrun) d->counters[1]++;
  }
  return 0;
}

/* ======================================================================== */
/* slice_data(): write (corpus encoder)                                     */
/* ======================================================================== */
/* The caller has already parsed its own slice header through
 * h264i_parse_slice_header (so reference lists, POC and the picture exist)
 * and appended the same header bits to `w`. */
void h264i_enc_begin_slice(H264Oracle* d, BitWriter* w) {
  Slice* sl = &d->sl;
  d->qp = sl->qp;
  d->last_qp_delta_nonzero = 0;
  d->counters[0]++;
  d->cenc.outstanding = 0;
  if (d->apps->cabac) {
    while (w->pos & 7) bw_put(w, 1, 1); /* cabac_alignment_one_bit */
    cabac_init_contexts(&d->cabac, sl->type, sl->cabac_init_idc, sl->qp);
    d->cov_ctx_table = sl->type == 2 ? 0 : (sl->type == 0 ? 1 : 4) + sl->cabac_init_idc;
    cenc_start(&d->cenc, w);
  }
}

int h264i_enc_mb(H264Oracle* d, BitWriter* w, int addr, const MbPlan* plan, int last_in_slice) {
  const int cabac = d->apps->cabac;
  Rd r = {d, NULL, cabac, 1, w, plan};
  int rc;
  if (cabac) {
    if (d->sl.type != 2) cenc_bin(&d->cenc, &d->cabac.state[skip_flag_ctx(d, addr)], plan->is_skip);
    rc = plan->is_skip ? code_skip_mb(d, addr) : code_mb(&r, addr);
    if (rc < 0) return rc;
    cenc_terminate(&d->cenc, last_in_slice);
    return 0;
  }
  /* CAVLC: skip runs are written lazily in front of the next coded MB (or
   * at the end of the slice); the count lives in cenc.outstanding, which is
   * otherwise unused without CABAC. */
  if (plan->is_skip) {
    rc = code_skip_mb(d, addr);
    d->cenc.outstanding++;
    if (last_in_slice) {
      bw_ue(w, d->cenc.outstanding);
      d->cenc.outstanding = 0;
    }
    return rc;
  }
  if (d->sl.type != 2) {
    bw_ue(w, d->cenc.outstanding);
    d->cenc.outstanding = 0;
  }
  return code_mb(&r, addr);
}

void h264i_enc_end_slice(H264Oracle* d, BitWriter* w) {
  if (!d->apps->cabac) bw_put(w, 1, 1); /* rbsp_stop_one_bit (CABAC: written by the flush) */
  while (w->pos & 7) bw_put(w, 0, 1);
}
