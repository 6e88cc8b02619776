/* TEST INFRASTRUCTURE -- oracle: sample reconstruction.  Intra prediction
 * (8.3), inter prediction with quarter-sample interpolation and weighting
 * (8.4.2), scaling + inverse transforms (8.5) and the deblocking filter
 * (8.7).  Restates FFmpeg's h264_mb.c / h264pred / h264qpel / h264chroma /
 * h264idct / h264_loopfilter.c as used behind
 * software_video_decoder.cpp:353-380.  These are the per-sample definitions
 * the HIP kernels in hwang_amd/csrc/amd are checked against. */
#include "h264_internal.h"

/* ======================================================================== */
/* transforms                                                               */
/* ======================================================================== */
static void idct4x4(const int* in, int* out) { /* raster in -> residual (already >> 6) */
  int t[16];
  for (int i = 0; i < 4; ++i) { /* rows */
    const int* d = in + 4 * i;
    int e0 = d[0] + d[2], e1 = d[0] - d[2];
    int e2 = (d[1] >> 1) - d[3], e3 = d[1] + (d[3] >> 1);
    t[4 * i + 0] = e0 + e3;
    t[4 * i + 1] = e1 + e2;
    t[4 * i + 2] = e1 - e2;
    t[4 * i + 3] = e0 - e3;
  }
  for (int j = 0; j < 4; ++j) { /* columns */
    int d0 = t[j], d1 = t[4 + j], d2 = t[8 + j], d3 = t[12 + j];
    int e0 = d0 + d2, e1 = d0 - d2;
    int e2 = (d1 >> 1) - d3, e3 = d1 + (d3 >> 1);
    out[j] = (e0 + e3 + 32) >> 6;
    out[4 + j] = (e1 + e2 + 32) >> 6;
    out[8 + j] = (e1 - e2 + 32) >> 6;
    out[12 + j] = (e0 - e3 + 32) >> 6;
  }
}

static void idct8_1d(const int* d, int stride, int* o, int ostride) {
  int d0 = d[0], d1 = d[stride], d2 = d[2 * stride], d3 = d[3 * stride];
  int d4 = d[4 * stride], d5 = d[5 * stride], d6 = d[6 * stride], d7 = d[7 * stride];
  int e0 = d0 + d4;
  int e1 = -d3 + d5 - d7 - (d7 >> 1);
  int e2 = d0 - d4;
  int e3 = d1 + d7 - d3 - (d3 >> 1);
  int e4 = (d2 >> 1) - d6;
  int e5 = -d1 + d7 + d5 + (d5 >> 1);
  int e6 = d2 + (d6 >> 1);
  int e7 = d3 + d5 + d1 + (d1 >> 1);
  int f0 = e0 + e6, f1 = e1 + (e7 >> 2), f2 = e2 + e4, f3 = e3 + (e5 >> 2);
  int f4 = e2 - e4, f5 = (e3 >> 2) - e5, f6 = e0 - e6, f7 = e7 - (e1 >> 2);
  o[0] = f0 + f7;
  o[ostride] = f2 + f5;
  o[2 * ostride] = f4 + f3;
  o[3 * ostride] = f6 + f1;
  o[4 * ostride] = f6 - f1;
  o[5 * ostride] = f4 - f3;
  o[6 * ostride] = f2 - f5;
  o[7 * ostride] = f0 - f7;
}

static void idct8x8(const int* in, int* out) {
  int t[64];
  for (int i = 0; i < 8; ++i) idct8_1d(in + 8 * i, 1, t + 8 * i, 1);
  for (int j = 0; j < 8; ++j) idct8_1d(t + j, 8, out + j, 8);
  for (int k = 0; k < 64; ++k) out[k] = (out[k] + 32) >> 6;
}

int h264i_level_scale4(const uint8_t* weight, int qp, int pos) {
  const int i = pos >> 2, j = pos & 3;
  const int cls = ((i & 1) == 0 && (j & 1) == 0) ? 0 : (((i & 1) && (j & 1)) ? 1 : 2);
  return weight[pos] * norm_adjust4x4[qp % 6][cls];
}

int h264i_level_scale8(const uint8_t* weight, int qp, int pos) {
  const int i = pos >> 3, j = pos & 7;
  int cls;
  if ((i & 3) == 0 && (j & 3) == 0) cls = 0;
  else if ((i & 1) && (j & 1)) cls = 1;
  else if ((i & 3) == 2 && (j & 3) == 2) cls = 2;
  else if (((i & 3) == 0 && (j & 1)) || ((i & 1) && (j & 3) == 0)) cls = 3;
  else if (((i & 3) == 0 && (j & 3) == 2) || ((i & 3) == 2 && (j & 3) == 0)) cls = 4;
  else cls = 5;
  return weight[pos] * norm_adjust8x8[qp % 6][cls];
}

/* scale a 4x4 block of levels; dc_given: element 0 is already scaled */
static void scale4x4(const int16_t* lv, const uint8_t* weight, int qp, int dc_given, int dc, int* out) {
  for (int k = 0; k < 16; ++k) {
    if (k == 0 && dc_given) {
      out[0] = dc;
      continue;
    }
    int v = lv[k] * h264i_level_scale4(weight, qp, k);
    out[k] = qp >= 24 ? v << (qp / 6 - 4) : (v + (1 << (3 - qp / 6))) >> (4 - qp / 6);
  }
}

static void add_residual(uint8_t* dst, int stride, const int* res, int n) {
  for (int y = 0; y < n; ++y)
    for (int x = 0; x < n; ++x) dst[y * stride + x] = (uint8_t)clip8(dst[y * stride + x] + res[y * n + x]);
}

/* ======================================================================== */
/* intra prediction                                                         */
/* ======================================================================== */
/* availability of the neighbouring macroblock for INTRA prediction */
int h264i_intra_mb_ok(const H264Oracle* d, int mbx, int mby) {
  if (mbx < 0 || mby < 0 || mbx >= d->mbw || mby >= d->mbh) return 0;
  const int a = mby * d->mbw + mbx;
  if (d->mb_slice[a] != d->slice_index) return 0;
  if (d->apps->constrained_intra_pred && d->mb_kind[a] == MB_INTER) return 0;
  return 1;
}

/* Is the 4x4 luma block at MB-relative (rx, ry) (may be -1 / 4) usable as an
 * intra neighbour of the block whose decode index is cur_zz? */
int h264i_intra_blk_ok(const H264Oracle* d, int rx, int ry, int cur_zz) {
  const MB* m = &d->mb;
  if (rx > 3 && ry >= 0) return 0;
  if (rx >= 0 && rx <= 3 && ry >= 0) return raster_to_blk4[ry * 4 + rx] < cur_zz;
  return h264i_intra_mb_ok(d, m->mbx + (rx < 0 ? -1 : (rx > 3 ? 1 : 0)), m->mby + (ry < 0 ? -1 : 0));
}

void h264i_pred4x4(uint8_t* dst, int stride, int mode, int has_left, int has_top, int has_tl, int has_tr) {
  int t[8], l[4], tl = 128;
  for (int i = 0; i < 4; ++i) {
    l[i] = has_left ? dst[i * stride - 1] : 128;
    t[i] = has_top ? dst[-stride + i] : 128;
  }
  for (int i = 4; i < 8; ++i) t[i] = has_top ? (has_tr ? dst[-stride + i] : dst[-stride + 3]) : 128;
  if (has_tl) tl = dst[-stride - 1];
  /* p(x,-1) = t[x], p(-1,y) = l[y], p(-1,-1) = tl; index -1 via helper */
#define T(x) ((x) < 0 ? tl : t[x])
#define L(y) ((y) < 0 ? tl : l[y])
  for (int y = 0; y < 4; ++y)
    for (int x = 0; x < 4; ++x) {
      int v;
      switch (mode) {
        case 0: v = t[x]; break;
        case 1: v = l[y]; break;
        case 2:
          if (has_top && has_left) v = (t[0] + t[1] + t[2] + t[3] + l[0] + l[1] + l[2] + l[3] + 4) >> 3;
          else if (has_left) v = (l[0] + l[1] + l[2] + l[3] + 2) >> 2;
          else if (has_top) v = (t[0] + t[1] + t[2] + t[3] + 2) >> 2;
          else v = 128;
          break;
        case 3: /* diagonal down-left */
          v = (x == 3 && y == 3) ? (t[6] + 3 * t[7] + 2) >> 2 : (t[x + y] + 2 * t[x + y + 1] + t[x + y + 2] + 2) >> 2;
          break;
        case 4: /* diagonal down-right */
          if (x > y) v = (T(x - y - 2) + 2 * T(x - y - 1) + T(x - y) + 2) >> 2;
          else if (x < y) v = (L(y - x - 2) + 2 * L(y - x - 1) + L(y - x) + 2) >> 2;
          else v = (t[0] + 2 * tl + l[0] + 2) >> 2;
          break;
        case 5: { /* vertical-right */
          int z = 2 * x - y;
          if (z >= 0 && !(z & 1)) v = (T(x - (y >> 1) - 1) + T(x - (y >> 1)) + 1) >> 1;
          else if (z >= 0) v = (T(x - (y >> 1) - 2) + 2 * T(x - (y >> 1) - 1) + T(x - (y >> 1)) + 2) >> 2;
          else if (z == -1) v = (l[0] + 2 * tl + t[0] + 2) >> 2;
          else v = (L(y - 1) + 2 * L(y - 2) + L(y - 3) + 2) >> 2;
          break;
        }
        case 6: { /* horizontal-down */
          int z = 2 * y - x;
          if (z >= 0 && !(z & 1)) v = (L(y - (x >> 1) - 1) + L(y - (x >> 1)) + 1) >> 1;
          else if (z >= 0) v = (L(y - (x >> 1) - 2) + 2 * L(y - (x >> 1) - 1) + L(y - (x >> 1)) + 2) >> 2;
          else if (z == -1) v = (l[0] + 2 * tl + t[0] + 2) >> 2;
          else v = (T(x - 1) + 2 * T(x - 2) + T(x - 3) + 2) >> 2;
          break;
        }
        case 7: /* vertical-left */
          if (!(y & 1)) v = (t[x + (y >> 1)] + t[x + (y >> 1) + 1] + 1) >> 1;
          else v = (t[x + (y >> 1)] + 2 * t[x + (y >> 1) + 1] + t[x + (y >> 1) + 2] + 2) >> 2;
          break;
        default: { /* 8 horizontal-up */
          int z = x + 2 * y;
          if (z > 5) v = l[3];
          else if (z == 5) v = (l[2] + 3 * l[3] + 2) >> 2;
          else if (!(z & 1)) v = (l[y + (x >> 1)] + l[y + (x >> 1) + 1] + 1) >> 1;
          else v = (l[y + (x >> 1)] + 2 * l[y + (x >> 1) + 1] + l[y + (x >> 1) + 2] + 2) >> 2;
          break;
        }
      }
      /* write into a scratch first: predictions read dst's neighbours only,
       * which lie outside the 4x4, so in-place is safe */
      dst[y * stride + x] = (uint8_t)v;
    }
#undef T
#undef L
}

void h264i_pred8x8l(uint8_t* dst, int stride, int mode, int has_left, int has_top, int has_tl, int has_tr) {
  int rt[16], rl[8], rtl = 128; /* raw */
  int t[16], l[8], tl = 128;    /* filtered */
  for (int i = 0; i < 8; ++i) {
    rl[i] = has_left ? dst[i * stride - 1] : 128;
    rt[i] = has_top ? dst[-stride + i] : 128;
  }
  for (int i = 8; i < 16; ++i) rt[i] = has_top ? (has_tr ? dst[-stride + i] : dst[-stride + 7]) : 128;
  if (has_tl) rtl = dst[-stride - 1];
  if (has_top) {
    t[0] = has_tl ? (rtl + 2 * rt[0] + rt[1] + 2) >> 2 : (3 * rt[0] + rt[1] + 2) >> 2;
    for (int i = 1; i < 15; ++i) t[i] = (rt[i - 1] + 2 * rt[i] + rt[i + 1] + 2) >> 2;
    t[15] = (rt[14] + 3 * rt[15] + 2) >> 2;
  } else {
    for (int i = 0; i < 16; ++i) t[i] = 128;
  }
  if (has_left) {
    l[0] = has_tl ? (rtl + 2 * rl[0] + rl[1] + 2) >> 2 : (3 * rl[0] + rl[1] + 2) >> 2;
    for (int i = 1; i < 7; ++i) l[i] = (rl[i - 1] + 2 * rl[i] + rl[i + 1] + 2) >> 2;
    l[7] = (rl[6] + 3 * rl[7] + 2) >> 2;
  } else {
    for (int i = 0; i < 8; ++i) l[i] = 128;
  }
  if (has_tl) {
    if (has_top && has_left) tl = (rt[0] + 2 * rtl + rl[0] + 2) >> 2;
    else if (has_top) tl = (3 * rtl + rt[0] + 2) >> 2;
    else if (has_left) tl = (3 * rtl + rl[0] + 2) >> 2;
    else tl = rtl;
  }
#define T(x) ((x) < 0 ? tl : t[x])
#define L(y) ((y) < 0 ? tl : l[y])
  uint8_t out[64];
  for (int y = 0; y < 8; ++y)
    for (int x = 0; x < 8; ++x) {
      int v;
      switch (mode) {
        case 0: v = t[x]; break;
        case 1: v = l[y]; break;
        case 2: {
          int st = 0, sl = 0;
          for (int i = 0; i < 8; ++i) {
            st += t[i];
            sl += l[i];
          }
          if (has_top && has_left) v = (st + sl + 8) >> 4;
          else if (has_left) v = (sl + 4) >> 3;
          else if (has_top) v = (st + 4) >> 3;
          else v = 128;
          break;
        }
        case 3:
          v = (x == 7 && y == 7) ? (t[14] + 3 * t[15] + 2) >> 2 : (t[x + y] + 2 * t[x + y + 1] + t[x + y + 2] + 2) >> 2;
          break;
        case 4:
          if (x > y) v = (T(x - y - 2) + 2 * T(x - y - 1) + T(x - y) + 2) >> 2;
          else if (x < y) v = (L(y - x - 2) + 2 * L(y - x - 1) + L(y - x) + 2) >> 2;
          else v = (t[0] + 2 * tl + l[0] + 2) >> 2;
          break;
        case 5: {
          int z = 2 * x - y;
          if (z >= 0 && !(z & 1)) v = (T(x - (y >> 1) - 1) + T(x - (y >> 1)) + 1) >> 1;
          else if (z >= 0) v = (T(x - (y >> 1) - 2) + 2 * T(x - (y >> 1) - 1) + T(x - (y >> 1)) + 2) >> 2;
          else if (z == -1) v = (l[0] + 2 * tl + t[0] + 2) >> 2;
          else v = (L(y - 2 * x - 1) + 2 * L(y - 2 * x - 2) + L(y - 2 * x - 3) + 2) >> 2;
          break;
        }
        case 6: {
          int z = 2 * y - x;
          if (z >= 0 && !(z & 1)) v = (L(y - (x >> 1) - 1) + L(y - (x >> 1)) + 1) >> 1;
          else if (z >= 0) v = (L(y - (x >> 1) - 2) + 2 * L(y - (x >> 1) - 1) + L(y - (x >> 1)) + 2) >> 2;
          else if (z == -1) v = (l[0] + 2 * tl + t[0] + 2) >> 2;
          else v = (T(x - 2 * y - 1) + 2 * T(x - 2 * y - 2) + T(x - 2 * y - 3) + 2) >> 2;
          break;
        }
        case 7:
          if (!(y & 1)) v = (t[x + (y >> 1)] + t[x + (y >> 1) + 1] + 1) >> 1;
          else v = (t[x + (y >> 1)] + 2 * t[x + (y >> 1) + 1] + t[x + (y >> 1) + 2] + 2) >> 2;
          break;
        default: {
          int z = x + 2 * y;
          if (z > 13) v = l[7];
          else if (z == 13) v = (l[6] + 3 * l[7] + 2) >> 2;
          else if (!(z & 1)) v = (l[y + (x >> 1)] + l[y + (x >> 1) + 1] + 1) >> 1;
          else v = (l[y + (x >> 1)] + 2 * l[y + (x >> 1) + 1] + l[y + (x >> 1) + 2] + 2) >> 2;
          break;
        }
      }
      out[y * 8 + x] = (uint8_t)v;
    }
#undef T
#undef L
  for (int y = 0; y < 8; ++y) memcpy(dst + y * stride, out + y * 8, 8);
}

void h264i_pred16x16(uint8_t* dst, int stride, int mode, int has_left, int has_top) {
  int t[16], l[16], tl = has_left && has_top ? dst[-stride - 1] : 128;
  for (int i = 0; i < 16; ++i) {
    t[i] = has_top ? dst[-stride + i] : 128;
    l[i] = has_left ? dst[i * stride - 1] : 128;
  }
  int dc = 128, a = 0, bb = 0, cc = 0;
  if (mode == 2) {
    int st = 0, sl = 0;
    for (int i = 0; i < 16; ++i) {
      st += t[i];
      sl += l[i];
    }
    if (has_top && has_left) dc = (st + sl + 16) >> 5;
    else if (has_left) dc = (sl + 8) >> 4;
    else if (has_top) dc = (st + 8) >> 4;
  } else if (mode == 3) {
    int H = 0, V = 0;
    for (int k = 0; k < 8; ++k) {
      H += (k + 1) * (t[8 + k] - (6 - k >= 0 ? t[6 - k] : tl));
      V += (k + 1) * (l[8 + k] - (6 - k >= 0 ? l[6 - k] : tl));
    }
    a = 16 * (l[15] + t[15]);
    bb = (5 * H + 32) >> 6;
    cc = (5 * V + 32) >> 6;
  }
  for (int y = 0; y < 16; ++y)
    for (int x = 0; x < 16; ++x) {
      int v = mode == 0 ? t[x] : mode == 1 ? l[y] : mode == 2 ? dc : clip8((a + bb * (x - 7) + cc * (y - 7) + 16) >> 5);
      dst[y * stride + x] = (uint8_t)v;
    }
}

void h264i_pred_chroma8x8(uint8_t* dst, int stride, int mode, int has_left, int has_top) {
  int t[8], l[8], tl = has_left && has_top ? dst[-stride - 1] : 128;
  for (int i = 0; i < 8; ++i) {
    t[i] = has_top ? dst[-stride + i] : 128;
    l[i] = has_left ? dst[i * stride - 1] : 128;
  }
  if (mode == 0) {
    for (int by = 0; by < 2; ++by)
      for (int bx = 0; bx < 2; ++bx) {
        int st = t[4 * bx] + t[4 * bx + 1] + t[4 * bx + 2] + t[4 * bx + 3];
        int sl = l[4 * by] + l[4 * by + 1] + l[4 * by + 2] + l[4 * by + 3];
        int dc;
        if (bx == by) {
          if (has_top && has_left) dc = (st + sl + 4) >> 3;
          else if (has_top) dc = (st + 2) >> 2;
          else if (has_left) dc = (sl + 2) >> 2;
          else dc = 128;
        } else if (bx == 1) { /* top-right: prefers top */
          if (has_top) dc = (st + 2) >> 2;
          else if (has_left) dc = (sl + 2) >> 2;
          else dc = 128;
        } else { /* bottom-left: prefers left */
          if (has_left) dc = (sl + 2) >> 2;
          else if (has_top) dc = (st + 2) >> 2;
          else dc = 128;
        }
        for (int y = 0; y < 4; ++y)
          for (int x = 0; x < 4; ++x) dst[(4 * by + y) * stride + 4 * bx + x] = (uint8_t)dc;
      }
    return;
  }
  int a = 0, bb = 0, cc = 0;
  if (mode == 3) {
    int H = 0, V = 0;
    for (int k = 0; k < 4; ++k) {
      H += (k + 1) * (t[4 + k] - (2 - k >= 0 ? t[2 - k] : tl));
      V += (k + 1) * (l[4 + k] - (2 - k >= 0 ? l[2 - k] : tl));
    }
    a = 16 * (l[7] + t[7]);
    bb = (34 * H + 32) >> 6;
    cc = (34 * V + 32) >> 6;
  }
  for (int y = 0; y < 8; ++y)
    for (int x = 0; x < 8; ++x) {
      int v = mode == 1 ? l[y] : mode == 2 ? t[x] : clip8((a + bb * (x - 3) + cc * (y - 3) + 16) >> 5);
      dst[y * stride + x] = (uint8_t)v;
    }
}

/* ======================================================================== */
/* inter prediction                                                         */
/* ======================================================================== */
static inline int ref_px(const Picture* p, int plane, int w, int h, int x, int y) {
  x = iclip(0, w - 1, x);
  y = iclip(0, h - 1, y);
  return p->plane[plane][y * p->stride[plane] + x];
}

static inline int tap6(int a, int b, int c, int dd, int e, int f) { return a - 5 * b + 20 * c + 20 * dd - 5 * e + f; }

/* one luma sample at integer (x, y) + fraction (fx, fy) in quarter units */
static int luma_sample(const Picture* p, int w, int h, int x, int y, int fx, int fy) {
#define PX(xx, yy) ref_px(p, 0, w, h, (xx), (yy))
#define B1(xx, yy) tap6(PX((xx)-2, yy), PX((xx)-1, yy), PX(xx, yy), PX((xx) + 1, yy), PX((xx) + 2, yy), PX((xx) + 3, yy))
#define H1(xx, yy) tap6(PX(xx, (yy)-2), PX(xx, (yy)-1), PX(xx, yy), PX(xx, (yy) + 1), PX(xx, (yy) + 2), PX(xx, (yy) + 3))
#define HB(xx, yy) clip8((B1(xx, yy) + 16) >> 5)
#define HH(xx, yy) clip8((H1(xx, yy) + 16) >> 5)
  if (fx == 0 && fy == 0) return PX(x, y);
  int G = PX(x, y);
  if (fy == 0) {
    int b = HB(x, y);
    return fx == 2 ? b : (fx == 1 ? (G + b + 1) >> 1 : (PX(x + 1, y) + b + 1) >> 1);
  }
  if (fx == 0) {
    int hh = HH(x, y);
    return fy == 2 ? hh : (fy == 1 ? (G + hh + 1) >> 1 : (PX(x, y + 1) + hh + 1) >> 1);
  }
  if (fx == 2 || fy == 2) {
    int j1 = tap6(B1(x, y - 2), B1(x, y - 1), B1(x, y), B1(x, y + 1), B1(x, y + 2), B1(x, y + 3));
    int j = clip8((j1 + 512) >> 10);
    if (fx == 2 && fy == 2) return j;
    if (fx == 2) return fy == 1 ? (HB(x, y) + j + 1) >> 1 : (HB(x, y + 1) + j + 1) >> 1;
    return fx == 1 ? (HH(x, y) + j + 1) >> 1 : (HH(x + 1, y) + j + 1) >> 1;
  }
  /* diagonal quarter positions e, g, p, r */
  int b = fy == 1 ? HB(x, y) : HB(x, y + 1);
  int hh = fx == 1 ? HH(x, y) : HH(x + 1, y);
  return (b + hh + 1) >> 1;
#undef PX
#undef B1
#undef H1
#undef HB
#undef HH
}

static int chroma_sample(const Picture* p, int plane, int w, int h, int x, int y, int fx, int fy) {
  int A = ref_px(p, plane, w, h, x, y), B = ref_px(p, plane, w, h, x + 1, y);
  int C = ref_px(p, plane, w, h, x, y + 1), D = ref_px(p, plane, w, h, x + 1, y + 1);
  return ((8 - fx) * (8 - fy) * A + fx * (8 - fy) * B + (8 - fx) * fy * C + fx * fy * D + 32) >> 6;
}

/* Block form of the two sample functions above: identical arithmetic,
 * evaluated separably over a (w+5) x (h+5) window. */
void h264i_mc_block(const Picture* ref, int plane, int pw, int ph, int x, int y, int mvx, int mvy,
                    int w, int h, uint8_t* dst, int ds) {
  if (plane != 0) {
    const int fx = mvx & 7, fy = mvy & 7, ix = x + (mvx >> 3), iy = y + (mvy >> 3);
    for (int j = 0; j < h; ++j)
      for (int i = 0; i < w; ++i) dst[j * ds + i] = (uint8_t)chroma_sample(ref, plane, pw, ph, ix + i, iy + j, fx, fy);
    return;
  }
  const int fx = mvx & 3, fy = mvy & 3, ix = x + (mvx >> 2), iy = y + (mvy >> 2);
  int src[21][22]; /* rows -2..h+2, cols -2..w+3 */
  for (int j = 0; j < h + 5; ++j) {
    const int yy = iclip(0, ph - 1, iy - 2 + j);
    const uint8_t* row = ref->plane[0] + yy * ref->stride[0];
    for (int i = 0; i < w + 6; ++i) src[j][i] = row[iclip(0, pw - 1, ix - 2 + i)];
  }
#define G(i, j) src[(j) + 2][(i) + 2]
  if (fx == 0 && fy == 0) {
    for (int j = 0; j < h; ++j)
      for (int i = 0; i < w; ++i) dst[j * ds + i] = (uint8_t)G(i, j);
    return;
  }
  int B1[21][17]; /* horizontal 6-tap, rows -2..h+2, cols 0..w */
  int H1[16][17]; /* vertical 6-tap, rows 0..h-1, cols 0..w */
  const int need_b = fy == 0 || fx == 2 || (fx & 1 && fy & 1) || fy == 2;
  const int need_h = fx == 0 || fy == 2 || (fx & 1 && fy & 1) || fx == 2;
  if (need_b || fx == 2 || fy == 2)
    for (int j = 0; j < h + 5; ++j)
      for (int i = 0; i <= w; ++i)
        B1[j][i] = tap6(src[j][i], src[j][i + 1], src[j][i + 2], src[j][i + 3], src[j][i + 4], src[j][i + 5]);
  if (need_h)
    for (int j = 0; j < h; ++j)
      for (int i = 0; i <= w; ++i)
        H1[j][i] = tap6(src[j][i + 2], src[j + 1][i + 2], src[j + 2][i + 2], src[j + 3][i + 2], src[j + 4][i + 2],
                        src[j + 5][i + 2]);
#define HB(i, j) clip8((B1[(j) + 2][i] + 16) >> 5)
#define HH(i, j) clip8((H1[j][i] + 16) >> 5)
#define JJ(i, j) \
  clip8((tap6(B1[j][i], B1[(j) + 1][i], B1[(j) + 2][i], B1[(j) + 3][i], B1[(j) + 4][i], B1[(j) + 5][i]) + 512) >> 10)
  for (int j = 0; j < h; ++j)
    for (int i = 0; i < w; ++i) {
      int v;
      if (fy == 0) {
        int b = HB(i, j);
        v = fx == 2 ? b : (fx == 1 ? (G(i, j) + b + 1) >> 1 : (G(i + 1, j) + b + 1) >> 1);
      } else if (fx == 0) {
        int hh = HH(i, j);
        v = fy == 2 ? hh : (fy == 1 ? (G(i, j) + hh + 1) >> 1 : (G(i, j + 1) + hh + 1) >> 1);
      } else if (fx == 2 || fy == 2) {
        int jj = JJ(i, j);
        if (fx == 2 && fy == 2) v = jj;
        else if (fx == 2) v = fy == 1 ? (HB(i, j) + jj + 1) >> 1 : (HB(i, j + 1) + jj + 1) >> 1;
        else v = fx == 1 ? (HH(i, j) + jj + 1) >> 1 : (HH(i + 1, j) + jj + 1) >> 1;
      } else {
        int b = fy == 1 ? HB(i, j) : HB(i, j + 1);
        int hh = fx == 1 ? HH(i, j) : HH(i + 1, j);
        v = (b + hh + 1) >> 1;
      }
      dst[j * ds + i] = (uint8_t)v;
    }
#undef G
#undef HB
#undef HH
#undef JJ
}

/* Inter prediction of the current macroblock from d->mb.{ref,mv} into the
 * current picture.  Prediction is defined per sample, so blocks that share
 * motion are predicted in one piece (16x16 / 8x8 / 4x4) purely for speed. */
static void inter_pred_region(H264Oracle* d, int bx, int by, int n4) {
  const MB* m = &d->mb;
  const Slice* sl = &d->sl;
  Picture* cur = d->cur;
  const int W = d->mbw * 16, H = d->mbh * 16;
  const int blk = by * 4 + bx, b8 = (by >> 1) * 2 + (bx >> 1);
  const int r0 = m->ref[0][b8], r1 = m->ref[1][b8];
  const Picture* p0 = r0 >= 0 ? sl->ref_list[0][r0] : NULL;
  const Picture* p1 = r1 >= 0 ? sl->ref_list[1][r1] : NULL;
  for (int plane = 0; plane < (d->pred_luma_only ? 1 : 3); ++plane) {
    const int n = plane ? n4 * 2 : n4 * 4;
    const int pw = plane ? W / 2 : W, ph = plane ? H / 2 : H;
    const int x0 = (m->mbx * 16 + bx * 4) >> (plane ? 1 : 0), y0 = (m->mby * 16 + by * 4) >> (plane ? 1 : 0);
    uint8_t* dst = cur->plane[plane] + y0 * cur->stride[plane] + x0;
    uint8_t pb[2][256];
    if (plane == 0) COV(d, H264O_COV_WP + sl->use_weight * 3 + (p0 && p1 ? 2 : (p0 ? 0 : 1)));
    for (int l = 0; l < 2; ++l) {
      const Picture* p = l ? p1 : p0;
      if (!p) continue;
      if (plane == 0) COV(d, H264O_COV_LUMA_MC + ((m->mv[l][blk][1] & 3) * 4 + (m->mv[l][blk][0] & 3)) * 3 + (n4 == 4 ? 0 : (n4 == 2 ? 1 : 2)));
      else if (plane == 1) COV(d, H264O_COV_CHROMA_MC + (m->mv[l][blk][1] & 7) * 8 + (m->mv[l][blk][0] & 7));
      h264i_mc_block(p, plane, pw, ph, x0, y0, m->mv[l][blk][0], m->mv[l][blk][1], n, n, pb[l], 16);
    }
    for (int y = 0; y < n; ++y)
      for (int x = 0; x < n; ++x) {
        int s[2] = {p0 ? pb[0][y * 16 + x] : 0, p1 ? pb[1][y * 16 + x] : 0};
        int v;
        if (sl->use_weight == 1) {
          const int ld = plane ? sl->chroma_log2_denom : sl->luma_log2_denom;
          int w0 = 0, w1 = 0, o0 = 0, o1 = 0;
          if (p0) {
            w0 = plane ? sl->chroma_weight[0][r0][plane - 1] : sl->luma_weight[0][r0];
            o0 = plane ? sl->chroma_offset[0][r0][plane - 1] : sl->luma_offset[0][r0];
          }
          if (p1) {
            w1 = plane ? sl->chroma_weight[1][r1][plane - 1] : sl->luma_weight[1][r1];
            o1 = plane ? sl->chroma_offset[1][r1][plane - 1] : sl->luma_offset[1][r1];
          }
          if (p0 && p1)
            v = ((s[0] * w0 + s[1] * w1 + (1 << ld)) >> (ld + 1)) + ((o0 + o1 + 1) >> 1);
          else {
            int sv = p0 ? s[0] : s[1], w = p0 ? w0 : w1, o = p0 ? o0 : o1;
            v = ld >= 1 ? ((sv * w + (1 << (ld - 1))) >> ld) + o : sv * w + o;
          }
          v = clip8(v);
        } else if (p0 && p1) {
          if (sl->use_weight == 2) {
            int w1 = sl->implicit_weight[r0][r1], w0 = 64 - w1;
            v = clip8((s[0] * w0 + s[1] * w1 + 32) >> 6);
          } else {
            v = (s[0] + s[1] + 1) >> 1;
          }
        } else {
          v = p0 ? s[0] : s[1];
        }
        dst[y * cur->stride[plane] + x] = (uint8_t)v;
      }
  }
}

static int same_motion(const MB* m, int a, int b) {
  const int a8 = (a >> 3) * 2 + ((a & 3) >> 1), b8 = (b >> 3) * 2 + ((b & 3) >> 1);
  for (int l = 0; l < 2; ++l) {
    if (m->ref[l][a8] != m->ref[l][b8]) return 0;
    if (m->ref[l][a8] >= 0 && (m->mv[l][a][0] != m->mv[l][b][0] || m->mv[l][a][1] != m->mv[l][b][1])) return 0;
  }
  return 1;
}

void h264i_inter_pred_mb(H264Oracle* d) {
  const MB* m = &d->mb;
  int uniform = 1;
  for (int k = 1; k < 16 && uniform; ++k) uniform = same_motion(m, 0, k);
  if (uniform) {
    inter_pred_region(d, 0, 0, 4);
    return;
  }
  for (int q = 0; q < 4; ++q) {
    const int bx = (q & 1) * 2, by = (q >> 1) * 2, b = by * 4 + bx;
    if (same_motion(m, b, b + 1) && same_motion(m, b, b + 4) && same_motion(m, b, b + 5)) {
      inter_pred_region(d, bx, by, 2);
    } else {
      inter_pred_region(d, bx, by, 1);
      inter_pred_region(d, bx + 1, by, 1);
      inter_pred_region(d, bx, by + 1, 1);
      inter_pred_region(d, bx + 1, by + 1, 1);
    }
  }
}

/* ======================================================================== */
/* macroblock reconstruction                                                */
/* ======================================================================== */
static void recon_luma_residual_4x4(H264Oracle* d, int blk_raster, int dc_given, int dc, const uint8_t* weight) {
  const MB* m = &d->mb;
  Picture* cur = d->cur;
  const int bx = blk_raster & 3, by = blk_raster >> 2;
  uint8_t* dst = cur->plane[0] + (m->mby * 16 + by * 4) * cur->stride[0] + m->mbx * 16 + bx * 4;
  int any = dc_given && dc;
  for (int k = 0; k < 16 && !any; ++k) any = m->luma[blk_raster][k] != 0;
  if (!any) return;
  int sc[16], res[16];
  COV(d, H264O_COV_DEQUANT + (m->qp >= 24));
  scale4x4(m->luma[blk_raster], weight, m->qp, dc_given, dc, sc);
  idct4x4(sc, res);
  add_residual(dst, cur->stride[0], res, 4);
}

static void recon_luma_residual_8x8(H264Oracle* d, int b8, const uint8_t* weight) {
  const MB* m = &d->mb;
  Picture* cur = d->cur;
  uint8_t* dst = cur->plane[0] + (m->mby * 16 + (b8 >> 1) * 8) * cur->stride[0] + m->mbx * 16 + (b8 & 1) * 8;
  int any = 0;
  for (int k = 0; k < 64 && !any; ++k) any = m->luma8[b8][k] != 0;
  if (!any) return;
  int sc[64], res[64];
  const int qp = m->qp;
  COV(d, H264O_COV_DEQUANT + 2 + (qp >= 36));
  for (int k = 0; k < 64; ++k) {
    int v = m->luma8[b8][k] * h264i_level_scale8(weight, qp, k);
    sc[k] = qp >= 36 ? v << (qp / 6 - 6) : (v + (1 << (5 - qp / 6))) >> (6 - qp / 6);
  }
  idct8x8(sc, res);
  add_residual(dst, cur->stride[0], res, 8);
}

static void recon_chroma_residual(H264Oracle* d) {
  const MB* m = &d->mb;
  Picture* cur = d->cur;
  const int intra = m->kind != MB_INTER;
  if (!(m->cbp >> 4)) return;
  for (int p = 0; p < 2; ++p) {
    const uint8_t* weight = d->apps->scaling4[(intra ? 0 : 3) + 1 + p];
    const int qp = m->qpc[p];
    const int16_t* c = m->chroma_dc[p];
    int f[4] = {c[0] + c[1] + c[2] + c[3], c[0] - c[1] + c[2] - c[3], c[0] + c[1] - c[2] - c[3],
                c[0] - c[1] - c[2] + c[3]};
    const int ls = h264i_level_scale4(weight, qp, 0);
    if (f[0] | f[1] | f[2] | f[3]) COV(d, H264O_COV_DEQUANT + 6);
    for (int s = 0; s < 4; ++s) {
      int dc = ((f[s] * ls) << (qp / 6)) >> 5;
      int sc[16], res[16];
      scale4x4(m->chroma_ac[p][s], weight, qp, 1, dc, sc);
      int any = 0;
      for (int k = 0; k < 16 && !any; ++k) any = sc[k] != 0;
      if (!any) continue;
      idct4x4(sc, res);
      uint8_t* dst = cur->plane[1 + p] + (m->mby * 8 + (s >> 1) * 4) * cur->stride[1 + p] + m->mbx * 8 + (s & 1) * 4;
      add_residual(dst, cur->stride[1 + p], res, 4);
    }
  }
}

void h264i_recon_mb(H264Oracle* d) {
  MB* m = &d->mb;
  Picture* cur = d->cur;
  const PPS* pps = d->apps;
  uint8_t* y0 = cur->plane[0] + m->mby * 16 * cur->stride[0] + m->mbx * 16;
  const int sy = cur->stride[0];

  if (m->kind == MB_IPCM) {
    for (int y = 0; y < 16; ++y) memcpy(y0 + y * sy, m->pcm + y * 16, 16);
    for (int p = 0; p < 2; ++p)
      for (int y = 0; y < 8; ++y)
        memcpy(cur->plane[1 + p] + (m->mby * 8 + y) * cur->stride[1 + p] + m->mbx * 8, m->pcm + 256 + p * 64 + y * 8, 8);
    return;
  }

  if (m->kind == MB_INTER) {
    h264i_inter_pred_mb(d);
    const uint8_t* w4 = pps->scaling4[3];
    if (m->t8x8) {
      for (int b8 = 0; b8 < 4; ++b8) recon_luma_residual_8x8(d, b8, pps->scaling8[1]);
    } else {
      for (int k = 0; k < 16; ++k) recon_luma_residual_4x4(d, k, 0, 0, w4);
    }
    recon_chroma_residual(d);
    return;
  }

  const int left_ok = h264i_intra_mb_ok(d, m->mbx - 1, m->mby), top_ok = h264i_intra_mb_ok(d, m->mbx, m->mby - 1);
  if (m->kind == MB_I16x16) {
    h264i_pred16x16(y0, sy, m->i16_mode, left_ok, top_ok);
    COV(d, H264O_COV_I16 + m->i16_mode * 4 + (left_ok | top_ok << 1));
    /* DC: inverse 4x4 Hadamard, then scaling (8.5.10) */
    const uint8_t* w4 = pps->scaling4[0];
    int c[16], f[16], t[16];
    for (int k = 0; k < 16; ++k) c[k] = m->luma_dc[k];
    for (int i = 0; i < 4; ++i) { /* rows */
      int a = c[4 * i], b = c[4 * i + 1], cc = c[4 * i + 2], dd = c[4 * i + 3];
      t[4 * i] = a + b + cc + dd;
      t[4 * i + 1] = a + b - cc - dd;
      t[4 * i + 2] = a - b - cc + dd;
      t[4 * i + 3] = a - b + cc - dd;
    }
    for (int j = 0; j < 4; ++j) {
      int a = t[j], b = t[4 + j], cc = t[8 + j], dd = t[12 + j];
      f[j] = a + b + cc + dd;
      f[4 + j] = a + b - cc - dd;
      f[8 + j] = a - b - cc + dd;
      f[12 + j] = a - b + cc - dd;
    }
    const int qp = m->qp, ls = h264i_level_scale4(w4, qp, 0);
    if (m->cbf_dc & 1) COV(d, H264O_COV_DEQUANT + 4 + (qp >= 36));
    for (int k = 0; k < 16; ++k) {
      int dc = qp >= 36 ? (f[k] * ls) << (qp / 6 - 6) : (f[k] * ls + (1 << (5 - qp / 6))) >> (6 - qp / 6);
      recon_luma_residual_4x4(d, k, 1, dc, w4); /* f is indexed [row=by][col=bx] = raster block */
    }
  } else if (m->kind == MB_I4x4) {
    const uint8_t* w4 = pps->scaling4[0];
    for (int k = 0; k < 16; ++k) {
      const int bx = blk4_x[k], by = blk4_y[k];
      uint8_t* dst = y0 + by * 4 * sy + bx * 4;
      int hl = h264i_intra_blk_ok(d, bx - 1, by, k), ht = h264i_intra_blk_ok(d, bx, by - 1, k);
      int htl = h264i_intra_blk_ok(d, bx - 1, by - 1, k), htr = h264i_intra_blk_ok(d, bx + 1, by - 1, k);
      h264i_pred4x4(dst, sy, m->i4_mode[by * 4 + bx], hl, ht, htl, htr);
      COV(d, H264O_COV_I4 + m->i4_mode[by * 4 + bx] * 16 + (hl | ht << 1 | htl << 2 | htr << 3));
      recon_luma_residual_4x4(d, by * 4 + bx, 0, 0, w4);
    }
  } else { /* MB_I8x8 */
    for (int b8 = 0; b8 < 4; ++b8) {
      const int bx = (b8 & 1) * 2, by = (b8 >> 1) * 2;
      uint8_t* dst = y0 + by * 4 * sy + bx * 4;
      const int zz = b8 * 4;
      int hl = h264i_intra_blk_ok(d, bx - 1, by, zz), ht = h264i_intra_blk_ok(d, bx, by - 1, zz);
      int htl = h264i_intra_blk_ok(d, bx - 1, by - 1, zz), htr = h264i_intra_blk_ok(d, bx + 2, by - 1, zz);
      h264i_pred8x8l(dst, sy, m->i4_mode[by * 4 + bx], hl, ht, htl, htr);
      COV(d, H264O_COV_I8 + m->i4_mode[by * 4 + bx] * 16 + (hl | ht << 1 | htl << 2 | htr << 3));
      recon_luma_residual_8x8(d, b8, pps->scaling8[0]);
    }
  }
  for (int p = 0; p < 2; ++p) {
    uint8_t* c0 = cur->plane[1 + p] + m->mby * 8 * cur->stride[1 + p] + m->mbx * 8;
    h264i_pred_chroma8x8(c0, cur->stride[1 + p], m->chroma_mode, left_ok, top_ok);
  }
  COV(d, H264O_COV_IC + m->chroma_mode * 4 + (left_ok | top_ok << 1));
  recon_chroma_residual(d);
}

/* ======================================================================== */
/* deblocking filter (8.7)                                                  */
/* ======================================================================== */
static void filter_edge_sample(uint8_t* pix, int step, int bs, int alpha, int beta, int tc0, int chroma, int64_t* cov) {
  /* pix points at q0; p_i = pix[-(i+1)*step], q_i = pix[i*step] */
#define FCOV(i) do { if (cov) cov[H264O_COV_FILT + (i)]++; } while (0)
  int p0 = pix[-step], p1 = pix[-2 * step], q0 = pix[0], q1 = pix[step];
  if (iabs(p0 - q0) >= alpha || iabs(p1 - p0) >= beta || iabs(q1 - q0) >= beta) {
    FCOV(10 + (chroma != 0));
    return;
  }
  if (bs < 4) {
    int tc;
    if (chroma) {
      tc = tc0 + 1;
      FCOV(8);
    } else {
      int p2 = pix[-3 * step], q2 = pix[2 * step];
      int ap = iabs(p2 - p0), aq = iabs(q2 - q0);
      FCOV((ap < beta) * 2 + (aq < beta));
      tc = tc0 + (ap < beta) + (aq < beta);
      if (ap < beta) pix[-2 * step] = (uint8_t)(p1 + iclip(-tc0, tc0, (p2 + ((p0 + q0 + 1) >> 1) - (p1 << 1)) >> 1));
      if (aq < beta) pix[step] = (uint8_t)(q1 + iclip(-tc0, tc0, (q2 + ((p0 + q0 + 1) >> 1) - (q1 << 1)) >> 1));
    }
    int delta = iclip(-tc, tc, (((q0 - p0) << 2) + (p1 - q1) + 4) >> 3);
    pix[-step] = (uint8_t)clip8(p0 + delta);
    pix[0] = (uint8_t)clip8(q0 - delta);
    return;
  }
  if (chroma) {
    FCOV(9);
    pix[-step] = (uint8_t)((2 * p1 + p0 + q1 + 2) >> 2);
    pix[0] = (uint8_t)((2 * q1 + q0 + p1 + 2) >> 2);
    return;
  }
  int p2 = pix[-3 * step], q2 = pix[2 * step];
  int ap = iabs(p2 - p0), aq = iabs(q2 - q0);
  int strong = iabs(p0 - q0) < ((alpha >> 2) + 2);
  FCOV(4 + (ap < beta && strong) * 2 + (aq < beta && strong));
#undef FCOV
  if (ap < beta && strong) {
    int p3 = pix[-4 * step];
    pix[-step] = (uint8_t)((p2 + 2 * p1 + 2 * p0 + 2 * q0 + q1 + 4) >> 3);
    pix[-2 * step] = (uint8_t)((p2 + p1 + p0 + q0 + 2) >> 2);
    pix[-3 * step] = (uint8_t)((2 * p3 + 3 * p2 + p1 + p0 + q0 + 4) >> 3);
  } else {
    pix[-step] = (uint8_t)((2 * p1 + p0 + q1 + 2) >> 2);
  }
  if (aq < beta && strong) {
    int q3 = pix[3 * step];
    pix[0] = (uint8_t)((p1 + 2 * p0 + 2 * q0 + 2 * q1 + q2 + 4) >> 3);
    pix[step] = (uint8_t)((p0 + q0 + q1 + q2 + 2) >> 2);
    pix[2 * step] = (uint8_t)((2 * q3 + 3 * q2 + q1 + q0 + p0 + 4) >> 3);
  } else {
    pix[0] = (uint8_t)((2 * q1 + q0 + p1 + 2) >> 2);
  }
}

/* boundary strength between the 4x4 blocks (pa, pblk) and (qa, qblk) */
static int boundary_strength(const H264Oracle* d, int pa, int pblk, int qa, int qblk, int mb_edge) {
  const Picture* cur = d->cur;
  if (cur->mb_intra[pa] || cur->mb_intra[qa]) return mb_edge ? 4 : 3;
  /* non-zero coefficients in the transform block containing the sample */
  int pn, qn;
  if (d->mb_t8x8[pa]) {
    int q = (pblk >> 3) * 2 + ((pblk & 3) >> 1);
    int o = (q >> 1) * 8 + (q & 1) * 2;
    pn = d->nnz[pa * 24 + o] | d->nnz[pa * 24 + o + 1] | d->nnz[pa * 24 + o + 4] | d->nnz[pa * 24 + o + 5];
  } else {
    pn = d->nnz[pa * 24 + pblk];
  }
  if (d->mb_t8x8[qa]) {
    int q = (qblk >> 3) * 2 + ((qblk & 3) >> 1);
    int o = (q >> 1) * 8 + (q & 1) * 2;
    qn = d->nnz[qa * 24 + o] | d->nnz[qa * 24 + o + 1] | d->nnz[qa * 24 + o + 4] | d->nnz[qa * 24 + o + 5];
  } else {
    qn = d->nnz[qa * 24 + qblk];
  }
  if (pn || qn) return 2;
  const int p8 = pa * 4 + (pblk >> 3) * 2 + ((pblk & 3) >> 1), q8 = qa * 4 + (qblk >> 3) * 2 + ((qblk & 3) >> 1);
  int pr0 = cur->ref_idx[0][p8] >= 0 ? cur->ref_uid[0][p8] : -1;
  int pr1 = cur->ref_idx[1][p8] >= 0 ? cur->ref_uid[1][p8] : -1;
  int qr0 = cur->ref_idx[0][q8] >= 0 ? cur->ref_uid[0][q8] : -1;
  int qr1 = cur->ref_idx[1][q8] >= 0 ? cur->ref_uid[1][q8] : -1;
  const int16_t* pm0 = cur->mv[0][pa * 16 + pblk];
  const int16_t* pm1 = cur->mv[1][pa * 16 + pblk];
  const int16_t* qm0 = cur->mv[0][qa * 16 + qblk];
  const int16_t* qm1 = cur->mv[1][qa * 16 + qblk];
#define FAR(a, b) (iabs((a)[0] - (b)[0]) >= 4 || iabs((a)[1] - (b)[1]) >= 4)
  /* same set of reference pictures? */
  if (!((pr0 == qr0 && pr1 == qr1) || (pr0 == qr1 && pr1 == qr0))) return 1;
  if (pr0 == pr1) {
    if (pr0 < 0) return 0; /* cannot happen for inter blocks */
    /* both lists point at the same picture */
    int straight = FAR(pm0, qm0) || FAR(pm1, qm1);
    int cross = FAR(pm0, qm1) || FAR(pm1, qm0);
    return (straight && cross) ? 1 : 0;
  }
  if (pr0 == qr0) { /* lists line up */
    if (pr0 >= 0 && FAR(pm0, qm0)) return 1;
    if (pr1 >= 0 && FAR(pm1, qm1)) return 1;
    return 0;
  }
  /* crossed: p's L0 picture is q's L1 picture */
  if (pr0 >= 0 && FAR(pm0, qm1)) return 1;
  if (pr1 >= 0 && FAR(pm1, qm0)) return 1;
  return 0;
#undef FAR
}

void h264i_deblock_picture(H264Oracle* d) {
  Picture* cur = d->cur;
  for (int mby = 0; mby < d->mbh; ++mby)
    for (int mbx = 0; mbx < d->mbw; ++mbx) {
      const int a = mby * d->mbw + mbx;
      const MbDeblock db = d->mb_deblock[a];
      if (db.disable == 1) continue;
      const int t8 = d->mb_t8x8[a];
      for (int dir = 0; dir < 2; ++dir) { /* 0: vertical edges, 1: horizontal */
        for (int e = 0; e < 4; ++e) {
          int na = a; /* MB holding the p samples */
          if (e == 0) {
            if (dir == 0) {
              if (mbx == 0) continue;
              na = a - 1;
            } else {
              if (mby == 0) continue;
              na = a - d->mbw;
            }
            if (db.disable == 2 && d->mb_slice[na] != d->mb_slice[a]) continue;
          }
          const int luma_edge = !(t8 && (e & 1));
          int bs[4];
          for (int k = 0; k < 4; ++k) {
            int qblk = dir == 0 ? k * 4 + e : e * 4 + k;
            int pblk = e == 0 ? (dir == 0 ? k * 4 + 3 : 12 + k) : (dir == 0 ? k * 4 + e - 1 : (e - 1) * 4 + k);
            bs[k] = boundary_strength(d, na, pblk, a, qblk, e == 0);
          }
          if (d->cov)
            for (int k = 0; k < 4; ++k) {
              if (luma_edge) COV(d, H264O_COV_BS + ((dir * 2 + (e == 0)) * 3 + 0) * 5 + bs[k]);
              if (!(e & 1)) {
                COV(d, H264O_COV_BS + ((dir * 2 + (e == 0)) * 3 + 1) * 5 + bs[k]);
                COV(d, H264O_COV_BS + ((dir * 2 + (e == 0)) * 3 + 2) * 5 + bs[k]);
              }
            }
          if (!(bs[0] | bs[1] | bs[2] | bs[3])) continue;
          /* luma */
          if (luma_edge) {
            int qpav = (d->mb_qp[na] + d->mb_qp[a] + 1) >> 1;
            int ia = iclip(0, 51, qpav + db.alpha), ib = iclip(0, 51, qpav + db.beta);
            int alpha = deblock_alpha[ia], beta = deblock_beta[ib];
            uint8_t* base = cur->plane[0] + mby * 16 * cur->stride[0] + mbx * 16;
            for (int i = 0; i < 16; ++i) {
              int s = bs[i >> 2];
              if (!s) continue;
              uint8_t* pix = dir == 0 ? base + i * cur->stride[0] + e * 4 : base + e * 4 * cur->stride[0] + i;
              filter_edge_sample(pix, dir == 0 ? 1 : cur->stride[0], s, alpha, beta, s < 4 ? deblock_tc0[ia][s - 1] : 0, 0, d->cov);
            }
          }
          /* chroma: 8x8 planes have edges at 0 and 4, i.e. luma edges 0 and 2 */
          if (!(e & 1)) {
            for (int p = 0; p < 2; ++p) {
              int qpav = (d->mb_qpc[p][na] + d->mb_qpc[p][a] + 1) >> 1;
              int ia = iclip(0, 51, qpav + db.alpha), ib = iclip(0, 51, qpav + db.beta);
              int alpha = deblock_alpha[ia], beta = deblock_beta[ib];
              const int cs = cur->stride[1 + p];
              uint8_t* base = cur->plane[1 + p] + mby * 8 * cs + mbx * 8;
              for (int i = 0; i < 8; ++i) {
                int s = bs[i >> 1];
                if (!s) continue;
                uint8_t* pix = dir == 0 ? base + i * cs + e * 2 : base + e * 2 * cs + i;
                filter_edge_sample(pix, dir == 0 ? 1 : cs, s, alpha, beta, s < 4 ? deblock_tc0[ia][s - 1] : 0, 1, d->cov);
              }
            }
          }
        }
      }
    }
}
