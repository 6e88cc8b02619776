/* TEST INFRASTRUCTURE - HEVC oracle: from parsed coding units to samples.  8.4.4.2 intra sample prediction (reference
 * substitution, smoothing, planar / DC / angular), 8.5.3.3 fractional sample interpolation and weighted prediction,
 * 8.6.2 - 8.6.4 scaling and inverse transforms (DCT 4..32, DST 4, transform skip, bypass), 8.7.2 deblocking, 8.7.3
 * sample adaptive offset.  Scalar, one sample at a time, written for reading.  See hevc_oracle.h: parity unpinned. */
#include "hevc_internal.h"

/* ------------------------------------------------------------------------------------------------ intra, 8.4.4.2 */
/* left[0] = top[0] = p[-1][-1]; left[1 + k] = p[-1][k]; top[1 + k] = p[k][-1]; k = 0 .. 2n - 1 */
static void intra_filter_refs(int* left, int* top, int n, int c_idx, int mode, int strong_enabled) {
  if (c_idx != 0 || mode == 1 || n == 4) return;
  const int dist = hv_min(hv_abs(mode - 26), hv_abs(mode - 10));
  const int thres = n == 8 ? 7 : n == 16 ? 1 : 0;
  if (dist <= thres) return;
  int fl[130], ft[130];
  const int corner = left[0];
  if (strong_enabled && n == 32 && hv_abs(corner + top[2 * n] - 2 * top[n]) < 8 && hv_abs(corner + left[2 * n] - 2 * left[n]) < 8) {
    fl[0] = ft[0] = corner;
    for (int k = 0; k < 63; ++k) {
      fl[1 + k] = ((63 - k) * corner + (k + 1) * left[64] + 32) >> 6;
      ft[1 + k] = ((63 - k) * corner + (k + 1) * top[64] + 32) >> 6;
    }
    fl[64] = left[64];
    ft[64] = top[64];
  } else {
    fl[0] = ft[0] = (left[1] + 2 * corner + top[1] + 2) >> 2;
    for (int k = 1; k < 2 * n; ++k) {
      fl[k] = (left[k + 1] + 2 * left[k] + left[k - 1] + 2) >> 2;
      ft[k] = (top[k + 1] + 2 * top[k] + top[k - 1] + 2) >> 2;
    }
    /* (left[k - 1] for k = 1 is the corner: left[0]) */
    fl[2 * n] = left[2 * n];
    ft[2 * n] = top[2 * n];
  }
  memcpy(left, fl, sizeof(int) * (size_t)(2 * n + 1));
  memcpy(top, ft, sizeof(int) * (size_t)(2 * n + 1));
}

static void intra_pred_core(const int* left, const int* top, int n, int log2, int c_idx, int mode, uint8_t* dst, int stride) {
  if (mode == 0) { /* planar, 8.4.4.2.4 */
    for (int y = 0; y < n; ++y)
      for (int x = 0; x < n; ++x)
        dst[y * stride + x] = (uint8_t)(((n - 1 - x) * left[1 + y] + (x + 1) * top[1 + n] + (n - 1 - y) * top[1 + x] + (y + 1) * left[1 + n] + n) >> (log2 + 1));
    return;
  }
  if (mode == 1) { /* DC, 8.4.4.2.5 */
    int sum = n;
    for (int k = 0; k < n; ++k) sum += left[1 + k] + top[1 + k];
    const int dc = sum >> (log2 + 1);
    for (int y = 0; y < n; ++y)
      for (int x = 0; x < n; ++x) dst[y * stride + x] = (uint8_t)dc;
    if (c_idx == 0 && n < 32) {
      dst[0] = (uint8_t)((left[1] + 2 * dc + top[1] + 2) >> 2);
      for (int x = 1; x < n; ++x) dst[x] = (uint8_t)((top[1 + x] + 3 * dc + 2) >> 2);
      for (int y = 1; y < n; ++y) dst[y * stride] = (uint8_t)((left[1 + y] + 3 * dc + 2) >> 2);
    }
    return;
  }
  /* angular, 8.4.4.2.6 */
  const int angle = hevc_intra_angle[mode];
  int ref_buf[3 * 32 + 4];
  int* ref = ref_buf + 32 + 1; /* ref[-n .. 2n] */
  const int vertical = mode >= 18;
  const int* main_ref = vertical ? top : left;  /* main_ref[1 + k] runs along the prediction direction's source */
  const int* side_ref = vertical ? left : top;
  for (int k = 0; k <= n; ++k) ref[k] = main_ref[k]; /* ref[0] = corner, ref[k] = p[-1 + k][-1] */
  if (angle < 0) {
    const int last = (n * angle) >> 5;
    if (last < -1) {
      const int inv = hevc_inv_angle[mode - 11];
      for (int k = -1; k >= last; --k) ref[k] = side_ref[(k * inv + 128) >> 8]; /* p[-1][-1 + ((k * inv + 128) >> 8)] */
    }
  } else {
    for (int k = n + 1; k <= 2 * n; ++k) ref[k] = main_ref[k];
  }
  for (int j = 0; j < n; ++j) {     /* along the prediction direction: rows if vertical, columns if horizontal */
    const int idx = ((j + 1) * angle) >> 5, fact = ((j + 1) * angle) & 31;
    for (int i = 0; i < n; ++i) {
      const int v = fact ? ((32 - fact) * ref[i + idx + 1] + fact * ref[i + idx + 2] + 16) >> 5 : ref[i + idx + 1];
      if (vertical)
        dst[j * stride + i] = (uint8_t)v;
      else
        dst[i * stride + j] = (uint8_t)v;
    }
  }
  if (c_idx == 0 && n < 32) { /* the edge of a purely vertical / horizontal prediction follows the other neighbour's gradient */
    if (mode == 26)
      for (int y = 0; y < n; ++y) dst[y * stride] = (uint8_t)hv_clip8(top[1] + ((left[1 + y] - left[0]) >> 1));
    if (mode == 10)
      for (int x = 0; x < n; ++x) dst[x] = (uint8_t)hv_clip8(left[1] + ((top[1 + x] - top[0]) >> 1));
  }
}

void hv_intra_predict(HevcOracle* d, int c_idx, int x, int y, int log2, int mode) {
  const HvSps* s = d->cur_sps;
  const int n = 1 << log2, sh = c_idx ? 1 : 0;
  const int stride = s->width >> sh;
  uint8_t* plane = d->cur->plane[c_idx];
  const int xl = x << sh, yl = y << sh; /* luma location of the block */
  const int cip = d->cur_pps->constrained_intra_pred;
  int left[130], top[130];
  uint8_t al[130], at[130];
  /* 8.4.4.2.2: which neighbouring samples exist */
  for (int k = -1; k < 2 * n; ++k) {
    const int xn = x - 1, yn = y + k;
    int ok = hv_zscan_available(d, xl, yl, xn << sh, yn << sh);
    if (ok && cip && d->m_pred[((yn << sh) >> 2) * d->w4 + ((xn << sh) >> 2)] != 2) ok = 0;
    al[k + 1] = (uint8_t)ok;
    left[k + 1] = ok ? plane[yn * stride + xn] : 0;
  }
  for (int k = 0; k < 2 * n; ++k) {
    const int xn = x + k, yn = y - 1;
    int ok = hv_zscan_available(d, xl, yl, xn << sh, yn << sh);
    if (ok && cip && d->m_pred[((yn << sh) >> 2) * d->w4 + ((xn << sh) >> 2)] != 2) ok = 0;
    at[k + 1] = (uint8_t)ok;
    top[k + 1] = ok ? plane[yn * stride + xn] : 0;
  }
  at[0] = al[0];
  top[0] = left[0];
  /* substitution: search from the bottom of the left column up to the corner, then along the top row */
  int any = 0;
  for (int k = 0; k <= 2 * n; ++k) any |= al[k] | at[k];
  if (!any) {
    for (int k = 0; k <= 2 * n; ++k) left[k] = top[k] = 128;
  } else {
    if (!al[2 * n]) {
      int found = -1, v = 0;
      for (int k = 2 * n - 1; k >= 0 && found < 0; --k)
        if (al[k]) { found = k; v = left[k]; }
      for (int k = 1; k <= 2 * n && found < 0; ++k)
        if (at[k]) { found = k; v = top[k]; }
      left[2 * n] = v;
      al[2 * n] = 1;
    }
    for (int k = 2 * n - 1; k >= 0; --k)
      if (!al[k]) { left[k] = left[k + 1]; al[k] = 1; }
    top[0] = left[0];
    for (int k = 1; k <= 2 * n; ++k)
      if (!at[k]) top[k] = top[k - 1];
  }
  intra_filter_refs(left, top, n, c_idx, mode, s->strong_intra_smoothing);
  intra_pred_core(left, top, n, log2, c_idx, mode, plane + y * stride + x, stride);
}

void hevco_intra_block(const uint8_t* corner_and_left, const uint8_t* top_in, int log2, int c_idx, int mode, int strong, uint8_t* pred) {
  const int n = 1 << log2;
  int left[130], top[130];
  for (int k = 0; k <= 2 * n; ++k) left[k] = corner_and_left[k];
  top[0] = left[0];
  for (int k = 0; k < 2 * n; ++k) top[1 + k] = top_in[k];
  intra_filter_refs(left, top, n, c_idx, mode, strong);
  intra_pred_core(left, top, n, log2, c_idx, mode, pred, n);
}

/* ------------------------------------------------------------------------------------------------ inter, 8.5.3.3 */
static inline int ref_sample(const uint8_t* ref, int w, int h, int x, int y) { return ref[hv_clip3(0, h - 1, y) * w + hv_clip3(0, w - 1, x)]; }

/* one block of prediction samples at 14-bit precision (predSamplesLX, 8.5.3.3.3): rows of the reference are fetched
 * once, clamped to the picture (the padding of 8.5.3.3.3.1), filtered horizontally at full precision (shift1 = 0 at
 * 8 bits), then vertically */
void hv_interp_block(const uint8_t* ref, int rw, int rh, int chroma, int x, int y, int w, int h, int mvx, int mvy, int16_t* out) {
  const int fbits = chroma ? 3 : 2, taps = chroma ? 4 : 8, before = chroma ? 1 : 3;
  const int xf = mvx & ((1 << fbits) - 1), yf = mvy & ((1 << fbits) - 1);
  const int xi = x + (mvx >> fbits), yi = y + (mvy >> fbits);
  const int8_t* fx = chroma ? hevc_chroma_filter[xf] : hevc_luma_filter[xf];
  const int8_t* fy = chroma ? hevc_chroma_filter[yf] : hevc_luma_filter[yf];
  static int hor[(64 + 7) * 64];
  int row[64 + 8];
  const int rows = yf ? h + taps - 1 : h, y0 = yf ? yi - before : yi;
  for (int j = 0; j < rows; ++j) {
    const uint8_t* src = ref + (size_t)hv_clip3(0, rh - 1, y0 + j) * rw;
    if (xf) {
      for (int i = 0; i < w + taps - 1; ++i) row[i] = src[hv_clip3(0, rw - 1, xi - before + i)];
      for (int i = 0; i < w; ++i) {
        int v = 0;
        for (int t = 0; t < taps; ++t) v += fx[t] * row[i + t];
        hor[j * w + i] = v;
      }
    } else {
      for (int i = 0; i < w; ++i) hor[j * w + i] = src[hv_clip3(0, rw - 1, xi + i)] << (yf ? 0 : 6);
    }
  }
  for (int j = 0; j < h; ++j)
    for (int i = 0; i < w; ++i) {
      int v;
      if (!yf) {
        v = hor[j * w + i];
      } else {
        v = 0;
        for (int t = 0; t < taps; ++t) v += fy[t] * hor[(j + t) * w + i];
        if (xf) v >>= 6; /* shift2 */
      }
      out[j * w + i] = (int16_t)v;
    }
}

void hv_inter_predict_pu(HevcOracle* d, const HvPu* pu) {
  const HvSps* s = d->cur_sps;
  const HvSliceHeader* sh = &d->sh;
  const int weighted = sh->slice_type == HV_SLICE_P ? d->cur_pps->weighted_pred : d->cur_pps->weighted_bipred;
  static int16_t tmp[2][64 * 64];
  for (int c = 0; c < 3; ++c) {
    const int sub = c ? 1 : 0;
    const int x = pu->x >> sub, y = pu->y >> sub, w = pu->w >> sub, h = pu->h >> sub;
    const int pw = s->width >> sub, ph = s->height >> sub;
    for (int l = 0; l < 2; ++l)
      if (pu->mv.pred_flag & (1 << l))
        hv_interp_block(d->ref_list[l][pu->mv.ref_idx[l]]->plane[c], pw, ph, c != 0, x, y, w, h, pu->mv.mv[l][0], pu->mv.mv[l][1], tmp[l]);
    uint8_t* dst = d->cur->plane[c] + y * pw + x;
    const int bi = pu->mv.pred_flag == 3, l0 = (pu->mv.pred_flag & 1) ? 0 : 1;
    if (!weighted) {
      for (int j = 0; j < h; ++j)
        for (int i = 0; i < w; ++i)
          dst[j * pw + i] = (uint8_t)hv_clip8(bi ? (tmp[0][j * w + i] + tmp[1][j * w + i] + 64) >> 7 : (tmp[l0][j * w + i] + 32) >> 6);
      continue;
    }
    /* 8.5.3.3.4.3 */
    const int log2wd = (c ? sh->chroma_log2_wd : sh->luma_log2_wd) + 6;
    int wt[2] = {0, 0}, of[2] = {0, 0};
    for (int l = 0; l < 2; ++l)
      if (pu->mv.pred_flag & (1 << l)) {
        const int r = pu->mv.ref_idx[l];
        wt[l] = c ? sh->chroma_weight[l][r][c - 1] : sh->luma_weight[l][r];
        of[l] = c ? sh->chroma_offset[l][r][c - 1] : sh->luma_offset[l][r];
      }
    for (int j = 0; j < h; ++j)
      for (int i = 0; i < w; ++i) {
        int v;
        if (bi)
          v = (tmp[0][j * w + i] * wt[0] + tmp[1][j * w + i] * wt[1] + ((of[0] + of[1] + 1) << log2wd)) >> (log2wd + 1);
        else
          v = ((tmp[l0][j * w + i] * wt[l0] + (1 << (log2wd - 1))) >> log2wd) + of[l0]; /* (log2WD >= 6 at 8 bits) */
        dst[j * pw + i] = (uint8_t)hv_clip8(v);
      }
  }
}

void hevco_inter_block(const uint8_t* ref, int ref_w, int ref_h, int c_idx, int x, int y, int w, int h, int mvx, int mvy, uint8_t* pred) {
  int16_t* t = (int16_t*)malloc(sizeof(int16_t) * (size_t)w * h);
  hv_interp_block(ref, ref_w, ref_h, c_idx != 0, x, y, w, h, mvx, mvy, t);
  for (int i = 0; i < w * h; ++i) pred[i] = (uint8_t)hv_clip8((t[i] + 32) >> 6);
  free(t);
}

/* --------------------------------------------------------------------------------------- residual, 8.6.2 - 8.6.4 */
static void inverse_core(const int16_t* lev, int log2, int c_idx, int qp, int intra, int tskip, int bypass, const uint8_t* m, int16_t* res) {
  static const int level_scale[6] = {40, 45, 51, 57, 64, 72};
  const int n = 1 << log2;
  if (bypass) {
    for (int i = 0; i < n * n; ++i) res[i] = lev[i];
    return;
  }
  static int dq[32 * 32], g[32 * 32];
  const int bd_shift = log2 + 3; /* BitDepth + Log2(nTbS) - 5 */
  for (int i = 0; i < n * n; ++i) {
    const int64_t v = ((int64_t)lev[i] * (m ? m[i] : 16) * level_scale[qp % 6] << (qp / 6)) + ((int64_t)1 << (bd_shift - 1));
    const int64_t q = v >> bd_shift;
    dq[i] = (int)(q < -32768 ? -32768 : q > 32767 ? 32767 : q);
  }
  if (tskip) {
    for (int i = 0; i < n * n; ++i) res[i] = (int16_t)(((dq[i] << 7) + (1 << 11)) >> 12);
    return;
  }
  const int dst = intra && c_idx == 0 && log2 == 2;
  const int step = 32 >> log2;
  /* first stage: every column (vertical) */
  for (int x = 0; x < n; ++x)
    for (int i = 0; i < n; ++i) {
      int64_t e = 0;
      for (int j = 0; j < n; ++j) e += (int64_t)(dst ? hevc_dst4[j][i] : hevc_trans_matrix[j * step][i]) * dq[j * n + x];
      const int64_t t = (e + 64) >> 7;
      g[i * n + x] = (int)(t < -32768 ? -32768 : t > 32767 ? 32767 : t);
    }
  /* second stage: every row (horizontal), then the final shift of 20 - BitDepth */
  for (int y = 0; y < n; ++y)
    for (int i = 0; i < n; ++i) {
      int64_t e = 0;
      for (int j = 0; j < n; ++j) e += (int64_t)(dst ? hevc_dst4[j][i] : hevc_trans_matrix[j * step][i]) * g[y * n + j];
      res[y * n + i] = (int16_t)((e + (1 << 11)) >> 12);
    }
}

void hv_inverse_transform(const HevcOracle* d, const int16_t* lev, int log2, int c_idx, int qp, int intra, int tskip, int bypass, int16_t* res) {
  const uint8_t* m = NULL;
  if (d->cur_sps->scaling_list_enabled && !(tskip && log2 > 2)) {
    const HvScaling* sc = d->cur_pps->scaling_list_present ? &d->cur_pps->scaling : &d->cur_sps->scaling;
    const int mid = (intra ? 0 : 3) + c_idx;
    m = log2 == 2 ? sc->f4[mid] : log2 == 3 ? sc->f8[mid] : log2 == 4 ? sc->f16[mid] : sc->f32[intra ? 0 : 1];
  }
  inverse_core(lev, log2, c_idx, qp, intra, tskip, bypass, m, res);
}

void hevco_residual_block(const int16_t* levels, int log2, int c_idx, int qp, int intra, int tskip, int16_t* res) {
  inverse_core(levels, log2, c_idx, qp, intra, tskip, 0, NULL, res);
}

void hv_residual_add(HevcOracle* d, const HvCu* cu, const HvTu* tu, int c) {
  if (!tu->cbf[c] || tu->coef[c] < 0) return;
  const HvSps* s = d->cur_sps;
  const int log2 = c ? tu->clog2 : tu->log2, n = 1 << log2;
  const int x = c ? tu->cx : tu->x, y = c ? tu->cy : tu->y;
  const int stride = s->width >> (c ? 1 : 0);
  const HvSliceFacts* f = &d->slice_facts[d->slice_idx];
  int qp = cu->qp_y;
  if (c) qp = hv_chroma_qp(cu->qp_y, (c == 1 ? d->cur_pps->cb_qp_offset + f->cb_qp_offset : d->cur_pps->cr_qp_offset + f->cr_qp_offset));
  static int16_t res[32 * 32];
  hv_inverse_transform(d, &d->coefs[tu->coef[c]], log2, c, qp, cu->intra, tu->tskip[c], cu->tq_bypass, res);
  uint8_t* p = d->cur->plane[c] + y * stride + x;
  for (int j = 0; j < n; ++j)
    for (int i = 0; i < n; ++i) p[j * stride + i] = (uint8_t)hv_clip8(p[j * stride + i] + res[j * n + i]);
}

/* transform / prediction block edges and the "has levels" map the deblocking filter reads (8.7.2.2, 8.7.2.3) */
void hv_mark_edges(HevcOracle* d, const HvCu* cu) {
  const int n = 1 << cu->log2;
  const HvSliceFacts* f = &d->slice_facts[d->slice_idx];
  const int w4 = d->w4;
  int filter_left = cu->x > 0, filter_top = cu->y > 0;
  if (filter_left && !f->loop_filter_across_slices && !hv_zscan_available(d, cu->x, cu->y, cu->x - 1, cu->y)) filter_left = 0;
  if (filter_top && !f->loop_filter_across_slices && !hv_zscan_available(d, cu->x, cu->y, cu->x, cu->y - 1)) filter_top = 0;
  const int on = !f->deblocking_disabled;
  for (int k = 0; k < n; k += 4) {
    d->m_edge[((cu->y + k) >> 2) * w4 + (cu->x >> 2)] |= (uint8_t)(1 | 2 | (on && filter_left ? 16 : 0));
    d->m_edge[(cu->y >> 2) * w4 + ((cu->x + k) >> 2)] |= (uint8_t)(4 | 8 | (on && filter_top ? 32 : 0));
  }
  for (int t = 0; t < cu->num_tu; ++t) {
    const HvTu* tu = &d->tus[cu->first_tu + t];
    const int tn = 1 << tu->log2;
    for (int k = 0; k < tn; k += 4) {
      if (tu->x > cu->x) d->m_edge[((tu->y + k) >> 2) * w4 + (tu->x >> 2)] |= (uint8_t)(1 | (on ? 16 : 0));
      if (tu->y > cu->y) d->m_edge[(tu->y >> 2) * w4 + ((tu->x + k) >> 2)] |= (uint8_t)(4 | (on ? 32 : 0));
    }
    if (tu->cbf[0])
      for (int j = 0; j < tn; j += 4)
        for (int i = 0; i < tn; i += 4) d->m_nz[((tu->y + j) >> 2) * w4 + ((tu->x + i) >> 2)] = 1;
  }
  if (!cu->intra)
    for (int p = 1; p < cu->num_pu; ++p) {
      const HvPu* pu = &cu->pu[p];
      if (pu->x > cu->x)
        for (int k = 0; k < pu->h; k += 4) d->m_edge[((pu->y + k) >> 2) * w4 + (pu->x >> 2)] |= (uint8_t)(2 | (on ? 16 : 0));
      if (pu->y > cu->y)
        for (int k = 0; k < pu->w; k += 4) d->m_edge[(pu->y >> 2) * w4 + ((pu->x + k) >> 2)] |= (uint8_t)(8 | (on ? 32 : 0));
    }
}

void hv_recon_cu(HevcOracle* d, const HvCu* cu) {
  const HvSps* s = d->cur_sps;
  const int n = 1 << cu->log2;
  hv_mark_edges(d, cu);
  if (cu->pcm) {
    const int16_t* smp = &d->coefs[cu->pcm_samples];
    for (int c = 0; c < 3; ++c) {
      const int sub = c ? 1 : 0, m = n >> sub, stride = s->width >> sub, shift = 8 - (c ? s->pcm_bits_chroma : s->pcm_bits_luma);
      uint8_t* p = d->cur->plane[c] + (cu->y >> sub) * stride + (cu->x >> sub);
      for (int j = 0; j < m; ++j)
        for (int i = 0; i < m; ++i) p[j * stride + i] = (uint8_t)(*smp++ << shift);
    }
    return;
  }
  if (cu->intra) {
    const int half = n >> 1;
    for (int t = 0; t < cu->num_tu; ++t) {
      const HvTu* tu = &d->tus[cu->first_tu + t];
      const int pi = cu->part_mode == HV_PART_NxN ? ((tu->y - cu->y >= half) << 1 | (tu->x - cu->x >= half)) : 0;
      hv_intra_predict(d, 0, tu->x, tu->y, tu->log2, cu->luma_mode[pi]);
      hv_residual_add(d, cu, tu, 0);
      if (tu->has_chroma)
        for (int c = 1; c < 3; ++c) {
          hv_intra_predict(d, c, tu->cx, tu->cy, tu->clog2, cu->chroma_mode);
          hv_residual_add(d, cu, tu, c);
        }
    }
    return;
  }
  for (int p = 0; p < cu->num_pu; ++p) hv_inter_predict_pu(d, &cu->pu[p]);
  for (int t = 0; t < cu->num_tu; ++t)
    for (int c = 0; c < 3; ++c) hv_residual_add(d, cu, &d->tus[cu->first_tu + t], c);
}

/* --------------------------------------------------------------------------------------------- deblocking, 8.7.2 */
static int motion_bs(const HvMv* p, const HvMv* q) {
  const int np = (p->pred_flag & 1) + (p->pred_flag >> 1), nq = (q->pred_flag & 1) + (q->pred_flag >> 1);
  if (np != nq) return 1;
#define FAR(a, b) (hv_abs((a)[0] - (b)[0]) >= 4 || hv_abs((a)[1] - (b)[1]) >= 4)
  if (np == 1) {
    const int lp = (p->pred_flag & 1) ? 0 : 1, lq = (q->pred_flag & 1) ? 0 : 1;
    if (p->ref_poc[lp] != q->ref_poc[lq]) return 1;
    return FAR(p->mv[lp], q->mv[lq]);
  }
  if (np == 2) {
    const int pa = p->ref_poc[0], pb = p->ref_poc[1], qa = q->ref_poc[0], qb = q->ref_poc[1];
    if (!((pa == qa && pb == qb) || (pa == qb && pb == qa))) return 1;
    if (pa != pb) {
      if (pa == qa) return FAR(p->mv[0], q->mv[0]) || FAR(p->mv[1], q->mv[1]);
      return FAR(p->mv[0], q->mv[1]) || FAR(p->mv[1], q->mv[0]);
    }
    return (FAR(p->mv[0], q->mv[0]) || FAR(p->mv[1], q->mv[1])) && (FAR(p->mv[0], q->mv[1]) || FAR(p->mv[1], q->mv[0]));
  }
#undef FAR
  return 0;
}

static void derive_bs(HevcOracle* d) {
  const int w4 = d->w4, h4 = d->h4;
  for (int dir = 0; dir < 2; ++dir)
    for (int y = 0; y < h4; ++y)
      for (int x = 0; x < w4; ++x) {
        const int i = y * w4 + x;
        d->bs[dir][i] = 0;
        if (((dir ? y : x) & 1) || !(d->m_edge[i] & (dir ? 32 : 16))) continue; /* the 8x8 grid; filterEdgeFlag */
        const int j = dir ? i - w4 : i - 1;
        if (d->m_pred[i] == 2 || d->m_pred[j] == 2)
          d->bs[dir][i] = 2;
        else if ((d->m_edge[i] & (dir ? 4 : 1)) && (d->m_nz[i] || d->m_nz[j]))
          d->bs[dir][i] = 1;
        else
          d->bs[dir][i] = (uint8_t)motion_bs(&d->cur->mvf[j], &d->cur->mvf[i]);
      }
}

/* one 4-sample segment of a luma edge; p points at q0 of the first line, `across` steps over the edge, `along` along it */
static void filter_luma_segment(uint8_t* q0, int across, int along, int beta, int tc, int no_p, int no_q) {
#define P(k, l) q0[-((k) + 1) * across + (l) * along]
#define Q(k, l) q0[(k) * across + (l) * along]
  const int dp0 = hv_abs(P(2, 0) - 2 * P(1, 0) + P(0, 0)), dp3 = hv_abs(P(2, 3) - 2 * P(1, 3) + P(0, 3));
  const int dq0 = hv_abs(Q(2, 0) - 2 * Q(1, 0) + Q(0, 0)), dq3 = hv_abs(Q(2, 3) - 2 * Q(1, 3) + Q(0, 3));
  const int dpq0 = dp0 + dq0, dpq3 = dp3 + dq3, dp = dp0 + dp3, dq = dq0 + dq3;
  if (dpq0 + dpq3 >= beta) return;
  const int strong0 = 2 * dpq0 < (beta >> 2) && hv_abs(P(3, 0) - P(0, 0)) + hv_abs(Q(0, 0) - Q(3, 0)) < (beta >> 3) && hv_abs(P(0, 0) - Q(0, 0)) < ((5 * tc + 1) >> 1);
  const int strong3 = 2 * dpq3 < (beta >> 2) && hv_abs(P(3, 3) - P(0, 3)) + hv_abs(Q(0, 3) - Q(3, 3)) < (beta >> 3) && hv_abs(P(0, 3) - Q(0, 3)) < ((5 * tc + 1) >> 1);
  const int strong = strong0 && strong3;
  const int dep = dp < ((beta + (beta >> 1)) >> 3), deq = dq < ((beta + (beta >> 1)) >> 3);
  for (int l = 0; l < 4; ++l) {
    const int p0 = P(0, l), p1 = P(1, l), p2 = P(2, l), p3 = P(3, l), q0v = Q(0, l), q1 = Q(1, l), q2 = Q(2, l), q3 = Q(3, l);
    if (strong) {
      if (!no_p) {
        P(0, l) = (uint8_t)hv_clip3(p0 - 2 * tc, p0 + 2 * tc, (p2 + 2 * p1 + 2 * p0 + 2 * q0v + q1 + 4) >> 3);
        P(1, l) = (uint8_t)hv_clip3(p1 - 2 * tc, p1 + 2 * tc, (p2 + p1 + p0 + q0v + 2) >> 2);
        P(2, l) = (uint8_t)hv_clip3(p2 - 2 * tc, p2 + 2 * tc, (2 * p3 + 3 * p2 + p1 + p0 + q0v + 4) >> 3);
      }
      if (!no_q) {
        Q(0, l) = (uint8_t)hv_clip3(q0v - 2 * tc, q0v + 2 * tc, (p1 + 2 * p0 + 2 * q0v + 2 * q1 + q2 + 4) >> 3);
        Q(1, l) = (uint8_t)hv_clip3(q1 - 2 * tc, q1 + 2 * tc, (p0 + q0v + q1 + q2 + 2) >> 2);
        Q(2, l) = (uint8_t)hv_clip3(q2 - 2 * tc, q2 + 2 * tc, (p0 + q0v + q1 + 3 * q2 + 2 * q3 + 4) >> 3);
      }
    } else {
      int delta = (9 * (q0v - p0) - 3 * (q1 - p1) + 8) >> 4;
      if (hv_abs(delta) >= tc * 10) continue;
      delta = hv_clip3(-tc, tc, delta);
      if (!no_p) {
        P(0, l) = (uint8_t)hv_clip8(p0 + delta);
        if (dep) P(1, l) = (uint8_t)hv_clip8(p1 + hv_clip3(-(tc >> 1), tc >> 1, (((p2 + p0 + 1) >> 1) - p1 + delta) >> 1));
      }
      if (!no_q) {
        Q(0, l) = (uint8_t)hv_clip8(q0v - delta);
        if (deq) Q(1, l) = (uint8_t)hv_clip8(q1 + hv_clip3(-(tc >> 1), tc >> 1, (((q2 + q0v + 1) >> 1) - q1 - delta) >> 1));
      }
    }
  }
#undef P
#undef Q
}

void hv_deblock_picture(HevcOracle* d) {
  const HvSps* s = d->cur_sps;
  const HvPps* pps = d->cur_pps;
  const int w4 = d->w4, h4 = d->h4;
  derive_bs(d);
  for (int dir = 0; dir < 2; ++dir) {
    for (int y = 0; y < h4; ++y)
      for (int x = 0; x < w4; ++x) {
        const int i = y * w4 + x, bs = d->bs[dir][i];
        if (!bs) continue;
        const int j = dir ? i - w4 : i - 1;
        const HvSliceFacts* f = &d->slice_facts[d->m_slice[((y * 4) >> s->log2_ctb) * s->width_ctb + ((x * 4) >> s->log2_ctb)]];
        const int qpl = (d->m_qp[i] + d->m_qp[j] + 1) >> 1;
        const int beta = hevc_beta_table[hv_clip3(0, 51, qpl + (f->beta_offset_div2 << 1))];
        const int tc = hevc_tc_table[hv_clip3(0, 53, qpl + 2 * (bs - 1) + (f->tc_offset_div2 << 1))];
        const int no_p = (d->m_flags[j] & 4) != 0, no_q = (d->m_flags[i] & 4) != 0;
        uint8_t* q0 = d->cur->plane[0] + (y * 4) * s->width + x * 4;
        filter_luma_segment(q0, dir ? s->width : 1, dir ? 1 : s->width, beta, tc, no_p, no_q);
      }
    /* chroma: edges of strength 2 on the 8-sample chroma grid; a 4-sample luma segment is two chroma samples long */
    const int cw = s->width / 2;
    for (int y = 0; y < h4; ++y)
      for (int x = 0; x < w4; ++x) {
        const int i = y * w4 + x;
        if (d->bs[dir][i] != 2 || ((dir ? y : x) & 3)) continue;
        const int j = dir ? i - w4 : i - 1;
        const HvSliceFacts* f = &d->slice_facts[d->m_slice[((y * 4) >> s->log2_ctb) * s->width_ctb + ((x * 4) >> s->log2_ctb)]];
        const int no_p = (d->m_flags[j] & 4) != 0, no_q = (d->m_flags[i] & 4) != 0;
        for (int c = 1; c < 3; ++c) {
          const int qpi = ((d->m_qp[i] + d->m_qp[j] + 1) >> 1) + (c == 1 ? pps->cb_qp_offset : pps->cr_qp_offset);
          const int qpc = qpi < 30 ? qpi : qpi >= 43 ? qpi - 6 : hevc_qpc_table[qpi - 30];
          const int tc = hevc_tc_table[hv_clip3(0, 53, qpc + 2 + (f->tc_offset_div2 << 1))];
          uint8_t* q0 = d->cur->plane[c] + (y * 2) * cw + x * 2;
          const int across = dir ? cw : 1, along = dir ? 1 : cw;
          for (int l = 0; l < 2; ++l) {
            uint8_t* q = q0 + l * along;
            const int p0 = q[-across], p1 = q[-2 * across], q0v = q[0], q1 = q[across];
            const int delta = hv_clip3(-tc, tc, ((((q0v - p0) << 2) + p1 - q1 + 4) >> 3));
            if (!no_p) q[-across] = (uint8_t)hv_clip8(p0 + delta);
            if (!no_q) q[0] = (uint8_t)hv_clip8(q0v - delta);
          }
        }
      }
  }
}

/* ---------------------------------------------------------------------------------------------------- SAO, 8.7.3 */
void hv_sao_picture(HevcOracle* d) {
  const HvSps* s = d->cur_sps;
  if (!s->sao_enabled) return;
  memcpy(d->deblocked[0], d->cur->plane[0], (size_t)s->width * s->height * 3 / 2);
  static const int8_t hpos[4][2] = {{-1, 1}, {0, 0}, {-1, 1}, {1, -1}}, vpos[4][2] = {{0, 0}, {-1, 1}, {-1, 1}, {-1, 1}};
  for (int ry = 0; ry < s->height_ctb; ++ry)
    for (int rx = 0; rx < s->width_ctb; ++rx) {
      const int addr = ry * s->width_ctb + rx;
      const HvSao* p = &d->sao[addr];
      if (d->m_slice[addr] == 0xffff) continue;
      const HvSliceFacts* f = &d->slice_facts[d->m_slice[addr]];
      for (int c = 0; c < 3; ++c) {
        if (!p->type[c] || !(c ? f->sao_chroma : f->sao_luma)) continue;
        const int sub = c ? 1 : 0, cs = s->ctb_size >> sub, pw = s->width >> sub, ph = s->height >> sub;
        const uint8_t* src = d->deblocked[c];
        uint8_t* dst = d->cur->plane[c];
        int band[32];
        memset(band, 0, sizeof(band));
        for (int k = 0; k < 4; ++k) band[(k + p->band_pos[c]) & 31] = k + 1;
        for (int y = ry * cs; y < hv_min((ry + 1) * cs, ph); ++y)
          for (int x = rx * cs; x < hv_min((rx + 1) * cs, pw); ++x) {
            const int xl = x << sub, yl = y << sub;
            if (d->m_flags[(yl >> 2) * d->w4 + (xl >> 2)] & 4) continue; /* pcm with the loop filter off, or bypass */
            const int v = src[y * pw + x];
            int idx;
            if (p->type[c] == 1) {
              idx = band[v >> 3];
            } else {
              const int cl = p->eo_class[c];
              int sum = 2, ok = 1;
              for (int k = 0; k < 2 && ok; ++k) {
                const int xn = x + hpos[cl][k], yn = y + vpos[cl][k];
                if (xn < 0 || yn < 0 || xn >= pw || yn >= ph) { ok = 0; break; }
                const int an = ((yn << sub) >> s->log2_ctb) * s->width_ctb + ((xn << sub) >> s->log2_ctb);
                if (an != addr && d->m_slice[an] != d->m_slice[addr]) {
                  /* across a slice boundary: the later of the two slices decides */
                  const HvSliceFacts* fn = d->m_slice[an] == 0xffff ? NULL : &d->slice_facts[d->m_slice[an]];
                  if (!fn || fn->slice_addr_rs != f->slice_addr_rs) {
                    const int later_is_cur = an < addr;
                    if (later_is_cur ? !f->loop_filter_across_slices : !(fn && fn->loop_filter_across_slices)) { ok = 0; break; }
                  }
                }
                sum += hv_sign(v - src[yn * pw + xn]);
              }
              if (!ok) continue;
              idx = sum == 2 ? 0 : sum < 2 ? sum + 1 : sum;
            }
            if (idx) dst[y * pw + x] = (uint8_t)hv_clip8(v + p->offset[c][idx - 1]);
          }
      }
    }
}

/* ---------------------------------------------------------------------- hooks for tests/test_hevc_independent.py */
/* one 4-line luma edge segment: blk[line * 8 + k], k = 0..7 across the edge (p3 p2 p1 p0 | q0 q1 q2 q3) */
void hevco_deblock_luma(uint8_t* blk, int bs, int qp_p, int qp_q, int beta_offset_div2, int tc_offset_div2, int no_p, int no_q) {
  const int qpl = (qp_p + qp_q + 1) >> 1;
  const int beta = hevc_beta_table[hv_clip3(0, 51, qpl + (beta_offset_div2 << 1))];
  const int tc = hevc_tc_table[hv_clip3(0, 53, qpl + 2 * (bs - 1) + (tc_offset_div2 << 1))];
  filter_luma_segment(blk + 4, 1, 8, beta, tc, no_p, no_q);
}
/* one chroma edge sample row: p1 p0 | q0 q1, bS = 2 */
void hevco_deblock_chroma(uint8_t* row, int qp_p, int qp_q, int c_qp_pic_offset, int tc_offset_div2) {
  const int qpi = ((qp_p + qp_q + 1) >> 1) + c_qp_pic_offset;
  const int qpc = qpi < 30 ? qpi : qpi >= 43 ? qpi - 6 : hevc_qpc_table[qpi - 30];
  const int tc = hevc_tc_table[hv_clip3(0, 53, qpc + 2 + (tc_offset_div2 << 1))];
  const int p1 = row[0], p0 = row[1], q0 = row[2], q1 = row[3];
  const int delta = hv_clip3(-tc, tc, ((((q0 - p0) << 2) + p1 - q1 + 4) >> 3));
  row[1] = (uint8_t)hv_clip8(p0 + delta);
  row[2] = (uint8_t)hv_clip8(q0 - delta);
}
/* SAO of a w x h block that is a whole picture (samples whose neighbour lies outside stay as they are) */
void hevco_sao_block(const uint8_t* in, int w, int h, int type, int band_pos, int eo_class, const int8_t* offsets, uint8_t* out) {
  static const int8_t hpos[4][2] = {{-1, 1}, {0, 0}, {-1, 1}, {1, -1}}, vpos[4][2] = {{0, 0}, {-1, 1}, {-1, 1}, {-1, 1}};
  int band[32];
  memset(band, 0, sizeof(band));
  for (int k = 0; k < 4; ++k) band[(k + band_pos) & 31] = k + 1;
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x) {
      const int v = in[y * w + x];
      int idx = 0;
      if (type == 1) {
        idx = band[v >> 3];
      } else if (type == 2) {
        const int xa = x + hpos[eo_class][0], ya = y + vpos[eo_class][0], xb = x + hpos[eo_class][1], yb = y + vpos[eo_class][1];
        if (xa >= 0 && ya >= 0 && xb >= 0 && yb >= 0 && xa < w && xb < w && ya < h && yb < h) {
          const int sum = 2 + hv_sign(v - in[ya * w + xa]) + hv_sign(v - in[yb * w + xb]);
          idx = sum == 2 ? 0 : sum < 2 ? sum + 1 : sum;
        }
      }
      out[y * w + x] = (uint8_t)(idx ? hv_clip8(v + offsets[idx - 1]) : v);
    }
}
/* the constant tables, by name (tests/test_hevc_tables.py) */
const void* hevco_table(const char* name, int* bytes) {
#define T(n, t) if (!strcmp(name, n)) { *bytes = (int)sizeof(t); return t; }
  T("cabac_init", hevc_cabac_init) T("range_lps", hevc_range_lps) T("trans_lps", hevc_trans_lps) T("trans_matrix", hevc_trans_matrix)
  T("dst4", hevc_dst4) T("luma_filter", hevc_luma_filter) T("chroma_filter", hevc_chroma_filter) T("intra_angle", hevc_intra_angle)
  T("inv_angle", hevc_inv_angle) T("beta", hevc_beta_table) T("tc", hevc_tc_table) T("qpc", hevc_qpc_table)
  T("scaling_intra8", hevc_default_scaling_intra8) T("scaling_inter8", hevc_default_scaling_inter8) T("ctx_idx_map4", hevc_ctx_idx_map4)
#undef T
  if (!strncmp(name, "scan", 4) && strlen(name) == 6) { /* "scanLS": log2 size L, scan index S */
    *bytes = 1 << (2 * (name[4] - '0'));
    return hv_scan(name[4] - '0', name[5] - '0');
  }
  *bytes = 0;
  return NULL;
}
