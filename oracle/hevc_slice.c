/* TEST INFRASTRUCTURE - HEVC oracle: the CABAC engine of 9.3 (decoder and encoder), the slice segment data syntax of
 * 7.3.8 written ONCE and walked in either direction (d->enc: the corpus encoder writes streams through the very
 * routines the oracle parses them with, as corpus/h264_encoder.c does with the H.264 oracle's syntax engine), and the
 * derivations that sit between syntax and samples: most probable intra modes (8.4.2), merge and AMVP candidates with
 * the temporal one (8.5.3.2.1 - 8.5.3.2.9), the luma QP predictor (8.6.1).  See hevc_oracle.h: parity unpinned. */
#include <stdio.h>

#include "hevc_internal.h"

/* ------------------------------------------------------------------------------------------- CABAC, 9.3.2 / 9.3.4 */
void hv_cabac_init_contexts(HevcOracle* d) {
  const HvSliceHeader* sh = &d->sh;
  int init_type = 0;
  if (sh->slice_type == HV_SLICE_P) init_type = sh->cabac_init_flag ? 2 : 1;
  if (sh->slice_type == HV_SLICE_B) init_type = sh->cabac_init_flag ? 1 : 2;
  const int qp = hv_clip3(0, 51, sh->slice_qp);
  for (int i = 0; i < HC_CTX_COUNT; ++i) {
    const int v = hevc_cabac_init[init_type][i];
    const int m = (v >> 4) * 5 - 45, n = ((v & 15) << 3) - 16;
    const int pre = hv_clip3(1, 126, ((m * qp) >> 4) + n);
    const int mps = pre > 63;
    d->ctx[i] = (uint8_t)((mps ? pre - 64 : 63 - pre) << 1 | mps);
  }
}

static inline unsigned cd_bit(HvCabacDec* c) {
  unsigned bit = 0;
  if ((c->pos >> 3) < c->size)
    bit = (c->p[c->pos >> 3] >> (7 - (c->pos & 7))) & 1;
  else
    c->overrun = 1;
  ++c->pos;
  return bit;
}
static void cd_init_engine(HvCabacDec* c) {
  c->range = 510;
  c->offset = 0;
  for (int i = 0; i < 9; ++i) c->offset = c->offset << 1 | cd_bit(c);
}
void hv_cabac_dec_start(HvCabacDec* c, const uint8_t* p, size_t size, size_t byte_pos) {
  c->p = p;
  c->size = size;
  c->pos = byte_pos * 8;
  c->overrun = 0;
  cd_init_engine(c);
}
static inline int cd_decision(HvCabacDec* c, uint8_t* ctx) {
  int state = *ctx >> 1, mps = *ctx & 1, bin;
  const unsigned lps = hevc_range_lps[state][(c->range >> 6) & 3];
  c->range -= lps;
  if (c->offset >= c->range) {
    bin = !mps;
    c->offset -= c->range;
    c->range = lps;
    if (state == 0) mps = !mps;
    state = hevc_trans_lps[state];
  } else {
    bin = mps;
    if (state < 62) ++state;
  }
  *ctx = (uint8_t)(state << 1 | mps);
  while (c->range < 256) {
    c->range <<= 1;
    c->offset = c->offset << 1 | cd_bit(c);
  }
  return bin;
}
static inline int cd_bypass(HvCabacDec* c) {
  c->offset = c->offset << 1 | cd_bit(c);
  if (c->offset >= c->range) {
    c->offset -= c->range;
    return 1;
  }
  return 0;
}
static inline int cd_terminate(HvCabacDec* c) {
  c->range -= 2;
  /* The encoder's flush (9.3.4.5) ends on a one bit, and a decoder that started with nine bits has just read it: the
   * last bit consumed is that one, what follows is zeros up to the byte boundary (rbsp_slice_segment_trailing_bits
   * seen from behind its stop bit; the alignment in front of pcm_sample() and of a wavefront substream). */
  if (c->offset >= c->range) return 1;
  while (c->range < 256) {
    c->range <<= 1;
    c->offset = c->offset << 1 | cd_bit(c);
  }
  return 0;
}

static void ce_write(HvCabacEnc* e, int bit) {
  if ((e->nbits >> 3) >= e->cap) {
    e->cap = e->cap * 2 + 4096;
    e->buf = (uint8_t*)realloc(e->buf, e->cap);
    memset(e->buf + (e->nbits >> 3), 0, e->cap - (e->nbits >> 3));
  }
  if (bit) e->buf[e->nbits >> 3] |= (uint8_t)(0x80 >> (e->nbits & 7));
  ++e->nbits;
}
static void ce_put(HvCabacEnc* e, int bit) {
  if (e->first)
    e->first = 0;
  else
    ce_write(e, bit);
  for (; e->outstanding > 0; --e->outstanding) ce_write(e, !bit);
}
static void ce_renorm(HvCabacEnc* e) {
  while (e->range < 256) {
    if (e->low < 256) {
      ce_put(e, 0);
    } else if (e->low >= 512) {
      e->low -= 512;
      ce_put(e, 1);
    } else {
      e->low -= 256;
      ++e->outstanding;
    }
    e->range <<= 1;
    e->low <<= 1;
  }
}
static void ce_init_engine(HvCabacEnc* e) {
  e->low = 0;
  e->range = 510;
  e->first = 1;
  e->outstanding = 0;
}
void hv_cabac_enc_start(HvCabacEnc* e) {
  if (e->buf) memset(e->buf, 0, e->cap);
  e->nbits = 0;
  ce_init_engine(e);
}
static void ce_decision(HvCabacEnc* e, uint8_t* ctx, int bin) {
  int state = *ctx >> 1, mps = *ctx & 1;
  const unsigned lps = hevc_range_lps[state][(e->range >> 6) & 3];
  e->range -= lps;
  if (bin != mps) {
    e->low += e->range;
    e->range = lps;
    if (state == 0) mps = !mps;
    state = hevc_trans_lps[state];
  } else if (state < 62) {
    ++state;
  }
  *ctx = (uint8_t)(state << 1 | mps);
  ce_renorm(e);
}
static void ce_bypass(HvCabacEnc* e, int bin) {
  e->low <<= 1;
  if (bin) e->low += e->range;
  if (e->low >= 1024) {
    ce_put(e, 1);
    e->low -= 1024;
  } else if (e->low < 512) {
    ce_put(e, 0);
  } else {
    e->low -= 512;
    ++e->outstanding;
  }
}
/* 9.3.4.5 / EncodeFlush: ends on a one bit - the rbsp_stop_one_bit of a slice, or the bit in front of the alignment
 * zeros of pcm_sample() and of a wavefront substream */
static void ce_terminate(HvCabacEnc* e, int bin) {
  e->range -= 2;
  if (!bin) {
    ce_renorm(e);
    return;
  }
  e->low += e->range;
  e->range = 2;
  ce_renorm(e);
  ce_put(e, (e->low >> 9) & 1);
  ce_write(e, (e->low >> 8) & 1);
  ce_write(e, 1);
}
void hv_cabac_enc_finish_slice(HvCabacEnc* e) {
  while (e->nbits & 7) ce_write(e, 0);
}

/* the two directions behind one call */
static inline int cbin(HevcOracle* d, int ctx, int v) {
  if (d->enc) {
    ce_decision(&d->ce, &d->ctx[ctx], v);
    return v;
  }
  return cd_decision(&d->cd, &d->ctx[ctx]);
}
static inline int cbyp(HevcOracle* d, int v) {
  if (d->enc) {
    ce_bypass(&d->ce, v);
    return v;
  }
  return cd_bypass(&d->cd);
}
static inline unsigned cbyp_bits(HevcOracle* d, unsigned v, int n) {
  unsigned r = 0;
  for (int i = n - 1; i >= 0; --i) r = r << 1 | (unsigned)cbyp(d, (int)((v >> i) & 1));
  return r;
}
static inline int cterm(HevcOracle* d, int v) {
  if (d->enc) {
    ce_terminate(&d->ce, v);
    return v;
  }
  return cd_terminate(&d->cd);
}
/* truncated unary with every bin in bypass */
static int cbyp_tu(HevcOracle* d, int v, int cmax) {
  int n = 0;
  while (n < cmax && cbyp(d, n < v)) ++n;
  return n;
}
/* k-th order Exp-Golomb in bypass (9.3.3.6) */
static unsigned cbyp_egk(HevcOracle* d, unsigned v, int k) {
  if (d->enc) {
    unsigned s = v;
    for (;;) {
      if (s >= (1u << k)) {
        ce_bypass(&d->ce, 1);
        s -= 1u << k;
        ++k;
      } else {
        ce_bypass(&d->ce, 0);
        while (k--) ce_bypass(&d->ce, (int)((s >> k) & 1));
        break;
      }
    }
    return v;
  }
  unsigned r = 0;
  while (cd_bypass(&d->cd) && k < 31) {
    r += 1u << k;
    ++k;
  }
  while (k--) r += (unsigned)cd_bypass(&d->cd) << k;
  return r;
}

/* --------------------------------------------------------------------------------------------------------- scans */
static uint8_t scan_tab[4][3][64]; /* [log2 of the square][scanIdx: 0 diagonal, 1 horizontal, 2 vertical]: x | y << 4 */
static int scan_ready;
static void build_scans(void) {
  for (int l = 0; l < 4; ++l) {
    const int n = 1 << l;
    int i = 0, x = 0, y = 0;
    for (;;) { /* 6.5.3 */
      while (y >= 0) {
        if (x < n && y < n) scan_tab[l][0][i++] = (uint8_t)(x | y << 4);
        --y;
        ++x;
      }
      y = x;
      x = 0;
      if (i >= n * n) break;
    }
    i = 0;
    for (y = 0; y < n; ++y)
      for (x = 0; x < n; ++x) scan_tab[l][1][i++] = (uint8_t)(x | y << 4);
    i = 0;
    for (x = 0; x < n; ++x)
      for (y = 0; y < n; ++y) scan_tab[l][2][i++] = (uint8_t)(x | y << 4);
  }
  scan_ready = 1;
}
const uint8_t* hv_scan(int log2_blocks, int scan_idx) {
  if (!scan_ready) build_scans();
  return scan_tab[log2_blocks][scan_idx];
}
const uint8_t* hv_diag_scan(int log2_blocks) { return hv_scan(log2_blocks, 0); }

/* --------------------------------------------------------------------------------------------------------- pools */
HvCu* hv_new_cu(HevcOracle* d) {
  if (d->num_cu >= d->cap_cu) return NULL;
  HvCu* cu = &d->cus[d->num_cu++];
  memset(cu, 0, sizeof(*cu));
  cu->first_node = d->num_nodes;
  cu->first_tu = d->num_tu;
  cu->pcm_samples = -1;
  return cu;
}
HvTu* hv_new_tu(HevcOracle* d) {
  if (d->num_tu >= d->cap_tu) return NULL;
  HvTu* tu = &d->tus[d->num_tu++];
  memset(tu, 0, sizeof(*tu));
  tu->coef[0] = tu->coef[1] = tu->coef[2] = -1;
  return tu;
}
int hv_new_nodes(HevcOracle* d, int n) {
  if (d->num_nodes + n > d->cap_nodes) return -1;
  const int at = d->num_nodes;
  memset(&d->nodes[at], 0, sizeof(HvNode) * (size_t)n);
  d->num_nodes += n;
  return at;
}
int32_t hv_new_coefs(HevcOracle* d, int n) {
  if (d->num_coefs + n > d->cap_coefs) return -1;
  const int32_t at = d->num_coefs;
  memset(&d->coefs[at], 0, sizeof(int16_t) * (size_t)n);
  d->num_coefs += n;
  return at;
}

/* ------------------------------------------------------------------------------------------------- availability */
static inline int zaddr(const HevcOracle* d, int x, int y) {
  const HvSps* s = d->cur_sps;
  const int ctb = (y >> s->log2_ctb) * s->width_ctb + (x >> s->log2_ctb);
  const int bits = s->log2_ctb - 2;
  int xi = (x >> 2) & ((1 << bits) - 1), yi = (y >> 2) & ((1 << bits) - 1), m = 0;
  for (int i = 0; i < bits; ++i) m |= ((xi >> i) & 1) << (2 * i) | ((yi >> i) & 1) << (2 * i + 1);
  return (ctb << (2 * bits)) | m;
}
/* 6.4.1 */
int hv_zscan_available(const HevcOracle* d, int xc, int yc, int xn, int yn) {
  const HvSps* s = d->cur_sps;
  if (xn < 0 || yn < 0 || xn >= s->width || yn >= s->height) return 0;
  if (zaddr(d, xn, yn) > zaddr(d, xc, yc)) return 0;
  const int cn = (yn >> s->log2_ctb) * s->width_ctb + (xn >> s->log2_ctb);
  const int cc = (yc >> s->log2_ctb) * s->width_ctb + (xc >> s->log2_ctb);
  if (d->m_slice[cn] == 0xffff || d->m_slice[cc] == 0xffff) return 0;
  return d->slice_facts[d->m_slice[cn]].slice_addr_rs == d->slice_facts[d->m_slice[cc]].slice_addr_rs;
}
/* 6.4.2 */
static int pb_available(const HevcOracle* d, const HvCu* cu, int xp, int yp, int w, int h, int part_idx, int xn, int yn) {
  const int n = 1 << cu->log2;
  int avail;
  if (xn >= cu->x && xn < cu->x + n && yn >= cu->y && yn < cu->y + n) {
    avail = !((w << 1) == n && (h << 1) == n && part_idx == 1 && cu->y + h <= yn && cu->x + w > xn);
  } else {
    avail = hv_zscan_available(d, xp, yp, xn, yn);
  }
  if (avail && d->m_pred[(yn >> 2) * d->w4 + (xn >> 2)] != 1) avail = 0; /* intra (or, in a damaged stream, nothing) */
  return avail;
}

/* --------------------------------------------------------------------------------------------- intra modes, 8.4.2 */
void hv_mpm_candidates(const HevcOracle* d, int x, int y, int cand[3]) {
  const HvSps* s = d->cur_sps;
  int a = 1, b = 1; /* INTRA_DC */
  if (hv_zscan_available(d, x, y, x - 1, y)) {
    const int i = (y >> 2) * d->w4 + ((x - 1) >> 2);
    if (d->m_pred[i] == 2 && !(d->m_flags[i] & 1)) a = d->m_imode[i];
  }
  if (hv_zscan_available(d, x, y, x, y - 1) && y - 1 >= ((y >> s->log2_ctb) << s->log2_ctb)) {
    const int i = ((y - 1) >> 2) * d->w4 + (x >> 2);
    if (d->m_pred[i] == 2 && !(d->m_flags[i] & 1)) b = d->m_imode[i];
  }
  if (a == b) {
    if (a < 2) {
      cand[0] = 0;
      cand[1] = 1;
      cand[2] = 26;
    } else {
      cand[0] = a;
      cand[1] = 2 + ((a + 29) % 32);
      cand[2] = 2 + ((a - 2 + 1) % 32);
    }
  } else {
    cand[0] = a;
    cand[1] = b;
    cand[2] = (a != 0 && b != 0) ? 0 : (a != 1 && b != 1) ? 1 : 26;
  }
}
int hv_chroma_mode_from_syntax(int v, int luma_mode) {
  static const uint8_t tab[4] = {0, 26, 10, 1};
  if (v == 4) return luma_mode;
  return tab[v] == luma_mode ? 34 : tab[v];
}
void hv_derive_intra_modes(HevcOracle* d, HvCu* cu) {
  const int parts = cu->part_mode == HV_PART_NxN ? 4 : 1;
  const int pb = (1 << cu->log2) >> (parts == 4);
  for (int i = 0; i < parts; ++i) {
    const int x = cu->x + (i & 1) * pb, y = cu->y + (i >> 1) * pb;
    int cand[3], mode;
    hv_mpm_candidates(d, x, y, cand);
    if (cu->prev_intra_luma_pred_flag[i]) {
      mode = cand[cu->mpm_idx[i]];
    } else {
      if (cand[0] > cand[1]) { int t = cand[0]; cand[0] = cand[1]; cand[1] = t; }
      if (cand[0] > cand[2]) { int t = cand[0]; cand[0] = cand[2]; cand[2] = t; }
      if (cand[1] > cand[2]) { int t = cand[1]; cand[1] = cand[2]; cand[2] = t; }
      mode = cu->rem_intra_luma_pred_mode[i];
      for (int k = 0; k < 3; ++k) mode += mode >= cand[k];
    }
    cu->luma_mode[i] = (uint8_t)mode;
    for (int yy = 0; yy < pb; yy += 4)
      for (int xx = 0; xx < pb; xx += 4) d->m_imode[((y + yy) >> 2) * d->w4 + ((x + xx) >> 2)] = (uint8_t)mode;
  }
  if (parts == 1) cu->luma_mode[1] = cu->luma_mode[2] = cu->luma_mode[3] = cu->luma_mode[0];
  cu->chroma_mode = (uint8_t)hv_chroma_mode_from_syntax(cu->intra_chroma_pred_mode, cu->luma_mode[0]);
}

/* ------------------------------------------------------------------------------------------ motion, 8.5.3.2.x */
int hv_num_parts(int part_mode) { return part_mode == HV_PART_2Nx2N ? 1 : part_mode == HV_PART_NxN ? 4 : 2; }
void hv_pu_geometry(const HvCu* cu, int i, int* x, int* y, int* w, int* h) {
  const int n = 1 << cu->log2, q = n >> 2;
  *x = cu->x;
  *y = cu->y;
  *w = n;
  *h = n;
  switch (cu->part_mode) {
    case HV_PART_2NxN: *h = n / 2; *y += i * (n / 2); break;
    case HV_PART_Nx2N: *w = n / 2; *x += i * (n / 2); break;
    case HV_PART_NxN: *w = *h = n / 2; *x += (i & 1) * (n / 2); *y += (i >> 1) * (n / 2); break;
    case HV_PART_2NxnU: *h = i ? n - q : q; *y += i ? q : 0; break;
    case HV_PART_2NxnD: *h = i ? q : n - q; *y += i ? n - q : 0; break;
    case HV_PART_nLx2N: *w = i ? n - q : q; *x += i ? q : 0; break;
    case HV_PART_nRx2N: *w = i ? q : n - q; *x += i ? n - q : 0; break;
    default: break;
  }
}

static inline const HvMv* mv_at(const HevcOracle* d, int x, int y) { return &d->cur->mvf[(y >> 2) * d->w4 + (x >> 2)]; }
static int same_motion(const HvMv* a, const HvMv* b) {
  if (a->pred_flag != b->pred_flag) return 0;
  for (int l = 0; l < 2; ++l)
    if (a->pred_flag & (1 << l))
      if (a->ref_idx[l] != b->ref_idx[l] || a->mv[l][0] != b->mv[l][0] || a->mv[l][1] != b->mv[l][1]) return 0;
  return 1;
}
static void scale_mv(int16_t mv[2], int td, int tb) {
  td = hv_clip3(-128, 127, td);
  tb = hv_clip3(-128, 127, tb);
  if (td == 0) return;
  const int tx = (16384 + (hv_abs(td) >> 1)) / td;
  const int f = hv_clip3(-4096, 4095, (tb * tx + 32) >> 6);
  for (int c = 0; c < 2; ++c) {
    const int v = f * mv[c];
    mv[c] = (int16_t)hv_clip3(-32768, 32767, hv_sign(v) * ((hv_abs(v) + 127) >> 8));
  }
}

/* 8.5.3.2.8 + 8.5.3.2.9: temporal luma motion vector prediction for list X and refIdxLX */
static int temporal_mv(HevcOracle* d, int xp, int yp, int w, int h, int ref_idx, int X, int16_t out[2]) {
  const HvSps* s = d->cur_sps;
  const HvSliceHeader* sh = &d->sh;
  if (!sh->temporal_mvp) return 0;
  const HvPic* col = d->ref_list[(sh->slice_type == HV_SLICE_B && !sh->collocated_from_l0) ? 1 : 0][sh->collocated_ref_idx];
  if (!col || !col->mvf) return 0;
  const HvPic* target = d->ref_list[X][ref_idx];
  const int cur_long = target->ref_state == 2;
  for (int attempt = 0; attempt < 2; ++attempt) {
    int xc, yc;
    if (attempt == 0) {
      xc = xp + w;
      yc = yp + h;
      if ((yp >> s->log2_ctb) != (yc >> s->log2_ctb) || yc >= s->height || xc >= s->width) continue;
    } else {
      xc = xp + (w >> 1);
      yc = yp + (h >> 1);
    }
    xc = (xc >> 4) << 4;
    yc = (yc >> 4) << 4;
    const HvMv* c = &col->mvf[(yc >> 2) * d->w4 + (xc >> 2)];
    if (!c->pred_flag) continue;
    int lc;
    if (!(c->pred_flag & 1)) {
      lc = 1;
    } else if (c->pred_flag == 1) {
      lc = 0;
    } else {
      int no_backward = 1; /* NoBackwardPredFlag: no reference of the slice follows the current picture */
      for (int l = 0; l < (sh->slice_type == HV_SLICE_B ? 2 : 1); ++l)
        for (int i = 0; i < sh->num_ref_idx[l]; ++i) no_backward &= d->ref_list[l][i]->poc <= d->cur->poc;
      lc = no_backward ? X : sh->collocated_from_l0;
    }
    if (cur_long != c->ref_long[lc]) continue;
    out[0] = c->mv[lc][0];
    out[1] = c->mv[lc][1];
    const int col_diff = col->poc - c->ref_poc[lc], cur_diff = d->cur->poc - target->poc;
    if (!cur_long && col_diff != cur_diff) scale_mv(out, col_diff, cur_diff);
    return 1;
  }
  return 0;
}

/* 8.5.3.2.2 - 8.5.3.2.5: the whole merging candidate list (MaxNumMergeCand entries) */
void hv_merge_candidates(HevcOracle* d, const HvCu* cu, int xp, int yp, int w, int h, int part_idx, HvMv out[5], int* count) {
  const HvSliceHeader* sh = &d->sh;
  const int pml = d->cur_pps->log2_parallel_merge_level;
  int part_mode = cu->part_mode;
  if (pml > 2 && cu->log2 == 3) { /* singleMCLFlag: all prediction units of an 8x8 unit share the 2Nx2N list */
    xp = cu->x;
    yp = cu->y;
    w = h = 8;
    part_idx = 0;
    part_mode = HV_PART_2Nx2N;
  }
  const int max_cand = sh->max_num_merge_cand;
  int n = 0;
  memset(out, 0, sizeof(HvMv) * 5);
#define SAME_MER(xn, yn) ((xp >> pml) == ((xn) >> pml) && (yp >> pml) == ((yn) >> pml))
  const int xa1 = xp - 1, ya1 = yp + h - 1, xb1 = xp + w - 1, yb1 = yp - 1, xb0 = xp + w, yb0 = yp - 1, xa0 = xp - 1, ya0 = yp + h,
            xb2 = xp - 1, yb2 = yp - 1;
  int av_a1 = !SAME_MER(xa1, ya1) &&
              !(part_idx == 1 && (part_mode == HV_PART_Nx2N || part_mode == HV_PART_nLx2N || part_mode == HV_PART_nRx2N)) &&
              pb_available(d, cu, xp, yp, w, h, part_idx, xa1, ya1);
  const HvMv* a1 = av_a1 ? mv_at(d, xa1, ya1) : NULL;
  int av_b1 = !SAME_MER(xb1, yb1) &&
              !(part_idx == 1 && (part_mode == HV_PART_2NxN || part_mode == HV_PART_2NxnU || part_mode == HV_PART_2NxnD)) &&
              pb_available(d, cu, xp, yp, w, h, part_idx, xb1, yb1);
  const HvMv* b1 = av_b1 ? mv_at(d, xb1, yb1) : NULL;
  if (av_b1 && av_a1 && same_motion(a1, b1)) av_b1 = 0;
  int av_b0 = !SAME_MER(xb0, yb0) && pb_available(d, cu, xp, yp, w, h, part_idx, xb0, yb0);
  const HvMv* b0 = av_b0 ? mv_at(d, xb0, yb0) : NULL;
  if (av_b0 && b1 && same_motion(b1, b0)) av_b0 = 0;
  int av_a0 = !SAME_MER(xa0, ya0) && pb_available(d, cu, xp, yp, w, h, part_idx, xa0, ya0);
  const HvMv* a0 = av_a0 ? mv_at(d, xa0, ya0) : NULL;
  if (av_a0 && a1 && same_motion(a1, a0)) av_a0 = 0;
  int av_b2 = !SAME_MER(xb2, yb2) && pb_available(d, cu, xp, yp, w, h, part_idx, xb2, yb2);
  const HvMv* b2 = av_b2 ? mv_at(d, xb2, yb2) : NULL;
  if (av_b2 && ((a1 && same_motion(a1, b2)) || (b1 && same_motion(b1, b2)))) av_b2 = 0;
  if (av_a0 + av_a1 + av_b0 + av_b1 == 4) av_b2 = 0;
#undef SAME_MER
  /* (b1 / a1 above are the neighbours' motion whether or not they were pruned: the comparisons of 8.5.3.2.3 are
   * against the blocks, and a pruned block is still "available" for them only if it was available itself) */
  if (av_a1) out[n++] = *a1;
  if (av_b1) out[n++] = *b1;
  if (av_b0) out[n++] = *b0;
  if (av_a0) out[n++] = *a0;
  if (av_b2) out[n++] = *b2;
  if (n < max_cand || 1) { /* temporal candidate: refIdxLXCol = 0 */
    HvMv t;
    memset(&t, 0, sizeof(t));
    int16_t mv[2];
    if (temporal_mv(d, xp, yp, w, h, 0, 0, mv)) {
      t.pred_flag |= 1;
      t.mv[0][0] = mv[0];
      t.mv[0][1] = mv[1];
    }
    if (sh->slice_type == HV_SLICE_B && temporal_mv(d, xp, yp, w, h, 0, 1, mv)) {
      t.pred_flag |= 2;
      t.mv[1][0] = mv[0];
      t.mv[1][1] = mv[1];
    }
    if (t.pred_flag && n < 5) out[n++] = t;
  }
  if (n > max_cand) n = max_cand;
  /* 8.5.3.2.4: combined bi-predictive candidates */
  if (sh->slice_type == HV_SLICE_B && n > 1 && n < max_cand) {
    static const uint8_t l0c[12] = {0, 1, 0, 2, 1, 2, 0, 3, 1, 3, 2, 3}, l1c[12] = {1, 0, 2, 0, 2, 1, 3, 0, 3, 1, 3, 2};
    const int orig = n;
    for (int k = 0; k < orig * (orig - 1) && n < max_cand; ++k) {
      const HvMv* c0 = &out[l0c[k]];
      const HvMv* c1 = &out[l1c[k]];
      if ((c0->pred_flag & 1) && (c1->pred_flag & 2) &&
          (d->ref_list[0][c0->ref_idx[0]]->poc != d->ref_list[1][c1->ref_idx[1]]->poc || c0->mv[0][0] != c1->mv[1][0] ||
           c0->mv[0][1] != c1->mv[1][1])) {
        HvMv m;
        memset(&m, 0, sizeof(m));
        m.pred_flag = 3;
        m.ref_idx[0] = c0->ref_idx[0];
        m.ref_idx[1] = c1->ref_idx[1];
        m.mv[0][0] = c0->mv[0][0];
        m.mv[0][1] = c0->mv[0][1];
        m.mv[1][0] = c1->mv[1][0];
        m.mv[1][1] = c1->mv[1][1];
        out[n++] = m;
      }
    }
  }
  /* 8.5.3.2.5: zero candidates */
  const int num_ref = sh->slice_type == HV_SLICE_P ? sh->num_ref_idx[0] : hv_min(sh->num_ref_idx[0], sh->num_ref_idx[1]);
  for (int zero = 0; n < max_cand; ++zero) {
    HvMv m;
    memset(&m, 0, sizeof(m));
    const int r = zero < num_ref ? zero : 0;
    m.pred_flag = sh->slice_type == HV_SLICE_P ? 1 : 3;
    m.ref_idx[0] = (int8_t)r;
    m.ref_idx[1] = (int8_t)(sh->slice_type == HV_SLICE_P ? -1 : r);
    out[n++] = m;
  }
  *count = n;
}

/* 8.5.3.2.6 + 8.5.3.2.7: the two motion vector predictor candidates of list X */
void hv_amvp_candidates(HevcOracle* d, const HvCu* cu, int xp, int yp, int w, int h, int part_idx, int X, int ref_idx,
                        int16_t out[2][2]) {
  const HvPic* target = d->ref_list[X][ref_idx];
  const int target_long = target->ref_state == 2;
  const int nx[5] = {xp - 1, xp - 1, xp + w, xp + w - 1, xp - 1}, ny[5] = {yp + h, yp + h - 1, yp - 1, yp - 1, yp - 1}; /* A0 A1 B0 B1 B2 */
  int av[5];
  for (int k = 0; k < 5; ++k) av[k] = pb_available(d, cu, xp, yp, w, h, part_idx, nx[k], ny[k]);
  int16_t mva[2] = {0, 0}, mvb[2] = {0, 0};
  int have_a = 0, have_b = 0;
  const int is_scaled = av[0] || av[1];
  /* A without scaling, then with */
  for (int k = 0; k < 2 && !have_a; ++k) {
    if (!av[k]) continue;
    const HvMv* m = mv_at(d, nx[k], ny[k]);
    for (int t = 0; t < 2 && !have_a; ++t) {
      const int l = t ? !X : X;
      if ((m->pred_flag & (1 << l)) && d->ref_list[l][m->ref_idx[l]]->poc == target->poc) {
        have_a = 1;
        mva[0] = m->mv[l][0];
        mva[1] = m->mv[l][1];
      }
    }
  }
  for (int k = 0; k < 2 && !have_a; ++k) {
    if (!av[k]) continue;
    const HvMv* m = mv_at(d, nx[k], ny[k]);
    for (int t = 0; t < 2 && !have_a; ++t) {
      const int l = t ? !X : X;
      if (!(m->pred_flag & (1 << l))) continue;
      const HvPic* r = d->ref_list[l][m->ref_idx[l]];
      if ((r->ref_state == 2) != target_long) continue;
      have_a = 1;
      mva[0] = m->mv[l][0];
      mva[1] = m->mv[l][1];
      if (!target_long && r->ref_state != 2) scale_mv(mva, d->cur->poc - r->poc, d->cur->poc - target->poc);
    }
  }
  /* B without scaling */
  for (int k = 2; k < 5 && !have_b; ++k) {
    if (!av[k]) continue;
    const HvMv* m = mv_at(d, nx[k], ny[k]);
    for (int t = 0; t < 2 && !have_b; ++t) {
      const int l = t ? !X : X;
      if ((m->pred_flag & (1 << l)) && d->ref_list[l][m->ref_idx[l]]->poc == target->poc) {
        have_b = 1;
        mvb[0] = m->mv[l][0];
        mvb[1] = m->mv[l][1];
      }
    }
  }
  if (!is_scaled && have_b) {
    have_a = 1;
    mva[0] = mvb[0];
    mva[1] = mvb[1];
  }
  if (!is_scaled) {
    have_b = 0;
    for (int k = 2; k < 5 && !have_b; ++k) {
      if (!av[k]) continue;
      const HvMv* m = mv_at(d, nx[k], ny[k]);
      for (int t = 0; t < 2 && !have_b; ++t) {
        const int l = t ? !X : X;
        if (!(m->pred_flag & (1 << l))) continue;
        const HvPic* r = d->ref_list[l][m->ref_idx[l]];
        if ((r->ref_state == 2) != target_long) continue;
        have_b = 1;
        mvb[0] = m->mv[l][0];
        mvb[1] = m->mv[l][1];
        if (!target_long && r->ref_state != 2) scale_mv(mvb, d->cur->poc - r->poc, d->cur->poc - target->poc);
      }
    }
  }
  int n = 0;
  if (have_a) {
    out[n][0] = mva[0];
    out[n++][1] = mva[1];
  }
  if (have_b && !(have_a && mva[0] == mvb[0] && mva[1] == mvb[1])) {
    out[n][0] = mvb[0];
    out[n++][1] = mvb[1];
  }
  if (n < 2) {
    int16_t mvc[2];
    if (temporal_mv(d, xp, yp, w, h, ref_idx, X, mvc)) {
      out[n][0] = mvc[0];
      out[n++][1] = mvc[1];
    }
  }
  for (; n < 2; ++n) out[n][0] = out[n][1] = 0;
}

void hv_store_pu_motion(HevcOracle* d, const HvPu* pu) {
  HvMv m = pu->mv;
  for (int l = 0; l < 2; ++l) {
    if (m.pred_flag & (1 << l)) {
      const HvPic* r = d->ref_list[l][m.ref_idx[l]];
      m.ref_poc[l] = r->poc;
      m.ref_long[l] = r->ref_state == 2;
    } else {
      m.ref_idx[l] = -1;
      m.mv[l][0] = m.mv[l][1] = 0;
      m.ref_poc[l] = 0;
      m.ref_long[l] = 0;
    }
  }
  for (int y = pu->y; y < pu->y + pu->h; y += 4)
    for (int x = pu->x; x < pu->x + pu->w; x += 4) d->cur->mvf[(y >> 2) * d->w4 + (x >> 2)] = m;
}

/* 8.5.3.2.1: from the syntax elements of a prediction unit to its motion */
void hv_derive_pu_motion(HevcOracle* d, HvCu* cu, int part_idx) {
  HvPu* pu = &cu->pu[part_idx];
  hv_pu_geometry(cu, part_idx, &pu->x, &pu->y, &pu->w, &pu->h);
  if (pu->merge_flag) {
    HvMv cand[5];
    int n;
    hv_merge_candidates(d, cu, pu->x, pu->y, pu->w, pu->h, part_idx, cand, &n);
    pu->mv = cand[pu->merge_idx < n ? pu->merge_idx : 0];
    if (pu->mv.pred_flag == 3 && pu->w + pu->h == 12) { /* 8x4 / 4x8: uni-prediction only */
      pu->mv.pred_flag = 1;
      pu->mv.ref_idx[1] = -1;
    }
  } else {
    memset(&pu->mv, 0, sizeof(pu->mv));
    pu->mv.pred_flag = (uint8_t)(pu->inter_pred_idc == 2 ? 3 : pu->inter_pred_idc == 1 ? 2 : 1);
    for (int l = 0; l < 2; ++l) {
      pu->mv.ref_idx[l] = -1;
      if (!(pu->mv.pred_flag & (1 << l))) continue;
      pu->mv.ref_idx[l] = pu->ref_idx[l];
      int16_t cand[2][2];
      hv_amvp_candidates(d, cu, pu->x, pu->y, pu->w, pu->h, part_idx, l, pu->ref_idx[l], cand);
      for (int c = 0; c < 2; ++c) pu->mv.mv[l][c] = (int16_t)(cand[pu->mvp_flag[l]][c] + pu->mvd[l][c]); /* (8-272 ..: wraps at 16 bits) */
    }
  }
  hv_store_pu_motion(d, pu);
}

/* ------------------------------------------------------------------------------------------------------ QP, 8.6.1 */
int hv_chroma_qp(int qp_y, int offset) {
  const int qpi = hv_clip3(0, 57, qp_y + offset);
  return qpi < 30 ? qpi : qpi >= 43 ? qpi - 6 : hevc_qpc_table[qpi - 30];
}
int hv_qp_y_pred(HevcOracle* d, int xq, int yq) {
  const HvSps* s = d->cur_sps;
  int a = d->qp_y_prev, b = d->qp_y_prev;
  if (xq > 0 && ((xq - 1) >> s->log2_ctb) == (xq >> s->log2_ctb)) a = d->m_qp[(yq >> 2) * d->w4 + ((xq - 1) >> 2)];
  if (yq > 0 && ((yq - 1) >> s->log2_ctb) == (yq >> s->log2_ctb)) b = d->m_qp[((yq - 1) >> 2) * d->w4 + (xq >> 2)];
  return (a + b + 1) >> 1;
}

void hv_set_cu_maps(HevcOracle* d, const HvCu* cu) {
  const int n = 1 << cu->log2;
  const int lf_off = cu->tq_bypass || (cu->pcm && d->cur_sps->pcm_loop_filter_disabled);
  for (int y = cu->y; y < cu->y + n; y += 4)
    for (int x = cu->x; x < cu->x + n; x += 4) {
      const int i = (y >> 2) * d->w4 + (x >> 2);
      d->m_pred[i] = cu->intra ? 2 : 1;
      d->m_skip[i] = cu->skip;
      d->m_depth[i] = cu->depth;
      d->m_flags[i] = (uint8_t)((cu->pcm ? 1 : 0) | (cu->tq_bypass ? 2 : 0) | (lf_off ? 4 : 0));
      d->m_qp[i] = cu->qp_y;
      if (cu->intra) {
        d->m_imode[i] = 1;
        memset(&d->cur->mvf[i], 0, sizeof(HvMv));
      }
    }
}
static void set_cu_qp_map(HevcOracle* d, const HvCu* cu) {
  const int n = 1 << cu->log2;
  for (int y = cu->y; y < cu->y + n; y += 4)
    for (int x = cu->x; x < cu->x + n; x += 4) d->m_qp[(y >> 2) * d->w4 + (x >> 2)] = cu->qp_y;
}

/* ------------------------------------------------------------------------------------- residual_coding, 7.3.8.11 */
static int code_last_prefix(HevcOracle* d, int base, int v, int log2, int c_idx) {
  int off, shift;
  if (c_idx == 0) {
    off = 3 * (log2 - 2) + ((log2 - 1) >> 2);
    shift = (log2 + 1) >> 2;
  } else {
    off = 15;
    shift = log2 - 2;
  }
  const int cmax = (log2 << 1) - 1;
  int n = 0;
  while (n < cmax && cbin(d, base + off + (n >> shift), n < v)) ++n;
  return n;
}

static int residual_coding(HevcOracle* d, const HvCu* cu, HvTu* tu, int c_idx, int log2, int16_t* lev /* raster n x n */, int intra_mode) {
  const HvPps* pps = d->cur_pps;
  const int n = 1 << log2;
  if (pps->transform_skip_enabled && !cu->tq_bypass && log2 == 2)
    tu->tskip[c_idx] = (uint8_t)cbin(d, HC_TRANSFORM_SKIP + (c_idx ? 1 : 0), tu->tskip[c_idx]);
  int scan_idx = 0;
  if (cu->intra && (log2 == 2 || (log2 == 3 && c_idx == 0))) {
    if (intra_mode >= 6 && intra_mode <= 14) scan_idx = 2;
    if (intra_mode >= 22 && intra_mode <= 30) scan_idx = 1;
  }
  tu->scan_idx[c_idx] = (uint8_t)scan_idx;
  const uint8_t* sb_scan = hv_scan(log2 - 2, scan_idx);
  const uint8_t* pos_scan = hv_scan(2, scan_idx);
  int last_x = 0, last_y = 0;
  if (d->enc) { /* the last significant level in scan order */
    int found = 0;
    for (int i = (1 << (2 * (log2 - 2))) - 1; i >= 0 && !found; --i)
      for (int k = 15; k >= 0 && !found; --k) {
        const int x = ((sb_scan[i] & 15) << 2) + (pos_scan[k] & 15), y = ((sb_scan[i] >> 4) << 2) + (pos_scan[k] >> 4);
        if (lev[y * n + x]) {
          last_x = x;
          last_y = y;
          found = 1;
        }
      }
    if (!found) return -1; /* a coded block flag promised a level */
    if (scan_idx == 2) { int t = last_x; last_x = last_y; last_y = t; } /* coded swapped */
  }
  {
    int px = 0, py = 0;
    if (d->enc) { /* binarisation of a position: prefix + fixed-length suffix */
      static const uint8_t group[32] = {0, 1, 2, 3, 4, 4, 5, 5, 6, 6, 6, 6, 7, 7, 7, 7, 8, 8, 8, 8, 8, 8, 8, 8, 9, 9, 9, 9, 9, 9, 9, 9};
      px = group[last_x];
      py = group[last_y];
    }
    px = code_last_prefix(d, HC_LAST_X, px, log2, c_idx);
    py = code_last_prefix(d, HC_LAST_Y, py, log2, c_idx);
    int vx = px, vy = py;
    if (px > 3) {
      const int bits = (px >> 1) - 1, base = (1 << bits) * (2 + (px & 1));
      vx = base + (int)cbyp_bits(d, (unsigned)(last_x - base), bits);
    }
    if (py > 3) {
      const int bits = (py >> 1) - 1, base = (1 << bits) * (2 + (py & 1));
      vy = base + (int)cbyp_bits(d, (unsigned)(last_y - base), bits);
    }
    last_x = vx;
    last_y = vy;
  }
  if (scan_idx == 2) { int t = last_x; last_x = last_y; last_y = t; }
  if (last_x >= n || last_y >= n) return -1;
  /* locate the last position in the scan */
  int last_sb = (1 << (2 * (log2 - 2))) - 1, last_pos = 16;
  for (;;) {
    if (last_pos == 0) {
      last_pos = 16;
      --last_sb;
      if (last_sb < 0) return -1;
    }
    --last_pos;
    const int x = ((sb_scan[last_sb] & 15) << 2) + (pos_scan[last_pos] & 15), y = ((sb_scan[last_sb] >> 4) << 2) + (pos_scan[last_pos] >> 4);
    if (x == last_x && y == last_y) break;
  }
  uint8_t csbf[8][8];
  memset(csbf, 0, sizeof(csbf));
  const int sbw = 1 << (log2 - 2);
  int c1 = 1; /* greater1Ctx state carried from the previous coded sub-block (9.3.4.2.6) */
  int first_coded_sb = 1;
  for (int i = last_sb; i >= 0; --i) {
    const int xs = sb_scan[i] & 15, ys = sb_scan[i] >> 4;
    int infer_dc = 0;
    int coded = 1;
    if (i < last_sb && i > 0) {
      const int right = xs + 1 < sbw ? csbf[ys][xs + 1] : 0, below = ys + 1 < sbw ? csbf[ys + 1][xs] : 0;
      if (d->enc) {
        coded = 0;
        for (int k = 0; k < 16; ++k) coded |= lev[((ys << 2) + (pos_scan[k] >> 4)) * n + (xs << 2) + (pos_scan[k] & 15)] != 0;
      }
      coded = cbin(d, HC_CODED_SUB_BLOCK + hv_min(right | below, 1) + (c_idx ? 2 : 0), coded);
      infer_dc = 1;
    }
    csbf[ys][xs] = (uint8_t)coded;
    uint8_t sig[16];
    memset(sig, 0, sizeof(sig));
    const int start = i == last_sb ? last_pos - 1 : 15;
    if (i == last_sb) sig[last_pos] = 1;
    if (coded) {
      const int right = xs + 1 < sbw ? csbf[ys][xs + 1] : 0, below = ys + 1 < sbw ? csbf[ys + 1][xs] : 0;
      const int prev_csbf = right | below << 1;
      for (int k = start; k >= 0; --k) {
        const int xp = pos_scan[k] & 15, yp = pos_scan[k] >> 4, xc = (xs << 2) + xp, yc = (ys << 2) + yp;
        if (k > 0 || !infer_dc) {
          int sctx;
          if (log2 == 2) {
            sctx = hevc_ctx_idx_map4[(yc << 2) + xc];
          } else if (xc + yc == 0) {
            sctx = 0;
          } else {
            if (prev_csbf == 0)
              sctx = (xp + yp == 0) ? 2 : (xp + yp < 3) ? 1 : 0;
            else if (prev_csbf == 1)
              sctx = yp == 0 ? 2 : yp == 1 ? 1 : 0;
            else if (prev_csbf == 2)
              sctx = xp == 0 ? 2 : xp == 1 ? 1 : 0;
            else
              sctx = 2;
            if (c_idx == 0) {
              if (xs || ys) sctx += 3;
              sctx += log2 == 3 ? (scan_idx == 0 ? 9 : 15) : 21;
            } else {
              sctx += log2 == 3 ? 9 : 12;
            }
          }
          sig[k] = (uint8_t)cbin(d, HC_SIG_COEFF + (c_idx ? 27 + sctx : sctx), d->enc ? lev[yc * n + xc] != 0 : 0);
          if (sig[k]) infer_dc = 0;
        } else {
          sig[k] = 1; /* k == 0 and nothing else in a sub-block that was flagged as coded */
        }
      }
    }
    int nsig = 0;
    for (int k = 0; k < 16; ++k) nsig += sig[k];
    if (!nsig) continue;
    /* greater-than-1 flags of the first eight, one greater-than-2 flag */
    int ctx_set = (i == 0 || c_idx > 0) ? 0 : 2;
    if (!first_coded_sb && c1 == 0) ++ctx_set;
    first_coded_sb = 0;
    c1 = 1;
    uint8_t g1[16], g2[16];
    memset(g1, 0, sizeof(g1));
    memset(g2, 0, sizeof(g2));
    int first_sig = 16, last_sig = -1, num_g1 = 0, last_g1_pos = -1;
    int absval[16];
    memset(absval, 0, sizeof(absval));
    if (d->enc)
      for (int k = 0; k < 16; ++k) absval[k] = hv_abs(lev[((ys << 2) + (pos_scan[k] >> 4)) * n + (xs << 2) + (pos_scan[k] & 15)]);
    for (int k = 15; k >= 0; --k) {
      if (!sig[k]) continue;
      if (num_g1 < 8) {
        g1[k] = (uint8_t)cbin(d, HC_GREATER1 + ctx_set * 4 + c1 + (c_idx ? 16 : 0), absval[k] > 1);
        ++num_g1;
        if (g1[k]) {
          c1 = 0;
          if (last_g1_pos == -1) last_g1_pos = k;
        } else if (c1 > 0 && c1 < 3) {
          ++c1;
        }
      }
      if (last_sig == -1) last_sig = k;
      first_sig = k;
    }
    const int sign_hidden = !cu->tq_bypass && pps->sign_data_hiding && last_sig - first_sig > 3;
    if (last_g1_pos != -1) g2[last_g1_pos] = (uint8_t)cbin(d, HC_GREATER2 + ctx_set + (c_idx ? 4 : 0), absval[last_g1_pos] > 2);
    uint8_t neg[16];
    memset(neg, 0, sizeof(neg));
    for (int k = 15; k >= 0; --k) {
      if (!sig[k]) continue;
      const int yc = (ys << 2) + (pos_scan[k] >> 4), xc = (xs << 2) + (pos_scan[k] & 15);
      if (!sign_hidden || k != first_sig) neg[k] = (uint8_t)cbyp(d, d->enc ? lev[yc * n + xc] < 0 : 0);
    }
    int num_sig = 0, sum_abs = 0, rice = 0;
    for (int k = 15; k >= 0; --k) {
      if (!sig[k]) continue;
      const int yc = (ys << 2) + (pos_scan[k] >> 4), xc = (xs << 2) + (pos_scan[k] & 15);
      const int base = 1 + g1[k] + g2[k];
      int a = base;
      if (base == ((num_sig < 8) ? ((k == last_g1_pos) ? 3 : 2) : 1)) {
        /* coeff_abs_level_remaining: prefix of up to four ones + rice bits, then Exp-Golomb of order rice + 1 (9.3.3.11) */
        unsigned rem = d->enc ? (unsigned)(absval[k] - base) : 0;
        if (d->enc) {
          if ((rem >> rice) < 4) {
            cbyp_tu(d, (int)(rem >> rice), 4);
            cbyp_bits(d, rem & ((1u << rice) - 1), rice);
          } else {
            cbyp_tu(d, 4, 4);
            cbyp_egk(d, rem - (4u << rice), rice + 1);
          }
        } else {
          const int pre = cbyp_tu(d, 0, 4);
          if (pre < 4)
            rem = ((unsigned)pre << rice) + cbyp_bits(d, 0, rice);
          else
            rem = (4u << rice) + cbyp_egk(d, 0, rice + 1);
          if (rem > 32768) return -1;
        }
        a = base + (int)rem;
        if (a > 3 * (1 << rice)) rice = hv_min(rice + 1, 4);
      }
      if (!d->enc) {
        int v = neg[k] ? -a : a;
        if (sign_hidden) {
          sum_abs += a;
          if (k == first_sig && (sum_abs & 1)) v = -v;
        }
        lev[yc * n + xc] = (int16_t)hv_clip3(-32768, 32767, v);
      }
      ++num_sig;
    }
  }
  return 0;
}

/* sign data hiding on the encoder's side: make the parity of every eligible sub-block say what the sign of its
 * first level (in scan order) is, by moving the level at the high-frequency end one step away from zero */
void hv_sdh_fix(int16_t* lev, int log2, int scan_idx) {
  const int n = 1 << log2;
  const uint8_t* sb_scan = hv_scan(log2 - 2, scan_idx);
  const uint8_t* pos_scan = hv_scan(2, scan_idx);
  for (int i = 0; i < (1 << (2 * (log2 - 2))); ++i) {
    int first = 16, last = -1, sum = 0;
    int16_t* at[16];
    for (int k = 0; k < 16; ++k) {
      at[k] = &lev[(((sb_scan[i] >> 4) << 2) + (pos_scan[k] >> 4)) * n + ((sb_scan[i] & 15) << 2) + (pos_scan[k] & 15)];
      if (*at[k]) {
        if (first == 16) first = k;
        last = k;
        sum += hv_abs(*at[k]);
      }
    }
    if (last - first <= 3) continue;
    if ((sum & 1) != (*at[first] < 0)) *at[last] = (int16_t)(*at[last] + (*at[last] < 0 ? -1 : 1));
  }
}

/* --------------------------------------------------------------------------------- transform_tree, 7.3.8.8 - 10 */
typedef struct {
  HevcOracle* d;
  HvCu* cu;
  int next_node; /* enc: position in the unit's stored node list */
  int next_tu;
  int max_depth;
  int fail;
} TtWalk;

static void code_cu_qp_delta(HevcOracle* d, HvCu* cu) {
  int v = cu->qp_delta, a = hv_abs(v);
  /* cu_qp_delta_abs: prefix TU(5) with two contexts, suffix EG0 */
  int pre = 0;
  while (pre < 5 && cbin(d, HC_QP_DELTA + (pre ? 1 : 0), pre < a)) ++pre;
  int val = pre;
  if (pre == 5) val = 5 + (int)cbyp_egk(d, (unsigned)(a - 5), 0);
  int neg = 0;
  if (val) neg = cbyp(d, v < 0);
  if (!d->enc) {
    cu->qp_delta = (int8_t)(neg ? -val : val);
    cu->qp_delta_coded = 1;
  }
  d->is_cu_qp_delta_coded = 1;
  d->cu_qp_delta_val = cu->qp_delta;
  if (!d->enc) {
    const int pred = hv_qp_y_pred(d, d->qg_x, d->qg_y);
    cu->qp_y = (int8_t)(((pred + cu->qp_delta + 52) % 52));
  }
}

static void transform_tree(TtWalk* w, int x0, int y0, int xb, int yb, int log2, int depth, int blk, int parent_cb, int parent_cr) {
  HevcOracle* d = w->d;
  HvCu* cu = w->cu;
  const HvSps* s = d->cur_sps;
  if (w->fail) return;
  HvNode* node;
  if (d->enc) {
    node = &d->nodes[cu->first_node + w->next_node++];
  } else {
    const int at = hv_new_nodes(d, 1);
    if (at < 0) { w->fail = 1; return; }
    node = &d->nodes[at];
    ++cu->num_nodes;
  }
  const int intra_split = cu->intra && cu->part_mode == HV_PART_NxN;
  const int inter_split = s->max_th_depth_inter == 0 && !cu->intra && cu->part_mode != HV_PART_2Nx2N && depth == 0;
  int split;
  if (log2 <= s->log2_max_tb && log2 > s->log2_min_tb && depth < w->max_depth && !(intra_split && depth == 0))
    split = cbin(d, HC_SPLIT_TRANSFORM + 5 - log2, node->split);
  else
    split = log2 > s->log2_max_tb || (intra_split && depth == 0) || inter_split;
  node->split = (uint8_t)split;
  int cb = 0, cr = 0;
  if (log2 > 2) {
    if (depth == 0 || parent_cb) cb = cbin(d, HC_CBF_CHROMA + depth, node->cbf_cb);
    if (depth == 0 || parent_cr) cr = cbin(d, HC_CBF_CHROMA + depth, node->cbf_cr);
  } else if (depth > 0) {
    cb = parent_cb;
    cr = parent_cr;
  }
  node->cbf_cb = (uint8_t)cb;
  node->cbf_cr = (uint8_t)cr;
  if (split) {
    const int h = 1 << (log2 - 1);
    for (int i = 0; i < 4; ++i) transform_tree(w, x0 + (i & 1) * h, y0 + (i >> 1) * h, x0, y0, log2 - 1, depth + 1, i, cb, cr);
    return;
  }
  /* transform_unit */
  HvTu* tu;
  if (d->enc) {
    tu = &d->tus[cu->first_tu + w->next_tu++];
  } else {
    tu = hv_new_tu(d);
    if (!tu) { w->fail = 1; return; }
    ++cu->num_tu;
    tu->x = (uint16_t)x0;
    tu->y = (uint16_t)y0;
    tu->log2 = (uint8_t)log2;
    if (log2 > 2) {
      tu->has_chroma = 1;
      tu->cx = (uint16_t)(x0 >> 1);
      tu->cy = (uint16_t)(y0 >> 1);
      tu->clog2 = (uint8_t)(log2 - 1);
    } else if (blk == 3) {
      tu->has_chroma = 1;
      tu->cx = (uint16_t)(xb >> 1);
      tu->cy = (uint16_t)(yb >> 1);
      tu->clog2 = 2;
    }
  }
  int cbf_luma = 1;
  if (cu->intra || depth != 0 || cb || cr) cbf_luma = cbin(d, HC_CBF_LUMA + (depth == 0 ? 1 : 0), node->cbf_luma);
  node->cbf_luma = (uint8_t)cbf_luma;
  /* (for a 4x4 luma block cb / cr are the flags of the 8x8 parent: the chroma residual travels with the fourth block) */
  const int chroma_here = log2 > 2 || blk == 3;
  if (!d->enc) {
    tu->cbf[0] = (uint8_t)cbf_luma;
    tu->cbf[1] = (uint8_t)(chroma_here && cb);
    tu->cbf[2] = (uint8_t)(chroma_here && cr);
  }
  if (cbf_luma || cb || cr) {
    if (d->cur_pps->cu_qp_delta_enabled && !d->is_cu_qp_delta_coded) code_cu_qp_delta(d, cu);
    const int pu_idx = intra_split ? (((y0 - cu->y) >= (1 << (cu->log2 - 1))) << 1 | ((x0 - cu->x) >= (1 << (cu->log2 - 1)))) : 0;
    for (int c = 0; c < 3; ++c) {
      if (c == 0 ? !cbf_luma : !(chroma_here && (c == 1 ? cb : cr))) continue;
      const int l2 = c == 0 ? log2 : (log2 > 2 ? log2 - 1 : 2);
      if (!d->enc) {
        tu->coef[c] = hv_new_coefs(d, 1 << (2 * l2));
        if (tu->coef[c] < 0) { w->fail = 1; return; }
      }
      const int mode = c == 0 ? cu->luma_mode[pu_idx] : cu->chroma_mode;
      if (residual_coding(d, cu, tu, c, l2, &d->coefs[tu->coef[c]], mode)) { w->fail = 1; return; }
    }
  }
  ++d->counters[11];
}

/* ----------------------------------------------------------------------------------- prediction_unit, 7.3.8.6 / 9 */
static void code_mvd(HevcOracle* d, int16_t mvd[2]) {
  int g0[2], g1[2] = {0, 0};
  for (int c = 0; c < 2; ++c) g0[c] = cbin(d, HC_MVD_G0, mvd[c] != 0);
  for (int c = 0; c < 2; ++c)
    if (g0[c]) g1[c] = cbin(d, HC_MVD_G1, hv_abs(mvd[c]) > 1);
  for (int c = 0; c < 2; ++c) {
    int a = g0[c];
    int neg = 0;
    if (g0[c]) {
      if (g1[c]) a = 2 + (int)cbyp_egk(d, (unsigned)(hv_abs(mvd[c]) - 2), 1);
      neg = cbyp(d, mvd[c] < 0);
    }
    if (!d->enc) mvd[c] = (int16_t)(neg ? -a : a);
  }
}

static void prediction_unit(HevcOracle* d, HvCu* cu, int part_idx) {
  const HvSliceHeader* sh = &d->sh;
  HvPu* pu = &cu->pu[part_idx];
  int x, y, w, h;
  hv_pu_geometry(cu, part_idx, &x, &y, &w, &h);
  if (cu->skip) {
    pu->merge_flag = 1;
  } else {
    pu->merge_flag = (uint8_t)cbin(d, HC_MERGE_FLAG, pu->merge_flag);
  }
  if (pu->merge_flag) {
    if (sh->max_num_merge_cand > 1) {
      int v = 0;
      if (cbin(d, HC_MERGE_IDX, pu->merge_idx > 0)) {
        v = 1;
        while (v < sh->max_num_merge_cand - 1 && cbyp(d, pu->merge_idx > v)) ++v;
      }
      pu->merge_idx = (uint8_t)v;
    } else {
      pu->merge_idx = 0;
    }
    return;
  }
  if (sh->slice_type == HV_SLICE_B) {
    int idc;
    if (w + h != 12 && cbin(d, HC_INTER_PRED_IDC + cu->depth, pu->inter_pred_idc == 2))
      idc = 2;
    else
      idc = cbin(d, HC_INTER_PRED_IDC + 4, pu->inter_pred_idc == 1);
    pu->inter_pred_idc = (uint8_t)idc;
  } else {
    pu->inter_pred_idc = 0;
  }
  for (int l = 0; l < 2; ++l) {
    if (pu->inter_pred_idc == (l ? 0 : 1)) continue;
    int r = 0;
    const int cmax = sh->num_ref_idx[l] - 1;
    if (cmax > 0) { /* truncated unary: two context-coded bins, the rest bypass */
      while (r < cmax) {
        const int more = r < 2 ? cbin(d, HC_REF_IDX + r, pu->ref_idx[l] > r) : cbyp(d, pu->ref_idx[l] > r);
        if (!more) break;
        ++r;
      }
    }
    pu->ref_idx[l] = (int8_t)r;
    if (l == 1 && sh->mvd_l1_zero && pu->inter_pred_idc == 2) {
      pu->mvd[1][0] = pu->mvd[1][1] = 0;
    } else {
      code_mvd(d, pu->mvd[l]);
    }
    pu->mvp_flag[l] = (uint8_t)cbin(d, HC_MVP_FLAG, pu->mvp_flag[l]);
  }
}

/* --------------------------------------------------------------------------------------- coding_unit, 7.3.8.5 */
static int code_part_mode(HevcOracle* d, const HvCu* cu) {
  const HvSps* s = d->cur_sps;
  const int pm = cu->part_mode;
  if (cu->intra) return cbin(d, HC_PART_MODE, pm == HV_PART_2Nx2N) ? HV_PART_2Nx2N : HV_PART_NxN;
  if (cbin(d, HC_PART_MODE, pm == HV_PART_2Nx2N)) return HV_PART_2Nx2N;
  const int horizontal = pm == HV_PART_2NxN || pm == HV_PART_2NxnU || pm == HV_PART_2NxnD;
  if (cu->log2 == s->log2_min_cb) {
    if (cu->log2 == 3) return cbin(d, HC_PART_MODE + 1, pm == HV_PART_2NxN) ? HV_PART_2NxN : HV_PART_Nx2N;
    if (cbin(d, HC_PART_MODE + 1, pm == HV_PART_2NxN)) return HV_PART_2NxN;
    return cbin(d, HC_PART_MODE + 2, pm == HV_PART_Nx2N) ? HV_PART_Nx2N : HV_PART_NxN;
  }
  if (!s->amp_enabled) return cbin(d, HC_PART_MODE + 1, pm == HV_PART_2NxN) ? HV_PART_2NxN : HV_PART_Nx2N;
  const int hor = cbin(d, HC_PART_MODE + 1, horizontal);
  if (cbin(d, HC_PART_MODE + 3, pm == HV_PART_2NxN || pm == HV_PART_Nx2N)) return hor ? HV_PART_2NxN : HV_PART_Nx2N;
  const int second = cbyp(d, pm == HV_PART_2NxnD || pm == HV_PART_nRx2N);
  return hor ? (second ? HV_PART_2NxnD : HV_PART_2NxnU) : (second ? HV_PART_nRx2N : HV_PART_nLx2N);
}

static int pcm_sample(HevcOracle* d, HvCu* cu) {
  const HvSps* s = d->cur_sps;
  const int n = 1 << cu->log2, total = n * n * 3 / 2;
  if (d->enc) {
    HvCabacEnc* e = &d->ce;
    while (e->nbits & 7) ce_write(e, 0); /* pcm_alignment_zero_bit */
    const int16_t* smp = &d->coefs[cu->pcm_samples];
    for (int i = 0; i < total; ++i) {
      const int bits = i < n * n ? s->pcm_bits_luma : s->pcm_bits_chroma;
      for (int b = bits - 1; b >= 0; --b) ce_write(e, (smp[i] >> b) & 1);
    }
    ce_init_engine(e);
    return 0;
  }
  HvCabacDec* c = &d->cd;
  while (c->pos & 7)
    if (cd_bit(c)) return -1; /* pcm_alignment_zero_bit */
  cu->pcm_samples = hv_new_coefs(d, total);
  if (cu->pcm_samples < 0) return -1;
  int16_t* smp = &d->coefs[cu->pcm_samples];
  for (int i = 0; i < total; ++i) {
    const int bits = i < n * n ? s->pcm_bits_luma : s->pcm_bits_chroma;
    int v = 0;
    for (int b = 0; b < bits; ++b) v = v << 1 | (int)cd_bit(c);
    smp[i] = (int16_t)v;
  }
  cd_init_engine(c); /* 9.3.2.5 */
  return c->overrun ? -1 : 0;
}

static int coding_unit(HevcOracle* d, int x0, int y0, int log2, int depth) {
  const HvSps* s = d->cur_sps;
  const HvPps* pps = d->cur_pps;
  const HvSliceHeader* sh = &d->sh;
  HvCu* cu;
  if (d->enc) {
    if (d->walk_cu >= d->num_cu) return hv_fail(d, "encoder plan: ran out of coding units"), -1;
    cu = &d->cus[d->walk_cu++];
    if (cu->x != x0 || cu->y != y0 || cu->log2 != log2) return hv_fail(d, "encoder plan: unit %d is %d,%d/%d, the walk is at %d,%d/%d", d->walk_cu - 1, cu->x, cu->y, cu->log2, x0, y0, log2), -1;
  } else {
    cu = hv_new_cu(d);
    if (!cu) return hv_fail(d, "more coding units than the picture can hold"), -1;
    cu->x = (uint16_t)x0;
    cu->y = (uint16_t)y0;
    cu->log2 = (uint8_t)log2;
    cu->depth = (uint8_t)depth;
  }
  ++d->counters[2];
  if (pps->transquant_bypass_enabled) cu->tq_bypass = (uint8_t)cbin(d, HC_TQ_BYPASS, cu->tq_bypass);
  if (sh->slice_type != HV_SLICE_I) {
    const int l = hv_zscan_available(d, x0, y0, x0 - 1, y0) && d->m_skip[(y0 >> 2) * d->w4 + ((x0 - 1) >> 2)];
    const int a = hv_zscan_available(d, x0, y0, x0, y0 - 1) && d->m_skip[((y0 - 1) >> 2) * d->w4 + (x0 >> 2)];
    cu->skip = (uint8_t)cbin(d, HC_CU_SKIP + l + a, cu->skip);
  }
  if (!d->enc) {
    /* the luma QP the unit has unless it codes a cu_qp_delta of its own */
    cu->qp_y = (int8_t)(pps->cu_qp_delta_enabled ? (hv_qp_y_pred(d, d->qg_x, d->qg_y) + d->cu_qp_delta_val + 52) % 52 : sh->slice_qp);
  }
  if (cu->skip) {
    if (!d->enc) {
      cu->intra = 0;
      cu->part_mode = HV_PART_2Nx2N;
      cu->num_pu = 1;
      hv_set_cu_maps(d, cu);
    }
    prediction_unit(d, cu, 0);
    if (!d->enc) hv_derive_pu_motion(d, cu, 0);
    ++d->counters[8];
  } else {
    if (sh->slice_type != HV_SLICE_I)
      cu->intra = (uint8_t)cbin(d, HC_PRED_MODE, cu->intra);
    else
      cu->intra = 1;
    if (!cu->intra || log2 == s->log2_min_cb) {
      const int pm = code_part_mode(d, cu);
      if (!d->enc) cu->part_mode = (uint8_t)pm;
      if (cu->intra && pm == HV_PART_NxN && log2 <= s->log2_min_tb) return hv_fail(d, "intra NxN in a coding block of the minimum transform size"), -1;
      if (!cu->intra && pm == HV_PART_NxN && log2 == 3) return hv_fail(d, "inter NxN in an 8x8 coding block"), -1;
    } else if (!d->enc) {
      cu->part_mode = HV_PART_2Nx2N;
    }
    if (cu->intra) {
      ++d->counters[7];
      if (cu->part_mode == HV_PART_2Nx2N && s->pcm_enabled && log2 >= s->log2_min_pcm && log2 <= s->log2_max_pcm)
        cu->pcm = (uint8_t)cterm(d, cu->pcm);
      if (!d->enc) hv_set_cu_maps(d, cu);
      if (cu->pcm) {
        if (pcm_sample(d, cu)) return hv_fail(d, "bad pcm_sample()"), -1;
        if (!d->enc) {
          cu->luma_mode[0] = cu->luma_mode[1] = cu->luma_mode[2] = cu->luma_mode[3] = 1;
          cu->chroma_mode = 1;
        }
      } else {
        const int parts = cu->part_mode == HV_PART_NxN ? 4 : 1;
        for (int i = 0; i < parts; ++i) cu->prev_intra_luma_pred_flag[i] = (uint8_t)cbin(d, HC_PREV_INTRA_LUMA, cu->prev_intra_luma_pred_flag[i]);
        for (int i = 0; i < parts; ++i) {
          if (cu->prev_intra_luma_pred_flag[i])
            cu->mpm_idx[i] = (uint8_t)cbyp_tu(d, cu->mpm_idx[i], 2);
          else
            cu->rem_intra_luma_pred_mode[i] = (uint8_t)cbyp_bits(d, cu->rem_intra_luma_pred_mode[i], 5);
        }
        int v = 4;
        if (cbin(d, HC_INTRA_CHROMA, cu->intra_chroma_pred_mode != 4)) v = (int)cbyp_bits(d, cu->intra_chroma_pred_mode, 2);
        cu->intra_chroma_pred_mode = (uint8_t)v;
        if (!d->enc) hv_derive_intra_modes(d, cu);
      }
    } else {
      if (!d->enc) {
        cu->num_pu = (uint8_t)hv_num_parts(cu->part_mode);
        hv_set_cu_maps(d, cu);
      }
      for (int i = 0; i < cu->num_pu; ++i) {
        prediction_unit(d, cu, i);
        if (!d->enc) hv_derive_pu_motion(d, cu, i);
      }
    }
    if (!cu->pcm) {
      int root = 1;
      if (!cu->intra && !(cu->part_mode == HV_PART_2Nx2N && cu->pu[0].merge_flag)) root = cbin(d, HC_RQT_ROOT_CBF, cu->rqt_root_cbf);
      if (!d->enc) cu->rqt_root_cbf = (uint8_t)root;
      if (root) {
        TtWalk w;
        memset(&w, 0, sizeof(w));
        w.d = d;
        w.cu = cu;
        w.max_depth = cu->intra ? s->max_th_depth_intra + (cu->part_mode == HV_PART_NxN) : s->max_th_depth_inter;
        transform_tree(&w, x0, y0, x0, y0, log2, 0, 0, 0, 0);
        if (w.fail) return hv_fail(d, "transform tree of the unit at %d,%d does not parse", x0, y0), -1;
      }
    }
  }
  if (!d->enc) {
    set_cu_qp_map(d, cu);
  }
  d->last_cu_qp = cu->qp_y;
  if (getenv("HV_TRACE"))
    fprintf(stderr, "%s cu %d,%d/%d intra=%d pm=%d skip=%d pcm=%d modes=%d,%d,%d,%d c=%d qp=%d ntu=%d root=%d bits=%zu\n", d->enc ? "E" : "D", x0, y0, log2, cu->intra,
            cu->part_mode, cu->skip, cu->pcm, cu->luma_mode[0], cu->luma_mode[1], cu->luma_mode[2], cu->luma_mode[3], cu->chroma_mode, cu->qp_y, cu->num_tu,
            cu->rqt_root_cbf, d->enc ? d->ce.nbits : d->cd.pos);
  if (getenv("HV_TRACE") && !cu->intra)
    for (int i = 0; i < cu->num_pu; ++i) {
      const HvPu* pu = &cu->pu[i];
      fprintf(stderr, "%s   pu%d mf=%d mi=%d idc=%d ref=%d,%d mvd=%d,%d/%d,%d mvp=%d,%d -> pf=%d mv=%d,%d/%d,%d ri=%d,%d\n", d->enc ? "E" : "D", i, pu->merge_flag, pu->merge_idx,
              pu->inter_pred_idc, pu->ref_idx[0], pu->ref_idx[1], pu->mvd[0][0], pu->mvd[0][1], pu->mvd[1][0], pu->mvd[1][1], pu->mvp_flag[0], pu->mvp_flag[1],
              pu->mv.pred_flag, pu->mv.mv[0][0], pu->mv.mv[0][1], pu->mv.mv[1][0], pu->mv.mv[1][1], pu->mv.ref_idx[0], pu->mv.ref_idx[1]);
    }
  return 0;
}

/* ------------------------------------------------------------------------------------ coding_quadtree, 7.3.8.4 */
static int coding_quadtree(HevcOracle* d, int x0, int y0, int log2, int depth) {
  const HvSps* s = d->cur_sps;
  const HvPps* pps = d->cur_pps;
  const int n = 1 << log2;
  int split;
  if (x0 + n <= s->width && y0 + n <= s->height && log2 > s->log2_min_cb) {
    const int l = hv_zscan_available(d, x0, y0, x0 - 1, y0) && d->m_depth[(y0 >> 2) * d->w4 + ((x0 - 1) >> 2)] > depth;
    const int a = hv_zscan_available(d, x0, y0, x0, y0 - 1) && d->m_depth[((y0 - 1) >> 2) * d->w4 + (x0 >> 2)] > depth;
    split = cbin(d, HC_SPLIT_CU + l + a, d->enc ? d->m_depth[(y0 >> 2) * d->w4 + (x0 >> 2)] > depth : 0);
  } else {
    split = log2 > s->log2_min_cb;
  }
  if (pps->cu_qp_delta_enabled && log2 >= s->log2_ctb - pps->diff_cu_qp_delta_depth) {
    d->is_cu_qp_delta_coded = 0;
    d->cu_qp_delta_val = 0;
    d->qg_x = x0;
    d->qg_y = y0;
    d->qp_y_prev = d->last_cu_qp;
  }
  if (split) {
    const int h = n >> 1;
    for (int i = 0; i < 4; ++i) {
      const int x = x0 + (i & 1) * h, y = y0 + (i >> 1) * h;
      if (x < s->width && y < s->height)
        if (coding_quadtree(d, x, y, log2 - 1, depth + 1)) return -1;
    }
    return 0;
  }
  return coding_unit(d, x0, y0, log2, depth);
}

/* ------------------------------------------------------------------------------------------------ sao, 7.3.8.3 */
static void sao_syntax(HevcOracle* d, int rx, int ry) {
  const HvSps* s = d->cur_sps;
  const HvSliceHeader* sh = &d->sh;
  const int addr = ry * s->width_ctb + rx;
  HvSao* p = &d->sao[addr];
  int merge_left = 0, merge_up = 0;
  if (rx > 0 && addr - 1 >= sh->slice_addr_rs && d->m_slice[addr - 1] != 0xffff &&
      d->slice_facts[d->m_slice[addr - 1]].slice_addr_rs == sh->slice_addr_rs)
    merge_left = cbin(d, HC_SAO_MERGE, p->merge_left);
  if (ry > 0 && !merge_left && addr - s->width_ctb >= sh->slice_addr_rs && d->m_slice[addr - s->width_ctb] != 0xffff &&
      d->slice_facts[d->m_slice[addr - s->width_ctb]].slice_addr_rs == sh->slice_addr_rs)
    merge_up = cbin(d, HC_SAO_MERGE, p->merge_up);
  if (!d->enc) {
    memset(p, 0, sizeof(*p));
    p->merge_left = (uint8_t)merge_left;
    p->merge_up = (uint8_t)merge_up;
    if (merge_left || merge_up) {
      const HvSao* q = &d->sao[merge_left ? addr - 1 : addr - s->width_ctb];
      memcpy(p->type, q->type, sizeof(p->type));
      memcpy(p->band_pos, q->band_pos, sizeof(p->band_pos));
      memcpy(p->eo_class, q->eo_class, sizeof(p->eo_class));
      memcpy(p->offset, q->offset, sizeof(p->offset));
    }
  }
  if (merge_left || merge_up) return;
  for (int c = 0; c < 3; ++c) {
    if (!(c == 0 ? sh->sao_luma : sh->sao_chroma)) {
      if (!d->enc) p->type[c] = 0;
      continue;
    }
    if (c != 2) {
      int t = 0;
      if (cbin(d, HC_SAO_TYPE, p->type[c] != 0)) t = cbyp(d, p->type[c] == 2) ? 2 : 1;
      p->type[c] = (uint8_t)t;
    } else {
      p->type[2] = p->type[1];
    }
    if (!p->type[c]) continue;
    int a[4];
    for (int i = 0; i < 4; ++i) a[i] = cbyp_tu(d, hv_abs(p->offset[c][i]), 7);
    if (p->type[c] == 1) {
      for (int i = 0; i < 4; ++i) {
        int neg = 0;
        if (a[i]) neg = cbyp(d, p->offset[c][i] < 0);
        if (!d->enc) p->offset[c][i] = (int8_t)(neg ? -a[i] : a[i]);
      }
      p->band_pos[c] = (uint8_t)cbyp_bits(d, p->band_pos[c], 5);
    } else {
      if (c != 2)
        p->eo_class[c] = (uint8_t)cbyp_bits(d, p->eo_class[c], 2);
      else
        p->eo_class[2] = p->eo_class[1];
      if (!d->enc)
        for (int i = 0; i < 4; ++i) p->offset[c][i] = (int8_t)(i < 2 ? a[i] : -a[i]);
    }
  }
}

/* -------------------------------------------------------------------------- slice_segment_data, 7.3.8.1 */
int hv_slice_data(HevcOracle* d) {
  const HvSps* s = d->cur_sps;
  const HvPps* pps = d->cur_pps;
  const HvSliceHeader* sh = &d->sh;
  const int total = s->width_ctb * s->height_ctb;
  hv_cabac_init_contexts(d);
  d->last_cu_qp = sh->slice_qp;
  d->qp_y_prev = sh->slice_qp;
  d->is_cu_qp_delta_coded = 0;
  d->cu_qp_delta_val = 0;
  int addr = sh->segment_address;
  size_t wpp_sub_start = d->enc ? 0 : d->sh.data_byte_offset;  /* dec: where the current wavefront substream began */
  int wpp_zeros = 0, wpp_sub = 0;
  for (;;) {
    const int rx = addr % s->width_ctb, ry = addr / s->width_ctb;
    d->ctb_addr_rs = addr;
    if (pps->entropy_coding_sync && rx == 0 && addr != sh->segment_address) {
      /* 9.3.1: a new row starts from what the row above had learnt after its second block - if that block is there
       * and belongs to this slice - else from the tables */
      const int above_right = addr - s->width_ctb + 1;
      if (s->width_ctb > 1 && above_right >= sh->slice_addr_rs && d->m_slice[above_right] != 0xffff)
        memcpy(d->ctx, d->ctx_wpp, sizeof(d->ctx));
      else
        hv_cabac_init_contexts(d);
      d->last_cu_qp = sh->slice_qp; /* (8.6.1: the first quantisation group of a row predicts from SliceQpY) */
    }
    d->m_slice[addr] = (uint16_t)d->slice_idx;
    if (sh->sao_luma || sh->sao_chroma) sao_syntax(d, rx, ry);
    if (coding_quadtree(d, rx << s->log2_ctb, ry << s->log2_ctb, s->log2_ctb, 0)) return -1;
    if (pps->entropy_coding_sync && rx == (s->width_ctb > 1 ? 1 : 0)) memcpy(d->ctx_wpp, d->ctx, sizeof(d->ctx));
    if (!d->enc && d->cd.overrun) return hv_fail(d, "slice data ran past the end of the NAL unit at coding tree block %d", addr), -1;
    const int last = cterm(d, d->enc ? addr + 1 == d->enc_end_ctb : 0);
    ++addr;
    if (last) break;
    if (addr >= total) return hv_fail(d, "slice data goes on behind the last coding tree block"), -1;
    if (pps->entropy_coding_sync && addr % s->width_ctb == 0) {
      /* end_of_subset_one_bit + byte_alignment(): the substream of the next row starts on a byte boundary */
      if (!cterm(d, 1)) return hv_fail(d, "end_of_subset_one_bit is not set at the end of row %d", ry), -1;
      if (d->enc) {
        while (d->ce.nbits & 7) ce_write(&d->ce, 0);
        if (d->enc_nsub < 256) d->enc_sub_start[d->enc_nsub++] = d->ce.nbits / 8;
        ce_init_engine(&d->ce);
      } else {
        while (d->cd.pos & 7)
          if (cd_bit(&d->cd)) return hv_fail(d, "wavefront substream does not end on zero alignment bits"), -1;
        /* a slice header that lists entry points must list THIS one: the substream just ended is entry_point_offset
           bytes long in the NAL unit - the bytes here plus the emulation prevention bytes that were taken out of it
           (7.4.7.1; the zero count starts afresh behind the header's alignment bit, and runs on across substreams) */
        if (d->sh.num_entry_points > 0) {
          const size_t end = d->cd.pos >> 3;
          uint32_t size = 0;
          for (size_t i = wpp_sub_start; i < end; ++i) {
            if (wpp_zeros >= 2 && d->cd.p[i] <= 3) { ++size; wpp_zeros = 0; }
            ++size;
            wpp_zeros = d->cd.p[i] == 0 ? wpp_zeros + 1 : 0;
          }
          if (wpp_sub >= d->sh.num_entry_points) return hv_fail(d, "more wavefront rows than the slice header's %d entry points", d->sh.num_entry_points), -1;
          if (wpp_sub < 256 && d->sh.entry_point[wpp_sub] != size)
            return hv_fail(d, "entry point %d says %u bytes, the substream is %u", wpp_sub, d->sh.entry_point[wpp_sub], size), -1;
          ++wpp_sub;
          wpp_sub_start = end;
        }
        cd_init_engine(&d->cd);
      }
    }
  }
  if (!d->enc) {
    /* the desync check: the next bit is rbsp_stop_one_bit, zeros fill its byte, nothing but cabac_zero_words follows */
    HvCabacDec* c = &d->cd;
    if (getenv("HV_TRACE")) {
      fprintf(stderr, "D end: pos %zu of %zu bits; next:", c->pos, c->size * 8);
      for (size_t i = c->pos; i < c->pos + 24 && i < c->size * 8; ++i) fprintf(stderr, "%d", (c->p[i >> 3] >> (7 - (i & 7))) & 1);
      fprintf(stderr, "\n");
    }
    int clean = c->pos >= 1 && ((c->p[(c->pos - 1) >> 3] >> (7 - ((c->pos - 1) & 7))) & 1) == 1; /* the stop bit, already consumed */
    while (c->pos & 7) clean &= cd_bit(c) == 0;
    for (size_t i = c->pos >> 3; i < c->size; ++i) clean &= c->p[i] == 0;
    if (clean && !c->overrun) ++d->counters[1];
  }
  return 0;
}
