/* TEST INFRASTRUCTURE - HEVC oracle: NAL units, parameter sets (7.3.2), slice segment headers (7.3.6), picture order
 * count and reference picture sets (8.3.1, 8.3.2), reference picture lists (8.3.4), the output order of C.5.2, the
 * decoded picture hash SEI (D.2.19 / D.3.19) and the public entry points of hevc_oracle.h.  See hevc_oracle.h. */
#include <stdarg.h>
#include <stdio.h>

#include "hevc_internal.h"

void hv_fail(HevcOracle* d, const char* fmt, ...) {
  if (d->error[0]) return; /* the first error stays */
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(d->error, sizeof(d->error), fmt, ap);
  va_end(ap);
}

/* 7.4.2: emulation prevention bytes out; the two-byte NAL unit header is NOT part of `nal` here */
size_t hv_unescape(const uint8_t* nal, size_t n, uint8_t* out) {
  size_t o = 0;
  int zeros = 0;
  for (size_t i = 0; i < n; ++i) {
    if (zeros >= 2 && nal[i] == 3) {
      zeros = 0;
      continue;
    }
    out[o++] = nal[i];
    zeros = nal[i] == 0 ? zeros + 1 : 0;
  }
  return o;
}

/* ------------------------------------------------------------------------------------------------ MD5 (RFC 1321) */
static const uint32_t md5_k[64] = {
    0xd76aa478, 0xe8c7b756, 0x242070db, 0xc1bdceee, 0xf57c0faf, 0x4787c62a, 0xa8304613, 0xfd469501, 0x698098d8, 0x8b44f7af,
    0xffff5bb1, 0x895cd7be, 0x6b901122, 0xfd987193, 0xa679438e, 0x49b40821, 0xf61e2562, 0xc040b340, 0x265e5a51, 0xe9b6c7aa,
    0xd62f105d, 0x02441453, 0xd8a1e681, 0xe7d3fbc8, 0x21e1cde6, 0xc33707d6, 0xf4d50d87, 0x455a14ed, 0xa9e3e905, 0xfcefa3f8,
    0x676f02d9, 0x8d2a4c8a, 0xfffa3942, 0x8771f681, 0x6d9d6122, 0xfde5380c, 0xa4beea44, 0x4bdecfa9, 0xf6bb4b60, 0xbebfbc70,
    0x289b7ec6, 0xeaa127fa, 0xd4ef3085, 0x04881d05, 0xd9d4d039, 0xe6db99e5, 0x1fa27cf8, 0xc4ac5665, 0xf4292244, 0x432aff97,
    0xab9423a7, 0xfc93a039, 0x655b59c3, 0x8f0ccc92, 0xffeff47d, 0x85845dd1, 0x6fa87e4f, 0xfe2ce6e0, 0xa3014314, 0x4e0811a1,
    0xf7537e82, 0xbd3af235, 0x2ad7d2bb, 0xeb86d391};
static const uint8_t md5_s[64] = {7, 12, 17, 22, 7, 12, 17, 22, 7, 12, 17, 22, 7, 12, 17, 22, 5, 9,  14, 20, 5, 9,
                                  14, 20, 5, 9,  14, 20, 5, 9,  14, 20, 4, 11, 16, 23, 4, 11, 16, 23, 4, 11, 16, 23,
                                  4, 11, 16, 23, 6, 10, 15, 21, 6, 10, 15, 21, 6, 10, 15, 21, 6, 10, 15, 21};
static void md5_block(uint32_t st[4], const uint8_t* p) {
  uint32_t m[16], a = st[0], b = st[1], c = st[2], dd = st[3];
  for (int i = 0; i < 16; ++i) m[i] = (uint32_t)p[4 * i] | (uint32_t)p[4 * i + 1] << 8 | (uint32_t)p[4 * i + 2] << 16 | (uint32_t)p[4 * i + 3] << 24;
  for (int i = 0; i < 64; ++i) {
    uint32_t f;
    int g;
    if (i < 16) {
      f = (b & c) | (~b & dd);
      g = i;
    } else if (i < 32) {
      f = (dd & b) | (~dd & c);
      g = (5 * i + 1) & 15;
    } else if (i < 48) {
      f = b ^ c ^ dd;
      g = (3 * i + 5) & 15;
    } else {
      f = c ^ (b | ~dd);
      g = (7 * i) & 15;
    }
    f += a + md5_k[i] + m[g];
    a = dd;
    dd = c;
    c = b;
    b += f << md5_s[i] | f >> (32 - md5_s[i]);
  }
  st[0] += a;
  st[1] += b;
  st[2] += c;
  st[3] += dd;
}
void hv_md5(const uint8_t* data, size_t n, uint8_t out[16]) {
  uint32_t st[4] = {0x67452301, 0xefcdab89, 0x98badcfe, 0x10325476};
  size_t i = 0;
  for (; i + 64 <= n; i += 64) md5_block(st, data + i);
  uint8_t tail[128];
  size_t r = n - i;
  memcpy(tail, data + i, r);
  tail[r++] = 0x80;
  size_t padded = r <= 56 ? 64 : 128;
  memset(tail + r, 0, padded - r);
  uint64_t bits = (uint64_t)n * 8;
  for (int k = 0; k < 8; ++k) tail[padded - 8 + k] = (uint8_t)(bits >> (8 * k));
  md5_block(st, tail);
  if (padded == 128) md5_block(st, tail + 64);
  for (int k = 0; k < 4; ++k)
    for (int j = 0; j < 4; ++j) out[4 * k + j] = (uint8_t)(st[k] >> (8 * j));
}
/* D.3.19: per colour component over the decoded (not cropped) picture, one byte per 8-bit sample, raster order */
void hv_picture_md5(const HevcOracle* d, const HvPic* p, uint8_t out[3][16]) {
  const HvSps* s = d->cur_sps;
  hv_md5(p->plane[0], (size_t)s->width * s->height, out[0]);
  hv_md5(p->plane[1], (size_t)(s->width / 2) * (s->height / 2), out[1]);
  hv_md5(p->plane[2], (size_t)(s->width / 2) * (s->height / 2), out[2]);
}

/* ------------------------------------------------------------------------------------------- scaling lists 7.3.4 */
static void scaling_expand(HvScaling* s, int size_id, int m, const uint8_t* list, int dc) {
  /* 7.4.5: the list is in up-right diagonal order; 16x16 and 32x32 replicate an 8x8 list and replace the DC */
  if (size_id == 0) {
    const uint8_t* sc = hv_diag_scan(2);
    for (int i = 0; i < 16; ++i) s->f4[m][(sc[i] >> 4) * 4 + (sc[i] & 15)] = list[i];
    return;
  }
  const uint8_t* sc = hv_diag_scan(3);
  uint8_t m8[64];
  for (int i = 0; i < 64; ++i) m8[(sc[i] >> 4) * 8 + (sc[i] & 15)] = list[i];
  if (size_id == 1) {
    memcpy(s->f8[m], m8, 64);
  } else if (size_id == 2) {
    for (int y = 0; y < 16; ++y)
      for (int x = 0; x < 16; ++x) s->f16[m][y * 16 + x] = m8[(y >> 1) * 8 + (x >> 1)];
    s->f16[m][0] = (uint8_t)dc;
  } else {
    for (int y = 0; y < 32; ++y)
      for (int x = 0; x < 32; ++x) s->f32[m][y * 32 + x] = m8[(y >> 2) * 8 + (x >> 2)];
    s->f32[m][0] = (uint8_t)dc;
  }
}
static void scaling_default_list(int size_id, int m, uint8_t* list) {
  if (size_id == 0) {
    memset(list, 16, 16);
    return;
  }
  const int inter = size_id == 3 ? m == 1 : m >= 3;
  memcpy(list, inter ? hevc_default_scaling_inter8 : hevc_default_scaling_intra8, 64);
}
void hv_scaling_default(HvScaling* s) {
  uint8_t list[64];
  for (int size_id = 0; size_id < 4; ++size_id)
    for (int m = 0; m < (size_id == 3 ? 2 : 6); ++m) {
      scaling_default_list(size_id, m, list);
      scaling_expand(s, size_id, m, list, 16);
    }
}
int hv_parse_scaling_list(HvBits* b, HvScaling* s) {
  static uint8_t lists[4][6][64]; /* (parsing is single-threaded per process in the tests; kept off the stack) */
  uint8_t (*L)[6][64] = lists;
  int dc[4][6];
  for (int size_id = 0; size_id < 4; ++size_id)
    for (int m = 0; m < (size_id == 3 ? 2 : 6); ++m) {
      const int n = hv_min(64, 1 << (4 + (size_id << 1)));
      dc[size_id][m] = 16;
      if (!hvb_u(b, 1)) { /* scaling_list_pred_mode_flag = 0: copy */
        const int delta = (int)hvb_ue(b);
        if (delta == 0) {
          scaling_default_list(size_id, m, L[size_id][m]);
        } else {
          if (delta > m) return -1;
          memcpy(L[size_id][m], L[size_id][m - delta], (size_t)n);
          dc[size_id][m] = dc[size_id][m - delta];
        }
      } else {
        int next = 8;
        if (size_id > 1) {
          next = hvb_se(b) + 8;
          dc[size_id][m] = next;
        }
        for (int i = 0; i < n; ++i) {
          next = (next + hvb_se(b) + 256) % 256;
          L[size_id][m][i] = (uint8_t)next;
        }
      }
      scaling_expand(s, size_id, m, L[size_id][m], dc[size_id][m]);
    }
  return b->overrun ? -1 : 0;
}

/* ------------------------------------------------------------------------------------ st_ref_pic_set 7.3.7 / 7.4.8 */
int hv_parse_st_rps(HvBits* b, HvRps* out, int idx, int num_sps_sets, const HvRps* sets) {
  memset(out, 0, sizeof(*out));
  int inter = 0;
  if (idx != 0) inter = (int)hvb_u(b, 1);
  if (inter) {
    int delta_idx = 1;
    if (idx == num_sps_sets) delta_idx = (int)hvb_ue(b) + 1;
    if (delta_idx > idx) return -1;
    const HvRps* r = &sets[idx - delta_idx];
    const int sign = (int)hvb_u(b, 1);
    const int delta_rps = (1 - 2 * sign) * ((int)hvb_ue(b) + 1);
    const int n_ref = r->num_negative + r->num_positive;
    uint8_t used[33], use_delta[33];
    for (int j = 0; j <= n_ref; ++j) {
      used[j] = (uint8_t)hvb_u(b, 1);
      use_delta[j] = 1;
      if (!used[j]) use_delta[j] = (uint8_t)hvb_u(b, 1);
    }
    /* (7-61) negative pictures of the new set */
    int i = 0;
    for (int j = r->num_positive - 1; j >= 0; --j) {
      const int dp = r->delta_poc_s1[j] + delta_rps;
      if (dp < 0 && use_delta[r->num_negative + j]) {
        if (i >= 16) return -1;
        out->delta_poc_s0[i] = dp;
        out->used_s0[i++] = used[r->num_negative + j];
      }
    }
    if (delta_rps < 0 && use_delta[n_ref]) {
      if (i >= 16) return -1;
      out->delta_poc_s0[i] = delta_rps;
      out->used_s0[i++] = used[n_ref];
    }
    for (int j = 0; j < r->num_negative; ++j) {
      const int dp = r->delta_poc_s0[j] + delta_rps;
      if (dp < 0 && use_delta[j]) {
        if (i >= 16) return -1;
        out->delta_poc_s0[i] = dp;
        out->used_s0[i++] = used[j];
      }
    }
    out->num_negative = i;
    /* (7-62) positive pictures */
    i = 0;
    for (int j = r->num_negative - 1; j >= 0; --j) {
      const int dp = r->delta_poc_s0[j] + delta_rps;
      if (dp > 0 && use_delta[j]) {
        if (i >= 16) return -1;
        out->delta_poc_s1[i] = dp;
        out->used_s1[i++] = used[j];
      }
    }
    if (delta_rps > 0 && use_delta[n_ref]) {
      if (i >= 16) return -1;
      out->delta_poc_s1[i] = delta_rps;
      out->used_s1[i++] = used[n_ref];
    }
    for (int j = 0; j < r->num_positive; ++j) {
      const int dp = r->delta_poc_s1[j] + delta_rps;
      if (dp > 0 && use_delta[r->num_negative + j]) {
        if (i >= 16) return -1;
        out->delta_poc_s1[i] = dp;
        out->used_s1[i++] = used[r->num_negative + j];
      }
    }
    out->num_positive = i;
  } else {
    out->num_negative = (int)hvb_ue(b);
    out->num_positive = (int)hvb_ue(b);
    if (out->num_negative > 16 || out->num_positive > 16) return -1;
    int poc = 0;
    for (int i = 0; i < out->num_negative; ++i) {
      poc -= (int)hvb_ue(b) + 1;
      out->delta_poc_s0[i] = poc;
      out->used_s0[i] = (uint8_t)hvb_u(b, 1);
    }
    poc = 0;
    for (int i = 0; i < out->num_positive; ++i) {
      poc += (int)hvb_ue(b) + 1;
      out->delta_poc_s1[i] = poc;
      out->used_s1[i] = (uint8_t)hvb_u(b, 1);
    }
  }
  return b->overrun ? -1 : 0;
}

/* ---------------------------------------------------------------------------------- profile_tier_level 7.3.3 */
static void skip_profile_tier_level(HvBits* b, int max_sub_layers_minus1) {
  hvb_u(b, 8);  /* profile space, tier, profile idc */
  hvb_u(b, 32); /* compatibility flags */
  hvb_u(b, 32); /* progressive, interlaced, non-packed, frame-only, 28 of the 43 reserved bits */
  hvb_u(b, 16); /* the other 15, general_inbld / reserved */
  hvb_u(b, 8);  /* general_level_idc */
  uint8_t prof[8], lev[8];
  for (int i = 0; i < max_sub_layers_minus1; ++i) {
    prof[i] = (uint8_t)hvb_u(b, 1);
    lev[i] = (uint8_t)hvb_u(b, 1);
  }
  if (max_sub_layers_minus1 > 0)
    for (int i = max_sub_layers_minus1; i < 8; ++i) hvb_u(b, 2);
  for (int i = 0; i < max_sub_layers_minus1; ++i) {
    if (prof[i]) {
      hvb_u(b, 32);
      hvb_u(b, 32);
      hvb_u(b, 24);
    }
    if (lev[i]) hvb_u(b, 8);
  }
}

/* ------------------------------------------------------------------------------------------------- SPS 7.3.2.2 */
int hv_parse_sps(HevcOracle* d, HvBits* b) {
  HvSps s;
  memset(&s, 0, sizeof(s));
  s.vps_id = (int)hvb_u(b, 4);
  const int msl = (int)hvb_u(b, 3);
  s.max_sub_layers = msl + 1;
  hvb_u(b, 1); /* sps_temporal_id_nesting_flag */
  skip_profile_tier_level(b, msl);
  s.id = (int)hvb_ue(b);
  if (s.id > 15) return hv_fail(d, "SPS id %d out of range", s.id), -1;
  s.chroma_format_idc = (int)hvb_ue(b);
  if (s.chroma_format_idc == 3) hvb_u(b, 1);
  s.width = (int)hvb_ue(b);
  s.height = (int)hvb_ue(b);
  if (hvb_u(b, 1)) { /* conformance window, in chroma sample units */
    s.conf_left = 2 * (int)hvb_ue(b);
    s.conf_right = 2 * (int)hvb_ue(b);
    s.conf_top = 2 * (int)hvb_ue(b);
    s.conf_bottom = 2 * (int)hvb_ue(b);
  }
  s.bit_depth_luma = 8 + (int)hvb_ue(b);
  s.bit_depth_chroma = 8 + (int)hvb_ue(b);
  s.log2_max_poc_lsb = 4 + (int)hvb_ue(b);
  const int ordering = (int)hvb_u(b, 1);
  for (int i = ordering ? 0 : msl; i <= msl; ++i) {
    s.max_dec_pic_buffering = (int)hvb_ue(b) + 1;
    s.max_num_reorder = (int)hvb_ue(b);
    s.max_latency_increase_plus1 = (int)hvb_ue(b);
  }
  s.log2_min_cb = 3 + (int)hvb_ue(b);
  s.log2_ctb = s.log2_min_cb + (int)hvb_ue(b);
  s.log2_min_tb = 2 + (int)hvb_ue(b);
  s.log2_max_tb = s.log2_min_tb + (int)hvb_ue(b);
  s.max_th_depth_inter = (int)hvb_ue(b);
  s.max_th_depth_intra = (int)hvb_ue(b);
  s.scaling_list_enabled = (int)hvb_u(b, 1);
  hv_scaling_default(&s.scaling);
  if (s.scaling_list_enabled) {
    s.scaling_list_present = (int)hvb_u(b, 1);
    if (s.scaling_list_present && hv_parse_scaling_list(b, &s.scaling)) return hv_fail(d, "SPS: bad scaling_list_data"), -1;
  }
  s.amp_enabled = (int)hvb_u(b, 1);
  s.sao_enabled = (int)hvb_u(b, 1);
  s.pcm_enabled = (int)hvb_u(b, 1);
  if (s.pcm_enabled) {
    s.pcm_bits_luma = 1 + (int)hvb_u(b, 4);
    s.pcm_bits_chroma = 1 + (int)hvb_u(b, 4);
    s.log2_min_pcm = 3 + (int)hvb_ue(b);
    s.log2_max_pcm = s.log2_min_pcm + (int)hvb_ue(b);
    s.pcm_loop_filter_disabled = (int)hvb_u(b, 1);
  }
  s.num_st_rps = (int)hvb_ue(b);
  if (s.num_st_rps > 64) return hv_fail(d, "SPS: %d short-term reference picture sets", s.num_st_rps), -1;
  for (int i = 0; i < s.num_st_rps; ++i)
    if (hv_parse_st_rps(b, &s.st_rps[i], i, s.num_st_rps, s.st_rps)) return hv_fail(d, "SPS: bad st_ref_pic_set %d", i), -1;
  s.long_term_present = (int)hvb_u(b, 1);
  if (s.long_term_present) {
    s.num_lt_sps = (int)hvb_ue(b);
    if (s.num_lt_sps > 32) return hv_fail(d, "SPS: %d long-term pictures", s.num_lt_sps), -1;
    for (int i = 0; i < s.num_lt_sps; ++i) {
      s.lt_poc_lsb_sps[i] = (int)hvb_u(b, s.log2_max_poc_lsb);
      s.lt_used_sps[i] = (int)hvb_u(b, 1);
    }
  }
  s.temporal_mvp_enabled = (int)hvb_u(b, 1);
  s.strong_intra_smoothing = (int)hvb_u(b, 1);
  /* vui_parameters_present_flag and everything behind it: nothing a decoder of the samples needs */
  if (b->overrun) return hv_fail(d, "SPS: ran out of data"), -1;
  if (s.chroma_format_idc != 1) return hv_fail(d, "HEVC chroma_format_idc %d is not supported (4:2:0 only)", s.chroma_format_idc), -1;
  if (s.bit_depth_luma != 8 || s.bit_depth_chroma != 8)
    return hv_fail(d, "HEVC bit depth %d / %d is not supported (8 bit only)", s.bit_depth_luma, s.bit_depth_chroma), -1;
  if (s.log2_ctb > 6 || s.log2_ctb < 4 || s.log2_max_tb > 5 || s.log2_min_tb >= s.log2_min_cb || s.log2_max_tb > s.log2_ctb)
    return hv_fail(d, "SPS: block sizes out of range (CTB %d, min CB %d, TB %d..%d)", s.log2_ctb, s.log2_min_cb, s.log2_min_tb, s.log2_max_tb), -1;
  if (s.width <= 0 || s.height <= 0 || s.width > 8192 || s.height > 8192 || (s.width & ((1 << s.log2_min_cb) - 1)) ||
      (s.height & ((1 << s.log2_min_cb) - 1)))
    return hv_fail(d, "SPS: picture size %dx%d", s.width, s.height), -1;
  s.ctb_size = 1 << s.log2_ctb;
  s.width_ctb = (s.width + s.ctb_size - 1) >> s.log2_ctb;
  s.height_ctb = (s.height + s.ctb_size - 1) >> s.log2_ctb;
  s.width_min_cb = s.width >> s.log2_min_cb;
  s.height_min_cb = s.height >> s.log2_min_cb;
  s.valid = 1;
  d->sps[s.id] = s;
  return 0;
}

/* ------------------------------------------------------------------------------------------------- PPS 7.3.2.3 */
int hv_parse_pps(HevcOracle* d, HvBits* b) {
  HvPps p;
  memset(&p, 0, sizeof(p));
  p.id = (int)hvb_ue(b);
  p.sps_id = (int)hvb_ue(b);
  if (p.id > 63 || p.sps_id > 15) return hv_fail(d, "PPS id %d / SPS id %d out of range", p.id, p.sps_id), -1;
  p.dependent_slices_enabled = (int)hvb_u(b, 1);
  p.output_flag_present = (int)hvb_u(b, 1);
  p.num_extra_slice_header_bits = (int)hvb_u(b, 3);
  p.sign_data_hiding = (int)hvb_u(b, 1);
  p.cabac_init_present = (int)hvb_u(b, 1);
  p.num_ref_idx_default[0] = 1 + (int)hvb_ue(b);
  p.num_ref_idx_default[1] = 1 + (int)hvb_ue(b);
  p.init_qp = 26 + hvb_se(b);
  p.constrained_intra_pred = (int)hvb_u(b, 1);
  p.transform_skip_enabled = (int)hvb_u(b, 1);
  p.cu_qp_delta_enabled = (int)hvb_u(b, 1);
  if (p.cu_qp_delta_enabled) p.diff_cu_qp_delta_depth = (int)hvb_ue(b);
  p.cb_qp_offset = hvb_se(b);
  p.cr_qp_offset = hvb_se(b);
  p.slice_chroma_qp_offsets_present = (int)hvb_u(b, 1);
  p.weighted_pred = (int)hvb_u(b, 1);
  p.weighted_bipred = (int)hvb_u(b, 1);
  p.transquant_bypass_enabled = (int)hvb_u(b, 1);
  p.tiles_enabled = (int)hvb_u(b, 1);
  p.entropy_coding_sync = (int)hvb_u(b, 1);
  if (p.tiles_enabled) return hv_fail(d, "HEVC tiles are not supported"), -1;
  p.loop_filter_across_slices = (int)hvb_u(b, 1);
  p.deblocking_control_present = (int)hvb_u(b, 1);
  if (p.deblocking_control_present) {
    p.deblocking_override_enabled = (int)hvb_u(b, 1);
    p.deblocking_disabled = (int)hvb_u(b, 1);
    if (!p.deblocking_disabled) {
      p.beta_offset_div2 = hvb_se(b);
      p.tc_offset_div2 = hvb_se(b);
    }
  }
  p.scaling_list_present = (int)hvb_u(b, 1);
  hv_scaling_default(&p.scaling);
  if (p.scaling_list_present && hv_parse_scaling_list(b, &p.scaling)) return hv_fail(d, "PPS: bad scaling_list_data"), -1;
  p.lists_modification_present = (int)hvb_u(b, 1);
  p.log2_parallel_merge_level = 2 + (int)hvb_ue(b);
  p.slice_header_extension_present = (int)hvb_u(b, 1);
  if (b->overrun) return hv_fail(d, "PPS: ran out of data"), -1;
  p.valid = 1;
  d->pps[p.id] = p;
  return 0;
}

/* ----------------------------------------------------------------------------- slice segment header 7.3.6.1 */
static int parse_pred_weight_table(HvBits* b, HvSliceHeader* sh) {
  sh->luma_log2_wd = (int)hvb_ue(b);
  sh->chroma_log2_wd = sh->luma_log2_wd + hvb_se(b);
  if (sh->luma_log2_wd > 7 || sh->chroma_log2_wd < 0 || sh->chroma_log2_wd > 7) return -1;
  for (int l = 0; l < (sh->slice_type == HV_SLICE_B ? 2 : 1); ++l) {
    uint8_t lf[16], cf[16];
    for (int i = 0; i < sh->num_ref_idx[l]; ++i) lf[i] = (uint8_t)hvb_u(b, 1);
    for (int i = 0; i < sh->num_ref_idx[l]; ++i) cf[i] = (uint8_t)hvb_u(b, 1);
    for (int i = 0; i < sh->num_ref_idx[l]; ++i) {
      sh->luma_weight[l][i] = (int16_t)(1 << sh->luma_log2_wd);
      sh->luma_offset[l][i] = 0;
      if (lf[i]) {
        sh->luma_weight[l][i] = (int16_t)(sh->luma_weight[l][i] + hvb_se(b));
        sh->luma_offset[l][i] = (int16_t)hvb_se(b);
      }
      for (int j = 0; j < 2; ++j) {
        sh->chroma_weight[l][i][j] = (int16_t)(1 << sh->chroma_log2_wd);
        sh->chroma_offset[l][i][j] = 0;
      }
      if (cf[i])
        for (int j = 0; j < 2; ++j) {
          const int w = (1 << sh->chroma_log2_wd) + hvb_se(b);
          const int doff = hvb_se(b);
          sh->chroma_weight[l][i][j] = (int16_t)w;
          /* (7-56): the offset is coded relative to what the weight does to mid-grey */
          sh->chroma_offset[l][i][j] = (int16_t)hv_clip3(-128, 127, 128 + doff - ((128 * w) >> sh->chroma_log2_wd));
        }
    }
  }
  return 0;
}

int hv_parse_slice_header(HevcOracle* d, HvBits* b, int nal_type, HvSliceHeader* sh) {
  memset(sh, 0, sizeof(*sh));
  sh->first_slice_segment_in_pic = (int)hvb_u(b, 1);
  if (nal_type >= 16 && nal_type <= 23) sh->no_output_of_prior_pics = (int)hvb_u(b, 1);
  sh->pps_id = (int)hvb_ue(b);
  if (sh->pps_id > 63 || !d->pps[sh->pps_id].valid) return hv_fail(d, "slice refers to picture parameter set %d, which was never sent", sh->pps_id), -1;
  const HvPps* pps = &d->pps[sh->pps_id];
  if (!d->sps[pps->sps_id].valid) return hv_fail(d, "slice refers to sequence parameter set %d, which was never sent", pps->sps_id), -1;
  const HvSps* sps = &d->sps[pps->sps_id];
  d->cur_pps = pps;
  d->cur_sps = sps;
  if (!sh->first_slice_segment_in_pic) {
    if (pps->dependent_slices_enabled) sh->dependent = (int)hvb_u(b, 1);
    sh->segment_address = (int)hvb_u(b, hv_ceil_log2((unsigned)(sps->width_ctb * sps->height_ctb)));
    if (sh->segment_address >= sps->width_ctb * sps->height_ctb) return hv_fail(d, "slice_segment_address %d outside the picture", sh->segment_address), -1;
  }
  if (sh->dependent) return hv_fail(d, "HEVC dependent slice segments are not supported"), -1;
  for (int i = 0; i < pps->num_extra_slice_header_bits; ++i) hvb_u(b, 1);
  sh->slice_type = (int)hvb_ue(b);
  if (sh->slice_type > 2) return hv_fail(d, "slice_type %d", sh->slice_type), -1;
  sh->pic_output_flag = 1;
  if (pps->output_flag_present) sh->pic_output_flag = (int)hvb_u(b, 1);
  const int idr = nal_type == 19 || nal_type == 20;
  sh->rps = NULL;
  if (!idr) {
    sh->poc_lsb = (int)hvb_u(b, sps->log2_max_poc_lsb);
    sh->st_rps_sps_flag = (int)hvb_u(b, 1);
    if (!sh->st_rps_sps_flag) {
      if (hv_parse_st_rps(b, &sh->st_rps_local, sps->num_st_rps, sps->num_st_rps, sps->st_rps)) return hv_fail(d, "slice header: bad st_ref_pic_set"), -1;
      sh->rps = &sh->st_rps_local;
    } else {
      if (sps->num_st_rps == 0) return hv_fail(d, "slice header selects a reference picture set of an SPS that has none"), -1;
      if (sps->num_st_rps > 1) sh->st_rps_idx = (int)hvb_u(b, hv_ceil_log2((unsigned)sps->num_st_rps));
      if (sh->st_rps_idx >= sps->num_st_rps) return hv_fail(d, "short_term_ref_pic_set_idx %d", sh->st_rps_idx), -1;
      sh->rps = &sps->st_rps[sh->st_rps_idx];
    }
    if (sps->long_term_present) {
      int num_lt_sps = 0;
      if (sps->num_lt_sps > 0) num_lt_sps = (int)hvb_ue(b);
      const int num_lt_pics = (int)hvb_ue(b);
      sh->num_lt = num_lt_sps + num_lt_pics;
      if (sh->num_lt > 32 || num_lt_sps > sps->num_lt_sps) return hv_fail(d, "slice header: %d long-term pictures", sh->num_lt), -1;
      for (int i = 0; i < sh->num_lt; ++i) {
        if (i < num_lt_sps) {
          int idx = 0;
          if (sps->num_lt_sps > 1) idx = (int)hvb_u(b, hv_ceil_log2((unsigned)sps->num_lt_sps));
          if (idx >= sps->num_lt_sps) return hv_fail(d, "lt_idx_sps %d", idx), -1;
          sh->lt_poc_lsb[i] = sps->lt_poc_lsb_sps[idx];
          sh->lt_used[i] = sps->lt_used_sps[idx];
        } else {
          sh->lt_poc_lsb[i] = (int)hvb_u(b, sps->log2_max_poc_lsb);
          sh->lt_used[i] = (int)hvb_u(b, 1);
        }
        sh->lt_msb_present[i] = (int)hvb_u(b, 1);
        sh->lt_delta_msb_cycle[i] = 0;
        if (sh->lt_msb_present[i]) sh->lt_delta_msb_cycle[i] = (int)hvb_ue(b);
        /* (7-52): the cycles accumulate within each of the two groups */
        if (i != 0 && i != num_lt_sps) sh->lt_delta_msb_cycle[i] += sh->lt_delta_msb_cycle[i - 1];
      }
    }
    if (sps->temporal_mvp_enabled) sh->temporal_mvp = (int)hvb_u(b, 1);
  }
  if (sps->sao_enabled) {
    sh->sao_luma = (int)hvb_u(b, 1);
    sh->sao_chroma = (int)hvb_u(b, 1);
  }
  sh->num_ref_idx[0] = sh->num_ref_idx[1] = 0;
  sh->collocated_from_l0 = 1;
  if (sh->slice_type != HV_SLICE_I) {
    sh->num_ref_idx[0] = pps->num_ref_idx_default[0];
    sh->num_ref_idx[1] = sh->slice_type == HV_SLICE_B ? pps->num_ref_idx_default[1] : 0;
    if (hvb_u(b, 1)) {
      sh->num_ref_idx[0] = 1 + (int)hvb_ue(b);
      if (sh->slice_type == HV_SLICE_B) sh->num_ref_idx[1] = 1 + (int)hvb_ue(b);
    }
    if (sh->num_ref_idx[0] > 15 || sh->num_ref_idx[1] > 15) return hv_fail(d, "num_ref_idx_active %d / %d", sh->num_ref_idx[0], sh->num_ref_idx[1]), -1;
    int total = 0; /* NumPicTotalCurr (7-55) */
    if (sh->rps) {
      for (int i = 0; i < sh->rps->num_negative; ++i) total += sh->rps->used_s0[i];
      for (int i = 0; i < sh->rps->num_positive; ++i) total += sh->rps->used_s1[i];
    }
    for (int i = 0; i < sh->num_lt; ++i) total += sh->lt_used[i];
    if (total == 0) return hv_fail(d, "a P or B slice whose reference picture set is empty"), -1;
    if (pps->lists_modification_present && total > 1) {
      const int bits = hv_ceil_log2((unsigned)total);
      for (int l = 0; l < (sh->slice_type == HV_SLICE_B ? 2 : 1); ++l) {
        sh->rplm_flag[l] = (int)hvb_u(b, 1);
        if (sh->rplm_flag[l])
          for (int i = 0; i < sh->num_ref_idx[l]; ++i) {
            sh->list_entry[l][i] = (int)hvb_u(b, bits);
            if (sh->list_entry[l][i] >= total) return hv_fail(d, "list_entry %d of %d", sh->list_entry[l][i], total), -1;
          }
      }
    }
    if (sh->slice_type == HV_SLICE_B) sh->mvd_l1_zero = (int)hvb_u(b, 1);
    if (pps->cabac_init_present) sh->cabac_init_flag = (int)hvb_u(b, 1);
    if (sh->temporal_mvp) {
      if (sh->slice_type == HV_SLICE_B) sh->collocated_from_l0 = (int)hvb_u(b, 1);
      if ((sh->collocated_from_l0 && sh->num_ref_idx[0] > 1) || (!sh->collocated_from_l0 && sh->num_ref_idx[1] > 1))
        sh->collocated_ref_idx = (int)hvb_ue(b);
      if (sh->collocated_ref_idx >= sh->num_ref_idx[sh->collocated_from_l0 ? 0 : 1]) return hv_fail(d, "collocated_ref_idx %d", sh->collocated_ref_idx), -1;
    }
    if ((pps->weighted_pred && sh->slice_type == HV_SLICE_P) || (pps->weighted_bipred && sh->slice_type == HV_SLICE_B)) {
      if (parse_pred_weight_table(b, sh)) return hv_fail(d, "bad pred_weight_table"), -1;
    }
    sh->max_num_merge_cand = 5 - (int)hvb_ue(b);
    if (sh->max_num_merge_cand < 1 || sh->max_num_merge_cand > 5) return hv_fail(d, "MaxNumMergeCand %d", sh->max_num_merge_cand), -1;
  }
  sh->qp_delta = hvb_se(b);
  if (pps->slice_chroma_qp_offsets_present) {
    sh->cb_qp_offset = hvb_se(b);
    sh->cr_qp_offset = hvb_se(b);
  }
  int override = 0;
  if (pps->deblocking_override_enabled) override = (int)hvb_u(b, 1);
  sh->deblocking_disabled = pps->deblocking_disabled;
  sh->beta_offset_div2 = pps->beta_offset_div2;
  sh->tc_offset_div2 = pps->tc_offset_div2;
  if (override) {
    sh->deblocking_disabled = (int)hvb_u(b, 1);
    if (!sh->deblocking_disabled) {
      sh->beta_offset_div2 = hvb_se(b);
      sh->tc_offset_div2 = hvb_se(b);
    }
  }
  sh->loop_filter_across_slices = pps->loop_filter_across_slices;
  if (pps->loop_filter_across_slices && (sh->sao_luma || sh->sao_chroma || !sh->deblocking_disabled))
    sh->loop_filter_across_slices = (int)hvb_u(b, 1);
  if (pps->entropy_coding_sync) {
    sh->num_entry_points = (int)hvb_ue(b);
    if (sh->num_entry_points > 0) {
      const int len = 1 + (int)hvb_ue(b);
      if (len > 32) return hv_fail(d, "offset_len_minus1 %d", len - 1), -1;
      for (int i = 0; i < sh->num_entry_points; ++i) {
        const uint32_t v = hvb_u(b, len) + 1;
        if (i < 256) sh->entry_point[i] = v;   /* (the walk finds the rows by itself, and compares: hv_slice_data) */
      }
    }
  }
  if (pps->slice_header_extension_present) {
    const int n = (int)hvb_ue(b);
    for (int i = 0; i < n; ++i) hvb_u(b, 8);
  }
  /* byte_alignment(): a one, then zeros */
  if (hvb_u(b, 1) != 1) return hv_fail(d, "slice header does not end on its alignment bit"), -1;
  while (b->pos & 7)
    if (hvb_u(b, 1)) return hv_fail(d, "slice header alignment bits are not zero"), -1;
  if (b->overrun) return hv_fail(d, "slice header: ran out of data"), -1;
  sh->data_byte_offset = b->pos >> 3;
  sh->slice_qp = pps->init_qp + sh->qp_delta;
  if (sh->slice_qp < 0 || sh->slice_qp > 51) return hv_fail(d, "SliceQpY %d", sh->slice_qp), -1;
  sh->slice_addr_rs = sh->segment_address;
  return 0;
}

/* --------------------------------------------------------------------------------------------------------- DPB */
static void free_pic_buffers(HvPic* p) {
  free(p->plane[0]);
  free(p->mvf);
  memset(p, 0, sizeof(*p));
}

HvPic* hv_alloc_pic(HevcOracle* d) {
  const HvSps* s = d->cur_sps;
  for (int i = 0; i < HV_MAX_DPB; ++i) {
    HvPic* p = &d->dpb[i];
    if (p->in_use) continue;
    const size_t ysz = (size_t)s->width * s->height;
    if (!p->plane[0]) {
      p->plane[0] = (uint8_t*)malloc(ysz * 3 / 2);
      p->mvf = (HvMv*)malloc(sizeof(HvMv) * (size_t)(s->width / 4) * (s->height / 4));
      if (!p->plane[0] || !p->mvf) return NULL;
    }
    p->plane[1] = p->plane[0] + ysz;
    p->plane[2] = p->plane[1] + ysz / 4;
    memset(p->mvf, 0, sizeof(HvMv) * (size_t)(s->width / 4) * (s->height / 4));
    p->in_use = 1;
    p->ref_state = 0;
    p->needed_for_output = 0;
    p->latency_count = 0;
    p->missing = 0;
    return p;
  }
  return NULL;
}

static void release_unused(HevcOracle* d) {
  for (int i = 0; i < HV_MAX_DPB; ++i) {
    HvPic* p = &d->dpb[i];
    if (p->in_use && p != d->cur && !p->ref_state && !p->needed_for_output) p->in_use = 0;
  }
}

static void output_pic(HevcOracle* d, HvPic* p) {
  const HvSps* s = d->cur_sps;
  p->needed_for_output = 0;
  if (d->out_count >= 64) {
    hv_fail(d, "output queue overflow (frames are not being fetched)");
    return;
  }
  const int w = s->width - s->conf_left - s->conf_right, h = s->height - s->conf_top - s->conf_bottom;
  const int cw = (w + 1) / 2, ch = (h + 1) / 2;
  uint8_t* o = (uint8_t*)malloc((size_t)w * h + 2 * (size_t)cw * ch);
  uint8_t* q = o;
  for (int y = 0; y < h; ++y, q += w) memcpy(q, p->plane[0] + (size_t)(y + s->conf_top) * s->width + s->conf_left, (size_t)w);
  for (int c = 1; c < 3; ++c)
    for (int y = 0; y < ch; ++y, q += cw) memcpy(q, p->plane[c] + (size_t)(y + s->conf_top / 2) * (s->width / 2) + s->conf_left / 2, (size_t)cw);
  const int slot = (d->out_head + d->out_count) % 64;
  d->outq[slot].i420 = o;
  d->outq[slot].w = w;
  d->outq[slot].h = h;
  d->outq[slot].poc = p->poc;
  ++d->out_count;
}

/* C.5.2.4 "bumping": the picture with the smallest POC among those waiting for output goes out */
static int bump(HevcOracle* d) {
  HvPic* best = NULL;
  for (int i = 0; i < HV_MAX_DPB; ++i) {
    HvPic* p = &d->dpb[i];
    if (p->in_use && p->needed_for_output && (!best || p->poc < best->poc)) best = p;
  }
  if (!best) return 0;
  output_pic(d, best);
  return 1;
}
static int num_waiting(const HevcOracle* d) {
  int n = 0;
  for (int i = 0; i < HV_MAX_DPB; ++i) n += d->dpb[i].in_use && d->dpb[i].needed_for_output;
  return n;
}
static int dpb_fullness(const HevcOracle* d) {
  int n = 0;
  for (int i = 0; i < HV_MAX_DPB; ++i) n += d->dpb[i].in_use && &d->dpb[i] != d->cur;
  return n;
}
static int latency_exceeded(const HevcOracle* d) {
  const HvSps* s = d->cur_sps;
  if (!s->max_latency_increase_plus1) return 0;
  const int max_latency = s->max_num_reorder + s->max_latency_increase_plus1 - 1;
  for (int i = 0; i < HV_MAX_DPB; ++i)
    if (d->dpb[i].in_use && d->dpb[i].needed_for_output && d->dpb[i].latency_count >= max_latency) return 1;
  return 0;
}

static HvPic* find_ref(HevcOracle* d, int poc, int lsb_mask, int want_state) {
  for (int i = 0; i < HV_MAX_DPB; ++i) {
    HvPic* p = &d->dpb[i];
    if (!p->in_use || p == d->cur || !p->ref_state) continue;
    if (want_state == 1 && p->ref_state != 1) continue;
    if (lsb_mask ? (p->poc & lsb_mask) == poc : p->poc == poc) return p;
  }
  return NULL;
}

/* 8.3.3: a reference that is not there (a stream entered at a CRA picture, or damaged) is made up: mid-grey, intra */
static HvPic* generate_missing(HevcOracle* d, int poc, int state) {
  HvPic* keep = d->cur;
  HvPic* p = hv_alloc_pic(d);
  if (!p) {
    for (int i = 0; i < 3 && !p; ++i) /* (make room: the oldest picture waiting for output goes) */
      if (bump(d)) {
        release_unused(d);
        p = hv_alloc_pic(d);
      }
  }
  d->cur = keep;
  if (!p) return NULL;
  const HvSps* s = d->cur_sps;
  memset(p->plane[0], 128, (size_t)s->width * s->height * 3 / 2);
  p->poc = poc;
  p->ref_state = state;
  p->missing = 1;
  p->decode_order = d->decode_counter++;
  return p;
}

/* 8.3.1 + 8.3.2 + C.5.2.2 for the picture whose first slice segment header is in d->sh */
static int ensure_maps(HevcOracle* d);

int hv_start_picture(HevcOracle* d, int nal_type, int temporal_id) {
  const HvSps* sps = d->cur_sps;
  if (ensure_maps(d)) return -1; /* (a new picture size empties the DPB: before the current picture takes a buffer) */
  const HvSliceHeader* sh = &d->sh;
  const int irap = nal_type >= 16 && nal_type <= 23;
  const int idr = nal_type == 19 || nal_type == 20;
  d->no_rasl_output = irap && (idr || nal_type <= 18 || d->first_picture); /* NoRaslOutputFlag: IDR, BLA, or a CRA that starts the sequence */
  /* 8.3.1 */
  const int max_lsb = 1 << sps->log2_max_poc_lsb;
  int msb = 0;
  if (!(irap && d->no_rasl_output)) {
    const int prev_lsb = d->prev_tid0_poc & (max_lsb - 1), prev_msb = d->prev_tid0_poc - prev_lsb;
    if (sh->poc_lsb < prev_lsb && prev_lsb - sh->poc_lsb >= max_lsb / 2)
      msb = prev_msb + max_lsb;
    else if (sh->poc_lsb > prev_lsb && sh->poc_lsb - prev_lsb > max_lsb / 2)
      msb = prev_msb - max_lsb;
    else
      msb = prev_msb;
  }
  d->poc = msb + sh->poc_lsb;
  /* RASL / RADL pictures and sub-layer non-reference pictures do not move prevTid0Pic */
  if (temporal_id == 0 && !(nal_type >= 6 && nal_type <= 9) && !(nal_type <= 14 && (nal_type & 1) == 0)) d->prev_tid0_poc = d->poc;

  /* 8.3.2 */
  d->cur = NULL;
  if (irap && d->no_rasl_output) {
    /* C.5.2.2: the pictures of the previous sequence go out (no_output_of_prior_pics_flag is not honoured: the
     * callers of this oracle count one frame per sample) and the DPB is emptied */
    while (bump(d)) {
    }
    for (int i = 0; i < HV_MAX_DPB; ++i) {
      d->dpb[i].ref_state = 0;
      d->dpb[i].in_use = 0;
    }
  }
  d->n_st_before = d->n_st_after = d->n_lt = 0;
  int marked[HV_MAX_DPB] = {0};
  if (!idr && sh->rps) {
    const HvRps* r = sh->rps;
    for (int i = 0; i < sh->num_lt; ++i) { /* long-term first: a picture that is long-term is not looked at as short-term */
      int poc_lt = sh->lt_poc_lsb[i];
      if (sh->lt_msb_present[i]) poc_lt += d->poc - sh->lt_delta_msb_cycle[i] * max_lsb - (d->poc & (max_lsb - 1));
      HvPic* p = find_ref(d, poc_lt, sh->lt_msb_present[i] ? 0 : max_lsb - 1, 0);
      if (!p && sh->lt_used[i]) p = generate_missing(d, poc_lt, 2);
      if (p) {
        p->ref_state = 2;
        marked[p - d->dpb] = 1;
        if (sh->lt_used[i] && d->n_lt < 16) d->rps_lt[d->n_lt++] = p;
      }
    }
    for (int i = 0; i < r->num_negative; ++i) {
      const int poc = d->poc + r->delta_poc_s0[i];
      HvPic* p = find_ref(d, poc, 0, 1);
      if (!p && r->used_s0[i]) p = generate_missing(d, poc, 1);
      if (p) {
        marked[p - d->dpb] = 1;
        if (r->used_s0[i]) d->rps_st_before[d->n_st_before++] = p;
      }
    }
    for (int i = 0; i < r->num_positive; ++i) {
      const int poc = d->poc + r->delta_poc_s1[i];
      HvPic* p = find_ref(d, poc, 0, 1);
      if (!p && r->used_s1[i]) p = generate_missing(d, poc, 1);
      if (p) {
        marked[p - d->dpb] = 1;
        if (r->used_s1[i]) d->rps_st_after[d->n_st_after++] = p;
      }
    }
  }
  for (int i = 0; i < HV_MAX_DPB; ++i)
    if (d->dpb[i].in_use && !marked[i]) d->dpb[i].ref_state = 0;
  /* C.5.2.2: room for the current picture */
  release_unused(d);
  while (num_waiting(d) > sps->max_num_reorder || latency_exceeded(d) || dpb_fullness(d) >= hv_max(sps->max_dec_pic_buffering, 1)) {
    if (!bump(d)) break;
    release_unused(d);
  }
  d->cur = hv_alloc_pic(d);
  if (!d->cur) return hv_fail(d, "decoded picture buffer is full (%d pictures held)", dpb_fullness(d)), -1;
  d->cur->poc = d->poc;
  d->cur->decode_order = d->decode_counter++;
  d->first_picture = 0;
  return 0;
}

/* 8.3.4 */
int hv_build_ref_lists(HevcOracle* d) {
  const HvSliceHeader* sh = &d->sh;
  if (sh->slice_type == HV_SLICE_I) return 0;
  const int total = d->n_st_before + d->n_st_after + d->n_lt;
  if (total == 0) return hv_fail(d, "no reference pictures for a P / B slice"), -1;
  for (int l = 0; l < (sh->slice_type == HV_SLICE_B ? 2 : 1); ++l) {
    HvPic* temp[32];
    const int n_temp = hv_max(sh->num_ref_idx[l], total);
    int r = 0;
    HvPic** first = l == 0 ? d->rps_st_before : d->rps_st_after;
    HvPic** second = l == 0 ? d->rps_st_after : d->rps_st_before;
    const int n_first = l == 0 ? d->n_st_before : d->n_st_after, n_second = l == 0 ? d->n_st_after : d->n_st_before;
    while (r < n_temp) {
      for (int i = 0; i < n_first && r < n_temp; ++i) temp[r++] = first[i];
      for (int i = 0; i < n_second && r < n_temp; ++i) temp[r++] = second[i];
      for (int i = 0; i < d->n_lt && r < n_temp; ++i) temp[r++] = d->rps_lt[i];
    }
    for (int i = 0; i < sh->num_ref_idx[l]; ++i) d->ref_list[l][i] = sh->rplm_flag[l] ? temp[sh->list_entry[l][i]] : temp[i];
  }
  return 0;
}

/* C.5.2.3: the current picture is decoded */
void hv_finish_picture(HevcOracle* d) {
  const HvSps* sps = d->cur_sps;
  HvPic* cur = d->cur;
  if (!cur) return;
  for (int i = 0; i < HV_MAX_DPB; ++i)
    if (d->dpb[i].in_use && d->dpb[i].needed_for_output && &d->dpb[i] != cur) ++d->dpb[i].latency_count;
  cur->needed_for_output = 1; /* (pic_output_flag = 0 and RASL pictures are output all the same: one frame per sample) */
  cur->latency_count = 0;
  cur->ref_state = 1;
  d->cur = NULL;
  while (num_waiting(d) > sps->max_num_reorder || latency_exceeded(d))
    if (!bump(d)) break;
  release_unused(d);
}

/* ------------------------------------------------------------------------------------------------ public API */
HevcOracle* hevco_open(void) {
  HevcOracle* d = (HevcOracle*)calloc(1, sizeof(HevcOracle));
  if (d) {
    d->first_picture = 1;
    d->nal_length_size = 4;
  }
  return d;
}

static void free_maps(HevcOracle* d) {
  free(d->m_pred); free(d->m_skip); free(d->m_depth); free(d->m_imode); free(d->m_flags); free(d->m_qp); free(d->m_nz);
  free(d->m_edge); free(d->m_slice); free(d->sao); free(d->bs[0]); free(d->bs[1]); free(d->cus); free(d->tus);
  free(d->nodes); free(d->coefs); free(d->deblocked[0]);
  d->m_pred = d->m_skip = d->m_depth = d->m_imode = d->m_flags = d->m_nz = d->m_edge = NULL;
  d->m_qp = NULL; d->m_slice = NULL; d->sao = NULL; d->bs[0] = d->bs[1] = NULL; d->cus = NULL; d->tus = NULL;
  d->nodes = NULL; d->coefs = NULL; d->deblocked[0] = NULL;
  d->w4 = d->h4 = 0;
}

void hevco_close(HevcOracle* d) {
  if (!d) return;
  for (int i = 0; i < HV_MAX_DPB; ++i) free_pic_buffers(&d->dpb[i]);
  while (d->out_count) hevco_pop_frame(d, NULL, 0);
  free_maps(d);
  free(d->ce.buf);
  free(d);
}

const char* hevco_error(HevcOracle* d) { return d->error; }

static int decode_nal(HevcOracle* d, const uint8_t* nal, size_t n);

/* ISO/IEC 14496-15 8.3.3.1.2: HEVCDecoderConfigurationRecord */
int hevco_configure(HevcOracle* d, const uint8_t* c, size_t size) {
  d->error[0] = 0;
  if (size < 23 || c[0] != 1) return hv_fail(d, "hvcC: not a version-1 HEVCDecoderConfigurationRecord (%zu bytes)", size), -1;
  d->nal_length_size = (c[21] & 3) + 1;
  const int arrays = c[22];
  size_t o = 23;
  for (int a = 0; a < arrays; ++a) {
    if (o + 3 > size) return hv_fail(d, "hvcC: truncated array header"), -1;
    const int count = c[o + 1] << 8 | c[o + 2];
    o += 3;
    for (int i = 0; i < count; ++i) {
      if (o + 2 > size) return hv_fail(d, "hvcC: truncated NAL length"), -1;
      const size_t len = (size_t)c[o] << 8 | c[o + 1];
      o += 2;
      if (o + len > size) return hv_fail(d, "hvcC: NAL unit runs past the record"), -1;
      if (decode_nal(d, c + o, len)) return -1;
      o += len;
    }
  }
  d->first_picture = 1;
  return 0;
}

static int ensure_maps(HevcOracle* d) {
  const HvSps* s = d->cur_sps;
  const int w4 = s->width / 4, h4 = s->height / 4;
  if (d->w4 == w4 && d->h4 == h4 && d->m_pred) return 0;
  free_maps(d);
  for (int i = 0; i < HV_MAX_DPB; ++i) free_pic_buffers(&d->dpb[i]); /* a new picture size: nothing of the old one survives */
  const size_t n = (size_t)w4 * h4;
  d->w4 = w4;
  d->h4 = h4;
  d->m_pred = (uint8_t*)malloc(n); d->m_skip = (uint8_t*)malloc(n); d->m_depth = (uint8_t*)malloc(n);
  d->m_imode = (uint8_t*)malloc(n); d->m_flags = (uint8_t*)malloc(n); d->m_qp = (int8_t*)malloc(n);
  d->m_nz = (uint8_t*)malloc(n); d->m_edge = (uint8_t*)malloc(n);
  d->bs[0] = (uint8_t*)malloc(n); d->bs[1] = (uint8_t*)malloc(n);
  d->m_slice = (uint16_t*)malloc(sizeof(uint16_t) * (size_t)s->width_ctb * s->height_ctb);
  d->sao = (HvSao*)malloc(sizeof(HvSao) * (size_t)s->width_ctb * s->height_ctb);
  d->cap_cu = (s->width / 8) * (s->height / 8);
  d->cus = (HvCu*)malloc(sizeof(HvCu) * (size_t)d->cap_cu);
  d->cap_tu = (int)n;
  d->tus = (HvTu*)malloc(sizeof(HvTu) * n);
  d->cap_nodes = (int)(n * 2);
  d->nodes = (HvNode*)malloc(sizeof(HvNode) * (size_t)d->cap_nodes);
  d->cap_coefs = s->width * s->height * 3 / 2 + 64;
  d->coefs = (int16_t*)malloc(sizeof(int16_t) * (size_t)d->cap_coefs);
  d->deblocked[0] = (uint8_t*)malloc((size_t)s->width * s->height * 3 / 2);
  if (!d->m_pred || !d->m_skip || !d->m_depth || !d->m_imode || !d->m_flags || !d->m_qp || !d->m_nz || !d->m_edge || !d->bs[0] ||
      !d->bs[1] || !d->m_slice || !d->sao || !d->cus || !d->tus || !d->nodes || !d->coefs || !d->deblocked[0])
    return hv_fail(d, "out of memory"), -1;
  d->deblocked[1] = d->deblocked[0] + (size_t)s->width * s->height;
  d->deblocked[2] = d->deblocked[1] + (size_t)s->width * s->height / 4;
  return 0;
}

/* what a picture needs before its first slice: empty maps and pools */
int hv_begin_picture_state(HevcOracle* d) {
  if (ensure_maps(d)) return -1;
  const size_t n = (size_t)d->w4 * d->h4;
  memset(d->m_pred, 0, n); memset(d->m_skip, 0, n); memset(d->m_depth, 0, n); memset(d->m_imode, 1, n);
  memset(d->m_flags, 0, n); memset(d->m_qp, 0, n); memset(d->m_nz, 0, n); memset(d->m_edge, 0, n);
  memset(d->bs[0], 0, n); memset(d->bs[1], 0, n);
  memset(d->sao, 0, sizeof(HvSao) * (size_t)d->cur_sps->width_ctb * d->cur_sps->height_ctb);
  memset(d->m_slice, 0xff, sizeof(uint16_t) * (size_t)d->cur_sps->width_ctb * d->cur_sps->height_ctb);
  d->num_cu = d->num_tu = d->num_nodes = d->num_coefs = d->num_slices = 0;
  d->recon_next = d->walk_cu = 0;
  d->have_md5 = 0;
  return 0;
}

/* everything after the last slice of a picture: reconstruct, filter, check the hash, hand over to the DPB */
int hv_complete_picture(HevcOracle* d) {
  if (!d->cur) return 0;
  if (!d->parse_only) {
    hv_deblock_picture(d);
    hv_sao_picture(d);
    if (d->have_md5) {
      uint8_t got[3][16];
      hv_picture_md5(d, d->cur, got);
      ++d->counters[9];
      if (memcmp(got, d->md5, 48)) ++d->counters[10];
    }
  } else {
    memset(d->cur->plane[0], 128, (size_t)d->cur_sps->width * d->cur_sps->height * 3 / 2);
  }
  ++d->counters[3];
  hv_finish_picture(d);
  return 0;
}

static int decode_sei(HevcOracle* d, HvBits* b, int suffix) {
  while (b->pos + 16 <= b->size * 8) {
    int type = 0, size = 0, v;
    do {
      v = (int)hvb_u(b, 8);
      type += v;
    } while (v == 255);
    do {
      v = (int)hvb_u(b, 8);
      size += v;
    } while (v == 255);
    if (b->overrun || b->pos / 8 + (size_t)size > b->size) return 0; /* (a damaged SEI is not a decoding error) */
    if (suffix && type == 132 && size >= 1) {
      const uint8_t* p = b->p + b->pos / 8;
      if (p[0] == 0 && size >= 49) {
        memcpy(d->md5, p + 1, 48);
        d->have_md5 = 1;
      }
    }
    b->pos += (size_t)size * 8;
    if (b->pos / 8 + 1 >= b->size) break; /* only the trailing bits are left */
  }
  return 0;
}

static int decode_nal(HevcOracle* d, const uint8_t* nal, size_t n) {
  if (n < 2) return hv_fail(d, "NAL unit of %zu bytes", n), -1;
  if (nal[0] & 0x80) return hv_fail(d, "forbidden_zero_bit is set"), -1;
  const int type = (nal[0] >> 1) & 63;
  const int layer = (nal[0] & 1) << 5 | nal[1] >> 3;
  const int tid = (nal[1] & 7) - 1;
  if (layer != 0) return 0; /* (other layers: not ours) */
  uint8_t* rbsp = (uint8_t*)malloc(n);
  HvBits b;
  memset(&b, 0, sizeof(b));
  b.p = rbsp;
  b.size = hv_unescape(nal + 2, n - 2, rbsp);
  int rc = 0;
  if (type == 33) {
    rc = hv_parse_sps(d, &b);
  } else if (type == 34) {
    rc = hv_parse_pps(d, &b);
  } else if (type == 39 || type == 40) {
    rc = decode_sei(d, &b, type == 40);
  } else if (type <= 9 || (type >= 16 && type <= 21)) {
    HvSliceHeader sh;
    rc = hv_parse_slice_header(d, &b, type, &sh);
    if (!rc) {
      if (sh.first_slice_segment_in_pic) {
        if (d->cur) rc = hv_complete_picture(d); /* (a sample with two pictures: not expected, but well-defined) */
        d->sh = sh;
        d->sh.rps = sh.st_rps_sps_flag || !sh.rps ? sh.rps : &d->sh.st_rps_local;
        if (!rc) rc = hv_start_picture(d, type, tid);
        if (!rc) rc = hv_begin_picture_state(d);
      } else {
        if (!d->cur) rc = (hv_fail(d, "a slice segment that is not the first of its picture, and no picture was started"), -1);
        d->sh = sh;
        d->sh.rps = sh.st_rps_sps_flag || !sh.rps ? sh.rps : &d->sh.st_rps_local;
      }
      if (!rc) rc = hv_build_ref_lists(d);
      if (!rc) {
        if (d->num_slices >= HV_MAX_SLICES) {
          rc = (hv_fail(d, "more than %d slices in a picture", HV_MAX_SLICES), -1);
        } else {
          HvSliceFacts* f = &d->slice_facts[d->num_slices];
          f->slice_addr_rs = d->sh.slice_addr_rs;
          f->deblocking_disabled = d->sh.deblocking_disabled;
          f->beta_offset_div2 = d->sh.beta_offset_div2;
          f->tc_offset_div2 = d->sh.tc_offset_div2;
          f->loop_filter_across_slices = d->sh.loop_filter_across_slices;
          f->cb_qp_offset = d->sh.cb_qp_offset;
          f->cr_qp_offset = d->sh.cr_qp_offset;
          f->sao_luma = d->sh.sao_luma;
          f->sao_chroma = d->sh.sao_chroma;
          d->slice_idx = d->num_slices++;
          d->enc = 0;
          hv_cabac_dec_start(&d->cd, rbsp, b.size, d->sh.data_byte_offset);
          ++d->counters[0];
          ++d->counters[4 + (d->sh.slice_type == HV_SLICE_I ? 0 : d->sh.slice_type == HV_SLICE_P ? 1 : 2)];
          rc = hv_slice_data(d);
          /* samples a slice at a time: prediction weights and reference lists are the slice's */
          if (!rc && !d->parse_only)
            for (; d->recon_next < d->num_cu; ++d->recon_next) hv_recon_cu(d, &d->cus[d->recon_next]);
        }
      }
    }
  } /* VPS (32), AUD (35), EOS / EOB (36, 37), filler (38), reserved: nothing to do */
  free(rbsp);
  return rc;
}

int hevco_decode(HevcOracle* d, const uint8_t* sample, size_t size) {
  if (d->error[0]) return -1;
  size_t o = 0;
  while (o + (size_t)d->nal_length_size <= size) {
    size_t len = 0;
    for (int i = 0; i < d->nal_length_size; ++i) len = len << 8 | sample[o + i];
    o += (size_t)d->nal_length_size;
    if (len > size - o) return hv_fail(d, "NAL unit of %zu bytes runs past the sample (%zu left)", len, size - o), -1;
    if (len && decode_nal(d, sample + o, len)) return -1;
    o += len;
  }
  if (hv_complete_picture(d)) return -1;
  return d->error[0] ? -1 : 0;
}

int hevco_flush(HevcOracle* d) {
  if (d->cur) hv_complete_picture(d);
  while (bump(d)) {
  }
  for (int i = 0; i < HV_MAX_DPB; ++i) {
    d->dpb[i].ref_state = 0;
    d->dpb[i].in_use = 0;
  }
  d->first_picture = 1;
  d->prev_tid0_poc = 0;
  return d->error[0] ? -1 : 0;
}

int hevco_frames_ready(HevcOracle* d) { return d->out_count; }

int hevco_frame_info(HevcOracle* d, int* width, int* height, int* poc) {
  if (!d->out_count) return -1;
  *width = d->outq[d->out_head].w;
  *height = d->outq[d->out_head].h;
  *poc = d->outq[d->out_head].poc;
  return 0;
}

int hevco_pop_frame(HevcOracle* d, uint8_t* dst, size_t capacity) {
  if (!d->out_count) return -1;
  const int w = d->outq[d->out_head].w, h = d->outq[d->out_head].h;
  const size_t n = (size_t)w * h + 2 * (size_t)((w + 1) / 2) * ((h + 1) / 2);
  int rc = 0;
  if (dst) {
    if (capacity < n)
      rc = -1;
    else
      memcpy(dst, d->outq[d->out_head].i420, n);
  }
  free(d->outq[d->out_head].i420);
  d->out_head = (d->out_head + 1) % 64;
  --d->out_count;
  return rc;
}

void hevco_counters(HevcOracle* d, int64_t out[12]) { memcpy(out, d->counters, sizeof(d->counters)); }
void hevco_set_parse_only(HevcOracle* d, int on) { d->parse_only = on; }
