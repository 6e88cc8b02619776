/* Corpus generator (input synthesis for tests and bench; not the product): a seeded HEVC Main-profile encoder + MP4
 * muxer.  It exists because no HEVC stream and no HEVC encoder is available in the build image or on the GPU boxes
 * (VERDICT r5 item 2); what it is for is syntax and arithmetic COVERAGE, not compression.
 *
 * It reconstructs through the oracle's own routines (oracle/hevc_recon.c) and writes slice data through the oracle's
 * bidirectional syntax walk (oracle/hevc_slice.c, d->enc = 1): parameter sets and slice headers are written here and
 * read back through the oracle's parsers before anything depends on them.  So a stream and the oracle's decode of it
 * agree by construction - which pins nothing (a mirrored misunderstanding is mirrored in both); what stands against
 * that is listed in oracle/hevc_oracle.h.  Every access unit carries the decoded picture hash SEI (MD5, D.3.19) of
 * the reconstruction.
 *
 *   hevc_encoder --width W --height H --frames N --out clip.mp4 [--yuv-out recon.yuv] [options, see usage()]
 *
 * Tools it emits (each behind an option so that tests can isolate them): CTB 16/32/64, coding quadtree down to 8x8,
 * intra 2Nx2N / NxN over all 35 modes (best match or --random-intra), I_PCM, inter 2Nx2N / 2NxN / Nx2N / NxN / AMP
 * with merge (incl. temporal, combined, zero candidates; parallel merge level), skip, AMVP with temporal candidates,
 * bi-prediction, explicit weighted prediction, residual quadtree with DCT 4-32 / DST / transform skip / transquant
 * bypass, sign data hiding, cu_qp_delta, scaling lists, deblocking (offsets, off, across slices), SAO (band / edge /
 * merge), several slices per picture, wavefront entry points, constrained intra prediction, reference picture sets in
 * the SPS or the slice header, POC wrap, non-reference B pictures and a B pyramid. */
#include <math.h>
#include <stdio.h>

#include "../oracle/hevc_internal.h"

typedef struct {
  int width, height, frames, gop, bframes, refs, qp, seed, fps;
  int ctb, min_tb, max_tb, th_intra, th_inter;
  int amp, sao, pcm, tskip, bypass, sdh, qp_swing, scaling, wp, tmvp, strong, merge_cand, pml, slices, wpp;
  int deblock_off, beta, tc, cip, rps_sps, poc_bits, random_intra, cabac_init, cb_off, cr_off, nal_length_size, extra_nals, open_gop;
  int inter_rps;   /* reference picture sets are coded by prediction from an earlier one wherever a deltaRps exists (7.3.7);
                      needs --rps-sps: a slice header's own set is predicted from one of the SPS's */
  int multi_pps;   /* two picture parameter sets, taken in turn by the pictures: the second has another init_qp_minus26 and
                      num_ref_idx_default_active of 1 / 1 (the slice headers make up for both) */
  int sh_extras;   /* the optional bits of the slice segment header: output_flag_present_flag (pic_output_flag = 1),
                      num_extra_slice_header_bits = 2, slice_segment_header_extension of a few bytes */
  int sub_layers;  /* 2: the sequence says it has two temporal sub-layers (profile_tier_level with a sub-layer level, ordering
                      info per sub-layer) and pictures nothing refers to are in the upper one (nuh_temporal_id_plus1 = 2, TSA_N) */
  int keep_unused; /* every other picture keeps the farthest picture of its set without using it (used_by_curr_pic_flag = 0:
                      RefPicSetStFoll - in the DPB, in no list, not counted in NumPicTotalCurr) */
  int rplm;        /* ref_pic_lists_modification() in the slice headers: list entries drawn from the whole set */
  int long_term;   /* the GOP's first picture stays as a long-term reference once the short-term set has let go of it:
                      1 - listed in every slice header (poc_lsb_lt), 2 - as the SPS's one long-term candidate (lt_idx_sps) */
  int bpyramid, lf_across, mvd_l1_zero, hvc1, intra_only, cover, slice_cb_off;
  const char* out;
  const char* yuv_out;
} Cfg;

static uint32_t hash32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}
static uint32_t hash3(uint32_t a, uint32_t b, uint32_t c) { return hash32(a * 0x9e3779b1U ^ hash32(b * 0x85ebca77U ^ hash32(c + 0xc2b2ae3dU))); }

/* ------------------------------------------------------------------------------------------------------- content */
/* smooth value noise panning at a fractional velocity, moving textured boxes, a little grain */
typedef struct {
  int W, H, dw, dh;
  uint32_t seed;
  uint8_t* plane[3];
} Scene;
static int lat(uint32_t seed, int x, int y) { return (int)(hash3(seed, (uint32_t)x, (uint32_t)y) & 255); }
static int vnoise(uint32_t seed, int64_t x, int64_t y) { /* 1/256 lattice units */
  const int ix = (int)(x >> 8), iy = (int)(y >> 8), fx = (int)(x & 255), fy = (int)(y & 255);
  const int t = lat(seed, ix, iy) * (256 - fx) + lat(seed, ix + 1, iy) * fx, b = lat(seed, ix, iy + 1) * (256 - fx) + lat(seed, ix + 1, iy + 1) * fx;
  return (t * (256 - fy) + b * fy) >> 16;
}
static void scene_render(Scene* s, int t) {
  const int64_t ox = 7 * (int64_t)t, oy = 3 * (int64_t)t; /* quarter samples per frame */
  for (int c = 0; c < 3; ++c) {
    const int sub = c ? 1 : 0, w = s->W >> sub, h = s->H >> sub;
    for (int y = 0; y < h; ++y)
      for (int x = 0; x < w; ++x) {
        const int xx = hv_min(x << sub, s->dw - 1), yy = hv_min(y << sub, s->dh - 1);
        const int64_t wx = (int64_t)xx * 4 + ox, wy = (int64_t)yy * 4 + oy;
        int v;
        if (c == 0)
          v = 36 + (vnoise(s->seed, wx, wy) * 5 >> 3) + (vnoise(s->seed + 1, wx * 5, wy * 5) >> 2) + (vnoise(s->seed + 2, wx * 19, wy * 19) >> 4);
        else
          v = (c == 1 ? 100 : 150) + (c == 1 ? 1 : -1) * (vnoise(s->seed + 2 + c, wx / 2, wy / 2) >> 2);
        for (int k = 0; k < 5; ++k) { /* boxes */
          const uint32_t hs = hash3(s->seed, 900 + k, 0);
          const int bw = 40 + (int)(hs % 90), bh = 30 + (int)((hs >> 8) % 70);
          int bx = (int)(hash3(s->seed, 900 + k, 1) % (unsigned)(s->dw * 4)) + ((int)(hash3(s->seed, 900 + k, 2) % 25) - 12) * t;
          int by = (int)(hash3(s->seed, 900 + k, 3) % (unsigned)(s->dh * 4)) + ((int)(hash3(s->seed, 900 + k, 4) % 13) - 6) * t;
          int dx = (xx * 4 - bx) % (s->dw * 4), dy = (yy * 4 - by) % (s->dh * 4);
          if (dx < 0) dx += s->dw * 4;
          if (dy < 0) dy += s->dh * 4;
          if (dx < bw * 4 && dy < bh * 4)
            v = c == 0 ? 28 + (vnoise(hs, (int64_t)dx * 9, (int64_t)dy * 9) * 3 >> 2) + (vnoise(hs + 5, (int64_t)dx * 37, (int64_t)dy * 37) >> 3)
                       : 64 + (int)(hash3(hs, (uint32_t)c, 7) % 128);
        }
        if (c == 0) {
          const uint32_t g = hash3(s->seed + 77, (uint32_t)(y * w + x), (uint32_t)t);
          v += ((int)(g & 7) + (int)((g >> 3) & 7) - 7) >> 1;
        }
        s->plane[c][y * w + x] = (uint8_t)hv_clip8(v);
      }
  }
}

/* --------------------------------------------------------------------------------------------------- bit writer */
typedef struct {
  uint8_t* b;
  size_t bits, cap;
} BitW;
static void bw_put(BitW* w, uint32_t v, int n) {
  for (int i = n - 1; i >= 0; --i) {
    if ((w->bits >> 3) >= w->cap) {
      w->cap = w->cap * 2 + 256;
      w->b = (uint8_t*)realloc(w->b, w->cap);
      memset(w->b + (w->bits >> 3), 0, w->cap - (w->bits >> 3));
    }
    if ((v >> i) & 1) w->b[w->bits >> 3] |= (uint8_t)(0x80 >> (w->bits & 7));
    ++w->bits;
  }
}
static void bw_ue(BitW* w, uint32_t v) {
  int n = 0;
  while (((v + 1) >> (n + 1)) != 0) ++n;
  bw_put(w, 0, n);
  bw_put(w, v + 1, n + 1);
}
static void bw_se(BitW* w, int v) { bw_ue(w, v > 0 ? (uint32_t)(2 * v - 1) : (uint32_t)(-2 * v)); }
static void bw_trailing(BitW* w) {
  bw_put(w, 1, 1);
  while (w->bits & 7) bw_put(w, 0, 1);
}

typedef struct {
  uint8_t* b;
  size_t n, cap;
} Buf;
static void bput(Buf* b, const void* p, size_t n) {
  if (b->n + n > b->cap) {
    b->cap = (b->n + n) * 2 + 1024;
    b->b = (uint8_t*)realloc(b->b, b->cap);
  }
  memcpy(b->b + b->n, p, n);
  b->n += n;
}
static void b8(Buf* b, uint32_t v) { uint8_t t = (uint8_t)v; bput(b, &t, 1); }
static void b16(Buf* b, uint32_t v) { b8(b, v >> 8); b8(b, v); }
static void b32(Buf* b, uint32_t v) { b16(b, v >> 16); b16(b, v); }
static void bzero(Buf* b, size_t n) { while (n--) b8(b, 0); }
static size_t box_begin(Buf* b, const char* t) { const size_t at = b->n; b32(b, 0); bput(b, t, 4); return at; }
static void box_end(Buf* b, size_t at) {
  const uint32_t n = (uint32_t)(b->n - at);
  b->b[at] = (uint8_t)(n >> 24); b->b[at + 1] = (uint8_t)(n >> 16); b->b[at + 2] = (uint8_t)(n >> 8); b->b[at + 3] = (uint8_t)n;
}

/* NAL unit: two-byte header + RBSP with emulation prevention (7.4.2) */
static int g_nal_tid_plus1 = 1; /* nuh_temporal_id_plus1 of the NAL units written next */
static void nal_append(Buf* out, int type, const uint8_t* rbsp, size_t n) {
  b8(out, (uint32_t)type << 1);
  b8(out, (uint32_t)g_nal_tid_plus1);
  int zeros = 0;
  for (size_t i = 0; i < n; ++i) {
    if (zeros >= 2 && rbsp[i] <= 3) { b8(out, 3); zeros = 0; }
    b8(out, rbsp[i]);
    zeros = rbsp[i] == 0 ? zeros + 1 : 0;
  }
}
static void sample_add_nal(Buf* sample, int length_size, const Buf* nal) {
  for (int i = length_size - 1; i >= 0; --i) b8(sample, (uint32_t)(nal->n >> (8 * i)));
  bput(sample, nal->b, nal->n);
}

/* ----------------------------------------------------------------------------------------------------- encoder */
typedef struct {
  int disp;       /* display index within the GOP = POC */
  int type;       /* HV_SLICE_I / P / B */
  int is_ref, idr;
  int cra, rasl;  /* --open-gop 1: a CRA picture at the head of a GOP, the leading pictures (RASL_N) behind it in decoding order */
  int radl;       /* --open-gop 2: an IDR_W_RADL picture there instead, its leading pictures (RADL_N) predict from it alone */
} PicPlan;

typedef struct {
  Cfg c;
  HevcOracle* d; /* the reconstruction context: DPB, maps, pools, syntax walk */
  Scene scene;
  int cw, ch;    /* coded size */
  Buf vps, sps, pps, pps1;   /* (pps1: the second picture parameter set of --multi-pps) */
  HvRps sps_sets[16];
  int num_sps_sets;
  int refset[HV_MAX_REFS], n_refset; /* POCs of the reference pictures kept */
  int purge_below;                   /* --open-gop: the CRA picture's POC while what precedes it is still kept for its leading pictures, else -1 */
  uint32_t pic_counter;
  /* per picture */
  HvSliceHeader slice_sh[64];
  int slice_start[65];
  int n_slices;
  uint32_t h; /* hash base of the current picture */
} Enc;

static void put_profile_tier_level_sub(BitW* w, int max_sub_layers_minus1);
static void put_profile_tier_level(BitW* w) { put_profile_tier_level_sub(w, 0); }
static void put_profile_tier_level_sub(BitW* w, int msl) {
  bw_put(w, 0, 2); bw_put(w, 0, 1); bw_put(w, 1, 5);   /* profile space 0, Main tier, Main profile */
  bw_put(w, 0x60000000u, 32);                           /* compatible with Main and Main 10 */
  bw_put(w, 1, 1); bw_put(w, 0, 1); bw_put(w, 0, 1); bw_put(w, 1, 1);
  bw_put(w, 0, 32); bw_put(w, 0, 11); bw_put(w, 0, 1);  /* 43 reserved zero bits, general_inbld_flag */
  bw_put(w, 120, 8);                                    /* level 4 */
  for (int i = 0; i < msl; ++i) { bw_put(w, 0, 1); bw_put(w, 1, 1); }   /* sub_layer_profile_present_flag 0, sub_layer_level_present_flag 1 */
  if (msl > 0)
    for (int i = msl; i < 8; ++i) bw_put(w, 0, 2);      /* reserved_zero_2bits */
  for (int i = 0; i < msl; ++i) bw_put(w, 93, 8);        /* sub_layer_level_idc: 3.1 */
}

static void put_scaling_lists(BitW* w, uint32_t seed) {
  for (int size_id = 0; size_id < 4; ++size_id)
    for (int m = 0; m < (size_id == 3 ? 2 : 6); ++m) {
      const uint32_t h = hash3(seed, (uint32_t)size_id, (uint32_t)m);
      const int kind = (int)(h % 4); /* 0 default, 1 copy of the previous, 2 / 3 explicit */
      if (kind == 0 || (kind == 1 && m == 0)) {
        bw_put(w, 0, 1); bw_ue(w, 0);
      } else if (kind == 1) {
        bw_put(w, 0, 1); bw_ue(w, 1);
      } else {
        bw_put(w, 1, 1);
        const int n = hv_min(64, 1 << (4 + (size_id << 1)));
        int next = 8;
        if (size_id > 1) {
          const int dc = 8 + (int)((h >> 8) % 40);
          bw_se(w, dc - 8);
          next = dc;
        }
        for (int i = 0; i < n; ++i) {
          const int target = hv_clip3(1, 255, 10 + i / 2 + (int)(hash3(h, (uint32_t)i, 1) % 9));
          int delta = target - next;
          if (delta > 127) delta -= 256;
          if (delta < -128) delta += 256;
          bw_se(w, delta);
          next = target;
        }
      }
    }
}

static void put_st_rps(BitW* w, const HvRps* r, int idx) {
  if (idx != 0) bw_put(w, 0, 1); /* inter_ref_pic_set_prediction_flag */
  bw_ue(w, (uint32_t)r->num_negative);
  bw_ue(w, (uint32_t)r->num_positive);
  int prev = 0;
  for (int i = 0; i < r->num_negative; ++i) {
    bw_ue(w, (uint32_t)(prev - r->delta_poc_s0[i] - 1));
    bw_put(w, r->used_s0[i], 1);
    prev = r->delta_poc_s0[i];
  }
  prev = 0;
  for (int i = 0; i < r->num_positive; ++i) {
    bw_ue(w, (uint32_t)(r->delta_poc_s1[i] - prev - 1));
    bw_put(w, r->used_s1[i], 1);
    prev = r->delta_poc_s1[i];
  }
}

/* `r` coded by prediction from `ref` = sets[idx - delta_idx] (inter_ref_pic_set_prediction_flag = 1): possible when one
   deltaRps maps entries of ref (or zero: the picture ref belonged to) onto every entry of r; returns 0 if there is none */
static int put_st_rps_inter(BitW* w, const HvRps* r, const HvRps* ref, int idx, int num_sets, int delta_idx) {
  int tgt[32], nt = 0, src[33], ns = 0;
  for (int i = 0; i < r->num_negative; ++i) tgt[nt++] = r->delta_poc_s0[i];
  for (int i = 0; i < r->num_positive; ++i) tgt[nt++] = r->delta_poc_s1[i];
  for (int i = 0; i < ref->num_negative; ++i) src[ns++] = ref->delta_poc_s0[i];
  for (int i = 0; i < ref->num_positive; ++i) src[ns++] = ref->delta_poc_s1[i];
  src[ns] = 0; /* index NumDeltaPocs: deltaRps itself */
  for (int i = 0; i < r->num_negative; ++i) if (!r->used_s0[i]) return 0;
  for (int i = 0; i < r->num_positive; ++i) if (!r->used_s1[i]) return 0;
  for (int d = -20; d <= 20; ++d) {
    if (!d) continue;
    int covered = 0, clash = 0;
    for (int t = 0; t < nt; ++t) {
      int hit = 0;
      for (int j = 0; j <= ns; ++j) hit += src[j] + d == tgt[t];
      covered += hit > 0;
    }
    for (int j = 0; j <= ns; ++j) clash |= src[j] + d == 0; /* an entry that lands on the picture itself must be switched off: it is (its flags are 0) */
    (void)clash;
    if (covered != nt) continue;
    bw_put(w, 1, 1);
    if (idx == num_sets) bw_ue(w, (uint32_t)delta_idx - 1);
    bw_put(w, d < 0, 1);
    bw_ue(w, (uint32_t)(d < 0 ? -d : d) - 1);
    for (int j = 0; j <= ns; ++j) {
      int member = 0;
      for (int t = 0; t < nt; ++t) member |= src[j] + d == tgt[t];
      bw_put(w, (uint32_t)member, 1);        /* used_by_curr_pic_flag */
      if (!member) bw_put(w, 0, 1);          /* use_delta_flag */
    }
    return 1;
  }
  return 0;
}

static int log2i(int v) { int n = 0; while ((1 << n) < v) ++n; return n; }

static void write_parameter_sets(Enc* e) {
  const Cfg* c = &e->c;
  static const uint8_t vps_rbsp[] = {0x0c, 0x01, 0xff, 0xff, 0x01, 0x60, 0x00, 0x00, 0x00, 0x90, 0x00, 0x00, 0x00, 0x00, 0x00, 0x78, 0x95, 0x98, 0x09};
  e->vps.n = 0;
  nal_append(&e->vps, 32, vps_rbsp, sizeof(vps_rbsp));
  BitW w = {0};
  const int msl = c->sub_layers == 2 ? 1 : 0;
  bw_put(&w, 0, 4); bw_put(&w, (uint32_t)msl, 3); bw_put(&w, 1, 1);
  put_profile_tier_level_sub(&w, msl);
  bw_ue(&w, 0); bw_ue(&w, 1);
  bw_ue(&w, (uint32_t)e->cw); bw_ue(&w, (uint32_t)e->ch);
  if (e->cw != c->width || e->ch != c->height) {
    bw_put(&w, 1, 1); bw_ue(&w, 0); bw_ue(&w, (uint32_t)(e->cw - c->width) / 2); bw_ue(&w, 0); bw_ue(&w, (uint32_t)(e->ch - c->height) / 2);
  } else {
    bw_put(&w, 0, 1);
  }
  bw_ue(&w, 0); bw_ue(&w, 0);
  bw_ue(&w, (uint32_t)c->poc_bits - 4);
  bw_put(&w, 1, 1);                 /* sps_sub_layer_ordering_info_present_flag: one triple per sub-layer */
  for (int i = 0; i <= msl; ++i) {
    bw_ue(&w, (uint32_t)c->refs + 1); /* sps_max_dec_pic_buffering_minus1: the references and the current picture, one to spare */
    bw_ue(&w, c->bframes ? (c->bpyramid ? 2u : 1u) : 0u);
    bw_ue(&w, 0);
  }
  bw_ue(&w, 0);                                   /* minimum coding block 8 */
  bw_ue(&w, (uint32_t)log2i(c->ctb) - 3);
  bw_ue(&w, (uint32_t)log2i(c->min_tb) - 2);
  bw_ue(&w, (uint32_t)(log2i(c->max_tb) - log2i(c->min_tb)));
  bw_ue(&w, (uint32_t)c->th_inter); bw_ue(&w, (uint32_t)c->th_intra);
  bw_put(&w, c->scaling ? 1 : 0, 1);
  if (c->scaling) {
    bw_put(&w, (c->scaling & 1) ? 1 : 0, 1);
    if (c->scaling & 1) put_scaling_lists(&w, (uint32_t)c->seed * 31 + 5);
  }
  bw_put(&w, (uint32_t)c->amp, 1); bw_put(&w, c->sao ? 1 : 0, 1); bw_put(&w, c->pcm ? 1 : 0, 1);
  if (c->pcm) {
    bw_put(&w, c->pcm == 2 ? 5 : 7, 4); bw_put(&w, c->pcm == 2 ? 4 : 7, 4); /* sample bit depths 8 / 8, or 6 / 5 */
    bw_ue(&w, 0); bw_ue(&w, (uint32_t)hv_min(log2i(c->ctb), 5) - 3);         /* 8 .. min(CTB, 32) */
    bw_put(&w, c->pcm == 2 ? 1 : 0, 1);                                        /* pcm_loop_filter_disabled_flag */
  }
  bw_ue(&w, (uint32_t)e->num_sps_sets);
  for (int i = 0; i < e->num_sps_sets; ++i)
    if (!(c->inter_rps && i > 0 && put_st_rps_inter(&w, &e->sps_sets[i], &e->sps_sets[i - 1], i, e->num_sps_sets, 1))) put_st_rps(&w, &e->sps_sets[i], i);
  bw_put(&w, c->long_term ? 1 : 0, 1);    /* long_term_ref_pics_present_flag */
  if (c->long_term) {
    bw_ue(&w, c->long_term == 2 ? 1 : c->long_term == 3 ? 2 : 0);    /* num_long_term_ref_pics_sps */
    if (c->long_term == 3) { bw_put(&w, 5, (int)c->poc_bits); bw_put(&w, 0, 1); }  /* (3) a candidate nobody picks in front of the one that is */
    if (c->long_term >= 2) { bw_put(&w, 0, (int)c->poc_bits); bw_put(&w, 1, 1); }  /* lt_ref_pic_poc_lsb_sps = 0, used_by_curr_pic_lt_sps_flag */
  }
  bw_put(&w, (uint32_t)c->tmvp, 1); bw_put(&w, (uint32_t)c->strong, 1);
  bw_put(&w, 0, 1); bw_put(&w, 0, 1);     /* vui, extension */
  bw_trailing(&w);
  e->sps.n = 0;
  nal_append(&e->sps, 33, w.b, w.bits / 8);
  free(w.b);
  e->pps.n = 0;
  e->pps1.n = 0;
  for (int pps_id = 0; pps_id < (c->multi_pps ? 2 : 1); ++pps_id) {
  memset(&w, 0, sizeof(w));
  bw_ue(&w, (uint32_t)pps_id); bw_ue(&w, 0);
  bw_put(&w, 0, 1); bw_put(&w, c->sh_extras ? 1 : 0, 1); bw_put(&w, c->sh_extras ? 2 : 0, 3); /* dependent slices off, output_flag_present_flag, num_extra_slice_header_bits */
  bw_put(&w, (uint32_t)c->sdh, 1); bw_put(&w, c->cabac_init ? 1 : 0, 1);
  bw_ue(&w, pps_id ? 0u : (uint32_t)hv_min(c->refs, 15) - 1); bw_ue(&w, pps_id ? 0u : (uint32_t)hv_min(c->refs, 15) - 1);
  bw_se(&w, pps_id ? 3 : 0);              /* init_qp_minus26: slices carry the difference */
  bw_put(&w, (uint32_t)c->cip, 1); bw_put(&w, (uint32_t)c->tskip, 1);
  bw_put(&w, c->qp_swing ? 1 : 0, 1);
  if (c->qp_swing) bw_ue(&w, (uint32_t)hv_min(c->qp_swing - 1, log2i(c->ctb) - 3)); /* diff_cu_qp_delta_depth */
  bw_se(&w, c->cb_off); bw_se(&w, c->cr_off);
  bw_put(&w, c->slice_cb_off ? 1 : 0, 1);
  bw_put(&w, c->wp ? 1 : 0, 1); bw_put(&w, c->wp ? 1 : 0, 1);
  bw_put(&w, (uint32_t)c->bypass, 1);
  bw_put(&w, 0, 1); bw_put(&w, (uint32_t)c->wpp, 1);
  bw_put(&w, 1, 1);                       /* pps_loop_filter_across_slices_enabled_flag: the slices say */
  bw_put(&w, 1, 1);                       /* deblocking_filter_control_present_flag */
  bw_put(&w, 1, 1); bw_put(&w, 0, 1);     /* override enabled; not disabled in the PPS */
  bw_se(&w, 0); bw_se(&w, 0);
  bw_put(&w, (c->scaling & 2) ? 1 : 0, 1);
  if (c->scaling & 2) put_scaling_lists(&w, (uint32_t)c->seed * 17 + 3);
  bw_put(&w, c->rplm ? 1 : 0, 1);         /* lists_modification_present_flag */
  bw_ue(&w, (uint32_t)c->pml - 2);
  bw_put(&w, c->sh_extras ? 1 : 0, 1); bw_put(&w, 0, 1);   /* slice_segment_header_extension_present_flag, pps_extension */
  bw_trailing(&w);
  nal_append(pps_id ? &e->pps1 : &e->pps, 34, w.b, w.bits / 8);
  free(w.b);
  }

}

static size_t build_hvcc(const Enc* e, uint8_t* out) {
  size_t o = 0;
  out[o++] = 1; out[o++] = 1;
  out[o++] = 0x60; out[o++] = 0; out[o++] = 0; out[o++] = 0;
  out[o++] = 0x90; out[o++] = 0; out[o++] = 0; out[o++] = 0; out[o++] = 0; out[o++] = 0;
  out[o++] = 120;
  out[o++] = 0xf0; out[o++] = 0; out[o++] = 0xfc; out[o++] = 0xfd; out[o++] = 0xf8; out[o++] = 0xf8;
  out[o++] = 0; out[o++] = 0;
  out[o++] = (uint8_t)(1 << 3 | 1 << 2 | (e->c.nal_length_size - 1));
  out[o++] = 3;
  const Buf* nals[4] = {&e->vps, &e->sps, &e->pps, &e->pps1};
  for (int a = 0; a < 3; ++a) {
    const int count = a == 2 && e->pps1.n ? 2 : 1;   /* the array of picture parameter sets may hold two */
    out[o++] = (uint8_t)(0x80 | (32 + a));
    out[o++] = 0; out[o++] = (uint8_t)count;
    for (int k = 0; k < count; ++k) {
      out[o++] = (uint8_t)(nals[a + k]->n >> 8); out[o++] = (uint8_t)nals[a + k]->n;
      memcpy(out + o, nals[a + k]->b, nals[a + k]->n);
      o += nals[a + k]->n;
    }
  }
  return o;
}

/* ---- GOP plan: decode order ---- */
static int plan_gop(const Cfg* c, int n, PicPlan* out) {
  int k = 0;
  out[k++] = (PicPlan){0, HV_SLICE_I, 1, 1};
  if (c->intra_only) {
    for (int i = 1; i < n; ++i) out[k++] = (PicPlan){i, HV_SLICE_I, 1, 0};
    return k;
  }
  int prev = 0;
  while (prev < n - 1) {
    int next = hv_min(prev + c->bframes + 1, n - 1);
    out[k++] = (PicPlan){next, HV_SLICE_P, 1, 0};
    const int nb = next - prev - 1;
    if (c->bpyramid && nb >= 3) {
      const int mid = prev + (nb + 1) / 2;
      out[k++] = (PicPlan){mid, HV_SLICE_B, 1, 0};
      for (int i = prev + 1; i < next; ++i)
        if (i != mid) out[k++] = (PicPlan){i, HV_SLICE_B, 0, 0};
    } else {
      for (int i = prev + 1; i < next; ++i) out[k++] = (PicPlan){i, HV_SLICE_B, 0, 0};
    }
    prev = next;
  }
  return k;
}

/* --open-gop, every GOP but the first: a CRA picture (21) in place of the IDR picture, displayed behind `bframes` leading
 * pictures that are decoded after it and predict from it AND from the last picture of the GOP before (RASL_N, 8): what
 * x265 writes by default.  Decoding that starts at the CRA picture cannot decode them (8.3.3: their references are made
 * up); everything behind the CRA picture in output order refers to nothing in front of it. */
static int plan_open_gop(const Cfg* c, int n, PicPlan* out) {
  int k = 0;
  const int lead = hv_min(c->bframes, n - 1);
  out[k++] = (PicPlan){lead, HV_SLICE_I, 1, 0, 1, 0};
  for (int i = 0; i < lead; ++i) out[k++] = (PicPlan){i, HV_SLICE_B, 0, 0, 0, 1};
  int prev = lead;
  while (prev < n - 1) {
    int next = hv_min(prev + c->bframes + 1, n - 1);
    out[k++] = (PicPlan){next, HV_SLICE_P, 1, 0, 0, 0};
    for (int i = prev + 1; i < next; ++i) out[k++] = (PicPlan){i, HV_SLICE_B, 0, 0, 0, 0};
    prev = next;
  }
  return k;
}

/* the reference picture set of a picture at `poc`: everything in refset, nearest first on either side */
static void rps_for(const Enc* e, int poc, HvRps* r) {
  memset(r, 0, sizeof(*r));
  int neg[16], pos[16], nn = 0, np = 0;
  for (int i = 0; i < e->n_refset; ++i) {
    if (e->refset[i] < poc) neg[nn++] = e->refset[i];
    if (e->refset[i] > poc) pos[np++] = e->refset[i];
  }
  for (int i = 0; i < nn; ++i)
    for (int j = i + 1; j < nn; ++j)
      if (neg[j] > neg[i]) { int t = neg[i]; neg[i] = neg[j]; neg[j] = t; }
  for (int i = 0; i < np; ++i)
    for (int j = i + 1; j < np; ++j)
      if (pos[j] < pos[i]) { int t = pos[i]; pos[i] = pos[j]; pos[j] = t; }
  r->num_negative = nn;
  r->num_positive = np;
  for (int i = 0; i < nn; ++i) { r->delta_poc_s0[i] = neg[i] - poc; r->used_s0[i] = 1; }
  for (int i = 0; i < np; ++i) { r->delta_poc_s1[i] = pos[i] - poc; r->used_s1[i] = 1; }
  if (e->c.keep_unused && nn + np > 1 && (hash3((uint32_t)e->c.seed, 0x6b75, (uint32_t)poc) & 1)) {
    if (nn > 1) r->used_s0[nn - 1] = 0;
    else if (np > 1) r->used_s1[np - 1] = 0;
  }
}
static int rps_used(const HvRps* r) {
  int n = 0;
  for (int i = 0; i < r->num_negative; ++i) n += r->used_s0[i];
  for (int i = 0; i < r->num_positive; ++i) n += r->used_s1[i];
  return n;
}
static int rps_equal(const HvRps* a, const HvRps* b) {
  if (a->num_negative != b->num_negative || a->num_positive != b->num_positive) return 0;
  for (int i = 0; i < a->num_negative; ++i)
    if (a->delta_poc_s0[i] != b->delta_poc_s0[i] || a->used_s0[i] != b->used_s0[i]) return 0;
  for (int i = 0; i < a->num_positive; ++i)
    if (a->delta_poc_s1[i] != b->delta_poc_s1[i] || a->used_s1[i] != b->used_s1[i]) return 0;
  return 1;
}
static void refset_add(Enc* e, int poc) {
  e->refset[e->n_refset++] = poc;
  if (e->n_refset > e->c.refs) { /* the one furthest back in output order goes */
    int lo = 0;
    for (int i = 1; i < e->n_refset; ++i)
      if (e->refset[i] < e->refset[lo]) lo = i;
    e->refset[lo] = e->refset[--e->n_refset];
  }
}

static int lt_wanted(const Enc* e, const PicPlan* pp) {
  if (!e->c.long_term || pp->idr) return 0;
  for (int i = 0; i < e->n_refset; ++i)
    if (e->refset[i] == 0) return 0;
  return 1;
}

/* ---- slice header (7.3.6.1), mirrored by hv_parse_slice_header ---- */
static void write_slice_header(Enc* e, BitW* w, const PicPlan* pp, int first_ctb, const HvRps* rps, int nl0, int nl1, int n_entry, const uint32_t* entry) {
  const Cfg* c = &e->c;
  const HvSps* sps = &e->d->sps[0];
  const uint32_t h = hash3(e->h, 0x51, (uint32_t)first_ctb);
  bw_put(w, first_ctb == 0, 1);
  if (pp->idr || pp->cra) bw_put(w, 0, 1);   /* no_output_of_prior_pics_flag: every IRAP picture has it */
  const int pps_id = c->multi_pps ? (int)(e->pic_counter & 1) : 0;
  bw_ue(w, (uint32_t)pps_id);
  if (first_ctb) bw_put(w, (uint32_t)first_ctb, hv_ceil_log2((unsigned)(sps->width_ctb * sps->height_ctb)));
  if (c->sh_extras) bw_put(w, (h >> 24) & 3, 2);  /* slice_reserved_flag[0..1]: whatever */
  bw_ue(w, (uint32_t)pp->type);
  if (c->sh_extras) bw_put(w, 1, 1);              /* pic_output_flag */
  if (!pp->idr) {
    bw_put(w, (uint32_t)pp->disp & ((1u << c->poc_bits) - 1), c->poc_bits);
    int idx = -1;
    for (int i = 0; i < e->num_sps_sets; ++i)
      if (rps_equal(&e->sps_sets[i], rps)) idx = i;
    if (idx >= 0) {
      bw_put(w, 1, 1);
      if (e->num_sps_sets > 1) bw_put(w, (uint32_t)idx, hv_ceil_log2((unsigned)e->num_sps_sets));
    } else {
      bw_put(w, 0, 1);
      int predicted = 0;
      if (c->inter_rps)
        for (int k = 1; k <= e->num_sps_sets && !predicted; ++k) predicted = put_st_rps_inter(w, rps, &e->sps_sets[e->num_sps_sets - k], e->num_sps_sets, e->num_sps_sets, k);
      if (!predicted) put_st_rps(w, rps, e->num_sps_sets);
    }
    if (c->long_term) { /* the GOP's first picture (POC 0), as soon as it is in no short-term set any more */
      const int use_lt = lt_wanted(e, pp);
      if (c->long_term >= 2) bw_ue(w, (uint32_t)use_lt);             /* num_long_term_sps */
      bw_ue(w, (uint32_t)(c->long_term == 1 ? use_lt : 0));           /* num_long_term_pics */
      if (use_lt) {
        if (c->long_term == 3) bw_put(w, 1, 1);                        /* lt_idx_sps: the second of two candidates (one bit) */
        if (c->long_term == 1) { bw_put(w, 0, c->poc_bits); bw_put(w, 1, 1); } /* poc_lsb_lt, used_by_curr_pic_lt_flag */
        const int cycle = pp->disp >> c->poc_bits, present = cycle > 0 || ((h >> 21) & 1);
        bw_put(w, (uint32_t)present, 1);                               /* delta_poc_msb_present_flag */
        if (present) bw_ue(w, (uint32_t)cycle);                        /* delta_poc_msb_cycle_lt */
      }
    }
    if (c->tmvp) bw_put(w, pp->type != HV_SLICE_I, 1); /* slice_temporal_mvp_enabled_flag */
  }
  if (c->sao) { bw_put(w, (c->sao & 4) ? 0 : 1, 1); bw_put(w, (c->sao & 8) ? 0 : 1, 1); }
  if (pp->type != HV_SLICE_I) {
    const int def = pps_id ? 1 : hv_min(c->refs, 15);
    const int ovr = nl0 != def || (pp->type == HV_SLICE_B && nl1 != def);
    bw_put(w, (uint32_t)ovr, 1);
    if (ovr) {
      bw_ue(w, (uint32_t)nl0 - 1);
      if (pp->type == HV_SLICE_B) bw_ue(w, (uint32_t)nl1 - 1);
    }
    const int total_curr = rps_used(rps) + lt_wanted(e, pp);   /* NumPicTotalCurr */
    if (c->rplm && total_curr > 1) { /* ref_pic_lists_modification() */
      const int bits = hv_ceil_log2((unsigned)total_curr);
      for (int l = 0; l < (pp->type == HV_SLICE_B ? 2 : 1); ++l) {
        const int on = (int)(hash3(h, 0x726c, (uint32_t)l) % 3) != 0;
        bw_put(w, (uint32_t)on, 1);
        if (on)
          for (int i = 0; i < (l ? nl1 : nl0); ++i) bw_put(w, hash3(h, 0x726d + (uint32_t)l, (uint32_t)i) % (unsigned)total_curr, bits);
      }
    }
    if (pp->type == HV_SLICE_B) bw_put(w, (uint32_t)c->mvd_l1_zero, 1);
    if (c->cabac_init) bw_put(w, c->cabac_init == 2 ? (h >> 3) & 1 : 1, 1);
    if (c->tmvp) {
      int from_l0 = 1;
      if (pp->type == HV_SLICE_B) { from_l0 = (int)((h >> 5) & 1); bw_put(w, (uint32_t)from_l0, 1); }
      const int nl = from_l0 ? nl0 : nl1;
      if (nl > 1) bw_ue(w, (h >> 7) % (unsigned)nl);
    }
    if (c->wp) { /* pred_weight_table */
      const int ld = (int)((h >> 9) % 8), cd = hv_clip3(0, 7, ld + (int)((h >> 12) % 3) - 1);
      bw_ue(w, (uint32_t)ld);
      bw_se(w, cd - ld);
      for (int l = 0; l < (pp->type == HV_SLICE_B ? 2 : 1); ++l) {
        const int n = l ? nl1 : nl0;
        uint32_t hh[16];
        for (int i = 0; i < n; ++i) hh[i] = hash3(h, 0x77 + (uint32_t)l, (uint32_t)i);
        for (int i = 0; i < n; ++i) bw_put(w, hh[i] & 1, 1);
        for (int i = 0; i < n; ++i) bw_put(w, (hh[i] >> 1) & 1, 1);
        for (int i = 0; i < n; ++i) {
          if (hh[i] & 1) {
            /* weights near one (in units of the denominator), small offsets */
            bw_se(w, hv_clip3(-128, 127, ((int)((hh[i] >> 4) % 9) - 4) * (1 << ld) / 16));
            bw_se(w, (int)((hh[i] >> 8) % 9) - 4);
          }
          if ((hh[i] >> 1) & 1)
            for (int j = 0; j < 2; ++j) {
              bw_se(w, hv_clip3(-128, 127, ((int)((hh[i] >> (12 + 4 * j)) % 7) - 3) * (1 << cd) / 16));
              bw_se(w, (int)((hh[i] >> (20 + 4 * j)) % 7) - 3);
            }
        }
      }
    }
    bw_ue(w, (uint32_t)(5 - c->merge_cand));
  }
  bw_se(w, hv_clip3(0, 51, c->qp + ((pp->type == HV_SLICE_B && !pp->is_ref) ? 2 : 0) - (pp->type == HV_SLICE_I ? 2 : 0)) - 26 - (pps_id ? 3 : 0));
  if (c->slice_cb_off) { bw_se(w, c->slice_cb_off); bw_se(w, -c->slice_cb_off); }
  const int override = c->deblock_off || c->beta || c->tc;
  bw_put(w, (uint32_t)override, 1);
  if (override) {
    const int off = c->deblock_off == 1 || (c->deblock_off == 2 && (h & 0x10000));
    bw_put(w, (uint32_t)off, 1);
    if (!off) { bw_se(w, c->beta); bw_se(w, c->tc); }
    if ((c->sao && (c->sao & 12) != 12) || !off) bw_put(w, (uint32_t)c->lf_across, 1);
  } else {
    bw_put(w, (uint32_t)c->lf_across, 1);
  }
  if (c->wpp) {
    /* entry points (7.3.6.1): the bytes of every wavefront row's substream but the last, emulation prevention bytes
     * counted, 32 bits each.  They are known once the slice data is written: the header is written twice - without
     * them for the analysis (which needs a header to parse, not its length), with them for the stream. */
    bw_ue(w, (uint32_t)n_entry);
    if (n_entry) {
      bw_ue(w, 31);
      for (int i = 0; i < n_entry; ++i) bw_put(w, entry[i] - 1, 32);
    }
  }
  if (c->sh_extras) { /* slice_segment_header_extension: length, then bytes nobody interprets */
    const int n = (int)((h >> 26) % 4);
    bw_ue(w, (uint32_t)n);
    for (int i = 0; i < n; ++i) bw_put(w, hash3(h, 0x7368, (uint32_t)i) & 0xff, 8);
  }
  bw_trailing(w);
}

/* ---- forward transform + quantisation (HM-like integer arithmetic) ---- */
static void forward_quant(const HevcOracle* d, const int* res, int log2, int c_idx, int qp, int intra, int tskip, int bypass, int16_t* lev) {
  static const int qscale[6] = {26214, 23302, 20560, 18396, 16384, 14564};
  const int n = 1 << log2;
  if (bypass) {
    for (int i = 0; i < n * n; ++i) lev[i] = (int16_t)res[i];
    return;
  }
  static int64_t coef[32 * 32], tmp[32 * 32];
  if (tskip) {
    for (int i = 0; i < n * n; ++i) coef[i] = (int64_t)res[i] << (7 - log2 + 0); /* 15 - bitDepth - log2 */
  } else {
    const int dst = intra && c_idx == 0 && log2 == 2, step = 32 >> log2;
    const int s1 = log2 - 1, s2 = log2 + 6;
    for (int y = 0; y < n; ++y)      /* rows */
      for (int k = 0; k < n; ++k) {
        int64_t a = 0;
        for (int x = 0; x < n; ++x) a += (int64_t)(dst ? hevc_dst4[k][x] : hevc_trans_matrix[k * step][x]) * res[y * n + x];
        tmp[y * n + k] = (a + (s1 ? (1 << (s1 - 1)) : 0)) >> s1;
      }
    for (int k = 0; k < n; ++k)      /* columns */
      for (int j = 0; j < n; ++j) {
        int64_t a = 0;
        for (int y = 0; y < n; ++y) a += (int64_t)(dst ? hevc_dst4[j][y] : hevc_trans_matrix[j * step][y]) * tmp[y * n + k];
        coef[j * n + k] = (a + (1 << (s2 - 1))) >> s2;
      }
  }
  const uint8_t* m = NULL;
  if (d->cur_sps->scaling_list_enabled && !(tskip && log2 > 2)) {
    const HvScaling* sc = d->cur_pps->scaling_list_present ? &d->cur_pps->scaling : &d->cur_sps->scaling;
    const int mid = (intra ? 0 : 3) + c_idx;
    m = log2 == 2 ? sc->f4[mid] : log2 == 3 ? sc->f8[mid] : log2 == 4 ? sc->f16[mid] : sc->f32[intra ? 0 : 1];
  }
  const int qbits = 14 + qp / 6 + (7 - log2);
  const int64_t round = (int64_t)(intra ? 171 : 85) << (qbits - 9);
  for (int i = 0; i < n * n; ++i) {
    int64_t a = coef[i] < 0 ? -coef[i] : coef[i];
    if (m) a = a * 16 / m[i];
    const int64_t q = (a * qscale[qp % 6] + round) >> qbits;
    lev[i] = (int16_t)hv_clip3(-32768, 32767, (int)(coef[i] < 0 ? -q : q));
  }
}

static int sad_plane(const Enc* e, int c, int x, int y, int w, int h) {
  const int stride = e->cw >> (c ? 1 : 0);
  const uint8_t* a = e->scene.plane[c] + y * stride + x;
  const uint8_t* b = e->d->cur->plane[c] + y * stride + x;
  int s = 0;
  for (int j = 0; j < h; ++j)
    for (int i = 0; i < w; ++i) s += hv_abs(a[j * stride + i] - b[j * stride + i]);
  return s;
}

/* ---- transform tree analysis: decides, quantises, reconstructs; builds nodes and units in the walk's order ---- */
typedef struct {
  Enc* e;
  HvCu* cu;
  int max_depth;
  int any_cbf;
} TtEnc;

static void code_block(TtEnc* t, HvTu* tu, int c) {
  Enc* e = t->e;
  HevcOracle* d = e->d;
  const HvCu* cu = t->cu;
  const int log2 = c ? tu->clog2 : tu->log2, n = 1 << log2;
  const int x = c ? tu->cx : tu->x, y = c ? tu->cy : tu->y;
  const int stride = e->cw >> (c ? 1 : 0);
  const HvSliceFacts* f = &d->slice_facts[d->slice_idx];
  int qp = cu->qp_y;
  if (c) qp = hv_chroma_qp(cu->qp_y, c == 1 ? d->cur_pps->cb_qp_offset + f->cb_qp_offset : d->cur_pps->cr_qp_offset + f->cr_qp_offset);
  static int res[32 * 32];
  const uint8_t* src = e->scene.plane[c] + y * stride + x;
  const uint8_t* rec = d->cur->plane[c] + y * stride + x;
  for (int j = 0; j < n; ++j)
    for (int i = 0; i < n; ++i) res[j * n + i] = src[j * stride + i] - rec[j * stride + i];
  const uint32_t h = hash3(e->h, (uint32_t)(tu->x * 4 + c), (uint32_t)tu->y);
  tu->tskip[c] = (uint8_t)(d->cur_pps->transform_skip_enabled && !cu->tq_bypass && log2 == 2 && (h % 5) == 0);
  static int16_t lev[32 * 32];
  forward_quant(d, res, log2, c, qp, cu->intra, tu->tskip[c], cu->tq_bypass, lev);
  if (e->c.cover && (h >> 8) % 97 == 0 && !cu->tq_bypass) lev[(h >> 16) % (unsigned)(n * n)] = (int16_t)(((h >> 4) & 1) ? 300 + (int)(h % 4000) : -(17 + (int)(h % 90))); /* long escapes */
  int nz = 0;
  for (int i = 0; i < n * n; ++i) nz |= lev[i] != 0;
  tu->cbf[c] = (uint8_t)nz;
  tu->coef[c] = -1;
  if (!nz) return;
  int scan_idx = 0;
  if (cu->intra && (log2 == 2 || (log2 == 3 && c == 0))) {
    const int half = (1 << cu->log2) >> 1;
    const int pi = cu->part_mode == HV_PART_NxN ? ((tu->y - cu->y >= half) << 1 | (tu->x - cu->x >= half)) : 0;
    const int mode = c ? cu->chroma_mode : cu->luma_mode[pi];
    if (mode >= 6 && mode <= 14) scan_idx = 2;
    if (mode >= 22 && mode <= 30) scan_idx = 1;
  }
  if (d->cur_pps->sign_data_hiding && !cu->tq_bypass) hv_sdh_fix(lev, log2, scan_idx);
  tu->coef[c] = hv_new_coefs(d, n * n);
  memcpy(&d->coefs[tu->coef[c]], lev, sizeof(int16_t) * (size_t)(n * n));
  t->any_cbf = 1;
  hv_residual_add(d, cu, tu, c);
}

/* returns cbf_cb | cbf_cr << 1 of the subtree */
static int tt_analyse(TtEnc* t, int x0, int y0, int xb, int yb, int log2, int depth, int blk) {
  Enc* e = t->e;
  HevcOracle* d = e->d;
  HvCu* cu = t->cu;
  const HvSps* s = d->cur_sps;
  const int ni = hv_new_nodes(d, 1);
  ++cu->num_nodes;
  const int intra_split = cu->intra && cu->part_mode == HV_PART_NxN;
  const int inter_split = s->max_th_depth_inter == 0 && !cu->intra && cu->part_mode != HV_PART_2Nx2N && depth == 0;
  int split;
  if (log2 <= s->log2_max_tb && log2 > s->log2_min_tb && depth < t->max_depth && !(intra_split && depth == 0))
    split = hash3(e->h, (uint32_t)(x0 * 8 + depth), (uint32_t)(y0 + 40000)) % 3 == 0;
  else
    split = log2 > s->log2_max_tb || (intra_split && depth == 0) || inter_split;
  d->nodes[ni].split = (uint8_t)split;
  if (split) {
    const int h = 1 << (log2 - 1);
    int sub = 0;
    int child[4];
    for (int i = 0; i < 4; ++i) {
      child[i] = d->num_nodes;
      sub |= tt_analyse(t, x0 + (i & 1) * h, y0 + (i >> 1) * h, x0, y0, log2 - 1, depth + 1, i);
    }
    if (log2 == 3) /* the chroma of four 4x4 luma blocks travels with the fourth; its flags sit on this node */
      for (int i = 0; i < 4; ++i) {
        d->nodes[child[i]].cbf_cb = (uint8_t)(sub & 1);
        d->nodes[child[i]].cbf_cr = (uint8_t)(sub >> 1);
      }
    d->nodes[ni].cbf_cb = (uint8_t)(sub & 1);
    d->nodes[ni].cbf_cr = (uint8_t)(sub >> 1);
    return sub;
  }
  HvTu* tu = hv_new_tu(d);
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
  if (cu->intra) {
    const int half = (1 << cu->log2) >> 1;
    const int pi = intra_split ? ((y0 - cu->y >= half) << 1 | (x0 - cu->x >= half)) : 0;
    hv_intra_predict(d, 0, x0, y0, log2, cu->luma_mode[pi]);
  }
  code_block(t, tu, 0);
  int sub = 0;
  if (tu->has_chroma)
    for (int c = 1; c < 3; ++c) {
      if (cu->intra) hv_intra_predict(d, c, tu->cx, tu->cy, tu->clog2, cu->chroma_mode);
      code_block(t, tu, c);
      sub |= tu->cbf[c] << (c - 1);
    }
  d->nodes[ni].cbf_luma = tu->cbf[0];
  d->nodes[ni].cbf_cb = (uint8_t)(sub & 1);
  d->nodes[ni].cbf_cr = (uint8_t)(sub >> 1);
  return sub;
}

/* a unit whose prediction is in the picture already (inter) or is made block by block (intra): residual quadtree */
static void analyse_residual(Enc* e, HvCu* cu) {
  HevcOracle* d = e->d;
  const HvSps* s = d->cur_sps;
  TtEnc t;
  memset(&t, 0, sizeof(t));
  t.e = e;
  t.cu = cu;
  t.max_depth = cu->intra ? s->max_th_depth_intra + (cu->part_mode == HV_PART_NxN) : s->max_th_depth_inter;
  const int node0 = d->num_nodes, tu0 = d->num_tu, coef0 = d->num_coefs;
  cu->first_node = node0;
  cu->first_tu = tu0;
  cu->num_nodes = cu->num_tu = 0;
  tt_analyse(&t, cu->x, cu->y, cu->x, cu->y, cu->log2, 0, 0);
  /* an intra unit always carries its tree; an inter unit with nothing coded carries none (rqt_root_cbf = 0) */
  if (!cu->intra) {
    cu->rqt_root_cbf = (uint8_t)t.any_cbf;
    if (!t.any_cbf) {
      d->num_nodes = node0;
      d->num_tu = tu0;
      d->num_coefs = coef0;
      cu->num_nodes = cu->num_tu = 0;
    } else if (cu->part_mode == HV_PART_2Nx2N && cu->pu[0].merge_flag) {
      /* (merge 2Nx2N with a residual: rqt_root_cbf is inferred to be one) */
    }
  } else {
    cu->rqt_root_cbf = 1;
  }
  /* cbf_luma of an inter unit's undivided tree without chroma is inferred to be one: such a tree must have levels */
  if (!cu->intra && t.any_cbf) {
    HvNode* root = &d->nodes[node0];
    if (!root->split && !root->cbf_cb && !root->cbf_cr && !root->cbf_luma) cu->rqt_root_cbf = 0; /* (cannot happen: any_cbf) */
  }
}

/* ---- intra unit ---- */
static void set_intra_syntax(HevcOracle* d, HvCu* cu, int i, int x, int y, int pb, int mode) {
  int cand[3];
  hv_mpm_candidates(d, x, y, cand);
  cu->prev_intra_luma_pred_flag[i] = 0;
  for (int k = 0; k < 3; ++k)
    if (cand[k] == mode) {
      cu->prev_intra_luma_pred_flag[i] = 1;
      cu->mpm_idx[i] = (uint8_t)k;
    }
  if (!cu->prev_intra_luma_pred_flag[i]) {
    int rem = mode;
    for (int k = 0; k < 3; ++k) rem -= cand[k] < mode;
    cu->rem_intra_luma_pred_mode[i] = (uint8_t)rem;
  }
  cu->luma_mode[i] = (uint8_t)mode;
  for (int yy = 0; yy < pb; yy += 4)
    for (int xx = 0; xx < pb; xx += 4) d->m_imode[((y + yy) >> 2) * d->w4 + ((x + xx) >> 2)] = (uint8_t)mode;
}

static void analyse_intra(Enc* e, HvCu* cu) {
  HevcOracle* d = e->d;
  const HvSps* s = d->cur_sps;
  const Cfg* c = &e->c;
  const uint32_t h = hash3(e->h, (uint32_t)cu->x * 3 + 1, (uint32_t)cu->y);
  cu->intra = 1;
  cu->skip = 0;
  cu->num_pu = 0;
  cu->part_mode = HV_PART_2Nx2N;
  if (cu->log2 == s->log2_min_cb && cu->log2 > s->log2_min_tb && h % 3 == 0) cu->part_mode = HV_PART_NxN;
  cu->pcm = (uint8_t)(c->pcm && cu->part_mode == HV_PART_2Nx2N && cu->log2 >= s->log2_min_pcm && cu->log2 <= s->log2_max_pcm && (h >> 4) % 11 == 0);
  hv_set_cu_maps(d, cu);
  if (cu->pcm) {
    const int n = 1 << cu->log2;
    cu->pcm_samples = hv_new_coefs(d, n * n * 3 / 2);
    int16_t* smp = &d->coefs[cu->pcm_samples];
    for (int cc = 0; cc < 3; ++cc) {
      const int sub = cc ? 1 : 0, m = n >> sub, stride = e->cw >> sub, bits = cc ? s->pcm_bits_chroma : s->pcm_bits_luma;
      const uint8_t* src = e->scene.plane[cc] + (cu->y >> sub) * stride + (cu->x >> sub);
      for (int j = 0; j < m; ++j)
        for (int i = 0; i < m; ++i) *smp++ = (int16_t)(src[j * stride + i] >> (8 - bits));
    }
    for (int i = 0; i < 4; ++i) cu->luma_mode[i] = 1;
    cu->chroma_mode = 1;
    cu->rqt_root_cbf = 0;
    cu->first_node = d->num_nodes;
    cu->first_tu = d->num_tu;
    hv_recon_cu(d, cu);
    return;
  }
  const int parts = cu->part_mode == HV_PART_NxN ? 4 : 1, pb = (1 << cu->log2) >> (parts == 4);
  const int plog2 = cu->log2 - (parts == 4);
  for (int i = 0; i < parts; ++i) {
    const int x = cu->x + (i & 1) * pb, y = cu->y + (i >> 1) * pb;
    int best = 1;
    if (c->random_intra) {
      best = (int)(hash3(h, (uint32_t)i, 9) % 35);
    } else {
      /* best match of a prediction of the whole block (the blocks of a divided tree predict one by one later) */
      int best_sad = 1 << 30;
      const int tl = hv_min(plog2, s->log2_max_tb);
      for (int m = 0; m < 35; ++m) {
        if (pb > 16 && m > 1 && (m & 1) && m != 10 && m != 26) continue; /* (every other angle for the big ones: speed) */
        int sad = 0;
        for (int yy = 0; yy < pb; yy += 1 << tl)
          for (int xx = 0; xx < pb; xx += 1 << tl) hv_intra_predict(d, 0, x + xx, y + yy, tl, m);
        sad = sad_plane(e, 0, x, y, pb, pb) + (m > 1 ? 2 * pb : 0);
        if (sad < best_sad) { best_sad = sad; best = m; }
      }
    }
    set_intra_syntax(d, cu, i, x, y, pb, best);
    if (parts == 4) { /* the next block's candidates see this one's samples only roughly; its mode exactly */
    }
  }
  if (parts == 1) cu->luma_mode[1] = cu->luma_mode[2] = cu->luma_mode[3] = cu->luma_mode[0];
  cu->intra_chroma_pred_mode = (uint8_t)((h >> 9) % 5 == 0 ? (h >> 12) % 4 : 4);
  if (c->random_intra) cu->intra_chroma_pred_mode = (uint8_t)((h >> 9) % 5);
  cu->chroma_mode = (uint8_t)hv_chroma_mode_from_syntax(cu->intra_chroma_pred_mode, cu->luma_mode[0]);
  analyse_residual(e, cu);
}

/* ---- inter unit ---- */
static int pu_luma_sad(Enc* e, const HvPu* pu, const HvMv* mv) {
  HevcOracle* d = e->d;
  if (mv->pred_flag != 3) { /* one list, whole-sample vector: no interpolation to do */
    const int l = (mv->pred_flag & 1) ? 0 : 1;
    if (!((mv->mv[l][0] | mv->mv[l][1]) & 3)) {
      const uint8_t* ref = d->ref_list[l][mv->ref_idx[l]]->plane[0];
      const uint8_t* src = e->scene.plane[0];
      const int dx = mv->mv[l][0] >> 2, dy = mv->mv[l][1] >> 2;
      int sad = 0;
      for (int y = pu->y; y < pu->y + pu->h; ++y) {
        const uint8_t* row = ref + hv_clip3(0, e->ch - 1, y + dy) * e->cw;
        for (int x = pu->x; x < pu->x + pu->w; ++x) sad += hv_abs(src[y * e->cw + x] - row[hv_clip3(0, e->cw - 1, x + dx)]);
      }
      return sad;
    }
  }
  static int16_t t[2][64 * 64];
  const int n = pu->w * pu->h;
  for (int l = 0; l < 2; ++l)
    if (mv->pred_flag & (1 << l)) hv_interp_block(d->ref_list[l][mv->ref_idx[l]]->plane[0], e->cw, e->ch, 0, pu->x, pu->y, pu->w, pu->h, mv->mv[l][0], mv->mv[l][1], t[l]);
  const uint8_t* src = e->scene.plane[0] + pu->y * e->cw + pu->x;
  int sad = 0;
  const int l0 = (mv->pred_flag & 1) ? 0 : 1;
  for (int i = 0; i < n; ++i) {
    const int v = mv->pred_flag == 3 ? (t[0][i] + t[1][i] + 64) >> 7 : (t[l0][i] + 32) >> 6;
    sad += hv_abs(src[(i / pu->w) * e->cw + (i % pu->w)] - hv_clip8(v));
  }
  return sad;
}

static int search_list(Enc* e, const HvPu* pu, int l, int r, const int16_t start[2][2], int16_t best_mv[2]) {
  HvMv m;
  memset(&m, 0, sizeof(m));
  m.pred_flag = (uint8_t)(1 << l);
  m.ref_idx[l] = (int8_t)r;
  int best = 1 << 30;
  int16_t cands[4][2] = {{0, 0}, {start[0][0], start[0][1]}, {start[1][0], start[1][1]}, {0, 0}};
  const int dist = e->d->cur->poc - e->d->ref_list[l][r]->poc;
  cands[3][0] = (int16_t)(-7 * dist); /* the pan */
  cands[3][1] = (int16_t)(-3 * dist);
  for (int k = 0; k < 4; ++k) {
    m.mv[l][0] = (int16_t)(cands[k][0] & ~3);
    m.mv[l][1] = (int16_t)(cands[k][1] & ~3);
    const int sad = pu_luma_sad(e, pu, &m);
    if (sad < best) { best = sad; best_mv[0] = m.mv[l][0]; best_mv[1] = m.mv[l][1]; }
  }
  for (int step = 8; step >= 1; step >>= 1) { /* integer diamond down to quarter samples */
    int improved = 1;
    for (int it = 0; it < (step >= 4 ? 3 : 1) && improved; ++it) {
      improved = 0;
      const int cx = best_mv[0], cy = best_mv[1];
      for (int k = 0; k < 8; ++k) {
        static const int8_t dx[8] = {1, -1, 0, 0, 1, -1, 1, -1}, dy[8] = {0, 0, 1, -1, 1, 1, -1, -1};
        m.mv[l][0] = (int16_t)(cx + dx[k] * step);
        m.mv[l][1] = (int16_t)(cy + dy[k] * step);
        const int sad = pu_luma_sad(e, pu, &m);
        if (sad < best) { best = sad; best_mv[0] = m.mv[l][0]; best_mv[1] = m.mv[l][1]; improved = 1; }
      }
    }
  }
  return best;
}

static int mvd_cost(const int16_t a[2], const int16_t b[2]) { return hv_abs(a[0] - b[0]) + hv_abs(a[1] - b[1]); }

static void analyse_pu(Enc* e, HvCu* cu, int pi) {
  HevcOracle* d = e->d;
  const HvSliceHeader* sh = &d->sh;
  const Cfg* c = &e->c;
  HvPu* pu = &cu->pu[pi];
  memset(pu, 0, sizeof(*pu));
  hv_pu_geometry(cu, pi, &pu->x, &pu->y, &pu->w, &pu->h);
  const uint32_t h = hash3(e->h, (uint32_t)pu->x * 5 + 2, (uint32_t)pu->y * 3 + (uint32_t)pi);
  HvMv mc[5];
  int nm;
  hv_merge_candidates(d, cu, pu->x, pu->y, pu->w, pu->h, pi, mc, &nm);
  int best_merge = -1, best_merge_sad = 1 << 30;
  for (int k = 0; k < nm; ++k) {
    HvMv m = mc[k];
    if (m.pred_flag == 3 && pu->w + pu->h == 12) { m.pred_flag = 1; m.ref_idx[1] = -1; }
    const int sad = pu_luma_sad(e, pu, &m);
    if (sad < best_merge_sad) { best_merge_sad = sad; best_merge = k; }
  }
  /* motion search per list (the first two references of each) */
  int best_sad[2] = {1 << 30, 1 << 30}, best_ref[2] = {0, 0};
  int16_t best_mv[2][2] = {{0, 0}, {0, 0}};
  for (int l = 0; l < (sh->slice_type == HV_SLICE_B ? 2 : 1); ++l)
    for (int r = 0; r < hv_min(sh->num_ref_idx[l], (c->long_term ? 16 : 2) + c->cover); ++r) {
      int16_t cand[2][2], mv[2];
      hv_amvp_candidates(d, cu, pu->x, pu->y, pu->w, pu->h, pi, l, r, cand);
      int sad = search_list(e, pu, l, r, cand, mv) + 4 * r;
      /* --long-term: a share of the blocks is pushed to the list's last entry - the long-term picture when one is listed -
         whatever it costs, so that unscaled predictors and the long / short-term candidate rules are exercised */
      if (c->long_term && r == sh->num_ref_idx[l] - 1 && (hash3(h, 0x4c54, (uint32_t)l) % 3) == 0) sad -= 1 << 20;
      if (sad < best_sad[l]) { best_sad[l] = sad; best_ref[l] = r; best_mv[l][0] = mv[0]; best_mv[l][1] = mv[1]; }
    }
  int idc = 0, inter_sad = best_sad[0];
  if (sh->slice_type == HV_SLICE_B) {
    if (best_sad[1] < inter_sad) { idc = 1; inter_sad = best_sad[1]; }
    if (pu->w + pu->h != 12) {
      HvMv m;
      memset(&m, 0, sizeof(m));
      m.pred_flag = 3;
      for (int l = 0; l < 2; ++l) { m.ref_idx[l] = (int8_t)best_ref[l]; m.mv[l][0] = best_mv[l][0]; m.mv[l][1] = best_mv[l][1]; }
      const int sad = pu_luma_sad(e, pu, &m);
      if (sad < inter_sad) { idc = 2; inter_sad = sad; }
    }
  }
  if (c->cover && h % 7 == 0) { /* wild vectors now and then: long mvd codes, references far outside the picture */
    best_mv[idc == 1][0] = (int16_t)((int)(h >> 8) % 2000 - 1000);
    best_mv[idc == 1][1] = (int16_t)((int)(h >> 20) % 1200 - 600);
  }
  const int use_merge = best_merge >= 0 && (best_merge_sad <= inter_sad + pu->w * pu->h / 8 || (h >> 4) % 9 == 0) && !(c->cover && h % 5 == 1);
  if (use_merge) {
    pu->merge_flag = 1;
    pu->merge_idx = (uint8_t)best_merge;
  } else {
    pu->merge_flag = 0;
    pu->inter_pred_idc = (uint8_t)idc;
    for (int l = 0; l < 2; ++l) {
      if (idc == (l ? 0 : 1)) continue;
      pu->ref_idx[l] = (int8_t)best_ref[l];
      int16_t cand[2][2];
      hv_amvp_candidates(d, cu, pu->x, pu->y, pu->w, pu->h, pi, l, best_ref[l], cand);
      pu->mvp_flag[l] = (uint8_t)(mvd_cost(best_mv[l], cand[1]) < mvd_cost(best_mv[l], cand[0]));
      if (l == 1 && sh->mvd_l1_zero && idc == 2) {
        pu->mvd[1][0] = pu->mvd[1][1] = 0; /* (the vector becomes the predictor) */
      } else {
        pu->mvd[l][0] = (int16_t)(best_mv[l][0] - cand[pu->mvp_flag[l]][0]);
        pu->mvd[l][1] = (int16_t)(best_mv[l][1] - cand[pu->mvp_flag[l]][1]);
      }
    }
  }
  hv_derive_pu_motion(d, cu, pi); /* from the syntax elements, as a decoder would */
  hv_inter_predict_pu(d, pu);
}

static void analyse_inter(Enc* e, HvCu* cu) {
  HevcOracle* d = e->d;
  const HvSps* s = d->cur_sps;
  const uint32_t h = hash3(e->h, (uint32_t)cu->x * 7 + 3, (uint32_t)cu->y);
  cu->intra = 0;
  cu->pcm = 0;
  cu->part_mode = HV_PART_2Nx2N;
  if (h % 5 == 0) {
    int allowed[8], na = 0;
    allowed[na++] = HV_PART_2NxN;
    allowed[na++] = HV_PART_Nx2N;
    if (cu->log2 == s->log2_min_cb && cu->log2 > 3) allowed[na++] = HV_PART_NxN;
    if (s->amp_enabled && cu->log2 > s->log2_min_cb) {
      allowed[na++] = HV_PART_2NxnU; allowed[na++] = HV_PART_2NxnD; allowed[na++] = HV_PART_nLx2N; allowed[na++] = HV_PART_nRx2N;
    }
    cu->part_mode = (uint8_t)allowed[(h >> 8) % (unsigned)na];
  }
  cu->num_pu = (uint8_t)hv_num_parts(cu->part_mode);
  cu->skip = 0;
  hv_set_cu_maps(d, cu);
  for (int i = 0; i < cu->num_pu; ++i) analyse_pu(e, cu, i);
  analyse_residual(e, cu);
  if (cu->part_mode == HV_PART_2Nx2N && cu->pu[0].merge_flag && !cu->rqt_root_cbf) {
    cu->skip = 1;
    hv_set_cu_maps(d, cu); /* (the skip flags of the neighbours are contexts of the next units' flags) */
  }
}

/* ---- coding quadtree analysis ---- */
static int block_activity(const Enc* e, int x, int y, int n) {
  const uint8_t* p = e->scene.plane[0] + y * e->cw + x;
  int64_t sum = 0;
  for (int j = 0; j < n; j += 2)
    for (int i = 0; i < n; i += 2) sum += p[j * e->cw + i];
  const int mean = (int)(sum / ((n / 2) * (n / 2)));
  int dev = 0;
  for (int j = 0; j < n; j += 2)
    for (int i = 0; i < n; i += 2) dev += hv_abs(p[j * e->cw + i] - mean);
  return dev / ((n / 2) * (n / 2));
}

static int analyse_cu(Enc* e, int x0, int y0, int log2, int depth) {
  HevcOracle* d = e->d;
  const HvPps* pps = d->cur_pps;
  const HvSliceHeader* sh = &d->sh;
  const Cfg* c = &e->c;
  HvCu* cu = hv_new_cu(d);
  if (!cu) return -1;
  cu->x = (uint16_t)x0;
  cu->y = (uint16_t)y0;
  cu->log2 = (uint8_t)log2;
  cu->depth = (uint8_t)depth;
  const uint32_t h = hash3(e->h, (uint32_t)x0 * 11 + 5, (uint32_t)y0);
  cu->tq_bypass = (uint8_t)(pps->transquant_bypass_enabled && h % 13 == 0);
  /* luma QP: what the quantisation group predicts, unless this unit is the one that codes the group's difference */
  int want = sh->slice_qp;
  const int pred = pps->cu_qp_delta_enabled ? hv_qp_y_pred(d, d->qg_x, d->qg_y) : sh->slice_qp;
  if (pps->cu_qp_delta_enabled) {
    want = d->is_cu_qp_delta_coded ? (pred + d->cu_qp_delta_val + 52) % 52
                                   : hv_clip3(0, 51, sh->slice_qp + (int)(hash3(e->h, (uint32_t)d->qg_x + 77, (uint32_t)d->qg_y) % 13) - 6 +
                                                         ((h >> 20) % 31 == 0 ? ((h >> 25) & 1 ? 25 : -25) : 0));
  }
  cu->qp_y = (int8_t)want;
  int intra = sh->slice_type == HV_SLICE_I;
  if (!intra) intra = (h >> 6) % (c->cover ? 5 : 12) == 0;
  if (intra)
    analyse_intra(e, cu);
  else
    analyse_inter(e, cu);
  /* did anything of it carry a coded block flag?  then the group's difference is (or was) coded */
  int any = 0;
  for (int t = 0; t < cu->num_tu; ++t) {
    const HvTu* tu = &d->tus[cu->first_tu + t];
    any |= tu->cbf[0] | tu->cbf[1] | tu->cbf[2];
  }
  if (pps->cu_qp_delta_enabled) {
    if (any && !d->is_cu_qp_delta_coded) {
      int delta = want - pred;
      if (delta > 25) delta -= 52;
      if (delta < -26) delta += 52;
      cu->qp_delta = (int8_t)delta;
      cu->qp_delta_coded = 1;
      d->is_cu_qp_delta_coded = 1;
      d->cu_qp_delta_val = delta;
    } else if (!any && !d->is_cu_qp_delta_coded) {
      cu->qp_y = (int8_t)pred; /* nothing coded: the unit has the predicted QP (it only matters to the deblocking filter) */
    }
  }
  const int n = 1 << log2;
  for (int y = y0; y < y0 + n; y += 4)
    for (int x = x0; x < x0 + n; x += 4) d->m_qp[(y >> 2) * d->w4 + (x >> 2)] = cu->qp_y;
  d->last_cu_qp = cu->qp_y;
  if (!cu->pcm) hv_mark_edges(d, cu); /* (an I_PCM unit went through hv_recon_cu, which marks them) */
  return 0;
}

static int analyse_quadtree(Enc* e, int x0, int y0, int log2, int depth) {
  HevcOracle* d = e->d;
  const HvSps* s = d->cur_sps;
  const HvPps* pps = d->cur_pps;
  const int n = 1 << log2;
  int split;
  if (x0 + n <= s->width && y0 + n <= s->height && log2 > s->log2_min_cb) {
    const uint32_t h = hash3(e->h, (uint32_t)x0 * 13 + (uint32_t)log2, (uint32_t)y0);
    const int act = block_activity(e, x0, y0, n);
    split = act > (log2 == 6 ? 6 : log2 == 5 ? 10 : 16);
    if (h % 8 == 0) split = !split;
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
    const int hh = n >> 1;
    /* the walk asks the depth map whether a block is divided: mark the whole block deeper before the children fill it in */
    for (int i = 0; i < 4; ++i) {
      const int x = x0 + (i & 1) * hh, y = y0 + (i >> 1) * hh;
      if (x < s->width && y < s->height)
        if (analyse_quadtree(e, x, y, log2 - 1, depth + 1)) return -1;
    }
    return 0;
  }
  return analyse_cu(e, x0, y0, log2, depth);
}

/* ---- SAO parameters per coding tree block, after deblocking ---- */
static void decide_sao(Enc* e) {
  HevcOracle* d = e->d;
  const HvSps* s = d->cur_sps;
  for (int addr = 0; addr < s->width_ctb * s->height_ctb; ++addr) {
    HvSao* p = &d->sao[addr];
    memset(p, 0, sizeof(*p));
    const int rx = addr % s->width_ctb, ry = addr / s->width_ctb;
    const HvSliceFacts* f = &d->slice_facts[d->m_slice[addr]];
    const uint32_t h = hash3(e->h, 0x5a0 + (uint32_t)addr, 1);
    const int left_ok = rx > 0 && addr - 1 >= f->slice_addr_rs, up_ok = ry > 0 && addr - s->width_ctb >= f->slice_addr_rs;
    if (left_ok && h % 5 == 0) {
      *p = d->sao[addr - 1];
      p->merge_left = 1;
      p->merge_up = 0;
      continue;
    }
    if (up_ok && h % 5 == 1) {
      *p = d->sao[addr - s->width_ctb];
      p->merge_left = 0;
      p->merge_up = 1;
      continue;
    }
    for (int c = 0; c < 3; ++c) {
      if (!(c ? f->sao_chroma : f->sao_luma)) continue;
      const uint32_t hc = hash3(h, (uint32_t)c, 3);
      if (c != 2) {
        p->type[c] = (uint8_t)(hc % 3);
        p->eo_class[c] = (uint8_t)((hc >> 4) % 4);
      } else {
        p->type[2] = p->type[1];
        p->eo_class[2] = p->eo_class[1];
      }
      if (!p->type[c]) continue;
      p->band_pos[c] = (uint8_t)((hc >> 8) % 32);
      /* offsets from what the deblocked block is off by, category by category */
      const int sub = c ? 1 : 0, cs = s->ctb_size >> sub, pw = s->width >> sub, ph = s->height >> sub;
      const uint8_t* rec = d->cur->plane[c];
      const uint8_t* src = e->scene.plane[c];
      if (p->type[c] == 1) { /* the four bands around the block's mean */
        int64_t sum = 0, cnt = 0;
        for (int y = ry * cs; y < hv_min((ry + 1) * cs, ph); ++y)
          for (int x = rx * cs; x < hv_min((rx + 1) * cs, pw); ++x) { sum += rec[y * pw + x]; ++cnt; }
        p->band_pos[c] = (uint8_t)(((int)(sum / cnt) >> 3) + 30 + (int)((hc >> 13) % 3)) & 31;
      }
      int64_t acc[5] = {0}, num[5] = {0};
      static const int8_t hpos[4][2] = {{-1, 1}, {0, 0}, {-1, 1}, {1, -1}}, vpos[4][2] = {{0, 0}, {-1, 1}, {-1, 1}, {-1, 1}};
      for (int y = ry * cs; y < hv_min((ry + 1) * cs, ph); ++y)
        for (int x = rx * cs; x < hv_min((rx + 1) * cs, pw); ++x) {
          const int v = rec[y * pw + x];
          int idx = 0;
          if (p->type[c] == 1) {
            const int k = ((v >> 3) - p->band_pos[c]) & 31;
            idx = k < 4 ? k + 1 : 0;
          } else {
            const int cl = p->eo_class[c];
            const int xa = x + hpos[cl][0], ya = y + vpos[cl][0], xb = x + hpos[cl][1], yb = y + vpos[cl][1];
            if (xa < 0 || ya < 0 || xb < 0 || yb < 0 || xa >= pw || xb >= pw || ya >= ph || yb >= ph) continue;
            const int sum = 2 + hv_sign(v - rec[ya * pw + xa]) + hv_sign(v - rec[yb * pw + xb]);
            idx = sum == 2 ? 0 : sum < 2 ? sum + 1 : sum;
          }
          if (idx) { acc[idx] += src[y * pw + x] - v; ++num[idx]; }
        }
      for (int k = 0; k < 4; ++k) {
        int o = num[k + 1] ? (int)lround((double)acc[k + 1] / (double)num[k + 1]) : 0;
        if (e->c.sao & 2) o += (int)(hash3(hc, (uint32_t)k, 5) % 5) - 2; /* --sao 3: noisy offsets, the full range */
        o = hv_clip3(-7, 7, o);
        if (p->type[c] == 2) o = k < 2 ? hv_max(o, 0) : hv_min(o, 0);
        p->offset[c][k] = (int8_t)o;
      }
    }
  }
}

/* ---- one picture ---- */
static int encode_picture(Enc* e, const PicPlan* pp, Buf* sample) {
  const Cfg* c = &e->c;
  HevcOracle* d = e->d;
  e->h = hash3((uint32_t)c->seed, e->pic_counter++, 0xabc);
  if (pp->idr) e->n_refset = 0;
  if (pp->cra) e->purge_below = pp->disp;
  else if (!pp->rasl && e->purge_below >= 0) { /* the first trailing picture: nothing in front of the CRA picture is kept any longer */
    int m = 0;
    for (int i = 0; i < e->n_refset; ++i)
      if (e->refset[i] >= e->purge_below) e->refset[m++] = e->refset[i];
    e->n_refset = m;
    e->purge_below = -1;
  }
  HvRps rps;
  rps_for(e, pp->disp, &rps);
  int nl0 = 0, nl1 = 0;
  if (pp->type != HV_SLICE_I) {
    const int total = rps_used(&rps) + lt_wanted(e, pp);
    nl0 = hv_min(hv_max(total, 1), hv_min(c->refs + (c->long_term ? 1 : 0), 15));
    nl1 = nl0;
    if (c->cover && total > 1 && (e->h & 3) == 0) nl0 = nl1 = hv_min(total + 1, 15); /* lists longer than the set: entries repeat */
  }
  scene_render(&e->scene, (int)(e->pic_counter - 1) + 0);
  const int nal_type = pp->idr ? 19 : pp->cra ? 21 : pp->rasl ? 8 : pp->radl ? 6 : pp->is_ref ? 1 : 0;
  const HvSps* sps = &d->sps[0];
  const int total_ctb = sps->width_ctb * sps->height_ctb;
  e->n_slices = hv_min(hv_max(c->slices, 1), total_ctb);
  for (int k = 0; k <= e->n_slices; ++k) e->slice_start[k] = (int)((int64_t)total_ctb * k / e->n_slices);
  Buf hdr_rbsp[64];
  memset(hdr_rbsp, 0, sizeof(hdr_rbsp));
  /* analysis, slice by slice */
  for (int k = 0; k < e->n_slices; ++k) {
    BitW w = {0};
    write_slice_header(e, &w, pp, e->slice_start[k], &rps, nl0, nl1, 0, NULL);
    bput(&hdr_rbsp[k], w.b, w.bits / 8);
    HvBits b = {w.b, w.bits / 8, 0, 0};
    HvSliceHeader sh;
    if (hv_parse_slice_header(d, &b, nal_type, &sh)) return fprintf(stderr, "own slice header does not parse: %s\n", d->error), -1;
    free(w.b);
    d->sh = sh;
    d->sh.rps = sh.st_rps_sps_flag || !sh.rps ? sh.rps : &d->sh.st_rps_local;
    if (k == 0) {
      if (hv_start_picture(d, nal_type, 0) || hv_begin_picture_state(d)) return fprintf(stderr, "%s\n", d->error), -1;
    }
    if (hv_build_ref_lists(d)) return fprintf(stderr, "%s\n", d->error), -1;
    e->slice_sh[k] = d->sh;
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
    d->last_cu_qp = d->sh.slice_qp;
    d->qp_y_prev = d->sh.slice_qp;
    d->is_cu_qp_delta_coded = 0;
    d->cu_qp_delta_val = 0;
    for (int addr = e->slice_start[k]; addr < e->slice_start[k + 1]; ++addr) {
      const int rx = addr % sps->width_ctb, ry = addr / sps->width_ctb;
      if (c->wpp && rx == 0 && addr != e->slice_start[k]) d->last_cu_qp = d->sh.slice_qp;
      d->m_slice[addr] = (uint16_t)d->slice_idx;
      if (analyse_quadtree(e, rx << sps->log2_ctb, ry << sps->log2_ctb, sps->log2_ctb, 0)) return fprintf(stderr, "analysis failed\n"), -1;
    }
  }
  d->recon_next = d->num_cu;
  hv_deblock_picture(d);
  if (c->sao) {
    decide_sao(e);
    hv_sao_picture(d);
  }
  /* syntax, slice by slice, through the oracle's walk */
  if (c->extra_nals) {
    Buf nal = {0};
    const uint8_t aud[1] = {(uint8_t)((pp->type == HV_SLICE_I ? 0 : pp->type == HV_SLICE_P ? 1 : 2) << 5 | 0x10)};
    nal_append(&nal, 35, aud, 1);
    sample_add_nal(sample, c->nal_length_size, &nal);
    if (pp->idr) {
      sample_add_nal(sample, c->nal_length_size, &e->vps);
      sample_add_nal(sample, c->nal_length_size, &e->sps);
      sample_add_nal(sample, c->nal_length_size, &e->pps);
      if (e->pps1.n) sample_add_nal(sample, c->nal_length_size, &e->pps1);
    }
    free(nal.b);
  }
  d->walk_cu = 0;
  for (int k = 0; k < e->n_slices; ++k) {
    d->sh = e->slice_sh[k];
    d->sh.rps = NULL;
    d->slice_idx = k;
    d->enc = 1;
    d->enc_end_ctb = e->slice_start[k + 1];
    d->enc_nsub = 0;
    hv_cabac_enc_start(&d->ce);
    const int rc = hv_slice_data(d);
    d->enc = 0;
    if (rc) return fprintf(stderr, "slice data walk failed: %s\n", d->error), -1;
    hv_cabac_enc_finish_slice(&d->ce);
    Buf rbsp = {0}, nal = {0};
    if (c->wpp && d->enc_nsub > 0) {
      /* the header once more, now with the entry points: a substream's size in the NAL unit, i.e. with the emulation
         prevention bytes that fall into it (the header ends on its alignment one bit: the count of zeros starts afresh) */
      uint32_t entry[256];
      const size_t nbytes = d->ce.nbits / 8;
      size_t start = 0;
      int zeros = 0, sub = 0;
      uint32_t size = 0;
      for (size_t i = 0; i <= nbytes; ++i) {
        if (sub < d->enc_nsub && i == d->enc_sub_start[sub]) { entry[sub++] = size; size = 0; start = i; }
        if (i == nbytes) break;
        if (zeros >= 2 && d->ce.buf[i] <= 3) { ++size; zeros = 0; }
        ++size;
        zeros = d->ce.buf[i] == 0 ? zeros + 1 : 0;
      }
      (void)start;
      BitW w2 = {0};
      write_slice_header(e, &w2, pp, e->slice_start[k], &rps, nl0, nl1, d->enc_nsub, entry);
      hdr_rbsp[k].n = 0;
      bput(&hdr_rbsp[k], w2.b, w2.bits / 8);
      free(w2.b);
    }
    bput(&rbsp, hdr_rbsp[k].b, hdr_rbsp[k].n);
    bput(&rbsp, d->ce.buf, d->ce.nbits / 8);
    g_nal_tid_plus1 = (c->sub_layers == 2 && !pp->is_ref && !pp->idr && !pp->rasl && !pp->radl) ? 2 : 1;
    nal_append(&nal, (c->sub_layers == 2 && !pp->is_ref && !pp->idr && !pp->rasl && !pp->radl) ? 2 /* TSA_N */ : nal_type, rbsp.b, rbsp.n);
    g_nal_tid_plus1 = 1;
    sample_add_nal(sample, c->nal_length_size, &nal);
    free(rbsp.b);
    free(nal.b);
    free(hdr_rbsp[k].b);
  }
  if (d->walk_cu != d->num_cu) return fprintf(stderr, "the walk wrote %d of %d coding units\n", d->walk_cu, d->num_cu), -1;
  { /* decoded picture hash, suffix SEI */
    uint8_t md5[3][16];
    hv_picture_md5(d, d->cur, md5);
    uint8_t sei[2 + 1 + 48 + 1];
    sei[0] = 132;
    sei[1] = 49;
    sei[2] = 0;
    memcpy(sei + 3, md5, 48);
    sei[51] = 0x80;
    Buf nal = {0};
    nal_append(&nal, 40, sei, sizeof(sei));
    sample_add_nal(sample, c->nal_length_size, &nal);
    free(nal.b);
  }
  ++d->counters[3];
  hv_finish_picture(d);
  if (!pp->is_ref) {
    /* (hv_finish_picture marks every picture as a reference; the next picture's reference picture set drops it) */
  } else {
    refset_add(e, pp->disp);
  }
  return 0;
}

/* ------------------------------------------------------------------------------------------------------- muxer */
static void write_mp4(const Enc* e, const Buf* samples, const uint8_t* keys, const int32_t* cts, size_t n) {
  const Cfg* c = &e->c;
  const uint32_t ts = (uint32_t)c->fps * 1000, delta = 1000;
  Buf f = {0};
  size_t a = box_begin(&f, "ftyp");
  bput(&f, "isom", 4); b32(&f, 512); bput(&f, "isomiso2mp41", 12);
  box_end(&f, a);
  /* moov first: its size does not depend on the offsets */
  for (int pass = 0; pass < 2; ++pass) {
    const size_t moov_at = f.n;
    const uint64_t payload = pass ? f.n : 0;
    (void)payload;
    size_t moov = box_begin(&f, "moov");
    a = box_begin(&f, "mvhd");
    b32(&f, 0); b32(&f, 0); b32(&f, 0); b32(&f, ts); b32(&f, (uint32_t)(n * delta));
    b32(&f, 0x00010000); b16(&f, 0x0100); bzero(&f, 10);
    const uint32_t mat[9] = {0x10000, 0, 0, 0, 0x10000, 0, 0, 0, 0x40000000};
    for (int i = 0; i < 9; ++i) b32(&f, mat[i]);
    bzero(&f, 24); b32(&f, 2);
    box_end(&f, a);
    size_t trak = box_begin(&f, "trak");
    a = box_begin(&f, "tkhd");
    b32(&f, 3); b32(&f, 0); b32(&f, 0); b32(&f, 1); b32(&f, 0); b32(&f, (uint32_t)(n * delta));
    bzero(&f, 8); b16(&f, 0); b16(&f, 0); b16(&f, 0); b16(&f, 0);
    for (int i = 0; i < 9; ++i) b32(&f, mat[i]);
    b32(&f, (uint32_t)c->width << 16); b32(&f, (uint32_t)c->height << 16);
    box_end(&f, a);
    size_t mdia = box_begin(&f, "mdia");
    a = box_begin(&f, "mdhd");
    b32(&f, 0); b32(&f, 0); b32(&f, 0); b32(&f, ts); b32(&f, (uint32_t)(n * delta)); b16(&f, 0x55c4); b16(&f, 0);
    box_end(&f, a);
    a = box_begin(&f, "hdlr");
    b32(&f, 0); b32(&f, 0); bput(&f, "vide", 4); bzero(&f, 12); bput(&f, "VideoHandler", 13);
    box_end(&f, a);
    size_t minf = box_begin(&f, "minf");
    a = box_begin(&f, "vmhd");
    b32(&f, 1); bzero(&f, 8);
    box_end(&f, a);
    a = box_begin(&f, "dinf");
    size_t dref = box_begin(&f, "dref");
    b32(&f, 0); b32(&f, 1);
    size_t url = box_begin(&f, "url ");
    b32(&f, 1);
    box_end(&f, url);
    box_end(&f, dref);
    box_end(&f, a);
    size_t stbl = box_begin(&f, "stbl");
    a = box_begin(&f, "stsd");
    b32(&f, 0); b32(&f, 1);
    size_t v = box_begin(&f, c->hvc1 ? "hvc1" : "hev1");
    bzero(&f, 6); b16(&f, 1); bzero(&f, 16);
    b16(&f, (uint32_t)c->width); b16(&f, (uint32_t)c->height);
    b32(&f, 0x00480000); b32(&f, 0x00480000); b32(&f, 0); b16(&f, 1); bzero(&f, 32); b16(&f, 0x18); b16(&f, 0xffff);
    size_t cc = box_begin(&f, "hvcC");
    uint8_t hvcc[2048];
    bput(&f, hvcc, build_hvcc(e, hvcc));
    box_end(&f, cc);
    box_end(&f, v);
    box_end(&f, a);
    a = box_begin(&f, "stts");
    b32(&f, 0); b32(&f, 1); b32(&f, (uint32_t)n); b32(&f, delta);
    box_end(&f, a);
    int reorder = 0;
    for (size_t i = 0; i < n; ++i) reorder |= cts[i] != 0;
    if (reorder) {
      a = box_begin(&f, "ctts");
      b32(&f, 0); b32(&f, (uint32_t)n);
      for (size_t i = 0; i < n; ++i) { b32(&f, 1); b32(&f, (uint32_t)(cts[i] * (int32_t)delta)); }
      box_end(&f, a);
    }
    uint32_t nk = 0;
    for (size_t i = 0; i < n; ++i) nk += keys[i];
    a = box_begin(&f, "stss");
    b32(&f, 0); b32(&f, nk);
    for (size_t i = 0; i < n; ++i)
      if (keys[i]) b32(&f, (uint32_t)i + 1);
    box_end(&f, a);
    a = box_begin(&f, "stsc");
    b32(&f, 0); b32(&f, 1); b32(&f, 1); b32(&f, (uint32_t)n); b32(&f, 1);
    box_end(&f, a);
    a = box_begin(&f, "stsz");
    b32(&f, 0); b32(&f, 0); b32(&f, (uint32_t)n);
    for (size_t i = 0; i < n; ++i) b32(&f, (uint32_t)samples[i].n);
    box_end(&f, a);
    a = box_begin(&f, "stco");
    b32(&f, 0); b32(&f, 1);
    static size_t moov_size;
    b32(&f, (uint32_t)(moov_at + moov_size + 8)); /* first pass: a placeholder of the right width */
    box_end(&f, a);
    box_end(&f, stbl);
    box_end(&f, minf);
    box_end(&f, mdia);
    box_end(&f, trak);
    box_end(&f, moov);
    if (!pass) {
      moov_size = f.n - moov_at;
      f.n = moov_at; /* again, now that the offset of the payload is known */
    }
  }
  uint64_t total = 8;
  for (size_t i = 0; i < n; ++i) total += samples[i].n;
  b32(&f, (uint32_t)total);
  bput(&f, "mdat", 4);
  for (size_t i = 0; i < n; ++i) bput(&f, samples[i].b, samples[i].n);
  FILE* fp = fopen(c->out, "wb");
  if (!fp) { perror(c->out); exit(1); }
  fwrite(f.b, 1, f.n, fp);
  fclose(fp);
  free(f.b);
}

/* ---------------------------------------------------------------------------------------------------------- main */
static int arg_int(int argc, char** argv, const char* name, int def) {
  for (int i = 1; i + 1 < argc; ++i)
    if (!strcmp(argv[i], name)) return atoi(argv[i + 1]);
  return def;
}
static const char* arg_str(int argc, char** argv, const char* name, const char* def) {
  for (int i = 1; i + 1 < argc; ++i)
    if (!strcmp(argv[i], name)) return argv[i + 1];
  return def;
}

static void usage(void) {
  fprintf(stderr,
          "hevc_encoder --width W --height H --frames N --out clip.mp4 [--yuv-out f] [--gop 30] [--bframes 0] [--refs 2]\n"
          "  [--qp 30] [--seed 1] [--fps 30] [--ctb 16|32|64] [--min-tb 4] [--max-tb 32] [--th-intra 1] [--th-inter 1]\n"
          "  [--amp 0] [--sao 0|1|3 (+4: luma off, +8: chroma off)] [--pcm 0|1|2] [--tskip 0] [--bypass 0] [--sdh 0]\n"
          "  [--qp-swing 0|depth+1] [--scaling 0..3] [--wp 0] [--tmvp 1] [--strong 1] [--merge-cand 5] [--pml 2]\n"
          "  [--slices 1] [--wpp 0] [--deblock-off 0|1|2] [--beta 0] [--tc 0] [--cip 0] [--rps-sps 0] [--poc-bits 8] [--long-term 0|1|2|3] [--inter-rps 0] [--rplm 0] [--keep-unused 0] [--sub-layers 1|2] [--sh-extras 0] [--multi-pps 0]\n"
          "  [--random-intra 0] [--cabac-init 0|1|2] [--cb-off 0] [--cr-off 0] [--slice-cb-off 0] [--nal-length-size 4]\n"
          "  [--extra-nals 0] [--bpyramid 0] [--lf-across 1] [--mvd-l1-zero 0] [--hvc1 0] [--intra-only 0] [--cover 0]\n");
}

int main(int argc, char** argv) {
  Enc* e = (Enc*)calloc(1, sizeof(Enc));
  e->purge_below = -1;
  Cfg* c = &e->c;
#define A(name, flag, def) c->name = arg_int(argc, argv, flag, def)
  A(width, "--width", 0); A(height, "--height", 0); A(frames, "--frames", 0); A(gop, "--gop", 30); A(bframes, "--bframes", 0);
  A(refs, "--refs", 2); A(qp, "--qp", 30); A(seed, "--seed", 1); A(fps, "--fps", 30); A(ctb, "--ctb", 32); A(min_tb, "--min-tb", 4);
  A(max_tb, "--max-tb", 32); A(th_intra, "--th-intra", 1); A(th_inter, "--th-inter", 1); A(amp, "--amp", 0); A(sao, "--sao", 0);
  A(pcm, "--pcm", 0); A(tskip, "--tskip", 0); A(bypass, "--bypass", 0); A(sdh, "--sdh", 0); A(qp_swing, "--qp-swing", 0);
  A(scaling, "--scaling", 0); A(wp, "--wp", 0); A(tmvp, "--tmvp", 1); A(strong, "--strong", 1); A(merge_cand, "--merge-cand", 5);
  A(pml, "--pml", 2); A(slices, "--slices", 1); A(wpp, "--wpp", 0); A(deblock_off, "--deblock-off", 0); A(beta, "--beta", 0);
  A(tc, "--tc", 0); A(cip, "--cip", 0); A(rps_sps, "--rps-sps", 0); A(poc_bits, "--poc-bits", 8); A(random_intra, "--random-intra", 0); A(open_gop, "--open-gop", 0);
  A(cabac_init, "--cabac-init", 0); A(cb_off, "--cb-off", 0); A(cr_off, "--cr-off", 0); A(slice_cb_off, "--slice-cb-off", 0);
  A(nal_length_size, "--nal-length-size", 4); A(extra_nals, "--extra-nals", 0); A(bpyramid, "--bpyramid", 0); A(long_term, "--long-term", 0); A(inter_rps, "--inter-rps", 0); A(rplm, "--rplm", 0); A(keep_unused, "--keep-unused", 0); A(sub_layers, "--sub-layers", 1); A(sh_extras, "--sh-extras", 0); A(multi_pps, "--multi-pps", 0);
  A(lf_across, "--lf-across", 1); A(mvd_l1_zero, "--mvd-l1-zero", 0); A(hvc1, "--hvc1", 0); A(intra_only, "--intra-only", 0);
  A(cover, "--cover", 0);
#undef A
  c->out = arg_str(argc, argv, "--out", NULL);
  c->yuv_out = arg_str(argc, argv, "--yuv-out", NULL);
  if (!c->width || !c->height || !c->frames || !c->out) return usage(), 2;
  if (c->bframes && c->refs < 2) c->refs = 2;
  if (c->bpyramid && c->refs < 3) c->refs = 3;
  if (c->max_tb > c->ctb) c->max_tb = c->ctb;
  if (c->max_tb > 32) c->max_tb = 32;
  c->gop = hv_min(c->gop, (1 << c->poc_bits) * 8);
  e->cw = (c->width + 7) & ~7;
  e->ch = (c->height + 7) & ~7;
  e->d = hevco_open();
  e->scene.W = e->cw; e->scene.H = e->ch; e->scene.dw = c->width; e->scene.dh = c->height;
  e->scene.seed = (uint32_t)c->seed * 2654435761u + 99;
  e->scene.plane[0] = (uint8_t*)malloc((size_t)e->cw * e->ch * 3 / 2);
  e->scene.plane[1] = e->scene.plane[0] + (size_t)e->cw * e->ch;
  e->scene.plane[2] = e->scene.plane[1] + (size_t)e->cw * e->ch / 4;
  /* --rps-sps: the reference picture sets of one GOP's pictures go into the SPS */
  if (c->rps_sps) {
    PicPlan plan[600];
    const int n = plan_gop(c, hv_min(c->gop, c->frames), plan);
    for (int i = 0; i < n && e->num_sps_sets < 12; ++i) {
      if (plan[i].idr) { e->n_refset = 0; refset_add(e, 0); continue; }
      HvRps r;
      rps_for(e, plan[i].disp, &r);
      int known = 0;
      for (int k = 0; k < e->num_sps_sets; ++k) known |= rps_equal(&e->sps_sets[k], &r);
      if (!known && !(c->rps_sps == 2 && (i & 1))) e->sps_sets[e->num_sps_sets++] = r;   /* (2: every other picture's set only - the rest is the slice headers' business) */
      if (plan[i].is_ref) refset_add(e, plan[i].disp);
    }
    e->n_refset = 0;
  }
  write_parameter_sets(e);
  {
    uint8_t hvcc[2048];
    const size_t n = build_hvcc(e, hvcc);
    if (hevco_configure(e->d, hvcc, n)) return fprintf(stderr, "own parameter sets do not parse: %s\n", hevco_error(e->d)), 1;
    e->d->cur_sps = &e->d->sps[0];
    e->d->cur_pps = &e->d->pps[0];
  }
  Buf* samples = (Buf*)calloc((size_t)c->frames, sizeof(Buf));
  uint8_t* keys = (uint8_t*)calloc((size_t)c->frames, 1);
  int32_t* cts = (int32_t*)calloc((size_t)c->frames, sizeof(int32_t));
  FILE* yuv = c->yuv_out ? fopen(c->yuv_out, "wb") : NULL;
  size_t k = 0;
  for (int g0 = 0; g0 < c->frames; g0 += c->gop) {
    PicPlan plan[600];
    const int open = c->open_gop && g0 > 0 && c->bframes > 0 && !c->intra_only;
    const int n = open ? plan_open_gop(c, hv_min(c->gop, c->frames - g0), plan) : plan_gop(c, hv_min(c->gop, c->frames - g0), plan);
    /* (--open-gop: one IDR picture at the start, the picture order count runs on through the GOPs) */
    const int poc_base = c->open_gop == 1 ? g0 : (c->open_gop == 2 && open) ? -hv_min(c->bframes, n - 1) : 0;
    for (int i = 0; i < n; ++i, ++k) {
      /* the scene is rendered at the DISPLAY position */
      const uint32_t keep = e->pic_counter;
      e->pic_counter = (uint32_t)(g0 + plan[i].disp);
      plan[i].disp += poc_base;
      if (c->open_gop == 1 && g0 > 0) plan[i].idr = 0, plan[i].cra = plan[i].cra || (i == 0);
      if (c->open_gop == 2 && open) { /* closed after all: the head is an IDR picture (POC 0), the leading pictures have negative POCs */
        plan[i].idr = plan[i].cra;
        plan[i].cra = 0;
        plan[i].radl = plan[i].rasl;
        plan[i].rasl = 0;
      }
      const int rc = encode_picture(e, &plan[i], &samples[k]);
      plan[i].disp -= poc_base;
      e->pic_counter = keep + 1;
      if (rc) return 1;
      keys[k] = (uint8_t)(plan[i].idr || plan[i].cra);
      cts[k] = g0 + plan[i].disp - (int)k + (c->bframes ? (c->bpyramid ? 2 : 1) : 0);
      while (hevco_frames_ready(e->d)) {
        int w, h, poc;
        hevco_frame_info(e->d, &w, &h, &poc);
        const size_t sz = (size_t)w * h + 2 * (size_t)((w + 1) / 2) * ((h + 1) / 2);
        uint8_t* buf = (uint8_t*)malloc(sz);
        hevco_pop_frame(e->d, buf, sz);
        if (yuv) fwrite(buf, 1, sz, yuv);
        free(buf);
      }
    }
  }
  hevco_flush(e->d);
  while (hevco_frames_ready(e->d)) {
    int w, h, poc;
    hevco_frame_info(e->d, &w, &h, &poc);
    const size_t sz = (size_t)w * h + 2 * (size_t)((w + 1) / 2) * ((h + 1) / 2);
    uint8_t* buf = (uint8_t*)malloc(sz);
    hevco_pop_frame(e->d, buf, sz);
    if (yuv) fwrite(buf, 1, sz, yuv);
    free(buf);
  }
  if (yuv) fclose(yuv);
  write_mp4(e, samples, keys, cts, (size_t)c->frames);
  size_t bytes = 0;
  for (int i = 0; i < c->frames; ++i) bytes += samples[i].n;
  fprintf(stderr, "wrote %s: %d samples, %zu bytes of HEVC\n", c->out, c->frames, bytes);
  return 0;
}
