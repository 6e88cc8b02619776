/* Synthetic-corpus generator: procedural 4:2:0 video -> H.264 (CAVLC or
 * CABAC, I/P/B, 4x4 + 8x8 transform, quarter-sample motion, multi-slice) in
 * an MP4 that hwang's indexer accepts.
 *
 * Why it exists: no encoder (x264 / FFmpeg) and no network are available, and
 * BASELINE.json's configs are all synthetic.  This is
 * INPUT GENERATION (test/bench infrastructure), not part of the decode path:
 * libhwang_amd.so never links it.  It is built on the oracle's syntax engine
 * (oracle/h264_slice.c is written once for parsing and writing) and on the
 * oracle's reconstruction, so the encoder's reconstructed pictures are by
 * construction what the oracle decodes from the stream; the reference's own
 * dead encoder wrapper (hwang/video_encoder.cpp) played the same role via
 * libx264.
 *
 * The encoder aims at legal, varied streams (every macroblock type, all nine
 * intra modes, all sixteen sub-sample positions, bi-prediction, direct modes,
 * qp deltas, I_PCM, several slices), not at rate-distortion optimality.
 */
#include "../oracle/h264_internal.h"

#include <math.h>
#include <pthread.h>
#include <stdio.h>

/* ------------------------------------------------------------------------- */
typedef struct {
  int width, height, frames, gop, bframes, qp, cabac, t8x8, refs, slices, seed, threads, repeat;
  int chroma_qp_offset, deblock_idc, alpha_div2, beta_div2, implicit_wp, temporal_direct;
  int intra_only_pcm_period;
  /* syntax real encoders use by default that the first corpus never emitted */
  int explicit_wp;  /* PPS weighted_pred_flag = 1, weighted_bipred_idc = 1: pred_weight_table in every P/B slice */
  int wp_denom;     /* luma_log2_weight_denom (chroma uses a different one) */
  int rplm;         /* ref_pic_list_modification on a hash-chosen subset of slices */
  int mmco;         /* 1: adaptive marking script (ops 1,2,3,4,6, long-term IDR); 2: also op 5 (needs bframes 0) */
  /* container variants (not fragmented): */
  int moov_last;       /* ftyp, mdat, moov - the layout of files that were not "fast-started" */
  int free_boxes;      /* a free box between ftyp and what follows, a uuid box at the end of the file */
  int large_mdat;      /* mdat with a 64-bit size field whatever its size */
  int audio_track;     /* a second track (handler 'soun', own sample tables, its samples behind the video's in mdat) in front
                          of the video track */
  int random_intra;    /* intra modes (chroma, 16x16, 4x4, 8x8) and the intra family drawn by hash among the valid ones instead of
                          by best match: every mode of every family with every availability pattern shows up */
  int no_stss;         /* leave the sync sample box out when every sample is one (--gop 1) */
  int hevc_entry;      /* container tests: 1 = the sample entry is 'hev1' with an hvcC box, 2 = 'hvc1' (the samples stay
                          what they are: only indexers and format switches look at such a file) */
  int chunk_samples;   /* > 0: chunks of this many samples (every third chunk one more), the sound track's chunks between
                          them: several stsc runs, one chunk offset per chunk with gaps */
  int cover;           /* widens syntax coverage by hash (tests/coverage_model.py): intra macroblocks of every family in P and B slices,
                        * Intra16x16 with chroma but without luma AC, B 16x8 / 8x16 whose halves use different lists, and - at
                        * QP <= 3 - now and then a level so large that CAVLC needs level_prefix >= 16 */
  int wild_mv;         /* now and then a P macroblock takes a vector hundreds of samples long, mostly to somewhere outside the
                          reference picture: long mvd codes, prediction from clamped coordinates only */
  int global_mv;       /* known-answer pictures: what they decode to follows from the previous picture(s) and a few numbers
                          that depend on address / position / picture number only, so tests/test_*_independent.py can
                          compute it from the standard's clauses without a decoder.  1: every P (and, with --bframes, B)
                          picture is its reference(s) shifted by one vector per list (all quarter-sample positions, far
                          outside the picture; explicit / implicit weights), nothing coded; 2: a checkerboard of two vectors
                          (every macroblock edge: bS 1); 3: ... of inter and I_PCM macroblocks (bS 4, qP 0); 4: known
                          coefficient levels in every macroblock (4x4 or, with --t8x8 1, 8x8 transform; filter on: bS 2);
                          5 / 6 / 8: ... with Intra16x16 / Intra4x4 / Intra8x8 macroblocks of known modes in between
                          (filter on: every bS); 7: known vector differences and reference indices, 16x16 / 16x8 / 8x16 /
                          P_Skip (vector prediction); 9: all-B_Skip pictures (temporal direct with --temporal-direct 1);
                          10: coded and B_Skip macroblocks in a checkerboard (spatial direct); 11 (with --gop 1): intra
                          pictures of I_PCM and known intra macroblocks, nothing taken from a previous picture; 12: B
                          macroblocks with 16x16 / 16x8 / 8x16 partitions from list 0, 1 or both and known differences
                          per list.  --scaling 4 adds scaling matrices with known entries */
  int qp_swing;        /* every few macroblocks the QP jumps anywhere in 0..51: large mb_qp_delta values, incl. the ones that
                          only fit the -26..25 range by wrapping around 52 */
  int slice_mbs;       /* > 0: a new slice every so many macroblocks - boundaries in the middle of rows (overrides --slices) */
  int mixed_slices;    /* slices of other types inside a picture: I slices in P pictures, P and I slices in B pictures
                          (needs --slices >= 2) */
  int crop_left, crop_top; /* frame_crop_left / top_offset (two-sample units), taken out of the right / bottom padding */
  int multi_pps;       /* a second picture parameter set (id 1: other pic_init_qp, other chroma QP offsets) used by every
                          other coded picture */
  int nal_length_size; /* avcC lengthSizeMinusOne + 1: 4 (default), 2 or 1 */
  int extra_nals;      /* access unit delimiter in front of every picture, in-band SPS + PPS and an SEI at every IDR,
                          filler data behind the slices: what muxers and other encoders put into samples */
  int fn_bits, poc_bits; /* log2_max_frame_num, log2_max_pic_order_cnt_lsb (default 8 / 8; 4 makes both wrap every few pictures) */
  int bpyramid;     /* the middle B picture between two anchors is a reference (x264's b-pyramid): needs bframes >= 2, refs >= 4 */
  int slice_qp_delta; /* PPS pic_init_qp = qp - this, every slice header adds it back */
  int poc_type;     /* pic_order_cnt_type 0 / 1 (one-frame cycle, per-slice delta) / 2 (needs bframes 0) */
  int constrained_intra; /* PPS constrained_intra_pred_flag */
  int direct8x8;    /* SPS direct_8x8_inference_flag (default 1) */
  int rate_kbit;    /* > 0: rate control - kbit per P picture (I: x4, B: x0.5), QP steered per macroblock row */
  int scaling;      /* bit 0: SPS scaling matrices, bit 1: PPS scaling matrices (fall-back rules A and B, default flag) */
  /* container variants (hwang/mp4_index_creator.cpp:516-733 and the stz2 / co64 branches of the moov parser) */
  int fragmented;   /* 1..3: mvex/trex + one moof+mdat per GOP, three different tfhd / trun flag combinations */
  int co64, stz2;   /* 64-bit chunk offsets; compact sample sizes (16-bit fields) */
  const char* out;
  const char* yuv_out;
  int fps;
} Cfg;

static uint32_t hash32(uint32_t x) {
  x ^= x >> 16;
  x *= 0x7feb352dU;
  x ^= x >> 15;
  x *= 0x846ca68bU;
  x ^= x >> 16;
  return x;
}
static uint32_t hash3(uint32_t a, uint32_t b, uint32_t c) { return hash32(a * 0x9e3779b1U ^ hash32(b * 0x85ebca77U ^ hash32(c + 0xc2b2ae3dU))); }

/* ------------------------------------------------------------------------- */
/* procedural content                                                        */
/* ------------------------------------------------------------------------- */
typedef struct {
  int x, y, w, h;     /* quarter-sample position at t=0, size in samples */
  int vx, vy;         /* quarter samples per frame */
  uint32_t seed;
  int cu, cv;
} Rect;

typedef struct {
  int W, H;           /* coded size */
  int dw, dh;         /* display size */
  uint32_t seed;
  int pan_x, pan_y;   /* quarter samples per frame */
  Rect rects[8];
  uint8_t* plane[3];
} Scene;

static int lattice(uint32_t seed, int ix, int iy) { return (int)(hash3(seed, (uint32_t)ix, (uint32_t)iy) & 255); }

/* bilinear value noise; (x, y) in 1/256 lattice units */
static int vnoise(uint32_t seed, int64_t x, int64_t y) {
  int ix = (int)(x >> 8), iy = (int)(y >> 8), fx = (int)(x & 255), fy = (int)(y & 255);
  int a = lattice(seed, ix, iy), b = lattice(seed, ix + 1, iy);
  int c = lattice(seed, ix, iy + 1), d = lattice(seed, ix + 1, iy + 1);
  int top = a * (256 - fx) + b * fx, bot = c * (256 - fx) + d * fx;
  return (top * (256 - fy) + bot * fy) >> 16;
}

static void scene_init(Scene* s, const Cfg* c) {
  memset(s, 0, sizeof(*s));
  s->dw = c->width;
  s->dh = c->height;
  s->W = (c->width + 15) & ~15;
  s->H = (c->height + 15) & ~15;
  s->seed = (uint32_t)c->seed * 2654435761U + 17;
  s->pan_x = 7;  /* 1.75 samples / frame */
  s->pan_y = 2;  /* 0.5 */
  for (int i = 0; i < 8; ++i) {
    Rect* r = &s->rects[i];
    uint32_t h = hash3(s->seed, 1000 + i, 0);
    r->w = 48 + (int)(h % (unsigned)(c->width / 6 + 1));
    r->h = 40 + (int)((h >> 8) % (unsigned)(c->height / 5 + 1));
    r->x = (int)(hash3(s->seed, 1000 + i, 1) % (unsigned)(c->width * 4));
    r->y = (int)(hash3(s->seed, 1000 + i, 2) % (unsigned)(c->height * 4));
    r->vx = (int)(hash3(s->seed, 1000 + i, 3) % 33) - 16;
    r->vy = (int)(hash3(s->seed, 1000 + i, 4) % 17) - 8;
    r->seed = hash3(s->seed, 1000 + i, 5);
    r->cu = 64 + (int)(hash3(s->seed, 1000 + i, 6) % 128);
    r->cv = 64 + (int)(hash3(s->seed, 1000 + i, 7) % 128);
  }
  s->plane[0] = (uint8_t*)malloc((size_t)s->W * s->H);
  s->plane[1] = (uint8_t*)malloc((size_t)s->W * s->H / 4);
  s->plane[2] = (uint8_t*)malloc((size_t)s->W * s->H / 4);
}

static void scene_free(Scene* s) {
  for (int i = 0; i < 3; ++i) free(s->plane[i]);
}

static int wrap(int v, int m) {
  v %= m;
  return v < 0 ? v + m : v;
}

/* frame t (absolute display index) */
static void scene_render(Scene* s, int t) {
  const int W = s->W, H = s->H;
  /* sub-sample positions are realised by evaluating the continuous noise
   * field at quarter-sample offsets */
  const int64_t ox = (int64_t)s->pan_x * t, oy = (int64_t)s->pan_y * t; /* quarter samples */
  int rx[8], ry[8];
  for (int i = 0; i < 8; ++i) {
    rx[i] = wrap(s->rects[i].x + s->rects[i].vx * t, s->dw * 4);
    ry[i] = wrap(s->rects[i].y + s->rects[i].vy * t, s->dh * 4);
  }
  for (int y = 0; y < H; ++y) {
    const int yy = y < s->dh ? y : s->dh - 1;
    for (int x = 0; x < W; ++x) {
      const int xx = x < s->dw ? x : s->dw - 1;
      /* world coordinates in quarter samples */
      int64_t wx = (int64_t)xx * 4 + ox, wy = (int64_t)yy * 4 + oy;
      int v = 40 + (vnoise(s->seed, wx, wy) * 5 >> 3)            /* 64-sample lattice */
              + (vnoise(s->seed + 1, wx * 4, wy * 4) >> 2)        /* 16 */
              + (vnoise(s->seed + 2, wx * 16, wy * 16) >> 4);     /* 4 */
      for (int i = 0; i < 8; ++i) {
        const Rect* r = &s->rects[i];
        int dx = wrap(xx * 4 - rx[i], s->dw * 4), dy = wrap(yy * 4 - ry[i], s->dh * 4);
        if (dx < r->w * 4 && dy < r->h * 4)
          v = 30 + (vnoise(r->seed, (int64_t)dx * 8, (int64_t)dy * 8) * 3 >> 2) + (vnoise(r->seed + 9, (int64_t)dx * 40, (int64_t)dy * 40) >> 3);
      }
      /* film grain, triangular, sigma ~ 2 */
      uint32_t g = hash3(s->seed + 77, (uint32_t)(y * W + x), (uint32_t)t);
      v += (int)(g & 7) + (int)((g >> 3) & 7) - 7 >> 1;
      s->plane[0][y * W + x] = (uint8_t)clip8(v);
    }
  }
  for (int y = 0; y < H / 2; ++y) {
    const int yy = imin(y * 2, s->dh - 1);
    for (int x = 0; x < W / 2; ++x) {
      const int xx = imin(x * 2, s->dw - 1);
      int64_t wx = (int64_t)xx * 4 + ox, wy = (int64_t)yy * 4 + oy;
      int u = 96 + (vnoise(s->seed + 3, wx / 2, wy / 2) >> 2);
      int v = 160 - (vnoise(s->seed + 4, wx / 2, wy / 2) >> 2);
      for (int i = 0; i < 8; ++i) {
        const Rect* r = &s->rects[i];
        int dx = wrap(xx * 4 - rx[i], s->dw * 4), dy = wrap(yy * 4 - ry[i], s->dh * 4);
        if (dx < r->w * 4 && dy < r->h * 4) {
          u = r->cu;
          v = r->cv;
        }
      }
      s->plane[1][y * (W / 2) + x] = (uint8_t)clip8(u);
      s->plane[2][y * (W / 2) + x] = (uint8_t)clip8(v);
    }
  }
}

/* ------------------------------------------------------------------------- */
/* forward transform + quantisation by projection on the inverse basis       */
/* ------------------------------------------------------------------------- */
static double basis4[4][4];     /* [coef][sample] 1-D inverse basis */
static double norm4[4];
static double basis8[8][8];
static double norm8[8];

static void init_bases(void) {
  static const double a[4][4] = {{1, 1, 1, 1}, {1, .5, -.5, -1}, {1, -1, -1, 1}, {.5, -1, 1, -.5}};
  for (int k = 0; k < 4; ++k) {
    norm4[k] = 0;
    for (int i = 0; i < 4; ++i) {
      basis4[k][i] = a[k][i];
      norm4[k] += a[k][i] * a[k][i];
    }
  }
  /* 8-point: run the inverse butterfly of 8.5.13 in floating point on unit
   * vectors */
  for (int k = 0; k < 8; ++k) {
    double d[8] = {0}, e[8], f[8], g[8];
    d[k] = 1;
    e[0] = d[0] + d[4];
    e[1] = -d[3] + d[5] - d[7] - d[7] / 2;
    e[2] = d[0] - d[4];
    e[3] = d[1] + d[7] - d[3] - d[3] / 2;
    e[4] = d[2] / 2 - d[6];
    e[5] = -d[1] + d[7] + d[5] + d[5] / 2;
    e[6] = d[2] + d[6] / 2;
    e[7] = d[3] + d[5] + d[1] + d[1] / 2;
    f[0] = e[0] + e[6];
    f[1] = e[1] + e[7] / 4;
    f[2] = e[2] + e[4];
    f[3] = e[3] + e[5] / 4;
    f[4] = e[2] - e[4];
    f[5] = e[3] / 4 - e[5];
    f[6] = e[0] - e[6];
    f[7] = e[7] - e[1] / 4;
    g[0] = f[0] + f[7];
    g[1] = f[2] + f[5];
    g[2] = f[4] + f[3];
    g[3] = f[6] + f[1];
    g[4] = f[6] - f[1];
    g[5] = f[4] - f[3];
    g[6] = f[2] - f[5];
    g[7] = f[0] - f[7];
    norm8[k] = 0;
    for (int i = 0; i < 8; ++i) {
      basis8[k][i] = g[i];
      norm8[k] += g[i] * g[i];
    }
  }
}

/* residual (n x n, raster) -> "d" coefficients (the values the decoder's
 * scaling stage should produce), raster (row = vertical frequency) */
static void project(const int* res, int n, double* out) {
  double tmp[64];
  for (int y = 0; y < n; ++y)
    for (int k = 0; k < n; ++k) {
      double s = 0;
      for (int x = 0; x < n; ++x) s += res[y * n + x] * (n == 4 ? basis4[k][x] : basis8[k][x]);
      tmp[y * n + k] = s / (n == 4 ? norm4[k] : norm8[k]);
    }
  for (int k2 = 0; k2 < n; ++k2)
    for (int k = 0; k < n; ++k) {
      double s = 0;
      for (int y = 0; y < n; ++y) s += tmp[y * n + k] * (n == 4 ? basis4[k2][y] : basis8[k2][y]);
      out[k2 * n + k] = 64.0 * s / (n == 4 ? norm4[k2] : norm8[k2]);
    }
}

static int quant(double v, double step, double dead) {
  double a = fabs(v) / step + dead;
  int q = (int)a;
  if (q > 2000) q = 2000;
  return v < 0 ? -q : q;
}

/* ------------------------------------------------------------------------- */
/* encoder state                                                             */
/* ------------------------------------------------------------------------- */
typedef struct {
  const Cfg* cfg;
  H264Oracle* d;
  Scene scene;
  const uint8_t* src[3];
  int sstride[3];
  BitWriter w;
  uint8_t* sps_nal;
  int sps_len;
  uint8_t* pps_nal;
  int pps_len;
  uint8_t* pps1_nal; /* --multi-pps */
  int pps1_len;
  int coded_index;   /* pictures coded so far in this GOP (chooses the PPS) */
  int frame_abs;   /* absolute display index of the picture being coded */
  int poc1_prev_fn, poc1_fn_offset; /* pic_order_cnt_type 1: FrameNumOffset bookkeeping (8.2.1.2) */
  int rc_qp;       /* rate control: the QP the next macroblock row starts from (reset at every I picture, so that
                      GOPs stay independent of each other and of the number of encoder threads) */
  int hint_n;
  int hints[10][2]; /* candidate motion per frame of distance, quarter samples */
} Enc;

typedef struct {
  uint8_t* data;
  size_t size;
  int display;  /* display index inside the GOP */
  int key;
} Sample;

/* --- NAL ------------------------------------------------------------------ */
static size_t nal_escape(const uint8_t* rbsp, size_t n, int header, uint8_t* out) {
  size_t o = 0;
  out[o++] = (uint8_t)header;
  int zeros = 0;
  for (size_t i = 0; i < n; ++i) {
    if (zeros >= 2 && rbsp[i] <= 3) {
      out[o++] = 3;
      zeros = 0;
    }
    out[o++] = rbsp[i];
    zeros = rbsp[i] == 0 ? zeros + 1 : 0;
  }
  return o;
}

static void bw_trailing(BitWriter* w) {
  bw_put(w, 1, 1);
  while (w->pos & 7) bw_put(w, 0, 1);
}

static int level_for(const Cfg* c) {
  int mbs = ((c->width + 15) / 16) * ((c->height + 15) / 16);
  if (mbs <= 396) return 20;
  if (mbs <= 1620) return 30;
  if (mbs <= 3600) return 31;
  if (mbs <= 8192) return 40;
  if (mbs <= 22080) return 50;
  return 51;
}

static int profile_for(const Cfg* c) {
  if (c->t8x8 || c->scaling) return 100;
  return (c->cabac || c->multi_pps == 2 || c->bframes || c->explicit_wp) ? 77 : 66;
}

/* scaling_list() syntax (7.3.2.1.1.1).  kind 0: not present (fall-back rule),
 * 1: all `size` entries explicit, 2: useDefaultScalingMatrixFlag (first
 * nextScale = 0), 3: explicit up to a hash-chosen position where nextScale
 * becomes 0 and the last value repeats.  Values 6..56, rising with frequency. */
static void put_scaling_list(BitWriter* w, int size, int kind, uint32_t seed) {
  if (kind == 0) {
    bw_put(w, 0, 1);
    return;
  }
  bw_put(w, 1, 1);
  if (kind == 2) {
    bw_se(w, -8);
    return;
  }
  int last = 8;
  const int stop = kind == 3 ? size / 2 + (int)(seed % 5) : size;
  for (int j = 0; j < size; ++j) {
    if (j == stop) {
      int delta = -last;
      if (delta < -128) delta += 256;
      bw_se(w, delta);
      break;
    }
    int v = 6 + j * (size == 16 ? 40 : 10) / 16 + (int)(hash3(seed, (uint32_t)j, (uint32_t)size) % 7);
    int delta = v - last;
    bw_se(w, delta);
    last = v;
  }
}

static void write_sps(Enc* e) {
  const Cfg* c = e->cfg;
  BitWriter w = {0};
  const int profile = profile_for(c);
  bw_put(&w, profile, 8);
  bw_put(&w, 0, 8);
  bw_put(&w, level_for(c), 8);
  bw_ue(&w, 0);
  if (profile == 100) {
    bw_ue(&w, 1); /* chroma_format_idc */
    bw_ue(&w, 0);
    bw_ue(&w, 0);
    bw_put(&w, 0, 1); /* qpprime_y_zero_transform_bypass */
    if (c->scaling & 4) {
      /* all eight lists explicit, entry j (zig-zag order) of list i = 8 + (i * 7 + j * 3) % 25: a scaling matrix that
       * tests/test_residual_independent.py can restate without the hash */
      bw_put(&w, 1, 1);
      for (int i = 0; i < 8; ++i) {
        int last = 8;
        bw_put(&w, 1, 1);
        for (int j = 0; j < (i < 6 ? 16 : 64); ++j) {
          const int v = 8 + (i * 7 + j * 3) % 25;
          bw_se(&w, v - last);
          last = v;
        }
      }
    } else if (c->scaling & 1) {
      /* Y-intra explicit, Cb-intra falls back to it, Cr-intra explicit, Y-inter "use default",
       * Cb-inter falls back to that, Cr-inter stops early; 8x8 intra explicit, 8x8 inter absent
       * (fall-back rule A: the default list) */
      static const int kinds[8] = {1, 0, 1, 2, 0, 3, 1, 0};
      bw_put(&w, 1, 1);
      for (int i = 0; i < 8; ++i) put_scaling_list(&w, i < 6 ? 16 : 64, kinds[i], (uint32_t)c->seed * 31 + i);
    } else {
      bw_put(&w, 0, 1); /* seq_scaling_matrix_present */
    }
  }
  bw_ue(&w, c->fn_bits - 4);      /* log2_max_frame_num_minus4 */
  bw_ue(&w, c->poc_type); /* pic_order_cnt_type */
  if (c->poc_type == 0) {
    bw_ue(&w, c->poc_bits - 4);      /* log2_max_pic_order_cnt_lsb_minus4 */
  } else if (c->poc_type == 1) {
    bw_put(&w, 0, 1);  /* delta_pic_order_always_zero_flag: every slice carries delta_pic_order_cnt[0] */
    bw_se(&w, -1);     /* offset_for_non_ref_pic */
    bw_se(&w, 0);      /* offset_for_top_to_bottom_field */
    bw_ue(&w, 1);      /* num_ref_frames_in_pic_order_cnt_cycle */
    bw_se(&w, 2 * (c->bframes + 1)); /* offset_for_ref_frame[0] */
  }
  bw_ue(&w, c->refs);
  bw_put(&w, 0, 1);  /* gaps */
  bw_ue(&w, (c->width + 15) / 16 - 1);
  bw_ue(&w, (c->height + 15) / 16 - 1);
  bw_put(&w, 1, 1);  /* frame_mbs_only */
  bw_put(&w, c->direct8x8 ? 1 : 0, 1);  /* direct_8x8_inference */
  const int cr = (((c->width + 15) & ~15) - c->width) / 2, cb = (((c->height + 15) & ~15) - c->height) / 2;
  if (cr || cb) {
    bw_put(&w, 1, 1);
    bw_ue(&w, imin(c->crop_left, cr));
    bw_ue(&w, cr - imin(c->crop_left, cr));
    bw_ue(&w, imin(c->crop_top, cb));
    bw_ue(&w, cb - imin(c->crop_top, cb));
  } else {
    bw_put(&w, 0, 1);
  }
  bw_put(&w, 1, 1); /* vui_parameters_present */
  bw_put(&w, 0, 1); /* aspect_ratio_info */
  bw_put(&w, 0, 1); /* overscan */
  bw_put(&w, 0, 1); /* video_signal_type */
  bw_put(&w, 0, 1); /* chroma_loc */
  bw_put(&w, 1, 1); /* timing_info */
  bw_put(&w, 1, 32);
  bw_put(&w, 2 * c->fps, 32);
  bw_put(&w, 1, 1);
  bw_put(&w, 0, 1); /* nal_hrd */
  bw_put(&w, 0, 1); /* vcl_hrd */
  bw_put(&w, 0, 1); /* pic_struct_present */
  bw_put(&w, 1, 1); /* bitstream_restriction */
  bw_put(&w, 1, 1); /* motion_vectors_over_pic_boundaries */
  bw_ue(&w, 0);
  bw_ue(&w, 0);
  bw_ue(&w, 16);
  bw_ue(&w, 16);
  bw_ue(&w, c->bframes ? (c->bpyramid ? 2 : 1) : 0);           /* num_reorder_frames */
  bw_ue(&w, imax(c->refs, c->bframes ? (c->bpyramid ? 3 : 2) : 1)); /* max_dec_frame_buffering */
  bw_trailing(&w);
  e->sps_nal = (uint8_t*)malloc(w.pos / 8 * 2 + 8);
  e->sps_len = (int)nal_escape(w.buf, w.pos / 8, 0x67, e->sps_nal);
  free(w.buf);
}

/* the second parameter set differs where a decoder must follow the change per picture */
static int pps_init_qp(const Cfg* c, int id) { return c->qp - c->slice_qp_delta + (id ? 4 : 0); }
/* --multi-pps 2: the second set also switches the entropy coder, so CABAC and CAVLC pictures alternate in one stream */
static int pps_cabac(const Cfg* c, int id) { return (id && c->multi_pps == 2) ? !c->cabac : c->cabac; }
static int pps_chroma_offset(const Cfg* c, int id) { return id ? iclip(-12, 12, c->chroma_qp_offset + 2) : c->chroma_qp_offset; }
static void write_pps_id(Enc* e, int id) {
  const Cfg* c = e->cfg;
  BitWriter w = {0};
  bw_ue(&w, id);
  bw_ue(&w, 0);
  bw_put(&w, pps_cabac(c, id), 1);
  bw_put(&w, 0, 1);
  bw_ue(&w, 0); /* slice groups */
  bw_ue(&w, c->refs - 1);
  bw_ue(&w, 0);
  bw_put(&w, c->explicit_wp ? 1 : 0, 1);                           /* weighted_pred */
  bw_put(&w, c->explicit_wp ? 1 : (c->implicit_wp ? 2 : 0), 2);    /* weighted_bipred_idc */
  bw_se(&w, pps_init_qp(c, id) - 26);
  bw_se(&w, 0);
  bw_se(&w, pps_chroma_offset(c, id));
  bw_put(&w, 1, 1); /* deblocking_filter_control_present */
  bw_put(&w, c->constrained_intra ? 1 : 0, 1); /* constrained_intra_pred */
  bw_put(&w, 0, 1); /* redundant_pic_cnt_present */
  if (c->t8x8 || (c->scaling & 2)) {
    bw_put(&w, c->t8x8 ? 1 : 0, 1);
    if (c->scaling & 2) {
      /* fall-back rule B: absent lists 0 / 3 / 6 / 7 take the SPS's, absent 1, 2, 4, 5 the previous list */
      static const int kinds[8] = {0, 1, 0, 1, 2, 0, 0, 3};
      bw_put(&w, 1, 1);
      for (int i = 0; i < 6 + (c->t8x8 ? 2 : 0); ++i) put_scaling_list(&w, i < 6 ? 16 : 64, kinds[i], (uint32_t)c->seed * 57 + 100 + i);
    } else {
      bw_put(&w, 0, 1);
    }
    const int co = pps_chroma_offset(c, id);
    bw_se(&w, co > 0 ? co - 1 : co);
  }
  bw_trailing(&w);
  uint8_t* nal = (uint8_t*)malloc(w.pos / 8 * 2 + 8);
  const int len = (int)nal_escape(w.buf, w.pos / 8, 0x68, nal);
  if (id) e->pps1_nal = nal, e->pps1_len = len;
  else e->pps_nal = nal, e->pps_len = len;
  free(w.buf);
}
static void write_pps(Enc* e) {
  write_pps_id(e, 0);
  if (e->cfg->multi_pps) write_pps_id(e, 1);
}

static size_t make_avcc(const Enc* e, uint8_t* out) {
  size_t o = 0;
  out[o++] = 1;
  out[o++] = e->sps_nal[1];
  out[o++] = e->sps_nal[2];
  out[o++] = e->sps_nal[3];
  out[o++] = (uint8_t)(0xfc | (e->cfg->nal_length_size - 1));
  out[o++] = 0xe1;
  out[o++] = (uint8_t)(e->sps_len >> 8);
  out[o++] = (uint8_t)e->sps_len;
  memcpy(out + o, e->sps_nal, e->sps_len);
  o += e->sps_len;
  out[o++] = e->cfg->multi_pps ? 2 : 1;
  out[o++] = (uint8_t)(e->pps_len >> 8);
  out[o++] = (uint8_t)e->pps_len;
  memcpy(out + o, e->pps_nal, e->pps_len);
  o += e->pps_len;
  if (e->cfg->multi_pps) {
    out[o++] = (uint8_t)(e->pps1_len >> 8);
    out[o++] = (uint8_t)e->pps1_len;
    memcpy(out + o, e->pps1_nal, e->pps1_len);
    o += e->pps1_len;
  }
  return o;
}

/* ------------------------------------------------------------------------- */
/* macroblock decisions                                                      */
/* ------------------------------------------------------------------------- */
static int sad_region(const Enc* e, int plane, int x0, int y0, int w, int h) {
  const Picture* cur = e->d->cur;
  const uint8_t* a = cur->plane[plane] + y0 * cur->stride[plane] + x0;
  const uint8_t* b = e->src[plane] + y0 * e->sstride[plane] + x0;
  int s = 0;
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x) s += iabs(a[y * cur->stride[plane] + x] - b[y * e->sstride[plane] + x]);
  return s;
}

static void residual_of(const Enc* e, int plane, int x0, int y0, int n, int* res) {
  const Picture* cur = e->d->cur;
  for (int y = 0; y < n; ++y)
    for (int x = 0; x < n; ++x)
      res[y * n + x] = e->src[plane][(y0 + y) * e->sstride[plane] + x0 + x] - cur->plane[plane][(y0 + y) * cur->stride[plane] + x0 + x];
}

/* quantise the 4x4 luma block at raster index blk from the prediction that
 * currently sits in the picture; returns 1 if any level is non-zero */
static int quant_luma4(Enc* e, MB* p, int blk, int qp, int intra, int skip_dc) {
  H264Oracle* d = e->d;
  int res[16];
  double c[16];
  const int x0 = d->mb.mbx * 16 + (blk & 3) * 4, y0 = d->mb.mby * 16 + (blk >> 2) * 4;
  residual_of(e, 0, x0, y0, 4, res);
  project(res, 4, c);
  const uint8_t* wt = d->apps->scaling4[intra ? 0 : 3];
  int any = 0;
  for (int k = skip_dc; k < 16; ++k) {
    double step = h264i_level_scale4(wt, qp, k) * pow(2.0, qp / 6 - 4);
    int q = quant(c[k], step, intra ? 1.0 / 3 : 1.0 / 6);
    p->luma[blk][k] = (int16_t)q;
    any |= q != 0;
  }
  if (skip_dc) p->luma[blk][0] = 0;
  if (e->cfg->cover && qp <= 1 && !skip_dc && wt[0] == 16) {
    /* a level beyond what level_prefix 15 can carry (|level| > 2528), alone in its block: scaled (x 10 or x 11 at
     * QP 0 / 1, flat weights) it is 27000 / 29700 - inside the 16-bit range the standard bounds coefficients and
     * transform intermediates to (8.5.12.1), and the block decodes to clipped samples */
    const uint32_t hl = hash3((uint32_t)e->cfg->seed + 29, (uint32_t)e->frame_abs, (uint32_t)(d->mb.addr * 16 + blk));
    if (hl % 5 == 0) {
      memset(p->luma[blk], 0, sizeof(p->luma[blk]));
      p->luma[blk][0] = (int16_t)((hl >> 8) & 1 ? -2700 : 2700);
      any = 1;
    }
  }
  return any;
}

static int quant_luma8(Enc* e, MB* p, int b8, int qp, int intra) {
  H264Oracle* d = e->d;
  int res[64];
  double c[64];
  const int x0 = d->mb.mbx * 16 + (b8 & 1) * 8, y0 = d->mb.mby * 16 + (b8 >> 1) * 8;
  residual_of(e, 0, x0, y0, 8, res);
  project(res, 8, c);
  const uint8_t* wt = d->apps->scaling8[intra ? 0 : 1];
  int any = 0;
  for (int k = 0; k < 64; ++k) {
    double step = h264i_level_scale8(wt, qp, k) * pow(2.0, qp / 6 - 6);
    int q = quant(c[k], step, intra ? 1.0 / 3 : 1.0 / 6);
    p->luma8[b8][k] = (int16_t)q;
    any |= q != 0;
  }
  return any;
}

/* chroma of the whole MB from the prediction in the picture; returns cbp chroma */
static int quant_chroma(Enc* e, MB* p, int qp, int intra) {
  H264Oracle* d = e->d;
  int cbp = 0;
  for (int pl = 0; pl < 2; ++pl) {
    const int qpc = chroma_qp_table[iclip(0, 51, qp + d->apps->chroma_qp_offset[pl])];
    const uint8_t* wt = d->apps->scaling4[(intra ? 0 : 3) + 1 + pl];
    double dc[4];
    for (int s = 0; s < 4; ++s) {
      int res[16];
      double c[16];
      residual_of(e, 1 + pl, d->mb.mbx * 8 + (s & 1) * 4, d->mb.mby * 8 + (s >> 1) * 4, 4, res);
      project(res, 4, c);
      dc[s] = c[0];
      for (int k = 1; k < 16; ++k) {
        double step = h264i_level_scale4(wt, qpc, k) * pow(2.0, qpc / 6 - 4);
        int q = quant(c[k], step, intra ? 1.0 / 3 : 1.0 / 6);
        p->chroma_ac[pl][s][k] = (int16_t)q;
        if (q) cbp = 2;
      }
      p->chroma_ac[pl][s][0] = 0;
    }
    /* decoder: f = H2 L H2 ; dc = (f * LS(0,0) << (qp/6)) >> 5  =>  L = H2 f H2 / 4 */
    const double step = h264i_level_scale4(wt, qpc, 0) * pow(2.0, qpc / 6 - 5);
    double f[4] = {dc[0] / step, dc[1] / step, dc[2] / step, dc[3] / step};
    double L[4] = {(f[0] + f[1] + f[2] + f[3]) / 4, (f[0] - f[1] + f[2] - f[3]) / 4, (f[0] + f[1] - f[2] - f[3]) / 4,
                   (f[0] - f[1] - f[2] + f[3]) / 4};
    for (int k = 0; k < 4; ++k) {
      int q = quant(L[k], 1.0, intra ? 1.0 / 3 : 1.0 / 6);
      p->chroma_dc[pl][k] = (int16_t)q;
      if (q && !cbp) cbp = 1;
    }
  }
  if (cbp < 2)
    memset(p->chroma_ac, 0, sizeof(p->chroma_ac));
  if (cbp < 1) memset(p->chroma_dc, 0, sizeof(p->chroma_dc));
  return cbp;
}

/* Reconstructs one transform block in place (prediction must be in the
 * picture) by running the oracle's reconstruction on a scratch MB.  Used
 * while deciding Intra4x4 / Intra8x8, where later blocks predict from
 * earlier reconstructed ones. */
static void recon_luma_block_now(Enc* e, const MB* p, int blk_or_b8, int is8, int qp, int intra) {
  /* small local dequant + idct through the public recon entry: build an MB
   * that only carries this block and call the residual adders via
   * h264i_recon_mb is too coarse, so replicate the few lines here */
  H264Oracle* d = e->d;
  Picture* cur = d->cur;
  if (!is8) {
    int sc[16], t[16], r[16];
    const uint8_t* wt = d->apps->scaling4[intra ? 0 : 3];
    int any = 0;
    for (int k = 0; k < 16; ++k) {
      int v = p->luma[blk_or_b8][k] * h264i_level_scale4(wt, qp, k);
      sc[k] = qp >= 24 ? v << (qp / 6 - 4) : (v + (1 << (3 - qp / 6))) >> (4 - qp / 6);
      any |= sc[k];
    }
    if (!any) return;
    for (int i = 0; i < 4; ++i) {
      const int* x = sc + 4 * i;
      int e0 = x[0] + x[2], e1 = x[0] - x[2], e2 = (x[1] >> 1) - x[3], e3 = x[1] + (x[3] >> 1);
      t[4 * i] = e0 + e3; t[4 * i + 1] = e1 + e2; t[4 * i + 2] = e1 - e2; t[4 * i + 3] = e0 - e3;
    }
    for (int j = 0; j < 4; ++j) {
      int d0 = t[j], d1 = t[4 + j], d2 = t[8 + j], d3 = t[12 + j];
      int e0 = d0 + d2, e1 = d0 - d2, e2 = (d1 >> 1) - d3, e3 = d1 + (d3 >> 1);
      r[j] = (e0 + e3 + 32) >> 6; r[4 + j] = (e1 + e2 + 32) >> 6; r[8 + j] = (e1 - e2 + 32) >> 6; r[12 + j] = (e0 - e3 + 32) >> 6;
    }
    uint8_t* dst = cur->plane[0] + (d->mb.mby * 16 + (blk_or_b8 >> 2) * 4) * cur->stride[0] + d->mb.mbx * 16 + (blk_or_b8 & 3) * 4;
    for (int y = 0; y < 4; ++y)
      for (int x = 0; x < 4; ++x) dst[y * cur->stride[0] + x] = (uint8_t)clip8(dst[y * cur->stride[0] + x] + r[y * 4 + x]);
    return;
  }
  /* 8x8: let the oracle do it -- temporarily make d->mb an I8x8/inter MB with
   * only this block and no prediction step.  Simpler: call the full MB recon
   * later; for the decision loop an approximate neighbour (prediction only
   * plus projected residual) would bias nothing in the bitstream's legality,
   * but intra 8x8 blocks predict from reconstructed neighbours, so be exact:
   * run the 8x8 inverse here as well. */
  int sc[64], t[64], o[64];
  const uint8_t* wt = d->apps->scaling8[intra ? 0 : 1];
  int any = 0;
  for (int k = 0; k < 64; ++k) {
    int v = p->luma8[blk_or_b8][k] * h264i_level_scale8(wt, qp, k);
    sc[k] = qp >= 36 ? v << (qp / 6 - 6) : (v + (1 << (5 - qp / 6))) >> (6 - qp / 6);
    any |= sc[k];
  }
  if (!any) return;
  for (int pass = 0; pass < 2; ++pass) {
    const int* in = pass ? t : sc;
    int* out = pass ? o : t;
    for (int i = 0; i < 8; ++i) {
      const int st = pass ? 8 : 1;
      const int* x = pass ? in + i : in + 8 * i;
      int d0 = x[0], d1 = x[st], d2 = x[2 * st], d3 = x[3 * st], d4 = x[4 * st], d5 = x[5 * st], d6 = x[6 * st], d7 = x[7 * st];
      int e0 = d0 + d4, e1 = -d3 + d5 - d7 - (d7 >> 1), e2 = d0 - d4, e3 = d1 + d7 - d3 - (d3 >> 1);
      int e4 = (d2 >> 1) - d6, e5 = -d1 + d7 + d5 + (d5 >> 1), e6 = d2 + (d6 >> 1), e7 = d3 + d5 + d1 + (d1 >> 1);
      int f0 = e0 + e6, f1 = e1 + (e7 >> 2), f2 = e2 + e4, f3 = e3 + (e5 >> 2);
      int f4 = e2 - e4, f5 = (e3 >> 2) - e5, f6 = e0 - e6, f7 = e7 - (e1 >> 2);
      int* y = pass ? out + i : out + 8 * i;
      y[0] = f0 + f7; y[st] = f2 + f5; y[2 * st] = f4 + f3; y[3 * st] = f6 + f1;
      y[4 * st] = f6 - f1; y[5 * st] = f4 - f3; y[6 * st] = f2 - f5; y[7 * st] = f0 - f7;
    }
  }
  uint8_t* dst = cur->plane[0] + (d->mb.mby * 16 + (blk_or_b8 >> 1) * 8) * cur->stride[0] + d->mb.mbx * 16 + (blk_or_b8 & 1) * 8;
  for (int y = 0; y < 8; ++y)
    for (int x = 0; x < 8; ++x) dst[y * cur->stride[0] + x] = (uint8_t)clip8(dst[y * cur->stride[0] + x] + ((o[y * 8 + x] + 32) >> 6));
}

static int sad_block(const Enc* e, int x0, int y0, int n) { return sad_region(e, 0, x0, y0, n, n); }

/* Intra decision for the current MB (h264i_begin_mb done).  Fills plan->mb
 * (kind, modes, levels, cbp, mb_type_raw) and returns its SAD-ish cost. */
static int decide_intra(Enc* e, MbPlan* plan, int qp, int slice_type) {
  H264Oracle* d = e->d;
  MB* p = &plan->mb;
  Picture* cur = d->cur;
  const int mbx = d->mb.mbx, mby = d->mb.mby;
  uint8_t* y0 = cur->plane[0] + mby * 16 * cur->stride[0] + mbx * 16;
  const int sy = cur->stride[0];
  const int left = h264i_intra_mb_ok(d, mbx - 1, mby), top = h264i_intra_mb_ok(d, mbx, mby - 1);
  const int tl = h264i_intra_mb_ok(d, mbx - 1, mby - 1);
  const uint32_t hv = hash3((uint32_t)e->cfg->seed, (uint32_t)e->frame_abs, (uint32_t)d->mb.addr);
  const int type_off = slice_type == 2 ? 0 : (slice_type == 0 ? 5 : 23);

  /* rare I_PCM */
  if (e->cfg->intra_only_pcm_period && hv % (unsigned)e->cfg->intra_only_pcm_period == 0) {
    p->kind = MB_IPCM;
    p->mb_type_raw = type_off + 25;
    for (int y = 0; y < 16; ++y) memcpy(p->pcm + y * 16, e->src[0] + (mby * 16 + y) * e->sstride[0] + mbx * 16, 16);
    for (int pl = 0; pl < 2; ++pl)
      for (int y = 0; y < 8; ++y)
        memcpy(p->pcm + 256 + pl * 64 + y * 8, e->src[1 + pl] + (mby * 8 + y) * e->sstride[1 + pl] + mbx * 8, 8);
    return 0;
  }

  /* chroma mode */
  int best_c = 0, best_cc = 1 << 30;
  for (int mode = 0; mode < 4; ++mode) {
    if ((mode == 1 && !left) || (mode == 2 && !top) || (mode == 3 && !(left && top && tl))) continue;
    int cost = 0;
    for (int pl = 0; pl < 2; ++pl) {
      h264i_pred_chroma8x8(cur->plane[1 + pl] + mby * 8 * cur->stride[1 + pl] + mbx * 8, cur->stride[1 + pl], mode, left, top);
      cost += sad_region(e, 1 + pl, mbx * 8, mby * 8, 8, 8);
    }
    if (cost < best_cc || (e->cfg->random_intra && (hv >> 20) % 4 == (uint32_t)mode)) {
      best_cc = cost;
      best_c = mode;
      if (e->cfg->random_intra && (hv >> 20) % 4 == (uint32_t)mode) break;
    }
  }
  p->chroma_mode = best_c;

  /* Intra16x16 */
  int best16 = 2, cost16 = 1 << 30;
  for (int mode = 0; mode < 4; ++mode) {
    if ((mode == 0 && !top) || (mode == 1 && !left) || (mode == 3 && !(left && top && tl))) continue;
    h264i_pred16x16(y0, sy, mode, left, top);
    int cost = sad_block(e, mbx * 16, mby * 16, 16);
    if (cost < cost16 || (e->cfg->random_intra && (hv >> 22) % 4 == (uint32_t)mode)) {
      cost16 = cost;
      best16 = mode;
      if (e->cfg->random_intra && (hv >> 22) % 4 == (uint32_t)mode) break;
    }
  }
  /* pick the family: smooth MBs -> 16x16; detailed ones -> 4x4 or 8x8 */
  int family = 16;
  if (cost16 > 256 * 5 || (hv >> 8) % 5 == 0) family = (e->cfg->t8x8 && ((hv >> 12) & 1)) ? 8 : 4;
  if (e->cfg->random_intra) family = (hv >> 24) % 3 == 0 ? 16 : ((e->cfg->t8x8 && ((hv >> 12) & 1)) ? 8 : 4);

  if (family == 16) {
    p->kind = MB_I16x16;
    p->i16_mode = best16;
    h264i_pred16x16(y0, sy, best16, left, top);
    /* AC per block + DC through the Hadamard */
    double dcs[16];
    int any_ac = 0;
    const uint8_t* wt = d->apps->scaling4[0];
    for (int blk = 0; blk < 16; ++blk) {
      int res[16];
      double c[16];
      residual_of(e, 0, mbx * 16 + (blk & 3) * 4, mby * 16 + (blk >> 2) * 4, 4, res);
      project(res, 4, c);
      dcs[blk] = c[0];
      p->luma[blk][0] = 0;
      for (int k = 1; k < 16; ++k) {
        double step = h264i_level_scale4(wt, qp, k) * pow(2.0, qp / 6 - 4);
        int q = quant(c[k], step, 1.0 / 3);
        p->luma[blk][k] = (int16_t)q;
        any_ac |= q != 0;
      }
    }
    if (e->cfg->cover && (hv >> 27) % 3 == 0) any_ac = 0; /* Intra16x16 types 1..12: DC (and chroma) only */
    if (!any_ac) memset(p->luma, 0, sizeof(p->luma));
    /* decoder: f = H L H; dcY = f * LS * 2^(qp/6-6)  =>  L = H f H / 16 */
    const double step = h264i_level_scale4(wt, qp, 0) * pow(2.0, qp / 6 - 6);
    double f[16], t[16];
    for (int k = 0; k < 16; ++k) f[k] = dcs[k] / step;
    for (int i = 0; i < 4; ++i) {
      double a = f[4 * i], b = f[4 * i + 1], c2 = f[4 * i + 2], dd = f[4 * i + 3];
      t[4 * i] = a + b + c2 + dd; t[4 * i + 1] = a + b - c2 - dd; t[4 * i + 2] = a - b - c2 + dd; t[4 * i + 3] = a - b + c2 - dd;
    }
    for (int j = 0; j < 4; ++j) {
      double a = t[j], b = t[4 + j], c2 = t[8 + j], dd = t[12 + j];
      double o[4] = {a + b + c2 + dd, a + b - c2 - dd, a - b - c2 + dd, a - b + c2 - dd};
      for (int i = 0; i < 4; ++i) p->luma_dc[4 * i + j] = (int16_t)quant(o[i] / 16, 1.0, 1.0 / 3);
    }
    if (e->cfg->cover && slice_type != 2 && (hv >> 29) % 3 == 0) memset(p->luma_dc, 0, sizeof(p->luma_dc)); /* coded_block_flag 0 for the DC block */
    for (int pl = 0; pl < 2; ++pl)
      h264i_pred_chroma8x8(cur->plane[1 + pl] + mby * 8 * cur->stride[1 + pl] + mbx * 8, cur->stride[1 + pl], best_c, left, top);
    int cc = quant_chroma(e, p, qp, 1);
    p->cbp = (any_ac ? 15 : 0) | cc << 4;
    p->mb_type_raw = type_off + 1 + best16 + 4 * cc + (any_ac ? 12 : 0);
    p->qp = qp;
    return cost16 + best_cc;
  }

  /* Intra4x4 / Intra8x8: sequential, each block predicts from reconstructed
   * neighbours */
  p->kind = family == 4 ? MB_I4x4 : MB_I8x8;
  p->t8x8 = family == 8;
  p->mb_type_raw = type_off;
  d->mb.kind = p->kind; /* availability helpers look at d->mb */
  int total = 0, cbp = 0;
  const int nblk = family == 4 ? 16 : 4;
  for (int k = 0; k < nblk; ++k) {
    const int bx = family == 4 ? blk4_x[k] : (k & 1) * 2, by = family == 4 ? blk4_y[k] : (k >> 1) * 2;
    const int zz = family == 4 ? k : k * 4, n = family;
    const int hl = h264i_intra_blk_ok(d, bx - 1, by, zz), ht = h264i_intra_blk_ok(d, bx, by - 1, zz);
    const int htl = h264i_intra_blk_ok(d, bx - 1, by - 1, zz);
    const int htr = h264i_intra_blk_ok(d, bx + n / 4, by - 1, zz);
    uint8_t* dst = y0 + by * 4 * sy + bx * 4;
    int best = 2, bc = 1 << 30;
    /* start the mode scan at a hash-chosen offset so ties break differently
     * from block to block and every mode shows up in flat regions */
    for (int mm = 0; mm < 9; ++mm) {
      const int mode = (mm + (int)((hv >> (k & 15)) % 9)) % 9;
      const int need_t = mode == 0 || mode == 3 || mode == 7, need_l = mode == 1 || mode == 8;
      const int need_all = mode == 4 || mode == 5 || mode == 6;
      if ((need_t && !ht) || (need_l && !hl) || (need_all && !(ht && hl && htl))) continue;
      if (n == 4) h264i_pred4x4(dst, sy, mode, hl, ht, htl, htr);
      else h264i_pred8x8l(dst, sy, mode, hl, ht, htl, htr);
      int cost = sad_block(e, mbx * 16 + bx * 4, mby * 16 + by * 4, n);
      if (cost < bc) {
        bc = cost;
        best = mode;
      }
      if (e->cfg->random_intra) break; /* the first valid mode of the hash-rotated scan */
    }
    if (n == 4) {
      h264i_pred4x4(dst, sy, best, hl, ht, htl, htr);
      p->i4_mode[by * 4 + bx] = (int8_t)best;
      if (quant_luma4(e, p, by * 4 + bx, qp, 1, 0)) cbp |= 1 << ((by >> 1) * 2 + (bx >> 1));
      recon_luma_block_now(e, p, by * 4 + bx, 0, qp, 1);
    } else {
      h264i_pred8x8l(dst, sy, best, hl, ht, htl, htr);
      for (int s = 0; s < 4; ++s) p->i4_mode[(by + (s >> 1)) * 4 + bx + (s & 1)] = (int8_t)best;
      if (quant_luma8(e, p, k, qp, 1)) cbp |= 1 << k;
      recon_luma_block_now(e, p, k, 1, qp, 1);
    }
    total += bc;
  }
  /* blocks whose 8x8 has its CBP bit clear must carry no levels */
  for (int b8 = 0; b8 < 4; ++b8)
    if (!((cbp >> b8) & 1)) {
      memset(p->luma8[b8], 0, sizeof(p->luma8[b8]));
      for (int s = 0; s < 4; ++s) memset(p->luma[((b8 >> 1) * 2 + (s >> 1)) * 4 + (b8 & 1) * 2 + (s & 1)], 0, sizeof(p->luma[0]));
    }
  for (int pl = 0; pl < 2; ++pl)
    h264i_pred_chroma8x8(cur->plane[1 + pl] + mby * 8 * cur->stride[1 + pl] + mbx * 8, cur->stride[1 + pl], best_c, left, top);
  p->cbp = cbp | quant_chroma(e, p, qp, 1) << 4;
  p->qp = qp;
  return total + best_cc + 64;
}

/* evaluates the motion currently in d->mb (ref/mv): predicts, returns SAD */
static int eval_inter(Enc* e) {
  H264Oracle* d = e->d;
  d->pred_luma_only = 1;
  h264i_inter_pred_mb(d);
  d->pred_luma_only = 0;
  return sad_block(e, d->mb.mbx * 16, d->mb.mby * 16, 16);
}

static void set_all_motion(MB* m, int list, int ref, int mvx, int mvy) {
  for (int q = 0; q < 4; ++q) m->ref[list][q] = (int8_t)ref;
  for (int k = 0; k < 16; ++k) {
    m->mv[list][k][0] = (int16_t)(ref >= 0 ? mvx : 0);
    m->mv[list][k][1] = (int16_t)(ref >= 0 ? mvy : 0);
  }
}

static int mv_ok(const Enc* e, int mvx, int mvy) {
  /* stay inside what level limits allow and keep the reference window near
   * the picture (samples outside are edge-clamped, which is legal) */
  (void)e;
  return mvx > -2000 && mvx < 2000 && mvy > -1000 && mvy < 1000;
}

/* quantises luma+chroma of an inter MB whose prediction sits in the picture */
static void quant_inter(Enc* e, MbPlan* plan, int qp, int allow8) {
  MB* p = &plan->mb;
  int cbp = 0;
  if (allow8 && p->t8x8) {
    for (int b8 = 0; b8 < 4; ++b8) {
      if (quant_luma8(e, p, b8, qp, 0)) cbp |= 1 << b8;
      else memset(p->luma8[b8], 0, sizeof(p->luma8[b8]));
    }
  } else {
    p->t8x8 = 0;
    for (int blk = 0; blk < 16; ++blk)
      if (quant_luma4(e, p, blk, qp, 0, 0)) cbp |= 1 << ((blk >> 3) * 2 + ((blk & 3) >> 1));
    for (int b8 = 0; b8 < 4; ++b8)
      if (!((cbp >> b8) & 1))
        for (int s = 0; s < 4; ++s) memset(p->luma[((b8 >> 1) * 2 + (s >> 1)) * 4 + (b8 & 1) * 2 + (s & 1)], 0, sizeof(p->luma[0]));
  }
  p->cbp = cbp | quant_chroma(e, p, qp, 0) << 4;
  if (!(p->cbp & 15)) p->t8x8 = 0;
  p->qp = qp;
}

static void copy_motion_from_state(const H264Oracle* d, MB* p) {
  memcpy(p->mv, d->mb.mv, sizeof(p->mv));
  memcpy(p->ref, d->mb.ref, sizeof(p->ref));
}

/* --global-mv: the one vector of P picture number k (quarter samples).  tests/test_inter_independent.py restates this
 * function; keep the two in step. */
static void global_mv_of(int k, int width, int height, int* mvx, int* mvy) {
  const int fx = k & 3, fy = (k >> 2) & 3;                 /* all 16 fractional positions over 16 pictures */
  int ix = ((k * 5) % 9) - 4, iy = ((k * 7) % 5) - 2;      /* whole samples, both parities (chroma eighths) */
  if (k % 6 == 5) ix = (k % 12 == 5 ? -1 : 1) * (width / 2 + 3 * k);   /* well outside, left / right */
  if (k % 7 == 6) iy = (k % 14 == 6 ? -1 : 1) * (height / 2 + 2 * k);  /* ... above / below */
  *mvx = ix * 4 + fx;
  *mvy = iy * 4 + fy;
}

/* --global-mv 4: the level of coefficient `pos` (raster, 0..15) of 4x4 block `blk` of macroblock `addr` in picture k;
 * comp 0 luma (blk raster 0..15), 1 / 2 chroma AC of Cb / Cr (blk 0..3, pos 0 unused), 3 / 4 chroma DC of Cb / Cr (pos =
 * the block), 5 luma with the 8x8 transform (blk = the 8x8 block, pos raster 0..63), 6 the Intra16x16 luma DC levels
 * (pos = the block's position in the macroblock, raster).  Restated in
 * tests/test_residual_independent.py: sparse, small, now and then a large one. */
static int global_level(int k, int addr, int comp, int blk, int pos) {
  const int t = addr * 7 + blk * 5 + pos * 3 + k * 11 + comp * 13;
  if ((comp <= 2 || comp == 5) && (addr + blk + k) % 3 == 0) return 0; /* a third of the blocks carry nothing (their edges: bS 0) */
  if (t % 5 != 0) return 0;
  int v = (addr + blk * 3 + pos + comp) % 9 - 4;
  if (t % 35 == 0) v *= 9;
  return v;
}

/* --global-mv 3: what an I_PCM macroblock's sample at (x, y) of plane pl differs by from the prediction there
 * (restated in tests/test_deblock_independent.py): mostly a few levels, now and then a step the filter must leave alone */
static int global_pcm_delta(int k, int pl, int x, int y) {
  int e = (x * 7 + y * 13 + k * 5 + pl * 3) % 13 - 6;
  if (((x >> 2) + (y >> 2) + k) % 5 == 0) e += 24;
  return e;
}

/* --global-mv 11 (with --gop 1): intra pictures whose content follows from position and picture number alone - a
 * checkerboard of I_PCM macroblocks (global_pcm_sample) and Intra16x16 / Intra4x4 / Intra8x8 macroblocks with known
 * modes and levels (as --global-mv 5 / 6 / 8).  tests/test_residual_independent.py rebuilds them from nothing. */
static int global_pcm_sample(int k, int pl, int x, int y) {
  return (x * 7 + y * 13 + k * 29 + pl * 31 + ((x >> 2) ^ (y >> 3)) * 5) & 0xff;
}
static void decide_known_intra(Enc* e, MbPlan* plan) {
  H264Oracle* d = e->d;
  MB* p = &plan->mb;
  const int addr = d->mb.addr, k = e->frame_abs, mbx = d->mb.mbx, mby = d->mb.mby;
  if ((mbx + mby) & 1) {
    p->kind = MB_IPCM;
    p->mb_type_raw = 25;
    for (int y = 0; y < 16; ++y)
      for (int x = 0; x < 16; ++x) p->pcm[y * 16 + x] = (uint8_t)global_pcm_sample(k, 0, mbx * 16 + x, mby * 16 + y);
    for (int pl = 0; pl < 2; ++pl)
      for (int y = 0; y < 8; ++y)
        for (int x = 0; x < 8; ++x) p->pcm[256 + pl * 64 + y * 8 + x] = (uint8_t)global_pcm_sample(k, 1 + pl, mbx * 8 + x, mby * 8 + y);
    return;
  }
  const int inside = mbx > 0 && mby > 0;
  int family = (addr / 2 + k) % 3; /* 0 Intra16x16, 1 Intra4x4, 2 Intra8x8 (Intra4x4 without --t8x8) */
  int cmode = (addr * 3 + k) % 4, chroma = 0;
  if (family == 2 && !e->cfg->t8x8) family = 1;
  if ((cmode == 1 && mbx == 0) || (cmode == 2 && mby == 0) || (cmode == 3 && !inside)) cmode = 0;
  p->chroma_mode = cmode;
  for (int pl = 0; pl < 2; ++pl)
    for (int blk = 0; blk < 4; ++blk) {
      p->chroma_dc[pl][blk] = (int16_t)global_level(k, addr, 3 + pl, 0, blk);
      if (p->chroma_dc[pl][blk] && chroma < 1) chroma = 1;
      for (int pos = 1; pos < 16; ++pos) {
        p->chroma_ac[pl][blk][pos] = (int16_t)global_level(k, addr, 1 + pl, blk, pos);
        if (p->chroma_ac[pl][blk][pos]) chroma = 2;
      }
    }
  p->qp = e->cfg->qp;
  if (family == 0) {
    int mode = (addr + k) % 4, any_ac = 0;
    if ((mode == 0 && mby == 0) || (mode == 1 && mbx == 0) || (mode == 3 && !inside)) mode = 2;
    p->kind = MB_I16x16;
    p->i16_mode = mode;
    for (int pos = 0; pos < 16; ++pos) p->luma_dc[pos] = (int16_t)global_level(k, addr, 6, 0, pos);
    for (int blk = 0; blk < 16; ++blk)
      for (int pos = 1; pos < 16; ++pos) {
        p->luma[blk][pos] = (int16_t)global_level(k, addr, 0, blk, pos);
        if (p->luma[blk][pos]) any_ac = 1;
      }
    p->cbp = (any_ac ? 15 : 0) | chroma << 4;
    p->mb_type_raw = 1 + mode + 4 * chroma + (any_ac ? 12 : 0);
    return;
  }
  int cbp = 0;
  p->mb_type_raw = 0;
  if (family == 1) {
    p->kind = MB_I4x4;
    for (int blk = 0; blk < 16; ++blk) {
      p->i4_mode[blk] = (int8_t)(inside ? (addr * 5 + blk * 7 + k) % 9 : 2);
      for (int pos = 0; pos < 16; ++pos) {
        p->luma[blk][pos] = (int16_t)global_level(k, addr, 0, blk, pos);
        if (p->luma[blk][pos]) cbp |= 1 << ((blk >> 3) * 2 + ((blk & 3) >> 1));
      }
    }
  } else {
    p->kind = MB_I8x8;
    p->t8x8 = 1;
    for (int b8 = 0; b8 < 4; ++b8) {
      const int mode = inside ? (addr * 5 + b8 * 7 + k) % 9 : 2;
      for (int s4 = 0; s4 < 4; ++s4) p->i4_mode[((b8 >> 1) * 2 + (s4 >> 1)) * 4 + (b8 & 1) * 2 + (s4 & 1)] = (int8_t)mode;
      for (int pos = 0; pos < 64; ++pos) {
        p->luma8[b8][pos] = (int16_t)global_level(k, addr, 5, b8, pos);
        if (p->luma8[b8][pos]) cbp |= 1 << b8;
      }
    }
  }
  p->cbp = cbp | chroma << 4;
}

/* P macroblock */
static void decide_p(Enc* e, MbPlan* plan, int qp, int dist /* frames to ref 0 */) {
  H264Oracle* d = e->d;
  MB* p = &plan->mb;
  const int addr = d->mb.addr;
  const uint32_t hv = hash3((uint32_t)e->cfg->seed + 5, (uint32_t)e->frame_abs, (uint32_t)addr);
  const int nref = d->sl.num_ref_idx[0];
  int skip_mv[2];
  h264i_pskip_mv(d, skip_mv);

  if (e->cfg->global_mv == 7) {
    /* --global-mv 7: the vector DIFFERENCES and reference indices follow from address and picture number; the vectors
     * are whatever the prediction of 8.4.1.3 makes of them (every fourth macroblock is P_Skip: 8.4.1.1).  Nothing is
     * coded, one 16x16 partition per macroblock: what the picture decodes to says whether the decoder predicts
     * vectors as the clause does (tests/test_mvpred_independent.py restates it) */
    const int k = e->frame_abs;
    p->kind = MB_INTER;
    p->mb_type_raw = 0;
    if (addr % 4 == 3) {
      set_all_motion(&d->mb, 0, 0, skip_mv[0], skip_mv[1]);
      plan->is_skip = 1;
    } else if (addr % 7 == 5) {
      set_all_motion(&d->mb, 0, 0, 0, 0); /* (reference 0, vector 0: what the P_Skip rule looks for in a neighbour) */
    } else if ((addr * 3 + k) % 5 == 4) {
      /* P_8x8: a reference per quadrant, sub-macroblock types 8x8 / 8x4 / 4x8 / 4x4 by address, quadrant and picture
       * number, a difference per sub-macroblock partition; all four reference indices first (as the syntax has them:
       * a later quadrant's index is known while its vectors are not), then the partitions in decoding order */
      for (int q = 0; q < 4; ++q) h264i_set_ref8(d, q, (addr + q + k) % nref);
      for (int q = 0; q < 4; ++q) {
        const int st = (addr + q * 3 + k) % 4, r = (addr + q + k) % nref;
        const int pw = (st == 0 || st == 1) ? 2 : 1, ph = (st == 0 || st == 2) ? 2 : 1; /* 0 8x8, 1 8x4, 2 4x8, 3 4x4 */
        int n = 0;
        p->sub_type[q] = st;
        for (int sy = 0; sy < 2; sy += ph)
          for (int sx = 0; sx < 2; sx += pw, ++n) {
            const int bx = (q & 1) * 2 + sx, by = (q >> 1) * 2 + sy;
            int mvp[2];
            h264i_pred_mv_blk(d, bx, by, pw, r, mvp);
            h264i_set_blk(d, bx, by, pw, ph, mvp[0] + (addr * 5 + k * 3 + q * 4 + n) % 13 - 6, mvp[1] + (addr * 7 + k + q * 2 + n * 3) % 9 - 4);
          }
      }
      p->mb_type_raw = 3;
    } else if ((addr + k) % 3 != 0) {
      /* two partitions, 16x8 or 8x16: each its own reference and difference; the second one's prediction sees the first */
      const int shape = (addr + k) % 3;
      for (int part = 0; part < 2; ++part) {
        const int r = (addr * 3 + k + part) % nref;
        int mvp[2];
        h264i_pred_mv_part(d, shape, part, r, mvp);
        h264i_set_part(d, shape, part, r, mvp[0] + (addr * 5 + k * 3 + part * 4) % 13 - 6, mvp[1] + (addr * 7 + k + part * 2) % 9 - 4);
      }
      p->mb_type_raw = shape;
    } else {
      const int r = (addr * 3 + k) % nref;
      int mvp[2];
      h264i_pred_mv16(d, 0, r, mvp);
      set_all_motion(&d->mb, 0, r, mvp[0] + (addr * 5 + k * 3) % 13 - 6, mvp[1] + (addr * 7 + k) % 9 - 4);
    }
    copy_motion_from_state(d, p);
    h264i_inter_pred_mb(d);
    p->qp = qp;
    return;
  }
  if (e->cfg->global_mv) {
    int gx, gy;
    global_mv_of(e->frame_abs, e->cfg->width, e->cfg->height, &gx, &gy);
    /* --global-mv 2: a checkerboard of two vectors - every macroblock edge inside the picture then separates different
     * motion and gets boundary strength 1 (nothing is coded, all macroblocks keep the slice's QP): the in-loop filter
     * runs on known edges with known parameters */
    if (e->cfg->global_mv == 2 && ((d->mb.mbx + d->mb.mby) & 1)) {
      gx += 9;
      gy -= 6;
    }
    /* --global-mv 10: every third anchor hardly moves - (1, -1) - so that for the B pictures in front of it the co-located
     * macroblocks "do not move" (8.4.1.2.2: colZeroFlag) */
    if (e->cfg->global_mv == 10 && e->frame_abs % 9 == 0) {
      gx = 1;
      gy = -1;
    }
    p->kind = MB_INTER;
    set_all_motion(&d->mb, 0, 0, gx, gy);
    copy_motion_from_state(d, p);
    p->mb_type_raw = 0;
    h264i_inter_pred_mb(d);
    p->qp = qp; /* (plan is zeroed: no coefficients, cbp 0) */
    /* --global-mv 3: a checkerboard of such macroblocks and I_PCM macroblocks whose samples are what the vector predicts
     * there plus a small known pattern (global_pcm_delta): every macroblock edge inside the picture has an intra
     * macroblock on one side - boundary strength 4 - with QP 0 on that side (8.7.2.2) and the slice's QP on the other */
    if (e->cfg->global_mv == 3 && ((d->mb.mbx + d->mb.mby) & 1)) {
      const Picture* cur = d->cur;
      p->kind = MB_IPCM;
      p->mb_type_raw = 5 + 25;
      for (int y = 0; y < 16; ++y)
        for (int x = 0; x < 16; ++x) {
          const int X = d->mb.mbx * 16 + x, Y = d->mb.mby * 16 + y;
          p->pcm[y * 16 + x] = (uint8_t)iclip(0, 255, cur->plane[0][Y * cur->stride[0] + X] + global_pcm_delta(e->frame_abs, 0, X, Y));
        }
      for (int pl = 0; pl < 2; ++pl)
        for (int y = 0; y < 8; ++y)
          for (int x = 0; x < 8; ++x) {
            const int X = d->mb.mbx * 8 + x, Y = d->mb.mby * 8 + y;
            p->pcm[256 + pl * 64 + y * 8 + x] =
                (uint8_t)iclip(0, 255, cur->plane[1 + pl][Y * cur->stride[1 + pl] + X] + global_pcm_delta(e->frame_abs, 1 + pl, X, Y));
          }
      return;
    }
    /* --global-mv 4: the prediction plus a residual whose levels follow from position and picture number alone
     * (global_level): dequantisation and the 4x4 inverse transforms (luma, chroma AC, chroma DC) on known input.  The
     * macroblock's QP is the slice's (no mb_qp_delta worth the name), the 8x8 transform stays off */
    /* --global-mv 5: as 4, but every other macroblock of a checkerboard is an Intra16x16 macroblock with known levels
     * (the luma DC levels of comp 6 and the AC levels of comp 0) whose prediction modes follow from address and picture
     * number where the neighbours allow them: Intra16x16 and chroma prediction from neighbouring inter macroblocks, the
     * luma DC path (8.5.10) */
    if (e->cfg->global_mv == 5 && ((d->mb.mbx + d->mb.mby) & 1)) {
      const int k = e->frame_abs, left = d->mb.mbx > 0, top = d->mb.mby > 0;
      int mode = (addr + k) % 4, cmode = (addr * 3 + k) % 4, any_ac = 0, chroma = 0;
      if ((mode == 0 && !top) || (mode == 1 && !left) || (mode == 3 && !(left && top))) mode = 2;
      if ((cmode == 1 && !left) || (cmode == 2 && !top) || (cmode == 3 && !(left && top))) cmode = 0;
      p->kind = MB_I16x16;
      p->i16_mode = mode;
      p->chroma_mode = cmode;
      for (int pos = 0; pos < 16; ++pos) p->luma_dc[pos] = (int16_t)global_level(k, addr, 6, 0, pos);
      for (int blk = 0; blk < 16; ++blk)
        for (int pos = 1; pos < 16; ++pos) {
          p->luma[blk][pos] = (int16_t)global_level(k, addr, 0, blk, pos);
          if (p->luma[blk][pos]) any_ac = 1;
        }
      for (int pl = 0; pl < 2; ++pl)
        for (int blk = 0; blk < 4; ++blk) {
          p->chroma_dc[pl][blk] = (int16_t)global_level(k, addr, 3 + pl, 0, blk);
          if (p->chroma_dc[pl][blk] && chroma < 1) chroma = 1;
          for (int pos = 1; pos < 16; ++pos) {
            p->chroma_ac[pl][blk][pos] = (int16_t)global_level(k, addr, 1 + pl, blk, pos);
            if (p->chroma_ac[pl][blk][pos]) chroma = 2;
          }
        }
      p->cbp = (any_ac ? 15 : 0) | chroma << 4;
      p->mb_type_raw = 5 + 1 + mode + 4 * chroma + (any_ac ? 12 : 0);
      p->qp = e->cfg->qp;
      return;
    }
    /* --global-mv 6: as 5 with Intra4x4 macroblocks: each 4x4 block's prediction mode follows from address, block and
     * picture number (inside the picture all nine are legal everywhere; a macroblock on the left or top border takes DC
     * throughout), levels of comp 0 incl. position 0 */
    if (e->cfg->global_mv == 6 && ((d->mb.mbx + d->mb.mby) & 1)) {
      const int k = e->frame_abs, inside = d->mb.mbx > 0 && d->mb.mby > 0;
      int cmode = (addr * 3 + k) % 4, cbp = 0, chroma = 0;
      if ((cmode == 1 && d->mb.mbx == 0) || (cmode == 2 && d->mb.mby == 0) || (cmode == 3 && !inside)) cmode = 0;
      p->kind = MB_I4x4;
      p->mb_type_raw = 5;
      p->chroma_mode = cmode;
      for (int blk = 0; blk < 16; ++blk) {
        p->i4_mode[blk] = (int8_t)(inside ? (addr * 5 + blk * 7 + k) % 9 : 2);
        for (int pos = 0; pos < 16; ++pos) {
          p->luma[blk][pos] = (int16_t)global_level(k, addr, 0, blk, pos);
          if (p->luma[blk][pos]) cbp |= 1 << ((blk >> 3) * 2 + ((blk & 3) >> 1));
        }
      }
      for (int pl = 0; pl < 2; ++pl)
        for (int blk = 0; blk < 4; ++blk) {
          p->chroma_dc[pl][blk] = (int16_t)global_level(k, addr, 3 + pl, 0, blk);
          if (p->chroma_dc[pl][blk] && chroma < 1) chroma = 1;
          for (int pos = 1; pos < 16; ++pos) {
            p->chroma_ac[pl][blk][pos] = (int16_t)global_level(k, addr, 1 + pl, blk, pos);
            if (p->chroma_ac[pl][blk][pos]) chroma = 2;
          }
        }
      p->cbp = cbp | chroma << 4;
      p->qp = e->cfg->qp;
      return;
    }
    /* --global-mv 8 (needs --t8x8 1): as 6 with Intra8x8 macroblocks - one mode per 8x8 block, the luma levels of comp 5 */
    if (e->cfg->global_mv == 8 && ((d->mb.mbx + d->mb.mby) & 1)) {
      const int k = e->frame_abs, inside = d->mb.mbx > 0 && d->mb.mby > 0;
      int cmode = (addr * 3 + k) % 4, cbp = 0, chroma = 0;
      if ((cmode == 1 && d->mb.mbx == 0) || (cmode == 2 && d->mb.mby == 0) || (cmode == 3 && !inside)) cmode = 0;
      p->kind = MB_I8x8;
      p->t8x8 = 1;
      p->mb_type_raw = 5;
      p->chroma_mode = cmode;
      for (int b8 = 0; b8 < 4; ++b8) {
        const int mode = inside ? (addr * 5 + b8 * 7 + k) % 9 : 2;
        for (int s4 = 0; s4 < 4; ++s4) p->i4_mode[((b8 >> 1) * 2 + (s4 >> 1)) * 4 + (b8 & 1) * 2 + (s4 & 1)] = (int8_t)mode;
        for (int pos = 0; pos < 64; ++pos) {
          p->luma8[b8][pos] = (int16_t)global_level(k, addr, 5, b8, pos);
          if (p->luma8[b8][pos]) cbp |= 1 << b8;
        }
      }
      for (int pl = 0; pl < 2; ++pl)
        for (int blk = 0; blk < 4; ++blk) {
          p->chroma_dc[pl][blk] = (int16_t)global_level(k, addr, 3 + pl, 0, blk);
          if (p->chroma_dc[pl][blk] && chroma < 1) chroma = 1;
          for (int pos = 1; pos < 16; ++pos) {
            p->chroma_ac[pl][blk][pos] = (int16_t)global_level(k, addr, 1 + pl, blk, pos);
            if (p->chroma_ac[pl][blk][pos]) chroma = 2;
          }
        }
      p->cbp = cbp | chroma << 4;
      p->qp = e->cfg->qp;
      return;
    }
    if (e->cfg->global_mv >= 4 && e->cfg->global_mv <= 8) {
      const int k = e->frame_abs;
      int cbp = 0, chroma = 0;
      if (e->cfg->t8x8) { /* (--t8x8 1: the luma levels are those of four 8x8 blocks, comp 5, raster positions 0..63) */
        p->t8x8 = 1;
        for (int b8 = 0; b8 < 4; ++b8)
          for (int pos = 0; pos < 64; ++pos) {
            p->luma8[b8][pos] = (int16_t)global_level(k, addr, 5, b8, pos);
            if (p->luma8[b8][pos]) cbp |= 1 << b8;
          }
      } else
      for (int blk = 0; blk < 16; ++blk)
        for (int pos = 0; pos < 16; ++pos) {
          p->luma[blk][pos] = (int16_t)global_level(k, addr, 0, blk, pos);
          if (p->luma[blk][pos]) cbp |= 1 << ((blk >> 3) * 2 + ((blk & 3) >> 1));
        }
      for (int pl = 0; pl < 2; ++pl) {
        for (int blk = 0; blk < 4; ++blk) {
          p->chroma_dc[pl][blk] = (int16_t)global_level(k, addr, 3 + pl, 0, blk);
          if (p->chroma_dc[pl][blk] && chroma < 1) chroma = 1;
          for (int pos = 1; pos < 16; ++pos) {
            p->chroma_ac[pl][blk][pos] = (int16_t)global_level(k, addr, 1 + pl, blk, pos);
            if (p->chroma_ac[pl][blk][pos]) chroma = 2;
          }
        }
      }
      p->cbp = cbp | chroma << 4;
      p->qp = e->cfg->qp;
      return;
    }
    if (gx == skip_mv[0] && gy == skip_mv[1]) plan->is_skip = 1;
    return;
  }

  /* candidates: (ref, mvx, mvy) */
  int cand[40][3], nc = 0;
#define ADD(r, x, y) do { if (nc < 40 && mv_ok(e, (x), (y))) { cand[nc][0] = (r); cand[nc][1] = (x); cand[nc][2] = (y); ++nc; } } while (0)
  ADD(0, skip_mv[0], skip_mv[1]);
  ADD(0, 0, 0);
  {
    int mvp[2];
    h264i_pred_mv16(d, 0, 0, mvp);
    ADD(0, mvp[0], mvp[1]);
    const int dd0 = (d->cur->poc - d->sl.ref_list[0][0]->poc) / 2;
    ADD(0, e->hints[0][0] * dd0, e->hints[0][1] * dd0); /* background */
  }
  const int n_first = nc;
  for (int r = 0; r < nref; ++r) {
    const Picture* rp = d->sl.ref_list[0][r];
    const int dd = (d->cur->poc - rp->poc) / 2;
    if (r > 0) {
      int mvp[2];
      h264i_pred_mv16(d, 0, r, mvp);
      ADD(r, mvp[0], mvp[1]);
    }
    for (int h = r > 0 ? 0 : 1; h < e->hint_n; ++h) ADD(r, e->hints[h][0] * dd, e->hints[h][1] * dd);
  }
  (void)dist;
  /* stage 1: the usual suspects; the long list only when none of them fits
   * (film grain alone costs about 256 * 2.3 of SAD) */
  int best = 0, bc = 1 << 30;
  for (int i = 0; i < nc; ++i) {
    if (i == n_first && bc < 256 * 3) break;
    set_all_motion(&d->mb, 0, cand[i][0], cand[i][1], cand[i][2]);
    int c = eval_inter(e) + (i == 0 ? 0 : 24);
    if (c < bc) {
      bc = c;
      best = i;
    }
  }
  int br = cand[best][0], bx = cand[best][1], by = cand[best][2];
  /* quarter-sample refinement when the match is still poor */
  for (int it = 0; it < 2 && bc > 256 * 3; ++it) {
    static const int dx[4] = {1, -1, 0, 0}, dy[4] = {0, 0, 1, -1};
    int improved = 0;
    for (int k = 0; k < 4; ++k) {
      if (!mv_ok(e, bx + dx[k], by + dy[k])) continue;
      set_all_motion(&d->mb, 0, br, bx + dx[k], by + dy[k]);
      int c = eval_inter(e) + 24;
      if (c < bc) {
        bc = c;
        bx += dx[k];
        by += dy[k];
        improved = 1;
        break;
      }
    }
    if (!improved) break;
  }

  /* intra fallback for uncovered content */
  if (bc > 256 * 10) {
    MbPlan ip;
    memset(&ip, 0, sizeof(ip));
    h264i_begin_mb(d, addr);
    int ic = decide_intra(e, &ip, qp, d->sl.type);
    if (ic + 256 < bc) {
      *plan = ip;
      return;
    }
    h264i_begin_mb(d, addr);
  }

  if (e->cfg->wild_mv && hv % 17 == 5) {
    br = (int)((hv >> 3) % (uint32_t)nref);
    bx = (int)((hv >> 5) % 3999) - 1999;
    by = (int)((hv >> 17) % 1999) - 999;
    bc = 0; /* keep it one 16x16 partition */
  }
  p->kind = MB_INTER;
  int shape = 0; /* 0 16x16, 1 16x8, 2 8x16, 3 8x8 */
  set_all_motion(&d->mb, 0, br, bx, by);
  /* partitioned motion: when the 16x16 match is poor, or on a hash-chosen
   * subset (syntax coverage), give every 8x8 its own best candidate */
  if (bc > 256 * 4 || hv % 11 == 0) {
    int pm[4][3], qb[4] = {1 << 30, 1 << 30, 1 << 30, 1 << 30};
    for (int i = 0; i < nc + 1; ++i) {
      int r = i < nc ? cand[i][0] : br, x = i < nc ? cand[i][1] : bx, y = i < nc ? cand[i][2] : by;
      set_all_motion(&d->mb, 0, r, x, y);
      d->pred_luma_only = 1;
      h264i_inter_pred_mb(d);
      d->pred_luma_only = 0;
      for (int q = 0; q < 4; ++q) {
        int c = sad_block(e, d->mb.mbx * 16 + (q & 1) * 8, d->mb.mby * 16 + (q >> 1) * 8, 8) + (i == nc ? -16 : 0);
        if (c < qb[q]) {
          qb[q] = c;
          pm[q][0] = r; pm[q][1] = x; pm[q][2] = y;
        }
      }
    }
    int same01 = !memcmp(pm[0], pm[1], sizeof(pm[0])), same23 = !memcmp(pm[2], pm[3], sizeof(pm[0]));
    int same02 = !memcmp(pm[0], pm[2], sizeof(pm[0])), same13 = !memcmp(pm[1], pm[3], sizeof(pm[0]));
    if (same01 && same23 && same02) shape = 0;
    else if (same01 && same23) shape = 1;
    else if (same02 && same13) shape = 2;
    else shape = 3;
    if (shape == 0 && hv % 11 == 0) { /* force some partitioned syntax */
      shape = 1 + (int)((hv >> 5) % 3);
      pm[3][1] += (int)((hv >> 9) & 3) - 1;
      pm[3][2] += (int)((hv >> 11) & 3) - 1;
      if (shape == 1) memcpy(pm[2], pm[3], sizeof(pm[0]));
      if (shape == 2) memcpy(pm[1], pm[3], sizeof(pm[0]));
    }
    for (int q = 0; q < 4; ++q) {
      d->mb.ref[0][q] = (int8_t)pm[q][0];
      for (int s = 0; s < 4; ++s) {
        int k = ((q >> 1) * 2 + (s >> 1)) * 4 + (q & 1) * 2 + (s & 1);
        d->mb.mv[0][k][0] = (int16_t)pm[q][1];
        d->mb.mv[0][k][1] = (int16_t)pm[q][2];
      }
    }
    if (shape == 3) {
      for (int q = 0; q < 4; ++q) {
        int st = 0;
        uint32_t hq = hash32(hv + q * 977);
        if (hq % 5 == 0) { /* sub-8x8 partitions with slightly different vectors */
          st = 1 + (int)((hq >> 4) % 3);
          for (int s = 1; s < 4; ++s) {
            int k = ((q >> 1) * 2 + (s >> 1)) * 4 + (q & 1) * 2 + (s & 1);
            int k0 = ((q >> 1) * 2) * 4 + (q & 1) * 2;
            int jx = (int)((hq >> (6 + 2 * s)) & 3) - 1, jy = (int)((hq >> (14 + 2 * s)) & 3) - 1;
            if (st == 1 && (s & 1)) { d->mb.mv[0][k][0] = d->mb.mv[0][k - 1][0]; d->mb.mv[0][k][1] = d->mb.mv[0][k - 1][1]; continue; }
            if (st == 2 && (s >> 1)) { d->mb.mv[0][k][0] = d->mb.mv[0][k - 4][0]; d->mb.mv[0][k][1] = d->mb.mv[0][k - 4][1]; continue; }
            d->mb.mv[0][k][0] = (int16_t)(d->mb.mv[0][k0][0] + jx);
            d->mb.mv[0][k][1] = (int16_t)(d->mb.mv[0][k0][1] + jy);
          }
        }
        p->sub_type[q] = st;
      }
    }
  }
  copy_motion_from_state(d, p);
  p->mb_type_raw = shape == 0 ? 0 : (shape == 1 ? 1 : (shape == 2 ? 2 : 3));
  /* P_8x8ref0 (CAVLC only, 7.4.5): four quadrants that all use reference 0 need no ref_idx at all - every other time */
  if (shape == 3 && !d->apps->cabac && d->sl.num_ref_idx[0] > 1 && p->ref[0][0] == 0 && p->ref[0][1] == 0 && p->ref[0][2] == 0 &&
      p->ref[0][3] == 0 && ((hv >> 7) & 1))
    p->mb_type_raw = 4;
  h264i_inter_pred_mb(d);
  int sub8 = 1;
  if (shape == 3)
    for (int q = 0; q < 4; ++q) sub8 &= p->sub_type[q] == 0;
  p->t8x8 = e->cfg->t8x8 && sub8 && (hv >> 3) % 3 != 0;
  quant_inter(e, plan, qp, 1);
  /* P_Skip: 16x16, ref 0, predicted skip vector, nothing coded */
  if (shape == 0 && br == 0 && bx == skip_mv[0] && by == skip_mv[1] && p->cbp == 0) plan->is_skip = 1;
}

/* B macroblock */
static void decide_b(Enc* e, MbPlan* plan, int qp) {
  H264Oracle* d = e->d;
  MB* p = &plan->mb;
  const int addr = d->mb.addr;
  const uint32_t hv = hash3((uint32_t)e->cfg->seed + 9, (uint32_t)e->frame_abs, (uint32_t)addr);
  const Picture* r0 = d->sl.ref_list[0][0];
  const Picture* r1 = d->sl.ref_list[1][0];
  const int d0 = (d->cur->poc - r0->poc) / 2, d1 = (d->cur->poc - r1->poc) / 2; /* d1 < 0 */

  if (e->cfg->global_mv == 12) {
    /* --global-mv 12: B macroblocks of one 16x16 or two 16x8 / 8x16 partitions; which lists a partition uses (0, 1 or
     * both) and the vector DIFFERENCE per list follow from address, partition and picture number - the vectors are what
     * the prediction of 8.4.1.3 makes of them, list by list (a list a neighbour does not use counts as reference -1) */
    const int k = e->frame_abs, shape = (addr + k) % 3; /* 0 16x16, 1 16x8, 2 8x16 */
    int use[2];
    for (int part = 0; part < (shape ? 2 : 1); ++part) {
      use[part] = (addr * 2 + k + part) % 3; /* 0 list 0, 1 list 1, 2 both */
      for (int l = 0; l < 2; ++l) {
        const int on = use[part] == 2 || use[part] == l;
        int mvp[2] = {0, 0};
        if (on) h264i_pred_mv_part_l(d, l, shape, part, 0, mvp);
        h264i_set_part_l(d, l, shape, part, on ? 0 : -1, mvp[0] + (addr * 5 + k * 3 + part * 4 + l * 2) % 13 - 6,
                         mvp[1] + (addr * 7 + k + part * 2 + l * 5) % 9 - 4);
      }
    }
    static const int tab[3][3] = {{4, 8, 12}, {10, 6, 14}, {16, 18, 20}}; /* [first partition][second] -> the 16x8 type */
    p->kind = MB_INTER;
    p->mb_type_raw = shape == 0 ? 1 + use[0] : tab[use[0]][use[1]] + (shape == 2 ? 1 : 0);
    copy_motion_from_state(d, p);
    h264i_inter_pred_mb(d);
    p->qp = qp;
    return;
  }
  if (e->cfg->global_mv == 10 && ((d->mb.mbx + d->mb.mby) & 1)) {
    /* --global-mv 10: a checkerboard of coded macroblocks (below: list 0, list 1 or both by address and picture number,
     * one vector per list and picture) and B_Skip macroblocks whose references and vectors SPATIAL direct prediction
     * derives from those neighbours (8.4.1.2.2) */
    if (h264i_pred_direct_mb(d) < 0) return;
    p->kind = MB_INTER;
    p->mb_type_raw = 0;
    copy_motion_from_state(d, p);
    h264i_inter_pred_mb(d);
    p->qp = qp;
    plan->is_skip = 1;
    return;
  }
  if (e->cfg->global_mv == 9) {
    /* every macroblock B_Skip: the vectors are what direct prediction (temporal with --temporal-direct 1) derives from
     * the co-located macroblocks of the anchor behind - which all carry that picture's one vector */
    if (h264i_pred_direct_mb(d) < 0) return;
    p->kind = MB_INTER;
    p->mb_type_raw = 0;
    copy_motion_from_state(d, p);
    h264i_inter_pred_mb(d);
    p->qp = qp;
    plan->is_skip = 1;
    return;
  }
  if (e->cfg->global_mv) {
    /* every macroblock one 16x16 partition from list 0, list 1 or both (by the picture's number), one vector per list
     * and picture, nothing coded */
    const int k = e->frame_abs, kind = 1 + (e->cfg->global_mv == 10 ? k + addr : k) % 3; /* 1 L0, 2 L1, 3 Bi */
    int m0[2], m1[2];
    global_mv_of(k, e->cfg->width, e->cfg->height, &m0[0], &m0[1]);
    global_mv_of(k + 17, e->cfg->width, e->cfg->height, &m1[0], &m1[1]);
    p->kind = MB_INTER;
    /* (--global-mv 10 with several references: the coded macroblocks name different ones, so that the B_Skip
     * macroblocks between them have something to choose from) */
    const int r0 = e->cfg->global_mv == 10 ? (addr / 2 + k) % d->sl.num_ref_idx[0] : 0;
    const int r1 = e->cfg->global_mv == 10 ? (addr / 2 + k + 1) % d->sl.num_ref_idx[1] : 0;
    set_all_motion(&d->mb, 0, kind != 2 ? r0 : -1, m0[0], m0[1]);
    set_all_motion(&d->mb, 1, kind != 1 ? r1 : -1, m1[0], m1[1]);
    p->mb_type_raw = kind;
    copy_motion_from_state(d, p);
    h264i_inter_pred_mb(d);
    p->qp = qp;
    return;
  }

  /* direct first: it also tells whether B_Skip is possible */
  if (h264i_pred_direct_mb(d) < 0) return;
  MB direct = d->mb;
  int best_kind = 0; /* 0 direct, 1 L0, 2 L1, 3 Bi */
  int bc = eval_inter(e);
  int bmv[2][2] = {{0, 0}, {0, 0}};
  for (int h = 0; h < e->hint_n + 1; ++h) {
    if (h == 1 && bc < 256 * 3) break; /* direct or the background already fits */
    int hx = h < e->hint_n ? e->hints[h][0] : 0, hy = h < e->hint_n ? e->hints[h][1] : 0;
    int m0[2] = {hx * d0, hy * d0}, m1[2] = {hx * d1, hy * d1};
    for (int kind = 1; kind <= 3; ++kind) {
      if (!mv_ok(e, m0[0], m0[1]) || !mv_ok(e, m1[0], m1[1])) continue;
      set_all_motion(&d->mb, 0, kind != 2 ? 0 : -1, m0[0], m0[1]);
      set_all_motion(&d->mb, 1, kind != 1 ? 0 : -1, m1[0], m1[1]);
      int c = eval_inter(e) + 40;
      if (c < bc) {
        bc = c;
        best_kind = kind;
        bmv[0][0] = m0[0]; bmv[0][1] = m0[1]; bmv[1][0] = m1[0]; bmv[1][1] = m1[1];
      }
    }
  }
  if (bc > 256 * 11) {
    MbPlan ip;
    memset(&ip, 0, sizeof(ip));
    h264i_begin_mb(d, addr);
    int ic = decide_intra(e, &ip, qp, 1);
    if (ic + 256 < bc) {
      *plan = ip;
      return;
    }
    h264i_begin_mb(d, addr);
  }
  p->kind = MB_INTER;
  if (best_kind == 0) {
    const uint32_t pick = hv % 13;
    if (pick == 0 || pick == 1) {
      /* B_8x8 mixing direct sub-blocks with explicit ones that repeat the
       * direct motion (legal, exercises sub_mb_type + ref/mvd syntax) */
      d->mb = direct;
      p->mb_type_raw = 22;
      for (int q = 0; q < 4; ++q) {
        int uses0 = direct.ref[0][q] >= 0, uses1 = direct.ref[1][q] >= 0;
        int explicit_ = ((hv >> (4 + q)) & 1);
        if (!explicit_) p->sub_type[q] = 0;
        else p->sub_type[q] = uses0 && uses1 ? 3 : (uses0 ? 1 : 2);
      }
      /* explicit 8x8 sub-blocks need one vector: take the direct one of
       * their first 4x4; every third explicit quadrant is split further (8x4, 4x8, 4x4: sub_mb_type 4..12) with
       * slightly different vectors per part, the same in both lists it uses */
      for (int q = 0; q < 4; ++q)
        if (p->sub_type[q]) {
          int k0 = ((q >> 1) * 2) * 4 + (q & 1) * 2;
          const uint32_t hs = hash3((uint32_t)e->cfg->seed + 17, hv, (uint32_t)q);
          const int split = hs % 3 == 0 ? 1 + (int)((hs >> 3) % 3) : 0; /* 0 8x8, 1 8x4, 2 4x8, 3 4x4 */
          for (int l = 0; l < 2; ++l)
            for (int s = 0; s < 4; ++s) {
              int k = ((q >> 1) * 2 + (s >> 1)) * 4 + (q & 1) * 2 + (s & 1);
              int jx = 0, jy = 0;
              /* the part a 4x4 block belongs to: 8x4 -> its row, 4x8 -> its column, 4x4 -> itself */
              const int part = split == 1 ? (s >> 1) : split == 2 ? (s & 1) : split == 3 ? s : 0;
              if (part) {
                jx = (int)((hs >> (6 + 2 * part)) & 3) - 1;
                jy = (int)((hs >> (14 + 2 * part)) & 3) - 1;
              }
              d->mb.mv[l][k][0] = (int16_t)(d->mb.mv[l][k0][0] + (s ? jx : 0));
              d->mb.mv[l][k][1] = (int16_t)(d->mb.mv[l][k0][1] + (s ? jy : 0));
            }
          if (e->cfg->cover && (hs >> 24) % 3 == 0) /* an explicit quadrant may name any reference: index 0 next to neighbours > 0 */
            for (int l = 0; l < 2; ++l)
              if (d->mb.ref[l][q] > 0) d->mb.ref[l][q] = 0;
          if (split) { /* L0 / L1 / Bi 8x8 (1, 2, 3) -> 8x4 (4, 6, 8), 4x8 (5, 7, 9), 4x4 (10, 11, 12) */
            const int t = p->sub_type[q];
            p->sub_type[q] = split == 3 ? 9 + t : 2 + 2 * t + (split == 2 ? 1 : 0);
          }
        }
      copy_motion_from_state(d, p);
    } else {
      d->mb = direct;
      p->mb_type_raw = 0;
      copy_motion_from_state(d, p);
    }
  } else {
    set_all_motion(&d->mb, 0, best_kind != 2 ? 0 : -1, bmv[0][0], bmv[0][1]);
    set_all_motion(&d->mb, 1, best_kind != 1 ? 0 : -1, bmv[1][0], bmv[1][1]);
    p->mb_type_raw = best_kind;
    if (hv % 7 == 0) {
      /* two-partition types: same motion in both halves; for Bi pick an
       * asymmetric L0 / Bi style split on some of them */
      int shape = 1 + (int)((hv >> 3) & 1); /* 1 16x8, 2 8x16 */
      int pa = best_kind - 1, pb = best_kind - 1; /* 0 L0 1 L1 2 Bi */
      if (best_kind == 3 && ((hv >> 5) & 1)) pb = (int)((hv >> 6) & 1);
      if (e->cfg->cover && best_kind != 3 && ((hv >> 5) & 1)) pb = (pa + 1 + (int)((hv >> 6) & 1)) % 3; /* L0_L1, L1_L0, L0_Bi, L1_Bi */
      static const int tab[3][3] = {{4, 8, 12}, {10, 6, 14}, {16, 18, 20}}; /* [predA][predB] -> 16x8 type */
      p->mb_type_raw = tab[pa][pb] + (shape == 2 ? 1 : 0);
      for (int q = 0; q < 4; ++q) {
        int second = shape == 1 ? (q >> 1) : (q & 1);
        int pr = second ? pb : pa;
        d->mb.ref[0][q] = (int8_t)(pr != 1 ? 0 : -1);
        d->mb.ref[1][q] = (int8_t)(pr != 0 ? 0 : -1);
        for (int s = 0; s < 4; ++s) {
          int k = ((q >> 1) * 2 + (s >> 1)) * 4 + (q & 1) * 2 + (s & 1);
          for (int l = 0; l < 2; ++l) {
            d->mb.mv[l][k][0] = (int16_t)(d->mb.ref[l][q] >= 0 ? bmv[l][0] : 0);
            d->mb.mv[l][k][1] = (int16_t)(d->mb.ref[l][q] >= 0 ? bmv[l][1] : 0);
          }
        }
      }
    }
    copy_motion_from_state(d, p);
  }
  h264i_inter_pred_mb(d);
  p->t8x8 = e->cfg->t8x8 && (hv >> 9) % 3 != 0 && (p->mb_type_raw != 22 || 1);
  quant_inter(e, plan, qp, 1);
  if (best_kind == 0 && p->mb_type_raw == 0 && p->cbp == 0) plan->is_skip = 1;
}

/* ------------------------------------------------------------------------- */
/* picture / GOP                                                             */
/* ------------------------------------------------------------------------- */
typedef struct {
  int type;      /* 0 P 1 B 2 I(IDR) */
  int display;   /* index inside GOP */
  int frame_num;
  int is_ref;
  int ref_ordinal; /* k-th reference picture of the GOP (IDR = 0) */
  int mmco5;       /* this picture carries memory_management_control_operation 5 */
  int poc_base;    /* display index of the last MMCO-5 picture before this one (POC restarts there) */
} PicPlan;

/* dec_ref_pic_marking of one picture (identical in all its slices) */
typedef struct {
  int idr_long;    /* IDR: long_term_reference_flag */
  int adaptive;
  int n;
  int op[12], a[12], b[12];
} Marking;

/* The reference pictures in the oracle's DPB as the slice header syntax names
 * them: PicNum of the short-term ones (8.2.4.1, relative to the current
 * frame_num), LongTermPicNum of the long-term ones. */
typedef struct {
  int n_st, st_pn[MAX_DPB];
  int n_lt, lt_idx[MAX_DPB];
  int max_lt;
} DpbView;

static void dpb_view(const H264Oracle* d, int cur_frame_num, DpbView* v) {
  memset(v, 0, sizeof(*v));
  const int max_fn = 1 << d->sps[0].log2_max_frame_num;
  for (int i = 0; i < MAX_DPB; ++i) {
    const Picture* p = &d->pool[i];
    if (!p->in_pool_use || p == d->cur) continue;
    if (p->ref == 1) v->st_pn[v->n_st++] = p->frame_num > cur_frame_num ? p->frame_num - max_fn : p->frame_num;
    else if (p->ref == 2) v->lt_idx[v->n_lt++] = p->long_term_idx;
  }
  v->max_lt = d->max_long_term_frame_idx;
}

static int dpb_refs(const H264Oracle* d) {
  int n = 0;
  for (int i = 0; i < MAX_DPB; ++i) n += d->pool[i].in_pool_use && &d->pool[i] != d->cur && d->pool[i].ref != 0;
  return n;
}

/* Adaptive marking script.  Keeps the standard's constraint that at most
 * max_num_ref_frames pictures are marked after the current one is added, and
 * walks through every operation over the reference pictures of a GOP. */
static void plan_marking(const Enc* e, const PicPlan* pp, Marking* m) {
  const Cfg* c = e->cfg;
  memset(m, 0, sizeof(*m));
  if (!c->mmco || !pp->is_ref) return;
  if (pp->type == 2) {
    m->idr_long = (e->frame_abs / c->gop) & 1;
    return;
  }
  DpbView v;
  dpb_view(e->d, pp->frame_num, &v);
  const int maxrefs = c->refs, cur = pp->frame_num;
  const uint32_t h = hash3((uint32_t)c->seed + 21, (uint32_t)e->frame_abs, 7);
#define OP(o, x, y) do { m->op[m->n] = (o); m->a[m->n] = (x); m->b[m->n] = (y); m->n++; } while (0)
  int cur_long = 0, cur_idx = -1;
  if (pp->mmco5) {
    OP(5, 0, 0);
    m->adaptive = 1;
    return;
  }
  /* helpers on the simulated state */
  int oldest, newest;
#define FIND_ST() do { oldest = newest = -1; for (int i = 0; i < v.n_st; ++i) { if (oldest < 0 || v.st_pn[i] < v.st_pn[oldest]) oldest = i; if (newest < 0 || v.st_pn[i] > v.st_pn[newest]) newest = i; } } while (0)
#define DROP_ST(i) do { v.st_pn[(i)] = v.st_pn[--v.n_st]; } while (0)
#define DROP_LT_IDX(x) do { for (int q = 0; q < v.n_lt; ++q) if (v.lt_idx[q] == (x)) { v.lt_idx[q] = v.lt_idx[--v.n_lt]; --q; } } while (0)
  switch (pp->ref_ordinal % 6) {
    case 1: /* oldest short-term picture becomes long-term */
      FIND_ST();
      if (oldest >= 0) {
        if (v.max_lt < 1) {
          OP(4, 2, 0);
          v.max_lt = 1;
        }
        int idx = 0;
        for (int q = 0; q < v.n_lt; ++q) if (v.lt_idx[q] == 0) idx = 1;
        OP(3, cur - v.st_pn[oldest] - 1, idx);
        DROP_LT_IDX(idx);
        DROP_ST(oldest);
        v.lt_idx[v.n_lt++] = idx;
      }
      break;
    case 2: { /* the current picture becomes long-term */
      if (v.max_lt < 0) {
        OP(4, 1, 0);
        v.max_lt = 0;
      }
      const int idx = v.max_lt >= 1 ? (int)(h & 1) : 0;
      OP(6, idx, 0);
      DROP_LT_IDX(idx);
      cur_long = 1;
      cur_idx = idx;
      v.lt_idx[v.n_lt++] = idx;
      break;
    }
    case 3: /* sliding window */
      break;
    case 4: /* a long-term picture goes, else the newest short-term one */
      FIND_ST();
      if (v.n_lt) {
        const int idx = v.lt_idx[0];
        OP(2, idx, 0);
        DROP_LT_IDX(idx);
      } else if (newest >= 0 && v.n_st >= 2) {
        OP(1, cur - v.st_pn[newest] - 1, 0);
        DROP_ST(newest);
      }
      break;
    case 5: /* the second newest short-term picture goes */
      if (v.n_st >= 2) {
        FIND_ST();
        int second = -1;
        for (int i = 0; i < v.n_st; ++i)
          if (i != newest && (second < 0 || v.st_pn[i] > v.st_pn[second])) second = i;
        OP(1, cur - v.st_pn[second] - 1, 0);
        DROP_ST(second);
      }
      break;
    default: { /* max long-term index shrinks */
      const int keep = (int)(h & 1); /* MaxLongTermFrameIdx = keep - 1 */
      if (v.n_lt || v.max_lt >= keep) {
        OP(4, keep, 0);
        for (int q = 0; q < v.n_lt; ++q)
          if (v.lt_idx[q] > keep - 1) {
            v.lt_idx[q] = v.lt_idx[--v.n_lt];
            --q;
          }
        v.max_lt = keep - 1;
      }
      break;
    }
  }
  if (m->n == 0) {
    /* sliding window: legal when it can find a short-term picture to drop, or nothing has to go */
    if (v.n_st + v.n_lt < maxrefs || v.n_st > 0) return;
  }
  /* make room for the current picture (adaptive mode has no sliding window) */
  while (v.n_st + v.n_lt + (cur_long ? 0 : 1) > maxrefs || (cur_long && v.n_st + v.n_lt > maxrefs)) {
    FIND_ST();
    if (oldest >= 0) {
      OP(1, cur - v.st_pn[oldest] - 1, 0);
      DROP_ST(oldest);
    } else {
      int big = -1;
      for (int q = 0; q < v.n_lt; ++q)
        if (v.lt_idx[q] != cur_idx && (big < 0 || v.lt_idx[q] > v.lt_idx[big])) big = q;
      if (big < 0) break; /* only the current picture is left */
      OP(2, v.lt_idx[big], 0);
      v.lt_idx[big] = v.lt_idx[--v.n_lt];
    }
  }
  m->adaptive = m->n > 0;
#undef OP
#undef FIND_ST
#undef DROP_ST
#undef DROP_LT_IDX
}

/* ref_pic_list_modification (7.3.3.1) for list `list` with `nactive` entries:
 * moves one or two hash-chosen reference pictures (short-term by PicNum
 * difference, both signs; long-term by LongTermPicNum) to the front. */
static void write_rplm(const Enc* e, BitWriter* w, const PicPlan* pp, int nactive, uint32_t h) {
  DpbView v;
  dpb_view(e->d, pp->frame_num, &v);
  if (!e->cfg->rplm || v.n_st + v.n_lt < 2 || h % 3 == 0) {
    bw_put(w, 0, 1);
    return;
  }
  bw_put(w, 1, 1);
  const int max_pn = 1 << e->d->sps[0].log2_max_frame_num;
  int pred = pp->frame_num; /* picNumLXPred, kept in [0, MaxPicNum) like the decoder does */
  const int nops = 1 + (int)((h >> 4) % (unsigned)imin(2, nactive));
  int used_st[MAX_DPB] = {0}, used_lt[MAX_DPB] = {0};
  for (int k = 0; k < nops; ++k) {
    const uint32_t hk = hash32(h + 77u * (uint32_t)k);
    int done = 0;
    if (v.n_lt && ((hk & 3) == 0 || !v.n_st)) {
      for (int t = 0; t < v.n_lt && !done; ++t) {
        const int j = (int)((hk >> 3) + t) % v.n_lt;
        if (used_lt[j]) continue;
        used_lt[j] = 1;
        bw_ue(w, 2);
        bw_ue(w, v.lt_idx[j]);
        done = 1;
      }
    }
    for (int t = 0; t < v.n_st && !done; ++t) {
      const int j = (int)((hk >> 3) + t) % v.n_st;
      if (used_st[j]) continue;
      const int target = v.st_pn[j] < 0 ? v.st_pn[j] + max_pn : v.st_pn[j];
      if (target == pred) continue;
      used_st[j] = 1;
      if (target < pred) {
        bw_ue(w, 0);
        bw_ue(w, pred - target - 1);
      } else {
        bw_ue(w, 1);
        bw_ue(w, target - pred - 1);
      }
      pred = target;
      done = 1;
    }
  }
  bw_ue(w, 3);
}

/* pred_weight_table (7.3.3.2): per reference a hash-chosen mix of flags,
 * weights around 2^denom and small offsets; chroma uses its own denominator. */
static void write_pwt(const Enc* e, BitWriter* w, const PicPlan* pp, int nl0, int nl1, uint32_t h) {
  const int ld = e->cfg->wp_denom, cd = (ld + 6) & 7;
  bw_ue(w, ld);
  bw_ue(w, cd);
  if (e->cfg->global_mv) {
    /* weights and offsets that follow from the picture's number and the list alone (tests/test_inter_independent.py
     * restates them): every entry of a list the same.  B slices: w0 + w1 <= 128 (127 at denominator 7), hence the caps */
    for (int l = 0; l < (pp->type == 1 ? 2 : 1); ++l) {
      const int k = e->frame_abs + 7 * l;
      const int cap_l = pp->type == 1 && ld >= 6 ? (ld == 7 ? 63 : 64) : 127;
      const int cap_c = pp->type == 1 && cd >= 6 ? (cd == 7 ? 63 : 64) : 127;
      for (int i = 0; i < (l ? nl1 : nl0); ++i) {
        bw_put(w, 1, 1);
        bw_se(w, iclip(-128, cap_l, (1 << ld) + (ld >= 3 ? (k * 3) % 7 - 3 : (ld >= 1 ? k % 3 - 1 : 0))));
        bw_se(w, (k * 5) % 11 - 5);
        bw_put(w, 1, 1);
        for (int pl = 0; pl < 2; ++pl) {
          bw_se(w, iclip(-128, cap_c, (1 << cd) + (cd >= 3 ? (k + pl) % 5 - 2 : (cd >= 1 ? (k + pl) % 3 - 1 : 0))));
          bw_se(w, (k * 3 + pl * 2) % 9 - 4);
        }
      }
    }
    return;
  }
  for (int l = 0; l < (pp->type == 1 ? 2 : 1); ++l)
    for (int i = 0; i < (l ? nl1 : nl0); ++i) {
      const uint32_t hi = hash3(h, (uint32_t)(l * 32 + i), 5);
      /* B slices: w0 + w1 <= 128 (127 at denominator 7) */
      const int cap_l = pp->type == 1 && ld >= 6 ? (ld == 7 ? 63 : 64) : 127;
      const int cap_c = pp->type == 1 && cd >= 6 ? (cd == 7 ? 63 : 64) : 127;
      const int lf = (hi & 3) != 0;
      bw_put(w, lf, 1);
      if (lf) {
        int wv = (1 << ld) + (ld >= 3 ? (int)((hi >> 2) % 7) - 3 : (ld >= 1 ? (int)((hi >> 2) % 3) - 1 : 0));
        bw_se(w, iclip(-128, cap_l, wv));
        bw_se(w, (int)((hi >> 8) % 11) - 5);
      }
      const int cf = ((hi >> 12) & 3) != 0;
      bw_put(w, cf, 1);
      if (cf)
        for (int pl = 0; pl < 2; ++pl) {
          int wv = (1 << cd) + (cd >= 3 ? (int)((hi >> (14 + 3 * pl)) % 5) - 2 : (cd >= 1 ? (int)((hi >> (14 + 3 * pl)) % 3) - 1 : 0));
          bw_se(w, iclip(-128, cap_c, wv));
          bw_se(w, (int)((hi >> (20 + 4 * pl)) % 9) - 4);
        }
    }
}

/* st: this slice's type (the picture's, or a "lower" one with --mixed-slices); pp->type == 2 still means an IDR picture */
static void write_slice_header(Enc* e, BitWriter* w, const PicPlan* pp, int st, int first_mb, int nrefs_l0, const Marking* mk) {
  const Cfg* c = e->cfg;
  PicPlan as_slice = *pp;
  as_slice.type = st;
  const PicPlan* sp = &as_slice;
  const uint32_t hs = hash3((uint32_t)c->seed + 13, (uint32_t)e->frame_abs, (uint32_t)first_mb);
  const int pps_id = c->multi_pps ? (e->coded_index & 1) : 0;
  bw_ue(w, first_mb);
  bw_ue(w, st);
  bw_ue(w, pps_id);
  const int fn_mask = (1 << c->fn_bits) - 1;
  bw_put(w, pp->frame_num & fn_mask, c->fn_bits);
  if (pp->type == 2) bw_ue(w, (e->frame_abs / c->gop) & 1); /* idr_pic_id */
  if (c->poc_type == 0) {
    bw_put(w, ((pp->display - pp->poc_base) * 2) & ((1 << c->poc_bits) - 1), c->poc_bits);
  } else if (c->poc_type == 1) {
    /* 8.2.1.2 with a one-frame cycle: expected = absFrameNum * offset_for_ref_frame[0] (+ offset_for_non_ref_pic);
     * delta_pic_order_cnt[0] makes up the difference to 2 * display */
    if (first_mb == 0) {
      if (pp->type == 2) e->poc1_fn_offset = 0;
      else if (e->poc1_prev_fn > (pp->frame_num & fn_mask)) e->poc1_fn_offset += fn_mask + 1;
      e->poc1_prev_fn = pp->frame_num & fn_mask;
    }
    int abs_fn = pp->type == 2 ? 0 : e->poc1_fn_offset + (pp->frame_num & fn_mask);
    if (!pp->is_ref && abs_fn > 0) --abs_fn;
    int expected = abs_fn > 0 ? abs_fn * 2 * (c->bframes + 1) : 0;
    if (!pp->is_ref) expected += -1;
    bw_se(w, (pp->display - pp->poc_base) * 2 - expected);
  }
  if (st == 1) bw_put(w, c->temporal_direct ? 0 : 1, 1);
  if (st != 2) {
    /* B slices predict from one picture per list; with a reference B picture as the co-located one, temporal direct
     * may map to any picture it references, so list 0 then carries everything the DPB holds */
    const int nl0 = st == 1 ? ((c->bpyramid && c->temporal_direct) ? imin(c->refs, dpb_refs(e->d)) : 1) : nrefs_l0;
    bw_put(w, 1, 1); /* num_ref_idx_active_override */
    bw_ue(w, nl0 - 1);
    if (st == 1) bw_ue(w, 0);
    /* temporal direct needs the co-located block's reference in list 0: leave B lists alone there */
    const int mod_ok = st == 0 || !c->temporal_direct;
    if (mod_ok) write_rplm(e, w, sp, nl0, hs);
    else bw_put(w, 0, 1); /* ref_pic_list_modification_flag_l0 */
    if (st == 1) {
      if (mod_ok) write_rplm(e, w, sp, 1, hash32(hs + 3));
      else bw_put(w, 0, 1);
    }
    if (c->explicit_wp) write_pwt(e, w, sp, nl0, 1, hash32(hs + 9));
  }
  if (pp->is_ref) {
    if (pp->type == 2) {
      bw_put(w, 0, 1);
      bw_put(w, mk->idr_long, 1);
    } else {
      bw_put(w, mk->adaptive, 1);
      if (mk->adaptive) {
        for (int k = 0; k < mk->n; ++k) {
          const int op = mk->op[k];
          bw_ue(w, op);
          if (op == 1 || op == 3) bw_ue(w, mk->a[k]);
          if (op == 2) bw_ue(w, mk->a[k]);
          if (op == 3) bw_ue(w, mk->b[k]);
          if (op == 6) bw_ue(w, mk->a[k]);
          if (op == 4) bw_ue(w, mk->a[k]);
        }
        bw_ue(w, 0);
      }
    }
  }
  if (pps_cabac(c, pps_id) && st != 2) bw_ue(w, (e->frame_abs + first_mb) % 3);
  bw_se(w, c->qp - pps_init_qp(c, pps_id)); /* slice_qp_delta: the slice starts at --qp whatever the parameter set says */
  bw_ue(w, c->deblock_idc);
  if (c->deblock_idc != 1) {
    bw_se(w, c->alpha_div2);
    bw_se(w, c->beta_div2);
  }
}

/* codes one picture (all slices) into one MP4 sample */
static int encode_picture(Enc* e, const PicPlan* pp, int nrefs_l0, Sample* out) {
  const Cfg* c = e->cfg;
  H264Oracle* d = e->d;
  const int mbw = (c->width + 15) / 16, mbh = (c->height + 15) / 16;
  uint8_t* sample = NULL;
  size_t ssize = 0, scap = 0;
  const int nal_type = pp->type == 2 ? 5 : 1;
  const int ref_idc = pp->is_ref ? (pp->type == 2 ? 3 : 2) : 0;
  const int nmb_pic = mbw * mbh;
  const int nslices = c->slice_mbs > 0 ? (nmb_pic + c->slice_mbs - 1) / c->slice_mbs : imin(c->slices, mbh);
  Marking mk;
  plan_marking(e, pp, &mk); /* from the DPB as it is before this picture; the same in every slice */

  /* Rate control (--rate-kbit): a bit budget per picture by type, spent evenly over the macroblock rows; after
   * every row the QP moves one step towards the budget (within +-8 of --qp), so mb_qp_delta runs in long
   * monotonic stretches and neighbouring macroblocks (and deblocked edges) see different QPs all the time. */
  if (pp->type == 2 || e->rc_qp == 0) e->rc_qp = c->qp;
  const double rc_budget = c->rate_kbit * 1000.0 * (pp->type == 2 ? 4.0 : pp->type == 0 ? 1.0 : 0.5);
  size_t rc_bits_before = 0;
  if (c->extra_nals) { /* AUD; at an IDR also the parameter sets in band and a user-data SEI */
    const int ls = c->nal_length_size;
    scap = 512 + (size_t)e->sps_len + e->pps_len + e->pps1_len;
    sample = (uint8_t*)malloc(scap);
    const uint8_t aud[2] = {0x09, (uint8_t)((pp->type == 2 ? 0 : pp->type == 0 ? 1 : 2) << 5 | 0x10)}; /* (an upper bound with --mixed-slices) */
    const uint8_t sei[22] = {0x06, 0x05, 0x12, 'h', 'w', 'a', 'n', 'g', '-', 'c', 'o', 'r', 'p', 'u', 's', '-', 's', 'e', 'i', 'x', 'y', 0x80};
    const uint8_t* nal[5] = {aud, e->sps_nal, e->pps_nal, sei, e->pps1_nal};
    const size_t len[5] = {sizeof(aud), (size_t)e->sps_len, (size_t)e->pps_len, sizeof(sei), (size_t)e->pps1_len};
    for (int q = 0; q < (pp->type == 2 ? (c->multi_pps ? 5 : 4) : 1); ++q) {
      for (int k = 0; k < ls; ++k) sample[ssize + k] = (uint8_t)(len[q] >> (8 * (ls - 1 - k)));
      memcpy(sample + ssize + ls, nal[q], len[q]);
      ssize += ls + len[q];
    }
  }
  for (int s = 0; s < nslices; ++s) {
    const int row0 = s * mbh / nslices, row1 = (s + 1) * mbh / nslices;
    const int mb0 = c->slice_mbs > 0 ? s * c->slice_mbs : row0 * mbw;
    const int mb1 = c->slice_mbs > 0 ? imin((s + 1) * c->slice_mbs, nmb_pic) : row1 * mbw;
    BitWriter* w = &e->w;
    w->pos = 0;
    /* --mixed-slices: every second slice of a P picture is an I slice; in a B picture the slices cycle B, P, I */
    int st = pp->type;
    if (c->mixed_slices && pp->type == 0 && (s & 1)) st = 2;
    if (c->mixed_slices && pp->type == 1) st = s % 3 == 0 ? 1 : s % 3 == 1 ? 0 : 2;
    write_slice_header(e, w, pp, st, mb0, nrefs_l0, &mk);
    /* let the oracle parse the header it will later be given: builds the
     * reference lists, POC, starts the picture */
    {
      const size_t hb = w->pos;
      Bits b = {w->buf, (int)hb, 0, 0};
      if (h264i_parse_slice_header(d, &b, ref_idc, nal_type) != 0 || b.pos != (int)hb || b.overrun) {
        fprintf(stderr, "encoder: slice header self-parse failed: %s (pos %d vs %zu)\n", d->error, b.pos, hb);
        return -1;
      }
    }
    h264i_enc_begin_slice(d, w);
    for (int addr = mb0; addr < mb1; ++addr) {
      MbPlan plan;
      memset(&plan, 0, sizeof(plan));
      h264i_begin_mb(d, addr);
      int qp = c->rate_kbit > 0 ? e->rc_qp : c->qp;
      uint32_t hq = hash3((uint32_t)c->seed + 3, (uint32_t)e->frame_abs, (uint32_t)addr);
      if (hq % 9 == 0) qp = iclip(10, 48, qp + (int)((hq >> 4) % 5) - 2);
      if (c->qp_swing && hq % 7 == 3) qp = (int)((hq >> 8) % 52);
      if (st == 2 && c->global_mv == 11) decide_known_intra(e, &plan);
      else if (st == 2) decide_intra(e, &plan, qp, 2);
      else if (c->cover && hq % 5 == 1) decide_intra(e, &plan, qp, st);
      else if (st == 0) decide_p(e, &plan, qp, 1);
      else decide_b(e, &plan, qp);
      if (h264i_enc_mb(d, w, addr, &plan, addr == mb1 - 1) < 0) {
        fprintf(stderr, "encoder: %s\n", d->error);
        return -1;
      }
      if (c->rate_kbit > 0 && (addr + 1) % mbw == 0) {
        const double spent = (double)(rc_bits_before + w->pos), due = rc_budget * ((addr + 1) / mbw) / mbh;
        if (spent > due * 1.05) e->rc_qp = imin(imin(e->rc_qp + 1, c->qp + 8), 48);
        else if (spent < due * 0.95) e->rc_qp = imax(imax(e->rc_qp - 1, c->qp - 8), 10);
      }
    }
    h264i_enc_end_slice(d, w);
    rc_bits_before += w->pos;
    const size_t rb = w->pos / 8;
    if (ssize + rb * 3 / 2 + 16 > scap) {
      scap = (ssize + rb * 3 / 2 + 16) * 2;
      sample = (uint8_t*)realloc(sample, scap);
    }
    const int ls = c->nal_length_size;
    size_t n = nal_escape(w->buf, rb, (ref_idc << 5) | nal_type, sample + ssize + ls);
    if (ls < 4 && n >= ((size_t)1 << (8 * ls))) {
      fprintf(stderr, "encoder: a %zu-byte NAL unit does not fit --nal-length-size %d\n", n, ls);
      return -1;
    }
    for (int k = 0; k < ls; ++k) sample[ssize + k] = (uint8_t)(n >> (8 * (ls - 1 - k)));
    ssize += ls + n;
    if (c->extra_nals && s == nslices - 1) { /* filler data (nal_unit_type 12) behind the last slice */
      const uint8_t filler[6] = {0x0c, 0xff, 0xff, 0xff, 0xff, 0x80};
      for (int k = 0; k < ls; ++k) sample[ssize + k] = (uint8_t)(sizeof(filler) >> (8 * (ls - 1 - k)));
      memcpy(sample + ssize + ls, filler, sizeof(filler));
      ssize += ls + sizeof(filler);
    }
  }
  if (h264i_finish_picture(d) < 0) {
    fprintf(stderr, "encoder: finish_picture: %s\n", d->error);
    return -1;
  }
  out->data = sample;
  out->size = ssize;
  out->display = pp->display;
  out->key = pp->type == 2;
  return 0;
}

/* GOP structure in coding order */
static int plan_gop(const Cfg* c, int n, PicPlan* out) {
  const int fn_mask = (1 << c->fn_bits) - 1;
  int k = 0, fn = 0, ordinal = 0, poc_base = 0;
  out[k++] = (PicPlan){2, 0, 0, 1, 0, 0, 0};
  int prev_anchor = 0;
  while (prev_anchor < n - 1) {
    int anchor = imin(prev_anchor + c->bframes + 1, n - 1);
    /* closed GOP: the tail is coded as P pictures */
    fn = (fn + 1) & fn_mask;
    ++ordinal;
    /* MMCO 5 on every 5th reference picture (only without B pictures: it acts like an IDR for output order) */
    const int mmco5 = c->mmco == 2 && c->bframes == 0 && ordinal % 5 == 4;
    out[k++] = (PicPlan){0, anchor, fn, 1, ordinal, mmco5, poc_base};
    int mid = -1;
    if (c->bpyramid && anchor - prev_anchor >= 3) { /* the middle B picture is coded right after the anchor, as a reference */
      mid = (prev_anchor + anchor) / 2;
      fn = (fn + 1) & fn_mask;
      ++ordinal;
      out[k++] = (PicPlan){1, mid, fn, 1, ordinal, 0, poc_base};
    }
    for (int b = prev_anchor + 1; b < anchor; ++b)
      if (b != mid) out[k++] = (PicPlan){1, b, (fn + 1) & fn_mask, 0, ordinal, 0, poc_base};
    if (mmco5) { /* the picture counts as frame_num 0 / POC 0 from here on (8.2.1) */
      fn = 0;
      poc_base = anchor;
    }
    prev_anchor = anchor;
  }
  return k;
}

typedef struct {
  Sample* samples;
  int count;
  uint8_t* recon; /* display order, cropped I420, when requested */
} GopOut;

static int encode_gop(Enc* e, int gop_index, int nframes, GopOut* out, int want_recon) {
  const Cfg* c = e->cfg;
  H264Oracle* d = e->d;
  PicPlan plans[1024];
  if (nframes > 1000) return -1;
  const int n = plan_gop(c, nframes, plans);
  out->samples = (Sample*)calloc(n, sizeof(Sample));
  out->count = n;
  const int W = e->scene.W;
  e->src[0] = e->scene.plane[0];
  e->src[1] = e->scene.plane[1];
  e->src[2] = e->scene.plane[2];
  e->sstride[0] = W;
  e->sstride[1] = e->sstride[2] = W / 2;
  const size_t fsz = (size_t)c->width * c->height * 3 / 2;
  if (want_recon) out->recon = (uint8_t*)malloc(fsz * nframes);
  int k = 0;
  for (int i = 0; i <= n; ++i) {
    if (i < n) {
      e->frame_abs = gop_index * c->gop + plans[i].display;
      e->coded_index = i;
      scene_render(&e->scene, e->frame_abs);
      /* what the DPB holds (marking operations make this more than a count of coded pictures) */
      const int refs_available = plans[i].type == 2 ? 0 : dpb_refs(d);
      int nl0 = imax(1, imin(c->temporal_direct ? 1 : c->refs, refs_available));
      if (encode_picture(e, &plans[i], nl0, &out->samples[i]) < 0) return -1;
    } else {
      h264o_flush(d);
    }
    while (h264o_frames_ready(d)) { /* reconstruction in display order = golden YUV */
      if (want_recon && k < nframes) h264o_pop_frame(d, out->recon + fsz * k, fsz);
      else h264o_pop_frame(d, NULL, 0);
      ++k;
    }
  }
  if (k != nframes) {
    fprintf(stderr, "encoder: GOP produced %d of %d frames\n", k, nframes);
    return -1;
  }
  return 0;
}

/* ------------------------------------------------------------------------- */
/* MP4 writer                                                                */
/* ------------------------------------------------------------------------- */
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
static void b32(Buf* b, uint32_t v) {
  uint8_t t[4] = {(uint8_t)(v >> 24), (uint8_t)(v >> 16), (uint8_t)(v >> 8), (uint8_t)v};
  bput(b, t, 4);
}
static void b16(Buf* b, uint32_t v) {
  uint8_t t[2] = {(uint8_t)(v >> 8), (uint8_t)v};
  bput(b, t, 2);
}
static void b64(Buf* b, uint64_t v) {
  b32(b, (uint32_t)(v >> 32));
  b32(b, (uint32_t)v);
}
static size_t box_begin(Buf* b, const char* type) {
  size_t at = b->n;
  b32(b, 0);
  bput(b, type, 4);
  return at;
}
static void box_end(Buf* b, size_t at) {
  uint32_t n = (uint32_t)(b->n - at);
  b->b[at] = (uint8_t)(n >> 24);
  b->b[at + 1] = (uint8_t)(n >> 16);
  b->b[at + 2] = (uint8_t)(n >> 8);
  b->b[at + 3] = (uint8_t)n;
}
static void bzero_(Buf* b, size_t n) {
  static const uint8_t z[64] = {0};
  while (n) {
    size_t k = n > 64 ? 64 : n;
    bput(b, z, k);
    n -= k;
  }
}

/* HEVCDecoderConfigurationRecord (ISO/IEC 14496-15 8.3.3.1) for --hevc-entry: Main profile, level 4, one VPS / SPS / PPS.
 * The SPS follows ITU-T H.265 7.3.2.2 as far as a parser of the picture geometry reads (profile_tier_level, ids,
 * chroma format, size, conformance window, bit depths, POC bits, sub-layer ordering, coding / transform block sizes);
 * VPS and PPS are stubs.  Written from the syntax tables - no HEVC encoder or decoder has seen it. */
static size_t hevc_nal(const uint8_t* rbsp, size_t n, int type, uint8_t* out) {
  size_t o = 0;
  out[o++] = (uint8_t)(type << 1);  /* forbidden_zero_bit, nal_unit_type, top bit of nuh_layer_id */
  out[o++] = 1;                     /* nuh_layer_id 0, nuh_temporal_id_plus1 1 */
  int zeros = 0;
  for (size_t i = 0; i < n; ++i) {
    if (zeros >= 2 && rbsp[i] <= 3) {
      out[o++] = 3;
      zeros = 0;
    }
    out[o++] = rbsp[i];
    zeros = rbsp[i] == 0 ? zeros + 1 : 0;
  }
  return o;
}
static size_t build_hvcc(const Cfg* c, uint8_t* out) {
  const int cw = (c->width + 7) & ~7, ch = (c->height + 7) & ~7;  /* multiples of the minimum coding block (8) */
  BitWriter w = {0};
  bw_put(&w, 0, 4);  /* sps_video_parameter_set_id */
  bw_put(&w, 0, 3);  /* sps_max_sub_layers_minus1 */
  bw_put(&w, 1, 1);  /* sps_temporal_id_nesting_flag */
  bw_put(&w, 0, 2); bw_put(&w, 0, 1); bw_put(&w, 1, 5);        /* general profile space 0, Main tier, profile 1 (Main) */
  bw_put(&w, 0x60000000u, 32);                                 /* compatibility: Main, Main 10 */
  bw_put(&w, 1, 1); bw_put(&w, 0, 1); bw_put(&w, 0, 1); bw_put(&w, 1, 1);  /* progressive, !interlaced, !non-packed, frame-only */
  bw_put(&w, 0, 32); bw_put(&w, 0, 11);                        /* 43 reserved zero bits */
  bw_put(&w, 0, 1);                                            /* general_inbld_flag / reserved */
  bw_put(&w, 120, 8);                                          /* general_level_idc: 4.0 */
  bw_ue(&w, 0);      /* sps_seq_parameter_set_id */
  bw_ue(&w, 1);      /* chroma_format_idc 4:2:0 */
  bw_ue(&w, (unsigned)cw);
  bw_ue(&w, (unsigned)ch);
  if (cw != c->width || ch != c->height) {
    bw_put(&w, 1, 1);  /* conformance window, in chroma sample units */
    bw_ue(&w, 0); bw_ue(&w, (unsigned)(cw - c->width) / 2); bw_ue(&w, 0); bw_ue(&w, (unsigned)(ch - c->height) / 2);
  } else {
    bw_put(&w, 0, 1);
  }
  bw_ue(&w, 0); bw_ue(&w, 0);  /* bit depths 8 */
  bw_ue(&w, 4);                /* log2_max_pic_order_cnt_lsb_minus4 */
  bw_put(&w, 1, 1);            /* sps_sub_layer_ordering_info_present_flag */
  bw_ue(&w, 4); bw_ue(&w, 2); bw_ue(&w, 0);  /* max_dec_pic_buffering_minus1, num_reorder, max_latency_increase_plus1 */
  bw_ue(&w, 0);                /* log2_min_luma_coding_block_size_minus3: 8 */
  bw_ue(&w, 3);                /* log2_diff_max_min: CTB 64 */
  bw_ue(&w, 0);                /* log2_min_luma_transform_block_size_minus2: 4 */
  bw_ue(&w, 3);                /* log2_diff_max_min transform: 32 */
  bw_ue(&w, 1); bw_ue(&w, 1);  /* max_transform_hierarchy_depth inter / intra */
  bw_put(&w, 0, 1); bw_put(&w, 0, 1); bw_put(&w, 1, 1); bw_put(&w, 0, 1);  /* scaling lists, amp, sao, pcm */
  bw_ue(&w, 0);                /* num_short_term_ref_pic_sets */
  bw_put(&w, 0, 1); bw_put(&w, 1, 1); bw_put(&w, 0, 1); bw_put(&w, 0, 1); bw_put(&w, 0, 1);  /* long-term, tmvp, strong intra, vui, ext */
  bw_put(&w, 1, 1);
  while (w.pos & 7) bw_put(&w, 0, 1);
  uint8_t sps[256], vps[16], pps[16];
  const size_t sps_len = hevc_nal(w.buf, w.pos / 8, 33, sps);
  free(w.buf);
  static const uint8_t vps_rbsp[] = {0x0c, 0x01, 0xff, 0xff, 0x01, 0x60, 0x00, 0x00, 0x00, 0x90};
  static const uint8_t pps_rbsp[] = {0xc1, 0x72, 0xb4, 0x62, 0x40};
  const size_t vps_len = hevc_nal(vps_rbsp, sizeof(vps_rbsp), 32, vps);
  const size_t pps_len = hevc_nal(pps_rbsp, sizeof(pps_rbsp), 34, pps);
  size_t o = 0;
  out[o++] = 1;                 /* configurationVersion */
  out[o++] = 1;                 /* profile space 0, Main tier, profile_idc 1 */
  out[o++] = 0x60; out[o++] = 0; out[o++] = 0; out[o++] = 0;  /* general_profile_compatibility_flags */
  out[o++] = 0x90; out[o++] = 0; out[o++] = 0; out[o++] = 0; out[o++] = 0; out[o++] = 0;  /* constraint indicator flags */
  out[o++] = 120;               /* general_level_idc */
  out[o++] = 0xf0; out[o++] = 0;  /* reserved 1111, min_spatial_segmentation_idc 0 */
  out[o++] = 0xfc;              /* reserved 111111, parallelismType 0 */
  out[o++] = 0xfc | 1;          /* reserved, chromaFormat 1 */
  out[o++] = 0xf8;              /* reserved, bitDepthLumaMinus8 0 */
  out[o++] = 0xf8;              /* reserved, bitDepthChromaMinus8 0 */
  out[o++] = 0; out[o++] = 0;   /* avgFrameRate */
  out[o++] = (uint8_t)(0 << 6 | 1 << 3 | 1 << 2 | (c->nal_length_size - 1));  /* constantFrameRate, numTemporalLayers 1, nested, length size */
  out[o++] = 3;                 /* numOfArrays */
  const uint8_t* nals[3] = {vps, sps, pps};
  const size_t lens[3] = {vps_len, sps_len, pps_len};
  for (int a = 0; a < 3; ++a) {
    out[o++] = (uint8_t)(0x80 | (32 + a));  /* array_completeness, NAL_unit_type */
    out[o++] = 0; out[o++] = 1;             /* numNalus */
    out[o++] = (uint8_t)(lens[a] >> 8); out[o++] = (uint8_t)lens[a];
    memcpy(out + o, nals[a], lens[a]);
    o += lens[a];
  }
  return o;
}


/* samples in chunk k of the video track (--chunk-samples; 0 = everything in one chunk) */
static size_t chunk_len(const Cfg* c, size_t k, size_t first, size_t n) {
  if (c->chunk_samples <= 0) return n;
  const size_t want = (size_t)c->chunk_samples + (k % 3 == 2 ? 1 : 0);
  return first + want > n ? n - first : want;
}

/* sizes[], keys[], cts_off[] per sample in decode order */
static void build_moov(Buf* m, const Cfg* c, const uint8_t* avcc, size_t avcc_len, const uint32_t* sizes,
                       const uint8_t* keys, const int32_t* cts, size_t n, uint64_t mdat_payload_offset, int has_b) {
  const uint32_t ts = (uint32_t)c->fps * 1000, delta = 1000;
  size_t moov = box_begin(m, "moov");
  {
    size_t a = box_begin(m, "mvhd");
    b32(m, 0); b32(m, 0); b32(m, 0); b32(m, ts); b32(m, (uint32_t)(n * delta));
    b32(m, 0x00010000); b16(m, 0x0100); bzero_(m, 10);
    const uint32_t mat[9] = {0x10000, 0, 0, 0, 0x10000, 0, 0, 0, 0x40000000};
    for (int i = 0; i < 9; ++i) b32(m, mat[i]);
    bzero_(m, 24); b32(m, 2);
    box_end(m, a);
  }
  if (c->audio_track) { /* n four-byte "audio" samples in one chunk behind the video payload */
    uint64_t vbytes = 0;
    for (size_t i = 0; i < n; ++i) vbytes += sizes[i];
    size_t t = box_begin(m, "trak");
    size_t a = box_begin(m, "tkhd");
    b32(m, 3); b32(m, 0); b32(m, 0); b32(m, 2); b32(m, 0); b32(m, (uint32_t)(n * delta));
    bzero_(m, 8); b16(m, 0); b16(m, 1); b16(m, 0x0100); b16(m, 0);
    const uint32_t amat[9] = {0x10000, 0, 0, 0, 0x10000, 0, 0, 0, 0x40000000};
    for (int i = 0; i < 9; ++i) b32(m, amat[i]);
    b32(m, 0); b32(m, 0);
    box_end(m, a);
    size_t md = box_begin(m, "mdia");
    a = box_begin(m, "mdhd");
    b32(m, 0); b32(m, 0); b32(m, 0); b32(m, 48000); b32(m, (uint32_t)(n * 1024)); b16(m, 0x55c4); b16(m, 0);
    box_end(m, a);
    a = box_begin(m, "hdlr");
    b32(m, 0); b32(m, 0); bput(m, "soun", 4); bzero_(m, 12); bput(m, "SoundHandler", 13);
    box_end(m, a);
    size_t mi = box_begin(m, "minf");
    a = box_begin(m, "smhd");
    b32(m, 0); b32(m, 0);
    box_end(m, a);
    a = box_begin(m, "dinf");
    size_t d2 = box_begin(m, "dref");
    b32(m, 0); b32(m, 1);
    size_t u = box_begin(m, "url ");
    b32(m, 1);
    box_end(m, u);
    box_end(m, d2);
    box_end(m, a);
    size_t st = box_begin(m, "stbl");
    a = box_begin(m, "stsd");
    b32(m, 0); b32(m, 1);
    size_t en = box_begin(m, "mp4a");
    bzero_(m, 6); b16(m, 1); bzero_(m, 8); b16(m, 2); b16(m, 16); b32(m, 0); b32(m, 48000u << 16);
    box_end(m, en);
    box_end(m, a);
    a = box_begin(m, "stts");
    b32(m, 0); b32(m, 1); b32(m, (uint32_t)n); b32(m, 1024);
    box_end(m, a);
    if (c->chunk_samples > 0) { /* one sound chunk behind every video chunk, as many samples as that one has */
      size_t nch = 0;
      for (size_t f = 0, k = 0; f < n; f += chunk_len(c, k, f, n), ++k) ++nch;
      a = box_begin(m, "stsc");
      size_t cnt_at = m->n + 4;
      b32(m, 0); b32(m, 0);
      uint32_t runs = 0;
      size_t prev = 0;
      for (size_t f = 0, k = 0; f < n; ++k) {
        const size_t len = chunk_len(c, k, f, n);
        if (len != prev) { b32(m, (uint32_t)k + 1); b32(m, (uint32_t)len); b32(m, 1); ++runs; prev = len; }
        f += len;
      }
      m->b[cnt_at] = (uint8_t)(runs >> 24); m->b[cnt_at + 1] = (uint8_t)(runs >> 16); m->b[cnt_at + 2] = (uint8_t)(runs >> 8); m->b[cnt_at + 3] = (uint8_t)runs;
      box_end(m, a);
      a = box_begin(m, "stsz");
      b32(m, 0); b32(m, 4); b32(m, (uint32_t)n);
      box_end(m, a);
      a = box_begin(m, "stco");
      b32(m, 0); b32(m, (uint32_t)nch);
      uint64_t at = mdat_payload_offset;
      for (size_t f = 0, k = 0; f < n; ++k) {
        const size_t len = chunk_len(c, k, f, n);
        for (size_t i = 0; i < len; ++i) at += sizes[f + i];
        b32(m, (uint32_t)at);
        at += 4 * len;
        f += len;
      }
      box_end(m, a);
    } else {
      a = box_begin(m, "stsc");
      b32(m, 0); b32(m, 1); b32(m, 1); b32(m, (uint32_t)n); b32(m, 1);
      box_end(m, a);
      a = box_begin(m, "stsz");
      b32(m, 0); b32(m, 4); b32(m, (uint32_t)n);
      box_end(m, a);
      a = box_begin(m, "stco");
      b32(m, 0); b32(m, 1); b32(m, (uint32_t)(mdat_payload_offset + vbytes));
      box_end(m, a);
    }
    box_end(m, st);
    box_end(m, mi);
    box_end(m, md);
    box_end(m, t);
  }
  size_t trak = box_begin(m, "trak");
  {
    size_t a = box_begin(m, "tkhd");
    b32(m, 3); b32(m, 0); b32(m, 0); b32(m, 1); b32(m, 0); b32(m, (uint32_t)(n * delta));
    bzero_(m, 8); b16(m, 0); b16(m, 0); b16(m, 0); b16(m, 0);
    const uint32_t mat[9] = {0x10000, 0, 0, 0, 0x10000, 0, 0, 0, 0x40000000};
    for (int i = 0; i < 9; ++i) b32(m, mat[i]);
    b32(m, (uint32_t)c->width << 16); b32(m, (uint32_t)c->height << 16);
    box_end(m, a);
  }
  size_t mdia = box_begin(m, "mdia");
  {
    size_t a = box_begin(m, "mdhd");
    b32(m, 0); b32(m, 0); b32(m, 0); b32(m, ts); b32(m, (uint32_t)(n * delta)); b16(m, 0x55c4); b16(m, 0);
    box_end(m, a);
    a = box_begin(m, "hdlr");
    b32(m, 0); b32(m, 0); bput(m, "vide", 4); bzero_(m, 12); bput(m, "VideoHandler", 13);
    box_end(m, a);
  }
  size_t minf = box_begin(m, "minf");
  {
    size_t a = box_begin(m, "vmhd");
    b32(m, 1); bzero_(m, 8);
    box_end(m, a);
    a = box_begin(m, "dinf");
    size_t d2 = box_begin(m, "dref");
    b32(m, 0); b32(m, 1);
    size_t u = box_begin(m, "url ");
    b32(m, 1);
    box_end(m, u);
    box_end(m, d2);
    box_end(m, a);
  }
  size_t stbl = box_begin(m, "stbl");
  {
    size_t a = box_begin(m, "stsd");
    b32(m, 0); b32(m, 1);
    size_t v = box_begin(m, c->hevc_entry == 1 ? "hev1" : c->hevc_entry == 2 ? "hvc1" : "avc1");
    bzero_(m, 6); b16(m, 1); bzero_(m, 16);
    b16(m, (uint32_t)c->width); b16(m, (uint32_t)c->height);
    b32(m, 0x00480000); b32(m, 0x00480000); b32(m, 0); b16(m, 1); bzero_(m, 32); b16(m, 0x18); b16(m, 0xffff);
    size_t cc = box_begin(m, c->hevc_entry ? "hvcC" : "avcC");
    if (c->hevc_entry) {
      uint8_t hvcc[512];
      bput(m, hvcc, build_hvcc(c, hvcc));
    } else {
      bput(m, avcc, avcc_len);
    }
    box_end(m, cc);
    box_end(m, v);
    box_end(m, a);

    if (c->fragmented) n = 0; /* the sample tables of a fragmented file are empty: everything comes from the moofs */
    a = box_begin(m, "stts");
    b32(m, 0); b32(m, n ? 1 : 0);
    if (n) { b32(m, (uint32_t)n); b32(m, delta); }
    box_end(m, a);
    if (has_b && n) {
      a = box_begin(m, "ctts");
      b32(m, 0); b32(m, (uint32_t)n);
      for (size_t i = 0; i < n; ++i) { b32(m, 1); b32(m, (uint32_t)(cts[i] * (int32_t)delta)); }
      box_end(m, a);
    }
    uint32_t nk = 0;
    for (size_t i = 0; i < n; ++i) nk += keys[i];
    if (!(c->no_stss && nk == n)) { /* --no-stss: an all-intra track may leave the box out (every sample is a sync sample) */
      a = box_begin(m, "stss");
      b32(m, 0); b32(m, nk);
      for (size_t i = 0; i < n; ++i) if (keys[i]) b32(m, (uint32_t)i + 1);
      box_end(m, a);
    }
    a = box_begin(m, "stsc");
    if (c->chunk_samples > 0 && n) {
      size_t cnt_at = m->n + 4;
      b32(m, 0); b32(m, 0);
      uint32_t runs = 0;
      size_t prev = 0;
      for (size_t f = 0, k = 0; f < n; ++k) {
        const size_t len = chunk_len(c, k, f, n);
        if (len != prev) { b32(m, (uint32_t)k + 1); b32(m, (uint32_t)len); b32(m, 1); ++runs; prev = len; }
        f += len;
      }
      m->b[cnt_at] = (uint8_t)(runs >> 24); m->b[cnt_at + 1] = (uint8_t)(runs >> 16); m->b[cnt_at + 2] = (uint8_t)(runs >> 8); m->b[cnt_at + 3] = (uint8_t)runs;
    } else {
      b32(m, 0); b32(m, n ? 1 : 0);
      if (n) { b32(m, 1); b32(m, (uint32_t)n); b32(m, 1); }
    }
    box_end(m, a);
    int fits16 = c->stz2;
    for (size_t i = 0; i < n; ++i) fits16 &= sizes[i] < 65536;
    if (fits16 && n) {
      a = box_begin(m, "stz2");
      b32(m, 0); b32(m, 16); b32(m, (uint32_t)n); /* reserved(24) + field_size(8) */
      for (size_t i = 0; i < n; ++i) b16(m, sizes[i]);
    } else {
      a = box_begin(m, "stsz");
      b32(m, 0); b32(m, 0); b32(m, (uint32_t)n);
      for (size_t i = 0; i < n; ++i) b32(m, sizes[i]);
    }
    box_end(m, a);
    {
      const int wide = mdat_payload_offset > 0xfffffff0ULL || c->co64;
      size_t nch = n ? 1 : 0;
      if (c->chunk_samples > 0 && n) {
        nch = 0;
        for (size_t f = 0, k = 0; f < n; f += chunk_len(c, k, f, n), ++k) ++nch;
      }
      a = box_begin(m, wide ? "co64" : "stco");
      b32(m, 0); b32(m, (uint32_t)nch);
      uint64_t at = mdat_payload_offset;
      for (size_t f = 0, k = 0; f < n && k < nch; ++k) {
        const size_t len = chunk_len(c, k, f, n);
        if (wide) b64(m, at);
        else b32(m, (uint32_t)at);
        for (size_t i = 0; i < len; ++i) at += sizes[f + i];
        if (c->audio_track && c->chunk_samples > 0) at += 4 * len; /* the sound chunk in between */
        f += len;
      }
      box_end(m, a);
    }
  }
  box_end(m, stbl);
  box_end(m, minf);
  box_end(m, mdia);
  box_end(m, trak);
  if (c->fragmented) {
    size_t a = box_begin(m, "mvex");
    size_t t = box_begin(m, "trex");
    /* track 1, description 1, duration, default size 0, default flags: not a sync sample */
    b32(m, 0); b32(m, 1); b32(m, 1); b32(m, 1000); b32(m, 0); b32(m, 0x00010000);
    box_end(m, t);
    box_end(m, a);
  }
  box_end(m, moov);
}

/* One movie fragment: moof for the samples [first, first + count) followed by
 * the mdat header.  `variant` picks the tfhd / trun flag combination; `at` is
 * the file offset the moof will have.  Returns the bytes (moof + 8 of mdat). */
static void build_moof(Buf* m, int variant, uint32_t seq, uint64_t at, const uint32_t* sizes, const uint8_t* keys,
                       const int32_t* cts, size_t count) {
  const size_t moof = box_begin(m, "moof");
  size_t a = box_begin(m, "mfhd");
  b32(m, 0); b32(m, seq);
  box_end(m, a);
  const size_t traf = box_begin(m, "traf");
  /* two passes: the data offsets depend on the moof's own size */
  const size_t tfhd_at = m->n;
  for (int pass = 0; pass < 2; ++pass) {
    const uint64_t moof_size = pass ? m->n - moof : 0; /* pass 0 writes placeholders, sizes do not change */
    m->n = tfhd_at;
    const uint64_t payload = at + moof_size + 8; /* first byte after the mdat header */
    a = box_begin(m, "tfhd");
    if (variant == 1) { /* default-base-is-moof; flags and sizes from trun / trex */
      b32(m, 0x020000); b32(m, 1);
    } else if (variant == 2) { /* explicit base data offset + default sample flags (non-sync) */
      b32(m, 0x000001 | 0x000020); b32(m, 1); b64(m, payload); b32(m, 0x00010000);
    } else { /* nothing: base = this moof (first traf), trex defaults */
      b32(m, 0); b32(m, 1);
    }
    box_end(m, a);
    const size_t nruns = variant == 3 && count > 1 ? 2 : 1;
    size_t done = 0;
    for (size_t r = 0; r < nruns; ++r) {
      const size_t cnt = nruns == 1 ? count : (r == 0 ? count / 2 : count - count / 2);
      a = box_begin(m, "trun");
      if (variant == 1) {
        b32(m, 0x000001 | 0x000004 | 0x000200); b32(m, (uint32_t)cnt);
        b32(m, (uint32_t)(moof_size + 8));       /* data offset, relative to the moof */
        b32(m, keys[0] ? 0x02000000 : 0x00010000); /* first_sample_flags */
        for (size_t i = 0; i < cnt; ++i) b32(m, sizes[i]);
      } else if (variant == 2) {
        b32(m, 0x000100 | 0x000200 | 0x000400 | 0x000800); b32(m, (uint32_t)cnt); /* no data offset: continues at the base */
        for (size_t i = 0; i < cnt; ++i) {
          b32(m, 1000); b32(m, sizes[i]); b32(m, keys[i] ? 0x02000000 : 0x00010000); b32(m, (uint32_t)(cts[i] * 1000));
        }
      } else {
        /* first run: data offset + per-sample flags; second run: neither (follows the first) */
        uint64_t off = moof_size + 8;
        if (r == 0) { b32(m, 0x000001 | 0x000200 | 0x000400); b32(m, (uint32_t)cnt); b32(m, (uint32_t)off); }
        else { b32(m, 0x000200 | 0x000400); b32(m, (uint32_t)cnt); }
        for (size_t i = 0; i < cnt; ++i) { b32(m, sizes[done + i]); b32(m, keys[done + i] ? 0x02000000 : 0x00010000); }
      }
      box_end(m, a);
      done += cnt;
    }
    box_end(m, traf);
    box_end(m, moof);
  }
  uint64_t payload_bytes = 0;
  for (size_t i = 0; i < count; ++i) payload_bytes += sizes[i];
  b32(m, (uint32_t)(payload_bytes + 8));
  bput(m, "mdat", 4);
}

/* ------------------------------------------------------------------------- */
/* driver                                                                    */
/* ------------------------------------------------------------------------- */
typedef struct {
  const Cfg* cfg;
  int next_gop, ngops;
  pthread_mutex_t mu;
  GopOut* gops;
  int want_recon;
  int failed;
} Work;

static void enc_free(Enc* e) {
  scene_free(&e->scene);
  free(e->w.buf);
  free(e->sps_nal);
  free(e->pps_nal);
  free(e->pps1_nal);
  h264o_close(e->d);
}

static int enc_init(Enc* e, const Cfg* c) {
  memset(e, 0, sizeof(*e));
  e->cfg = c;
  scene_init(&e->scene, c);
  write_sps(e);
  write_pps(e);
  uint8_t avcc[512];
  size_t n = make_avcc(e, avcc);
  e->d = h264o_open();
  if (h264o_configure(e->d, avcc, n) != 0) {
    fprintf(stderr, "encoder: own parameter sets rejected: %s\n", h264o_error(e->d));
    return -1;
  }
  e->hint_n = 0;
  /* hints: motion vector per frame of temporal distance towards the past.
   * Background: frame t shows world(x + pan t), so the sample at x came from
   * x + pan in the previous frame; a rectangle moving by v came from x - v. */
  e->hints[e->hint_n][0] = e->scene.pan_x;
  e->hints[e->hint_n][1] = e->scene.pan_y;
  e->hint_n++;
  for (int i = 0; i < 8 && e->hint_n < 10; ++i) {
    e->hints[e->hint_n][0] = -e->scene.rects[i].vx;
    e->hints[e->hint_n][1] = -e->scene.rects[i].vy;
    e->hint_n++;
  }
  return 0;
}

static void* worker(void* arg) {
  Work* wk = (Work*)arg;
  Enc e;
  if (enc_init(&e, wk->cfg) < 0) {
    wk->failed = 1;
    return NULL;
  }
  for (;;) {
    pthread_mutex_lock(&wk->mu);
    int g = wk->next_gop++;
    pthread_mutex_unlock(&wk->mu);
    if (g >= wk->ngops || wk->failed) break;
    int nframes = imin(wk->cfg->gop, wk->cfg->frames - g * wk->cfg->gop);
    if (encode_gop(&e, g, nframes, &wk->gops[g], wk->want_recon) < 0) wk->failed = 1;
  }
  enc_free(&e);
  return NULL;
}

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

int main(int argc, char** argv) {
  Cfg c;
  memset(&c, 0, sizeof(c));
  c.width = arg_int(argc, argv, "--width", 320);
  c.height = arg_int(argc, argv, "--height", 240);
  c.frames = arg_int(argc, argv, "--frames", 30);
  c.gop = arg_int(argc, argv, "--gop", 30);
  c.bframes = arg_int(argc, argv, "--bframes", 2);
  c.qp = arg_int(argc, argv, "--qp", 26);
  c.cabac = arg_int(argc, argv, "--cabac", 1);
  c.t8x8 = arg_int(argc, argv, "--t8x8", 1);
  c.refs = arg_int(argc, argv, "--refs", 2);
  c.slices = arg_int(argc, argv, "--slices", 1);
  c.seed = arg_int(argc, argv, "--seed", 1);
  c.threads = arg_int(argc, argv, "--threads", 1);
  c.repeat = arg_int(argc, argv, "--repeat", 1);
  c.fps = arg_int(argc, argv, "--fps", 30);
  c.chroma_qp_offset = arg_int(argc, argv, "--chroma-qp-offset", 0);
  c.deblock_idc = arg_int(argc, argv, "--deblock", 0);
  c.alpha_div2 = arg_int(argc, argv, "--alpha", 0);
  c.beta_div2 = arg_int(argc, argv, "--beta", 0);
  c.implicit_wp = arg_int(argc, argv, "--implicit-wp", 0);
  c.temporal_direct = arg_int(argc, argv, "--temporal-direct", 0);
  c.intra_only_pcm_period = arg_int(argc, argv, "--pcm-period", 4099);
  c.explicit_wp = arg_int(argc, argv, "--explicit-wp", 0);
  c.wp_denom = arg_int(argc, argv, "--wp-denom", 5);
  c.rplm = arg_int(argc, argv, "--rplm", 0);
  c.mmco = arg_int(argc, argv, "--mmco", 0);
  c.scaling = arg_int(argc, argv, "--scaling", 0);
  c.fragmented = arg_int(argc, argv, "--fragmented", 0);
  c.co64 = arg_int(argc, argv, "--co64", 0);
  c.stz2 = arg_int(argc, argv, "--stz2", 0);
  c.rate_kbit = arg_int(argc, argv, "--rate-kbit", 0);
  c.moov_last = arg_int(argc, argv, "--moov-last", 0);
  c.free_boxes = arg_int(argc, argv, "--free-boxes", 0);
  c.large_mdat = arg_int(argc, argv, "--large-mdat", 0);
  c.audio_track = arg_int(argc, argv, "--audio-track", 0);
  c.random_intra = arg_int(argc, argv, "--random-intra", 0);
  c.cover = arg_int(argc, argv, "--cover", 0);
  c.no_stss = arg_int(argc, argv, "--no-stss", 0);
  c.hevc_entry = arg_int(argc, argv, "--hevc-entry", 0);
  c.chunk_samples = arg_int(argc, argv, "--chunk-samples", 0);
  c.wild_mv = arg_int(argc, argv, "--wild-mv", 0);
  c.global_mv = arg_int(argc, argv, "--global-mv", 0);
  c.qp_swing = arg_int(argc, argv, "--qp-swing", 0);
  c.slice_mbs = arg_int(argc, argv, "--slice-mbs", 0);
  c.mixed_slices = arg_int(argc, argv, "--mixed-slices", 0);
  c.crop_left = arg_int(argc, argv, "--crop-left", 0);
  c.crop_top = arg_int(argc, argv, "--crop-top", 0);
  c.multi_pps = arg_int(argc, argv, "--multi-pps", 0);
  c.nal_length_size = arg_int(argc, argv, "--nal-length-size", 4);
  c.extra_nals = arg_int(argc, argv, "--extra-nals", 0);
  if (c.nal_length_size != 1 && c.nal_length_size != 2 && c.nal_length_size != 4) {
    fprintf(stderr, "encoder: --nal-length-size 1, 2 or 4\n");
    return 2;
  }
  c.fn_bits = arg_int(argc, argv, "--fn-bits", 8);
  c.poc_bits = arg_int(argc, argv, "--poc-bits", 8);
  if (c.fn_bits < 4 || c.fn_bits > 12 || c.poc_bits < 4 || c.poc_bits > 12 || (1 << c.fn_bits) <= c.refs ||
      4 * (c.bframes + 1) >= (1 << c.poc_bits)) {
    fprintf(stderr, "encoder: --fn-bits / --poc-bits 4..12; MaxFrameNum > refs; 4 * (bframes + 1) < MaxPicOrderCntLsb\n");
    return 2;
  }
  c.bpyramid = arg_int(argc, argv, "--bpyramid", 0);
  c.slice_qp_delta = arg_int(argc, argv, "--slice-qp-delta", 0);
  if (c.bpyramid && (c.bframes < 2 || c.refs < 4 || c.mmco)) {
    fprintf(stderr, "encoder: --bpyramid needs --bframes >= 2 and --refs >= 4 (sliding window: two anchors + two middle pictures), and does not combine with --mmco\n");
    return 2;
  }
  c.poc_type = arg_int(argc, argv, "--poc-type", 0);
  c.constrained_intra = arg_int(argc, argv, "--constrained-intra", 0);
  c.direct8x8 = arg_int(argc, argv, "--direct8x8", 1);
  if (c.poc_type < 0 || c.poc_type > 2 || (c.poc_type == 2 && c.bframes) || (c.poc_type && c.mmco == 2)) {
    fprintf(stderr, "encoder: --poc-type 2 needs --bframes 0; types 1 and 2 do not combine with --mmco 2\n");
    return 2;
  }
  if (c.wp_denom < 0 || c.wp_denom > 7) c.wp_denom = 5;
  if (c.explicit_wp) c.implicit_wp = 0;
  c.out = arg_str(argc, argv, "--out", "out.mp4");
  c.yuv_out = arg_str(argc, argv, "--yuv-out", NULL);
  if (c.refs < 1) c.refs = 1;
  if (c.bframes && c.refs < 2) c.refs = 2;
  if (c.width & 1 || c.height & 1 || c.width < 16 || c.height < 16 || c.gop < 1 || c.frames < 1) {
    fprintf(stderr, "bad geometry\n");
    return 2;
  }
  init_bases();

  Work wk;
  memset(&wk, 0, sizeof(wk));
  wk.cfg = &c;
  wk.ngops = (c.frames + c.gop - 1) / c.gop;
  wk.gops = (GopOut*)calloc(wk.ngops, sizeof(GopOut));
  wk.want_recon = c.yuv_out != NULL;
  pthread_mutex_init(&wk.mu, NULL);
  int nt = imax(1, imin(c.threads, wk.ngops));
  pthread_t th[512];
  if (nt > 512) nt = 512;
  for (int i = 0; i < nt; ++i) pthread_create(&th[i], NULL, worker, &wk);
  for (int i = 0; i < nt; ++i) pthread_join(th[i], NULL);
  if (wk.failed) {
    fprintf(stderr, "encoding failed\n");
    return 1;
  }

  /* mux */
  Enc e0;
  memset(&e0, 0, sizeof(e0));
  e0.cfg = &c;
  write_sps(&e0);
  write_pps(&e0);
  uint8_t avcc[512];
  size_t avcc_len = make_avcc(&e0, avcc);

  size_t n1 = 0;
  for (int g = 0; g < wk.ngops; ++g) n1 += wk.gops[g].count;
  const size_t n = n1 * (size_t)c.repeat;
  uint32_t* sizes = (uint32_t*)malloc(n * 4);
  uint8_t* keys = (uint8_t*)malloc(n);
  int32_t* cts = (int32_t*)malloc(n * 4);
  uint64_t payload = 0;
  size_t k = 0;
  for (int rep = 0; rep < c.repeat; ++rep)
    for (int g = 0; g < wk.ngops; ++g) {
      const GopOut* go = &wk.gops[g];
      for (int i = 0; i < go->count; ++i, ++k) {
        sizes[k] = (uint32_t)go->samples[i].size;
        keys[k] = (uint8_t)go->samples[i].key;
        /* composition offset in frames; +1 keeps it non-negative with one
         * reorder frame */
        cts[k] = go->samples[i].display - i + (c.bframes ? 1 : 0);
        payload += go->samples[i].size;
      }
    }
  Buf head = {0};
  {
    size_t a = box_begin(&head, "ftyp");
    bput(&head, "isom", 4); b32(&head, 512); bput(&head, "isomiso2avc1mp41", 16);
    box_end(&head, a);
  }
  /* moov's size does not depend on the chunk offset's value, only on its
   * width: build once to learn the size, then for real */
  if (c.free_boxes) {
    size_t a = box_begin(&head, "free");
    bzero_(&head, 24);
    box_end(&head, a);
  }
  const uint64_t audio_bytes = c.audio_track ? (uint64_t)n * 4 : 0;
  const int really_big = payload + audio_bytes + head.n + (size_t)n * 16 + 4096 > 0xfffffff0ULL;
  const int big = c.large_mdat || really_big; /* a forced 64-bit mdat size does not make the chunk offsets 64-bit */
  Buf moov = {0};
  build_moov(&moov, &c, avcc, avcc_len, sizes, keys, cts, n, really_big ? 0x100000000ULL : 0, c.bframes > 0);
  const uint64_t off = head.n + (c.moov_last ? 0 : moov.n) + (big ? 16 : 8);
  moov.n = 0;
  build_moov(&moov, &c, avcc, avcc_len, sizes, keys, cts, n, off, c.bframes > 0);
  FILE* f = fopen(c.out, "wb");
  if (!f) {
    perror(c.out);
    return 1;
  }
  fwrite(head.b, 1, head.n, f);
  if (!c.moov_last || c.fragmented) fwrite(moov.b, 1, moov.n, f);
  if (c.fragmented) {
    /* one moof + mdat per GOP (DASH-style); sizes / keys / cts are indexed in decode order */
    uint64_t at = head.n + moov.n;
    size_t k0 = 0;
    uint32_t seq = 1;
    for (int rep = 0; rep < c.repeat; ++rep)
      for (int g = 0; g < wk.ngops; ++g) {
        const size_t cnt = (size_t)wk.gops[g].count;
        Buf frag = {0};
        build_moof(&frag, c.fragmented, seq++, at, sizes + k0, keys + k0, cts + k0, cnt);
        fwrite(frag.b, 1, frag.n, f);
        at += frag.n;
        for (size_t i = 0; i < cnt; ++i) {
          fwrite(wk.gops[g].samples[i].data, 1, wk.gops[g].samples[i].size, f);
          at += wk.gops[g].samples[i].size;
        }
        free(frag.b);
        k0 += cnt;
      }
    fclose(f);
    goto wrote;
  }
  if (big) {
    Buf h = {0};
    b32(&h, 1); bput(&h, "mdat", 4); b64(&h, payload + audio_bytes + 16);
    fwrite(h.b, 1, h.n, f);
  } else {
    Buf h = {0};
    b32(&h, (uint32_t)(payload + audio_bytes + 8)); bput(&h, "mdat", 4);
    fwrite(h.b, 1, h.n, f);
  }
  {
    size_t k = 0, in_chunk = 0, first = 0, s_index = 0;
    size_t len = chunk_len(&c, 0, 0, n);
    for (int rep = 0; rep < c.repeat; ++rep)
      for (int g = 0; g < wk.ngops; ++g)
        for (int i = 0; i < wk.gops[g].count; ++i, ++s_index) {
          fwrite(wk.gops[g].samples[i].data, 1, wk.gops[g].samples[i].size, f);
          if (++in_chunk == len && c.chunk_samples > 0) { /* chunk complete: its sound chunk follows */
            if (c.audio_track)
              for (size_t b = 0; b < 4 * len; ++b) fputc((int)(0xa0 + (b & 15)), f);
            first += len;
            len = chunk_len(&c, ++k, first, n);
            in_chunk = 0;
          }
        }
    if (c.audio_track && c.chunk_samples <= 0)
      for (uint64_t i = 0; i < audio_bytes; ++i) fputc((int)(0xa0 + (i & 15)), f);
  }
  if (c.moov_last) fwrite(moov.b, 1, moov.n, f);
  if (c.free_boxes) {
    Buf u = {0};
    size_t a = box_begin(&u, "uuid");
    bput(&u, "hwang-corpus-box", 16);
    b64(&u, 0x0123456789abcdefULL);
    box_end(&u, a);
    fwrite(u.b, 1, u.n, f);
    free(u.b);
  }
  fclose(f);
wrote:

  if (c.yuv_out) {
    FILE* y = fopen(c.yuv_out, "wb");
    const size_t fsz = (size_t)c.width * c.height * 3 / 2;
    for (int g = 0; g < wk.ngops; ++g) {
      int nframes = imin(c.gop, c.frames - g * c.gop);
      fwrite(wk.gops[g].recon, 1, fsz * nframes, y);
    }
    fclose(y);
  }
  fprintf(stderr, "wrote %s: %zu samples, %.1f kbit/frame, profile %d, %s\n", c.out, n,
          payload * 8.0 / 1000.0 / (double)n, profile_for(&c), c.cabac ? "CABAC" : "CAVLC");
  return 0;
}
