// See hevc_parser.h.  Clause numbers are ITU-T H.265 (04/2013).
#include "hevc_parser.h"

#include "hevc_tables.h"

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstring>

namespace hwang {
namespace hevc {
namespace {

using amd::hevc::EdgeInfo;
using amd::hevc::InterPu;
using amd::hevc::IntraTu;
using amd::hevc::ResTu;
using amd::hevc::SaoCtb;
using amd::hevc::SliceLoop;
using amd::hevc::WeightEntry;

inline int clip3(int lo, int hi, int v) { return v < lo ? lo : v > hi ? hi : v; }
inline int iabs(int v) { return v < 0 ? -v : v; }
inline int ceil_log2(unsigned v) {
  int n = 0;
  while ((1u << n) < v) ++n;
  return n;
}

// ------------------------------------------------------------------------------------------------ plain bit reader
class Bits {
 public:
  Bits(const uint8_t* p, size_t n) : p_(p), n_(n) {}
  unsigned u(int bits) {
    unsigned v = 0;
    for (int i = 0; i < bits; ++i) {
      unsigned b = 0;
      if ((pos_ >> 3) < n_)
        b = (p_[pos_ >> 3] >> (7 - (pos_ & 7))) & 1;
      else
        bad_ = true;
      ++pos_;
      v = v << 1 | b;
    }
    return v;
  }
  unsigned ue() {
    int z = 0;
    while (z < 32 && !u(1) && !bad_) ++z;
    if (z >= 32) {  // (no syntax element is that long: 9.2 allows 32 bits in all)
      bad_ = true;
      return 0;
    }
    return z ? ((1u << z) - 1 + u(z)) : 0;
  }
  int se() {
    const unsigned k = ue();
    return (k & 1) ? static_cast<int>((k + 1) >> 1) : -static_cast<int>(k >> 1);
  }
  bool bad() const { return bad_; }
  size_t pos() const { return pos_; }
  size_t size_bits() const { return n_ * 8; }
  const uint8_t* data() const { return p_; }

 private:
  const uint8_t* p_;
  size_t n_, pos_ = 0;
  bool bad_ = false;
};

// --------------------------------------------------------------------------------------------- parameter sets
struct Rps {
  int nneg = 0, npos = 0;
  int dneg[16], dpos[16];
  uint8_t uneg[16], upos[16];
};

struct Scaling {  // ScalingFactor, raster [y * n + x]
  uint8_t f4[6][16], f8[6][64], f16[6][256], f32[2][1024];
};
static_assert(sizeof(Scaling) == 4064, "the seam carries the table as bytes");

struct Sps {
  bool valid = false;
  int max_sub_layers = 1, width = 0, height = 0, crop_l = 0, crop_r = 0, crop_t = 0, crop_b = 0;
  int log2_max_poc_lsb = 4, max_dec_pic_buffering = 1, max_num_reorder = 0, max_latency_plus1 = 0;
  int log2_min_cb = 3, log2_ctb = 4, log2_min_tb = 2, log2_max_tb = 5, th_inter = 0, th_intra = 0;
  bool scaling_enabled = false, amp = false, sao = false, pcm = false, pcm_lf_off = false, lt_present = false, tmvp = false, strong = false;
  Scaling scaling;
  int pcm_bits_y = 8, pcm_bits_c = 8, log2_min_pcm = 3, log2_max_pcm = 3;
  int num_rps = 0;
  Rps rps[65];
  int num_lt = 0, lt_lsb[32];
  uint8_t lt_used[32];
  int wctb = 0, hctb = 0;
};

struct Pps {
  bool valid = false;
  int sps_id = 0;
  bool dependent_slices = false, output_flag_present = false, sign_hiding = false, cabac_init_present = false;
  int extra_bits = 0, nref_default[2] = {1, 1}, init_qp = 26;
  bool cip = false, tskip = false, qp_delta = false, slice_chroma_offsets = false, wp = false, wbp = false, bypass = false, wpp = false;
  int qp_delta_depth = 0, cb_off = 0, cr_off = 0;
  bool lf_across = false, db_override = false, db_off = false, scaling_present = false, lists_mod = false, sh_ext = false;
  int beta_off = 0, tc_off = 0, pml = 2;
  Scaling scaling;
};

enum { SL_B = 0, SL_P = 1, SL_I = 2 };

struct SliceHdr {
  bool first = false, dependent = false;
  int pps_id = 0, address = 0, type = SL_I, poc_lsb = 0;
  bool rps_from_sps = false;
  Rps rps;
  bool have_rps = false;
  int num_lt = 0, lt_lsb[32], lt_cycle[32];
  uint8_t lt_used[32], lt_msb[32];
  bool tmvp = false, sao_y = false, sao_c = false;
  int nref[2] = {0, 0};
  bool rplm[2] = {false, false};
  int entry[2][16];
  bool mvd_l1_zero = false, cabac_init = false, col_from_l0 = true;
  int col_ref = 0;
  int wd_y = 0, wd_c = 0;
  int16_t wy[2][16], oy[2][16], wc[2][16][2], oc[2][16][2];
  int max_merge = 5, qp = 26, cb_off = 0, cr_off = 0;
  bool db_off = false, across = false;
  int beta_off = 0, tc_off = 0;
  size_t data_offset = 0;
};

void scaling_put(Scaling* s, int size_id, int m, const uint8_t* list, int dc) {
  static const auto diag = [](int log2, uint8_t* out) {  // 6.5.3: x | y << 4
    const int n = 1 << log2;
    int i = 0, x = 0, y = 0;
    for (;;) {
      while (y >= 0) {
        if (x < n && y < n) out[i++] = static_cast<uint8_t>(x | y << 4);
        --y;
        ++x;
      }
      y = x;
      x = 0;
      if (i >= n * n) break;
    }
  };
  uint8_t sc[64];
  if (size_id == 0) {
    diag(2, sc);
    for (int i = 0; i < 16; ++i) s->f4[m][(sc[i] >> 4) * 4 + (sc[i] & 15)] = list[i];
    return;
  }
  diag(3, sc);
  uint8_t m8[64];
  for (int i = 0; i < 64; ++i) m8[(sc[i] >> 4) * 8 + (sc[i] & 15)] = list[i];
  if (size_id == 1) {
    std::memcpy(s->f8[m], m8, 64);
  } else if (size_id == 2) {
    for (int y = 0; y < 16; ++y)
      for (int x = 0; x < 16; ++x) s->f16[m][y * 16 + x] = m8[(y >> 1) * 8 + (x >> 1)];
    s->f16[m][0] = static_cast<uint8_t>(dc);
  } else {
    for (int y = 0; y < 32; ++y)
      for (int x = 0; x < 32; ++x) s->f32[m][y * 32 + x] = m8[(y >> 2) * 8 + (x >> 2)];
    s->f32[m][0] = static_cast<uint8_t>(dc);
  }
}
void scaling_default_list(int size_id, int m, uint8_t* list) {
  if (size_id == 0) {
    std::memset(list, 16, 16);
    return;
  }
  const bool inter = size_id == 3 ? m == 1 : m >= 3;
  std::memcpy(list, inter ? hevc_default_scaling_inter8 : hevc_default_scaling_intra8, 64);
}
void scaling_defaults(Scaling* s) {
  uint8_t list[64];
  for (int size_id = 0; size_id < 4; ++size_id)
    for (int m = 0; m < (size_id == 3 ? 2 : 6); ++m) {
      scaling_default_list(size_id, m, list);
      scaling_put(s, size_id, m, list, 16);
    }
}
bool scaling_parse(Bits& b, Scaling* s) {  // 7.3.4
  uint8_t lists[4][6][64];
  int dc[4][6];
  for (int size_id = 0; size_id < 4; ++size_id)
    for (int m = 0; m < (size_id == 3 ? 2 : 6); ++m) {
      const int n = std::min(64, 1 << (4 + (size_id << 1)));
      dc[size_id][m] = 16;
      if (!b.u(1)) {
        const int delta = static_cast<int>(b.ue());
        if (delta == 0) {
          scaling_default_list(size_id, m, lists[size_id][m]);
        } else {
          if (delta > m) return false;
          std::memcpy(lists[size_id][m], lists[size_id][m - delta], static_cast<size_t>(n));
          dc[size_id][m] = dc[size_id][m - delta];
        }
      } else {
        int next = 8;
        if (size_id > 1) {
          next = b.se() + 8;
          dc[size_id][m] = next;
        }
        for (int i = 0; i < n; ++i) {
          next = (next + b.se() + 256) % 256;
          lists[size_id][m][i] = static_cast<uint8_t>(next);
        }
      }
      scaling_put(s, size_id, m, lists[size_id][m], dc[size_id][m]);
    }
  return !b.bad();
}

// 7.3.7 with the derivation of 7.4.8 for predicted sets
bool parse_rps(Bits& b, Rps* out, int idx, int num_sets, const Rps* sets) {
  *out = Rps();
  bool predicted = false;
  if (idx != 0) predicted = b.u(1) != 0;
  if (!predicted) {
    out->nneg = static_cast<int>(b.ue());
    out->npos = static_cast<int>(b.ue());
    if (out->nneg > 16 || out->npos > 16) return false;
    int poc = 0;
    for (int i = 0; i < out->nneg; ++i) {
      poc -= static_cast<int>(b.ue()) + 1;
      out->dneg[i] = poc;
      out->uneg[i] = static_cast<uint8_t>(b.u(1));
    }
    poc = 0;
    for (int i = 0; i < out->npos; ++i) {
      poc += static_cast<int>(b.ue()) + 1;
      out->dpos[i] = poc;
      out->upos[i] = static_cast<uint8_t>(b.u(1));
    }
    return !b.bad();
  }
  int back = 1;
  if (idx == num_sets) back = static_cast<int>(b.ue()) + 1;
  if (back > idx) return false;
  const Rps& r = sets[idx - back];
  const int sign = static_cast<int>(b.u(1));
  const int drps = (1 - 2 * sign) * (static_cast<int>(b.ue()) + 1);
  const int nr = r.nneg + r.npos;
  uint8_t used[33], keep[33];
  for (int j = 0; j <= nr; ++j) {
    used[j] = static_cast<uint8_t>(b.u(1));
    keep[j] = 1;
    if (!used[j]) keep[j] = static_cast<uint8_t>(b.u(1));
  }
  // the reference set's pictures, shifted by drps, plus the reference picture itself (shift drps), sorted into the
  // two halves: negative ones closest first, positive ones closest first
  int i = 0;
  for (int j = r.npos - 1; j >= 0; --j) {
    const int d = r.dpos[j] + drps;
    if (d < 0 && keep[r.nneg + j]) {
      if (i >= 16) return false;
      out->dneg[i] = d;
      out->uneg[i++] = used[r.nneg + j];
    }
  }
  if (drps < 0 && keep[nr]) {
    if (i >= 16) return false;
    out->dneg[i] = drps;
    out->uneg[i++] = used[nr];
  }
  for (int j = 0; j < r.nneg; ++j) {
    const int d = r.dneg[j] + drps;
    if (d < 0 && keep[j]) {
      if (i >= 16) return false;
      out->dneg[i] = d;
      out->uneg[i++] = used[j];
    }
  }
  out->nneg = i;
  i = 0;
  for (int j = r.nneg - 1; j >= 0; --j) {
    const int d = r.dneg[j] + drps;
    if (d > 0 && keep[j]) {
      if (i >= 16) return false;
      out->dpos[i] = d;
      out->upos[i++] = used[j];
    }
  }
  if (drps > 0 && keep[nr]) {
    if (i >= 16) return false;
    out->dpos[i] = drps;
    out->upos[i++] = used[nr];
  }
  for (int j = 0; j < r.npos; ++j) {
    const int d = r.dpos[j] + drps;
    if (d > 0 && keep[r.nneg + j]) {
      if (i >= 16) return false;
      out->dpos[i] = d;
      out->upos[i++] = used[r.nneg + j];
    }
  }
  out->npos = i;
  return !b.bad();
}

// ------------------------------------------------------------------------------------------------------- CABAC
// 9.3.4.3 with the offset kept scaled: `value` holds ivlOffset in its top bits and `bits` not yet consumed bits
// below it, so a renormalisation is a subtraction from `bits` and a refill happens once per 32 bits.
struct NextCtx {
  uint8_t mlps[256];   // the next context byte: [128 + state] after the more probable symbol, [127 - state] after the other
  uint32_t lps4[128];  // [state = pStateIdx << 1 | valMps]: rangeTabLps for qRangeIdx 0 .. 3, a byte each - the load depends
                       // on the context only, so it is off the range -> range chain that paces consecutive bins
  NextCtx() {
    for (int st = 0; st < 64; ++st)
      for (int mps = 0; mps < 2; ++mps) {
        const int s = st << 1 | mps;
        mlps[128 + s] = static_cast<uint8_t>((st < 62 ? st + 1 : st) << 1 | mps);
        mlps[127 - s] = static_cast<uint8_t>(hevc_trans_lps[st] << 1 | (st == 0 ? !mps : mps));
        lps4[s] = 0;
        for (int q = 0; q < 4; ++q) lps4[s] |= static_cast<uint32_t>(hevc_range_lps[st][q]) << (8 * q);
      }
  }
};
static const NextCtx kNextCtx;

class Cabac {
 public:
  void start(const uint8_t* p, size_t n, size_t at) {
    p_ = p;
    n_ = n;
    broken_ = false;
    restart(at);
  }
  void restart(size_t at) {
    pos_ = at;
    value_ = 0;
    bits_ = -9;
    range_ = 510;
    refill();
    // 9.3.2.5: ivlOffset is never 510 or 511 in a conforming stream; in a damaged one it may be, and every step of the
    // engine relies on offset < range (a bypass read would return more than its n bits).  Forced inside; the slice ends
    // in an error through its end-of-slice checks.
    const uint64_t limit = static_cast<uint64_t>(range_) << bits_;
    if (value_ >= limit) {
      value_ = limit - 1;
      broken_ = true;
    }
  }
  // bytes consumed once the engine has terminated: the stop bit is the last bit the nine-bit window took
  size_t aligned_end() const {
    const size_t consumed = pos_ * 8 - static_cast<size_t>(bits_);
    return (consumed + 7) >> 3;
  }
  bool overrun() const { return broken_ || pos_ > n_ + 8; }

  // 9.3.4.3.2, as h264_parser.cpp's cabac_step has it (counted there on the box's CPU: profiles/r05_host_perf_counters.txt).
  // The context is state << 1 | MPS in a 16-bit word - a byte store may alias anything, which would force the engine's
  // own members through memory around every bin.  The outcome is a sign: state ^ ~sign indexes ONE table for the next
  // state on either side and its low bit is the bin; offset, range and next state are selected without a branch (which
  // way a bin goes is the least predictable branch of the parser).
  int decision(uint16_t* ctx) {
    const int32_t s = *ctx;
    const uint32_t lps = (kNextCtx.lps4[s] >> ((range_ >> 3) & 24)) & 0xff;
    uint32_t r = range_ - lps;
    const uint64_t scaled = static_cast<uint64_t>(r) << bits_;
    const int64_t d = static_cast<int64_t>(value_ - scaled);  // >= 0: the less probable symbol
    value_ = d >= 0 ? static_cast<uint64_t>(d) : value_;
    r = d >= 0 ? lps : r;
    const int32_t sx = s ^ static_cast<int32_t>(~(d >> 63));
    *ctx = kNextCtx.mlps[128 + sx];
    const int n = __builtin_clz(r) - 23;  // 0 when r >= 256
    range_ = r << n;
    bits_ -= n;
    if (bits_ < 0) refill();
    return sx & 1;
  }
  int bypass() {
    if (--bits_ < 0) refill();
    const uint64_t scaled = static_cast<uint64_t>(range_) << bits_;
    const int64_t d = static_cast<int64_t>(value_ - scaled);  // (offset < 2^62: the sign of the difference is the bin)
    value_ = d >= 0 ? static_cast<uint64_t>(d) : value_;
    return static_cast<int>(~d >> 63) & 1;
  }
  // n bypass bins at once (n <= 16): the offset divided by the range is the n bins read as a number
  unsigned bypass_bits(int n) {
    if (n <= 1) return n ? static_cast<unsigned>(bypass()) : 0u;
    if (bits_ < n) refill();
    bits_ -= n;
    const uint64_t scaled = static_cast<uint64_t>(range_) << bits_;
    const uint64_t q = value_ / scaled;
    value_ -= q * scaled;
    return static_cast<unsigned>(q);
  }
  int terminate() {
    range_ -= 2;
    const uint64_t scaled = static_cast<uint64_t>(range_) << bits_;
    if (value_ >= scaled) return 1;
    if (range_ < 256) {
      range_ <<= 1;
      if (--bits_ < 0) refill();
    }
    return 0;
  }
  // raw bits behind a terminated engine (pcm_sample)
  const uint8_t* raw() const { return p_; }
  size_t size() const { return n_; }

 private:
  void refill() {  // (bits_ >= -9: at most 9 + 32 + ... bits of offset stay well below 64)
    uint32_t w = 0;
    for (int i = 0; i < 4; ++i) w = w << 8 | (pos_ + i < n_ ? p_[pos_ + i] : 0);
    pos_ += 4;
    value_ = value_ << 32 | w;
    bits_ += 32;
  }
  const uint8_t* p_ = nullptr;
  size_t n_ = 0, pos_ = 0;
  uint64_t value_ = 0;
  int bits_ = 0;
  uint32_t range_ = 510;
  bool broken_ = false;  // the engine was started on bits that are not an arithmetic codeword
};

// ------------------------------------------------------------------------------------------------- pictures
struct Motion {  // per prediction block, 16 bytes
  int16_t mv[2][2];
  int8_t ref[2];
  uint8_t flags;   // bit 0 / 1: list used
  uint8_t rid[2];  // which picture each list refers to: its DPB entry while the picture that holds this field was
                   // decoded (Pic::ref_poc_tab / ref_lt_tab say what that entry was: later pictures' temporal candidates)
  uint8_t pad[3];
};
static_assert(sizeof(Motion) == 16, "Motion");

struct Pic {
  bool in_use = false;
  int slot = -1;
  int poc = 0, ref = 0 /* 0 none, 1 short, 2 long */, latency = 0;
  bool waiting = false;  // for output
  bool placeholder = false;  // skipped on purpose / made up for a missing reference
  bool made_up = false;      // ... the latter: gets a surface of 128s when something predicts from it
  int32_t ref_poc_tab[amd::hevc::kMaxSurfaces] = {0};  // POC / long-term state of DPB entry i when this picture referred to it
  uint8_t ref_lt_tab[amd::hevc::kMaxSurfaces] = {0};
  // Motion: one entry per prediction block ([0]: none - intra, or a placeholder picture), and per 4x4 luma block, in
  // raster order, which entry covers it.  (A field of 16-byte motions per 4x4 block was 2 MB written per 1080p picture, 13 %
  // of a P picture's parse; the index is a quarter of it, and two blocks of one prediction block compare equal by index.)
  std::vector<Motion> pu_motion;
  std::vector<uint32_t> pu_of;
};

// sig_coeff_flag's context (9.3.4.2.5) by scan position: the pattern of a 4x4 sub-block given which of its right / lower
// neighbours are coded (sigCtx 0..2 before the per-size offsets), and Table 9-41's map for 4x4 blocks - per scan type.
struct SigTables {
  uint8_t pat[3][4][16];
  uint8_t map4[3][16];
};
struct Scans {
  uint8_t t[4][3][64];
  uint8_t inv[4][3][64];  // position (x | y << log2) -> index in the scan
  Scans() {
    for (int l = 0; l < 4; ++l) {
      const int n = 1 << l;
      int i = 0, x = 0, y = 0;
      for (;;) {
        while (y >= 0) {
          if (x < n && y < n) t[l][0][i++] = static_cast<uint8_t>(x | y << 4);
          --y;
          ++x;
        }
        y = x;
        x = 0;
        if (i >= n * n) break;
      }
      i = 0;
      for (y = 0; y < n; ++y)
        for (x = 0; x < n; ++x) t[l][1][i++] = static_cast<uint8_t>(x | y << 4);
      i = 0;
      for (x = 0; x < n; ++x)
        for (y = 0; y < n; ++y) t[l][2][i++] = static_cast<uint8_t>(x | y << 4);
      for (int sc = 0; sc < 3; ++sc)
        for (int k = 0; k < n * n; ++k) inv[l][sc][(t[l][sc][k] & 15) | (t[l][sc][k] >> 4) << l] = static_cast<uint8_t>(k);
    }
  }
};
const Scans& scans() {
  static const Scans s;
  return s;
}
const SigTables& sig_tables() {
  static const SigTables t = [] {
    SigTables r;
    const Scans& sc = scans();
    for (int scan = 0; scan < 3; ++scan)
      for (int k = 0; k < 16; ++k) {
        const int xp = sc.t[2][scan][k] & 15, yp = sc.t[2][scan][k] >> 4;
        r.pat[scan][0][k] = static_cast<uint8_t>(xp + yp == 0 ? 2 : xp + yp < 3 ? 1 : 0);
        r.pat[scan][1][k] = static_cast<uint8_t>(yp == 0 ? 2 : yp == 1 ? 1 : 0);
        r.pat[scan][2][k] = static_cast<uint8_t>(xp == 0 ? 2 : xp == 1 ? 1 : 0);
        r.pat[scan][3][k] = 2;
        r.map4[scan][k] = hevc_ctx_idx_map4[(yp << 2) + xp];
      }
    return r;
  }();
  return t;
}

enum { PART_2Nx2N = 0, PART_2NxN, PART_Nx2N, PART_NxN, PART_2NxnU, PART_2NxnD, PART_nLx2N, PART_nRx2N };

}  // namespace

// ====================================================================================================== Parser
struct Parser::Impl {
  PictureSink* sink = nullptr;
  std::string error;
  int nal_length_size = 4;
  Sps sps[16];
  Pps pps[64];
  const Sps* S = nullptr;
  const Pps* P = nullptr;
  int64_t decoded = 0, skipped = 0;
  std::vector<uint64_t> wanted;
  std::vector<uint64_t> keyframes;   // decoding-order ordinals of the interval's closed-GOP sync samples (optional)
  uint64_t pictures_started = 0;     // pictures (= samples) since the interval began
  uint64_t outputs_emitted = 0;

  // DPB
  Pic dpb[amd::hevc::kMaxSurfaces];
  Pic* cur = nullptr;
  int prev_tid0_poc = 0;
  bool first_picture = true;
  int poc = 0;
  Pic* st_before[16];
  Pic* st_after[16];
  Pic* lt_cur[16];
  int n_before = 0, n_after = 0, n_lt = 0;
  Pic* list[2][16];

  // current picture
  bool pic_open = false, skipping = false;
  PicBuffers buf;
  PicDone done;
  int w4 = 0, h4 = 0;
  int next_ctb = 0;  // the picture's coding tree blocks before this address are decoded
  bool pic_deblock = false;  // a coding unit of the picture lies in a slice with the deblocking filter on
  int pic_w = 0, pic_h = 0, pic_lc = 0, pic_wctb = 0, pic_zmask = 0;  // the active SPS's, where every neighbour test wants them
  std::vector<uint8_t> m_pred, m_skip, m_depth, m_mode, m_flags, m_nz, m_edge;
  std::vector<int8_t> m_qp;
  std::vector<uint16_t> ctb_slice;   // slice index per coding tree block, 0xffff = not decoded
  std::vector<uint32_t> slice_addr;  // SliceAddrRs per slice
  uint32_t zmap[256];                // Morton order of a 16 x 16 grid of 4x4 blocks
  int num_slices = 0;

  // current slice
  SliceHdr sh;
  int slice_idx = 0;
  Cabac cabac;
  uint16_t ctx[HC_CTX_COUNT], ctx_saved[HC_CTX_COUNT];  // (16-bit on purpose: Cabac::decision)
  bool qp_delta_coded = false;
  int qp_delta_val = 0, qp_prev = 0, last_qp = 0, qg_x = 0, qg_y = 0;
  int weights_base[2] = {0, 0};

  bool fail(const char* fmt, ...) {
    if (error.empty()) {
      char tmp[256];
      va_list ap;
      va_start(ap, fmt);
      vsnprintf(tmp, sizeof(tmp), fmt, ap);
      va_end(ap);
      error = tmp;
    }
    return false;
  }

  Impl() {
    for (int i = 0; i < 256; ++i) {
      uint32_t m = 0;
      for (int b = 0; b < 4; ++b) m |= ((i >> b) & 1u) << (2 * b) | ((i >> (4 + b)) & 1u) << (2 * b + 1);
      zmap[i] = m;  // i = x | y << 4
    }
  }

  // ------------------------------------------------------------------------------------------ NAL layer
  static size_t unescape(const uint8_t* p, size_t n, uint8_t* out) {
    size_t o = 0;
    int zeros = 0;
    for (size_t i = 0; i < n; ++i) {
      if (zeros >= 2 && p[i] == 3) {
        zeros = 0;
        continue;
      }
      out[o++] = p[i];
      zeros = p[i] == 0 ? zeros + 1 : 0;
    }
    return o;
  }

  bool parse_sps(Bits& b) {
    Sps s;
    b.u(4);
    const int msl = static_cast<int>(b.u(3));
    s.max_sub_layers = msl + 1;
    b.u(1);
    // profile_tier_level(1, msl)
    b.u(8); b.u(32); b.u(32); b.u(16); b.u(8);
    uint8_t pp[8], lp[8];
    for (int i = 0; i < msl; ++i) { pp[i] = static_cast<uint8_t>(b.u(1)); lp[i] = static_cast<uint8_t>(b.u(1)); }
    if (msl > 0) for (int i = msl; i < 8; ++i) b.u(2);
    for (int i = 0; i < msl; ++i) {
      if (pp[i]) { b.u(32); b.u(32); b.u(24); }
      if (lp[i]) b.u(8);
    }
    const unsigned id = b.ue();
    if (id > 15) return fail("SPS id %u", id);
    const unsigned chroma = b.ue();
    if (chroma == 3) b.u(1);
    s.width = static_cast<int>(b.ue());
    s.height = static_cast<int>(b.ue());
    if (b.u(1)) {
      s.crop_l = 2 * static_cast<int>(b.ue()); s.crop_r = 2 * static_cast<int>(b.ue());
      s.crop_t = 2 * static_cast<int>(b.ue()); s.crop_b = 2 * static_cast<int>(b.ue());
    }
    const unsigned bdy = b.ue(), bdc = b.ue();
    s.log2_max_poc_lsb = 4 + static_cast<int>(b.ue());
    const bool ordering = b.u(1) != 0;
    for (int i = ordering ? 0 : msl; i <= msl; ++i) {
      s.max_dec_pic_buffering = static_cast<int>(b.ue()) + 1;
      s.max_num_reorder = static_cast<int>(b.ue());
      s.max_latency_plus1 = static_cast<int>(b.ue());
    }
    s.log2_min_cb = 3 + static_cast<int>(b.ue());
    s.log2_ctb = s.log2_min_cb + static_cast<int>(b.ue());
    s.log2_min_tb = 2 + static_cast<int>(b.ue());
    s.log2_max_tb = s.log2_min_tb + static_cast<int>(b.ue());
    s.th_inter = static_cast<int>(b.ue());
    s.th_intra = static_cast<int>(b.ue());
    s.scaling_enabled = b.u(1) != 0;
    scaling_defaults(&s.scaling);
    if (s.scaling_enabled && b.u(1) && !scaling_parse(b, &s.scaling)) return fail("SPS: bad scaling_list_data");
    s.amp = b.u(1) != 0;
    s.sao = b.u(1) != 0;
    s.pcm = b.u(1) != 0;
    if (s.pcm) {
      s.pcm_bits_y = 1 + static_cast<int>(b.u(4));
      s.pcm_bits_c = 1 + static_cast<int>(b.u(4));
      s.log2_min_pcm = 3 + static_cast<int>(b.ue());
      s.log2_max_pcm = s.log2_min_pcm + static_cast<int>(b.ue());
      s.pcm_lf_off = b.u(1) != 0;
    }
    s.num_rps = static_cast<int>(b.ue());
    if (s.num_rps > 64) return fail("SPS: %d reference picture sets", s.num_rps);
    for (int i = 0; i < s.num_rps; ++i)
      if (!parse_rps(b, &s.rps[i], i, s.num_rps, s.rps)) return fail("SPS: bad st_ref_pic_set %d", i);
    s.lt_present = b.u(1) != 0;
    if (s.lt_present) {
      s.num_lt = static_cast<int>(b.ue());
      if (s.num_lt > 32) return fail("SPS: %d long-term pictures", s.num_lt);
      for (int i = 0; i < s.num_lt; ++i) {
        s.lt_lsb[i] = static_cast<int>(b.u(s.log2_max_poc_lsb));
        s.lt_used[i] = static_cast<uint8_t>(b.u(1));
      }
    }
    s.tmvp = b.u(1) != 0;
    s.strong = b.u(1) != 0;
    if (b.bad()) return fail("SPS: ran out of data");
    if (chroma != 1) return fail("HEVC chroma_format_idc %u is not supported by this backend (4:2:0 only)", chroma);
    if (bdy || bdc) return fail("HEVC bit depth %u / %u is not supported by this backend (8 bit only)", 8 + bdy, 8 + bdc);
    if (s.log2_ctb < 4 || s.log2_ctb > 6 || s.log2_max_tb > 5 || s.log2_min_tb >= s.log2_min_cb || s.log2_max_tb > s.log2_ctb || s.log2_min_cb > s.log2_ctb)
      return fail("SPS: block sizes out of range (CTB %d, CB %d, TB %d..%d)", s.log2_ctb, s.log2_min_cb, s.log2_min_tb, s.log2_max_tb);
    const int mcb = 1 << s.log2_min_cb;
    if (s.width <= 0 || s.height <= 0 || s.width > 8192 || s.height > 8192 || s.width % mcb || s.height % mcb) return fail("SPS: picture size %dx%d", s.width, s.height);
    if (s.pcm && (s.log2_max_pcm > 5 || s.log2_max_pcm > s.log2_ctb || s.log2_min_pcm < s.log2_min_cb || s.pcm_bits_y > 8 || s.pcm_bits_c > 8))
      return fail("SPS: pcm parameters out of range");
    // (every one of these is a shift count, a loop bound or a table index further down)
    if (s.log2_max_poc_lsb > 16) return fail("SPS: log2_max_pic_order_cnt_lsb %d", s.log2_max_poc_lsb);
    if (s.max_dec_pic_buffering < 1 || s.max_dec_pic_buffering > 16 || s.max_num_reorder < 0 || s.max_num_reorder >= s.max_dec_pic_buffering + (s.max_dec_pic_buffering == 1))
      return fail("SPS: decoded picture buffer of %d pictures, %d reordered", s.max_dec_pic_buffering, s.max_num_reorder);
    if (s.th_inter < 0 || s.th_intra < 0 || s.th_inter > s.log2_ctb - s.log2_min_tb || s.th_intra > s.log2_ctb - s.log2_min_tb)
      return fail("SPS: max_transform_hierarchy_depth %d / %d", s.th_inter, s.th_intra);
    if (s.crop_l < 0 || s.crop_r < 0 || s.crop_t < 0 || s.crop_b < 0 || s.crop_l + s.crop_r >= s.width || s.crop_t + s.crop_b >= s.height)
      return fail("SPS: conformance window %d %d %d %d of %dx%d", s.crop_l, s.crop_r, s.crop_t, s.crop_b, s.width, s.height);
    s.wctb = (s.width + (1 << s.log2_ctb) - 1) >> s.log2_ctb;
    s.hctb = (s.height + (1 << s.log2_ctb) - 1) >> s.log2_ctb;
    s.valid = true;
    sps[id] = s;
    return true;
  }

  bool parse_pps(Bits& b) {
    Pps p;
    const unsigned id = b.ue(), sid = b.ue();
    if (id > 63 || sid > 15) return fail("PPS id %u / SPS id %u", id, sid);
    p.sps_id = static_cast<int>(sid);
    p.dependent_slices = b.u(1) != 0;
    p.output_flag_present = b.u(1) != 0;
    p.extra_bits = static_cast<int>(b.u(3));
    p.sign_hiding = b.u(1) != 0;
    p.cabac_init_present = b.u(1) != 0;
    p.nref_default[0] = 1 + static_cast<int>(b.ue());
    p.nref_default[1] = 1 + static_cast<int>(b.ue());
    p.init_qp = 26 + b.se();
    p.cip = b.u(1) != 0;
    p.tskip = b.u(1) != 0;
    p.qp_delta = b.u(1) != 0;
    if (p.qp_delta) p.qp_delta_depth = static_cast<int>(b.ue());
    p.cb_off = b.se();
    p.cr_off = b.se();
    p.slice_chroma_offsets = b.u(1) != 0;
    p.wp = b.u(1) != 0;
    p.wbp = b.u(1) != 0;
    p.bypass = b.u(1) != 0;
    const bool tiles = b.u(1) != 0;
    p.wpp = b.u(1) != 0;
    if (tiles) return fail("HEVC tiles are not supported by this backend");
    p.lf_across = b.u(1) != 0;
    if (b.u(1)) {
      p.db_override = b.u(1) != 0;
      p.db_off = b.u(1) != 0;
      if (!p.db_off) {
        p.beta_off = b.se();
        p.tc_off = b.se();
      }
    }
    p.scaling_present = b.u(1) != 0;
    scaling_defaults(&p.scaling);
    if (p.scaling_present && !scaling_parse(b, &p.scaling)) return fail("PPS: bad scaling_list_data");
    p.lists_mod = b.u(1) != 0;
    p.pml = 2 + static_cast<int>(b.ue());
    p.sh_ext = b.u(1) != 0;
    if (b.bad()) return fail("PPS: ran out of data");
    if (p.nref_default[0] > 15 || p.nref_default[1] > 15 || p.init_qp < 0 || p.init_qp > 51 || p.qp_delta_depth < 0 || p.qp_delta_depth > 3 || p.cb_off < -12 ||
        p.cb_off > 12 || p.cr_off < -12 || p.cr_off > 12 || p.beta_off < -6 || p.beta_off > 6 || p.tc_off < -6 || p.tc_off > 6 || p.pml < 2 || p.pml > 6)
      return fail("PPS: a parameter is out of its range (init_qp %d, offsets %d %d %d %d, depth %d, merge level %d)", p.init_qp, p.cb_off, p.cr_off, p.beta_off,
                  p.tc_off, p.qp_delta_depth, p.pml);
    p.valid = true;
    pps[id] = p;
    return true;
  }

  bool parse_slice_header(Bits& b, int nal_type, SliceHdr* h) {
    *h = SliceHdr();
    h->first = b.u(1) != 0;
    if (nal_type >= 16 && nal_type <= 23) b.u(1);
    const unsigned pid = b.ue();
    if (pid > 63 || !pps[pid].valid) return fail("slice refers to picture parameter set %u, which was never sent", pid);
    P = &pps[pid];
    if (!sps[P->sps_id].valid) return fail("slice refers to sequence parameter set %d, which was never sent", P->sps_id);
    S = &sps[P->sps_id];
    if (P->pml > S->log2_ctb || P->qp_delta_depth > S->log2_ctb - S->log2_min_cb)
      return fail("picture parameter set %u does not fit its sequence parameter set (merge level %d, quantisation group depth %d)", pid, P->pml, P->qp_delta_depth);
    h->pps_id = static_cast<int>(pid);
    const int ctbs = S->wctb * S->hctb;
    if (!h->first) {
      if (P->dependent_slices) h->dependent = b.u(1) != 0;
      h->address = static_cast<int>(b.u(ceil_log2(static_cast<unsigned>(ctbs))));
      if (h->address >= ctbs) return fail("slice_segment_address %d outside the picture", h->address);
    }
    if (h->dependent) return fail("HEVC dependent slice segments are not supported by this backend");
    for (int i = 0; i < P->extra_bits; ++i) b.u(1);
    h->type = static_cast<int>(b.ue());
    if (h->type > 2) return fail("slice_type %d", h->type);
    if (P->output_flag_present) b.u(1);
    const bool idr = nal_type == 19 || nal_type == 20;
    if (!idr) {
      h->poc_lsb = static_cast<int>(b.u(S->log2_max_poc_lsb));
      h->rps_from_sps = b.u(1) != 0;
      if (!h->rps_from_sps) {
        if (!parse_rps(b, &h->rps, S->num_rps, S->num_rps, S->rps)) return fail("slice header: bad st_ref_pic_set");
      } else {
        if (!S->num_rps) return fail("slice header selects a reference picture set of an SPS that has none");
        int idx = 0;
        if (S->num_rps > 1) idx = static_cast<int>(b.u(ceil_log2(static_cast<unsigned>(S->num_rps))));
        if (idx >= S->num_rps) return fail("short_term_ref_pic_set_idx %d", idx);
        h->rps = S->rps[idx];
      }
      h->have_rps = true;
      if (S->lt_present) {
        int from_sps = 0;
        if (S->num_lt > 0) from_sps = static_cast<int>(b.ue());
        const int own = static_cast<int>(b.ue());
        h->num_lt = from_sps + own;
        if (h->num_lt > 32 || from_sps > S->num_lt) return fail("slice header: %d long-term pictures", h->num_lt);
        for (int i = 0; i < h->num_lt; ++i) {
          if (i < from_sps) {
            int idx = 0;
            if (S->num_lt > 1) idx = static_cast<int>(b.u(ceil_log2(static_cast<unsigned>(S->num_lt))));
            if (idx >= S->num_lt) return fail("lt_idx_sps %d", idx);
            h->lt_lsb[i] = S->lt_lsb[idx];
            h->lt_used[i] = S->lt_used[idx];
          } else {
            h->lt_lsb[i] = static_cast<int>(b.u(S->log2_max_poc_lsb));
            h->lt_used[i] = static_cast<uint8_t>(b.u(1));
          }
          h->lt_msb[i] = static_cast<uint8_t>(b.u(1));
          h->lt_cycle[i] = h->lt_msb[i] ? static_cast<int>(b.ue()) : 0;
          if (i != 0 && i != from_sps) h->lt_cycle[i] += h->lt_cycle[i - 1];
        }
      }
      if (S->tmvp) h->tmvp = b.u(1) != 0;
    }
    if (S->sao) {
      h->sao_y = b.u(1) != 0;
      h->sao_c = b.u(1) != 0;
    }
    if (h->type != SL_I) {
      h->nref[0] = P->nref_default[0];
      h->nref[1] = h->type == SL_B ? P->nref_default[1] : 0;
      if (b.u(1)) {
        h->nref[0] = 1 + static_cast<int>(b.ue());
        if (h->type == SL_B) h->nref[1] = 1 + static_cast<int>(b.ue());
      }
      if (h->nref[0] > 15 || h->nref[1] > 15) return fail("num_ref_idx_active %d / %d", h->nref[0], h->nref[1]);
      int total = 0;
      for (int i = 0; i < h->rps.nneg; ++i) total += h->rps.uneg[i];
      for (int i = 0; i < h->rps.npos; ++i) total += h->rps.upos[i];
      for (int i = 0; i < h->num_lt; ++i) total += h->lt_used[i];
      if (!total) return fail("a P or B slice whose reference picture set is empty");
      if (P->lists_mod && total > 1) {
        const int bits = ceil_log2(static_cast<unsigned>(total));
        for (int l = 0; l < (h->type == SL_B ? 2 : 1); ++l) {
          h->rplm[l] = b.u(1) != 0;
          if (h->rplm[l])
            for (int i = 0; i < h->nref[l]; ++i) {
              h->entry[l][i] = static_cast<int>(b.u(bits));
              if (h->entry[l][i] >= total) return fail("list_entry %d of %d", h->entry[l][i], total);
            }
        }
      }
      if (h->type == SL_B) h->mvd_l1_zero = b.u(1) != 0;
      if (P->cabac_init_present) h->cabac_init = b.u(1) != 0;
      if (h->tmvp) {
        if (h->type == SL_B) h->col_from_l0 = b.u(1) != 0;
        if (h->nref[h->col_from_l0 ? 0 : 1] > 1) h->col_ref = static_cast<int>(b.ue());
        if (h->col_ref >= h->nref[h->col_from_l0 ? 0 : 1]) return fail("collocated_ref_idx %d", h->col_ref);
      }
      if ((P->wp && h->type == SL_P) || (P->wbp && h->type == SL_B)) {  // 7.3.6.3
        h->wd_y = static_cast<int>(b.ue());
        h->wd_c = h->wd_y + b.se();
        if (h->wd_y > 7 || h->wd_c < 0 || h->wd_c > 7) return fail("bad pred_weight_table");
        for (int l = 0; l < (h->type == SL_B ? 2 : 1); ++l) {
          uint8_t fy[16], fc[16];
          for (int i = 0; i < h->nref[l]; ++i) fy[i] = static_cast<uint8_t>(b.u(1));
          for (int i = 0; i < h->nref[l]; ++i) fc[i] = static_cast<uint8_t>(b.u(1));
          for (int i = 0; i < h->nref[l]; ++i) {
            h->wy[l][i] = static_cast<int16_t>(1 << h->wd_y);
            h->oy[l][i] = 0;
            if (fy[i]) {
              h->wy[l][i] = static_cast<int16_t>(h->wy[l][i] + b.se());
              h->oy[l][i] = static_cast<int16_t>(b.se());
            }
            for (int j = 0; j < 2; ++j) {
              h->wc[l][i][j] = static_cast<int16_t>(1 << h->wd_c);
              h->oc[l][i][j] = 0;
              if (fc[i]) {
                const int w = (1 << h->wd_c) + b.se();
                const int d = b.se();
                h->wc[l][i][j] = static_cast<int16_t>(w);
                h->oc[l][i][j] = static_cast<int16_t>(clip3(-128, 127, 128 + d - ((128 * w) >> h->wd_c)));
              }
            }
          }
        }
      }
      h->max_merge = 5 - static_cast<int>(b.ue());
      if (h->max_merge < 1 || h->max_merge > 5) return fail("MaxNumMergeCand %d", h->max_merge);
    }
    h->qp = P->init_qp + b.se();
    if (P->slice_chroma_offsets) {
      h->cb_off = b.se();
      h->cr_off = b.se();
    }
    bool override = false;
    if (P->db_override) override = b.u(1) != 0;
    h->db_off = P->db_off;
    h->beta_off = P->beta_off;
    h->tc_off = P->tc_off;
    if (override) {
      h->db_off = b.u(1) != 0;
      if (!h->db_off) {
        h->beta_off = b.se();
        h->tc_off = b.se();
      }
    }
    h->across = P->lf_across;
    if (P->lf_across && (h->sao_y || h->sao_c || !h->db_off)) h->across = b.u(1) != 0;
    if (P->wpp) {
      const unsigned n = b.ue();
      if (n > 0) {
        const int len = 1 + static_cast<int>(b.ue());
        if (len > 32 || n > 4096) return fail("entry points: %u of %d bits", n, len);
        for (unsigned i = 0; i < n; ++i) b.u(len);  // (rows are found by walking the data)
      }
    }
    if (P->sh_ext) {
      const unsigned n = b.ue();
      for (unsigned i = 0; i < n && !b.bad(); ++i) b.u(8);
    }
    if (b.u(1) != 1) return fail("slice header does not end on its alignment bit");
    while (b.pos() & 7)
      if (b.u(1)) return fail("slice header alignment bits are not zero");
    if (b.bad()) return fail("slice header: ran out of data");
    if (h->qp < 0 || h->qp > 51) return fail("SliceQpY %d", h->qp);
    if (h->beta_off < -6 || h->beta_off > 6 || h->tc_off < -6 || h->tc_off > 6 || P->cb_off + h->cb_off < -12 || P->cb_off + h->cb_off > 12 ||
        P->cr_off + h->cr_off < -12 || P->cr_off + h->cr_off > 12)
      return fail("slice header: deblocking / chroma offsets out of range (%d %d %d %d)", h->beta_off, h->tc_off, h->cb_off, h->cr_off);
    h->data_offset = b.pos() >> 3;
    return true;
  }

  // ------------------------------------------------------------------------------------------------ DPB
  Pic* grab_pic() {
    for (Pic& p : dpb)
      if (!p.in_use) {
        p = Pic();
        p.in_use = true;
        return &p;
      }
    return nullptr;
  }
  // The motion index maps (0.5 MB at 1080p) and the motion lists circulate between the pictures of the DPB instead of
  // being allocated - and paged in - per picture.
  std::vector<std::vector<uint32_t>> motion_pool;
  std::vector<std::vector<Motion>> pu_pool;
  void drop(Pic* p) {
    if (p->slot >= 0) sink->release_surface(p->slot);
    p->in_use = false;
    p->slot = -1;
    if (p->pu_of.capacity()) motion_pool.push_back(std::move(p->pu_of));
    if (p->pu_motion.capacity()) pu_pool.push_back(std::move(p->pu_motion));
    p->pu_of = std::vector<uint32_t>();
    p->pu_motion = std::vector<Motion>();
  }
  void release_unused() {
    for (Pic& p : dpb)
      if (p.in_use && &p != cur && !p.ref && !p.waiting) drop(&p);
  }
  bool bump() {
    Pic* best = nullptr;
    for (Pic& p : dpb)
      if (p.in_use && p.waiting && (!best || p.poc < best->poc)) best = &p;
    if (!best) return false;
    best->waiting = false;
    sink->output_picture(best->placeholder ? -1 : best->slot);
    ++outputs_emitted;
    return true;
  }
  int waiting() const {
    int n = 0;
    for (const Pic& p : dpb) n += p.in_use && p.waiting;
    return n;
  }
  int fullness() const {
    int n = 0;
    for (const Pic& p : dpb) n += p.in_use && &p != cur;
    return n;
  }
  bool latency_hit() const {
    if (!S->max_latency_plus1) return false;
    const int lim = S->max_num_reorder + S->max_latency_plus1 - 1;
    for (const Pic& p : dpb)
      if (p.in_use && p.waiting && p.latency >= lim) return true;
    return false;
  }
  Pic* find_ref(int want, int mask, bool short_only) {
    for (Pic& p : dpb) {
      if (!p.in_use || &p == cur || !p.ref) continue;
      if (short_only && p.ref != 1) continue;
      if (mask ? (p.poc & mask) == want : p.poc == want) return &p;
    }
    return nullptr;
  }
  // 8.3.3: a reference that was never decoded (a stream entered at a CRA picture): a stand-in without motion; it gets
  // a surface of 128s (8.3.3.2) when a prediction block first refers to it - the CRA picture's leading pictures do,
  // and they are output like every other picture (one frame per sample: the callers count)
  Pic* make_missing(int poc_value, int state) {
    Pic* p = grab_pic();
    if (!p) return nullptr;
    p->poc = poc_value;
    p->ref = state;
    p->placeholder = true;
    p->made_up = true;
    p->pu_of.assign(S ? static_cast<size_t>(S->width / 4) * (S->height / 4) : 0, 0u);
    p->pu_motion.assign(1, Motion());
    return p;
  }

  // 8.3.1, 8.3.2, C.5.2.2
  bool start_picture(int nal_type, int tid) {
    const bool irap = nal_type >= 16 && nal_type <= 23, idr = nal_type == 19 || nal_type == 20;
    const bool no_rasl = irap && (idr || nal_type <= 18 || first_picture);
    const int max_lsb = 1 << S->log2_max_poc_lsb;
    int msb = 0;
    if (!(irap && no_rasl)) {
      const int pl = prev_tid0_poc & (max_lsb - 1), pm = prev_tid0_poc - pl;
      if (sh.poc_lsb < pl && pl - sh.poc_lsb >= max_lsb / 2) msb = pm + max_lsb;
      else if (sh.poc_lsb > pl && sh.poc_lsb - pl > max_lsb / 2) msb = pm - max_lsb;
      else msb = pm;
    }
    poc = msb + sh.poc_lsb;
    if (tid == 0 && !(nal_type >= 6 && nal_type <= 9) && !(nal_type <= 14 && !(nal_type & 1))) prev_tid0_poc = poc;
    cur = nullptr;
    if (irap && no_rasl) {
      while (bump()) {
      }
      for (Pic& p : dpb)
        if (p.in_use) {
          p.ref = 0;
          drop(&p);
        }
    }
    n_before = n_after = n_lt = 0;
    bool keep[amd::hevc::kMaxSurfaces] = {false};
    if (!idr && sh.have_rps) {
      for (int i = 0; i < sh.num_lt; ++i) {
        int pl = sh.lt_lsb[i];
        if (sh.lt_msb[i]) pl += poc - sh.lt_cycle[i] * max_lsb - (poc & (max_lsb - 1));
        Pic* p = find_ref(pl, sh.lt_msb[i] ? 0 : max_lsb - 1, false);
        if (!p && sh.lt_used[i]) p = make_missing(pl, 2);
        if (p) {
          p->ref = 2;
          keep[p - dpb] = true;
          if (sh.lt_used[i] && n_lt < 16) lt_cur[n_lt++] = p;
        }
      }
      for (int i = 0; i < sh.rps.nneg; ++i) {
        Pic* p = find_ref(poc + sh.rps.dneg[i], 0, true);
        if (!p && sh.rps.uneg[i]) p = make_missing(poc + sh.rps.dneg[i], 1);
        if (p) {
          keep[p - dpb] = true;
          if (sh.rps.uneg[i]) st_before[n_before++] = p;
        }
      }
      for (int i = 0; i < sh.rps.npos; ++i) {
        Pic* p = find_ref(poc + sh.rps.dpos[i], 0, true);
        if (!p && sh.rps.upos[i]) p = make_missing(poc + sh.rps.dpos[i], 1);
        if (p) {
          keep[p - dpb] = true;
          if (sh.rps.upos[i]) st_after[n_after++] = p;
        }
      }
    }
    for (int i = 0; i < amd::hevc::kMaxSurfaces; ++i)
      if (dpb[i].in_use && !keep[i]) dpb[i].ref = 0;
    release_unused();
    while (waiting() > S->max_num_reorder || latency_hit() || fullness() >= std::max(S->max_dec_pic_buffering, 1)) {
      if (!bump()) break;
      release_unused();
    }
    cur = grab_pic();
    if (!cur) return fail("decoded picture buffer is full (%d pictures held)", fullness());
    cur->poc = poc;
    first_picture = false;
    return true;
  }

  bool build_lists() {  // 8.3.4
    if (sh.type == SL_I) return true;
    const int total = n_before + n_after + n_lt;
    if (!total) return fail("no reference pictures for a P / B slice");
    for (int l = 0; l < (sh.type == SL_B ? 2 : 1); ++l) {
      Pic* temp[32];
      const int nt = std::max(sh.nref[l], total);
      int r = 0;
      Pic** a = l ? st_after : st_before;
      Pic** c = l ? st_before : st_after;
      const int na = l ? n_after : n_before, nc = l ? n_before : n_after;
      while (r < nt) {
        for (int i = 0; i < na && r < nt; ++i) temp[r++] = a[i];
        for (int i = 0; i < nc && r < nt; ++i) temp[r++] = c[i];
        for (int i = 0; i < n_lt && r < nt; ++i) temp[r++] = lt_cur[i];
      }
      for (int i = 0; i < sh.nref[l]; ++i) list[l][i] = sh.rplm[l] ? temp[sh.entry[l][i]] : temp[i];
    }
    return true;
  }

  void finish_picture() {  // C.5.2.3
    for (Pic& p : dpb)
      if (p.in_use && p.waiting && &p != cur) ++p.latency;
    cur->waiting = true;
    cur->latency = 0;
    cur->ref = 1;
    cur = nullptr;
    while (waiting() > S->max_num_reorder || latency_hit())
      if (!bump()) break;
    release_unused();
  }

  // ----------------------------------------------------------------------------------- picture-level state
  bool open_picture() {
    w4 = S->width / 4;
    pic_w = S->width; pic_h = S->height; pic_lc = S->log2_ctb; pic_wctb = S->wctb; pic_zmask = (1 << (S->log2_ctb - 2)) - 1;
    h4 = S->height / 4;
    const size_t n = static_cast<size_t>(w4) * h4;
    // Only what is OR-ed into starts from zero; every other map is read only where this picture has already written
    // it (availability goes by z-order and slice, 6.4.1, not by what a map holds).
    if (m_pred.size() != n) {
      m_pred.assign(n, 0); m_skip.assign(n, 0); m_depth.assign(n, 0); m_mode.assign(n, 1); m_flags.assign(n, 0); m_qp.assign(n, 0);
    }
    m_nz.assign(n, 0); m_edge.assign(n, 0);
    ctb_slice.assign(static_cast<size_t>(S->wctb) * S->hctb, 0xffff);
    slice_addr.clear();
    num_slices = 0;
    next_ctb = 0;
    pic_deblock = false;
    if (!motion_pool.empty()) {
      cur->pu_of = std::move(motion_pool.back());
      motion_pool.pop_back();
    }
    if (!pu_pool.empty()) {
      cur->pu_motion = std::move(pu_pool.back());
      pu_pool.pop_back();
    }
    cur->pu_motion.assign(1, Motion());
    const size_t msize = n;
    if (cur->pu_of.size() != msize) cur->pu_of.assign(msize, 0u);  // (a recycled map holds its last picture: every unit overwrites its part)
    PicSetup setup;
    setup.width = S->width;
    setup.height = S->height;
    setup.crop_left = S->crop_l; setup.crop_right = S->crop_r; setup.crop_top = S->crop_t; setup.crop_bottom = S->crop_b;
    setup.log2_ctb = S->log2_ctb;
    buf = PicBuffers();
    if (!sink->begin_picture(setup, &buf, &error)) return false;
    cur->slot = buf.slot;
    done = PicDone();
    done.slot = buf.slot;
    done.cookie = buf.cookie;
    done.cb_qp_offset = P->cb_off;
    done.cr_qp_offset = P->cr_off;
    done.pcm_bits_luma = S->pcm_bits_y;
    done.pcm_bits_chroma = S->pcm_bits_c;
    const int ctbs = S->wctb * S->hctb;
    for (int c = 0; c < 3; ++c) std::fill(buf.intra_ctb_first[c], buf.intra_ctb_first[c] + ctbs + 1, 0u);
    std::memset(buf.sao, 0, sizeof(SaoCtb) * static_cast<size_t>(ctbs));
    pic_open = true;
    return true;
  }

  void note_ref_slot(int slot) {
    for (int i = 0; i < done.num_ref_slots; ++i)
      if (done.ref_slots[i] == slot) return;
    if (done.num_ref_slots < amd::hevc::kMaxSurfaces) done.ref_slots[done.num_ref_slots++] = slot;
  }

  bool close_picture() {
    done.deblock = pic_deblock;  // (some slice has the deblocking filter on; which edges, how strongly: k_hevc_deblock)
    done.num_slices = num_slices;
    // the per-plane intra lists are complete: the per-block counts become offsets
    {
      const int ctbs = S->wctb * S->hctb;
      for (int c = 0; c < 3; ++c) {
        uint32_t at = 0;
        for (int k = 0; k <= ctbs; ++k) {
          const uint32_t nk = buf.intra_ctb_first[c][k];
          buf.intra_ctb_first[c][k] = at;
          at += nk;
        }
      }
    }
    if (S->scaling_enabled) {
      done.scaling = true;
      std::memcpy(buf.scaling, P->scaling_present ? &P->scaling : &S->scaling, sizeof(Scaling));
    }
    pic_open = false;
    ++decoded;
    if (!sink->end_picture(done, &error)) return false;
    finish_picture();
    return true;
  }

  // ------------------------------------------------------------------------------- availability, 6.4.1 / 6.4.2
  __attribute__((always_inline)) bool zavail(int xc, int yc, int xn, int yn) const {
    if (static_cast<unsigned>(xn) >= static_cast<unsigned>(pic_w) || static_cast<unsigned>(yn) >= static_cast<unsigned>(pic_h)) return false;
    // by coding tree block first (raster order, and the only place a slice can begin), inside one by the small z-map
    const int lc = pic_lc;
    const int cn = (yn >> lc) * pic_wctb + (xn >> lc), cc = (yc >> lc) * pic_wctb + (xc >> lc);
    if (cn == cc) return zmap[((xn >> 2) & pic_zmask) | ((yn >> 2) & pic_zmask) << 4] <= zmap[((xc >> 2) & pic_zmask) | ((yc >> 2) & pic_zmask) << 4];
    if (cn > cc) return false;
    const uint16_t sn = ctb_slice[cn], sc = ctb_slice[cc];
    if (sn == sc) return sn != 0xffff;
    return sn != 0xffff && sc != 0xffff && slice_addr[sn] == slice_addr[sc];
  }

  struct Cu {
    int x, y, log2, depth;
    bool bypass = false, skip = false, intra = false, pcm = false;
    int part = PART_2Nx2N, qp = 0;
    int luma_mode[4] = {1, 1, 1, 1}, chroma_mode = 1;
  };
  struct Pu {
    int x, y, w, h;
    Motion m;
  };

  bool pb_avail(const Cu& cu, int xp, int yp, int w, int h, int part_idx, int xn, int yn) const {
    const int n = 1 << cu.log2;
    bool ok;
    if (xn >= cu.x && xn < cu.x + n && yn >= cu.y && yn < cu.y + n)
      ok = !((w << 1) == n && (h << 1) == n && part_idx == 1 && cu.y + h <= yn && cu.x + w > xn);
    else
      ok = zavail(xp, yp, xn, yn);
    return ok && m_pred[(yn >> 2) * w4 + (xn >> 2)] == 1;
  }
  const Motion& mot(int x, int y) const { return cur->pu_motion[cur->pu_of[(y >> 2) * w4 + (x >> 2)]]; }

  static void scale_mv(int16_t* mv, int td, int tb) {
    td = clip3(-128, 127, td);
    tb = clip3(-128, 127, tb);
    if (!td) return;
    const int tx = (16384 + (iabs(td) >> 1)) / td;
    const int f = clip3(-4096, 4095, (tb * tx + 32) >> 6);
    for (int c = 0; c < 2; ++c) {
      const int v = f * mv[c];
      mv[c] = static_cast<int16_t>(clip3(-32768, 32767, (v < 0 ? -1 : 1) * ((iabs(v) + 127) >> 8)));
    }
  }

  // 8.5.3.2.8
  bool temporal(int xp, int yp, int w, int h, int ref_idx, int X, int16_t* out) {
    if (!sh.tmvp) return false;
    const Pic* col = list[(sh.type == SL_B && !sh.col_from_l0) ? 1 : 0][sh.col_ref];
    if (!col || col->pu_of.empty()) return false;
    const Pic* target = list[X][ref_idx];
    const bool cur_long = target->ref == 2;
    for (int attempt = 0; attempt < 2; ++attempt) {
      int xc, yc;
      if (attempt == 0) {
        xc = xp + w;
        yc = yp + h;
        if ((yp >> S->log2_ctb) != (yc >> S->log2_ctb) || yc >= S->height || xc >= S->width) continue;
      } else {
        xc = xp + (w >> 1);
        yc = yp + (h >> 1);
      }
      const Motion& c = col->pu_motion[col->pu_of[(((yc >> 4) << 4) >> 2) * w4 + (((xc >> 4) << 4) >> 2)]];
      if (!c.flags) continue;
      int lc;
      if (!(c.flags & 1)) lc = 1;
      else if (c.flags == 1) lc = 0;
      else {
        bool no_backward = true;
        for (int l = 0; l < (sh.type == SL_B ? 2 : 1); ++l)
          for (int i = 0; i < sh.nref[l]; ++i) no_backward &= list[l][i]->poc <= cur->poc;
        lc = no_backward ? X : (sh.col_from_l0 ? 1 : 0);
      }
      if (cur_long != (col->ref_lt_tab[c.rid[lc]] != 0)) continue;
      out[0] = c.mv[lc][0];
      out[1] = c.mv[lc][1];
      const int cd = col->poc - col->ref_poc_tab[c.rid[lc]], td = cur->poc - target->poc;
      if (!cur_long && cd != td) scale_mv(out, cd, td);
      return true;
    }
    return false;
  }

  static bool same_motion(const Motion& a, const Motion& b) {
    if (a.flags != b.flags) return false;
    for (int l = 0; l < 2; ++l)
      if ((a.flags >> l) & 1)
        if (a.ref[l] != b.ref[l] || a.mv[l][0] != b.mv[l][0] || a.mv[l][1] != b.mv[l][1]) return false;
    return true;
  }

  // 8.5.3.2.2 .. 8.5.3.2.5: candidate `want` of the merging list
  Motion merge_candidate(const Cu& cu, int xp, int yp, int w, int h, int part_idx, int want) {
    const int pml = P->pml;
    int part = cu.part;
    if (pml > 2 && cu.log2 == 3) {
      xp = cu.x;
      yp = cu.y;
      w = h = 8;
      part_idx = 0;
      part = PART_2Nx2N;
    }
    Motion c[6];
    int n = 0;
    auto mer = [&](int xn, int yn) { return (xp >> pml) == (xn >> pml) && (yp >> pml) == (yn >> pml); };
    const int xa1 = xp - 1, ya1 = yp + h - 1, xb1 = xp + w - 1, yb1 = yp - 1, xb0 = xp + w, yb0 = yp - 1, xa0 = xp - 1, ya0 = yp + h, xb2 = xp - 1, yb2 = yp - 1;
    // in list order, each candidate as soon as it is known: most blocks want the first one
    const bool ra1 = !mer(xa1, ya1) && !(part_idx == 1 && (part == PART_Nx2N || part == PART_nLx2N || part == PART_nRx2N)) && pb_avail(cu, xp, yp, w, h, part_idx, xa1, ya1);
    if (ra1) {
      if (want == 0) return mot(xa1, ya1);
      c[n++] = mot(xa1, ya1);
    }
    const bool rb1 = !mer(xb1, yb1) && !(part_idx == 1 && (part == PART_2NxN || part == PART_2NxnU || part == PART_2NxnD)) && pb_avail(cu, xp, yp, w, h, part_idx, xb1, yb1);
    const bool fb1 = rb1 && !(ra1 && same_motion(mot(xa1, ya1), mot(xb1, yb1)));
    if (fb1) {
      if (want == n) return mot(xb1, yb1);
      c[n++] = mot(xb1, yb1);
    }
    const bool rb0 = !mer(xb0, yb0) && pb_avail(cu, xp, yp, w, h, part_idx, xb0, yb0);
    const bool fb0 = rb0 && !(rb1 && same_motion(mot(xb1, yb1), mot(xb0, yb0)));
    if (fb0) {
      if (want == n) return mot(xb0, yb0);
      c[n++] = mot(xb0, yb0);
    }
    const bool ra0 = !mer(xa0, ya0) && pb_avail(cu, xp, yp, w, h, part_idx, xa0, ya0);
    const bool fa0 = ra0 && !(ra1 && same_motion(mot(xa1, ya1), mot(xa0, ya0)));
    if (fa0) {
      if (want == n) return mot(xa0, ya0);
      c[n++] = mot(xa0, ya0);
    }
    if (n < 4) {
      const bool rb2 = !mer(xb2, yb2) && pb_avail(cu, xp, yp, w, h, part_idx, xb2, yb2);
      if (rb2 && !(ra1 && same_motion(mot(xa1, ya1), mot(xb2, yb2))) && !(rb1 && same_motion(mot(xb1, yb1), mot(xb2, yb2)))) c[n++] = mot(xb2, yb2);
    }
    if (n > want) return c[want];
    {
      Motion t = Motion();
      int16_t mv[2];
      if (temporal(xp, yp, w, h, 0, 0, mv)) {
        t.flags |= 1;
        t.mv[0][0] = mv[0];
        t.mv[0][1] = mv[1];
      }
      if (sh.type == SL_B && temporal(xp, yp, w, h, 0, 1, mv)) {
        t.flags |= 2;
        t.mv[1][0] = mv[0];
        t.mv[1][1] = mv[1];
      }
      if (t.flags) c[n++] = t;
    }
    const int maxc = sh.max_merge;
    if (n > maxc) n = maxc;
    if (n > want) return c[want];
    Motion list5[5];
    for (int i = 0; i < n; ++i) list5[i] = c[i];
    if (sh.type == SL_B && n > 1 && n < maxc) {
      static const uint8_t a0[12] = {0, 1, 0, 2, 1, 2, 0, 3, 1, 3, 2, 3}, a1[12] = {1, 0, 2, 0, 2, 1, 3, 0, 3, 1, 3, 2};
      const int orig = n;
      for (int k = 0; k < orig * (orig - 1) && n < maxc; ++k) {
        const Motion& p0 = list5[a0[k]];
        const Motion& p1 = list5[a1[k]];
        if ((p0.flags & 1) && (p1.flags & 2) &&
            (list[0][p0.ref[0]]->poc != list[1][p1.ref[1]]->poc || p0.mv[0][0] != p1.mv[1][0] || p0.mv[0][1] != p1.mv[1][1])) {
          Motion m = Motion();
          m.flags = 3;
          m.ref[0] = p0.ref[0];
          m.ref[1] = p1.ref[1];
          m.mv[0][0] = p0.mv[0][0]; m.mv[0][1] = p0.mv[0][1];
          m.mv[1][0] = p1.mv[1][0]; m.mv[1][1] = p1.mv[1][1];
          list5[n++] = m;
        }
      }
    }
    const int nr = sh.type == SL_P ? sh.nref[0] : std::min(sh.nref[0], sh.nref[1]);
    for (int z = 0; n < maxc; ++z) {
      Motion m = Motion();
      const int r = z < nr ? z : 0;
      m.flags = sh.type == SL_P ? 1 : 3;
      m.ref[0] = static_cast<int8_t>(r);
      m.ref[1] = static_cast<int8_t>(sh.type == SL_P ? -1 : r);
      list5[n++] = m;
    }
    return list5[want < n ? want : 0];
  }

  // 8.5.3.2.6 / .7
  void amvp(const Cu& cu, int xp, int yp, int w, int h, int part_idx, int X, int ref_idx, int16_t out[2][2]) {
    const Pic* target = list[X][ref_idx];
    const bool tl = target->ref == 2;
    const int nx[5] = {xp - 1, xp - 1, xp + w, xp + w - 1, xp - 1}, ny[5] = {yp + h, yp + h - 1, yp - 1, yp - 1, yp - 1};
    bool av[5];
    for (int k = 0; k < 5; ++k) av[k] = pb_avail(cu, xp, yp, w, h, part_idx, nx[k], ny[k]);
    int16_t a[2] = {0, 0}, bb[2] = {0, 0};
    bool ha = false, hb = false;
    const bool scaled = av[0] || av[1];
    auto exact = [&](int k0, int k1, int16_t* mv) {
      for (int k = k0; k < k1; ++k) {
        if (!av[k]) continue;
        const Motion& m = mot(nx[k], ny[k]);
        for (int t = 0; t < 2; ++t) {
          const int l = t ? !X : X;
          if (((m.flags >> l) & 1) && list[l][m.ref[l]]->poc == target->poc) {
            mv[0] = m.mv[l][0];
            mv[1] = m.mv[l][1];
            return true;
          }
        }
      }
      return false;
    };
    auto any = [&](int k0, int k1, int16_t* mv) {
      for (int k = k0; k < k1; ++k) {
        if (!av[k]) continue;
        const Motion& m = mot(nx[k], ny[k]);
        for (int t = 0; t < 2; ++t) {
          const int l = t ? !X : X;
          if (!((m.flags >> l) & 1)) continue;
          const Pic* r = list[l][m.ref[l]];
          if ((r->ref == 2) != tl) continue;
          mv[0] = m.mv[l][0];
          mv[1] = m.mv[l][1];
          if (!tl && r->ref != 2) scale_mv(mv, cur->poc - r->poc, cur->poc - target->poc);
          return true;
        }
      }
      return false;
    };
    ha = exact(0, 2, a) || any(0, 2, a);
    hb = exact(2, 5, bb);
    if (!scaled && hb) {
      ha = true;
      a[0] = bb[0];
      a[1] = bb[1];
    }
    if (!scaled) hb = any(2, 5, bb);
    int n = 0;
    if (ha) { out[n][0] = a[0]; out[n++][1] = a[1]; }
    if (hb && !(ha && a[0] == bb[0] && a[1] == bb[1])) { out[n][0] = bb[0]; out[n++][1] = bb[1]; }
    if (n < 2) {
      int16_t c[2];
      if (temporal(xp, yp, w, h, ref_idx, X, c)) { out[n][0] = c[0]; out[n++][1] = c[1]; }
    }
    for (; n < 2; ++n) out[n][0] = out[n][1] = 0;
  }

  // ------------------------------------------------------------------------------------------- slice data
  void init_contexts() {
    int t = 0;
    if (sh.type == SL_P) t = sh.cabac_init ? 2 : 1;
    if (sh.type == SL_B) t = sh.cabac_init ? 1 : 2;
    const int qp = clip3(0, 51, sh.qp);
    for (int i = 0; i < HC_CTX_COUNT; ++i) {
      const int v = hevc_cabac_init[t][i];
      const int pre = clip3(1, 126, ((((v >> 4) * 5 - 45) * qp) >> 4) + ((v & 15) << 3) - 16);
      const int mps = pre > 63;
      ctx[i] = static_cast<uint16_t>((mps ? pre - 64 : 63 - pre) << 1 | mps);
    }
  }
  int bin(int c) { return cabac.decision(&ctx[c]); }

  int qp_pred() const {
    int a = qp_prev, b = qp_prev;
    if (qg_x > 0 && ((qg_x - 1) >> S->log2_ctb) == (qg_x >> S->log2_ctb)) a = m_qp[(qg_y >> 2) * w4 + ((qg_x - 1) >> 2)];
    if (qg_y > 0 && ((qg_y - 1) >> S->log2_ctb) == (qg_y >> S->log2_ctb)) b = m_qp[((qg_y - 1) >> 2) * w4 + (qg_x >> 2)];
    return (a + b + 1) >> 1;
  }
  static int chroma_qp(int qpy, int off) {
    const int q = clip3(0, 57, qpy + off);
    return q < 30 ? q : q >= 43 ? q - 6 : hevc_qpc_table[q - 30];
  }

  // n4 bytes of a row of a coding unit: 2, 4, 8 or 16 (a fixed size is a store or two, not a call)
  static void fill_run(uint8_t* p, uint8_t v, int n4) {
    switch (n4) {
      case 2: std::memset(p, v, 2); break;
      case 4: std::memset(p, v, 4); break;
      case 8: std::memset(p, v, 8); break;
      case 16: std::memset(p, v, 16); break;
      default: std::memset(p, v, static_cast<size_t>(n4)); break;
    }
  }
  void mark_cu(const Cu& cu) {
    const int n = 1 << cu.log2;
    const uint8_t fl = static_cast<uint8_t>((cu.pcm ? 1 : 0) | (cu.bypass ? 2 : 0) | ((cu.bypass || (cu.pcm && S->pcm_lf_off)) ? 4 : 0));
    const int n4 = n >> 2;
    for (int y = cu.y; y < cu.y + n; y += 4) {
      const size_t i = static_cast<size_t>(y >> 2) * w4 + (cu.x >> 2);
      fill_run(&m_pred[i], cu.intra ? 2 : 1, n4);
      fill_run(&m_skip[i], cu.skip, n4);
      fill_run(&m_depth[i], static_cast<uint8_t>(cu.depth), n4);
      fill_run(&m_flags[i], fl, n4);
      fill_run(reinterpret_cast<uint8_t*>(&m_qp[i]), static_cast<uint8_t>(static_cast<int8_t>(cu.qp)), n4);
      if (cu.intra) fill_run(&m_mode[i], 1, n4);
    }
    // "intra" as far as later pictures' temporal candidates and this picture's strengths go
    if (cu.intra)
      for (int y = cu.y; y < cu.y + n; y += 4) std::memset(&cur->pu_of[static_cast<size_t>(y >> 2) * w4 + (cu.x >> 2)], 0, sizeof(uint32_t) * n4);
  }

  void mpm(int x, int y, int cand[3]) const {  // 8.4.2
    int a = 1, b = 1;
    if (zavail(x, y, x - 1, y)) {
      const int i = (y >> 2) * w4 + ((x - 1) >> 2);
      if (m_pred[i] == 2 && !(m_flags[i] & 1)) a = m_mode[i];
    }
    if (zavail(x, y, x, y - 1) && y - 1 >= ((y >> S->log2_ctb) << S->log2_ctb)) {
      const int i = ((y - 1) >> 2) * w4 + (x >> 2);
      if (m_pred[i] == 2 && !(m_flags[i] & 1)) b = m_mode[i];
    }
    if (a == b) {
      if (a < 2) { cand[0] = 0; cand[1] = 1; cand[2] = 26; }
      else { cand[0] = a; cand[1] = 2 + ((a + 29) % 32); cand[2] = 2 + ((a - 2 + 1) % 32); }
    } else {
      cand[0] = a;
      cand[1] = b;
      cand[2] = (a != 0 && b != 0) ? 0 : (a != 1 && b != 1) ? 1 : 26;
    }
  }

  // one intra-predicted block of one plane: the record k_hevc_intra works from
  bool emit_intra(int c, int x, int y, int log2, int mode) {
    if (done.num_intra[c] >= buf.cap_intra[c]) return fail("intra block list overflow");
    IntraTu& t = buf.intra[c][done.num_intra[c]++];
    ++buf.intra_ctb_first[c][((y << (c ? 1 : 0)) >> S->log2_ctb) * S->wctb + ((x << (c ? 1 : 0)) >> S->log2_ctb)];  // (a count until close_picture)
    t.x = static_cast<uint16_t>(x);
    t.y = static_cast<uint16_t>(y);
    t.log2 = static_cast<uint8_t>(log2);
    t.mode = static_cast<uint8_t>(mode);
    t.flags = S->strong ? amd::hevc::INTRA_STRONG : 0;
    t.pad = 0;
    t.avail_left = t.avail_top = t.wait_left = t.wait_top = 0;
    if (mode == amd::hevc::kIntraPcm) return true;
    const int sub = c ? 1 : 0, unit = c ? 2 : 4, n = 1 << log2;
    const int xl = x << sub, yl = y << sub;
    // 6.4.1 by blocks, not by samples: the n samples to the left, the n below them, the n above and the n above-right each
    // lie in one aligned n x n square, and a square is decoded before this block or after it as a whole (z-order; a slice
    // begins with a coding tree block, which is no smaller) - one test per square, the picture's edges cut the rest.
    // Only constrained intra prediction (and which neighbours are waited for) looks at the units one by one.
    const int pw = pic_w >> sub, ph = pic_h >> sub, m = 1 << sub, nu = n / unit;
    auto square = [&](int xn, int yn) { return zavail(xl, yl, xn * m, yn * m); };  // (xn, yn may be -1)
    auto intra_written = [&](int xn, int yn) { return m_pred[((yn << sub) >> 2) * w4 + ((xn << sub) >> 2)] == 2; };
    auto ok_unit = [&](int xn, int yn) { return !P->cip || intra_written(xn, yn); };
    const bool all_intra = sh.type == SL_I;  // every neighbour there is is an intra block of this picture: waited for
    if (square(x - 1, y - 1) && ok_unit(x - 1, y - 1))
      t.flags |= amd::hevc::INTRA_CORNER | ((all_intra || intra_written(x - 1, y - 1)) ? amd::hevc::INTRA_WAIT_CORNER : 0);
    const bool sq_left = x > 0 && square(x - 1, y), sq_below = x > 0 && y + n < ph && square(x - 1, y + n);
    const bool sq_top = y > 0 && square(x, y - 1), sq_right = y > 0 && x + n < pw && square(x + n, y - 1);
    for (int half = 0; half < 2; ++half) {
      if (half ? sq_below : sq_left)
        for (int k = half * nu; k < (half + 1) * nu && y + k * unit < ph; ++k)
          if (ok_unit(x - 1, y + k * unit)) {
            t.avail_left |= 1u << k;
            if (all_intra || intra_written(x - 1, y + k * unit)) t.wait_left |= 1u << k;
          }
      if (half ? sq_right : sq_top)
        for (int k = half * nu; k < (half + 1) * nu && x + k * unit < pw; ++k)
          if (ok_unit(x + k * unit, y - 1)) {
            t.avail_top |= 1u << k;
            if (all_intra || intra_written(x + k * unit, y - 1)) t.wait_top |= 1u << k;
          }
    }
    return true;
  }

  // ---- residual_coding (7.3.8.11) straight into the level stream
  bool residual(const Cu& cu, int c, int x, int y, int log2, int mode) {
    const int n = 1 << log2;
    bool tskip = false;
    if (P->tskip && !cu.bypass && log2 == 2) tskip = bin(HC_TRANSFORM_SKIP + (c ? 1 : 0)) != 0;
    int scan = 0;
    if (cu.intra && (log2 == 2 || (log2 == 3 && c == 0))) {
      if (mode >= 6 && mode <= 14) scan = 2;
      if (mode >= 22 && mode <= 30) scan = 1;
    }
    static const Scans& sc_tab = scans();
    static const SigTables& sg_tab = sig_tables();
    const uint8_t* sbs = sc_tab.t[log2 - 2][scan];
    const uint8_t* pss = sc_tab.t[2][scan];
    // last significant position
    int off, shift;
    if (c == 0) { off = 3 * (log2 - 2) + ((log2 - 1) >> 2); shift = (log2 + 1) >> 2; }
    else { off = 15; shift = log2 - 2; }
    const int cmax = (log2 << 1) - 1;
    int px = 0, py = 0;
    while (px < cmax && bin(HC_LAST_X + off + (px >> shift))) ++px;
    while (py < cmax && bin(HC_LAST_Y + off + (py >> shift))) ++py;
    int lx = px, ly = py;
    if (px > 3) { const int nb = (px >> 1) - 1; lx = (1 << nb) * (2 + (px & 1)) + static_cast<int>(cabac.bypass_bits(nb)); }
    if (py > 3) { const int nb = (py >> 1) - 1; ly = (1 << nb) * (2 + (py & 1)) + static_cast<int>(cabac.bypass_bits(nb)); }
    if (scan == 2) std::swap(lx, ly);
    if (lx >= n || ly >= n) return fail("last significant coefficient outside its block");
    // where the last significant position sits in the scan: sub-block, then position within it
    const int last_sb = sc_tab.inv[log2 - 2][scan][(lx >> 2) | (ly >> 2) << (log2 - 2)];
    const int last_pos = sc_tab.inv[2][scan][(lx & 3) | (ly & 3) << 2];
    if (done.num_res >= buf.cap_res) return fail("transform block list overflow");
    ResTu& r = buf.res[done.num_res];
    r.x = static_cast<uint16_t>(x);
    r.y = static_cast<uint16_t>(y);
    r.log2 = static_cast<uint8_t>(log2);
    r.c_idx = static_cast<uint8_t>(c);
    const SliceLoop& sl = buf.slices[slice_idx];
    (void)sl;
    r.qp = static_cast<uint8_t>(c == 0 ? cu.qp : chroma_qp(cu.qp, c == 1 ? P->cb_off + sh.cb_off : P->cr_off + sh.cr_off));
    r.flags = static_cast<uint8_t>((cu.intra && c == 0 && log2 == 2 ? amd::hevc::RES_DST : 0) | (tskip ? amd::hevc::RES_TSKIP : 0) | (cu.bypass ? amd::hevc::RES_BYPASS : 0));
    r.matrix = 0xff;
    if (S->scaling_enabled && !(tskip && log2 > 2)) r.matrix = static_cast<uint8_t>(log2 == 5 ? (cu.intra ? 0 : 1) : (cu.intra ? 0 : 3) + c);
    r.pad[0] = r.pad[1] = r.pad[2] = 0;
    r.coef_offset = static_cast<uint32_t>(done.num_coefs);
    uint8_t csbf[8][8];
    std::memset(csbf, 0, sizeof(csbf));
    const int sbw = 1 << (log2 - 2);
    int c1 = 1;
    bool first_sb = true;
    for (int i = last_sb; i >= 0; --i) {
      const int xs = sbs[i] & 15, ys = sbs[i] >> 4;
      const int right = xs + 1 < sbw ? csbf[ys][xs + 1] : 0, below = ys + 1 < sbw ? csbf[ys + 1][xs] : 0;
      bool infer_dc = false, coded = true;
      if (i < last_sb && i > 0) {
        coded = bin(HC_CODED_SUB_BLOCK + std::min(right | below, 1) + (c ? 2 : 0)) != 0;
        infer_dc = true;
      }
      csbf[ys][xs] = coded;
      uint16_t sig = 0;
      if (i == last_sb) sig |= 1u << last_pos;
      if (coded) {
        const uint8_t* pat;
        int base;
        if (log2 == 2) {
          pat = sg_tab.map4[scan];
          base = HC_SIG_COEFF + (c ? 27 : 0);
        } else {
          pat = sg_tab.pat[scan][right | below << 1];
          base = HC_SIG_COEFF + (c ? 27 + (log2 == 3 ? 9 : 12) : ((xs || ys) ? 3 : 0) + (log2 == 3 ? (scan == 0 ? 9 : 15) : 21));
        }
        const bool dc_block = log2 != 2 && !xs && !ys;  // the block's DC coefficient has a context of its own
        int k = i == last_sb ? last_pos - 1 : 15;
        for (; k > 0; --k)
          if (bin(base + pat[k])) {
            sig |= 1u << k;
            infer_dc = false;
          }
        if (k == 0) {
          if (infer_dc) sig |= 1;
          else if (bin(dc_block ? HC_SIG_COEFF + (c ? 27 : 0) : base + pat[0])) sig |= 1;
        }
      }
      if (!sig) continue;
      int ctx_set = (i == 0 || c > 0) ? 0 : 2;
      if (!first_sb && c1 == 0) ++ctx_set;
      first_sb = false;
      c1 = 1;
      uint16_t g1 = 0, g2 = 0;
      const int first_sig = __builtin_ctz(sig), last_sig = 31 - __builtin_clz(static_cast<uint32_t>(sig));
      int g1pos = -1;
      {
        uint32_t m = sig;
        for (int ng1 = 0; m && ng1 < 8; ++ng1) {  // (the first eight levels of a sub-block only)
          const int k = 31 - __builtin_clz(m);
          m &= ~(1u << k);
          if (bin(HC_GREATER1 + ctx_set * 4 + c1 + (c ? 16 : 0))) {
            g1 |= 1u << k;
            c1 = 0;
            if (g1pos < 0) g1pos = k;
          } else if (c1 > 0 && c1 < 3) {
            ++c1;
          }
        }
      }
      const bool hidden = !cu.bypass && P->sign_hiding && last_sig - first_sig > 3;
      if (g1pos >= 0 && bin(HC_GREATER2 + ctx_set + (c ? 4 : 0))) g2 |= 1u << g1pos;
      uint16_t neg = 0;
      {  // coeff_sign_flag of every level whose sign is coded, highest scan position first: one read
        uint16_t coded = sig;
        if (hidden) coded &= static_cast<uint16_t>(~(1u << first_sig));
        const int ns = __builtin_popcount(coded);
        unsigned bits = cabac.bypass_bits(ns);
        int left = ns;
        for (uint32_t m = coded; m;) {
          const int k = 31 - __builtin_clz(m);
          m &= ~(1u << k);
          --left;
          if ((bits >> left) & 1) neg |= 1u << k;
        }
      }
      int nsig = 0, sum = 0, rice = 0;
      for (uint32_t m = sig; m;) {
        const int k = 31 - __builtin_clz(m);
        m &= ~(1u << k);
        const int base = 1 + ((g1 >> k) & 1) + ((g2 >> k) & 1);
        int a = base;
        if (base == (nsig < 8 ? (k == g1pos ? 3 : 2) : 1)) {
          int pre = 0;
          while (pre < 4 && cabac.bypass()) ++pre;
          unsigned rem;
          if (pre < 4) {
            rem = (static_cast<unsigned>(pre) << rice) + cabac.bypass_bits(rice);
          } else {
            int kk = rice + 1;
            unsigned e = 0;
            while (cabac.bypass() && kk < 31) e += 1u << kk++;
            e += cabac.bypass_bits(kk);
            rem = (4u << rice) + e;
          }
          if (rem > 40000) return fail("coefficient level out of range");
          a = base + static_cast<int>(rem);
          if (a > 3 * (1 << rice)) rice = std::min(rice + 1, 4);
        }
        int v = ((neg >> k) & 1) ? -a : a;
        if (hidden) {
          sum += a;
          if (k == first_sig && (sum & 1)) v = -v;
        }
        v = clip3(-32768, 32767, v);
        if (done.num_coefs >= buf.cap_coefs) return fail("level stream overflow");
        const int pos = ((ys << 2) + (pss[k] >> 4)) * n + (xs << 2) + (pss[k] & 15);
        buf.coefs[done.num_coefs++] = static_cast<uint32_t>(pos) << 16 | (static_cast<uint32_t>(v) & 0xffffu);
        ++nsig;
      }
    }
    r.ncoef = static_cast<uint32_t>(done.num_coefs) - r.coef_offset;
    ++done.num_res;
    return !cabac.overrun();
  }

  struct TreeCtx {
    Cu* cu;
    int max_depth;
  };

  bool transform_tree(TreeCtx& t, int x0, int y0, int xb, int yb, int log2, int depth, int blk, bool pcb, bool pcr) {
    Cu& cu = *t.cu;
    const bool isplit = cu.intra && cu.part == PART_NxN;
    const bool inter_split = S->th_inter == 0 && !cu.intra && cu.part != PART_2Nx2N && depth == 0;
    bool split;
    if (log2 <= S->log2_max_tb && log2 > S->log2_min_tb && depth < t.max_depth && !(isplit && depth == 0))
      split = bin(HC_SPLIT_TRANSFORM + 5 - log2) != 0;
    else
      split = log2 > S->log2_max_tb || (isplit && depth == 0) || inter_split;
    bool cb = false, cr = false;
    if (log2 > 2) {
      if (depth == 0 || pcb) cb = bin(HC_CBF_CHROMA + depth) != 0;
      if (depth == 0 || pcr) cr = bin(HC_CBF_CHROMA + depth) != 0;
    } else if (depth > 0) {
      cb = pcb;
      cr = pcr;
    }
    if (split) {
      const int h = 1 << (log2 - 1);
      for (int i = 0; i < 4; ++i)
        if (!transform_tree(t, x0 + (i & 1) * h, y0 + (i >> 1) * h, x0, y0, log2 - 1, depth + 1, i, cb, cr)) return false;
      return true;
    }
    bool luma = true;
    if (cu.intra || depth != 0 || cb || cr) luma = bin(HC_CBF_LUMA + (depth == 0 ? 1 : 0)) != 0;
    const bool chroma_here = log2 > 2 || blk == 3;
    const int half = (1 << cu.log2) >> 1;
    const int pi = isplit ? ((y0 - cu.y >= half) << 1 | (x0 - cu.x >= half)) : 0;
    // transform block edges and the "has levels" map for the deblocking filter
    const int tn = 1 << log2;
    for (int k = 0; k < tn; k += 4) {
      if (x0 > cu.x) m_edge[((y0 + k) >> 2) * w4 + (x0 >> 2)] |= 1 | edge_on_v;
      if (y0 > cu.y) m_edge[(y0 >> 2) * w4 + ((x0 + k) >> 2)] |= 4 | edge_on_h;
    }
    if (luma)
      for (int j = 0; j < tn; j += 4)
        for (int i = 0; i < tn; i += 4) m_nz[((y0 + j) >> 2) * w4 + ((x0 + i) >> 2)] = 1;
    if (luma || cb || cr) {
      if (P->qp_delta && !qp_delta_coded) {  // cu_qp_delta_abs, cu_qp_delta_sign_flag
        int pre = 0;
        while (pre < 5 && bin(HC_QP_DELTA + (pre ? 1 : 0))) ++pre;
        int v = pre;
        if (pre == 5) {
          int k = 0;
          unsigned e = 0;
          while (cabac.bypass() && k < 31) e += 1u << k++;
          e += cabac.bypass_bits(k);
          v = 5 + static_cast<int>(e);
        }
        if (v && cabac.bypass()) v = -v;
        qp_delta_coded = true;
        qp_delta_val = v;
        cu.qp = (qp_pred() + v + 52) % 52;
        if (v < -26 || v > 25) return fail("CuQpDeltaVal %d", v);
      }
    }
    // intra: the prediction of this block is a record whether or not it has levels
    if (cu.intra) {
      if (!emit_intra(0, x0, y0, log2, cu.luma_mode[pi])) return false;
    }
    if (luma && !residual(cu, 0, x0, y0, log2, cu.luma_mode[pi])) return false;
    if (chroma_here) {
      const int cx = (log2 > 2 ? x0 : xb) >> 1, cy = (log2 > 2 ? y0 : yb) >> 1, cl = log2 > 2 ? log2 - 1 : 2;
      for (int c = 1; c < 3; ++c) {
        if (cu.intra && !emit_intra(c, cx, cy, cl, cu.chroma_mode)) return false;
        if ((c == 1 ? cb : cr) && !residual(cu, c, cx, cy, cl, cu.chroma_mode)) return false;
      }
    }
    return true;
  }

  uint8_t edge_on_v = 0, edge_on_h = 0;  // filterEdgeFlag bits of internal edges of the unit being parsed

  void read_mvd(int16_t* mvd) {
    const int g0x = bin(HC_MVD_G0), g0y = bin(HC_MVD_G0);
    int g1x = 0, g1y = 0;
    if (g0x) g1x = bin(HC_MVD_G1);
    if (g0y) g1y = bin(HC_MVD_G1);
    auto rest = [&](int g0, int g1) {
      if (!g0) return 0;
      int a = 1;
      if (g1) {
        int k = 1;
        unsigned e = 0;
        while (cabac.bypass() && k < 31) e += 1u << k++;
        e += cabac.bypass_bits(k);
        a = 2 + static_cast<int>(e);
      }
      return cabac.bypass() ? -a : a;
    };
    mvd[0] = static_cast<int16_t>(rest(g0x, g1x));
    mvd[1] = static_cast<int16_t>(rest(g0y, g1y));
  }

  static void pu_geometry(const Cu& cu, int i, int* x, int* y, int* w, int* h) {
    const int n = 1 << cu.log2, q = n >> 2;
    *x = cu.x; *y = cu.y; *w = n; *h = n;
    switch (cu.part) {
      case PART_2NxN: *h = n / 2; *y += i * (n / 2); break;
      case PART_Nx2N: *w = n / 2; *x += i * (n / 2); break;
      case PART_NxN: *w = *h = n / 2; *x += (i & 1) * (n / 2); *y += (i >> 1) * (n / 2); break;
      case PART_2NxnU: *h = i ? n - q : q; *y += i ? q : 0; break;
      case PART_2NxnD: *h = i ? q : n - q; *y += i ? n - q : 0; break;
      case PART_nLx2N: *w = i ? n - q : q; *x += i ? q : 0; break;
      case PART_nRx2N: *w = i ? q : n - q; *x += i ? n - q : 0; break;
      default: break;
    }
  }

  bool prediction_unit(Cu& cu, int part_idx) {
    int x, y, w, h;
    pu_geometry(cu, part_idx, &x, &y, &w, &h);
    Motion m = Motion();
    bool merge = cu.skip;
    if (!cu.skip) merge = bin(HC_MERGE_FLAG) != 0;
    if (part_idx == 0) first_pu_merge = merge;
    if (merge) {
      int idx = 0;
      if (sh.max_merge > 1 && bin(HC_MERGE_IDX)) {
        idx = 1;
        while (idx < sh.max_merge - 1 && cabac.bypass()) ++idx;
      }
      m = merge_candidate(cu, x, y, w, h, part_idx, idx);
      if (m.flags == 3 && w + h == 12) {
        m.flags = 1;
        m.ref[1] = -1;
      }
    } else {
      int idc = 0;
      if (sh.type == SL_B) {
        if (w + h != 12 && bin(HC_INTER_PRED_IDC + cu.depth)) idc = 2;
        else idc = bin(HC_INTER_PRED_IDC + 4);
      }
      m.flags = static_cast<uint8_t>(idc == 2 ? 3 : idc == 1 ? 2 : 1);
      for (int l = 0; l < 2; ++l) {
        m.ref[l] = -1;
        if (!((m.flags >> l) & 1)) continue;
        int r = 0;
        const int cmax = sh.nref[l] - 1;
        while (r < cmax) {
          if (!(r < 2 ? bin(HC_REF_IDX + r) : cabac.bypass())) break;
          ++r;
        }
        m.ref[l] = static_cast<int8_t>(r);
        int16_t mvd[2] = {0, 0};
        if (!(l == 1 && sh.mvd_l1_zero && idc == 2)) read_mvd(mvd);
        const int flag = bin(HC_MVP_FLAG);
        int16_t cand[2][2];
        amvp(cu, x, y, w, h, part_idx, l, r, cand);
        m.mv[l][0] = static_cast<int16_t>(cand[flag][0] + mvd[0]);
        m.mv[l][1] = static_cast<int16_t>(cand[flag][1] + mvd[1]);
      }
    }
    // store, and emit the prediction record
    if (done.num_pus >= buf.cap_pus) return fail("prediction block list overflow");
    InterPu& p = buf.pus[done.num_pus++];
    p.x = static_cast<uint16_t>(x);
    p.y = static_cast<uint16_t>(y);
    p.w = static_cast<uint8_t>(w);
    p.h = static_cast<uint8_t>(h);
    p.pred_flag = m.flags;
    p.weighted = (sh.type == SL_P ? P->wp : P->wbp) ? 1 : 0;
    p.pad = 0;
    for (int l = 0; l < 2; ++l) {
      p.ref_slot[l] = 0;
      p.mv[l][0] = p.mv[l][1] = 0;
      p.wp[l] = 0;
      if ((m.flags >> l) & 1) {
        if (m.ref[l] < 0 || m.ref[l] >= sh.nref[l]) return fail("reference index %d of %d", m.ref[l], sh.nref[l]);
        const Pic* r = list[l][m.ref[l]];
        m.rid[l] = static_cast<uint8_t>(r - dpb);
        cur->ref_poc_tab[m.rid[l]] = r->poc;
        cur->ref_lt_tab[m.rid[l]] = r->ref == 2;
        if (r->slot < 0) {
          if (!r->made_up) return fail("prediction from a picture that was not decoded (POC %d)", r->poc);
          std::string why;
          const int made = sink->missing_surface(&why);
          if (made < 0) return fail("prediction from a picture that was never in the stream (POC %d): %s", r->poc, why.c_str());
          dpb[m.rid[l]].slot = made;
        }
        p.ref_slot[l] = static_cast<uint8_t>(r->slot);
        note_ref_slot(r->slot);
        p.mv[l][0] = m.mv[l][0];
        p.mv[l][1] = m.mv[l][1];
        p.wp[l] = static_cast<uint16_t>(weights_base[l] + m.ref[l]);
      } else {
        m.ref[l] = -1;
        m.rid[l] = 0;
        m.mv[l][0] = m.mv[l][1] = 0;
      }
    }
    {
      const uint32_t pu_index = static_cast<uint32_t>(cur->pu_motion.size());
      cur->pu_motion.push_back(m);
      for (int yy = y; yy < y + h; yy += 4) {
        uint32_t* run = &cur->pu_of[static_cast<size_t>(yy >> 2) * w4 + (x >> 2)];
        for (int k = 0; k < (w >> 2); ++k) run[k] = pu_index;
      }
    }
    // a prediction block edge inside the unit
    if (x > cu.x)
      for (int k = 0; k < h; k += 4) m_edge[((y + k) >> 2) * w4 + (x >> 2)] |= 2 | edge_on_v;
    if (y > cu.y)
      for (int k = 0; k < w; k += 4) m_edge[(y >> 2) * w4 + ((x + k) >> 2)] |= 8 | edge_on_h;
    return true;
  }
  bool first_pu_merge = false;

  bool coding_unit(int x0, int y0, int log2, int depth) {
    Cu cu;
    cu.x = x0; cu.y = y0; cu.log2 = log2; cu.depth = depth;
    const int n = 1 << log2;
    if (P->bypass) cu.bypass = bin(HC_TQ_BYPASS) != 0;
    if (sh.type != SL_I) {
      const int l = zavail(x0, y0, x0 - 1, y0) && m_skip[(y0 >> 2) * w4 + ((x0 - 1) >> 2)];
      const int a = zavail(x0, y0, x0, y0 - 1) && m_skip[((y0 - 1) >> 2) * w4 + (x0 >> 2)];
      cu.skip = bin(HC_CU_SKIP + l + a) != 0;
    }
    cu.qp = P->qp_delta ? (qp_pred() + qp_delta_val + 52) % 52 : sh.qp;
    // the unit's outer edges: filterEdgeFlag (8.7.2.3) and the strengths come later, the flags now
    const SliceLoop& sl = buf.slices[slice_idx];
    {
      bool left = x0 > 0, top = y0 > 0;
      if (left && !sl.across && !zavail(x0, y0, x0 - 1, y0)) left = false;
      if (top && !sl.across && !zavail(x0, y0, x0, y0 - 1)) top = false;
      const bool on = !sh.db_off;
      pic_deblock = pic_deblock || on;
      edge_on_v = on ? 16 : 0;
      edge_on_h = on ? 32 : 0;
      for (int k = 0; k < n; k += 4) {
        m_edge[((y0 + k) >> 2) * w4 + (x0 >> 2)] |= static_cast<uint8_t>(1 | 2 | (on && left ? 16 : 0));
        m_edge[(y0 >> 2) * w4 + ((x0 + k) >> 2)] |= static_cast<uint8_t>(4 | 8 | (on && top ? 32 : 0));
      }
    }
    if (cu.skip) {
      cu.intra = false;
      mark_cu(cu);
      if (!prediction_unit(cu, 0)) return false;
    } else {
      cu.intra = sh.type == SL_I ? true : bin(HC_PRED_MODE) != 0;
      if (!cu.intra || log2 == S->log2_min_cb) {
        // part_mode, 9.3.3.7
        if (cu.intra) {
          cu.part = bin(HC_PART_MODE) ? PART_2Nx2N : PART_NxN;
        } else if (bin(HC_PART_MODE)) {
          cu.part = PART_2Nx2N;
        } else if (log2 == S->log2_min_cb) {
          if (log2 == 3) cu.part = bin(HC_PART_MODE + 1) ? PART_2NxN : PART_Nx2N;
          else if (bin(HC_PART_MODE + 1)) cu.part = PART_2NxN;
          else cu.part = bin(HC_PART_MODE + 2) ? PART_Nx2N : PART_NxN;
        } else if (!S->amp) {
          cu.part = bin(HC_PART_MODE + 1) ? PART_2NxN : PART_Nx2N;
        } else {
          const bool hor = bin(HC_PART_MODE + 1) != 0;
          if (bin(HC_PART_MODE + 3)) cu.part = hor ? PART_2NxN : PART_Nx2N;
          else {
            const bool second = cabac.bypass() != 0;
            cu.part = hor ? (second ? PART_2NxnD : PART_2NxnU) : (second ? PART_nRx2N : PART_nLx2N);
          }
        }
        if (cu.intra && cu.part == PART_NxN && log2 <= S->log2_min_tb) return fail("intra NxN in a coding block of the minimum transform size");
      }
      if (cu.intra) {
        if (cu.part == PART_2Nx2N && S->pcm && log2 >= S->log2_min_pcm && log2 <= S->log2_max_pcm) cu.pcm = cabac.terminate() != 0;
        mark_cu(cu);
        if (cu.pcm) {
          if (!pcm_samples(cu)) return false;
        } else {
          const int parts = cu.part == PART_NxN ? 4 : 1, pb = n >> (parts == 4);
          bool prev[4];
          for (int i = 0; i < parts; ++i) prev[i] = bin(HC_PREV_INTRA_LUMA) != 0;
          for (int i = 0; i < parts; ++i) {
            const int x = x0 + (i & 1) * pb, y = y0 + (i >> 1) * pb;
            int cand[3], mode;
            mpm(x, y, cand);
            if (prev[i]) {
              int k = 0;
              while (k < 2 && cabac.bypass()) ++k;
              mode = cand[k];
            } else {
              mode = static_cast<int>(cabac.bypass_bits(5));
              std::sort(cand, cand + 3);
              for (int k = 0; k < 3; ++k) mode += mode >= cand[k];
            }
            cu.luma_mode[i] = mode;
            for (int yy = 0; yy < pb; yy += 4)
              for (int xx = 0; xx < pb; xx += 4) m_mode[((y + yy) >> 2) * w4 + ((x + xx) >> 2)] = static_cast<uint8_t>(mode);
          }
          int v = 4;
          if (bin(HC_INTRA_CHROMA)) v = static_cast<int>(cabac.bypass_bits(2));
          static const uint8_t tab[4] = {0, 26, 10, 1};
          cu.chroma_mode = v == 4 ? cu.luma_mode[0] : (tab[v] == cu.luma_mode[0] ? 34 : tab[v]);
        }
      } else {
        mark_cu(cu);
        const int parts = cu.part == PART_2Nx2N ? 1 : cu.part == PART_NxN ? 4 : 2;
        if (cu.part == PART_NxN && log2 == 3) return fail("inter NxN in an 8x8 coding block");
        for (int i = 0; i < parts; ++i)
          if (!prediction_unit(cu, i)) return false;
      }
      if (!cu.pcm) {
        bool root = true;
        if (!cu.intra && !(cu.part == PART_2Nx2N && first_pu_merge)) root = bin(HC_RQT_ROOT_CBF) != 0;
        if (root) {
          TreeCtx t{&cu, cu.intra ? S->th_intra + (cu.part == PART_NxN) : S->th_inter};
          if (!transform_tree(t, x0, y0, x0, y0, log2, 0, 0, false, false)) return false;
        }
      }
    }
    {
      // the unit is complete: what the in-loop filters need to know about each of its 4x4 blocks - which sides are block
      // edges that may be filtered (m_edge, collected while the unit was parsed), levels, intra, QpY.  The boundary
      // strengths themselves are the deblocking kernel's business (8.7.2.4 needs nothing but these and the motion).
      // (a record is a word - edge, QpY, flags, slice from the low byte up - and a unit's row of them a short loop the
      // compiler can widen: 130 000 records a 1080p picture)
      static_assert(sizeof(EdgeInfo) == 4 && offsetof(EdgeInfo, edge) == 0 && offsetof(EdgeInfo, qp) == 1 && offsetof(EdgeInfo, flags) == 2 && offsetof(EdgeInfo, slice) == 3, "EdgeInfo is written as a word");
      const uint32_t upper = (static_cast<uint32_t>(static_cast<uint8_t>(cu.qp)) << 8) | ((cu.bypass || (cu.pcm && S->pcm_lf_off)) ? 1u << 16 : 0u) |
                             (static_cast<uint32_t>(static_cast<uint8_t>(slice_idx)) << 24) | (cu.intra ? amd::hevc::EDGE_INTRA : 0u);
      const int n4 = n >> 2;
      for (int y = y0; y < y0 + n; y += 4) {
        const size_t i0 = static_cast<size_t>(y >> 2) * w4 + (x0 >> 2);
        const uint8_t* __restrict me = &m_edge[i0];
        const uint8_t* __restrict mn = &m_nz[i0];
        uint32_t* __restrict out = reinterpret_cast<uint32_t*>(buf.edges + i0);
        for (int k = 0; k < n4; ++k) out[k] = upper | me[k] | (mn[k] ? static_cast<uint32_t>(amd::hevc::EDGE_LEVELS) : 0u);
        fill_run(reinterpret_cast<uint8_t*>(&m_qp[i0]), static_cast<uint8_t>(cu.qp), n4);
      }
    }
    last_qp = cu.qp;
    return !cabac.overrun();
  }

  // pcm_sample(): the samples become "levels" of bypassed transform blocks, the block an intra record that adds
  // them to nothing
  bool pcm_samples(const Cu& cu) {
    const int n = 1 << cu.log2;
    size_t at = cabac.aligned_end();
    const uint8_t* p = cabac.raw();
    const size_t size = cabac.size();
    size_t bitpos = at * 8;
    auto get = [&](int bits) {
      int v = 0;
      for (int i = 0; i < bits; ++i, ++bitpos) v = v << 1 | (((bitpos >> 3) < size ? p[bitpos >> 3] >> (7 - (bitpos & 7)) : 0) & 1);
      return v;
    };
    for (int c = 0; c < 3; ++c) {
      const int sub = c ? 1 : 0, m = n >> sub, bits = c ? S->pcm_bits_c : S->pcm_bits_y;
      const int log2 = cu.log2 - sub;
      if (done.num_res >= buf.cap_res) return fail("transform block list overflow");
      ResTu& r = buf.res[done.num_res++];
      r.x = static_cast<uint16_t>(cu.x >> sub);
      r.y = static_cast<uint16_t>(cu.y >> sub);
      r.log2 = static_cast<uint8_t>(log2);
      r.c_idx = static_cast<uint8_t>(c);
      r.qp = 0;
      r.flags = amd::hevc::RES_BYPASS;
      r.matrix = 0xff;
      r.pad[0] = r.pad[1] = r.pad[2] = 0;
      r.coef_offset = static_cast<uint32_t>(done.num_coefs);
      for (int i = 0; i < m * m; ++i) {
        const int v = get(bits) << (8 - bits);
        if (!v) continue;
        if (done.num_coefs >= buf.cap_coefs) return fail("level stream overflow");
        buf.coefs[done.num_coefs++] = static_cast<uint32_t>(i) << 16 | static_cast<uint32_t>(v);
      }
      r.ncoef = static_cast<uint32_t>(done.num_coefs) - r.coef_offset;
      if (!emit_intra(c, cu.x >> sub, cu.y >> sub, log2, amd::hevc::kIntraPcm)) return false;
    }
    if ((bitpos + 7) / 8 > size) return fail("pcm_sample() runs past the slice data");
    cabac.restart((bitpos + 7) >> 3);  // 9.3.2.5
    return true;
  }

  bool coding_quadtree(int x0, int y0, int log2, int depth) {
    const int n = 1 << log2;
    bool split;
    if (x0 + n <= S->width && y0 + n <= S->height && log2 > S->log2_min_cb) {
      const int l = zavail(x0, y0, x0 - 1, y0) && m_depth[(y0 >> 2) * w4 + ((x0 - 1) >> 2)] > depth;
      const int a = zavail(x0, y0, x0, y0 - 1) && m_depth[((y0 - 1) >> 2) * w4 + (x0 >> 2)] > depth;
      split = bin(HC_SPLIT_CU + l + a) != 0;
    } else {
      split = log2 > S->log2_min_cb;
    }
    if (P->qp_delta && log2 >= S->log2_ctb - P->qp_delta_depth) {
      qp_delta_coded = false;
      qp_delta_val = 0;
      qg_x = x0;
      qg_y = y0;
      qp_prev = last_qp;
    }
    if (!split) return coding_unit(x0, y0, log2, depth);
    const int h = n >> 1;
    for (int i = 0; i < 4; ++i) {
      const int x = x0 + (i & 1) * h, y = y0 + (i >> 1) * h;
      if (x < S->width && y < S->height && !coding_quadtree(x, y, log2 - 1, depth + 1)) return false;
    }
    return true;
  }

  void sao_syntax(int rx, int ry) {  // 7.3.8.3
    const int addr = ry * S->wctb + rx;
    SaoCtb& p = buf.sao[addr];
    const uint32_t first = slice_addr[slice_idx];
    bool left = false, up = false;
    if (rx > 0 && static_cast<uint32_t>(addr - 1) >= first && ctb_slice[addr - 1] != 0xffff && slice_addr[ctb_slice[addr - 1]] == first) left = bin(HC_SAO_MERGE) != 0;
    if (ry > 0 && !left && static_cast<uint32_t>(addr - S->wctb) >= first && addr - S->wctb >= 0 && ctb_slice[addr - S->wctb] != 0xffff &&
        slice_addr[ctb_slice[addr - S->wctb]] == first)
      up = bin(HC_SAO_MERGE) != 0;
    if (left || up) {
      p = buf.sao[left ? addr - 1 : addr - S->wctb];
      return;
    }
    p = SaoCtb();
    for (int c = 0; c < 3; ++c) {
      if (!(c ? sh.sao_c : sh.sao_y)) continue;
      if (c != 2) {
        p.type[c] = 0;
        if (bin(HC_SAO_TYPE)) p.type[c] = cabac.bypass() ? 2 : 1;
      } else {
        p.type[2] = p.type[1];
      }
      if (!p.type[c]) continue;
      int a[4];
      for (int i = 0; i < 4; ++i) {
        a[i] = 0;
        while (a[i] < 7 && cabac.bypass()) ++a[i];
      }
      if (p.type[c] == 1) {
        for (int i = 0; i < 4; ++i) p.offset[c][i] = static_cast<int8_t>((a[i] && cabac.bypass()) ? -a[i] : a[i]);
        p.band_pos[c] = static_cast<uint8_t>(cabac.bypass_bits(5));
      } else {
        if (c != 2) p.eo_class[c] = static_cast<uint8_t>(cabac.bypass_bits(2));
        else p.eo_class[2] = p.eo_class[1];
        for (int i = 0; i < 4; ++i) p.offset[c][i] = static_cast<int8_t>(i < 2 ? a[i] : -a[i]);
      }
    }
  }

  bool slice_data(const uint8_t* rbsp, size_t size) {
    const int total = S->wctb * S->hctb;
    cabac.start(rbsp, size, sh.data_offset);
    init_contexts();
    last_qp = qp_prev = sh.qp;
    qp_delta_coded = false;
    qp_delta_val = 0;
    int addr = sh.address;
    for (;;) {
      const int rx = addr % S->wctb, ry = addr / S->wctb;
      if (P->wpp && rx == 0 && addr != sh.address) {
        const int ar = addr - S->wctb + 1;
        if (S->wctb > 1 && static_cast<uint32_t>(ar) >= slice_addr[slice_idx] && ctb_slice[ar] != 0xffff) std::memcpy(ctx, ctx_saved, sizeof(ctx));
        else init_contexts();
        last_qp = sh.qp;
      }
      ctb_slice[addr] = static_cast<uint16_t>(slice_idx);
      next_ctb = addr + 1;
      if (sh.sao_y || sh.sao_c) sao_syntax(rx, ry);
      if (!coding_quadtree(rx << S->log2_ctb, ry << S->log2_ctb, S->log2_ctb, 0)) return false;
      if (P->wpp && rx == (S->wctb > 1 ? 1 : 0)) std::memcpy(ctx_saved, ctx, sizeof(ctx));
      if (cabac.overrun()) return fail("slice data ran past the end of the NAL unit at coding tree block %d", addr);
      const bool last = cabac.terminate() != 0;
      ++addr;
      if (last) break;
      if (addr >= total) return fail("slice data goes on behind the last coding tree block");
      if (P->wpp && addr % S->wctb == 0) {
        if (!cabac.terminate()) return fail("end_of_subset_one_bit is not set at the end of row %d", ry);
        cabac.restart(cabac.aligned_end());
      }
    }
    return true;
  }

  // ------------------------------------------------------------------------------------------ access units
  bool decode_nal(const uint8_t* nal, size_t n) {
    if (n < 2) return fail("NAL unit of %zu bytes", n);
    if (nal[0] & 0x80) return fail("forbidden_zero_bit is set");
    const int type = (nal[0] >> 1) & 63, layer = (nal[0] & 1) << 5 | nal[1] >> 3, tid = (nal[1] & 7) - 1;
    if (layer) return true;
    const bool vcl = type <= 9 || (type >= 16 && type <= 21);
    if (!vcl && type != 33 && type != 34) return true;  // VPS, AUD, SEI, filler, end of sequence: nothing a decoder of the samples needs
    std::vector<uint8_t> rbsp(n);
    const size_t rn = unescape(nal + 2, n - 2, rbsp.data());
    Bits b(rbsp.data(), rn);
    if (type == 33) return parse_sps(b);
    if (type == 34) return parse_pps(b);
    SliceHdr h;
    if (!parse_slice_header(b, type, &h)) return false;
    if (h.first) {
      if (pic_open && !close_picture()) return false;
      sh = h;
      skipping = false;
      // Sparse decode, the tail of a GOP or of the interval: a picture at decoding position d is output at position
      // d - sps_max_num_reorder_pics or later, so once d exceeds the last wanted position of its GOP by more than
      // that, neither it nor anything that follows up to the next closed GOP can be wanted or be referred to by a
      // wanted picture.  Whatever the DPB holds goes out first (every wanted position is among it or out already),
      // then each such sample is one void output position and no work.  (h264_parser.cpp: start_picture does the same.)
      {
        const uint64_t ordinal = pictures_started++;
        if (!wanted.empty()) {
          uint64_t g0 = 0, g1 = UINT64_MAX;
          if (!keyframes.empty()) {
            auto it = std::upper_bound(keyframes.begin(), keyframes.end(), ordinal);
            if (it != keyframes.end()) g1 = *it;
            if (it != keyframes.begin()) g0 = *(it - 1);
          }
          auto hi = std::lower_bound(wanted.begin(), wanted.end(), g1);
          const bool any = hi != wanted.begin() && *(hi - 1) >= g0;
          if (!any || ordinal > *(hi - 1) + static_cast<uint64_t>(S ? S->max_num_reorder : 0)) {
            cur = nullptr;
            // (what is still waiting beyond the last wanted position would come out in the wrong places - the samples
            // skipped from here on have positions among them: void positions, like theirs)
            const uint64_t keep_to = any ? *(hi - 1) : 0;
            for (Pic& p : dpb)
              if (p.in_use && p.waiting && !any) p.placeholder = true;
            for (;;) {
              if (any && outputs_emitted > keep_to)
                for (Pic& p : dpb)
                  if (p.in_use && p.waiting) p.placeholder = true;
              if (!bump()) break;
            }
            for (Pic& p : dpb)
              if (p.in_use) {
                p.ref = 0;
                drop(&p);
              }
            sink->output_picture(-1);
            ++outputs_emitted;
            skipping = true;
            ++skipped;
            return true;
          }
        }
      }
      if (!start_picture(type, tid)) return false;
      // sparse decode: a picture nothing refers to, at an output position nobody asked for
      const bool sub_layer_non_ref = type <= 14 && !(type & 1);
      if (!wanted.empty() && sub_layer_non_ref && tid == S->max_sub_layers - 1) {
        uint64_t pos = outputs_emitted;
        for (const Pic& p : dpb)
          if (p.in_use && &p != cur && p.waiting && p.poc < cur->poc) ++pos;
        if (!std::binary_search(wanted.begin(), wanted.end(), pos)) {
          cur->placeholder = true;
          cur->waiting = true;
          cur->ref = 0;
          cur->slot = -1;
          skipping = true;
          ++skipped;
          Pic* c = cur;
          cur = nullptr;
          for (Pic& p : dpb)
            if (p.in_use && p.waiting && &p != c) ++p.latency;
          while (waiting() > S->max_num_reorder || latency_hit())
            if (!bump()) break;
          release_unused();
          return true;
        }
      }
      if (!open_picture()) return false;
    } else {
      if (skipping) return true;
      if (!pic_open) return fail("a slice segment that is not the first of its picture, and no picture was started");
      sh = h;
    }
    if (!build_lists()) return false;
    if (num_slices >= amd::hevc::kMaxSlices) return fail("more than %d slices in a picture", amd::hevc::kMaxSlices);
    // 7.4.7.1: slice segment addresses increase within a picture.  The kernels depend on it too: the intra records are
    // listed by coding tree block in the order the blocks were decoded, which must be raster order.
    if (sh.address < next_ctb) return fail("slice segment address %d, but the picture's blocks up to %d are decoded already", sh.address, next_ctb);
    slice_idx = num_slices++;
    slice_addr.push_back(static_cast<uint32_t>(sh.address));
    SliceLoop& sl = buf.slices[slice_idx];
    sl.beta_offset = static_cast<int8_t>(sh.beta_off * 2);
    sl.tc_offset = static_cast<int8_t>(sh.tc_off * 2);
    sl.across = sh.across;
    sl.sao_luma = sh.sao_y;
    sl.sao_chroma = sh.sao_c;
    sl.pad[0] = sl.pad[1] = sl.pad[2] = 0;
    sl.first_ctb = static_cast<uint32_t>(sh.address);
    done.sao |= sh.sao_y || sh.sao_c;
    // explicit weights of this slice's reference lists (8.5.3.3.4.3)
    weights_base[0] = weights_base[1] = 0;
    if (sh.type != SL_I && (sh.type == SL_P ? P->wp : P->wbp)) {
      for (int l = 0; l < (sh.type == SL_B ? 2 : 1); ++l) {
        weights_base[l] = static_cast<int>(done.num_weights);
        for (int i = 0; i < sh.nref[l]; ++i) {
          if (done.num_weights >= buf.cap_weights) return fail("weight table overflow");
          WeightEntry& w = buf.weights[done.num_weights++];
          w.w[0] = sh.wy[l][i]; w.o[0] = sh.oy[l][i];
          w.w[1] = sh.wc[l][i][0]; w.o[1] = sh.oc[l][i][0];
          w.w[2] = sh.wc[l][i][1]; w.o[2] = sh.oc[l][i][1];
          w.log2wd[0] = static_cast<uint8_t>(sh.wd_y);
          w.log2wd[1] = static_cast<uint8_t>(sh.wd_c);
          w.pad[0] = w.pad[1] = 0;
        }
      }
    }
    return slice_data(rbsp.data(), rn);
  }
};

Parser::Parser(PictureSink* sink) : impl_(new Impl) { impl_->sink = sink; }
Parser::~Parser() {
  Impl& d = *impl_;
  if (d.pic_open) {
    d.sink->abandon_picture(d.done);
    d.pic_open = false;
  }
  for (Pic& p : d.dpb)
    if (p.in_use) d.drop(&p);
}

bool Parser::configure(const uint8_t* c, size_t size) {
  Impl& d = *impl_;
  d.error.clear();
  if (size < 23 || c[0] != 1) return d.fail("hvcC: not a version-1 HEVCDecoderConfigurationRecord (%zu bytes)", size);
  d.nal_length_size = (c[21] & 3) + 1;
  size_t o = 23;
  for (int a = 0; a < c[22]; ++a) {
    if (o + 3 > size) return d.fail("hvcC: truncated array header");
    const int count = c[o + 1] << 8 | c[o + 2];
    o += 3;
    for (int i = 0; i < count; ++i) {
      if (o + 2 > size) return d.fail("hvcC: truncated NAL length");
      const size_t len = static_cast<size_t>(c[o]) << 8 | c[o + 1];
      o += 2;
      if (o + len > size) return d.fail("hvcC: NAL unit runs past the record");
      if (len && !d.decode_nal(c + o, len)) return false;
      o += len;
    }
  }
  for (const Sps& s : d.sps)
    if (s.valid && !d.S) d.S = &s;
  if (!d.S) return d.fail("hvcC: no sequence parameter set");
  d.first_picture = true;
  return true;
}

bool Parser::decode(const uint8_t* sample, size_t size) {
  Impl& d = *impl_;
  if (!d.error.empty()) return false;
  size_t o = 0;
  bool ok = true;
  while (ok && o + static_cast<size_t>(d.nal_length_size) <= size) {
    size_t len = 0;
    for (int i = 0; i < d.nal_length_size; ++i) len = len << 8 | sample[o + i];
    o += static_cast<size_t>(d.nal_length_size);
    if (len > size - o) {
      ok = d.fail("NAL unit of %zu bytes runs past the sample (%zu left)", len, size - o);
      break;
    }
    if (len) ok = d.decode_nal(sample + o, len);
    o += len;
  }
  if (ok && d.pic_open) ok = d.close_picture();
  if (!ok && d.pic_open) {
    d.sink->abandon_picture(d.done);
    d.pic_open = false;
    if (d.cur) {  // the picture still takes its output position
      d.cur->placeholder = true;
      d.cur->slot = -1;
      d.sink->release_surface(d.done.slot);
      d.finish_picture();
    }
  }
  d.skipping = false;
  // (the slice data readers report running past the end of their NAL unit by returning false, nothing else)
  if (!ok && d.error.empty()) d.fail("the slice data is damaged: the arithmetic decoder ran past the end of its NAL unit");
  return ok;
}

bool Parser::flush() {
  Impl& d = *impl_;
  while (d.bump()) {
  }
  for (Pic& p : d.dpb)
    if (p.in_use) {
      p.ref = 0;
      d.drop(&p);
    }
  d.first_picture = true;
  d.prev_tid0_poc = 0;
  d.outputs_emitted = 0;
  d.pictures_started = 0;
  d.wanted.clear();
  d.keyframes.clear();
  d.error.clear();
  return true;
}

void Parser::set_wanted(const std::vector<uint64_t>& ordinals) {
  impl_->wanted = ordinals;
  std::sort(impl_->wanted.begin(), impl_->wanted.end());
}
void Parser::set_keyframes(const std::vector<uint64_t>& ordinals) {
  impl_->keyframes = ordinals;
  std::sort(impl_->keyframes.begin(), impl_->keyframes.end());
}
int Parser::width() const { return impl_->S ? impl_->S->width - impl_->S->crop_l - impl_->S->crop_r : 0; }
int Parser::height() const { return impl_->S ? impl_->S->height - impl_->S->crop_t - impl_->S->crop_b : 0; }
const std::string& Parser::error() const { return impl_->error; }
int64_t Parser::pictures_decoded() const { return impl_->decoded; }
int64_t Parser::pictures_skipped() const { return impl_->skipped; }
std::string Parser::describe() const {
  const Sps* s = impl_->S;
  if (!s) return "no sequence parameter set";
  char tmp[160];
  snprintf(tmp, sizeof(tmp), "HEVC Main %dx%d (coded %dx%d), CTB %d", width(), height(), s->width, s->height, 1 << s->log2_ctb);
  return tmp;
}

}  // namespace hevc
}  // namespace hwang
