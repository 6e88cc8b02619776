#include "h264_parser.h"

#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#if defined(HWANG_CABAC_TRACE) || defined(HWANG_REGION_COUNTERS)
#include <stdio.h>
#include <vector>
#endif
#include <emmintrin.h>
#include <immintrin.h>

namespace hwang {
namespace h264 {

#include "h264_tables.h"

using amd::MbInfo;
using amd::MbMotion;
using amd::SliceParams;

namespace {

// kMaxDpb: entries of the parser's picture table - references (<= 16), pictures waiting for output (reorder depth
// <= 16) and void positions of skipped pictures among them
constexpr int kMaxSps = 32, kMaxPps = 256, kMaxDpb = 40, kMaxRefs = 32;

inline int iclip(int lo, int hi, int v) { return v < lo ? lo : (v > hi ? hi : v); }
inline int imin(int a, int b) { return a < b ? a : b; }
inline int imax(int a, int b) { return a > b ? a : b; }
inline int iabs(int a) { return a < 0 ? -a : a; }
inline int median3(int a, int b, int c) {
  return a > b ? (b > c ? b : (a > c ? c : a)) : (a > c ? a : (b > c ? c : b));
}

// ---------------------------------------------------------------------------
// Bit reader over an RBSP: 64-bit window, refilled bytewise.
// ---------------------------------------------------------------------------
struct Bits {
  const uint8_t* buf = nullptr;
  int size_bits = 0;
  int pos = 0;
  bool overrun = false;

  int left() const { return size_bits - pos; }
  unsigned peek(int n) const {  // n <= 25
    uint32_t v = 0;
    const int byte = pos >> 3;
    const int nbytes = (size_bits + 7) >> 3;
    for (int i = 0; i < 4; ++i) v = (v << 8) | (byte + i < nbytes ? buf[byte + i] : 0);
    return (v << (pos & 7)) >> (32 - n);
  }
  unsigned get(int n) {
    if (n == 0) return 0;
    if (n < 0 || n > 32) {  // never produced by a conforming stream; a crafted one must not move pos backwards
      overrun = true;
      pos = size_bits;
      return 0;
    }
    if (n > 25) {  // one level only: n - 16 <= 16
      unsigned hi = get(n - 16);
      return (hi << 16) | get(16);
    }
    if (pos + n > size_bits) {
      overrun = true;
      pos = size_bits;
      return 0;
    }
    unsigned v = peek(n);
    pos += n;
    return v;
  }
  unsigned bit() { return get(1); }
  unsigned ue() {
    int zeros = 0;
    while (!bit()) {
      if (++zeros > 31 || overrun) {  // codeNum must fit 32 bits (9.1)
        overrun = true;
        return 0;
      }
    }
    return zeros ? ((1u << zeros) - 1 + get(zeros)) : 0;
  }
  int se() {
    unsigned k = ue();
    return (k & 1) ? static_cast<int>((k + 1) >> 1) : -static_cast<int>(k >> 1);
  }
  bool more_rbsp_data() const {
    int last = size_bits - 1;
    while (last >= 0 && !((buf[last >> 3] >> (7 - (last & 7))) & 1)) --last;
    return pos < last;
  }
};

// ---------------------------------------------------------------------------
// CABAC decoding engine (9.3.3.2).  `value` carries the 9-bit offset plus
// `extra` prefetched stream bits, so renormalisation is a counter decrement
// and the stream is touched once per byte instead of once per bit.
// ---------------------------------------------------------------------------
// CABAC decoding engine (9.3.1.2, 9.3.3.2).  `value` holds the arithmetic
// decoder's offset scaled by 2^extra: `extra` bits of the stream beyond the
// nine the standard keeps are already shifted in, so renormalisation is a
// shift of `range` plus bookkeeping and the bitstream is touched once per 32
// bits.  Decisions are branch-free (the LPS/MPS outcome is data, not control
// flow - about half of all bins sit in contexts near p = 0.5 and mispredict).
struct CabacTables {
  uint32_t lps4[128];     // [state = pStateIdx << 1 | valMPS]: rangeTabLPS for qCodIRangeIdx 0..3, one byte each - the load
                          // depends on the context only, so it is off the range -> range chain that paces consecutive bins
  uint8_t mlps[256];      // the next state: [128 + state] after the MPS, [128 + ~state] = [127 - state] after the LPS
  CabacTables() {
    for (int st = 0; st < 128; ++st) {
      const int ps = st >> 1, mps = st & 1;
      mlps[128 + st] = static_cast<uint8_t>(((ps < 62 ? ps + 1 : ps) << 1) | mps);
      mlps[127 - st] = static_cast<uint8_t>((cabac_trans_lps[ps] << 1) | (ps == 0 ? !mps : mps));
      lps4[st] = 0;
      for (int q = 0; q < 4; ++q) lps4[st] |= static_cast<uint32_t>(cabac_range_lps[ps][q]) << (8 * q);
    }
  }
};
static const CabacTables g_cabac_tables;

// One decision of the arithmetic decoder (9.3.3.2.1) without the refill: the engine's registers by reference, so that
// every caller keeps them where it has them (members, or locals for a run of bins).  The outcome is a sign: state ^
// ~sign indexes ONE table for the next state on either side and its low bit is the decoded bin - 8-10 % fewer
// instructions per picture than a mask, a two-column table and an exclusive-or with the MPS, and the rangeTabLPS load
// stays off the range -> range chain (counted on the GPU box, profiles/r05_host_perf_counters.txt: -3.5 % cycles of an
// intra picture, -2.5 % of a P picture; the same with the load ON the chain - one table indexed by range and state -
// has as few instructions and gains less).
static inline __attribute__((always_inline)) int cabac_step(uint16_t* state, int ctx, uint64_t& value, uint32_t& range, int& extra) {
  const int32_t s = state[ctx];
  const uint32_t lps = (g_cabac_tables.lps4[s] >> ((range >> 3) & 24)) & 0xff;
  uint32_t r = range - lps;
  const uint64_t scaled = static_cast<uint64_t>(r) << extra;
  const int64_t d = static_cast<int64_t>(value - scaled);  // >= 0: the LPS
  value = d >= 0 ? static_cast<uint64_t>(d) : value;
  r = d >= 0 ? lps : r;
  const int32_t sx = s ^ static_cast<int32_t>(~(d >> 63));
  state[ctx] = g_cabac_tables.mlps[128 + sx];
  const int shift = __builtin_clz(r) - 23;  // 0 when r >= 256
  range = r << shift;
  extra -= shift;
  return sx & 1;
}

// cold path of the arithmetic decoder: called with extra < 8 (or -9 at the
// start of a slice), returns with extra >= 24
__attribute__((noinline)) static uint64_t cabac_refill_slow(uint64_t value, int* extra_io, const uint8_t** p_io,
                                                           const uint8_t* end, bool* overrun) {
  int extra = *extra_io;
  const uint8_t* p = *p_io;
  if (extra >= 0 && p + 4 <= end) {
    const uint32_t w = (static_cast<uint32_t>(p[0]) << 24) | (static_cast<uint32_t>(p[1]) << 16) |
                       (static_cast<uint32_t>(p[2]) << 8) | p[3];
    p += 4;
    value = (value << 32) | w;
    extra += 32;
  } else {
    while (extra < 24) {
      uint32_t b = 0;
      if (p < end) b = *p;
      else if (p > end + 8) *overrun = true;
      ++p;
      value = (value << 8) | b;
      extra += 8;
    }
  }
  *extra_io = extra;
  *p_io = p;
  return value;
}

#ifdef HWANG_REGION_COUNTERS
// Development aid (tools/hostprof/build_variant.sh NAME -DHWANG_REGION_COUNTERS): user-mode instructions retired per
// REGION of the macroblock layer, exact - the counter is read (a system call whose kernel side is not counted) whenever
// the parser passes from one region to the next; every switch costs the region it leaves ~25-30 instructions.
#include <linux/perf_event.h>
#include <sys/syscall.h>
#include <unistd.h>
enum { R_OTHER, R_MBHEAD, R_PSKIP, R_BEGIN, R_MBTYPE, R_INTER_REFS, R_INTER_MVD, R_INTRAMODES, R_CBP, R_T8X8, R_QPD, R_RES_FILL, R_RESIDUAL, R_RES_BLOCK, R_COMMIT, R_COUNT };
struct RegionCounters {
  int fd = -1;
  uint64_t instr[R_COUNT] = {0}, enters[R_COUNT] = {0}, last = 0;
  int cur = R_OTHER;
  RegionCounters() {
    perf_event_attr a;
    memset(&a, 0, sizeof(a));
    a.size = sizeof(a);
    a.type = PERF_TYPE_HARDWARE;
#ifdef HWANG_REGION_CYCLES  // (cycles instead: the reads disturb the pipeline, the shares still say where the time is)
    a.config = PERF_COUNT_HW_CPU_CYCLES;
#else
    a.config = PERF_COUNT_HW_INSTRUCTIONS;
#endif
    a.exclude_kernel = 1;
    a.exclude_hv = 1;
    fd = static_cast<int>(syscall(SYS_perf_event_open, &a, 0, -1, -1, 0));
  }
  void enter(int r) {
    uint64_t now = 0;
    if (fd < 0 || read(fd, &now, 8) != 8) return;
    instr[cur] += now - last;
    last = now;
    cur = r;
    ++enters[r];
  }
  ~RegionCounters() {
    static const char* const names[R_COUNT] = {"outside the macroblock layer", "position + skip flag + end_of_slice", "P_Skip", "begin_mb", "mb_type",
                                               "inter: shapes + references", "inter: differences", "intra prediction modes", "coded_block_pattern",
                                               "transform_size_8x8_flag", "mb_qp_delta", "residual: fill_nz_cache", "residual: around the blocks",
                                               "residual: cabac_block", "commit_mb"};
    enter(R_OTHER);
    uint64_t total = 0;
    for (uint64_t v : instr) total += v;
    for (int i = 0; i < R_COUNT; ++i)
      fprintf(stderr, "region %-36s %14llu instructions %5.1f %%  %10llu entries  %8.1f per entry\n", names[i], (unsigned long long)instr[i],
              total ? 100.0 * instr[i] / total : 0.0, (unsigned long long)enters[i], enters[i] ? 1.0 * instr[i] / enters[i] : 0.0);
  }
};
static RegionCounters g_regions;
#define REGION(r) g_regions.enter(r)
#else
#define REGION(r) do { } while (0)
#endif

#ifdef HWANG_CABAC_TRACE
// Development aid (tools/hostprof/engine_bench.cpp): the context index of every bin of ONE slice (HWANG_CABAC_TRACE_SLICE
// = its number in the run, default 0), with the contexts and the bytes the slice started from - so that formulations of
// the arithmetic decoder can be timed on a real bin sequence with no syntax around it.  0xffff = bypass, 0xfffe = terminate.
struct CabacTrace {
  std::vector<uint16_t> ctx;
  int slice = -1, want = getenv("HWANG_CABAC_TRACE_SLICE") ? atoi(getenv("HWANG_CABAC_TRACE_SLICE")) : 0;
  bool on = false;
  void begin(const uint16_t* state, int nstate, const uint8_t* data, const uint8_t* end) {
    flush();
    on = ++slice == want;
    if (!on) return;
    FILE* f = fopen("cabac_trace_init.bin", "wb");
    const uint32_t n = static_cast<uint32_t>(end - data), ns = static_cast<uint32_t>(nstate);
    fwrite(&ns, 4, 1, f);
    fwrite(state, 2, ns, f);
    fwrite(&n, 4, 1, f);
    fwrite(data, 1, n, f);
    fclose(f);
  }
  void flush() {
    if (!on) return;
    FILE* f = fopen("cabac_trace_ctx.bin", "wb");
    fwrite(ctx.data(), 2, ctx.size(), f);
    fclose(f);
    on = false;
  }
  ~CabacTrace() { flush(); }
};
static CabacTrace g_cabac_trace;
#define CABAC_TRACE(c) do { if (g_cabac_trace.on) g_cabac_trace.ctx.push_back(static_cast<uint16_t>(c)); } while (0)
#else
#define CABAC_TRACE(c) do { } while (0)
#endif

struct Cabac {
  const uint8_t* p = nullptr;
  const uint8_t* end = nullptr;
  const uint8_t* start = nullptr;
  uint64_t value = 0;
  int extra = 0;
  uint32_t range = 0;
  bool overrun = false;
  // 16-bit on purpose: a byte store may alias anything, which would force the
  // engine's registers (value, range, extra, p) through memory around every bin
  uint16_t state[CABAC_CTX_COUNT];

  // called with extra < 8 (or -9 at the start): afterwards extra >= 24
  void refill() {
    if (extra >= 0 && p + 4 <= end) {
      const uint32_t w = (static_cast<uint32_t>(p[0]) << 24) | (static_cast<uint32_t>(p[1]) << 16) |
                         (static_cast<uint32_t>(p[2]) << 8) | p[3];
      p += 4;
      value = (value << 32) | w;
      extra += 32;
      return;
    }
    while (extra < 24) {
      uint32_t b = 0;
      if (p < end) b = *p;
      else if (p > end + 8) overrun = true;
      ++p;
      value = (value << 8) | b;
      extra += 8;
    }
  }
  void begin(const uint8_t* data, const uint8_t* data_end) {
    start = p = data;
    end = data_end;
    value = 0;
    extra = -9;  // the first nine bits become the offset
    overrun = false;
    range = 510;
    refill();
#ifdef HWANG_CABAC_TRACE
    g_cabac_trace.begin(state, CABAC_CTX_COUNT, data, data_end);
#endif
  }
  // bits of the stream consumed so far
  int bits_consumed() const { return static_cast<int>(p - start) * 8 - extra; }

  void init_contexts(int slice_type, int idc, int qp) {
    const int t = slice_type == 2 ? 0 : 1 + idc;
    qp = iclip(0, 51, qp);
    for (int i = 0; i < CABAC_CTX_COUNT; ++i) {
      int m = cabac_init_mn[t][i][0], n = cabac_init_mn[t][i][1];
      int pre = iclip(1, 126, ((m * qp) >> 4) + n);
      state[i] = pre <= 63 ? static_cast<uint8_t>((63 - pre) << 1)
                           : static_cast<uint8_t>(((pre - 64) << 1) | 1);
    }
  }
  __attribute__((always_inline)) inline int bin(int ctx) {
    CABAC_TRACE(ctx);
    const int b = cabac_step(state, ctx, value, range, extra);
    if (__builtin_expect(extra < 8, 0)) refill();
    return b;
  }
  __attribute__((always_inline)) inline int bypass() {
    CABAC_TRACE(0xffff);
    extra -= 1;
    if (__builtin_expect(extra < 8, 0)) refill();
    const uint64_t scaled = static_cast<uint64_t>(range) << extra;
    const uint64_t mask = ~static_cast<uint64_t>(static_cast<int64_t>(value - scaled) >> 63);
    value -= scaled & mask;
    return static_cast<int>(mask & 1);
  }
  inline int terminate() {
    CABAC_TRACE(0xfffe);
    range -= 2;
    const uint64_t scaled = static_cast<uint64_t>(range) << extra;
    if (value >= scaled) return 1;
    if (range < 256) {
      range <<= 1;
      extra -= 1;
      if (extra < 8) refill();
    }
    return 0;
  }
};

// The engine's registers as locals of one syntax-element reader (leaf functions
// only: nothing else may touch the Cabac object while one of these is alive).
// Through the object every bin loads and stores offset, range and bit budget;
// as locals whose address never escapes they stay in registers for the run of
// bins and are written back once, in the destructor.
struct CabacLocal {
  Cabac& c;
  uint64_t value;
  uint32_t range;
  int extra;
  const uint8_t* p;
  const uint8_t* const end;
  uint16_t* const state;
  bool overrun = false;
  __attribute__((always_inline)) explicit CabacLocal(Cabac& cc)
      : c(cc), value(cc.value), range(cc.range), extra(cc.extra), p(cc.p), end(cc.end), state(cc.state) {}
  __attribute__((always_inline)) ~CabacLocal() {
    c.value = value;
    c.range = range;
    c.extra = extra;
    c.p = p;
    if (overrun) c.overrun = true;
  }
  CabacLocal(const CabacLocal&) = delete;
  CabacLocal& operator=(const CabacLocal&) = delete;
  __attribute__((always_inline)) void refill() {
    int e = extra;
    const uint8_t* q = p;
    bool ov = false;
    value = cabac_refill_slow(value, &e, &q, end, &ov);
    extra = e;
    p = q;
    overrun |= ov;
  }
  __attribute__((always_inline)) int bin(int ctx) {
    CABAC_TRACE(ctx);
    const int b = cabac_step(state, ctx, value, range, extra);
    if (__builtin_expect(extra < 8, 0)) refill();
    return b;
  }
  __attribute__((always_inline)) int bypass() {
    CABAC_TRACE(0xffff);
    extra -= 1;
    if (__builtin_expect(extra < 8, 0)) refill();
    const uint64_t scaled = static_cast<uint64_t>(range) << extra;
    const uint64_t mask = ~static_cast<uint64_t>(static_cast<int64_t>(value - scaled) >> 63);
    value -= scaled & mask;
    return static_cast<int>(mask & 1);
  }
  __attribute__((always_inline)) int terminate() {
    CABAC_TRACE(0xfffe);
    range -= 2;
    const uint64_t scaled = static_cast<uint64_t>(range) << extra;
    if (value >= scaled) return 1;
    if (range < 256) {
      range <<= 1;
      extra -= 1;
      if (extra < 8) refill();
    }
    return 0;
  }
};

// ---------------------------------------------------------------------------
struct Sps {
  bool valid = false;
  int profile_idc = 0, level_idc = 0, constraint_flags = 0;
  int log2_max_frame_num = 4, poc_type = 0, log2_max_poc_lsb = 4;
  int delta_pic_order_always_zero = 0, offset_for_non_ref_pic = 0, offset_for_top_to_bottom = 0;
  int num_ref_frames_in_poc_cycle = 0;
  int offset_for_ref_frame[256];
  int max_num_ref_frames = 0;
  int mb_width = 0, mb_height = 0;
  int direct_8x8_inference = 0;
  int crop_left = 0, crop_right = 0, crop_top = 0, crop_bottom = 0;
  int bitstream_restriction = 0, num_reorder_frames = -1;
  int scaling_matrix_present = 0;
  uint8_t scaling4[6][16];
  uint8_t scaling8[2][64];
};

struct Pps {
  bool valid = false;
  int sps_id = 0, cabac = 0, bottom_field_pic_order_present = 0;
  int num_ref_idx_default[2] = {1, 1};
  int weighted_pred = 0, weighted_bipred_idc = 0;
  int init_qp = 26;
  int chroma_qp_offset[2] = {0, 0};
  int deblocking_control_present = 0, constrained_intra_pred = 0, redundant_pic_cnt_present = 0;
  int transform_8x8_mode = 0;
  uint8_t scaling4[6][16];
  uint8_t scaling8[2][64];
};

// What the slice-data decoders share of a picture: its motion for co-located
// lookups and the surface it is reconstructed into.  In the serial mode the
// parser thread is the only user; with entropy workers (Parser's `pool`) a
// picture's records are written by the worker that decodes it and read by the
// workers of later B pictures (direct prediction), so the object is reference
// counted - the DPB entry, the picture's own task and every task that has it in
// a reference list hold it - and its surface goes back to the sink when the
// last holder lets go.
struct PicData {
  int slot = -1;          // GPU surface; -1 for a skipped picture
  int uid = 0;
  MbMotion* motion = nullptr;         // owned by the sink (pinned, per surface)
  std::vector<int32_t> ref_uid[2];    // per 8x8
  std::vector<uint8_t> mb_intra;
  // P macroblocks are parsed without deriving their motion vectors (motion[]
  // holds the differences); mvrec keeps what the derivation needs per
  // macroblock: MbInfo::mv_flags, mv_shape, mv_sub.  `derived` is the lazily
  // computed final motion, only ever needed when a B picture uses this one as
  // its co-located picture.
  std::vector<uint8_t> mvrec;
  bool mv_deferred = false;
  // compact hand-over (PicBuffers::pmo / pmx): kept for the lazy derivation
  const amd::PMotion* pmo = nullptr;
  const uint32_t* pmx = nullptr;
  std::vector<MbMotion> derived;
  bool derived_valid = false;
  // entropy workers: `parsed` is raised (under mu) when every macroblock record
  // of the picture is final
  std::mutex mu;
  std::condition_variable cv;
  bool parsed = true;
};

// A decoded (or skipped) picture in the DPB: the sequence-level bookkeeping,
// touched by the parser thread only.
struct Pic {
  bool in_use = false;
  int slot = -1;          // = d->slot
  int uid = 0;
  int poc = 0;
  int frame_num = 0, frame_num_wrap = 0, pic_num = 0;
  int ref = 0;            // 0 unused, 1 short-term, 2 long-term
  int long_term_idx = -1;
  bool needed_for_output = false;
  std::shared_ptr<PicData> d;
};

// A reference-list entry as the slice-data decoder sees it: frozen when the
// slice header is parsed (marking and POC of the DPB entry change later).
struct RefSnap {
  PicData* d = nullptr;
  int slot = -1, uid = 0, poc = 0;
  int marking = 0;        // 1 short-term, 2 long-term at the time of this slice
};

struct SliceHdr {
  int first_mb = 0, type = 2, pps_id = 0, frame_num = 0;
  int poc_lsb = 0, delta_poc_bottom = 0, delta_poc[2] = {0, 0};
  int direct_spatial = 0;
  int num_ref_idx[2] = {0, 0};
  int cabac_init_idc = 0, qp = 26;
  int disable_deblock = 0, alpha_offset = 0, beta_offset = 0;
  int luma_log2_denom = 0, chroma_log2_denom = 0;
  int luma_weight[2][kMaxRefs], luma_offset[2][kMaxRefs];
  int chroma_weight[2][kMaxRefs][2], chroma_offset[2][kMaxRefs][2];
  int use_weight = 0;
  int nal_ref_idc = 0;
  bool is_idr = false;
  int long_term_reference_flag = 0, adaptive_marking = 0;
  int mmco_count = 0;
  struct { int op, a, b; } mmco[66];
  Pic* ref_list[2][kMaxRefs + 1];   // parser thread only
  RefSnap ref[2][kMaxRefs + 1];      // what the slice data is decoded against
  int cur_poc = 0;
};

enum { MB_NONE = 0, MB_I4x4, MB_I8x8, MB_I16x16, MB_IPCM, MB_INTER };

struct NB {
  int ref;  // -2 partition not available, -1 list unused / intra, >= 0 refIdx
  int mv[2];
};

enum { PRED_MEDIAN = 0, PRED_16x8_TOP, PRED_16x8_BOTTOM, PRED_8x16_LEFT, PRED_8x16_RIGHT };

const uint8_t b_mb_info[23][3] = {
    {0, 0, 0}, {0, 0, 0}, {0, 1, 1}, {0, 2, 2}, {1, 0, 0}, {2, 0, 0}, {1, 1, 1}, {2, 1, 1},
    {1, 0, 1}, {2, 0, 1}, {1, 1, 0}, {2, 1, 0}, {1, 0, 2}, {2, 0, 2}, {1, 1, 2}, {2, 1, 2},
    {1, 2, 0}, {2, 2, 0}, {1, 2, 1}, {2, 2, 1}, {1, 2, 2}, {2, 2, 2}, {3, 0, 0}};
const uint8_t b_sub_info[13][2] = {{0, 3}, {0, 0}, {0, 1}, {0, 2}, {1, 0}, {2, 0}, {1, 1},
                                   {2, 1}, {1, 2}, {2, 2}, {3, 0}, {3, 1}, {3, 2}};
inline bool uses_list(int pred, int list) { return pred == 2 || pred == list; }

const uint8_t identity_scan16[16] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15};
const uint8_t dc_scan4[4] = {0, 1, 2, 3};


// Recycles PicData objects (their vectors are sized per picture geometry) and
// gives a picture's surface back to the sink when the last holder lets go.
struct PicPool : std::enable_shared_from_this<PicPool> {
  PictureSink* sink = nullptr;
  std::mutex mu;
  std::vector<PicData*> free_;
  ~PicPool() {
    for (PicData* p : free_) delete p;
  }
  std::shared_ptr<PicData> grab() {
    PicData* p = nullptr;
    {
      std::lock_guard<std::mutex> lk(mu);
      if (!free_.empty()) {
        p = free_.back();
        free_.pop_back();
      }
    }
    if (!p) p = new PicData();
    p->slot = -1;
    p->motion = nullptr;
    p->pmo = nullptr;
    p->pmx = nullptr;
    p->mv_deferred = false;
    p->derived_valid = false;
    p->parsed = true;
    std::shared_ptr<PicPool> self = shared_from_this();
    return std::shared_ptr<PicData>(p, [self](PicData* q) {
      if (q->slot >= 0) self->sink->release_surface(q->slot);
      q->slot = -1;
      std::lock_guard<std::mutex> lk(self->mu);
      self->free_.push_back(q);
    });
  }
};

// ===========================================================================
// Entropy decoding of one picture's slice data (7.3.4 slice_data() down to the
// residual blocks): everything below the slice header.  Self-contained per
// picture - what it knows of other pictures comes frozen in the SliceHdr
// (reference snapshots) and through PicData of the co-located picture - so
// several of these may work on consecutive pictures at the same time.  The
// parser thread owns one (serial mode); every entropy worker has its own.
// ===========================================================================
struct SliceData {
  std::string err;  // first error of the picture in hand
  int64_t slices = 0, slices_clean_end = 0;
  bool fail(const char* msg) {
    if (err.empty()) err = msg;
    return false;
  }

  const Sps* asps = nullptr;
  const Pps* apps = nullptr;
  int mbw = 0, mbh = 0;
  PicData* cur = nullptr;
  PicBuffers bufs;
  size_t coef_used = 0;
  bool has_inter = false, has_intra = false;
  int intra_mbs = 0;  // of this picture (the residual kernel's share of the work)

  SliceHdr sl;
  int slice_index = 0;
  // per-picture macroblock state (host only)
  std::vector<int16_t> mb_slice;
  std::vector<uint8_t> mb_kind, mb_skip, mb_direct, direct8, mb_t8x8, mb_cbp, mb_cbf_dc, mb_chroma_mode;
  std::vector<int8_t> i4mode;
  std::vector<uint8_t> nnz;  // mb*24
  std::vector<uint8_t> mvd;  // mb*[2][16][2]
  Bits bits;
  Cabac cabac;
  bool use_cabac = false;
  int last_qp_delta_nonzero = 0;
  int qp = 26;
  int mbs_decoded = 0;  // of the picture in progress
  bool pos_valid = false;  // mb.mbx / mb.mby / mb.addr describe a macroblock of the picture in progress
  const MbMotion* col_motion = nullptr;  // this slice's co-located motion (pred_direct fetches it with its first direct macroblock)

  // P_Skip is 40 % of the macroblocks of a typical P picture and carries no
  // syntax at all: everything it leaves behind is known when the slice starts,
  // except its position, its neighbours and the running QP.  These templates are
  // what begin_mb + decode_skip_mb + commit_mb would have produced field by field.
  MbMotion pskip_motion;
  int32_t pskip_ref_uid = -1;
  MbInfo pskip_info;
  int pskip_info_qp = -1;
  // compact motion hand-over of the picture in progress (bufs.pmo / bufs.pmx)
  bool compact = false;        // macroblocks of P / I slices are written as PMotion
  bool any_compact = false, any_full_motion = false;
  size_t pmx_used = 0;         // words, row table included

  // current macroblock
  struct Mb {
    int mbx, mby, addr, kind, skip, direct16, t8x8, i16_mode, chroma_mode, cbp, qp;
    // neighbour macroblocks in the same slice (and inside the picture), by
    // offset: av[(oy + 1) * 3 + ox + 1] for ox in -1..1, oy in -1..0
    bool av[6];
    bool avA, avB;
    uint8_t av_bits;  // A | B << 1 | C << 2 | D << 3: the order of MBF_AVAIL_A .. D (<< 2) and MVF_AVAIL_A .. D (<< 4)
    int mv_shape, mv_sub, mv_pskip, mv_defer;  // deferred motion-vector derivation (P macroblocks)
    int pm_parts;  // compact hand-over: partitions whose difference has been written so far
    // motion record of each available, non-intra neighbour (same index as
    // av[]; [4] = this macroblock), nullptr otherwise; filled by begin_mb
    const MbMotion* nbmo[6];
    int sub_type[4];
    int8_t i4_mode[16];
    uint8_t* nnz;       // this macroblock's 24 counts IN the picture's array (begin_mb): written in place - a copy at the end would
                        // load 16 bytes that byte stores still in the store queue had just written, and wait for them to drain
    uint16_t nz_mask;   // CABAC: the 4x4 luma blocks (raster) with levels, as the residual reader saw them
    uint8_t nzc[48];                // fill_nz_cache(): coded flags of this macroblock's blocks and of those along its edges
    uint8_t dc_left, dc_top;        // ... and the DC blocks' flags of the two neighbour macroblocks
    int cbf_dc;
    uint32_t coded_mask;
    // the macroblock's non-zero levels as they will be handed over (MbInfo::ncoef): (group * 16 + position) << 16 | level
    uint32_t ent[26 * 16];
    int nent;
    int16_t scratch[26][16];  // dense groups: CAVLC staging, and the fallback form of macroblocks with very many levels
  } mb;

  // A new picture: `c` receives its motion, `b` its records.
  void begin_picture(const Sps* sps, PicData* c, const PicBuffers& b) {
    asps = sps;
    mbw = sps->mb_width;
    mbh = sps->mb_height;
    const int nmb = mbw * mbh;
    cur = c;
    bufs = b;
    err.clear();
    if (bufs.pmx && bufs.pmx_capacity < static_cast<size_t>(mbh) + static_cast<size_t>(nmb) * 15) bufs.pmo = nullptr, bufs.pmx = nullptr;
    // a sink that derives vectors itself takes them compactly; without the buffers for that the host derives
    if (!bufs.pmo || !bufs.pmx) bufs.device_mv = false, bufs.pmo = nullptr, bufs.pmx = nullptr;
    if (!bufs.device_mv) bufs.pmo = nullptr, bufs.pmx = nullptr;
    compact = false;
    any_compact = any_full_motion = false;
    pmx_used = bufs.pmx ? static_cast<size_t>(mbh) : 0;
    // Per-macroblock state is (re)written by begin_mb / commit_mb before any
    // later macroblock of this picture reads it (neighbours precede in raster
    // order; FMO/ASO are rejected), so nothing is cleared per picture.
    if (mb_slice.size() != static_cast<size_t>(nmb)) {
      mb_slice.assign(nmb, -1);
      mb_kind.assign(nmb, 0);
      mb_skip.assign(nmb, 0);
      mb_direct.assign(nmb, 0);
      direct8.assign(nmb * 4, 0);
      mb_t8x8.assign(nmb, 0);
      mb_cbp.assign(nmb, 0);
      mb_cbf_dc.assign(nmb, 0);
      mb_chroma_mode.assign(nmb, 0);
      i4mode.assign(nmb * 16, 2);
      nnz.assign(nmb * 24, 0);
      mvd.assign(static_cast<size_t>(nmb) * 64, 0);
    }
    for (int l = 0; l < 2; ++l)
      if (c->ref_uid[l].size() != static_cast<size_t>(nmb) * 4) c->ref_uid[l].assign(nmb * 4, -1);
    if (c->mb_intra.size() != static_cast<size_t>(nmb)) c->mb_intra.assign(nmb, 0);
    if (c->mvrec.size() != static_cast<size_t>(nmb) * 3) c->mvrec.assign(static_cast<size_t>(nmb) * 3, 0);
    c->mv_deferred = false;
    c->derived_valid = false;
    mbs_decoded = 0;
    coef_used = 0;
    has_inter = has_intra = false;
    intra_mbs = 0;
    pos_valid = false;
  }

  // One slice: header `h` (with its reference snapshots), the picture parameter set it names, its 1-based
  // position in the picture, and the bit reader standing at slice_data().
  bool decode_slice(const SliceHdr& h, const Pps* pps, int index, Bits b) {
    sl = h;
    apps = pps;
    slice_index = index;
    return decode_slice_data(b) && err.empty();
  }

  // All slices are in: what the slice data contributes to PicDone.
  bool finish_picture(PicDone* done) {
    if (mbs_decoded != mbw * mbh) return fail("picture ended with undecoded macroblocks");
    if (cur->mv_deferred && !bufs.device_mv) {
      derive_deferred(cur, cur->motion);  // in place: nothing has been submitted yet
      cur->mv_deferred = false;
    }
    done->mv_deferred = cur->mv_deferred;
    done->compact_motion = any_compact;
    done->full_motion = any_full_motion;
    done->pmx_count = pmx_used;
    done->coef_count = coef_used;
    done->has_inter = has_inter;
    done->has_intra = has_intra;
    done->num_intra_mbs = intra_mbs;
    return err.empty();
  }

  // =================================================================== MBs
  static_assert(amd::MBF_AVAIL_A == 4 && amd::MBF_AVAIL_B == 8 && amd::MBF_AVAIL_C == 16 && amd::MBF_AVAIL_D == 32, "Mb::av_bits << 2");
  static_assert(amd::MVF_AVAIL_A == 16 && amd::MVF_AVAIL_B == 32 && amd::MVF_AVAIL_C == 64 && amd::MVF_AVAIL_D == 128, "Mb::av_bits << 4");
  // Sets the current macroblock's position and which neighbours are available
  // (6.4.9: inside the picture and in the same slice).  Everything else asks
  // mb.av[] instead of re-deriving it per syntax element.
  inline void set_position(int addr) {
    if (addr == mb.addr + 1 && pos_valid) {  // the next macroblock in raster order: no division
      if (++mb.mbx == mbw) {
        mb.mbx = 0;
        ++mb.mby;
      }
    } else {
      mb.mbx = addr % mbw;
      mb.mby = addr / mbw;
      pos_valid = true;
    }
    mb.addr = addr;
    if (bufs.pmx && mb.mbx == 0) bufs.pmx[mb.mby] = static_cast<uint32_t>(pmx_used);  // this row's extras start here
    const bool left = mb.mbx > 0, top = mb.mby > 0, right = mb.mbx + 1 < mbw;
    // (a slice is a run of macroblocks in raster order - FMO / ASO are refused - and starts where the one before ended:
    // a macroblock in front of this one belongs to this slice exactly if it is not in front of the slice's first)
    const int first = sl.first_mb;
    mb.av[0] = left && top && addr - mbw - 1 >= first;   // D
    mb.av[1] = top && addr - mbw >= first;               // B
    mb.av[2] = right && top && addr - mbw + 1 >= first;  // C
    mb.av[3] = left && addr - 1 >= first;                // A
    mb.av[4] = true;                                     // the macroblock itself
    mb.av[5] = false;                                    // right neighbour: never decoded yet
    mb.avA = mb.av[3];
    mb.avB = mb.av[1];
    mb.av_bits = static_cast<uint8_t>((mb.av[3] ? 1 : 0) | (mb.av[1] ? 2 : 0) | (mb.av[2] ? 4 : 0) | (mb.av[0] ? 8 : 0));
  }
  inline bool nb_avail(int ox, int oy) const { return mb.av[(oy + 1) * 3 + ox + 1]; }
  inline bool mb_avail(int mbx, int mby) const {
    if (mbx < 0 || mby < 0 || mbx >= mbw || mby >= mbh) return false;
    return mb_slice[mby * mbw + mbx] == slice_index;
  }
  inline bool intra_mb_ok(int mbx, int mby) const {
    if (!nb_avail(mbx - mb.mbx, mby - mb.mby)) return false;
    return !(apps->constrained_intra_pred && mb_kind[mby * mbw + mbx] == MB_INTER);
  }
  uint8_t* mvd_at(int addr, int list, int blk) { return &mvd[(static_cast<size_t>(addr) * 2 + list) * 32 + blk * 2]; }

  NB fetch_nb(int list, int rx, int ry, int cur_zz) const {
    NB nb = {-2, {0, 0}};
    if (rx > 3 && ry >= 0) return nb;
    const int k = (ry < 0 ? 0 : 3) + (rx < 0 ? 0 : (rx > 3 ? 2 : 1));
    const int bx = rx & 3, by = ry & 3;
    if (k == 4) {
      if (raster_to_blk4[by * 4 + bx] >= cur_zz) return nb;
    } else if (!mb.av[k]) {
      return nb;
    }
    nb.ref = -1;
    if (!mb.nbmo[k]) return nb;  // intra neighbour
    const MbMotion& mo = *mb.nbmo[k];
    int r = mo.ref_idx[list][(by >> 1) * 2 + (bx >> 1)];
    if (r < 0) return nb;
    nb.ref = r;
    nb.mv[0] = mo.mv[list][by * 4 + bx][0];
    nb.mv[1] = mo.mv[list][by * 4 + bx][1];
    return nb;
  }

  void pred_mv(int list, int bx, int by, int bw, int ref, int dir, int out[2]) const {
    const int zz = raster_to_blk4[by * 4 + bx];
    NB A = fetch_nb(list, bx - 1, by, zz);
    NB B = fetch_nb(list, bx, by - 1, zz);
    NB C = fetch_nb(list, bx + bw, by - 1, zz);
    if (C.ref == -2) C = fetch_nb(list, bx - 1, by - 1, zz);
    pred_mv_from(A, B, C, ref, dir, out);
  }
  // 8.4.1.3 with the three neighbours in hand (C replaced by D already where C is not available)
  static void pred_mv_from(const NB& A, const NB& B, const NB& C, int ref, int dir, int out[2]) {
    const NB* pick = nullptr;
    if (dir == PRED_16x8_TOP && B.ref == ref) pick = &B;
    if (dir == PRED_16x8_BOTTOM && A.ref == ref) pick = &A;
    if (dir == PRED_8x16_LEFT && A.ref == ref) pick = &A;
    if (dir == PRED_8x16_RIGHT && C.ref == ref) pick = &C;
    if (!pick) {
      int match = (A.ref == ref) + (B.ref == ref) + (C.ref == ref);
      if (match == 1) pick = A.ref == ref ? &A : (B.ref == ref ? &B : &C);
      else if (match == 0 && B.ref == -2 && C.ref == -2 && A.ref != -2) pick = &A;
    }
    if (pick) {
      out[0] = pick->mv[0];
      out[1] = pick->mv[1];
    } else {
      out[0] = median3(A.mv[0], B.mv[0], C.mv[0]);
      out[1] = median3(A.mv[1], B.mv[1], C.mv[1]);
    }
  }

  void set_mv_block(int list, int bx, int by, int bw, int bh, int mvx, int mvy) {
    MbMotion& mo = cur->motion[mb.addr];
    const uint32_t packed = static_cast<uint16_t>(mvx) | (static_cast<uint32_t>(static_cast<uint16_t>(mvy)) << 16);
    uint32_t* dst = reinterpret_cast<uint32_t*>(mo.mv[list]);
    for (int y = by; y < by + bh; ++y)
      for (int x = bx; x < bx + bw; ++x) dst[y * 4 + x] = packed;
  }
  void set_mvd_block(int list, int bx, int by, int bw, int bh, int dx, int dy) {
    uint8_t ax = static_cast<uint8_t>(imin(iabs(dx), 255)), ay = static_cast<uint8_t>(imin(iabs(dy), 255));
    for (int y = by; y < by + bh; ++y)
      for (int x = bx; x < bx + bw; ++x) {
        uint8_t* m = mvd_at(mb.addr, list, y * 4 + x);
        m[0] = ax;
        m[1] = ay;
      }
  }
  void set_ref8(int list, int b8, int ref) {
    MbMotion& mo = cur->motion[mb.addr];
    mo.ref_idx[list][b8] = static_cast<int8_t>(ref);
    if (ref >= 0 && ref < sl.num_ref_idx[list] && sl.ref[list][ref].d) {
      const RefSnap& r = sl.ref[list][ref];
      mo.ref_slot[list][b8] = static_cast<int8_t>(r.slot);
      cur->ref_uid[list][mb.addr * 4 + b8] = r.uid;
    } else {
      mo.ref_slot[list][b8] = -1;
      cur->ref_uid[list][mb.addr * 4 + b8] = -1;
    }
  }

  // ------------------------------------------------------------------------
  // Motion vectors of the deferred (P) macroblocks of a whole picture, from the
  // records the parser left: differences in mv[0], reference indices, partition
  // shapes and neighbour availability (8.4.1.1, 8.4.1.3).  `motion` is written
  // in place; macroblocks are visited in raster order, so every neighbour a
  // prediction looks at is final already.  The GPU does the same in k_mv; this
  // host version serves sinks without it and co-located lookups of B pictures.
  // ------------------------------------------------------------------------
  void derive_deferred(const PicData* p, MbMotion* motion) const {
    const int nmb = mbw * mbh;
    for (int addr = 0; addr < nmb; ++addr) {
      const uint8_t* rec = &p->mvrec[static_cast<size_t>(addr) * 3];
      const int flags = rec[0];
      if (!(flags & amd::MVF_DEFERRED)) continue;
      MbMotion& mo = motion[addr];
      // neighbour sample: ref -2 unavailable, -1 intra / list unused
      auto fetch = [&](int rx, int ry, int cur_zz) -> NB {
        NB nb = {-2, {0, 0}};
        if (rx > 3 && ry >= 0) return nb;
        const int ox = rx < 0 ? -1 : (rx > 3 ? 1 : 0), oy = ry < 0 ? -1 : 0;
        const int bx = rx & 3, by = ry & 3;
        const MbMotion* m;
        if (ox == 0 && oy == 0) {
          if (raster_to_blk4[by * 4 + bx] >= cur_zz) return nb;
          m = &mo;
        } else {
          const int bit = oy == 0 ? amd::MVF_AVAIL_A : (ox < 0 ? amd::MVF_AVAIL_D : (ox == 0 ? amd::MVF_AVAIL_B : amd::MVF_AVAIL_C));
          if (!(flags & bit)) return nb;
          const int na = addr + oy * mbw + ox;
          nb.ref = -1;
          if (p->mb_intra[na]) return nb;
          m = &motion[na];
        }
        nb.ref = -1;
        const int r = m->ref_idx[0][(by >> 1) * 2 + (bx >> 1)];
        if (r < 0) return nb;
        nb.ref = r;
        nb.mv[0] = m->mv[0][by * 4 + bx][0];
        nb.mv[1] = m->mv[0][by * 4 + bx][1];
        return nb;
      };
      auto predict = [&](int bx, int by, int bw, int ref, int dir, int out[2]) {
        const int zz = raster_to_blk4[by * 4 + bx];
        NB A = fetch(bx - 1, by, zz);
        NB B = fetch(bx, by - 1, zz);
        NB C = fetch(bx + bw, by - 1, zz);
        if (C.ref == -2) C = fetch(bx - 1, by - 1, zz);
        const NB* pick = nullptr;
        if (dir == PRED_16x8_TOP && B.ref == ref) pick = &B;
        if (dir == PRED_16x8_BOTTOM && A.ref == ref) pick = &A;
        if (dir == PRED_8x16_LEFT && A.ref == ref) pick = &A;
        if (dir == PRED_8x16_RIGHT && C.ref == ref) pick = &C;
        if (!pick) {
          const int match = (A.ref == ref) + (B.ref == ref) + (C.ref == ref);
          if (match == 1) pick = A.ref == ref ? &A : (B.ref == ref ? &B : &C);
          else if (match == 0 && B.ref == -2 && C.ref == -2 && A.ref != -2) pick = &A;
        }
        if (pick) {
          out[0] = pick->mv[0];
          out[1] = pick->mv[1];
        } else {
          out[0] = median3(A.mv[0], B.mv[0], C.mv[0]);
          out[1] = median3(A.mv[1], B.mv[1], C.mv[1]);
        }
      };
      auto fill = [&](int bx, int by, int bw, int bh, int mvx, int mvy) {
        const uint32_t packed = static_cast<uint16_t>(mvx) | (static_cast<uint32_t>(static_cast<uint16_t>(mvy)) << 16);
        uint32_t* dst = reinterpret_cast<uint32_t*>(mo.mv[0]);
        for (int y = by; y < by + bh; ++y)
          for (int x = bx; x < bx + bw; ++x) dst[y * 4 + x] = packed;
      };
      if (flags & amd::MVF_PSKIP) {
        const NB A = fetch(-1, 0, 0), B = fetch(0, -1, 0);
        int mv[2] = {0, 0};
        if (!(A.ref == -2 || B.ref == -2 || (A.ref == 0 && A.mv[0] == 0 && A.mv[1] == 0) ||
              (B.ref == 0 && B.mv[0] == 0 && B.mv[1] == 0)))
          predict(0, 0, 4, 0, PRED_MEDIAN, mv);
        fill(0, 0, 4, 4, mv[0], mv[1]);
        continue;
      }
      const int shape = rec[1], sub = rec[2];
      const int nparts = shape == 0 ? 1 : (shape == 3 ? 4 : 2);
      for (int i = 0; i < nparts; ++i) {
        int px, py, pw, ph;
        if (shape == 0) { px = 0; py = 0; pw = 4; ph = 4; }
        else if (shape == 1) { px = 0; py = i * 2; pw = 4; ph = 2; }
        else if (shape == 2) { px = i * 2; py = 0; pw = 2; ph = 4; }
        else { px = (i & 1) * 2; py = (i >> 1) * 2; pw = 2; ph = 2; }
        int sw = pw, sh = ph;
        if (shape == 3) {
          const int ss = (sub >> (2 * i)) & 3;
          if (ss == 2 || ss == 3) sw = 1;
          if (ss == 1 || ss == 3) sh = 1;
        }
        const int ref = mo.ref_idx[0][(py >> 1) * 2 + (px >> 1)];
        for (int sy = 0; sy < ph; sy += sh)
          for (int sx = 0; sx < pw; sx += sw) {
            const int bx = px + sx, by = py + sy;
            int dir = PRED_MEDIAN;
            if (shape == 1) dir = i == 0 ? PRED_16x8_TOP : PRED_16x8_BOTTOM;
            if (shape == 2) dir = i == 0 ? PRED_8x16_LEFT : PRED_8x16_RIGHT;
            int mvp[2];
            predict(bx, by, sw, ref, dir, mvp);
            const int dx = mo.mv[0][by * 4 + bx][0], dy = mo.mv[0][by * 4 + bx][1];
            fill(bx, by, sw, sh, mvp[0] + dx, mvp[1] + dy);
          }
      }
    }
  }

  // Macroblocks that were handed over as PMotion + extra differences (MVF_COMPACT), put back into the
  // MbMotion form derive_deferred works on: reference indices, and the differences replicated over each
  // partition's blocks.  Same partition walk, same order as the stream was written in.
  void expand_compact(const PicData* p, MbMotion* motion) const {
    if (!p->pmo || !p->pmx) return;
    size_t x = 0;
    for (int addr = 0; addr < mbw * mbh; ++addr) {
      if (addr % mbw == 0) x = p->pmx[addr / mbw];
      const uint8_t* rec = &p->mvrec[static_cast<size_t>(addr) * 3];
      if (!(rec[0] & amd::MVF_COMPACT)) continue;
      MbMotion& mo = motion[addr];
      const amd::PMotion& pm = p->pmo[addr];
      memset(&mo, 0, sizeof(mo));
      memcpy(mo.ref_idx[0], pm.ref_idx, 4);
      memcpy(mo.ref_slot[0], pm.ref_slot, 4);
      memset(mo.ref_idx[1], -1, 4);
      memset(mo.ref_slot[1], -1, 4);
      if (!(rec[0] & amd::MVF_DEFERRED) || (rec[0] & amd::MVF_PSKIP)) continue;
      const int shape = rec[1], sub = rec[2];
      const int nparts = shape == 0 ? 1 : (shape == 3 ? 4 : 2);
      int part = 0;
      for (int i = 0; i < nparts; ++i) {
        int px, py, pw, ph;
        if (shape == 0) { px = 0; py = 0; pw = 4; ph = 4; }
        else if (shape == 1) { px = 0; py = i * 2; pw = 4; ph = 2; }
        else if (shape == 2) { px = i * 2; py = 0; pw = 2; ph = 4; }
        else { px = (i & 1) * 2; py = (i >> 1) * 2; pw = 2; ph = 2; }
        int sw = pw, sh = ph;
        if (shape == 3) {
          const int ss = (sub >> (2 * i)) & 3;
          if (ss == 2 || ss == 3) sw = 1;
          if (ss == 1 || ss == 3) sh = 1;
        }
        for (int sy = 0; sy < ph; sy += sh)
          for (int sx = 0; sx < pw; sx += sw) {
            uint32_t word;
            if (part++ == 0) word = static_cast<uint16_t>(pm.mvd0[0]) | (static_cast<uint32_t>(static_cast<uint16_t>(pm.mvd0[1])) << 16);
            else word = p->pmx[x++];
            uint32_t* dst = reinterpret_cast<uint32_t*>(mo.mv[0]);
            for (int yy = py + sy; yy < py + sy + sh; ++yy)
              for (int xx = px + sx; xx < px + sx + sw; ++xx) dst[yy * 4 + xx] = word;
          }
      }
    }
  }

  // Final motion of a (possibly deferred) picture for co-located lookups.
  // With entropy workers the co-located picture may still be in another
  // worker's hands: wait for its records (it precedes this picture in decoding
  // order, so its task started earlier and does not depend on this one); the
  // derivation runs once, under the picture's lock.
  const MbMotion* final_motion(PicData* p) {
    std::unique_lock<std::mutex> lk(p->mu);
    p->cv.wait(lk, [&] { return p->parsed; });
    if (!p->mv_deferred) return p->motion;
    if (!p->derived_valid) {
      const size_t nmb = static_cast<size_t>(mbw) * mbh;
      p->derived.assign(p->motion, p->motion + nmb);  // the surface's buffer may still be uploading: work on a copy
      expand_compact(p, p->derived.data());
      derive_deferred(p, p->derived.data());
      p->derived_valid = true;
    }
    return p->derived.data();
  }

  bool pred_direct(int only_b8) {
    PicData* col = sl.ref[1][0].d;
    if (!col || !col->motion) return fail("direct prediction without RefPicList1[0]");
    const RefSnap& colref = sl.ref[1][0];
    const int inf8 = asps->direct_8x8_inference;
    const int bsz = inf8 ? 2 : 1;  // 4x4 blocks per side that share one derivation
    const int addr = mb.addr;
    if (!col_motion) col_motion = final_motion(col);  // (once per slice: it takes the co-located picture's lock)
    const MbMotion& cm = col_motion[addr];
    if (sl.direct_spatial) {
      int ref[2], mvp[2][2] = {{0, 0}, {0, 0}};
      for (int l = 0; l < 2; ++l) {
        NB A = fetch_nb(l, -1, 0, 0), B = fetch_nb(l, 0, -1, 0), C = fetch_nb(l, 4, -1, 0);
        if (C.ref == -2) C = fetch_nb(l, -1, -1, 0);
        int r = -1;
        if (A.ref >= 0) r = A.ref;
        if (B.ref >= 0 && (r < 0 || B.ref < r)) r = B.ref;
        if (C.ref >= 0 && (r < 0 || C.ref < r)) r = C.ref;
        ref[l] = r;
        if (r >= 0) pred_mv_from(A, B, C, r, PRED_MEDIAN, mvp[l]);  // (the 16x16 partition's neighbours: the same three)
      }
      if (ref[0] < 0 && ref[1] < 0) ref[0] = ref[1] = 0;
      const bool col_short = colref.marking == 1;
      if (only_b8 < 0 && inf8) {
        // The whole macroblock (B_Skip, B_Direct_16x16) with one derivation per quadrant - most macroblocks of a B
        // picture: references and vectors are the same for the four quadrants except where the co-located corner block
        // stands still (colZeroFlag), so they are written a quadrant row at a time, not a 4x4 block at a time.
        // (Round 5, sampled on the GPU box: this function and the stores it made were 29 % of the instructions of a
        // B picture.)
        MbMotion& mo = cur->motion[addr];
        unsigned zero_q = 0;  // quadrants whose co-located block has reference 0 and a vector within +-1
        if (col_short && !col->mb_intra[addr])
          for (int b8 = 0; b8 < 4; ++b8) {
            const int cx = (b8 & 1) * 3, cy = (b8 >> 1) * 3;
            const int cl = cm.ref_idx[0][b8] >= 0 ? 0 : 1;
            if (cm.ref_idx[cl][b8] == 0) {
              const int16_t* v = cm.mv[cl][cy * 4 + cx];
              if (v[0] >= -1 && v[0] <= 1 && v[1] >= -1 && v[1] <= 1) zero_q |= 1u << b8;
            }
          }
        for (int l = 0; l < 2; ++l) {
          const int r = ref[l];
          int8_t slot = -1;
          int uid = -1;
          if (r >= 0 && r < sl.num_ref_idx[l] && sl.ref[l][r].d) {
            slot = static_cast<int8_t>(sl.ref[l][r].slot);
            uid = sl.ref[l][r].uid;
          }
          memset(mo.ref_idx[l], r, 4);
          memset(mo.ref_slot[l], slot, 4);
          int* const u = &cur->ref_uid[l][addr * 4];
          u[0] = u[1] = u[2] = u[3] = uid;
          const uint32_t packed = r < 0 ? 0u : (static_cast<uint16_t>(mvp[l][0]) | (static_cast<uint32_t>(static_cast<uint16_t>(mvp[l][1])) << 16));
          const uint64_t pair = packed | (static_cast<uint64_t>(packed) << 32);
          const unsigned zq = r == 0 ? zero_q : 0;
          uint64_t* const dst = reinterpret_cast<uint64_t*>(mo.mv[l]);  // two blocks a word, two words a row
          const uint64_t q0 = (zq & 1) ? 0 : pair, q1 = (zq & 2) ? 0 : pair, q2 = (zq & 4) ? 0 : pair, q3 = (zq & 8) ? 0 : pair;
          dst[0] = dst[2] = q0;
          dst[1] = dst[3] = q1;
          dst[4] = dst[6] = q2;
          dst[5] = dst[7] = q3;
        }
        return true;
      }
      for (int b8 = 0; b8 < 4; ++b8) {
        if (only_b8 >= 0 && b8 != only_b8) continue;
        for (int l = 0; l < 2; ++l) set_ref8(l, b8, ref[l]);
        // with direct_8x8_inference the four 4x4 blocks of a quadrant share the
        // co-located corner block: one evaluation, one 2x2 fill
        for (int s = 0; s < (inf8 ? 1 : 4); ++s) {
          const int bx = (b8 & 1) * 2 + (s & 1), by = (b8 >> 1) * 2 + (s >> 1);
          const int cx = inf8 ? (b8 & 1) * 3 : bx, cy = inf8 ? (b8 >> 1) * 3 : by;
          bool col_zero = false;
          if (col_short && !col->mb_intra[addr]) {
            const int c8 = (cy >> 1) * 2 + (cx >> 1);
            const int cl = cm.ref_idx[0][c8] >= 0 ? 0 : 1;
            if (cm.ref_idx[cl][c8] == 0) {
              const int16_t* v = cm.mv[cl][cy * 4 + cx];
              col_zero = v[0] >= -1 && v[0] <= 1 && v[1] >= -1 && v[1] <= 1;
            }
          }
          for (int l = 0; l < 2; ++l) {
            bool zero = ref[l] < 0 || (ref[l] == 0 && col_zero);
            set_mv_block(l, bx, by, bsz, bsz, zero ? 0 : mvp[l][0], zero ? 0 : mvp[l][1]);
          }
        }
      }
      return true;
    }
    for (int b8 = 0; b8 < 4; ++b8) {
      if (only_b8 >= 0 && b8 != only_b8) continue;
      for (int s = 0; s < (inf8 ? 1 : 4); ++s) {
        const int bx = (b8 & 1) * 2 + (s & 1), by = (b8 >> 1) * 2 + (s >> 1);
        const int cx = inf8 ? (b8 & 1) * 3 : bx, cy = inf8 ? (b8 >> 1) * 3 : by;
        int ref0 = 0, mvc[2] = {0, 0};
        if (!col->mb_intra[addr]) {
          const int c8 = (cy >> 1) * 2 + (cx >> 1);
          const int cl = cm.ref_idx[0][c8] >= 0 ? 0 : 1;
          if (cm.ref_idx[cl][c8] >= 0) {
            mvc[0] = cm.mv[cl][cy * 4 + cx][0];
            mvc[1] = cm.mv[cl][cy * 4 + cx][1];
            const int uid = col->ref_uid[cl][addr * 4 + c8];
            ref0 = -1;
            for (int i = 0; i < sl.num_ref_idx[0]; ++i)
              if (sl.ref[0][i].d && sl.ref[0][i].uid == uid) {
                ref0 = i;
                break;
              }
            if (ref0 < 0) return fail("temporal direct: co-located reference not in RefPicList0");
          }
        }
        const RefSnap& p0 = sl.ref[0][ref0];
        int mv0[2], mv1[2];
        const int td = iclip(-128, 127, colref.poc - p0.poc);
        if (p0.marking == 2 || td == 0) {
          mv0[0] = mvc[0];
          mv0[1] = mvc[1];
          mv1[0] = mv1[1] = 0;
        } else {
          const int tb = iclip(-128, 127, sl.cur_poc - p0.poc);
          const int tx = (16384 + iabs(td / 2)) / td;
          const int dsf = iclip(-1024, 1023, (tb * tx + 32) >> 6);
          mv0[0] = (dsf * mvc[0] + 128) >> 8;
          mv0[1] = (dsf * mvc[1] + 128) >> 8;
          mv1[0] = mv0[0] - mvc[0];
          mv1[1] = mv0[1] - mvc[1];
        }
        if (s == 0) {
          set_ref8(0, b8, ref0);
          set_ref8(1, b8, 0);
        }
        set_mv_block(0, bx, by, bsz, bsz, mv0[0], mv0[1]);
        set_mv_block(1, bx, by, bsz, bsz, mv1[0], mv1[1]);
      }
    }
    return true;
  }

  // ------------------------------------------------------------- residual
  int read_vlc(const uint8_t* len, const uint8_t* code, int n) {
    const unsigned window = bits.peek(16);
    for (int i = 0; i < n; ++i) {
      if (!len[i]) continue;
      if ((window >> (16 - len[i])) == code[i]) {
        if (bits.left() < len[i]) break;
        bits.pos += len[i];
        return i;
      }
    }
    bits.overrun = true;
    return -1;
  }
  int cavlc_nc(int plane, int rx, int ry) const {
    const int n = plane == 0 ? 4 : 2;
    const int base = plane == 0 ? 0 : (plane == 1 ? 16 : 20);
    int na = -1, nb = -1;
    if (rx > 0) na = mb.nnz[base + ry * n + rx - 1];
    else if (mb.avA) na = nnz[(mb.addr - 1) * 24 + base + ry * n + n - 1];
    if (ry > 0) nb = mb.nnz[base + (ry - 1) * n + rx];
    else if (mb.avB) nb = nnz[(mb.addr - mbw) * 24 + base + (n - 1) * n + rx];
    if (na >= 0 && nb >= 0) return (na + nb + 1) >> 1;
    if (na >= 0) return na;
    if (nb >= 0) return nb;
    return 0;
  }
  int cavlc_block(int nc, int max_coeff, const uint8_t* scan, int start, int16_t* out) {
    Bits& b = bits;
    int token;
    if (nc == -1) {
      token = read_vlc(chroma_dc_coeff_token_len, chroma_dc_coeff_token_bits, 20);
    } else {
      const int t = nc < 2 ? 0 : (nc < 4 ? 1 : (nc < 8 ? 2 : 3));
      token = read_vlc(coeff_token_len[t], coeff_token_bits[t], 68);
    }
    if (token < 0) return fail("CAVLC: bad coeff_token") ? 0 : -1;
    const int total = token >> 2, t1 = token & 3;
    if (total == 0) return 0;
    if (total > max_coeff) return fail("CAVLC: total_coeff exceeds block size") ? 0 : -1;
    int level[16];
    for (int i = 0; i < t1; ++i) level[i] = b.bit() ? -1 : 1;
    int suffix_len = (total > 10 && t1 < 3) ? 1 : 0;
    for (int i = t1; i < total; ++i) {
      int prefix = 0;
      while (!b.bit()) {
        if (++prefix > 31 || b.overrun) return fail("CAVLC: bad level_prefix") ? 0 : -1;
      }
      int code = imin(15, prefix) << suffix_len;
      int size = suffix_len;
      if (prefix == 14 && suffix_len == 0) size = 4;
      if (prefix >= 15) size = prefix - 3;
      if (size > 0) code += b.get(size);
      if (prefix >= 15 && suffix_len == 0) code += 15;
      if (prefix >= 16) code += (1 << (prefix - 3)) - 4096;
      if (i == t1 && t1 < 3) code += 2;
      level[i] = (code & 1) ? (-code - 1) >> 1 : (code + 2) >> 1;
      if (suffix_len == 0) suffix_len = 1;
      if (iabs(level[i]) > (3 << (suffix_len - 1)) && suffix_len < 6) ++suffix_len;
    }
    int zeros_left = 0;
    if (total < max_coeff) {
      int tz = nc == -1 ? read_vlc(chroma_dc_total_zeros_len[total - 1], chroma_dc_total_zeros_bits[total - 1], 4)
                        : read_vlc(total_zeros_len[total - 1], total_zeros_bits[total - 1],
                                   total_zeros_count[total - 1]);
      if (tz < 0) return fail("CAVLC: bad total_zeros") ? 0 : -1;
      zeros_left = tz;
    }
    if (zeros_left + total > max_coeff) return fail("CAVLC: total_zeros exceeds block size") ? 0 : -1;
    int pos = zeros_left + total - 1;
    for (int i = 0; i < total; ++i) {
      out[scan[start + pos]] = static_cast<int16_t>(level[i]);
      if (i == total - 1) break;
      int run = 0;
      if (zeros_left > 0) {
        const int t = imin(zeros_left, 7) - 1;
        run = read_vlc(run_before_len[t], run_before_bits[t], run_before_count[t]);
        if (run < 0 || run > zeros_left) return fail("CAVLC: bad run_before") ? 0 : -1;
        zeros_left -= run;
      }
      pos -= run + 1;
    }
    return total;
  }

  // The residual of one macroblock (7.3.5.3 with 9.3.2.x / 7.3.5.3.3), CABAC.  Most of a picture's bins are decoded
  // here, so the arithmetic decoder's registers (offset, range, bit budget, read pointer) live in locals for ALL the
  // blocks of the macroblock - up to 27 of them, many with a single coded_block_flag bin and nothing behind it.  The
  // block reader is instantiated per ctxBlockCat (its bounds, scan and context bases are constants then) and the callers
  // are plain loops per kind of block: a block's coded_block_flag context is two loads from the flag cache
  // (fill_nz_cache) and what it leaves behind two stores and an OR into a register.  (Round 5, counted on the GPU
  // box: a list of blocks around ONE copy of the reader, with a switch before and after every block, spent ~170
  // instructions per block outside the reader - a quarter of an intra picture's instructions.)  The context words
  // are 16-bit on purpose (see Cabac::state).
  // Non-zero levels are appended to mb.ent with flat index `gbase` + raster position (gbase = group * 16).
  template <int N> struct Cat { static constexpr int value = N; };
  static constexpr int nzc_luma(int which) { return 9 + (which >> 2) * 8 + (which & 3); }
  static constexpr int nzc_chroma(int p, int s) { return (1 + 3 * p + (s >> 1)) * 8 + 6 + (s & 1); }
  bool cabac_residual() {
    REGION(R_RES_FILL);
    fill_nz_cache();
    REGION(R_RESIDUAL);
    const bool i16 = mb.kind == MB_I16x16;

    Cabac& c = cabac;
    uint64_t value = c.value;
    uint32_t range = c.range;
    int extra = c.extra;
    const uint8_t* p = c.p;
    const uint8_t* const end = c.end;
    uint16_t* const state = c.state;
    bool overrun = false;
    // The refill is the cold path and stays out of line; it works on copies, so
    // that taking their addresses does not pin the hot variables (value, extra,
    // p) to the stack - a by-reference capture in an un-inlined lambda did, and
    // every bin then went through a store and a reload of the offset.
    auto refill = [&]() __attribute__((always_inline)) {
      int e = extra;
      const uint8_t* q = p;
      bool ov = false;
      value = cabac_refill_slow(value, &e, &q, end, &ov);
      extra = e;
      p = q;
      overrun |= ov;
    };
    auto bin = [&](int ctx) __attribute__((always_inline)) -> int {
      CABAC_TRACE(ctx);
      const int b = cabac_step(state, ctx, value, range, extra);
      if (__builtin_expect(extra < 8, 0)) refill();
      return b;
    };
    auto bypass = [&]() __attribute__((always_inline)) -> int {
      CABAC_TRACE(0xffff);
      extra -= 1;
      if (__builtin_expect(extra < 8, 0)) refill();
      const uint64_t scaled = static_cast<uint64_t>(range) << extra;
      const uint64_t mask = ~static_cast<uint64_t>(static_cast<int64_t>(value - scaled) >> 63);
      value -= scaled & mask;
      return static_cast<int>(mask & 1);
    };
    int nent = mb.nent;
    uint32_t coded = mb.coded_mask, nzm = 0;
    auto leave = [&]() {
      c.value = value;
      c.range = range;
      c.extra = extra;
      c.p = p;
      if (overrun) c.overrun = true;
      mb.nent = nent;
      mb.coded_mask = coded;
      mb.nz_mask = static_cast<uint16_t>(nzm);
    };

    // One block of category `cat` whose coded_block_flag context increment is `inc`: the number of its non-zero
    // levels, -1 for a level that does not end.
    auto block = [&](auto cat_c, int inc, int gbase) __attribute__((always_inline)) -> int {
      constexpr int cat = decltype(cat_c)::value;
      constexpr int max_coeff = cat == 5 ? 64 : cat == 3 ? 4 : (cat == 1 || cat == 4) ? 15 : 16;
      constexpr int start = (cat == 1 || cat == 4) ? 1 : 0;
      constexpr int cbf_ctx = 85 + (cat == 1 ? 4 : cat == 2 ? 8 : cat == 3 ? 12 : cat == 4 ? 16 : 0);
      constexpr int sig_off = cat == 1 ? 15 : cat == 2 ? 29 : cat == 3 ? 44 : cat == 4 ? 47 : 0;
      constexpr int abs_off = cat == 1 ? 10 : cat == 2 ? 20 : cat == 3 ? 30 : cat == 4 ? 39 : 0;
      static_assert(cat >= 0 && cat <= 5, "ctxBlockCat");
      if (cat != 5 && !bin(cbf_ctx + inc)) return 0;  // (an 8x8 block has no flag of its own outside 4:4:4)
      constexpr int sig_base = cat == 5 ? 402 : 105 + sig_off;
      constexpr int last_base = cat == 5 ? 417 : 166 + sig_off;
      constexpr int abs_base = cat == 5 ? 426 : 227 + abs_off;
      const uint8_t* const scan = cat == 5 ? zigzag8x8 : cat == 3 ? dc_scan4 : zigzag4x4;
      uint8_t idx[64];
      int n = 0, i;
      constexpr int lastpos = max_coeff - 1;
      if (cat == 5) {
        for (i = 0; i < lastpos; ++i)
          if (bin(sig_base + sig_coeff_offset_8x8[i])) {
            idx[n++] = static_cast<uint8_t>(i);
            if (bin(last_base + last_coeff_offset_8x8[i])) break;
          }
      } else if (cat == 3) {
        for (i = 0; i < lastpos; ++i) {
          const int ci = imin(i, 2);
          if (bin(sig_base + ci)) {
            idx[n++] = static_cast<uint8_t>(i);
            if (bin(last_base + ci)) break;
          }
        }
      } else {
        for (i = 0; i < lastpos; ++i)
          if (bin(sig_base + i)) {
            idx[n++] = static_cast<uint8_t>(i);
            if (bin(last_base + i)) break;
          }
      }
      if (i == lastpos) idx[n++] = static_cast<uint8_t>(lastpos);
      int gt1 = 0, eq1 = 0;
      constexpr int gt_cap = 4 - (cat == 3 ? 1 : 0);
      uint32_t* const ent = mb.ent + nent;  // n <= 64 entries; a macroblock never has more than 26 * 16
      for (int k = n - 1; k >= 0; --k) {
        const int inc0 = gt1 ? 0 : imin(4, 1 + eq1);
        int val;
        if (!bin(abs_base + inc0)) {
          val = 1;
          ++eq1;
        } else {
          const int ainc = 5 + imin(gt_cap, gt1);
          val = 2;
          while (val < 15 && bin(abs_base + ainc)) ++val;
          if (val == 15) {
            int k0 = 0;
            while (bypass()) {
              val += 1 << k0;
              if (++k0 > 24) return -1;
            }
            while (k0--) val += bypass() << k0;
          }
          ++gt1;
        }
        if (bypass()) val = -val;
        ent[k] = (static_cast<uint32_t>(gbase + scan[start + idx[k]]) << 16) | static_cast<uint16_t>(val);
      }
      nent += n;
      return n;
    };
    auto overflow = [&]() {
      leave();
      return fail("CABAC: coeff_abs_level overflow");
    };
    // the DC blocks' increments: the neighbour macroblocks' flags (bit 0 Intra16x16 DC, bits 1 / 2 Cb / Cr DC)
    auto dc_inc = [&](int bit) { return ((mb.dc_left >> bit) & 1) + 2 * ((mb.dc_top >> bit) & 1); };
    uint8_t* const nzc = mb.nzc;

    if (i16) {
      const int n = block(Cat<0>(), dc_inc(0), 16 * amd::kMaskLumaDc);
      if (__builtin_expect(n < 0, 0)) return overflow();
      if (n) {
        coded |= 1u << amd::kMaskLumaDc;
        mb.cbf_dc |= 1;
      }
    }
    if (mb.cbp & 15) {
      if (mb.t8x8) {
        for (int b8 = 0; b8 < 4; ++b8) {
          if (!((mb.cbp >> b8) & 1)) continue;
          const int n = block(Cat<5>(), 0, 64 * b8);
          if (__builtin_expect(n < 0, 0)) return overflow();
          const int w0 = (b8 >> 1) * 8 + (b8 & 1) * 2, pos = nzc_luma(w0);
          const uint8_t cnt = static_cast<uint8_t>(n), f = n != 0;
          mb.nnz[w0] = mb.nnz[w0 + 1] = mb.nnz[w0 + 4] = mb.nnz[w0 + 5] = cnt;
          nzc[pos] = nzc[pos + 1] = nzc[pos + 8] = nzc[pos + 9] = f;
          if (n) {
            coded |= 0xfu << (4 * b8);
            nzm |= 0x33u << w0;
          }
        }
      } else {
        auto luma4 = [&](auto cat_c) __attribute__((always_inline)) -> bool {
          for (int b8 = 0; b8 < 4; ++b8) {
            if (!((mb.cbp >> b8) & 1)) continue;
            for (int s = 0; s < 4; ++s) {
              const int which = ((b8 >> 1) * 2 + (s >> 1)) * 4 + (b8 & 1) * 2 + (s & 1), pos = nzc_luma(which);
              const int n = block(cat_c, nzc[pos - 1] + 2 * nzc[pos - 8], 16 * which);
              if (__builtin_expect(n < 0, 0)) return false;
              mb.nnz[which] = static_cast<uint8_t>(n);
              nzc[pos] = n != 0;
              coded |= static_cast<uint32_t>(n != 0) << which;
              nzm |= static_cast<uint32_t>(n != 0) << which;
            }
          }
          return true;
        };
        if (!(i16 ? luma4(Cat<1>()) : luma4(Cat<2>()))) return overflow();
      }
    }
    const int cc = mb.cbp >> 4;
    if (cc) {
      for (int pl = 0; pl < 2; ++pl) {
        const int n = block(Cat<3>(), dc_inc(1 + pl), 16 * amd::kMaskChromaDc + 4 * pl);
        if (__builtin_expect(n < 0, 0)) return overflow();
        if (n) {
          coded |= 1u << amd::kMaskChromaDc;
          mb.cbf_dc |= 2 << pl;
        }
      }
      if (cc == 2)
        for (int k = 0; k < 8; ++k) {
          const int pos = nzc_chroma(k >> 2, k & 3);
          const int n = block(Cat<4>(), nzc[pos - 1] + 2 * nzc[pos - 8], 16 * (amd::kMaskChromaAc + k));
          if (__builtin_expect(n < 0, 0)) return overflow();
          mb.nnz[16 + k] = static_cast<uint8_t>(n);
          nzc[pos] = n != 0;
          coded |= static_cast<uint32_t>(n != 0) << (amd::kMaskChromaAc + k);
        }
    }
    leave();
    return true;
  }

  // coded_block_flag context inputs (9.3.3.1.1.9) of the macroblock's blocks, one byte each, 0 / 1: a block's
  // increment is the flag to its left + twice the flag above it, both found by position - luma block (x, y) at 9 + 8 y +
  // x, its left neighbour one before it and the one above it eight before it; the Cb blocks in columns 6-7 of rows 1-2
  // and the Cr blocks of rows 4-5 likewise (nzc_luma / nzc_chroma).  Filled here, once per macroblock that has a
  // residual: the macroblock's own positions 0, and along the edges the flags of the neighbour macroblocks' blocks -
  // "unavailable" counts as coded for intra macroblocks, I_PCM neighbours as coded - then kept by the block loops.
  void fill_nz_cache() {
    const uint8_t unavailable = mb.kind != MB_INTER;
    const int a = mb.addr;
    // DC blocks: the same rules; the Intra16x16 DC flag of a neighbour that is not Intra16x16 is 0
    auto dc_flags = [&](bool avail, int na) -> uint8_t {
      if (!avail) return unavailable ? 7 : 0;
      const int k = mb_kind[na];
      if (k == MB_IPCM) return 7;
      const int f = mb_cbf_dc[na];
      return static_cast<uint8_t>((f & 6) | (k == MB_I16x16 ? (f & 1) : 0));
    };
    mb.dc_left = dc_flags(mb.av[3], a - 1);
    mb.dc_top = dc_flags(mb.av[1], a - mbw);
    uint8_t* const z = mb.nzc;
    memset(z, 0, sizeof(mb.nzc));
    if (mb.av[3]) {
      const uint8_t* n = &nnz[(a - 1) * 24];
      const bool pcm = mb_kind[a - 1] == MB_IPCM;
      for (int y = 0; y < 4; ++y) z[8 * (y + 1)] = pcm || n[y * 4 + 3] != 0;
      for (int pl = 0; pl < 2; ++pl)
        for (int y = 0; y < 2; ++y) z[nzc_chroma(pl, 2 * y) - 1] = pcm || n[16 + 4 * pl + y * 2 + 1] != 0;
    } else {
      for (int y = 0; y < 4; ++y) z[8 * (y + 1)] = unavailable;
      for (int pl = 0; pl < 2; ++pl)
        for (int y = 0; y < 2; ++y) z[nzc_chroma(pl, 2 * y) - 1] = unavailable;
    }
    if (mb.av[1]) {
      const uint8_t* n = &nnz[(a - mbw) * 24];
      const bool pcm = mb_kind[a - mbw] == MB_IPCM;
      for (int x = 0; x < 4; ++x) z[1 + x] = pcm || n[12 + x] != 0;
      for (int pl = 0; pl < 2; ++pl)
        for (int x = 0; x < 2; ++x) z[nzc_chroma(pl, x) - 8] = pcm || n[16 + 4 * pl + 2 + x] != 0;
    } else {
      for (int x = 0; x < 4; ++x) z[1 + x] = unavailable;
      for (int pl = 0; pl < 2; ++pl)
        for (int x = 0; x < 2; ++x) z[nzc_chroma(pl, x) - 8] = unavailable;
    }
  }

  // ---------------------------------------------------- syntax elements
  int rd_intra_mb_type(int ctx_base, int intra_slice) {
    CabacLocal c(cabac);
    if (intra_slice) {
      int inc = 0;
      if (mb.avA) {
        int k = mb_kind[mb.addr - 1];
        inc += k != MB_I4x4 && k != MB_I8x8;
      }
      if (mb.avB) {
        int k = mb_kind[mb.addr - mbw];
        inc += k != MB_I4x4 && k != MB_I8x8;
      }
      if (!c.bin(ctx_base + inc)) return 0;
      ctx_base += 2;
    } else if (!c.bin(ctx_base)) {
      return 0;
    }
    if (c.terminate()) return 25;
    int t = 1;
    t += 12 * c.bin(ctx_base + 1);
    if (c.bin(ctx_base + 2)) t += 4 + 4 * c.bin(ctx_base + 2 + intra_slice);
    t += 2 * c.bin(ctx_base + 3 + intra_slice);
    t += c.bin(ctx_base + 3 + 2 * intra_slice);
    return t;
  }
  int rd_mb_type() {
    const int st = sl.type;
    if (!use_cabac) return static_cast<int>(bits.ue());
    Cabac& c = cabac;
    if (st == 2) return rd_intra_mb_type(3, 1);
    if (st == 0) {
      {
        CabacLocal l(cabac);  // (up to three bins: the engine in registers)
        if (!l.bin(14)) {
          if (!l.bin(15)) return 3 * l.bin(16);
          return 2 - l.bin(17);
        }
      }
      return 5 + rd_intra_mb_type(17, 0);
    }
    int inc = 0;
    if (mb.avA && !mb_direct[mb.addr - 1]) inc++;
    if (mb.avB && !mb_direct[mb.addr - mbw]) inc++;
    if (!c.bin(27 + inc)) return 0;
    if (!c.bin(27 + 3)) return 1 + c.bin(27 + 5);
    int v = c.bin(27 + 4) << 3;
    v |= c.bin(27 + 5) << 2;
    v |= c.bin(27 + 5) << 1;
    v |= c.bin(27 + 5);
    if (v < 8) return v + 3;
    if (v == 13) return 23 + rd_intra_mb_type(32, 0);
    if (v == 14) return 11;
    if (v == 15) return 22;
    v = (v << 1) | c.bin(27 + 5);
    return v - 4;
  }
  int rd_sub_mb_type() {
    if (!use_cabac) return static_cast<int>(bits.ue());
    CabacLocal c(cabac);
    if (sl.type == 0) {
      if (c.bin(21)) return 0;
      if (!c.bin(22)) return 1;
      return c.bin(23) ? 2 : 3;
    }
    if (!c.bin(36)) return 0;
    if (!c.bin(37)) return 1 + c.bin(39);
    int t = 3;
    if (c.bin(38)) {
      if (c.bin(39)) return 11 + c.bin(39);
      t += 4;
    }
    t += 2 * c.bin(39);
    t += c.bin(39);
    return t;
  }
  int rd_ref_idx(int list, int b8) {
    const int n = sl.num_ref_idx[list];
    if (n <= 1) return 0;
    if (!use_cabac) return n == 2 ? !bits.bit() : static_cast<int>(bits.ue());
    const int bx = (b8 & 1) * 2, by = (b8 >> 1) * 2;
    int inc = 0;
    for (int k = 0; k < 2; ++k) {
      const int rx = k == 0 ? bx - 1 : bx, ry = k == 0 ? by : by - 1;
      const int nx = mb.mbx + (rx < 0 ? -1 : 0), ny = mb.mby + (ry < 0 ? -1 : 0);
      const bool inside = rx >= 0 && ry >= 0;
      if (!inside && !nb_avail(nx - mb.mbx, ny - mb.mby)) continue;
      const int addr = ny * mbw + nx;
      const int q = ((ry & 3) >> 1) * 2 + ((rx & 3) >> 1);
      if (cur->mb_intra[addr]) continue;
      if (sl.type == 1 && direct8[addr * 4 + q]) continue;
      if (cur->motion[addr].ref_idx[list][q] > 0) inc += k == 0 ? 1 : 2;
    }
    int ref = 0;
    CabacLocal c(cabac);
    while (c.bin(54 + inc)) {
      ++ref;
      inc = (inc >> 2) + 4;
      if (ref >= 32) return -1;
    }
    return ref;
  }
  // mvd_l0/l1 of one partition, both components (7.3.5.1 order: x then y): the
  // two neighbours are looked up once for both (9.3.3.1.1.7).
  void rd_mvd_pair(int list, int bx, int by, int* dx, int* dy) {
    if (!use_cabac) {
      *dx = bits.se();
      *dy = bits.se();
      return;
    }
    int sum[2] = {0, 0};
    if (bx > 0) {
      const uint8_t* m = mvd_at(mb.addr, list, by * 4 + bx - 1);
      sum[0] += m[0];
      sum[1] += m[1];
    } else if (mb.av[3]) {  // left macroblock
      const uint8_t* m = mvd_at(mb.addr - 1, list, by * 4 + 3);
      sum[0] += m[0];
      sum[1] += m[1];
    }
    if (by > 0) {
      const uint8_t* m = mvd_at(mb.addr, list, (by - 1) * 4 + bx);
      sum[0] += m[0];
      sum[1] += m[1];
    } else if (mb.av[1]) {  // macroblock above
      const uint8_t* m = mvd_at(mb.addr - mbw, list, 12 + bx);
      sum[0] += m[0];
      sum[1] += m[1];
    }
    CabacLocal c(cabac);
    int out[2];
    for (int comp = 0; comp < 2; ++comp) {
      const int base = comp == 0 ? 40 : 47;
      const int inc = sum[comp] < 3 ? 0 : (sum[comp] > 32 ? 2 : 1);
      int v = 0;
      if (c.bin(base + inc)) {
        v = 1;
        int ctx = base + 3;
        while (v < 9 && c.bin(ctx)) {
          if (v < 4) ++ctx;
          ++v;
        }
        if (v >= 9) {
          int k = 3;
          while (c.bypass()) {
            v += 1 << k;
            if (++k > 24) { v = 0; break; }
          }
          if (k <= 24) while (k--) v += c.bypass() << k;
        }
        if (v && c.bypass()) v = -v;
      }
      out[comp] = v;
    }
    *dx = out[0];
    *dy = out[1];
  }
  int rd_cbp(bool intra) {
    if (!use_cabac) {
      unsigned code = bits.ue();
      if (code > 47) return -1;
      return intra ? cbp_intra_from_code[code] : cbp_inter_from_code[code];
    }
    CabacLocal c(cabac);
    const bool la = mb.avA, ta = mb.avB;
    const int cbp_a = la ? mb_cbp[mb.addr - 1] : 0x0f;
    const int cbp_b = ta ? mb_cbp[mb.addr - mbw] : 0x0f;
    int cbp = 0;
    for (int b8 = 0; b8 < 4; ++b8) {
      int bit_a = (b8 & 1) ? (cbp >> (b8 - 1)) & 1 : (cbp_a >> (b8 + 1)) & 1;
      int bit_b = (b8 & 2) ? (cbp >> (b8 - 2)) & 1 : (cbp_b >> (b8 + 2)) & 1;
      cbp |= c.bin(73 + !bit_a + 2 * !bit_b) << b8;
    }
    const int ca = la ? mb_cbp[mb.addr - 1] >> 4 : 0;
    const int cb = ta ? mb_cbp[mb.addr - mbw] >> 4 : 0;
    if (c.bin(77 + (ca != 0) + 2 * (cb != 0))) cbp |= (1 + c.bin(77 + 4 + (ca == 2) + 2 * (cb == 2))) << 4;
    return cbp;
  }
  int rd_qp_delta() {
    if (!use_cabac) return bits.se();
    int ctx = last_qp_delta_nonzero ? 1 : 0, v = 0;
    CabacLocal c(cabac);
    while (c.bin(60 + ctx)) {
      ctx = 2 + (ctx >> 1);
      if (++v > 104) return 1000;
    }
    return (v & 1) ? (v + 1) >> 1 : -((v + 1) >> 1);
  }
  int rd_chroma_pred_mode() {
    if (!use_cabac) return static_cast<int>(bits.ue());
    int inc = 0;
    if (mb.avA && mb_chroma_mode[mb.addr - 1]) inc++;
    if (mb.avB && mb_chroma_mode[mb.addr - mbw]) inc++;
    CabacLocal c(cabac);
    if (!c.bin(64 + inc)) return 0;
    if (!c.bin(64 + 3)) return 1;
    return 2 + c.bin(64 + 3);
  }
  int rd_t8x8_flag() {
    if (!use_cabac) return bits.bit();
    int inc = 0;
    if (mb.avA && mb_t8x8[mb.addr - 1]) inc++;
    if (mb.avB && mb_t8x8[mb.addr - mbw]) inc++;
    return cabac.bin(399 + inc);
  }
  int rd_intra_pred_mode() {
    if (!use_cabac) return bits.bit() ? -1 : static_cast<int>(bits.get(3));
    CabacLocal c(cabac);
    if (c.bin(68)) return -1;
    int v = c.bin(69);
    v |= c.bin(69) << 1;
    v |= c.bin(69) << 2;
    return v;
  }
  int predicted_intra_mode(int bx, int by) const {
    int mode[2];
    for (int k = 0; k < 2; ++k) {
      const int rx = k == 0 ? bx - 1 : bx, ry = k == 0 ? by : by - 1;
      if (rx >= 0 && ry >= 0) {
        mode[k] = mb.i4_mode[ry * 4 + rx];
        continue;
      }
      const int nx = mb.mbx + (rx < 0 ? -1 : 0), ny = mb.mby + (ry < 0 ? -1 : 0);
      if (!nb_avail(nx - mb.mbx, ny - mb.mby)) return 2;
      const int addr = ny * mbw + nx;
      const int kind = mb_kind[addr];
      if (kind == MB_INTER && apps->constrained_intra_pred) return 2;
      mode[k] = (kind != MB_I4x4 && kind != MB_I8x8) ? 2 : i4mode[addr * 16 + (ry & 3) * 4 + (rx & 3)];
    }
    return imin(mode[0], mode[1]);
  }

  // CAVLC staging -> level entries of group `g` (positions offset .. offset + n - 1); marks the group coded.
  // `always`: mark it even when all n levels are zero (the four groups of an 8x8 block are coded together).
  void emit_levels(int g, int offset, const int16_t* lv, int n, bool always = false) {
    bool any = always;
    for (int k = 0; k < n; ++k)
      if (lv[k]) {
        mb.ent[mb.nent++] = (static_cast<uint32_t>(g * 16 + offset + k) << 16) | static_cast<uint16_t>(lv[k]);
        any = true;
      }
    if (any) mb.coded_mask |= 1u << g;
  }

  bool parse_residual() {
    if (use_cabac) return cabac_residual();
    const bool i16 = mb.kind == MB_I16x16;
    if (i16) {
      int16_t tmp[16] = {0};
      const int n = cavlc_block(cavlc_nc(0, 0, 0), 16, zigzag4x4, 0, tmp);
      if (n < 0) return false;
      if (n > 0) emit_levels(amd::kMaskLumaDc, 0, tmp, 16);
      if (n) mb.cbf_dc |= 1;
    }
    for (int b8 = 0; b8 < 4; ++b8) {
      if (!((mb.cbp >> b8) & 1)) continue;
      if (mb.t8x8) {
        // 64 coefficients in raster 8x8 order spread over groups 4*b8..4*b8+3
        int16_t blk[64];
        memset(blk, 0, sizeof(blk));
        int total = 0;
        for (int s = 0; s < 4; ++s) {
          const int bx = (b8 & 1) * 2 + (s & 1), by = (b8 >> 1) * 2 + (s >> 1);
          int16_t tmp[16] = {0};
          int n = cavlc_block(cavlc_nc(0, bx, by), 16, identity_scan16, 0, tmp);
          if (n < 0) return false;
          for (int k = 0; k < 16; ++k)
            if (tmp[k]) blk[zigzag8x8[4 * k + s]] = tmp[k];
          mb.nnz[by * 4 + bx] = static_cast<uint8_t>(n);
          total += n;
        }
        if (total) {
          for (int g = 0; g < 4; ++g) emit_levels(4 * b8 + g, 0, blk + 16 * g, 16, true);
        }
        continue;
      }
      for (int s = 0; s < 4; ++s) {
        const int bx = (b8 & 1) * 2 + (s & 1), by = (b8 >> 1) * 2 + (s >> 1);
        const int ri = by * 4 + bx;
        int16_t tmp[16] = {0};
        const int n = i16 ? cavlc_block(cavlc_nc(0, bx, by), 15, zigzag4x4, 1, tmp) : cavlc_block(cavlc_nc(0, bx, by), 16, zigzag4x4, 0, tmp);
        if (n < 0) return false;
        if (n > 0) emit_levels(ri, 0, tmp, 16);
        mb.nnz[ri] = static_cast<uint8_t>(n);
      }
    }
    const int cc = mb.cbp >> 4;
    if (cc) {
      for (int p = 0; p < 2; ++p) {
        int16_t tmp[4] = {0, 0, 0, 0};
        const int n = cavlc_block(-1, 4, dc_scan4, 0, tmp);
        if (n < 0) return false;
        if (n > 0) emit_levels(amd::kMaskChromaDc, 4 * p, tmp, 4);
        if (n) mb.cbf_dc |= 2 << p;
      }
    }
    if (cc == 2) {
      for (int p = 0; p < 2; ++p)
        for (int s = 0; s < 4; ++s) {
          const int bx = s & 1, by = s >> 1;
          const int g = amd::kMaskChromaAc + 4 * p + s;
          int16_t tmp[16] = {0};
          const int n = cavlc_block(cavlc_nc(1 + p, bx, by), 15, zigzag4x4, 1, tmp);
          if (n < 0) return false;
          if (n > 0) emit_levels(g, 0, tmp, 16);
          mb.nnz[16 + 4 * p + s] = static_cast<uint8_t>(n);
        }
    }
    if (bits.overrun) return fail("residual ran past the end of the slice");
    return true;
  }

  bool parse_inter(int mb_type) {
    const bool is_b = sl.type == 1;
    int shape, pred[4], sub_shape[4] = {0, 0, 0, 0};
    if (!is_b) {
      shape = mb_type == 0 ? 0 : (mb_type == 1 ? 1 : (mb_type == 2 ? 2 : 3));
      pred[0] = pred[1] = pred[2] = pred[3] = 0;
    } else {
      shape = b_mb_info[mb_type][0];
      pred[0] = b_mb_info[mb_type][1];
      pred[1] = b_mb_info[mb_type][2];
    }
    const int nparts = shape == 0 ? 1 : (shape == 3 ? 4 : 2);
    if (shape == 3) {
      for (int i = 0; i < 4; ++i) {
        int t = rd_sub_mb_type();
        if (t < 0 || t > (is_b ? 12 : 3)) return fail("bad sub_mb_type");
        mb.sub_type[i] = t;
        if (is_b) {
          sub_shape[i] = b_sub_info[t][0];
          pred[i] = b_sub_info[t][1];
        } else {
          sub_shape[i] = t;
          pred[i] = 0;
        }
      }
      if (is_b)
        for (int i = 0; i < 4; ++i)
          if (pred[i] == 3) {
            direct8[mb.addr * 4 + i] = 1;
            if (!pred_direct(i)) return false;
          }
    }
    for (int l = 0; l < (is_b ? 2 : 1); ++l)
      for (int i = 0; i < nparts; ++i) {
        if (pred[i] == 3) continue;
        int ref = -1;
        const int b8 = shape == 0 ? 0 : (shape == 1 ? i * 2 : i);
        if (uses_list(pred[i], l)) {
          if (!is_b && mb_type == 4) {
            ref = 0;
          } else {
            ref = rd_ref_idx(l, b8);
            if (ref < 0 || ref >= sl.num_ref_idx[l]) return fail("ref_idx out of range");
          }
        }
        if (shape == 0) {
          for (int q = 0; q < 4; ++q) set_ref8(l, q, ref);
        } else if (shape == 1) {
          set_ref8(l, i * 2, ref);
          set_ref8(l, i * 2 + 1, ref);
        } else if (shape == 2) {
          set_ref8(l, i, ref);
          set_ref8(l, i + 2, ref);
        } else {
          set_ref8(l, i, ref);
        }
      }
    const bool defer = !is_b;  // P: record the differences, the vectors are derived per picture later
    if (defer) {
      mb.mv_defer = 1;
      mb.mv_shape = shape;
      mb.mv_sub = sub_shape[0] | (sub_shape[1] << 2) | (sub_shape[2] << 4) | (sub_shape[3] << 6);
    }
    const MbMotion& mo = cur->motion[mb.addr];
    for (int l = 0; l < (is_b ? 2 : 1); ++l)
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
        const int ref = mo.ref_idx[l][(py >> 1) * 2 + (px >> 1)];
        for (int sy = 0; sy < ph; sy += sh)
          for (int sx = 0; sx < pw; sx += sw) {
            const int bx = px + sx, by = py + sy;
            int dir = PRED_MEDIAN;
            if (shape == 1) dir = i == 0 ? PRED_16x8_TOP : PRED_16x8_BOTTOM;
            if (shape == 2) dir = i == 0 ? PRED_8x16_LEFT : PRED_8x16_RIGHT;
            int mvp[2] = {0, 0};
            if (!defer) pred_mv(l, bx, by, sw, ref, dir, mvp);
            int dx, dy;
            rd_mvd_pair(l, bx, by, &dx, &dy);
            set_mvd_block(l, bx, by, sw, sh, dx, dy);
            if (compact) {
              // decode order = the order k_mv walks the partitions in
              const uint32_t word = static_cast<uint16_t>(dx) | (static_cast<uint32_t>(static_cast<uint16_t>(dy)) << 16);
              if (mb.pm_parts++ == 0) {
                bufs.pmo[mb.addr].mvd0[0] = static_cast<int16_t>(dx);
                bufs.pmo[mb.addr].mvd0[1] = static_cast<int16_t>(dy);
              } else {
                if (pmx_used >= bufs.pmx_capacity) return fail("motion extras overflow");
                bufs.pmx[pmx_used++] = word;
              }
            } else {
              set_mv_block(l, bx, by, sw, sh, mvp[0] + dx, mvp[1] + dy);
            }
          }
      }
    return true;
  }

  // The motion syntax of a macroblock of a P slice (7.3.5.1 / 7.3.5.2, mb_type 0 .. 4 of Table 7-13): exactly what
  // parse_inter does for it - the same reads in the same order, the same stores - with the shapes spelled out.  Through
  // parse_inter's loops over lists, partitions and sub-partitions a P macroblock took ~1 400 instructions here, of which
  // ~350 belonged to its ten bins (round 5, counted per region on the GPU box's EPYC: tools/hostprof, -DHWANG_REGION_COUNTERS);
  // in a P picture that was a quarter of the parser's instructions.
  bool parse_inter_p(int mb_type) {
    const int a = mb.addr;
    MbMotion& mo = cur->motion[a];
    int32_t* const uid = &cur->ref_uid[0][static_cast<size_t>(a) * 4];
    const int nref = sl.num_ref_idx[0];
    // the reference of the 8x8 quadrants in `mask`: index into the list, its surface, its picture (set_ref8)
    auto set_refs = [&](int mask, int ref) __attribute__((always_inline)) {
      int slot = -1, u = -1;
      if (sl.ref[0][ref].d) {
        slot = sl.ref[0][ref].slot;
        u = sl.ref[0][ref].uid;
      }
      for (int q = 0; q < 4; ++q)
        if ((mask >> q) & 1) {
          mo.ref_idx[0][q] = static_cast<int8_t>(ref);
          mo.ref_slot[0][q] = static_cast<int8_t>(slot);
          uid[q] = u;
        }
    };
    // ref_idx_l0 of the partition that starts at quadrant b8 (absent - inferred 0 - with one reference and in P_8x8ref0)
    // (the contexts of 9.3.3.1.1.6 for list 0 of a P slice: "the neighbouring partition's reference index is above 0" of
    // the 8x8 to the left and above - in this macroblock, or the adjoining quadrant of the macroblock next door; an intra,
    // skipped or not yet decoded quadrant holds -1 / 0 there and an unavailable macroblock does not count)
    auto read_ref = [&](int b8, int* ref) __attribute__((always_inline)) -> bool {
      if (mb_type == 4 || nref <= 1) {
        *ref = 0;
        return nref >= 1;
      }
      if (!use_cabac) {
        *ref = rd_ref_idx(0, b8);
        return *ref >= 0 && *ref < nref;
      }
      bool ca, cb;
      switch (b8) {
        case 0:
          ca = mb.avA && cur->motion[a - 1].ref_idx[0][1] > 0;
          cb = mb.avB && cur->motion[a - mbw].ref_idx[0][2] > 0;
          break;
        case 1:
          ca = mo.ref_idx[0][0] > 0;
          cb = mb.avB && cur->motion[a - mbw].ref_idx[0][3] > 0;
          break;
        case 2:
          ca = mb.avA && cur->motion[a - 1].ref_idx[0][3] > 0;
          cb = mo.ref_idx[0][0] > 0;
          break;
        default:
          ca = mo.ref_idx[0][2] > 0;
          cb = mo.ref_idx[0][1] > 0;
          break;
      }
      int inc = (ca ? 1 : 0) + (cb ? 2 : 0), v = 0;
      CabacLocal c(cabac);
      while (c.bin(54 + inc)) {
        ++v;
        inc = (inc >> 2) + 4;
        if (v >= 32) return false;
      }
      *ref = v;
      return v < nref;
    };
    const int shape = mb_type >= 3 ? 3 : mb_type;
    int sub = 0;
    if (shape == 3)
      for (int i = 0; i < 4; ++i) {
        const int t = rd_sub_mb_type();
        if (t < 0 || t > 3) return fail("bad sub_mb_type");
        mb.sub_type[i] = t;
        sub |= t << (2 * i);
      }
    int r;
    switch (shape) {
      case 0:
        if (!read_ref(0, &r)) return fail("ref_idx out of range");
        set_refs(15, r);
        break;
      case 1:
        if (!read_ref(0, &r)) return fail("ref_idx out of range");
        set_refs(3, r);
        if (!read_ref(2, &r)) return fail("ref_idx out of range");
        set_refs(12, r);
        break;
      case 2:
        if (!read_ref(0, &r)) return fail("ref_idx out of range");
        set_refs(5, r);
        if (!read_ref(1, &r)) return fail("ref_idx out of range");
        set_refs(10, r);
        break;
      default:
        for (int i = 0; i < 4; ++i) {
          if (!read_ref(i, &r)) return fail("ref_idx out of range");
          set_refs(1 << i, r);
        }
        break;
    }
    REGION(R_INTER_MVD);
    mb.mv_defer = 1;  // the differences are recorded, the vectors derived per picture later (k_mv / derive_deferred)
    mb.mv_shape = shape;
    mb.mv_sub = sub;
    uint8_t* const mvd0 = mvd_at(a, 0, 0);
    bool ok = true;
    // mvd_l0 of the (sub-)partition whose top-left 4x4 block is (bx, by), bw x bh blocks: read, kept as |mvd| for the
    // contexts of what follows (set_mvd_block), handed over
    auto one = [&](int bx, int by, int bw, int bh) __attribute__((always_inline)) {
      int dx, dy;
      rd_mvd_pair(0, bx, by, &dx, &dy);
      const uint16_t pat = static_cast<uint16_t>(imin(iabs(dx), 255) | (imin(iabs(dy), 255) << 8));
      if (bw == 4) {  // whole rows of the macroblock's sixteen {|mvd_x|, |mvd_y|} pairs: 8 bytes each
        const __m128i v = _mm_set1_epi16(static_cast<int16_t>(pat));
        for (int y = by; y < by + bh; y += 2) _mm_storeu_si128(reinterpret_cast<__m128i*>(mvd0 + y * 8), v);
      } else {
        for (int y = by; y < by + bh; ++y)
          for (int x = bx; x < bx + bw; ++x) memcpy(mvd0 + (y * 4 + x) * 2, &pat, 2);
      }
      if (compact) {
        // decode order = the order k_mv walks the partitions in
        if (mb.pm_parts++ == 0) {
          bufs.pmo[a].mvd0[0] = static_cast<int16_t>(dx);
          bufs.pmo[a].mvd0[1] = static_cast<int16_t>(dy);
        } else if (pmx_used >= bufs.pmx_capacity) {
          ok = false;
        } else {
          bufs.pmx[pmx_used++] = static_cast<uint16_t>(dx) | (static_cast<uint32_t>(static_cast<uint16_t>(dy)) << 16);
        }
      } else {
        set_mv_block(0, bx, by, bw, bh, dx, dy);
      }
    };
    switch (shape) {
      case 0:
        one(0, 0, 4, 4);
        break;
      case 1:
        one(0, 0, 4, 2);
        one(0, 2, 4, 2);
        break;
      case 2:
        one(0, 0, 2, 4);
        one(2, 0, 2, 4);
        break;
      default:
        for (int i = 0; i < 4 && ok; ++i) {
          const int px = (i & 1) * 2, py = (i >> 1) * 2;
          switch ((sub >> (2 * i)) & 3) {
            case 0:
              one(px, py, 2, 2);
              break;
            case 1:
              one(px, py, 2, 1);
              one(px, py + 1, 2, 1);
              break;
            case 2:
              one(px, py, 1, 2);
              one(px + 1, py, 1, 2);
              break;
            default:
              one(px, py, 1, 1);
              one(px + 1, py, 1, 1);
              one(px, py + 1, 1, 1);
              one(px + 1, py + 1, 1, 1);
              break;
          }
        }
        break;
    }
    return ok || fail("motion extras overflow");
  }

  // (the caller has called set_position(addr))
  void begin_mb(int addr) {
    if (sl.type == 1 || !compact) {  // (what reads them - pred_mv, pred_direct - runs for B slices and for sinks that leave the vectors to the host)
      static const int off_x[4] = {-1, 0, 1, -1}, off_y[4] = {-1, -1, -1, 0};
      for (int k = 0; k < 4; ++k) {
        const int na = addr + off_y[k] * mbw + off_x[k];
        mb.nbmo[k] = (mb.av[k] && !cur->mb_intra[na]) ? &cur->motion[na] : nullptr;
      }
      mb.nbmo[4] = &cur->motion[addr];
      mb.nbmo[5] = nullptr;
    }
    mb.kind = MB_NONE;
    mb.skip = mb.direct16 = mb.t8x8 = mb.i16_mode = mb.chroma_mode = mb.cbp = 0;
    mb.mv_shape = mb.mv_sub = mb.mv_pskip = mb.mv_defer = 0;
    mb.qp = qp;
    mb.cbf_dc = 0;
    mb.coded_mask = 0;
    mb.nent = 0;
    memset(mb.sub_type, 0, sizeof(mb.sub_type));
    memset(mb.i4_mode, 2, sizeof(mb.i4_mode));
    mb.nnz = &nnz[static_cast<size_t>(addr) * 24];
    memset(mb.nnz, 0, 24);
    mb.nz_mask = 0;
    MbMotion& mo = cur->motion[addr];
    if (!compact) memset(mo.mv, 0, sizeof(mo.mv));  // compact slices never read or upload these vectors
    memset(mo.ref_slot, -1, sizeof(mo.ref_slot));
    memset(mo.ref_idx, -1, sizeof(mo.ref_idx));
    mb.pm_parts = 0;
    for (int l = 0; l < 2; ++l)
      for (int q = 0; q < 4; ++q) cur->ref_uid[l][addr * 4 + q] = -1;
    memset(mvd_at(addr, 0, 0), 0, 64);
    memset(&direct8[addr * 4], 0, 4);
    cur->mb_intra[addr] = 0;
    mb_slice[addr] = static_cast<int16_t>(slice_index);
    mb_kind[addr] = MB_NONE;
  }

  bool commit_mb(const uint8_t* pcm) {
    const int a = mb.addr;
    ++mbs_decoded;
    const Pps& p = *apps;
    MbInfo& o = bufs.mb[a];
    memset(&o, 0, sizeof(o));
    mb_kind[a] = static_cast<uint8_t>(mb.kind);
    mb_skip[a] = static_cast<uint8_t>(mb.skip);
    mb_direct[a] = static_cast<uint8_t>((mb.skip || mb.direct16) && sl.type == 1);
    mb_t8x8[a] = static_cast<uint8_t>(mb.t8x8);
    mb_cbp[a] = static_cast<uint8_t>(mb.kind == MB_IPCM ? 0x2f : mb.cbp);
    mb_cbf_dc[a] = static_cast<uint8_t>(mb.cbf_dc);
    mb_chroma_mode[a] = static_cast<uint8_t>((mb.kind != MB_INTER && mb.kind != MB_IPCM) ? mb.chroma_mode : 0);
    memcpy(&i4mode[a * 16], mb.i4_mode, 16);
    const bool intra = mb.kind != MB_INTER;
    cur->mb_intra[a] = intra;
    if (intra) { has_intra = true; ++intra_mbs; } else has_inter = true;

    static const uint8_t kind_map[6] = {0, amd::MBK_I4x4, amd::MBK_I8x8, amd::MBK_I16x16, amd::MBK_IPCM, amd::MBK_INTER};
    o.kind = kind_map[mb.kind];
    uint8_t f = 0;
    if (mb.t8x8) f |= amd::MBF_T8X8;
    if (intra) f |= amd::MBF_INTRA;
    if (!p.constrained_intra_pred) {  // (intra_mb_ok is the neighbour's availability then)
      f |= static_cast<uint8_t>(mb.av_bits << 2);
    } else {
      if (intra_mb_ok(mb.mbx - 1, mb.mby)) f |= amd::MBF_AVAIL_A;
      if (intra_mb_ok(mb.mbx, mb.mby - 1)) f |= amd::MBF_AVAIL_B;
      if (intra_mb_ok(mb.mbx + 1, mb.mby - 1)) f |= amd::MBF_AVAIL_C;
      if (intra_mb_ok(mb.mbx - 1, mb.mby - 1)) f |= amd::MBF_AVAIL_D;
    }
    if (sl.disable_deblock != 1) {
      if (mb.mbx > 0 && !(sl.disable_deblock == 2 && mb_slice[a - 1] != mb_slice[a])) f |= amd::MBF_DB_LEFT;
      if (mb.mby > 0 && !(sl.disable_deblock == 2 && mb_slice[a - mbw] != mb_slice[a])) f |= amd::MBF_DB_TOP;
      o.deblock_internal = 1;
    }
    o.flags = f;
    const int qy = mb.kind == MB_IPCM ? 0 : mb.qp;
    o.qp = static_cast<uint8_t>(qy);
    o.qpc[0] = chroma_qp_table[iclip(0, 51, qy + p.chroma_qp_offset[0])];
    o.qpc[1] = chroma_qp_table[iclip(0, 51, qy + p.chroma_qp_offset[1])];
    o.pred_modes = static_cast<uint8_t>((mb.i16_mode & 15) | (mb.chroma_mode << 4));
    o.slice = static_cast<uint8_t>(slice_index - 1);
    {
      uint8_t mf = 0;
      if (compact) {
        // every macroblock of a compact slice has a PMotion record (intra: no reference)
        mf |= amd::MVF_COMPACT;
        amd::PMotion& pm = bufs.pmo[a];
        const MbMotion& mo = cur->motion[a];
        memcpy(pm.ref_idx, mo.ref_idx[0], 4);
        memcpy(pm.ref_slot, mo.ref_slot[0], 4);
        if (mb.pm_parts == 0) pm.mvd0[0] = pm.mvd0[1] = 0;
        any_compact = true;
      } else if (!intra) {
        any_full_motion = true;
      }
      if (mb.mv_defer) mf |= amd::MVF_DEFERRED;
      if (mb.mv_pskip) mf |= amd::MVF_PSKIP;
      mf |= static_cast<uint8_t>(mb.av_bits << 4);
      o.mv_flags = mf;
      o.mv_shape = static_cast<uint8_t>(mb.mv_shape);
      o.mv_sub = static_cast<uint8_t>(mb.mv_sub);
      uint8_t* rec = &cur->mvrec[static_cast<size_t>(a) * 3];
      rec[0] = mf;
      rec[1] = o.mv_shape;
      rec[2] = o.mv_sub;
      if (mb.mv_defer) cur->mv_deferred = true;
    }
    if (intra) {
      for (int k = 0; k < 8; ++k) o.i4_modes[k] = static_cast<uint8_t>((mb.i4_mode[2 * k] & 15) | (mb.i4_mode[2 * k + 1] << 4));
    } else {
      memset(o.i4_modes, 0x22, sizeof(o.i4_modes));  // (begin_mb's value - DC - in every nibble: an inter macroblock has no modes)
    }
    uint16_t nz = mb.nz_mask;  // (CABAC: the residual reader's; an 8x8 block marks its four 4x4 blocks)
    if (!use_cabac) {
      nz = static_cast<uint16_t>(~_mm_movemask_epi8(
          _mm_cmpeq_epi8(_mm_loadu_si128(reinterpret_cast<const __m128i*>(mb.nnz)), _mm_setzero_si128())));
      if (mb.t8x8)  // the transform block is the 8x8: any coefficient marks all four
        for (int b8 = 0; b8 < 4; ++b8) {
          const int o4 = (b8 >> 1) * 8 + (b8 & 1) * 2;
          const uint16_t m = (1u << o4) | (1u << (o4 + 1)) | (1u << (o4 + 4)) | (1u << (o4 + 5));
          if (nz & m) nz |= m;
        }
    }
    if (mb.kind == MB_IPCM) nz = 0xffff;
    o.nz_mask = nz;

    // coefficients: one word per non-zero level of the groups that were touched (MbInfo::ncoef)
    o.coef_offset = static_cast<uint32_t>(coef_used);
    if (coef_used + amd::kMaxSparseCoefs > bufs.coef_capacity) return fail("coefficient buffer overflow");
    if (mb.kind == MB_IPCM) {
      memcpy(bufs.coefs + coef_used, pcm, 384);
      coef_used += 96;
      o.coded_mask = 0;
    } else if (mb.coded_mask) {
      uint32_t* dst = bufs.coefs + coef_used;
      if (mb.nent <= amd::kMaxSparseCoefs) {
        memcpy(dst, mb.ent, sizeof(uint32_t) * mb.nent);
        o.ncoef = static_cast<uint16_t>(mb.nent);
        coef_used += mb.nent;
      } else {  // more levels than the sparse form is worth: dense groups in mask order (at most 26 x 8 words)
        for (uint32_t m = mb.coded_mask; m; m &= m - 1) memset(mb.scratch[__builtin_ctz(m)], 0, 32);
        for (int i = 0; i < mb.nent; ++i) (&mb.scratch[0][0])[mb.ent[i] >> 16] = static_cast<int16_t>(mb.ent[i] & 0xffff);
        int16_t* d16 = reinterpret_cast<int16_t*>(dst);
        for (uint32_t m = mb.coded_mask; m; m &= m - 1) {
          memcpy(d16, mb.scratch[__builtin_ctz(m)], 32);
          d16 += 16;
        }
        o.ncoef = amd::kDenseCoefs;
        coef_used += static_cast<size_t>(__builtin_popcount(mb.coded_mask)) * 8;
      }
      o.coded_mask = mb.coded_mask;
    }
    return true;
  }

  // What every P_Skip macroblock of this slice shares (reference 0 of list 0, zero differences).
  void prepare_pskip() {
    memset(&pskip_motion, 0, sizeof(pskip_motion));
    memset(pskip_motion.ref_slot, -1, sizeof(pskip_motion.ref_slot));
    memset(pskip_motion.ref_idx, -1, sizeof(pskip_motion.ref_idx));
    pskip_ref_uid = -1;
    pskip_info_qp = -1;  // (the record template: made for the first P_Skip macroblock, and again when QP_Y has moved)
    if (sl.num_ref_idx[0] > 0 && sl.ref[0][0].d) {
      const RefSnap& r = sl.ref[0][0];
      for (int q = 0; q < 4; ++q) {
        pskip_motion.ref_idx[0][q] = 0;
        pskip_motion.ref_slot[0][q] = static_cast<int8_t>(r.slot);
      }
      pskip_ref_uid = r.uid;
    }
  }

  // P_Skip in one pass: exactly the stores begin_mb + decode_skip_mb + commit_mb make for it
  // (tests/golden/parser_digest.json pins the records byte for byte).
  bool decode_pskip(int addr) {
    REGION(R_PSKIP);
    const Pps& p = *apps;
    last_qp_delta_nonzero = 0;
    // host-side context state the neighbours will ask for
    mb_slice[addr] = static_cast<int16_t>(slice_index);
    mb_kind[addr] = MB_INTER;
    mb_skip[addr] = 1;
    mb_direct[addr] = 0;
    mb_t8x8[addr] = 0;
    mb_cbp[addr] = 0;
    mb_cbf_dc[addr] = 0;
    mb_chroma_mode[addr] = 0;
    memset(&i4mode[addr * 16], 2, 16);
    memset(&nnz[addr * 24], 0, 24);
    memset(mvd_at(addr, 0, 0), 0, 64);
    memset(&direct8[addr * 4], 0, 4);
    cur->mb_intra[addr] = 0;
    if (compact) {
      // the 12-byte record is all the GPU needs; the host keeps the reference indices for its contexts
      amd::PMotion& pm = bufs.pmo[addr];
      memcpy(pm.ref_idx, pskip_motion.ref_idx[0], 4);
      memcpy(pm.ref_slot, pskip_motion.ref_slot[0], 4);
      pm.mvd0[0] = pm.mvd0[1] = 0;
      memcpy(cur->motion[addr].ref_idx, pskip_motion.ref_idx, sizeof(pskip_motion.ref_idx));
      any_compact = true;
    } else {
      cur->motion[addr] = pskip_motion;
      any_full_motion = true;
    }
    for (int q = 0; q < 4; ++q) {
      cur->ref_uid[0][addr * 4 + q] = pskip_ref_uid;
      cur->ref_uid[1][addr * 4 + q] = -1;
    }
    has_inter = true;
    ++mbs_decoded;
    // the record: what does not change from one P_Skip macroblock of the slice to the next comes from a template
    if (pskip_info_qp != qp) {
      memset(&pskip_info, 0, sizeof(pskip_info));
      pskip_info.kind = amd::MBK_INTER;
      pskip_info.deblock_internal = sl.disable_deblock != 1;
      pskip_info.qp = static_cast<uint8_t>(qp);
      pskip_info.qpc[0] = chroma_qp_table[iclip(0, 51, qp + p.chroma_qp_offset[0])];
      pskip_info.qpc[1] = chroma_qp_table[iclip(0, 51, qp + p.chroma_qp_offset[1])];
      pskip_info.slice = static_cast<uint8_t>(slice_index - 1);
      memset(pskip_info.i4_modes, 0x22, sizeof(pskip_info.i4_modes));
      pskip_info_qp = qp;
    }
    MbInfo o = pskip_info;
    uint8_t f = 0;
    if (!p.constrained_intra_pred) {
      f |= static_cast<uint8_t>(mb.av_bits << 2);
    } else {
      if (intra_mb_ok(mb.mbx - 1, mb.mby)) f |= amd::MBF_AVAIL_A;
      if (intra_mb_ok(mb.mbx, mb.mby - 1)) f |= amd::MBF_AVAIL_B;
      if (intra_mb_ok(mb.mbx + 1, mb.mby - 1)) f |= amd::MBF_AVAIL_C;
      if (intra_mb_ok(mb.mbx - 1, mb.mby - 1)) f |= amd::MBF_AVAIL_D;
    }
    if (sl.disable_deblock != 1) {
      if (mb.mbx > 0 && !(sl.disable_deblock == 2 && mb_slice[addr - 1] != mb_slice[addr])) f |= amd::MBF_DB_LEFT;
      if (mb.mby > 0 && !(sl.disable_deblock == 2 && mb_slice[addr - mbw] != mb_slice[addr])) f |= amd::MBF_DB_TOP;
    }
    o.flags = f;
    const uint8_t mf = static_cast<uint8_t>(amd::MVF_DEFERRED | amd::MVF_PSKIP | (compact ? amd::MVF_COMPACT : 0) | (mb.av_bits << 4));
    o.mv_flags = mf;
    uint8_t* rec = &cur->mvrec[static_cast<size_t>(addr) * 3];
    rec[0] = mf;
    rec[1] = rec[2] = 0;
    cur->mv_deferred = true;
    o.coef_offset = static_cast<uint32_t>(coef_used);
    bufs.mb[addr] = o;
    REGION(R_MBHEAD);
    return true;
  }

  bool decode_skip_mb(int addr) {
    if (sl.type == 0) return decode_pskip(addr);
    begin_mb(addr);
    mb.kind = MB_INTER;
    mb.skip = 1;
    last_qp_delta_nonzero = 0;
    if (sl.type == 0) {
      for (int q = 0; q < 4; ++q) set_ref8(0, q, 0);
      mb.mv_defer = 1;   // mv[0] stays zero (begin_mb); the inference (8.4.1.1) runs with the picture's derivation
      mb.mv_pskip = 1;
    } else {
      mb.direct16 = 1;
      memset(&direct8[addr * 4], 1, 4);
      if (!pred_direct(-1)) return false;
    }
    return commit_mb(nullptr);
  }

  bool decode_mb(int addr) {
    const Pps& p = *apps;
    REGION(R_BEGIN);
    begin_mb(addr);
    const int st = sl.type;
    REGION(R_MBTYPE);
    const int t = rd_mb_type();
    int intra_t = -1;
    if (st == 2) {
      intra_t = t;
    } else if (st == 0) {
      if (t > 30 || t < 0) return fail("bad mb_type (P)");
      if (t >= 5) intra_t = t - 5;
    } else {
      if (t > 48 || t < 0) return fail("bad mb_type (B)");
      if (t >= 23) intra_t = t - 23;
    }
    if (intra_t > 25) return fail("bad mb_type (I)");

    if (intra_t == 25) {
      mb.kind = MB_IPCM;
      uint8_t pcm[384];
      if (use_cabac) {
        // the terminating bin consumed the encoder's flush; align and read
        int pos = cabac.bits_consumed();
        pos = (pos + 7) & ~7;
        const uint8_t* src = cabac.start + (pos >> 3);
        if (src + 384 > cabac.end) return fail("truncated I_PCM");
        memcpy(pcm, src, 384);
        Cabac saved = cabac;
        cabac.begin(src + 384, saved.end);
        memcpy(cabac.state, saved.state, sizeof(cabac.state));
        cabac.start = saved.start;  // keep positions relative to the slice data
      } else {
        while (bits.pos & 7) bits.bit();
        for (int i = 0; i < 384; ++i) pcm[i] = static_cast<uint8_t>(bits.get(8));
        if (bits.overrun) return fail("truncated I_PCM");
      }
      memset(mb.nnz, 16, 24);
      mb.cbf_dc = 7;
      last_qp_delta_nonzero = 0;
      return commit_mb(pcm);
    }

    if (intra_t >= 0) {
      REGION(R_INTRAMODES);
      if (intra_t == 0) {
        mb.kind = MB_I4x4;
        if (p.transform_8x8_mode && rd_t8x8_flag()) {
          mb.kind = MB_I8x8;
          mb.t8x8 = 1;
        }
        const int nblk = mb.kind == MB_I4x4 ? 16 : 4;
        if (use_cabac) {
          // The modes of the sixteen blocks (or four) in one pass: the arithmetic decoder in registers throughout, and
          // the neighbours' modes (8.3.1.1 / 8.3.2.1) in a 5 x 5 grid whose row -1 / column -1 are the macroblocks
          // above / to the left: -1 = not available for intra prediction (then the prediction is DC whatever the
          // other neighbour says), 2 = a neighbour that is not Intra4x4 / Intra8x8.  (predicted_intra_mode +
          // rd_intra_pred_mode per block were 1 100 of an Intra4x4 macroblock's instructions outside its bins.)
          int8_t grid[25];
          {
            const int a = mb.addr;
            auto edge = [&](bool avail, int na, int8_t* dst, int stride, int first, int step) {
              int8_t fill = 2;
              const int kind = avail ? mb_kind[na] : MB_NONE;
              if (!avail || (kind == MB_INTER && p.constrained_intra_pred)) fill = -1;
              const bool modes = avail && (kind == MB_I4x4 || kind == MB_I8x8);
              for (int i = 0; i < 4; ++i) dst[i * stride] = modes ? static_cast<int8_t>(i4mode[na * 16 + first + i * step]) : fill;
            };
            edge(mb.av[1], a - mbw, grid + 1, 1, 12, 1);   // above: its bottom row
            edge(mb.av[3], a - 1, grid + 5, 5, 3, 4);      // left: its right column
          }
          CabacLocal c(cabac);
          for (int k = 0; k < nblk; ++k) {
            const int bx = nblk == 16 ? blk4_x[k] : (k & 1) * 2, by = nblk == 16 ? blk4_y[k] : (k >> 1) * 2;
            const int m = imin(grid[(by + 1) * 5 + bx], grid[by * 5 + bx + 1]);
            const int pm = m < 0 ? 2 : m;
            int mode = pm;
            if (!c.bin(68)) {
              int rem = c.bin(69);
              rem |= c.bin(69) << 1;
              rem |= c.bin(69) << 2;
              mode = rem < pm ? rem : rem + 1;
            }
            if (nblk == 16) {
              mb.i4_mode[by * 4 + bx] = static_cast<int8_t>(mode);
              grid[(by + 1) * 5 + bx + 1] = static_cast<int8_t>(mode);
            } else {
              for (int s = 0; s < 4; ++s) {
                mb.i4_mode[(by + (s >> 1)) * 4 + bx + (s & 1)] = static_cast<int8_t>(mode);
                grid[(by + (s >> 1) + 1) * 5 + bx + (s & 1) + 1] = static_cast<int8_t>(mode);
              }
            }
          }
        } else
        for (int k = 0; k < nblk; ++k) {
          const int bx = nblk == 16 ? blk4_x[k] : (k & 1) * 2, by = nblk == 16 ? blk4_y[k] : (k >> 1) * 2;
          const int pm = predicted_intra_mode(bx, by);
          const int rem = rd_intra_pred_mode();
          const int mode = rem < 0 ? pm : (rem < pm ? rem : rem + 1);
          if (nblk == 16) mb.i4_mode[by * 4 + bx] = static_cast<int8_t>(mode);
          else
            for (int s = 0; s < 4; ++s) mb.i4_mode[(by + (s >> 1)) * 4 + bx + (s & 1)] = static_cast<int8_t>(mode);
        }
      } else {
        mb.kind = MB_I16x16;
        mb.i16_mode = (intra_t - 1) & 3;
        mb.cbp = (((intra_t - 1) >> 2) % 3) << 4 | ((intra_t - 1) >= 12 ? 15 : 0);
      }
      mb.chroma_mode = rd_chroma_pred_mode();
      if (mb.chroma_mode > 3 || mb.chroma_mode < 0) return fail("bad intra_chroma_pred_mode");
      cur->mb_intra[addr] = 1;
    } else {
      REGION(R_INTER_REFS);
      mb.kind = MB_INTER;
      if (st == 1 && t == 0) {
        mb.direct16 = 1;
        memset(&direct8[addr * 4], 1, 4);
        if (!pred_direct(-1)) return false;
      } else if (!(st == 0 ? parse_inter_p(t) : parse_inter(t))) {
        return false;
      }
    }

    REGION(R_CBP);
    if (mb.kind != MB_I16x16) {
      int cbp = rd_cbp(mb.kind != MB_INTER);
      if (cbp < 0) return fail("bad coded_block_pattern");
      mb.cbp = cbp;
      if ((cbp & 15) && p.transform_8x8_mode && mb.kind == MB_INTER) {
        bool ok = true;
        if (st == 1 && t == 0) ok = asps->direct_8x8_inference;
        if ((st == 0 && (t == 3 || t == 4)) || (st == 1 && t == 22)) {
          for (int i = 0; i < 4; ++i) {
            if (st == 1 && mb.sub_type[i] == 0) {
              if (!asps->direct_8x8_inference) ok = false;
            } else if (st == 1 ? b_sub_info[mb.sub_type[i]][0] != 0 : mb.sub_type[i] != 0) {
              ok = false;
            }
          }
        }
        REGION(R_T8X8);
        if (ok) mb.t8x8 = rd_t8x8_flag();
      }
    }
    REGION(R_QPD);
    if (mb.cbp || mb.kind == MB_I16x16) {
      const int dq = rd_qp_delta();
      if (dq < -26 || dq > 25) return fail("mb_qp_delta out of range");
      last_qp_delta_nonzero = dq != 0;
      qp = (qp + dq + 52) % 52;
      mb.qp = qp;
      REGION(R_RESIDUAL);
      if (!parse_residual()) return false;
    } else {
      last_qp_delta_nonzero = 0;
    }
    REGION(R_COMMIT);
    const bool ok = commit_mb(nullptr);
    REGION(R_MBHEAD);
    return ok;
  }

  bool decode_slice_data(Bits& b) {
    const Pps& p = *apps;
    const int nmb = mbw * mbh;
    int addr = sl.first_mb;
    if (addr >= nmb) return fail("first_mb_in_slice out of range");
    // Slices arrive in macroblock order (FMO / ASO are rejected): a slice that starts anywhere but at the
    // first macroblock not decoded yet is a duplicate, an overlap or a gap.  A slice that re-decodes
    // macroblocks would append to the per-picture streams (coefficients, motion extras) twice.
    if (addr != mbs_decoded) return fail("slice does not start at the next undecoded macroblock");
    pos_valid = false;
    col_motion = nullptr;
    // P and I slices hand their motion over compactly when the sink derives the vectors (k_mv)
    compact = bufs.device_mv && bufs.pmo && bufs.pmx && sl.type != 1;
    if (sl.type == 0) prepare_pskip();
    qp = sl.qp;
    last_qp_delta_nonzero = 0;
    use_cabac = p.cabac;
    if (use_cabac) {
      while (b.pos & 7)
        if (!b.bit()) return fail("cabac_alignment_one_bit is zero");
      cabac.init_contexts(sl.type, sl.cabac_init_idc, sl.qp);
      cabac.begin(b.buf + (b.pos >> 3), b.buf + ((b.size_bits + 7) >> 3));
      REGION(R_MBHEAD);
      // mb_skip_flag of the macroblock at `addr` (the caller has called set_position): 9.3.3.1.1.1
      const int skip_base = sl.type == 0 ? 11 : 24;
      auto skip_ctx = [&]() __attribute__((always_inline)) {
        return skip_base + ((mb.avA && !mb_skip[addr - 1]) ? 1 : 0) + ((mb.avB && !mb_skip[addr - mbw]) ? 1 : 0);
      };
      bool skipped = false;
      set_position(addr);
      if (sl.type != 2) skipped = cabac.bin(skip_ctx());
      for (;;) {
        if (!(skipped ? decode_skip_mb(addr) : decode_mb(addr))) return false;
        if (cabac.overrun) return fail("slice data ran past the end of the NAL unit");
        ++addr;
        // end_of_slice_flag and the next macroblock's skip flag in one go: the engine in registers for the pair
        bool past_end = false;
        {
          CabacLocal c(cabac);
          if (c.terminate()) break;
          if (addr >= nmb) {
            past_end = true;
          } else {
            set_position(addr);
            if (sl.type != 2) skipped = c.bin(skip_ctx());
          }
        }
        if (past_end) return fail("slice runs past the end of the picture");
      }
      REGION(R_OTHER);
      {  // desync check: last consumed bit is the stop bit, only zeros follow
        const int pos = (b.pos & ~7) + cabac.bits_consumed();
        bool ok = pos > 0 && pos <= b.size_bits && ((b.buf[(pos - 1) >> 3] >> (7 - ((pos - 1) & 7))) & 1);
        for (int i = pos; ok && i < b.size_bits; ++i)
          if ((b.buf[i >> 3] >> (7 - (i & 7))) & 1) ok = false;
        ++slices;
        if (ok) ++slices_clean_end;
      }
      return true;
    }
    bits = b;
    for (;;) {
      if (sl.type != 2) {
        unsigned run = bits.ue();
        if (run > static_cast<unsigned>(nmb - addr)) return fail("mb_skip_run out of range");
        while (run--) {
          set_position(addr);
          if (!decode_skip_mb(addr)) return false;
          ++addr;
        }
        if (!bits.more_rbsp_data()) break;
      }
      if (addr >= nmb) return fail("slice runs past the end of the picture");
      set_position(addr);
      if (!decode_mb(addr)) return false;
      if (bits.overrun) return fail("slice data ran past the end of the NAL unit");
      ++addr;
      if (!bits.more_rbsp_data()) break;
    }
    {
      Bits t = bits;
      bool ok = t.bit() == 1;
      while (ok && t.pos < t.size_bits)
        if (t.bit()) ok = false;
      ++slices;
      if (ok && !t.overrun) ++slices_clean_end;
    }
    return true;
  }

};

// One picture on its way through the entropy workers.
// Emulation prevention bytes out of src[i0 .. n) (what lies before i0 has been dealt with), appended at out + o; returns
// the new length.  A 0x03 is dropped when the two bytes before it - in the escaped stream - are zero; everything
// between two 0x03 bytes moves with memcpy (they are ~256 bytes apart on average).
size_t unescape_span(const uint8_t* src, size_t n, size_t i0, uint8_t* out, size_t o) {
  size_t i = i0;
  while (i < n) {
    const uint8_t* p = static_cast<const uint8_t*>(memchr(src + i, 3, n - i));
    const size_t j = p ? static_cast<size_t>(p - src) : n;
    memcpy(out + o, src + i, j - i);
    o += j - i;
    if (!p) break;
    if (!(j >= 2 && src[j - 1] == 0 && src[j - 2] == 0)) out[o++] = 3;
    i = j + 1;
  }
  return o;
}

// Where the pipeline's threads spend their time (process-wide, nanoseconds; read by pipe_debug_stats()).
std::atomic<int64_t> g_ns_room{0}, g_ns_surface{0}, g_ns_worker_busy{0}, g_ns_handover{0}, g_n_tasks{0};
inline int64_t now_ns() {
  return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

struct PicTask {
  std::shared_ptr<PicData> cur;
  std::vector<std::shared_ptr<PicData>> keep;  // every picture a slice of this one refers to
  Sps sps;
  PicBuffers bufs;
  struct Slice {
    SliceHdr hdr;
    Pps pps;
    std::vector<uint8_t> raw;  // the NAL unit's payload as it came (escaped): the worker removes the emulation prevention bytes
    int pos = 0, index = 0;    // pos: first bit of the slice data in the RBSP
  };
  std::deque<Slice> slices;
  PicDone done;
  Pps done_pps;  // done.scaling4 / scaling8 point here
  uint64_t epoch = 0;     // which interval (Parser::Impl::epoch)
  bool finished = false;  // slice data decoded (or failed): ready to be handed to the sink, in decoding order
  std::string err;
  int64_t slices_n = 0, clean_n = 0;
};

}  // namespace

void pipe_debug_stats(int64_t out[6]) {
  out[0] = g_ns_room.load();
  out[1] = g_ns_surface.load();
  out[2] = g_ns_worker_busy.load();
  out[3] = g_ns_handover.load();
  out[4] = g_n_tasks.load();
  out[5] = 0;
}

// ===========================================================================
struct WorkerPool::State {
  std::mutex mu;
  std::condition_variable cv;
  std::deque<std::function<void()>> jobs;
  bool stop = false;
  std::vector<std::thread> threads;
};

WorkerPool::WorkerPool(int threads) : state_(new State()) {
  State* s = state_;
  for (int i = 0; i < threads; ++i)
    s->threads.emplace_back([s] {
      for (;;) {
        std::function<void()> job;
        {
          std::unique_lock<std::mutex> lk(s->mu);
          s->cv.wait(lk, [&] { return s->stop || !s->jobs.empty(); });
          if (s->jobs.empty()) return;  // stop, and nothing left to run
          job = std::move(s->jobs.front());
          s->jobs.pop_front();
        }
        job();
      }
    });
}

WorkerPool::~WorkerPool() {
  {
    std::lock_guard<std::mutex> lk(state_->mu);
    state_->stop = true;
  }
  state_->cv.notify_all();
  for (std::thread& t : state_->threads) t.join();
  delete state_;
}

int WorkerPool::size() const { return static_cast<int>(state_->threads.size()); }

void WorkerPool::submit(std::function<void()> job) {
  {
    std::lock_guard<std::mutex> lk(state_->mu);
    state_->jobs.push_back(std::move(job));
  }
  state_->cv.notify_one();
}

// ===========================================================================
// The parser thread's side: NAL units, parameter sets, slice headers, the DPB
// (reference lists, marking, POC, display order).  Slice data goes to `sd` at
// once (serial mode) or, picture by picture, to the entropy workers.
// ===========================================================================
struct Parser::Impl {
  Parser* owner;
  PictureSink* sink;
  std::string* err;

  int nal_length_size = 4;
  Sps sps[kMaxSps];
  Pps pps[kMaxPps];
  const Sps* asps = nullptr;
  const Pps* apps = nullptr;
  int mbw = 0, mbh = 0;

  std::vector<uint8_t> rbsp;

  Pic dpb[kMaxDpb + 8];
  std::shared_ptr<PicPool> pool;
  int next_uid = 0;
  Pic* cur = nullptr;
  bool pic_in_progress = false;
  bool skipping_picture = false;  // remaining slices of a picture skipped on purpose
  PicBuffers bufs;
  int num_slices = 0;
  bool any_deblock = false;
  int slice_type_mask = 0;

  int prev_poc_msb = 0, prev_poc_lsb = 0, prev_frame_num = 0, prev_frame_num_offset = 0;
  bool prev_ref_has_mmco5 = false;
  int max_long_term_frame_idx = -1;

  // sparse decoding
  std::vector<uint64_t> wanted;
  std::vector<uint64_t> keyframes;  // decoding-order ordinals of the interval's sync samples (optional)
  uint64_t outputs_emitted = 0;  // since last flush/configure
  uint64_t pictures_started = 0; // since last flush/configure: the decoding-order ordinal of the next picture

  SliceHdr sl;
  int slice_index = 0;
  SliceData sd;  // serial mode: slice data is decoded on the parser thread

  bool fail(const char* msg) {
    if (err->empty()) *err = msg;
    return false;
  }

  // ------------------------------------------------------- entropy workers
  // With a WorkerPool the parser thread runs ahead: per picture it parses the
  // headers, settles the DPB (references, marking, output order need nothing
  // from the slice data) and queues a PicTask; workers decode the slice data of
  // up to `max_in_flight` consecutive pictures at once.  A picture waits for
  // another only where a B slice's direct prediction reads the co-located
  // picture's motion (SliceData::final_motion).  Finished pictures are handed
  // to the sink strictly in decoding order - by whichever worker completes the
  // oldest one.  Display-order output does NOT wait for that: output_picture is
  // called from the parser thread as soon as the DPB releases a picture, so the
  // caller's protocol (DecoderAutomata: one interval after the other) advances
  // at header speed and the pictures of several intervals are with the workers
  // at a time; the sink defers what it is asked to do with a frame until the
  // frame's end_picture has come (or abandon_picture, if it never will).
  WorkerPool* workers = nullptr;
  int max_in_flight = 0;
  std::mutex pipe_mu;
  std::condition_variable pipe_cv;
  std::deque<std::shared_ptr<PicTask>> inflight;  // decoding order
  bool handing_over = false;  // a thread is inside the in-order hand-over loop
  // A picture that fails with the workers takes the rest of its interval with it (what follows refers to it);
  // the next interval starts afresh.  Intervals are told apart by `epoch` (raised by flush()).  The failure is
  // reported once, by the next decode(), flush() or wait_idle().
  uint64_t epoch = 1;
  uint64_t failed_epoch = 0;
  bool pipe_unreported = false;
  std::string pipe_error;
  std::shared_ptr<PicTask> building;

  bool pipe_check_locked() {
    if (!pipe_unreported) return true;
    pipe_unreported = false;
    return fail(pipe_error.c_str());
  }
  bool pipe_wait_for_room() {
    std::unique_lock<std::mutex> lk(pipe_mu);
    if (static_cast<int>(inflight.size()) < max_in_flight) return true;
    const int64_t t0 = now_ns();
    pipe_cv.wait(lk, [&] { return static_cast<int>(inflight.size()) < max_in_flight; });
    g_ns_room += now_ns() - t0;
    return true;
  }
  // waits until one picture fewer is in flight; false if none is
  bool pipe_wait_one() {
    std::unique_lock<std::mutex> lk(pipe_mu);
    const size_t n = inflight.size();
    if (n == 0) return false;
    const int64_t t0 = now_ns();
    pipe_cv.wait(lk, [&] { return inflight.size() < n; });
    g_ns_surface += now_ns() - t0;
    return true;
  }
  // Nothing in flight and no worker inside hand_over any more (the destructor relies on the second part: a worker
  // that has popped the last picture still takes pipe_mu once more).
  bool pipe_idle_locked() const { return inflight.empty() && !handing_over; }
  bool pipe_drain() {
    std::unique_lock<std::mutex> lk(pipe_mu);
    pipe_cv.wait(lk, [&] { return pipe_idle_locked(); });
    return pipe_check_locked();
  }
  void pipe_begin_task(Pic* c) {
    building = std::make_shared<PicTask>();
    building->cur = c->d;
    building->sps = *asps;
    building->bufs = bufs;
    building->epoch = epoch;
  }
  void pipe_add_slice(const Bits& b) {
    building->slices.emplace_back();
    PicTask::Slice& s = building->slices.back();
    s.hdr = sl;
    s.pps = *apps;
    s.raw.assign(ue.src, ue.src + ue.n);  // (the caller's sample need not outlive feed())
    s.pos = b.pos;
    s.index = slice_index;
    for (int l = 0; l < 2; ++l)
      for (int i = 0; i < sl.num_ref_idx[l]; ++i)
        if (sl.ref_list[l][i]) building->keep.push_back(sl.ref_list[l][i]->d);
  }
  void pipe_drop_task() { building.reset(); }
  void pipe_submit_task(const PicDone& front_done) {
    std::shared_ptr<PicTask> t = std::move(building);
    building.reset();
    t->done = front_done;
    t->done_pps = *apps;
    t->done.scaling4 = t->done_pps.scaling4;
    t->done.scaling8 = t->done_pps.scaling8;
    {
      std::lock_guard<std::mutex> lk(pipe_mu);
      inflight.push_back(t);
    }
    workers->submit([this, t] { run_task(t); });
  }
  // worker thread
  void run_task(const std::shared_ptr<PicTask>& t) {
    static thread_local std::unique_ptr<SliceData> tls;
    if (!tls) tls.reset(new SliceData());
    SliceData& w = *tls;
    const int64_t t_begin = now_ns();
    w.begin_picture(&t->sps, t->cur.get(), t->bufs);
    bool ok = true;
    static thread_local std::vector<uint8_t> rbsp_tls;
    for (PicTask::Slice& s : t->slices) {
      rbsp_tls.resize(s.raw.size() + 8);
      const size_t rn = unescape_span(s.raw.data(), s.raw.size(), 0, rbsp_tls.data(), 0);
      memset(rbsp_tls.data() + rn, 0, 8);
      Bits b;
      b.buf = rbsp_tls.data();
      b.size_bits = static_cast<int>(rn) * 8;
      b.pos = s.pos;
      if (!w.decode_slice(s.hdr, &s.pps, s.index, b)) {
        ok = false;
        break;
      }
    }
    if (ok) ok = w.finish_picture(&t->done);
    if (!ok) t->err = w.err.empty() ? "slice data could not be decoded" : w.err;
    t->slices_n = w.slices;
    t->clean_n = w.slices_clean_end;
    w.slices = w.slices_clean_end = 0;
    {
      // every record of the picture is final (or the picture failed): B pictures waiting for it go on
      std::lock_guard<std::mutex> lk(t->cur->mu);
      t->cur->parsed = true;
    }
    t->cur->cv.notify_all();
    const int64_t t_parsed = now_ns();
    g_ns_worker_busy += t_parsed - t_begin;
    ++g_n_tasks;
    hand_over(t);
    g_ns_handover += now_ns() - t_parsed;
  }
  // Pictures go to the sink in decoding order: whoever finishes the oldest one
  // in flight hands over it and every finished one behind it.
  void hand_over(const std::shared_ptr<PicTask>& done_task) {
    std::unique_lock<std::mutex> lk(pipe_mu);
    done_task->finished = true;
    if (handing_over) return;  // the thread in the loop below will get to it
    handing_over = true;
    while (!inflight.empty() && inflight.front()->finished) {
      std::shared_ptr<PicTask> t = inflight.front();
      if (!t->err.empty() && failed_epoch != t->epoch) {
        failed_epoch = t->epoch;
        pipe_error = t->err;
        pipe_unreported = true;
      }
      const bool drop = failed_epoch == t->epoch;  // this picture or an earlier one of its interval failed
      lk.unlock();
      std::string sink_error;
      bool sink_ok = true;
      if (!drop) sink_ok = sink->end_picture(t->done, &sink_error);
      if (drop || !sink_ok) sink->abandon_picture(t->done);
      lk.lock();
      if (!sink_ok && failed_epoch != t->epoch) {
        failed_epoch = t->epoch;
        pipe_error = sink_error.empty() ? "the sink refused a picture" : sink_error;
        pipe_unreported = true;
      }
      owner->slices_ += t->slices_n;
      owner->slices_clean_end_ += t->clean_n;
      inflight.pop_front();
      // the task's references go here, outside the lock (the last holder of a picture returns its surface)
      lk.unlock();
      t.reset();
      pipe_cv.notify_all();
      lk.lock();
    }
    handing_over = false;
    pipe_cv.notify_all();  // (under the lock: the parser may be waiting for this to tear the object down)
  }

  // ----------------------------------------------------------------- NAL
  // Emulation-prevention bytes out, in two steps where that saves work: a slice NAL is unescaped only as far as its
  // header can reach before the header is parsed, and the rest only if the picture is decoded at all (with a sparse
  // request two samples out of three are non-reference pictures nobody wants: 20-50 KB each that are never touched).
  struct {
    const uint8_t* src = nullptr;
    size_t n = 0, i = 0, o = 0;
    int zeros = 0;
  } ue;
  void unescape_begin(const uint8_t* src, size_t n) {
    rbsp.resize(n + 8);
    ue.src = src;
    ue.n = n;
    ue.i = ue.o = 0;
    ue.zeros = 0;
  }
  // consumes input up to `upto` bytes; returns the RBSP bytes produced so far (8 zero bytes follow them)
  int unescape_upto(size_t upto) {
    const size_t end = std::min(upto, ue.n);
    size_t o = ue.o;
    int zeros = ue.zeros;
    uint8_t* out = rbsp.data();
    for (size_t i = ue.i; i < end; ++i) {
      const uint8_t c = ue.src[i];
      if (zeros >= 2 && c == 3) {
        zeros = 0;
        continue;
      }
      out[o++] = c;
      zeros = c == 0 ? zeros + 1 : 0;
    }
    ue.i = std::max(ue.i, end);
    ue.o = o;
    ue.zeros = zeros;
    memset(out + o, 0, 8);
    return static_cast<int>(o);
  }

  // the rest of the unit at once
  int unescape_rest() {
    uint8_t* out = rbsp.data();
    ue.o = unescape_span(ue.src, ue.n, ue.i, out, ue.o);
    ue.i = ue.n;
    memset(out + ue.o, 0, 8);
    return static_cast<int>(ue.o);
  }

  // -------------------------------------------------------- parameter sets
  void scaling_list(Bits& b, uint8_t* out, int size, const uint8_t* fallback, const uint8_t* default_zz) {
    const uint8_t* scan = size == 16 ? zigzag4x4 : zigzag8x8;
    if (!b.bit()) {
      memcpy(out, fallback, size);
      return;
    }
    int last = 8, next = 8;
    for (int j = 0; j < size; ++j) {
      if (next != 0) {
        int delta = b.se();
        next = (last + delta + 256) & 255;
        if (j == 0 && next == 0) {
          for (int k = 0; k < size; ++k) out[scan[k]] = default_zz[k];
          return;
        }
      }
      last = next == 0 ? last : next;
      out[scan[j]] = static_cast<uint8_t>(last);
    }
  }
  void scaling_matrices(Bits& b, bool want8, uint8_t s4[6][16], uint8_t s8[2][64], const uint8_t fb4[6][16],
                        const uint8_t fb8[2][64], bool sps_fallback) {
    uint8_t d4i[16], d4p[16], d8i[64], d8p[64];
    for (int k = 0; k < 16; ++k) {
      d4i[zigzag4x4[k]] = default_4x4_intra[k];
      d4p[zigzag4x4[k]] = default_4x4_inter[k];
    }
    for (int k = 0; k < 64; ++k) {
      d8i[zigzag8x8[k]] = default_8x8_intra[k];
      d8p[zigzag8x8[k]] = default_8x8_inter[k];
    }
    const uint8_t* f0 = sps_fallback ? fb4[0] : d4i;
    const uint8_t* f3 = sps_fallback ? fb4[3] : d4p;
    const uint8_t* f6 = sps_fallback ? fb8[0] : d8i;
    const uint8_t* f7 = sps_fallback ? fb8[1] : d8p;
    scaling_list(b, s4[0], 16, f0, default_4x4_intra);
    scaling_list(b, s4[1], 16, s4[0], default_4x4_intra);
    scaling_list(b, s4[2], 16, s4[1], default_4x4_intra);
    scaling_list(b, s4[3], 16, f3, default_4x4_inter);
    scaling_list(b, s4[4], 16, s4[3], default_4x4_inter);
    scaling_list(b, s4[5], 16, s4[4], default_4x4_inter);
    if (want8) {
      scaling_list(b, s8[0], 64, f6, default_8x8_intra);
      scaling_list(b, s8[1], 64, f7, default_8x8_inter);
    } else {
      memcpy(s8[0], f6, 64);
      memcpy(s8[1], f7, 64);
    }
  }

  int max_dpb_frames(const Sps& s) const {
    static const int lv[][2] = {{10, 396},   {11, 900},   {12, 2376},  {13, 2376},   {20, 2376},  {21, 4752},
                                {22, 8100},  {30, 8100},  {31, 18000}, {32, 20480},  {40, 32768}, {41, 32768},
                                {42, 34816}, {50, 110400}, {51, 184320}, {52, 184320}};
    int mbs = 184320;
    for (auto& l : lv)
      if (l[0] == s.level_idc) mbs = l[1];
    if (s.level_idc == 11 && (s.constraint_flags & 0x10)) mbs = 396;
    return iclip(1, 16, mbs / (s.mb_width * s.mb_height));
  }

  bool parse_sps(Bits& b) {
    Sps s;
    memset(s.scaling4, 16, sizeof(s.scaling4));
    memset(s.scaling8, 16, sizeof(s.scaling8));
    s.profile_idc = b.get(8);
    s.constraint_flags = b.get(8);
    s.level_idc = b.get(8);
    unsigned id = b.ue();
    if (id >= kMaxSps) return fail("sps_id out of range");
    int p = s.profile_idc;
    if (p == 100 || p == 110 || p == 122 || p == 244 || p == 44 || p == 83 || p == 86 || p == 118 || p == 128 ||
        p == 138 || p == 139 || p == 134 || p == 135) {
      int chroma_format_idc = b.ue();
      if (chroma_format_idc == 3) b.bit();
      int bdl = b.ue(), bdc = b.ue(), bypass = b.bit();
      if (chroma_format_idc != 1 || bdl || bdc || bypass)
        return fail("unsupported stream: only 8-bit 4:2:0 without transform bypass is decoded");
      s.scaling_matrix_present = b.bit();
      if (s.scaling_matrix_present) {
        uint8_t flat4[6][16], flat8[2][64];
        memset(flat4, 16, sizeof(flat4));
        memset(flat8, 16, sizeof(flat8));
        scaling_matrices(b, true, s.scaling4, s.scaling8, flat4, flat8, false);
      }
    }
    // 7.4.2.1.1 ranges; unchecked values used to drive Bits::get(n) with absurd n
    // (stack overflow / backwards cursor on a crafted avcC).
    unsigned l2fn = b.ue();
    if (l2fn > 12) return fail("SPS: log2_max_frame_num_minus4 out of range");
    s.log2_max_frame_num = static_cast<int>(l2fn) + 4;
    s.poc_type = b.ue();
    if (s.poc_type == 0) {
      unsigned l2poc = b.ue();
      if (l2poc > 12) return fail("SPS: log2_max_pic_order_cnt_lsb_minus4 out of range");
      s.log2_max_poc_lsb = static_cast<int>(l2poc) + 4;
    } else if (s.poc_type == 1) {
      s.delta_pic_order_always_zero = b.bit();
      s.offset_for_non_ref_pic = b.se();
      s.offset_for_top_to_bottom = b.se();
      s.num_ref_frames_in_poc_cycle = b.ue();
      if (s.num_ref_frames_in_poc_cycle > 255) return fail("bad poc cycle");
      for (int i = 0; i < s.num_ref_frames_in_poc_cycle; ++i) s.offset_for_ref_frame[i] = b.se();
    } else if (s.poc_type != 2) {
      return fail("bad pic_order_cnt_type");
    }
    unsigned nref = b.ue();
    if (nref > 16) return fail("SPS: max_num_ref_frames out of range");
    s.max_num_ref_frames = static_cast<int>(nref);
    b.bit();  // gaps_in_frame_num_value_allowed_flag
    unsigned mbw = b.ue(), mbh = b.ue();
    if (mbw >= 512 || mbh >= 512) return fail("picture size out of range");
    s.mb_width = static_cast<int>(mbw) + 1;
    s.mb_height = static_cast<int>(mbh) + 1;
    if (!b.bit()) return fail("unsupported stream: interlaced coding (frame_mbs_only_flag = 0)");
    s.direct_8x8_inference = b.bit();
    if (b.bit()) {
      // crop offsets are in chroma units (x2 for 4:2:0); the sum must leave at least one luma column/row (7.4.2.1.1)
      uint64_t cl = b.ue(), cr = b.ue(), ct = b.ue(), cb = b.ue();
      if (2 * (cl + cr) >= 16ull * s.mb_width || 2 * (ct + cb) >= 16ull * s.mb_height)
        return fail("SPS: frame cropping exceeds the picture");
      s.crop_left = static_cast<int>(2 * cl);
      s.crop_right = static_cast<int>(2 * cr);
      s.crop_top = static_cast<int>(2 * ct);
      s.crop_bottom = static_cast<int>(2 * cb);
    }
    if (b.bit()) {  // VUI
      if (b.bit() && b.get(8) == 255) b.get(32);
      if (b.bit()) b.bit();
      if (b.bit()) {
        b.get(4);
        if (b.bit()) b.get(24);
      }
      if (b.bit()) {
        b.ue();
        b.ue();
      }
      if (b.bit()) {
        b.get(32);
        b.get(32);
        b.bit();
      }
      bool hrd = false;
      for (int pass = 0; pass < 2; ++pass) {
        if (!b.bit()) continue;
        hrd = true;
        int cpb = b.ue() + 1;
        b.get(8);
        for (int i = 0; i < cpb && !b.overrun; ++i) {
          b.ue();
          b.ue();
          b.bit();
        }
        b.get(20);
      }
      if (hrd) b.bit();
      b.bit();
      s.bitstream_restriction = b.bit();
      if (s.bitstream_restriction) {
        b.bit();
        b.ue();
        b.ue();
        b.ue();
        b.ue();
        s.num_reorder_frames = b.ue();
        b.ue();
      }
    }
    if (b.overrun) return fail("truncated SPS");
    if (s.mb_width > 512 || s.mb_height > 512 || s.mb_width < 1 || s.mb_height < 1)
      return fail("picture size out of range");
    s.valid = true;
    sps[id] = s;
    return true;
  }

  bool parse_pps(Bits& b) {
    Pps p;
    unsigned id = b.ue();
    if (id >= kMaxPps) return fail("pps_id out of range");
    p.sps_id = b.ue();
    if (p.sps_id >= kMaxSps || !sps[p.sps_id].valid) return fail("PPS refers to a missing SPS");
    const Sps& s = sps[p.sps_id];
    p.cabac = b.bit();
    p.bottom_field_pic_order_present = b.bit();
    if (b.ue() != 0) return fail("unsupported stream: slice groups (FMO)");
    unsigned nr0 = b.ue(), nr1 = b.ue();
    if (nr0 > 31 || nr1 > 31) return fail("PPS: num_ref_idx_default_active out of range");
    p.num_ref_idx_default[0] = static_cast<int>(nr0) + 1;
    p.num_ref_idx_default[1] = static_cast<int>(nr1) + 1;
    p.weighted_pred = b.bit();
    p.weighted_bipred_idc = b.get(2);
    if (p.weighted_bipred_idc > 2) return fail("PPS: weighted_bipred_idc out of range");
    int iq = b.se();
    if (iq < -26 || iq > 25) return fail("PPS: pic_init_qp out of range");
    p.init_qp = 26 + iq;
    b.se();
    p.chroma_qp_offset[0] = b.se();
    if (p.chroma_qp_offset[0] < -12 || p.chroma_qp_offset[0] > 12) return fail("PPS: chroma_qp_index_offset out of range");
    p.deblocking_control_present = b.bit();
    p.constrained_intra_pred = b.bit();
    p.redundant_pic_cnt_present = b.bit();
    p.chroma_qp_offset[1] = p.chroma_qp_offset[0];
    memcpy(p.scaling4, s.scaling4, sizeof(p.scaling4));
    memcpy(p.scaling8, s.scaling8, sizeof(p.scaling8));
    if (b.more_rbsp_data()) {
      p.transform_8x8_mode = b.bit();
      if (b.bit())
        scaling_matrices(b, p.transform_8x8_mode, p.scaling4, p.scaling8, s.scaling4, s.scaling8,
                         s.scaling_matrix_present);
      p.chroma_qp_offset[1] = b.se();
      if (p.chroma_qp_offset[1] < -12 || p.chroma_qp_offset[1] > 12)
        return fail("PPS: second_chroma_qp_index_offset out of range");
    }
    if (b.overrun) return fail("truncated PPS");
    p.valid = true;
    pps[id] = p;
    return true;
  }

  // ------------------------------------------------------------------ DPB
  void release_unused() {
    for (Pic& p : dpb) {
      if (p.in_use && &p != cur && !p.ref && !p.needed_for_output) {
        p.in_use = false;
        p.slot = -1;
        p.d.reset();  // the surface goes back to the sink with the last holder (PicPool::Recycler)
      }
    }
  }
  Pic* grab_pic() {
    for (Pic& p : dpb)
      if (!p.in_use) {
        p.in_use = true;
        p.uid = ++next_uid;
        p.ref = 0;
        p.needed_for_output = false;
        p.long_term_idx = -1;
        p.slot = -1;
        p.d = pool->grab();
        p.d->uid = p.uid;
        return &p;
      }
    return nullptr;
  }
  void emit(Pic* p) {
    sink->output_picture(p->slot);
    if (p->slot < 0 && !wanted.empty() &&
        std::binary_search(wanted.begin(), wanted.end(), outputs_emitted)) {
      // A picture was skipped on the prediction that nobody wants its output
      // position, and the prediction was wrong.  Never hand out garbage.
      fail("sparse decode mispredicted the display position of a skipped picture");
    }
    ++outputs_emitted;
    p->needed_for_output = false;
  }
  Pic* lowest_poc_pending(int* count) {
    Pic* best = nullptr;
    int n = 0;
    for (Pic& p : dpb) {
      if (!p.in_use || !p.needed_for_output || &p == cur) continue;
      ++n;
      if (!best || p.poc < best->poc) best = &p;
    }
    if (count) *count = n;
    return best;
  }
  void output_all() {
    while (Pic* p = lowest_poc_pending(nullptr)) emit(p);
    release_unused();
  }
  int reorder_depth() const {
    if (asps->poc_type == 2) return 0;
    if (asps->bitstream_restriction && asps->num_reorder_frames >= 0) return asps->num_reorder_frames;
    return const_cast<Impl*>(this)->max_dpb_frames(*asps);
  }

  // ------------------------------------------------------------------ POC
  void compute_poc(Pic* c) {
    const Sps& s = *asps;
    const int max_frame_num = 1 << s.log2_max_frame_num;
    int fno;
    if (sl.is_idr) fno = 0;
    else if (prev_frame_num > sl.frame_num) fno = prev_frame_num_offset + max_frame_num;
    else fno = prev_frame_num_offset;
    int top, bottom;
    if (s.poc_type == 0) {
      const int max_lsb = 1 << s.log2_max_poc_lsb;
      if (sl.is_idr) prev_poc_msb = prev_poc_lsb = 0;
      int msb;
      if (sl.poc_lsb < prev_poc_lsb && prev_poc_lsb - sl.poc_lsb >= max_lsb / 2) msb = prev_poc_msb + max_lsb;
      else if (sl.poc_lsb > prev_poc_lsb && sl.poc_lsb - prev_poc_lsb > max_lsb / 2) msb = prev_poc_msb - max_lsb;
      else msb = prev_poc_msb;
      top = msb + sl.poc_lsb;
      bottom = top + sl.delta_poc_bottom;
      if (sl.nal_ref_idc) {
        prev_poc_msb = msb;
        prev_poc_lsb = sl.poc_lsb;
      }
    } else if (s.poc_type == 1) {
      int abs_frame_num = s.num_ref_frames_in_poc_cycle ? fno + sl.frame_num : 0;
      if (!sl.nal_ref_idc && abs_frame_num > 0) abs_frame_num--;
      int per_cycle = 0;
      for (int i = 0; i < s.num_ref_frames_in_poc_cycle; ++i) per_cycle += s.offset_for_ref_frame[i];
      int expected = 0;
      if (abs_frame_num > 0) {
        int cycle = (abs_frame_num - 1) / s.num_ref_frames_in_poc_cycle;
        int in_cycle = (abs_frame_num - 1) % s.num_ref_frames_in_poc_cycle;
        expected = cycle * per_cycle;
        for (int i = 0; i <= in_cycle; ++i) expected += s.offset_for_ref_frame[i];
      }
      if (!sl.nal_ref_idc) expected += s.offset_for_non_ref_pic;
      top = expected + sl.delta_poc[0];
      bottom = top + s.offset_for_top_to_bottom + sl.delta_poc[1];
    } else {
      top = bottom = sl.is_idr ? 0 : 2 * (fno + sl.frame_num) - (sl.nal_ref_idc ? 0 : 1);
    }
    c->poc = imin(top, bottom);
    prev_frame_num_offset = fno;
    prev_frame_num = sl.frame_num;
  }

  // ----------------------------------------------------------- ref lists
  void init_ref_lists() {
    const int max_frame_num = 1 << asps->log2_max_frame_num;
    std::vector<Pic*> st, lt;
    for (Pic& p : dpb) {
      if (!p.in_use || &p == cur) continue;
      if (p.ref == 1) {
        p.frame_num_wrap = p.frame_num > sl.frame_num ? p.frame_num - max_frame_num : p.frame_num;
        p.pic_num = p.frame_num_wrap;
        st.push_back(&p);
      } else if (p.ref == 2) {
        p.pic_num = p.long_term_idx;
        lt.push_back(&p);
      }
    }
    std::sort(lt.begin(), lt.end(), [](Pic* a, Pic* b) { return a->long_term_idx < b->long_term_idx; });
    for (int l = 0; l < 2; ++l) memset(sl.ref_list[l], 0, sizeof(sl.ref_list[l]));
    if (sl.type == 0) {
      std::sort(st.begin(), st.end(), [](Pic* a, Pic* b) { return a->pic_num > b->pic_num; });
      int n = 0;
      for (Pic* p : st) if (n < kMaxRefs) sl.ref_list[0][n++] = p;
      for (Pic* p : lt) if (n < kMaxRefs) sl.ref_list[0][n++] = p;
    } else if (sl.type == 1) {
      std::vector<Pic*> before, after;
      for (Pic* p : st) (p->poc < cur->poc ? before : after).push_back(p);
      std::sort(before.begin(), before.end(), [](Pic* a, Pic* b) { return a->poc > b->poc; });
      std::sort(after.begin(), after.end(), [](Pic* a, Pic* b) { return a->poc < b->poc; });
      int n0 = 0, n1 = 0;
      for (Pic* p : before) if (n0 < kMaxRefs) sl.ref_list[0][n0++] = p;
      for (Pic* p : after) if (n0 < kMaxRefs) sl.ref_list[0][n0++] = p;
      for (Pic* p : lt) if (n0 < kMaxRefs) sl.ref_list[0][n0++] = p;
      for (Pic* p : after) if (n1 < kMaxRefs) sl.ref_list[1][n1++] = p;
      for (Pic* p : before) if (n1 < kMaxRefs) sl.ref_list[1][n1++] = p;
      for (Pic* p : lt) if (n1 < kMaxRefs) sl.ref_list[1][n1++] = p;
      if (n1 > 1 && n0 == n1 && !memcmp(sl.ref_list[0], sl.ref_list[1], n1 * sizeof(Pic*)))
        std::swap(sl.ref_list[1][0], sl.ref_list[1][1]);
    }
  }

  bool modify_ref_list(Bits& b, int list) {
    const int max_pic_num = 1 << asps->log2_max_frame_num;
    int pred = sl.frame_num, idx = 0;
    const int n = sl.num_ref_idx[list];
    Pic** L = sl.ref_list[list];
    for (;;) {
      unsigned op = b.ue();
      if (op == 3) break;
      if (op > 3 || b.overrun || idx >= n) return fail("bad ref_pic_list_modification");
      Pic* pic = nullptr;
      if (op < 2) {
        int diff = b.ue() + 1;
        if (op == 0) {
          pred -= diff;
          if (pred < 0) pred += max_pic_num;
        } else {
          pred += diff;
          if (pred >= max_pic_num) pred -= max_pic_num;
        }
        int pic_num = pred > sl.frame_num ? pred - max_pic_num : pred;
        for (Pic& p : dpb)
          if (p.in_use && &p != cur && p.ref == 1 && p.pic_num == pic_num) pic = &p;
      } else {
        int lt_num = b.ue();
        for (Pic& p : dpb)
          if (p.in_use && &p != cur && p.ref == 2 && p.long_term_idx == lt_num) pic = &p;
      }
      if (!pic) return fail("ref_pic_list_modification names a missing picture");
      for (int c = n; c > idx; --c) L[c] = L[c - 1];
      L[idx++] = pic;
      int w = idx;
      for (int c = idx; c <= n; ++c)
        if (L[c] != pic) L[w++] = L[c];
    }
    return true;
  }

  // --------------------------------------------------------- slice header
  // returns 0 ok, 1 skip this NAL, -1 error
  int parse_slice_header(Bits& b, int nal_ref_idc, int nal_type) {
    sl = SliceHdr();
    sl.nal_ref_idc = nal_ref_idc;
    sl.is_idr = nal_type == 5;
    sl.first_mb = b.ue();
    unsigned st = b.ue();
    if (st > 9) return fail("bad slice_type") ? 0 : -1;
    st %= 5;
    if (st > 2) return fail("unsupported stream: SP/SI slices") ? 0 : -1;
    sl.type = st;
    sl.pps_id = b.ue();
    if (sl.pps_id >= kMaxPps || !pps[sl.pps_id].valid) return fail("slice refers to a missing PPS") ? 0 : -1;
    const Pps* p = &pps[sl.pps_id];
    const Sps* s = &sps[p->sps_id];
    sl.frame_num = b.get(s->log2_max_frame_num);
    if (sl.is_idr) b.ue();
    if (s->poc_type == 0) {
      sl.poc_lsb = b.get(s->log2_max_poc_lsb);
      if (p->bottom_field_pic_order_present) sl.delta_poc_bottom = b.se();
    } else if (s->poc_type == 1 && !s->delta_pic_order_always_zero) {
      sl.delta_poc[0] = b.se();
      if (p->bottom_field_pic_order_present) sl.delta_poc[1] = b.se();
    }
    if (p->redundant_pic_cnt_present && b.ue() != 0) return 1;
    if (sl.type == 1) sl.direct_spatial = b.bit();
    sl.num_ref_idx[0] = p->num_ref_idx_default[0];
    sl.num_ref_idx[1] = p->num_ref_idx_default[1];
    if (sl.type != 2) {
      if (b.bit()) {
        sl.num_ref_idx[0] = b.ue() + 1;
        if (sl.type == 1) sl.num_ref_idx[1] = b.ue() + 1;
      }
      if (sl.num_ref_idx[0] > kMaxRefs || sl.num_ref_idx[1] > kMaxRefs)
        return fail("num_ref_idx_active out of range") ? 0 : -1;
    }
    if (sl.type != 1) sl.num_ref_idx[1] = 0;
    if (sl.type == 2) sl.num_ref_idx[0] = 0;

    if (sl.first_mb != 0 && skipping_picture) return 1;
    skipping_picture = false;
    if (sl.first_mb == 0 || !pic_in_progress) {
      if (pic_in_progress && !finish_picture()) return -1;
      apps = p;
      asps = s;
      int r = start_picture();
      if (r != 0) return r;
    } else if (p->sps_id != apps->sps_id) {
      return fail("SPS changed inside a picture") ? 0 : -1;
    } else {
      apps = p;
    }
    if (sl.type != 2) {
      init_ref_lists();
      if (b.bit() && !modify_ref_list(b, 0)) return -1;
      if (sl.type == 1 && b.bit() && !modify_ref_list(b, 1)) return -1;
      for (int l = 0; l < 2; ++l)
        for (int i = 0; i < sl.num_ref_idx[l]; ++i) {
          if (!sl.ref_list[l][i]) return fail("reference list has a missing picture") ? 0 : -1;
          if (sl.ref_list[l][i]->slot < 0)
            return fail("a skipped picture is used as a reference") ? 0 : -1;
          const Pic* rp = sl.ref_list[l][i];
          RefSnap& rs = sl.ref[l][i];
          rs.d = rp->d.get();
          rs.slot = rp->slot;
          rs.uid = rp->uid;
          rs.poc = rp->poc;
          rs.marking = rp->ref;
        }
    }
    sl.cur_poc = cur->poc;
    sl.use_weight = 0;
    if ((p->weighted_pred && sl.type == 0) || (p->weighted_bipred_idc == 1 && sl.type == 1)) {
      sl.use_weight = 1;
      sl.luma_log2_denom = b.ue();
      sl.chroma_log2_denom = b.ue();
      if (sl.luma_log2_denom > 7 || sl.chroma_log2_denom > 7) return fail("bad weight denominator") ? 0 : -1;
      for (int l = 0; l < (sl.type == 1 ? 2 : 1); ++l)
        for (int i = 0; i < sl.num_ref_idx[l]; ++i) {
          sl.luma_weight[l][i] = 1 << sl.luma_log2_denom;
          sl.luma_offset[l][i] = 0;
          if (b.bit()) {
            sl.luma_weight[l][i] = b.se();
            sl.luma_offset[l][i] = b.se();
            if (sl.luma_weight[l][i] < -128 || sl.luma_weight[l][i] > 127 || sl.luma_offset[l][i] < -128 ||
                sl.luma_offset[l][i] > 127)
              return fail("prediction weight out of range") ? 0 : -1;
          }
          for (int c = 0; c < 2; ++c) {
            sl.chroma_weight[l][i][c] = 1 << sl.chroma_log2_denom;
            sl.chroma_offset[l][i][c] = 0;
          }
          if (b.bit())
            for (int c = 0; c < 2; ++c) {
              sl.chroma_weight[l][i][c] = b.se();
              sl.chroma_offset[l][i][c] = b.se();
              if (sl.chroma_weight[l][i][c] < -128 || sl.chroma_weight[l][i][c] > 127 ||
                  sl.chroma_offset[l][i][c] < -128 || sl.chroma_offset[l][i][c] > 127)
                return fail("prediction weight out of range") ? 0 : -1;
            }
        }
    } else if (p->weighted_bipred_idc == 2 && sl.type == 1) {
      sl.use_weight = 2;
    }
    if (nal_ref_idc) {
      if (sl.is_idr) {
        b.bit();
        sl.long_term_reference_flag = b.bit();
      } else {
        sl.adaptive_marking = b.bit();
        if (sl.adaptive_marking) {
          for (;;) {
            unsigned op = b.ue();
            if (op == 0) break;
            if (op > 6 || sl.mmco_count >= 64 || b.overrun) return fail("bad MMCO") ? 0 : -1;
            int a = 0, c = 0;
            if (op == 1 || op == 3) a = b.ue();
            if (op == 2) a = b.ue();
            if (op == 3 || op == 6) c = b.ue();
            if (op == 4) a = b.ue();
            sl.mmco[sl.mmco_count].op = op;
            sl.mmco[sl.mmco_count].a = a;
            sl.mmco[sl.mmco_count].b = c;
            sl.mmco_count++;
          }
        }
      }
    }
    if (p->cabac && sl.type != 2) {
      sl.cabac_init_idc = b.ue();
      if (sl.cabac_init_idc > 2) return fail("bad cabac_init_idc") ? 0 : -1;
    }
    sl.qp = p->init_qp + b.se();
    if (sl.qp < 0 || sl.qp > 51) return fail("slice QP out of range") ? 0 : -1;
    if (p->deblocking_control_present) {
      sl.disable_deblock = b.ue();
      if (sl.disable_deblock > 2) return fail("bad disable_deblocking_filter_idc") ? 0 : -1;
      if (sl.disable_deblock != 1) {
        sl.alpha_offset = b.se() * 2;
        sl.beta_offset = b.se() * 2;
      }
    }
    if (b.overrun) return fail("truncated slice header") ? 0 : -1;
    if (num_slices >= bufs.max_slices) return fail("too many slices in one picture") ? 0 : -1;
    slice_index = num_slices + 1;
    fill_slice_params(&bufs.slices[num_slices]);
    ++num_slices;
    slice_type_mask |= sl.type == 0 ? 1 : (sl.type == 1 ? 2 : 4);
    if (sl.disable_deblock != 1) any_deblock = true;
    return 0;
  }

  void fill_slice_params(SliceParams* sp) {
    memset(sp, 0, sizeof(*sp));
    sp->alpha_offset = static_cast<int8_t>(sl.alpha_offset);
    sp->beta_offset = static_cast<int8_t>(sl.beta_offset);
    sp->weight_mode = static_cast<uint8_t>(sl.use_weight);
    sp->luma_log2_denom = static_cast<uint8_t>(sl.luma_log2_denom);
    sp->chroma_log2_denom = static_cast<uint8_t>(sl.chroma_log2_denom);
    sp->cur_poc = cur->poc;
    for (int l = 0; l < 2; ++l)
      for (int i = 0; i < sl.num_ref_idx[l]; ++i) {
        sp->luma_weight[l][i] = static_cast<int16_t>(sl.luma_weight[l][i]);
        sp->luma_offset[l][i] = static_cast<int16_t>(sl.luma_offset[l][i]);
        for (int c = 0; c < 2; ++c) {
          sp->chroma_weight[l][i][c] = static_cast<int16_t>(sl.chroma_weight[l][i][c]);
          sp->chroma_offset[l][i][c] = static_cast<int16_t>(sl.chroma_offset[l][i][c]);
        }
        sp->ref_poc[l][i] = sl.ref_list[l][i]->poc;
        sp->ref_long[l][i] = sl.ref_list[l][i]->ref == 2;
      }
  }

  // returns 0 ok, 1 picture skipped on purpose, -1 error
  int start_picture() {
    if (workers && (mbw != asps->mb_width || mbh != asps->mb_height) && !pipe_drain()) return -1;  // the sink re-makes its buffers
    mbw = asps->mb_width;
    mbh = asps->mb_height;
    // Sparse decode, the tail of an interval: a picture at decoding position d is displayed at d - reorder depth or
    // later, so once d exceeds the last wanted position by more than the reorder depth neither it nor anything that
    // follows can be wanted or be referred to by a wanted picture.  Whatever the DPB still holds goes out first (all
    // wanted positions are among it or out already), then every further sample is one void output position and no
    // work at all.  (The reference's DecoderAutomata feeds every sample of an interval, decoder_automata.cpp:287-331;
    // this is where that stops costing anything.)
    const uint64_t ordinal = pictures_started++;
    bool useless = false;
    if (!wanted.empty()) {
      // the same inside an interval of several GOPs when the caller has said where the sync samples are: what lies
      // between a GOP's last wanted position (+ reorder depth) and the next sync sample serves nobody - the GOPs
      // are closed, which is what starting an interval at a sync sample assumes in the first place
      uint64_t g0 = 0, g1 = UINT64_MAX;
      if (!keyframes.empty()) {
        auto it = std::upper_bound(keyframes.begin(), keyframes.end(), ordinal);
        if (it != keyframes.end()) g1 = *it;
        if (it != keyframes.begin()) g0 = *(it - 1);
      }
      auto hi = std::lower_bound(wanted.begin(), wanted.end(), g1);
      const bool any = hi != wanted.begin() && *(hi - 1) >= g0;
      useless = !any || ordinal > *(hi - 1) + static_cast<uint64_t>(reorder_depth());
    }
    if (useless) {
      cur = nullptr;
      output_all();
      for (Pic& p : dpb) p.ref = 0;
      release_unused();
      sink->output_picture(-1);
      ++outputs_emitted;
      pic_in_progress = false;
      skipping_picture = true;
      ++owner->pictures_skipped_;
      return 1;
    }
    if (sl.is_idr) {
      cur = nullptr;
      output_all();
      for (Pic& p : dpb) p.ref = 0;
      release_unused();
      prev_frame_num = prev_frame_num_offset = 0;
      max_long_term_frame_idx = -1;
    } else if (prev_ref_has_mmco5) {
      prev_frame_num = prev_frame_num_offset = 0;
    }
    prev_ref_has_mmco5 = false;
    release_unused();
    Pic* c = grab_pic();
    if (!c) return fail("decoded picture buffer exhausted") ? 0 : -1;
    cur = c;
    c->frame_num = sl.frame_num;
    compute_poc(c);

    // Sparse decode: an unwanted non-reference picture is not decoded.  Its
    // display position is predicted as (already output) + (pending pictures
    // that precede it); emit() verifies the prediction.
    if (!wanted.empty() && !sl.nal_ref_idc) {
      uint64_t pos = outputs_emitted;
      for (Pic& p : dpb)
        if (p.in_use && &p != c && p.needed_for_output && p.poc < c->poc) ++pos;
      if (!std::binary_search(wanted.begin(), wanted.end(), pos)) {
        c->slot = -1;
        c->needed_for_output = true;
        cur = nullptr;
        pic_in_progress = false;
        skipping_picture = true;
        ++owner->pictures_skipped_;
        bump();
        return 1;
      }
    }

    PicSetup setup{mbw, mbh, asps->crop_left, asps->crop_right, asps->crop_top, asps->crop_bottom};
    if (workers && !pipe_wait_for_room()) {
      c->in_use = false;
      c->d.reset();
      cur = nullptr;
      return -1;
    }
    bufs = PicBuffers();
    for (;;) {
      if (sink->begin_picture(setup, &bufs, err)) break;
      // every surface held by pictures that are still with the workers: one of them finishing frees one
      if (workers && bufs.no_surface && pipe_wait_one()) {
        err->clear();
        bufs = PicBuffers();
        continue;
      }
      c->in_use = false;
      c->d.reset();
      cur = nullptr;
      return -1;
    }
    c->slot = bufs.slot;
    PicData& d = *c->d;
    d.slot = bufs.slot;
    d.motion = bufs.motion;
    d.pmo = bufs.pmo;
    d.pmx = bufs.pmx;
    d.parsed = !workers;
    num_slices = 0;
    slice_index = 0;
    any_deblock = false;
    slice_type_mask = 0;
    pic_in_progress = true;
    if (workers) pipe_begin_task(c);
    else sd.begin_picture(asps, c->d.get(), bufs);
    return 0;
  }

  void unmark_long_idx(int idx, Pic* except) {
    for (Pic& p : dpb)
      if (p.in_use && &p != except && p.ref == 2 && p.long_term_idx == idx) p.ref = 0;
  }

  void mark_references(Pic* c) {
    if (!sl.nal_ref_idc) return;
    if (sl.is_idr) {
      if (sl.long_term_reference_flag) {
        c->ref = 2;
        c->long_term_idx = 0;
        max_long_term_frame_idx = 0;
      } else {
        c->ref = 1;
        max_long_term_frame_idx = -1;
      }
      return;
    }
    bool cur_is_long = false;
    if (sl.adaptive_marking) {
      for (int k = 0; k < sl.mmco_count; ++k) {
        const int op = sl.mmco[k].op, a = sl.mmco[k].a, bb = sl.mmco[k].b;
        if (op == 1 || op == 3) {
          const int pic_num = sl.frame_num - (a + 1);
          for (Pic& p : dpb) {
            if (!p.in_use || &p == c || p.ref != 1 || p.pic_num != pic_num) continue;
            if (op == 1) {
              p.ref = 0;
            } else {
              unmark_long_idx(bb, &p);
              p.ref = 2;
              p.long_term_idx = bb;
            }
          }
        } else if (op == 2) {
          for (Pic& p : dpb)
            if (p.in_use && &p != c && p.ref == 2 && p.long_term_idx == a) p.ref = 0;
        } else if (op == 4) {
          max_long_term_frame_idx = a - 1;
          for (Pic& p : dpb)
            if (p.in_use && &p != c && p.ref == 2 && p.long_term_idx > a - 1) p.ref = 0;
        } else if (op == 5) {
          for (Pic& p : dpb)
            if (&p != c) p.ref = 0;
          max_long_term_frame_idx = -1;
          prev_ref_has_mmco5 = true;
          prev_poc_msb = prev_poc_lsb = 0;
          c->frame_num = 0;
        } else if (op == 6) {
          unmark_long_idx(a, c);
          c->ref = 2;
          c->long_term_idx = a;
          cur_is_long = true;
        }
      }
    } else {
      int n = 0;
      Pic* oldest = nullptr;
      for (Pic& p : dpb) {
        if (!p.in_use || &p == c || !p.ref) continue;
        ++n;
        if (p.ref == 1 && (!oldest || p.frame_num_wrap < oldest->frame_num_wrap)) oldest = &p;
      }
      if (n >= imax(asps->max_num_ref_frames, 1) && oldest) oldest->ref = 0;
    }
    if (!cur_is_long) c->ref = 1;
  }

  void bump() {
    const int depth = reorder_depth();
    for (;;) {
      int n = 0;
      Pic* p = lowest_poc_pending(&n);
      if (!p || n <= depth) break;
      emit(p);
    }
    release_unused();
  }

  bool finish_picture() {
    if (!pic_in_progress) return true;
    pic_in_progress = false;
    Pic* c = cur;
    PicDone done;
    memset(&done, 0, sizeof(done));
    done.slot = c->slot;
    done.cookie = bufs.cookie;
    done.num_slices = num_slices;
    done.any_deblock = any_deblock;
    done.slice_type_mask = slice_type_mask;
    done.scaling4 = apps->scaling4;
    done.scaling8 = apps->scaling8;
    done.num_ref_slots = 0;
    for (Pic& p : dpb)
      if (p.in_use && &p != c && p.ref && p.slot >= 0 && done.num_ref_slots < 32)
        done.ref_slots[done.num_ref_slots++] = p.slot;
    if (workers) {
      pipe_submit_task(done);  // the slice data is decoded, and the picture handed to the sink, by the workers
    } else {
      const bool ok = sd.finish_picture(&done);
      owner->slices_ += sd.slices;
      owner->slices_clean_end_ += sd.slices_clean_end;
      sd.slices = sd.slices_clean_end = 0;
      if (!ok) return fail(sd.err.c_str());
      if (!sink->end_picture(done, err)) return false;
    }
    ++owner->pictures_decoded_;

    const int max_frame_num = 1 << asps->log2_max_frame_num;
    for (Pic& p : dpb)
      if (p.in_use && &p != c && p.ref == 1) {
        p.frame_num_wrap = p.frame_num > sl.frame_num ? p.frame_num - max_frame_num : p.frame_num;
        p.pic_num = p.frame_num_wrap;
      }
    mark_references(c);
    c->needed_for_output = true;
    if (prev_ref_has_mmco5) {
      cur = nullptr;
      c->needed_for_output = false;
      output_all();
      c->needed_for_output = true;
      c->poc = 0;
    }
    cur = nullptr;
    bump();
    return err->empty();
  }

  bool decode_nal(const uint8_t* nal, size_t n) {
    if (n < 1) return true;
    const int ref_idc = (nal[0] >> 5) & 3, type = nal[0] & 31;
    if (type != 1 && type != 5 && type != 7 && type != 8) return true;
    // The longest slice header: two reference list modifications of 33 entries, prediction weights of 2 x 32
    // references, 66 marking operations - about 1.2 KB.
    constexpr size_t kHeaderReach = 2048;
    const bool slice = type == 1 || type == 5;
    unescape_begin(nal + 1, n - 1);
    int rn = unescape_upto(slice ? kHeaderReach : n - 1);
    Bits b;
    b.buf = rbsp.data();
    b.size_bits = rn * 8;
    if (type == 7) return parse_sps(b);
    if (type == 8) return parse_pps(b);
    const int r = parse_slice_header(b, ref_idc, type);
    if (r < 0 || !err->empty()) return false;
    if (r > 0) return true;
    if (workers) {  // (a worker unescapes the whole unit; the header's bit position is the same there)
      pipe_add_slice(b);
      return true;
    }
    if (ue.i < ue.n) {  // the slice data
      rn = unescape_rest();
      b.size_bits = rn * 8;
    }
    if (!sd.decode_slice(sl, apps, slice_index, b)) return fail(sd.err.c_str());
    return true;
  }

  void abort_picture() {
    if (pic_in_progress && cur && cur->slot >= 0) {  // begin_picture without an end_picture: the sink takes its buffers back
      PicDone gone;
      memset(&gone, 0, sizeof(gone));
      gone.slot = cur->slot;
      gone.cookie = bufs.cookie;
      sink->abandon_picture(gone);
    }
    pic_in_progress = false;
    if (workers) pipe_drop_task();
    if (cur) {
      cur->in_use = false;
      cur->slot = -1;
      cur->d.reset();
      cur = nullptr;
    }
  }
};

// ===========================================================================
Parser::Parser(PictureSink* sink, WorkerPool* workers, int max_in_flight) : impl_(new Impl()) {
  impl_->owner = this;
  impl_->sink = sink;
  impl_->err = &error_;
  impl_->pool = std::make_shared<PicPool>();
  impl_->pool->sink = sink;
  if (workers && workers->size() > 0) {
    impl_->workers = workers;
    impl_->max_in_flight = max_in_flight > 0 ? max_in_flight : 2 * workers->size();
  }
}
Parser::~Parser() {
  if (impl_->workers) {
    impl_->pipe_drop_task();
    std::string ignored;
    impl_->err = &ignored;
    (void)impl_->pipe_drain();  // no task may outlive the object it points into
    impl_->err = &error_;
  }
  delete impl_;
}

int Parser::width() const {
  const Sps* s = impl_->asps;
  if (!s)
    for (const Sps& c : impl_->sps)
      if (c.valid) { s = &c; break; }
  return s ? s->mb_width * 16 - s->crop_left - s->crop_right : 0;
}
int Parser::height() const {
  const Sps* s = impl_->asps;
  if (!s)
    for (const Sps& c : impl_->sps)
      if (c.valid) { s = &c; break; }
  return s ? s->mb_height * 16 - s->crop_top - s->crop_bottom : 0;
}

bool Parser::configure(const uint8_t* avcc, size_t size) {
  error_.clear();
  Impl& d = *impl_;
  if (size < 7 || avcc[0] != 1) return d.fail("extradata is not an avcC record");
  d.nal_length_size = (avcc[4] & 3) + 1;
  size_t pos = 5;
  for (int pass = 0; pass < 2; ++pass) {
    if (pos >= size) return d.fail("truncated avcC");
    int count = pass == 0 ? (avcc[pos] & 31) : avcc[pos];
    pos++;
    for (int i = 0; i < count; ++i) {
      if (pos + 2 > size) return d.fail("truncated avcC");
      size_t len = (avcc[pos] << 8) | avcc[pos + 1];
      pos += 2;
      if (pos + len > size) return d.fail("truncated avcC");
      if (!d.decode_nal(avcc + pos, len)) return false;
      pos += len;
    }
  }
  return true;
}

bool Parser::decode(const uint8_t* sample, size_t size) {
  error_.clear();
  Impl& d = *impl_;
  if (d.workers) {  // a picture that failed with the workers is reported at the next call
    std::lock_guard<std::mutex> lk(d.pipe_mu);
    if (!d.pipe_check_locked()) return false;
  }
  size_t pos = 0;
  const int ls = d.nal_length_size;
  while (pos + ls <= size) {
    size_t len = 0;
    for (int i = 0; i < ls; ++i) len = (len << 8) | sample[pos + i];
    pos += ls;
    if (len > size - pos) {
      d.abort_picture();
      return d.fail("NAL length exceeds sample");
    }
    if (!d.decode_nal(sample + pos, len)) {
      d.abort_picture();
      return false;
    }
    pos += len;
  }
  d.skipping_picture = false;
  if (!d.finish_picture()) {
    d.abort_picture();
    return false;
  }
  return true;
}

bool Parser::flush() {
  error_.clear();  // a failed decode() has been reported by that call; flush() answers for itself
  Impl& d = *impl_;
  if (d.pic_in_progress && !d.finish_picture()) d.abort_picture();
  if (d.workers) {
    // Does not wait for the workers: the interval's pictures keep decoding while the caller moves on to the next
    // interval (wait_idle() is the barrier).  A failure that is known by now and that the caller has not heard of is
    // reported here, once; a picture of this interval that fails later is reported by the next decode(), flush() or
    // wait_idle().  Pictures of the next interval do not depend on a failed one (it starts with an IDR picture).
    std::unique_lock<std::mutex> lk(d.pipe_mu);
    (void)d.pipe_check_locked();
    ++d.epoch;
  }
  if (d.asps) {
    d.cur = nullptr;
    d.output_all();
    for (Pic& p : d.dpb) p.ref = 0;
    d.release_unused();
  }
  d.outputs_emitted = 0;
  d.pictures_started = 0;
  d.wanted.clear();
  d.keyframes.clear();
  return error_.empty();
}

bool Parser::wait_idle(std::string* error) {
  Impl& d = *impl_;
  if (!d.workers) return true;
  std::unique_lock<std::mutex> lk(d.pipe_mu);
  d.pipe_cv.wait(lk, [&] { return d.pipe_idle_locked(); });
  const bool failed = d.pipe_unreported;
  if (failed && error) *error = d.pipe_error;
  d.pipe_unreported = false;
  return !failed;
}

void Parser::set_keyframes(const std::vector<uint64_t>& ordinals) {
  impl_->keyframes = ordinals;
  std::sort(impl_->keyframes.begin(), impl_->keyframes.end());
}

void Parser::set_wanted(const std::vector<uint64_t>& ordinals) {
  impl_->wanted = ordinals;
  std::sort(impl_->wanted.begin(), impl_->wanted.end());
}

}  // namespace h264
}  // namespace hwang
