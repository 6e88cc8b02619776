// Development aid: formulations of the CABAC arithmetic decoder (9.3.3.2) timed on a REAL bin sequence with no syntax
// around it.  Input: the trace a -DHWANG_CABAC_TRACE build of the parser writes (cabac_trace_init.bin: the contexts
// and the bytes a slice started from; cabac_trace_ctx.bin: the context index of every bin, 0xffff bypass, 0xfffe
// terminate).  Every variant must produce the same bins (checksum) and end in the same state.
// usage: engine_bench init.bin ctx.bin [reps]
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "h264/h264_tables.h"


static std::vector<uint8_t> rd(const char* p) {
  FILE* f = fopen(p, "rb");
  if (!f) { perror(p); exit(2); }
  fseek(f, 0, SEEK_END);
  long n = ftell(f);
  fseek(f, 0, SEEK_SET);
  std::vector<uint8_t> v(n);
  if (fread(v.data(), 1, n, f) != (size_t)n) exit(2);
  fclose(f);
  return v;
}

struct Tables {
  uint8_t next[128][2];
  uint32_t lps4[128];
  uint16_t next2[128];  // next[st][0] | next[st][1] << 8
  uint64_t pack[128];   // lps4 | next2 << 32
  uint64_t lpsn[128];   // per quarter 16 bits: normalised LPS range (9 bits) | its shift << 9
  Tables() {
    for (int st = 0; st < 128; ++st) {
      const int ps = st >> 1, mps = st & 1;
      next[st][0] = (uint8_t)(((ps < 62 ? ps + 1 : ps) << 1) | mps);
      next[st][1] = (uint8_t)((cabac_trans_lps[ps] << 1) | (ps == 0 ? !mps : mps));
      next2[st] = (uint16_t)(next[st][0] | (next[st][1] << 8));
      lps4[st] = 0;
      lpsn[st] = 0;
      for (int q = 0; q < 4; ++q) {
        const uint32_t l = cabac_range_lps[ps][q];
        lps4[st] |= l << (8 * q);
        const int sh = __builtin_clz(l) - 23;
        lpsn[st] |= (uint64_t)((l << sh) | (sh << 9)) << (16 * q);
      }
      pack[st] = lps4[st] | ((uint64_t)next2[st] << 32);
    }
  }
};
static const Tables T;

struct Eng {
  const uint8_t* p;
  const uint8_t* end;
  uint64_t value = 0;
  int extra = -9;
  uint32_t range = 510;
  uint16_t state[1024];
  __attribute__((noinline)) void refill() {
    if (extra >= 0 && p + 4 <= end) {
      const uint32_t w = ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3];
      p += 4;
      value = (value << 32) | w;
      extra += 32;
      return;
    }
    while (extra < 24) {
      uint32_t b = p < end ? *p : 0;
      ++p;
      value = (value << 8) | b;
      extra += 8;
    }
  }
};

// V0: the parser's formulation
template <int V> static inline __attribute__((always_inline)) int bin(Eng& e, uint64_t& value, uint32_t& range, int& extra, int ctx);
template <> inline __attribute__((always_inline)) int bin<0>(Eng& e, uint64_t& value, uint32_t& range, int& extra, int ctx) {
  const uint32_t st = e.state[ctx];
  const uint32_t lps = (T.lps4[st] >> ((range >> 3) & 24)) & 0xff;
  uint32_t r = range - lps;
  const uint64_t scaled = (uint64_t)r << extra;
  const uint64_t lmask = ~(uint64_t)((int64_t)(value - scaled) >> 63);
  value -= scaled & lmask;
  r += (lps - r) & (uint32_t)lmask;
  const uint32_t l = (uint32_t)lmask & 1;
  e.state[ctx] = T.next[st][l];
  const int shift = __builtin_clz(r) - 23;
  range = r << shift;
  extra -= shift;
  return (int)((st & 1) ^ l);
}
// V1: both outcomes prepared, selected by the comparison (conditional moves)
template <> inline __attribute__((always_inline)) int bin<1>(Eng& e, uint64_t& value, uint32_t& range, int& extra, int ctx) {
  const uint32_t st = e.state[ctx];
  const uint32_t q8 = (range >> 3) & 24;
  const uint32_t lps = (T.lps4[st] >> q8) & 0xff;
  const uint32_t en = (uint32_t)(T.lpsn[st] >> (2 * q8)) & 0xffff;
  const uint32_t r = range - lps;
  const uint64_t scaled = (uint64_t)r << extra;
  // most probable symbol: at most one bit of renormalisation
  const uint32_t m1 = (r >> 8) ^ 1;
  const uint32_t rm = r << m1;
  const int em = extra - (int)m1;
  // least probable symbol: range and shift from the table
  const uint32_t rl = en & 0x1ff;
  const int el = extra - (int)(en >> 9);
  const uint64_t vl = value - scaled;
  const bool is_lps = value >= scaled;
  value = is_lps ? vl : value;
  range = is_lps ? rl : rm;
  extra = is_lps ? el : em;
  const uint32_t l = is_lps;
  e.state[ctx] = T.next[st][l];
  return (int)((st & 1) ^ l);
}
// V2: as V1 with the comparison as a sign mask (no flags)
template <> inline __attribute__((always_inline)) int bin<2>(Eng& e, uint64_t& value, uint32_t& range, int& extra, int ctx) {
  const uint32_t st = e.state[ctx];
  const uint32_t q8 = (range >> 3) & 24;
  const uint32_t lps = (T.lps4[st] >> q8) & 0xff;
  const uint32_t en = (uint32_t)(T.lpsn[st] >> (2 * q8)) & 0xffff;
  const uint32_t r = range - lps;
  const uint64_t scaled = (uint64_t)r << extra;
  const uint32_t m1 = (r >> 8) ^ 1;
  const uint32_t rm = r << m1;
  const uint32_t rl = en & 0x1ff;
  const uint32_t sl = en >> 9;
  const uint64_t vl = value - scaled;
  const uint64_t lmask = ~(uint64_t)((int64_t)vl >> 63);
  value -= scaled & lmask;
  range = rm ^ ((rm ^ rl) & (uint32_t)lmask);
  extra -= (int)(m1 ^ ((m1 ^ sl) & (uint32_t)lmask));
  const uint32_t l = (uint32_t)lmask & 1;
  e.state[ctx] = T.next[st][l];
  return (int)((st & 1) ^ l);
}

// V3: V0 with the two candidate next states fetched with the state (one packed word), selected by the outcome
template <> inline __attribute__((always_inline)) int bin<3>(Eng& e, uint64_t& value, uint32_t& range, int& extra, int ctx) {
  const uint32_t st = e.state[ctx];
  const uint64_t pk = T.pack[st];
  const uint32_t lps = ((uint32_t)pk >> ((range >> 3) & 24)) & 0xff;
  uint32_t r = range - lps;
  const uint64_t scaled = (uint64_t)r << extra;
  const uint64_t lmask = ~(uint64_t)((int64_t)(value - scaled) >> 63);
  value -= scaled & lmask;
  r += (lps - r) & (uint32_t)lmask;
  const uint32_t l = (uint32_t)lmask & 1;
  e.state[ctx] = (uint16_t)((pk >> (32 + (lmask & 8))) & 0xff);
  const int shift = __builtin_clz(r) - 23;
  range = r << shift;
  extra -= shift;
  return (int)((st & 1) ^ l);
}
// V4: V2 + V3
template <> inline __attribute__((always_inline)) int bin<4>(Eng& e, uint64_t& value, uint32_t& range, int& extra, int ctx) {
  const uint32_t st = e.state[ctx];
  const uint64_t pk = T.pack[st];
  const uint32_t q8 = (range >> 3) & 24;
  const uint32_t lps = ((uint32_t)pk >> q8) & 0xff;
  const uint32_t en = (uint32_t)(T.lpsn[st] >> (2 * q8)) & 0xffff;
  const uint32_t r = range - lps;
  const uint64_t scaled = (uint64_t)r << extra;
  const uint32_t m1 = (r >> 8) ^ 1;
  const uint32_t rm = r << m1;
  const uint32_t rl = en & 0x1ff;
  const uint32_t sl = en >> 9;
  const uint64_t vl = value - scaled;
  const uint64_t lmask = ~(uint64_t)((int64_t)vl >> 63);
  value -= scaled & lmask;
  range = rm ^ ((rm ^ rl) & (uint32_t)lmask);
  extra -= (int)(m1 ^ ((m1 ^ sl) & (uint32_t)lmask));
  const uint32_t l = (uint32_t)lmask & 1;
  e.state[ctx] = (uint16_t)((pk >> (32 + (lmask & 8))) & 0xff);
  return (int)((st & 1) ^ l);
}

template <int V> static uint64_t run(Eng e, const uint16_t* tr, size_t n, uint64_t* end_state) {
  uint64_t value = e.value, sum = 0;
  uint32_t range = e.range;
  int extra = e.extra;
  auto refill = [&]() __attribute__((always_inline)) {
    e.value = value; e.extra = extra; e.refill(); value = e.value; extra = e.extra;
  };
  extra = -9;
  refill();
  for (size_t i = 0; i < n; ++i) {
    const int t = tr[i];
    int b;
    if (__builtin_expect(t >= 0xfffe, 0)) {
      if (t == 0xffff) {
        extra -= 1;
        if (__builtin_expect(extra < 8, 0)) refill();
        const uint64_t scaled = (uint64_t)range << extra;
        const uint64_t mask = ~(uint64_t)((int64_t)(value - scaled) >> 63);
        value -= scaled & mask;
        b = (int)(mask & 1);
      } else {
        range -= 2;
        const uint64_t scaled = (uint64_t)range << extra;
        if (value >= scaled) { b = 1; }
        else {
          b = 0;
          if (range < 256) { range <<= 1; extra -= 1; if (extra < 8) refill(); }
        }
      }
    } else {
      b = bin<V>(e, value, range, extra, t);
      if (__builtin_expect(extra < 8, 0)) refill();
    }
    sum = sum * 1099511628211ull + (uint64_t)b;
  }
  *end_state = value ^ ((uint64_t)range << 40) ^ (uint64_t)(uint32_t)extra;
  return sum;
}

// Two independent streams decoded in ONE loop, a bin of each in turn: what instruction-level parallelism across two
// arithmetic decoders is worth (each stream's bins form one dependent chain; two chains can overlap in the core).
static uint64_t run_dual(Eng ea, Eng eb, const uint16_t* tr, size_t n, uint64_t* end_state) {
  uint64_t va = ea.value, vb = eb.value, sum = 0;
  uint32_t ra = ea.range, rb = eb.range;
  int xa = -9, xb = -9;
  auto refill_a = [&]() __attribute__((always_inline)) { ea.value = va; ea.extra = xa; ea.refill(); va = ea.value; xa = ea.extra; };
  auto refill_b = [&]() __attribute__((always_inline)) { eb.value = vb; eb.extra = xb; eb.refill(); vb = eb.value; xb = eb.extra; };
  refill_a();
  refill_b();
  auto step = [&](Eng& e, uint64_t& value, uint32_t& range, int& extra, int t, auto& refill) __attribute__((always_inline)) -> int {
    int b;
    if (__builtin_expect(t >= 0xfffe, 0)) {
      if (t == 0xffff) {
        extra -= 1;
        if (__builtin_expect(extra < 8, 0)) refill();
        const uint64_t scaled = (uint64_t)range << extra;
        const uint64_t mask = ~(uint64_t)((int64_t)(value - scaled) >> 63);
        value -= scaled & mask;
        b = (int)(mask & 1);
      } else {
        range -= 2;
        const uint64_t scaled = (uint64_t)range << extra;
        if (value >= scaled) { b = 1; }
        else {
          b = 0;
          if (range < 256) { range <<= 1; extra -= 1; if (extra < 8) refill(); }
        }
      }
    } else {
      b = bin<0>(e, value, range, extra, t);
      if (__builtin_expect(extra < 8, 0)) refill();
    }
    return b;
  };
  for (size_t i = 0; i < n; ++i) {
    const int t = tr[i];
    const int b0 = step(ea, va, ra, xa, t, refill_a);
    const int b1 = step(eb, vb, rb, xb, t, refill_b);
    sum = sum * 1099511628211ull + (uint64_t)(b0 + 2 * b1);
  }
  *end_state = va ^ vb;
  return sum;
}

int main(int argc, char** argv) {
  auto init = rd(argv[1]);
  auto ctx = rd(argv[2]);
  const int reps = argc > 3 ? atoi(argv[3]) : 20;
  uint32_t ns;
  memcpy(&ns, init.data(), 4);
  Eng e;
  memset(e.state, 0, sizeof(e.state));
  memcpy(e.state, init.data() + 4, 2 * ns);
  uint32_t nb;
  memcpy(&nb, init.data() + 4 + 2 * ns, 4);
  e.p = init.data() + 8 + 2 * ns;
  e.end = e.p + nb;
  const uint16_t* tr = (const uint16_t*)ctx.data();
  const size_t n = ctx.size() / 2;
  uint64_t (*fns[])(Eng, const uint16_t*, size_t, uint64_t*) = {run<0>, run<1>, run<2>, run<3>, run<4>};
  uint64_t ref_sum = 0, ref_end = 0;
  for (int v = 0; v < 5; ++v) {
    double best = 1e9;
    uint64_t sum = 0, end_state = 0;
    for (int r = 0; r < reps; ++r) {
      auto t0 = std::chrono::steady_clock::now();
      sum = fns[v](e, tr, n, &end_state);
      best = std::min(best, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
    }
    if (v == 0) { ref_sum = sum; ref_end = end_state; }
    printf("V%d: %.3f ns per bin (%zu bins)%s\n", v, best * 1e9 / n, n, (sum == ref_sum && end_state == ref_end) ? "" : "  MISMATCH");
  }
  {
    double best = 1e9;
    uint64_t es = 0;
    for (int r = 0; r < reps; ++r) {
      auto t0 = std::chrono::steady_clock::now();
      run_dual(e, e, tr, n, &es);
      best = std::min(best, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
    }
    printf("two streams interleaved (the same slice twice, independent engines): %.3f ns per PAIR of bins -> %.3f ns per bin\n", best * 1e9 / n,
           best * 1e9 / n / 2);
  }
}
