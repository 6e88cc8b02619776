// Host harness for hwang_amd/csrc/amd/inter_filters.h (test infrastructure, not part of the product): binds the four
// hardware operations to plain C restatements of the gfx950 instructions and checks the packed interpolation against
// a scalar restatement of ITU-T H.264 8.4.2.2.1 / 8.4.2.2.2 on random windows, every fractional position, plus the
// residual add.  Prints "ok <cases>" or the first mismatch.  Built by tests/test_inter_filters.py with clang++.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

static inline int clip8(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }
constexpr int kWinLumaDw = 6, kWinChromaDw = 3;

// v_alignbyte_b32: ({S0, S1} >> (8 * S2[1:0])) [31:0]
static inline uint32_t emu_alignbyte(uint32_t hi, uint32_t lo, int n) {
  const uint64_t v = ((uint64_t)hi << 32) | lo;
  return (uint32_t)(v >> (8 * (n & 3)));
}
// v_perm_b32: byte k of the result = in[sel byte k], in[0..3] = S1, in[4..7] = S0, selector 12 = 0x00, >= 13 = 0xff
static inline uint32_t emu_perm(uint32_t hi, uint32_t lo, uint32_t sel) {
  const uint64_t v = ((uint64_t)hi << 32) | lo;
  uint32_t out = 0;
  for (int k = 0; k < 4; ++k) {
    const unsigned s = (sel >> (8 * k)) & 0xff;
    unsigned b;
    if (s >= 13) b = 0xff;
    else if (s == 12) b = 0;
    else if (s >= 8) b = ((v >> (8 * (2 * (s - 8) + 1) + 7)) & 1) ? 0xff : 0;  // sign of a 16-bit half
    else b = (unsigned)((v >> (8 * s)) & 0xff);
    out |= b << (8 * k);
  }
  return out;
}
static inline int emu_sdot4(int a, int b, int acc) {
  for (int k = 0; k < 4; ++k) acc += (int)(int8_t)((uint32_t)a >> (8 * k)) * (int)(int8_t)((uint32_t)b >> (8 * k));
  return acc;
}
static inline uint32_t emu_udot4(uint32_t a, uint32_t b, uint32_t acc) {
  for (int k = 0; k < 4; ++k) acc += ((a >> (8 * k)) & 0xff) * ((b >> (8 * k)) & 0xff);
  return acc;
}

#define HW_FN static inline
#define HW_ALIGNBYTE(hi, lo, n) emu_alignbyte(hi, lo, n)
#define HW_PERM(hi, lo, sel) emu_perm(hi, lo, sel)
#define HW_SDOT4(a, b, acc) emu_sdot4(a, b, acc)
#define HW_UDOT4(a, b, acc) emu_udot4(a, b, acc)
#include "../../hwang_amd/csrc/amd/inter_filters.h"

// ---- the standard's formulas, one sample at a time -------------------------
static int tap(const int* p) { return p[0] - 5 * p[1] + 20 * p[2] + 20 * p[3] - 5 * p[4] + p[5]; }
// win[r][c]: sample (x - 2 + c, y - 2 + r); returns the prediction at (x + fx/4, y + fy/4)
static int luma_ref(const uint8_t win[6][12], int fx, int fy) {
  auto G = [&](int dx, int dy) { return (int)win[2 + dy][2 + dx]; };
  auto b1 = [&](int dy) { int p[6]; for (int k = 0; k < 6; ++k) p[k] = win[2 + dy][k]; return tap(p); };          // between G and H of row dy
  auto h1 = [&](int dx) { int p[6]; for (int k = 0; k < 6; ++k) p[k] = win[k][2 + dx]; return tap(p); };          // between G and M of column dx
  auto bs = [&](int dy) { return clip8((b1(dy) + 16) >> 5); };
  auto hs = [&](int dx) { return clip8((h1(dx) + 16) >> 5); };
  auto js = [&]() { int p[6]; for (int k = 0; k < 6; ++k) { int q[6]; for (int c = 0; c < 6; ++c) q[c] = win[k][c]; p[k] = tap(q); } return clip8((tap(p) + 512) >> 10); };
  const int b = bs(0), h = hs(0), s = bs(1), m = hs(1);
  switch (fy * 4 + fx) {
    case 0: return G(0, 0);
    case 1: return (G(0, 0) + b + 1) >> 1;                // a
    case 2: return b;
    case 3: return (b + G(1, 0) + 1) >> 1;                // c
    case 4: return (G(0, 0) + h + 1) >> 1;                // d
    case 5: return (b + h + 1) >> 1;                      // e
    case 6: return (b + js() + 1) >> 1;                   // f
    case 7: return (b + m + 1) >> 1;                      // g
    case 8: return h;
    case 9: return (h + js() + 1) >> 1;                   // i
    case 10: return js();
    case 11: return (js() + m + 1) >> 1;                  // k
    case 12: return (G(0, 1) + h + 1) >> 1;               // n
    case 13: return (h + s + 1) >> 1;                     // p
    case 14: return (js() + s + 1) >> 1;                  // q
    default: return (m + s + 1) >> 1;                     // r
  }
}

static uint32_t rnd_state = 12345;
static uint32_t rnd() { rnd_state = rnd_state * 1664525u + 1013904223u; return rnd_state >> 8; }

int main() {
  long cases = 0;
  for (int iter = 0; iter < 4000; ++iter) {
    // a lane's view of the luma window: six rows of kWinLumaDw dwords, the lane uses dwords 0..2 of each
    uint32_t lds[6 * kWinLumaDw];
    uint8_t rows[6][12];
    const int mode = iter % 4;  // random, extremes, low contrast, all 255
    for (int r = 0; r < 6; ++r)
      for (int c = 0; c < 12; ++c)
        rows[r][c] = mode == 0 ? rnd() & 0xff : mode == 1 ? ((rnd() & 1) ? 255 : 0) : mode == 2 ? 120 + (rnd() & 15) : 255;
    memset(lds, 0xee, sizeof(lds));
    for (int r = 0; r < 6; ++r) memcpy(&lds[r * kWinLumaDw], rows[r], 12);
    for (int fy = 0; fy < 4; ++fy)
      for (int fx = 0; fx < 4; ++fx) {
        const uint32_t got = luma_win4(lds, fx, fy);
        for (int i = 0; i < 4; ++i) {
          uint8_t w[6][12];
          for (int r = 0; r < 6; ++r) memcpy(w[r], rows[r] + i, 12 - i);
          const int want = luma_ref(w, fx, fy);
          ++cases;
          if ((int)((got >> (8 * i)) & 0xff) != want) {
            printf("luma mismatch iter %d fx %d fy %d sample %d: got %u want %d\n", iter, fx, fy, i, (got >> (8 * i)) & 0xff, want);
            return 1;
          }
        }
      }
    // chroma: two rows of kWinChromaDw dwords; the lane's samples start at byte 0 (even) or 2 (odd) of its dword
    uint32_t cl[2 * kWinChromaDw];
    uint8_t cr[2][12];
    for (int r = 0; r < 2; ++r)
      for (int c = 0; c < 12; ++c) cr[r][c] = mode == 1 ? ((rnd() & 1) ? 255 : 0) : rnd() & 0xff;
    for (int r = 0; r < 2; ++r) memcpy(&cl[r * kWinChromaDw], cr[r], 12);
    for (int fy = 0; fy < 8; ++fy)
      for (int fx = 0; fx < 8; ++fx)
        for (int cbcol = 0; cbcol < 4; ++cbcol) {
          const uint32_t wts = (uint32_t)((8 - fx) * (8 - fy)) | ((uint32_t)(fx * (8 - fy)) << 8) | ((uint32_t)((8 - fx) * fy) << 16) |
                               ((uint32_t)(fx * fy) << 24);
          const uint32_t got = chroma_win2(cl + (cbcol >> 1), cbcol & 1, wts);
          for (int i = 0; i < 2; ++i) {
            const int c = 2 * cbcol + i;
            const int want = ((8 - fx) * (8 - fy) * cr[0][c] + fx * (8 - fy) * cr[0][c + 1] + (8 - fx) * fy * cr[1][c] + fx * fy * cr[1][c + 1] + 32) >> 6;
            ++cases;
            if ((int)((got >> (8 * i)) & 0xff) != want || (got >> 16)) {
              printf("chroma mismatch iter %d fx %d fy %d cbcol %d sample %d: got %x want %d\n", iter, fx, fy, cbcol, i, got, want);
              return 1;
            }
          }
        }
    // residual add and the byte-wise average
    const uint32_t pred = rnd() | (rnd() << 16), other = rnd() | (rnd() << 16);
    int16_t res[4];
    for (int i = 0; i < 4; ++i) res[i] = (int16_t)((int)(rnd() % 1200) - 600);
    uint32_t r01, r23;
    memcpy(&r01, res, 4);
    memcpy(&r23, res + 2, 4);
    const uint32_t sum4 = add_res4(pred, r01, r23), sum2 = add_res2(pred, r01), av = avg4(pred, other);
    for (int i = 0; i < 4; ++i) {
      const int p = (pred >> (8 * i)) & 0xff, o = (other >> (8 * i)) & 0xff;
      ++cases;
      if ((int)((sum4 >> (8 * i)) & 0xff) != clip8(p + res[i]) || (int)((av >> (8 * i)) & 0xff) != ((p + o + 1) >> 1) ||
          (i < 2 && (int)((sum2 >> (8 * i)) & 0xff) != clip8(p + res[i])) || (sum2 >> 16)) {
        printf("residual / average mismatch iter %d sample %d: pred %08x other %08x res %d sum4 %08x sum2 %08x avg %08x\n", iter, i, pred, other,
               res[i], sum4, sum2, av);
        return 1;
      }
    }
  }
  printf("ok %ld\n", cases);
  return 0;
}
