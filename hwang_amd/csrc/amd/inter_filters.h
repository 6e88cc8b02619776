// Packed arithmetic of k_inter's window path (k_recon.hip): sub-sample interpolation (8.4.2.2.1, 8.4.2.2.2) and the
// residual add for four luma / two chroma samples of one lane, on data as it lies in the LDS window.
//
// Included by k_recon.hip inside its namespace with the four hardware operations bound to the gfx950
// instructions, and by tests/fixtures/inter_filters_host.cpp with them bound to plain C restatements, which is
// how the arithmetic is checked against a scalar restatement of the standard's formulas on a machine without a GPU
// (tests/test_inter_filters.py).  The includer defines:
//   HW_FN                    function qualifiers
//   HW_ALIGNBYTE(hi, lo, n)  bytes n .. n+3 of the eight bytes lo, hi          (v_alignbyte_b32)
//   HW_PERM(hi, lo, sel)     byte k = byte sel[k] of lo (0-3) / hi (4-7), 0x0c = 0   (v_perm_b32)
//   HW_SDOT4(a, b, acc)      acc + sum of the four signed byte products           (v_dot4c_i32_i8)
//   HW_UDOT4(a, b, acc)      the same on unsigned bytes                           (v_dot4_u32_u8)
//   clip8(int), kWinLumaDw, kWinChromaDw
// ---- packed arithmetic of the window path ---------------------------------
// Four samples travel as the four bytes of a dword, intermediate sums as pairs
// of 16-bit lanes (v_pk_*_i16: samples 0,2 in one register, 1,3 in another), the
// horizontal six-tap sums come from the 8-bit dot product (v_dot4c_i32_i8).
typedef short pk16 __attribute__((ext_vector_type(2)));
HW_FN pk16 as_pk(uint32_t v) { return __builtin_bit_cast(pk16, v); }
HW_FN uint32_t as_u32(pk16 v) { return __builtin_bit_cast(uint32_t, v); }
HW_FN pk16 clamp255(pk16 v) {
  return __builtin_elementwise_min(__builtin_elementwise_max(v, (pk16)(0)), (pk16)(255));
}
// (a + b + 1) >> 1 on each of the four bytes
HW_FN uint32_t avg4(uint32_t a, uint32_t b) { return (a | b) - (((a ^ b) >> 1) & 0x7f7f7f7fu); }
HW_FN uint32_t pack4(int v0, int v1, int v2, int v3) {
  return (uint32_t)v0 | ((uint32_t)v1 << 8) | ((uint32_t)v2 << 16) | ((uint32_t)v3 << 24);
}
// clip8 of four sums that fit 16 bits, packed.  Done on 16-bit pairs on purpose: written as clip8(x) | clip8(y) << 8 on
// 32-bit values, ROCm 7.2's compiler selects gfx950's v_ashr_pk_u8_i32 and assumes bits 31:16 of its result are zero -
// the hardware sets them when the first operand is negative (tests/fixtures/inter_filters_gpu.hip caught it;
// tests/test_inter_filters.py keeps the instruction out of every kernel of this repo).
HW_FN uint32_t clip_pack4(int v0, int v1, int v2, int v3) {
  const pk16 a = clamp255(as_pk(((uint32_t)v0 & 0xffffu) | ((uint32_t)v2 << 16)));
  const pk16 b = clamp255(as_pk(((uint32_t)v1 & 0xffffu) | ((uint32_t)v3 << 16)));
  return as_u32(a) | (as_u32(b) << 8);
}
// bytes n .. n+3 of the eight bytes lo, hi
HW_FN uint32_t bytes_at(uint32_t lo, uint32_t hi, int n) { return HW_ALIGNBYTE(hi, lo, n); }

// Six-tap sums (1, -5, 20, 20, -5, 1) of one row for four adjacent samples: a, b, c
// are bytes 0..11 of the row, sample i takes bytes i .. i+5.  The bytes go in as
// signed (x - 128); the taps add up to 32, so 4096 in the accumulator restores
// the exact unsigned sum.  `bias` is added on top (the rounding term).
constexpr uint32_t kTapLo = 0x1414fb01u;  // 1, -5, 20, 20
constexpr uint32_t kTapHi = 0x000001fbu;  // -5, 1, 0, 0
HW_FN void htap4(uint32_t a, uint32_t b, uint32_t c, int bias, int* t) {
  a ^= 0x80808080u;
  b ^= 0x80808080u;
  c ^= 0x80808080u;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const uint32_t lo = i ? bytes_at(a, b, i) : a, hi = i ? bytes_at(b, c, i) : b;
    t[i] = HW_SDOT4((int)hi, (int)kTapHi, HW_SDOT4((int)lo, (int)kTapLo, 4096 + bias));
  }
}
// horizontal half samples b (8-241), clipped, packed
HW_FN uint32_t hhalf4(uint32_t a, uint32_t b, uint32_t c) {
  int t[4];
  htap4(a, b, c, 16, t);
  return clip_pack4(t[0] >> 5, t[1] >> 5, t[2] >> 5, t[3] >> 5);
}
// Vertical six-tap sums of four columns, accumulated row by row on 16-bit pairs (samples 0,2 in `e`, 1,3 in `o`), so
// that only the running sums stay live.  g = the four samples of one row, wt = that row's tap (1, -5, 20, 20, -5, 1).
struct VAcc {
  pk16 e, o;
};
HW_FN void vacc_add(VAcc& a, uint32_t g, int wt) {
  a.e += as_pk(g & 0x00ff00ffu) * (short)wt;
  a.o += as_pk((g >> 8) & 0x00ff00ffu) * (short)wt;
}
// vertical half samples h (8-242): (sum + 16) >> 5, clipped, packed; the accumulator starts at 16
HW_FN uint32_t vacc_half(const VAcc& a) { return as_u32(clamp255(a.e >> 5)) | (as_u32(clamp255(a.o >> 5)) << 8); }
HW_FN int tap_of_row(int r) { return (r == 0 || r == 5) ? 1 : ((r == 1 || r == 4) ? -5 : 20); }

// 8.4.2.2.1 for the lane's four luma samples from the window: row r of the lane
// (sample row y-2+r) is base[r * kWinLumaDw + 0..2], byte 0 = column x-2.  fx, fy
// are wave uniform: the branches are scalar.  Rows are consumed one at a time.
HW_FN uint32_t luma_win4(const uint32_t* base, int fx, int fy) {
  if (fy == 0) {
    const uint32_t a = base[2 * kWinLumaDw], b = base[2 * kWinLumaDw + 1], c = base[2 * kWinLumaDw + 2];
    const uint32_t g = bytes_at(a, b, 2);
    if (fx == 0) return g;
    const uint32_t h = hhalf4(a, b, c);
    return fx == 2 ? h : avg4(h, fx == 1 ? g : bytes_at(a, b, 3));
  }
  const int col = fx == 3 ? 3 : 2;  // column of the vertical half sample
  if (fx != 2) {                    // d, h, n (fx 0) / e, g, p, r (fx odd, fy odd) / i, k (fx odd, fy 2): a vertical half sample is needed
    VAcc acc = {(pk16)(16), (pk16)(16)};
    uint32_t gsel = 0;
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      const uint32_t g = bytes_at(base[r * kWinLumaDw], base[r * kWinLumaDw + 1], col);
      vacc_add(acc, g, tap_of_row(r));
      if (r == 2 || r == 3) gsel = (r == 2) == (fy != 3) ? g : gsel;
    }
    const uint32_t v = vacc_half(acc);
    if (fx == 0) return fy == 2 ? v : avg4(v, gsel);
    if (fy != 2) {                  // e, g, p, r: nearest horizontal and vertical half samples
      const uint32_t* row = base + (fy == 3 ? 3 : 2) * kWinLumaDw;
      return avg4(hhalf4(row[0], row[1], row[2]), v);
    }
    // i, k: centre sample j averaged with h: fall through to j below, keeping v
    int j1[4] = {512, 512, 512, 512};
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      int t[4];
      htap4(base[r * kWinLumaDw], base[r * kWinLumaDw + 1], base[r * kWinLumaDw + 2], 0, t);
#pragma unroll
      for (int i = 0; i < 4; ++i) j1[i] += tap_of_row(r) * t[i];
    }
    return avg4(clip_pack4(j1[0] >> 10, j1[1] >> 10, j1[2] >> 10, j1[3] >> 10), v);
  }
  // fx == 2: the centre sample j from the six rows' horizontal sums; alone (fy 2) or averaged with b (f, q)
  int j1[4] = {512, 512, 512, 512}, bb[4] = {0, 0, 0, 0};
#pragma unroll
  for (int r = 0; r < 6; ++r) {
    int t[4];
    htap4(base[r * kWinLumaDw], base[r * kWinLumaDw + 1], base[r * kWinLumaDw + 2], 0, t);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      j1[i] += tap_of_row(r) * t[i];
      if (r == 2 || r == 3) bb[i] = (r == 2) == (fy != 3) ? t[i] : bb[i];
    }
  }
  const uint32_t jp = clip_pack4(j1[0] >> 10, j1[1] >> 10, j1[2] >> 10, j1[3] >> 10);
  if (fy == 2) return jp;
  return avg4(jp, clip_pack4((bb[0] + 16) >> 5, (bb[1] + 16) >> 5, (bb[2] + 16) >> 5, (bb[3] + 16) >> 5));
}

// 8.4.2.2.2 for the lane's two chroma samples: rowp points at the dword of window row cy that holds column
// 2 * cbcol (the next row is kWinChromaDw further on); wts = the four bilinear weights, one byte each.
HW_FN uint32_t chroma_win2(const uint32_t* rowp, bool odd, uint32_t wts) {
  const uint32_t a0 = rowp[0], a1 = rowp[1], c0 = rowp[kWinChromaDw], c1 = rowp[kWinChromaDw + 1];
  const uint32_t a = odd ? bytes_at(a0, a1, 2) : a0, c = odd ? bytes_at(c0, c1, 2) : c0;
  const uint32_t v0 = HW_PERM(c, a, 0x05040100u);  // A, B, C, D of sample 0
  const uint32_t v1 = HW_PERM(c, a, 0x06050201u);  // ... of sample 1
  return (HW_UDOT4(v0, wts, 32u) >> 6) | ((HW_UDOT4(v1, wts, 32u) >> 6) << 8);
}

// prediction bytes + 16-bit residuals, clipped
HW_FN uint32_t add_res4(uint32_t pred, uint32_t r01, uint32_t r23) {
  const pk16 s01 = clamp255(as_pk(HW_PERM(0u, pred, 0x0c010c00u)) + as_pk(r01));
  const pk16 s23 = clamp255(as_pk(HW_PERM(0u, pred, 0x0c030c02u)) + as_pk(r23));
  return HW_PERM(as_u32(s23), as_u32(s01), 0x06040200u);
}
HW_FN uint32_t add_res2(uint32_t pred, uint32_t r01) {
  const pk16 s = clamp255(as_pk(HW_PERM(0u, pred, 0x0c010c00u)) + as_pk(r01));
  return HW_PERM(0u, as_u32(s), 0x0c0c0200u);
}

