// GPU check of hwang_amd/csrc/amd/inter_filters.h (test infrastructure): the same header compiled twice - once bound
// to the gfx950 instructions and run on the device, once bound to the plain C restatements of
// tests/fixtures/inter_filters_host.cpp's kind and run on the host - must agree bit for bit on random inputs.  That
// pins the restatements of v_alignbyte_b32 / v_perm_b32 / v_dot4c_i32_i8 / v_dot4_u32_u8 to the hardware, so the
// CPU test of the arithmetic (tests/test_inter_filters.py) speaks for what the kernel computes.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

namespace dev {
__device__ __forceinline__ int clip8(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }
constexpr int kWinLumaDw = 6, kWinChromaDw = 3;
#define HW_FN __device__ __forceinline__
#define HW_ALIGNBYTE(hi, lo, n) __builtin_amdgcn_alignbyte(hi, lo, n)
#define HW_PERM(hi, lo, sel) __builtin_amdgcn_perm(hi, lo, sel)
#define HW_SDOT4(a, b, acc) __builtin_amdgcn_sdot4(a, b, acc, false)
#define HW_UDOT4(a, b, acc) __builtin_amdgcn_udot4(a, b, acc, false)
#include "../../hwang_amd/csrc/amd/inter_filters.h"
#undef HW_FN
#undef HW_ALIGNBYTE
#undef HW_PERM
#undef HW_SDOT4
#undef HW_UDOT4
}  // namespace dev

namespace host {
static inline int clip8(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }
constexpr int kWinLumaDw = 6, kWinChromaDw = 3;
static inline uint32_t emu_alignbyte(uint32_t hi, uint32_t lo, int n) { return (uint32_t)((((uint64_t)hi << 32) | lo) >> (8 * (n & 3))); }
static inline uint32_t emu_perm(uint32_t hi, uint32_t lo, uint32_t sel) {
  const uint64_t v = ((uint64_t)hi << 32) | lo;
  uint32_t out = 0;
  for (int k = 0; k < 4; ++k) {
    const unsigned s = (sel >> (8 * k)) & 0xff;
    const unsigned b = s >= 13 ? 0xff : s == 12 ? 0 : s >= 8 ? (((v >> (8 * (2 * (s - 8) + 1) + 7)) & 1) ? 0xff : 0) : (unsigned)((v >> (8 * s)) & 0xff);
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
}  // namespace host

constexpr int kIn = 40;   // dwords of input per case: 36 luma window, 4 spare used for chroma / residual
constexpr int kOut = 16 + 1 + 3;  // luma for 16 fractions, one chroma, add_res4 / add_res2 / avg4

__global__ void run(const uint32_t* in, uint32_t* out, int n) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  const uint32_t* w = in + (size_t)t * kIn;
  uint32_t* o = out + (size_t)t * kOut;
  for (int f = 0; f < 16; ++f) o[f] = dev::luma_win4(w, f & 3, f >> 2);
  const int fx = w[36] & 7, fy = (w[36] >> 3) & 7, odd = (w[36] >> 6) & 1;
  const uint32_t wts = (uint32_t)((8 - fx) * (8 - fy)) | ((uint32_t)(fx * (8 - fy)) << 8) | ((uint32_t)((8 - fx) * fy) << 16) | ((uint32_t)(fx * fy) << 24);
  o[16] = dev::chroma_win2(w, odd, wts);
  o[17] = dev::add_res4(w[37], w[38], w[39]);
  o[18] = dev::add_res2(w[37], w[38]);
  o[19] = dev::avg4(w[37], w[0]);
}

int main() {
  const int n = 1 << 16;
  std::vector<uint32_t> in((size_t)n * kIn), out((size_t)n * kOut);
  uint32_t s = 99;
  for (auto& v : in) {
    s = s * 1664525u + 1013904223u;
    uint32_t x = s;
    s = s * 1664525u + 1013904223u;
    v = (x >> 8) ^ (s << 8);
  }
  for (int t = 0; t < n; ++t) {  // residuals in a plausible range, some cases saturated
    if (t % 5 == 0) for (int k = 0; k < 36; ++k) in[(size_t)t * kIn + k] = (in[(size_t)t * kIn + k] & 0x01010101u) * 255u;
    int16_t r[4];
    for (int k = 0; k < 4; ++k) r[k] = (int16_t)((int)((in[(size_t)t * kIn + 38 + (k >> 1)] >> (16 * (k & 1))) % 1400) - 700);
    memcpy(&in[(size_t)t * kIn + 38], r, 8);
  }
  uint32_t *din, *dout;
  if (hipMalloc(&din, in.size() * 4) != hipSuccess || hipMalloc(&dout, out.size() * 4) != hipSuccess) return 2;
  if (hipMemcpy(din, in.data(), in.size() * 4, hipMemcpyHostToDevice) != hipSuccess) return 2;
  hipLaunchKernelGGL(run, dim3(n / 256), dim3(256), 0, 0, din, dout, n);
  if (hipMemcpy(out.data(), dout, out.size() * 4, hipMemcpyDeviceToHost) != hipSuccess) return 2;
  for (int t = 0; t < n; ++t) {
    const uint32_t* w = &in[(size_t)t * kIn];
    uint32_t want[kOut];
    for (int f = 0; f < 16; ++f) want[f] = host::luma_win4(w, f & 3, f >> 2);
    const int fx = w[36] & 7, fy = (w[36] >> 3) & 7, odd = (w[36] >> 6) & 1;
    const uint32_t wts = (uint32_t)((8 - fx) * (8 - fy)) | ((uint32_t)(fx * (8 - fy)) << 8) | ((uint32_t)((8 - fx) * fy) << 16) | ((uint32_t)(fx * fy) << 24);
    want[16] = host::chroma_win2(w, odd, wts);
    want[17] = host::add_res4(w[37], w[38], w[39]);
    want[18] = host::add_res2(w[37], w[38]);
    want[19] = host::avg4(w[37], w[0]);
    for (int k = 0; k < kOut; ++k)
      if (want[k] != out[(size_t)t * kOut + k]) {
        printf("mismatch case %d output %d: device %08x host %08x\n", t, k, out[(size_t)t * kOut + k], want[k]);
        return 1;
      }
  }
  printf("ok %d\n", n * kOut);
  return 0;
}
