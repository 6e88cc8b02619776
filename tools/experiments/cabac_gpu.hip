// Development aid: how fast can ONE wave run an H.264 CABAC decoding engine?
// (entropy decoding on the GPU needs hundreds of streams
// in flight to beat the host; this measures the per-stream rate instead of
// guessing it.)  Each wave decodes `nbins` bins from a pseudo-random byte stream
// (every byte stream is a valid arithmetic-coded stream) with 256 adaptive
// contexts chosen by a function of the previous bins, which reproduces the
// dependence structure of real slice data: bin -> next context -> state ->
// range -> offset.  Everything is wave-uniform, so the arithmetic runs on the
// scalar unit; the context states and the two tables live in VGPRs and are
// reached with v_readlane / v_writelane (dynamic lane, static register), which
// avoids a ~100-cycle LDS round trip per bin.
// build: hipcc --offload-arch=gfx950 -O3 cabac_gpu.hip -o cabac_gpu_test
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

static const uint8_t range_lps[64][4] = {
    {128, 176, 208, 240}, {128, 167, 197, 227}, {128, 158, 187, 216}, {123, 150, 178, 205}, {116, 142, 169, 195}, {111, 135, 160, 185}, {105, 128, 152, 175}, {100, 122, 144, 166},
    {95, 116, 137, 158},  {90, 110, 130, 150},  {85, 104, 123, 142},  {81, 99, 117, 135},   {77, 94, 111, 128},   {73, 89, 105, 122},   {69, 85, 100, 116},   {66, 80, 95, 110},
    {62, 76, 90, 104},    {59, 72, 86, 99},     {56, 69, 81, 94},     {53, 65, 77, 89},     {51, 62, 73, 85},     {48, 59, 69, 80},     {46, 56, 66, 76},     {43, 53, 63, 72},
    {41, 50, 59, 69},     {39, 48, 56, 65},     {37, 45, 54, 62},     {35, 43, 51, 59},     {33, 41, 48, 56},     {32, 39, 46, 53},     {30, 37, 43, 50},     {29, 35, 41, 48},
    {27, 33, 39, 45},     {26, 31, 37, 43},     {24, 30, 35, 41},     {23, 28, 33, 39},     {22, 27, 32, 37},     {21, 26, 30, 35},     {20, 24, 29, 33},     {19, 23, 27, 31},
    {18, 22, 26, 30},     {17, 21, 25, 28},     {16, 20, 23, 27},     {15, 19, 22, 25},     {14, 18, 21, 24},     {14, 17, 20, 23},     {13, 16, 19, 22},     {12, 15, 18, 21},
    {12, 14, 17, 20},     {11, 14, 16, 19},     {11, 13, 15, 18},     {10, 12, 15, 17},     {10, 12, 14, 16},     {9, 11, 13, 15},      {9, 11, 12, 14},      {8, 10, 12, 14},
    {8, 9, 11, 13},       {7, 9, 11, 12},       {7, 9, 10, 12},       {7, 8, 10, 11},       {6, 8, 9, 11},        {6, 7, 9, 10},        {6, 7, 8, 9},         {2, 2, 2, 2}};
static const uint8_t trans_lps[64] = {0,  0,  1,  2,  2,  4,  4,  5,  6,  7,  8,  9,  9,  11, 11, 12, 13, 13, 15, 15, 16, 16,
                                      18, 18, 19, 19, 21, 21, 22, 22, 23, 24, 24, 25, 26, 26, 27, 27, 28, 29, 29, 30, 30, 30,
                                      31, 32, 32, 33, 33, 33, 34, 34, 35, 35, 35, 36, 36, 36, 37, 37, 37, 38, 38, 63};

__device__ __forceinline__ uint32_t writelane(uint32_t val, uint32_t lane, uint32_t old) {
  // lane `lane` of `old` := val (both wave-uniform, i.e. SGPRs)
  // (one SGPR operand per VALU instruction on gfx9: the lane select goes through M0)
  asm volatile("s_mov_b32 m0, %2\n\tv_writelane_b32 %0, %1, m0" : "+v"(old) : "s"(val), "s"(lane) : "m0");
  return old;
}

struct Tables { uint32_t lps4[128]; uint32_t next[128]; };  // next: [7:0] after MPS, [15:8] after LPS

__global__ __launch_bounds__(256) void k_cabac(const uint32_t* __restrict__ stream, uint32_t stream_words, const Tables* __restrict__ tab,
                                               int nbins, uint32_t* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  // tables and context states in registers: entry e of a 128-entry table is lane e & 63 of register e >> 6
  uint32_t l0 = tab->lps4[lane], l1 = tab->lps4[64 + lane];
  uint32_t n0 = tab->next[lane], n1 = tab->next[64 + lane];
  uint32_t s0 = (lane * 2) & 127, s1 = (lane * 3 + 1) & 127, s2 = (lane * 5 + 2) & 127, s3 = (lane * 7 + 3) & 127;  // 256 states
  uint32_t pos = __builtin_amdgcn_readfirstlane((wave * 977u) % (stream_words - 8));
  uint64_t value = ((uint64_t)__builtin_bswap32(stream[pos]) << 32) | __builtin_bswap32(stream[pos + 1]);
  pos += 2;
  int extra = 64 - 9;  // bits of the stream in `value` beyond the nine of the offset
  uint32_t range = 510;
  uint32_t hist = 1, acc = 0;
  for (int i = 0; i < nbins; ++i) {
    const uint32_t ctx = (hist * 0x9e3779b1u >> 24) & 255;  // depends on the previous bins, like ctxIdxInc does
    const uint32_t sl = ctx & 63;
    // branch-free selects: a scalar branch costs more than the three extra readlanes
    const uint32_t hi = ctx >> 6;
    const uint32_t c0 = __builtin_amdgcn_readlane(s0, sl), c1 = __builtin_amdgcn_readlane(s1, sl),
                   c2 = __builtin_amdgcn_readlane(s2, sl), c3 = __builtin_amdgcn_readlane(s3, sl);
    const uint32_t st = hi == 0 ? c0 : hi == 1 ? c1 : hi == 2 ? c2 : c3;
    const uint32_t tl = st & 63;
    const uint32_t la = __builtin_amdgcn_readlane(l0, tl), lb = __builtin_amdgcn_readlane(l1, tl);
    const uint32_t na = __builtin_amdgcn_readlane(n0, tl), nb = __builtin_amdgcn_readlane(n1, tl);
    const uint32_t lw = st < 64 ? la : lb, nw = st < 64 ? na : nb;
    const uint32_t lps = (lw >> ((range >> 3) & 24)) & 0xff;
    uint32_t r = range - lps;
    const uint64_t scaled = (uint64_t)r << extra;
    const bool is_lps = value >= scaled;
    value -= is_lps ? scaled : 0;
    r = is_lps ? lps : r;
    const uint32_t ns = is_lps ? (nw >> 8) & 0xff : nw & 0xff;
    s0 = writelane(hi == 0 ? ns : c0, sl, s0);
    s1 = writelane(hi == 1 ? ns : c1, sl, s1);
    s2 = writelane(hi == 2 ? ns : c2, sl, s2);
    s3 = writelane(hi == 3 ? ns : c3, sl, s3);
    const int shift = __builtin_clz(r) - 23;
    range = r << shift;
    extra -= shift;
    if (extra < 24) {  // refill 32 bits
      value = (value << 32) | __builtin_bswap32(stream[pos]);
      pos = pos + 1 < stream_words ? pos + 1 : 0;
      extra += 32;
    }
    const uint32_t bin = (st & 1) ^ (is_lps ? 1u : 0u);
    hist = (hist << 1) | bin;
    acc += bin;
  }
  if (lane == 0) out[wave] = acc;
}

__global__ __launch_bounds__(256) void k_cabac1024(const uint32_t* __restrict__ stream, uint32_t stream_words, const Tables* __restrict__ tab,
                                               int nbins, uint32_t* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  // tables and context states in registers: entry e of a 128-entry table is lane e & 63 of register e >> 6
  uint32_t l0 = tab->lps4[lane], l1 = tab->lps4[64 + lane];
  uint32_t n0 = tab->next[lane], n1 = tab->next[64 + lane];
  uint32_t S[16];  // 1024 states: context c is lane c & 63 of S[c >> 6]; the index is wave-uniform
#pragma unroll
  for (int q = 0; q < 16; ++q) S[q] = (lane * (2 * q + 3) + q) & 127;
  uint32_t pos = __builtin_amdgcn_readfirstlane((wave * 977u) % (stream_words - 8));
  uint64_t value = ((uint64_t)__builtin_bswap32(stream[pos]) << 32) | __builtin_bswap32(stream[pos + 1]);
  pos += 2;
  int extra = 64 - 9;  // bits of the stream in `value` beyond the nine of the offset
  uint32_t range = 510;
  uint32_t hist = 1, acc = 0;
  for (int i = 0; i < nbins; ++i) {
    const uint32_t ctx = (hist * 0x9e3779b1u >> 22) & 1023;  // depends on the previous bins, like ctxIdxInc does
    const uint32_t sl = ctx & 63;
    const uint32_t hi = ctx >> 6;
    uint32_t sreg = S[hi];   // dynamic register index (v_movrels / gpr index mode)
    const uint32_t st = __builtin_amdgcn_readlane(sreg, sl);
    const uint32_t tl = st & 63;
    const uint32_t la = __builtin_amdgcn_readlane(l0, tl), lb = __builtin_amdgcn_readlane(l1, tl);
    const uint32_t na = __builtin_amdgcn_readlane(n0, tl), nb = __builtin_amdgcn_readlane(n1, tl);
    const uint32_t lw = st < 64 ? la : lb, nw = st < 64 ? na : nb;
    const uint32_t lps = (lw >> ((range >> 3) & 24)) & 0xff;
    uint32_t r = range - lps;
    const uint64_t scaled = (uint64_t)r << extra;
    const bool is_lps = value >= scaled;
    value -= is_lps ? scaled : 0;
    r = is_lps ? lps : r;
    const uint32_t ns = is_lps ? (nw >> 8) & 0xff : nw & 0xff;
    S[hi] = writelane(ns, sl, sreg);
    const int shift = __builtin_clz(r) - 23;
    range = r << shift;
    extra -= shift;
    if (extra < 24) {  // refill 32 bits
      value = (value << 32) | __builtin_bswap32(stream[pos]);
      pos = pos + 1 < stream_words ? pos + 1 : 0;
      extra += 32;
    }
    const uint32_t bin = (st & 1) ^ (is_lps ? 1u : 0u);
    hist = (hist << 1) | bin;
    acc += bin;
  }
  if (lane == 0) out[wave] = acc;
}

int main(int argc, char** argv) {
  const int nbins = argc > 1 ? atoi(argv[1]) : 200000;
  Tables t;
  for (int st = 0; st < 128; ++st) {
    const int ps = st >> 1, mps = st & 1;
    t.lps4[st] = 0;
    for (int q = 0; q < 4; ++q) t.lps4[st] |= (uint32_t)range_lps[ps][q] << (8 * q);
    const uint32_t after_mps = ((ps < 62 ? ps + 1 : ps) << 1) | mps;
    const uint32_t after_lps = (trans_lps[ps] << 1) | (ps == 0 ? !mps : mps);
    t.next[st] = after_mps | (after_lps << 8);
  }
  const uint32_t words = 1 << 20;
  std::vector<uint32_t> h(words);
  uint64_t x = 88172645463325252ull;
  for (auto& w : h) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; w = (uint32_t)(x >> 16); }
  uint32_t *d_stream, *d_out; Tables* d_tab;
  (void)hipMalloc(&d_stream, words * 4); (void)hipMalloc(&d_out, 65536 * 4); (void)hipMalloc(&d_tab, sizeof(Tables));
  (void)hipMemcpy(d_stream, h.data(), words * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(d_tab, &t, sizeof(t), hipMemcpyHostToDevice);
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  for (int variant = 0; variant < 2; ++variant)
  for (int waves : {1, 1024, 4096}) {
    auto kern = variant ? k_cabac1024 : k_cabac;
    const int blocks = (waves + 3) / 4, threads = waves < 4 ? waves * 64 : 256;
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, d_stream, words, d_tab, 1000, d_out);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(a);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, d_stream, words, d_tab, nbins, d_out);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    std::vector<uint32_t> o(waves);
    (void)hipMemcpy(o.data(), d_out, waves * 4, hipMemcpyDeviceToHost);
    printf("%s %5d waves: %8.2f ms for %d bins each -> %7.1f ns/bin per stream, %8.1f Mbin/s aggregate (ones: %.3f)\n", variant ? "1024 ctx" : " 256 ctx", waves, ms, nbins,
           ms * 1e6 / nbins, waves * (double)nbins / ms / 1e3, o[0] / (double)nbins);
  }
  return 0;
}
