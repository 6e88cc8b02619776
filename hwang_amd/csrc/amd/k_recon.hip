// The sample-parallel half of H.264 decoding on CDNA4 (gfx950).  Every kernel
// takes an array of JobDesc, one per picture of the batch.
//
//   k_upload    the batch's records pulled from pinned host memory (one launch)
//   k_inter     every inter macroblock at once, one wave each: dequant + 4x4 /
//               8x8 inverse transforms, quarter-sample luma and eighth-sample
//               chroma motion compensation from unaligned dword fetches
//               (default, explicit and implicit weighting), + residual
//   k_mv        motion vectors of P macroblocks from the parsed differences (a row pipeline like k_intra's; the
//               prediction branch-free on packed vectors, a fifth wave per workgroup imports the band above's line)
//   k_residual  dequant + inverse transforms of the intra macroblocks -> int16
//   k_intra     intra prediction in dependency order: a wave per macroblock
//               row, four rows (one workgroup) per band, progress handed down
//               through LDS inside a band and through agent-scope words in
//               global memory between bands
//   k_bs        boundary strengths, alpha / beta / tc0, edge masks per macroblock
//   k_deblock   in-loop deblocking filter, same banded row pipeline; a row's vertical edges on its dwords and a
//               column's horizontal edges on its bytes, in registers
//   k_chain     k_inter + k_intra + k_deblock of a lane's dependent pictures as ONE launch of overlapping stages
//
// What the row pipelines cost: a lone wave issues an instruction every four to five clocks whatever unit executes it,
// so a macroblock's step is the number of instructions on its path ("What a step costs"): their
// hot paths are written without branches on values, and nothing that is a load from global memory sits in a
// macroblock loop unless it is a prefetch.
//
// These replace what the reference gets from libavcodec's hl_decode_mb /
// loop_filter behind avcodec_send_packet
// (impls/software/software_video_decoder.cpp:353-357) and from NVDEC's
// fixed-function pipeline (impls/nvidia/nvidia_video_decoder.cpp:335,489).
// All arithmetic is 8-bit samples with <= 20-bit integer intermediates: these
// are stencils on bytes, bound by memory latency/bandwidth -- no MFMA.
//
// Bit-exactness: every formula below is the oracle's (oracle/h264_recon.c),
// i.e. ITU-T H.264 8.3 (intra), 8.4.2.2 (interpolation), 8.4.2.3 (weighting),
// 8.5 (transforms), 8.7 (deblocking); tests/test_gpu_decode.py compares whole
// pictures bit for bit.
#include "frame_job.h"
#include "prof.h"

// How long a wavefront that waits for another unit's progress word sleeps between two looks, in units of the original
// 64 / 128 clocks (an experiment knob: the HEVC kernels gained from looking less often).
#ifndef HW_POLL_SCALE
#define HW_POLL_SCALE 1
#endif

#include <hip/hip_runtime.h>

namespace hwang {
namespace amd {

namespace {

__constant__ uint8_t c_norm4[6][3] = {{10, 16, 13}, {11, 18, 14}, {13, 20, 16},
                                      {14, 23, 18}, {16, 25, 20}, {18, 29, 23}};
__constant__ uint8_t c_norm8[6][6] = {{20, 18, 32, 19, 25, 24}, {22, 19, 35, 21, 28, 26},
                                      {26, 23, 42, 24, 33, 31}, {28, 25, 45, 26, 35, 33},
                                      {32, 28, 51, 30, 40, 38}, {36, 32, 58, 34, 46, 43}};
__constant__ uint8_t c_alpha[52] = {0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,
                                    0,  0,  0,  4,  4,  5,  6,  7,  8,  9,  10, 12, 13,
                                    15, 17, 20, 22, 25, 28, 32, 36, 40, 45, 50, 56, 63,
                                    71, 80, 90, 101, 113, 127, 144, 162, 182, 203, 226, 255, 255};
__constant__ uint8_t c_beta[52] = {0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,
                                   0,  0,  0,  2,  2,  2,  3,  3,  3,  3,  4,  4,  4,
                                   6,  6,  7,  7,  8,  8,  9,  9,  10, 10, 11, 11, 12,
                                   12, 13, 13, 14, 14, 15, 15, 16, 16, 17, 17, 18, 18};
__constant__ uint8_t c_tc0[52][3] = {
    {0, 0, 0},   {0, 0, 0},   {0, 0, 0},   {0, 0, 0},   {0, 0, 0},   {0, 0, 0},   {0, 0, 0},
    {0, 0, 0},   {0, 0, 0},   {0, 0, 0},   {0, 0, 0},   {0, 0, 0},   {0, 0, 0},   {0, 0, 0},
    {0, 0, 0},   {0, 0, 0},   {0, 0, 0},   {0, 0, 1},   {0, 0, 1},   {0, 0, 1},   {0, 0, 1},
    {0, 1, 1},   {0, 1, 1},   {1, 1, 1},   {1, 1, 1},   {1, 1, 1},   {1, 1, 1},   {1, 1, 2},
    {1, 1, 2},   {1, 1, 2},   {1, 1, 2},   {1, 2, 3},   {1, 2, 3},   {2, 2, 3},   {2, 2, 4},
    {2, 3, 4},   {2, 3, 4},   {3, 3, 5},   {3, 4, 6},   {3, 4, 6},   {4, 5, 7},   {4, 5, 8},
    {4, 6, 9},   {5, 7, 10},  {6, 8, 11},  {6, 8, 13},  {7, 10, 14}, {8, 11, 16}, {9, 12, 18},
    {10, 13, 20}, {11, 15, 23}, {13, 17, 25}};

__device__ __forceinline__ int iclip(int lo, int hi, int v) { return v < lo ? lo : (v > hi ? hi : v); }
__device__ __forceinline__ int clip8(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }
__device__ __forceinline__ int iabs(int v) { return v < 0 ? -v : v; }
__device__ __forceinline__ int imin(int a, int b) { return a < b ? a : b; }

// Pointers that the kernels read out of the job table are "generic" to the
// compiler, which then emits flat_load / flat_store.  Flat operations also
// count on the LDS counter (lgkmcnt), so every wait for an LDS result would
// wait for the global prefetches in flight as well - fatal for the row
// pipelines, whose whole point is to keep loads and stores in flight across
// LDS work.  All of these pointers are HBM (or pinned host memory mapped for
// the GPU), so the accessors below go through address space 1 explicitly.
#define HW_AS1 __attribute__((address_space(1)))
__device__ __forceinline__ uint32_t gld32(const void* p) { return *(const HW_AS1 uint32_t*)p; }
__device__ __forceinline__ uint16_t gld16(const void* p) { return *(const HW_AS1 uint16_t*)p; }
__device__ __forceinline__ int gld16s(const void* p) { return *(const HW_AS1 int16_t*)p; }
__device__ __forceinline__ uint8_t gld8(const void* p) { return *(const HW_AS1 uint8_t*)p; }
__device__ __forceinline__ void gst32(void* p, uint32_t v) { *(HW_AS1 uint32_t*)p = v; }
__device__ __forceinline__ void gst16(void* p, uint16_t v) { *(HW_AS1 uint16_t*)p = v; }
__device__ __forceinline__ void gst8(void* p, uint8_t v) { *(HW_AS1 uint8_t*)p = v; }
template <class T>
__device__ __forceinline__ T gload(const T* p) {  // whole record, dword-wise
  static_assert(sizeof(T) % 4 == 0, "dword-sized records only");
  T out;
  uint32_t* o = reinterpret_cast<uint32_t*>(&out);
#pragma unroll
  for (unsigned i = 0; i < sizeof(T) / 4; ++i) o[i] = gld32(reinterpret_cast<const uint32_t*>(p) + i);
  return out;
}

__device__ __forceinline__ int level_scale4(const uint8_t* weight, int qp, int pos) {
  const int i = pos >> 2, j = pos & 3;
  const int cls = ((i & 1) == 0 && (j & 1) == 0) ? 0 : (((i & 1) && (j & 1)) ? 1 : 2);
  return weight[pos] * c_norm4[qp % 6][cls];
}
__device__ __forceinline__ int level_scale8(const uint8_t* weight, int qp, int pos) {
  const int i = pos >> 3, j = pos & 7;
  int cls;
  if ((i & 3) == 0 && (j & 3) == 0) cls = 0;
  else if ((i & 1) && (j & 1)) cls = 1;
  else if ((i & 3) == 2 && (j & 3) == 2) cls = 2;
  else if (((i & 3) == 0 && (j & 1)) || ((i & 1) && (j & 3) == 0)) cls = 3;
  else if (((i & 3) == 0 && (j & 3) == 2) || ((i & 3) == 2 && (j & 3) == 0)) cls = 4;
  else cls = 5;
  return weight[pos] * c_norm8[qp % 6][cls];
}

// Agent-scope accesses (sc1 on gfx950): they bypass the per-XCD L2 (and the CU's L1), so it does not matter which
// XCDs - or which workgroups of a launch - the two ends of a hand-off land on.
__device__ __forceinline__ uint64_t agent_ld64(const uint64_t* p) {
  return __hip_atomic_load((const HW_AS1 uint64_t*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ uint32_t agent_ld32(const void* p) {
  return __hip_atomic_load((const HW_AS1 uint32_t*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ int agent_ld8(const void* p) {
  return __hip_atomic_load((const HW_AS1 uint8_t*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void agent_st32(void* p, uint32_t v) {
  __hip_atomic_store((HW_AS1 uint32_t*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void agent_st16(void* p, uint16_t v) {
  __hip_atomic_store((HW_AS1 uint16_t*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void agent_st64(uint64_t* p, uint64_t v) {
  __hip_atomic_store((HW_AS1 uint64_t*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// Sample accessors of the reconstruction stages.  FRESH = the samples are produced / consumed by other workgroups of
// the SAME launch (chain launches, k_chain below: no kernel boundary makes them visible), so every access is
// agent-scope; otherwise the plain cached forms.
template <bool FRESH> __device__ __forceinline__ uint32_t sld32(const void* p) { return FRESH ? agent_ld32(p) : gld32(p); }
template <bool FRESH> __device__ __forceinline__ int sld8(const void* p) { return FRESH ? agent_ld8(p) : (int)gld8(p); }
template <bool FRESH> __device__ __forceinline__ void sst32(void* p, uint32_t v) { if (FRESH) agent_st32(p, v); else gst32(p, v); }
template <bool FRESH> __device__ __forceinline__ void sst16(void* p, uint16_t v) { if (FRESH) agent_st16(p, v); else gst16(p, v); }

constexpr int kLevels = 26 * 16;
__device__ __forceinline__ void wave_sync();

__device__ __forceinline__ int tap6(int a, int b, int c, int d, int e, int f) { return a - 5 * b + 20 * c + 20 * d - 5 * e + f; }

__device__ int implicit_w1(const SliceParams& sp, int r0, int r1) {
  if (sp.ref_long[0][r0] || sp.ref_long[1][r1]) return 32;
  const int tb = iclip(-128, 127, sp.cur_poc - sp.ref_poc[0][r0]);
  const int td = iclip(-128, 127, sp.ref_poc[1][r1] - sp.ref_poc[0][r0]);
  if (td == 0) return 32;
  const int tx = (16384 + iabs(td / 2)) / td;
  const int dsf = iclip(-1024, 1023, (tb * tx + 32) >> 6);
  const int w = dsf >> 2;
  return (w >= -64 && w <= 128) ? w : 32;
}

// weighted combination of up to two predictions (8.4.2.3); plane 0 luma, 1/2 chroma
__device__ int combine_pred(const SliceParams& sp, int plane, bool use0, bool use1, int s0, int s1, int r0, int r1) {
  if (sp.weight_mode == 1) {
    const int ld = plane ? sp.chroma_log2_denom : sp.luma_log2_denom;
    int w0 = 0, w1 = 0, o0 = 0, o1 = 0;
    if (use0) {
      w0 = plane ? sp.chroma_weight[0][r0][plane - 1] : sp.luma_weight[0][r0];
      o0 = plane ? sp.chroma_offset[0][r0][plane - 1] : sp.luma_offset[0][r0];
    }
    if (use1) {
      w1 = plane ? sp.chroma_weight[1][r1][plane - 1] : sp.luma_weight[1][r1];
      o1 = plane ? sp.chroma_offset[1][r1][plane - 1] : sp.luma_offset[1][r1];
    }
    int v;
    if (use0 && use1) {
      v = ((s0 * w0 + s1 * w1 + (1 << ld)) >> (ld + 1)) + ((o0 + o1 + 1) >> 1);
    } else {
      const int sv = use0 ? s0 : s1, w = use0 ? w0 : w1, o = use0 ? o0 : o1;
      v = ld >= 1 ? ((sv * w + (1 << (ld - 1))) >> ld) + o : sv * w + o;
    }
    return clip8(v);
  }
  if (use0 && use1) {
    if (sp.weight_mode == 2) {
      const int w1 = implicit_w1(sp, r0, r1), w0 = 64 - w1;
      return clip8((s0 * w0 + s1 * w1 + 32) >> 6);
    }
    return (s0 + s1 + 1) >> 1;
  }
  return use0 ? s0 : s1;
}

// Everything one wave wrote to LDS is visible to its other lanes afterwards.
// LDS instructions of a wave execute in issue order, so this only has to stop
// the COMPILER from moving LDS accesses across it; a memory-model fence (even
// at wavefront scope the backend pairs it with s_waitcnt vmcnt(0)) would drain
// the global prefetches and stores that are deliberately left in flight.
__device__ __forceinline__ void wave_sync() {
  __asm__ volatile("" ::: "memory");
  __builtin_amdgcn_wave_barrier();
  __asm__ volatile("" ::: "memory");
}

// ---------------------------------------------------------------------------
// Residual of one macroblock, computed by ONE wave (64 lanes) into LDS.
//   work[384] ints scratch, dcs[24] ints, res_y[256], res_c[128] int16, lv[kLevels] int16
//
// The coefficient stream holds one word per NON-ZERO level ((group * 16 +
// position) << 16 | level, MbInfo::ncoef of them; a coded 4x4 block of a P
// picture has one or two).  Each lane takes an entry, scales it (8.5.9, 8.5.12.1,
// 8.5.13) and drops it at its place in the zeroed work arrays - no loop over the
// 384 coefficient slots, no address arithmetic over the mask.  DC levels of
// Intra16x16 luma and of chroma go through their Hadamard transforms first
// (8.5.10, 8.5.11).  kDenseCoefs (more than kMaxSparseCoefs levels): the coded
// groups follow densely and every slot is visited.
// ---------------------------------------------------------------------------
__device__ void wave_residual(const MbInfo& mb, const uint32_t* coefs, const PictureParams& pp, int lane,
                              int* work, int* dcs, int16_t* res_y, int16_t* res_c, int16_t* lv) {
  const bool intra = mb.flags & MBF_INTRA;
  const bool t8 = mb.flags & MBF_T8X8;
  const bool i16 = mb.kind == MBK_I16x16;
  const int qp = mb.qp;
  int* dy = work;
  int* dc = work + 256;
  // zero the work arrays and the two DC staging rows (lv[0..15] luma DC, lv[16..23] chroma DC)
#pragma unroll
  for (int t = lane; t < 384; t += 64) work[t] = 0;
  if (lane < 12) reinterpret_cast<uint32_t*>(lv)[lane] = 0;
  wave_sync();
  // wave-uniform scaling parameters
  const int qs = qp / 6, qr = qp - 6 * qs;
  const int qcs0 = mb.qpc[0] / 6, qcr0 = mb.qpc[0] - 6 * qcs0, qcs1 = mb.qpc[1] / 6, qcr1 = mb.qpc[1] - 6 * qcs1;
  auto place = [&](int idx, int level) {
    const int g = idx >> 4, pos = idx & 15;
    if (g < 16) {
      if (t8) {
        const int k = idx & 63;
        const int sc = level * level_scale8(pp.scaling8[intra ? 0 : 1], qr, k);
        dy[idx] = qs >= 6 ? sc << (qs - 6) : (sc + (1 << (5 - qs))) >> (6 - qs);
      } else {
        const int sc = level * level_scale4(pp.scaling4[intra ? 0 : 3], qr, pos);
        dy[idx] = qs >= 4 ? sc << (qs - 4) : (sc + (1 << (3 - qs))) >> (4 - qs);
      }
    } else if (g == kMaskLumaDc) {
      lv[pos] = (int16_t)level;
    } else if (g == kMaskChromaDc) {
      lv[16 + pos] = (int16_t)level;
    } else {
      const int c = idx - 16 * kMaskChromaAc, p = c >> 6;  // c: plane * 64 + block * 16 + position
      const int s6 = p ? qcs1 : qcs0, r6 = p ? qcr1 : qcr0;
      const int sc = level * level_scale4(pp.scaling4[(intra ? 0 : 3) + 1 + p], r6, pos);
      dc[c] = s6 >= 4 ? sc << (s6 - 4) : (sc + (1 << (3 - s6))) >> (4 - s6);
    }
  };
  const uint32_t* src = coefs + mb.coef_offset;
  if (mb.ncoef != kDenseCoefs) {
    for (int t = lane; t < mb.ncoef; t += 64) {
      const uint32_t w = gld32(src + t);
      place((int)(w >> 16), (int)(int16_t)(w & 0xffff));
    }
  } else {
    for (int t = lane; t < kLevels; t += 64) {
      const int g = t >> 4;
      if ((mb.coded_mask >> g) & 1) {
        const int level = gld16s(reinterpret_cast<const int16_t*>(src) + 16 * __popc(mb.coded_mask & ((1u << g) - 1)) + (t & 15));
        if (level) place(t, level);
      }
    }
  }
  wave_sync();
  if (i16 && lane < 16 && ((mb.coded_mask >> kMaskLumaDc) & 1)) {
    const int i = lane >> 2, j = lane & 3;
    int f = 0;
    for (int a = 0; a < 4; ++a) {
      const int sa = (i == 0 || (i == 1 && a < 2) || (i == 2 && (a == 0 || a == 3)) || (i == 3 && (a & 1) == 0)) ? 1 : -1;
      for (int b = 0; b < 4; ++b) {
        const int sb = (j == 0 || (j == 1 && b < 2) || (j == 2 && (b == 0 || b == 3)) || (j == 3 && (b & 1) == 0)) ? 1 : -1;
        f += sa * sb * lv[a * 4 + b];
      }
    }
    const int ls = level_scale4(pp.scaling4[0], qr, 0);
    dy[lane * 16] = qs >= 6 ? (f * ls) << (qs - 6) : (f * ls + (1 << (5 - qs))) >> (6 - qs);
  }
  if (lane >= 16 && lane < 24 && ((mb.coded_mask >> kMaskChromaDc) & 1)) {
    const int p = (lane - 16) >> 2, k = (lane - 16) & 3;
    const int16_t* c = lv + 16 + 4 * p;
    const int c0 = c[0], c1 = c[1], c2 = c[2], c3 = c[3];
    const int f = k == 0 ? c0 + c1 + c2 + c3 : k == 1 ? c0 - c1 + c2 - c3 : k == 2 ? c0 + c1 - c2 - c3 : c0 - c1 - c2 + c3;
    const int s6 = p ? qcs1 : qcs0, r6 = p ? qcr1 : qcr0;
    const int ls = level_scale4(pp.scaling4[(intra ? 0 : 3) + 1 + p], r6, 0);
    dc[p * 64 + k * 16] = ((f * ls) << s6) >> 5;
  }
  wave_sync();
  // Inverse transforms, per plane group only where something was coded (wave uniform): a P macroblock often
  // carries luma or chroma levels only.
  const bool has_luma = (mb.coded_mask & 0x1ffffu) != 0, has_chroma = (mb.coded_mask >> kMaskChromaDc) != 0;
  if (!has_luma) {
    reinterpret_cast<uint32_t*>(res_y)[lane] = 0;
    reinterpret_cast<uint32_t*>(res_y)[64 + lane] = 0;
  }
  if (!has_chroma) reinterpret_cast<uint32_t*>(res_c)[lane] = 0;
  // rows
  if (!has_luma) {
  } else if (mb.flags & MBF_T8X8) {
    if (lane < 32) {
      int* d = dy + (lane >> 3) * 64 + (lane & 7) * 8;
      const int d0 = d[0], d1 = d[1], d2 = d[2], d3 = d[3], d4 = d[4], d5 = d[5], d6 = d[6], d7 = d[7];
      const int e0 = d0 + d4, e1 = -d3 + d5 - d7 - (d7 >> 1), e2 = d0 - d4, e3 = d1 + d7 - d3 - (d3 >> 1);
      const int e4 = (d2 >> 1) - d6, e5 = -d1 + d7 + d5 + (d5 >> 1), e6 = d2 + (d6 >> 1), e7 = d3 + d5 + d1 + (d1 >> 1);
      const int f0 = e0 + e6, f1 = e1 + (e7 >> 2), f2 = e2 + e4, f3 = e3 + (e5 >> 2);
      const int f4 = e2 - e4, f5 = (e3 >> 2) - e5, f6 = e0 - e6, f7 = e7 - (e1 >> 2);
      d[0] = f0 + f7; d[1] = f2 + f5; d[2] = f4 + f3; d[3] = f6 + f1;
      d[4] = f6 - f1; d[5] = f4 - f3; d[6] = f2 - f5; d[7] = f0 - f7;
    }
  } else {
    int* d = dy + (lane >> 2) * 16 + (lane & 3) * 4;
    const int e0 = d[0] + d[2], e1 = d[0] - d[2], e2 = (d[1] >> 1) - d[3], e3 = d[1] + (d[3] >> 1);
    d[0] = e0 + e3; d[1] = e1 + e2; d[2] = e1 - e2; d[3] = e0 - e3;
  }
  if (has_chroma && lane < 32) {
    int* d = dc + (lane >> 2) * 16 + (lane & 3) * 4;
    const int e0 = d[0] + d[2], e1 = d[0] - d[2], e2 = (d[1] >> 1) - d[3], e3 = d[1] + (d[3] >> 1);
    d[0] = e0 + e3; d[1] = e1 + e2; d[2] = e1 - e2; d[3] = e0 - e3;
  }
  wave_sync();
  // columns
  if (!has_luma) {
  } else if (mb.flags & MBF_T8X8) {
    if (lane < 32) {
      const int b8 = lane >> 3, c = lane & 7;
      const int* d = dy + b8 * 64 + c;
      const int d0 = d[0], d1 = d[8], d2 = d[16], d3 = d[24], d4 = d[32], d5 = d[40], d6 = d[48], d7 = d[56];
      const int e0 = d0 + d4, e1 = -d3 + d5 - d7 - (d7 >> 1), e2 = d0 - d4, e3 = d1 + d7 - d3 - (d3 >> 1);
      const int e4 = (d2 >> 1) - d6, e5 = -d1 + d7 + d5 + (d5 >> 1), e6 = d2 + (d6 >> 1), e7 = d3 + d5 + d1 + (d1 >> 1);
      const int f0 = e0 + e6, f1 = e1 + (e7 >> 2), f2 = e2 + e4, f3 = e3 + (e5 >> 2);
      const int f4 = e2 - e4, f5 = (e3 >> 2) - e5, f6 = e0 - e6, f7 = e7 - (e1 >> 2);
      const int o[8] = {f0 + f7, f2 + f5, f4 + f3, f6 + f1, f6 - f1, f4 - f3, f2 - f5, f0 - f7};
      const int x = (b8 & 1) * 8 + c, y0 = (b8 >> 1) * 8;
#pragma unroll
      for (int r = 0; r < 8; ++r) res_y[(y0 + r) * 16 + x] = (int16_t)((o[r] + 32) >> 6);
    }
  } else {
    const int blk = lane >> 2, c = lane & 3;
    const int* d = dy + blk * 16 + c;
    const int e0 = d[0] + d[8], e1 = d[0] - d[8], e2 = (d[4] >> 1) - d[12], e3 = d[4] + (d[12] >> 1);
    const int x = (blk & 3) * 4 + c, y0 = (blk >> 2) * 4;
    res_y[(y0 + 0) * 16 + x] = (int16_t)((e0 + e3 + 32) >> 6);
    res_y[(y0 + 1) * 16 + x] = (int16_t)((e1 + e2 + 32) >> 6);
    res_y[(y0 + 2) * 16 + x] = (int16_t)((e1 - e2 + 32) >> 6);
    res_y[(y0 + 3) * 16 + x] = (int16_t)((e0 - e3 + 32) >> 6);
  }
  if (has_chroma && lane < 32) {
    const int pb = lane >> 2, c = lane & 3;
    const int* d = dc + pb * 16 + c;
    const int e0 = d[0] + d[8], e1 = d[0] - d[8], e2 = (d[4] >> 1) - d[12], e3 = d[4] + (d[12] >> 1);
    const int p = pb >> 2, blk = pb & 3;
    const int x = (blk & 1) * 4 + c, y0 = (blk >> 1) * 4;
    int16_t* r = res_c + p * 64;
    r[(y0 + 0) * 8 + x] = (int16_t)((e0 + e3 + 32) >> 6);
    r[(y0 + 1) * 8 + x] = (int16_t)((e1 + e2 + 32) >> 6);
    r[(y0 + 2) * 8 + x] = (int16_t)((e1 - e2 + 32) >> 6);
    r[(y0 + 3) * 8 + x] = (int16_t)((e0 - e3 + 32) >> 6);
  }
  wave_sync();
}


// ---------------------------------------------------------------------------
// Motion compensation (8.4.2.2), register resident.
//
// A lane predicts 4 horizontally adjacent luma samples (one row of a 4x4
// block, so one motion vector) and 2 chroma samples.  What the 6-tap filter
// needs - bytes x-2 .. x+6 of rows y-2 .. y+3, or less when the vector is
// integer in one direction - is fetched straight from the reference surface
// as UNALIGNED dwords (gfx950 serves them from the vector L1, which neighbour
// lanes' overlapping windows keep hot), at most 18 per lane and list, and
// filtered in registers: no LDS staging, no barrier.  A window that leaves the
// picture takes a byte-wise path with clamped coordinates (the padding of
// unrestricted motion vectors); that is decided per lane.
// ---------------------------------------------------------------------------
__device__ __forceinline__ uint32_t ld32u(const uint8_t* p) { return gld32(p); }

// 12 bytes (3 dwords) of row y starting at column xs, coordinates clamped
template <bool FRESH>
__device__ __forceinline__ void row_clamped(const uint8_t* plane, int pitch, int W, int H, int xs, int y, int nd,
                                            uint32_t* w) {
  const uint8_t* row = plane + (size_t)iclip(0, H - 1, y) * pitch;
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    if (d >= nd) break;
    uint32_t v = 0;
#pragma unroll
    for (int b = 0; b < 4; ++b) v |= (uint32_t)sld8<FRESH>(row + iclip(0, W - 1, xs + 4 * d + b)) << (8 * b);
    w[d] = v;
  }
}

#define BYTE_OF(w, k) (((w)[(k) >> 2] >> (8 * ((k) & 3))) & 0xff)

__device__ __forceinline__ int tap6_row(const uint32_t* w, int i) {  // bytes i .. i+5 of the row
  return tap6(BYTE_OF(w, i), BYTE_OF(w, i + 1), BYTE_OF(w, i + 2), BYTE_OF(w, i + 3), BYTE_OF(w, i + 4), BYTE_OF(w, i + 5));
}

__device__ __forceinline__ void luma_filter4(const uint32_t (*w)[3], int fx, int fy, int* out);

// plane: reference luma; (px, py): position of the lane's first sample; mv in quarter samples
template <bool FRESH>
__device__ __forceinline__ void luma_pred4(const uint8_t* plane, int pitch, int W, int H, int px, int py, int mvx,
                                           int mvy, int* out) {
  const int fx = mvx & 3, fy = mvy & 3;
  const int x = px + (mvx >> 2), y = py + (mvy >> 2);
  const bool horiz = fx != 0, vert = fy != 0;
  const int xs = horiz ? x - 2 : x;      // first byte that is used
  const int nd = horiz ? 3 : 1;          // dwords per row (9 resp. 4 bytes used)
  const int r0 = vert ? 0 : 2, r1 = vert ? 5 : 2;  // rows y-2+r
  const bool inside = xs >= 0 && xs + (horiz ? 9 : 4) <= W && y - 2 + r0 >= 0 && y - 2 + r1 < H;
  uint32_t w[6][3];
  if (inside) {
    const uint8_t* base = plane + (size_t)(y - 2) * pitch + xs;
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      if (r < r0 || r > r1) continue;
      w[r][0] = sld32<FRESH>(base + r * pitch);
      if (horiz) {
        w[r][1] = sld32<FRESH>(base + r * pitch + 4);
        w[r][2] = sld32<FRESH>(base + r * pitch + 8);
      }
    }
  } else {
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      if (r < r0 || r > r1) continue;
      row_clamped<FRESH>(plane, pitch, W, H, xs, y - 2 + r, nd, w[r]);
    }
  }
  luma_filter4(w, fx, fy, out);
}

// The arithmetic of 8.4.2.2.1 on a lane's 6 x 12-byte neighbourhood.  w[r] is row
// y - 2 + r; with a horizontal fraction byte 0 is column x - 2 (nine bytes used),
// without one byte 0 is column x (four bytes used); without a vertical fraction
// only w[2] is read.
__device__ __forceinline__ void luma_filter4(const uint32_t (*w)[3], int fx, int fy, int* out) {
  const bool horiz = fx != 0, vert = fy != 0;
  if (!horiz && !vert) {
#pragma unroll
    for (int i = 0; i < 4; ++i) out[i] = BYTE_OF(w[2], i);
    return;
  }
  if (!vert) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int b = clip8((tap6_row(w[2], i) + 16) >> 5);
      out[i] = fx == 2 ? b : ((fx == 1 ? BYTE_OF(w[2], i + 2) : BYTE_OF(w[2], i + 3)) + b + 1) >> 1;
    }
    return;
  }
  if (!horiz) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int h = clip8((tap6(BYTE_OF(w[0], i), BYTE_OF(w[1], i), BYTE_OF(w[2], i), BYTE_OF(w[3], i), BYTE_OF(w[4], i),
                                BYTE_OF(w[5], i)) + 16) >> 5);
      out[i] = fy == 2 ? h : ((fy == 1 ? BYTE_OF(w[2], i) : BYTE_OF(w[3], i)) + h + 1) >> 1;
    }
    return;
  }
  if (fx == 2 || fy == 2) {
    int j1[4] = {0, 0, 0, 0}, bsel[4] = {0, 0, 0, 0};
    const int brow = fy == 3 ? 3 : 2;
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      const int wt = (r == 0 || r == 5) ? 1 : ((r == 1 || r == 4) ? -5 : 20);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int bb = tap6_row(w[r], i);
        j1[i] += wt * bb;
        if (r == brow) bsel[i] = bb;
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int j = clip8((j1[i] + 512) >> 10);
      if (fx == 2 && fy == 2) {
        out[i] = j;
      } else if (fx == 2) {
        out[i] = (clip8((bsel[i] + 16) >> 5) + j + 1) >> 1;
      } else {
        const int c = i + 2 + (fx == 3);  // column of the half sample h; resolved per lane below
        const int h = fx == 3 ? tap6(BYTE_OF(w[0], i + 3), BYTE_OF(w[1], i + 3), BYTE_OF(w[2], i + 3), BYTE_OF(w[3], i + 3),
                                     BYTE_OF(w[4], i + 3), BYTE_OF(w[5], i + 3))
                              : tap6(BYTE_OF(w[0], i + 2), BYTE_OF(w[1], i + 2), BYTE_OF(w[2], i + 2), BYTE_OF(w[3], i + 2),
                                     BYTE_OF(w[4], i + 2), BYTE_OF(w[5], i + 2));
        (void)c;
        out[i] = (clip8((h + 16) >> 5) + j + 1) >> 1;
      }
    }
    return;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int bb = fy == 3 ? tap6_row(w[3], i) : tap6_row(w[2], i);
    const int h = fx == 3 ? tap6(BYTE_OF(w[0], i + 3), BYTE_OF(w[1], i + 3), BYTE_OF(w[2], i + 3), BYTE_OF(w[3], i + 3),
                                 BYTE_OF(w[4], i + 3), BYTE_OF(w[5], i + 3))
                          : tap6(BYTE_OF(w[0], i + 2), BYTE_OF(w[1], i + 2), BYTE_OF(w[2], i + 2), BYTE_OF(w[3], i + 2),
                                 BYTE_OF(w[4], i + 2), BYTE_OF(w[5], i + 2));
    out[i] = (clip8((bb + 16) >> 5) + clip8((h + 16) >> 5) + 1) >> 1;
  }
}

__device__ __forceinline__ void chroma_filter2(uint32_t a, uint32_t c, int fx, int fy, int* out);
// two horizontally adjacent chroma samples (8.4.2.2.2); mv in eighth samples
template <bool FRESH>
__device__ __forceinline__ void chroma_pred2(const uint8_t* plane, int pitch, int W, int H, int px, int py, int mvx,
                                             int mvy, int* out) {
  const int fx = mvx & 7, fy = mvy & 7;
  const int x = px + (mvx >> 3), y = py + (mvy >> 3);
  uint32_t a, c;
  if (x >= 0 && x + 3 <= W && y >= 0 && y + 1 < H) {
    a = sld32<FRESH>(plane + (size_t)y * pitch + x);
    c = sld32<FRESH>(plane + (size_t)(y + 1) * pitch + x);
  } else {
    uint32_t t[3];
    row_clamped<FRESH>(plane, pitch, W, H, x, y, 1, t);
    a = t[0];
    row_clamped<FRESH>(plane, pitch, W, H, x, y + 1, 1, t);
    c = t[0];
  }
  chroma_filter2(a, c, fx, fy, out);
}
// a: three bytes of row y from column x, c: of row y + 1
__device__ __forceinline__ void chroma_filter2(uint32_t a, uint32_t c, int fx, int fy, int* out) {
  const int a0 = a & 0xff, a1 = (a >> 8) & 0xff, a2 = (a >> 16) & 0xff;
  const int c0 = c & 0xff, c1 = (c >> 8) & 0xff, c2 = (c >> 16) & 0xff;
  out[0] = ((8 - fx) * (8 - fy) * a0 + fx * (8 - fy) * a1 + (8 - fx) * fy * c0 + fx * fy * c1 + 32) >> 6;
  out[1] = ((8 - fx) * (8 - fy) * a1 + fx * (8 - fy) * a2 + (8 - fx) * fy * c1 + fx * fy * c2 + 32) >> 6;
}

constexpr int kMaxRows = kMaxMbRows;
// Row pipelines whose pictures are spread over several workgroups ("bands" of
// kBandWaves rows, one wave per row; k_mv explains why).  BandCtl is the
// per-picture hand-off area in global memory (JobDesc::band_ctl); every word in
// it is tagged with JobDesc::epoch, nothing is cleared between pictures.
constexpr int kBandWaves = 4;
constexpr int kMaxBands = kMaxRows / kBandWaves;
struct BandCtl {
  uint64_t db_prog[kMaxBands];   // k_deblock: epoch << 32 | macroblocks of the band's last row that are final
  uint64_t in_prog[kMaxBands];   // k_intra (neighbours read from the surface): likewise
  // Chain launches (k_chain): progress of the picture's stages per macroblock ROW, for the other stages of the same
  // picture and for the pictures that predict from it inside the same launch.
  uint64_t fin[kMaxRows];        // deblocking: epoch << 32 | macroblocks of row r that row r's own filtering has left final
  uint64_t intra_prog[kMaxRows]; // intra prediction: epoch << 32 | macroblocks of row r in front of which every intra
                                 // macroblock has landed (inter macroblocks are not its business)
  uint64_t inter_mask[kMaxRows]; // motion compensation: bit j = macroblocks 8 j .. 8 j + 7 of row r (a run, kChainRun:
                                 // one inter unit) have landed; zeroed by k_upload
  // tagged words {datum, epoch}; with B = bands of the picture:
  //   [0, B * kLineMbs * 5)                      k_mv: [band][mbx] MvEdge as five words
  //   [B * kLineMbs * 5, B * kLineMbs * 13)      k_intra (LDS neighbours): [band][mbx] bottom rows, 32 bytes as eight words
  uint64_t lines[1];
};
constexpr int kMvLineWords = 5, kIntraLineWords = 8;
__device__ __forceinline__ int bands_of(int mb_height) { return (mb_height + kBandWaves - 1) / kBandWaves; }

// A bounded poll of another band expired.  The polls are bounded so that a
// protocol error can never hang the GPU; this makes the same error loud on the
// host (GpuExecutor::complete fails the batch) instead of silent wrong samples.
// Chain launches carry four control words (ChainArgs::tickets = JobDesc::chain_ctl, zeroed by the host with every
// launch's unit table): [0] / [1] the tickets, [2] the first fault of THIS launch - once one poll has expired the
// others give up at their next look instead of each sitting out its own bound, so a launch with a broken hand-off
// ends in a fraction of a second; nothing outlives the launch - and [3] a heartbeat that is raised whenever a row of a
// unit or a run of inter macroblocks is through (a same-address atomic per published macroblock was measured: it
// serialises in the L2 and cost the reference's caller a quarter of its rate).  A poll only counts tries during which
// neither the word it watches nor the heartbeat moved: its bound is "neither my producer nor anything else in the
// launch has got anywhere for that long", not "my producer took that long" - a row unit may sit behind a chain of
// pictures, on a device it shares with other engines' launches and other processes.
constexpr int kChainFaultWord = 2, kChainBeatWord = 3;
constexpr int kChainTries = 1 << 17;  // ~0.1 - 0.2 s of polling with the whole launch standing still
static_assert(kChainRun == 8, "chain_need_inter: one bit of BandCtl::inter_mask per run of 8 macroblocks (upto_mb >> 3)");
__device__ __forceinline__ void report_fault(const JobDesc& job, uint32_t code) {
  if (job.fault) __hip_atomic_store((HW_AS1 uint32_t*)job.fault, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  if (job.chain_ctl) __hip_atomic_store((HW_AS1 uint32_t*)(job.chain_ctl + kChainFaultWord), code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// (one lane calls this)
__device__ __forceinline__ void chain_beat(const uint32_t* chain_ctl) {
  (void)__hip_atomic_fetch_add((HW_AS1 uint32_t*)(chain_ctl + kChainBeatWord), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// Development aid: what every wave of a chain launch is doing (hwang_debug_chain_state reads it while the launch is
// still running): code << 24 | a << 12 | b per workgroup and wave.
__device__ uint32_t g_chain_dbg[256 * 8];
__device__ __forceinline__ void chain_dbg(uint32_t code, int a, int b) {
  if ((threadIdx.x & 63) == 0 && blockIdx.x < 256)
    __hip_atomic_store((HW_AS1 uint32_t*)&g_chain_dbg[blockIdx.x * 8 + (threadIdx.x >> 6)], code << 24 | (static_cast<uint32_t>(a) & 0xfff) << 12 | (static_cast<uint32_t>(b) & 0xfff),
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
#ifdef EXP_STAMPS
// Experiment (tools/experiments/exp_stamps.py): where one wave's time goes inside a row kernel - s_memtime at up to 8
// points per macroblock of ONE row of the first picture of the launch.
__device__ uint64_t g_stamps[256 * 8];
#define STAMP(on, mb, slot) do { if ((on) && (threadIdx.x & 63) == 0 && (mb) < 256) g_stamps[(mb) * 8 + (slot)] = __builtin_amdgcn_s_memtime(); } while (0)
#define STAMP_VALUE(on, mb, v) do { if ((on) && (threadIdx.x & 63) == 0 && (mb) < 256) g_stamps[(mb) * 8 + 7] = (v); } while (0)
#else
#define STAMP(on, mb, slot) do { } while (0)
#define STAMP_VALUE(on, mb, v) do { } while (0)
#endif
// Every 256th try of a chain poll: has another poll of this launch given up (then so does this one), has anything in
// the launch moved since this poll last looked (then its count starts again)?
struct ChainPoll {
  const uint32_t* ctl;
  uint32_t beat = 0;
  uint64_t watched = 0;
  int tries = 0;
  bool armed = false;  // (the heartbeat is first read at the 256th try: a wait that is over at once costs nothing extra)
  __device__ __forceinline__ explicit ChainPoll(const uint32_t* chain_ctl) : ctl(chain_ctl) {}
  // `seen`: the word this poll watches, as just read.  false: give up
  __device__ __forceinline__ bool again(uint64_t seen) {
    if (seen != watched) {  // my producer moved (not far enough yet)
      watched = seen;
      tries = 0;
    }
    if ((++tries & 255) == 0) {
      if (__hip_atomic_load((const HW_AS1 uint32_t*)(ctl + kChainFaultWord), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return false;
      const uint32_t b = __hip_atomic_load((const HW_AS1 uint32_t*)(ctl + kChainBeatWord), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (armed && b != beat) tries = 0;
      beat = b;
      armed = true;
    }
    return tries < kChainTries;
  }
};

constexpr int kInterWaves = 4;
// Reference window of one macroblock and one list, staged in LDS with the
// macroblock's (integer) displacement already applied: luma rows y0-2 .. y0+18,
// 24 bytes from column x0-2 (21 used), chroma 9 rows of 12 bytes (9 used).
constexpr int kWinLumaRows = 21, kWinLumaDw = 6, kWinChromaRows = 9, kWinChromaDw = 3;
constexpr int kWinLumaWords = kWinLumaRows * kWinLumaDw;                                   // 126
constexpr int kWinDwords = kWinLumaWords + 2 * kWinChromaRows * kWinChromaDw;              // 180
struct InterWave {
  int work[384];
  int dcs[24];
  int16_t res_y[256];
  int16_t res_c[128];
  int16_t lv[kLevels];
  alignas(8) uint32_t win[2][kWinDwords];
};

typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ u32x2 gld64u(const uint8_t* p) { return *(const HW_AS1 u32x2*)p; }  // any byte alignment
#define HW_FN __device__ __forceinline__
#define HW_ALIGNBYTE(hi, lo, n) __builtin_amdgcn_alignbyte(hi, lo, n)
#define HW_PERM(hi, lo, sel) __builtin_amdgcn_perm(hi, lo, sel)
#define HW_SDOT4(a, b, acc) __builtin_amdgcn_sdot4(a, b, acc, false)
#define HW_UDOT4(a, b, acc) __builtin_amdgcn_udot4(a, b, acc, false)
#include "inter_filters.h"

}  // namespace

// ===========================================================================
// k_inter: one WAVE per inter macroblock, four macroblocks (64 luma bytes per
// row) per workgroup, no workgroup barriers.  Everything that is the same for
// the 64 lanes - the macroblock's index and records, the decision below, the
// fractions of its vector, the surface pointers - lives in SGPRs (the wave index
// is made provably uniform), so the vector pipe only sees per-sample work.
//   residual (dequant + IDCT) into LDS unless nothing was coded;
//   prediction.  A macroblock whose sixteen blocks share one vector and one
//   reference per list (P_Skip, 16x16, B direct with uniform motion: ~ 88 % of
//   the inter macroblocks of the bench corpus) takes the WINDOW path: the wave
//   fetches the (16 + 5)^2 luma and two (8 + 1)^2 chroma neighbourhoods of each
//   list ONCE, cooperatively - 63 eight-byte + 54 four-byte unaligned loads,
//   scalar base + 32-bit lane offset, all issued before anything waits: one
//   memory round trip - into LDS, shifted so that every lane's bytes start
//   dword-aligned.  The filters read LDS and work on packed data: horizontal
//   six-tap sums by the 8-bit dot product, vertical ones on 16-bit pairs,
//   quarter-sample averages and the default bi-prediction on four bytes at once
//   (luma_win4 / chroma_win2 above).  Other macroblocks (partitions with
//   different vectors, windows that leave the picture) take the per-lane
//   register path.  Weighted combination, + residual, one dword / one halfword store.
// Workgroups are numbered so that each XCD (blockIdx % 8) owns a contiguous
// band of the picture: its L2 then sees whole cache lines of the output and
// the overlapping reference windows of neighbouring macroblocks.
// ===========================================================================
namespace {
// agent-scope forms of the window loads (any byte alignment, like gld64u / ld32u)
HW_FN u32x2 agent_ld64u(const uint8_t* p) {
  const uint64_t v = __hip_atomic_load((const HW_AS1 uint64_t*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  u32x2 r;
  r.x = (uint32_t)v;
  r.y = (uint32_t)(v >> 32);
  return r;
}

// Chain launches: a macroblock that predicts from a picture of the same launch waits until that picture's
// deblocking has left final everything the prediction reads: luma rows `top` .. `bottom`, columns up to `right`
// (samples; chroma lies inside the same macroblocks).  BandCtl::fin[r] counts the macroblocks of row r that row r's
// own filtering - stores landed - is through with.  A macroblock of row r is final once rows r AND r + 1 are through
// with its column (the top edges of row r + 1 still change row r's last lines), and a row only stays behind the row
// above where it has top edges to filter - so every row the window touches, and the one below, is asked.  The polls
// are bounded like every other one.
HW_FN void chain_wait_reference(const JobDesc& ref, const JobDesc& job, int top, int bottom, int right) {
  const int mbh = ref.pp.mb_height, mbw = ref.pp.mb_width;
  const int first = iclip(0, mbh - 1, top >> 4), last = imin(iclip(0, mbh - 1, bottom >> 4) + 1, mbh - 1);
  const uint32_t need = (uint32_t)imin(iclip(0, mbw - 1, right >> 4) + 1, mbw);
  const uint64_t* words = static_cast<const BandCtl*>(ref.band_ctl)->fin;
  const uint32_t epoch = ref.epoch;
  for (int row = last; row >= first; --row) {  // (bottom-up: the lowest row is normally the last to get there)
    bool ok = false;
    chain_dbg(3, row, need);
    ChainPoll poll(job.chain_ctl);
    uint64_t v;
    do {
      v = agent_ld64(words + row);
      ok = static_cast<uint32_t>(v >> 32) == epoch && static_cast<uint32_t>(v) >= need;
      if (!ok) __builtin_amdgcn_s_sleep(2 * HW_POLL_SCALE);
    } while (!ok && poll.again(v));
    if (!ok) {
      report_fault(job, FAULT_CHAIN_REFERENCE);
      return;
    }
  }
}
// A stage's bounded poll of another stage's per-row word; what was seen last is kept, so a stage that trails the
// other closely asks once per stretch the other has advanced, not once per macroblock.
HW_FN bool chain_need_inter(const uint32_t* chain_ctl, const BandCtl* ctl, int row, int upto_mb, uint64_t& seen) {
  const uint64_t want = (2ull << (upto_mb >> 3)) - 1;  // every run up to the one that holds macroblock upto_mb (run 63: all ones)
  if ((seen & want) == want) return true;
  chain_dbg(1, row, upto_mb);
  ChainPoll poll(chain_ctl);
  do {
    seen = agent_ld64(&ctl->inter_mask[row]);
    if ((seen & want) == want) return true;
    __builtin_amdgcn_s_sleep(1 * HW_POLL_SCALE);
  } while (poll.again(seen));
  return false;
}
HW_FN bool chain_need_intra(const uint32_t* chain_ctl, const BandCtl* ctl, uint32_t epoch, int row, uint32_t count, uint32_t& seen) {
  if (seen >= count) return true;
  chain_dbg(2, row, count);
  ChainPoll poll(chain_ctl);
  uint64_t v;
  do {
    v = agent_ld64(&ctl->intra_prog[row]);
    if (static_cast<uint32_t>(v >> 32) == epoch) seen = static_cast<uint32_t>(v);
    if (seen >= count) return true;
    __builtin_amdgcn_s_sleep(1 * HW_POLL_SCALE);
  } while (poll.again(v));
  return false;
}
}  // namespace

// One inter macroblock by one wave.  CHAIN: as part of a unit of k_chain - references of the same launch are waited for
// and read with agent-scope loads, the macroblock's samples are stored agent-scope (k_chain raises the run's bit when
// the run's last store has landed).
struct InterPicture {  // what a wave needs from the descriptor, requested in one go
  int mb_width, mb_height, y_pitch, c_pitch;
  const MbInfo* mbs;
  const MbMotion* motions;
  const SliceParams* slices;
  Surface cur;
};
__device__ __forceinline__ InterPicture inter_picture(const JobDesc& job) {
  const PictureParams& pp = job.pp;
  InterPicture ip;
  ip.mb_width = pp.mb_width;
  ip.mb_height = pp.mb_height;
  ip.y_pitch = pp.y_pitch;
  ip.c_pitch = pp.c_pitch;
  ip.mbs = job.mbs;
  ip.motions = job.motions;
  ip.slices = job.slices;
  ip.cur = pp.surfaces[pp.cur_slot];
  return ip;
}
template <bool CHAIN>
__device__ __forceinline__ void inter_mb(const JobDesc* __restrict__ jobs, const JobDesc& job, const InterPicture& ip, const int mbi,
                                         const int lane, InterWave& L) {
  const PictureParams& pp = job.pp;
  const int mb_width = ip.mb_width, mb_height = ip.mb_height, y_pitch = ip.y_pitch, c_pitch = ip.c_pitch;
  const MbInfo* const mbs = ip.mbs;
  const MbMotion* const motions = ip.motions;
  const SliceParams* const slices = ip.slices;
  const Surface cur = ip.cur;
  const MbMotion* __restrict__ mo = motions + mbi;
  // the lane's motion data: luma block (row y, 4x4 column lane&3)
  const int y = lane >> 2, bcol = lane & 3;
  const int blk = (y >> 2) * 4 + bcol, b8 = (y >> 3) * 2 + (bcol >> 1);
  const int cplane = lane >> 5, ct = lane & 31, cy = ct >> 2, cbcol = ct & 3;
  const uint32_t mvw0 = gld32(reinterpret_cast<const uint32_t*>(mo->mv[0]) + blk);
  const uint32_t mvw1 = gld32(reinterpret_cast<const uint32_t*>(mo->mv[1]) + blk);
  const uint32_t slots0 = gld32(mo->ref_slot[0]);
  const uint32_t slots1 = gld32(mo->ref_slot[1]);
  const uint32_t idx0 = gld32(mo->ref_idx[0]);
  const uint32_t idx1 = gld32(mo->ref_idx[1]);
  const MbInfo mb = gload(&mbs[mbi]);
  if (mb.kind != MBK_INTER) return;
  const SliceParams& sp = slices[mb.slice];
  const int weight_mode = gld8(&sp.weight_mode);
  const int mbx = mbi % mb_width, mby = mbi / mb_width;
  const bool has_res = mb.coded_mask != 0;
  const int W = mb_width * 16, H = mb_height * 16;
  auto byte_of = [](uint32_t w, int k) { return (int)(int8_t)((w >> (8 * k)) & 0xff); };
  // surface numbers are bytes 0 .. kMaxSurfaces - 1 (0xff: the list is not used - and then the byte is not looked at)
  auto slot_of = [](uint32_t w, int k) { return (int)((w >> (8 * k)) & 0xff); };

  // ---- window path: is the macroblock's motion uniform per list?  (wave-uniform decisions: every lane holds one
  // of the sixteen vectors, all four reference bytes.)  A window that lies inside the reference picture is fetched
  // with wide loads; one that leaves it - every macroblock along the picture's border, vectors that point outside -
  // with byte loads from clamped coordinates (8.4.2.2.1's edge extension) into the same registers, so both kinds
  // continue on the same path.
  bool use_win[2], win_inside[2];
  int wmvx[2], wmvy[2], wx0[2], wy0[2];
  const uint8_t* wsrc[2][3];
  const uint8_t* wplane[2][3];
  bool window = true;
#pragma unroll
  for (int l = 0; l < 2; ++l) {
    const uint32_t mvw = l ? mvw1 : mvw0, idx = l ? idx1 : idx0, slots = l ? slots1 : slots0;
    const uint32_t first = __builtin_amdgcn_readfirstlane(mvw);
    const int r = byte_of(idx, 0);
    use_win[l] = r >= 0;
    const bool uniform = __all(mvw == first) && idx == (idx & 0xff) * 0x01010101u && slots == (slots & 0xff) * 0x01010101u;
    wmvx[l] = (int16_t)(first & 0xffff);
    wmvy[l] = (int16_t)(first >> 16);
    const int x0 = mbx * 16 + (wmvx[l] >> 2) - 2, y0 = mby * 16 + (wmvy[l] >> 2) - 2;
    const int cx0 = mbx * 8 + (wmvx[l] >> 3), cy0 = mby * 8 + (wmvy[l] >> 3);
    const bool inside = x0 >= 0 && x0 + 21 <= W && y0 >= 0 && y0 + 21 <= H && cx0 >= 0 && cx0 + 9 <= W / 2 && cy0 >= 0 &&
                        cy0 + 9 <= H / 2;
    if (idx != 0xffffffffu) window = window && uniform;  // a list nobody uses constrains nothing
    else use_win[l] = false;
    win_inside[l] = inside;
    wx0[l] = x0;
    wy0[l] = y0;
    if (use_win[l]) {
      const Surface& sf = pp.surfaces[slot_of(slots, 0)];
      wplane[l][0] = sf.y;
      wplane[l][1] = sf.u;
      wplane[l][2] = sf.v;
      wsrc[l][0] = sf.y + (ptrdiff_t)y0 * y_pitch + x0;
      wsrc[l][1] = sf.u + (ptrdiff_t)cy0 * c_pitch + cx0;
      wsrc[l][2] = sf.v + (ptrdiff_t)cy0 * c_pitch + cx0;
    }
  }
  window = window && (use_win[0] || use_win[1]);
  // chain: references that this launch reconstructs (wave-uniform: the record's words are the same in every lane)
  bool fresh[2] = {false, false};
  if constexpr (CHAIN) {
    if (window) {
#pragma unroll
      for (int l = 0; l < 2; ++l) {
        if (!use_win[l]) continue;
        const int rj = job.slot_job[__builtin_amdgcn_readfirstlane(slot_of(l ? slots1 : slots0, 0))];
        if (rj >= 0) {
          fresh[l] = true;
          chain_wait_reference(jobs[rj], job, wy0[l], wy0[l] + 20, wx0[l] + 20);
        }
      }
    } else {
      // partitions with vectors of their own: how far down / right any lane reads (six-tap margins included), then
      // every reference of the record that this launch reconstructs has to have got that far
      int top = 1 << 20, bottom = 0, right = 0;
#pragma unroll
      for (int l = 0; l < 2; ++l) {
        const uint32_t mvw = l ? mvw1 : mvw0;
        if (byte_of(l ? idx1 : idx0, b8) < 0) continue;
        const int yy = mby * 16 + y + ((int)(int16_t)(mvw >> 16) >> 2);
        top = min(top, yy - 2);
        bottom = max(bottom, yy + 3);
        right = max(right, mbx * 16 + bcol * 4 + 3 + ((int)(int16_t)(mvw & 0xffff) >> 2) + 3);
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        top = min(top, __shfl_xor(top, o));
        bottom = max(bottom, __shfl_xor(bottom, o));
        right = max(right, __shfl_xor(right, o));
      }
      top = __builtin_amdgcn_readfirstlane(top);
      bottom = __builtin_amdgcn_readfirstlane(bottom);
      right = __builtin_amdgcn_readfirstlane(right);
#pragma unroll
      for (int l = 0; l < 2; ++l) {
        const uint32_t idx = __builtin_amdgcn_readfirstlane(l ? idx1 : idx0), slots = __builtin_amdgcn_readfirstlane(l ? slots1 : slots0);
        int seen = -1;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          if (byte_of(idx, k) < 0) continue;
          const int sl = slot_of(slots, k);
          if (sl == seen) continue;
          seen = sl;
          const int rj = job.slot_job[sl];
          if (rj >= 0) chain_wait_reference(jobs[rj], job, top, bottom, right);
        }
      }
    }
  }
  uint8_t* const out_y = cur.y + (size_t)(mby * 16) * y_pitch + mbx * 16;
  uint8_t* const out_c = (cplane ? cur.v : cur.u) + (size_t)(mby * 8) * c_pitch + mbx * 8;
  const uint32_t out_yoff = (uint32_t)(y * y_pitch + bcol * 4), out_coff = (uint32_t)(cy * c_pitch + cbcol * 2);

  if (window) {
    // issue every reference load of the macroblock now; they land while the residual is computed
    const int lr = (lane * 43) >> 7, lc = lane - 3 * lr;                       // luma: row 0..20, eight-byte column 0..2 (lane < 63)
    const int cpl = lane >= 27, cq = lane - 27 * cpl, cr = (cq * 43) >> 7, cd = cq - 3 * cr;  // chroma: plane, row, dword (lane < 54)
    const uint32_t loff = (uint32_t)(lr * y_pitch + 8 * lc), coff = (uint32_t)(cr * c_pitch + 4 * cd);
    u32x2 wy[2] = {0, 0};
    uint32_t wc[2] = {0, 0};
#pragma unroll
    for (int l = 0; l < 2; ++l) {
      if (!use_win[l]) continue;
      if (CHAIN && fresh[l]) {  // (the picture is being written by this launch: nothing of it may come from a cache)
        if (win_inside[l]) {
          if (lane < 63) wy[l] = agent_ld64u(wsrc[l][0] + loff);
          if (lane < 54) wc[l] = agent_ld32((cpl ? wsrc[l][2] : wsrc[l][1]) + coff);
        } else {
          if (lane < 63) {
            const uint8_t* row = wplane[l][0] + (size_t)iclip(0, H - 1, wy0[l] + lr) * y_pitch;
            uint32_t lo = 0, hi = 0;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
              lo |= (uint32_t)agent_ld8(row + iclip(0, W - 1, wx0[l] + 8 * lc + b)) << (8 * b);
              hi |= (uint32_t)agent_ld8(row + iclip(0, W - 1, wx0[l] + 8 * lc + 4 + b)) << (8 * b);
            }
            wy[l].x = lo;
            wy[l].y = hi;
          }
          if (lane < 54) {
            const int cx0 = mbx * 8 + (wmvx[l] >> 3), cy0 = mby * 8 + (wmvy[l] >> 3);
            const uint8_t* row = (cpl ? wplane[l][2] : wplane[l][1]) + (size_t)iclip(0, H / 2 - 1, cy0 + cr) * c_pitch;
            uint32_t v = 0;
#pragma unroll
            for (int b = 0; b < 4; ++b) v |= (uint32_t)agent_ld8(row + iclip(0, W / 2 - 1, cx0 + 4 * cd + b)) << (8 * b);
            wc[l] = v;
          }
        }
      } else if (win_inside[l]) {
        if (lane < 63) wy[l] = gld64u(wsrc[l][0] + loff);
        if (lane < 54) wc[l] = ld32u((cpl ? wsrc[l][2] : wsrc[l][1]) + coff);
      } else {  // the same bytes, one at a time, from clamped coordinates
        if (lane < 63) {
          const uint8_t* row = wplane[l][0] + (size_t)iclip(0, H - 1, wy0[l] + lr) * y_pitch;
          uint32_t lo = 0, hi = 0;
#pragma unroll
          for (int b = 0; b < 4; ++b) {
            lo |= (uint32_t)gld8(row + iclip(0, W - 1, wx0[l] + 8 * lc + b)) << (8 * b);
            hi |= (uint32_t)gld8(row + iclip(0, W - 1, wx0[l] + 8 * lc + 4 + b)) << (8 * b);
          }
          wy[l].x = lo;
          wy[l].y = hi;
        }
        if (lane < 54) {
          const int cx0 = mbx * 8 + (wmvx[l] >> 3), cy0 = mby * 8 + (wmvy[l] >> 3);
          const uint8_t* row = (cpl ? wplane[l][2] : wplane[l][1]) + (size_t)iclip(0, H / 2 - 1, cy0 + cr) * c_pitch;
          uint32_t v = 0;
#pragma unroll
          for (int b = 0; b < 4; ++b) v |= (uint32_t)gld8(row + iclip(0, W / 2 - 1, cx0 + 4 * cd + b)) << (8 * b);
          wc[l] = v;
        }
      }
    }
    if (has_res) wave_residual(mb, job.coefs, pp, lane, L.work, L.dcs, L.res_y, L.res_c, L.lv);  // ends with wave_sync
#pragma unroll
    for (int l = 0; l < 2; ++l) {
      if (!use_win[l]) continue;
      if (lane < 63) *reinterpret_cast<u32x2*>(&L.win[l][lr * kWinLumaDw + 2 * lc]) = wy[l];
      if (lane < 54) L.win[l][kWinLumaWords + lane] = wc[l];
    }
    wave_sync();

    const bool bi = use_win[0] && use_win[1];
    const int r0 = byte_of(idx0, 0), r1 = byte_of(idx1, 0);
    {  // luma
      uint32_t p[2] = {0, 0};
#pragma unroll
      for (int l = 0; l < 2; ++l)
        if (use_win[l]) p[l] = luma_win4(L.win[l] + y * kWinLumaDw + bcol, wmvx[l] & 3, wmvy[l] & 3);
      uint32_t pred;
      if (weight_mode == 0) {
        pred = bi ? avg4(p[0], p[1]) : (use_win[0] ? p[0] : p[1]);
      } else {
        int v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
          v[i] = combine_pred(sp, 0, use_win[0], use_win[1], (p[0] >> (8 * i)) & 0xff, (p[1] >> (8 * i)) & 0xff, r0, r1);
        pred = pack4(v[0], v[1], v[2], v[3]);
      }
      if (has_res) {
        const uint32_t* rp = reinterpret_cast<const uint32_t*>(L.res_y + y * 16 + bcol * 4);
        pred = add_res4(pred, rp[0], rp[1]);
      }
      sst32<CHAIN>(out_y + out_yoff, pred);
    }
    {  // chroma: lanes 0..31 Cb, 32..63 Cr
      uint32_t p[2] = {0, 0};
#pragma unroll
      for (int l = 0; l < 2; ++l) {
        if (!use_win[l]) continue;
        const int fx = wmvx[l] & 7, fy = wmvy[l] & 7;
        const uint32_t wts = (uint32_t)((8 - fx) * (8 - fy)) | ((uint32_t)(fx * (8 - fy)) << 8) | ((uint32_t)((8 - fx) * fy) << 16) |
                             ((uint32_t)(fx * fy) << 24);
        p[l] = chroma_win2(L.win[l] + kWinLumaWords + cplane * kWinChromaRows * kWinChromaDw + cy * kWinChromaDw + (cbcol >> 1),
                           cbcol & 1, wts);
      }
      uint32_t pred;
      if (weight_mode == 0) {
        pred = bi ? avg4(p[0], p[1]) : (use_win[0] ? p[0] : p[1]);
      } else {
        int v[2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
          v[i] = combine_pred(sp, 1 + cplane, use_win[0], use_win[1], (p[0] >> (8 * i)) & 0xff, (p[1] >> (8 * i)) & 0xff, r0, r1);
        pred = (uint32_t)v[0] | ((uint32_t)v[1] << 8);
      }
      if (has_res) pred = add_res2(pred, *reinterpret_cast<const uint32_t*>(L.res_c + cplane * 64 + cy * 8 + cbcol * 2));
      sst16<CHAIN>(out_c + out_coff, (uint16_t)pred);
    }
    if constexpr (!CHAIN) return;
  } else {
  // ---- per-lane register path: partitions with their own vectors, windows that leave the picture
  const int cblk = (cy >> 1) * 4 + cbcol, cb8 = (cy >> 2) * 2 + (cbcol >> 1);
  const uint32_t cmvw0 = gld32(reinterpret_cast<const uint32_t*>(mo->mv[0]) + cblk);
  const uint32_t cmvw1 = gld32(reinterpret_cast<const uint32_t*>(mo->mv[1]) + cblk);
  if (has_res) wave_residual(mb, job.coefs, pp, lane, L.work, L.dcs, L.res_y, L.res_c, L.lv);  // ends with wave_sync
  const bool plain = idx1 == 0xffffffffu && weight_mode == 0 && !((idx0 & 0x80808080u));
  {  // luma
    const int r0 = byte_of(idx0, b8), r1 = byte_of(idx1, b8);
    const int px = mbx * 16 + bcol * 4, py = mby * 16 + y;
    int s0[4] = {0, 0, 0, 0}, s1[4] = {0, 0, 0, 0};
    if (r0 >= 0)
      luma_pred4<CHAIN>(pp.surfaces[slot_of(slots0, b8)].y, y_pitch, W, H, px, py, (int16_t)(mvw0 & 0xffff), (int16_t)(mvw0 >> 16), s0);
    if (r1 >= 0)
      luma_pred4<CHAIN>(pp.surfaces[slot_of(slots1, b8)].y, y_pitch, W, H, px, py, (int16_t)(mvw1 & 0xffff), (int16_t)(mvw1 >> 16), s1);
    uint32_t packed = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int pred = plain ? s0[i] : combine_pred(sp, 0, r0 >= 0, r1 >= 0, s0[i], s1[i], r0, r1);
      const int v = has_res ? clip8(pred + L.res_y[y * 16 + bcol * 4 + i]) : pred;
      packed |= (uint32_t)v << (8 * i);
    }
    sst32<CHAIN>(out_y + out_yoff, packed);
  }
  {  // chroma: lanes 0..31 Cb, 32..63 Cr
    const int r0 = byte_of(idx0, cb8), r1 = byte_of(idx1, cb8);
    const int px = mbx * 8 + cbcol * 2, py = mby * 8 + cy;
    int s0[2] = {0, 0}, s1[2] = {0, 0};
    if (r0 >= 0) {
      const Surface& sf = pp.surfaces[slot_of(slots0, cb8)];
      chroma_pred2<CHAIN>(cplane ? sf.v : sf.u, c_pitch, W / 2, H / 2, px, py, (int16_t)(cmvw0 & 0xffff), (int16_t)(cmvw0 >> 16), s0);
    }
    if (r1 >= 0) {
      const Surface& sf = pp.surfaces[slot_of(slots1, cb8)];
      chroma_pred2<CHAIN>(cplane ? sf.v : sf.u, c_pitch, W / 2, H / 2, px, py, (int16_t)(cmvw1 & 0xffff), (int16_t)(cmvw1 >> 16), s1);
    }
    uint32_t packed = 0;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int pred = plain ? s0[i] : combine_pred(sp, 1 + cplane, r0 >= 0, r1 >= 0, s0[i], s1[i], r0, r1);
      const int v = has_res ? clip8(pred + L.res_c[cplane * 64 + cy * 8 + cbcol * 2 + i]) : pred;
      packed |= (uint32_t)v << (8 * i);
    }
    sst16<CHAIN>(out_c + out_coff, (uint16_t)packed);
  }
  }
}

__global__ __launch_bounds__(kInterWaves * 64, 7) void k_inter(const JobDesc* __restrict__ jobs) {
  const JobDesc& job = jobs[blockIdx.y];
  const int do_inter = job.do_inter;
  const InterPicture ip = inter_picture(job);
  if (!do_inter) return;
  const int nmb = ip.mb_width * ip.mb_height;
  const int ngroups = (nmb + kInterWaves - 1) / kInterWaves;
  const int band = (ngroups + 7) / 8;
  const int group = (blockIdx.x % 8) * band + blockIdx.x / 8;
  if (blockIdx.x / 8 >= band || group >= ngroups) return;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int mbi = group * kInterWaves + wave;
  if (mbi >= nmb) return;
  __shared__ InterWave lds[kInterWaves];
  inter_mb<false>(jobs, job, ip, mbi, lane, lds[wave]);
}

// ===========================================================================
// Row pipelines (k_mv, k_intra, k_deblock).
//
// Motion-vector prediction, intra prediction and the deblocking filter have
// true raster-order dependencies: a macroblock needs its left, top-left, top
// and top-right neighbours finished.  Rows can run concurrently as long as
// each stays two macroblocks behind the row above, so a picture takes about
// mb_width + 2 * mb_height steps if every row has a wave of its own.  A step
// is bound by the dependent chain inside one macroblock (measured
// 1.3 - 2.7 us for a lone wave), and several waves on one SIMD only stretch
// it, so a picture is cut into bands of kBandWaves = 4 rows, one workgroup of
// four waves (one per SIMD) per band:
//   * inside a band rows hand over through LDS: counters (row_wait) and, where
//     the neighbours are small, the data itself (line buffers);
//   * between bands the hand-off goes through global memory with agent-scope
//     (sc1) accesses, which bypass the per-XCD L2 - either words that carry
//     their own validity tag (k_mv, k_intra's LDS-neighbour path) or the
//     picture's own samples plus a counter raised once they have landed
//     (k_deblock, k_intra's surface path);
//   * bands are handed out by an atomic ticket, band-major over the pictures
//     of a batch, so a workgroup only ever waits for one that started before
//     it: no deadlock whatever the dispatch order, and every poll is bounded
//     anyway (a logic error would show as wrong samples in the parity tests,
//     not as a hung GPU).
// ===========================================================================
namespace {

// The counters live in LDS and must be addressed as such: through a generic
// volatile pointer they become flat_load ... sc0 sc1 followed by s_waitcnt
// vmcnt(0), i.e. every poll would drain the wave's global traffic.
typedef volatile __attribute__((address_space(3))) int lds_counter;

// progress[r + 1] = macroblocks of row r that are complete; progress[0] is
// the (always complete) row above the picture
// The producer completed its global stores (s_waitcnt vmcnt(0)) before it
// raised the counter, all waves of the workgroup share the CU's L1, and this
// wave issues its dependent loads after the poll returns: no acquire fence
// (which would also drain this wave's own outstanding stores and prefetches).
__device__ __forceinline__ void row_wait(lds_counter* progress, int row, int need) {
  while (progress[row] < need) __builtin_amdgcn_s_sleep(1);
  __asm__ volatile("" ::: "memory");
}

// ---- intra prediction helpers working on an LDS tile -----------------------
// tile: 17 rows x 28 columns (stride 28): row 0 / col 0 hold the neighbours
// (tile[0][0] = top-left, tile[0][1..16] = top, tile[0][17..24] = top-right of
// the MB, tile[1..16][0] = left), interior = current macroblock.
constexpr int TS = 28;

// Directional intra prediction as data (8.3.1.2.x, 8.3.2.2.x): every sample of
// every mode is a 1-, 2- or 3-tap combination of entries of the block's edge
// array E (left column bottom-up, corner, top row incl. top-right), so the
// per-sample code has no branches: entry = i0 | i1 << 4 | i2 << 8 | taps << 12
// (4x4, E[13]) resp. i0 | i1 << 5 | i2 << 10 | taps << 15 (8x8, E[25]).
__constant__ uint16_t c_p4tab[9][16] = {
    {0x1005, 0x1006, 0x1007, 0x1008, 0x1005, 0x1006, 0x1007, 0x1008, 0x1005, 0x1006, 0x1007, 0x1008, 0x1005, 0x1006, 0x1007, 0x1008},
    {0x1003, 0x1003, 0x1003, 0x1003, 0x1002, 0x1002, 0x1002, 0x1002, 0x1001, 0x1001, 0x1001, 0x1001, 0x1000, 0x1000, 0x1000, 0x1000},
    {0x0000, 0x0000, 0x0000, 0x0000, 0x0000, 0x0000, 0x0000, 0x0000, 0x0000, 0x0000, 0x0000, 0x0000, 0x0000, 0x0000, 0x0000, 0x0000},
    {0x3765, 0x3876, 0x3987, 0x3a98, 0x3876, 0x3987, 0x3a98, 0x3ba9, 0x3987, 0x3a98, 0x3ba9, 0x3cba, 0x3a98, 0x3ba9, 0x3cba, 0x3ccb},
    {0x3345, 0x3654, 0x3765, 0x3876, 0x3234, 0x3345, 0x3654, 0x3765, 0x3123, 0x3234, 0x3345, 0x3654, 0x3012, 0x3123, 0x3234, 0x3345},
    {0x2054, 0x2065, 0x2076, 0x2087, 0x3543, 0x3654, 0x3765, 0x3876, 0x3432, 0x2054, 0x2065, 0x2076, 0x3321, 0x3543, 0x3654, 0x3765},
    {0x2034, 0x3543, 0x3456, 0x3567, 0x2023, 0x3234, 0x2034, 0x3543, 0x2012, 0x3123, 0x2023, 0x3234, 0x2001, 0x3012, 0x2012, 0x3123},
    {0x2065, 0x2076, 0x2087, 0x2098, 0x3765, 0x3876, 0x3987, 0x3a98, 0x2076, 0x2087, 0x2098, 0x20a9, 0x3876, 0x3987, 0x3a98, 0x3ba9},
    {0x2023, 0x3123, 0x2012, 0x3012, 0x2012, 0x3012, 0x2001, 0x3001, 0x2001, 0x3001, 0x1000, 0x1000, 0x1000, 0x1000, 0x1000, 0x1000},
};
__constant__ uint32_t c_p8tab[9][64] = {
    {0x08009, 0x0800a, 0x0800b, 0x0800c, 0x0800d, 0x0800e, 0x0800f, 0x08010, 0x08009, 0x0800a, 0x0800b, 0x0800c, 0x0800d, 0x0800e, 0x0800f, 0x08010, 0x08009, 0x0800a, 0x0800b, 0x0800c, 0x0800d, 0x0800e, 0x0800f, 0x08010, 0x08009, 0x0800a, 0x0800b, 0x0800c, 0x0800d, 0x0800e, 0x0800f, 0x08010, 0x08009, 0x0800a, 0x0800b, 0x0800c, 0x0800d, 0x0800e, 0x0800f, 0x08010, 0x08009, 0x0800a, 0x0800b, 0x0800c, 0x0800d, 0x0800e, 0x0800f, 0x08010, 0x08009, 0x0800a, 0x0800b, 0x0800c, 0x0800d, 0x0800e, 0x0800f, 0x08010, 0x08009, 0x0800a, 0x0800b, 0x0800c, 0x0800d, 0x0800e, 0x0800f, 0x08010},
    {0x08007, 0x08007, 0x08007, 0x08007, 0x08007, 0x08007, 0x08007, 0x08007, 0x08006, 0x08006, 0x08006, 0x08006, 0x08006, 0x08006, 0x08006, 0x08006, 0x08005, 0x08005, 0x08005, 0x08005, 0x08005, 0x08005, 0x08005, 0x08005, 0x08004, 0x08004, 0x08004, 0x08004, 0x08004, 0x08004, 0x08004, 0x08004, 0x08003, 0x08003, 0x08003, 0x08003, 0x08003, 0x08003, 0x08003, 0x08003, 0x08002, 0x08002, 0x08002, 0x08002, 0x08002, 0x08002, 0x08002, 0x08002, 0x08001, 0x08001, 0x08001, 0x08001, 0x08001, 0x08001, 0x08001, 0x08001, 0x08000, 0x08000, 0x08000, 0x08000, 0x08000, 0x08000, 0x08000, 0x08000},
    {0x00000, 0x00000, 0x00000, 0x00000, 0x00000, 0x00000, 0x00000, 0x00000, 0x00000, 0x00000, 0x00000, 0x00000, 0x00000, 0x00000, 0x00000, 0x00000, 0x00000, 0x00000, 0x00000, 0x00000, 0x00000, 0x00000, 0x00000, 0x00000, 0x00000, 0x00000, 0x00000, 0x00000, 0x00000, 0x00000, 0x00000, 0x00000, 0x00000, 0x00000, 0x00000, 0x00000, 0x00000, 0x00000, 0x00000, 0x00000, 0x00000, 0x00000, 0x00000, 0x00000, 0x00000, 0x00000, 0x00000, 0x00000, 0x00000, 0x00000, 0x00000, 0x00000, 0x00000, 0x00000, 0x00000, 0x00000, 0x00000, 0x00000, 0x00000, 0x00000, 0x00000, 0x00000, 0x00000, 0x00000},
    {0x1ad49, 0x1b16a, 0x1b58b, 0x1b9ac, 0x1bdcd, 0x1c1ee, 0x1c60f, 0x1ca30, 0x1b16a, 0x1b58b, 0x1b9ac, 0x1bdcd, 0x1c1ee, 0x1c60f, 0x1ca30, 0x1ce51, 0x1b58b, 0x1b9ac, 0x1bdcd, 0x1c1ee, 0x1c60f, 0x1ca30, 0x1ce51, 0x1d272, 0x1b9ac, 0x1bdcd, 0x1c1ee, 0x1c60f, 0x1ca30, 0x1ce51, 0x1d272, 0x1d693, 0x1bdcd, 0x1c1ee, 0x1c60f, 0x1ca30, 0x1ce51, 0x1d272, 0x1d693, 0x1dab4, 0x1c1ee, 0x1c60f, 0x1ca30, 0x1ce51, 0x1d272, 0x1d693, 0x1dab4, 0x1ded5, 0x1c60f, 0x1ca30, 0x1ce51, 0x1d272, 0x1d693, 0x1dab4, 0x1ded5, 0x1e2f6, 0x1ca30, 0x1ce51, 0x1d272, 0x1d693, 0x1dab4, 0x1ded5, 0x1e2f6, 0x1e317},
    {0x19d09, 0x1a928, 0x1ad49, 0x1b16a, 0x1b58b, 0x1b9ac, 0x1bdcd, 0x1c1ee, 0x198e8, 0x19d09, 0x1a928, 0x1ad49, 0x1b16a, 0x1b58b, 0x1b9ac, 0x1bdcd, 0x194c7, 0x198e8, 0x19d09, 0x1a928, 0x1ad49, 0x1b16a, 0x1b58b, 0x1b9ac, 0x190a6, 0x194c7, 0x198e8, 0x19d09, 0x1a928, 0x1ad49, 0x1b16a, 0x1b58b, 0x18c85, 0x190a6, 0x194c7, 0x198e8, 0x19d09, 0x1a928, 0x1ad49, 0x1b16a, 0x18864, 0x18c85, 0x190a6, 0x194c7, 0x198e8, 0x19d09, 0x1a928, 0x1ad49, 0x18443, 0x18864, 0x18c85, 0x190a6, 0x194c7, 0x198e8, 0x19d09, 0x1a928, 0x18022, 0x18443, 0x18864, 0x18c85, 0x190a6, 0x194c7, 0x198e8, 0x19d09},
    {0x10128, 0x10149, 0x1016a, 0x1018b, 0x101ac, 0x101cd, 0x101ee, 0x1020f, 0x1a507, 0x1a928, 0x1ad49, 0x1b16a, 0x1b58b, 0x1b9ac, 0x1bdcd, 0x1c1ee, 0x1a0e6, 0x10128, 0x10149, 0x1016a, 0x1018b, 0x101ac, 0x101cd, 0x101ee, 0x19cc5, 0x1a507, 0x1a928, 0x1ad49, 0x1b16a, 0x1b58b, 0x1b9ac, 0x1bdcd, 0x198a4, 0x1a0e6, 0x10128, 0x10149, 0x1016a, 0x1018b, 0x101ac, 0x101cd, 0x19483, 0x19cc5, 0x1a507, 0x1a928, 0x1ad49, 0x1b16a, 0x1b58b, 0x1b9ac, 0x19062, 0x198a4, 0x1a0e6, 0x10128, 0x10149, 0x1016a, 0x1018b, 0x101ac, 0x18c41, 0x19483, 0x19cc5, 0x1a507, 0x1a928, 0x1ad49, 0x1b16a, 0x1b58b},
    {0x100e8, 0x1a507, 0x1a12a, 0x1a54b, 0x1a96c, 0x1ad8d, 0x1b1ae, 0x1b5cf, 0x100c7, 0x198e8, 0x100e8, 0x1a507, 0x1a12a, 0x1a54b, 0x1a96c, 0x1ad8d, 0x100a6, 0x194c7, 0x100c7, 0x198e8, 0x100e8, 0x1a507, 0x1a12a, 0x1a54b, 0x10085, 0x190a6, 0x100a6, 0x194c7, 0x100c7, 0x198e8, 0x100e8, 0x1a507, 0x10064, 0x18c85, 0x10085, 0x190a6, 0x100a6, 0x194c7, 0x100c7, 0x198e8, 0x10043, 0x18864, 0x10064, 0x18c85, 0x10085, 0x190a6, 0x100a6, 0x194c7, 0x10022, 0x18443, 0x10043, 0x18864, 0x10064, 0x18c85, 0x10085, 0x190a6, 0x10001, 0x18022, 0x10022, 0x18443, 0x10043, 0x18864, 0x10064, 0x18c85},
    {0x10149, 0x1016a, 0x1018b, 0x101ac, 0x101cd, 0x101ee, 0x1020f, 0x10230, 0x1ad49, 0x1b16a, 0x1b58b, 0x1b9ac, 0x1bdcd, 0x1c1ee, 0x1c60f, 0x1ca30, 0x1016a, 0x1018b, 0x101ac, 0x101cd, 0x101ee, 0x1020f, 0x10230, 0x10251, 0x1b16a, 0x1b58b, 0x1b9ac, 0x1bdcd, 0x1c1ee, 0x1c60f, 0x1ca30, 0x1ce51, 0x1018b, 0x101ac, 0x101cd, 0x101ee, 0x1020f, 0x10230, 0x10251, 0x10272, 0x1b58b, 0x1b9ac, 0x1bdcd, 0x1c1ee, 0x1c60f, 0x1ca30, 0x1ce51, 0x1d272, 0x101ac, 0x101cd, 0x101ee, 0x1020f, 0x10230, 0x10251, 0x10272, 0x10293, 0x1b9ac, 0x1bdcd, 0x1c1ee, 0x1c60f, 0x1ca30, 0x1ce51, 0x1d272, 0x1d693},
    {0x100c7, 0x194c7, 0x100a6, 0x190a6, 0x10085, 0x18c85, 0x10064, 0x18864, 0x100a6, 0x190a6, 0x10085, 0x18c85, 0x10064, 0x18864, 0x10043, 0x18443, 0x10085, 0x18c85, 0x10064, 0x18864, 0x10043, 0x18443, 0x10022, 0x18022, 0x10064, 0x18864, 0x10043, 0x18443, 0x10022, 0x18022, 0x10001, 0x18001, 0x10043, 0x18443, 0x10022, 0x18022, 0x10001, 0x18001, 0x08000, 0x08000, 0x10022, 0x18022, 0x10001, 0x18001, 0x08000, 0x08000, 0x08000, 0x08000, 0x10001, 0x18001, 0x08000, 0x08000, 0x08000, 0x08000, 0x08000, 0x08000, 0x08000, 0x08000, 0x08000, 0x08000, 0x08000, 0x08000, 0x08000, 0x08000},
};
// Order in which the sixteen 4x4 blocks of an Intra4x4 macroblock are predicted:
// a block needs its left, top-left, top and top-right neighbours, so the
// blocks (x, y) with equal x + 2y are independent - 10 steps instead of 16,
// two blocks (32 lanes) in the six middle ones.

// luma4x4BlkIdx of a raster 4x4 block ({0,1,4,5,2,3,6,7,8,9,12,13,10,11,14,15} as nibbles)
__device__ __forceinline__ int zz_of_raster(int r) { return (int)((0xFEBADC9876325410ull >> (4 * r)) & 15); }
__device__ __forceinline__ int tap_sample(const uint8_t* e, unsigned ent, int shift, int mask) {
  const int a = e[ent & mask], b = e[(ent >> shift) & mask], c = e[(ent >> (2 * shift)) & mask];
  const int taps = ent >> (3 * shift);
  return taps == 1 ? a : taps == 2 ? (a + b + 1) >> 1 : (a + 2 * b + c + 2) >> 2;
}

struct IntraWaveLds {
  int16_t res_y[256];
  int16_t res_c[128];
  uint8_t tile[17 * TS];
  uint8_t ctile[2][9 * 12];
  uint8_t e4[2][16];  // Intra4x4 edge arrays of the (up to) two blocks of a step
  uint8_t e8[32];     // Intra8x8 filtered edge array
};

}  // namespace

// ===========================================================================
// k_mv: motion vectors of the P macroblocks whose derivation the host left to
// the GPU (MbInfo::mv_flags & MVF_DEFERRED): MbMotion::mv[0] arrives holding
// the parsed differences and leaves holding the vectors (8.4.1.1 P_Skip
// inference, 8.4.1.3 median / directional prediction).  Prediction looks at
// the left, top, top-right and top-left neighbours, i.e. it is the same
// raster-order chain as intra prediction: a wave per macroblock row, four
// rows per workgroup, neighbours handed on through LDS (the previous
// macroblock's right column, a line buffer with every macroblock's bottom
// row) and, between workgroups, through tagged words in global memory.
// Pictures wider than kLineMbs macroblocks are
// derived on the host instead (h264_parser: PicBuffers::device_mv).
// ===========================================================================
namespace {
struct MvEdge {        // what a neighbour below / to the right needs of a macroblock: five dwords
  uint32_t mv[4];      // packed (x, y) of four 4x4 blocks; 0 where the block has no reference (intra)
  uint32_t refs;       // reference index of the two 8x8 quadrants they lie in, one byte each (-1: intra / unused)
};
struct MvCell {        // one 4x4 block as a neighbour
  uint32_t mv;         // packed (x, y); 0 unless ref >= 0
  int ref;             // -2 not available (outside the slice / picture, or later in decoding order), -1 intra or list unused, >= 0
};
constexpr int kMvChunk = 4;  // macroblocks whose records travel together (prefetch distance)
// A picture's rows are spread over workgroups ("bands") of kMvWaves rows, one
// wave per row.  Inside a band rows hand on through LDS; the last row of a
// band exports its line to MvCtl in global memory and the band below imports it.
// Measured on the skeleton
// (tools/experiments/rowpipe_bands.hip): a picture takes ~mb_width + 2 *
// mb_height steps instead of ~mb_width * mb_height / 16, and a step is as fast
// as a lone wave's because at most one row wave sits on each SIMD.
//
// Hand-off words carry their own validity: every 32-bit datum travels as a
// 64-bit store {datum, JobDesc::mv_epoch}, which is single-copy atomic, so the
// producer neither waits for its stores nor maintains a counter, and the
// consumer needs one round trip, not two (the scheme of low-latency collective
// protocols).  Nothing is cleared between pictures: a slot left over from an
// earlier picture carries an earlier epoch.
//
// What a step costs (round 4, s_memtime stamps inside the kernel: tools/experiments/exp_stamps.py): a lone wave issues
// an instruction every four to five clocks WHATEVER unit executes it, so the step is the number of instructions on
// the macroblock's path and nothing else - moving the prediction to the scalar unit (round 2) had bought nothing,
// its branches cost.  The path is therefore written without branches on values: neighbours as {vector, reference}
// cells with the availability folded into the reference (-2), vectors packed (x, y) and compared / selected /
// medianed as such (v_pk_min_i16 / v_pk_max_i16), every lane computing the same prediction; partitioned macroblocks
// gather their surroundings into a 6 x 5 grid of cells in LDS once and index it.  No store follows a load in the
// row waves' queue conditionally, so the wait for the prefetched records is a count, not a drain.
constexpr int kMvWaves = kBandWaves;
constexpr int kMvGridW = 6;  // cells of rows -1 .. 3, columns -1 .. 4 around / inside the macroblock
struct MvWaveLds {
  uint32_t x[64];                  // the chunk's window of the extra-difference stream
  alignas(8) MvCell grid[32];      // [(y + 1) * kMvGridW + x + 1]
  uint32_t left[8];                // MvEdge of the previous macroblock's right column (blocks 3, 7, 11, 15; quadrants 1, 3)
};
__device__ __forceinline__ uint32_t pk_min_i16(uint32_t a, uint32_t b) {
  uint32_t r;
  asm("v_pk_min_i16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ uint32_t pk_max_i16(uint32_t a, uint32_t b) {
  uint32_t r;
  asm("v_pk_max_i16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ uint32_t pk_add_u16(uint32_t a, uint32_t b) {
  uint32_t r;
  asm("v_pk_add_u16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
// 8.4.1.3: the prediction for a partition with reference `ref` from its neighbours A (left), B (top), C (top-right, or
// top-left where that one is not available); dir: 0 median, 1 16x8 top, 2 16x8 bottom, 3 8x16 left, 4 8x16 right.
__device__ __forceinline__ uint32_t mv_predict(const MvCell A, const MvCell B, const MvCell C, int ref, int dir) {
  const bool ea = A.ref == ref, eb = B.ref == ref, ec = C.ref == ref;
  const int matches = static_cast<int>(ea) + static_cast<int>(eb) + static_cast<int>(ec);
  const uint32_t lo = pk_min_i16(A.mv, B.mv), hi = pk_max_i16(A.mv, B.mv);
  const uint32_t median = pk_max_i16(lo, pk_min_i16(hi, C.mv));
  const uint32_t the_one = ea ? A.mv : (eb ? B.mv : C.mv);
  const bool only_a = matches == 0 && B.ref == -2 && C.ref == -2 && A.ref != -2;
  uint32_t p = matches == 1 ? the_one : (only_a ? A.mv : median);
  p = (dir == 1 && eb) ? B.mv : p;
  p = ((dir == 2 || dir == 3) && ea) ? A.mv : p;
  p = (dir == 4 && ec) ? C.mv : p;
  return p;
}
}  // namespace

__global__ __launch_bounds__((kMvWaves + 1) * 64) void k_mv(JobDesc* __restrict__ jobs, int njobs) {
  // Bands are handed out by ticket, band-major: whoever holds band b of a
  // picture started after the holder of band b - 1, so a poll below never waits
  // for a workgroup that is not running yet - whatever order the dispatcher
  // uses.  The counter is jobs[0].ticket_mv, zeroed by the host with the table.
  __shared__ unsigned s_ticket;
  __shared__ int s_progress[kMvWaves + 1];
  __shared__ MvWaveLds s_lds[kMvWaves];
  __shared__ MvEdge s_line[kMvWaves + 1][kLineMbs];  // [kMvWaves]: the line imported from the band above
  if (threadIdx.x == 0) s_ticket = atomicAdd(&jobs[0].ticket_mv, 1u);
  if (threadIdx.x <= kMvWaves) s_progress[threadIdx.x] = 0;
  __syncthreads();
  const int ticket = __builtin_amdgcn_readfirstlane(static_cast<int>(s_ticket));
  const int band = ticket / njobs;
  const JobDesc& job = jobs[ticket % njobs];
  if (!job.do_mv) return;
  const PictureParams& pp = job.pp;
  const MbInfo* __restrict__ mbs = job.mbs;
  MbMotion* __restrict__ motions = const_cast<MbMotion*>(job.motions);
  const PMotion* __restrict__ pmos = job.pmo;
  const uint32_t* __restrict__ pmx = job.pmx;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int W = pp.mb_width;
  BandCtl* ctl = static_cast<BandCtl*>(job.band_ctl);
  const uint32_t epoch = job.epoch;
  lds_counter* progress = (lds_counter*)s_progress;
  if (wave == kMvWaves) {
    // The band's IMPORTER.  The line of the band above comes from global memory, a round trip of a microsecond or more
    // per look; this wave does nothing but look: the next eight entries of the band above (lane -> entry lane >> 3,
    // word lane & 7), the complete ones in front of the first missing word go to LDS - where the band's first row
    // finds them like any other row finds the row above - and the counter that row waits on follows.  Whatever else
    // has arrived comes along.  A poll that never succeeds (cannot happen by construction) is reported and the
    // counter released: wrong vectors and a failed request, not a hung GPU.
    if (band == 0 || band * kMvWaves >= pp.mb_height) return;
    const uint64_t* src = ctl->lines + static_cast<size_t>(band - 1) * kLineMbs * kMvLineWords;
    int seen = 0;  // macroblocks of the row above whose edges are in s_line[kMvWaves]
    for (int tries = 0; tries < (1 << 21) && seen < W; ++tries) {
      const int e = seen + (lane >> 3), d = lane & 7;
      const bool act = d < 5 && e < W;
      uint64_t v = 0;
      if (act) v = agent_ld64(src + e * 5 + d);
      const uint64_t missing = __ballot(act && static_cast<uint32_t>(v >> 32) != epoch);
      const int nvalid = min(missing ? __builtin_ctzll(missing) >> 3 : 8, W - seen);
      if (nvalid) {
        if (act && (lane >> 3) < nvalid) reinterpret_cast<uint32_t*>(&s_line[kMvWaves][e])[d] = static_cast<uint32_t>(v);
        seen += nvalid;
        __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        progress[0] = seen;
      } else {
        __builtin_amdgcn_s_sleep(1);
      }
    }
    if (seen < W) {
      report_fault(job, FAULT_MV_IMPORT);
      progress[0] = 1 << 30;
    }
    return;
  }
  const int mby = band * kMvWaves + wave;
  if (mby >= pp.mb_height) return;  // no workgroup barrier below this point
  MvWaveLds& L = s_lds[wave];
  const uint32_t* up = reinterpret_cast<const uint32_t*>(wave == 0 ? s_line[kMvWaves] : s_line[wave - 1]);  // 5 dwords a macroblock
  uint32_t* line_me = reinterpret_cast<uint32_t*>(s_line[wave]);
  const bool exporting = wave == kMvWaves - 1 && mby + 1 < pp.mb_height;
  const int export_lanes = exporting ? 5 : 0;
  if (wave == 0 && band == 0 && lane == 0) s_progress[0] = 1 << 30;  // the row above the picture: only this wave reads it

  // ---- what a lane is in the 6 x 5 grid of a partitioned macroblock's surroundings (lanes 0 .. 29), fixed for the row:
  // which neighbour supplies the cell, where in the line above / the left column its vector and reference byte are
  const int g_row = lane / kMvGridW, g_col = lane % kMvGridW;
  int g_flag = 0;        // the MVF_AVAIL_* bit that makes the cell available (0: never - inside / right of the macroblock)
  int g_up = 0;          // row 0: dword of `up` relative to 5 * mbx that holds the vector
  int g_up_refs = 4;     //        ... and the one with the reference bytes
  int g_shift = 0;       // which byte of it
  if (lane < 30) {
    if (g_row == 0) {
      if (g_col == 0) { g_flag = MVF_AVAIL_D; g_up = -5 + 3; g_up_refs = -5 + 4; g_shift = 8; }
      else if (g_col == 5) { g_flag = MVF_AVAIL_C; g_up = 5; g_up_refs = 5 + 4; g_shift = 0; }
      else { g_flag = MVF_AVAIL_B; g_up = g_col - 1; g_up_refs = 4; g_shift = 8 * ((g_col - 1) >> 1); }
    } else if (g_col == 0) {
      g_flag = MVF_AVAIL_A;
      g_shift = 8 * ((g_row - 1) >> 1);
    }
  }
  // lanes 0 .. 35 of the MbMotion record (36 dwords: vectors of list 0, list 1 unused, surfaces, indices)
  const bool rec_vec = lane < 16, rec_slot = lane == 32, rec_ref = lane == 34, rec_ones = lane == 33 || lane == 35;

  {
    // Records travel kMvChunk macroblocks at a time, one chunk ahead: a single
    // macroblock of lead is less than the memory latency once the per-macroblock
    // work is down to a fraction of a microsecond.  Lanes 0..3 hold one macroblock each.
    // The macroblocks' motion arrives as PMotion (12 B: reference indices, their surfaces, the first
    // difference) plus, for partitioned macroblocks, further differences in the row's extra stream; a
    // 64-word window of that stream travels with every chunk.
    uint32_t n_mv = 0, n_ref = 0, n_slot = 0, n_info = 0, n_kind = 0, n_sub = 0, n_x = 0;
    uint32_t xptr = __builtin_amdgcn_readfirstlane(gld32(pmx + mby));  // this row's first extra word
    auto prefetch = [&](int mbx0, uint32_t xat) {
      if (lane < kMvChunk) {
        const int q = min(mbx0 + lane, W - 1);
        const MbInfo* mi = &mbs[mby * W + q];
        n_info = gld32(reinterpret_cast<const uint8_t*>(mi) + 24);
        n_kind = gld32(mi);
        n_sub = gld8(&mi->mv_sub);
        const uint32_t* pm = reinterpret_cast<const uint32_t*>(pmos + mby * W + q);
        n_ref = gld32(pm);
        n_slot = gld32(pm + 1);
        n_mv = gld32(pm + 2);
      }
      n_x = gld32(pmx + xat + lane);
    };
    // extra words a macroblock consumes: its partitions beyond the first (lanes 0..3 hold the chunk's records)
    auto extras_of = [&](uint32_t info, uint32_t sub) -> uint32_t {
      const uint32_t flags = (info >> 16) & 0xff, shape = (info >> 24) & 3;
      if (!(flags & MVF_DEFERRED) || (flags & MVF_PSKIP) || shape == 0) return 0;
      if (shape != 3) return 1;
      uint32_t n = 0;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const uint32_t ss = (sub >> (2 * q)) & 3;
        n += ss == 0 ? 1 : (ss == 3 ? 4 : 2);
      }
      return n - 1;
    };
    prefetch(0, xptr);
    asm volatile("" : "+v"(n_mv), "+v"(n_ref), "+v"(n_slot), "+v"(n_info), "+v"(n_kind), "+v"(n_sub), "+v"(n_x));
    int left_ref = -1;        // left neighbour's top-right quadrant: reference index ...
    uint32_t left_mv = 0;     // ... and the vector of its block 3 (registers; L.left is the LDS copy for partitioned macroblocks)
    uint32_t* out = reinterpret_cast<uint32_t*>(motions + mby * W) + lane;  // this lane's dword of the row's first record
    for (int mbx0 = 0; mbx0 < W; mbx0 += kMvChunk) {
      L.x[lane] = n_x;
      // the chunk's records stay in these registers (lane k holds macroblock k's
      // words): readlane hands them on without an LDS round trip
      const uint32_t c_info = n_info, c_kind = n_kind, c_sub = n_sub, c_ref = n_ref, c_slot = n_slot, c_mv = n_mv;
      uint32_t xo = 0;  // extras of this chunk consumed so far (index into L.x)
      {
        // where the next chunk's extras start: behind this chunk's
        const uint32_t e = (lane < kMvChunk && mbx0 + lane < W) ? extras_of(c_info, c_sub) : 0;
        const uint32_t total = __builtin_amdgcn_readlane(e, 0) + __builtin_amdgcn_readlane(e, 1) + __builtin_amdgcn_readlane(e, 2) +
                               __builtin_amdgcn_readlane(e, 3);
        xptr += total;
      }
      if (mbx0 + kMvChunk < W) prefetch(mbx0 + kMvChunk, xptr);
      wave_sync();
      for (int k = 0; k < kMvChunk && mbx0 + k < W; ++k) {
        const int mbx = mbx0 + k;
#ifdef EXP_STAMPS
        const bool st_on = EXP_STAMPS == 1 && ticket % njobs == 0 && mby == EXP_STAMP_ROW;
#endif
        STAMP(st_on, mbx, 0);
        const uint32_t info = __builtin_amdgcn_readlane(c_info, k);
        const int flags = (info >> 16) & 0xff, shape = (info >> 24) & 3, sub = __builtin_amdgcn_readlane(c_sub, k);
        const bool intra = (__builtin_amdgcn_readlane(c_kind, k) >> 8) & MBF_INTRA;  // MbInfo::flags is byte 1
        const bool deferred = flags & MVF_DEFERRED;
        const uint32_t refw = intra ? 0xffffffffu : __builtin_amdgcn_readlane(c_ref, k);
        const uint32_t slotw = intra ? 0xffffffffu : __builtin_amdgcn_readlane(c_slot, k);
        const uint32_t mvd0 = __builtin_amdgcn_readlane(c_mv, k);  // PMotion::mvd0
        auto cref = [&](int b8) { return (int)(int8_t)((refw >> (8 * b8)) & 0xff); };
        if (deferred && (flags & (MVF_AVAIL_B | MVF_AVAIL_C | MVF_AVAIL_D))) row_wait(progress, wave, min(mbx + 2, W));
        STAMP(st_on, mbx, 1);
        const bool whole = deferred && ((flags & MVF_PSKIP) || shape == 0);
        // what leaves the macroblock: lanes 0..15 their block's vector, lanes 0..3 the right column's / the bottom
        // row's; block 3's for the next macroblock's prediction (0 where there is no reference: intra, not derived here)
        uint32_t blk = 0, right = 0, bottom = 0, block3 = 0;
        if (whole) {
          // One prediction for the macroblock, from neighbours that all lie outside it.  The LDS reads go out together
          // and unconditionally (a value behind a cleared availability bit is simply not used).
          const int xc = min(mbx + 1, W - 1), xd = max(mbx - 1, 0);
          const uint32_t b_mv = up[5 * mbx], b_refs = up[5 * mbx + 4];
          const uint32_t c_mvw = up[5 * xc], c_refs = up[5 * xc + 4];
          const uint32_t d_mvw = up[5 * xd + 3], d_refs = up[5 * xd + 4];
          const bool av_a = flags & MVF_AVAIL_A, av_b = flags & MVF_AVAIL_B, av_c = flags & MVF_AVAIL_C, av_d = flags & MVF_AVAIL_D;
          MvCell A, B, C;
          A.ref = av_a ? max(left_ref, -1) : -2;
          A.mv = av_a ? left_mv : 0u;
          B.ref = av_b ? max((int)(int8_t)(b_refs & 0xff), -1) : -2;
          B.mv = av_b ? b_mv : 0u;
          const int c_ref_ = max((int)(int8_t)(c_refs & 0xff), -1), d_ref_ = max((int)(int8_t)((d_refs >> 8) & 0xff), -1);
          C.ref = av_c ? c_ref_ : (av_d ? d_ref_ : -2);
          C.mv = av_c ? c_mvw : (av_d ? d_mvw : 0u);
          const bool skip = flags & MVF_PSKIP;
          const uint32_t pred = mv_predict(A, B, C, skip ? 0 : cref(0), 0);
          const bool zero = A.ref == -2 || B.ref == -2 || (A.ref == 0 && A.mv == 0) || (B.ref == 0 && B.mv == 0);
          const uint32_t whole_mv = skip ? (zero ? 0u : pred) : pk_add_u16(pred, mvd0);
          blk = right = bottom = block3 = whole_mv;
        } else if (deferred) {
          // ---- the surroundings, one cell a lane
          if (lane < 30) {
            // (the cells of neighbours that do not exist - left of the first, right of the last macroblock - read some
            // word of the line: their availability bit is clear)
            const int last = 5 * W - 1;
            const uint32_t t_mv = up[min(max(5 * mbx + g_up, 0), last)], t_refs = up[min(max(5 * mbx + g_up_refs, 0), last)];
            const uint32_t l_mv = L.left[(g_row - 1) & 3], l_refs = L.left[4];
            const bool top = g_row == 0;
            const bool avail = (flags & g_flag) != 0;
            const int r = max((int)(int8_t)(((top ? t_refs : l_refs) >> g_shift) & 0xff), -1);
            MvCell c;
            c.ref = avail ? r : -2;
            c.mv = avail ? (top ? t_mv : l_mv) : 0u;
            L.grid[lane] = c;
          }
          wave_sync();
          int part = 0;  // partitions in decode order: the first difference is mvd0, the others follow in the stream
          const int nparts = shape == 3 ? 4 : 2;  // 16x16 took the path above
          for (int i = 0; i < nparts; ++i) {
            int px, py, pw, ph;
            if (shape == 1) { px = 0; py = i * 2; pw = 4; ph = 2; }
            else if (shape == 2) { px = i * 2; py = 0; pw = 2; ph = 4; }
            else { px = (i & 1) * 2; py = (i >> 1) * 2; pw = 2; ph = 2; }
            int sw = pw, sh = ph;
            if (shape == 3) {
              const int ss = (sub >> (2 * i)) & 3;
              if (ss == 2 || ss == 3) sw = 1;
              if (ss == 1 || ss == 3) sh = 1;
            }
            const int ref = cref((py >> 1) * 2 + (px >> 1));
            const int dir = shape == 1 ? (i == 0 ? 1 : 2) : (shape == 2 ? (i == 0 ? 3 : 4) : 0);
            for (int sy = 0; sy < ph; sy += sh)
              for (int sx = 0; sx < pw; sx += sw) {
                const int bx = px + sx, by = py + sy;
                // cells: A (bx - 1, by), B (bx, by - 1), C (bx + sw, by - 1), D (bx - 1, by - 1)
                const int at = (by + 1) * kMvGridW + bx;
                const MvCell A = L.grid[at], B = L.grid[at - kMvGridW + 1], Cr = L.grid[at - kMvGridW + 1 + sw], D = L.grid[at - kMvGridW];
                const uint32_t d = part == 0 ? mvd0 : L.x[(xo + part - 1) & 63];
                ++part;
                MvCell C;
                C.ref = Cr.ref == -2 ? D.ref : Cr.ref;
                C.mv = Cr.ref == -2 ? D.mv : Cr.mv;
                MvCell f;
                f.mv = pk_add_u16(mv_predict(A, B, C, ref, dir), d);
                f.ref = ref;
                const int x = lane & 3, y = (lane >> 2) & 3;
                if (lane < 16 && x >= bx && x < bx + sw && y >= by && y < by + sh) L.grid[(y + 1) * kMvGridW + x + 1] = f;
                wave_sync();
              }
          }
          xo += part - 1;
          blk = L.grid[(((lane >> 2) & 3) + 1) * kMvGridW + (lane & 3) + 1].mv;
          right = L.grid[((lane & 3) + 1) * kMvGridW + 4].mv;
          bottom = L.grid[4 * kMvGridW + (lane & 3) + 1].mv;
          block3 = L.grid[kMvGridW + 4].mv;
        }
        wave_sync();
        STAMP(st_on, mbx, 2);
        // The prefetched registers have arrived by now; saying so before the store
        // below is issued keeps the next chunk from waiting on that store (in-order
        // vmcnt, see k_deblock).
        if (k == kMvChunk - 1 || mbx == W - 1)
          asm volatile("" : "+v"(n_mv), "+v"(n_ref), "+v"(n_slot), "+v"(n_info), "+v"(n_kind), "+v"(n_sub), "+v"(n_x));
        // The macroblock's full MbMotion for k_inter and k_bs - written here for every compact macroblock, intra ones
        // included (no reference).  (Whether it is written is a matter of the lane mask, not of a branch: see above.)
        {
          uint32_t v = rec_vec ? blk : 0u;
          v = rec_slot ? slotw : v;
          v = rec_ref ? refw : v;
          v = rec_ones ? 0xffffffffu : v;
          if (lane < ((flags & MVF_COMPACT) ? 36 : 0)) gst32(out + mbx * 36, v);
        }
        STAMP(st_on, mbx, 3);
        const uint32_t refs_bottom = static_cast<uint32_t>(static_cast<uint8_t>(cref(2))) | (static_cast<uint32_t>(static_cast<uint8_t>(cref(3))) << 8);
        const uint32_t refs_right = static_cast<uint32_t>(static_cast<uint8_t>(cref(1))) | (static_cast<uint32_t>(static_cast<uint8_t>(cref(3))) << 8);
        if (lane < 5) {
          line_me[5 * mbx + lane] = lane < 4 ? bottom : refs_bottom;
          L.left[lane] = lane < 4 ? right : refs_right;
        }
        left_ref = cref(1);
        left_mv = block3;
        // the band below reads this row's edges from global memory
        if (lane < export_lanes)
          agent_st64(ctl->lines + (static_cast<size_t>(band) * kLineMbs + mbx) * kMvLineWords + lane,
                     (static_cast<uint64_t>(epoch) << 32) | (lane < 4 ? bottom : refs_bottom));
        __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        progress[wave + 1] = mbx + 1;
        STAMP(st_on, mbx, 4);
        STAMP_VALUE(st_on, mbx, (whole ? 1 : 0) | (deferred ? 2 : 0) | (shape << 4));
      }
      wave_sync();  // the chunk buffer is rewritten at the top of the next round
    }
  }
}

// ===========================================================================
// k_residual: dequantisation + inverse transforms of every intra macroblock,
// fully parallel (one wave per macroblock, 4 per workgroup), written as int16
// to the job's residual scratch.  It takes the coefficient loads, the scaling
// and the IDCT out of k_intra's raster-order chain: there the residual of the
// next macroblock is then a plain prefetch at an address that depends on
// nothing.
// ===========================================================================
__global__ __launch_bounds__(kInterWaves * 64) void k_residual(const JobDesc* __restrict__ jobs) {
  const JobDesc& job = jobs[blockIdx.y];
  const PictureParams& pp = job.pp;
  if (!job.do_intra) return;
  const int nmb = pp.mb_width * pp.mb_height;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;  // provably uniform: SGPRs, scalar loads
  const int mbi = blockIdx.x * kInterWaves + wave;
  if (mbi >= nmb) return;
  const MbInfo mb = gload(&job.mbs[mbi]);
  if (!(mb.flags & MBF_INTRA) || mb.kind == MBK_IPCM) return;
  __shared__ InterWave lds[kInterWaves];
  InterWave& L = lds[wave];
  uint32_t* out = reinterpret_cast<uint32_t*>(job.resid + (size_t)mbi * 384);
  if (mb.coded_mask == 0) {
    gst32(out + lane, 0);
    gst32(out + 64 + lane, 0);
    gst32(out + 128 + lane, 0);
    return;
  }
  wave_residual(mb, job.coefs, pp, lane, L.work, L.dcs, L.res_y, L.res_c, L.lv);  // ends with wave_sync
  const uint32_t* ry = reinterpret_cast<const uint32_t*>(L.res_y);
  const uint32_t* rc = reinterpret_cast<const uint32_t*>(L.res_c);
  gst32(out + lane, ry[lane]);
  gst32(out + 64 + lane, ry[64 + lane]);
  gst32(out + 128 + lane, rc[lane]);
}

// ===========================================================================
// k_intra: one workgroup of kBandWaves waves (= rows) per band of a picture; k_mv explains the bands.
//
// Two neighbour paths, chosen per picture:
//  * intra-only pictures no wider than kLineMbs macroblocks (every I picture
//    up to 2K): a macroblock's neighbours never leave LDS.  The left column
//    is the previous macroblock's tile, the row above comes from a line buffer
//    the wave of that row fills (bottom luma row + both chroma rows, 32 B per
//    macroblock).  Progress is published after an LDS-only wait, global stores
//    drain in the background and nothing in this kernel reads them back, so the
//    per-macroblock chain holds no global-memory round trip but the records.
//  * pictures with inter macroblocks (k_inter wrote those to HBM) and wider
//    pictures: neighbours are read from the surface; a release fence precedes
//    the progress counter.
// ===========================================================================

namespace {
// LDS of one band's four waves (k_intra: the workgroup; k_chain: its row waves)
struct IntraGroupLds {
  int progress[kBandWaves + 1];
  IntraWaveLds w[kBandWaves];
  uint8_t line[kBandWaves + 1][kLineMbs][32];  // [kBandWaves]: the line imported from the band above
};
// The Intra4x4 / Intra8x8 tap tables, in LDS: every one of a macroblock's (up to sixteen) dependent prediction
// steps looks its samples' taps up by (mode, position) per lane - from constant memory that was a vector-memory
// round trip inside each step of the chain.
struct IntraTables {
  uint16_t p4[9][16];
  uint32_t p8[9][64];
};
__device__ __forceinline__ void load_intra_tables(IntraTables& T, int tid, int nthreads) {
  for (int t = tid; t < 9 * 64; t += nthreads) T.p8[0][t] = c_p8tab[0][t];
  for (int t = tid; t < 9 * 16; t += nthreads) T.p4[0][t] = c_p4tab[0][t];
}
}  // namespace

// One band (kBandWaves rows, a wave each) of one picture's intra prediction.  G.progress[1 ..] must be zero on entry.
// CHAIN: as a unit of k_chain - see there.
template <bool CHAIN>
__device__ __forceinline__ void intra_band(const JobDesc& job, const int band, const int wave, const int lane, IntraGroupLds& G,
                                           const IntraTables& T) {
  const PictureParams& pp = job.pp;
  // (copies: a field read through `pp` is a load from global memory wherever it is used - the compiler cannot know that
  // the stores below leave the descriptor alone - and a load in the macroblock loop is a round trip, and a drain of
  // everything in flight, on the chain)
  const int W = pp.mb_width, H = pp.mb_height;
  const MbInfo* __restrict__ mbs = job.mbs;
  const uint32_t* __restrict__ coefs = job.coefs;
  const int mby = band * kBandWaves + wave;
  if (mby >= H) return;  // no workgroup barrier below this point
  lds_counter* progress = (lds_counter*)G.progress;
  IntraWaveLds& L = G.w[wave];
  uint8_t* tile = L.tile;
  const Surface cur = pp.surfaces[pp.cur_slot];
  const int yp = pp.y_pitch, cp = pp.c_pitch;
  const bool do_inter = job.do_inter;
  const bool lds_nb = !do_inter && W <= kLineMbs;  // wave uniform
  uint8_t (*line_up)[32] = wave == 0 ? G.line[kBandWaves] : G.line[wave - 1];
  uint8_t (*line_me)[32] = G.line[wave];
  BandCtl* ctl = static_cast<BandCtl*>(job.band_ctl);
  const uint32_t epoch = job.epoch;
  const bool importing = wave == 0 && band > 0;
  const bool exporting = wave == kBandWaves - 1 && mby + 1 < H;
  uint64_t* const xline = ctl->lines + static_cast<size_t>(bands_of(H)) * kLineMbs * kMvLineWords;
  if (wave == 0 && band == 0 && lane == 0) G.progress[0] = 1 << 30;  // the row above the picture: only this wave reads it
  int seen = 0;  // importing: macroblocks of the row above that are known final (and, with lds_nb, sit in s_line[kBandWaves])
  // lds_nb pictures: fetch the next eight entries of the band above, keep the
  // complete ones in front of the first missing word (k_mv has the same scheme)
  auto import_line = [&](int need) {
    if (seen >= need) return;
    const uint64_t* src = xline + static_cast<size_t>(band - 1) * kLineMbs * kIntraLineWords;
    for (int tries = 0; tries < (1 << 21) && seen < need; ++tries) {
      const int e = seen + (lane >> 3), d = lane & 7;
      const bool act = e < W;
      uint64_t v = 0;
      if (act) v = agent_ld64(src + e * kIntraLineWords + d);
      const uint64_t missing = __ballot(act && static_cast<uint32_t>(v >> 32) != epoch);
      const int nvalid = min(missing ? __builtin_ctzll(missing) >> 3 : 8, W - seen);
      if (act && (lane >> 3) < nvalid) reinterpret_cast<uint32_t*>(G.line[kBandWaves][e])[d] = static_cast<uint32_t>(v);
      seen += nvalid;
      if (seen < need) {
        if (nvalid) __builtin_amdgcn_s_sleep(1);
        else __builtin_amdgcn_s_sleep(1 * HW_POLL_SCALE);
      }
    }
    if (seen < need) {  // impossible by construction; reported, not hung
      report_fault(job, FAULT_INTRA_LINE);
      seen = need;
    }
    __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    wave_sync();
  };
  // other pictures: the band above publishes a counter once its bottom rows have landed
  auto import_wait = [&](int need) {
    for (int tries = 0; tries < (1 << 21) && seen < need; ++tries) {
      const uint64_t v = agent_ld64(&ctl->in_prog[band - 1]);
      if (static_cast<uint32_t>(v >> 32) == epoch) seen = static_cast<int>(static_cast<uint32_t>(v));
      if (seen < need) __builtin_amdgcn_s_sleep(2 * HW_POLL_SCALE);
    }
    if (seen < need) {
      report_fault(job, FAULT_INTRA_PROGRESS);
      seen = need;
    }
    __asm__ volatile("" ::: "memory");
  };
  // a sample of the row above: through the agent-coherent path when another workgroup wrote it
  auto up8 = [&](const uint8_t* q) -> int { return (CHAIN || importing) ? agent_ld8(q) : gld8(q); };
  auto publish = [&](int done) {  // surface path: this wave's stores land first
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    // (chain: the samples went out write-through and the row below reads them past its L1 - they have to have landed)
    if constexpr (CHAIN) __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (exporting) {
      // the workgroup-scope fence above does not wait for outstanding stores
      // (inside a workgroup the shared L1 orders them); another workgroup must
      // not see the counter before the samples
      __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (lane == 0) agent_st64(&ctl->in_prog[band], (static_cast<uint64_t>(epoch) << 32) | static_cast<uint32_t>(done));
    } else {
      progress[wave + 1] = done;
    }
    // (chain: the deblocking of this launch follows this row's and the next row's intra stage macroblock by macroblock)
    if constexpr (CHAIN) {
      if (lane == 0) agent_st64(&ctl->intra_prog[mby], (static_cast<uint64_t>(epoch) << 32) | static_cast<uint32_t>(done));
    }
  };

  uint64_t seen_inter_up = 0, seen_inter_me = 0;  // chain: the inter runs of the row above / this row known to have landed
  const uint32_t* __restrict__ resid = reinterpret_cast<const uint32_t*>(job.resid);
  {
    int prev_mbx = -2;  // macroblock whose samples the tile still holds
    for (int mbx0 = 0; mbx0 < W; mbx0 += 64) {
      // which of the next 64 macroblocks of this row are intra?
      const int probe = mbx0 + lane;
      const bool is_intra = probe < W && (gld8(&mbs[mby * W + probe].flags) & MBF_INTRA);
      unsigned long long todo = __ballot(is_intra);
      // Surface path (pictures with inter macroblocks): those are final since k_inter, so the row is complete up to its
      // next intra macroblock the moment the previous one's samples have landed - say so at once, not only "up to the
      // one just finished".  A row below then waits only where an intra macroblock really sits above an intra
      // macroblock (in a P picture they are few and scattered: the wavefront used to crawl through every row at the
      // pace of the rows above whether or not anything there concerned it).
      if (!lds_nb) publish(todo ? mbx0 + __builtin_ctzll(todo) : min(mbx0 + 64, W));
      // residual of the first macroblock of the run; inside the loop the next one's is fetched a macroblock ahead
      // (and its record: a load at the top of the macroblock would sit on the chain)
      uint32_t r0 = 0, r1 = 0, r2 = 0;
      MbInfo next_mb = {};
      if (todo) {
        const int first = mby * W + mbx0 + __builtin_ctzll(todo);
        const uint32_t* rp = resid + (size_t)first * 192;
        r0 = gld32(rp + lane);
        r1 = gld32(rp + 64 + lane);
        r2 = gld32(rp + 128 + lane);
        next_mb = gload(&mbs[first]);
      }
      while (todo) {
        const int mbx = mbx0 + __builtin_ctzll(todo);
        todo &= todo - 1;
        const MbInfo mb = next_mb;
#ifdef EXP_STAMPS
        const bool si_on = EXP_STAMPS == 3 && !CHAIN && mby == EXP_STAMP_ROW;
#endif
        STAMP(si_on, mbx, 0);
        // land this macroblock's residual in LDS, start the next one's loads
        reinterpret_cast<uint32_t*>(L.res_y)[lane] = r0;
        reinterpret_cast<uint32_t*>(L.res_y)[64 + lane] = r1;
        reinterpret_cast<uint32_t*>(L.res_c)[lane] = r2;
        if (todo) {
          const int nxt = mby * W + mbx0 + __builtin_ctzll(todo);
          const uint32_t* rp = resid + (size_t)nxt * 192;
          r0 = gld32(rp + lane);
          r1 = gld32(rp + 64 + lane);
          r2 = gld32(rp + 128 + lane);
          next_mb = gload(&mbs[nxt]);
        }
        uint8_t* dy = cur.y + (size_t)(mby * 16) * yp + mbx * 16;
        uint8_t* du = cur.u + (size_t)(mby * 8) * cp + mbx * 8;
        uint8_t* dv = cur.v + (size_t)(mby * 8) * cp + mbx * 8;
        const bool pcm = mb.kind == MBK_IPCM;
        const bool hA = mb.flags & MBF_AVAIL_A, hB = mb.flags & MBF_AVAIL_B;
        const bool hC = mb.flags & MBF_AVAIL_C, hD = mb.flags & MBF_AVAIL_D;
        if constexpr (CHAIN) {
          // the neighbours among the INTER macroblocks come from this same launch: the runs of the row above up to the
          // top-right neighbour's, and of this row up to the left neighbour's, have to have landed
          if (do_inter) {
            bool ok = true;
            if ((hB || hC || hD) && mby > 0) ok = chain_need_inter(job.chain_ctl, ctl, mby - 1, min(mbx + 1, W - 1), seen_inter_up);
            if (ok && hA && mbx > 0) ok = chain_need_inter(job.chain_ctl, ctl, mby, mbx - 1, seen_inter_me);
            if (!ok) report_fault(job, FAULT_CHAIN_INTER);
          }
        }
        // the row above must be complete up to the top-right neighbour
        if (hB || hC || hD) {
          const int need = min(mbx + 2, W);
          if (!importing) row_wait(progress, wave, need);
          else if (lds_nb) import_line(need);
          else import_wait(need);
        }
        STAMP(si_on, mbx, 1);
        // ---- neighbours into the tiles
        if (lds_nb) {
          // left column: the previous macroblock of this row is still in the tile
          if (lane >= 32 && lane < 48) {
            const int r = lane - 31;
            tile[r * TS] = (hA && prev_mbx == mbx - 1) ? tile[r * TS + 16] : 128;
          } else if (lane >= 48) {
            const int p = (lane - 48) >> 3, j = (lane - 48) & 7;
            L.ctile[p][(j + 1) * 12] = (hA && prev_mbx == mbx - 1) ? L.ctile[p][(j + 1) * 12 + 8] : 128;
          }
          if (lane < 25) {
            int v = 128;
            if (lane == 0) { if (hD) v = line_up[mbx - 1][15]; }
            else if (lane <= 16) { if (hB) v = line_up[mbx][lane - 1]; }
            else if (hC) v = line_up[mbx + 1][lane - 17];
            tile[lane] = (uint8_t)v;
          } else if (lane < 32) {
            // nothing
          }
          if (lane < 18) {
            const int p = lane / 9, i = lane % 9;
            int v = 128;
            if (i == 0) { if (hD) v = line_up[mbx - 1][16 + 8 * p + 7]; }
            else if (hB) v = line_up[mbx][16 + 8 * p + i - 1];
            L.ctile[p][i] = (uint8_t)v;
          }
        } else {
          if (lane < 25)
            tile[lane] = (lane == 0) ? (hD ? up8(dy - yp - 1) : 128) : (lane <= 16) ? (hB ? up8(dy - yp + lane - 1) : 128)
                                                                                    : (hC ? up8(dy - yp + lane - 1) : 128);
          else if (lane >= 32 && lane < 48)
            tile[(lane - 31) * TS] = hA ? sld8<CHAIN>(dy + (lane - 32) * yp - 1) : 128;
          if (lane < 18) {
            const int p = lane / 9, i = lane % 9;
            const uint8_t* d = p ? dv : du;
            L.ctile[p][i] = (i == 0) ? (hD ? up8(d - cp - 1) : 128) : (hB ? up8(d - cp + i - 1) : 128);
          } else if (lane >= 32 && lane < 48) {
            const int p = (lane - 32) >> 3, j = (lane - 32) & 7;
            const uint8_t* d = p ? dv : du;
            L.ctile[p][(j + 1) * 12] = hA ? sld8<CHAIN>(d + j * cp - 1) : 128;
          }
        }
        wave_sync();
        STAMP(si_on, mbx, 2);

        if (pcm) {
          const uint8_t* src = reinterpret_cast<const uint8_t*>(coefs + mb.coef_offset);
          for (int t = lane; t < 256; t += 64) tile[(1 + (t >> 4)) * TS + 1 + (t & 15)] = gld8(src + t);
          for (int t = lane; t < 128; t += 64) L.ctile[t >> 6][(1 + ((t >> 3) & 7)) * 12 + 1 + (t & 7)] = gld8(src + 256 + t);
        } else if (mb.kind == MBK_I16x16) {
          const int mode = mb.pred_modes & 15;
          int dcv = 128, a = 0, b = 0, c = 0;
          if (mode == 2) {
            int st = 0, sl = 0;
            for (int i = 0; i < 16; ++i) {
              st += tile[1 + i];
              sl += tile[(1 + i) * TS];
            }
            dcv = (hB && hA) ? (st + sl + 16) >> 5 : hA ? (sl + 8) >> 4 : hB ? (st + 8) >> 4 : 128;
          } else if (mode == 3) {
            int Hh = 0, Vv = 0;
            for (int k = 0; k < 8; ++k) {
              Hh += (k + 1) * (tile[1 + 8 + k] - tile[1 + 6 - k]);
              Vv += (k + 1) * (tile[(1 + 8 + k) * TS] - tile[(1 + 6 - k) * TS]);
            }
            a = 16 * (tile[16 * TS] + tile[16]);
            b = (5 * Hh + 32) >> 6;
            c = (5 * Vv + 32) >> 6;
          }
          int outv[4];
#pragma unroll
          for (int it = 0; it < 4; ++it) {
            const int t = lane + it * 64, x = t & 15, y = t >> 4;
            const int v = mode == 0 ? tile[1 + x] : mode == 1 ? tile[(1 + y) * TS] : mode == 2 ? dcv
                                                  : clip8((a + b * (x - 7) + c * (y - 7) + 16) >> 5);
            outv[it] = clip8(v + L.res_y[t]);
          }
          wave_sync();  // every lane has read its neighbours before the tile is overwritten
#pragma unroll
          for (int it = 0; it < 4; ++it) {
            const int t = lane + it * 64;
            tile[(1 + (t >> 4)) * TS + 1 + (t & 15)] = (uint8_t)outv[it];
          }
        } else if (mb.kind == MBK_I4x4) {
          // the blocks' sixteen prediction modes as one word: nibble r = raster block r (a register shift instead of
          // indexing the record per lane, which went through scratch memory)
          const uint64_t modes4 = (uint64_t)mb.i4_modes[0] | ((uint64_t)mb.i4_modes[1] << 8) | ((uint64_t)mb.i4_modes[2] << 16) |
                                  ((uint64_t)mb.i4_modes[3] << 24) | ((uint64_t)mb.i4_modes[4] << 32) | ((uint64_t)mb.i4_modes[5] << 40) |
                                  ((uint64_t)mb.i4_modes[6] << 48) | ((uint64_t)mb.i4_modes[7] << 56);
          for (int st = 0; st < 10; ++st) {
            const int half = lane >> 4;  // lanes 0..15 first block of the step, 16..31 second
            // step -> raster block: first {0,1,2,3,6,7,10,11,14,15}, second {-,-,4,5,8,9,12,13,-,-} (nibbles of constants)
            const int raster = half == 0 ? (int)((0xFEBA763210ull >> (4 * st)) & 15)
                                         : (half == 1 && st >= 2 && st <= 7) ? (int)((0x00DC985400ull >> (4 * st)) & 15) : -1;
            const int bx = raster & 3, by = raster >> 2;
            const bool hl = bx > 0 || hA, ht = by > 0 || hB;
            const uint8_t* t = tile + (1 + by * 4) * TS + 1 + bx * 4;
            if (raster >= 0 && (lane & 15) < 13) {
              const int i = lane & 15;
              const bool htl = (bx > 0 && by > 0) ? true : (bx > 0) ? hB : (by > 0) ? hA : hD;
              bool htr;
              if (by == 0) htr = bx < 3 ? hB : hC;
              else if (bx == 3) htr = false;
              else htr = zz_of_raster((by - 1) * 4 + bx + 1) < zz_of_raster(raster);
              int v = 128;
              if (i < 4) { if (hl) v = t[(3 - i) * TS - 1]; }
              else if (i == 4) { if (htl) v = t[-TS - 1]; }
              else if (ht) v = (i > 8 && !htr) ? t[-TS + 3] : t[-TS + i - 5];
              L.e4[half][i] = (uint8_t)v;
            }
            wave_sync();
            if (raster >= 0) {
              const int px = lane & 3, py = (lane >> 2) & 3;
              const int mode = (int)((modes4 >> (4 * raster)) & 15);
              const uint8_t* e = L.e4[half];
              int v;
              if (mode == 2) {
                const int st4 = e[5] + e[6] + e[7] + e[8], sl4 = e[0] + e[1] + e[2] + e[3];
                v = (ht && hl) ? (st4 + sl4 + 4) >> 3 : hl ? (sl4 + 2) >> 2 : ht ? (st4 + 2) >> 2 : 128;
              } else {
                v = tap_sample(e, T.p4[mode > 8 ? 8 : mode][py * 4 + px], 4, 15);
              }
              const int out = clip8(v + L.res_y[(by * 4 + py) * 16 + bx * 4 + px]);
              tile[(1 + by * 4 + py) * TS + 1 + bx * 4 + px] = (uint8_t)out;
            }
            wave_sync();
          }
        } else {  // MBK_I8x8
          for (int b8 = 0; b8 < 4; ++b8) {
            const int bx = (b8 & 1) * 8, by = (b8 >> 1) * 8;
            const bool hl = bx > 0 || hA, ht = by > 0 || hB;
            const bool htl = (bx > 0 && by > 0) ? true : (bx > 0) ? hB : (by > 0) ? hA : hD;
            const bool htr = b8 == 0 ? hB : b8 == 1 ? hC : b8 == 2;
            const uint8_t* t = tile + (1 + by) * TS + 1 + bx;
            // reference sample filtering (8.3.2.2.1) into E: [0..7] left bottom-up, [8] corner, [9..24] top
            if (lane < 16) {
              int v = 128;
              if (ht) {
                auto R = [&](int i) -> int { return (i > 7 && !htr) ? t[-TS + 7] : t[-TS + i]; };
                if (lane == 0) v = htl ? (t[-TS - 1] + 2 * R(0) + R(1) + 2) >> 2 : (3 * R(0) + R(1) + 2) >> 2;
                else if (lane == 15) v = (R(14) + 3 * R(15) + 2) >> 2;
                else v = (R(lane - 1) + 2 * R(lane) + R(lane + 1) + 2) >> 2;
              }
              L.e8[9 + lane] = (uint8_t)v;
            } else if (lane < 24) {
              const int j = lane - 16;
              int v = 128;
              if (hl) {
                auto Lr = [&](int i) -> int { return t[i * TS - 1]; };
                if (j == 0) v = htl ? (t[-TS - 1] + 2 * Lr(0) + Lr(1) + 2) >> 2 : (3 * Lr(0) + Lr(1) + 2) >> 2;
                else if (j == 7) v = (Lr(6) + 3 * Lr(7) + 2) >> 2;
                else v = (Lr(j - 1) + 2 * Lr(j) + Lr(j + 1) + 2) >> 2;
              }
              L.e8[7 - j] = (uint8_t)v;
            } else if (lane == 24) {
              int v = 128;
              if (htl) {
                const int cc = t[-TS - 1];
                if (ht && hl) v = (t[-TS] + 2 * cc + t[-1] + 2) >> 2;
                else if (ht) v = (3 * cc + t[-TS] + 2) >> 2;
                else if (hl) v = (3 * cc + t[-1] + 2) >> 2;
                else v = cc;
              }
              L.e8[8] = (uint8_t)v;
            }
            wave_sync();
            {
              const int px = lane & 7, py = lane >> 3;
              const int raster = (by >> 2) * 4 + (bx >> 2);
              const int mode = (mb.i4_modes[raster >> 1] >> ((raster & 1) * 4)) & 15;  // raster: one of four constants
              const uint8_t* e = L.e8;
              int v;
              if (mode == 2) {
                int st8 = 0, sl8 = 0;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                  st8 += e[9 + i];
                  sl8 += e[i];
                }
                v = (ht && hl) ? (st8 + sl8 + 8) >> 4 : hl ? (sl8 + 4) >> 3 : ht ? (st8 + 4) >> 3 : 128;
              } else {
                v = tap_sample(e, T.p8[mode > 8 ? 8 : mode][py * 8 + px], 5, 31);
              }
              tile[(1 + by + py) * TS + 1 + bx + px] = (uint8_t)clip8(v + L.res_y[(by + py) * 16 + bx + px]);
            }
            wave_sync();
          }
        }

        STAMP(si_on, mbx, 3);
        // chroma 8x8, both planes: two samples per lane
        if (!pcm) {
          int outc[2];
#pragma unroll
          for (int it = 0; it < 2; ++it) {
            const int p = it, cx = lane & 7, cy = lane >> 3;
            const uint8_t* ct = L.ctile[p];
            const int mode = mb.pred_modes >> 4;
            int v;
            if (mode == 0) {
              const int bx = cx >> 2, by = cy >> 2;
              int st = 0, sl = 0;
              for (int i = 0; i < 4; ++i) {
                st += ct[1 + 4 * bx + i];
                sl += ct[(1 + 4 * by + i) * 12];
              }
              if (bx == by) v = (hB && hA) ? (st + sl + 4) >> 3 : hB ? (st + 2) >> 2 : hA ? (sl + 2) >> 2 : 128;
              else if (bx == 1) v = hB ? (st + 2) >> 2 : hA ? (sl + 2) >> 2 : 128;
              else v = hA ? (sl + 2) >> 2 : hB ? (st + 2) >> 2 : 128;
            } else if (mode == 1) {
              v = ct[(1 + cy) * 12];
            } else if (mode == 2) {
              v = ct[1 + cx];
            } else {
              int Hh = 0, Vv = 0;
              for (int k = 0; k < 4; ++k) {
                Hh += (k + 1) * (ct[1 + 4 + k] - ct[1 + 2 - k]);
                Vv += (k + 1) * (ct[(1 + 4 + k) * 12] - ct[(1 + 2 - k) * 12]);
              }
              const int a = 16 * (ct[8 * 12] + ct[8]);
              const int b = (34 * Hh + 32) >> 6, c = (34 * Vv + 32) >> 6;
              v = clip8((a + b * (cx - 3) + c * (cy - 3) + 16) >> 5);
            }
            outc[it] = clip8(v + L.res_c[p * 64 + cy * 8 + cx]);
          }
          wave_sync();
          L.ctile[0][(1 + (lane >> 3)) * 12 + 1 + (lane & 7)] = (uint8_t)outc[0];
          L.ctile[1][(1 + (lane >> 3)) * 12 + 1 + (lane & 7)] = (uint8_t)outc[1];
        }
        wave_sync();

        if constexpr (CHAIN) {
          if (lds_nb) {
            // (the LDS hand-over to the row below needs no stores to have landed; the deblocking of this launch does:
            // what is in flight here is the PREVIOUS macroblock's samples - say that everything before this one is there)
            __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) agent_st64(&ctl->intra_prog[mby], (static_cast<uint64_t>(epoch) << 32) | static_cast<uint32_t>(mbx));
          }
        }
        STAMP(si_on, mbx, 4);
        // ---- tiles -> surface: one dword per lane (luma), 32 lanes x dword (chroma)
        {
          const int r = lane >> 2, c4 = (lane & 3) * 4;
          const uint8_t* tp = tile + (1 + r) * TS + 1 + c4;
          const uint32_t v = tp[0] | (tp[1] << 8) | (tp[2] << 16) | ((uint32_t)tp[3] << 24);
          // (surface path: the band below reads the bottom rows from another workgroup)
          const bool coherent = exporting && !lds_nb;
          if (CHAIN || (coherent && r == 15)) agent_st32(dy + (size_t)r * yp + c4, v);
          else gst32(dy + (size_t)r * yp + c4, v);
          if (lane < 32) {
            const int p = lane >> 4, rr = (lane >> 1) & 7, cc = (lane & 1) * 4;
            const uint8_t* cpx = L.ctile[p] + (1 + rr) * 12 + 1 + cc;
            const uint32_t w = cpx[0] | (cpx[1] << 8) | (cpx[2] << 16) | ((uint32_t)cpx[3] << 24);
            if (CHAIN || (coherent && rr == 7)) agent_st32((p ? dv : du) + (size_t)rr * cp + cc, w);
            else gst32((p ? dv : du) + (size_t)rr * cp + cc, w);
          }
        }
        if (lds_nb) {
          // bottom rows for the row below, then the counter: LDS only
          if (lane < 16) line_me[mbx][lane] = tile[16 * TS + 1 + lane];
          else if (lane < 32) line_me[mbx][lane] = L.ctile[(lane - 16) >> 3][8 * 12 + 1 + ((lane - 16) & 7)];
          __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          if (exporting) {
            // the same 32 bytes, as eight tagged words, for the band below
            wave_sync();
            if (lane < 8)
              agent_st64(xline + (static_cast<size_t>(band) * kLineMbs + mbx) * kIntraLineWords + lane,
                         (static_cast<uint64_t>(epoch) << 32) | reinterpret_cast<const uint32_t*>(line_me[mbx])[lane]);
            __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          } else {
            progress[wave + 1] = mbx + 1;
          }
        } else {
          publish(todo ? mbx0 + __builtin_ctzll(todo) : min(mbx0 + 64, W));  // up to the next intra macroblock
        }
        prev_mbx = mbx;
        STAMP(si_on, mbx, 5);
        STAMP_VALUE(si_on, mbx, mb.kind);
      }
    }
    // inter macroblocks at the end of the row need no work: the row is done
    if (lds_nb) {
      __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (!exporting) progress[wave + 1] = W;  // (every macroblock of an lds_nb picture exported its own entry)
    } else {
      publish(W);
    }
    if constexpr (CHAIN) {
      // the row is through: its samples have landed
      __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (lane == 0) agent_st64(&ctl->intra_prog[mby], (static_cast<uint64_t>(epoch) << 32) | static_cast<uint32_t>(W));
    }
  }
}


__global__ __launch_bounds__(kBandWaves * 64) void k_intra(JobDesc* __restrict__ jobs, int njobs) {
  // one workgroup per band of kBandWaves rows, bands handed out by ticket (see k_mv)
  __shared__ unsigned s_ticket;
  __shared__ IntraGroupLds s_group;
  __shared__ IntraTables s_tables;
  if (threadIdx.x == 0) s_ticket = atomicAdd(&jobs[0].ticket_in, 1u);
  if (threadIdx.x <= kBandWaves) s_group.progress[threadIdx.x] = 0;
  load_intra_tables(s_tables, threadIdx.x, kBandWaves * 64);
  __syncthreads();
  const int band = static_cast<int>(s_ticket) / njobs;
  const JobDesc& job = jobs[static_cast<int>(s_ticket) % njobs];
  if (!job.do_intra) return;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;  // provably uniform: SGPRs, scalar loads
  intra_band<false>(job, band, wave, lane, s_group, s_tables);
}

// ===========================================================================
// k_deblock: one workgroup of kBandWaves waves (= rows) per band of a picture.
// ===========================================================================
namespace {

// One line across an edge: p3 p2 p1 p0 | q0 q1 q2 q3 (chroma uses p1 .. q1 only), filtered in place (8.7.2.3 / 8.7.2.4).
// Written without a branch on a value: a lone wave issues an instruction every four to five clocks whatever it is, a
// row's step is the number of instructions on its path (see k_mv), and the lanes of one edge differ in everything -
// strength, plane, whether the thresholds let the filter act at all.  Every lane computes the bS < 4 filter; the
// bS = 4 formulas are computed where SOME lane of the wave needs them (`some_strong`, uniform).
struct EdgeLine {
  int p3, p2, p1, p0, q0, q1, q2, q3;
};
__device__ __forceinline__ void filter_line(EdgeLine& v, int bs, int alpha, int beta, int tc0, bool chroma, bool some_strong) {
  const int p0 = v.p0, p1 = v.p1, p2 = v.p2, q0 = v.q0, q1 = v.q1, q2 = v.q2;
  const bool on = bs != 0 && iabs(p0 - q0) < alpha && iabs(p1 - p0) < beta && iabs(q1 - q0) < beta;
  const bool ap = !chroma && iabs(p2 - p0) < beta, aq = !chroma && iabs(q2 - q0) < beta;
  const int tc = tc0 + (chroma ? 1 : static_cast<int>(ap) + static_cast<int>(aq));
  const int delta = iclip(-tc, tc, (((q0 - p0) << 2) + (p1 - q1) + 4) >> 3);
  const int avg = (p0 + q0 + 1) >> 1;
  int np0 = clip8(p0 + delta), nq0 = clip8(q0 - delta);
  // (KEEP: the value exists before the select that may discard it - otherwise the compiler sinks its computation into
  // an exec-mask region of its own, and a region costs more issue slots than the three or four instructions it guards)
#define KEEP(x) asm volatile("" : "+v"(x))
  int mp1 = p1 + iclip(-tc0, tc0, (p2 + avg - (p1 << 1)) >> 1), mq1 = q1 + iclip(-tc0, tc0, (q2 + avg - (q1 << 1)) >> 1);
  KEEP(mp1);
  KEEP(mq1);
  int np1 = ap ? mp1 : p1;
  int nq1 = aq ? mq1 : q1;
  int np2 = p2, nq2 = q2;
  if (some_strong) {
    const bool strong = bs == 4;
    const bool close = iabs(p0 - q0) < ((alpha >> 2) + 2);
    const bool sp = ap && close, sq = aq && close;
    int a0 = (p2 + 2 * p1 + 2 * p0 + 2 * q0 + q1 + 4) >> 3, b0 = (2 * p1 + p0 + q1 + 2) >> 2;
    int c0 = (p1 + 2 * p0 + 2 * q0 + 2 * q1 + q2 + 4) >> 3, d0 = (2 * q1 + q0 + p1 + 2) >> 2;
    int a1 = (p2 + p1 + p0 + q0 + 2) >> 2, c1 = (p0 + q0 + q1 + q2 + 2) >> 2;
    int a2 = (2 * v.p3 + 3 * p2 + p1 + p0 + q0 + 4) >> 3, c2 = (2 * v.q3 + 3 * q2 + q1 + q0 + p0 + 4) >> 3;
    KEEP(a0);
    KEEP(b0);
    KEEP(c0);
    KEEP(d0);
    KEEP(a1);
    KEEP(c1);
    KEEP(a2);
    KEEP(c2);
    const int s_p0 = sp ? a0 : b0, s_q0 = sq ? c0 : d0;
    const int s_p1 = sp ? a1 : p1, s_q1 = sq ? c1 : q1;
    const int s_p2 = sp ? a2 : p2, s_q2 = sq ? c2 : q2;
    np0 = strong ? s_p0 : np0;
    nq0 = strong ? s_q0 : nq0;
    np1 = strong ? s_p1 : np1;
    nq1 = strong ? s_q1 : nq1;
    np2 = strong ? s_p2 : np2;
    nq2 = strong ? s_q2 : nq2;
  }
  v.p0 = on ? np0 : p0;
  v.q0 = on ? nq0 : q0;
  v.p1 = on ? np1 : p1;
  v.q1 = on ? nq1 : q1;
  v.p2 = on ? np2 : p2;
  v.q2 = on ? nq2 : q2;
#undef KEEP
}
// The same across a VERTICAL edge, on the two dwords of a row that meet there (lo = p3 p2 p1 p0, hi = q0 q1 q2 q3).
__device__ __forceinline__ void filter_dwords(uint32_t& lo, uint32_t& hi, int bs, int alpha, int beta, int tc0, bool chroma, bool some_strong) {
  EdgeLine v;
  v.p3 = lo & 0xff;
  v.p2 = (lo >> 8) & 0xff;
  v.p1 = (lo >> 16) & 0xff;
  v.p0 = lo >> 24;
  v.q0 = hi & 0xff;
  v.q1 = (hi >> 8) & 0xff;
  v.q2 = (hi >> 16) & 0xff;
  v.q3 = hi >> 24;
  filter_line(v, bs, alpha, beta, tc0, chroma, some_strong);
  lo = static_cast<uint32_t>(v.p3) | (static_cast<uint32_t>(v.p2) << 8) | (static_cast<uint32_t>(v.p1) << 16) | (static_cast<uint32_t>(v.p0) << 24);
  hi = static_cast<uint32_t>(v.q0) | (static_cast<uint32_t>(v.q1) << 8) | (static_cast<uint32_t>(v.q2) << 16) | (static_cast<uint32_t>(v.q3) << 24);
}

__device__ int boundary_strength(const MbInfo& pm, const MbInfo& qm, const MbMotion* pmo, const MbMotion* qmo,
                                 int pblk, int qblk, bool mb_edge) {
  if ((pm.flags | qm.flags) & MBF_INTRA) return mb_edge ? 4 : 3;
  if (((pm.nz_mask >> pblk) & 1) || ((qm.nz_mask >> qblk) & 1)) return 2;
  const int p8 = (pblk >> 3) * 2 + ((pblk & 3) >> 1), q8 = (qblk >> 3) * 2 + ((qblk & 3) >> 1);
  const int pr0 = pmo->ref_slot[0][p8], pr1 = pmo->ref_slot[1][p8];
  const int qr0 = qmo->ref_slot[0][q8], qr1 = qmo->ref_slot[1][q8];
  const int16_t* pm0 = pmo->mv[0][pblk];
  const int16_t* pm1 = pmo->mv[1][pblk];
  const int16_t* qm0 = qmo->mv[0][qblk];
  const int16_t* qm1 = qmo->mv[1][qblk];
#define FAR(a, b) (iabs((a)[0] - (b)[0]) >= 4 || iabs((a)[1] - (b)[1]) >= 4)
  if (!((pr0 == qr0 && pr1 == qr1) || (pr0 == qr1 && pr1 == qr0))) return 1;
  if (pr0 == pr1) {
    if (pr0 == -1) return 0;  // (-1 = no reference; other negative bytes are surfaces 128 .. 254)
    const bool straight = FAR(pm0, qm0) || FAR(pm1, qm1);
    const bool cross = FAR(pm0, qm1) || FAR(pm1, qm0);
    return (straight && cross) ? 1 : 0;
  }
  if (pr0 == qr0) {
    if (pr0 != -1 && FAR(pm0, qm0)) return 1;
    if (pr1 != -1 && FAR(pm1, qm1)) return 1;
    return 0;
  }
  if (pr0 != -1 && FAR(pm0, qm1)) return 1;
  if (pr1 != -1 && FAR(pm1, qm0)) return 1;
  return 0;
#undef FAR
}

// Per-macroblock deblocking record, produced by k_bs (fully parallel) so the
// serial row pipeline touches no MbInfo / motion data: 32 boundary strengths
// (16 vertical-edge segments, then 16 horizontal; 0 = edge not filtered) and
// the table indices indexA / indexB per plane for the left edge, the top edge
// and the internal edges.
struct DbInfo {
  uint8_t bs[2][4][4];   // [0 vertical / 1 horizontal edges][position along the edge / 4][edge 0..3]:
                         // the four strengths one filter lane needs are one dword
  uint8_t alpha[3][4];   // [plane][0 left, 1 top, 2 internal, -]: a plane's thresholds are one dword for the lane
  uint8_t beta[3][4];
  uint8_t tc0[3][3][4];  // [plane][class][bS - 1, -]
  uint8_t t8x8;
  uint8_t any;           // some bs != 0
  uint8_t emask;         // bit e: vertical edge e has a non-zero bS, bit 4 + e: horizontal edge e
  uint8_t pad[1];
};
static_assert(sizeof(DbInfo) == 96, "DbInfo must stay 96 bytes");

// A macroblock's samples while it is filtered: three tiles (Y, Cb, Cr) of 20 rows x 24 bytes, twice (the tile of the
// macroblock before stays readable: its last columns are this one's left neighbours).
//   luma   row y (-4 .. 15) -> tile row y + 4;  column x (0 .. 15) -> byte 4 + x
//   chroma row j (-2 .. 7)  -> tile row j + 4 for j < 2, j + 8 from 2 on;  column x -> byte 4 + x for x < 4, 8 + x from 4 on
// so that the horizontal edges of all three planes are the same code: edge e of a column reads tile rows 4e .. 4e + 7
// (p3 .. q3; chroma edge 1 of the standard is e = 2 and uses p1 .. q1 = rows 4e + 2 .. 4e + 5: chroma rows 2 .. 5), and
// a row's two dwords that meet at a vertical edge are dwords e, e + 1 of the tile row (chroma: 1 | 3 for its second edge).
constexpr int TR = 24, kTileBytes = 20 * TR;
struct DeblockWaveLds {
  alignas(16) uint8_t tile[2][3 * kTileBytes];
  DbInfo db;
};

}  // namespace

// k_bs: 32 lanes per macroblock, 8 macroblocks per 256-thread workgroup.
__global__ __launch_bounds__(256) void k_bs(const JobDesc* __restrict__ jobs) {
  const JobDesc& job = jobs[blockIdx.y];
  if (!job.do_deblock) return;
  const PictureParams& pp = job.pp;
  const MbInfo* __restrict__ mbs = job.mbs;
  const MbMotion* __restrict__ motions = job.motions;
  const SliceParams* __restrict__ slices = job.slices;
  DbInfo* __restrict__ out = static_cast<DbInfo*>(job.db);
  const int nmb = pp.mb_width * pp.mb_height;
  // Workgroups are numbered so that each XCD (blockIdx % 8) owns a contiguous stretch of the picture, as in
  // k_inter: a macroblock's records are read by its own workgroup and by the one of the macroblock below, and
  // with round-robin numbering those sat on different XCDs - every record was fetched from memory twice
  // (PMC: 2.9 MB read per picture for 1.4 MB of records).
  const int ngroups = (nmb + 7) / 8;
  const int band = (ngroups + 7) / 8;
  const int group = (blockIdx.x % 8) * band + blockIdx.x / 8;
  if (blockIdx.x / 8 >= band || group >= ngroups) return;
  const int mbi = group * 8 + (threadIdx.x >> 5);
  const int t = threadIdx.x & 31;
  if (mbi >= nmb) return;
  const MbInfo mb = mbs[mbi];
  const bool do_left = mb.flags & MBF_DB_LEFT, do_top = mb.flags & MBF_DB_TOP;
  const int dir = t >> 4, e = (t >> 2) & 3, k = t & 3;
  int bs = 0;
  const bool edge_on = e == 0 ? (dir == 0 ? do_left : do_top) : (mb.deblock_internal != 0);
  const bool skip_luma_only = (mb.flags & MBF_T8X8) && (e & 1);  // odd edges: neither luma nor chroma
  if (edge_on && !skip_luma_only) {
    const int qblk = dir == 0 ? k * 4 + e : e * 4 + k;
    if (e == 0) {
      const int nbi = dir == 0 ? mbi - 1 : mbi - pp.mb_width;
      const MbInfo nb = mbs[nbi];
      const int pblk = dir == 0 ? k * 4 + 3 : 12 + k;
      if ((nb.flags | mb.flags) & MBF_INTRA) bs = 4;
      else bs = boundary_strength(nb, mb, &motions[nbi], &motions[mbi], pblk, qblk, true);
    } else {
      const int pblk = dir == 0 ? k * 4 + e - 1 : (e - 1) * 4 + k;
      if (mb.flags & MBF_INTRA) bs = 3;
      else bs = boundary_strength(mb, mb, &motions[mbi], &motions[mbi], pblk, qblk, false);
    }
  }
  // The record is assembled in LDS and leaves as 24 dwords: single-byte global
  // stores from 32 lanes cost ~14x the record's size in HBM write traffic
  // (rocprofv3 WRITE_SIZE: 11.2 MB per 1080p picture for 0.78 MB of records).
  __shared__ DbInfo s_db[8];
  DbInfo* o = &s_db[threadIdx.x >> 5];
  o->bs[dir][k][e] = (uint8_t)bs;
  const unsigned long long nz = __ballot(bs != 0);
  if (t < 9) {
    const int plane = t / 3, cls = t % 3;  // cls: 0 left, 1 top, 2 internal
    const SliceParams& sp = slices[mb.slice];
    // (no dynamic indexing into the header copies: that would put them in scratch memory)
    const int qc = plane == 0 ? mb.qp : (plane == 1 ? mb.qpc[0] : mb.qpc[1]);
    int qn = qc;
    if (!(cls == 2 || (cls == 0 && !do_left) || (cls == 1 && !do_top))) {
      const MbInfo nb = gload(&mbs[cls == 0 ? mbi - 1 : mbi - pp.mb_width]);
      qn = plane == 0 ? nb.qp : (plane == 1 ? nb.qpc[0] : nb.qpc[1]);
    }
    const int qpav = (qn + qc + 1) >> 1;
    const int ia = iclip(0, 51, qpav + sp.alpha_offset), ib = iclip(0, 51, qpav + sp.beta_offset);
    o->alpha[plane][cls] = c_alpha[ia];
    o->beta[plane][cls] = c_beta[ib];
    for (int b = 0; b < 3; ++b) o->tc0[plane][cls][b] = c_tc0[ia][b];
  }
  if (t == 0) {
    o->t8x8 = (mb.flags & MBF_T8X8) ? 1 : 0;
    // lanes of this macroblock are bits [32*(mb&1) ...) of the wave's ballot
    const unsigned mine = (unsigned)(nz >> (32 * ((threadIdx.x >> 5) & 1)));
    o->any = mine != 0;
    unsigned em = 0;  // lane t = dir * 16 + e * 4 + k: one nibble per edge
    for (int g = 0; g < 8; ++g)
      if ((mine >> (4 * g)) & 0xf) em |= 1u << g;
    o->emask = (uint8_t)em;
  }
  wave_sync();  // the 32 lanes of a macroblock are half a wave
  if (t < 24) gst32(reinterpret_cast<uint32_t*>(&out[mbi]) + t, reinterpret_cast<const uint32_t*>(o)[t]);
}

// ===========================================================================
// k_deblock: one workgroup of kBandWaves waves per band; every wave walks its
// row one macroblock at a time.  Per macroblock the only dependent memory
// traffic is the four rows above (after the row above has passed); the
// macroblock's own samples and its DbInfo are prefetched one iteration ahead
// and its stores are drained one iteration late.
// ===========================================================================
namespace {
struct DeblockGroupLds {
  int progress[kBandWaves + 1];
  DeblockWaveLds w[kBandWaves];
};
}  // namespace

// One band (kBandWaves rows, a wave each) of one picture's deblocking.  G.progress[1 ..] must be zero on entry.
// CHAIN: as a unit of k_chain - see there.
template <bool CHAIN>
__device__ __forceinline__ void deblock_band(const JobDesc& job, const int band, const int wave, const int lane, DeblockGroupLds& G) {
  const PictureParams& pp = job.pp;
  const DbInfo* __restrict__ dbs = static_cast<const DbInfo*>(job.db);
  const int mby = band * kBandWaves + wave;
  if (mby >= pp.mb_height) return;  // no workgroup barrier below this point
  lds_counter* progress = (lds_counter*)G.progress;
  DeblockWaveLds& L = G.w[wave];
  const Surface cur = pp.surfaces[pp.cur_slot];
  const int yp = pp.y_pitch, cp = pp.c_pitch;
  const int W = pp.mb_width;
  BandCtl* ctl = static_cast<BandCtl*>(job.band_ctl);
  const uint32_t epoch32 = job.epoch;  // (locals: see the lesson about descriptor fields in loops)
  const uint64_t epoch = static_cast<uint64_t>(epoch32) << 32;
  const bool has_row_below = mby + 1 < pp.mb_height;
  const bool importing = wave == 0 && band > 0;
  const bool exporting = wave == kBandWaves - 1 && has_row_below;
  if (wave == 0 && band == 0 && lane == 0) G.progress[0] = 1 << 30;  // the row above the picture: only this wave reads it
  int seen = 0;  // importing: macroblocks of the row above known to be final
  // test hook: a band whose progress words carry a wrong tag looks exactly like a band that never reports
  const uint64_t epoch_out = job.inject_fault == 1 ? epoch ^ (1ull << 63) : epoch;
  auto import_wait = [&](int need) {
    for (int tries = 0; tries < (1 << 21) && seen < need; ++tries) {
      const uint64_t v = agent_ld64(&ctl->db_prog[band - 1]);
      if ((v & 0xffffffff00000000ull) == epoch) seen = static_cast<int>(static_cast<uint32_t>(v));
      if (seen < need) __builtin_amdgcn_s_sleep(2 * HW_POLL_SCALE);
    }
    if (seen < need) {  // a poll that never succeeds (impossible by construction) is reported to the host, and does not hang
      report_fault(job, FAULT_DEBLOCK_PROGRESS);
      seen = need;
    }
    __asm__ volatile("" ::: "memory");
  };

  // Chain: the row's samples come from this same launch, so before anything of macroblocks <= m of the row is read:
  //  * the inter runs of the row up to m's have landed;
  //  * the row's intra stage is past m (its intra macroblocks in front of m + 1 have landed) - and so is the intra stage
  //    of the row BELOW, by one macroblock more: intra prediction reads UNFILTERED neighbours, top-right included, so
  //    this row may not be touched where the row below still has to predict from it.
  uint64_t seen_inter = 0;
  uint32_t seen_intra_me = 0, seen_intra_below = 0;
  const bool chain_inter = CHAIN && job.do_inter, chain_intra = CHAIN && job.do_intra;
  auto chain_need = [&](int m) {
    if constexpr (CHAIN) {
      if (m < 0) return;
      m = min(m, W - 1);
      bool ok = true;
      if (chain_inter) ok = chain_need_inter(job.chain_ctl, ctl, mby, m, seen_inter);
      if (!ok) report_fault(job, FAULT_CHAIN_INTER);
      if (chain_intra) {
        // (m + 2 in both rows: the intra macroblock to the right of m predicts from m's column too - top-left in the
        // row below, and in THIS row from the row above, whose last lines this row's top edges change)
        ok = chain_need_intra(job.chain_ctl, ctl, epoch32, mby, static_cast<uint32_t>(min(m + 2, W)), seen_intra_me);
        if (ok && has_row_below) ok = chain_need_intra(job.chain_ctl, ctl, epoch32, mby + 1, static_cast<uint32_t>(min(m + 2, W)), seen_intra_below);
        if (!ok) report_fault(job, FAULT_CHAIN_INTRA);
      }
    }
  };
  // (chain: what the rows of this launch read of each other's progress per ROW, and what the pictures predicted from
  // this one follow)
  auto publish_fin = [&](int v) {
    if constexpr (CHAIN) {
      if (lane == 0) agent_st64(&ctl->fin[mby], epoch | static_cast<uint32_t>(v));
    }
  };
  {
    uint8_t* rowy = cur.y + (size_t)(mby * 16) * yp;
    uint8_t* rowu = cur.u + (size_t)(mby * 8) * cp;
    uint8_t* rowv = cur.v + (size_t)(mby * 8) * cp;
    const DbInfo* rowdb = dbs + mby * W;
    int tile_mbx = -2;  // macroblock whose samples the tile before the current one holds (its last four columns are the left neighbours)
    int ts = 0;         // which of the two tiles the current macroblock uses
    // ---- what a lane is, fixed for the row.  As a ROW of the macroblock (vertical edges, stores): lanes 0..15 luma
    // rows, 16..31 chroma rows of both planes; as a COLUMN (horizontal edges): lanes 0..15 luma columns, 16..31
    // chroma columns, lanes 32..63 the same again (they take the second edge of a pair).
    const int l32 = lane & 31;
    const bool v_chroma = l32 >= 16, h_chroma = l32 >= 16;
    const int c_plane = (l32 - 16) >> 3, c_idx = (l32 - 16) & 7;   // chroma lanes: plane, row or column
    const int v_off = v_chroma ? (1 + c_plane) * kTileBytes + (c_idx < 2 ? c_idx + 4 : c_idx + 8) * TR : (4 + l32) * TR;
    const int h_off = h_chroma ? (1 + c_plane) * kTileBytes + (c_idx < 4 ? 4 + c_idx : 8 + c_idx) : 4 + l32;
    uint8_t* const v_grow = v_chroma ? (c_plane ? rowv : rowu) + c_idx * cp : rowy + l32 * yp;  // the row's first sample in the picture
    const int v_xs = v_chroma ? 3 : 4;                                 // log2 of a macroblock's width in that plane
    // the band below reads the last rows of this one through agent-scope loads: a band's last row stores everything
    // that way (which rows they are is lane data; one kind of store per wave keeps the store phase half as long)
    const bool far = CHAIN || exporting;
    // the four strengths of a lane's edge segment are one dword of the record; the plane's thresholds another
    const int kk = l32 < 16 ? (l32 >> 2) : (((l32 - 16) & 7) >> 1);
    const int pl = l32 < 16 ? 0 : 1 + ((l32 - 16) >> 3);

    for (int mbx0 = 0; mbx0 < W; mbx0 += 64) {
      // Macroblocks without a single edge to filter are skipped outright: they
      // load nothing, store nothing; the counter simply jumps over them.
      const int probe = mbx0 + lane;
      unsigned long long todo = __ballot(probe < W && gld8(&rowdb[probe].any) != 0);
      // "Macroblocks < P of this row are final" - the value is normally raised one macroblock late, in the iteration of
      // the next macroblock that has edges (its stores have had that long to land).  Where that next one is further
      // away the rows below would wait for nothing: a macroblock without an edge of its own is final as soon as its
      // neighbours' edges are - all but its last columns, which the next filtered macroblock's left edge may still
      // touch.  So before a gap the counter is raised at once to (next macroblock with edges) - 1.
      auto publish_db = [&](int v) {
        // (chain: "macroblocks < v are final" is also what the row below and the pictures predicted from this one go
        // by for macroblocks this row never touches - a stretch without edges is final once its SAMPLES have landed)
        if (v > 0) chain_need(v - 1);
        if (exporting) { if (lane == 0) agent_st64(&ctl->db_prog[band], epoch_out | static_cast<uint32_t>(v)); }
        else progress[wave + 1] = v;
        publish_fin(v);
      };
      {
        const int first = todo ? mbx0 + __builtin_ctzll(todo) : (mbx0 + 64 < W ? mbx0 + 64 : W + 1);
        if (first - 1 > mbx0) {  // (what lies before this stretch is final: the previous one's last iteration saw to it)
          if (mbx0 > 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          publish_db(first - 1 < W ? first - 1 : W);
        }
      }
      // Interior samples arrive a GROUP of four macroblocks at a time: 64 bytes of every luma row (lane = row x
      // macroblock of the group, 16 bytes each) and 32 bytes of every chroma row (lane = plane x row x macroblock,
      // 8 bytes each): one 16-byte and one 8-byte load per lane every fourth macroblock instead of two dword loads
      // per macroblock (1 % off the kernel; the memory-side traffic did not change - what did was not re-reading the
      // left neighbour's columns, below).  g_* is the group being filtered, n_g* the next one with anything to filter,
      // in flight.
      u32x4 g_luma = 0, n_gluma = 0;
      u32x2 g_chroma = 0, n_gchroma = 0;
      int gcur = -1, gnext = -1;
      auto prefetch_group = [&](int g) {
        chain_need(4 * g + 3);
        const int mb = min(4 * g + (lane & 3), W - 1);
        if constexpr (CHAIN) {  // (written by this launch: past the caches)
          const uint64_t* ly = reinterpret_cast<const uint64_t*>(rowy + mb * 16 + (lane >> 2) * yp);
          const uint64_t l0 = agent_ld64(ly), l1 = agent_ld64(ly + 1);
          const uint64_t c0 = agent_ld64(reinterpret_cast<const uint64_t*>(((lane >> 5) ? rowv : rowu) + mb * 8 + ((lane >> 2) & 7) * cp));
          n_gluma.x = (uint32_t)l0;
          n_gluma.y = (uint32_t)(l0 >> 32);
          n_gluma.z = (uint32_t)l1;
          n_gluma.w = (uint32_t)(l1 >> 32);
          n_gchroma.x = (uint32_t)c0;
          n_gchroma.y = (uint32_t)(c0 >> 32);
        } else {
          n_gluma = *(const HW_AS1 u32x4*)(rowy + mb * 16 + (lane >> 2) * yp);
          n_gchroma = *(const HW_AS1 u32x2*)(((lane >> 5) ? rowv : rowu) + mb * 8 + ((lane >> 2) & 7) * cp);
        }
        gnext = g;
      };
      // first group beyond macroblock `mbx` (of this 64-macroblock stretch) that has anything to filter, or -1
      auto group_after = [&](int mbx, unsigned long long left) {
        const int rel_group = (mbx - mbx0) >> 2;
        const unsigned long long beyond = rel_group >= 15 ? 0ull : left & ~((1ull << (4 * (rel_group + 1))) - 1);
        return beyond ? (mbx0 + __builtin_ctzll(beyond)) >> 2 : -1;
      };
      // per macroblock: its deblocking record and, in case the left neighbour was skipped, that one's last columns
      uint32_t n_db = 0, n_left = 0;
      auto prefetch = [&](int mbx, bool left_in_tile) {
        chain_need(mbx);
        const uint8_t* gy = rowy + mbx * 16;
        if (lane < 24) n_db = gld32(reinterpret_cast<const uint32_t*>(rowdb + mbx) + lane);
        // the four columns to the left when the left neighbour was skipped (otherwise they slide over inside the
        // tile): lanes 0..15 luma rows, 16..31 chroma rows of both planes
        if (mbx > 0 && !left_in_tile) {
          if (lane < 16) n_left = sld32<CHAIN>(gy + lane * yp - 4);
          else if (lane < 32) n_left = sld32<CHAIN>((((lane - 16) >> 3) ? rowv : rowu) + mbx * 8 + ((lane - 16) & 7) * cp - 4);
        }
      };
      if (todo) {
        prefetch(mbx0 + __builtin_ctzll(todo), tile_mbx == mbx0 + __builtin_ctzll(todo) - 1);
        prefetch_group((mbx0 + __builtin_ctzll(todo)) >> 2);
      }
      // (complete before the loop, so that inside it the registers count as ready - see below)
      asm volatile("" : "+v"(n_gluma), "+v"(n_gchroma), "+v"(n_db), "+v"(n_left));

      while (todo) {
        const int mbx = mbx0 + __builtin_ctzll(todo);
        todo &= todo - 1;
#ifdef EXP_STAMPS
        const bool sd_on = EXP_STAMPS == 2 && !CHAIN && mby == EXP_STAMP_ROW;
#endif
        STAMP(sd_on, mbx, 0);
        uint8_t* gy = rowy + mbx * 16;
        uint8_t* gu = rowu + mbx * 8;
        uint8_t* gv = rowv + mbx * 8;
        // ---- land the prefetched interior + record, start the next prefetches
        uint8_t* T = L.tile[ts];
        const uint8_t* Tprev = L.tile[ts ^ 1];
        const bool left_in_tile = tile_mbx == mbx - 1;
        if ((mbx >> 2) != gcur) {  // a new group: what was in flight becomes current, the group after it is requested
          g_luma = n_gluma;
          g_chroma = n_gchroma;
          gcur = gnext;
          const int g = group_after(mbx, todo);
          if (g >= 0) prefetch_group(g);
        }
        if ((lane & 3) == (mbx & 3)) {  // the lanes that hold this macroblock's part of the group
          uint32_t* lrow = reinterpret_cast<uint32_t*>(T + (4 + (lane >> 2)) * TR + 4);
          lrow[0] = g_luma.x;
          lrow[1] = g_luma.y;
          lrow[2] = g_luma.z;
          lrow[3] = g_luma.w;
          const int j = (lane >> 2) & 7;
          uint32_t* crow = reinterpret_cast<uint32_t*>(T + (1 + (lane >> 5)) * kTileBytes + (j < 2 ? j + 4 : j + 8) * TR + 4);
          crow[0] = g_chroma.x;
          crow[2] = g_chroma.y;
        }
        if (lane < 24) reinterpret_cast<uint32_t*>(&L.db)[lane] = n_db;
        const uint32_t left_far = n_left;  // (the left neighbour's columns from memory, when it was skipped)
        if (todo) prefetch(mbx0 + __builtin_ctzll(todo), mbx0 + __builtin_ctzll(todo) == mbx + 1);
        wave_sync();
        STAMP(sd_on, mbx, 1);
        // ---- this lane's parameters for all eight edge steps, in registers
        const int emask = L.db.emask;
        const bool top_edge = mby > 0 && (emask & 0x10);
        const uint32_t vbs = *reinterpret_cast<const uint32_t*>(L.db.bs[0][kk]);
        const uint32_t hbs = *reinterpret_cast<const uint32_t*>(L.db.bs[1][kk]);
        const uint32_t a3 = *reinterpret_cast<const uint32_t*>(L.db.alpha[pl]);
        const uint32_t b3 = *reinterpret_cast<const uint32_t*>(L.db.beta[pl]);
        uint32_t tcw[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) tcw[c] = *reinterpret_cast<const uint32_t*>(L.db.tc0[pl][c]);
        // (bS = 4 exists on macroblock edges of intra macroblocks only: the formulas for it are skipped where no lane has it)
        // (and per edge: the macroblock edge - byte 0 of the strengths - is where an intra macroblock has it, its
        // inner edges have bS 3: three of an intra macroblock's four steps per direction skip the formulas)
        const bool strong_outer = __ballot(((vbs | hbs) & 0x00000004u) != 0) != 0;
        const bool strong_inner = __ballot(((vbs | hbs) & 0x04040400u) != 0) != 0;
        const bool some_strong = strong_outer || strong_inner;
        // ---- rows above: wait for the row above to be two macroblocks ahead
        uint32_t top_word = 0;
        if (top_edge) {
          const uint8_t* src = nullptr;
          if (lane < 16) src = gy + ((lane >> 2) - 4) * yp + (lane & 3) * 4;
          else if (lane < 24) {
            const int t = lane - 16, p = t >> 2, cr = (t >> 1) & 1, cc = t & 1;
            src = (p ? gv : gu) + (cr - 2) * cp + cc * 4;
          }
          if (importing) {
            import_wait(min(mbx + 1, W));
            if (lane < 24) top_word = agent_ld32(src);
          } else {
            row_wait(progress, wave, min(mbx + 1, W));
            if (lane < 24) top_word = sld32<CHAIN>(src);
          }
        }
        STAMP(sd_on, mbx, 2);
        // ---- vertical edges, in registers: lanes 0..15 one luma row each, 16..31 one chroma row each.  The row's dwords
        // R0 (the left neighbour's last four columns) R1 .. R4 (chroma: R1 and R3); edge e lies between R[e] and R[e + 1],
        // the chroma edges between R0 | R1 and R1 | R3.
        // (lanes 32..63 hold the same rows as 0..31: they take the second edge of a pair, below)
        uint32_t R0, R1, R2, R3, R4;
        {
          const uint32_t* row = reinterpret_cast<const uint32_t*>(T + v_off);
          R1 = row[1];
          R2 = row[2];
          R3 = row[3];
          R4 = row[4];
          if (left_in_tile) R0 = *reinterpret_cast<const uint32_t*>(Tprev + v_off + (v_chroma ? 12 : 16));
          else R0 = left_far;  // (lanes 0..31; the twins never use it)
        }
        const uint32_t O0 = R0, O1 = R1, O2 = R2, O3 = R3, O4 = R4;  // as they are in memory
        if (!(emask & 0xa) && (emask & 0x5)) {
          // Edges 0 and 2 touch disjoint samples, so when the odd edges are not filtered - every macroblock with the
          // 8x8 transform, and many without - both run in ONE step, the second on lanes 32..63, whose results come
          // back over v_permlane32_swap.  (Chroma: the two edges share the dword between them; edge 0 changes its first
          // sample, edge 2 its last.)
          const bool second = lane >= 32;
          const int e = second ? 2 : 0, cls = second ? 2 : 0;
          int bs = (vbs >> (8 * e)) & 0xff;
          if (!((emask >> e) & 1)) bs = 0;
          const int alpha = (a3 >> (8 * cls)) & 0xff, beta = (b3 >> (8 * cls)) & 0xff;
          const int tc0 = bs && bs < 4 ? ((second ? tcw[2] : tcw[0]) >> (8 * (bs - 1))) & 0xff : 0;
          uint32_t lo = second ? (v_chroma ? R1 : R2) : R0, hi = second ? R3 : R1;
          filter_dwords(lo, hi, bs, alpha, beta, tc0, v_chroma, some_strong);
          const uint32_t lo2 = __builtin_amdgcn_permlane32_swap(lo, lo, false, false)[1];  // every lane: the value of its twin in lanes 32..63
          const uint32_t hi2 = __builtin_amdgcn_permlane32_swap(hi, hi, false, false)[1];
          if (!second) {
            R0 = lo;
            R1 = v_chroma ? (hi & 0x0000ffffu) | (lo2 & 0xffff0000u) : hi;
            R2 = v_chroma ? R2 : lo2;
            R3 = hi2;
            uint32_t* row = reinterpret_cast<uint32_t*>(T + v_off);
            row[1] = R1;
            row[2] = R2;
            row[3] = R3;
            if ((emask & 1) && mbx > 0 && R0 != O0) {
              uint8_t* g = v_grow + (mbx << v_xs) - 4;
              if (far) agent_st32(g, R0);
              else gst32(g, R0);
            }
          }
        } else if (emask & 0xf) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            if (!((emask >> e) & 1)) continue;  // wave uniform: nothing on this edge
            const int cls = e == 0 ? 0 : 2;
            int bs = (vbs >> (8 * e)) & 0xff;
            if (lane >= 32 || (v_chroma && (e & 1))) bs = 0;
            const int alpha = (a3 >> (8 * cls)) & 0xff, beta = (b3 >> (8 * cls)) & 0xff;
            const int tc0 = bs && bs < 4 ? (tcw[cls] >> (8 * (bs - 1))) & 0xff : 0;
            if (e == 0) filter_dwords(R0, R1, bs, alpha, beta, tc0, v_chroma, strong_outer);
            else if (e == 1) filter_dwords(R1, R2, bs, alpha, beta, tc0, v_chroma, strong_inner);
            else if (e == 2) {
              uint32_t lo = v_chroma ? R1 : R2;
              filter_dwords(lo, R3, bs, alpha, beta, tc0, v_chroma, strong_inner);
              R1 = v_chroma ? lo : R1;
              R2 = v_chroma ? R2 : lo;
            } else filter_dwords(R3, R4, bs, alpha, beta, tc0, v_chroma, strong_inner);
          }
          if (lane < 32) {
            uint32_t* row = reinterpret_cast<uint32_t*>(T + v_off);
            row[1] = R1;
            row[2] = R2;
            row[3] = R3;
            row[4] = R4;
          }
          // ---- the left neighbour's last four columns are final now (only this macroblock's left edge touches them
          // from this row): store them at once, so that by the time this macroblock's progress is published - after
          // its horizontal edges, below - the left neighbour is final IN MEMORY and the row below may filter under it.
          if ((emask & 1) && mbx > 0 && lane < 32 && R0 != O0) {
            uint8_t* g = v_grow + (mbx << v_xs) - 4;
            if (far) agent_st32(g, R0);
            else gst32(g, R0);
          }
        }
        STAMP(sd_on, mbx, 3);
        // ---- the rows above arrive now (their latency hid behind the vertical edges)
        if (top_edge) {
          if (lane < 16) {
            *reinterpret_cast<uint32_t*>(T + (lane >> 2) * TR + 4 + (lane & 3) * 4) = top_word;
          } else if (lane < 24) {
            const int t = lane - 16, p = t >> 2, cr = (t >> 1) & 1, cc = t & 1;
            *reinterpret_cast<uint32_t*>(T + (1 + p) * kTileBytes + (2 + cr) * TR + (cc ? 12 : 4)) = top_word;
          }
        }
        wave_sync();
        STAMP(sd_on, mbx, 4);
        // ---- horizontal edges, in registers: lanes 0..15 one luma column each, 16..31 chroma columns.  Edges 0 and 2
        // touch disjoint rows, so when the odd edges are not filtered - every macroblock with the 8x8 transform, and
        // many without - both run in ONE step, the second on lanes 32..63.
        const int hm = ((emask >> 4) & 0xf) & (top_edge ? 0xf : 0xe);
        if (!(hm & 0xa) && (hm & 0x5)) {
          const int e = (lane >> 5) * 2;
          const int cls = e == 0 ? 1 : 2;
          int bs = (hbs >> (8 * e)) & 0xff;
          if (!((hm >> e) & 1)) bs = 0;
          const int alpha = (a3 >> (8 * cls)) & 0xff, beta = (b3 >> (8 * cls)) & 0xff;
          const int tc0 = bs && bs < 4 ? (tcw[cls] >> (8 * (bs - 1))) & 0xff : 0;
          uint8_t* col = T + h_off + (lane >> 5) * 8 * TR;
          EdgeLine v;
          v.p3 = col[0];
          v.p2 = col[TR];
          v.p1 = col[2 * TR];
          v.p0 = col[3 * TR];
          v.q0 = col[4 * TR];
          v.q1 = col[5 * TR];
          v.q2 = col[6 * TR];
          v.q3 = col[7 * TR];
          filter_line(v, bs, alpha, beta, tc0, h_chroma, some_strong);
          if (bs) {
            col[TR] = static_cast<uint8_t>(v.p2);
            col[2 * TR] = static_cast<uint8_t>(v.p1);
            col[3 * TR] = static_cast<uint8_t>(v.p0);
            col[4 * TR] = static_cast<uint8_t>(v.q0);
            col[5 * TR] = static_cast<uint8_t>(v.q1);
            col[6 * TR] = static_cast<uint8_t>(v.q2);
          }
        } else if (hm) {
          uint8_t* col = T + h_off;
          int v[20];
#pragma unroll
          for (int i = 0; i < 20; ++i) v[i] = col[i * TR];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            if (!((hm >> e) & 1)) continue;
            const int cls = e == 0 ? 1 : 2;
            int bs = (hbs >> (8 * e)) & 0xff;
            if (lane >= 32 || (h_chroma && (e & 1))) bs = 0;
            const int alpha = (a3 >> (8 * cls)) & 0xff, beta = (b3 >> (8 * cls)) & 0xff;
            const int tc0 = bs && bs < 4 ? (tcw[cls] >> (8 * (bs - 1))) & 0xff : 0;
            EdgeLine l;
            l.p3 = v[4 * e];
            l.p2 = v[4 * e + 1];
            l.p1 = v[4 * e + 2];
            l.p0 = v[4 * e + 3];
            l.q0 = v[4 * e + 4];
            l.q1 = v[4 * e + 5];
            l.q2 = v[4 * e + 6];
            l.q3 = v[4 * e + 7];
            filter_line(l, bs, alpha, beta, tc0, h_chroma, e == 0 ? strong_outer : strong_inner);
            v[4 * e + 1] = l.p2;
            v[4 * e + 2] = l.p1;
            v[4 * e + 3] = l.p0;
            v[4 * e + 4] = l.q0;
            v[4 * e + 5] = l.q1;
            v[4 * e + 6] = l.q2;
          }
          if (lane < 32) {
#pragma unroll
            for (int i = 1; i < 19; ++i) col[i * TR] = static_cast<uint8_t>(v[i]);
          }
        }
        wave_sync();
        // ---- the previous filtered macroblock's stores have had this whole
        // iteration to land: make sure, then everything before this macroblock
        // is final (skipped ones included)
        STAMP(sd_on, mbx, 5);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        STAMP(sd_on, mbx, 6);
        STAMP_VALUE(sd_on, mbx, emask | (top_edge ? 0x100 : 0));
        if (exporting) { if (lane == 0) agent_st64(&ctl->db_prog[band], epoch_out | static_cast<uint32_t>(mbx)); }
        else progress[wave + 1] = mbx;
        publish_fin(mbx);
        // The prefetched registers are complete too; saying so here keeps the
        // compiler from waiting for them at the top of the next iteration, where
        // its in-order counter would also cover the stores issued just below.
        asm volatile("" : "+v"(n_gluma), "+v"(n_gchroma), "+v"(n_db), "+v"(n_left));
        // ---- store what the filters changed: the row lanes compare their row with what it was in memory (a P
        // picture's filtered macroblocks change a fraction of their samples; writing all of them back cost 4.5 MB of
        // HBM writes per 3.1 MB picture, and fetches of the partly written lines), and the lanes that fetched the rows
        // above theirs.  The corner above-left belongs to other macroblocks' edges and is never written from here.
        if (lane < 32) {
          const uint32_t* row = reinterpret_cast<const uint32_t*>(T + v_off);
          const uint32_t F1 = row[1], F2 = row[2], F3 = row[3], F4 = row[4];
          uint8_t* g = v_grow + (mbx << v_xs);
          if (F1 != O1) { if (far) agent_st32(g, F1); else gst32(g, F1); }
          if (F3 != O3) { if (far) agent_st32(g + (v_chroma ? 4 : 8), F3); else gst32(g + (v_chroma ? 4 : 8), F3); }
          if (!v_chroma) {
            if (F2 != O2) { if (far) agent_st32(g + 4, F2); else gst32(g + 4, F2); }
            if (F4 != O4) { if (far) agent_st32(g + 12, F4); else gst32(g + 12, F4); }
          }
        }
        if (top_edge && lane < 24) {
          uint32_t f;
          uint8_t* g;
          if (lane < 16) {
            f = *reinterpret_cast<const uint32_t*>(T + (lane >> 2) * TR + 4 + (lane & 3) * 4);
            g = gy + ((lane >> 2) - 4) * yp + (lane & 3) * 4;
          } else {
            const int t = lane - 16, p = t >> 2, cr = (t >> 1) & 1, cc = t & 1;
            f = *reinterpret_cast<const uint32_t*>(T + (1 + p) * kTileBytes + (2 + cr) * TR + (cc ? 12 : 4));
            g = (p ? gv : gu) + (cr - 2) * cp + cc * 4;
          }
          if (f != top_word) {
            if (CHAIN) agent_st32(g, f);
            else gst32(g, f);
          }
        }
        tile_mbx = mbx;
        ts ^= 1;
        {
          // a gap behind this macroblock: let its stores land now and raise the counter over the gap
          const int next = todo ? mbx0 + __builtin_ctzll(todo) : (mbx0 + 64 < W ? mbx0 + 64 : W + 1);
          if (next - 1 > mbx + 1) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            publish_db(next - 1 < W ? next - 1 : W);
            asm volatile("" : "+v"(n_gluma), "+v"(n_gchroma), "+v"(n_db), "+v"(n_left));  // (they have landed too)
          }
        }
      }
    }
    chain_need(W - 1);  // (a row that ends in a stretch without edges: see publish_db)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (exporting) {
      // (the workgroup-scope fence does not wait for outstanding stores; the band below must not see the
      // counter before the last macroblock's samples)
      __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (lane == 0) agent_st64(&ctl->db_prog[band], epoch_out | static_cast<uint32_t>(W));
    } else {
      if constexpr (CHAIN) __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");
      progress[wave + 1] = W;
    }
    publish_fin(W);
  }
}


__global__ __launch_bounds__(kBandWaves * 64) void k_deblock(JobDesc* __restrict__ jobs, int njobs) {
  // One workgroup per band of kBandWaves rows, bands handed out by ticket (see
  // k_mv).  Rows hand over through the picture itself: inside a band the waves
  // share the CU's L1 and an LDS counter; across bands the last row of a band
  // writes its bottom rows with agent-scope stores and publishes its counter in
  // BandCtl::db_prog once they have landed, and the first row of the next band
  // polls that counter and reads the rows above with agent-scope loads.
  __shared__ unsigned s_ticket;
  __shared__ DeblockGroupLds s_group;
  if (threadIdx.x == 0) s_ticket = atomicAdd(&jobs[0].ticket_db, 1u);
  if (threadIdx.x <= kBandWaves) s_group.progress[threadIdx.x] = 0;
  __syncthreads();
  const int band = static_cast<int>(s_ticket) / njobs;
  const JobDesc& job = jobs[static_cast<int>(s_ticket) % njobs];
  if (!job.do_deblock) return;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  deblock_band<false>(job, band, wave, lane, s_group);
}

// ===========================================================================
// k_upload: the batch's host -> HBM transfers as ONE launch.
//
// Each picture brings up to four arrays (MbInfo, coefficients, SliceParams,
// MbMotion) from its lane's pinned staging buffers.  As separate
// hipMemcpyAsync calls they cost tens of microseconds each on the copy queue,
// so the kernel pulls them over PCIe itself: grid.y = picture * 4 + array,
// grid.x strides over 16-byte chunks, every wave has a full cache line in
// flight per load.  The job table (which carries the UploadDesc) is the only
// hipMemcpyAsync left per batch.
// ===========================================================================
constexpr int kUploadBlocks = 48;

__global__ __launch_bounds__(256) void k_upload(const JobDesc* __restrict__ jobs) {
  const JobDesc& job = jobs[blockIdx.y / kJobUploads];
  const int a = blockIdx.y % kJobUploads;
  if (blockIdx.x == 1 && a == 0 && job.band_ctl) {
    // (chain launches: the rows' inter-run masks start from zero with every picture - also for a picture that brings
    // nothing to upload: GpuExecutor::replay)
    uint64_t* mask = static_cast<BandCtl*>(job.band_ctl)->inter_mask;
    for (int r = threadIdx.x; r < job.pp.mb_height; r += 256) {
      gst32(reinterpret_cast<uint32_t*>(mask + r), 0u);
      gst32(reinterpret_cast<uint32_t*>(mask + r) + 1, 0u);
    }
  }
  if (a >= job.num_uploads) return;
  const UploadDesc u = job.uploads[a];
  const uint64_t n16 = u.bytes >> 4;
  typedef unsigned int v4u __attribute__((ext_vector_type(4)));
  const HW_AS1 v4u* __restrict__ src = (const HW_AS1 v4u*)u.src;
  HW_AS1 v4u* __restrict__ dst = (HW_AS1 v4u*)u.dst;
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (uint64_t)kUploadBlocks * 256)
    dst[i] = __builtin_nontemporal_load(&src[i]);
  if (blockIdx.x == 0) {
    const uint64_t tail = u.bytes & 15;
    if (threadIdx.x < tail)
      static_cast<uint8_t*>(u.dst)[(n16 << 4) + threadIdx.x] = static_cast<const uint8_t*>(u.src)[(n16 << 4) + threadIdx.x];
  }
}

// End-of-batch marker: one lane stores the batch's ticket into pinned host
// memory (system scope).  Stream order puts it behind everything the batch
// enqueued, copies included; the executor's poller thread reads it.
__global__ void k_done(uint32_t* flag, uint32_t ticket) {
  __hip_atomic_store(flag, ticket, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

hipError_t launch_done(uint32_t* flag, uint32_t ticket, hipStream_t stream) {
  hipLaunchKernelGGL(k_done, dim3(1), dim3(1), 0, stream, flag, ticket);
  return hipGetLastError();
}

// A descriptor table from pinned host memory to its device copy (a multiple of
// 16 bytes).  A kernel instead of hipMemcpyAsync: the copy needs no completion
// bookkeeping on the host, and it is one call path less into the runtime.
__global__ __launch_bounds__(256) void k_table(void* __restrict__ dst_, const void* __restrict__ src_, uint32_t n16) {
  typedef unsigned int v4u __attribute__((ext_vector_type(4)));
  const HW_AS1 v4u* __restrict__ src = (const HW_AS1 v4u*)src_;
  HW_AS1 v4u* __restrict__ dst = (HW_AS1 v4u*)dst_;
  for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n16; i += gridDim.x * 256) dst[i] = __builtin_nontemporal_load(&src[i]);
}
hipError_t launch_table_copy(void* dst, const void* src, size_t bytes, hipStream_t stream) {
  const uint32_t n16 = static_cast<uint32_t>((bytes + 15) >> 4);
  hipLaunchKernelGGL(k_table, dim3((n16 + 255) / 256 < 16 ? (n16 + 255) / 256 : 16), dim3(256), 0, stream, dst, src, n16);
  return hipGetLastError();
}

hipError_t launch_upload(const JobDesc* jobs, int njobs, hipStream_t stream) {
  hipLaunchKernelGGL(k_upload, dim3(kUploadBlocks, njobs * kJobUploads), dim3(256), 0, stream, jobs);
  return hipGetLastError();
}

// ===========================================================================
// k_chain: the sample stages of a batch - k_inter, k_intra, k_deblock - as ONE launch in which dependent pictures
// overlap.  The three kernels above are separated by kernel boundaries, so a lane's pictures (each predicted from the
// one before) cost the full latency of three wavefronts EACH: ~0.9 ms per 1080p P picture, and the reference's own
// caller (one DecoderAutomata, one interval at a time: python/hwang/decoder.py:60-67) sits through a GOP's worth of
// them per returned frame.  Here a picture's stages and the pictures of a chain follow each other row by row:
//   inter macroblock of picture n + 1   waits for picture n's deblocking to have left final what its vectors reach
//                                       (BandCtl::fin, per macroblock row; chain_wait_reference)
//   intra macroblock of picture n       waits for the inter runs that hold its neighbours (BandCtl::inter_mask: a bit
//                                       per run of kChainRun macroblocks of a row)
//   deblocking of picture n, row r      reads macroblocks <= m once the row's inter runs up to m's have landed and the
//                                       intra stage of rows r and r + 1 is past m (BandCtl::intra_prog; the row below
//                                       because intra prediction reads UNFILTERED neighbours)
// so picture n + 1 trails picture n by a few macroblocks and rows instead of by a picture.
//
// No kernel boundary orders anything here, so (1) every sample these stages hand to each other is stored and loaded
// agent-scope (CHAIN = true instantiations: write-through stores, loads past L1 / L2, every counter raised behind
// s_waitcnt vmcnt(0)); references that were final before the launch are read the cached way.  (2) Nothing may wait for
// work that is not running.  HIP promises no dispatch order, so work is handed out by ticket from two lists the host
// builds in dependency order (position by position of the lanes' chains; within a picture: inter runs top to bottom,
// then intra bands, then deblocking bands).  A workgroup is four ROW waves (one band at a time, as k_intra / k_deblock)
// plus four INTER waves (each takes runs of kChainRun macroblocks on its own).  The smallest unfinished row unit and
// the smallest unfinished inter unit are always held by resident waves; a row unit depends on row units before it and
// on inter units of its own picture, an inter unit on row units of EARLIER pictures: of the two smallest, one can
// always run - whatever the residency, with no assumption about other launches on the device.  Polls are bounded
// anyway and report to the batch's fault word.
// ===========================================================================
namespace {
constexpr int kChainInterWaves = 4;
struct ChainLds {
  union {
    IntraGroupLds intra;
    DeblockGroupLds deblock;
  } rows;
  IntraTables tables;
  InterWave inter[kChainInterWaves];
  // hand-over of a row unit to the band's four waves: wave 0 takes the ticket
  int generation;   // units handed out so far
  int arrived;      // row waves that are through with their part of the current unit
  uint32_t unit;
};
}  // namespace

__global__ __launch_bounds__((kBandWaves + kChainInterWaves) * 64) void k_chain(const ChainArgs args) {
  __shared__ ChainLds S;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  load_intra_tables(S.tables, threadIdx.x, (kBandWaves + kChainInterWaves) * 64);
  if (threadIdx.x == 0) {
    S.generation = 0;
    S.arrived = kBandWaves;  // "everybody is through with unit 0 - 1"
  }
  __syncthreads();  // (the only workgroup barrier: the two halves go their own ways from here)
  const JobDesc* __restrict__ jobs = args.jobs;

  if (wave >= kBandWaves) {
    // ---- inter waves
    InterWave& L = S.inter[wave - kBandWaves];
    for (;;) {
      uint32_t t = 0;
      if (lane == 0) t = __hip_atomic_fetch_add((HW_AS1 uint32_t*)(args.tickets + 1), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      t = __builtin_amdgcn_readfirstlane(t);
      if (t >= args.num_inter_units) {
        chain_dbg(8, 0, 0);
        break;
      }
      const uint32_t u = __builtin_amdgcn_readfirstlane(gld32(args.inter_units + t));
      const JobDesc& job = jobs[u >> 20];
      const InterPicture ip = inter_picture(job);
      chain_dbg(4, u >> 20, u & 0xfff);
      // runs are per row: run = row * runs_per_row + j, macroblocks 8 j .. 8 j + 7 of the row
      const int runs_per_row = (ip.mb_width + kChainRun - 1) / kChainRun;
      const int run = static_cast<int>(u & 0xfffffu), row = run / runs_per_row, j = run - row * runs_per_row;
      const int x0 = j * kChainRun, x1 = min(x0 + kChainRun, ip.mb_width);
      for (int x = x0; x < x1; ++x) {
        inter_mb<true>(jobs, job, ip, row * ip.mb_width + x, lane, L);
        wave_sync();  // (the next macroblock reuses the wave's LDS)
      }
      // the run's samples have landed (they were stored write-through): say so
      __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");
      // (test hook, HWANG_INJECT_FAULT=2: one run of the picture never reports - what a lost hand-off would look like)
      if (lane == 0 && !(job.inject_fault == 2 && row == ip.mb_height / 2 && j == 0))
        __hip_atomic_fetch_or((HW_AS1 uint64_t*)&static_cast<BandCtl*>(job.band_ctl)->inter_mask[row], 1ull << j, __ATOMIC_RELAXED,
                              __HIP_MEMORY_SCOPE_AGENT);
      if (lane == 0) chain_beat(job.chain_ctl);
    }
    return;
  }

  // ---- row waves
  lds_counter* generation = (lds_counter*)&S.generation;
  lds_counter* arrived = (lds_counter*)&S.arrived;
  for (int gen = 0;; ++gen) {
    if (wave == 0) {
      // everybody is through with the previous unit: take the next one, reset the band's counters, announce it
      while (*arrived < kBandWaves * (gen + 1)) __builtin_amdgcn_s_sleep(1);
      __asm__ volatile("" ::: "memory");
      uint32_t t = 0;
      if (lane == 0) t = __hip_atomic_fetch_add((HW_AS1 uint32_t*)args.tickets, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      t = __builtin_amdgcn_readfirstlane(t);
      const uint32_t u = t < args.num_row_units ? __builtin_amdgcn_readfirstlane(gld32(args.row_units + t)) : 0xffffffffu;
      if (lane == 0) {
        S.unit = u;
        for (int i = 0; i <= kBandWaves; ++i) S.rows.intra.progress[i] = 0;  // (the same words in both members)
      }
      __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (lane == 0) *generation = gen + 1;
    } else {
      while (*generation < gen + 1) __builtin_amdgcn_s_sleep(1);
      __asm__ volatile("" ::: "memory");
    }
    const uint32_t u = __builtin_amdgcn_readfirstlane(*(volatile __attribute__((address_space(3))) uint32_t*)&S.unit);
    if (u == 0xffffffffu) break;
    chain_dbg(5 + ((u >> 7) & 1), u >> 8, u & 0x7f);
    const JobDesc& job = jobs[u >> 8];
    const int band = static_cast<int>(u & 0x7f);
    if (u & 0x80) deblock_band<true>(job, band, wave, lane, S.rows.deblock);
    else intra_band<true>(job, band, wave, lane, S.rows.intra, S.tables);
    // through with this unit's LDS
    __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    chain_dbg(7, u >> 8, u & 0xff);
    if (lane == 0) {
      chain_beat(job.chain_ctl);  // (a row of the unit is through)
      __hip_atomic_fetch_add((__attribute__((address_space(3))) int*)&S.arrived, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
  }
  chain_dbg(9, 0, 0);
}

}  // namespace amd
}  // namespace hwang
#ifdef EXP_STAMPS
extern "C" int hwang_debug_stamps(uint64_t* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(hwang::amd::g_stamps), sizeof(uint64_t) * 256 * 8, 0, hipMemcpyDeviceToHost) == hipSuccess ? 0 : 1;
}
#endif
extern "C" int hwang_debug_chain_state(uint32_t* out, int n) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(hwang::amd::g_chain_dbg), sizeof(uint32_t) * (n < 2048 ? n : 2048), 0, hipMemcpyDeviceToHost) == hipSuccess ? 0 : 1;
}
namespace hwang {
namespace amd {
hipError_t launch_chain(const ChainArgs& args, int workgroups, hipStream_t stream) {
  const bool profiled = prof::enabled();
  prof::Span span;
  if (profiled) span = prof::begin(stream);
  hipLaunchKernelGGL(k_chain, dim3(workgroups), dim3((kBandWaves + kChainInterWaves) * 64), 0, stream, args);
  if (profiled) prof::end(span, stream, prof::K_CHAIN, args.bytes);
  return hipGetLastError();
}

// ===========================================================================
// launch helpers (host).  `jobs` is a device array of `njobs` descriptors;
// `max_mbs` the largest macroblock count among them.
// ===========================================================================
hipError_t launch_batch(const JobDesc* jobs, int njobs, int max_mbs, int max_rows, bool any_inter, bool any_intra,
                        bool any_deblock, bool any_mv, const uint64_t bytes[4], hipStream_t stream) {
  const bool profiled = prof::enabled();
  prof::Span span;
  if (any_mv) {
    if (profiled) span = prof::begin(stream);
    hipLaunchKernelGGL(k_mv, dim3(njobs * ((max_rows + kMvWaves - 1) / kMvWaves)), dim3((kMvWaves + 1) * 64), 0, stream,
                       const_cast<JobDesc*>(jobs), njobs);
    if (profiled) prof::end(span, stream, prof::K_MV, bytes[3]);
  }
  if (any_inter) {
    if (profiled) span = prof::begin(stream);
    hipLaunchKernelGGL(k_inter, dim3(((max_mbs + kInterWaves - 1) / kInterWaves + 7) / 8 * 8, njobs), dim3(kInterWaves * 64), 0,
                       stream, jobs);
    if (profiled) prof::end(span, stream, prof::K_INTER, bytes[0]);
  }
  if (any_intra) {
    if (profiled) span = prof::begin(stream);
    hipLaunchKernelGGL(k_residual, dim3((max_mbs + kInterWaves - 1) / kInterWaves, njobs), dim3(kInterWaves * 64), 0, stream,
                       jobs);
    hipLaunchKernelGGL(k_intra, dim3(njobs * ((max_rows + kBandWaves - 1) / kBandWaves)), dim3(kBandWaves * 64), 0, stream,
                       const_cast<JobDesc*>(jobs), njobs);
    if (profiled) prof::end(span, stream, prof::K_INTRA, bytes[1]);
  }
  if (any_deblock) {
    if (profiled) span = prof::begin(stream);
    hipLaunchKernelGGL(k_bs, dim3(((max_mbs + 7) / 8 + 7) / 8 * 8, njobs), dim3(256), 0, stream, jobs);
    hipLaunchKernelGGL(k_deblock, dim3(njobs * ((max_rows + kBandWaves - 1) / kBandWaves)), dim3(kBandWaves * 64), 0, stream,
                       const_cast<JobDesc*>(jobs), njobs);
    if (profiled) prof::end(span, stream, prof::K_DEBLOCK, bytes[2]);
  }
  return hipGetLastError();
}

// The same work as launch_batch in two stages.  What a picture's kernels need splits cleanly:
//   records only   k_mv (vectors from differences), k_bs (boundary strengths from records and vectors),
//                  k_residual (dequantisation + inverse transforms of intra macroblocks)
//   samples        k_inter -> k_intra -> k_deblock, each reading what the one before (and the reference pictures) wrote
// Only the second chain ties a picture to the one before it.  launch_pre runs the first for any set of pictures whose
// records have been uploaded - pictures of one lane included, far ahead of their turn; launch_main the second.
// bytes: [0] k_mv, [1] k_bs, [2] k_residual (algorithmic, for the spans).
hipError_t launch_pre(const JobDesc* jobs, int njobs, int max_mbs, int max_rows, bool any_mv, bool any_bs, bool any_residual,
                      const uint64_t bytes[3], hipStream_t stream) {
  const bool profiled = prof::enabled();
  prof::Span span;
  if (any_mv) {
    if (profiled) span = prof::begin(stream);
    hipLaunchKernelGGL(k_mv, dim3(njobs * ((max_rows + kMvWaves - 1) / kMvWaves)), dim3((kMvWaves + 1) * 64), 0, stream,
                       const_cast<JobDesc*>(jobs), njobs);
    if (profiled) prof::end(span, stream, prof::K_MV, bytes[0]);
  }
  if (any_bs) {
    if (profiled) span = prof::begin(stream);
    hipLaunchKernelGGL(k_bs, dim3(((max_mbs + 7) / 8 + 7) / 8 * 8, njobs), dim3(256), 0, stream, jobs);
    if (profiled) prof::end(span, stream, prof::K_BS, bytes[1]);
  }
  if (any_residual) {
    if (profiled) span = prof::begin(stream);
    hipLaunchKernelGGL(k_residual, dim3((max_mbs + kInterWaves - 1) / kInterWaves, njobs), dim3(kInterWaves * 64), 0, stream,
                       jobs);
    if (profiled) prof::end(span, stream, prof::K_RESIDUAL, bytes[2]);
  }
  return hipGetLastError();
}

// bytes: [0] k_inter, [1] k_intra, [2] k_deblock
hipError_t launch_main(const JobDesc* jobs, int njobs, int max_mbs, int max_rows, bool any_inter, bool any_intra,
                       bool any_deblock, const uint64_t bytes[3], hipStream_t stream) {
  const bool profiled = prof::enabled();
  prof::Span span;
  if (any_inter) {
    if (profiled) span = prof::begin(stream);
    hipLaunchKernelGGL(k_inter, dim3(((max_mbs + kInterWaves - 1) / kInterWaves + 7) / 8 * 8, njobs), dim3(kInterWaves * 64), 0,
                       stream, jobs);
    if (profiled) prof::end(span, stream, prof::K_INTER, bytes[0]);
  }
  if (any_intra) {
    if (profiled) span = prof::begin(stream);
    hipLaunchKernelGGL(k_intra, dim3(njobs * ((max_rows + kBandWaves - 1) / kBandWaves)), dim3(kBandWaves * 64), 0, stream,
                       const_cast<JobDesc*>(jobs), njobs);
    if (profiled) prof::end(span, stream, prof::K_INTRA, bytes[1]);
  }
  if (any_deblock) {
    if (profiled) span = prof::begin(stream);
    hipLaunchKernelGGL(k_deblock, dim3(njobs * ((max_rows + kBandWaves - 1) / kBandWaves)), dim3(kBandWaves * 64), 0, stream,
                       const_cast<JobDesc*>(jobs), njobs);
    if (profiled) prof::end(span, stream, prof::K_DEBLOCK, bytes[2]);
  }
  return hipGetLastError();
}

size_t deblock_scratch_bytes(int num_mbs) { return (size_t)num_mbs * sizeof(DbInfo); }
size_t band_ctl_bytes(int mb_height) {
  return offsetof(BandCtl, lines) +
         (size_t)((mb_height + kBandWaves - 1) / kBandWaves) * kLineMbs * (kMvLineWords + kIntraLineWords) * sizeof(uint64_t);
}

}  // namespace amd
}  // namespace hwang
