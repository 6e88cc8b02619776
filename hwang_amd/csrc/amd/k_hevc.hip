// HIP kernels of the HEVC path for gfx950 (wave64): the records of a BATCH of pictures (hevc_frame_job.h: up to 32 pictures
// of different decoders per launch, hevc_engine.cpp) -> samples.
//
//   k_hevc_clear      the words the intra kernels count in, zeroed.
//   k_hevc_residual   8.6.3 scaling + 8.6.4 inverse transforms (DCT 4..32, DST 4x4, transform skip, bypass): one
//                     wavefront per transform block, coefficients scattered into LDS, both matrix passes out of LDS,
//                     the residual written to the int16 residual planes (zero between pictures: readers clear them).
//   k_hevc_inter      8.5.3.3 fractional sample interpolation (8-tap luma, 4-tap chroma, references clamped to the
//                     picture), default and explicit weighting, + residual: one workgroup per prediction block, the
//                     reference window staged in LDS, separable passes; an eighth of a picture's blocks per XCD.
//   k_hevc_intra_*    8.4.4.2 intra prediction (substitution, [1 2 1] / strong smoothing, planar, DC, 33 angles, edge
//                     filters, I_PCM) + residual, one wavefront per block at a time.  _rows: a wavefront of coding tree
//                     block rows per colour plane, working out of an LDS tile (intra pictures); _flow: coding tree blocks
//                     taken by ticket, a block waiting for exactly the neighbouring units in OTHER coding tree blocks
//                     that it reads (the scattered intra blocks of P / B pictures).
//   k_hevc_deblock    8.7.2: boundary strengths from the blocks' properties and the prediction block records, then one
//                     thread per 4-sample edge segment, all vertical edges of the picture, then (a second launch) all
//                     horizontal ones - HEVC's filter is defined that way, nothing is serial in it.
//   k_hevc_sao        8.7.3: band / edge offsets from the deblocked picture into the DPB surface, four samples a thread,
//                     a band of rows per XCD.
// These are byte stencils bound by HBM and, for the intra kernels, by the dependent chain of blocks; no MFMA.
// What they compute is checked bit for bit against the CPU oracle and against the numpy statements of
// tests/hevc_numpy.py (tests/test_gpu_hevc.py, tests/test_gpu_hevc_kernels.py).  HEVC parity with FFmpeg is unpinned
// (DESIGN.md section 5a).
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "hevc_frame_job.h"
#include "hip_check.h"

namespace {

using namespace hwang::amd::hevc;

// --- tables (ITU-T H.265 8.6.4.2, Tables 8-4 / 8-5 / 8-11 / 8-12; tests/test_tables_match.py: same constants as hevc/hevc_tables.h)
__constant__ int8_t kTrans[32][32] = {
    {64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64},
    {90, 90, 88, 85, 82, 78, 73, 67, 61, 54, 46, 38, 31, 22, 13, 4, -4, -13, -22, -31, -38, -46, -54, -61, -67, -73, -78, -82, -85, -88, -90, -90},
    {90, 87, 80, 70, 57, 43, 25, 9, -9, -25, -43, -57, -70, -80, -87, -90, -90, -87, -80, -70, -57, -43, -25, -9, 9, 25, 43, 57, 70, 80, 87, 90},
    {90, 82, 67, 46, 22, -4, -31, -54, -73, -85, -90, -88, -78, -61, -38, -13, 13, 38, 61, 78, 88, 90, 85, 73, 54, 31, 4, -22, -46, -67, -82, -90},
    {89, 75, 50, 18, -18, -50, -75, -89, -89, -75, -50, -18, 18, 50, 75, 89, 89, 75, 50, 18, -18, -50, -75, -89, -89, -75, -50, -18, 18, 50, 75, 89},
    {88, 67, 31, -13, -54, -82, -90, -78, -46, -4, 38, 73, 90, 85, 61, 22, -22, -61, -85, -90, -73, -38, 4, 46, 78, 90, 82, 54, 13, -31, -67, -88},
    {87, 57, 9, -43, -80, -90, -70, -25, 25, 70, 90, 80, 43, -9, -57, -87, -87, -57, -9, 43, 80, 90, 70, 25, -25, -70, -90, -80, -43, 9, 57, 87},
    {85, 46, -13, -67, -90, -73, -22, 38, 82, 88, 54, -4, -61, -90, -78, -31, 31, 78, 90, 61, 4, -54, -88, -82, -38, 22, 73, 90, 67, 13, -46, -85},
    {83, 36, -36, -83, -83, -36, 36, 83, 83, 36, -36, -83, -83, -36, 36, 83, 83, 36, -36, -83, -83, -36, 36, 83, 83, 36, -36, -83, -83, -36, 36, 83},
    {82, 22, -54, -90, -61, 13, 78, 85, 31, -46, -90, -67, 4, 73, 88, 38, -38, -88, -73, -4, 67, 90, 46, -31, -85, -78, -13, 61, 90, 54, -22, -82},
    {80, 9, -70, -87, -25, 57, 90, 43, -43, -90, -57, 25, 87, 70, -9, -80, -80, -9, 70, 87, 25, -57, -90, -43, 43, 90, 57, -25, -87, -70, 9, 80},
    {78, -4, -82, -73, 13, 85, 67, -22, -88, -61, 31, 90, 54, -38, -90, -46, 46, 90, 38, -54, -90, -31, 61, 88, 22, -67, -85, -13, 73, 82, 4, -78},
    {75, -18, -89, -50, 50, 89, 18, -75, -75, 18, 89, 50, -50, -89, -18, 75, 75, -18, -89, -50, 50, 89, 18, -75, -75, 18, 89, 50, -50, -89, -18, 75},
    {73, -31, -90, -22, 78, 67, -38, -90, -13, 82, 61, -46, -88, -4, 85, 54, -54, -85, 4, 88, 46, -61, -82, 13, 90, 38, -67, -78, 22, 90, 31, -73},
    {70, -43, -87, 9, 90, 25, -80, -57, 57, 80, -25, -90, -9, 87, 43, -70, -70, 43, 87, -9, -90, -25, 80, 57, -57, -80, 25, 90, 9, -87, -43, 70},
    {67, -54, -78, 38, 85, -22, -90, 4, 90, 13, -88, -31, 82, 46, -73, -61, 61, 73, -46, -82, 31, 88, -13, -90, -4, 90, 22, -85, -38, 78, 54, -67},
    {64, -64, -64, 64, 64, -64, -64, 64, 64, -64, -64, 64, 64, -64, -64, 64, 64, -64, -64, 64, 64, -64, -64, 64, 64, -64, -64, 64, 64, -64, -64, 64},
    {61, -73, -46, 82, 31, -88, -13, 90, -4, -90, 22, 85, -38, -78, 54, 67, -67, -54, 78, 38, -85, -22, 90, 4, -90, 13, 88, -31, -82, 46, 73, -61},
    {57, -80, -25, 90, -9, -87, 43, 70, -70, -43, 87, 9, -90, 25, 80, -57, -57, 80, 25, -90, 9, 87, -43, -70, 70, 43, -87, -9, 90, -25, -80, 57},
    {54, -85, -4, 88, -46, -61, 82, 13, -90, 38, 67, -78, -22, 90, -31, -73, 73, 31, -90, 22, 78, -67, -38, 90, -13, -82, 61, 46, -88, 4, 85, -54},
    {50, -89, 18, 75, -75, -18, 89, -50, -50, 89, -18, -75, 75, 18, -89, 50, 50, -89, 18, 75, -75, -18, 89, -50, -50, 89, -18, -75, 75, 18, -89, 50},
    {46, -90, 38, 54, -90, 31, 61, -88, 22, 67, -85, 13, 73, -82, 4, 78, -78, -4, 82, -73, -13, 85, -67, -22, 88, -61, -31, 90, -54, -38, 90, -46},
    {43, -90, 57, 25, -87, 70, 9, -80, 80, -9, -70, 87, -25, -57, 90, -43, -43, 90, -57, -25, 87, -70, -9, 80, -80, 9, 70, -87, 25, 57, -90, 43},
    {38, -88, 73, -4, -67, 90, -46, -31, 85, -78, 13, 61, -90, 54, 22, -82, 82, -22, -54, 90, -61, -13, 78, -85, 31, 46, -90, 67, 4, -73, 88, -38},
    {36, -83, 83, -36, -36, 83, -83, 36, 36, -83, 83, -36, -36, 83, -83, 36, 36, -83, 83, -36, -36, 83, -83, 36, 36, -83, 83, -36, -36, 83, -83, 36},
    {31, -78, 90, -61, 4, 54, -88, 82, -38, -22, 73, -90, 67, -13, -46, 85, -85, 46, 13, -67, 90, -73, 22, 38, -82, 88, -54, -4, 61, -90, 78, -31},
    {25, -70, 90, -80, 43, 9, -57, 87, -87, 57, -9, -43, 80, -90, 70, -25, -25, 70, -90, 80, -43, -9, 57, -87, 87, -57, 9, 43, -80, 90, -70, 25},
    {22, -61, 85, -90, 73, -38, -4, 46, -78, 90, -82, 54, -13, -31, 67, -88, 88, -67, 31, 13, -54, 82, -90, 78, -46, 4, 38, -73, 90, -85, 61, -22},
    {18, -50, 75, -89, 89, -75, 50, -18, -18, 50, -75, 89, -89, 75, -50, 18, 18, -50, 75, -89, 89, -75, 50, -18, -18, 50, -75, 89, -89, 75, -50, 18},
    {13, -38, 61, -78, 88, -90, 85, -73, 54, -31, 4, 22, -46, 67, -82, 90, -90, 82, -67, 46, -22, -4, 31, -54, 73, -85, 90, -88, 78, -61, 38, -13},
    {9, -25, 43, -57, 70, -80, 87, -90, 90, -87, 80, -70, 57, -43, 25, -9, -9, 25, -43, 57, -70, 80, -87, 90, -90, 87, -80, 70, -57, 43, -25, 9},
    {4, -13, 22, -31, 38, -46, 54, -61, 67, -73, 78, -82, 85, -88, 90, -90, 90, -90, 88, -85, 82, -78, 73, -67, 61, -54, 46, -38, 31, -22, 13, -4}};
__constant__ int8_t kDst[4][4] = {{29, 55, 74, 84}, {74, 74, 0, -74}, {84, -29, -74, 55}, {55, -84, 74, -29}};
__constant__ int8_t kLumaTaps[4][8] = {{0, 0, 0, 64, 0, 0, 0, 0}, {-1, 4, -10, 58, 17, -5, 1, 0}, {-1, 4, -11, 40, 40, -11, 4, -1}, {0, 1, -5, 17, 58, -10, 4, -1}};
__constant__ int8_t kChromaTaps[8][4] = {{0, 64, 0, 0},   {-2, 58, 10, -2}, {-4, 54, 16, -2}, {-6, 46, 28, -4},
                                         {-4, 36, 36, -4}, {-4, 28, 46, -6}, {-2, 16, 54, -4}, {-2, 10, 58, -2}};
__constant__ int8_t kAngle[35] = {0,   0,   32,  26,  21,  17,  13,  9,  5,  2,  0,  -2, -5, -9, -13, -17, -21, -26,
                                  -32, -26, -21, -17, -13, -9,  -5,  -2, 0,  2,  5,  9,  13, 17, 21,  26,  32};
__constant__ int16_t kInvAngle[15] = {-4096, -1638, -910, -630, -482, -390, -315, -256, -315, -390, -482, -630, -910, -1638, -4096};
__constant__ uint8_t kBeta[52] = {0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  6,  7,  8,  9,  10, 11, 12, 13, 14, 15,
                                  16, 17, 18, 20, 22, 24, 26, 28, 30, 32, 34, 36, 38, 40, 42, 44, 46, 48, 50, 52, 54, 56, 58, 60, 62, 64};
__constant__ uint8_t kTc[54] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1,  1,  1,  1,  1,  1,
                                2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 5, 5, 6, 6, 7, 8, 9, 10, 11, 13, 14, 16, 18, 20, 22, 24};
__constant__ uint8_t kQpc[13] = {29, 30, 31, 32, 33, 33, 34, 34, 35, 35, 36, 36, 37};
__constant__ int kLevelScale[6] = {40, 45, 51, 57, 64, 72};

__device__ inline int clip3(int lo, int hi, int v) { return v < lo ? lo : v > hi ? hi : v; }
__device__ inline int clip8(int v) { return v < 0 ? 0 : v > 255 ? 255 : v; }
__device__ inline int iabs(int v) { return v < 0 ? -v : v; }
__device__ inline uint8_t* plane_of(const Surface& s, int c) { return c == 0 ? s.y : c == 1 ? s.u : s.v; }

// The words an intra kernel counts in - tickets, the rows' progress, the "done" bytes - start every picture at zero.
__global__ __launch_bounds__(256) void k_hevc_clear(const Batch batch) {
  const PicParams& pp = *batch.pic[blockIdx.y];
  if (!(pp.num_intra[0] | pp.num_intra[1] | pp.num_intra[2])) return;
  const int first = blockIdx.x * 1024 + threadIdx.x * 4;
  for (int k = 0; k < 4; ++k)
    if (first + k < pp.clear_words) pp.intra_ticket[first + k] = 0;
}

// ---------------------------------------------------------------------------------------------------- residual
constexpr int kResWaves = 4;

__global__ __launch_bounds__(64 * kResWaves) void k_hevc_residual(const Batch batch) {
  const PicParams& pp = *batch.pic[blockIdx.y];
  if (static_cast<int>(blockIdx.x) * kResWaves >= pp.num_res) return;
  __shared__ int d[kResWaves][1024];
  __shared__ int g[kResWaves][1024];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int t = blockIdx.x * kResWaves + wave;
  const bool active = t < pp.num_res;
  ResTu tu;
  if (active) tu = pp.res[t];
  const int log2 = active ? tu.log2 : 2, n = 1 << log2, nn = n * n;
  int* dd = d[wave];
  int* gg = g[wave];
  for (int i = lane; i < nn; i += 64) dd[i] = 0;
  __syncthreads();
  const bool bypass = active && (tu.flags & RES_BYPASS), tskip = active && (tu.flags & RES_TSKIP), dst = active && (tu.flags & RES_DST);
  if (active) {
    const uint8_t* m = nullptr;
    if (tu.matrix != 0xff && pp.scaling) {
      const int off = log2 == 2 ? 0 : log2 == 3 ? 96 : log2 == 4 ? 480 : 2016;
      m = pp.scaling + off + tu.matrix * nn;
    }
    const int bd_shift = log2 + 3, qp = tu.qp;
    for (uint32_t k = lane; k < tu.ncoef; k += 64) {
      const uint32_t e = pp.coefs[tu.coef_offset + k];
      const int pos = e >> 16, lev = static_cast<int16_t>(e & 0xffff);
      int v = lev;
      if (!bypass) {
        const long long s = ((static_cast<long long>(lev) * (m ? m[pos] : 16) * kLevelScale[qp % 6]) << (qp / 6)) + (1ll << (bd_shift - 1));
        const long long q = s >> bd_shift;
        v = static_cast<int>(q < -32768 ? -32768 : q > 32767 ? 32767 : q);
      }
      dd[pos] = v;
    }
  }
  __syncthreads();
  const int step = 32 >> log2;
  if (active && !bypass && !tskip) {  // columns
    for (int o = lane; o < nn; o += 64) {
      const int i = o / n, x = o % n;  // output row i of column x
      int e = 0;
      for (int j = 0; j < n; ++j) e += (dst ? kDst[j][i] : kTrans[j * step][i]) * dd[j * n + x];
      gg[o] = clip3(-32768, 32767, (e + 64) >> 7);
    }
  }
  __syncthreads();
  if (active) {
    const int pitch = tu.c_idx ? pp.c_pitch : pp.y_pitch;
    int16_t* out = pp.resid[tu.c_idx] + static_cast<size_t>(tu.y) * pitch + tu.x;
    for (int o = lane; o < nn; o += 64) {
      const int y = o / n, i = o % n;
      int r;
      if (bypass) {
        r = dd[o];
      } else if (tskip) {
        r = ((dd[o] << 7) + 2048) >> 12;
      } else {
        int e = 0;
        for (int j = 0; j < n; ++j) e += (dst ? kDst[j][i] : kTrans[j * step][i]) * gg[y * n + j];
        r = (e + 2048) >> 12;
      }
      out[static_cast<size_t>(y) * pitch + i] = static_cast<int16_t>(r);
    }
  }
}

// ------------------------------------------------------------------------------------------------------- inter
__device__ inline int ref_at(const uint8_t* p, int pitch, int w, int h, int x, int y) {
  return p[static_cast<size_t>(clip3(0, h - 1, y)) * pitch + clip3(0, w - 1, x)];
}

// One workgroup per prediction block.  Per colour plane and list: the reference window - the block plus the filter's
// 7 (3) extra rows and columns, clamped at the picture's edges as 8.5.3.3.3 says - is copied into LDS once (a sample of
// a 64x64 block was fetched 64 times when every output sample interpolated for itself), the horizontal pass writes
// 16-bit intermediates into LDS, the vertical pass reads them; list 0's 14-bit prediction waits in LDS for list 1's.
constexpr int kRefPitch = 72;  // 64 + 7 samples of a window row (+ 1)
__global__ __launch_bounds__(256) void k_hevc_inter(const Batch batch) {
  const PicParams& pp = *batch.pic[blockIdx.y];
  // Workgroups go to the eight XCDs in turn (the grid's x is a multiple of 8), and each XCD has an L2 of its own: XCD k
  // takes the k-th eighth of the picture's prediction blocks - a band of the picture in decoding order - so that a
  // reference picture's rows are fetched by one L2, not by all eight (measured with the blocks dealt out in turn:
  // 50.7 MB fetched from HBM per picture for 3.1 MB of reference plane, profiles/r06_hevc_pmc_summary_before.json).
  const int chunk = (pp.num_pu + 7) >> 3, k = static_cast<int>(blockIdx.x) >> 3;
  const int pi = static_cast<int>(blockIdx.x & 7) * chunk + k;
  if (k >= chunk || pi >= pp.num_pu) return;
  const InterPu pu = pp.pus[pi];
  __shared__ uint8_t s_ref[71 * kRefPitch];
  __shared__ int16_t s_tmp[71 * 64];
  __shared__ int16_t s_p0[64 * 64];
  const int tid = threadIdx.x;
  {  // which prediction block covers a 4x4 block: the deblocking kernel compares the motion across block edges (bS 3)
    const int bw = pu.w >> 2, bh = pu.h >> 2, w4 = pp.width >> 2;
    for (int i = tid; i < bw * bh; i += 256) pp.pu_map[((pu.y >> 2) + i / bw) * w4 + (pu.x >> 2) + i % bw] = static_cast<uint32_t>(pi);
  }
  for (int c = 0; c < 3; ++c) {
    const int sub = c ? 1 : 0;
    const int x0 = pu.x >> sub, y0 = pu.y >> sub, w = pu.w >> sub, h = pu.h >> sub;
    const int pw = pp.width >> sub, ph = pp.height >> sub, pitch = c ? pp.c_pitch : pp.y_pitch;
    const int fb = c ? 3 : 2, fm = (1 << fb) - 1, taps = c ? 4 : 8, before = c ? 1 : 3;
    uint8_t* dst = plane_of(pp.recon, c);
    int16_t* res = pp.resid[c];
    const bool use0 = pu.pred_flag & 1, use1 = pu.pred_flag & 2;
    WeightEntry w0 = {}, w1 = {};
    if (pu.weighted) {
      if (use0) w0 = pp.weights[pu.wp[0]];
      if (use1) w1 = pp.weights[pu.wp[1]];
    }
    for (int l = 0; l < 2; ++l) {
      if (!((pu.pred_flag >> l) & 1)) continue;
      const uint8_t* ref = plane_of(pp.refs[pu.ref_slot[l]], c);
      const int xf = pu.mv[l][0] & fm, yf = pu.mv[l][1] & fm;
      const int xi = x0 + (pu.mv[l][0] >> fb) - (xf ? before : 0), yi = y0 + (pu.mv[l][1] >> fb) - (yf ? before : 0);
      const int rw = w + (xf ? taps - 1 : 0), rh = h + (yf ? taps - 1 : 0);
      __syncthreads();  // (the passes of the previous list / plane are through with the buffers)
      for (int i = tid; i < rw * rh; i += 256) {
        const int ry = i / rw, rx = i - ry * rw;
        s_ref[ry * kRefPitch + rx] = static_cast<uint8_t>(ref_at(ref, pitch, pw, ph, xi + rx, yi + ry));
      }
      __syncthreads();
      const int8_t* tx = c ? kChromaTaps[xf] : kLumaTaps[xf];
      const int8_t* ty = c ? kChromaTaps[yf] : kLumaTaps[yf];
      for (int i = tid; i < w * rh; i += 256) {  // horizontal pass (or a copy): rh rows of w intermediates
        const int ry = i / w, x = i - ry * w;
        int v;
        if (xf) {
          v = 0;
          for (int t = 0; t < taps; ++t) v += tx[t] * s_ref[ry * kRefPitch + x + t];
        } else {
          v = s_ref[ry * kRefPitch + x];
        }
        s_tmp[ry * 64 + x] = static_cast<int16_t>(v);
      }
      __syncthreads();
      const bool last = l == 1 || !use1;
      for (int i = tid; i < w * h; i += 256) {
        const int yy = i / w, xx = i - yy * w;
        int p;  // predSamplesLX, 14 bit
        if (yf) {
          p = 0;
          for (int t = 0; t < taps; ++t) p += ty[t] * s_tmp[(yy + t) * 64 + xx];
          if (xf) p >>= 6;
        } else {
          p = xf ? s_tmp[yy * 64 + xx] : s_tmp[yy * 64 + xx] << 6;
        }
        if (!last) {
          s_p0[i] = static_cast<int16_t>(p);  // (read back by this very thread)
          continue;
        }
        const int p0 = use0 ? (l == 0 ? p : s_p0[i]) : 0, p1 = use1 ? p : 0;
        int v;
        if (!pu.weighted) {
          v = (use0 && use1) ? (p0 + p1 + 64) >> 7 : ((use0 ? p0 : p1) + 32) >> 6;
        } else {
          const int lwd = (c ? (use0 ? w0 : w1).log2wd[1] : (use0 ? w0 : w1).log2wd[0]) + 6;
          if (use0 && use1)
            v = (p0 * w0.w[c] + p1 * w1.w[c] + ((w0.o[c] + w1.o[c] + 1) << lwd)) >> (lwd + 1);
          else if (use0)
            v = ((p0 * w0.w[c] + (1 << (lwd - 1))) >> lwd) + w0.o[c];
          else
            v = ((p1 * w1.w[c] + (1 << (lwd - 1))) >> lwd) + w1.o[c];
        }
        const size_t at = static_cast<size_t>(y0 + yy) * pitch + x0 + xx;
        const int r = res[at];
        if (r) res[at] = 0;  // every sample's residual is read exactly once (by this kernel or by the intra one): the planes are zero again after the picture
        dst[at] = static_cast<uint8_t>(clip8(clip8(v) + r));
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------- intra
// Hand-offs between workgroups of ONE launch (the two intra kernels) use agent-scope accesses for the flags AND for the
// samples that cross (sc1 on gfx950: they go past the per-XCD L2), not acquire / release fences: at agent scope those are
// buffer_inv sc1 / buffer_wbl2 sc1 - the whole L2 of the XCD invalidated in every poll, written back behind every
// transform block - and every kernel running beside this one pays for them (measured: each kernel of a batch took as
// much longer as the batch had pictures, more streams bought nothing; profiles/r06_hevc_fence_free_*).  The producer's
// stores are through (s_waitcnt vmcnt(0): the workgroup is one wavefront) before it raises the flag; the consumer
// issues its loads behind the poll.
#define HW_AS1 __attribute__((address_space(1)))
__device__ __forceinline__ uint32_t agent_ld32(const void* p) { return __hip_atomic_load((const HW_AS1 uint32_t*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ int agent_ld8(const void* p) { return __hip_atomic_load((const HW_AS1 uint8_t*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void agent_st32(void* p, uint32_t v) { __hip_atomic_store((HW_AS1 uint32_t*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void agent_st8(void* p, uint8_t v) { __hip_atomic_store((HW_AS1 uint8_t*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void stores_through() { __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// intraPredAngle and invAngle (tables 8-4, 8-5) in LDS: a table in memory is a cache round trip in every block of the chain
__device__ inline void load_angles(int* angtab, int lane) {
  if (lane < 35) angtab[lane] = kAngle[lane];
  if (lane < 15) angtab[35 + lane] = kInvAngle[lane];
  __syncthreads();
}

// One intra-predicted transform block (8.4.4.2), by ONE wavefront (the workgroup: __syncthreads is a wave barrier).
// The 4n + 1 neighbours are fetched straight from where the substitution process of 8.4.4.2.2 would copy them from -
// which sample that is follows from the availability bits alone (the nearest existing sample below / before, else the
// first existing one), so no lane waits for another; smoothing, the extended reference of the angular modes and the
// prediction itself are lane-parallel over LDS.  line[i]: i < 2n left (bottom-up), 2n the corner, > 2n top.
// `out` is where neighbours are read from and the block goes, (x0, y0) the block's position there: the surface, or (row
// kernel, kTile) an LDS tile of the coding tree block with its rim; `res` likewise the residual plane or an LDS copy of
// the coding tree block's part of it (already cleared in HBM by whoever copied it).
template <bool kTile>
__device__ void intra_block(const IntraTu& tu, int x0, int y0, int c, uint8_t* out, int pitch, int16_t* res, int res_pitch, int* line, int* filt_line, int* refbuf, const int* angtab, int lane) {
  const int ushift = c ? 1 : 2;  // availability comes in units of 4 luma / 2 chroma samples
  const int log2 = tu.log2, n = 1 << log2, nu = (2 * n) >> ushift;
  uint8_t* nb = out;
  const int nb_pitch = pitch;
  uint8_t* dst = out + y0 * pitch + x0;
  int16_t* rs = res + y0 * res_pitch + x0;
  if (tu.mode == kIntraPcm) {
    for (int i = lane; i < n * n; i += 64) {
      const int r = rs[(i / n) * res_pitch + i % n];
      if (!kTile && r) rs[(i / n) * res_pitch + i % n] = 0;
      if (kTile) dst[(i / n) * pitch + i % n] = static_cast<uint8_t>(clip8(r));
      else agent_st8(dst + (i / n) * pitch + i % n, static_cast<uint8_t>(clip8(r)));
    }
  } else {
  const int mode = tu.mode;
  // availability in line order, one bit per unit: left units bottom-up, the corner, top units left to right
  uint64_t mask = 0;
  for (int k = 0; k < nu; ++k) mask |= static_cast<uint64_t>((tu.avail_left >> (nu - 1 - k)) & 1) << k;
  mask |= static_cast<uint64_t>(tu.flags & INTRA_CORNER ? 1 : 0) << nu;
  mask |= static_cast<uint64_t>(tu.avail_top & ((1u << nu) - 1)) << (nu + 1);
  for (int i = lane; i <= 4 * n; i += 64) {
    int v = 128;
    if (mask) {
      // the unit of sample i, and the unit its value comes from
      int u = i < 2 * n ? i >> ushift : i == 2 * n ? nu : nu + 1 + ((i - 2 * n - 1) >> ushift);
      int src = i;
      if (!((mask >> u) & 1)) {
        const uint64_t below = mask & ((1ull << u) - 1);
        if (below) {
          const int su = 63 - __builtin_clzll(below);  // nearest existing unit before: its LAST sample
          src = su < nu ? (su << ushift) + (1 << ushift) - 1 : su == nu ? 2 * n : 2 * n + 1 + ((su - nu - 1) << ushift) + (1 << ushift) - 1;
        } else {
          const int su = __builtin_ctzll(mask);        // nothing before: the FIRST existing sample of the line
          src = su < nu ? (su << ushift) : su == nu ? 2 * n : 2 * n + 1 + ((su - nu - 1) << ushift);
        }
      }
      // (surface variant: a neighbour may be another wavefront's work of this launch)
      const uint8_t* from = src < 2 * n ? nb + (y0 + 2 * n - 1 - src) * nb_pitch + x0 - 1 : src == 2 * n ? nb + (y0 - 1) * nb_pitch + x0 - 1 : nb + (y0 - 1) * nb_pitch + x0 + src - 2 * n - 1;
      v = kTile ? *from : agent_ld8(from);
    }
    line[i] = v;
  }
  __syncthreads();
  // 8.4.4.2.3: which smoothing, if any (every lane decides the same)
  int filt = 0;
  if (c == 0 && mode != 1 && n != 4) {
    const int a = iabs(mode - 26), b = iabs(mode - 10);
    const int dist = a < b ? a : b, thres = n == 8 ? 7 : n == 16 ? 1 : 0;
    if (dist > thres) {
      filt = 1;
      // corner = line[2n], p[-1][2n-1] = line[0], p[2n-1][-1] = line[4n], the middles line[n] (left) and line[3n] (top)
      if ((tu.flags & INTRA_STRONG) && n == 32 && iabs(line[64] + line[128] - 2 * line[96]) < 8 && iabs(line[64] + line[0] - 2 * line[32]) < 8) filt = 2;
    }
  }
  const int* ln = line;
  if (filt) {
    for (int i = lane; i <= 4 * n; i += 64) {
      int v;
      if (i == 0 || i == 4 * n) v = line[i];
      else if (filt == 2) {
        if (i == 64) v = line[64];
        else if (i < 64) v = (i * line[64] + (64 - i) * line[0] + 32) >> 6;        // left: distance from the corner is 64 - i
        else v = ((128 - i) * line[64] + (i - 64) * line[128] + 32) >> 6;
      } else {
        v = (line[i - 1] + 2 * line[i] + line[i + 1] + 2) >> 2;
      }
      filt_line[i] = v;
    }
    __syncthreads();
    ln = filt_line;
  }
#define LEFT(k) ln[2 * n - (k)]       /* k = 0: corner, k = 1 + j: p[-1][j] */
#define TOP(k) ln[2 * n + (k)]        /* k = 0: corner, k = 1 + j: p[j][-1] */
  int* ref = refbuf + 33;
  const int angle = mode >= 2 ? angtab[mode] : 0;
  const bool vertical = mode >= 18;
  if (mode >= 2) {
    const int last = (n * angle) >> 5;
    for (int k = lane - 32; k <= 2 * n; k += 64) {
      if (k >= 0) {
        if (k <= n || angle >= 0) ref[k] = vertical ? TOP(k) : LEFT(k);
      } else if (angle < 0 && last < -1 && k >= last) {
        const int t = (k * angtab[35 + mode - 11] + 128) >> 8;
        ref[k] = vertical ? LEFT(t) : TOP(t);
      }
    }
    __syncthreads();
  }
  int dc = 0;
  if (mode == 1) {
    int sum = n;
    for (int k = 1; k <= n; ++k) sum += LEFT(k) + TOP(k);
    dc = sum >> (log2 + 1);
  }
  for (int i = lane; i < n * n; i += 64) {
    const int x = i & (n - 1), y = i >> log2;
    int v;
    if (mode == 0) {
      v = ((n - 1 - x) * LEFT(1 + y) + (x + 1) * TOP(1 + n) + (n - 1 - y) * TOP(1 + x) + (y + 1) * LEFT(1 + n) + n) >> (log2 + 1);
    } else if (mode == 1) {
      v = dc;
      if (c == 0 && n < 32) {
        if (x == 0 && y == 0) v = (LEFT(1) + 2 * dc + TOP(1) + 2) >> 2;
        else if (y == 0) v = (TOP(1 + x) + 3 * dc + 2) >> 2;
        else if (x == 0) v = (LEFT(1 + y) + 3 * dc + 2) >> 2;
      }
    } else {
      const int j = vertical ? y : x, ii = vertical ? x : y;  // j: distance from the main reference
      const int idx = ((j + 1) * angle) >> 5, fact = ((j + 1) * angle) & 31;
      v = fact ? ((32 - fact) * ref[ii + idx + 1] + fact * ref[ii + idx + 2] + 16) >> 5 : ref[ii + idx + 1];
      if (c == 0 && n < 32) {
        if (mode == 26 && x == 0) v = clip8(TOP(1) + ((LEFT(1 + y) - LEFT(0)) >> 1));
        if (mode == 10 && y == 0) v = clip8(LEFT(1) + ((TOP(1 + x) - TOP(0)) >> 1));
      }
    }
    {
      const int r = rs[y * res_pitch + x];
      if (!kTile && r) rs[y * res_pitch + x] = 0;
      if (kTile) dst[y * pitch + x] = static_cast<uint8_t>(clip8(v + r));
      else agent_st8(dst + y * pitch + x, static_cast<uint8_t>(clip8(v + r)));
    }
  }
#undef LEFT
#undef TOP
  }
}

constexpr uint32_t kIntraSpins = 1u << 22;  // bounded wait for a neighbour (then: fault, the host fails the picture)

// Intra prediction, variant 1: a dataflow wavefront for pictures whose intra blocks are scattered (P and B pictures).
// Workgroups of one wavefront take the coding tree blocks of a colour plane by ticket, in raster order, and work through a
// block's records in decoding order; a transform block waits for exactly the neighbouring 4x4 units it reads that an
// earlier record of this picture writes in ANOTHER coding tree block (IntraTu::wait_*: the host knows which neighbours
// are intra) - each has a "done" byte - and marks its own units when its samples are out.  A ticket holder only ever
// waits for lower tickets (left, above, above-right in raster order), which are held by running wavefronts: no deadlock
// whatever the grid.  (A ticket per RECORD, the first version, handed every dependent pair of neighbours from one
// wavefront to another through a flag: most of the kernel's time was that.)  NOT for intra pictures: there the
// wavefronts would spin on their neighbours (415 ms per picture with a ticket per record,
// profiles/r06_hevc_dataflow_everywhere_kernel_stats.csv); those take variant 2.
__global__ __launch_bounds__(64) void k_hevc_intra_flow(const Batch batch) {
  const PicParams& pp = *batch.pic[blockIdx.z];
  const int c = blockIdx.y, lane = threadIdx.x;
  const int pitch = c ? pp.c_pitch : pp.y_pitch, ushift = c ? 1 : 2, sub = c ? 1 : 0, w4 = pp.width >> 2;
  uint8_t* plane = plane_of(pp.recon, c);
  const IntraTu* tus = pp.intra[c];
  const uint32_t* first = pp.intra_ctb_first[c];
  const uint32_t ctbs = static_cast<uint32_t>(pp.width_ctb * pp.height_ctb);
  uint32_t* ticket = pp.intra_ticket + c;
  uint8_t* done = pp.intra_done[c];
  const int n4 = 1 << (pp.log2_ctb - 2);  // a coding tree block in 4x4 luma units
  __shared__ int line[132], filt_line[132], refbuf[3 * 32 + 8], angtab[50];
  __shared__ uint32_t s_ticket;
  load_angles(angtab, lane);
  for (;;) {
    if (lane == 0) s_ticket = atomicAdd(ticket, 1u);
    __syncthreads();
    const uint32_t ctb = s_ticket;
    __syncthreads();
    if (ctb >= ctbs) return;
    const uint32_t t0 = first[ctb], t1 = first[ctb + 1];
    if (t0 == t1) continue;
    const int bx = static_cast<int>(ctb % static_cast<uint32_t>(pp.width_ctb)) * n4, by = static_cast<int>(ctb / static_cast<uint32_t>(pp.width_ctb)) * n4;
    for (uint32_t ti = t0; ti < t1; ++ti) {
      const IntraTu tu = tus[ti];
      const int n = 1 << tu.log2, ux = (tu.x << sub) >> 2, uy = (tu.y << sub) >> 2, nu = (2 * n) >> ushift;
      if (tu.mode != kIntraPcm && (tu.wait_left | tu.wait_top | (tu.flags & INTRA_WAIT_CORNER))) {
        // lane k < nu: left unit k (from the top); nu <= k < 2 nu: top unit k - nu; k = 2 nu: the corner - unless the
        // unit lies in this coding tree block: then this wavefront wrote it, some blocks ago
        const uint8_t* flag = nullptr;
        int fx = -1, fy = -1;
        if (lane < nu) {
          if ((tu.wait_left >> lane) & 1) { fx = ux - 1; fy = uy + lane; }
        } else if (lane < 2 * nu) {
          if ((tu.wait_top >> (lane - nu)) & 1) { fx = ux + lane - nu; fy = uy - 1; }
        } else if (lane == 2 * nu && (tu.flags & INTRA_WAIT_CORNER)) {
          fx = ux - 1;
          fy = uy - 1;
        }
        if (fx >= 0 && !(fx >= bx && fx < bx + n4 && fy >= by && fy < by + n4)) flag = done + fy * w4 + fx;
        if (__any(flag != nullptr)) {
          uint32_t spins = 0;
          for (;;) {
            const bool ok = !flag || agent_ld8(flag) != 0;
            if (__all(ok)) break;
            for (int k = 0; k < batch.spin_sleep; ++k) __builtin_amdgcn_s_sleep(8);
            if (++spins > kIntraSpins) {
              if (lane == 0) atomicExch(pp.fault, 1u);
              return;
            }
          }
        }
      }
      intra_block<false>(tu, tu.x, tu.y, c, plane, pitch, pp.resid[c], pitch, line, filt_line, refbuf, angtab, lane);
      // the block's samples are out (agent-scope stores, all through): mark the units it covers
      stores_through();
      __syncthreads();
      const int cnt = n >> ushift;
      if (lane < cnt * cnt) agent_st8(done + (uy + lane / cnt) * w4 + ux + lane % cnt, 1);
    }
  }
}

// Intra prediction, variant 2: a wavefront of coding tree block rows, for pictures that are mostly intra.  One
// wavefront per (row, colour plane); row r follows the progress word of row r - 1 two blocks behind (the above-right
// block is a neighbour); within a block the transform blocks are a dependent chain anyway (z-order: each quadrant leans
// on the ones before it).  17 rows at 1080p with 64x64 blocks is all the parallelism that leaves, so what counts is the
// time per transform block, and two dependent HBM round trips per block (its record, then its residual) were most of it:
// the coding tree block with its rim (the row above up to the above-right block, the column to the left), its part of
// the residual plane and its records are copied into LDS first, the chain runs out of LDS, the block goes back in rows.
constexpr int kTilePitch = 144;  // 16 bytes (the left column is the last of them) + 2 x 64 samples
constexpr int kTileRecs = 256;   // the records of a 64x64 block of 4x4s

__global__ __launch_bounds__(64) void k_hevc_intra_rows(const Batch batch) {
  const PicParams& pp = *batch.pic[blockIdx.z];
  const int c = blockIdx.y, row = blockIdx.x, lane = threadIdx.x;
  if (row >= pp.height_ctb) return;
  const int sub = c ? 1 : 0, pitch = c ? pp.c_pitch : pp.y_pitch, pw = pp.width >> sub, ph = pp.height >> sub;
  const int S = (1 << pp.log2_ctb) >> sub;  // the coding tree block in samples of this plane
  uint8_t* plane = plane_of(pp.recon, c);
  int16_t* resid = pp.resid[c];
  const IntraTu* tus = pp.intra[c];
  const uint32_t* first = pp.intra_ctb_first[c] + row * pp.width_ctb;
  uint32_t* progress = pp.progress + c * pp.height_ctb;
  __shared__ int line[132], filt_line[132], refbuf[3 * 32 + 8], angtab[50];
  load_angles(angtab, lane);
  __shared__ __attribute__((aligned(16))) uint8_t tile[(64 + 1) * kTilePitch];
  __shared__ __attribute__((aligned(16))) int16_t rtile[64 * 64];
  __shared__ uint32_t recs[kTileRecs * 6];
  static_assert(sizeof(IntraTu) == 24, "records are copied as six words");
  uint8_t* const origin = tile + kTilePitch + 16;  // the block's sample (0, 0); the rim is at -1
  uint32_t t_next = first[0];
  int published = 0, tile_col = -2;  // tile_col: the block the tile holds
  const int cy = row * S, wpr = S >> 2;
  const int lcpr = pp.log2_ctb - sub - 4, lrpr = pp.log2_ctb - sub - 3;  // log2 of the 16-byte pieces per row of the block / of its residual
  for (int col = 0; col < pp.width_ctb; ++col) {
    const uint32_t t0 = t_next, t1 = first[col + 1];
    t_next = t1;
    if (t1 == t0) continue;  // nothing intra in this block
    // the blocks passed without work are final too: say so before waiting, or the row below waits for this row's wait
    if (published < col) {
      if (lane == 0) agent_st32(&progress[row], static_cast<uint32_t>(col));
      published = col;
    }
    const int cx = col * S;
    // the column left of the block: the last column of the tile when the block to the left was this wavefront's previous
    // one (taken before the tile is overwritten), else what the inter kernel left in the surface
    int left_sample = 0;
    if (lane < S && cx > 0 && cy + lane < ph) left_sample = tile_col == col - 1 ? origin[lane * kTilePitch + S - 1] : plane[static_cast<size_t>(cy + lane) * pitch + cx - 1];
    tile_col = col;
    // what does not depend on the row above: the block itself (what inter prediction left there), its residual - cleared
    // behind the copy: every sample's residual is read once - and the column left of it (this wavefront's own work)
    // (16 bytes a lane and all of a copy's loads in flight at once: what is past the picture's width is pitch padding)
    if (lcpr >= 0) {
      uint4 blk[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int i = lane + 64 * k, r = i >> lcpr, x = cx + 16 * (i & ((1 << lcpr) - 1));
        blk[k] = (r < S && cy + r < ph && x < pw) ? *reinterpret_cast<const uint4*>(plane + static_cast<size_t>(cy + r) * pitch + x) : uint4{0, 0, 0, 0};
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int i = lane + 64 * k, r = i >> lcpr;
        if (r < S) *reinterpret_cast<uint4*>(origin + r * kTilePitch + 16 * (i & ((1 << lcpr) - 1))) = blk[k];
      }
    } else {
      for (int i = lane; i < S * wpr; i += 64) {
        const int r = i / wpr, x = cx + 4 * (i % wpr), y = cy + r;
        if (y < ph && x < pw) *reinterpret_cast<uint32_t*>(origin + r * kTilePitch + 4 * (i % wpr)) = *reinterpret_cast<const uint32_t*>(plane + static_cast<size_t>(y) * pitch + x);
      }
    }
    {
      uint4 blk[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int i = lane + 64 * k, r = i >> lrpr, x = cx + 8 * (i & ((1 << lrpr) - 1));
        blk[k] = (r < S && cy + r < ph && x < pw) ? *reinterpret_cast<const uint4*>(resid + static_cast<size_t>(cy + r) * pitch + x) : uint4{0, 0, 0, 0};
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int i = lane + 64 * k, r = i >> lrpr, x = cx + 8 * (i & ((1 << lrpr) - 1));
        if (blk[k].x | blk[k].y | blk[k].z | blk[k].w) *reinterpret_cast<uint4*>(resid + static_cast<size_t>(cy + r) * pitch + x) = uint4{0, 0, 0, 0};
        if (r < S) reinterpret_cast<uint4*>(rtile)[i] = blk[k];
      }
    }
    if (lane < S) origin[lane * kTilePitch - 1] = static_cast<uint8_t>(left_sample);
    if (row > 0) {
      const uint32_t need = static_cast<uint32_t>(col + 2 < pp.width_ctb ? col + 2 : pp.width_ctb);
      uint32_t spins = 0;
      while (agent_ld32(&progress[row - 1]) < need) {
        for (int k = 0; k < batch.spin_sleep; ++k) __builtin_amdgcn_s_sleep(8);
        if (++spins > kIntraSpins) {
          if (lane == 0) atomicExch(pp.fault, 1u);
          return;
        }
      }
    }
    __asm__ volatile("" ::: "memory");
    // the row above, from the sample left of the block to the end of the above-right block: the last row of another
    // wavefront's blocks (agent-scope stores there, agent-scope loads here), a word a lane - the first one for its last byte
    if (lane < 2 * (S >> 2) + 1) {
      const int x = cx - 4 + 4 * lane;
      const uint32_t w = (cy > 0 && x >= 0 && x < pw) ? agent_ld32(plane + static_cast<size_t>(cy - 1) * pitch + x) : 0u;
      *reinterpret_cast<uint32_t*>(origin - kTilePitch - 4 + 4 * lane) = w;
    }
    for (uint32_t base = t0; base < t1; base += kTileRecs) {
      const int m = static_cast<int>(t1 - base < static_cast<uint32_t>(kTileRecs) ? t1 - base : kTileRecs);
      const uint32_t* src = reinterpret_cast<const uint32_t*>(tus + base);
      for (int i = lane; i < m * 6; i += 64) recs[i] = src[i];
      __syncthreads();
      for (int k = 0; k < m; ++k) {
        const IntraTu& tu = reinterpret_cast<const IntraTu*>(recs)[k];
        intra_block<true>(tu, tu.x - cx, tu.y - cy, c, origin, kTilePitch, rtile, S, line, filt_line, refbuf, angtab, lane);
        __syncthreads();
      }
    }
    if (lcpr >= 0) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int i = lane + 64 * k, r = i >> lcpr, x = cx + 16 * (i & ((1 << lcpr) - 1));
        if (r < S - 1 && cy + r < ph && x < pw) *reinterpret_cast<uint4*>(plane + static_cast<size_t>(cy + r) * pitch + x) = *reinterpret_cast<const uint4*>(origin + r * kTilePitch + 16 * (i & ((1 << lcpr) - 1)));
      }
    } else {
      for (int i = lane; i < (S - 1) * wpr; i += 64) {
        const int r = i / wpr, x = cx + 4 * (i % wpr), y = cy + r;
        if (y < ph && x < pw) *reinterpret_cast<uint32_t*>(plane + static_cast<size_t>(y) * pitch + x) = *reinterpret_cast<const uint32_t*>(origin + r * kTilePitch + 4 * (i % wpr));
      }
    }
    // the block's last row is what the row below reads while this launch runs: agent-scope stores
    if (lane < wpr && cy + S - 1 < ph && cx + 4 * lane < pw)
      agent_st32(plane + static_cast<size_t>(cy + S - 1) * pitch + cx + 4 * lane, *reinterpret_cast<const uint32_t*>(origin + (S - 1) * kTilePitch + 4 * lane));
    // everything up to this block of the row is final as far as later rows are concerned
    stores_through();
    if (lane == 0) agent_st32(&progress[row], static_cast<uint32_t>(col + 1));
    published = col + 1;
  }
  if (published < pp.width_ctb && lane == 0) agent_st32(&progress[row], static_cast<uint32_t>(pp.width_ctb));
}

// -------------------------------------------------------------------------------------------------- deblocking
__device__ void luma_segment(uint8_t* q0, int across, int along, int beta, int tc, bool no_p, bool no_q) {
#define P(k, l) q0[-((k) + 1) * across + (l) * along]
#define Q(k, l) q0[(k) * across + (l) * along]
  const int dp0 = iabs(P(2, 0) - 2 * P(1, 0) + P(0, 0)), dp3 = iabs(P(2, 3) - 2 * P(1, 3) + P(0, 3));
  const int dq0 = iabs(Q(2, 0) - 2 * Q(1, 0) + Q(0, 0)), dq3 = iabs(Q(2, 3) - 2 * Q(1, 3) + Q(0, 3));
  if (dp0 + dq0 + dp3 + dq3 >= beta) return;
  const bool s0 = 2 * (dp0 + dq0) < (beta >> 2) && iabs(P(3, 0) - P(0, 0)) + iabs(Q(0, 0) - Q(3, 0)) < (beta >> 3) && iabs(P(0, 0) - Q(0, 0)) < ((5 * tc + 1) >> 1);
  const bool s3 = 2 * (dp3 + dq3) < (beta >> 2) && iabs(P(3, 3) - P(0, 3)) + iabs(Q(0, 3) - Q(3, 3)) < (beta >> 3) && iabs(P(0, 3) - Q(0, 3)) < ((5 * tc + 1) >> 1);
  const bool strong = s0 && s3;
  const int side = (beta + (beta >> 1)) >> 3;
  const bool dep = dp0 + dp3 < side, deq = dq0 + dq3 < side;
  for (int l = 0; l < 4; ++l) {
    const int p0 = P(0, l), p1 = P(1, l), p2 = P(2, l), p3 = P(3, l), q0v = Q(0, l), q1 = Q(1, l), q2 = Q(2, l), q3 = Q(3, l);
    if (strong) {
      if (!no_p) {
        P(0, l) = static_cast<uint8_t>(clip3(p0 - 2 * tc, p0 + 2 * tc, (p2 + 2 * p1 + 2 * p0 + 2 * q0v + q1 + 4) >> 3));
        P(1, l) = static_cast<uint8_t>(clip3(p1 - 2 * tc, p1 + 2 * tc, (p2 + p1 + p0 + q0v + 2) >> 2));
        P(2, l) = static_cast<uint8_t>(clip3(p2 - 2 * tc, p2 + 2 * tc, (2 * p3 + 3 * p2 + p1 + p0 + q0v + 4) >> 3));
      }
      if (!no_q) {
        Q(0, l) = static_cast<uint8_t>(clip3(q0v - 2 * tc, q0v + 2 * tc, (p1 + 2 * p0 + 2 * q0v + 2 * q1 + q2 + 4) >> 3));
        Q(1, l) = static_cast<uint8_t>(clip3(q1 - 2 * tc, q1 + 2 * tc, (p0 + q0v + q1 + q2 + 2) >> 2));
        Q(2, l) = static_cast<uint8_t>(clip3(q2 - 2 * tc, q2 + 2 * tc, (p0 + q0v + q1 + 3 * q2 + 2 * q3 + 4) >> 3));
      }
    } else {
      int delta = (9 * (q0v - p0) - 3 * (q1 - p1) + 8) >> 4;
      if (iabs(delta) >= tc * 10) continue;
      delta = clip3(-tc, tc, delta);
      if (!no_p) {
        P(0, l) = static_cast<uint8_t>(clip8(p0 + delta));
        if (dep) P(1, l) = static_cast<uint8_t>(clip8(p1 + clip3(-(tc >> 1), tc >> 1, (((p2 + p0 + 1) >> 1) - p1 + delta) >> 1)));
      }
      if (!no_q) {
        Q(0, l) = static_cast<uint8_t>(clip8(q0v - delta));
        if (deq) Q(1, l) = static_cast<uint8_t>(clip8(q1 + clip3(-(tc >> 1), tc >> 1, (((q2 + q0v + 1) >> 1) - q1 - delta) >> 1)));
      }
    }
  }
#undef P
#undef Q
}

// 8.7.2.4, the last rule: two prediction blocks differ enough for a filtered edge if they use different reference pictures
// or numbers of vectors, or a vector component differs by a whole sample or more (4 quarter samples).  "The same
// picture" is "the same surface" here; for two vectors each either pairing of the lists may match.
__device__ inline int motion_bs(const InterPu& p, const InterPu& q) {
  const int np = (p.pred_flag & 1) + (p.pred_flag >> 1), nq = (q.pred_flag & 1) + (q.pred_flag >> 1);
  if (np != nq) return 1;
  auto far = [](const int16_t* a, const int16_t* b) { return iabs(a[0] - b[0]) >= 4 || iabs(a[1] - b[1]) >= 4; };
  if (np == 1) {
    const int lp = (p.pred_flag & 1) ? 0 : 1, lq = (q.pred_flag & 1) ? 0 : 1;
    return p.ref_slot[lp] != q.ref_slot[lq] || far(p.mv[lp], q.mv[lq]);
  }
  const bool straight = p.ref_slot[0] == q.ref_slot[0] && p.ref_slot[1] == q.ref_slot[1];
  const bool crossed = p.ref_slot[0] == q.ref_slot[1] && p.ref_slot[1] == q.ref_slot[0];
  if (!straight && !crossed) return 1;
  if (p.ref_slot[0] != p.ref_slot[1]) {
    if (straight) return far(p.mv[0], q.mv[0]) || far(p.mv[1], q.mv[1]);
    return far(p.mv[0], q.mv[1]) || far(p.mv[1], q.mv[0]);
  }
  return (far(p.mv[0], q.mv[0]) || far(p.mv[1], q.mv[1])) && (far(p.mv[0], q.mv[1]) || far(p.mv[1], q.mv[0]));
}

__global__ __launch_bounds__(256) void k_hevc_deblock(const Batch batch, int dir) {
  const PicParams& pp = *batch.pic[blockIdx.y];
  if (!pp.deblock_enabled) return;
  const int w4 = pp.width >> 2, h4 = pp.height >> 2;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= w4 * h4) return;
  const int x = i % w4, y = i / w4;
  const EdgeInfo e = pp.edges[i];
  // an edge of the 8x8 grid that is a transform or prediction block edge and may be filtered
  if (((dir ? y : x) & 1) || !(e.edge & (dir ? EDGE_ON_TOP : EDGE_ON_LEFT))) return;
  const EdgeInfo p = pp.edges[dir ? i - w4 : i - 1];
  int bs;  // 8.7.2.4
  if ((e.edge | p.edge) & EDGE_INTRA) {
    bs = 2;
  } else if ((e.edge & (dir ? EDGE_TU_TOP : EDGE_TU_LEFT)) && ((e.edge | p.edge) & EDGE_LEVELS)) {
    bs = 1;
  } else {
    const uint32_t pj = pp.pu_map[dir ? i - w4 : i - 1], pq = pp.pu_map[i];
    if (pj == pq || !motion_bs(pp.pus[pj], pp.pus[pq])) return;
    bs = 1;
  }
  const SliceLoop sl = pp.slices[e.slice];
  const int qpl = (e.qp + p.qp + 1) >> 1;
  const bool no_p = p.flags & 1, no_q = e.flags & 1;
  {
    const int beta = kBeta[clip3(0, 51, qpl + sl.beta_offset)];
    const int tc = kTc[clip3(0, 53, qpl + 2 * (bs - 1) + sl.tc_offset)];
    uint8_t* q0 = pp.recon.y + static_cast<size_t>(y * 4) * pp.y_pitch + x * 4;
    luma_segment(q0, dir ? pp.y_pitch : 1, dir ? 1 : pp.y_pitch, beta, tc, no_p, no_q);
  }
  if (bs == 2 && !((dir ? y : x) & 3)) {
    for (int c = 1; c < 3; ++c) {
      const int qpi = qpl + (c == 1 ? pp.cb_qp_offset : pp.cr_qp_offset);
      const int qpc = qpi < 30 ? qpi : qpi >= 43 ? qpi - 6 : kQpc[qpi - 30];
      const int tc = kTc[clip3(0, 53, qpc + 2 + sl.tc_offset)];
      uint8_t* q0 = plane_of(pp.recon, c) + static_cast<size_t>(y * 2) * pp.c_pitch + x * 2;
      const int across = dir ? pp.c_pitch : 1, along = dir ? 1 : pp.c_pitch;
      for (int l = 0; l < 2; ++l) {
        uint8_t* q = q0 + l * along;
        const int p0 = q[-across], p1 = q[-2 * across], q0v = q[0], q1 = q[across];
        const int delta = clip3(-tc, tc, (((q0v - p0) << 2) + p1 - q1 + 4) >> 3);
        if (!no_p) q[-across] = static_cast<uint8_t>(clip8(p0 + delta));
        if (!no_q) q[0] = static_cast<uint8_t>(clip8(q0v - delta));
      }
    }
  }
}

// --------------------------------------------------------------------------------------------------------- SAO
__global__ __launch_bounds__(256) void k_hevc_sao(const Batch batch) {
  const PicParams& pp = *batch.pic[blockIdx.y / 3];
  if (!pp.sao_enabled) return;
  const int c = blockIdx.y % 3, sub = c ? 1 : 0;
  const int pw = pp.width >> sub, ph = pp.height >> sub, pitch = c ? pp.c_pitch : pp.y_pitch;
  // four samples of a row per thread, one 32-bit store (a byte per thread wrote 20 MB per 1080p picture for 3.1 MB of
  // samples - every byte store its own partial line, profiles/r06_hevc_pmc_summary_before.json); the plane's width is a
  // multiple of four and the four lie in one coding tree block, hence one slice
  // ... and a band of rows per XCD (workgroups go to the eight XCDs in turn, the grid's x is a multiple of 8): a row's
  // neighbours above and below are in the same L2 instead of in two others
  const int xcd = static_cast<int>(blockIdx.x & 7), j = static_cast<int>(blockIdx.x >> 3);
  const int xblocks = (pw / 4 + 255) / 256, band = (ph + 7) >> 3;
  const int row_in_band = j / xblocks;
  const int x0 = ((j % xblocks) * 256 + static_cast<int>(threadIdx.x)) * 4, y = xcd * band + row_in_band;
  if (row_in_band >= band || x0 >= pw || y >= ph) return;
  const uint8_t* src = plane_of(pp.recon, c);
  uint8_t* dst = plane_of(pp.dst, c);
  const int w4 = pp.width >> 2;
  const int yl = y << sub;
  const int ctb = (yl >> pp.log2_ctb) * pp.width_ctb + ((x0 << sub) >> pp.log2_ctb);
  const SaoCtb p = pp.sao[ctb];
  const uint32_t cur = *reinterpret_cast<const uint32_t*>(src + static_cast<size_t>(y) * pitch + x0);
  const EdgeInfo e0 = pp.edges[(yl >> 2) * w4 + ((x0 << sub) >> 2)];
  const SliceLoop sl = pp.slices[e0.slice];
  const int type = (c ? sl.sao_chroma : sl.sao_luma) ? p.type[c] : 0;
  uint32_t out4 = cur;
  if (type) {
    out4 = 0;
    const int cl = p.eo_class[c];
    const int dx = cl == 1 ? 0 : cl == 3 ? -1 : 1, dy = cl == 0 ? 0 : 1;  // the second neighbour; the first is its mirror image
    for (int k = 0; k < 4; ++k) {
      const int x = x0 + k, xl = x << sub;
      const int v = (cur >> (8 * k)) & 0xff;
      int out = v;
      const EdgeInfo e = (k < 2 || !sub) ? e0 : pp.edges[(yl >> 2) * w4 + (xl >> 2)];  // (four chroma samples cover two 4x4 luma blocks)
      if (!(e.flags & 1)) {
        int idx = 0;
        if (type == 1) {
          const int b = ((v >> 3) - p.band_pos[c]) & 31;
          idx = b < 4 ? b + 1 : 0;
        } else {
          const int xa = x - dx, ya = y - dy, xb = x + dx, yb = y + dy;
          bool ok = xa >= 0 && xb >= 0 && xa < pw && xb < pw && ya >= 0 && yb < ph;
          if (ok) {
            const EdgeInfo ea = pp.edges[((ya << sub) >> 2) * w4 + ((xa << sub) >> 2)], eb = pp.edges[((yb << sub) >> 2) * w4 + ((xb << sub) >> 2)];
            // across a slice boundary the later slice's flag decides
            if (ea.slice != e.slice) ok = ok && (pp.slices[ea.slice].first_ctb > sl.first_ctb ? pp.slices[ea.slice].across : sl.across);
            if (eb.slice != e.slice) ok = ok && (pp.slices[eb.slice].first_ctb > sl.first_ctb ? pp.slices[eb.slice].across : sl.across);
          }
          if (ok) {
            const int a = src[static_cast<size_t>(ya) * pitch + xa], b = src[static_cast<size_t>(yb) * pitch + xb];
            const int sgn = (v > a) - (v < a) + (v > b) - (v < b);
            idx = sgn == -2 ? 1 : sgn == -1 ? 2 : sgn == 1 ? 3 : sgn == 2 ? 4 : 0;
          }
        }
        if (idx) out = clip8(v + p.offset[c][idx - 1]);
      }
      out4 |= static_cast<uint32_t>(out) << (8 * k);
    }
  }
  *reinterpret_cast<uint32_t*>(dst + static_cast<size_t>(y) * pitch + x0) = out4;
}

}  // namespace

namespace hwang {
namespace amd {
namespace hevc {

// The kernels of `n` pictures of different decoders on `stream`, stage by stage: one launch per stage whatever `n` is.
// `pp[i]` is the host copy of picture i's parameters (for the grids), `batch.pic[i]` its device copy (already in the
// stream's past).  Pictures that take the row wavefront and pictures that take the dataflow kernel get a launch each.
hipError_t launch_batch(const PicParams* const* pp, const Batch& batch, int n, hipStream_t stream) {
  if (n <= 0) return hipSuccess;
  if (n > kMaxBatch) return hipErrorInvalidValue;
  int res = 0, pus = 0, units = 0, width = 0, height = 0, clear = 0;
  bool deblock = false, sao = false;
  static const int flow_cap = std::getenv("HWANG_HEVC_FLOW_WORKERS") ? std::atoi(std::getenv("HWANG_HEVC_FLOW_WORKERS")) : 32;
  static const int spin = std::getenv("HWANG_HEVC_SPIN_SLEEP") ? std::atoi(std::getenv("HWANG_HEVC_SPIN_SLEEP")) : 4;
  Batch rows, flow;
  static const int rows_spin = std::getenv("HWANG_HEVC_ROWS_SLEEP") ? std::atoi(std::getenv("HWANG_HEVC_ROWS_SLEEP")) : 16;
  flow.spin_sleep = spin;
  rows.spin_sleep = rows_spin;
  int n_rows = 0, n_flow = 0, rows_ctb = 0, flow_most = 0;
  for (int i = 0; i < n; ++i) {
    const PicParams& p = *pp[i];
    res = p.num_res > res ? p.num_res : res;
    pus = p.num_pu > pus ? p.num_pu : pus;
    if (p.deblock_enabled) {
      deblock = true;
      const int u = (p.width >> 2) * (p.height >> 2);
      units = u > units ? u : units;
    }
    if (p.sao_enabled) {
      sao = true;
      width = p.width > width ? p.width : width;
      height = p.height > height ? p.height : height;
    }
    if (p.num_intra[0] || p.num_intra[1] || p.num_intra[2]) {
      clear = p.clear_words > clear ? p.clear_words : clear;
      if (p.intra_rows) {
        rows.pic[n_rows++] = batch.pic[i];
        rows_ctb = p.height_ctb > rows_ctb ? p.height_ctb : rows_ctb;
      } else {
        // as many wavefronts as a sparse picture has independent coding tree blocks in flight; each takes blocks until none are left
        const int most = p.width_ctb * p.height_ctb;
        flow.pic[n_flow++] = batch.pic[i];
        flow_most = most > flow_most ? most : flow_most;
      }
    }
  }
  if (clear) hipLaunchKernelGGL(k_hevc_clear, dim3((clear + 1023) / 1024, n), dim3(256), 0, stream, batch);
  if (res) hipLaunchKernelGGL(k_hevc_residual, dim3((res + kResWaves - 1) / kResWaves, n), dim3(64 * kResWaves), 0, stream, batch);
  if (pus) hipLaunchKernelGGL(k_hevc_inter, dim3((pus + 7) & ~7, n), dim3(256), 0, stream, batch);
  if (n_rows) hipLaunchKernelGGL(k_hevc_intra_rows, dim3(rows_ctb, 3, n_rows), dim3(64), 0, stream, rows);
  if (n_flow) hipLaunchKernelGGL(k_hevc_intra_flow, dim3(flow_most < flow_cap ? flow_most : flow_cap, 3, n_flow), dim3(64), 0, stream, flow);
  if (deblock) {
    hipLaunchKernelGGL(k_hevc_deblock, dim3((units + 255) / 256, n), dim3(256), 0, stream, batch, 0);
    hipLaunchKernelGGL(k_hevc_deblock, dim3((units + 255) / 256, n), dim3(256), 0, stream, batch, 1);
  }
  if (sao) hipLaunchKernelGGL(k_hevc_sao, dim3(8 * ((width / 4 + 255) / 256) * ((height + 7) / 8), 3 * n), dim3(256), 0, stream, batch);
  return hipGetLastError();
}

}  // namespace hevc
}  // namespace amd
}  // namespace hwang
