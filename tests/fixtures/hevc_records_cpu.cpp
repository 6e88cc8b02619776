// TEST FIXTURE, not part of the product and never linked into libhwang_amd.so: runs the product's HOST half of the
// HEVC path (hwang_amd/csrc/hevc/hevc_parser.cpp) over the samples of a clip and interprets the records it emits
// (amd/hevc_frame_job.h) with plain scalar loops, so that the CPU test suite - no GPU in the build container - can
// check parser + seam against the oracle's pictures (tests/test_hevc_host.py).  What k_hevc.hip does with the same
// records is checked on the GPU (tests/test_gpu_hevc.py); this file shares no code with it.
//
//   hevc_records_cpu <hvcc file> <samples blob> <sizes file (one decimal per line)> <out.yuv> [wanted ordinals...] [kf keyframe ordinals...]
// Output: cropped I420 frames in display order; a skipped output position is written as a frame of zeros.
#include "../../hwang_amd/csrc/hevc/hevc_parser.h"
#include "../../hwang_amd/csrc/hevc/hevc_tables.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iterator>
#include <utility>
#include <vector>

using namespace hwang;
namespace gh = hwang::amd::hevc;

static int clip3(int lo, int hi, int v) { return v < lo ? lo : v > hi ? hi : v; }
static int clip8(int v) { return clip3(0, 255, v); }
static int iabs(int v) { return v < 0 ? -v : v; }

struct Plane3 {
  std::vector<uint8_t> p[3];
};

struct CpuSink : public hevc::PictureSink {
  hevc::PicSetup setup;
  std::vector<Plane3> surf = std::vector<Plane3>(gh::kMaxSurfaces);
  std::vector<int> held = std::vector<int>(gh::kMaxSurfaces, 0);
  std::vector<std::vector<uint8_t>> outputs;  // cropped I420 (empty = skipped position)
  // staging
  std::vector<gh::ResTu> res;
  std::vector<uint32_t> coefs;
  std::vector<gh::InterPu> pus;
  std::vector<gh::WeightEntry> weights;
  std::vector<gh::IntraTu> intra[3];
  std::vector<uint32_t> first[3];
  std::vector<gh::EdgeInfo> edges;
  std::vector<gh::SaoCtb> sao;
  std::vector<gh::SliceLoop> slices = std::vector<gh::SliceLoop>(gh::kMaxSlices);
  std::vector<uint8_t> scaling = std::vector<uint8_t>(4064);
  int W = 0, H = 0;
  int64_t n_res = 0, n_coef = 0, n_pu = 0, n_intra = 0;

  bool begin_picture(const hevc::PicSetup& s, hevc::PicBuffers* out, std::string* error) override {
    setup = s;
    W = s.width;
    H = s.height;
    const size_t w4h4 = static_cast<size_t>(W / 4) * (H / 4), ctbs = static_cast<size_t>((W + (1 << s.log2_ctb) - 1) >> s.log2_ctb) * ((H + (1 << s.log2_ctb) - 1) >> s.log2_ctb);
    res.assign(w4h4 * 3 / 2 + 64, gh::ResTu());
    coefs.assign(static_cast<size_t>(W) * H * 3 / 2 + 64, 0);
    pus.assign(w4h4 / 2 + 64, gh::InterPu());
    weights.assign(2 * 16 * gh::kMaxSlices, gh::WeightEntry());
    for (int c = 0; c < 3; ++c) {
      intra[c].assign((c ? w4h4 / 4 : w4h4) + 64, gh::IntraTu());
      first[c].assign(ctbs + 1, 0);
    }
    edges.assign(w4h4, gh::EdgeInfo());
    sao.assign(ctbs, gh::SaoCtb());
    int slot = -1;
    for (int i = 0; i < gh::kMaxSurfaces && slot < 0; ++i)
      if (!held[i]) slot = i;
    if (slot < 0) {
      *error = "no free surface";
      return false;
    }
    held[slot] = 1;
    surf[slot].p[0].assign(static_cast<size_t>(W) * H, 0);
    surf[slot].p[1].assign(static_cast<size_t>(W) * H / 4, 0);
    surf[slot].p[2].assign(static_cast<size_t>(W) * H / 4, 0);
    out->slot = slot;
    out->res = res.data(); out->cap_res = res.size();
    out->coefs = coefs.data(); out->cap_coefs = coefs.size();
    out->pus = pus.data(); out->cap_pus = pus.size();
    out->weights = weights.data(); out->cap_weights = weights.size();
    for (int c = 0; c < 3; ++c) {
      out->intra[c] = intra[c].data();
      out->cap_intra[c] = intra[c].size();
      out->intra_ctb_first[c] = first[c].data();
    }
    out->edges = edges.data();
    out->sao = sao.data();
    out->slices = slices.data();
    out->scaling = scaling.data();
    return true;
  }

  // ---- the records, interpreted
  static int interp(const std::vector<uint8_t>& ref, int w, int h, bool chroma, int xi, int yi, int xf, int yf) {
    const int taps = chroma ? 4 : 8, before = chroma ? 1 : 3;
    auto at = [&](int x, int y) { return static_cast<int>(ref[static_cast<size_t>(clip3(0, h - 1, y)) * w + clip3(0, w - 1, x)]); };
    auto fx = [&](int t) { return static_cast<int>(chroma ? hevc_chroma_filter[xf][t] : hevc_luma_filter[xf][t]); };
    auto fy = [&](int t) { return static_cast<int>(chroma ? hevc_chroma_filter[yf][t] : hevc_luma_filter[yf][t]); };
    if (!xf && !yf) return at(xi, yi) << 6;
    int v = 0;
    if (!yf) {
      for (int t = 0; t < taps; ++t) v += fx(t) * at(xi + t - before, yi);
      return v;
    }
    if (!xf) {
      for (int t = 0; t < taps; ++t) v += fy(t) * at(xi, yi + t - before);
      return v;
    }
    for (int u = 0; u < taps; ++u) {
      int r = 0;
      for (int t = 0; t < taps; ++t) r += fx(t) * at(xi + t - before, yi + u - before);
      v += fy(u) * r;
    }
    return v >> 6;
  }

  bool end_picture(const hevc::PicDone& d, std::string*) override {
    n_res += static_cast<int64_t>(d.num_res);
    n_coef += static_cast<int64_t>(d.num_coefs);
    n_pu += static_cast<int64_t>(d.num_pus);
    n_intra += static_cast<int64_t>(d.num_intra[0] + d.num_intra[1] + d.num_intra[2]);
    if (std::getenv("HEVC_RECORDS_VERBOSE")) {
      std::printf("picture (surface %d): res=%zu levels=%zu pus=%zu intra=%zu/%zu/%zu; prediction blocks per reference surface:", d.slot, d.num_res, d.num_coefs, d.num_pus,
                  d.num_intra[0], d.num_intra[1], d.num_intra[2]);
      int per_slot[gh::kMaxSurfaces] = {0};
      for (size_t t = 0; t < d.num_pus; ++t)
        for (int l = 0; l < 2; ++l)
          if ((pus[t].pred_flag >> l) & 1) ++per_slot[pus[t].ref_slot[l]];
      for (int k = 0; k < gh::kMaxSurfaces; ++k)
        if (per_slot[k]) std::printf(" %d:%d", k, per_slot[k]);
      std::printf("\n");
    }
    Plane3& cur = surf[d.slot];
    std::vector<int16_t> resid[3];
    for (int c = 0; c < 3; ++c) resid[c].assign(cur.p[c].size(), 0);
    // residual
    static const int level_scale[6] = {40, 45, 51, 57, 64, 72};
    for (size_t t = 0; t < d.num_res; ++t) {
      const gh::ResTu& tu = res[t];
      const int n = 1 << tu.log2, pw = W >> (tu.c_idx ? 1 : 0);
      std::vector<long long> dq(static_cast<size_t>(n) * n, 0), g(static_cast<size_t>(n) * n, 0);
      const uint8_t* m = nullptr;
      if (tu.matrix != 0xff && d.scaling) m = scaling.data() + (tu.log2 == 2 ? 0 : tu.log2 == 3 ? 96 : tu.log2 == 4 ? 480 : 2016) + tu.matrix * n * n;
      for (uint32_t k = 0; k < tu.ncoef; ++k) {
        const uint32_t e = coefs[tu.coef_offset + k];
        const int pos = static_cast<int>(e >> 16), lev = static_cast<int16_t>(e & 0xffff);
        long long v = lev;
        if (!(tu.flags & gh::RES_BYPASS)) {
          const int sh = tu.log2 + 3;
          v = ((static_cast<long long>(lev) * (m ? m[pos] : 16) * level_scale[tu.qp % 6] << (tu.qp / 6)) + (1ll << (sh - 1))) >> sh;
          v = v < -32768 ? -32768 : v > 32767 ? 32767 : v;
        }
        dq[static_cast<size_t>(pos)] = v;
      }
      for (int y = 0; y < n; ++y)
        for (int x = 0; x < n; ++x) {
          long long r;
          if (tu.flags & gh::RES_BYPASS) r = dq[static_cast<size_t>(y) * n + x];
          else if (tu.flags & gh::RES_TSKIP) r = ((dq[static_cast<size_t>(y) * n + x] << 7) + 2048) >> 12;
          else r = 0;
          g[static_cast<size_t>(y) * n + x] = r;
        }
      if (!(tu.flags & (gh::RES_BYPASS | gh::RES_TSKIP))) {
        const int step = 32 >> tu.log2;
        const bool dst = tu.flags & gh::RES_DST;
        std::vector<long long> e1(static_cast<size_t>(n) * n);
        for (int x = 0; x < n; ++x)
          for (int i = 0; i < n; ++i) {
            long long e = 0;
            for (int j = 0; j < n; ++j) e += static_cast<long long>(dst ? hevc_dst4[j][i] : hevc_trans_matrix[j * step][i]) * dq[static_cast<size_t>(j) * n + x];
            e = (e + 64) >> 7;
            e1[static_cast<size_t>(i) * n + x] = e < -32768 ? -32768 : e > 32767 ? 32767 : e;
          }
        for (int y = 0; y < n; ++y)
          for (int i = 0; i < n; ++i) {
            long long e = 0;
            for (int j = 0; j < n; ++j) e += static_cast<long long>(dst ? hevc_dst4[j][i] : hevc_trans_matrix[j * step][i]) * e1[static_cast<size_t>(y) * n + j];
            g[static_cast<size_t>(y) * n + i] = (e + 2048) >> 12;
          }
      }
      for (int y = 0; y < n; ++y)
        for (int x = 0; x < n; ++x) resid[tu.c_idx][static_cast<size_t>(tu.y + y) * pw + tu.x + x] = static_cast<int16_t>(g[static_cast<size_t>(y) * n + x]);
    }
    // inter
    for (size_t k = 0; k < d.num_pus; ++k) {
      const gh::InterPu& pu = pus[k];
      for (int c = 0; c < 3; ++c) {
        const int sub = c ? 1 : 0, x0 = pu.x >> sub, y0 = pu.y >> sub, w = pu.w >> sub, h = pu.h >> sub, pw = W >> sub, ph = H >> sub, fb = c ? 3 : 2, fm = (1 << fb) - 1;
        for (int y = y0; y < y0 + h; ++y)
          for (int x = x0; x < x0 + w; ++x) {
            int p[2] = {0, 0};
            for (int l = 0; l < 2; ++l)
              if ((pu.pred_flag >> l) & 1)
                p[l] = interp(surf[pu.ref_slot[l]].p[c], pw, ph, c != 0, x + (pu.mv[l][0] >> fb), y + (pu.mv[l][1] >> fb), pu.mv[l][0] & fm, pu.mv[l][1] & fm);
            int v;
            const bool bi = pu.pred_flag == 3;
            const int l0 = (pu.pred_flag & 1) ? 0 : 1;
            if (!pu.weighted) {
              v = bi ? (p[0] + p[1] + 64) >> 7 : (p[l0] + 32) >> 6;
            } else {
              const gh::WeightEntry& a = weights[pu.wp[l0]];
              const int lwd = a.log2wd[c ? 1 : 0] + 6;
              if (bi) {
                const gh::WeightEntry& b = weights[pu.wp[1]];
                v = (p[0] * a.w[c] + p[1] * b.w[c] + ((a.o[c] + b.o[c] + 1) << lwd)) >> (lwd + 1);
              } else {
                v = ((p[l0] * a.w[c] + (1 << (lwd - 1))) >> lwd) + a.o[c];
              }
            }
            const size_t at = static_cast<size_t>(y) * pw + x;
            cur.p[c][at] = static_cast<uint8_t>(clip8(clip8(v) + resid[c][at]));
          }
      }
    }
    // intra, plane by plane, in list order
    for (int c = 0; c < 3; ++c) {
      const int pw = W >> (c ? 1 : 0), unit = c ? 2 : 4;
      std::vector<uint8_t>& pl = cur.p[c];
      for (size_t k = 0; k < d.num_intra[c]; ++k) {
        const gh::IntraTu& tu = intra[c][k];
        const int n = 1 << tu.log2, x0 = tu.x, y0 = tu.y;
        if (tu.mode == gh::kIntraPcm) {
          for (int y = 0; y < n; ++y)
            for (int x = 0; x < n; ++x) pl[static_cast<size_t>(y0 + y) * pw + x0 + x] = static_cast<uint8_t>(clip8(resid[c][static_cast<size_t>(y0 + y) * pw + x0 + x]));
          continue;
        }
        int left[130], top[130];
        for (int i = 0; i < 2 * n; ++i) {
          left[1 + i] = ((tu.avail_left >> (i / unit)) & 1) ? pl[static_cast<size_t>(y0 + i) * pw + x0 - 1] : -1;
          top[1 + i] = ((tu.avail_top >> (i / unit)) & 1) ? pl[static_cast<size_t>(y0 - 1) * pw + x0 + i] : -1;
        }
        left[0] = (tu.flags & gh::INTRA_CORNER) ? pl[static_cast<size_t>(y0 - 1) * pw + x0 - 1] : -1;
        bool any = left[0] >= 0;
        for (int i = 1; i <= 2 * n; ++i) any = any || left[i] >= 0 || top[i] >= 0;
        if (!any) {
          for (int i = 0; i <= 2 * n; ++i) left[i] = top[i] = 128;
        } else {
          if (left[2 * n] < 0) {
            int v = -1;
            for (int i = 2 * n - 1; i >= 0 && v < 0; --i) v = left[i];
            for (int i = 1; i <= 2 * n && v < 0; ++i) v = top[i];
            left[2 * n] = v;
          }
          for (int i = 2 * n - 1; i >= 0; --i)
            if (left[i] < 0) left[i] = left[i + 1];
          top[0] = left[0];
          for (int i = 1; i <= 2 * n; ++i)
            if (top[i] < 0) top[i] = top[i - 1];
        }
        top[0] = left[0];
        const int mode = tu.mode;
        if (c == 0 && mode != 1 && n != 4) {
          const int dist = std::min(iabs(mode - 26), iabs(mode - 10)), thres = n == 8 ? 7 : n == 16 ? 1 : 0;
          if (dist > thres) {
            int fl[130], ft[130];
            if ((tu.flags & gh::INTRA_STRONG) && n == 32 && iabs(left[0] + top[64] - 2 * top[32]) < 8 && iabs(left[0] + left[64] - 2 * left[32]) < 8) {
              for (int i = 0; i <= 64; ++i) {
                fl[i] = (i == 0 || i == 64) ? left[i] : ((64 - i) * left[0] + i * left[64] + 32) >> 6;
                ft[i] = (i == 0 || i == 64) ? top[i] : ((64 - i) * top[0] + i * top[64] + 32) >> 6;
              }
            } else {
              fl[0] = ft[0] = (left[1] + 2 * left[0] + top[1] + 2) >> 2;
              for (int i = 1; i < 2 * n; ++i) {
                fl[i] = (left[i + 1] + 2 * left[i] + left[i - 1] + 2) >> 2;
                ft[i] = (top[i + 1] + 2 * top[i] + top[i - 1] + 2) >> 2;
              }
              fl[2 * n] = left[2 * n];
              ft[2 * n] = top[2 * n];
            }
            std::memcpy(left, fl, sizeof(int) * static_cast<size_t>(2 * n + 1));
            std::memcpy(top, ft, sizeof(int) * static_cast<size_t>(2 * n + 1));
          }
        }
        int refbuf[110];
        int* ref = refbuf + 33;
        const int angle = mode >= 2 ? hevc_intra_angle[mode] : 0;
        const bool vertical = mode >= 18;
        if (mode >= 2) {
          const int* mainr = vertical ? top : left;
          const int* side = vertical ? left : top;
          for (int i = 0; i <= n; ++i) ref[i] = mainr[i];
          if (angle < 0) {
            const int last = (n * angle) >> 5;
            if (last < -1)
              for (int i = -1; i >= last; --i) ref[i] = side[(i * hevc_inv_angle[mode - 11] + 128) >> 8];
          } else {
            for (int i = n + 1; i <= 2 * n; ++i) ref[i] = mainr[i];
          }
        }
        int dc = 0;
        if (mode == 1) {
          int sum = n;
          for (int i = 0; i < n; ++i) sum += left[1 + i] + top[1 + i];
          dc = sum >> (tu.log2 + 1);
        }
        for (int y = 0; y < n; ++y)
          for (int x = 0; x < n; ++x) {
            int v;
            if (mode == 0) {
              v = ((n - 1 - x) * left[1 + y] + (x + 1) * top[1 + n] + (n - 1 - y) * top[1 + x] + (y + 1) * left[1 + n] + n) >> (tu.log2 + 1);
            } else if (mode == 1) {
              v = dc;
              if (c == 0 && n < 32) {
                if (!x && !y) v = (left[1] + 2 * dc + top[1] + 2) >> 2;
                else if (!y) v = (top[1 + x] + 3 * dc + 2) >> 2;
                else if (!x) v = (left[1 + y] + 3 * dc + 2) >> 2;
              }
            } else {
              const int j = vertical ? y : x, i = vertical ? x : y;
              const int idx = ((j + 1) * angle) >> 5, fact = ((j + 1) * angle) & 31;
              v = fact ? ((32 - fact) * ref[i + idx + 1] + fact * ref[i + idx + 2] + 16) >> 5 : ref[i + idx + 1];
              if (c == 0 && n < 32) {
                if (mode == 26 && x == 0) v = clip8(top[1] + ((left[1 + y] - left[0]) >> 1));
                if (mode == 10 && y == 0) v = clip8(left[1] + ((top[1 + x] - top[0]) >> 1));
              }
            }
            const size_t at = static_cast<size_t>(y0 + y) * pw + x0 + x;
            pl[at] = static_cast<uint8_t>(clip8(v + resid[c][at]));
          }
      }
    }
    // deblocking: all vertical edges, then all horizontal ones
    const int w4 = W / 4, h4 = H / 4;
    // which prediction block covers a 4x4 block (8.7.2.4's last rule compares the motion of the two sides of an edge: another
    // reference picture or number of vectors, or a component a whole sample apart)
    std::vector<uint32_t> pu_of(static_cast<size_t>(w4) * h4, 0);
    for (size_t t = 0; t < d.num_pus; ++t)
      for (int yy = 0; yy < pus[t].h; yy += 4)
        for (int xx = 0; xx < pus[t].w; xx += 4) pu_of[static_cast<size_t>((pus[t].y + yy) / 4) * w4 + (pus[t].x + xx) / 4] = static_cast<uint32_t>(t);
    auto motion_differs = [&](const gh::InterPu& a, const gh::InterPu& b) {
      std::vector<std::pair<int, std::pair<int, int>>> va, vb;  // (surface, vector) of every list in use
      for (int l = 0; l < 2; ++l) {
        if ((a.pred_flag >> l) & 1) va.push_back({a.ref_slot[l], {a.mv[l][0], a.mv[l][1]}});
        if ((b.pred_flag >> l) & 1) vb.push_back({b.ref_slot[l], {b.mv[l][0], b.mv[l][1]}});
      }
      if (va.size() != vb.size()) return true;
      auto apart = [](const std::pair<int, int>& u, const std::pair<int, int>& v) { return iabs(u.first - v.first) >= 4 || iabs(u.second - v.second) >= 4; };
      if (va.size() == 1) return va[0].first != vb[0].first || apart(va[0].second, vb[0].second);
      const bool same_set = (va[0].first == vb[0].first && va[1].first == vb[1].first) || (va[0].first == vb[1].first && va[1].first == vb[0].first);
      if (!same_set) return true;
      const bool straight = apart(va[0].second, vb[0].second) || apart(va[1].second, vb[1].second);
      const bool crossed = apart(va[0].second, vb[1].second) || apart(va[1].second, vb[0].second);
      if (va[0].first == va[1].first) return straight && crossed;          // both vectors into one picture: either pairing may match
      return va[0].first == vb[0].first ? straight : crossed;
    };
    if (d.deblock)
      for (int dir = 0; dir < 2; ++dir)
        for (int y = 0; y < h4; ++y)
          for (int x = 0; x < w4; ++x) {
            const gh::EdgeInfo& e = edges[static_cast<size_t>(y) * w4 + x];
            // 8.7.2.4 from what the records say about the two blocks either side: only edges of the 8x8 grid that are block
            // edges and may be filtered; intra on either side: 2; a transform block edge with levels on either side: 1;
            // else the motion of the two prediction blocks decides
            if (((dir ? y : x) & 1) || !(e.edge & (dir ? gh::EDGE_ON_TOP : gh::EDGE_ON_LEFT))) continue;
            const gh::EdgeInfo& nb = edges[static_cast<size_t>(dir ? (y - 1) * w4 + x : y * w4 + x - 1)];
            int bs = 0;
            if ((e.edge & gh::EDGE_INTRA) || (nb.edge & gh::EDGE_INTRA)) bs = 2;
            else if ((e.edge & (dir ? gh::EDGE_TU_TOP : gh::EDGE_TU_LEFT)) && ((e.edge & gh::EDGE_LEVELS) || (nb.edge & gh::EDGE_LEVELS))) bs = 1;
            else {
              const uint32_t pq = pu_of[static_cast<size_t>(y) * w4 + x], pp_ = pu_of[static_cast<size_t>(dir ? (y - 1) * w4 + x : y * w4 + x - 1)];
              if (pq != pp_ && motion_differs(pus[pp_], pus[pq])) bs = 1;
            }
            if (!bs) continue;
            const gh::EdgeInfo& p = edges[static_cast<size_t>(dir ? (y - 1) * w4 + x : y * w4 + x - 1)];
            const gh::SliceLoop& sl = slices[e.slice];
            const int qpl = (e.qp + p.qp + 1) >> 1;
            const bool no_p = p.flags & 1, no_q = e.flags & 1;
            const int beta = hevc_beta_table[clip3(0, 51, qpl + sl.beta_offset)], tc = hevc_tc_table[clip3(0, 53, qpl + 2 * (bs - 1) + sl.tc_offset)];
            {
              uint8_t* q0 = cur.p[0].data() + static_cast<size_t>(y * 4) * W + x * 4;
              const int across = dir ? W : 1, along = dir ? 1 : W;
              auto P = [&](int k, int l) -> uint8_t& { return q0[-(k + 1) * across + l * along]; };
              auto Q = [&](int k, int l) -> uint8_t& { return q0[k * across + l * along]; };
              const int dp0 = iabs(P(2, 0) - 2 * P(1, 0) + P(0, 0)), dp3 = iabs(P(2, 3) - 2 * P(1, 3) + P(0, 3));
              const int dq0 = iabs(Q(2, 0) - 2 * Q(1, 0) + Q(0, 0)), dq3 = iabs(Q(2, 3) - 2 * Q(1, 3) + Q(0, 3));
              if (dp0 + dq0 + dp3 + dq3 < beta) {
                auto strong_line = [&](int l, int dpq) { return 2 * dpq < (beta >> 2) && iabs(P(3, l) - P(0, l)) + iabs(Q(0, l) - Q(3, l)) < (beta >> 3) && iabs(P(0, l) - Q(0, l)) < ((5 * tc + 1) >> 1); };
                const bool strong = strong_line(0, dp0 + dq0) && strong_line(3, dp3 + dq3);
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
              }
            }
            if (bs == 2 && !((dir ? y : x) & 3))
              for (int c = 1; c < 3; ++c) {
                const int qpi = qpl + (c == 1 ? d.cb_qp_offset : d.cr_qp_offset);
                const int qpc = qpi < 30 ? qpi : qpi >= 43 ? qpi - 6 : hevc_qpc_table[qpi - 30];
                const int tcc = hevc_tc_table[clip3(0, 53, qpc + 2 + sl.tc_offset)];
                const int cw = W / 2, across = dir ? cw : 1, along = dir ? 1 : cw;
                uint8_t* q0 = cur.p[c].data() + static_cast<size_t>(y * 2) * cw + x * 2;
                for (int l = 0; l < 2; ++l) {
                  uint8_t* q = q0 + l * along;
                  const int p0 = q[-across], p1 = q[-2 * across], q0v = q[0], q1 = q[across];
                  const int delta = clip3(-tcc, tcc, (((q0v - p0) << 2) + p1 - q1 + 4) >> 3);
                  if (!no_p) q[-across] = static_cast<uint8_t>(clip8(p0 + delta));
                  if (!no_q) q[0] = static_cast<uint8_t>(clip8(q0v - delta));
                }
              }
          }
    // SAO
    if (d.sao) {
      Plane3 src = cur;
      const int wctb = (W + (1 << setup.log2_ctb) - 1) >> setup.log2_ctb;
      for (int c = 0; c < 3; ++c) {
        const int sub = c ? 1 : 0, pw = W >> sub, ph = H >> sub;
        for (int y = 0; y < ph; ++y)
          for (int x = 0; x < pw; ++x) {
            const int xl = x << sub, yl = y << sub;
            const gh::SaoCtb& p = sao[static_cast<size_t>(yl >> setup.log2_ctb) * wctb + (xl >> setup.log2_ctb)];
            const gh::EdgeInfo& e = edges[static_cast<size_t>(yl >> 2) * w4 + (xl >> 2)];
            const gh::SliceLoop& sl = slices[e.slice];
            const int type = (c ? sl.sao_chroma : sl.sao_luma) ? p.type[c] : 0;
            if (!type || (e.flags & 1)) continue;
            const int v = src.p[c][static_cast<size_t>(y) * pw + x];
            int idx = 0;
            if (type == 1) {
              const int k = ((v >> 3) - p.band_pos[c]) & 31;
              idx = k < 4 ? k + 1 : 0;
            } else {
              static const int hx[4][2] = {{-1, 1}, {0, 0}, {-1, 1}, {1, -1}}, vy[4][2] = {{0, 0}, {-1, 1}, {-1, 1}, {-1, 1}};
              const int cl = p.eo_class[c];
              int s = 0;
              bool ok = true;
              for (int k = 0; k < 2 && ok; ++k) {
                const int xn = x + hx[cl][k], yn = y + vy[cl][k];
                if (xn < 0 || yn < 0 || xn >= pw || yn >= ph) { ok = false; break; }
                const gh::EdgeInfo& en = edges[static_cast<size_t>((yn << sub) >> 2) * w4 + ((xn << sub) >> 2)];
                if (en.slice != e.slice && !(slices[en.slice].first_ctb > sl.first_ctb ? slices[en.slice].across : sl.across)) { ok = false; break; }
                const int nv = src.p[c][static_cast<size_t>(yn) * pw + xn];
                s += (v > nv) - (v < nv);
              }
              if (ok) idx = s == -2 ? 1 : s == -1 ? 2 : s == 1 ? 3 : s == 2 ? 4 : 0;
            }
            if (idx) cur.p[c][static_cast<size_t>(y) * pw + x] = static_cast<uint8_t>(clip8(v + p.offset[c][idx - 1]));
          }
      }
    }
    return true;
  }

  void output_picture(int slot) override {
    std::vector<uint8_t> o;
    if (slot >= 0) {
      const int w = W - setup.crop_left - setup.crop_right, h = H - setup.crop_top - setup.crop_bottom;
      const Plane3& s = surf[slot];
      for (int y = 0; y < h; ++y) o.insert(o.end(), s.p[0].begin() + static_cast<long>(y + setup.crop_top) * W + setup.crop_left, s.p[0].begin() + static_cast<long>(y + setup.crop_top) * W + setup.crop_left + w);
      for (int c = 1; c < 3; ++c)
        for (int y = 0; y < (h + 1) / 2; ++y) {
          const long at = static_cast<long>(y + setup.crop_top / 2) * (W / 2) + setup.crop_left / 2;
          o.insert(o.end(), s.p[c].begin() + at, s.p[c].begin() + at + (w + 1) / 2);
        }
    }
    outputs.push_back(o);
  }
  void release_surface(int slot) override {
    if (slot >= 0) held[slot] = 0;
  }
  int missing_surface(std::string* error) override {
    int slot = -1;
    for (int i = 0; i < gh::kMaxSurfaces && slot < 0; ++i)
      if (!held[i]) slot = i;
    if (slot < 0) {
      *error = "no free surface";
      return -1;
    }
    held[slot] = 1;
    surf[slot].p[0].assign(static_cast<size_t>(W) * H, 128);
    surf[slot].p[1].assign(static_cast<size_t>(W) * H / 4, 128);
    surf[slot].p[2].assign(static_cast<size_t>(W) * H / 4, 128);
    return slot;
  }
};

static std::vector<uint8_t> slurp(const char* path) {
  std::ifstream f(path, std::ios::binary);
  return std::vector<uint8_t>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}

int main(int argc, char** argv) {
  if (argc < 5) return 2;
  const std::vector<uint8_t> hvcc = slurp(argv[1]), blob = slurp(argv[2]);
  std::vector<size_t> sizes;
  {
    std::ifstream f(argv[3]);
    size_t v;
    while (f >> v) sizes.push_back(v);
  }
  CpuSink sink;
  hevc::Parser parser(&sink);
  if (!parser.configure(hvcc.data(), hvcc.size())) {
    std::printf("configure: %s\n", parser.error().c_str());
    return 1;
  }
  std::vector<uint64_t> wanted, keyframes;   // [wanted ordinals ...] ["kf" closed-GOP sync sample ordinals ...]
  bool kf = false;
  for (int i = 5; i < argc; ++i) {
    if (std::string(argv[i]) == "kf") kf = true;
    else (kf ? keyframes : wanted).push_back(std::strtoull(argv[i], nullptr, 10));
  }
  if (!wanted.empty()) parser.set_wanted(wanted);
  if (!keyframes.empty()) parser.set_keyframes(keyframes);
  size_t at = 0;
  for (size_t i = 0; i < sizes.size(); ++i) {
    if (!parser.decode(blob.data() + at, sizes[i])) {
      std::printf("sample %zu: %s\n", i, parser.error().c_str());
      return 1;
    }
    at += sizes[i];
  }
  parser.flush();
  std::ofstream out(argv[4], std::ios::binary);
  const size_t w = static_cast<size_t>(parser.width()), h = static_cast<size_t>(parser.height());
  const size_t fs = w * h + 2 * ((w + 1) / 2) * ((h + 1) / 2);
  int skipped = 0;
  for (const auto& o : sink.outputs) {
    if (o.empty()) {
      std::vector<uint8_t> z(fs, 0);
      out.write(reinterpret_cast<const char*>(z.data()), static_cast<std::streamsize>(fs));
      ++skipped;
    } else {
      out.write(reinterpret_cast<const char*>(o.data()), static_cast<std::streamsize>(o.size()));
    }
  }
  std::printf("frames=%zu skipped_positions=%d decoded=%lld skipped=%lld res=%lld levels=%lld pus=%lld intra=%lld\n", sink.outputs.size(), skipped,
              static_cast<long long>(parser.pictures_decoded()), static_cast<long long>(parser.pictures_skipped()), static_cast<long long>(sink.n_res),
              static_cast<long long>(sink.n_coef), static_cast<long long>(sink.n_pu), static_cast<long long>(sink.n_intra));
  return 0;
}
