// TEST FIXTURE, not part of the product: feeds the HOST half of the H.264 path (hwang_amd/csrc/h264/h264_parser.cpp)
// damaged samples - flipped bits, overwritten bytes, bursts, truncated and swapped samples, damaged parameter sets - and
// checks (tests/test_h264_fuzz.py builds it with -fsanitize=address,undefined):
//   * the parser stays inside its memory and ends every sample with `true` or with an error text;
//   * every record it hands to the GPU kernels (amd/frame_job.h) is one they can follow without leaving their buffers:
//     coefficient runs inside the picture's level stream, level groups that exist, prediction modes the tables have,
//     reference surfaces the picture really holds, list indices inside the weight tables, the compact motion stream's
//     row table and length.  k_recon.hip trusts the records; a damaged stream must not produce one that points outside.
// Both hand-over forms: full MbMotion records, and PMotion + extra differences (k_mv derives the vectors).
//
//   h264_parser_fuzz <avcC file> <samples blob> <sizes file> <seed> <iterations> [damage kind 0..7]
#include "../../hwang_amd/csrc/h264/h264_parser.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iterator>
#include <string>
#include <vector>

using namespace hwang;

static std::vector<uint8_t> slurp(const char* path) {
  std::ifstream f(path, std::ios::binary);
  return std::vector<uint8_t>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}

struct CheckingSink : public h264::PictureSink {
  std::vector<amd::MbInfo> mb;
  std::vector<amd::MbMotion> mo[amd::kMaxSurfaces];
  std::vector<amd::PMotion> pmo[amd::kMaxSurfaces];
  std::vector<uint32_t> pmx[amd::kMaxSurfaces];
  bool busy[amd::kMaxSurfaces] = {false};
  std::vector<uint32_t> coefs;
  std::vector<amd::SliceParams> sl = std::vector<amd::SliceParams>(64);
  bool compact = false;
  int mbw = 0, mbh = 0, cur_slot = -1;
  long long pictures = 0;
  std::string bad;

  bool begin_picture(const h264::PicSetup& s, h264::PicBuffers* out, std::string* error) override {
    if (s.mb_width <= 0 || s.mb_height <= 0 || s.mb_width > amd::kMaxMbRows || s.mb_height > amd::kMaxMbRows) {
      bad = "begin_picture: " + std::to_string(s.mb_width) + " x " + std::to_string(s.mb_height) + " macroblocks";
      *error = bad;
      return false;
    }
    mbw = s.mb_width;
    mbh = s.mb_height;
    const size_t nmb = static_cast<size_t>(mbw) * mbh;
    int slot = -1;
    for (int i = 0; i < amd::kMaxSurfaces && slot < 0; ++i)
      if (!busy[i]) slot = i;
    if (slot < 0) {
      *error = "out of surfaces";
      return false;
    }
    busy[slot] = true;
    cur_slot = slot;
    // exactly the capacities the backend gives (amd_video_decoder.cpp), not a word more
    mb.assign(nmb, amd::MbInfo());
    mo[slot].assign(nmb, amd::MbMotion());
    coefs.assign(nmb * amd::kMaxSparseCoefs, 0);
    out->mb = mb.data();
    out->motion = mo[slot].data();
    out->coefs = coefs.data();
    out->coef_capacity = coefs.size();
    out->slices = sl.data();
    out->max_slices = 64;
    out->slot = slot;
    out->device_mv = compact;
    if (compact) {
      pmo[slot].assign(nmb, amd::PMotion());
      pmx[slot].assign(static_cast<size_t>(mbh) + nmb * 15, 0);
      out->pmo = pmo[slot].data();
      out->pmx = pmx[slot].data();
      out->pmx_capacity = pmx[slot].size();
    }
    return true;
  }

  bool check(bool ok, const char* what, long long a = 0, long long b = 0) {
    if (!ok && bad.empty()) bad = std::string(what) + " (" + std::to_string(a) + ", " + std::to_string(b) + ")";
    return ok;
  }

  bool end_picture(const h264::PicDone& d, std::string* error) override {
    ++pictures;
    const size_t nmb = static_cast<size_t>(mbw) * mbh;
    check(d.slot == cur_slot, "end_picture: slot", d.slot, cur_slot);
    check(d.num_slices >= 1 && d.num_slices <= 64, "slices", d.num_slices);
    check(d.coef_count <= coefs.size(), "level stream length", static_cast<long long>(d.coef_count));
    check(d.num_ref_slots >= 0 && d.num_ref_slots <= 32, "reference surfaces", d.num_ref_slots);
    bool is_ref[256] = {false};
    for (int i = 0; i < d.num_ref_slots && i < 32; ++i) {
      const int s = d.ref_slots[i];
      if (check(s >= 0 && s < amd::kMaxSurfaces && busy[s], "reference surface", s)) is_ref[s] = true;
    }
    if (d.compact_motion) check(compact && d.pmx_count >= static_cast<size_t>(mbh) && d.pmx_count <= pmx[d.slot].size(), "compact stream length", static_cast<long long>(d.pmx_count));
    size_t px = 0;  // position in the extra stream of the compact macroblocks of the current row
    for (size_t i = 0; i < nmb && bad.empty(); ++i) {
      const amd::MbInfo& m = mb[i];
      check(m.kind <= amd::MBK_INTER, "macroblock kind", m.kind);
      check(m.slice < d.num_slices, "macroblock's slice", m.slice, d.num_slices);
      check(m.qp <= 51 && m.qpc[0] <= 51 && m.qpc[1] <= 51, "quantiser", m.qp, m.qpc[0]);
      // the level stream
      if (m.kind == amd::MBK_IPCM) {
        check(static_cast<size_t>(m.coef_offset) + 96 <= d.coef_count, "I_PCM samples past the level stream", m.coef_offset);
      } else if (m.ncoef == amd::kDenseCoefs) {
        check(!(m.coded_mask >> 26), "coded_mask has groups that do not exist", m.coded_mask);
        check(static_cast<size_t>(m.coef_offset) + 8 * static_cast<size_t>(__builtin_popcount(m.coded_mask)) <= d.coef_count, "dense groups past the level stream", m.coef_offset);
      } else {
        check(m.ncoef <= amd::kMaxSparseCoefs, "sparse level count", m.ncoef);
        check(!(m.coded_mask >> 26), "coded_mask has groups that do not exist", m.coded_mask);
        if (check(static_cast<size_t>(m.coef_offset) + m.ncoef <= d.coef_count, "sparse levels past the level stream", m.coef_offset, m.ncoef))
          for (unsigned k = 0; k < m.ncoef && bad.empty(); ++k) {
            const uint32_t at = coefs[m.coef_offset + k] >> 16;
            check(at < 26 * 16 && ((m.coded_mask >> (at >> 4)) & 1), "level entry in a group the mask does not have", at, m.coded_mask);
          }
      }
      // prediction modes the kernels' tables have
      if (m.kind == amd::MBK_I16x16) check((m.pred_modes & 15) <= 3, "Intra16x16 mode", m.pred_modes & 15);
      if (m.kind <= amd::MBK_I16x16) check((m.pred_modes >> 4) <= 3, "chroma mode", m.pred_modes >> 4);
      if (m.kind == amd::MBK_I4x4 || m.kind == amd::MBK_I8x8)
        for (int k = 0; k < 16; ++k) check(((m.i4_modes[k >> 1] >> (4 * (k & 1))) & 15) <= 8, "Intra4x4 / 8x8 mode", k);
      check(m.mv_shape <= 3, "partition shape", m.mv_shape);
      {  // neighbours the kernels read on the strength of these flags must exist
        const int mbx = static_cast<int>(i % static_cast<size_t>(mbw)), mby = static_cast<int>(i / static_cast<size_t>(mbw));
        const bool a = mbx > 0, b = mby > 0, c = mby > 0 && mbx < mbw - 1, dd = mbx > 0 && mby > 0;
        check(!(m.flags & amd::MBF_AVAIL_A) || a, "left neighbour outside the picture", mbx, mby);
        check(!(m.flags & amd::MBF_AVAIL_B) || b, "top neighbour outside the picture", mbx, mby);
        check(!(m.flags & amd::MBF_AVAIL_C) || c, "top-right neighbour outside the picture", mbx, mby);
        check(!(m.flags & amd::MBF_AVAIL_D) || dd, "top-left neighbour outside the picture", mbx, mby);
        check(!(m.flags & amd::MBF_DB_LEFT) || a, "left edge filtered at the picture's edge", mbx, mby);
        check(!(m.flags & amd::MBF_DB_TOP) || b, "top edge filtered at the picture's edge", mbx, mby);
        check(!(m.mv_flags & amd::MVF_AVAIL_A) || a, "left vector neighbour outside the picture", mbx, mby);
        check(!(m.mv_flags & amd::MVF_AVAIL_B) || b, "top vector neighbour outside the picture", mbx, mby);
        check(!(m.mv_flags & amd::MVF_AVAIL_C) || c, "top-right vector neighbour outside the picture", mbx, mby);
        check(!(m.mv_flags & amd::MVF_AVAIL_D) || dd, "top-left vector neighbour outside the picture", mbx, mby);
      }
      // motion
      if (compact && (i % static_cast<size_t>(mbw)) == 0 && d.compact_motion) {
        px = pmx[d.slot][i / static_cast<size_t>(mbw)];
        check(px >= static_cast<size_t>(mbh) && px <= d.pmx_count, "row table entry outside the compact stream", static_cast<long long>(px));
      }
      if (m.kind == amd::MBK_INTER) {
        if (m.mv_flags & amd::MVF_COMPACT) {
          if (!check(compact && d.compact_motion, "compact macroblock in a picture without a compact stream")) break;
          const amd::PMotion& pm = pmo[d.slot][i];
          for (int b = 0; b < 4; ++b) {
            check(pm.ref_idx[b] >= 0 && pm.ref_idx[b] < 32, "compact reference index", pm.ref_idx[b]);
            const int s = static_cast<uint8_t>(pm.ref_slot[b]);
            check(s < amd::kMaxSurfaces && is_ref[s], "compact reference surface", s);
          }
          if ((m.mv_flags & amd::MVF_DEFERRED) && !(m.mv_flags & amd::MVF_PSKIP)) {
            int parts = m.mv_shape == 0 ? 1 : 2;
            if (m.mv_shape == 3) {
              parts = 0;
              for (int q = 0; q < 4; ++q) {
                const int ss = (m.mv_sub >> (2 * q)) & 3;
                parts += ss == 0 ? 1 : ss == 3 ? 4 : 2;
              }
            }
            px += static_cast<size_t>(parts - 1);
            check(px <= d.pmx_count, "extra differences past the compact stream", static_cast<long long>(px), static_cast<long long>(d.pmx_count));
          }
        } else {
          check(d.full_motion || !d.has_inter || !compact, "full motion record in a picture that says it has none");
          const amd::MbMotion& mm = mo[d.slot][i];
          for (int b = 0; b < 4; ++b) {
            bool any = false;
            for (int l = 0; l < 2; ++l) {
              const int s = static_cast<uint8_t>(mm.ref_slot[l][b]);
              if (s == 0xff) continue;
              any = true;
              check(s < amd::kMaxSurfaces && is_ref[s], "reference surface of a prediction", s);
              check(mm.ref_idx[l][b] >= 0 && mm.ref_idx[l][b] < 32, "reference index of a prediction", mm.ref_idx[l][b]);
            }
            check(any, "an 8x8 block of an inter macroblock predicts from nothing", static_cast<long long>(i), b);
          }
        }
      }
    }
    if (!bad.empty()) {
      *error = bad;
      return false;
    }
    return true;
  }
  void output_picture(int) override {}
  void release_surface(int s) override {
    if (s >= 0 && s < amd::kMaxSurfaces) busy[s] = false;
  }
};

static uint64_t rng_state;
static uint32_t rnd() {
  rng_state = rng_state * 6364136223846793005ull + 1442695040888963407ull;
  return static_cast<uint32_t>(rng_state >> 33);
}

int main(int argc, char** argv) {
  if (argc < 6) return 2;
  const std::vector<uint8_t> avcc0 = slurp(argv[1]), blob = slurp(argv[2]);
  std::vector<size_t> sizes;
  {
    std::ifstream f(argv[3]);
    size_t v;
    while (f >> v) sizes.push_back(v);
  }
  rng_state = std::strtoull(argv[4], nullptr, 10) * 2654435761u + 1;
  const int iterations = std::atoi(argv[5]);
  const int only_kind = argc > 6 ? std::atoi(argv[6]) : -1;
  long long pictures = 0, errors = 0;
  for (int it = 0; it < iterations; ++it) {
    std::vector<std::vector<uint8_t>> samples;
    size_t at = 0;
    for (size_t s : sizes) {
      samples.emplace_back(blob.begin() + at, blob.begin() + at + s);
      at += s;
    }
    std::vector<uint8_t> avcc = avcc0;
    const int kind = only_kind >= 0 ? only_kind : static_cast<int>(rnd() % 8);
    const int hits = 1 + static_cast<int>(rnd() % 6);
    for (int h = 0; h < hits; ++h) {
      std::vector<uint8_t>& s = samples[rnd() % samples.size()];
      if (s.size() < 8) continue;
      const size_t pos = 4 + rnd() % (s.size() - 4);
      switch (kind) {
        case 0: s[pos] ^= static_cast<uint8_t>(1u << (rnd() % 8)); break;
        case 1: s[pos] = static_cast<uint8_t>(rnd()); break;
        case 2: for (size_t k = pos; k < s.size() && k < pos + 16; ++k) s[k] = static_cast<uint8_t>(rnd()); break;
        case 3: s.resize(pos); break;
        case 4: std::swap(s, samples[rnd() % samples.size()]); break;
        case 5: for (size_t k = pos; k < s.size(); ++k) s[k] = 0; break;
        case 6: s[4 + rnd() % 4] = static_cast<uint8_t>(rnd()); break;   // the NAL header / first slice header bytes
        default: if (avcc.size() > 10) avcc[8 + rnd() % (avcc.size() - 8)] ^= static_cast<uint8_t>(1u << (rnd() % 8)); break;  // a parameter set
      }
    }
    for (int form = 0; form < 2; ++form) {
      CheckingSink sink;
      sink.compact = form != 0;
      h264::Parser parser(&sink);
      bool failed = !parser.configure(avcc.data(), avcc.size());
      for (size_t i = 0; i < samples.size() && !failed; ++i) failed = !parser.decode(samples[i].data(), samples[i].size());
      if (!failed) failed = !parser.flush();
      if (!sink.bad.empty()) {
        std::printf("iteration %d (damage kind %d, %s records): a record the kernels must not see: %s\n", it, kind, form ? "compact" : "full", sink.bad.c_str());
        return 1;
      }
      if (failed && parser.error().empty()) {
        std::printf("iteration %d: failed without an error text\n", it);
        return 1;
      }
      errors += failed;
      pictures += sink.pictures;
    }
  }
  std::printf("ok: %d damaged streams in both hand-over forms, %lld pictures checked, %lld runs ended in an error\n", iterations, pictures, errors);
  return 0;
}
