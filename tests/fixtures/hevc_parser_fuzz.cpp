// TEST FIXTURE, not part of the product: feeds the HOST half of the HEVC path (hwang_amd/csrc/hevc/hevc_parser.cpp)
// damaged samples - flipped bits, overwritten bytes, truncated and swapped samples, damaged parameter sets - and checks
// two things (tests/test_hevc_fuzz.py builds it with -fsanitize=address,undefined):
//   * the parser itself stays inside its memory and ends every sample with `true` or with an error text;
//   * every record it hands to the GPU kernels (amd/hevc_frame_job.h) stays inside the picture, its lists and its
//     surfaces - k_hevc.hip trusts the records, so a damaged stream must never produce one that points outside.
//
//   hevc_parser_fuzz <hvcc file> <samples blob> <sizes file> <seed> <iterations> [damage kind 0..7, default: all in turn]
// Prints "ok: <iterations> damaged streams, <n> pictures checked, <m> ended in an error"; exit code 1 on a bad record.
#include "../../hwang_amd/csrc/hevc/hevc_parser.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iterator>
#include <string>
#include <vector>

using namespace hwang;
namespace gh = hwang::amd::hevc;

static std::vector<uint8_t> slurp(const char* path) {
  std::ifstream f(path, std::ios::binary);
  return std::vector<uint8_t>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}

struct CheckingSink : public hevc::PictureSink {
  hevc::PicSetup setup;
  std::vector<int> held = std::vector<int>(gh::kMaxSurfaces, 0);
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
  int W = 0, H = 0, cur_slot = -1;
  long long pictures = 0;
  std::string bad;

  bool begin_picture(const hevc::PicSetup& s, hevc::PicBuffers* out, std::string* error) override {
    setup = s;
    W = s.width;
    H = s.height;
    if (W <= 0 || H <= 0 || (W & 7) || (H & 7) || s.log2_ctb < 4 || s.log2_ctb > 6 || W > 8192 || H > 8192) {
      bad = "begin_picture: geometry " + std::to_string(W) + "x" + std::to_string(H) + " ctb " + std::to_string(s.log2_ctb);
      *error = bad;
      return false;
    }
    const size_t w4h4 = static_cast<size_t>(W / 4) * (H / 4), ctbs = static_cast<size_t>((W + (1 << s.log2_ctb) - 1) >> s.log2_ctb) * ((H + (1 << s.log2_ctb) - 1) >> s.log2_ctb);
    // exactly the capacities the backend gives (hevc_video_decoder.cpp: ensure_geometry), and not a record more
    res.assign(w4h4 * 3 / 2 + 64, gh::ResTu());
    coefs.assign(static_cast<size_t>(W) * H * 3 / 2 + 64, 0);
    pus.assign(w4h4 / 2 + 64, gh::InterPu());
    weights.assign(2 * 16 * static_cast<size_t>(gh::kMaxSlices), gh::WeightEntry());
    intra[0].assign(w4h4 + 64, gh::IntraTu());
    intra[1].assign(w4h4 / 4 + 64, gh::IntraTu());
    intra[2].assign(w4h4 / 4 + 64, gh::IntraTu());
    for (int c = 0; c < 3; ++c) first[c].assign(ctbs + 1, 0);
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
    cur_slot = slot;
    out->slot = slot;
    out->cookie = nullptr;
    out->res = res.data();
    out->cap_res = res.size();
    out->coefs = coefs.data();
    out->cap_coefs = coefs.size();
    out->pus = pus.data();
    out->cap_pus = pus.size();
    out->weights = weights.data();
    out->cap_weights = weights.size();
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

  bool check(bool ok, const char* what, long long a = 0, long long b = 0) {
    if (!ok && bad.empty()) bad = std::string(what) + " (" + std::to_string(a) + ", " + std::to_string(b) + ")";
    return ok;
  }

  bool end_picture(const hevc::PicDone& d, std::string* error) override {
    ++pictures;
    const size_t ctbs = sao.size();
    check(d.slot == cur_slot, "end_picture: slot", d.slot, cur_slot);
    check(d.num_res <= res.size() && d.num_coefs <= coefs.size() && d.num_pus <= pus.size() && d.num_weights <= weights.size(), "list sizes");
    check(d.num_slices >= 0 && d.num_slices <= gh::kMaxSlices, "slices", d.num_slices);
    check(d.num_ref_slots >= 0 && d.num_ref_slots <= gh::kMaxSurfaces, "reference slots", d.num_ref_slots);
    bool is_ref[gh::kMaxSurfaces] = {false};
    for (int i = 0; i < d.num_ref_slots && i < gh::kMaxSurfaces; ++i) {
      const int s = d.ref_slots[i];
      if (check(s >= 0 && s < gh::kMaxSurfaces && held[s] && s != d.slot, "reference slot", s)) is_ref[s] = true;
    }
    for (size_t t = 0; t < d.num_res && bad.empty(); ++t) {
      const gh::ResTu& r = res[t];
      const int n = 1 << r.log2, pw = W >> (r.c_idx ? 1 : 0), ph = H >> (r.c_idx ? 1 : 0);
      check(r.log2 >= 2 && r.log2 <= 5 && r.c_idx <= 2, "transform block size / plane", r.log2, r.c_idx);
      check(r.x + n <= pw && r.y + n <= ph, "transform block position", r.x, r.y);
      check(r.qp <= 57 + 6 && (r.matrix == 0xff || r.matrix < 6), "transform block qp / matrix", r.qp, r.matrix);
      check(static_cast<size_t>(r.coef_offset) + r.ncoef <= d.num_coefs, "level run", r.coef_offset, r.ncoef);
      for (uint32_t k = 0; k < r.ncoef && bad.empty(); ++k) check((coefs[r.coef_offset + k] >> 16) < static_cast<uint32_t>(n * n), "level position", coefs[r.coef_offset + k] >> 16, n);
    }
    for (size_t t = 0; t < d.num_pus && bad.empty(); ++t) {
      const gh::InterPu& p = pus[t];
      check(p.w >= 4 && p.h >= 4 && p.w <= 64 && p.h <= 64 && p.x + p.w <= W && p.y + p.h <= H, "prediction block", p.x, p.y);
      check(p.pred_flag >= 1 && p.pred_flag <= 3, "prediction lists", p.pred_flag);
      for (int l = 0; l < 2; ++l)
        if ((p.pred_flag >> l) & 1) {
          check(p.ref_slot[l] < gh::kMaxSurfaces && is_ref[p.ref_slot[l]], "prediction block's reference surface", p.ref_slot[l]);
          if (p.weighted) check(p.wp[l] < d.num_weights, "weight entry", p.wp[l], static_cast<long long>(d.num_weights));
        }
    }
    for (int c = 0; c < 3; ++c) {
      check(d.num_intra[c] <= intra[c].size(), "intra list size", c);
      const int pw = W >> (c ? 1 : 0), ph = H >> (c ? 1 : 0);
      for (size_t t = 0; t < d.num_intra[c] && bad.empty(); ++t) {
        const gh::IntraTu& u = intra[c][t];
        const int n = 1 << u.log2;
        check(u.log2 >= 2 && u.log2 <= 5 && (u.mode <= 34 || u.mode == gh::kIntraPcm), "intra block size / mode", u.log2, u.mode);
        check(u.x + n <= pw && u.y + n <= ph, "intra block position", u.x, u.y);
        const int nu = 2 * n / (c ? 2 : 4);
        check(nu >= 32 || (!(u.avail_left >> nu) && !(u.avail_top >> nu)), "availability bits past the block's 2n samples");
        check(!(u.wait_left & ~u.avail_left) && !(u.wait_top & ~u.avail_top), "waits for an unavailable neighbour");
        // available neighbours lie inside the picture
        for (int k = 0; k < nu; ++k) {
          if ((u.avail_left >> k) & 1) check(u.x > 0 && u.y + k * (c ? 2 : 4) < ph, "left neighbour outside", u.x, u.y);
          if ((u.avail_top >> k) & 1) check(u.y > 0 && u.x + k * (c ? 2 : 4) < pw, "top neighbour outside", u.x, u.y);
        }
        if (u.flags & gh::INTRA_CORNER) check(u.x > 0 && u.y > 0, "corner outside");
      }
      // the per-block offsets: monotone, ending at the list's length, every record in the block its offset names
      if (d.num_intra[c] && bad.empty()) {
        check(first[c][0] == 0 && first[c][ctbs] == d.num_intra[c], "block offsets' ends", first[c][0], first[c][ctbs]);
        const int wctb = (W + (1 << setup.log2_ctb) - 1) >> setup.log2_ctb;
        for (size_t k = 0; k < ctbs && bad.empty(); ++k) {
          check(first[c][k] <= first[c][k + 1], "block offsets not monotone", static_cast<long long>(k));
          for (uint32_t t = first[c][k]; t < first[c][k + 1] && t < d.num_intra[c] && bad.empty(); ++t) {
            const gh::IntraTu& u = intra[c][t];
            const size_t at = static_cast<size_t>((u.y << (c ? 1 : 0)) >> setup.log2_ctb) * wctb + ((u.x << (c ? 1 : 0)) >> setup.log2_ctb);
            check(at == k, "record listed under another block", static_cast<long long>(at), static_cast<long long>(k));
          }
        }
      }
    }
    for (size_t i = 0; i < edges.size() && bad.empty(); ++i) {
      check(d.num_slices == 0 || edges[i].slice < d.num_slices, "edge record's slice", edges[i].slice, d.num_slices);
      check(edges[i].qp >= -64 && edges[i].qp <= 64, "edge record's qp", edges[i].qp);
    }
    for (size_t k = 0; k < ctbs && bad.empty(); ++k)
      for (int c = 0; c < 3; ++c) check(sao[k].type[c] <= 2 && sao[k].band_pos[c] <= 31 && sao[k].eo_class[c] <= 3, "SAO parameters", static_cast<long long>(k), c);
    if (!bad.empty()) {
      *error = bad;
      return false;
    }
    return true;
  }
  void output_picture(int) override {}
  void release_surface(int slot) override {
    if (slot >= 0 && slot < gh::kMaxSurfaces) held[slot] = 0;
  }
  int missing_surface(std::string* error) override {
    for (int i = 0; i < gh::kMaxSurfaces; ++i)
      if (!held[i]) {
        held[i] = 1;
        return i;
      }
    *error = "no free surface";
    return -1;
  }
};

static uint64_t rng_state;
static uint32_t rnd() {
  rng_state = rng_state * 6364136223846793005ull + 1442695040888963407ull;
  return static_cast<uint32_t>(rng_state >> 33);
}

int main(int argc, char** argv) {
  if (argc < 6) return 2;
  const std::vector<uint8_t> hvcc0 = slurp(argv[1]), blob = slurp(argv[2]);
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
    std::vector<uint8_t> hvcc = hvcc0;
    const int kind = only_kind >= 0 ? only_kind : static_cast<int>(rnd() % 8);
    const int hits = 1 + static_cast<int>(rnd() % 6);
    for (int h = 0; h < hits; ++h) {
      std::vector<uint8_t>& s = samples[rnd() % samples.size()];
      if (s.size() < 8) continue;
      const size_t pos = 4 + rnd() % (s.size() - 4);  // (the length prefix stays: the container's business)
      switch (kind) {
        case 0: s[pos] ^= static_cast<uint8_t>(1u << (rnd() % 8)); break;                       // a bit
        case 1: s[pos] = static_cast<uint8_t>(rnd()); break;                                    // a byte
        case 2: for (size_t k = pos; k < s.size() && k < pos + 16; ++k) s[k] = static_cast<uint8_t>(rnd()); break;  // a burst
        case 3: s.resize(pos); break;                                                           // truncated (the prefix now lies)
        case 4: std::swap(s, samples[rnd() % samples.size()]); break;                           // out of order
        case 5: for (size_t k = pos; k < s.size(); ++k) s[k] = 0; break;                        // zeros to the end
        case 6: s[4 + rnd() % 3] = static_cast<uint8_t>(rnd()); break;                          // the NAL header / first slice header bytes
        default: if (hvcc.size() > 30) hvcc[23 + rnd() % (hvcc.size() - 23)] ^= static_cast<uint8_t>(1u << (rnd() % 8)); break;  // a parameter set
      }
    }
    CheckingSink sink;
    hevc::Parser parser(&sink);
    bool failed = !parser.configure(hvcc.data(), hvcc.size());
    for (size_t i = 0; i < samples.size() && !failed; ++i) failed = !parser.decode(samples[i].data(), samples[i].size());
    if (!failed) parser.flush();
    if (!sink.bad.empty()) {
      std::printf("iteration %d (damage kind %d): a record the kernels must not see: %s\n", it, kind, sink.bad.c_str());
      return 1;
    }
    if (failed && parser.error().empty()) {
      std::printf("iteration %d: failed without an error text\n", it);
      return 1;
    }
    errors += failed;
    pictures += sink.pictures;
  }
  std::printf("ok: %d damaged streams, %lld pictures checked, %lld ended in an error\n", iterations, pictures, errors);
  return 0;
}
