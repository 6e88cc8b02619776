// Host-only timing of the HEVC parser: what a new interval costs (construct + configure), the intra picture, the rest.
#include "../../../hwang_amd/csrc/hevc/hevc_parser.h"
#include <chrono>
#include <cstdio>
#include <fstream>
#include <iterator>
#include <vector>
using namespace hwang;
namespace gh = hwang::amd::hevc;
struct NullSink : hevc::PictureSink {
  std::vector<gh::ResTu> res; std::vector<uint32_t> coefs; std::vector<gh::InterPu> pus; std::vector<gh::WeightEntry> weights;
  std::vector<gh::IntraTu> intra[3]; std::vector<uint32_t> first[3]; std::vector<gh::EdgeInfo> edges; std::vector<gh::SaoCtb> sao;
  std::vector<gh::SliceLoop> slices = std::vector<gh::SliceLoop>(255); std::vector<uint8_t> scaling = std::vector<uint8_t>(4064);
  int held[32] = {0}; int n_out = 0;
  bool begin_picture(const hevc::PicSetup& s, hevc::PicBuffers* out, std::string*) override {
    const size_t w4h4 = (size_t)(s.width / 4) * (s.height / 4), ctbs = (size_t)((s.width + 63) >> s.log2_ctb) * ((s.height + 63) >> s.log2_ctb) + 64;
    if (res.empty()) { res.resize(w4h4 * 3 / 2 + 64); coefs.resize((size_t)s.width * s.height * 3 / 2 + 64); pus.resize(w4h4 / 2 + 64); weights.resize(8192);
      for (int c = 0; c < 3; ++c) { intra[c].resize(w4h4 + 64); first[c].resize(ctbs * 20 + 1); } edges.resize(w4h4); sao.resize(ctbs * 20); }
    int slot = 0; while (held[slot]) ++slot; held[slot] = 1;
    out->slot = slot; out->res = res.data(); out->cap_res = res.size(); out->coefs = coefs.data(); out->cap_coefs = coefs.size();
    out->pus = pus.data(); out->cap_pus = pus.size(); out->weights = weights.data(); out->cap_weights = weights.size();
    for (int c = 0; c < 3; ++c) { out->intra[c] = intra[c].data(); out->cap_intra[c] = intra[c].size(); out->intra_ctb_first[c] = first[c].data(); }
    out->edges = edges.data(); out->sao = sao.data(); out->slices = slices.data(); out->scaling = scaling.data();
    return true;
  }
  bool end_picture(const hevc::PicDone&, std::string*) override { return true; }
  void output_picture(int) override { ++n_out; }
  void release_surface(int s) override { if (s >= 0) held[s] = 0; }
};
static std::vector<uint8_t> slurp(const char* p) { std::ifstream f(p, std::ios::binary); return std::vector<uint8_t>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>()); }
int main(int argc, char** argv) {
  auto hvcc = slurp(argv[1]), blob = slurp(argv[2]);
  std::vector<size_t> sizes; { std::ifstream f(argv[3]); size_t v; while (f >> v) sizes.push_back(v); }
  const int reps = argc > 4 ? atoi(argv[4]) : 3;
  for (int r = 0; r < reps; ++r) {
    NullSink sink;
    auto t0 = std::chrono::steady_clock::now();
    auto ms_since = [&]() { const auto t1 = std::chrono::steady_clock::now(); const double ms = std::chrono::duration<double, std::milli>(t1 - t0).count(); t0 = t1; return ms; };
    hevc::Parser* p = new hevc::Parser(&sink);
    if (!p->configure(hvcc.data(), hvcc.size())) { printf("%s\n", p->error().c_str()); return 1; }
    const double t_conf = ms_since();
    size_t at = 0;
    double t_first = 0, t_rest = 0;
    for (size_t i = 0; i < sizes.size(); ++i) {
      if (!p->decode(blob.data() + at, sizes[i])) { printf("%s\n", p->error().c_str()); return 1; }
      at += sizes[i];
      (i == 0 ? t_first : t_rest) += ms_since();
    }
    p->flush();
    const double t_flush = ms_since();
    delete p;
    const double t_del = ms_since();
    printf("configure %.3f ms, first picture %.3f ms, others %.3f ms each, flush %.3f, destroy %.3f\n", t_conf, t_first, t_rest / (sizes.size() - 1), t_flush, t_del);
  }
}
