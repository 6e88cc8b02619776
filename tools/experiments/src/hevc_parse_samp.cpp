#include "../../../hwang_amd/csrc/hevc/hevc_parser.h"
#include <signal.h>
#include <sys/time.h>
#include <ucontext.h>
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
static unsigned long long g_pc[1<<20]; static volatile int g_n;
static void on_prof(int, siginfo_t*, void* uc) { int i = __atomic_fetch_add(&g_n, 1, __ATOMIC_RELAXED); if (i < (1<<20)) g_pc[i] = ((ucontext_t*)uc)->uc_mcontext.gregs[REG_RIP]; }
int main(int argc, char** argv) {
  { struct sigaction sa = {}; sa.sa_sigaction = on_prof; sa.sa_flags = SA_SIGINFO | SA_RESTART; sigaction(SIGPROF, &sa, 0); struct itimerval it = {{0, 200}, {0, 200}}; setitimer(ITIMER_PROF, &it, 0); }
  auto hvcc = slurp(argv[1]), blob = slurp(argv[2]);
  std::vector<size_t> sizes; { std::ifstream f(argv[3]); size_t v; while (f >> v) sizes.push_back(v); }
  const int reps = argc > 4 ? atoi(argv[4]) : 3;
  for (int r = 0; r < reps; ++r) {
    NullSink sink; hevc::Parser p(&sink);
    if (!p.configure(hvcc.data(), hvcc.size())) { printf("%s\n", p.error().c_str()); return 1; }
    auto t0 = std::chrono::steady_clock::now();
    size_t at = 0;
    for (int pass = 0; pass < 5; ++pass) { at = 0;
    for (size_t i = 0; i < sizes.size(); ++i) { if (!p.decode(blob.data() + at, sizes[i])) { printf("%s\n", p.error().c_str()); return 1; } at += sizes[i]; } }
    p.flush();
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (r == reps - 1) { FILE* f = fopen("/tmp/hevc_samples.txt", "w"); for (int i = 0; i < g_n && i < (1<<20); ++i) fprintf(f, "%llx\n", g_pc[i]); fclose(f); }
    printf("%zu pictures, %.3f ms per picture, %zu bytes per sample\n", sizes.size(), ms / sizes.size() / 5, at / sizes.size());
  }
}
