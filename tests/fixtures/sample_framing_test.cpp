// util/avc_sample.h: which sync samples start a closed GOP, for AVC and HEVC framing (tests/test_host_sample_framing.py).
#include "util/avc_sample.h"

#include <cstdio>
#include <vector>

using namespace hwang;

static std::vector<uint8_t> nal(int length_size, std::vector<uint8_t> payload) {
  std::vector<uint8_t> out;
  for (int i = length_size - 1; i >= 0; --i) out.push_back(static_cast<uint8_t>(payload.size() >> (8 * i)));
  out.insert(out.end(), payload.begin(), payload.end());
  return out;
}
static std::vector<uint8_t> cat(std::vector<std::vector<uint8_t>> parts) {
  std::vector<uint8_t> out;
  for (auto& p : parts) out.insert(out.end(), p.begin(), p.end());
  return out;
}
static std::vector<uint8_t> hevc_nal(int length_size, int type) { return nal(length_size, {static_cast<uint8_t>(type << 1), 1, 0x80, 0x00}); }

#define CHECK(c) do { if (!(c)) { std::printf("failed: %s (line %d)\n", #c, __LINE__); return 1; } } while (0)

int main() {
  // an hvcC payload: version 1, Main (compatibility flags 0x60000000 - read as an avcC it says "1-byte lengths"), byte 21 = 0xfc | (length_size - 1)
  for (int ls : {1, 2, 4}) {
    std::vector<uint8_t> hvcc(23, 0);
    hvcc[0] = 1; hvcc[1] = 1; hvcc[2] = 0x60; hvcc[21] = static_cast<uint8_t>(0xfc | (ls - 1));
    const SampleFraming f = sample_framing(true, hvcc);
    CHECK(f.codec == 2 && f.length_size == ls);
    const auto vps = hevc_nal(ls, 32), sps = hevc_nal(ls, 33), pps = hevc_nal(ls, 34), aud = hevc_nal(ls, 35), sei = hevc_nal(ls, 39);
    for (int type : {16, 17, 18, 19, 20}) {  // BLA_W_LP, BLA_W_RADL, BLA_N_LP, IDR_W_RADL, IDR_N_LP
      auto s = cat({aud, vps, sps, pps, sei, hevc_nal(ls, type), hevc_nal(ls, type)});
      CHECK(sample_starts_closed_gop(s.data(), s.size(), f));
    }
    for (int type : {21, 0, 1, 8, 9}) {      // CRA, TRAIL_N, TRAIL_R, RASL_N, RASL_R
      auto s = cat({aud, sei, hevc_nal(ls, type)});
      CHECK(!sample_starts_closed_gop(s.data(), s.size(), f));
    }
    auto only_sets = cat({vps, sps, pps});
    CHECK(!sample_starts_closed_gop(only_sets.data(), only_sets.size(), f));
    auto cut = cat({aud, hevc_nal(ls, 19)});
    cut.resize(cut.size() - 2);              // a length that runs past the sample
    CHECK(!sample_starts_closed_gop(cut.data(), cut.size(), f));
    CHECK(!sample_starts_closed_gop(nullptr, 0, f));
    // what kind of sample: 1 closed GOP, 2 CRA, 3 leading, 0 other
    for (int type : {16, 19, 20}) { auto s = cat({aud, sei, hevc_nal(ls, type)}); CHECK(hevc_sample_kind(s.data(), s.size(), f) == 1); }
    { auto s = cat({aud, vps, sps, pps, hevc_nal(ls, 21)}); CHECK(hevc_sample_kind(s.data(), s.size(), f) == 2); }
    for (int type : {6, 7, 8, 9}) { auto s = cat({aud, hevc_nal(ls, type)}); CHECK(hevc_sample_kind(s.data(), s.size(), f) == 3); }
    for (int type : {0, 1, 2, 5}) { auto s = cat({hevc_nal(ls, type)}); CHECK(hevc_sample_kind(s.data(), s.size(), f) == 0); }
    CHECK(hevc_sample_kind(only_sets.data(), only_sets.size(), f) == 0 && hevc_sample_kind(cut.data(), cut.size(), f) == 0);
  }
  {  // an hvcC too short to say: the container's sync table is all there is
    const SampleFraming f = sample_framing(true, std::vector<uint8_t>{1, 1, 0x60});
    CHECK(f.codec == 0);
    CHECK(sample_starts_closed_gop(nullptr, 0, f));
  }
  {  // AVC as before
    std::vector<uint8_t> avcc = {1, 100, 0, 40, 0xff, 0xe1, 0};
    const SampleFraming f = sample_framing(false, avcc);
    CHECK(f.codec == 1 && f.length_size == 4);
    auto idr = cat({nal(4, {0x09, 0xf0}), nal(4, {0x67, 1}), nal(4, {0x68, 1}), nal(4, {0x65, 0x88})});
    auto rec = cat({nal(4, {0x09, 0xf0}), nal(4, {0x06, 5}), nal(4, {0x41, 0x9a})});
    CHECK(sample_starts_closed_gop(idr.data(), idr.size(), f));
    CHECK(!sample_starts_closed_gop(rec.data(), rec.size(), f));
    CHECK(sample_framing(false, std::vector<uint8_t>()).codec == 0);
  }
  std::printf("ok\n");
  return 0;
}
