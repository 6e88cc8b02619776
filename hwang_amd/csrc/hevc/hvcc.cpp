#include "hvcc.h"

#include <algorithm>
#include <vector>

namespace hwang {
namespace hevc {

namespace {

struct BitReader {
  const uint8_t* p;
  size_t n, pos = 0;  // pos in bits
  bool overrun = false;
  BitReader(const uint8_t* d, size_t bytes) : p(d), n(bytes) {}
  uint32_t u(int bits) {
    uint32_t v = 0;
    for (int i = 0; i < bits; ++i) {
      if (pos >= n * 8) {
        overrun = true;
        return 0;
      }
      v = (v << 1) | ((p[pos >> 3] >> (7 - (pos & 7))) & 1);
      ++pos;
    }
    return v;
  }
  void skip(int bits) {
    for (; bits > 0 && !overrun; bits -= 16) (void)u(bits > 16 ? 16 : bits);
  }
  uint32_t ue() {  // 9.2: Exp-Golomb
    int zeros = 0;
    while (!overrun && u(1) == 0)
      if (++zeros > 31) {
        overrun = true;
        return 0;
      }
    return zeros ? ((1u << zeros) - 1 + u(zeros)) : 0;
  }
};

// NAL payload without emulation prevention bytes (7.4.2: 00 00 03 -> 00 00)
std::vector<uint8_t> unescape(const uint8_t* d, size_t n) {
  std::vector<uint8_t> out;
  out.reserve(n);
  int zeros = 0;
  for (size_t i = 0; i < n; ++i) {
    if (zeros >= 2 && d[i] == 3) {
      zeros = 0;
      continue;
    }
    zeros = d[i] == 0 ? zeros + 1 : 0;
    out.push_back(d[i]);
  }
  return out;
}

// 7.3.3: profile_tier_level(1, max_sub_layers_minus1) - skipped, the record's header says the same
void skip_profile_tier_level(BitReader& b, int max_sub_layers_minus1) {
  b.skip(2 + 1 + 5 + 32 + 4 + 43 + 1 + 8);  // general_*: 96 bits
  bool profile_present[8] = {false}, level_present[8] = {false};
  for (int i = 0; i < max_sub_layers_minus1; ++i) {
    profile_present[i] = b.u(1);
    level_present[i] = b.u(1);
  }
  if (max_sub_layers_minus1 > 0)
    for (int i = max_sub_layers_minus1; i < 8; ++i) b.skip(2);
  for (int i = 0; i < max_sub_layers_minus1; ++i) {
    if (profile_present[i]) b.skip(88);
    if (level_present[i]) b.skip(8);
  }
}

// 7.3.2.2.1 as far as the coding block sizes
bool parse_sps(const uint8_t* nal, size_t size, Config* c) {
  if (size < 3) return false;
  const std::vector<uint8_t> rbsp = unescape(nal + 2, size - 2);  // two-byte NAL unit header (7.3.1.2)
  BitReader b(rbsp.data(), rbsp.size());
  b.skip(4);  // sps_video_parameter_set_id
  const int max_sub_layers_minus1 = static_cast<int>(b.u(3));
  b.skip(1);  // sps_temporal_id_nesting_flag
  skip_profile_tier_level(b, max_sub_layers_minus1);
  (void)b.ue();  // sps_seq_parameter_set_id
  c->sps_chroma_format = static_cast<int>(b.ue());
  if (c->sps_chroma_format == 3) b.skip(1);  // separate_colour_plane_flag
  c->width = static_cast<int>(b.ue());
  c->height = static_cast<int>(b.ue());
  if (b.u(1)) {  // conformance_window_flag; offsets in chroma units
    const int sub_w = (c->sps_chroma_format == 1 || c->sps_chroma_format == 2) ? 2 : 1, sub_h = c->sps_chroma_format == 1 ? 2 : 1;
    // (a hostile record: the offsets are kept unsigned until they have been held against the picture size)
    const uint64_t l = static_cast<uint64_t>(b.ue()) * sub_w, r = static_cast<uint64_t>(b.ue()) * sub_w;
    const uint64_t t = static_cast<uint64_t>(b.ue()) * sub_h, bo = static_cast<uint64_t>(b.ue()) * sub_h;
    if (c->width <= 0 || c->height <= 0 || l + r >= static_cast<uint64_t>(c->width) || t + bo >= static_cast<uint64_t>(c->height)) return false;
    c->crop_left = static_cast<int>(l);
    c->crop_right = static_cast<int>(r);
    c->crop_top = static_cast<int>(t);
    c->crop_bottom = static_cast<int>(bo);
  }
  c->sps_bit_depth_luma = 8 + static_cast<int>(b.ue());
  c->sps_bit_depth_chroma = 8 + static_cast<int>(b.ue());
  (void)b.ue();  // log2_max_pic_order_cnt_lsb_minus4
  const bool ordering_info = b.u(1);
  for (int i = ordering_info ? 0 : max_sub_layers_minus1; i <= max_sub_layers_minus1; ++i) {
    (void)b.ue();
    (void)b.ue();
    (void)b.ue();
  }
  c->log2_min_cb = 3 + static_cast<int>(b.ue());
  c->log2_ctb = c->log2_min_cb + static_cast<int>(b.ue());
  c->log2_min_tb = 2 + static_cast<int>(b.ue());
  c->log2_max_tb = c->log2_min_tb + static_cast<int>(b.ue());
  if (b.overrun || c->width <= 0 || c->height <= 0 || c->width > 65536 || c->height > 65536 || c->log2_ctb > 6 || c->log2_ctb < 4 ||
      c->sps_chroma_format > 3 || c->sps_bit_depth_luma > 16 || c->sps_bit_depth_chroma > 16)
    return false;
  return true;
}

}  // namespace

bool parse_hvcc(const uint8_t* d, size_t n, Config* c, std::string* error) {
  *c = Config();
  if (!d || n < 23) {
    *error = "the hvcC payload is shorter than the 23 bytes of its fixed part";
    return false;
  }
  c->configuration_version = d[0];
  if (c->configuration_version != 1) {
    *error = "configurationVersion " + std::to_string(d[0]) + " (1 expected)";
    return false;
  }
  c->profile_space = d[1] >> 6;
  c->tier = (d[1] >> 5) & 1;
  c->profile_idc = d[1] & 31;
  c->profile_compatibility = (uint32_t(d[2]) << 24) | (uint32_t(d[3]) << 16) | (uint32_t(d[4]) << 8) | d[5];
  c->level_idc = d[12];
  c->chroma_format = d[16] & 3;
  c->bit_depth_luma = 8 + (d[17] & 7);
  c->bit_depth_chroma = 8 + (d[18] & 7);
  c->num_temporal_layers = (d[21] >> 3) & 7;
  c->nal_length_size = (d[21] & 3) + 1;
  const int arrays = d[22];
  size_t pos = 23;
  for (int a = 0; a < arrays; ++a) {
    if (pos + 3 > n) {
      *error = "the NAL unit arrays run past the end of the box";
      return false;
    }
    const int type = d[pos] & 0x3f;
    const int count = (d[pos + 1] << 8) | d[pos + 2];
    pos += 3;
    for (int k = 0; k < count; ++k) {
      if (pos + 2 > n) {
        *error = "the NAL unit arrays run past the end of the box";
        return false;
      }
      const size_t len = (size_t(d[pos]) << 8) | d[pos + 1];
      pos += 2;
      if (len > n - pos) {
        *error = "a parameter set runs past the end of the box";
        return false;
      }
      if (type == 32) ++c->num_vps;
      else if (type == 33) {
        if (c->num_sps == 0) c->have_sps = parse_sps(d + pos, len, c);
        ++c->num_sps;
      } else if (type == 34) ++c->num_pps;
      else ++c->num_other;
      pos += len;
    }
  }
  return true;
}

std::string describe(const Config& c) {
  static const char* const profiles[] = {"profile 0", "Main", "Main 10", "Main Still Picture", "Range Extensions", "High Throughput",
                                         "Multiview Main", "Scalable Main", "3D Main", "Screen Content", "Scalable Range Extensions"};
  std::string s = "HEVC ";
  if (c.profile_idc >= 1 && c.profile_idc <= 10) s += std::string(profiles[c.profile_idc]) + " profile";
  else s += "profile " + std::to_string(c.profile_idc);
  s += c.tier ? ", High tier" : ", Main tier";
  s += ", level " + std::to_string(c.level_idc / 30) + "." + std::to_string((c.level_idc % 30) / 3);
  static const char* const chroma[] = {"4:0:0", "4:2:0", "4:2:2", "4:4:4"};
  if (c.have_sps) {
    const int w = c.width - c.crop_left - c.crop_right, h = c.height - c.crop_top - c.crop_bottom;
    s += "; " + std::to_string(w) + "x" + std::to_string(h);
    if (w != c.width || h != c.height) s += " (coded " + std::to_string(c.width) + "x" + std::to_string(c.height) + ")";
    s += std::string(", ") + chroma[c.sps_chroma_format & 3] + " " + std::to_string(c.sps_bit_depth_luma) + "-bit, CTB " +
         std::to_string(1 << c.log2_ctb);
  } else {
    s += std::string("; ") + chroma[c.chroma_format & 3] + " " + std::to_string(c.bit_depth_luma) + "-bit";
    s += c.num_sps ? " (its SPS could not be read)" : " (no SPS in the record)";
  }
  s += "; " + std::to_string(c.num_vps) + " VPS / " + std::to_string(c.num_sps) + " SPS / " + std::to_string(c.num_pps) + " PPS, " +
       std::to_string(c.nal_length_size) + "-byte NAL lengths";
  return s;
}

bool is_hevc_format(const std::string& f) { return f == "h265" || f == "hev1" || f == "hevc" || f == "hvc1"; }

}  // namespace hevc
}  // namespace hwang
