// What the container says about an HEVC track: the HEVCDecoderConfigurationRecord of ISO/IEC 14496-15 8.3.3.1 (the hvcC
// box payload the reference's indexer returns as the track's metadata, mp4_index_creator.cpp:454-469) and the first
// sequence parameter set in it (ITU-T H.265 7.3.2.2, as far as the picture size and the coding block sizes), as text
// for messages and for `hwang_hevc_describe_config`; and the format names that select the HEVC half of the backend
// (the reference's software backend: impls/software/software_video_decoder.cpp:128-132).  Decoding itself is
// hevc_parser.cpp + ../amd/k_hevc.hip.  Host only.
#pragma once

#include <cstddef>
#include <cstdint>
#include <string>

namespace hwang {
namespace hevc {

struct Config {
  int configuration_version = 0;
  int profile_space = 0, tier = 0, profile_idc = 0, level_idc = 0;
  uint32_t profile_compatibility = 0;
  int chroma_format = 0, bit_depth_luma = 8, bit_depth_chroma = 8;
  int num_temporal_layers = 0, nal_length_size = 0;
  int num_vps = 0, num_sps = 0, num_pps = 0, num_other = 0;
  // from the first SPS, if one is there and parses
  bool have_sps = false;
  int sps_chroma_format = 0, width = 0, height = 0;        // pic_width / height_in_luma_samples
  int crop_left = 0, crop_right = 0, crop_top = 0, crop_bottom = 0;  // conformance window, luma samples
  int sps_bit_depth_luma = 8, sps_bit_depth_chroma = 8;
  int log2_min_cb = 0, log2_ctb = 0, log2_min_tb = 0, log2_max_tb = 0;
};

// true: `out` filled; false: `error` says what is wrong with the record.
bool parse_hvcc(const uint8_t* data, size_t size, Config* out, std::string* error);
// "HEVC Main profile, Main tier, level 4.0; 1920x1080 (coded 1920x1088), 4:2:0 8-bit, CTB 64; 1 VPS / 1 SPS / 1 PPS, 4-byte NAL lengths"
std::string describe(const Config& c);
// Is `format` one of the names the reference's backends read as HEVC (software_video_decoder.cpp:128-132), or a sample
// entry type that carries it?
bool is_hevc_format(const std::string& format);

}  // namespace hevc
}  // namespace hwang
