// HEVC behind the same boundary - as far as this backend goes: it reads what the container says about an HEVC track and
// says no with a name.  The reference's software backend takes "h265" / "hev1" / "hevc" and hands the stream to
// libavcodec's HEVC decoder (impls/software/software_video_decoder.cpp:128-132); its indexer returns the hvcC box
// payload as the track's metadata (mp4_index_creator.cpp:454-469).  Here configure() parses that payload - the
// HEVCDecoderConfigurationRecord of ISO/IEC 14496-15 8.3.3.1 and the first sequence parameter set in it (ITU-T H.265
// 7.3.2.2, as far as the picture size and the coding block sizes) - and refuses the track with a message that says what
// it is.  No HEVC stream exists on the build image or on the GPU boxes to check a decoder against, which is why there
// is none.  Host only.
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
