// What a container's sync-sample table does not say: whether a sync sample is an IDR picture.  `stss` marks random
// access points; x264's open-GOP mode (and others) list recovery-point I pictures there, whose leading B pictures
// still refer to the GOP before.  Code that treats the stretch between two sync samples as self-contained - the
// per-GOP cut behind the last wanted frame (SparseDecodeHints::set_interval_keyframes), the retriever's per-GOP work
// items - asks here first.  Host only; AVCC framing (ISO/IEC 14496-15: length-prefixed NAL units, the length's size
// from the avcC box, which is what MP4IndexCreator hands out as metadata_bytes: mp4_index_creator.cpp:446-453).
#pragma once

#include <cstddef>
#include <cstdint>
#include <vector>

namespace hwang {
namespace avc {

// Bytes per NAL length field according to an avcC payload; 0 if `extradata` is not one.
inline int nal_length_size(const std::vector<uint8_t>& extradata) {
  if (extradata.size() < 7 || extradata[0] != 1) return 0;
  return (extradata[4] & 3) + 1;
}

// Does the sample hold a coded slice of an IDR picture (nal_unit_type 5)?  Malformed framing: false.
inline bool sample_is_idr(const uint8_t* data, size_t size, int length_size) {
  if (!data || length_size < 1 || length_size > 4) return false;
  size_t pos = 0;
  while (pos + length_size < size) {
    size_t len = 0;
    for (int i = 0; i < length_size; ++i) len = (len << 8) | data[pos + i];
    pos += length_size;
    if (len == 0 || len > size - pos) return false;
    const int type = data[pos] & 0x1f;
    if (type == 5) return true;
    if (type == 1) return false;  // a non-IDR slice: the picture is not an IDR picture
    pos += len;
  }
  return false;
}

}  // namespace avc

// The same question for either codec.  HEVC (ISO/IEC 14496-15 again: length-prefixed NAL units, the length's size in
// byte 21 of the hvcC box, nal_unit_type in bits 1 .. 6 of the first header byte): IDR_W_RADL / IDR_N_LP (19, 20) and
// the BLA types (16 .. 18: whatever leads them is not decodable by definition) start a closed GOP; a CRA picture (21)
// does not - its RASL pictures refer to the GOP before.  An avcC payload read as the answer for an hvcC one says "one
// length byte" and then "not an IDR picture" for every sample: correct but useless - no GOP of an HEVC clip was ever a
// work item of its own.
struct SampleFraming {
  int codec = 0;        // 0: unknown (trust the container's sync table), 1: AVC, 2: HEVC
  int length_size = 0;
};
inline SampleFraming sample_framing(bool hevc, const std::vector<uint8_t>& extradata) {
  SampleFraming f;
  if (hevc) {
    if (extradata.size() >= 23 && extradata[0] == 1) {
      f.codec = 2;
      f.length_size = (extradata[21] & 3) + 1;
    }
    return f;
  }
  f.length_size = avc::nal_length_size(extradata);
  f.codec = f.length_size ? 1 : 0;
  return f;
}
inline bool sample_starts_closed_gop(const uint8_t* data, size_t size, const SampleFraming& f) {
  if (f.codec == 0) return true;
  if (f.codec == 1) return avc::sample_is_idr(data, size, f.length_size);
  if (!data || f.length_size < 1 || f.length_size > 4) return false;
  size_t pos = 0;
  while (pos + f.length_size < size) {
    size_t len = 0;
    for (int i = 0; i < f.length_size; ++i) len = (len << 8) | data[pos + i];
    pos += f.length_size;
    if (len < 2 || len > size - pos) return false;
    const int type = (data[pos] >> 1) & 0x3f;
    if (type >= 16 && type <= 20) return true;   // BLA_W_LP .. IDR_N_LP
    if (type < 32) return false;                 // another coded slice segment first: CRA, or not a random access point at all
    pos += len;
  }
  return false;
}

// HEVC only: what the sample's first coded slice segment is - 1: an IDR or BLA picture (a closed GOP starts here), 2: a CRA
// picture, 3: a leading picture (RADL / RASL: nal_unit_type 6 .. 9), 0: anything else or malformed framing.
inline int hevc_sample_kind(const uint8_t* data, size_t size, const SampleFraming& f) {
  if (f.codec != 2 || !data || f.length_size < 1 || f.length_size > 4) return 0;
  size_t pos = 0;
  while (pos + f.length_size < size) {
    size_t len = 0;
    for (int i = 0; i < f.length_size; ++i) len = (len << 8) | data[pos + i];
    pos += f.length_size;
    if (len < 2 || len > size - pos) return 0;
    const int type = (data[pos] >> 1) & 0x3f;
    if (type >= 16 && type <= 20) return 1;
    if (type == 21) return 2;
    if (type >= 6 && type <= 9) return 3;
    if (type < 32) return 0;
    pos += len;
  }
  return 0;
}

}  // namespace hwang
