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
}  // namespace hwang
