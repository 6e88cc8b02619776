#include "video_index.h"

#include <algorithm>

#include <cstring>

namespace hwang {

// ---------------------------------------------------------------------------
// proto3 wire codec for message hwang.proto.VideoIndex
// (hwang/hwang_descriptors.proto:5-15):
//   1 frame_width u32 | 2 frame_height u32 | 3 sample_offsets packed u64 |
//   4 sample_sizes packed u64 | 5 keyframe_indices packed u64 |
//   6 metadata_bytes bytes | 7 timescale u32 | 8 duration u64 | 9 format string
// libprotobuf emits fields in field-number order and omits proto3 defaults
// (zero scalars, empty repeated/bytes/string); serialize() does the same so
// the bytes are identical to the reference's VideoIndex::serialize()
// (video_index.cpp:43-60).
// ---------------------------------------------------------------------------
namespace {

void put_varint(std::vector<uint8_t>& out, uint64_t v) {
  while (v >= 0x80) {
    out.push_back(static_cast<uint8_t>(v) | 0x80);
    v >>= 7;
  }
  out.push_back(static_cast<uint8_t>(v));
}

size_t varint_len(uint64_t v) {
  size_t n = 1;
  while (v >= 0x80) {
    v >>= 7;
    ++n;
  }
  return n;
}

void put_scalar(std::vector<uint8_t>& out, uint32_t field, uint64_t v) {
  if (v == 0) return;
  put_varint(out, (field << 3) | 0);
  put_varint(out, v);
}

void put_packed(std::vector<uint8_t>& out, uint32_t field,
                const std::vector<uint64_t>& vals) {
  if (vals.empty()) return;
  size_t payload = 0;
  for (uint64_t v : vals) payload += varint_len(v);
  put_varint(out, (field << 3) | 2);
  put_varint(out, payload);
  for (uint64_t v : vals) put_varint(out, v);
}

void put_bytes(std::vector<uint8_t>& out, uint32_t field, const uint8_t* p,
               size_t n) {
  if (n == 0) return;
  put_varint(out, (field << 3) | 2);
  put_varint(out, n);
  out.insert(out.end(), p, p + n);
}

struct Reader {
  const uint8_t* p;
  const uint8_t* end;
  bool fail = false;

  uint64_t varint() {
    uint64_t v = 0;
    for (int shift = 0; shift < 64; shift += 7) {
      if (p >= end) {
        fail = true;
        return 0;
      }
      uint8_t b = *p++;
      v |= static_cast<uint64_t>(b & 0x7f) << shift;
      if (!(b & 0x80)) return v;
    }
    fail = true;
    return 0;
  }
};

}  // namespace

std::vector<uint8_t> VideoIndex::serialize() const {
  std::vector<uint8_t> out;
  out.reserve(64 + 10 * 3 * sample_sizes_.size());
  put_scalar(out, 1, frame_width_);
  put_scalar(out, 2, frame_height_);
  put_packed(out, 3, sample_offsets_);
  put_packed(out, 4, sample_sizes_);
  put_packed(out, 5, keyframe_indices_);
  put_bytes(out, 6, metadata_bytes_.data(), metadata_bytes_.size());
  put_scalar(out, 7, timescale_);
  put_scalar(out, 8, duration_);
  put_bytes(out, 9, reinterpret_cast<const uint8_t*>(format_.data()),
            format_.size());
  return out;
}

bool VideoIndex::deserialize(const uint8_t* data, size_t size,
                             VideoIndex* out) {
  Reader r{data, data + size};
  uint32_t timescale = 0, width = 0, height = 0;
  uint64_t duration = 0;
  std::string format;
  std::vector<uint64_t> offs, sizes, kfs;
  std::vector<uint8_t> meta;

  while (r.p < r.end) {
    uint64_t key = r.varint();
    if (r.fail) return false;
    uint32_t field = static_cast<uint32_t>(key >> 3);
    uint32_t wire = static_cast<uint32_t>(key & 7);
    if (wire == 0) {
      uint64_t v = r.varint();
      if (r.fail) return false;
      switch (field) {
        case 1: width = static_cast<uint32_t>(v); break;
        case 2: height = static_cast<uint32_t>(v); break;
        case 7: timescale = static_cast<uint32_t>(v); break;
        case 8: duration = v; break;
        // A non-packed element of a repeated field is legal proto3 input.
        case 3: offs.push_back(v); break;
        case 4: sizes.push_back(v); break;
        case 5: kfs.push_back(v); break;
        default: break;
      }
    } else if (wire == 2) {
      uint64_t n = r.varint();
      if (r.fail || n > static_cast<uint64_t>(r.end - r.p)) return false;
      const uint8_t* body = r.p;
      r.p += n;
      if (field >= 3 && field <= 5) {
        std::vector<uint64_t>& dst = field == 3 ? offs : field == 4 ? sizes : kfs;
        Reader pr{body, body + n};
        while (pr.p < pr.end) {
          uint64_t v = pr.varint();
          if (pr.fail) return false;
          dst.push_back(v);
        }
      } else if (field == 6) {
        meta.assign(body, body + n);
      } else if (field == 9) {
        format.assign(reinterpret_cast<const char*>(body), n);
      }
    } else if (wire == 1) {
      if (r.end - r.p < 8) return false;
      r.p += 8;
    } else if (wire == 5) {
      if (r.end - r.p < 4) return false;
      r.p += 4;
    } else {
      return false;
    }
  }
  if (offs.size() != sizes.size()) return false;
  *out = VideoIndex(timescale, duration, width, height, format, offs, sizes,
                    kfs, meta);
  return true;
}

void VideoIndex::check() {
  problem_.clear();
  if (sample_offsets_.size() != sample_sizes_.size()) {
    problem_ = "sample_offsets and sample_sizes differ in length";
    num_frames_ = std::min(sample_offsets_.size(), sample_sizes_.size());
    return;
  }
  if (num_frames_ && keyframe_indices_.empty()) {
    problem_ = "index has samples but no keyframe";
    return;
  }
  for (size_t i = 0; i < keyframe_indices_.size(); ++i) {
    if (keyframe_indices_[i] >= num_frames_) {
      problem_ = "keyframe index beyond the last sample";
      return;
    }
    if (i && keyframe_indices_[i] <= keyframe_indices_[i - 1]) {
      problem_ = "keyframe indices are not strictly increasing";
      return;
    }
  }
}

VideoIndex VideoIndex::deserialize(const std::vector<uint8_t>& data) {
  VideoIndex idx;
  deserialize(data.data(), data.size(), &idx);
  return idx;
}

// ---------------------------------------------------------------------------
// Seek planner (reference: video_index.cpp:62-109).
//
// Walk the sorted rows with a cursor over GOP boundaries
// (keyframes + [frames]).  A row that falls into the GOP directly after the
// current interval extends it only when the two GOPs are byte-adjacent in the
// file; anything else (a skipped GOP, interleaved audio between them) closes
// the interval and opens a new one at the GOP that holds the row.  Quirks
// kept on purpose because callers can observe them: an empty row list yields
// one empty interval over the first GOP, and adjacency is only ever tested
// for the first boundary crossed.
// ---------------------------------------------------------------------------
VideoIntervals slice_into_video_intervals(const VideoIndex& index,
                                          const std::vector<uint64_t>& rows,
                                          bool* ok) {
  if (ok) *ok = true;
  VideoIntervals plan;
  std::vector<uint64_t> bounds = index.keyframe_indices();
  bounds.push_back(index.frames());
  auto fail = [&]() {
    if (ok) *ok = false;
    return VideoIntervals();
  };
  if (bounds.size() < 2 || !index.consistent()) return fail();

  const auto& offs = index.sample_offsets();
  const auto& sizes = index.sample_sizes();
  size_t first = 0;  // bounds index of the interval's first keyframe
  size_t last = 1;   // bounds index one past the interval's last GOP
  std::vector<uint64_t> wanted;

  for (uint64_t row : rows) {
    if (row >= index.frames() || row < bounds[0]) return fail();
    if (row >= bounds[last]) {
      const uint64_t edge = bounds[last];
      const bool contiguous = offs[edge - 1] + sizes[edge - 1] == offs[edge];
      ++last;
      if (row >= bounds[last] || !contiguous) {
        if (!wanted.empty()) {
          plan.sample_index_intervals.emplace_back(bounds[first],
                                                   bounds[last - 1]);
          plan.valid_frames.push_back(wanted);
          wanted.clear();
        }
        while (row >= bounds[last]) ++last;
        first = last - 1;
      }
    }
    wanted.push_back(row);
  }
  plan.sample_index_intervals.emplace_back(bounds[first], bounds[last]);
  plan.valid_frames.push_back(wanted);
  return plan;
}

}  // namespace hwang
