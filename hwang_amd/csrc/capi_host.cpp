// C ABI over the host-side components (index, indexer, planner, automata).
// See include/hwang_amd.h for the contract.
#include "../../include/hwang_amd.h"

#include "decoder_automata.h"
#include "mp4_index_creator.h"
#include "video_index.h"

#include <cstring>
#include <exception>
#include <new>
#include <string>

using namespace hwang;

#include "capi_types.h"
struct hwang_mp4_indexer {
  explicit hwang_mp4_indexer(uint64_t n) : v(n) {}
  MP4IndexCreator v;
};
struct hwang_video_intervals {
  VideoIntervals v;
};
struct hwang_automata {
  DecoderAutomata* a = nullptr;
  VideoDecoderInterface* backend = nullptr;  // borrowed (owned by `a`)
  std::string error;
};

extern "C" {

const char* hwang_amd_version(void) { return "hwang_amd 0.1 (gfx950)"; }

// ---- VideoIndex -----------------------------------------------------------
hwang_video_index* hwang_video_index_create(
    uint32_t timescale, uint64_t duration, uint32_t width, uint32_t height,
    const char* format, const uint64_t* sample_offsets,
    const uint64_t* sample_sizes, size_t num_samples,
    const uint64_t* keyframe_indices, size_t num_keyframes,
    const uint8_t* metadata, size_t metadata_size) {
  if ((num_samples && (!sample_offsets || !sample_sizes)) || (num_keyframes && !keyframe_indices) ||
      (metadata_size && !metadata))
    return nullptr;
  auto* h = new (std::nothrow) hwang_video_index;
  if (!h) return nullptr;
  try {  // nothing may unwind through the C ABI
    h->v = VideoIndex(
        timescale, duration, width, height, format ? format : "",
        std::vector<uint64_t>(sample_offsets, sample_offsets + num_samples),
        std::vector<uint64_t>(sample_sizes, sample_sizes + num_samples),
        std::vector<uint64_t>(keyframe_indices, keyframe_indices + num_keyframes),
        std::vector<uint8_t>(metadata, metadata + metadata_size));
  } catch (...) {
    delete h;
    return nullptr;
  }
  return h;
}
void hwang_video_index_destroy(hwang_video_index* index) { delete index; }

hwang_video_index* hwang_video_index_deserialize(const uint8_t* data,
                                                 size_t size) {
  auto* h = new (std::nothrow) hwang_video_index;
  if (!h) return nullptr;
  bool ok = false;
  try {
    ok = VideoIndex::deserialize(data, size, &h->v);
  } catch (...) {
  }
  if (!ok) {
    delete h;
    return nullptr;
  }
  return h;
}
size_t hwang_video_index_serialize(const hwang_video_index* index, uint8_t* out,
                                   size_t out_capacity) {
  std::vector<uint8_t> bytes = index->v.serialize();
  if (out && out_capacity >= bytes.size())
    std::memcpy(out, bytes.data(), bytes.size());
  return bytes.size();
}
uint32_t hwang_video_index_timescale(const hwang_video_index* i) { return i->v.timescale(); }
uint64_t hwang_video_index_duration(const hwang_video_index* i) { return i->v.duration(); }
double hwang_video_index_fps(const hwang_video_index* i) { return i->v.fps(); }
uint32_t hwang_video_index_frame_width(const hwang_video_index* i) { return i->v.frame_width(); }
uint32_t hwang_video_index_frame_height(const hwang_video_index* i) { return i->v.frame_height(); }
const char* hwang_video_index_format(const hwang_video_index* i) { return i->v.format().c_str(); }
uint64_t hwang_video_index_frames(const hwang_video_index* i) { return i->v.frames(); }
const char* hwang_video_index_problem(const hwang_video_index* i) { return i->v.problem().c_str(); }
const uint64_t* hwang_video_index_sample_offsets(const hwang_video_index* i) { return i->v.sample_offsets().data(); }
const uint64_t* hwang_video_index_sample_sizes(const hwang_video_index* i) { return i->v.sample_sizes().data(); }
size_t hwang_video_index_num_keyframes(const hwang_video_index* i) { return i->v.keyframe_indices().size(); }
const uint64_t* hwang_video_index_keyframe_indices(const hwang_video_index* i) { return i->v.keyframe_indices().data(); }
size_t hwang_video_index_metadata_size(const hwang_video_index* i) { return i->v.metadata_bytes().size(); }
const uint8_t* hwang_video_index_metadata_bytes(const hwang_video_index* i) { return i->v.metadata_bytes().data(); }

// ---- MP4IndexCreator ------------------------------------------------------
hwang_mp4_indexer* hwang_mp4_indexer_create(uint64_t file_size) {
  return new (std::nothrow) hwang_mp4_indexer(file_size);
}
void hwang_mp4_indexer_destroy(hwang_mp4_indexer* indexer) { delete indexer; }
int hwang_mp4_indexer_feed(hwang_mp4_indexer* indexer, const uint8_t* data,
                           size_t size, uint64_t* next_offset,
                           uint64_t* next_size) {
  uint64_t off = 0, sz = 0;
  bool more = false;
  try {
    more = indexer->v.feed(data, size, off, sz);
  } catch (const std::exception& e) {  // e.g. bad_alloc on a table the bounds checks let through
    indexer->v.abort(std::string("MP4IndexCreator: ") + e.what());
  } catch (...) {
    indexer->v.abort("MP4IndexCreator: unexpected exception");
  }
  if (next_offset) *next_offset = off;
  if (next_size) *next_size = sz;
  return more ? 1 : 0;
}
int hwang_mp4_indexer_is_done(hwang_mp4_indexer* i) { return i->v.is_done(); }
int hwang_mp4_indexer_is_error(hwang_mp4_indexer* i) { return i->v.is_error(); }
const char* hwang_mp4_indexer_error_message(hwang_mp4_indexer* i) { return i->v.error_message().c_str(); }
hwang_video_index* hwang_mp4_indexer_get_video_index(hwang_mp4_indexer* i) {
  auto* h = new (std::nothrow) hwang_video_index;
  try {
    if (h) h->v = i->v.get_video_index();
  } catch (...) {
    delete h;
    return nullptr;
  }
  return h;
}

// ---- planner --------------------------------------------------------------
hwang_video_intervals* hwang_slice_into_video_intervals(
    const hwang_video_index* index, const uint64_t* rows, size_t num_rows) {
  try {
    bool ok = true;
    VideoIntervals plan = slice_into_video_intervals(
        index->v, std::vector<uint64_t>(rows, rows + num_rows), &ok);
    if (!ok) return nullptr;
    auto* h = new (std::nothrow) hwang_video_intervals;
    if (h) h->v = std::move(plan);
    return h;
  } catch (...) {
    return nullptr;
  }
}
void hwang_video_intervals_destroy(hwang_video_intervals* p) { delete p; }
size_t hwang_video_intervals_count(const hwang_video_intervals* p) { return p->v.sample_index_intervals.size(); }
void hwang_video_intervals_range(const hwang_video_intervals* p, size_t i,
                                 uint64_t* start_sample, uint64_t* end_sample) {
  *start_sample = std::get<0>(p->v.sample_index_intervals[i]);
  *end_sample = std::get<1>(p->v.sample_index_intervals[i]);
}
size_t hwang_video_intervals_num_valid(const hwang_video_intervals* p, size_t i) { return p->v.valid_frames[i].size(); }
const uint64_t* hwang_video_intervals_valid(const hwang_video_intervals* p, size_t i) { return p->v.valid_frames[i].data(); }

// ---- DecoderAutomata ------------------------------------------------------
hwang_automata* hwang_automata_create(int device_type, int32_t device_id,
                                      int32_t num_devices, int decoder_type) {
  DeviceHandle handle{static_cast<DeviceType>(device_type), device_id};
  VideoDecoderInterface* backend = VideoDecoderFactory::make_from_config(
      handle, static_cast<uint32_t>(num_devices),
      static_cast<VideoDecoderType>(decoder_type));
  if (!backend) return nullptr;
  auto* h = new (std::nothrow) hwang_automata;
  if (!h) {
    delete backend;
    return nullptr;
  }
  h->backend = backend;
  h->a = DecoderAutomata::make_with_decoder(backend);
  return h;
}
void hwang_automata_destroy(hwang_automata* a) {
  if (!a) return;
  delete a->a;
  delete a;
}
int hwang_automata_initialize(hwang_automata* a, const hwang_encoded_data* data,
                              size_t num_data, const uint8_t* extradata,
                              size_t extradata_size) {
  std::vector<DecoderAutomata::EncodedData> segs(num_data);
  for (size_t i = 0; i < num_data; ++i) {
    const hwang_encoded_data& s = data[i];
    DecoderAutomata::EncodedData& d = segs[i];
    d.encoded_video.assign(s.encoded_video, s.encoded_video + s.encoded_video_size);
    d.width = s.width;
    d.height = s.height;
    d.start_keyframe = s.start_keyframe;
    d.end_keyframe = s.end_keyframe;
    d.format = s.format ? s.format : "";
    d.sample_offsets.assign(s.sample_offsets, s.sample_offsets + s.num_samples);
    d.sample_sizes.assign(s.sample_sizes, s.sample_sizes + s.num_samples);
    d.keyframes.assign(s.keyframes, s.keyframes + s.num_keyframes);
    d.valid_frames.assign(s.valid_frames, s.valid_frames + s.num_valid_frames);
  }
  Result r = a->a->initialize(
      segs, std::vector<uint8_t>(extradata, extradata + extradata_size));
  if (!r.ok) a->error = r.message;
  return r.ok ? 0 : 1;
}
int hwang_automata_get_frames(hwang_automata* a, uint8_t* buffer,
                              int32_t num_frames) {
  Result r = a->a->get_frames(buffer, num_frames);
  if (!r.ok) a->error = r.message;
  return r.ok ? 0 : 1;
}
const char* hwang_automata_last_error(hwang_automata* a) { return a->error.c_str(); }
void hwang_automata_stats(hwang_automata* a, int64_t stats[4]) {
  DecoderAutomata::Stats s = a->a->stats();
  stats[0] = s.samples_fed;
  stats[1] = s.frames_discarded;
  stats[2] = s.frames_returned;
  stats[3] = s.flushes;
}

}  // extern "C"
