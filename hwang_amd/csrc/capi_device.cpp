// C ABI over the backend boundary (VideoDecoderInterface + factory) and the
// parallel retriever.  See include/hwang_amd.h.
#include "../../include/hwang_amd.h"

#include "retriever.h"
#include "sparse_hints.h"
#include "video_decoder_factory.h"
#include "video_index.h"

#include <exception>
#include <memory>
#include <mutex>
#include <new>
#include <string>

using namespace hwang;

#include "capi_types.h"

struct hwang_decoder {
  std::unique_ptr<VideoDecoderInterface> d;
  // feed-side and get-side threads may both fail; keep their texts apart.
  std::mutex m;
  std::string error;
  int set(const Result& r) {
    if (r.ok) return 0;
    std::lock_guard<std::mutex> lk(m);
    error = r.message;
    return 1;
  }
};

struct hwang_retriever {
  std::unique_ptr<ParallelRetriever> r;
  std::string error;
};

extern "C" {

int hwang_decoder_type_supported(int decoder_type) {
  return VideoDecoderFactory::has_decoder_type(
             static_cast<VideoDecoderType>(decoder_type))
             ? 1
             : 0;
}

hwang_decoder* hwang_decoder_create(int device_type, int32_t device_id,
                                    uint32_t num_devices, int decoder_type) {
  DeviceHandle handle{static_cast<DeviceType>(device_type), device_id};
  VideoDecoderInterface* d = VideoDecoderFactory::make_from_config(
      handle, num_devices, static_cast<VideoDecoderType>(decoder_type));
  if (!d) return nullptr;
  auto* h = new (std::nothrow) hwang_decoder;
  if (!h) {
    delete d;
    return nullptr;
  }
  h->d.reset(d);
  return h;
}
void hwang_decoder_destroy(hwang_decoder* dec) { delete dec; }

int hwang_decoder_configure(hwang_decoder* dec, uint32_t width, uint32_t height,
                            const char* format, const uint8_t* extradata,
                            size_t extradata_size) {
  VideoDecoderInterface::FrameInfo info;
  info.width = width;
  info.height = height;
  info.format = format ? format : "";
  return dec->set(dec->d->configure(
      info, std::vector<uint8_t>(extradata, extradata + extradata_size)));
}
int hwang_decoder_feed(hwang_decoder* dec, const uint8_t* encoded,
                       size_t encoded_size, int keyframe) {
  return dec->set(dec->d->feed(encoded, encoded_size, keyframe != 0));
}
int hwang_decoder_flush(hwang_decoder* dec) { return dec->set(dec->d->flush()); }
int hwang_decoder_discard_frame(hwang_decoder* dec) {
  return dec->set(dec->d->discard_frame());
}
int hwang_decoder_get_frame(hwang_decoder* dec, uint8_t* decoded,
                            size_t decoded_size) {
  return dec->set(dec->d->get_frame(decoded, decoded_size));
}
int hwang_decoder_decoded_frames_buffered(hwang_decoder* dec) {
  return dec->d->decoded_frames_buffered();
}
int hwang_decoder_wait_until_frames_copied(hwang_decoder* dec) {
  return dec->set(dec->d->wait_until_frames_copied());
}
const char* hwang_decoder_last_error(hwang_decoder* dec) {
  std::lock_guard<std::mutex> lk(dec->m);
  return dec->error.c_str();
}
int hwang_decoder_set_wanted(hwang_decoder* dec, const uint64_t* ordinals,
                             size_t n) {
  auto* hints = dynamic_cast<SparseDecodeHints*>(dec->d.get());
  if (!hints) return dec->set(Result(false, "backend takes no sparse hints"));
  hints->set_wanted_frames(std::vector<uint64_t>(ordinals, ordinals + n));
  return 0;
}
int hwang_decoder_set_interval_keyframes(hwang_decoder* dec, const uint64_t* ordinals, size_t n) {
  auto* hints = dynamic_cast<SparseDecodeHints*>(dec->d.get());
  if (!hints) return dec->set(Result(false, "backend takes no sparse hints"));
  hints->set_interval_keyframes(std::vector<uint64_t>(ordinals, ordinals + n));
  return 0;
}
int hwang_decoder_get_frame_yuv(hwang_decoder* dec, uint8_t* decoded,
                                size_t decoded_size) {
  auto* planar = dynamic_cast<PlanarOutput*>(dec->d.get());
  if (!planar) return dec->set(Result(false, "backend has no planar output"));
  return dec->set(planar->get_frame_yuv(decoded, decoded_size));
}

// ---- parallel retriever ---------------------------------------------------
hwang_retriever* hwang_retriever_create(int device_type, int32_t device_id,
                                        int decoder_type, int32_t num_lanes) {
  DeviceHandle handle{static_cast<DeviceType>(device_type), device_id};
  ParallelRetriever* r = ParallelRetriever::make(
      handle, static_cast<VideoDecoderType>(decoder_type), num_lanes);
  if (!r) return nullptr;
  auto* h = new (std::nothrow) hwang_retriever;
  if (!h) {
    delete r;
    return nullptr;
  }
  h->r.reset(r);
  return h;
}
hwang_retriever* hwang_retriever_create_multi(int device_type, const int32_t* device_ids,
                                              size_t num_devices, int decoder_type,
                                              int32_t num_lanes) {
  if (!device_ids || !num_devices) return nullptr;
  std::vector<DeviceHandle> devices;
  for (size_t i = 0; i < num_devices; ++i)
    devices.push_back(DeviceHandle{static_cast<DeviceType>(device_type), device_ids[i]});
  ParallelRetriever* r = nullptr;
  try {
    r = ParallelRetriever::make(devices, static_cast<VideoDecoderType>(decoder_type), num_lanes);
  } catch (...) {
  }
  if (!r) return nullptr;
  auto* h = new (std::nothrow) hwang_retriever;
  if (!h) {
    delete r;
    return nullptr;
  }
  h->r.reset(r);
  return h;
}
int hwang_retriever_set_shard(hwang_retriever* r, int32_t rank, int32_t world) {
  Result res = r->r->set_shard(rank, world);
  if (!res.ok) r->error = res.message;
  return res.ok ? 0 : 1;
}
int32_t hwang_retriever_num_lanes(hwang_retriever* r) { return r->r->num_lanes(); }
int hwang_retriever_retrieve_batch(hwang_retriever* r, const hwang_retrieve_request* requests,
                                   size_t num_requests, int out_is_yuv) {
  Result res;
  try {
    std::vector<ParallelRetriever::Request> reqs(num_requests);
    for (size_t i = 0; i < num_requests; ++i) {
      if (!requests[i].index) {
        r->error = "retrieve_batch: request without an index";
        return 1;
      }
      reqs[i].index = &requests[i].index->v;
      reqs[i].file = requests[i].file;
      reqs[i].file_size = requests[i].file_size;
      reqs[i].rows = requests[i].rows;
      reqs[i].num_rows = requests[i].num_rows;
      reqs[i].out = requests[i].out;
      reqs[i].out_capacity = requests[i].out_capacity;
    }
    res = r->r->retrieve_batch(reqs, out_is_yuv != 0);
  } catch (const std::exception& e) {
    res = Result(false, std::string("retrieve_batch: ") + e.what());
  }
  if (!res.ok) r->error = res.message;
  return res.ok ? 0 : 1;
}
void hwang_retriever_device_stats(hwang_retriever* r, int32_t device_slot, int64_t stats[4]) {
  ParallelRetriever::DeviceStats s = r->r->device_stats(device_slot);
  stats[0] = s.lanes;
  stats[1] = s.items;
  stats[2] = s.frames_returned;
  stats[3] = s.samples_fed;
}
void hwang_retriever_destroy(hwang_retriever* r) { delete r; }
int hwang_retriever_retrieve(hwang_retriever* r, const hwang_video_index* index,
                             const uint8_t* file, size_t file_size,
                             const uint64_t* rows, size_t num_rows,
                             uint8_t* out, size_t out_capacity, int out_is_yuv) {
  Result res;
  try {
    res = r->r->retrieve(index->v, file, file_size, std::vector<uint64_t>(rows, rows + num_rows), out,
                         out_capacity, out_is_yuv != 0);
  } catch (const std::exception& e) {
    res = Result(false, std::string("retrieve: ") + e.what());
  }
  if (!res.ok) r->error = res.message;
  return res.ok ? 0 : 1;
}
const char* hwang_retriever_last_error(hwang_retriever* r) {
  return r->error.c_str();
}
void hwang_retriever_stats(hwang_retriever* r, int64_t stats[10]) {
  ParallelRetriever::Stats s = r->r->stats();
  stats[0] = s.intervals;
  stats[1] = s.samples_fed;
  stats[2] = s.pictures_reconstructed;
  stats[3] = s.pictures_skipped;
  stats[4] = s.frames_returned;
  stats[5] = s.bytes_uploaded;
  stats[6] = s.entropy_ns;
  stats[7] = s.wall_ns;
  stats[8] = s.stall_ns;
  stats[9] = s.lane_threads;
}
void hwang_retriever_host_counters(hwang_retriever* r, int64_t out[4]) {
  ParallelRetriever::Stats s = r->r->stats();
  out[0] = s.lane_instructions;
  out[1] = s.lane_cycles;
  out[2] = s.lane_branch_misses;
  out[3] = s.lane_cpu_ns;
}

}  // extern "C"
