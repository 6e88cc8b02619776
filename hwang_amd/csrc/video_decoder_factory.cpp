#include "video_decoder_factory.h"

#include "amd/amd_video_decoder.h"

namespace hwang {

// Reference: video_decoder_factory.cpp:31-98.  Only the MI355X backend exists
// in this build, plus whatever was registered at run time.
std::vector<VideoDecoderType> VideoDecoderFactory::get_supported_decoder_types() {
  std::vector<VideoDecoderType> types = {VideoDecoderType::AMD};
  for (VideoDecoderType t : registered_types()) types.push_back(t);
  return types;
}

bool VideoDecoderFactory::has_decoder_type(VideoDecoderType type) {
  for (VideoDecoderType t : get_supported_decoder_types())
    if (t == type) return true;
  return false;
}

VideoDecoderInterface* VideoDecoderFactory::make_from_config(
    DeviceHandle device_handle, uint32_t num_devices, VideoDecoderType type) {
  switch (type) {
    case VideoDecoderType::AMD:
      // The reconstruct kernels only exist for the GPU; a CPU handle is a
      // configuration error, not a cue to fall back.
      if (device_handle.type != DeviceType::GPU) return nullptr;
      return make_amd_video_decoder(device_handle.id, num_devices);
    default:
      // (0 is this library's own "the caller is the parallelism" - ParallelRetriever's lanes; the reference's callers
      // always pass >= 1, video_decoder_factory.cpp:90-91, and a registered backend is written against those)
      return make_registered(device_handle, num_devices ? num_devices : 1, type);
  }
}

int VideoDecoderFactory::device_numa_node(DeviceHandle device_handle) {
  return device_handle.type == DeviceType::GPU ? amd_device_numa_node(device_handle.id) : -1;
}

}  // namespace hwang
