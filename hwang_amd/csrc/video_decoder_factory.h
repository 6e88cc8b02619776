// Backend selection, same entry points as the reference's
// hwang/video_decoder_factory.h:23-37.  The new enumerator AMD selects the
// MI355X backend; SOFTWARE / NVIDIA / INTEL are kept so enum values stay
// source-compatible but are reported unsupported in this build (the
// reference gates them at compile time the same way,
// video_decoder_factory.cpp:31-53) -- there is deliberately no CPU fallback.
#pragma once

#include "common.h"
#include "video_decoder_interface.h"

#include <vector>

struct hwang_backend_vtable;

namespace hwang {

enum class VideoDecoderType {
  SOFTWARE = 0,
  NVIDIA = 1,
  INTEL = 2,
  AMD = 3,
  // values >= 100 (HWANG_DECODER_FIRST_REGISTERED): backends registered at run
  // time through hwang_register_decoder_backend (include/hwang_amd.h)
};

class VideoDecoderFactory {
 public:
  static std::vector<VideoDecoderType> get_supported_decoder_types();

  static bool has_decoder_type(VideoDecoderType type);

  static VideoDecoderInterface* make_from_config(DeviceHandle device_handle,
                                                 uint32_t num_devices,
                                                 VideoDecoderType type);

  // NUMA node a device's host-side work (lane threads, pinned staging) should run on; -1: no preference.
  static int device_numa_node(DeviceHandle device_handle);

  // Run-time registry behind the C plug-in boundary (registered_backend.cpp).
  static bool register_backend(int decoder_type, const struct hwang_backend_vtable* vtable);
  static std::vector<VideoDecoderType> registered_types();

 private:
  static VideoDecoderInterface* make_registered(DeviceHandle device_handle, uint32_t num_devices, VideoDecoderType type);
};

}  // namespace hwang
