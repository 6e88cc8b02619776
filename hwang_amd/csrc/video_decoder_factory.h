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

namespace hwang {

enum class VideoDecoderType {
  SOFTWARE = 0,
  NVIDIA = 1,
  INTEL = 2,
  AMD = 3,
  // Test fixture only: decodes nothing, emits one frame per fed sample whose
  // first four bytes are the sample's first four payload bytes.  Used to pin
  // the DecoderAutomata protocol on machines without a GPU.
  STAMP = 100,
};

class VideoDecoderFactory {
 public:
  static std::vector<VideoDecoderType> get_supported_decoder_types();

  static bool has_decoder_type(VideoDecoderType type);

  static VideoDecoderInterface* make_from_config(DeviceHandle device_handle,
                                                 uint32_t num_devices,
                                                 VideoDecoderType type);
};

}  // namespace hwang
