// MI355X backend behind VideoDecoderInterface (the reference's analogue is
// impls/nvidia/nvidia_video_decoder.h:38-106; nothing is shared with it).
#pragma once

#include "../video_decoder_interface.h"

#include <cstdint>

namespace hwang {

// Returns nullptr when the device cannot be opened.
VideoDecoderInterface* make_amd_video_decoder(int32_t device_id,
                                              uint32_t num_devices);

// NUMA node the GPU is attached to (from its PCI address), -1 if unknown.
int amd_device_numa_node(int32_t device_id);

}  // namespace hwang
