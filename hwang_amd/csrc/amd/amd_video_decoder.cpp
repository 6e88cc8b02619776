#include "amd_video_decoder.h"

namespace hwang {

VideoDecoderInterface* make_amd_video_decoder(int32_t, uint32_t) {
  return nullptr;  // placeholder until the reconstruct pipeline lands
}

}  // namespace hwang
