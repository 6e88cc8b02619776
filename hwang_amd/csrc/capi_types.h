// Opaque handle layouts shared by the capi_*.cpp translation units.
#pragma once

#include "video_index.h"

struct hwang_video_index {
  hwang::VideoIndex v;
};
