// HIP error -> hwang::Result (the reference's util/cuda.h:38-57 aborts via
// LOG(FATAL); a backend must never abort across the plugin boundary).
#pragma once

#include "../common.h"

#include <hip/hip_runtime.h>

#include <string>

#define HWANG_HIP_RETURN(expr__)                                              \
  do {                                                                        \
    hipError_t err__ = (expr__);                                              \
    if (err__ != hipSuccess) {                                                \
      return ::hwang::Result(false, std::string("HIP error: ") +              \
                                        hipGetErrorString(err__) + " at " +   \
                                        __FILE__ + ":" +                      \
                                        std::to_string(__LINE__));            \
    }                                                                         \
  } while (0)
