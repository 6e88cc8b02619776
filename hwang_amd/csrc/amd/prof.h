// Per-kernel HIP-event profiling used by bench.py's roofline block.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

namespace hwang {
namespace prof {

enum KernelId {
  K_RESIDUAL = 0,   // dequant + inverse transforms
  K_INTER = 1,      // motion compensation + residual add
  K_INTRA = 2,      // intra prediction wavefront + residual add
  K_DEBLOCK = 3,    // in-loop deblocking wavefront
  K_RGB = 4,        // I420 -> RGB24
  K_MV = 5,         // motion-vector derivation of P macroblocks
  K_BS = 6,         // boundary strengths + filter parameters (k_bs), when it runs as a stage of its own
  K_CHAIN = 7,      // k_chain: inter + intra + deblocking of a batch's pictures as one launch of overlapping stages
  K_COUNT = 8,
};

bool enabled();

// RAII-free helpers: begin() records a start event on `stream`, end() the
// stop event and files (kernel, algorithmic bytes); totals are resolved
// lazily in read() once the events have completed.
struct Span {
  hipEvent_t start = nullptr;
  hipEvent_t stop = nullptr;
};
Span begin(hipStream_t stream);
void end(const Span& span, hipStream_t stream, KernelId kernel,
         uint64_t algorithmic_bytes);

}  // namespace prof
}  // namespace hwang
