// FETCH_SIZE calibration for the access shapes of this repo's kernels (MI355X_MICROARCH.md, HBM:
// "other access widths are uncalibrated: calibrate on a known byte count in your own access pattern").  Every kernel
// reads a 1 GiB buffer exactly once (far beyond L2 + Infinity Cache), in the shape named:
//   stream16    16 B per lane, a wave covers 1 KiB contiguous (the guide's calibrated case: counter = bytes / 2)
//   rows64      k_deblock's tile loads: four lanes x 16 B = 64 B of one picture row, the next four lanes the row below
//               (pitch 2 048 B), i.e. 64-byte segments of sixteen different rows per wave
//   dword_una   k_inter's per-lane path: one unaligned dword per lane (offset + 1), lanes 4 bytes apart
//   qword_una   k_inter's window loads: 8 B per lane at byte offset 3 mod 8 ... lanes 8 bytes apart
//   rows24      k_inter's luma window: 3 lanes x 8 B = 24 B per row, 21 rows (pitch 2 048), per wave
// build: hipcc --offload-arch=gfx950 -O2 fetch_calib.hip -o fetch_calib ; run under
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace -d out -o calib --output-format csv -- ./fetch_calib
// and compare Counter_Value (KiB) per kernel with the bytes each kernel prints.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>

#define AS1 __attribute__((address_space(1)))
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

__global__ void stream16(const uint8_t* p, uint32_t* sink, size_t n) {
  size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 16;
  uint32_t acc = 0;
  for (; i + 16 <= n; i += (size_t)gridDim.x * blockDim.x * 16) {
    u32x4 v = *(const AS1 u32x4*)(p + i);
    acc += v.x ^ v.y ^ v.z ^ v.w;
  }
  if (acc == 0x12345678) *sink = acc;
}
// a "picture" of pitch 2048: every wave reads 64 B of 16 consecutive rows at column block c
__global__ void rows64(const uint8_t* p, uint32_t* sink, size_t n) {
  const int lane = threadIdx.x & 63;
  const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((size_t)gridDim.x * blockDim.x) >> 6;
  const size_t tiles = n / (16 * 2048) * 32;  // 16-row stripes x 32 column blocks of 64 B
  uint32_t acc = 0;
  for (size_t t = wave; t < tiles; t += nwaves) {
    const size_t stripe = t / 32, cb = t % 32;
    u32x4 v = *(const AS1 u32x4*)(p + stripe * 16 * 2048 + (size_t)(lane >> 2) * 2048 + cb * 64 + (lane & 3) * 16);
    acc += v.x ^ v.y ^ v.z ^ v.w;
  }
  if (acc == 0x12345678) *sink = acc;
}
__global__ void dword_una(const uint8_t* p, uint32_t* sink, size_t n) {
  size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4 + 1;
  uint32_t acc = 0;
  for (; i + 4 <= n; i += (size_t)gridDim.x * blockDim.x * 4) acc += *(const AS1 uint32_t*)(p + i);
  if (acc == 0x12345678) *sink = acc;
}
__global__ void qword_una(const uint8_t* p, uint32_t* sink, size_t n) {
  size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 8 + 3;
  uint32_t acc = 0;
  for (; i + 8 <= n; i += (size_t)gridDim.x * blockDim.x * 8) {
    u32x2 v = *(const AS1 u32x2*)(p + i);
    acc += v.x ^ v.y;
  }
  if (acc == 0x12345678) *sink = acc;
}
// k_inter's luma window: 63 lanes, lane -> (row lane / 3, 8-byte column lane % 3), 21 rows x 24 B at an odd column
__global__ void rows24(const uint8_t* p, uint32_t* sink, size_t n) {
  const int lane = threadIdx.x & 63;
  const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((size_t)gridDim.x * blockDim.x) >> 6;
  const size_t per_stripe = 2048 / 16;            // one window per 16 columns ("macroblock"), stripes of 16 rows
  const size_t windows = n / (16 * 2048) * per_stripe - per_stripe;  // (the last stripe's windows would run past the end)
  uint32_t acc = 0;
  for (size_t t = wave; t < windows; t += nwaves) {
    const size_t stripe = t / per_stripe, mb = t % per_stripe;
    if (lane < 63) {
      const int r = lane / 3, c = lane % 3;
      const size_t off = stripe * 16 * 2048 + (size_t)r * 2048 + mb * 16 + c * 8 + 5;
      if (off + 8 <= n) {
        u32x2 v = *(const AS1 u32x2*)(p + off);
        acc += v.x ^ v.y;
      }
    }
  }
  if (acc == 0x12345678) *sink = acc;
}

int main() {
  const size_t n = size_t(1) << 30;
  uint8_t* buf;
  uint32_t* sink;
  if (hipMalloc(&buf, n + 64) != hipSuccess || hipMalloc(&sink, 4) != hipSuccess) return 1;
  hipMemset(buf, 1, n + 64);
  hipDeviceSynchronize();
  struct { const char* name; void (*k)(const uint8_t*, uint32_t*, size_t); double bytes; } ks[] = {
      {"stream16", stream16, (double)n},
      {"rows64", rows64, (double)n},
      {"dword_una", dword_una, (double)n},
      {"qword_una", qword_una, (double)n},
      // 21 rows x 24 B per 16 x 16 window: the bytes REQUESTED are 1.97 x the buffer (windows overlap: rows 16-20 of a
      // stripe are the next stripe's rows 0-4, and 24 B per 16 columns); the bytes TOUCHED are the buffer once
      {"rows24", rows24, (double)n},
  };
  for (auto& k : ks) {
    for (int rep = 0; rep < 2; ++rep) {
      hipLaunchKernelGGL(k.k, dim3(4096), dim3(256), 0, 0, buf, sink, n);
      hipDeviceSynchronize();
    }
    printf("%s: buffer bytes %.0f\n", k.name, k.bytes);
  }
  return 0;
}
