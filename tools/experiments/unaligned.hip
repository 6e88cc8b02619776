// Does gfx950 under this driver allow unaligned dword loads from global memory and LDS?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void k(const uint8_t* src, uint32_t* out) {
  __shared__ uint8_t lds[512];
  const int t = threadIdx.x;
  for (int i = t; i < 512; i += 64) lds[i] = src[i];
  __syncthreads();
  out[t] = *reinterpret_cast<const uint32_t*>(src + t * 5 + 1);          // global, odd addresses
  out[64 + t] = *reinterpret_cast<const uint32_t*>(lds + t * 5 + 3);     // LDS, odd addresses
}
int main() {
  uint8_t h[512]; for (int i = 0; i < 512; ++i) h[i] = (uint8_t)(i * 7 + 3);
  uint8_t* d; uint32_t* o; uint32_t r[128];
  hipMalloc(&d, 512); hipMalloc(&o, 512); hipMemcpy(d, h, 512, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o);
  hipError_t e = hipDeviceSynchronize();
  printf("sync: %s\n", hipGetErrorString(e));
  hipMemcpy(r, o, 512, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int t = 0; t < 64; ++t) {
    uint32_t g = 0, l = 0;
    for (int b = 0; b < 4; ++b) { g |= (uint32_t)h[t * 5 + 1 + b] << (8 * b); l |= (uint32_t)h[t * 5 + 3 + b] << (8 * b); }
    bad += (r[t] != g) + (r[64 + t] != l);
  }
  printf("unaligned dword loads: %s (%d mismatches)\n", bad ? "WRONG" : "ok", bad);
  return 0;
}
