// How many kernels from different HIP streams does one process get to run at once?  N streams, each with K launches of a
// one-wavefront kernel that spins for about `us` microseconds; wall time / (K * us) = 1 / concurrency.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ void spin(long long ticks, int* sink) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) {}
  if (sink && ticks < 0) *sink = 1;
}

int main(int argc, char** argv) {
  const int us = argc > 1 ? std::atoi(argv[1]) : 500, K = argc > 2 ? std::atoi(argv[2]) : 40;
  int rate_khz = 0;
  hipDeviceGetAttribute(&rate_khz, hipDeviceAttributeWallClockRate, 0);
  const long long ticks = static_cast<long long>(us) * rate_khz / 1000;
  for (int n : {1, 2, 3, 4, 5, 6, 8, 12, 16, 24, 32}) {
    std::vector<hipStream_t> s(n);
    for (auto& q : s) hipStreamCreateWithFlags(&q, hipStreamNonBlocking);
    for (auto& q : s) hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, q, 1000, nullptr);
    hipDeviceSynchronize();
    const auto t0 = std::chrono::steady_clock::now();
    for (int k = 0; k < K; ++k)
      for (auto& q : s) hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, q, ticks, nullptr);
    hipDeviceSynchronize();
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    std::printf("streams=%d wall_ms=%.2f serial_ms=%.2f concurrency=%.2f\n", n, ms, n * K * us / 1000.0, n * K * us / 1000.0 / ms);
    for (auto& q : s) hipStreamDestroy(q);
  }
  return 0;
}
