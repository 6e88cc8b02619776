// Development aid: step latency of the row-pipeline skeleton used by k_mv /
// k_intra / k_deblock (one workgroup, W waves, wave w owns rows w, w+W, ...;
// row r may process macroblock x once row r-1 has completed x+2).
// Variants isolate what a step costs: the LDS poll, s_sleep, the LDS line
// hand-off, dependent ALU work, a global store that has to land before the
// counter is raised, prefetched global loads.
// build: hipcc --offload-arch=gfx950 -O3 rowpipe.hip -o rowpipe ; run: ./rowpipe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef volatile __attribute__((address_space(3))) int lds_counter;
#define HW_AS1 __attribute__((address_space(1)))
struct Params { int rows, cols, alu, flags; uint32_t* buf; };
enum { F_SLEEP = 1, F_LINE = 2, F_STORE = 4, F_LOAD = 8, F_LAG0 = 16 };

template <int W>
__global__ __launch_bounds__(W * 64) void k_pipe(Params p) {
  __shared__ int s_progress[513];
  __shared__ uint32_t s_line[W][128][4];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  lds_counter* progress = (lds_counter*)s_progress;
  for (int i = threadIdx.x; i <= p.rows; i += blockDim.x) s_progress[i] = i == 0 ? (1 << 30) : 0;
  __syncthreads();
  uint32_t acc = lane;
  HW_AS1 uint32_t* g = (HW_AS1 uint32_t*)p.buf + (size_t)blockIdx.x * 512 * 128 * 64;
  for (int r = wave; r < p.rows; r += W) {
    uint32_t pre = 0;
    if (p.flags & F_LOAD) pre = g[(r * 128 + 0) * 64 + lane];
    for (int x = 0; x < p.cols; ++x) {
      uint32_t nxt = 0;
      if ((p.flags & F_LOAD) && x + 1 < p.cols) nxt = g[(r * 128 + x + 1) * 64 + lane];
      const int need = (p.flags & F_LAG0) ? x + 1 : (x + 2 < p.cols ? x + 2 : p.cols);
      if (p.flags & F_SLEEP) { while (progress[r] < need) __builtin_amdgcn_s_sleep(1); }
      else { while (progress[r] < need) {} }
      __asm__ volatile("" ::: "memory");
      if (p.flags & F_LINE) acc += s_line[(wave + W - 1) % W][x][lane & 3];
      if (p.flags & F_LOAD) { __asm__ volatile("s_waitcnt vmcnt(1)" ::: "memory"); acc += pre; }
      for (int k = 0; k < p.alu; ++k) acc = acc * 1664525u + 1013904223u;   // dependent chain
      if (p.flags & F_LINE) { if (lane < 4) s_line[wave][x][lane] = acc; }
      if (p.flags & F_STORE) {
        g[(r * 128 + x) * 64 + lane] = acc;
        __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      progress[r + 1] = x + 1;
      pre = nxt;
    }
  }
  if (acc == 0x12345) p.buf[0] = acc;
}

template <int W> float run(Params p, int grid) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL(k_pipe<W>, dim3(grid), dim3(W * 64), 0, 0, p);
  hipDeviceSynchronize();
  hipEventRecord(a);
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(k_pipe<W>, dim3(grid), dim3(W * 64), 0, 0, p);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  return ms / 5 * 1000;
}
int main() {
  Params p; p.rows = 68; p.cols = 120;
  hipMalloc(&p.buf, (size_t)8 * 512 * 128 * 64 * 4);
  hipMemset(p.buf, 0, (size_t)8 * 512 * 128 * 64 * 4);
  struct V { const char* name; int flags, alu; } vs[] = {
    {"poll only (spin)", 0, 0}, {"poll only (s_sleep 1)", F_SLEEP, 0},
    {"+ LDS line", F_SLEEP | F_LINE, 0}, {"+ 100 dependent ALU", F_SLEEP | F_LINE, 100},
    {"+ 300 dependent ALU", F_SLEEP | F_LINE, 300},
    {"line + store landed", F_SLEEP | F_LINE | F_STORE, 0},
    {"line + prefetched load", F_SLEEP | F_LINE | F_LOAD, 0},
    {"line + load + store + 100 ALU", F_SLEEP | F_LINE | F_LOAD | F_STORE, 100},
    {"no lag (x+1), poll only", F_SLEEP | F_LAG0, 0},
  };
  for (auto& v : vs) {
    p.flags = v.flags; p.alu = v.alu;
    float t16 = run<16>(p, 1), t4 = run<4>(p, 1), t1 = run<1>(p, 1);
    // steps: W waves -> ceil(rows/W) passes of cols + 2 per wave of skew
    auto steps = [&](int W) { return ((p.rows + W - 1) / W) * p.cols + 2 * (W - 1); };
    printf("%-34s W16 %8.1f us (%5.0f ns/step)  W4 %8.1f us (%5.0f ns/step)  W1 %8.1f us (%5.0f ns/MB)\n", v.name, t16,
           t16 * 1000 / steps(16), t4, t4 * 1000 / steps(4), t1, t1 * 1000 / (p.rows * p.cols));
  }
  return 0;
}
