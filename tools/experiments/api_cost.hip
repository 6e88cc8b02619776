// Development aid: process CPU time (all threads, user + system) that the HIP
// runtime spends per call pattern the executor uses, to find what keeps the
// runtime's background threads in ioctl().  build: hipcc --offload-arch=gfx950 -O2
#include <hip/hip_runtime.h>
#include <sys/resource.h>
#include <chrono>
#include <cstdio>
#include <ctime>
#include <vector>
__global__ void k_nop(unsigned* p) { if (p && threadIdx.x == 12345) *p = 1; }
__global__ void k_flag(volatile unsigned* p, unsigned v) { if (threadIdx.x == 0) { __threadfence_system(); *p = v; } }
__global__ void k_spin(unsigned* p, int n) { for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(100); if (p && threadIdx.x == 12345) *p = 1; }
static double cpu_now() { rusage r; getrusage(RUSAGE_SELF, &r); return r.ru_utime.tv_sec + r.ru_utime.tv_usec * 1e-6 + r.ru_stime.tv_sec + r.ru_stime.tv_usec * 1e-6; }
static double sys_now() { rusage r; getrusage(RUSAGE_SELF, &r); return r.ru_stime.tv_sec + r.ru_stime.tv_usec * 1e-6; }
static double wall_now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static volatile unsigned* g_done;  // pinned; the executor's way of learning about completion: no runtime wait at all
static unsigned g_ticket = 0;
template <class F> void run(const char* name, int n, F f, hipStream_t s0, hipStream_t s1) {
  f(0); (void)hipStreamSynchronize(s0); (void)hipStreamSynchronize(s1);
  const double c0 = cpu_now(), y0 = sys_now(), w0 = wall_now();
  for (int i = 0; i < n; ++i) f(i);
  // s0 finishes last in every pattern below
  hipLaunchKernelGGL(k_flag, dim3(1), dim3(64), 0, s0, g_done, ++g_ticket);
  while (*g_done != g_ticket) { timespec ts = {0, 40000}; nanosleep(&ts, nullptr); }
  const double c1 = cpu_now(), y1 = sys_now(), w1 = wall_now();
  printf("%-58s %8.1f us cpu/op (%6.1f sys)  %8.1f us wall/op\n", name, (c1 - c0) / n * 1e6, (y1 - y0) / n * 1e6, (w1 - w0) / n * 1e6);
}
int main() {
  hipStream_t s0, s1;
  (void)hipStreamCreateWithFlags(&s0, hipStreamNonBlocking); (void)hipStreamCreateWithFlags(&s1, hipStreamNonBlocking);
  void *hbuf, *dbuf; (void)hipHostMalloc(&hbuf, 1 << 20, hipHostMallocDefault); (void)hipMalloc(&dbuf, 1 << 20);
  unsigned* flag; (void)hipHostMalloc((void**)&flag, 4096, hipHostMallocDefault);
  std::vector<hipEvent_t> ev(8); for (auto& e : ev) (void)hipEventCreateWithFlags(&e, hipEventDisableTiming);
  (void)hipHostMalloc((void**)&g_done, 4096, hipHostMallocDefault); *g_done = 0;
  const int n = 2000;
  run("kernel launch (nop)", n, [&](int) { hipLaunchKernelGGL(k_nop, dim3(1), dim3(64), 0, s0, nullptr); }, s0, s1);
  run("    [probe] 8 kernel launches again", n / 8, [&](int) { for (int k = 0; k < 8; ++k) hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s0, nullptr, 40); }, s0, s1);
  run("8 kernel launches (200 us of GPU work each)", n / 8, [&](int) { for (int k = 0; k < 8; ++k) hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s0, nullptr, 40); }, s0, s1);
  run("    [probe] 8 kernel launches again", n / 8, [&](int) { for (int k = 0; k < 8; ++k) hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s0, nullptr, 40); }, s0, s1);
  run("hipMemcpyAsync H2D 48 KB pinned", n, [&](int) { (void)hipMemcpyAsync(dbuf, hbuf, 48 << 10, hipMemcpyHostToDevice, s0); }, s0, s1);
  run("    [probe] 8 kernel launches again", n / 8, [&](int) { for (int k = 0; k < 8; ++k) hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s0, nullptr, 40); }, s0, s1);
  run("hipMemcpyAsync H2D 48 KB pinned + kernel", n, [&](int) { (void)hipMemcpyAsync(dbuf, hbuf, 48 << 10, hipMemcpyHostToDevice, s0); hipLaunchKernelGGL(k_nop, dim3(1), dim3(64), 0, s0, nullptr); }, s0, s1);
  run("    [probe] 8 kernel launches again", n / 8, [&](int) { for (int k = 0; k < 8; ++k) hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s0, nullptr, 40); }, s0, s1);
  run("event record on s1 + wait on s0 + kernel on each", n, [&](int i) {
    hipLaunchKernelGGL(k_nop, dim3(1), dim3(64), 0, s1, nullptr);
    (void)hipEventRecord(ev[i & 7], s1); (void)hipStreamWaitEvent(s0, ev[i & 7], 0);
    hipLaunchKernelGGL(k_nop, dim3(1), dim3(64), 0, s0, nullptr); }, s0, s1);
  run("    [probe] 8 kernel launches again", n / 8, [&](int) { for (int k = 0; k < 8; ++k) hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s0, nullptr, 40); }, s0, s1);
  run("kernel writing a pinned flag (k_done style)", n, [&](int) { hipLaunchKernelGGL(k_nop, dim3(1), dim3(64), 0, s0, flag); }, s0, s1);
  run("    [probe] 8 kernel launches again", n / 8, [&](int) { for (int k = 0; k < 8; ++k) hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s0, nullptr, 40); }, s0, s1);
  run("kernel + hipStreamSynchronize", n, [&](int) { hipLaunchKernelGGL(k_nop, dim3(1), dim3(64), 0, s0, nullptr); (void)hipStreamSynchronize(s0); }, s0, s1);
  run("    [probe] 8 kernel launches again", n / 8, [&](int) { for (int k = 0; k < 8; ++k) hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s0, nullptr, 40); }, s0, s1);
  run("executor pattern: memcpy, kernel, event, wait, 8 x 200us kernels", n / 8, [&](int i) {
    (void)hipMemcpyAsync(dbuf, hbuf, 48 << 10, hipMemcpyHostToDevice, s1);
    hipLaunchKernelGGL(k_nop, dim3(1), dim3(64), 0, s1, nullptr);
    (void)hipEventRecord(ev[i & 7], s1); (void)hipStreamWaitEvent(s0, ev[i & 7], 0);
    for (int k = 0; k < 8; ++k) hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s0, nullptr, 40);
    hipLaunchKernelGGL(k_nop, dim3(1), dim3(64), 0, s0, flag); }, s0, s1);
  run("    [probe] 8 kernel launches again", n / 8, [&](int) { for (int k = 0; k < 8; ++k) hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s0, nullptr, 40); }, s0, s1);
  run("  same without the memcpy", n / 8, [&](int i) {
    hipLaunchKernelGGL(k_nop, dim3(1), dim3(64), 0, s1, nullptr);
    (void)hipEventRecord(ev[i & 7], s1); (void)hipStreamWaitEvent(s0, ev[i & 7], 0);
    for (int k = 0; k < 8; ++k) hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s0, nullptr, 40);
    hipLaunchKernelGGL(k_nop, dim3(1), dim3(64), 0, s0, flag); }, s0, s1);
  run("    [probe] 8 kernel launches again", n / 8, [&](int) { for (int k = 0; k < 8; ++k) hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s0, nullptr, 40); }, s0, s1);
  run("  same without event/wait (memcpy + kernel on s1, independent)", n / 8, [&](int i) {
    (void)hipMemcpyAsync(dbuf, hbuf, 48 << 10, hipMemcpyHostToDevice, s1);
    hipLaunchKernelGGL(k_nop, dim3(1), dim3(64), 0, s1, nullptr);
    for (int k = 0; k < 8; ++k) hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s0, nullptr, 40);
    hipLaunchKernelGGL(k_nop, dim3(1), dim3(64), 0, s0, flag); }, s0, s1);
  run("    [probe] 8 kernel launches again", n / 8, [&](int) { for (int k = 0; k < 8; ++k) hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s0, nullptr, 40); }, s0, s1);
  run("  everything on one stream (memcpy, 9 kernels)", n / 8, [&](int i) {
    (void)hipMemcpyAsync(dbuf, hbuf, 48 << 10, hipMemcpyHostToDevice, s0);
    hipLaunchKernelGGL(k_nop, dim3(1), dim3(64), 0, s0, nullptr);
    for (int k = 0; k < 8; ++k) hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s0, nullptr, 40);
    hipLaunchKernelGGL(k_nop, dim3(1), dim3(64), 0, s0, flag); }, s0, s1);
  run("    [probe] 8 kernel launches again", n / 8, [&](int) { for (int k = 0; k < 8; ++k) hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s0, nullptr, 40); }, s0, s1);
  run("  one stream, no memcpy (table read from pinned memory by a kernel)", n / 8, [&](int i) {
    hipLaunchKernelGGL(k_nop, dim3(1), dim3(64), 0, s0, nullptr);
    for (int k = 0; k < 8; ++k) hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s0, nullptr, 40);
    hipLaunchKernelGGL(k_nop, dim3(1), dim3(64), 0, s0, flag); }, s0, s1);
  run("    [probe] 8 kernel launches again", n / 8, [&](int) { for (int k = 0; k < 8; ++k) hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s0, nullptr, 40); }, s0, s1);
  {
    std::vector<hipEvent_t> many(512); for (auto& e : many) (void)hipEventCreateWithFlags(&e, hipEventDisableTiming);
    run("  two streams, a fresh event per batch, no memcpy", n / 8, [&](int i) {
      hipLaunchKernelGGL(k_nop, dim3(1), dim3(64), 0, s1, nullptr);
      (void)hipEventRecord(many[i & 511], s1); (void)hipStreamWaitEvent(s0, many[i & 511], 0);
      for (int k = 0; k < 8; ++k) hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s0, nullptr, 40);
      hipLaunchKernelGGL(k_nop, dim3(1), dim3(64), 0, s0, flag); }, s0, s1);
  run("    [probe] 8 kernel launches again", n / 8, [&](int) { for (int k = 0; k < 8; ++k) hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s0, nullptr, 40); }, s0, s1);
  }
  return 0;
}
