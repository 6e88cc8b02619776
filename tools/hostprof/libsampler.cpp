// Development aid: process-wide SIGPROF sampler loadable from Python (ctypes).
// hw_sampler_start() arms ITIMER_PROF (it counts the CPU time of all threads
// and interrupts whichever is running); hw_sampler_dump(path) writes
// "pc tid thread-name" lines and /proc/self/maps so that the samples can be
// attributed offline (tools/hostprof/attribute.py).
#include <signal.h>
#include <sys/syscall.h>
#include <sys/time.h>
#include <ucontext.h>
#include <unistd.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
static uint64_t g_pc[1 << 20];
static int g_tid[1 << 20];
static volatile int g_n = 0;
static void on_prof(int, siginfo_t*, void* uc) {
  const int i = __atomic_fetch_add(&g_n, 1, __ATOMIC_RELAXED);
  if (i < (1 << 20)) {
    g_pc[i] = ((ucontext_t*)uc)->uc_mcontext.gregs[REG_RIP];
    g_tid[i] = (int)syscall(SYS_gettid);
  }
}
extern "C" void hw_sampler_start() {
  g_n = 0;
  struct sigaction sa = {};
  sa.sa_sigaction = on_prof;
  sa.sa_flags = SA_SIGINFO | SA_RESTART;
  sigaction(SIGPROF, &sa, 0);
  struct itimerval it = {{0, 500}, {0, 500}};
  setitimer(ITIMER_PROF, &it, 0);
}
extern "C" void hw_sampler_dump(const char* path) {
  struct itimerval it = {};
  setitimer(ITIMER_PROF, &it, 0);
  const int n = g_n < (1 << 20) ? g_n : (1 << 20);
  std::map<int, std::string> names;
  FILE* f = fopen(path, "w");
  for (int i = 0; i < n; ++i) {
    auto itn = names.find(g_tid[i]);
    if (itn == names.end()) {
      char p[64], nm[64] = "gone";
      snprintf(p, sizeof p, "/proc/self/task/%d/comm", g_tid[i]);
      if (FILE* c = fopen(p, "r")) { if (fgets(nm, sizeof nm, c)) nm[strcspn(nm, "\n")] = 0; fclose(c); }
      itn = names.emplace(g_tid[i], nm).first;
    }
    fprintf(f, "%llx %d %s\n", (unsigned long long)g_pc[i], g_tid[i], itn->second.c_str());
  }
  fclose(f);
  char mp[512];
  snprintf(mp, sizeof mp, "%s.maps", path);
  FILE* in = fopen("/proc/self/maps", "r");
  FILE* out = fopen(mp, "w");
  char line[1024];
  while (in && out && fgets(line, sizeof line, in)) if (strstr(line, " r-x")) fputs(line, out);
  if (in) fclose(in);
  if (out) fclose(out);
}
