#include "hwang_amd.h"
#include "sampler.h"
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <chrono>
#include <algorithm>
#include <cstring>
#include <string>
#include <linux/perf_event.h>
#include <sys/ioctl.h>
#include <sys/syscall.h>
#include <unistd.h>
// HWANG_PROBE_PERF=1: hardware counters of this thread around the timed calls (where the kernel lets a process count
// its own events: the GPU boxes' bare-metal hosts may, the build container's VM has no PMU)
struct Perf {
  struct Ev { const char* name; uint32_t type; uint64_t config; int fd; };
  std::vector<Ev> evs;
  Perf() {
    const Ev want[] = {{"cycles", PERF_TYPE_HARDWARE, PERF_COUNT_HW_CPU_CYCLES, -1}, {"instructions", PERF_TYPE_HARDWARE, PERF_COUNT_HW_INSTRUCTIONS, -1},
                       {"branches", PERF_TYPE_HARDWARE, PERF_COUNT_HW_BRANCH_INSTRUCTIONS, -1}, {"branch-misses", PERF_TYPE_HARDWARE, PERF_COUNT_HW_BRANCH_MISSES, -1},
                       {"stalled-frontend", PERF_TYPE_HARDWARE, PERF_COUNT_HW_STALLED_CYCLES_FRONTEND, -1},
                       {"stalled-backend", PERF_TYPE_HARDWARE, PERF_COUNT_HW_STALLED_CYCLES_BACKEND, -1},
                       {"L1d-load-misses", PERF_TYPE_HW_CACHE, PERF_COUNT_HW_CACHE_L1D | (PERF_COUNT_HW_CACHE_OP_READ << 8) | (PERF_COUNT_HW_CACHE_RESULT_MISS << 16), -1},
                       {"L1i-load-misses", PERF_TYPE_HW_CACHE, PERF_COUNT_HW_CACHE_L1I | (PERF_COUNT_HW_CACHE_OP_READ << 8) | (PERF_COUNT_HW_CACHE_RESULT_MISS << 16), -1}};
    for (Ev e : want) {
      perf_event_attr a;
      memset(&a, 0, sizeof(a));
      a.size = sizeof(a);
      a.type = e.type;
      a.config = e.config;
      a.disabled = 1;
      a.exclude_kernel = 1;
      a.exclude_hv = 1;
      e.fd = (int)syscall(SYS_perf_event_open, &a, 0, -1, -1, 0);
      if (e.fd < 0) fprintf(stderr, "perf: %s unavailable (%s)\n", e.name, strerror(errno));
      else evs.push_back(e);
    }
  }
  void start() { for (Ev& e : evs) { ioctl(e.fd, PERF_EVENT_IOC_RESET, 0); ioctl(e.fd, PERF_EVENT_IOC_ENABLE, 0); } }
  void stop(double pictures) {
    for (Ev& e : evs) {
      ioctl(e.fd, PERF_EVENT_IOC_DISABLE, 0);
      uint64_t v = 0;
      if (read(e.fd, &v, 8) == 8) printf("perf: %-18s %14llu  %12.0f per picture\n", e.name, (unsigned long long)v, v / pictures);
    }
  }
};
#include <cerrno>
#include <sys/mman.h>
// HWANG_PROBE_ISAMPLE=period: a profile by INSTRUCTIONS retired (one sample every `period` user-mode instructions of this
// thread, written to isamples.txt - addresses of the no-pie binary, for addr2line / annotate.py), where the kernel
// allows it.  Time-based samples over-weight the arithmetic decoder's dependent chain; this says where the
// instructions are.
struct ISampler {
  int fd = -1;
  void* ring = nullptr;
  size_t pages = 1024;
  bool start(long period, bool cycles = false) {
    perf_event_attr a;
    memset(&a, 0, sizeof(a));
    a.size = sizeof(a);
    a.type = PERF_TYPE_HARDWARE;
    a.config = cycles ? PERF_COUNT_HW_CPU_CYCLES : PERF_COUNT_HW_INSTRUCTIONS;  // HWANG_PROBE_CSAMPLE=period: by cycles instead
    a.sample_period = (uint64_t)period;
    a.sample_type = PERF_SAMPLE_IP;
    a.disabled = 1;
    a.exclude_kernel = 1;
    a.exclude_hv = 1;
    a.precise_ip = 2;
    fd = (int)syscall(SYS_perf_event_open, &a, 0, -1, -1, 0);
    if (fd < 0) { a.precise_ip = 0; fd = (int)syscall(SYS_perf_event_open, &a, 0, -1, -1, 0); }
    if (fd < 0) { fprintf(stderr, "isample: unavailable (%s)\n", strerror(errno)); return false; }
    ring = mmap(nullptr, (pages + 1) * 4096, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    if (ring == MAP_FAILED) { fprintf(stderr, "isample: mmap failed (%s)\n", strerror(errno)); ring = nullptr; return false; }
    ioctl(fd, PERF_EVENT_IOC_RESET, 0);
    ioctl(fd, PERF_EVENT_IOC_ENABLE, 0);
    return true;
  }
  void stop(const char* path) {
    if (!ring) return;
    ioctl(fd, PERF_EVENT_IOC_DISABLE, 0);
    perf_event_mmap_page* mp = (perf_event_mmap_page*)ring;
    const uint64_t head = mp->data_head, size = pages * 4096;
    __sync_synchronize();
    const uint8_t* data = (const uint8_t*)ring + 4096;
    uint64_t pos = head > size ? head - size : 0;   // (an overwritten ring: the newest `size` bytes; records may be cut at the start)
    FILE* f = fopen(path, "w");
    long n = 0;
    while (pos + 8 <= head) {
      perf_event_header h;
      for (size_t i = 0; i < sizeof(h); ++i) ((uint8_t*)&h)[i] = data[(pos + i) % size];
      if (h.size < 8) break;
      if (h.type == PERF_RECORD_SAMPLE && h.size >= 16) {
        uint64_t ip = 0;
        for (size_t i = 0; i < 8; ++i) ((uint8_t*)&ip)[i] = data[(pos + 8 + i) % size];
        fprintf(f, "%llx\n", (unsigned long long)ip);
        ++n;
      }
      pos += h.size;
    }
    fclose(f);
    fprintf(stderr, "isample: %ld samples -> %s%s\n", n, path, head > size ? " (the ring wrapped: the newest samples)" : "");
  }
};
static std::vector<uint8_t> rd(const char* p){FILE*f=fopen(p,"rb");fseek(f,0,SEEK_END);long n=ftell(f);fseek(f,0,SEEK_SET);std::vector<uint8_t> v(n);if(fread(v.data(),1,n,f)!=(size_t)n)exit(2);fclose(f);return v;}
int main(int argc,char**argv){
  auto avcc=rd("avcc.bin"); auto data=rd(argv[1]); auto offs=rd("offs.bin"); auto sizes=rd("sizes.bin");
  size_t n=offs.size()/8; int64_t out[8]; char err[256];
  int reps = argc>2?atoi(argv[2]):1; int gop = argc>3?atoi(argv[3]):60;
  // sparse: want the frame just before each keyframe-aligned GOP end -> B pictures are skipped
  size_t first = argc>4?atoi(argv[4]):gop-3;
  std::vector<uint64_t> wanted; for(size_t g=first; g<n; g+=(getenv("ALL")?1:gop)) wanted.push_back(g);   // ALL=1: every picture from `first` on
  const bool perf_on = getenv("HWANG_PROBE_PERF") != nullptr;
  Perf* perf = perf_on ? new Perf() : nullptr;
  const bool by_cycles = getenv("HWANG_PROBE_CSAMPLE") != nullptr;
  const long iperiod = by_cycles ? atol(getenv("HWANG_PROBE_CSAMPLE")) : getenv("HWANG_PROBE_ISAMPLE") ? atol(getenv("HWANG_PROBE_ISAMPLE")) : 0;
  ISampler isampler;
  if (!perf_on && !iperiod) sampler_start();
  if (iperiod) isampler.start(iperiod, by_cycles);
  if (perf) perf->start();
  double best=1e9;
  for(int r=0;r<reps;r++){
   auto t0=std::chrono::steady_clock::now();
   int rc=hwang_h264_probe(avcc.data(),avcc.size(),data.data(),(const uint64_t*)offs.data(),(const uint64_t*)sizes.data(),n,wanted.data(),wanted.size(),out,err,sizeof err);
   if(rc){printf("err %s\n",err);return 1;}
   double dt=std::chrono::duration<double>(std::chrono::steady_clock::now()-t0).count();
   best=std::min(best,dt);
  }
  if (perf) perf->stop((double)out[0] * reps);
  if (iperiod) isampler.stop("isamples.txt");
  if (!perf_on && !iperiod) sampler_dump("samples.txt");
  printf("decoded %lld skipped %lld: best %.3fs -> %.3f ms per decoded picture\n",(long long)out[0],(long long)out[1],best,best*1e3/out[0]);
}
