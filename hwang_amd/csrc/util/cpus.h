// CPUs this process may actually use: hardware threads, capped by a cgroup CPU-bandwidth quota when there is one
// (cgroup v2 cpu.max, v1 cfs_quota).  Entropy decoding is what the host spends CPU on; more runnable threads than
// CPUs only makes the quota throttle the whole process in bursts.
#pragma once

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <thread>

namespace hwang {

inline int32_t usable_cpus() {
  unsigned hw = std::thread::hardware_concurrency();
  double cpus = hw ? hw : 4;
  if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
    char quota[32];
    long period = 0;
    if (fscanf(f, "%31s %ld", quota, &period) == 2 && period > 0 && quota[0] != 'm')
      cpus = std::min(cpus, atof(quota) / period);
    fclose(f);
  } else {
    long q = -1, per = 0;
    if (FILE* fq = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {
      if (fscanf(fq, "%ld", &q) != 1) q = -1;
      fclose(fq);
    }
    if (FILE* fp = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) {
      if (fscanf(fp, "%ld", &per) != 1) per = 0;
      fclose(fp);
    }
    if (q > 0 && per > 0) cpus = std::min(cpus, static_cast<double>(q) / per);
  }
  return std::max<int32_t>(1, static_cast<int32_t>(cpus + 0.5));
}

}  // namespace hwang
