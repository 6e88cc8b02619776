// NUMA placement helpers ("pin lane threads to the GPU's NUMA node").  Host only, Linux sysfs.
// A GPU hangs off one socket's PCIe root; the threads that fill its pinned staging buffers, and the buffers
// themselves (pinned pages are allocated on the node of the thread that asks for them), belong on that socket when a
// node drives several GPUs.  Everything here fails soft: no sysfs, one node, or an affinity mask that leaves nothing
// -> nothing changes.
#pragma once

#include <sched.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

namespace hwang {
namespace numa {

// NUMA node of a PCI device given as "0000:c1:00.0"; -1 if unknown.
inline int node_of_pci_device(const char* bus_id) {
  std::string id(bus_id);
  for (char& c : id) c = static_cast<char>(tolower(c));
  const std::string path = "/sys/bus/pci/devices/" + id + "/numa_node";
  FILE* f = fopen(path.c_str(), "r");
  if (!f) return -1;
  int node = -1;
  if (fscanf(f, "%d", &node) != 1) node = -1;
  fclose(f);
  return node;
}

// A sysfs cpulist ("0-15,128-143\n") into `set`; returns how many CPUs it names.  Parsed in place with strtol: lane
// threads of several devices call this at the same time, so nothing with hidden state (strtok, which this used to
// be written with, keeps its cursor in a process-wide static: concurrent callers walked each other's buffers).
inline int parse_cpulist(const char* text, cpu_set_t* set) {
  CPU_ZERO(set);
  int n = 0;
  for (const char* p = text; *p;) {
    char* end = nullptr;
    const long lo = strtol(p, &end, 10);
    if (end == p) {  // a separator
      ++p;
      continue;
    }
    long hi = lo;
    p = end;
    if (*p == '-') {
      hi = strtol(p + 1, &end, 10);
      if (end == p + 1) hi = lo;
      p = end;
    }
    for (long c = lo < 0 ? 0 : lo; c <= hi && c < CPU_SETSIZE; ++c) {
      if (!CPU_ISSET(static_cast<int>(c), set)) ++n;
      CPU_SET(static_cast<int>(c), set);
    }
  }
  return n;
}

// CPUs of a node into `set`; false if the node has no list.
inline bool cpus_of_node(int node, cpu_set_t* set) {
  char path[96];
  snprintf(path, sizeof(path), "/sys/devices/system/node/node%d/cpulist", node);
  FILE* f = fopen(path, "r");
  if (!f) return false;
  char buf[4096];
  const bool got = fgets(buf, sizeof(buf), f) != nullptr;
  fclose(f);
  if (!got) return false;
  return parse_cpulist(buf, set) > 0;
}

// Restricts the calling thread to the CPUs of `node` that it is allowed to use anyway.  Returns the number of CPUs
// it ended up with, 0 if nothing was changed.
inline int bind_current_thread_to_node(int node) {
  if (node < 0) return 0;
  cpu_set_t want, have;
  if (!cpus_of_node(node, &want)) return 0;
  if (sched_getaffinity(0, sizeof(have), &have) != 0) return 0;
  cpu_set_t both;
  CPU_AND(&both, &want, &have);
  const int n = CPU_COUNT(&both);
  if (n == 0 || n == CPU_COUNT(&have)) return 0;  // nothing left, or nothing to gain
  return sched_setaffinity(0, sizeof(both), &both) == 0 ? n : 0;
}

// HWANG_NUMA_AFFINITY: "0" off, "1" on, unset -> `by_default`.
inline bool enabled(bool by_default) {
  const char* e = getenv("HWANG_NUMA_AFFINITY");
  if (!e || !*e) return by_default;
  return atoi(e) != 0;
}

}  // namespace numa
}  // namespace hwang
