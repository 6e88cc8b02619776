// tests/test_host_numa.py: hwang_amd/csrc/util/numa.h's cpulist parser, called from many threads at once
// (ParallelRetriever's lane threads do that when a process drives several GPUs).
#include "../../hwang_amd/csrc/util/numa.h"

#include <atomic>
#include <thread>
#include <vector>

static int expect_count(int t) { return 16 + 16 + 1 + (t % 7); }

int main() {
  std::atomic<int> bad{0};
  std::vector<std::thread> threads;
  for (int t = 0; t < 32; ++t)
    threads.emplace_back([t, &bad] {
      for (int round = 0; round < 2000; ++round) {
        char text[128];
        // "a-b,c-d,e,f-g\n": per thread a different list, so a parser that shares state mixes them up
        const int base = t * 40;
        snprintf(text, sizeof(text), "%d-%d,%d-%d,%d,%d-%d\n", base, base + 15, base + 1000, base + 1015, base + 20, base + 30,
                 base + 30 + (t % 7) - 1);
        cpu_set_t set;
        const int n = hwang::numa::parse_cpulist(text, &set);
        bool ok = n == expect_count(t) - ((t % 7) == 0 ? 0 : 0);
        // (t % 7 == 0: the last range is "x-(x-1)", i.e. empty)
        ok = ok && CPU_ISSET(base, &set) && CPU_ISSET(base + 15, &set) && !CPU_ISSET(base + 16, &set) && CPU_ISSET(base + 20, &set) &&
             (base + 1015 >= CPU_SETSIZE || CPU_ISSET(base + 1015, &set));
        if (base + 1015 >= CPU_SETSIZE) ok = CPU_COUNT(&set) == n;  // clipped at CPU_SETSIZE: only self-consistency
        if (!ok) bad.fetch_add(1);
      }
    });
  for (auto& th : threads) th.join();
  cpu_set_t set;
  if (hwang::numa::parse_cpulist("0-7\n", &set) != 8 || hwang::numa::parse_cpulist("3\n", &set) != 1 || !CPU_ISSET(3, &set) ||
      hwang::numa::parse_cpulist("", &set) != 0 || hwang::numa::parse_cpulist("0-3,2-5", &set) != 6 ||
      hwang::numa::parse_cpulist("junk", &set) != 0)
    bad.fetch_add(1);
  printf(bad.load() ? "bad %d\n" : "ok %d\n", bad.load());
  return bad.load() ? 1 : 0;
}
