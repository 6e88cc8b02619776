#include "prof.h"

#include "../../../include/hwang_amd.h"

#include <atomic>
#include <mutex>
#include <vector>

namespace hwang {
namespace prof {
namespace {

std::atomic<bool> g_enabled{false};
std::mutex g_m;
struct Pending {
  Span span;
  KernelId kernel;
  uint64_t bytes;
};
std::vector<Pending> g_pending;
std::vector<hipEvent_t> g_free_events;
struct Total {
  double launches = 0, usec = 0, bytes = 0;
};
Total g_totals[K_COUNT];

const char* kNames[K_COUNT] = {"residual_idct", "inter_pred", "intra_pred",
                               "deblock", "yuv420_to_rgb24", "mv_derivation", "boundary_strength", "chain"};

hipEvent_t take_event() {
  if (!g_free_events.empty()) {
    hipEvent_t e = g_free_events.back();
    g_free_events.pop_back();
    return e;
  }
  hipEvent_t e = nullptr;
  if (hipEventCreate(&e) != hipSuccess) return nullptr;
  return e;
}

// Folds finished spans into the totals.  With `wait`, blocks on each.
void resolve_locked(bool wait) {
  std::vector<Pending> keep;
  for (Pending& p : g_pending) {
    if (wait) (void)hipEventSynchronize(p.span.stop);
    if (hipEventQuery(p.span.stop) != hipSuccess) {
      keep.push_back(p);
      continue;
    }
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, p.span.start, p.span.stop) == hipSuccess) {
      Total& t = g_totals[p.kernel];
      t.launches += 1;
      t.usec += ms * 1000.0;
      t.bytes += static_cast<double>(p.bytes);
    }
    g_free_events.push_back(p.span.start);
    g_free_events.push_back(p.span.stop);
  }
  g_pending.swap(keep);
}

}  // namespace

bool enabled() { return g_enabled.load(std::memory_order_relaxed); }

Span begin(hipStream_t stream) {
  Span s;
  std::lock_guard<std::mutex> lk(g_m);
  s.start = take_event();
  s.stop = take_event();
  if (s.start) (void)hipEventRecord(s.start, stream);
  return s;
}

void end(const Span& span, hipStream_t stream, KernelId kernel,
         uint64_t algorithmic_bytes) {
  if (!span.start || !span.stop) return;
  (void)hipEventRecord(span.stop, stream);
  std::lock_guard<std::mutex> lk(g_m);
  g_pending.push_back({span, kernel, algorithmic_bytes});
  if (g_pending.size() > 4096) resolve_locked(false);
}

}  // namespace prof
}  // namespace hwang

using namespace hwang::prof;

extern "C" {

void hwang_prof_enable(int on) { g_enabled.store(on != 0); }
int hwang_prof_num_kernels(void) { return K_COUNT; }
const char* hwang_prof_kernel_name(int kernel_id) {
  return kernel_id >= 0 && kernel_id < K_COUNT ? kNames[kernel_id] : "";
}
int hwang_prof_read(int kernel_id, double out[3]) {
  if (kernel_id < 0 || kernel_id >= K_COUNT) return 1;
  std::lock_guard<std::mutex> lk(g_m);
  resolve_locked(true);
  out[0] = g_totals[kernel_id].launches;
  out[1] = g_totals[kernel_id].usec;
  out[2] = g_totals[kernel_id].bytes;
  return 0;
}
void hwang_prof_reset(void) {
  std::lock_guard<std::mutex> lk(g_m);
  resolve_locked(true);
  for (Total& t : g_totals) t = Total();
}

}  // extern "C"
