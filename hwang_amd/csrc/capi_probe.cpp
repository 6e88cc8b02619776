// C ABI: host-only probe of the H.264 parser (no GPU needed).  Runs the
// product's entropy decoder / DPB logic over a sequence of samples with a
// sink that only provides host buffers, and reports counters plus an FNV-1a
// digest of every record the GPU would have received.  Used by the CPU test
// suite to pin the host half of the backend (desync check, output order,
// sparse skipping) on machines without a GPU.
#include "../../include/hwang_amd.h"

#include "h264/h264_parser.h"
#include "hevc/hvcc.h"

#include <chrono>
#include <cstdio>
#include <linux/perf_event.h>
#include <sys/ioctl.h>
#include <sys/syscall.h>
#include <unistd.h>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

using namespace hwang;

namespace {

struct ProbeSink : public h264::PictureSink {
  struct Slot {
    bool busy = false;
    std::vector<amd::MbMotion> motion;
    std::vector<amd::PMotion> pmo;
    std::vector<uint32_t> pmx;
  };
  std::vector<Slot> slots = std::vector<Slot>(amd::kMaxSurfaces);
  // staging buffers of the pictures between begin_picture and end_picture: one set in serial mode, a ring with
  // entropy workers (PicBuffers::cookie names the set)
  struct Staging {
    std::vector<amd::MbInfo> mb;
    std::vector<uint32_t> coefs;
    std::vector<amd::SliceParams> slices = std::vector<amd::SliceParams>(256);  // as the decoder: MbInfo::slice is one byte
  };
  std::vector<Staging> staging = std::vector<Staging>(1);
  int staging_cursor = 0;
  bool pipelined = false;   // surfaces are handed out round robin
  int slot_cursor = 0;
  std::mutex mu;            // release_surface comes from any thread
  int nmb = 0;
  uint64_t digest = 1469598103934665603ull;
  uint64_t digest_noslot = 1469598103934665603ull;  // the same records with MbMotion::ref_slot blanked
  std::vector<int> outputs;  // slot per output (-1 skipped)
  int64_t coef_total = 0, inter_pics = 0;
  // HWANG_PROBE_TIMES=1 (serial decoding only): wall time between begin_picture and end_picture per picture kind
  bool times = getenv("HWANG_PROBE_TIMES") != nullptr;
  std::chrono::steady_clock::time_point t_begin;
  int64_t ns_kind[3] = {0, 0, 0}, n_kind[3] = {0, 0, 0}, coef_kind[3] = {0, 0, 0};  // intra-only, P, with B slices
  // ... and, where the kernel lets a process count its own events (the GPU boxes do, the build container's VM has no
  // PMU), user-mode instructions / cycles / branch misses per picture kind
  int pfd[3] = {-1, -1, -1};
  uint64_t pstart[3] = {0, 0, 0}, pkind[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
  void perf_open() {
    const uint64_t cfg[3] = {PERF_COUNT_HW_INSTRUCTIONS, PERF_COUNT_HW_CPU_CYCLES, PERF_COUNT_HW_BRANCH_MISSES};
    for (int i = 0; i < 3; ++i) {
      perf_event_attr a;
      std::memset(&a, 0, sizeof(a));
      a.size = sizeof(a);
      a.type = PERF_TYPE_HARDWARE;
      a.config = cfg[i];
      a.exclude_kernel = 1;
      a.exclude_hv = 1;
      pfd[i] = static_cast<int>(syscall(SYS_perf_event_open, &a, 0, -1, -1, 0));
    }
  }
  void perf_read(uint64_t out[3]) {
    for (int i = 0; i < 3; ++i) {
      out[i] = 0;
      if (pfd[i] >= 0 && read(pfd[i], &out[i], 8) != 8) out[i] = 0;
    }
  }
  ~ProbeSink() override {
    for (int fd : pfd)
      if (fd >= 0) close(fd);
  }

  bool fast = getenv("HWANG_PROBE_FAST") != nullptr;  // timing runs: no digest, no clearing
  void mix(const void* p, size_t n, bool both = true) {
    if (fast) return;
    const uint8_t* b = static_cast<const uint8_t*>(p);
    if (both)
      for (size_t i = 0; i < n; ++i) digest = (digest ^ b[i]) * 1099511628211ull;
    for (size_t i = 0; i < n; ++i) digest_noslot = (digest_noslot ^ b[i]) * 1099511628211ull;
  }
  bool begin_picture(const h264::PicSetup& s, h264::PicBuffers* out, std::string* error) override {
    nmb = s.mb_width * s.mb_height;
    int slot = -1;
    {
      std::lock_guard<std::mutex> lk(mu);
      if (pipelined) {
        // from where the last one was taken onwards: a surface a worker has only just let go of is not taken again at
        // once (when a worker lets go depends on timing; which surfaces the DPB holds does not)
        for (int k = 0; k < amd::kMaxSurfaces && slot < 0; ++k) {
          const int i = (slot_cursor + k) % amd::kMaxSurfaces;
          if (!slots[i].busy) slot = i;
        }
        if (slot >= 0) slot_cursor = (slot + 1) % amd::kMaxSurfaces;
      } else {
        for (int i = 0; i < amd::kMaxSurfaces; ++i)
          if (!slots[i].busy) {
            slot = i;
            break;
          }
      }
      if (slot >= 0) slots[slot].busy = true;
    }
    if (slot < 0) {
      *error = "probe: out of surfaces";
      out->no_surface = true;
      return false;
    }
    Staging& st = staging[staging_cursor];
    out->cookie = staging_cursor;
    staging_cursor = (staging_cursor + 1) % static_cast<int>(staging.size());
    if (!fast || slots[slot].motion.size() != static_cast<size_t>(nmb)) {
      slots[slot].motion.assign(nmb, amd::MbMotion());
      st.mb.assign(nmb, amd::MbInfo());
      st.coefs.assign(static_cast<size_t>(nmb) * amd::kMaxSparseCoefs, 0);
    }
    out->mb = st.mb.data();
    out->motion = slots[slot].motion.data();
    out->coefs = st.coefs.data();
    out->coef_capacity = st.coefs.size();
    out->slices = st.slices.data();
    out->max_slices = static_cast<int>(st.slices.size());
    out->slot = slot;
    if (times) {
      if (pfd[0] == -1) {
        perf_open();
        if (pfd[0] < 0) pfd[0] = -2;
      }
      perf_read(pstart);
      t_begin = std::chrono::steady_clock::now();
    }
    out->device_mv = fast;  // timing runs measure what the GPU path runs (P vectors left to k_mv); digests keep the host derivation
    if (fast) {
      if (slots[slot].pmo.size() != static_cast<size_t>(nmb)) {
        slots[slot].pmo.resize(nmb);
        slots[slot].pmx.resize(static_cast<size_t>(s.mb_height) + static_cast<size_t>(nmb) * 15);
      }
      out->pmo = slots[slot].pmo.data();
      out->pmx = slots[slot].pmx.data();
      out->pmx_capacity = slots[slot].pmx.size();
    }
    return true;
  }
  bool end_picture(const h264::PicDone& d, std::string*) override {
    if (times) {
      const int k = (d.slice_type_mask & 2) ? 2 : (d.has_inter ? 1 : 0);
      ns_kind[k] += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t_begin).count();
      ++n_kind[k];
      coef_kind[k] += static_cast<int64_t>(d.coef_count);
      uint64_t now[3];
      perf_read(now);
      for (int i = 0; i < 3; ++i) pkind[k][i] += now[i] - pstart[i];
    }
    Staging& st = staging[d.cookie];
    if (!fast)
      for (amd::MbInfo m : st.mb) {  // the derivation hints are not part of what the golden digests froze
        m.mv_flags = m.mv_shape = m.mv_sub = 0;
        mix(&m, sizeof(m));
      }
    mix(st.coefs.data(), d.coef_count * sizeof(uint32_t));
    if (d.has_inter && !fast) {
      // ref_slot depends on this sink's slot numbering, which is stable
      const std::vector<amd::MbMotion>& mo = slots[d.slot].motion;
      const uint64_t keep = digest_noslot;
      mix(mo.data(), mo.size() * sizeof(amd::MbMotion));
      digest_noslot = keep;
      for (amd::MbMotion m : mo) {
        std::memset(m.ref_slot, 0, sizeof(m.ref_slot));
        mix(&m, sizeof(m), false);
      }
    }
    if (d.has_inter) ++inter_pics;
    coef_total += static_cast<int64_t>(d.coef_count);
    return true;
  }
  void output_picture(int slot) override { outputs.push_back(slot); }
  void release_surface(int slot) override {
    std::lock_guard<std::mutex> lk(mu);
    if (slot >= 0) slots[slot].busy = false;
  }
};

}  // namespace

static int run_probe(const uint8_t* avcc, size_t avcc_size, const uint8_t* data, const uint64_t* sample_offsets,
                     const uint64_t* sample_sizes, size_t num_samples, const uint64_t* wanted, size_t num_wanted,
                     const uint64_t* keyframes, size_t num_keyframes, int workers, int64_t* out, int nout, char* error,
                     size_t error_capacity) {
  ProbeSink sink;
  std::unique_ptr<h264::WorkerPool> pool;
  const int in_flight = workers > 0 ? 2 * workers : 0;
  if (workers > 0) {
    pool.reset(new h264::WorkerPool(workers));
    sink.pipelined = true;
    sink.staging.resize(in_flight + 2);
  }
  h264::Parser parser(&sink, pool.get(), in_flight);
  auto report = [&](const std::string& e) {
    if (error && error_capacity) {
      std::strncpy(error, e.c_str(), error_capacity - 1);
      error[error_capacity - 1] = 0;
    }
    return 1;
  };
  if (!parser.configure(avcc, avcc_size)) return report(parser.error());
  if (num_wanted) parser.set_wanted(std::vector<uint64_t>(wanted, wanted + num_wanted));
  if (num_keyframes) parser.set_keyframes(std::vector<uint64_t>(keyframes, keyframes + num_keyframes));
  for (size_t i = 0; i < num_samples; ++i)
    if (!parser.decode(data + sample_offsets[i], sample_sizes[i])) return report(parser.error());
  if (!parser.flush()) return report(parser.error());
  {
    std::string worker_error;  // entropy workers: flush() does not wait for them
    if (!parser.wait_idle(&worker_error)) return report(worker_error);
  }
  if (sink.times)
    for (int k = 0; k < 3; ++k)
      if (sink.n_kind[k])
        fprintf(stderr, "probe: %lld %s pictures, %.3f ms and %.0f coefficient words each; %.0f instructions, %.0f cycles, %.0f branch misses each\n",
                (long long)sink.n_kind[k], k == 2 ? "B" : (k ? "inter" : "intra-only"), 1e-6 * sink.ns_kind[k] / sink.n_kind[k],
                1.0 * sink.coef_kind[k] / sink.n_kind[k], 1.0 * sink.pkind[k][0] / sink.n_kind[k], 1.0 * sink.pkind[k][1] / sink.n_kind[k],
                1.0 * sink.pkind[k][2] / sink.n_kind[k]);
  int64_t skipped_out = 0;
  for (int s : sink.outputs) skipped_out += s < 0;
  out[0] = parser.slices();
  out[1] = parser.slices_clean_end();
  out[2] = parser.pictures_decoded();
  out[3] = parser.pictures_skipped();
  out[4] = static_cast<int64_t>(sink.outputs.size());
  out[5] = skipped_out;
  out[6] = sink.coef_total;
  out[7] = static_cast<int64_t>(sink.digest & 0x7fffffffffffffffull);
  if (nout > 8) out[8] = static_cast<int64_t>(sink.digest_noslot & 0x7fffffffffffffffull);
  return 0;
}

extern "C" int hwang_h264_probe(const uint8_t* avcc, size_t avcc_size, const uint8_t* data,
                                const uint64_t* sample_offsets, const uint64_t* sample_sizes,
                                size_t num_samples, const uint64_t* wanted, size_t num_wanted,
                                int64_t out[8], char* error, size_t error_capacity) {
  return run_probe(avcc, avcc_size, data, sample_offsets, sample_sizes, num_samples, wanted, num_wanted, nullptr, 0, 0, out,
                   8, error, error_capacity);
}

extern "C" int hwang_h264_probe_workers(const uint8_t* avcc, size_t avcc_size, const uint8_t* data,
                                        const uint64_t* sample_offsets, const uint64_t* sample_sizes,
                                        size_t num_samples, const uint64_t* wanted, size_t num_wanted,
                                        const uint64_t* keyframes, size_t num_keyframes, int workers, int64_t out[9],
                                        char* error, size_t error_capacity) {
  return run_probe(avcc, avcc_size, data, sample_offsets, sample_sizes, num_samples, wanted, num_wanted, keyframes,
                   num_keyframes, workers, out, 9, error, error_capacity);
}

// Host only: what an hvcC box payload (the metadata MP4IndexCreator returns for an hev1 / hvc1 track) says about its
// track; 0 and the description, or 1 and what is wrong with the record.
extern "C" int hwang_hevc_describe_config(const uint8_t* hvcc, size_t size, char* out, size_t capacity) {
  hevc::Config c;
  std::string text;
  const bool ok = hevc::parse_hvcc(hvcc, size, &c, &text);
  if (ok) text = hevc::describe(c);
  if (out && capacity) {
    const size_t n = std::min(text.size(), capacity - 1);
    memcpy(out, text.data(), n);
    out[n] = 0;
  }
  return ok ? 0 : 1;
}
