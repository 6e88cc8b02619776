#include "retriever.h"

#include <linux/perf_event.h>
#include <sys/syscall.h>
#include <time.h>
#include <unistd.h>

#include "hevc/hvcc.h"
#include "util/avc_sample.h"
#include "util/cpus.h"
#include "util/numa.h"

#include <algorithm>
#include <chrono>
#include <pthread.h>
#include <cstdio>
#include <cstdlib>
#include <deque>
#include <thread>

namespace hwang {

int32_t ParallelRetriever::effective_cpus() { return usable_cpus(); }

ParallelRetriever* ParallelRetriever::make(const std::vector<DeviceHandle>& devices,
                                           VideoDecoderType type,
                                           int32_t num_lanes) {
  if (devices.empty()) return nullptr;
  // A lane waits for the GPU about as long as it parses (a picture's latency
  // through the batching pipeline is comparable to its parse time), so it takes
  // a few lanes per CPU to keep the CPUs - the bottleneck - busy; measured on a
  // 16-CPU quota: 32 / 40 / 48 / 64 lanes -> 660 / 676 / 698 / 651 frames/s.
  // The CPUs are the process's, however many GPUs it drives: more devices do
  // not mean more lanes, only fewer lanes per device.
  const bool auto_lanes = num_lanes <= 0;
  if (auto_lanes) num_lanes = std::min<int32_t>(3 * effective_cpus(), 128);
  num_lanes = std::max<int32_t>(num_lanes, auto_lanes ? static_cast<int32_t>(devices.size()) : 1);
  std::unique_ptr<ParallelRetriever> r(new ParallelRetriever());
  r->auto_lanes_ = auto_lanes;
  r->devices_ = devices;
  r->device_stats_.resize(devices.size());
  r->lanes_.resize(num_lanes);
  for (int32_t li = 0; li < num_lanes; ++li) {
    Lane& lane = r->lanes_[li];
    lane.device_slot = li % static_cast<int32_t>(devices.size());
    for (auto& b : lane.backend) {
      // num_devices = 0: "the caller is the parallelism" - a lane decodes on its own thread and its pictures travel in
      // the executor's wide batches.  (1 is what the reference's Python passes for ONE decoder that has the machine to
      // itself, decoder.py:28: there the AMD backend brings its own entropy workers; n > 1: that many.)
      b.reset(VideoDecoderFactory::make_from_config(devices[lane.device_slot], 0, type));
      if (!b) return nullptr;
    }
  }
  return r.release();
}

Result ParallelRetriever::set_shard(int32_t rank, int32_t world) {
  if (world < 1 || rank < 0 || rank >= world) return Result(false, "set_shard: need 0 <= rank < world");
  shard_rank_ = rank;
  shard_world_ = world;
  return Result();
}

ParallelRetriever::~ParallelRetriever() {}

// One work item on its way through a backend, a sample at a time (ItemRun::step), so that a thread can drive several
// items - on different backends - in turn: while one decoder's pictures are on the GPU the thread parses for another.
// The protocol is the reference automata's (decoder_automata.cpp:287-404): feed every sample up to the last wanted
// frame, fetch wanted frames and discard the others as they appear, (nullptr, 0) + flush at the end.
struct ParallelRetriever::ItemRun {
  const WorkItem* item = nullptr;
  VideoDecoderInterface* dec = nullptr;
  PlanarOutput* planar = nullptr;
  const VideoIndex* index = nullptr;
  const uint8_t* file = nullptr;
  size_t file_size = 0;
  uint8_t* out = nullptr;
  size_t frame_bytes = 0;
  bool yuv = false;
  uint64_t s = 0, next_out = 0;  // next sample to feed; display ordinal of the next output frame
  size_t got = 0, kf_cursor = 0;

  Result start(VideoDecoderInterface* d, const WorkItem& it, const VideoIndex& idx, const uint8_t* f, size_t fsize, uint8_t* o,
               size_t fbytes, bool want_yuv) {
    item = &it;
    dec = d;
    index = &idx;
    file = f;
    file_size = fsize;
    out = o;
    frame_bytes = fbytes;
    yuv = want_yuv;
    auto* hints = dynamic_cast<SparseDecodeHints*>(dec);
    planar = dynamic_cast<PlanarOutput*>(dec);
    if (yuv && !planar) return Result(false, "retrieve: backend has no planar (I420) output");
    const std::vector<uint64_t>& valid = *it.valid;
    if (hints) {
      std::vector<uint64_t> ordinals(valid.size());
      for (size_t i = 0; i < valid.size(); ++i) ordinals[i] = valid[i] - it.start;
      hints->set_wanted_frames(ordinals);
    }
    const auto& kfs = idx.keyframe_indices();
    kf_cursor = std::lower_bound(kfs.begin(), kfs.end(), it.start) - kfs.begin();
    s = next_out = it.start;
    got = 0;
    return Result();
  }
  Result consume(Lane& lane) {
    const std::vector<uint64_t>& valid = *item->valid;
    while (got < valid.size() && dec->decoded_frames_buffered() > 0) {
      if (next_out == valid[got]) {
        uint8_t* dst = out + (item->out_slot + got) * frame_bytes;
        HWANG_RETURN_ON_ERROR(yuv ? planar->get_frame_yuv(dst, frame_bytes) : dec->get_frame(dst, frame_bytes));
        ++got;
        ++lane.frames_returned;
      } else {
        HWANG_RETURN_ON_ERROR(dec->discard_frame());
      }
      ++next_out;
    }
    return Result();
  }
  // Feeds one sample, or - when there is none left to feed - ends the interval (*done).
  Result step(Lane& lane, bool* done) {
    const std::vector<uint64_t>& valid = *item->valid;
    *done = false;
    if (s < item->end && got < valid.size()) {
      const auto& offs = index->sample_offsets();
      const auto& sizes = index->sample_sizes();
      const auto& kfs = index->keyframe_indices();
      if (offs[s] > file_size || sizes[s] > file_size - offs[s])
        return Result(false, "retrieve: sample " + std::to_string(s) + " lies outside the file");
      bool keyframe = false;
      if (kf_cursor < kfs.size() && kfs[kf_cursor] == s) {
        keyframe = true;
        ++kf_cursor;
      }
      HWANG_RETURN_ON_ERROR(dec->feed(file + offs[s], sizes[s], keyframe));
      ++lane.samples_fed;
      ++s;
      return consume(lane);
    }
    // End of interval: empty packet + flush pushes out whatever the reorder
    // buffer still holds (decoder_automata.cpp:383-389 protocol).
    *done = true;
    HWANG_RETURN_ON_ERROR(dec->feed(nullptr, 0, false));
    HWANG_RETURN_ON_ERROR(dec->flush());
    HWANG_RETURN_ON_ERROR(consume(lane));
    while (dec->decoded_frames_buffered() > 0) HWANG_RETURN_ON_ERROR(dec->discard_frame());
    if (got != valid.size())
      return Result(false, "retrieve: interval [" + std::to_string(item->start) + "," + std::to_string(item->end) + ") produced " +
                               std::to_string(got) + " of " + std::to_string(valid.size()) + " wanted frames");
    return Result();
  }
};

Result ParallelRetriever::retrieve(const VideoIndex& index, const uint8_t* file,
                                   size_t file_size,
                                   const std::vector<uint64_t>& rows,
                                   uint8_t* out, size_t out_capacity, bool yuv) {
  Request rq;
  rq.index = &index;
  rq.file = file;
  rq.file_size = file_size;
  rq.rows = rows.data();
  rq.num_rows = rows.size();
  rq.out = out;
  rq.out_capacity = out_capacity;
  return retrieve_batch(std::vector<Request>{rq}, yuv);
}

namespace {
size_t frame_bytes_of(const VideoIndex& index, bool yuv) {
  const size_t w = index.frame_width(), h = index.frame_height();
  return yuv ? w * h + 2 * ((w + 1) / 2) * ((h + 1) / 2) : w * h * 3;
}
// A thread's own CPU time and - on request, where the kernel lets a process count its events - the user-mode
// instructions, cycles and mispredicted branches it retired (bench.py: host_lane_threads).
struct ThreadCounters {
  int fd[3] = {-1, -1, -1};
  int64_t cpu0 = 0;
  static int64_t thread_cpu_ns() {
    timespec ts;
    clock_gettime(CLOCK_THREAD_CPUTIME_ID, &ts);
    return static_cast<int64_t>(ts.tv_sec) * 1000000000 + ts.tv_nsec;
  }
  explicit ThreadCounters(bool hardware) {
    cpu0 = thread_cpu_ns();
    if (!hardware) return;
    const uint64_t cfg[3] = {PERF_COUNT_HW_INSTRUCTIONS, PERF_COUNT_HW_CPU_CYCLES, PERF_COUNT_HW_BRANCH_MISSES};
    for (int i = 0; i < 3; ++i) {
      perf_event_attr a;
      std::memset(&a, 0, sizeof(a));
      a.size = sizeof(a);
      a.type = PERF_TYPE_HARDWARE;
      a.config = cfg[i];
      a.exclude_kernel = 1;
      a.exclude_hv = 1;
      fd[i] = static_cast<int>(syscall(SYS_perf_event_open, &a, 0, -1, -1, 0));
    }
  }
  void stop(int64_t* cpu_ns, int64_t* instructions, int64_t* cycles, int64_t* branch_misses) {
    *cpu_ns = thread_cpu_ns() - cpu0;
    int64_t* out[3] = {instructions, cycles, branch_misses};
    for (int i = 0; i < 3; ++i) {
      uint64_t v = 0;
      if (fd[i] >= 0 && read(fd[i], &v, 8) == 8) *out[i] = static_cast<int64_t>(v);
      if (fd[i] >= 0) close(fd[i]);
      fd[i] = -1;
    }
  }
  ~ThreadCounters() {
    for (int f : fd)
      if (f >= 0) close(f);
  }
};

// (width, height, format, avcC) as one comparable blob
std::vector<uint8_t> config_key_of(const VideoIndex& index) {
  std::vector<uint8_t> k;
  const uint32_t wh[2] = {index.frame_width(), index.frame_height()};
  k.insert(k.end(), reinterpret_cast<const uint8_t*>(wh), reinterpret_cast<const uint8_t*>(wh) + sizeof(wh));
  k.insert(k.end(), index.format().begin(), index.format().end());
  k.push_back(0);
  k.insert(k.end(), index.metadata_bytes().begin(), index.metadata_bytes().end());
  return k;
}
}  // namespace

Result ParallelRetriever::retrieve_batch(const std::vector<Request>& requests, bool yuv) {
  const auto t0 = std::chrono::steady_clock::now();
  stats_ = Stats();
  for (DeviceStats& ds : device_stats_) ds = DeviceStats();
  for (const Lane& lane : lanes_) ++device_stats_[lane.device_slot].lanes;

  // Work items.  The planner merges byte-adjacent GOPs into one interval
  // (one seek for the reference's serial loop); every GOP still starts at a
  // sync sample, so for parallel decoding each GOP that holds a wanted row is
  // its own item, fed only up to its last wanted frame.
  std::vector<WorkItem> items;
  std::deque<std::vector<uint64_t>> item_rows;  // stable addresses while it grows
  std::vector<std::vector<uint8_t>> keys(requests.size());
  std::vector<size_t> frame_bytes(requests.size());
  int64_t largest_picture = 0;
  for (size_t ri = 0; ri < requests.size(); ++ri) {
    const Request& rq = requests[ri];
    if (!rq.index || (rq.num_rows && (!rq.rows || !rq.out || !rq.file)))
      return Result(false, "retrieve: request " + std::to_string(ri) + " has a null argument");
    const VideoIndex& index = *rq.index;
    frame_bytes[ri] = frame_bytes_of(index, yuv);
    if (rq.out_capacity < frame_bytes[ri] * rq.num_rows) return Result(false, "retrieve: output buffer too small");
    if (!rq.num_rows) continue;
    if (!index.consistent()) return Result(false, "retrieve: inconsistent index: " + index.problem());
    const std::vector<uint64_t> rows(rq.rows, rq.rows + rq.num_rows);
    if (!std::is_sorted(rows.begin(), rows.end())) return Result(false, "retrieve: rows must be sorted ascending");
    bool ok = true;
    VideoIntervals plan = slice_into_video_intervals(index, rows, &ok);
    if (!ok) return Result(false, "retrieve: row out of range");
    keys[ri] = config_key_of(index);
    largest_picture = std::max<int64_t>(largest_picture, static_cast<int64_t>(index.frame_width()) * index.frame_height());

    size_t slot = 0;
    const auto& sizes = index.sample_sizes();
    const auto& offsets = index.sample_offsets();
    const auto& kfs = index.keyframe_indices();
    // A GOP is an item of its own only if it is closed, i.e. if the sync sample behind it is an IDR picture; a
    // recovery-point I picture (open GOP) stays in the item of the GOP before it.
    const SampleFraming framing = sample_framing(hevc::is_hevc_format(index.format()), index.metadata_bytes());
    auto starts_closed_gop = [&](uint64_t sample) {
      if (framing.codec == 0) return true;  // no avcC / hvcC (another codec, a test backend): the sync table is all there is
      return offsets[sample] <= rq.file_size && sizes[sample] <= rq.file_size - offsets[sample] &&
             sample_starts_closed_gop(rq.file + offsets[sample], static_cast<size_t>(sizes[sample]), framing);
    };
    auto hevc_kind = [&](uint64_t sample) {
      return sample < offsets.size() && offsets[sample] <= rq.file_size && sizes[sample] <= rq.file_size - offsets[sample]
                 ? hevc_sample_kind(rq.file + offsets[sample], static_cast<size_t>(sizes[sample]), framing)
                 : 0;
    };
    for (size_t i = 0; i < plan.sample_index_intervals.size(); ++i) {
      const std::vector<uint64_t>& valid = plan.valid_frames[i];
      if (valid.empty()) continue;
      const uint64_t istart = std::get<0>(plan.sample_index_intervals[i]);
      const uint64_t iend = std::get<1>(plan.sample_index_intervals[i]);
      // ... and an HEVC GOP that starts with a CRA picture (x265's default) when none of the CRA picture's LEADING pictures
      // is wanted: they are the samples right behind it, displayed in front of it at the positions sample .. sample +
      // L - 1, and the only pictures of the GOP that refer to anything before it.  (Entered at the CRA picture they would
      // be decoded from made-up references, hevc_parser.h: missing_surface - a wanted one keeps its GOP in the item before.)
      auto may_start_item = [&](uint64_t sample) {
        if (starts_closed_gop(sample)) return true;
        if (framing.codec != 2 || hevc_kind(sample) != 2) return false;
        uint64_t leading = 0;
        while (sample + 1 + leading < iend && hevc_kind(sample + 1 + leading) == 3) ++leading;
        const auto lo = std::lower_bound(valid.begin(), valid.end(), sample);
        return lo == valid.end() || *lo >= sample + leading;
      };
      size_t k = std::upper_bound(kfs.begin(), kfs.end(), istart) - kfs.begin();  // first keyframe after istart
      uint64_t gop_start = istart;
      // The planner may start an interval AT a CRA picture (the GOP before it holds nothing wanted).  If one of that
      // picture's leading pictures is wanted, decoding starts a sync sample earlier: the leading picture has its
      // references then and is the sequential decode's frame - entered at the CRA picture it would be predicted from
      // made-up references (which is what DecoderAutomata's callers get there, as the reference's do worse: libavcodec
      // drops those frames).  The extra GOP is decoded without a wanted row in it, i.e. reference pictures only.
      if (framing.codec == 2 && k >= 2 && kfs[k - 1] == istart && !may_start_item(istart)) gop_start = kfs[k - 2];
      size_t v = 0;
      while (v < valid.size()) {
        while (k < kfs.size() && kfs[k] < iend && !may_start_item(kfs[k])) ++k;
        const uint64_t gop_end = (k < kfs.size() && kfs[k] < iend) ? kfs[k] : iend;
        size_t v2 = v;
        while (v2 < valid.size() && valid[v2] < gop_end) ++v2;
        if (v2 > v) {
          item_rows.emplace_back(valid.begin() + v, valid.begin() + v2);
          WorkItem it;
          it.request = static_cast<uint32_t>(ri);
          it.start = gop_start;
          it.end = gop_end;
          it.valid = &item_rows.back();
          it.out_slot = slot;
          slot += v2 - v;
          const uint64_t last = std::min<uint64_t>(it.end, item_rows.back().back() + 4);
          for (uint64_t s2 = it.start; s2 < last; ++s2) it.cost += sizes[s2];
          items.push_back(it);
        }
        v = v2;
        gop_start = gop_end;
        ++k;
        if (gop_end >= iend) break;
      }
    }
  }

  // Longest first.  The order is a pure function of the requests (ties broken
  // by request and position), so every rank of a sharded job computes the same.
  std::vector<size_t> order(items.size());
  for (size_t i = 0; i < order.size(); ++i) order[i] = i;
  std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) {
    return items[a].cost > items[b].cost;
  });
  if (shard_world_ > 1) {
    // LPT split: each item, longest first, to the share with the least work so far.
    std::vector<uint64_t> load(shard_world_, 0);
    std::vector<size_t> mine;
    for (size_t k : order) {
      const int32_t to = static_cast<int32_t>(std::min_element(load.begin(), load.end()) - load.begin());
      load[to] += items[k].cost + 1;
      if (to == shard_rank_) mine.push_back(k);
    }
    order.swap(mine);
  }
  stats_.intervals = static_cast<int64_t>(order.size());
  // Order of the shared work queue.  Longest-first (LPT) balances the tail,
  // but sorted items keep all lanes in lock-step: every lane reaches its next
  // keyframe at the same moment, the GPU sees a burst of intra-only pictures
  // followed by a burst of P pictures, and the two engines take turns instead
  // of overlapping.  So only the shortest quarter stays sorted (at the end,
  // where it evens out the finish); everything before it is shuffled with a
  // fixed seed.
  if (order.size() > 2 * lanes_.size()) {
    uint64_t state = 0x9e3779b97f4a7c15ull;
    const size_t head = order.size() - order.size() / 4;
    for (size_t i = head - 1; i > 0; --i) {
      state = state * 6364136223846793005ull + 1442695040888963407ull;
      std::swap(order[i], order[(state >> 33) % (i + 1)]);
    }
  }

  std::atomic<bool> abort{false};
  const bool host_counters = getenv("HWANG_HOST_COUNTERS") != nullptr;
  size_t nl = std::min(lanes_.size(), order.size());
  // A lane's working set grows with the picture (records of a 4K picture:
  // ~13 MB); beyond two lanes per CPU the parsers evict each other (measured
  // at 3840x2160 on 16 CPUs: 16 / 24 / 32 / 48 lanes -> 485 / 528 / 562 / 477
  // frames/s, 7.4 -> 8.3 ms of CPU per picture).
  if (auto_lanes_ && largest_picture > 2 * 1920 * 1088)
    nl = std::min<size_t>(nl, std::max<size_t>(devices_.size(), static_cast<size_t>(2 * effective_cpus())));
  // with fewer items than lanes, still spread them over every device
  if (nl < lanes_.size() && devices_.size() > 1) nl = std::min(lanes_.size(), std::max(nl, std::min(order.size(), devices_.size())));
  // Lane li starts with item li (so every lane, hence every device, has work from the first moment
  // whatever order the threads come up in); everything after that is pulled from the shared queue.
  std::atomic<size_t> cursor{nl};
  std::vector<std::thread> threads;
  for (size_t li = 0; li < nl; ++li) {
    Lane& lane = lanes_[li];
    lane.result = Result();
    lane.samples_fed = 0;
    lane.frames_returned = 0;
    lane.items = 0;
    lane.cpu_ns = lane.instructions = lane.cycles = lane.branch_misses = 0;
  }
  // A thread drives several lanes in turn, a sample of each at a time (HWANG_LANES_PER_THREAD overrides how many): the lanes -
  // the decode streams in flight, hence the width of the GPU's launches - stay what they were with half the threads.
  // A lane's thread waits for the GPU about as long as it parses; with a thread per lane that took three threads per
  // CPU to keep the CPUs busy, and the parser pays for being descheduled (counted on the box, tools/hostprof/
  // lib_counters_mt.py: 6.09 M cycles per P picture with a thread per CPU, 6.33 M with three).
  // Default: a thread per usable CPU (48 lanes on 16 CPUs: three lanes a thread) - when the queue is long.  A thread
  // that drives several items at once cannot give one away, so with few items per lane the end of the call is a few
  // threads finishing their two or three long items while the other CPUs idle (measured: one 4K60 stream of 20 GOPs,
  // 497 -> 315 frames/s; 4K every 10th frame, 60 items on 32 lanes, 638 -> 572): then a thread per lane, as in rounds
  // 1-4, and the operating system does the balancing.  The same threshold tells the backends whether they may trade a
  // picture's latency for wider launches (BatchingHints).
  size_t long_from = 8;  // work items per lane (HWANG_LONG_QUEUE_FROM: the tests set 1 to run small clips this way)
  if (const char* e = getenv("HWANG_LONG_QUEUE_FROM")) long_from = static_cast<size_t>(std::max(1, atoi(e)));
  const bool long_queue = order.size() >= long_from * nl;
  for (size_t li = 0; li < nl; ++li)
    for (auto& b : lanes_[li].backend)
      if (auto* h = dynamic_cast<BatchingHints*>(b.get())) h->set_long_queue(long_queue);
  size_t nthreads = long_queue ? std::min<size_t>(nl, static_cast<size_t>(std::max(1, effective_cpus()))) : nl;
  // (several processes sharing one host's CPUs - a rank per GPU - say how many threads are theirs: bench.py divides the
  // usable CPUs by the local world size)
  if (const char* e = getenv("HWANG_RETRIEVER_THREADS"))
    if (long_queue) nthreads = std::min<size_t>(nl, static_cast<size_t>(std::max(1, atoi(e))));
  if (const char* e = getenv("HWANG_LANES_PER_THREAD")) {
    const size_t share = static_cast<size_t>(std::max(1, std::min(atoi(e), 8)));
    nthreads = (nl + share - 1) / share;
  }
  // (lanes t, t + T, t + 2 T, ... of a thread: with several devices they are all lanes of ONE device when T is a
  // multiple of the device count, which the rounding below sees to)
  if (devices_.size() > 1) nthreads = std::min(nl, (nthreads + devices_.size() - 1) / devices_.size() * devices_.size());
  stats_.lane_threads = static_cast<int64_t>(nthreads);
  for (size_t ti = 0; ti < nthreads; ++ti) {
    threads.emplace_back([&, ti]() {
      pthread_setname_np(pthread_self(), "hw-lane");
      std::vector<size_t> mine;
      for (size_t li = ti; li < nl; li += nthreads) mine.push_back(li);
      // several GPUs on one host (one process driving them, or one rank per GPU): a lane's thread - and the pinned
      // staging buffers it is the first to ask for - stay on the socket its GPU hangs off (HWANG_NUMA_AFFINITY=0/1)
      if (numa::enabled(devices_.size() > 1 || shard_world_ > 1))
        numa::bind_current_thread_to_node(VideoDecoderFactory::device_numa_node(devices_[lanes_[mine[0]].device_slot]));
      ThreadCounters counters(host_counters);
      struct Slot {
        size_t li;
        ItemRun run;
        bool active = false, first = true, finished = false;
        int turn = 0;
        BatchingHints* hints = nullptr;  // of the backend the item in hand runs on
        Result r;
      };
      std::vector<Slot> slots(mine.size());
      for (size_t k = 0; k < mine.size(); ++k) {
        slots[k].li = mine[k];
        Lane& lane = lanes_[mine[k]];
        for (auto& b : lane.backend)
          if (auto* c = dynamic_cast<CountedBackend*>(b.get())) c->reset_counters();
        // configure() once per call and clip, not once per interval (the reference re-opens its codec
        // for every interval, decoder.py:65): a backend is re-configured only when the clip changes
        for (auto& key : lane.config_key) key.clear();
      }
      bool failed = false;
      size_t live = slots.size();
      size_t passed_over = 0;  // items in a row whose decoder would have started by waiting for the GPU
      while (live > 0 && !failed && !abort.load()) {
        for (Slot& sl : slots) {
          if (sl.finished) continue;
          Lane& lane = lanes_[sl.li];
          if (sl.active && live > 1 && passed_over + 1 < live && sl.hints && sl.hints->feed_would_wait()) {
            // another item of this thread first; when all of them would wait, the next one simply does
            ++passed_over;
            continue;
          }
          passed_over = 0;
          if (!sl.active) {
            // the lane's next item: lane li starts with item li (so every lane, hence every device, has work from the
            // first moment whatever order the threads come up in), everything after that comes from the shared queue
            const size_t k = sl.first ? sl.li : cursor.fetch_add(1);
            sl.first = false;
            if (k >= order.size()) {
              sl.finished = true;
              --live;
              continue;
            }
            const WorkItem& item = items[order[k]];
            const Request& rq = requests[item.request];
            if (lane.config_key[sl.turn] != keys[item.request]) {
              VideoDecoderInterface::FrameInfo info;
              info.width = rq.index->frame_width();
              info.height = rq.index->frame_height();
              info.format = rq.index->format();
              sl.r = lane.backend[sl.turn]->configure(info, rq.index->metadata_bytes());
              if (sl.r.ok) lane.config_key[sl.turn] = keys[item.request];
            }
            if (sl.r.ok)
              sl.r = sl.run.start(lane.backend[sl.turn].get(), item, *rq.index, rq.file, rq.file_size, rq.out, frame_bytes[item.request], yuv);
            sl.active = sl.r.ok;
            sl.hints = dynamic_cast<BatchingHints*>(lane.backend[sl.turn].get());
          }
          if (sl.active) {
            bool done = false;
            sl.r = sl.run.step(lane, &done);
            if (done || !sl.r.ok) {
              sl.active = false;
              ++lane.items;
              // (each lane alternates between its two backends from item to item: the next item's keyframe is on its
              // way while the previous one's last pictures finish on the other backend's executor lane)
              sl.turn = (sl.turn + 1) % kBackendsPerLane;
            }
          }
          if (!sl.r.ok) {
            failed = true;
            break;
          }
        }
      }
      int64_t cpu_ns = 0, instructions = 0, cycles = 0, branch_misses = 0;
      for (Slot& sl : slots) {
        Lane& lane = lanes_[sl.li];
        Result r = sl.r;
        // Every backend of every lane of this thread is drained whatever happened - after an error too (the failing
        // lane's or another thread's): conversions and copies into the caller's output may still be in flight, and the
        // caller is free to let go of that memory the moment retrieve() returns.  Only the first error is reported.
        for (auto& b : lane.backend) {
          const Result w = b->wait_until_frames_copied();
          if (r.ok && !failed && !w.ok) r = w;
        }
        if (!r.ok) abort.store(true);
        lane.result = r;
      }
      counters.stop(&cpu_ns, &instructions, &cycles, &branch_misses);
      Lane& first_lane = lanes_[mine[0]];  // (the thread's counters are booked on its first lane)
      first_lane.cpu_ns = cpu_ns;
      first_lane.instructions = instructions;
      first_lane.cycles = cycles;
      first_lane.branch_misses = branch_misses;
    });
  }
  for (auto& t : threads) t.join();

  Result result;
  for (size_t li = 0; li < nl; ++li) {
    Lane& lane = lanes_[li];
    if (!lane.result.ok && result.ok) result = lane.result;
    stats_.samples_fed += lane.samples_fed;
    stats_.frames_returned += lane.frames_returned;
    stats_.lane_cpu_ns += lane.cpu_ns;
    stats_.lane_instructions += lane.instructions;
    stats_.lane_cycles += lane.cycles;
    stats_.lane_branch_misses += lane.branch_misses;
    DeviceStats& ds = device_stats_[lane.device_slot];
    ds.items += lane.items;
    ds.frames_returned += lane.frames_returned;
    ds.samples_fed += lane.samples_fed;
    for (auto& b : lane.backend) {
      if (auto* c = dynamic_cast<CountedBackend*>(b.get())) {
        BackendCounters bc = c->counters();
        stats_.pictures_reconstructed += bc.pictures_reconstructed;
        stats_.pictures_skipped += bc.pictures_skipped;
        stats_.bytes_uploaded += bc.bytes_uploaded;
        stats_.entropy_ns += bc.entropy_ns;
        stats_.stall_ns += bc.stall_ns;
      }
    }
  }
  stats_.wall_ns = std::chrono::duration_cast<std::chrono::nanoseconds>(
                       std::chrono::steady_clock::now() - t0)
                       .count();
  return result;
}

}  // namespace hwang
