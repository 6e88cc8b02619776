// Sanitizer harness for the host half (no HIP): built by tests/test_sanitizers.py
// with -fsanitize=thread (retriever lanes, DecoderAutomata's feeder / caller
// threads, Queue<T>) and with -fsanitize=address,undefined (MP4 indexer, seek
// planner, proto codec and the H.264 parser on valid and on corrupted input).
// The GPU backend is replaced by a stub that is never selected.
#include "decoder_automata.h"
#include "h264/h264_parser.h"
#include "mp4_index_creator.h"
#include "retriever.h"
#include "../../include/hwang_amd.h"
#include "video_decoder_factory.h"
#include "video_index.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iterator>
#include <thread>
#include <vector>

namespace hwang {
VideoDecoderInterface* make_amd_video_decoder(int32_t, uint32_t) { return nullptr; }
int amd_device_numa_node(int32_t) { return -1; }
}
using namespace hwang;
extern "C" const hwang_backend_vtable hwang_stamp_vtable;  // tests/fixtures/stamp_backend.c
static const VideoDecoderType STAMP = static_cast<VideoDecoderType>(100);

static int fails = 0;
#define CHECK(c) do { if (!(c)) { std::fprintf(stderr, "CHECK failed %s:%d: %s\n", __FILE__, __LINE__, #c); ++fails; } } while (0)

static VideoIndex stamp_index(int n, int gop, std::vector<uint8_t>* blob) {
  std::vector<uint64_t> offs, sizes, kfs;
  for (int i = 0; i < n; ++i) {
    offs.push_back(blob->size());
    uint32_t v = static_cast<uint32_t>(i);
    const uint8_t* p = reinterpret_cast<const uint8_t*>(&v);
    blob->insert(blob->end(), p, p + 4);
    for (int k = 0; k < 3 + i % 5; ++k) blob->push_back('s');
    sizes.push_back(blob->size() - offs.back());
    if (i % gop == 0) kfs.push_back(i);
  }
  return VideoIndex(30000, n * 1001ull, 4, 2, "avc1", offs, sizes, kfs, {});
}

static void threads_part() {
  CHECK(VideoDecoderFactory::register_backend(100, &hwang_stamp_vtable));
  std::vector<uint8_t> blob;
  VideoIndex idx = stamp_index(1200, 30, &blob);
  std::vector<uint64_t> rows;
  for (uint64_t r = 3; r < 1200; r += 17) rows.push_back(r);
  // several devices, more lanes than CPUs, batch of two requests
  std::unique_ptr<ParallelRetriever> pr(ParallelRetriever::make(
      {DeviceHandle{DeviceType::CPU, 0}, DeviceHandle{DeviceType::CPU, 1}, DeviceHandle{DeviceType::CPU, 2}}, STAMP, 12));
  CHECK(pr != nullptr);
  std::vector<uint8_t> out1(rows.size() * 24), out2(rows.size() * 24);
  for (int rep = 0; rep < 3; ++rep) {
    ParallelRetriever::Request a, b;
    a.index = b.index = &idx;
    a.file = b.file = blob.data();
    a.file_size = b.file_size = blob.size();
    a.rows = b.rows = rows.data();
    a.num_rows = b.num_rows = rows.size();
    a.out = out1.data();
    b.out = out2.data();
    a.out_capacity = b.out_capacity = out1.size();
    Result r = pr->retrieve_batch({a, b}, false);
    CHECK(r.ok);
    for (size_t i = 0; i < rows.size(); ++i) {
      uint32_t v1, v2;
      std::memcpy(&v1, &out1[i * 24], 4);
      std::memcpy(&v2, &out2[i * 24], 4);
      CHECK(v1 == rows[i] && v2 == rows[i]);
    }
  }
  // DecoderAutomata: feeder thread vs caller thread
  std::unique_ptr<DecoderAutomata> da(DecoderAutomata::make_instance(DeviceHandle{DeviceType::CPU, 0}, 1, STAMP));
  CHECK(da != nullptr);
  for (int rep = 0; rep < 4; ++rep) {
    std::vector<DecoderAutomata::EncodedData> segs;
    std::vector<uint64_t> want;
    for (int g = rep; g < 30; g += 7) {
      DecoderAutomata::EncodedData d;
      const uint64_t s = g * 30, e = s + 30;
      d.encoded_video.assign(blob.begin() + idx.sample_offsets()[s], blob.begin() + idx.sample_offsets()[e - 1] + idx.sample_sizes()[e - 1]);
      d.width = 4; d.height = 2; d.format = "avc1";
      d.start_keyframe = s; d.end_keyframe = e;
      for (uint64_t k = s; k < e; ++k) { d.sample_offsets.push_back(idx.sample_offsets()[k] - idx.sample_offsets()[s]); d.sample_sizes.push_back(idx.sample_sizes()[k]); }
      d.keyframes = {s, e};
      d.valid_frames = {s + 1, s + 7, s + 29};
      for (uint64_t v : d.valid_frames) want.push_back(v);
      segs.push_back(d);
    }
    CHECK(da->initialize(segs, {}).ok);
    std::vector<uint8_t> frames(want.size() * 24);
    size_t got = 0;
    while (got < want.size()) {
      const int n = static_cast<int>(std::min<size_t>(4, want.size() - got));
      CHECK(da->get_frames(frames.data() + got * 24, n).ok);
      got += n;
    }
    for (size_t i = 0; i < want.size(); ++i) {
      uint32_t v;
      std::memcpy(&v, &frames[i * 24], 4);
      CHECK(v == want[i]);
    }
  }
}

struct NullSink : h264::PictureSink {
  std::vector<amd::MbInfo> mb;
  std::vector<amd::MbMotion> mo[amd::kMaxSurfaces];
  std::vector<amd::PMotion> pmo[amd::kMaxSurfaces];
  std::vector<uint32_t> pmx[amd::kMaxSurfaces];
  bool busy[amd::kMaxSurfaces] = {false};
  std::vector<uint32_t> coefs;
  std::vector<amd::SliceParams> sl = std::vector<amd::SliceParams>(64);
  bool compact = false;
  bool begin_picture(const h264::PicSetup& s, h264::PicBuffers* out, std::string* error) override {
    const size_t nmb = static_cast<size_t>(s.mb_width) * s.mb_height;
    int slot = -1;
    for (int i = 0; i < amd::kMaxSurfaces; ++i)
      if (!busy[i]) { slot = i; break; }
    if (slot < 0) { *error = "out of surfaces"; return false; }
    busy[slot] = true;
    mb.assign(nmb, amd::MbInfo());
    mo[slot].assign(nmb, amd::MbMotion());
    coefs.assign(nmb * amd::kMaxSparseCoefs, 0);
    out->mb = mb.data(); out->motion = mo[slot].data(); out->coefs = coefs.data(); out->coef_capacity = coefs.size();
    out->slices = sl.data(); out->max_slices = 64; out->slot = slot;
    out->device_mv = compact;
    if (compact) {
      pmo[slot].assign(nmb, amd::PMotion());
      pmx[slot].assign(s.mb_height + nmb * 15, 0);
      out->pmo = pmo[slot].data(); out->pmx = pmx[slot].data(); out->pmx_capacity = pmx[slot].size();
    }
    return true;
  }
  bool end_picture(const h264::PicDone&, std::string*) override { return true; }
  void output_picture(int) override {}
  void release_surface(int s) override { if (s >= 0) busy[s] = false; }
};

static std::vector<uint8_t> slurp(const char* p) {
  std::ifstream f(p, std::ios::binary);
  return std::vector<uint8_t>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}

static void memory_part(int argc, char** argv) {
  uint64_t rng = 0x9e3779b97f4a7c15ull;
  auto next = [&]() { rng = rng * 6364136223846793005ull + 1442695040888963407ull; return static_cast<uint32_t>(rng >> 33); };
  for (int a = 2; a < argc; ++a) {
    std::vector<uint8_t> file = slurp(argv[a]);
    CHECK(!file.empty());
    for (int trial = 0; trial < 25; ++trial) {
      std::vector<uint8_t> data = file;
      if (trial) for (int k = 0; k < 1 + trial % 6; ++k) data[next() % data.size()] ^= static_cast<uint8_t>(1u << (next() & 7));
      MP4IndexCreator ic(data.size());
      uint64_t off = 0, size = std::min<uint64_t>(1024, data.size());
      for (int it = 0; it < 4096 && !ic.is_done(); ++it) {
        if (off > data.size() || size > data.size() - off) break;
        uint64_t no = 0, ns = 0;
        ic.feed(data.data() + off, size, no, ns);
        off = no; size = ns;
      }
      if (!ic.is_done() || ic.is_error()) continue;
      VideoIndex idx = ic.get_video_index();
      std::vector<uint8_t> wire = idx.serialize();
      VideoIndex back;
      CHECK(VideoIndex::deserialize(wire.data(), wire.size(), &back));
      bool ok = true;
      std::vector<uint64_t> rows;
      for (uint64_t r = 0; r < idx.frames(); r += 3) rows.push_back(r);
      slice_into_video_intervals(idx, rows, &ok);
      if (!idx.consistent()) continue;
      {
        // the same (possibly corrupted) samples through the entropy workers: errors, never out-of-bounds accesses -
        // pictures behind a failed one still decode against whatever the failed one left behind
        int64_t out[9];
        char err[256];
        // (the probe takes the samples on trust, like VideoDecoderInterface::feed: a corrupted sample table that
        // points outside the file is the caller's to refuse - DecoderAutomata does - and is not fed here)
        bool inside = true;
        for (uint64_t f = 0; f < idx.frames(); ++f)
          inside &= idx.sample_offsets()[f] <= data.size() && idx.sample_sizes()[f] <= data.size() - idx.sample_offsets()[f];
        if (inside)
        for (int workers : {2, 4})
          (void)hwang_h264_probe_workers(idx.metadata_bytes().data(), idx.metadata_bytes().size(), data.data(),
                                         idx.sample_offsets().data(), idx.sample_sizes().data(), idx.frames(), nullptr, 0, nullptr, 0,
                                         workers, out, err, sizeof(err));
      }
      for (int compact = 0; compact < 2; ++compact) {
        NullSink sink;
        sink.compact = compact != 0;
        h264::Parser parser(&sink);
        if (!parser.configure(idx.metadata_bytes().data(), idx.metadata_bytes().size())) continue;
        for (uint64_t s = 0; s < idx.frames(); ++s) {
          const uint64_t o = idx.sample_offsets()[s], n = idx.sample_sizes()[s];
          if (o > data.size() || n > data.size() - o) break;
          if (!parser.decode(data.data() + o, n)) break;
        }
        parser.flush();
        if (trial == 0) CHECK(parser.slices() > 0 && parser.slices() == parser.slices_clean_end());
        if (trial == 0) {
          // the last slice NAL of every sample repeated many times (re-decoded macroblocks used to
          // append to the motion-extras stream past its end).  Must be refused, must not write out of bounds.
          NullSink sink2;
          sink2.compact = compact != 0;
          h264::Parser dup(&sink2);
          if (!dup.configure(idx.metadata_bytes().data(), idx.metadata_bytes().size())) continue;
          const int ls = (idx.metadata_bytes()[4] & 3) + 1;
          bool refused = false, expect = false;  // expect: some repeated NAL was a slice that does not start a picture
          for (uint64_t s = 0; s < idx.frames() && !refused; ++s) {
            const uint64_t o = idx.sample_offsets()[s], n = idx.sample_sizes()[s];
            if (o > data.size() || n > data.size() - o) break;
            // find the last NAL of the sample
            uint64_t pos = 0, last = 0, last_len = 0;
            while (pos + ls <= n) {
              uint64_t len = 0;
              for (int i = 0; i < ls; ++i) len = (len << 8) | data[o + pos + i];
              if (len > n - pos - ls) break;
              last = pos;
              last_len = ls + len;
              pos += ls + len;
            }
            std::vector<uint8_t> sample(data.begin() + o, data.begin() + o + n);
            const int type = last_len > static_cast<uint64_t>(ls) ? (data[o + last + ls] & 31) : 0;
            if ((type == 1 || type == 5) && last_len > static_cast<uint64_t>(ls) + 1 && !(data[o + last + ls + 1] & 0x80)) expect = true;
            if (type == 1 || type == 5)
              for (int k = 0; k < 250; ++k) sample.insert(sample.end(), data.begin() + o + last, data.begin() + o + last + last_len);
            if (!dup.decode(sample.data(), sample.size())) refused = true;
          }
          CHECK(refused || !expect);
        }
      }
    }
  }
}

// Entropy workers (num_devices > 1): the parser thread runs ahead through the headers while a pool decodes the slice
// data of consecutive pictures, B pictures wait for their co-located picture's records, finished pictures are handed
// to the sink in decoding order by whichever worker holds the oldest.  Same records as the serial decode.
static void workers_part(int argc, char** argv) {
  for (int a = 2; a < argc; ++a) {
    std::vector<uint8_t> data = slurp(argv[a]);
    CHECK(!data.empty());
    MP4IndexCreator ic(data.size());
    uint64_t off = 0, size = std::min<uint64_t>(1024, data.size());
    for (int it = 0; it < 4096 && !ic.is_done(); ++it) {
      uint64_t no = 0, ns = 0;
      ic.feed(data.data() + off, size, no, ns);
      off = no; size = ns;
    }
    CHECK(ic.is_done() && !ic.is_error());
    VideoIndex idx = ic.get_video_index();
    int64_t serial[9] = {0}, piped[9] = {0};
    char err[256] = {0};
    CHECK(hwang_h264_probe_workers(idx.metadata_bytes().data(), idx.metadata_bytes().size(), data.data(),
                                   idx.sample_offsets().data(), idx.sample_sizes().data(), idx.frames(), nullptr, 0, nullptr, 0, 0,
                                   serial, err, sizeof(err)) == 0);
    for (int workers : {1, 2, 5}) {
      for (int rep = 0; rep < 3; ++rep) {
        CHECK(hwang_h264_probe_workers(idx.metadata_bytes().data(), idx.metadata_bytes().size(), data.data(),
                                       idx.sample_offsets().data(), idx.sample_sizes().data(), idx.frames(), nullptr, 0, nullptr, 0,
                                       workers, piped, err, sizeof(err)) == 0);
        for (int k : {0, 1, 2, 4, 6, 8}) CHECK(piped[k] == serial[k]);
      }
    }
  }
}

int main(int argc, char** argv) {
  if (argc >= 3 && !std::strcmp(argv[1], "threads")) {
    threads_part();
    workers_part(argc, argv);
  } else if (argc >= 2 && !std::strcmp(argv[1], "threads")) threads_part();
  else if (argc >= 3 && !std::strcmp(argv[1], "memory")) memory_part(argc, argv);
  else return 2;
  std::printf("%s: %s\n", argv[1], fails ? "FAILED" : "ok");
  return fails ? 1 : 0;
}
