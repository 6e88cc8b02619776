// TEST FIXTURE: the reference's OWN DecoderAutomata (compiled from /root/reference/hwang/decoder_automata.cpp where it
// lies, with tests/stubs/glog/logging.h standing in for the logging library the image lacks) driving this library's
// backend through INTEGRATION.md's adapter, extracted verbatim from that file by tests/test_integration_shim.py.
// Nothing here is an oracle and nothing of it ships: it is the caller that sits ABOVE the drop-in seam
// (video_decoder_interface.h:26-55), proving that hwang's feeder/retriever state machine (decoder_automata.cpp:120-404)
// runs unmodified over `hwang_decoder_*`.
//
//   ref_automata_driver <plan> <blob> <out> <decoder_type> <device_type: 0 cpu, 1 gpu>
//
// <plan>: text, written by the test from this repo's planner (itself pinned to the reference's planner goldens):
//   width height format
//   extradata as hex ("-" = none)
//   n_intervals
//   per interval five lines:  start end blob_offset blob_length | sample_offsets... | sample_sizes... | keyframes... | valid_frames...
//   get_frames chunk sizes...
// EncodedData is filled the way the reference's own callers fill it (python/hwang/decoder.py:47-63: bytes of the
// interval, offsets rebased to its start).  <out>: the returned frames, w*h*3 bytes each, back to back.
#include "hwang/decoder_automata.h"
#include "hwang/impls/amd/amd_video_decoder_shim.h"
#include "hwang/video_decoder_factory.h"

#include <dlfcn.h>

#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <memory>
#include <sstream>

namespace hwang {
// The factory case of INTEGRATION.md section 1, test-local: the reference's factory translation unit includes FFmpeg
// and CUDA headers (video_decoder_factory.cpp:16-29) and cannot be compiled here.
VideoDecoderInterface* VideoDecoderFactory::make_from_config(DeviceHandle device_handle, uint32_t num_devices,
                                                             VideoDecoderType type) {
  return new AMDVideoDecoderShim(device_handle.id, device_handle.type, static_cast<int>(num_devices),
                                 static_cast<int>(type));
}
}  // namespace hwang

static std::vector<uint64_t> numbers(const std::string& line) {
  std::vector<uint64_t> v;
  std::istringstream is(line);
  uint64_t x;
  while (is >> x) v.push_back(x);
  return v;
}

int main(int argc, char** argv) {
  if (argc < 6) {
    std::fprintf(stderr, "usage: %s plan blob out decoder_type device_type\n", argv[0]);
    return 2;
  }
  const int decoder_type = std::atoi(argv[4]);
  const int device_type = std::atoi(argv[5]);
  void* fixture = nullptr;
  if (const char* so = std::getenv("HWANG_TEST_STAMP_LIB")) {  // a backend registered at run time (include/hwang_amd.h:44-59)
    fixture = dlopen(so, RTLD_NOW | RTLD_GLOBAL);
    if (!fixture) {
      std::fprintf(stderr, "dlopen %s: %s\n", so, dlerror());
      return 2;
    }
    auto* vt = static_cast<const hwang_backend_vtable*>(dlsym(fixture, "hwang_stamp_vtable"));
    if (!vt || hwang_register_decoder_backend(decoder_type, vt) != 0) {
      std::fprintf(stderr, "could not register the fixture backend as type %d\n", decoder_type);
      return 2;
    }
  }
  std::ifstream plan(argv[1]);
  std::ifstream blobf(argv[2], std::ios::binary);
  std::vector<uint8_t> blob((std::istreambuf_iterator<char>(blobf)), std::istreambuf_iterator<char>());
  std::string line;
  uint32_t width = 0, height = 0;
  std::string format;
  std::getline(plan, line);
  {
    std::istringstream is(line);
    is >> width >> height >> format;
  }
  std::vector<uint8_t> extradata;
  std::getline(plan, line);
  if (line != "-")
    for (size_t i = 0; i + 1 < line.size(); i += 2) extradata.push_back(static_cast<uint8_t>(std::stoi(line.substr(i, 2), nullptr, 16)));
  std::getline(plan, line);
  const size_t n_intervals = std::stoul(line);
  std::vector<hwang::DecoderAutomata::EncodedData> data(n_intervals);
  size_t total = 0;
  for (auto& d : data) {
    std::getline(plan, line);
    const auto head = numbers(line);
    d.width = width;
    d.height = height;
    d.format = format;
    d.start_keyframe = head[0];
    d.end_keyframe = head[1];
    d.encoded_video.assign(blob.begin() + head[2], blob.begin() + head[2] + head[3]);
    std::getline(plan, line);
    d.sample_offsets = numbers(line);
    std::getline(plan, line);
    d.sample_sizes = numbers(line);
    std::getline(plan, line);
    d.keyframes = numbers(line);
    std::getline(plan, line);
    d.valid_frames = numbers(line);
    total += d.valid_frames.size();
  }
  std::getline(plan, line);
  const auto chunks = numbers(line);

  hwang::DeviceHandle handle;
  handle.type = device_type ? hwang::DeviceType::GPU : hwang::DeviceType::CPU;
  handle.id = 0;
  std::unique_ptr<hwang::DecoderAutomata> automata(
      hwang::DecoderAutomata::make_instance(handle, 1, static_cast<hwang::VideoDecoderType>(decoder_type)));
  hwang::Result r = automata->initialize(data, extradata);
  if (!r.ok) {
    std::printf("initialize failed: %s\n", r.message.c_str());
    return 1;
  }
  const size_t frame_size = static_cast<size_t>(width) * height * 3;
  std::vector<uint8_t> frames(total * frame_size);
  size_t got = 0;
  for (uint64_t c : chunks) {
    r = automata->get_frames(frames.data() + got * frame_size, static_cast<int32_t>(c));
    if (!r.ok) {
      std::printf("get_frames failed: %s\n", r.message.c_str());
      return 1;
    }
    got += c;
  }
  automata.reset();  // joins the feeder (decoder_automata.cpp:55-78)
  std::ofstream(argv[3], std::ios::binary).write(reinterpret_cast<const char*>(frames.data()), static_cast<std::streamsize>(got * frame_size));
  std::printf("frames=%zu\n", got);
  if (fixture) {  // the fixture's call counters: configures, feeds, keyframe_feeds, empty_feeds, flushes, gets, discards, hints
    auto trace = reinterpret_cast<void (*)(int32_t*)>(dlsym(fixture, "hwang_stamp_trace"));
    int32_t t[8] = {0};
    if (trace) trace(t);
    std::printf("trace=%d %d %d %d %d %d %d %d\n", t[0], t[1], t[2], t[3], t[4], t[5], t[6], t[7]);
  }
  return 0;
}
