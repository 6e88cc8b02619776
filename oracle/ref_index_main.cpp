// TEST INFRASTRUCTURE: drives the reference's MP4IndexCreator (compiled from
// /root/reference where it lies, see Makefile `ref`) over a file and prints
// the VideoIndex as JSON, plus the (offset,size) pull-protocol trace.  Used
// to pin this repo's own indexer against the reference on arbitrary files.
#include "hwang/mp4_index_creator.h"

#include <cstdio>
#include <fstream>
#include <iterator>
#include <vector>

int main(int argc, char** argv) {
  if (argc < 2) return 2;
  std::ifstream f(argv[1], std::ios::binary);
  std::vector<uint8_t> data((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
  hwang::MP4IndexCreator indexer(data.size());
  uint64_t off = 0, size = data.size() < 1024 ? data.size() : 1024;
  printf("{\"trace\": [");
  int calls = 0;
  while (!indexer.is_done()) {
    printf("%s[%llu, %llu]", calls ? ", " : "", (unsigned long long)off, (unsigned long long)size);
    ++calls;
    uint64_t noff = 0, nsize = 0;
    indexer.feed(data.data() + off, size, noff, nsize);
    off = noff;
    size = nsize;
    if (off + size > data.size()) break;
  }
  printf("], \"error\": %s", indexer.is_error() ? "true" : "false");
  if (!indexer.is_error()) {
    hwang::VideoIndex idx = indexer.get_video_index();
    printf(", \"timescale\": %u, \"duration\": %llu, \"width\": %u, \"height\": %u, \"format\": \"%s\"",
           idx.timescale(), (unsigned long long)idx.duration(), idx.frame_width(), idx.frame_height(),
           idx.format().c_str());
    printf(", \"sample_offsets\": [");
    for (size_t i = 0; i < idx.sample_offsets().size(); ++i)
      printf("%s%llu", i ? "," : "", (unsigned long long)idx.sample_offsets()[i]);
    printf("], \"sample_sizes\": [");
    for (size_t i = 0; i < idx.sample_sizes().size(); ++i)
      printf("%s%llu", i ? "," : "", (unsigned long long)idx.sample_sizes()[i]);
    printf("], \"keyframes\": [");
    for (size_t i = 0; i < idx.keyframe_indices().size(); ++i)
      printf("%s%llu", i ? "," : "", (unsigned long long)idx.keyframe_indices()[i]);
    printf("], \"metadata\": \"");
    for (uint8_t b : idx.metadata_bytes()) printf("%02x", b);
    printf("\"");
  }
  printf("}\n");
  return 0;
}
