// Backends registered at run time through the C plug-in boundary
// (hwang_backend_vtable, include/hwang_amd.h): the reverse of INTEGRATION.md -
// there hwang's factory gains a case for this library's backend, here this
// library's factory (and with it DecoderAutomata and ParallelRetriever) accepts
// a backend someone else provides.  The reference's factory is compile-time
// only (video_decoder_factory.cpp:31-98).
#include "../../include/hwang_amd.h"

#include "sparse_hints.h"
#include "video_decoder_factory.h"

#include <map>
#include <mutex>

namespace hwang {
namespace {

class RegisteredBackend : public VideoDecoderInterface, public SparseDecodeHints {
 public:
  RegisteredBackend(const hwang_backend_vtable& v, void* self) : v_(v), self_(self) {}
  ~RegisteredBackend() override { v_.destroy(self_); }
  Result configure(const FrameInfo& m, const std::vector<uint8_t>& extradata) override {
    return wrap(v_.configure(self_, m.width, m.height, m.format.c_str(), extradata.data(), extradata.size()));
  }
  Result feed(const uint8_t* p, size_t n, bool keyframe) override { return wrap(v_.feed(self_, p, n, keyframe ? 1 : 0)); }
  Result flush() override { return wrap(v_.flush(self_)); }
  Result discard_frame() override { return wrap(v_.discard_frame(self_)); }
  Result get_frame(uint8_t* dst, size_t n) override { return wrap(v_.get_frame(self_, dst, n)); }
  int decoded_frames_buffered() override { return v_.decoded_frames_buffered(self_); }
  Result wait_until_frames_copied() override { return wrap(v_.wait_until_frames_copied(self_)); }
  void set_wanted_frames(const std::vector<uint64_t>& ordinals) override {
    if (v_.set_wanted) v_.set_wanted(self_, ordinals.data(), ordinals.size());
  }

 private:
  Result wrap(int rc) {
    if (rc == 0) return Result();
    const char* e = v_.last_error ? v_.last_error(self_) : nullptr;
    return Result(false, e && *e ? e : "registered backend failed");
  }
  const hwang_backend_vtable v_;
  void* const self_;
};

std::mutex g_mu;
std::map<int, hwang_backend_vtable>& registry() {
  static std::map<int, hwang_backend_vtable> r;
  return r;
}

}  // namespace

bool VideoDecoderFactory::register_backend(int decoder_type, const hwang_backend_vtable* vtable) {
  if (decoder_type < HWANG_DECODER_FIRST_REGISTERED || !vtable || !vtable->create || !vtable->destroy || !vtable->configure ||
      !vtable->feed || !vtable->flush || !vtable->discard_frame || !vtable->get_frame || !vtable->decoded_frames_buffered ||
      !vtable->wait_until_frames_copied)
    return false;
  std::lock_guard<std::mutex> lk(g_mu);
  registry()[decoder_type] = *vtable;
  return true;
}

std::vector<VideoDecoderType> VideoDecoderFactory::registered_types() {
  std::lock_guard<std::mutex> lk(g_mu);
  std::vector<VideoDecoderType> out;
  for (auto& kv : registry()) out.push_back(static_cast<VideoDecoderType>(kv.first));
  return out;
}

VideoDecoderInterface* VideoDecoderFactory::make_registered(DeviceHandle device, uint32_t num_devices, VideoDecoderType type) {
  hwang_backend_vtable v;
  {
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = registry().find(static_cast<int>(type));
    if (it == registry().end()) return nullptr;
    v = it->second;
  }
  void* self = v.create(static_cast<int>(device.type), device.id, num_devices);
  if (!self) return nullptr;
  return new RegisteredBackend(v, self);
}

}  // namespace hwang

extern "C" int hwang_register_decoder_backend(int decoder_type, const hwang_backend_vtable* vtable) {
  return hwang::VideoDecoderFactory::register_backend(decoder_type, vtable) ? 0 : 1;
}
