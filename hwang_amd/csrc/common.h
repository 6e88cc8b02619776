// Boundary value types shared by every host-side component.
//
// Mirrors the reference's hwang/common.h:18-68 (DeviceType, DeviceHandle,
// Result, HWANG_RETURN_ON_ERROR) so that a backend written against these
// types drops into the reference tree unchanged.  Differences on purpose:
//  * no glog dependency (the reference includes it at common.h:16);
//  * DeviceHandle::operator< is a real strict weak order (the reference's
//    `type < other.type && id < other.id`, common.h:33-35, is not one).
#pragma once

#include <cstdint>
#include <string>
#include <tuple>

namespace hwang {

enum class DeviceType {
  CPU = 0,
  GPU = 1,
};

struct DeviceHandle {
  bool operator==(const DeviceHandle& other) const {
    return type == other.type && id == other.id;
  }
  bool operator!=(const DeviceHandle& other) const { return !(*this == other); }
  bool operator<(const DeviceHandle& other) const {
    return std::tie(type, id) < std::tie(other.type, other.id);
  }
  bool can_copy_to(const DeviceHandle& other) const {
    return !(type == DeviceType::GPU && other.type == DeviceType::GPU &&
             id != other.id);
  }
  bool is_same_address_space(const DeviceHandle& other) const {
    return type == other.type &&
           (type == DeviceType::CPU || id == other.id);
  }

  DeviceType type;
  int32_t id;
};

static const DeviceHandle CPU_DEVICE = {DeviceType::CPU, 0};

struct Result {
  Result() : ok(true) {}
  Result(bool _ok, const std::string& _message) : ok(_ok), message(_message) {}

  bool ok;
  std::string message;
};

#define HWANG_RETURN_ON_ERROR(expr__) \
  do {                                \
    ::hwang::Result res__ = (expr__); \
    if (!res__.ok) {                  \
      return res__;                   \
    }                                 \
  } while (0);

}  // namespace hwang
