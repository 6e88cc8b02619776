// Test-only stand-in for <glog/logging.h> (not in this image): the reference's
// hwang/common.h includes it, the adapter under test logs nothing.  Used only
// by tests/test_integration_shim.py to COMPILE the INTEGRATION.md adapter
// against the reference's interface headers; nothing of the reference is
// built with it.
#pragma once
#include <iostream>
#include <string>
struct HwangTestNullLog {
  template <class T>
  HwangTestNullLog& operator<<(const T&) { return *this; }
};
#define LOG(x) HwangTestNullLog()
#define VLOG(x) HwangTestNullLog()
