// Library-level entry points of libamdretrieval (error string, device probes).
#include "common.hpp"

#include <cstring>

namespace amdr {

std::string& last_error_ref() {
  static thread_local std::string e;
  return e;
}

int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  last_error_ref() = buf;
  return code;
}

int check_device(int device) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0) return fail(AMDR_ENODEV, "no HIP device visible (%s)", hipGetErrorString(e));
  if (device < 0 || device >= n) return fail(AMDR_EINVAL, "device %d outside [0,%d)", device, n);
  AMDR_HIP(hipSetDevice(device));
  return AMDR_OK;
}

}  // namespace amdr

extern "C" {

const char* amdr_last_error(void) { return amdr::last_error_ref().c_str(); }

int amdr_version(void) { return 100; /* 0.1.0 */ }

int amdr_device_count(int32_t* count) {
  if (!count) return amdr::fail(AMDR_EINVAL, "device_count: null");
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) {
    *count = 0;
    return amdr::fail(AMDR_ENODEV, "hipGetDeviceCount: %s", hipGetErrorString(e));
  }
  *count = n;
  return AMDR_OK;
}

int amdr_device_name(int32_t device, char* buf, int32_t buf_len) {
  if (!buf || buf_len <= 0) return amdr::fail(AMDR_EINVAL, "device_name: bad buffer");
  hipDeviceProp_t prop;
  hipError_t e = hipGetDeviceProperties(&prop, device);
  if (e != hipSuccess) return amdr::fail(AMDR_ENODEV, "hipGetDeviceProperties: %s", hipGetErrorString(e));
  strncpy(buf, prop.gcnArchName, (size_t)buf_len - 1);
  buf[buf_len - 1] = 0;
  return AMDR_OK;
}

}  // extern "C"
