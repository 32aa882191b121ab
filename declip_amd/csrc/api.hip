// Error reporting, version and device query of libdeclip_hip.so.
#include "dh_common.h"
#include <string.h>

static thread_local char g_err[512] = "";

void dh_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* dh_last_error(void) { return g_err; }
extern "C" int dh_version(void) { return 100; }

extern "C" int dh_device_info(int device, int* out4) {
  DH_REQUIRE(out4, "dh_device_info: null out");
  hipDeviceProp_t p;
  hipError_t e = hipGetDeviceProperties(&p, device);
  if (e != hipSuccess) DH_FAIL(DH_ERR_LAUNCH, "hipGetDeviceProperties: %s", hipGetErrorString(e));
  out4[0] = p.multiProcessorCount;
  out4[1] = p.clockRate;
  out4[2] = (int)p.sharedMemPerMultiprocessor;
  int arch = 0;
  const char* g = strstr(p.gcnArchName, "gfx");
  if (g) arch = (int)strtol(g + 3, nullptr, 16);
  out4[3] = arch;
  return DH_OK;
}
