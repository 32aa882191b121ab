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

// A stream capture that was INVALIDATED (an operation that is illegal under capture ran on the capturing stream) stays open until somebody
// ends it; torch.cuda.graph's exit raises on exactly that call and then leaves the stream behind.  graph.GraphedStep(fallback=True) calls this
// on the abandoned capture stream before it re-runs the step eagerly.  Returns 1 if a capture was open and has been ended, 0 if none was.
extern "C" int dh_stream_abandon_capture(dh_stream_t stream) {
  hipStream_t st = (hipStream_t)stream;
  hipStreamCaptureStatus status = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(st, &status) != hipSuccess) { (void)hipGetLastError(); return 0; }
  if (status == hipStreamCaptureStatusNone) return 0;
  hipGraph_t g = nullptr;
  (void)hipStreamEndCapture(st, &g);          // (returns hipErrorStreamCaptureInvalidated for an invalidated capture: expected)
  if (g) (void)hipGraphDestroy(g);
  (void)hipGetLastError();
  return 1;
}
