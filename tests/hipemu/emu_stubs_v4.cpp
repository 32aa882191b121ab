// TEST INFRASTRUCTURE ONLY -- stand-ins for the entry points of gemm_v4.hip in emulation builds that leave that file out (the
// model-level emulation tests: a persistent 256 x 256 tile costs seconds of fiber switching per launch).
#include <hip/hip_runtime.h>

#include "../../include/declip_hip.h"

bool dh_gemm_try_v4(const dh_gemm_args*, int, hipStream_t) { return false; }
bool dh_gemm_try_v4_group(const dh_gemm_args*, int, hipStream_t) { return false; }
bool dh_maxsim_try_v4(const void*, const void*, int, int, int, int, int, float*, uint8_t*, hipStream_t) { return false; }
bool dh_ce_try_v4_fwd(const void*, const void*, const float*, const long long*, int, int, int, int, float*, float*, float*, int64_t, hipStream_t) { return false; }
bool dh_ce_try_v4_bwd(const void*, const void*, const float*, const long long*, const float*, const float*, int, int, int, int, void*, int64_t, hipStream_t) { return false; }
extern "C" int dh_gemm_v4_enable(int) { return 0; }
extern "C" int dh_gemm_v4_set_dynamic(int) { return 0; }
