// TEST INFRASTRUCTURE ONLY -- the GEMM families written with inline ISA (gemm_v4 / gemm_v3 / gemm_glds) cannot be emulated:
// on the host build the dispatcher of gemm.hip sees them decline every problem and falls through to the plain HIP C++ kernels
// (MFMA-builtin tiles and the generic VALU kernel), which are what the emulation covers.
#include <hip/hip_runtime.h>

#include "../../include/declip_hip.h"

bool dh_gemm_try_glds(const dh_gemm_args*, int, hipStream_t) { return false; }
bool dh_gemm_try_v3(const dh_gemm_args*, int, hipStream_t) { return false; }
bool dh_gemm_try_v4(const dh_gemm_args*, int, hipStream_t) { return false; }
bool dh_gemm_try_v5(const dh_gemm_args*, hipStream_t) { return false; }
bool dh_gemm_try_v4_group(const dh_gemm_args*, int, hipStream_t) { return false; }
bool dh_maxsim_try_v4(const void*, const void*, int, int, int, int, int, float*, uint8_t*, hipStream_t) { return false; }
bool dh_ce_try_v4_fwd(const void*, const void*, const float*, const long long*, int, int, int, int, float*, float*, float*, int64_t, hipStream_t) { return false; }
bool dh_ce_try_v4_bwd(const void*, const void*, const float*, const long long*, const float*, const float*, int, int, int, int, void*, int64_t, hipStream_t) { return false; }
