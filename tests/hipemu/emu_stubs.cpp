// TEST INFRASTRUCTURE ONLY -- the GEMM families written entirely in inline ISA (gemm_v3 / gemm_glds) cannot be emulated:
// on the host build the dispatcher of gemm.hip sees them decline every problem and falls through to the kernels the emulation
// covers.  gemm_v4 -- the benchmarked kernel -- keeps its inline ISA behind macros and DOES compile for the emulation (build it
// instead of emu_stubs_v4.cpp: tests/hipemu_util.py V4_SOURCES).
#include <hip/hip_runtime.h>

#include "../../include/declip_hip.h"

bool dh_gemm_try_glds(const dh_gemm_args*, int, hipStream_t) { return false; }
bool dh_gemm_try_v3(const dh_gemm_args*, int, hipStream_t) { return false; }
bool dh_gemm_try_v6(const dh_gemm_args*, hipStream_t) { return false; }   // gemm_v6.hip (round-5 experiment) is hardware-only
