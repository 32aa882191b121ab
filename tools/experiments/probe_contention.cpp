// What happens to a full-chip persistent GEMM when another stream holds a few CUs (RCCL kernels during the overlapped
// gradient all-reduce do exactly that)?  A "hog" kernel occupies `n` CUs (one 512-thread block with 100 KiB LDS each) for
// ~100 ms on a second stream while the v4 GEMM is timed on the first.
//   hipcc -O2 tools/probe_contention.cpp -Iinclude -Ldeclip_amd -ldeclip_hip -Wl,-rpath,'$ORIGIN/../declip_amd' -o tools/probe_contention
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "declip_hip.h"
__global__ __launch_bounds__(512) void hog(long cycles, float* sink) {
  extern __shared__ float lds[];
  lds[threadIdx.x] = threadIdx.x;
  const long t0 = wall_clock64();
  float x = threadIdx.x;
  while (wall_clock64() - t0 < cycles) { x = x * 1.0001f + lds[(threadIdx.x * 7) & 511]; __builtin_amdgcn_s_sleep(16); }
  if (x == 12345.f) sink[0] = x;
}
int main(int argc, char** argv) {
  const int M = 25600, N = 2304, K = 768;
  void *A, *B, *C; float *bias, *sink;
  hipMalloc(&A, (size_t)M * K * 2); hipMalloc(&B, (size_t)N * K * 2); hipMalloc(&C, (size_t)M * N * 2); hipMalloc(&bias, N * 4); hipMalloc(&sink, 4);
  hipMemset(A, 0x3c, (size_t)M * K * 2); hipMemset(B, 0x3c, (size_t)N * K * 2); hipMemset(bias, 0, N * 4);
  hipStream_t s1, s2; hipStreamCreate(&s1); hipStreamCreate(&s2);
  hipFuncSetAttribute((const void*)hog, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
  dh_gemm_args g; memset(&g, 0, sizeof(g));
  g.dtype = DH_BF16; g.c_dtype = DH_BF16; g.M = M; g.N = N; g.K = K; g.A = A; g.lda = K; g.B = B; g.ldb = K; g.C = C; g.ldc = N;
  g.bias = bias; g.alpha = 1.f; g.force_generic = 4; g.split_k = 1;
  printf("warmup\n"); fflush(stdout);
  for (int i = 0; i < 300; ++i) if (dh_gemm(&g, s1)) { printf("err %s\n", dh_last_error()); return 1; }
  printf("warm done\n"); fflush(stdout);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int nhog : {0, 4, 16, 32, 64}) {
    if (nhog) hipLaunchKernelGGL(hog, dim3(nhog), dim3(512), 100 * 1024, s2, 8000000L /* 80 ms at 100 MHz */, sink);
    // give the hog a moment to become resident
    for (int i = 0; i < 20; ++i) dh_gemm(&g, s1);
    hipEventRecord(e0, s1);
    for (int i = 0; i < 100; ++i) dh_gemm(&g, s1);
    hipEventRecord(e1, s1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("hog on %2d CUs: GEMM %.1f us  (x%.2f)\n", nhog, ms * 10, 0.0);
    hipDeviceSynchronize();
  }
  return 0;
}
