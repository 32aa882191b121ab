// How long does one workgroup s_barrier take on gfx950?  (v4's empty 8-barriers-per-K-tile skeleton measured
// ~160 ns per barrier in-step; this isolates the instruction.)
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k_bar(int n, long* out) {
  long t0 = __builtin_readcyclecounter();
  long w0 = wall_clock64();
  for (int i = 0; i < n; ++i) {
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  }
  long t1 = __builtin_readcyclecounter();
  long w1 = wall_clock64();
  if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = t1 - t0; out[1] = w1 - w0; }
}
// barrier + a little independent work between (8 dependent v_fma)
__global__ void k_bar_work(int n, long* out, float* sink) {
  float x = threadIdx.x;
  long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < n; ++i) {
#pragma unroll
    for (int j = 0; j < 8; ++j) x = x * 1.0001f + 0.5f;
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  }
  long t1 = __builtin_readcyclecounter();
  if (x == 1234.5f) sink[0] = x;
  if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = t1 - t0; }
}
int main() {
  long* d; float* s; hipMalloc(&d, 16); hipMalloc(&s, 4);
  const int n = 10000;
  for (int threads : {64, 128, 256, 512, 1024}) {
    for (int blocks : {1, 256, 512}) {
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      hipLaunchKernelGGL(k_bar, dim3(blocks), dim3(threads), 0, 0, n, d);
      hipDeviceSynchronize();
      hipEventRecord(e0);
      hipLaunchKernelGGL(k_bar, dim3(blocks), dim3(threads), 0, 0, n, d);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      long h[2]; hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
      printf("threads %4d blocks %3d: %.1f ns/barrier (event)  %.1f cyc/barrier (s_memtime)  wallclk ticks/barrier %.2f\n", threads, blocks, ms * 1e6 / n, (double)h[0] / n, (double)h[1] / n);
    }
  }
  hipLaunchKernelGGL(k_bar_work, dim3(256), dim3(512), 0, 0, n, d, s);
  hipDeviceSynchronize();
  long h[2]; hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
  printf("512 thr, barrier + 8 dependent fma: %.1f cyc/iter\n", (double)h[0] / n);
  printf("status: %s\n", hipGetErrorString(hipDeviceSynchronize()));
}
