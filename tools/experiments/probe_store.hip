// Per-CU store throughput: how fast can ONE workgroup per CU write full 512-byte row segments when only a few
// CUs are writing (HBM not the limit)?  blocks = number of CUs writing; each block writes `tiles` 256x256 bf16 tiles.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(4))) float f4;
template <int VEC>
__global__ __launch_bounds__(512) void k_store(uint16_t* __restrict__ C, int N, int tiles) {
  const int t = threadIdx.x;
  f4 v = {1.f, 2.f, 3.f, (float)t};
  for (int tile = 0; tile < tiles; ++tile) {
    uint16_t* base = C + ((long)blockIdx.x * tiles + tile) * 256L * N;      // 256 rows further down each tile
    if (VEC == 16) {
      const int cc = t & 31, r0 = t >> 5;
#pragma unroll 4
      for (int it = 0; it < 16; ++it) *reinterpret_cast<f4*>(base + (long)(r0 + 16 * it) * N + cc * 8) = v;
    } else {
      const int cc = t & 63, r0 = t >> 6;
#pragma unroll 4
      for (int it = 0; it < 32; ++it) *reinterpret_cast<float2*>(base + (long)(r0 + 8 * it) * N + cc * 4) = make_float2(v.x, v.y);
    }
  }
}
int main() {
  const int N = 3072;
  uint16_t* C; hipMalloc(&C, 4L << 30);
  hipMemset(C, 0, 4L << 30);
  for (int vec : {16, 8}) for (int blocks : {1, 8, 32, 64, 128, 256}) {
    int tiles = 2048 / blocks; if (tiles > 64) tiles = 64; if (tiles < 4) tiles = 4;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      if (vec == 16) hipLaunchKernelGGL(k_store<16>, dim3(blocks), dim3(512), 0, 0, C, N, tiles);
      else hipLaunchKernelGGL(k_store<8>, dim3(blocks), dim3(512), 0, 0, C, N, tiles);
      hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double bytes = (double)blocks * tiles * 256 * 512;
    printf("vec %2d B, %3d CUs x %2d tiles: %8.3f ms  %7.2f GB/s per CU  (%.2f TB/s total, %.1f B/clk/CU @2.4GHz)\n", vec, blocks, tiles, ms, bytes / blocks / ms / 1e6, bytes / ms / 1e9, bytes / blocks / ms / 1e6 / 2.4);
  }
  printf("status: %s\n", hipGetErrorString(hipDeviceSynchronize()));
}
