// HBM bandwidth probe for MI355X: what do pure-write, pure-read, copy and "GEMM-epilogue shaped" tile
// stores sustain?  (The tower GEMMs at b=512 write 150-320 MB per launch, so the store path sets their floor.)
//   hipcc --offload-arch=gfx950 -O3 tools/probe_bw.hip -o tools/probe_bw && tools/probe_bw
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>

typedef __attribute__((ext_vector_type(4))) float f4;

__global__ void k_fill(f4* __restrict__ p, long n4) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long stride = (long)gridDim.x * blockDim.x;
  f4 v = {1.f, 2.f, 3.f, 4.f};
  for (; i < n4; i += stride) p[i] = v;
}
__global__ void k_fill_nt(f4* __restrict__ p, long n4) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long stride = (long)gridDim.x * blockDim.x;
  f4 v = {1.f, 2.f, 3.f, 4.f};
  for (; i < n4; i += stride) __builtin_nontemporal_store(v, p + i);
}
__global__ void k_read(const f4* __restrict__ p, long n4, float* out) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long stride = (long)gridDim.x * blockDim.x;
  f4 acc = {0, 0, 0, 0};
  for (; i < n4; i += stride) acc += p[i];
  if (acc.x + acc.y + acc.z + acc.w == 12345.f) out[0] = 1.f;
}
__global__ void k_copy(const f4* __restrict__ s, f4* __restrict__ d, long n4) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long stride = (long)gridDim.x * blockDim.x;
  for (; i < n4; i += stride) d[i] = s[i];
}
// one block = one TM x TN bf16 tile of a row-major [M][N] bf16 matrix; 16 B per lane, lanes cover row segments
template <int TM, int TN>
__global__ void k_tile_store(uint16_t* __restrict__ C, int M, int N, int nt) {
  const int ntx = N / TN;
  const int tile = blockIdx.x;
  const int ty = tile / ntx, tx = tile % ntx;
  constexpr int CPR = TN / 8;                 // 16-byte chunks per tile row
  const int cc = threadIdx.x % CPR, r0 = threadIdx.x / CPR;
  const int rstep = blockDim.x / CPR;
  f4 v = {1.f, 2.f, 3.f, 4.f};
  for (int r = r0; r < TM; r += rstep) {
    const long off = (long)(ty * TM + r) * N + tx * TN + cc * 8;
    if (nt) __builtin_nontemporal_store(v, reinterpret_cast<f4*>(C + off));
    else *reinterpret_cast<f4*>(C + off) = v;
  }
}

template <typename F>
float timeit(F f, int iters = 10) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  f(); f();
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < iters; ++i) f();
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  return ms / iters;
}

int main() {
  const long bytes = 1L << 30;     // 1 GiB buffers (beyond the 256 MiB Infinity Cache)
  f4 *a, *b; float* o;
  hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMalloc(&o, 4);
  hipMemset(a, 0, bytes); hipMemset(b, 0, bytes);
  const long n4 = bytes / 16;
  for (int blocks : {256 * 4, 256 * 8, 256 * 16, 256 * 64}) {
    float t;
    t = timeit([&] { hipLaunchKernelGGL(k_fill, dim3(blocks), dim3(256), 0, 0, a, n4); });
    printf("fill     blocks %6d: %7.3f ms  %6.2f TB/s\n", blocks, t, bytes / t / 1e9);
    t = timeit([&] { hipLaunchKernelGGL(k_fill_nt, dim3(blocks), dim3(256), 0, 0, a, n4); });
    printf("fill_nt  blocks %6d: %7.3f ms  %6.2f TB/s\n", blocks, t, bytes / t / 1e9);
    t = timeit([&] { hipLaunchKernelGGL(k_read, dim3(blocks), dim3(256), 0, 0, a, n4, o); });
    printf("read     blocks %6d: %7.3f ms  %6.2f TB/s\n", blocks, t, bytes / t / 1e9);
    t = timeit([&] { hipLaunchKernelGGL(k_copy, dim3(blocks), dim3(256), 0, 0, a, b, n4); });
    printf("copy     blocks %6d: %7.3f ms  %6.2f TB/s (read+write)\n", blocks, t, 2.0 * bytes / t / 1e9);
  }
  // smaller working sets (what one GEMM launch writes): 157 MB
  {
    const long sz = 25600L * 3072 * 2;
    float t = timeit([&] { hipLaunchKernelGGL(k_fill, dim3(256 * 16), dim3(256), 0, 0, a, sz / 16); });
    printf("fill 157 MB (same buffer each iter): %7.3f ms  %6.2f TB/s\n", t, sz / t / 1e9);
    int it = 0;
    t = timeit([&] { hipLaunchKernelGGL(k_fill, dim3(256 * 16), dim3(256), 0, 0, (f4*)((char*)a + (it++ % 6) * sz), sz / 16); }, 12);
    printf("fill 157 MB (rotating 6 buffers)   : %7.3f ms  %6.2f TB/s\n", t, sz / t / 1e9);
  }
  // GEMM-epilogue shaped stores: C [25600][3072] bf16
  {
    const int M = 25600, N = 3072;
    const long sz = (long)M * N * 2;
    uint16_t* C = (uint16_t*)a;
    for (int nt = 0; nt < 2; ++nt) {
      int it = 0;
      float t = timeit([&] { hipLaunchKernelGGL((k_tile_store<128, 128>), dim3((M / 128) * (N / 128)), dim3(256), 0, 0, (uint16_t*)((char*)C + (it++ % 6) * sz), M, N, nt); }, 12);
      printf("tile store 128x128 nt=%d: %7.3f ms  %6.2f TB/s\n", nt, t, sz / t / 1e9);
      t = timeit([&] { hipLaunchKernelGGL((k_tile_store<256, 128>), dim3((M / 256) * (N / 128)), dim3(256), 0, 0, (uint16_t*)((char*)C + (it++ % 6) * sz), M, N, nt); }, 12);
      printf("tile store 256x128 nt=%d: %7.3f ms  %6.2f TB/s\n", nt, t, sz / t / 1e9);
      t = timeit([&] { hipLaunchKernelGGL((k_tile_store<256, 256>), dim3((M / 256) * (N / 256)), dim3(512), 0, 0, (uint16_t*)((char*)C + (it++ % 6) * sz), M, N, nt); }, 12);
      printf("tile store 256x256 nt=%d: %7.3f ms  %6.2f TB/s\n", nt, t, sz / t / 1e9);
      t = timeit([&] { hipLaunchKernelGGL((k_tile_store<64, 512>), dim3((M / 64) * (N / 512)), dim3(256), 0, 0, (uint16_t*)((char*)C + (it++ % 6) * sz), M, N, nt); }, 12);
      printf("tile store  64x512 nt=%d: %7.3f ms  %6.2f TB/s\n", nt, t, sz / t / 1e9);
    }
  }
  hipError_t e = hipDeviceSynchronize();
  printf("status: %s\n", hipGetErrorString(e));
  return 0;
}
