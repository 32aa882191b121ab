// Hardware-semantics probe (run once on the MI355X box; output parsed by hand):
//   1. ds_read_b64_tr_b16 : which LDS elements does lane l receive?
//   2. global_load_lds_dwordx4 : where do a wave's 64 x 16 B land in LDS?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(4))) short s4;

__global__ void probe_tr(uint16_t* out, int stride_elems) {
  __shared__ __attribute__((aligned(16))) uint16_t lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (uint16_t)i;
  __syncthreads();
  // lane l supplies the address of 4 contiguous elements: row (l) of a [64][stride] matrix, col 0
  s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)(lds + threadIdx.x * stride_elems));
  for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = (uint16_t)v[j];
}

__global__ void probe_glds(const uint32_t* g, uint32_t* out, int perm) {
  __shared__ __attribute__((aligned(16))) uint32_t lds[1024];
  for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = 0xdeadbeefu;
  __syncthreads();
  int src_lane = perm ? (threadIdx.x ^ 1) : threadIdx.x;
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + src_lane * 4),
                                   (__attribute__((address_space(3))) void*)(lds + 64), 16, 0, 0);
  __builtin_amdgcn_s_waitcnt(0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 1024; i += 64) out[i] = lds[i];
}

int main() {
  uint16_t* d_out; uint32_t *d_g, *d_o2;
  hipMalloc(&d_out, 256 * 2); hipMalloc(&d_g, 4096 * 4); hipMalloc(&d_o2, 1024 * 4);
  uint32_t hg[4096]; for (int i = 0; i < 4096; ++i) hg[i] = i;
  hipMemcpy(d_g, hg, sizeof(hg), hipMemcpyHostToDevice);
  for (int stride : {4, 16, 64}) {
    hipLaunchKernelGGL(probe_tr, dim3(1), dim3(64), 0, 0, d_out, stride);
    uint16_t h[256]; hipMemcpy(h, d_out, sizeof(h), hipMemcpyDeviceToHost);
    printf("== ds_read_tr16_b64, lane address = lds + lane*%d elems: result[lane] = 4 LDS element indices\n", stride);
    for (int l = 0; l < 64; ++l) printf("lane %2d: %4d %4d %4d %4d\n", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
  }
  for (int perm = 0; perm < 2; ++perm) {
    hipLaunchKernelGGL(probe_glds, dim3(1), dim3(64), 0, 0, d_g, d_o2, perm);
    uint32_t h[1024]; hipMemcpy(h, d_o2, sizeof(h), hipMemcpyDeviceToHost);
    printf("== global_load_lds 16B, perm=%d: LDS dwords 56..335 (base offset 64 dwords)\n", perm);
    for (int i = 56; i < 336; ++i) { if (h[i] == 0xdeadbeefu) printf(" ----"); else printf(" %4u", h[i]); if ((i - 56) % 16 == 15) printf("\n"); }
    printf("\n");
  }
  hipError_t e = hipDeviceSynchronize();
  printf("status: %s\n", hipGetErrorString(e));
  return 0;
}
