// Timeline of one v4 GEMM launch (trace build of the library: tools/build_trace.sh):
//   LD_LIBRARY_PATH=build/trace tools/gemm_trace M N K epi(0 none,1 gelu,2 dgelu [dX layout],3 residual,4 plain dX layout) dw stamps-per-tile
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "declip_hip.h"
int main(int argc, char** argv) {
  int M = argc > 1 ? atoi(argv[1]) : 25600, N = argc > 2 ? atoi(argv[2]) : 2304, K = argc > 3 ? atoi(argv[3]) : 768, epi = argc > 4 ? atoi(argv[4]) : 0;
  int dw = argc > 5 ? atoi(argv[5]) : 0;
  const int flavour = epi;                 // 2: dGELU (B contraction-major, aux = stored QuickGELU'); 3: residual; 4: plain dX layout
  const bool bkm = flavour == 2 || flavour == 4;
  if (flavour >= 3) epi = 0;   // 1: weight-gradient call (both operands contraction-major, fp32 accumulate, workspace, bias gradient)
  void* h = dlopen("libdeclip_hip.so", RTLD_NOW | RTLD_GLOBAL);
  auto rd = (int (*)(long*, int))dlsym(h, "dh_v4_trace_read");
  auto clr = (int (*)())dlsym(h, "dh_v4_trace_clear");
  if (!rd || !clr) { printf("not a trace build\n"); return 1; }
  void *A, *B, *C, *X; float* bias;
  hipMalloc(&A, (size_t)M * K * 2); hipMalloc(&B, (size_t)N * K * 2); hipMalloc(&C, (size_t)M * N * 4); hipMalloc(&X, (size_t)M * N * 2); hipMalloc(&bias, (N + M) * 4);
  void* ws; hipMalloc(&ws, 256u << 20); hipMemset(C, 0, (size_t)M * N * 4);
  std::vector<uint16_t> hb((size_t)(M > N ? M : N) * K);
  uint32_t s = 1; for (auto& v : hb) { s = s * 1664525u + 1013904223u; v = (uint16_t)(0x3c00 + ((s >> 9) & 0x3ff) + ((s >> 31) << 15)); }
  hipMemcpy(A, hb.data(), (size_t)M * K * 2, hipMemcpyHostToDevice);
  hipMemcpy(B, hb.data(), (size_t)N * K * 2, hipMemcpyHostToDevice);
  hipMemset(bias, 0, N * 4);
  dh_gemm_args g; memset(&g, 0, sizeof(g));
  g.dtype = DH_BF16; g.c_dtype = DH_BF16; g.M = M; g.N = N; g.K = K; g.A = A; g.lda = K; g.B = B; g.ldb = K; g.C = C; g.ldc = N;
  g.bias = bias; g.epilogue = epi; g.aux = epi ? X : nullptr; g.ldaux = N; g.alpha = 1.f; g.force_generic = 4; g.split_k = 1;
  if (bkm) { g.b_kmajor = 1; g.ldb = N; g.bias = nullptr; }
  if (flavour == 3) { g.residual = X; g.ldr = N; }
  if (flavour == 2) hipMemset(X, 0x3f, (size_t)M * N * 2);
  if (dw) { g.a_kmajor = g.b_kmajor = 1; g.lda = M; g.ldb = N; g.c_dtype = DH_F32; g.accumulate = 1; g.bias = nullptr; g.ws = ws; g.ws_bytes = 256u << 20; g.a_colsum = bias; g.split_k = 8; }
  for (int i = 0; i < 200; ++i) dh_gemm(&g, nullptr);     // warm clocks
  hipDeviceSynchronize();
  clr();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0); dh_gemm(&g, nullptr); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("M=%d N=%d K=%d flavour=%d dw=%d: %.1f us (incl. the reduce pass for dw)\n", M, N, K, flavour, dw, ms * 1e3);
  std::vector<long> tb(6 * 256);
  rd(tb.data(), 6 * 256);
  // stamps per tile: 10 since the continuous K-tile stream of round 3 (tile started, K-tiles 0..nk-3 done, K-tile nk-2 done, main
  // loop done, epilogue arguments loaded, quarter passes 0-3 staged, end of item); 7 for a trace build of the round-2 kernel
  const int spt = argc > 6 ? atoi(argv[6]) : 10;
  const char* names10[10] = {"", "steady", "kt(nk-2)", "kt(nk-1)", "args", "q0", "q1", "q2", "q3", "q4+tail"};
  const char* names7[7] = {"", "mainloop", "req-next", "stage0", "store0", "stage1", "store1"};
  // -DV4_TRACE=2: + load A done, barrier, MFMA A done, load B done inside K-tiles nk-2 and nk-1
  const char* names18[18] = {"", "steady", "LA", "bar", "MA+bar", "LB", "MB(kt nk-2)", "LA", "bar", "MA+bar", "LB", "MB+realign(kt nk-1)", "args", "q0", "q1", "q2", "q3", "q4+tail"};
  // -DV4_TRACE=3: load A done, barrier, MFMA A done, load B done inside K-tile 2 (steady state), then the 10 coarse stamps
  const char* names14[14] = {"", "kt0-1+LA(2)", "bar", "MA+bar", "LB", "MB(2)+rest of steady", "kt(nk-2)", "kt(nk-1)", "args", "q0", "q1", "q2", "q3", "q4+tail"};
  const char** names = spt == 10 ? names10 : spt == 18 ? names18 : spt == 14 ? names14 : names7;
  for (int slot = 0; slot < 6; ++slot) {
    long* p = tb.data() + slot * 256; int n = (int)p[0];
    printf("WG %d wave %d: %d stamps; per tile [cycles]: ", slot / 2 == 0 ? 0 : slot / 2 == 1 ? 100 : 200, (slot & 1) * 4, n);
    for (int i = 0; i + spt - 1 < n; i += spt) {
      printf("\n   tile %d: t0=%ld |", i / spt, p[1 + i] - p[1]);
      for (int j = 1; j < spt; ++j) printf(" %s %ld", names[j], p[1 + i + j] - p[1 + i + j - 1]);
      if (i + spt < n) printf(" | -> next tile %ld", p[1 + i + spt] - p[1 + i + spt - 1]);
    }
    printf("\n");
  }
  return 0;
}
