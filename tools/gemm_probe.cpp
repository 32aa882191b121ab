// Native GEMM probe (no Python start-up cost on a fresh GPU box): links libdeclip_hip.so through the C-ABI,
// checks the v4 kernel against the VALU fp32-FMA kernel on ragged mid-size problems and against v2 on the tower
// shapes, then times v2 (force 3) vs v4 (force 4) vs auto on every tower GEMM shape.
//   hipcc -O2 tools/gemm_probe.cpp -Iinclude -Ldeclip_amd -ldeclip_hip -Wl,-rpath,'$ORIGIN/../declip_amd' -o tools/gemm_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <vector>
#include <string>
#include "declip_hip.h"

static uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); return (uint16_t)(u >> 16); }
static float bf2f(uint16_t v) { uint32_t u = ((uint32_t)v) << 16; float f; memcpy(&f, &u, 4); return f; }

struct Buf { void* p; size_t n; };
static void* dalloc(size_t bytes) { void* p; if (hipMalloc(&p, bytes) != hipSuccess) { printf("hipMalloc failed\n"); exit(1); } return p; }
static uint16_t* rand_bf16(size_t n, uint32_t seed, float scale = 1.f) {
  std::vector<uint16_t> h(n);
  uint32_t s = seed * 2654435761u + 12345u;
  for (size_t i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; h[i] = f2bf(scale * (((s >> 8) & 0xffff) / 32768.f - 1.f)); }
  uint16_t* d = (uint16_t*)dalloc(n * 2);
  hipMemcpy(d, h.data(), n * 2, hipMemcpyHostToDevice);
  return d;
}
static float* rand_f32(size_t n, uint32_t seed) {
  std::vector<float> h(n);
  uint32_t s = seed * 2654435761u + 777u;
  for (size_t i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; h[i] = ((s >> 8) & 0xffff) / 32768.f - 1.f; }
  float* d = (float*)dalloc(n * 4);
  hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
  return d;
}

struct Case { const char* name; int ta, tb, M, N, K; int epi; bool bias, res, acc; int split; bool ws = false, cs = false; };
static void* g_ws = nullptr; static size_t g_ws_bytes = 0; static float* g_cs = nullptr;

static int run(const Case& c, int force, void* A, void* B, void* C, float* bias, void* res, void* aux) {
  dh_gemm_args g; memset(&g, 0, sizeof(g));
  g.dtype = DH_BF16; g.c_dtype = c.acc ? DH_F32 : DH_BF16;
  g.a_kmajor = c.ta; g.b_kmajor = c.tb; g.M = c.M; g.N = c.N; g.K = c.K;
  g.A = A; g.lda = c.ta ? c.M : c.K; g.B = B; g.ldb = c.tb ? c.N : c.K;
  g.C = C; g.ldc = c.N; g.bias = c.bias ? bias : nullptr; g.epilogue = c.epi;
  g.residual = c.res ? res : nullptr; g.ldr = c.N; g.aux = c.epi ? aux : nullptr; g.ldaux = c.N;
  g.accumulate = c.acc; g.split_k = c.split; g.alpha = 1.f; g.force_generic = force; g.pad_ok = 0;
  if ((c.ws || !c.acc) && force != 3) { g.ws = g_ws; g.ws_bytes = (int64_t)g_ws_bytes; }
  if (c.cs) g.a_colsum = g_cs;
  int rc = dh_gemm(&g, nullptr);
  if (rc != DH_OK) printf("  dh_gemm(force=%d) failed: %s\n", force, dh_last_error());
  return rc;
}

static double max_err(const Case& c, void* C1, void* C2, double* ref_max) {
  size_t n = (size_t)c.M * c.N;
  double e = 0, rm = 0;
  if (c.acc) {
    std::vector<float> a(n), b(n);
    hipMemcpy(a.data(), C1, n * 4, hipMemcpyDeviceToHost); hipMemcpy(b.data(), C2, n * 4, hipMemcpyDeviceToHost);
    for (size_t i = 0; i < n; ++i) { e = fmax(e, fabs((double)a[i] - b[i])); rm = fmax(rm, fabs((double)b[i])); }
  } else {
    std::vector<uint16_t> a(n), b(n);
    hipMemcpy(a.data(), C1, n * 2, hipMemcpyDeviceToHost); hipMemcpy(b.data(), C2, n * 2, hipMemcpyDeviceToHost);
    for (size_t i = 0; i < n; ++i) { e = fmax(e, fabs((double)bf2f(a[i]) - bf2f(b[i]))); rm = fmax(rm, fabs((double)bf2f(b[i]))); }
  }
  *ref_max = rm;
  return e;
}

static float timeit(const Case& c, int force, void* A, void* B, void* C, float* bias, void* res, void* aux, int iters) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 2; ++i) if (run(c, force, A, B, C, bias, res, aux)) return -1.f;
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < iters; ++i) run(c, force, A, B, C, bias, res, aux);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms / iters;
}

int main(int argc, char** argv) {
  int iters = argc > 1 ? atoi(argv[1]) : 20;
  bool quick = argc > 2 && !strcmp(argv[2], "quick");
  // ---------------- correctness: v4 vs the VALU kernel on ragged problems, all layouts / epilogues
  g_ws_bytes = 512u << 20; g_ws = dalloc(g_ws_bytes); g_cs = (float*)dalloc(8192 * 4);
  std::vector<Case> chk = {
      {"chk.NT", 0, 0, 512, 512, 256, 0, true, false, false, 1},       {"chk.NT.gelu", 0, 0, 512, 768, 192, 1, true, false, false, 1},
      {"chk.NT.res", 0, 0, 256, 256, 128, 0, true, true, false, 1},    {"chk.NN.dgelu", 0, 1, 768, 512, 320, 2, false, false, false, 1},
      {"chk.NN", 0, 1, 1024, 512, 128, 0, false, false, false, 1},      {"chk.NT.big", 0, 0, 2560, 1280, 704, 0, true, false, false, 1},
      {"chk.NT.res.big", 0, 0, 2816, 768, 512, 0, true, true, false, 1}, {"chk.NT.gelu.big", 0, 0, 2048, 2048, 512, 1, true, false, false, 1},
      {"chk.TT.acc", 1, 1, 512, 768, 1024, 0, false, false, true, 3},  {"chk.TT.acc1", 1, 1, 256, 512, 128, 0, false, false, true, 1},
      {"chk.TT.ws", 1, 1, 512, 768, 1024, 0, false, false, true, 3, true, false}, {"chk.TT.ws.cs", 1, 1, 768, 512, 4096, 0, false, false, true, 5, true, true},
      {"chk.TT.cs", 1, 1, 512, 1024, 1024, 0, false, false, true, 2, false, true},
  };
  int bad = 0;
  for (auto& c : chk) {
    size_t an = (size_t)c.M * c.K, bn = (size_t)c.N * c.K, cn = (size_t)c.M * c.N;
    uint16_t* A = rand_bf16(an, 1); uint16_t* B = rand_bf16(bn, 2, 0.25f);
    float* bias = rand_f32(c.N, 3); uint16_t* res = rand_bf16(cn, 4); uint16_t* aux_in = rand_bf16(cn, 5);
    size_t cb = cn * (c.acc ? 4 : 2);
    void *C1 = dalloc(cb), *C2 = dalloc(cb), *X1 = dalloc(cn * 2), *X2 = dalloc(cn * 2);
    hipMemset(C1, 0, cb); hipMemset(C2, 0, cb);
    if (c.epi == 2) { hipMemcpy(X1, aux_in, cn * 2, hipMemcpyDeviceToDevice); hipMemcpy(X2, aux_in, cn * 2, hipMemcpyDeviceToDevice); }
    std::vector<float> cs1(c.M, 0.f), cs2(c.M, 0.f);
    hipMemset(g_cs, 0, 8192 * 4);
    int rc1 = run(c, 4, A, B, C1, bias, res, X1);
    hipDeviceSynchronize();
    if (c.cs) { hipMemcpy(cs1.data(), g_cs, c.M * 4, hipMemcpyDeviceToHost); hipMemset(g_cs, 0, 8192 * 4); }
    int rc2 = run(c, 1, A, B, C2, bias, res, X2);
    hipDeviceSynchronize();
    if (c.cs) {
      hipMemcpy(cs2.data(), g_cs, c.M * 4, hipMemcpyDeviceToHost);
      double ce = 0, cm = 0;
      for (int i = 0; i < c.M; ++i) { ce = fmax(ce, fabs((double)cs1[i] - cs2[i])); cm = fmax(cm, fabs((double)cs2[i])); }
      printf("  colsum err %.3g (max %.3g)%s\n", ce, cm, ce <= 2e-3 * fmax(cm, 1.0) ? "" : "  <-- FAIL");
      if (ce > 2e-3 * fmax(cm, 1.0)) ++bad;
    }
    double rm, e = max_err(c, C1, C2, &rm);
    double tol = (c.acc ? 2e-3 : 1.6e-2) * fmax(rm, 1.0);
    bool ok = rc1 == 0 && rc2 == 0 && e <= tol;
    if (c.epi == 1 && ok) { Case cc = c; cc.acc = false; double rm2, e2 = max_err(cc, X1, X2, &rm2); ok = e2 <= 1.6e-2 * fmax(rm2, 1.0); printf("  aux err %.3g (max %.3g)\n", e2, rm2); }
    printf("%-14s M=%d N=%d K=%d: max err %.4g (ref max %.4g) %s\n", c.name, c.M, c.N, c.K, e, rm, ok ? "OK" : "FAIL");
    bad += !ok;
    hipFree(A); hipFree(B); hipFree(bias); hipFree(res); hipFree(aux_in); hipFree(C1); hipFree(C2); hipFree(X1); hipFree(X2);
  }
  // repeated runs of one shape: race screen (results must be bit-identical run to run)
  {
    Case c = {"race", 0, 0, 2048, 1024, 768, 0, true, false, false, 1};
    uint16_t* A = rand_bf16((size_t)c.M * c.K, 11); uint16_t* B = rand_bf16((size_t)c.N * c.K, 12, 0.25f);
    float* bias = rand_f32(c.N, 3);
    size_t cn = (size_t)c.M * c.N;
    void *C1 = dalloc(cn * 2), *C2 = dalloc(cn * 2);
    run(c, 4, A, B, C1, bias, nullptr, nullptr);
    int diff = 0;
    for (int r = 0; r < 20; ++r) {
      run(c, 4, A, B, C2, bias, nullptr, nullptr);
      hipDeviceSynchronize();
      double rm, e = max_err(c, C1, C2, &rm);
      if (e != 0) ++diff;
    }
    run(c, 3, A, B, C2, bias, nullptr, nullptr); hipDeviceSynchronize();
    double rm, e = max_err(c, C1, C2, &rm);
    printf("race screen: %d / 20 runs differ; v4 vs v2 max err %.4g (max %.4g)\n", diff, e, rm);
    bad += diff != 0;
    hipFree(A); hipFree(B); hipFree(bias); hipFree(C1); hipFree(C2);
  }
  printf("correctness: %s\n", bad ? "FAILURES" : "all OK");

  // ---------------- timing on the tower shapes (b = 512)
  std::vector<Case> cases;
  struct Tw { const char* n; int rows, d; } tws[2] = {{"vis", 512 * 50, 768}, {"txt", 512 * 77, 512}};
  static char names[64][32]; int ni = 0;
  for (auto& tw : tws) {
    struct L { const char* n; int out, in; int epi; bool res; } ls[4] = {{"qkv", 3 * tw.d, tw.d, 0, false}, {"out", tw.d, tw.d, 0, true}, {"fc", 4 * tw.d, tw.d, 1, false}, {"proj", tw.d, 4 * tw.d, 0, true}};
    for (auto& l : ls) {
      snprintf(names[ni], 32, "%s.%s.fwd", tw.n, l.n); cases.push_back({names[ni++], 0, 0, tw.rows, l.out, l.in, l.epi, true, l.res, false, 1});
      snprintf(names[ni], 32, "%s.%s.dX", tw.n, l.n);  cases.push_back({names[ni++], 0, 1, tw.rows, l.in, l.out, (l.epi == 0 && !strcmp(l.n, "proj")) ? 2 : 0, false, false, false, 1});
      int tiles = ((l.out + 255) / 256) * ((l.in + 255) / 256);
      int split = 256 / tiles; if (split < 1) split = 1; if (split > tw.rows / 512) split = tw.rows / 512;
      snprintf(names[ni], 32, "%s.%s.dW", tw.n, l.n);  cases.push_back({names[ni++], 1, 1, l.out, l.in, tw.rows, 0, false, false, true, split, true, true});
    }
  }
  size_t maxA = 0, maxC = 0;
  for (auto& c : cases) { maxA = std::max(maxA, (size_t)c.M * c.K); maxA = std::max(maxA, (size_t)c.N * c.K); maxC = std::max(maxC, (size_t)c.M * c.N); }
  uint16_t* A = rand_bf16(maxA, 21); uint16_t* B = rand_bf16(maxA, 22, 0.05f);
  float* bias = rand_f32(8192, 3); uint16_t* res = rand_bf16(maxC, 4); uint16_t* aux = rand_bf16(maxC, 5);
  void *C1 = dalloc(maxC * 4), *C2 = dalloc(maxC * 4);
  // pass 1: v4 vs v2 agreement (host-side compare: kept out of the timing pass so the GPU never idles there)
  std::vector<double> errs, rms;
  for (auto& c : cases) {
    if (quick && !(strstr(c.name, "vis.fc") || strstr(c.name, "txt.out"))) { errs.push_back(0); rms.push_back(0); continue; }
    hipMemset(C1, 0, maxC * 4); hipMemset(C2, 0, maxC * 4);
    run(c, 3, A, B, C1, bias, res, aux); run(c, 4, A, B, C2, bias, res, aux); hipDeviceSynchronize();
    double rm, e = max_err(c, C2, C1, &rm);
    errs.push_back(e); rms.push_back(rm);
  }
  {  // clock warm-up: ~1 s of back-to-back GEMMs right before the timing pass
    Case w = cases[0];
    for (int i = 0; i < 4000; ++i) run(w, 4, A, B, C1, bias, res, aux);
    hipDeviceSynchronize();
  }
  printf("%-14s %-2s %6s %6s %6s ep sk |   v2 ms    TF |   v4 ms    TF | v4/v2 | max err v4 vs v2\n", "case", "ly", "M", "N", "K");
  double tot2 = 0, tot4 = 0, flops = 0;
  int ci = -1;
  for (auto& c : cases) {
    ++ci;
    if (quick && !(strstr(c.name, "vis.fc") || strstr(c.name, "txt.out"))) continue;
    float t2 = timeit(c, 3, A, B, C1, bias, res, aux, iters);
    float t4 = timeit(c, 4, A, B, C2, bias, res, aux, iters);
    double fl = 2.0 * c.M * c.N * c.K;
    printf("%-14s %c%c %6d %6d %6d %d  %2d | %7.3f %5.0f | %7.3f %5.0f | %5.2f | %.3g (max %.3g)\n", c.name, c.ta ? 'T' : 'N', c.tb ? 'T' : 'N', c.M, c.N, c.K,
           c.epi, c.split, t2, fl / t2 / 1e9, t4, fl / t4 / 1e9, t2 / t4, errs[ci], rms[ci]);
    tot2 += t2; tot4 += t4; flops += fl;
  }
  printf("sum: v2 %.3f ms (%.0f TF avg)  v4 %.3f ms (%.0f TF avg)\n", tot2, flops / tot2 / 1e9, tot4, flops / tot4 / 1e9);
  printf("status: %s\n", hipGetErrorString(hipDeviceSynchronize()));
  return bad;
}
