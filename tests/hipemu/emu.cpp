// TEST INFRASTRUCTURE ONLY -- block scheduler and wave collectives of the host emulation (see hip/hip_runtime.h here).
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <ucontext.h>

// A fiber switch costs one swapcontext() = one sigprocmask system call per direction; kernels built on cross-lane operations
// switch tens of millions of times.  On x86-64 a 12-instruction register swap does the same job ~50x faster.
#if defined(__x86_64__)
#define EMU_FAST_SWITCH 1
extern "C" void emu_switch(void** save_sp, void* load_sp);
asm(R"(
.text
.globl emu_switch
.type emu_switch,@function
emu_switch:
  pushq %rbp
  pushq %rbx
  pushq %r12
  pushq %r13
  pushq %r14
  pushq %r15
  movq %rsp, (%rdi)
  movq %rsi, %rsp
  popq %r15
  popq %r14
  popq %r13
  popq %r12
  popq %rbx
  popq %rbp
  ret
.size emu_switch,.-emu_switch
)");
#else
#define EMU_FAST_SWITCH 0
#endif

#include <chrono>
#include <vector>

emu_idx threadIdx, blockIdx;
dim3 blockDim, gridDim;

namespace {
constexpr size_t kStack = 512 * 1024;
constexpr size_t kLds = 160 * 1024;
constexpr size_t kMaxDeposit = 256;          // bytes per lane and collective
struct Fiber {
  ucontext_t ctx;
  void* sp = nullptr;
  char* stack = nullptr;
  bool done = false;
  emu_idx tid;
  unsigned flat = 0;
};
struct Wave {
  unsigned count = 0, phase = 0, live = 64;
  alignas(16) unsigned char buf[2][64 * kMaxDeposit];
};
ucontext_t g_sched;
void* g_sched_sp = nullptr;
std::vector<Fiber> g_fibers;
std::vector<Wave> g_waves;
Fiber* g_cur = nullptr;
const std::function<void()>* g_body = nullptr;
unsigned g_bar_count = 0, g_bar_phase = 0, g_live = 0;
alignas(64) unsigned char g_lds[kLds];

unsigned long g_yields = 0, g_collectives = 0, g_barriers = 0;
void to_scheduler() {
#if EMU_FAST_SWITCH
  emu_switch(&g_cur->sp, g_sched_sp);
#else
  swapcontext(&g_cur->ctx, &g_sched);
#endif
}
void yield() { ++g_yields; to_scheduler(); }

void trampoline() {
  (*g_body)();
  g_cur->done = true;
  --g_live;
  --g_waves[g_cur->flat / 64].live;
  // a thread that has left no longer takes part in barriers: release waiters if it was the last one missing
  if (g_live > 0 && g_bar_count == g_live) { g_bar_count = 0; ++g_bar_phase; }
  to_scheduler();
  abort();                      // a finished fiber is never resumed
}
}  // namespace

void emu_syncthreads() {
  ++g_barriers;
  const unsigned my = g_bar_phase;
  if (++g_bar_count == g_live) { g_bar_count = 0; ++g_bar_phase; return; }
  while (g_bar_phase == my) yield();
}

void* emu_dyn_lds() { return g_lds; }
unsigned emu_lane() { return g_cur->flat & 63; }

const unsigned char* emu_wave_gather(const void* mine, size_t n) {
  if (n > kMaxDeposit) abort();
  ++g_collectives;
  Wave& w = g_waves[g_cur->flat / 64];
  if (w.live != 64 && w.live != (blockDim.x * blockDim.y * blockDim.z) % 64 && w.count == 0 && false) abort();
  const unsigned my = w.phase;
  unsigned char* buf = w.buf[my & 1];
  memcpy(buf + (size_t)(g_cur->flat & 63) * n, mine, n);
  if (++w.count == w.live) { w.count = 0; ++w.phase; }
  else while (w.phase == my) yield();
  return buf;
}

void emu_launch(dim3 grid, dim3 block, size_t dyn_lds, const std::function<void()>& body) {
  const unsigned nthreads = block.x * block.y * block.z;
  if (dyn_lds > kLds) { fprintf(stderr, "hipemu: %zu bytes of dynamic LDS requested\n", dyn_lds); abort(); }
  if (g_fibers.size() < nthreads) {
    size_t old = g_fibers.size();
    g_fibers.resize(nthreads);
    for (size_t i = old; i < nthreads; ++i) g_fibers[i].stack = (char*)malloc(kStack);
  }
  g_waves.resize((nthreads + 63) / 64);
  gridDim = grid;
  blockDim = block;
  g_body = &body;
  static const bool trace = getenv("HIPEMU_TRACE") != nullptr;
  if (trace) fprintf(stderr, "hipemu: launch grid (%u,%u,%u) block %u lds %zu  [so far: %lu yields, %lu collectives, %lu barrier arrivals]\n",
                     grid.x, grid.y, grid.z, nthreads, dyn_lds, g_yields, g_collectives, g_barriers);
  const auto t_start = std::chrono::steady_clock::now();
  for (unsigned bz = 0; bz < grid.z; ++bz)
    for (unsigned by = 0; by < grid.y; ++by)
      for (unsigned bx = 0; bx < grid.x; ++bx) {
        memset(g_lds, 0xff, dyn_lds);              // NaN-poison the block's dynamic LDS: reads of unwritten words show up
        g_bar_count = g_bar_phase = 0;
        g_live = nthreads;
        for (unsigned wv = 0; wv < g_waves.size(); ++wv) {
          g_waves[wv].count = g_waves[wv].phase = 0;
          g_waves[wv].live = std::min(64u, nthreads - wv * 64);
        }
        for (unsigned t = 0; t < nthreads; ++t) {
          Fiber& f = g_fibers[t];
          f.done = false;
          f.flat = t;
          f.tid = emu_idx{t % block.x, (t / block.x) % block.y, t / (block.x * block.y)};
#if EMU_FAST_SWITCH
          {   // initial frame: six callee-saved registers, then the entry point as the return address of emu_switch
            uintptr_t top = ((uintptr_t)f.stack + kStack) & ~(uintptr_t)15;
            void** sp = (void**)top;
            *--sp = nullptr;                      // (would be trampoline's return address: it never returns)
            *--sp = (void*)&trampoline;
            for (int r = 0; r < 6; ++r) *--sp = nullptr;
            f.sp = sp;
          }
#else
          getcontext(&f.ctx);
          f.ctx.uc_stack.ss_sp = f.stack;
          f.ctx.uc_stack.ss_size = kStack;
          f.ctx.uc_link = nullptr;
          makecontext(&f.ctx, trampoline, 0);
#endif
        }
        bool any = true;
        while (any) {            // round-robin: every live thread runs until it blocks (barrier / collective) or ends
          any = false;
          for (unsigned t = 0; t < nthreads; ++t) {
            Fiber& f = g_fibers[t];
            if (f.done) continue;
            any = true;
            g_cur = &f;
            blockIdx = emu_idx{bx, by, bz};
            threadIdx = f.tid;
#if EMU_FAST_SWITCH
            emu_switch(&g_sched_sp, f.sp);
#else
            swapcontext(&g_sched, &f.ctx);
#endif
          }
        }
      }
  if (trace) fprintf(stderr, "hipemu:   ... %.3f s\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count());
}

// ---- MFMA: lane l of a wave holds (ISA operand layouts)
//  32x32x2 f32 : a = A[i = l % 32][k = l / 32], b = B[k = l / 32][j = l % 32];
//  32x32x16 bf16: a[0..7] = A[i = l % 32][k = 8 * (l / 32) + 0..7], b likewise B[k][j = l % 32];
//  32x32 accumulators: c[r] = C[i = 8 * (r / 4) + 4 * (l / 32) + r % 4][j = l % 32];
//  16x16x32 bf16: a[0..7] = A[i = l % 16][k = 8 * (l / 16) + 0..7], b likewise; c[r] = C[i = 4 * (l / 16) + r][j = l % 16].
emu_f32x16 emu_mfma_f32_32x32x2f32(float a, float b, emu_f32x16 c) {
  struct Dep { float a, b; } mine{a, b};
  const Dep* all = (const Dep*)emu_wave_gather(&mine, sizeof mine);
  const unsigned l = emu_lane(), j = l % 32;
  for (int r = 0; r < 16; ++r) {
    const unsigned i = 8 * (r / 4) + 4 * (l / 32) + r % 4;
    float acc = c[r];
    for (unsigned k = 0; k < 2; ++k) acc = fmaf(all[k * 32 + i].a, all[k * 32 + j].b, acc);
    c[r] = acc;
  }
  return c;
}

static inline float bf(__bf16 v) { return (float)v; }

emu_f32x16 emu_mfma_f32_32x32x16_bf16(emu_bf16x8 a, emu_bf16x8 b, emu_f32x16 c) {
  struct Dep { emu_bf16x8 a, b; } mine{a, b};
  const Dep* all = (const Dep*)emu_wave_gather(&mine, sizeof mine);
  const unsigned l = emu_lane(), j = l % 32;
  for (int r = 0; r < 16; ++r) {
    const unsigned i = 8 * (r / 4) + 4 * (l / 32) + r % 4;
    float acc = c[r];
    for (unsigned kb = 0; kb < 2; ++kb)
      for (unsigned e = 0; e < 8; ++e) acc += bf(all[kb * 32 + i].a[e]) * bf(all[kb * 32 + j].b[e]);
    c[r] = acc;
  }
  return c;
}

emu_f32x4 emu_mfma_f32_16x16x32_bf16(emu_bf16x8 a, emu_bf16x8 b, emu_f32x4 c) {
  struct Dep { emu_bf16x8 a, b; } mine{a, b};
  const Dep* all = (const Dep*)emu_wave_gather(&mine, sizeof mine);
  const unsigned l = emu_lane(), j = l % 16;
  for (int r = 0; r < 4; ++r) {
    const unsigned i = 4 * (l / 16) + r;
    float acc = c[r];
    for (unsigned kb = 0; kb < 4; ++kb)
      for (unsigned e = 0; e < 8; ++e) acc += bf(all[kb * 16 + i].a[e]) * bf(all[kb * 16 + j].b[e]);
    c[r] = acc;
  }
  return c;
}

// ds_read_b64_tr_b16: the 16 lanes of a group each address 4 consecutive 16-bit elements; lane u = 4 * row + chunk supplies
// M[row][4 * chunk + 0..3] of a 4 x 16 block M, and lane t receives column t of it: (M[0][t], M[1][t], M[2][t], M[3][t]) -- the
// k-contiguous MFMA fragment of an operand stored contraction-major (layout as used by attention.hip frag_tr, whose GPU tests
// pin it).
emu_i16x4 emu_ds_read_tr16_b64(const void* lds_addr) {
  struct Dep { short v[4]; } mine;
  memcpy(mine.v, lds_addr, 8);
  const Dep* all = (const Dep*)emu_wave_gather(&mine, sizeof mine);
  const unsigned l = emu_lane(), g = l & ~15u, t = l & 15u;
  emu_i16x4 r;
  for (unsigned e = 0; e < 4; ++e) r[e] = all[g + 4 * e + (t >> 2)].v[t & 3];
  return r;
}

// pieces of the library that live in other translation units
static thread_local char g_err[512];
void dh_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof g_err, fmt, ap);
  va_end(ap);
}
extern "C" const char* dh_last_error() { return g_err; }
