// TEST INFRASTRUCTURE ONLY -- block scheduler of the host emulation (see hip/hip_runtime.h in this directory).
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <ucontext.h>

#include <vector>

emu_idx threadIdx, blockIdx;
dim3 blockDim, gridDim;

namespace {
constexpr size_t kStack = 256 * 1024;
struct Fiber {
  ucontext_t ctx;
  char* stack = nullptr;
  bool done = false;
  emu_idx tid;
};
ucontext_t g_sched;
std::vector<Fiber> g_fibers;
Fiber* g_cur = nullptr;
const std::function<void()>* g_body = nullptr;

void trampoline() {
  (*g_body)();
  g_cur->done = true;
  swapcontext(&g_cur->ctx, &g_sched);
}
}  // namespace

void emu_syncthreads() { swapcontext(&g_cur->ctx, &g_sched); }

void emu_launch(dim3 grid, dim3 block, const std::function<void()>& body) {
  const unsigned nthreads = block.x * block.y * block.z;
  if (g_fibers.size() < nthreads) {
    size_t old = g_fibers.size();
    g_fibers.resize(nthreads);
    for (size_t i = old; i < nthreads; ++i) g_fibers[i].stack = (char*)malloc(kStack);
  }
  gridDim = grid;
  blockDim = block;
  g_body = &body;
  for (unsigned bz = 0; bz < grid.z; ++bz)
    for (unsigned by = 0; by < grid.y; ++by)
      for (unsigned bx = 0; bx < grid.x; ++bx) {
        for (unsigned t = 0; t < nthreads; ++t) {
          Fiber& f = g_fibers[t];
          f.done = false;
          f.tid = emu_idx{t % block.x, (t / block.x) % block.y, t / (block.x * block.y)};
          getcontext(&f.ctx);
          f.ctx.uc_stack.ss_sp = f.stack;
          f.ctx.uc_stack.ss_size = kStack;
          f.ctx.uc_link = nullptr;
          makecontext(&f.ctx, trampoline, 0);
        }
        bool any = true;
        while (any) {            // one pass = every live thread runs to its next barrier (or to the end)
          any = false;
          for (unsigned t = 0; t < nthreads; ++t) {
            Fiber& f = g_fibers[t];
            if (f.done) continue;
            any = true;
            g_cur = &f;
            blockIdx = emu_idx{bx, by, bz};
            threadIdx = f.tid;
            swapcontext(&g_sched, &f.ctx);
          }
        }
      }
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
