// Communicator context of the data-parallel step: RCCL over xGMI behind the C-ABI.
//
// What it replaces: the reference reaches its collectives through the `linklink` Python facade (linklink/__init__.py:13-71):
//   * AllGather.forward / backward (model/clip.py:25-49): one all_gather per feature tensor, backward = all-reduce of the whole
//     [B, D] gradient + slice of the own rows;
//   * DistModule's per-parameter gradient all-reduce hooks (utils/dist.py:49-88).
// Here a context owns the RCCL communicator, a communication stream and the events that order it against the caller's compute
// streams, and exposes the three collectives of the step in the shape the engine uses them:
//   dh_allgather_packed      n feature tensors [rows, cols_k] -> ONE all-gather of [rows, sum cols] rows (rank-major result);
//   dh_reducescatter_packed  the backward of it: ONE reduce-scatter(SUM) of the [world * rows, sum cols] gradient, split back into
//                            the n per-tensor gradients (world x less traffic than the reference's all-reduce + slice, same sum);
//   dh_allreduce_bucket      SUM all-reduce of one contiguous bucket of the flat fp32 gradient buffer, optionally crossing the
//                            links as bf16 (cast -> all-reduce -> cast back, all on the communication stream).
// Every call only ENQUEUES: the communication stream first waits for an event recorded on the caller's stream (the producer of
// the operands), the caller orders consumers with dh_comm_wait(ctx, stream).  No host synchronisation anywhere.
//
// RCCL is bound at run time (dlopen of librccl.so, the copy the process already holds if PyTorch loaded one): the library itself
// has no link-time dependency on it and single-GPU users never load it.  No fallback: a missing symbol or any RCCL error is a
// failure of the call with the RCCL message in dh_last_error().
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <string.h>

#include <string>

#include "dh_common.h"

namespace {

struct Rccl {
  void* lib = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  decltype(&ncclAllGather) AllGather = nullptr;
  decltype(&ncclReduceScatter) ReduceScatter = nullptr;
  decltype(&ncclAllReduce) AllReduce = nullptr;
};
Rccl g_rccl;

// returns nullptr and sets the error message when RCCL cannot be bound
Rccl* rccl() {
  if (g_rccl.lib) return &g_rccl;
  const char* names[] = {getenv("DH_RCCL_LIB"), "librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
  void* h = nullptr;
  std::string tried;
  for (const char* n : names) {
    if (!n || !*n) continue;
    h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (h) break;
    tried += std::string(" ") + n;
  }
  if (!h) { dh_set_error("dh_init: cannot load RCCL (tried%s): %s", tried.c_str(), dlerror()); return nullptr; }
#define DH_BIND(field, sym)                                                              \
  g_rccl.field = reinterpret_cast<decltype(g_rccl.field)>(dlsym(h, sym));                \
  if (!g_rccl.field) { dh_set_error("dh_init: RCCL has no symbol %s", sym); return nullptr; }
  DH_BIND(GetUniqueId, "ncclGetUniqueId")
  DH_BIND(CommInitRank, "ncclCommInitRank")
  DH_BIND(CommDestroy, "ncclCommDestroy")
  DH_BIND(GetErrorString, "ncclGetErrorString")
  DH_BIND(AllGather, "ncclAllGather")
  DH_BIND(ReduceScatter, "ncclReduceScatter")
  DH_BIND(AllReduce, "ncclAllReduce")
#undef DH_BIND
  g_rccl.lib = h;
  return &g_rccl;
}

}  // namespace

struct dh_ctx {
  int rank, world, device;
  ncclComm_t comm;
  hipStream_t stream;        // the communication stream
  hipEvent_t ev_in, ev_out;  // producer stream -> comm stream, comm stream -> consumer stream
};

#define DH_HIP(call)                                                                                                   \
  do {                                                                                                                 \
    hipError_t e__ = (call);                                                                                           \
    if (e__ != hipSuccess) DH_FAIL(DH_ERR_LAUNCH, "%s:%d %s: %s", __FILE__, __LINE__, #call, hipGetErrorString(e__));    \
  } while (0)
#define DH_NCCL(call)                                                                                                  \
  do {                                                                                                                 \
    ncclResult_t r__ = (call);                                                                                         \
    if (r__ != ncclSuccess) DH_FAIL(DH_ERR_LAUNCH, "%s:%d %s: %s", __FILE__, __LINE__, #call, g_rccl.GetErrorString(r__)); \
  } while (0)

// ---------------------------------------------------------------------------------------------------------------- kernels
// pack / unpack between n row-major tensors [rows, cols_k] and one [rows, sum cols] buffer; 16-byte units (cols_k * elem % 16 == 0)
struct PackTable {
  const void* src[DH_COMM_MAX_TENSORS];
  void* dst[DH_COMM_MAX_TENSORS];
  int units[DH_COMM_MAX_TENSORS];       // 16-byte units per row of tensor k
  int first[DH_COMM_MAX_TENSORS + 1];   // first unit of tensor k inside a packed row
  int n;
};

__global__ __launch_bounds__(256) void pack_rows_kernel(PackTable t, uint4* __restrict__ packed, long rows, int row_units) {
  const long total = rows * row_units;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long r = i / row_units;
    const int u = (int)(i - r * row_units);
    int k = 0;
    while (k + 1 < t.n && u >= t.first[k + 1]) ++k;
    packed[i] = reinterpret_cast<const uint4*>(t.src[k])[r * t.units[k] + (u - t.first[k])];
  }
}
__global__ __launch_bounds__(256) void unpack_rows_kernel(PackTable t, const uint4* __restrict__ packed, long rows, int row_units) {
  const long total = rows * row_units;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long r = i / row_units;
    const int u = (int)(i - r * row_units);
    int k = 0;
    while (k + 1 < t.n && u >= t.first[k + 1]) ++k;
    reinterpret_cast<uint4*>(t.dst[k])[r * t.units[k] + (u - t.first[k])] = packed[i];
  }
}
__global__ __launch_bounds__(256) void f32_to_bf16_kernel(const float* __restrict__ x, bf16_t* __restrict__ y, long n) {
  // n is a multiple of 8 for every bucket the engine hands over (parameter slots are padded); the tail loop covers the rest
  const long n8 = n / 8;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n8; i += (long)gridDim.x * 256) {
    float v[8];
    ld8(x + 8 * i, v);
    st8_hw(y + 8 * i, v);
  }
  if (blockIdx.x == 0)
    for (long i = n8 * 8 + threadIdx.x; i < n; i += 256) y[i] = f2bf(x[i]);
}
__global__ __launch_bounds__(256) void bf16_to_f32_kernel(const bf16_t* __restrict__ x, float* __restrict__ y, long n) {
  const long n8 = n / 8;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n8; i += (long)gridDim.x * 256) {
    float v[8];
    ld8(x + 8 * i, v);
    *reinterpret_cast<float4*>(y + 8 * i) = make_float4(v[0], v[1], v[2], v[3]);
    *reinterpret_cast<float4*>(y + 8 * i + 4) = make_float4(v[4], v[5], v[6], v[7]);
  }
  if (blockIdx.x == 0)
    for (long i = n8 * 8 + threadIdx.x; i < n; i += 256) y[i] = bf2f(x[i]);
}

static int grid_for(long items) {
  long g = (items + 255) / 256;
  return (int)(g < 1 ? 1 : (g > 2048 ? 2048 : g));
}

static int make_table(PackTable& t, const void* const* cptrs, void* const* ptrs, const int* cols, int n, int elem_bytes, const char* who) {
  DH_REQUIRE(n >= 1 && n <= DH_COMM_MAX_TENSORS, "%s: %d tensors (1..%d)", who, n, DH_COMM_MAX_TENSORS);
  DH_REQUIRE(elem_bytes == 2 || elem_bytes == 4, "%s: element size %d (2 = bf16, 4 = fp32)", who, elem_bytes);
  t.n = n;
  t.first[0] = 0;
  for (int k = 0; k < n; ++k) {
    const long bytes = (long)cols[k] * elem_bytes;
    DH_REQUIRE(cols[k] > 0 && bytes % 16 == 0, "%s: tensor %d has %d columns: rows must be multiples of 16 bytes", who, k, cols[k]);
    const void* p = cptrs ? cptrs[k] : ptrs[k];
    DH_REQUIRE(p != nullptr && ((uintptr_t)p & 15) == 0, "%s: tensor %d is null or not 16-byte aligned", who, k);
    t.src[k] = cptrs ? cptrs[k] : nullptr;
    t.dst[k] = ptrs ? ptrs[k] : nullptr;
    t.units[k] = (int)(bytes / 16);
    t.first[k + 1] = t.first[k] + t.units[k];
  }
  return DH_OK;
}

// ---------------------------------------------------------------------------------------------------------------- C-ABI
extern "C" {

int dh_comm_unique_id(void* id_out, int64_t bytes) {
  DH_REQUIRE(id_out && bytes >= (int64_t)sizeof(ncclUniqueId), "dh_comm_unique_id: buffer of %lld bytes, need %zu", (long long)bytes,
             sizeof(ncclUniqueId));
  Rccl* r = rccl();
  if (!r) return DH_ERR_UNSUPPORTED;
  ncclUniqueId id;
  DH_NCCL(r->GetUniqueId(&id));
  memcpy(id_out, &id, sizeof(id));
  return DH_OK;
}

dh_ctx* dh_init(int rank, int world, int local_rank, const void* nccl_unique_id) {
  if (!(world >= 1 && rank >= 0 && rank < world && local_rank >= 0 && nccl_unique_id)) {
    dh_set_error("dh_init: rank %d of %d, local rank %d, id %p", rank, world, local_rank, nccl_unique_id);
    return nullptr;
  }
  Rccl* r = rccl();
  if (!r) return nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || local_rank >= ndev) {
    dh_set_error("dh_init: local rank %d but %d visible devices", local_rank, ndev);
    return nullptr;
  }
  if (hipSetDevice(local_rank) != hipSuccess) { dh_set_error("dh_init: hipSetDevice(%d) failed", local_rank); return nullptr; }
  dh_ctx* c = new dh_ctx();
  c->rank = rank; c->world = world; c->device = local_rank; c->comm = nullptr; c->stream = nullptr; c->ev_in = c->ev_out = nullptr;
  ncclUniqueId id;
  memcpy(&id, nccl_unique_id, sizeof(id));
  ncclResult_t nr = r->CommInitRank(&c->comm, world, id, rank);
  if (nr != ncclSuccess) {
    dh_set_error("dh_init: ncclCommInitRank(rank %d of %d): %s", rank, world, r->GetErrorString(nr));
    delete c;
    return nullptr;
  }
  if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess ||
      hipEventCreateWithFlags(&c->ev_in, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&c->ev_out, hipEventDisableTiming) != hipSuccess) {
    dh_set_error("dh_init: cannot create the communication stream / events");
    r->CommDestroy(c->comm);
    delete c;
    return nullptr;
  }
  (void)hipEventRecord(c->ev_out, c->stream);     // dh_comm_wait before any collective is a no-op, not an error
  return c;
}

int dh_finalize(dh_ctx* c) {
  DH_REQUIRE(c, "dh_finalize: null context");
  DH_HIP(hipStreamSynchronize(c->stream));
  DH_NCCL(g_rccl.CommDestroy(c->comm));
  (void)hipEventDestroy(c->ev_in);
  (void)hipEventDestroy(c->ev_out);
  (void)hipStreamDestroy(c->stream);
  delete c;
  return DH_OK;
}

int dh_ctx_info(const dh_ctx* c, int* out3) {
  DH_REQUIRE(c && out3, "dh_ctx_info: null argument");
  out3[0] = c->rank; out3[1] = c->world; out3[2] = c->device;
  return DH_OK;
}

void* dh_comm_stream(const dh_ctx* c) { return c ? (void*)c->stream : nullptr; }

int dh_comm_wait(dh_ctx* c, dh_stream_t stream) {
  DH_REQUIRE(c, "dh_comm_wait: null context");
  DH_HIP(hipStreamWaitEvent((hipStream_t)stream, c->ev_out, 0));
  return DH_OK;
}

static int comm_enter(dh_ctx* c, dh_stream_t producer) {       // the communication stream waits for what `producer` holds now
  DH_HIP(hipEventRecord(c->ev_in, (hipStream_t)producer));
  DH_HIP(hipStreamWaitEvent(c->stream, c->ev_in, 0));
  return DH_OK;
}
static int comm_leave(dh_ctx* c) {
  DH_HIP(hipEventRecord(c->ev_out, c->stream));
  return DH_OK;
}

int dh_allgather_packed(dh_ctx* c, const void* const* src, const int* cols, int n, int rows, int elem_bytes, void* gathered,
                        dh_stream_t stream) {
  DH_REQUIRE(c && src && cols && gathered && rows > 0, "dh_allgather_packed: bad args");
  PackTable t;
  int rc = make_table(t, src, nullptr, cols, n, elem_bytes, "dh_allgather_packed");
  if (rc) return rc;
  DH_REQUIRE(((uintptr_t)gathered & 15) == 0, "dh_allgather_packed: gathered buffer not 16-byte aligned");
  const int row_units = t.first[n];
  const size_t slot = (size_t)rows * row_units * 16;
  char* own = (char*)gathered + slot * c->rank;                 // in-place all-gather: the own rows are packed into the own slot
  if ((rc = comm_enter(c, stream))) return rc;
  hipLaunchKernelGGL(pack_rows_kernel, dim3(grid_for((long)rows * row_units)), dim3(256), 0, c->stream, t, (uint4*)own, (long)rows, row_units);
  DH_CHECK_LAUNCH();
  DH_NCCL(g_rccl.AllGather(own, gathered, slot, ncclUint8, c->comm, c->stream));
  return comm_leave(c);
}

int dh_reducescatter_packed(dh_ctx* c, const void* grad_gathered, void* const* dst, const int* cols, int n, int rows, int elem_bytes,
                            void* scratch, dh_stream_t stream) {
  DH_REQUIRE(c && grad_gathered && dst && cols && scratch && rows > 0, "dh_reducescatter_packed: bad args");
  PackTable t;
  int rc = make_table(t, nullptr, dst, cols, n, elem_bytes, "dh_reducescatter_packed");
  if (rc) return rc;
  DH_REQUIRE(((uintptr_t)scratch & 15) == 0 && ((uintptr_t)grad_gathered & 15) == 0, "dh_reducescatter_packed: buffers not 16-byte aligned");
  const int row_units = t.first[n];
  const size_t count = (size_t)rows * row_units * 16 / elem_bytes;   // elements per rank
  if ((rc = comm_enter(c, stream))) return rc;
  DH_NCCL(g_rccl.ReduceScatter(grad_gathered, scratch, count, elem_bytes == 4 ? ncclFloat32 : ncclBfloat16, ncclSum, c->comm, c->stream));
  hipLaunchKernelGGL(unpack_rows_kernel, dim3(grid_for((long)rows * row_units)), dim3(256), 0, c->stream, t, (const uint4*)scratch, (long)rows,
                     row_units);
  DH_CHECK_LAUNCH();
  return comm_leave(c);
}

int dh_allreduce_bucket(dh_ctx* c, float* grad, int64_t n, void* bf16_stage, dh_stream_t stream) {
  DH_REQUIRE(c && grad && n > 0, "dh_allreduce_bucket: bad args");
  DH_REQUIRE(!bf16_stage || (((uintptr_t)grad & 15) == 0 && ((uintptr_t)bf16_stage & 15) == 0), "dh_allreduce_bucket: buffers not 16-byte aligned");
  int rc = comm_enter(c, stream);
  if (rc) return rc;
  if (bf16_stage) {
    hipLaunchKernelGGL(f32_to_bf16_kernel, dim3(grid_for(n / 8)), dim3(256), 0, c->stream, grad, (bf16_t*)bf16_stage, (long)n);
    DH_CHECK_LAUNCH();
    DH_NCCL(g_rccl.AllReduce(bf16_stage, bf16_stage, (size_t)n, ncclBfloat16, ncclSum, c->comm, c->stream));
    hipLaunchKernelGGL(bf16_to_f32_kernel, dim3(grid_for(n / 8)), dim3(256), 0, c->stream, (const bf16_t*)bf16_stage, grad, (long)n);
    DH_CHECK_LAUNCH();
  } else {
    DH_NCCL(g_rccl.AllReduce(grad, grad, (size_t)n, ncclFloat32, ncclSum, c->comm, c->stream));
  }
  return comm_leave(c);
}

}  // extern "C"
