// Collectives of the element partition issued by the library itself: RCCL calls on the solver stream.
//
// The reference runs `mpirun -np N turtleFSI ...` with MPI reductions inside the solver [REF docs/simulation.md:14-32;
// src/vasp/simulations/simulation_common.py:217-220].  Rounds 1-2 left the wire to the host: every all-reduce was
// C -> Python callback -> H2D -> torch all_reduce -> .cpu(), every halo exchange ended in a device synchronise
// (vasp_amd/partition.py).  With `fsi_set_rccl` the library owns a communicator (one rank per GPU, xGMI underneath) and
// the Krylov loop queues  pack kernel -> ncclSend / ncclRecv group -> unpack kernel  and  ncclAllReduce on the coefficient
// vector in device memory -> update kernel  on ITS stream: the host waits once per Gram-Schmidt pass to read the numbers it
// needs for its bookkeeping, as in a single context, and never touches the payload.
//
// RCCL is resolved at run time (dlopen / dlsym) so that libvaspfsi.so has no link dependency on it and loads on a machine
// without ROCm's communication library (the CPU container of the test-suite); the copy PyTorch already mapped is preferred
// over a second one.  The torch.distributed transport (callbacks in FsiComm) stays the default and the tested fallback.
#include <dlfcn.h>

#include <cstdio>
#include <cstring>

#include "fsi_kernels.hpp"

namespace fsi {

namespace {

// the few declarations of rccl.h this file needs (ABI of NCCL 2.x / RCCL: stable since 2.0)
typedef struct { char internal[128]; } NcclUniqueId;
typedef void* NcclComm;
typedef int NcclResult;                    // 0 = ncclSuccess
constexpr int kNcclFloat64 = 8, kNcclSum = 0;

struct Api {
  NcclResult (*GetUniqueId)(NcclUniqueId*) = nullptr;
  NcclResult (*CommInitRank)(NcclComm*, int, NcclUniqueId, int) = nullptr;
  NcclResult (*CommDestroy)(NcclComm) = nullptr;
  NcclResult (*CommAbort)(NcclComm) = nullptr;
  NcclResult (*AllReduce)(const void*, void*, size_t, int, int, NcclComm, hipStream_t) = nullptr;
  NcclResult (*Send)(const void*, size_t, int, int, NcclComm, hipStream_t) = nullptr;
  NcclResult (*Recv)(void*, size_t, int, int, NcclComm, hipStream_t) = nullptr;
  NcclResult (*GroupStart)() = nullptr;
  NcclResult (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(NcclResult) = nullptr;
  bool ok = false;
  std::string where;
};

Api& api() {
  static Api a;
  static bool tried = false;
  if (tried) return a;
  tried = true;
  void* h = nullptr;
  for (const char* name : {"librccl.so", "librccl.so.1"}) {      // the copy that is already mapped (PyTorch's), if any
    h = dlopen(name, RTLD_NOW | RTLD_NOLOAD);
    if (h) { a.where = std::string(name) + " (already loaded)"; break; }
  }
  if (!h)
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (h) { a.where = name; break; }
    }
  if (!h) return a;
  auto sym = [&](const char* n) { return dlsym(h, n); };
  a.GetUniqueId = reinterpret_cast<decltype(a.GetUniqueId)>(sym("ncclGetUniqueId"));
  a.CommInitRank = reinterpret_cast<decltype(a.CommInitRank)>(sym("ncclCommInitRank"));
  a.CommDestroy = reinterpret_cast<decltype(a.CommDestroy)>(sym("ncclCommDestroy"));
  a.CommAbort = reinterpret_cast<decltype(a.CommAbort)>(sym("ncclCommAbort"));
  a.AllReduce = reinterpret_cast<decltype(a.AllReduce)>(sym("ncclAllReduce"));
  a.Send = reinterpret_cast<decltype(a.Send)>(sym("ncclSend"));
  a.Recv = reinterpret_cast<decltype(a.Recv)>(sym("ncclRecv"));
  a.GroupStart = reinterpret_cast<decltype(a.GroupStart)>(sym("ncclGroupStart"));
  a.GroupEnd = reinterpret_cast<decltype(a.GroupEnd)>(sym("ncclGroupEnd"));
  a.GetErrorString = reinterpret_cast<decltype(a.GetErrorString)>(sym("ncclGetErrorString"));
  a.ok = a.GetUniqueId && a.CommInitRank && a.CommDestroy && a.AllReduce && a.Send && a.Recv && a.GroupStart && a.GroupEnd;
  return a;
}

// A failed call leaves this rank outside a collective its peers may already be inside.  Returning alone would let them wait
// for ever, so the communicator is aborted: ncclCommAbort tears the rank's connections down, which the peers'
// pending operations see as a remote error instead of a silent stall, and every later call on this context fails at once
// (ctx->rccl_dead) - the run ends with FSI_ERR_DEVICE on every rank rather than hanging on some.
int fail(FsiCtx* ctx, const char* what, NcclResult r) {
  Api& a = api();
  ctx->err = std::string("RCCL: ") + what + ": " + (a.GetErrorString ? a.GetErrorString(r) : "error") + " (code " + std::to_string(r) +
             "); the communicator was aborted";
  if (ctx->rccl_comm && a.CommAbort) { (void)a.CommAbort(ctx->rccl_comm); ctx->rccl_comm = nullptr; }
  ctx->rccl_dead = true;
  return FSI_ERR_DEVICE;
}
int dead(FsiCtx* ctx) {
  if (ctx->err.empty()) ctx->err = "RCCL: the communicator of this context was aborted after an earlier error";
  return FSI_ERR_DEVICE;
}

}  // namespace

int rccl_unique_id(void* out128, std::string* err) {
  Api& a = api();
  if (!a.ok) { if (err) *err = "librccl.so could not be loaded (dlopen)"; return FSI_ERR_DEVICE; }
  NcclUniqueId id;
  const NcclResult r = a.GetUniqueId(&id);
  if (r != 0) { if (err) *err = std::string("ncclGetUniqueId: ") + (a.GetErrorString ? a.GetErrorString(r) : "error"); return FSI_ERR_DEVICE; }
  std::memcpy(out128, &id, sizeof id);
  return FSI_OK;
}

int rccl_init(FsiCtx* ctx, const void* id128, int rank, int world, const int64_t* send_counts, const int64_t* recv_counts) {
  Api& a = api();
  if (!a.ok) { ctx->err = "fsi_set_rccl: librccl.so could not be loaded (dlopen)"; return FSI_ERR_DEVICE; }
  if (!ctx->part) { ctx->err = "fsi_set_rccl: call fsi_set_partition first"; return FSI_ERR_INVALID; }
  int64_t ns = 0, nr = 0;
  for (int p = 0; p < world; ++p) {
    if (send_counts[p] < 0 || recv_counts[p] < 0 || (p == rank && (send_counts[p] || recv_counts[p]))) { ctx->err = "fsi_set_rccl: bad counts"; return FSI_ERR_INVALID; }
    ns += send_counts[p];
    nr += recv_counts[p];
  }
  if (ns != ctx->nsend || nr != ctx->nghost) { ctx->err = "fsi_set_rccl: the per-peer counts do not add up to the partition's send / ghost lists"; return FSI_ERR_INVALID; }
  if (hipSetDevice(ctx->device) != hipSuccess) { ctx->err = "fsi_set_rccl: hipSetDevice"; return FSI_ERR_DEVICE; }
  NcclUniqueId id;
  std::memcpy(&id, id128, sizeof id);
  NcclComm comm = nullptr;
  const NcclResult r = a.CommInitRank(&comm, world, id, rank);
  if (r != 0) return fail(ctx, "ncclCommInitRank", r);
  ctx->rccl_comm = comm;
  ctx->rccl_rank = rank;
  ctx->rccl_world = world;
  ctx->rccl_send.assign(send_counts, send_counts + world);
  ctx->rccl_recv.assign(recv_counts, recv_counts + world);
  if (ctx->rccl_red.alloc(4096) != hipSuccess) { ctx->err = "fsi_set_rccl: staging buffer"; return FSI_ERR_DEVICE; }
  ctx->rccl = true;
  if (getenv("FSI_DEBUG")) fprintf(stderr, "[fsi] rank %d of %d: collectives by the library through %s\n", rank, world, a.where.c_str());
  return FSI_OK;
}

void rccl_destroy(FsiCtx* ctx) {
  if (ctx->rccl_comm && !ctx->rccl_dead) {
    (void)hipStreamSynchronize(ctx->stream);
    api().CommDestroy(ctx->rccl_comm);
  }
  ctx->rccl_comm = nullptr;
  ctx->rccl = false;
  ctx->rccl_dead = false;
  ctx->rccl_red.release();
}

// in place, on the solver stream; n doubles in device memory
int rccl_allreduce_dev(FsiCtx* ctx, double* dptr, int64_t n) {
  if (ctx->rccl_dead) return dead(ctx);
  const NcclResult r = api().AllReduce(dptr, dptr, (size_t)n, kNcclFloat64, kNcclSum, ctx->rccl_comm, ctx->stream);
  if (r != 0) return fail(ctx, "ncclAllReduce", r);
  ctx->rccl_allreduces += 1;
  return FSI_OK;
}

// host values: staged through a small device buffer (norms and verdicts outside the Krylov iterations)
int rccl_allreduce_host(FsiCtx* ctx, double* v, int n) {
  if ((size_t)n > ctx->rccl_red.n) { if (ctx->rccl_red.alloc((size_t)n) != hipSuccess) { ctx->err = "RCCL staging buffer"; return FSI_ERR_DEVICE; } }
  if (hipMemcpyAsync(ctx->rccl_red.p, v, (size_t)n * sizeof(double), hipMemcpyHostToDevice, ctx->stream) != hipSuccess) { ctx->err = "RCCL staging copy"; return FSI_ERR_DEVICE; }
  const int rc = rccl_allreduce_dev(ctx, ctx->rccl_red.p, n);
  if (rc != FSI_OK) return rc;
  if (hipMemcpyAsync(v, ctx->rccl_red.p, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
      hipStreamSynchronize(ctx->stream) != hipSuccess) { ctx->err = "RCCL staging copy back"; return FSI_ERR_DEVICE; }
  return FSI_OK;
}

// sendbuf (packed, peers ascending) -> the peers' recvbuf: one grouped set of point-to-point transfers on the solver stream;
// over xGMI every neighbour pair has its own link, so the exchanges run side by side
int rccl_halo(FsiCtx* ctx) {
  Api& a = api();
  if (ctx->rccl_dead) return dead(ctx);
  NcclResult r = a.GroupStart();
  if (r != 0) return fail(ctx, "ncclGroupStart", r);
  int64_t so = 0, ro = 0;
  for (int p = 0; p < ctx->rccl_world; ++p) {
    if (ctx->rccl_send[p] > 0) { r = a.Send(ctx->sendbuf + so, (size_t)ctx->rccl_send[p], kNcclFloat64, p, ctx->rccl_comm, ctx->stream); if (r != 0) { (void)a.GroupEnd(); return fail(ctx, "ncclSend", r); } }
    if (ctx->rccl_recv[p] > 0) { r = a.Recv(ctx->recvbuf + ro, (size_t)ctx->rccl_recv[p], kNcclFloat64, p, ctx->rccl_comm, ctx->stream); if (r != 0) { (void)a.GroupEnd(); return fail(ctx, "ncclRecv", r); } }
    so += ctx->rccl_send[p];
    ro += ctx->rccl_recv[p];
  }
  r = a.GroupEnd();
  if (r != 0) return fail(ctx, "ncclGroupEnd", r);
  ctx->rccl_halos += 1;
  return FSI_OK;
}

}  // namespace fsi
