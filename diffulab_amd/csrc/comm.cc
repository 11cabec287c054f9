// libdiffulab_comm.so: the data-parallel gradient exchange of the training step behind the C ABI (SURVEY.md section 8b:
// dl_comm_{init,destroy}, dl_reduce_scatter_allgather_async, dl_comm_wait).  Replaces the DDP wrap that accelerator.prepare
// installs in the reference (training/trainers/base_trainer.py:277-279): one process per GPU, RCCL over xGMI, ONE communicator
// and ONE comm stream per process, collectives enqueued on that stream behind an event of the caller's compute stream, so a
// bucket's reduction overlaps the rest of the backward pass.  The only state is the opaque dl_comm_t.
//
// A separate library (not part of libdiffulab_hip.so) so that single-GPU use never needs librccl at load time.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#define DL_COMM_API extern "C" __attribute__((visibility("default")))

struct dl_comm_t {
  ncclComm_t comm;
  hipStream_t stream;  // the comm stream
  hipEvent_t ready;    // recorded on the caller's stream: "this gradient range is final"
  hipEvent_t done;     // recorded on the comm stream after the last enqueued collective
  int rank, world;
};

static thread_local char g_err[512] = "";
static int fail(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return -1;
}
DL_COMM_API const char* dl_comm_last_error(void) { return g_err; }

#define HIPC(x)                                                                    \
  do {                                                                             \
    hipError_t e__ = (x);                                                          \
    if (e__ != hipSuccess) return fail("%s: %s", #x, hipGetErrorString(e__));      \
  } while (0)
#define NCCLC(x)                                                                   \
  do {                                                                             \
    ncclResult_t r__ = (x);                                                        \
    if (r__ != ncclSuccess) return fail("%s: %s", #x, ncclGetErrorString(r__));    \
  } while (0)

// 128-byte rendezvous token: rank 0 creates it, the host code hands it to every rank (torch.distributed store / broadcast)
DL_COMM_API int dl_comm_unique_id(char out[128]) {
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
  ncclUniqueId id;
  NCCLC(ncclGetUniqueId(&id));
  memcpy(out, &id, 128);
  return 0;
}

static void comm_release(dl_comm_t* c) {  // every handle that was created so far (zero-initialised struct: nullptr = not yet)
  if (c->comm) (void)ncclCommDestroy(c->comm);
  if (c->ready) (void)hipEventDestroy(c->ready);
  if (c->done) (void)hipEventDestroy(c->done);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}
static int comm_setup(dl_comm_t* c, const char id[128], int rank, int world) {
  ncclUniqueId uid;
  memcpy(&uid, id, 128);
  NCCLC(ncclCommInitRank(&c->comm, world, uid, rank));
  HIPC(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
  HIPC(hipEventCreateWithFlags(&c->ready, hipEventDisableTiming));
  HIPC(hipEventCreateWithFlags(&c->done, hipEventDisableTiming));
  return 0;
}
DL_COMM_API int dl_comm_init(dl_comm_t** out, const char id[128], int rank, int world, int device) {
  if (!out || !id || world < 1 || rank < 0 || rank >= world) return fail("dl_comm_init: bad arguments");
  HIPC(hipSetDevice(device));
  dl_comm_t* c = new dl_comm_t();
  c->rank = rank;
  c->world = world;
  if (comm_setup(c, id, rank, world) != 0) {  // (the error text is already set)
    comm_release(c);
    return -1;
  }
  *out = c;
  return 0;
}

DL_COMM_API int dl_comm_destroy(dl_comm_t* c) {
  if (!c) return 0;
  (void)hipStreamSynchronize(c->stream);
  (void)ncclCommDestroy(c->comm);
  (void)hipEventDestroy(c->ready);
  (void)hipEventDestroy(c->done);
  (void)hipStreamDestroy(c->stream);
  delete c;
  return 0;
}

// SUM-all-reduce of grad[0, count) f32 in place as reduce-scatter + all-gather over the ring (what the survey calls
// dl_reduce_scatter_allgather_async), enqueued on the comm stream behind everything `after` (the caller's compute stream) has
// been given so far.  count need not divide by the world size (the ragged tail is all-reduced directly).
DL_COMM_API int dl_reduce_scatter_allgather_async(dl_comm_t* c, float* grad, int64_t count, hipStream_t after) {
  if (!c || !grad || count <= 0) return fail("dl_reduce_scatter_allgather_async: bad arguments");
  HIPC(hipEventRecord(c->ready, after));
  HIPC(hipStreamWaitEvent(c->stream, c->ready, 0));
  const int64_t per = count / c->world, body = per * c->world;
  if (per > 0) {
    NCCLC(ncclReduceScatter(grad, grad + (int64_t)c->rank * per, (size_t)per, ncclFloat32, ncclSum, c->comm, c->stream));
    NCCLC(ncclAllGather(grad + (int64_t)c->rank * per, grad, (size_t)per, ncclFloat32, c->comm, c->stream));
  }
  if (body < count) NCCLC(ncclAllReduce(grad + body, grad + body, (size_t)(count - body), ncclFloat32, ncclSum, c->comm, c->stream));
  HIPC(hipEventRecord(c->done, c->stream));
  return 0;
}

// rank-0 broadcast of the parameter arena (DDP constructor semantics)
DL_COMM_API int dl_comm_broadcast_async(dl_comm_t* c, float* buf, int64_t count, int root, hipStream_t after) {
  if (!c || !buf || count <= 0) return fail("dl_comm_broadcast_async: bad arguments");
  HIPC(hipEventRecord(c->ready, after));
  HIPC(hipStreamWaitEvent(c->stream, c->ready, 0));
  NCCLC(ncclBroadcast(buf, buf, (size_t)count, ncclFloat32, root, c->comm, c->stream));
  HIPC(hipEventRecord(c->done, c->stream));
  return 0;
}

// the next collective also waits for `event` (a producer on another stream: the side-stream weight-gradient GEMMs)
DL_COMM_API int dl_comm_after_event(dl_comm_t* c, hipEvent_t event) {
  if (!c || !event) return fail("dl_comm_after_event: bad arguments");
  HIPC(hipStreamWaitEvent(c->stream, event, 0));
  return 0;
}

// make `stream` wait for every collective enqueued so far (no host synchronisation)
DL_COMM_API int dl_comm_wait(dl_comm_t* c, hipStream_t stream) {
  if (!c) return fail("dl_comm_wait: null communicator");
  HIPC(hipStreamWaitEvent(stream, c->done, 0));
  return 0;
}

DL_COMM_API int dl_comm_rank(const dl_comm_t* c) { return c ? c->rank : -1; }
DL_COMM_API int dl_comm_world(const dl_comm_t* c) { return c ? c->world : -1; }
