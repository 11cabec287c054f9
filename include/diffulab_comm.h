/* C ABI of libdiffulab_comm.so: the data-parallel gradient exchange of the training step (RCCL over xGMI, one process per GPU).
 *
 * Reference interface replaced: the DistributedDataParallel wrap that `accelerator.prepare(denoiser, ...)` installs
 * (training/trainers/base_trainer.py:277-279; Accelerator built at training/trainers/common.py:103-109) and its bucketed
 * autograd-hook all-reduce.  Here the engine's backward declares contiguous ranges of the flat gradient arena final and each range
 * is reduced by ONE in-place collective on the communicator's own stream, overlapping the rest of the backward.
 * Conventions as in diffulab_hip.h: extern "C", plain pointers and sizes, caller-owned device buffers, int status
 * (0 = ok, negative = error; dl_comm_last_error() is thread-local), the caller's hipStream_t passed explicitly.
 * The communicator is the library's only state. */
#pragma once
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct dl_comm_t dl_comm_t;
typedef void* dl_stream_t;

const char* dl_comm_last_error(void);
/* rank 0 creates the 128-byte rendezvous token; the host distributes it (torch.distributed store / broadcast, a file, MPI ...) */
int dl_comm_unique_id(char out[128]);
int dl_comm_init(dl_comm_t** out, const char id[128], int rank, int world, int device);
int dl_comm_destroy(dl_comm_t* comm);
/* SUM over ranks of grad[0, count) f32, in place, as reduce-scatter + all-gather; enqueued on the comm stream behind the work
 * `after` has been given so far; returns immediately */
int dl_reduce_scatter_allgather_async(dl_comm_t* comm, float* grad, int64_t count, dl_stream_t after);
int dl_comm_broadcast_async(dl_comm_t* comm, float* buf, int64_t count, int root, dl_stream_t after);
/* the next collective also waits for `event` (hipEvent_t of a producer on another stream) */
int dl_comm_after_event(dl_comm_t* comm, void* event);
/* `stream` waits (on the device) for every collective enqueued so far */
int dl_comm_wait(dl_comm_t* comm, dl_stream_t stream);
int dl_comm_rank(const dl_comm_t* comm);
int dl_comm_world(const dl_comm_t* comm);

#ifdef __cplusplus
}
#endif
