/* dbcsr_amd_comm.h -- the panel exchange of the distributed multiply under the C-ABI.
 *
 * The reference's Cannon loop moves its A and B panels with device-pointer MPI isend / irecv
 * (multiply_cannon_g2g, src/mm/dbcsr_mm_cannon.F:2528-2557, 2655-2684; host-staged variant :1376-1463, 1497-1586),
 * preceded by an allgather of sizes and an index + data exchange in make_images (:532, :674-678, :1036).  On one
 * MI355X node those transfers belong on RCCL over xGMI: this header gives the MPI / Fortran host (or the Python
 * driver dbcsr_amd/cannon.py) the three primitives it needs, on a stream of its choice, ordered against the compute
 * stream with the events of dbcsr_acc.h (c_dbcsr_acc_event_record / c_dbcsr_acc_stream_wait_event -- where the
 * reference has acc_event_synchronize(...%acc_ready), dbcsr_mm_cannon.F:1427-1431, 1550-1554):
 *   - a communicator bootstrapped from a 128-byte id that rank 0 creates and the host broadcasts (MPI_Bcast or any
 *     other channel): RCCL needs no launcher of its own;
 *   - a grouped point-to-point exchange of device buffers (block data and int32 index arrays alike, counted in
 *     bytes): every send and receive of one tick / one redistribution step is posted as ONE group, so all xGMI links
 *     of the node carry traffic at once;
 *   - an allgather (sizes of what will be exchanged).
 * librccl.so is loaded on first use; without it (or without a second GPU) the create call fails and nothing else in
 * the library is affected.  Return value 0 = success, as in dbcsr_acc.h. */
#ifndef DBCSR_AMD_COMM_H
#define DBCSR_AMD_COMM_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define DBCSR_AMD_COMM_ID_BYTES 128

typedef struct dbcsr_amd_comm_op {
  void* buf;     /* device pointer */
  int64_t bytes; /* may be 0: the operation is skipped on both sides only if both sides pass 0 */
  int32_t peer;  /* rank in the communicator */
  int32_t reserved;
} dbcsr_amd_comm_op;

/* 1 when librccl could be loaded with every entry point this file uses, else 0.  Local, never blocks: the host calls it on every
 * rank and agrees on the answer (an allreduce of its own) BEFORE the collective create call, so that no rank waits in
 * ncclCommInitRank for a peer that cannot get there */
int dbcsr_amd_comm_available(void);
/* rank 0: fill `id` (ncclGetUniqueId); the host distributes it to every rank */
int dbcsr_amd_comm_unique_id(char id[DBCSR_AMD_COMM_ID_BYTES]);
/* collective over the nranks processes: communicator on the calling thread's current device (one process per GPU) */
int dbcsr_amd_comm_create(void** comm, const char id[DBCSR_AMD_COMM_ID_BYTES], int nranks, int rank);
int dbcsr_amd_comm_destroy(void* comm);
int dbcsr_amd_comm_rank(void* comm, int* rank, int* nranks);
/* one grouped exchange on `stream` (an acc stream handle, NULL = the null stream): asynchronous; matching sends and
 * receives must be posted by both peers in the same call sequence */
int dbcsr_amd_comm_exchange(void* comm, const dbcsr_amd_comm_op* sends, int nsend, const dbcsr_amd_comm_op* recvs, int nrecv, void* stream);
/* recv[r * bytes_per_rank ...] = send of rank r (device buffers) */
int dbcsr_amd_comm_allgather(void* comm, const void* send, void* recv, int64_t bytes_per_rank, void* stream);

#ifdef __cplusplus
}
#endif
#endif
