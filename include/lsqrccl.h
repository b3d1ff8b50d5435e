/*
 * lsqrccl.h -- the row-sharded runs' all-reduce as a DIRECT RCCL call (include/lsqhip.h:
 * lsq_options.row_allreduce / lsq_solver_set_row_allreduce).
 *
 * liblsqrccl.so is a thin shim: it binds librccl.so at run time (dlopen -- the copy the host process already uses, e.g.
 * PyTorch's, so that there is one RCCL in the process), owns an ncclComm_t per handle and exports a callback of the
 * lsq_device_allreduce_callback shape whose body is
 *     ncclAllReduce(d_buf, d_buf, count, ncclDouble, ncclSum, comm, hip_stream)
 * i.e. the collective is enqueued on the library's own stream: no host synchronisation, no staging copy, no Python in the
 * inner iteration.  Over xGMI the payload of an inner iteration at C4's width (n + 1 = 10001 doubles, 80 KB) is
 * latency-bound.  One process per GPU; the unique id travels over whatever the launcher offers (torch.distributed's store /
 * broadcast_object_list in leastsquaresoptim.jl_amd/rowshard.py; MPI_Bcast from Julia).
 */
#ifndef LSQRCCL_H
#define LSQRCCL_H
#ifdef __cplusplus
extern "C" {
#endif

/* bind librccl.so (NULL: the loader's search path); 0 on success.  Idempotent. */
int lsq_rccl_load(const char *librccl_path);
const char *lsq_rccl_last_error(void);
/* ncclGetUniqueId (rank 0), 128 bytes */
int lsq_rccl_unique_id(unsigned char out[128]);
/* ncclCommInitRank on the calling thread's current HIP device; collective over the `world` ranks */
int lsq_rccl_comm_create(const unsigned char id[128], int rank, int world, void **comm_out);
int lsq_rccl_comm_destroy(void *comm);
/* the callback to put into lsq_options.row_allreduce, with the comm handle as row_allreduce_user
 * (signature of lsq_device_allreduce_callback: (double *d_buf, int count, void *hip_stream, void *user) -> 0 on success) */
void *lsq_rccl_allreduce_callback(void);
/* how many collectives / doubles this comm handle has enqueued (diagnostics for the tests) */
int lsq_rccl_comm_stats(void *comm, long long *calls, long long *doubles);

#ifdef __cplusplus
}
#endif
#endif
