/*
 * lsqrccl.h -- the collectives of sharded runs as DIRECT RCCL calls: the row-sharded single problem's in-stream all-reduce
 * (include/lsqhip.h: lsq_options.row_allreduce / lsq_solver_set_row_allreduce) and, below it, the one scalar
 * exchange per outer iteration of independent problems (lsq_options.allreduce).
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

/* ---- the one exchange of INDEPENDENT problems (north_star: "a single RCCL all-reduce over xGMI for the global
 * ||r|| stopping test"; extends the reference's per-problem stop test, levenberg_marquardt.jl:123-124, to the whole job) ----
 * lsq_options.allreduce (include/lsqhip.h) served in C: per OUTER iteration ONE ncclAllReduce(sum) of world + 3 doubles
 * {sum of ssr, converged count, leaving count, one gradient-norm slot per rank} on a side stream of its own, staged through
 * page-locked memory; every rank recovers {sum ssr, max |g|, all converged}.  Semantics (the contract written at
 * lsq_allreduce_callback): a rank that reports converged or leaving, and every rank's first call, waits for its exchange
 * and gets this iteration's values; an ACTIVE rank gets the PREVIOUS exchange's values and leaves the new one in flight
 * under its iteration's device work (it never acts on "all converged": it is not converged itself); a rank that sees a
 * non-zero leaving count returns 2 (-> LSQ_ERCCL) and issues no further collective, so every rank issues the same number.
 * A wait is bounded by LSQ_EXCHANGE_TIMEOUT_S (default 120 s).
 *     void *comm, *x;  lsq_rccl_comm_create(id, rank, world, &comm);  lsq_rccl_xchg_create(comm, rank, world, &x);
 *     opt.allreduce = (lsq_allreduce_callback)lsq_rccl_xchg_callback();  opt.allreduce_user = x;
 *     lsq_optimize(...);  lsq_rccl_xchg_drain(x);  (before a barrier / before destroying the communicator)
 *     lsq_rccl_xchg_reset(x);  lsq_optimize(...);   (another run on the same handle)
 * The communicator stays the caller's (destroy the exchange first). */
int lsq_rccl_xchg_create(void *comm, int rank, int world, void **xchg_out);
int lsq_rccl_xchg_destroy(void *xchg);
/* the function to put into lsq_options.allreduce, with the exchange handle as allreduce_user
 * (signature of lsq_allreduce_callback: (double *h_vals, int count, void *user) -> 0 ok, 1 failed, 2 a peer has left) */
void *lsq_rccl_xchg_callback(void);
/* completes the exchange an active rank left in flight */
int lsq_rccl_xchg_drain(void *xchg);
/* between two lsq_optimize runs on the SAME handle (every rank, at the same point of its call sequence; no collective is
 * issued): drains, then forgets the previous run's last result and a seen abort -- the protocol state is per RUN: without it
 * an active rank's first call of the next run would return the previous run's final {sum ssr, max |g|, all converged}, and a
 * handle that has seen one abort would return 2 for ever */
int lsq_rccl_xchg_reset(void *xchg);
/* collectives issued, how many of them were waited for on the spot, and whether an abort has been seen */
int lsq_rccl_xchg_stats(void *xchg, long long *collectives, long long *synchronous, int *aborted);
/* The same protocol over ANY transport -- what the CPU tests use to run it over gloo next to its Python twin
 * (leastsquaresoptim.jl_amd/sharding.py), and what an MPI host would plug MPI_Iallreduce / MPI_Wait into:
 *   issue(h_buf, count, slot, user)   start a SUM all-reduce of h_buf[0..count) in place (may return before it is complete);
 *   finish(slot, user)                return once the all-reduce started for `slot` (0 or 1) is complete and h_buf holds it. */
typedef int (*lsq_xchg_issue_fn)(double *h_buf, int count, int slot, void *user);
typedef int (*lsq_xchg_finish_fn)(int slot, void *user);
int lsq_rccl_xchg_create_custom(int rank, int world, lsq_xchg_issue_fn issue, lsq_xchg_finish_fn finish, void *user,
                                void **xchg_out);

#ifdef __cplusplus
}
#endif
#endif
