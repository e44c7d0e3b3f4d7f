/*
 * rgbid_dist.h -- C-ABI of the multi-GPU helpers (librgbid_dist.so, links librccl directly).
 *
 * The tracker is sequential inside a sequence (pose prior from the previous frame src/visodo.cpp:1016-1032, keyframes persist
 * :2172-2211), so the independent unit is a CHUNK: a contiguous sub-sequence tracked from identity by one lane of one GPU's batched
 * engine (rgbid_engine.h).  One host process per GPU; ranks exchange ONLY fixed-size per-frame records -- what the reference
 * appends to odo_rmats_/odo_tvecs_/odo_covmats_ per frame (src/visodo.cpp:2150-2152) plus frame id and status -- with ONE
 * all-gather over RCCL/xGMI; no image data crosses GPUs and the path has no all-reduce.  Every rank (or rank 0) then composes the
 * global trajectory T_w,k = T_w,k-1 * dT_k on the host.
 *
 * A C++ host shards like this (one process per GPU, rank / world from its launcher):
 *     rgbid_dist_exchange_id(master_addr, port, world, rank, &id)        // rank 0 creates the RCCL id, the others receive it (TCP)
 *     rgbid_dist_init(&d, ctx, &id, world, rank)                         // ncclCommInitRank on the context's device
 *     rgbid_dist_chunk_ranges / rgbid_dist_rank_chunks                   // which frames this rank's engine lanes track
 *     ... rgbid_engine_step x chunk_len ...
 *     rgbid_engine_pack_gather_records(engine, 0, chunk_len, local_dev)  // device-side: pose-record ring -> [lanes][chunk_len] records
 *     rgbid_dist_gather_records(d, local_dev, lanes * chunk_len, all_dev)// ncclAllGather on the context's stream
 *     rgbid_dist_compose_trajectory(all_host, ...)                       // after one D2H of world * lanes * chunk_len * 392 bytes
 * Functions return 0, a positive hipError_t, a negative RGBID_E_* code, or RGBID_E_RCCL - ncclResult_t.
 */
#ifndef RGBID_DIST_H_
#define RGBID_DIST_H_

#include "rgbid_engine.h"

#ifdef __cplusplus
extern "C" {
#endif

#define RGBID_E_RCCL (-100)   /* RCCL failure: the return value is RGBID_E_RCCL - ncclResult_t */
#define RGBID_E_NET  (-90)    /* rendezvous (socket) failure */

/* rgbid_gather_record (392 bytes) and rgbid_engine_pack_gather_records: rgbid_engine.h */
typedef struct rgbid_dist rgbid_dist;
typedef struct rgbid_dist_id { char bytes[128]; } rgbid_dist_id;   /* ncclUniqueId */

/* ---- partitioning (pure host arithmetic) ----
 * n_frames frames -> n_chunks contiguous chunks that overlap by one frame (chunk c ends on the frame chunk c+1 starts on);
 * first[c] / last[c] inclusive; chunk lengths differ by at most one.  Needs n_frames >= n_chunks + 1. */
int rgbid_dist_chunk_ranges(int n_frames, int n_chunks, int* first, int* last);
/* the block of consecutive chunks owned by `rank`: [*start, *start + *count) */
int rgbid_dist_rank_chunks(int n_chunks, int world, int rank, int* start, int* count);

/* ---- rendezvous: rank 0 generates the RCCL unique id and serves it on addr:port (TCP), ranks 1..world-1 fetch it.  With
 * world == 1 no socket is opened.  Any out-of-band channel the host already has (MPI, a file, torch's store) works as well:
 * rgbid_dist_new_id on rank 0, ship the 128 bytes, rgbid_dist_init everywhere. ---- */
int rgbid_dist_new_id(rgbid_dist_id* id);
int rgbid_dist_exchange_id(const char* addr, int port, int world, int rank, rgbid_dist_id* id);
/* the transport alone (rank 0's `blob` of n bytes reaches every rank); exposed so that it can be tested without RCCL */
int rgbid_dist_broadcast_bytes(const char* addr, int port, int world, int rank, void* blob, size_t n);

/* ---- communicator bound to a context (its device and HIP stream) ---- */
int rgbid_dist_init(rgbid_dist** d, rgbid_ctx* ctx, const rgbid_dist_id* id, int world, int rank);
int rgbid_dist_destroy(rgbid_dist* d);
int rgbid_dist_world(const rgbid_dist* d);    /* the rank count RCCL reports (ncclCommCount) */
int rgbid_dist_rank(const rgbid_dist* d);

/* the only collective on the path: all-gather of n_local records per rank.  local_dev: device [n_local]; all_dev: device
 * [world * n_local], rank-major.  Asynchronous on the context's stream (follow with rgbid_ctx_sync or a stream-ordered D2H). */
int rgbid_dist_gather_records(rgbid_dist* d, const rgbid_gather_record* local_dev, int n_local, rgbid_gather_record* all_dev);
/* stream barrier across ranks (a 1-int all-reduce; used by benchmarks to bracket timed regions, not by the data path) */
int rgbid_dist_barrier(rgbid_dist* d);

/* ---- composition (host).  all: the gathered buffer [world][lanes_per_rank][chunk_len], rank-major -- chunk c sits in the block of the
 * rank that owns it (rgbid_dist_rank_chunks) at its index inside that block; ranks that own fewer than lanes_per_rank chunks pad with
 * lanes nobody reads (world = 1, lanes_per_rank = n_chunks: plain chunk-major).  first/last from rgbid_dist_chunk_ranges.  Writes the
 * global pose of every frame, T_w,k = T_w,k-1 * dT_k from the chunk that ENDS on or contains frame k: R [n_frames][9] row-major, t
 * [n_frames][3], frame 0 = identity; status (nullable) [n_frames] the status bits of the record that produced the frame; cov
 * (nullable) [n_frames][36] the frame-to-frame covariances (zero for frame 0). ---- */
int rgbid_dist_compose_trajectory(const rgbid_gather_record* all, int world, int lanes_per_rank, int n_chunks, int chunk_len,
                                  const int* first, const int* last, double* R, double* t, int* status, double* cov);

#ifdef __cplusplus
}
#endif
#endif
