// dist.cpp -- librgbid_dist.so: the multi-GPU helpers of include/rgbid_dist.h.
//
// Host code only (compiled by hipcc for the HIP / RCCL headers): chunk partitioning, a minimal TCP rendezvous for the RCCL unique id,
// the communicator bound to an rgbid context, ONE all-gather of 392-byte records on the context's stream, and the host-side
// composition of the global trajectory.  The reference has no counterpart (it is single-GPU); its per-frame outputs
// odo_rmats_/odo_tvecs_/odo_covmats_ (src/visodo.cpp:2150-2152) are what the records carry.
#include "../../include/rgbid_dist.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <arpa/inet.h>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <sys/socket.h>
#include <unistd.h>

#include <cerrno>
#include <cstdio>
#include <cstring>
#include <new>
#include <vector>

struct rgbid_dist {
  rgbid_ctx* ctx = nullptr;
  ncclComm_t comm = nullptr;
  hipStream_t stream = nullptr;
  int world = 0, rank = 0, device = 0;
  int* token = nullptr;   // device int for the barrier all-reduce
};

namespace {

inline int rccl_err(ncclResult_t r) { return r == ncclSuccess ? RGBID_OK : RGBID_E_RCCL - (int)r; }

bool send_all(int fd, const void* p, size_t n) {
  const char* c = static_cast<const char*>(p);
  while (n) {
    ssize_t k = ::send(fd, c, n, MSG_NOSIGNAL);
    if (k < 0) { if (errno == EINTR) continue; return false; }
    c += k; n -= (size_t)k;
  }
  return true;
}
bool recv_all(int fd, void* p, size_t n) {
  char* c = static_cast<char*>(p);
  while (n) {
    ssize_t k = ::recv(fd, c, n, 0);
    if (k < 0) { if (errno == EINTR) continue; return false; }
    if (k == 0) return false;
    c += k; n -= (size_t)k;
  }
  return true;
}

void mat3_mul(const double* A, const double* B, double* C) {
  double r[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) r[i * 3 + j] = A[i * 3] * B[j] + A[i * 3 + 1] * B[3 + j] + A[i * 3 + 2] * B[6 + j];
  memcpy(C, r, sizeof(r));
}

}  // namespace

extern "C" {

int rgbid_dist_chunk_ranges(int n_frames, int n_chunks, int* first, int* last) {
  if (!first || !last || n_chunks < 1 || n_frames < n_chunks + 1) return RGBID_E_INVALID;
  const int steps = n_frames - 1;            // frame-to-frame transitions to distribute
  const int base = steps / n_chunks, extra = steps % n_chunks;
  int s = 0;
  for (int c = 0; c < n_chunks; ++c) {
    int n = base + (c < extra ? 1 : 0);
    first[c] = s; last[c] = s + n;
    s += n;
  }
  return RGBID_OK;
}

int rgbid_dist_rank_chunks(int n_chunks, int world, int rank, int* start, int* count) {
  if (!start || !count || n_chunks < 0 || world < 1 || rank < 0 || rank >= world) return RGBID_E_INVALID;
  const int per = n_chunks / world, extra = n_chunks % world;
  *start = rank * per + (rank < extra ? rank : extra);
  *count = per + (rank < extra ? 1 : 0);
  return RGBID_OK;
}

int rgbid_dist_broadcast_bytes(const char* addr, int port, int world, int rank, void* blob, size_t n) {
  if (!blob || world < 1 || rank < 0 || rank >= world) return RGBID_E_INVALID;
  if (world == 1) return RGBID_OK;
  if (!addr || port <= 0 || port > 65535) return RGBID_E_INVALID;
  sockaddr_in sa{};
  sa.sin_family = AF_INET;
  sa.sin_port = htons((uint16_t)port);
  if (inet_pton(AF_INET, addr, &sa.sin_addr) != 1) return RGBID_E_NET;
  if (rank == 0) {
    int ls = ::socket(AF_INET, SOCK_STREAM, 0);
    if (ls < 0) return RGBID_E_NET;
    int one = 1;
    setsockopt(ls, SOL_SOCKET, SO_REUSEADDR, &one, sizeof(one));
    if (::bind(ls, (sockaddr*)&sa, sizeof(sa)) != 0 || ::listen(ls, world) != 0) { ::close(ls); return RGBID_E_NET; }
    timeval tv{120, 0};   // a rank that never shows up must not hang the job forever
    setsockopt(ls, SOL_SOCKET, SO_RCVTIMEO, &tv, sizeof(tv));
    int served = 0, rc = RGBID_OK;
    while (served < world - 1) {
      int fd = ::accept(ls, nullptr, nullptr);
      if (fd < 0) { if (errno == EINTR) continue; rc = RGBID_E_NET; break; }
      int peer = -1;
      bool ok = recv_all(fd, &peer, sizeof(peer)) && peer > 0 && peer < world && send_all(fd, blob, n);
      ::close(fd);
      if (!ok) { rc = RGBID_E_NET; break; }
      ++served;
    }
    ::close(ls);
    return rc;
  }
  // ranks > 0: rank 0 may not be listening yet -- retry for up to ~60 s
  for (int attempt = 0; attempt < 600; ++attempt) {
    int fd = ::socket(AF_INET, SOCK_STREAM, 0);
    if (fd < 0) return RGBID_E_NET;
    if (::connect(fd, (sockaddr*)&sa, sizeof(sa)) == 0) {
      timeval tv{120, 0};
      setsockopt(fd, SOL_SOCKET, SO_RCVTIMEO, &tv, sizeof(tv));
      bool ok = send_all(fd, &rank, sizeof(rank)) && recv_all(fd, blob, n);
      ::close(fd);
      return ok ? RGBID_OK : RGBID_E_NET;
    }
    ::close(fd);
    usleep(100 * 1000);
  }
  return RGBID_E_NET;
}

int rgbid_dist_new_id(rgbid_dist_id* id) {
  static_assert(sizeof(rgbid_dist_id) == sizeof(ncclUniqueId), "rgbid_dist_id carries an ncclUniqueId");
  if (!id) return RGBID_E_INVALID;
  ncclUniqueId u;
  ncclResult_t r = ncclGetUniqueId(&u);
  if (r != ncclSuccess) return rccl_err(r);
  memcpy(id->bytes, &u, sizeof(u));
  return RGBID_OK;
}

int rgbid_dist_exchange_id(const char* addr, int port, int world, int rank, rgbid_dist_id* id) {
  if (!id || world < 1 || rank < 0 || rank >= world) return RGBID_E_INVALID;
  if (rank == 0) { int r = rgbid_dist_new_id(id); if (r) return r; }
  return rgbid_dist_broadcast_bytes(addr, port, world, rank, id->bytes, sizeof(id->bytes));
}

int rgbid_dist_init(rgbid_dist** out, rgbid_ctx* ctx, const rgbid_dist_id* id, int world, int rank) {
  if (!out || !ctx || !id || world < 1 || rank < 0 || rank >= world) return RGBID_E_INVALID;
  *out = nullptr;
  rgbid_dist* d = new (std::nothrow) rgbid_dist();
  if (!d) return RGBID_E_NOMEM;
  d->ctx = ctx; d->world = world; d->rank = rank;
  void* s = nullptr;
  int r = rgbid_ctx_get_stream(ctx, &s);
  if (r) { delete d; return r; }
  d->stream = (hipStream_t)s;
  // the communicator lives on the device of the context's stream
  hipError_t he = hipStreamGetDevice(d->stream, &d->device);
  if (he == hipSuccess) he = hipSetDevice(d->device);
  if (he != hipSuccess) { delete d; return (int)he; }
  ncclUniqueId u;
  memcpy(&u, id->bytes, sizeof(u));
  ncclResult_t nr = ncclCommInitRank(&d->comm, world, u, rank);
  if (nr != ncclSuccess) { delete d; return rccl_err(nr); }
  int count = 0;
  nr = ncclCommCount(d->comm, &count);
  if (nr != ncclSuccess || count != world) { ncclCommDestroy(d->comm); delete d; return nr != ncclSuccess ? rccl_err(nr) : RGBID_E_INVALID; }
  he = hipMalloc((void**)&d->token, sizeof(int));
  if (he == hipSuccess) he = hipMemsetAsync(d->token, 0, sizeof(int), d->stream);
  if (he != hipSuccess) { ncclCommDestroy(d->comm); delete d; return (int)he; }
  *out = d;
  return RGBID_OK;
}

int rgbid_dist_destroy(rgbid_dist* d) {
  if (!d) return RGBID_OK;
  hipError_t he = hipSetDevice(d->device);
  if (he == hipSuccess) he = hipStreamSynchronize(d->stream);
  if (d->token) (void)hipFree(d->token);
  ncclResult_t nr = d->comm ? ncclCommDestroy(d->comm) : ncclSuccess;
  delete d;
  if (he != hipSuccess) return (int)he;
  return rccl_err(nr);
}

int rgbid_dist_world(const rgbid_dist* d) {
  if (!d) return 0;
  int count = 0;
  return ncclCommCount(d->comm, &count) == ncclSuccess ? count : 0;
}
int rgbid_dist_rank(const rgbid_dist* d) {
  if (!d) return -1;
  int r = -1;
  return ncclCommUserRank(d->comm, &r) == ncclSuccess ? r : -1;
}

int rgbid_dist_gather_records(rgbid_dist* d, const rgbid_gather_record* local_dev, int n_local, rgbid_gather_record* all_dev) {
  static_assert(sizeof(rgbid_gather_record) == 392, "SURVEY 8e record");
  if (!d || !local_dev || !all_dev || n_local < 0) return RGBID_E_INVALID;
  if (n_local == 0) return RGBID_OK;
  if (hipError_t he = hipSetDevice(d->device); he != hipSuccess) return (int)he;
  return rccl_err(ncclAllGather(local_dev, all_dev, (size_t)n_local * sizeof(rgbid_gather_record), ncclChar, d->comm, d->stream));
}

int rgbid_dist_barrier(rgbid_dist* d) {
  if (!d) return RGBID_E_INVALID;
  if (hipError_t e0 = hipSetDevice(d->device); e0 != hipSuccess) return (int)e0;
  ncclResult_t nr = ncclAllReduce(d->token, d->token, 1, ncclInt, ncclSum, d->comm, d->stream);
  if (nr != ncclSuccess) return rccl_err(nr);
  hipError_t he = hipStreamSynchronize(d->stream);
  return he == hipSuccess ? RGBID_OK : (int)he;
}

int rgbid_dist_compose_trajectory(const rgbid_gather_record* all, int world, int lanes_per_rank, int n_chunks, int chunk_len,
                                  const int* first, const int* last, double* R, double* t, int* status, double* cov) {
  if (!all || !first || !last || !R || !t || world < 1 || lanes_per_rank < 1 || n_chunks < 1 || chunk_len < 1) return RGBID_E_INVALID;
  const int F = last[n_chunks - 1] + 1;
  double Rw[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, tw[3] = {0, 0, 0};
  memcpy(R, Rw, sizeof(Rw)); memcpy(t, tw, sizeof(tw));
  if (status) status[0] = 0;
  if (cov) memset(cov, 0, sizeof(double) * 36);
  for (int c = 0; c < n_chunks; ++c) {
    const int n = last[c] - first[c] + 1;
    if (n < 1 || n > chunk_len || last[c] >= F || (c && first[c] != last[c - 1])) return RGBID_E_INVALID;
    // owner of chunk c and its index inside the owner's block
    int owner = -1, local = -1;
    for (int r = 0; r < world; ++r) {
      int s, cnt;
      rgbid_dist_rank_chunks(n_chunks, world, r, &s, &cnt);
      if (c >= s && c < s + cnt) { owner = r; local = c - s; break; }
    }
    if (owner < 0 || local >= lanes_per_rank) return RGBID_E_INVALID;
    const rgbid_gather_record* rec = all + ((size_t)owner * lanes_per_rank + local) * chunk_len;
    if (c == 0 && status) status[0] = rec[0].status;
    // the chunk's first frame IS the previous chunk's last frame: its pose is already composed; continue from there
    for (int j = 1; j < n; ++j) {
      const rgbid_gather_record& g = rec[j];
      double tn[3];
      for (int i = 0; i < 3; ++i) tn[i] = Rw[i * 3] * g.t[0] + Rw[i * 3 + 1] * g.t[1] + Rw[i * 3 + 2] * g.t[2] + tw[i];
      mat3_mul(Rw, g.R, Rw);
      memcpy(tw, tn, sizeof(tn));
      const int k = first[c] + j;
      memcpy(R + (size_t)k * 9, Rw, sizeof(Rw)); memcpy(t + (size_t)k * 3, tw, sizeof(tw));
      if (status) status[k] = g.status;
      if (cov) memcpy(cov + (size_t)k * 36, g.cov, sizeof(g.cov));
    }
  }
  return RGBID_OK;
}

}  // extern "C"
