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
#include <netdb.h>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <poll.h>
#include <sys/socket.h>
#include <unistd.h>

#include <algorithm>
#include <cerrno>
#include <map>
#include <mutex>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
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

// ---- TCP rendezvous ------------------------------------------------------------------------------------------------------------------
// One listening socket on rank 0, one short connection per other rank and exchange.  hello = {magic, rank, nonce, kind, seq, bytes}: the job
// (nonce), WHICH exchange this is (kind: broadcast / gather; seq: the number of exchanges this process has taken part in -- every rank calls the
// same sequence of exchanges, so the counters agree) and its payload size.  Rank 0 answers every hello with one byte: ACK -- the hello names
// the exchange rank 0 is serving and this rank has not been served in it -- or NAK.  A client that is refused (NAK), that finds nobody
// listening, or whose connection breaks before its payload went through, closes, waits and tries again until the deadline: a rank that
// finishes exchange k early and reaches rank 0's listener while it still serves exchange k (back-to-back exchanges on one port, world >= 3)
// is told "not yet" instead of failing the job.  Anything else that connects (a port scanner, a stale process of another job) is dropped.
constexpr uint32_t HELLO_MAGIC = 0x52474245u;   // "RGBE" (the hello of round 3, without kind / seq, was "RGBD")
struct Hello { uint32_t magic; int32_t rank; uint64_t nonce; uint32_t kind, seq; uint64_t bytes; };
constexpr unsigned char ACK = 1, NAK = 0;

uint64_t job_nonce() {
  if (const char* e = getenv("RGBID_DIST_NONCE")) return strtoull(e, nullptr, 0);
  uint64_t h = 0;
  if (const char* e = getenv("TORCHELASTIC_RUN_ID")) { h = 1469598103934665603ull; for (const char* p = e; *p; ++p) { h ^= (unsigned char)*p; h *= 1099511628211ull; } }
  return h;
}
int timeout_s() {
  if (const char* e = getenv("RGBID_DIST_TIMEOUT_S")) { int v = atoi(e); if (v > 0) return v; }
  return 120;
}
using Clock = std::chrono::steady_clock;
int ms_left(Clock::time_point deadline) {
  auto d = std::chrono::duration_cast<std::chrono::milliseconds>(deadline - Clock::now()).count();
  return d < 0 ? 0 : (int)d;
}
void set_io_timeout(int fd, int seconds) {
  timeval tv{seconds, 0};
  setsockopt(fd, SOL_SOCKET, SO_RCVTIMEO, &tv, sizeof(tv));
  setsockopt(fd, SOL_SOCKET, SO_SNDTIMEO, &tv, sizeof(tv));
}
// exchanges entered so far, per (port, rank): every rank calls the same sequence of exchanges on a port, so the counters agree across the ranks of
// a job whether they are processes (the product) or threads of one test process
std::mutex g_seq_mu;
std::map<std::pair<int, int>, uint32_t> g_seq;
uint32_t next_exchange_seq(int port, int rank) {
  std::lock_guard<std::mutex> lock(g_seq_mu);
  return g_seq[std::make_pair(port, rank)]++;
}
// a new job on this port starts counting at 0 again: a rank whose process was restarted, or a job that reuses the port of an earlier one while rank 0's
// process lived on, would otherwise be refused (NAK) until the timeout.  Called at the top of rgbid_dist_track_sequence -- the same logical point on every rank.
void reset_exchange_seq(int port, int rank) {
  std::lock_guard<std::mutex> lock(g_seq_mu);
  g_seq[std::make_pair(port, rank)] = 0;
}

// rank 0: blob -> every rank (gather == false), or every rank's n bytes -> all[world][n] on every rank (gather == true)
int tcp_exchange(const char* addr, int port, int world, int rank, void* blob, size_t n, void* all, bool gather) {
  if (!addr || port <= 0 || port > 65535) return RGBID_E_INVALID;
  addrinfo hints{};
  hints.ai_family = AF_UNSPEC; hints.ai_socktype = SOCK_STREAM;
  if (rank == 0) hints.ai_flags = AI_PASSIVE;
  addrinfo* res = nullptr;
  const std::string ports = std::to_string(port);
  const uint32_t seq = next_exchange_seq(port, rank), kind = gather ? 1u : 0u;   // counted before anything can fail: the ranks' counters stay in step
  if (getaddrinfo(addr, ports.c_str(), &hints, &res) != 0 || !res) return RGBID_E_NET;
  const uint64_t nonce = job_nonce();
  const auto deadline = Clock::now() + std::chrono::seconds(timeout_s());
  int rc = RGBID_E_NET;
  if (rank == 0) {
    int ls = -1;
    // the previous exchange's listener on this port may still be in TIME_WAIT teardown on some stacks: retry the bind briefly
    while (ls < 0 && ms_left(deadline) > 0) {
      for (addrinfo* a = res; a && ls < 0; a = a->ai_next) {
        ls = ::socket(a->ai_family, a->ai_socktype, a->ai_protocol);
        if (ls < 0) continue;
        int one = 1;
        setsockopt(ls, SOL_SOCKET, SO_REUSEADDR, &one, sizeof(one));
        if (::bind(ls, a->ai_addr, a->ai_addrlen) != 0 || ::listen(ls, world + 8) != 0) { ::close(ls); ls = -1; }
      }
      if (ls < 0) usleep(50 * 1000);
    }
    freeaddrinfo(res);
    if (ls < 0) return RGBID_E_NET;
    std::vector<int> fds(world, -1);        // gather: connections stay open until every rank's block has arrived
    std::vector<char> served(world, 0);
    if (gather) memcpy(all, blob, n);
    int n_served = 0, n_naks = 0;
    rc = RGBID_OK;
    while (n_served < world - 1) {
      pollfd pf{ls, POLLIN, 0};
      const int left = ms_left(deadline);
      if (left == 0) { rc = RGBID_E_NET; break; }
      const int pr = ::poll(&pf, 1, left);
      if (pr < 0) { if (errno == EINTR) continue; rc = RGBID_E_NET; break; }
      if (pr == 0) { rc = RGBID_E_NET; break; }
      int fd = ::accept(ls, nullptr, nullptr);
      if (fd < 0) { if (errno == EINTR || errno == EAGAIN || errno == ECONNABORTED) continue; rc = RGBID_E_NET; break; }
      set_io_timeout(fd, 2);                // a peer that connects and says nothing costs 2 s, not the job
      Hello h{};
      if (!recv_all(fd, &h, sizeof(h)) || h.magic != HELLO_MAGIC || h.nonce != nonce) { ::close(fd); continue; }   // not one of ours: no answer
      const bool mine = h.rank > 0 && h.rank < world && h.kind == kind && h.seq == seq && h.bytes == (uint64_t)n && !served[h.rank];
      const unsigned char answer = mine ? ACK : NAK;   // NAK: one of ours, but for another exchange (or served already): it will come back
      if (!mine && n_naks++ < 3)   // a desynchronised exchange counter must be visible, not a silent spin until the timeout
        fprintf(stderr, "rgbid_dist: rank 0 serves exchange (kind %u, seq %u, %zu bytes) on port %d and refused rank %d's hello (kind %u, seq %u, %llu bytes%s)\n", kind, seq, n, port,
                (int)h.rank, h.kind, h.seq, (unsigned long long)h.bytes, (h.rank > 0 && h.rank < world && served[h.rank]) ? ", served already" : "");
      if (!send_all(fd, &answer, 1) || !mine) { ::close(fd); continue; }
      set_io_timeout(fd, timeout_s());
      bool ok;
      if (gather) ok = recv_all(fd, (char*)all + (size_t)h.rank * n, n);
      else ok = send_all(fd, blob, n);
      if (!ok) { ::close(fd); continue; }                  // the rank reconnects until the deadline
      served[h.rank] = 1; ++n_served;
      if (gather) fds[h.rank] = fd; else ::close(fd);
    }
    if (gather) {
      for (int r = 1; r < world; ++r) {
        if (fds[r] < 0) continue;
        if (rc == RGBID_OK && !send_all(fds[r], all, (size_t)world * n)) rc = RGBID_E_NET;
        ::close(fds[r]);
      }
    }
    ::close(ls);
    return rc;
  }
  // ranks > 0: rank 0 may not be listening yet, may still be serving the previous exchange (NAK), or the connection may break before the payload
  // went through -- close, wait, try again until the deadline
  const Hello hello{HELLO_MAGIC, rank, nonce, kind, seq, (uint64_t)n};
  while (ms_left(deadline) > 0) {
    for (addrinfo* a = res; a; a = a->ai_next) {
      int fd = ::socket(a->ai_family, a->ai_socktype, a->ai_protocol);
      if (fd < 0) continue;
      if (::connect(fd, a->ai_addr, a->ai_addrlen) == 0) {
        set_io_timeout(fd, 5);
        unsigned char answer = NAK;
        bool ok = send_all(fd, &hello, sizeof(hello)) && recv_all(fd, &answer, 1) && answer == ACK;
        if (ok) {
          set_io_timeout(fd, timeout_s());
          if (gather) ok = send_all(fd, blob, n) && recv_all(fd, all, (size_t)world * n);
          else ok = recv_all(fd, blob, n);
        }
        ::close(fd);
        if (ok) { freeaddrinfo(res); return RGBID_OK; }
        break;                                             // refused or broken: back off, then start over
      }
      ::close(fd);
    }
    usleep(50 * 1000);
  }
  freeaddrinfo(res);
  return RGBID_E_NET;
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
  return tcp_exchange(addr, port, world, rank, blob, n, nullptr, false);
}

int rgbid_dist_allgather_bytes_tcp(const char* addr, int port, int world, int rank, const void* mine, size_t n, void* all) {
  if (!mine || !all || world < 1 || rank < 0 || rank >= world) return RGBID_E_INVALID;
  if (world == 1) { memcpy(all, mine, n); return RGBID_OK; }
  return tcp_exchange(addr, port, world, rank, const_cast<void*>(mine), n, all, true);
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

int rgbid_dist_renumber_warmed_chunks(rgbid_gather_record* all, int world, int lanes_per_rank, int n_chunks, int chunk_len, const int* first, const int* last,
                                      int warmup_frames) {
  if (!all || !first || !last || world < 1 || lanes_per_rank < 1 || n_chunks < 1 || chunk_len < 1 || warmup_frames < 0) return RGBID_E_INVALID;
  // a warmed-up lane numbers its chunk's first frame wc (its warm-up frames came first), not 0: renumber the chunk from its own head, so that the
  // composition's check of the ids (0 on the head, then 1 .. j, non-decreasing) holds what it held before.  A head that is not ahead of 0 although the
  // chunk warmed up, i.e. a lane that did not run its warm-up, is refused.
  for (int c = 0; c < n_chunks; ++c) {
    const int wc = std::min(warmup_frames, first[c]);
    if (!wc) continue;
    int s0 = 0, cnt = 0, owner = -1;
    for (int rk = 0; rk < world; ++rk) { rgbid_dist_rank_chunks(n_chunks, world, rk, &s0, &cnt); if (c >= s0 && c < s0 + cnt) { owner = rk; break; } }
    if (owner < 0 || c - s0 >= lanes_per_rank || last[c] - first[c] + 1 > chunk_len) return RGBID_E_INVALID;
    rgbid_gather_record* rec = all + ((size_t)owner * lanes_per_rank + (c - s0)) * chunk_len;
    const int head = rec[0].frame_id;
    if (head < 1 || head > wc) return RGBID_E_INVALID;
    // a lane that is lost across the chunk boundary (lost during warm-up, still lost on the chunk's first transitions) repeats the head's id
    // (visodo.cpp:2051-2117): after renumbering such a frame would read 0, which only a chunk HEAD may carry -- it keeps the smallest id of a
    // tracked position (1) and its LOST status, which is what the composed trajectory reports for it
    for (int j = 0; j < last[c] - first[c] + 1; ++j) { rec[j].frame_id -= head; if (j && rec[j].frame_id < 1) rec[j].frame_id = 1; }
  }
  return RGBID_OK;
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
    // a lane that really tracked chunk c numbers its frames 0, 1, 2, ... (rgbid_gather_record.frame_id = global_time_ of the frame inside the
    // lane's run; a frame that stays lost repeats the id, visodo.cpp:2051-2117): non-decreasing, 0 exactly on the first frame, never ahead
    // of the frame's position.  A padded lane, a zeroed buffer or a lane that tracked another chunk length fails this.
    for (int j = 0; j < n; ++j) {
      const int id = rec[j].frame_id;
      if (j == 0 ? id != 0 : (id < 1 || id > j || id < rec[j - 1].frame_id)) return RGBID_E_INVALID;
    }
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

// ---- the sharded-sequence driver ----------------------------------------------------------------------------------------------------------
namespace {
struct SeqScratch {   // everything the driver allocates, released on every exit path
  rgbid_engine* eng = nullptr;
  rgbid_dist* comm = nullptr;
  void *d_depth = nullptr, *d_rgb = nullptr, *d_local = nullptr, *d_all = nullptr;
  hipStream_t copy = nullptr;
  std::vector<hipEvent_t> ev;
  ~SeqScratch() {
    if (eng) rgbid_engine_destroy(eng);
    if (comm) rgbid_dist_destroy(comm);
    for (hipEvent_t e : ev) (void)hipEventDestroy(e);
    if (copy) (void)hipStreamDestroy(copy);
    for (void* p : {d_depth, d_rgb, d_local, d_all}) if (p) (void)hipFree(p);
  }
};
struct AsyncGuard {   // the driver enqueues asynchronously; the caller's context gets its own mode back on every exit path
  rgbid_ctx* c; int was = 0;
  explicit AsyncGuard(rgbid_ctx* c_) : c(c_) { rgbid_ctx_get_async(c, &was); rgbid_ctx_set_async(c, 1); }
  ~AsyncGuard() { rgbid_ctx_sync(c); rgbid_ctx_set_async(c, was); }
};
inline double ms_since(Clock::time_point t0) { return std::chrono::duration<double, std::milli>(Clock::now() - t0).count(); }
}  // namespace

int rgbid_dist_track_sequence(rgbid_ctx* ctx, const rgbid_seq_config* cfg, const uint16_t* depth_host, const uint8_t* rgb_host, int n_frames,
                              const rgbid_gather_record* inject, double* R, double* t, int* status, double* cov, rgbid_seq_report* report) {
  if (!cfg || !R || !t || cfg->world < 1 || cfg->rank < 0 || cfg->rank >= cfg->world || cfg->n_chunks < cfg->world || n_frames < cfg->n_chunks + 1 ||
      (cfg->exchange != RGBID_EXCHANGE_RCCL && cfg->exchange != RGBID_EXCHANGE_TCP))
    return RGBID_E_INVALID;
  if (rgbid_engine_config_size() != sizeof(rgbid_engine_config)) {   // librgbid_hip.so from another revision of rgbid_engine.h
    fprintf(stderr, "rgbid_dist_track_sequence: librgbid_hip.so and librgbid_dist.so disagree on rgbid_engine_config (%zu vs %zu bytes): rebuild both\n", rgbid_engine_config_size(),
            sizeof(rgbid_engine_config));
    return RGBID_E_INVALID;
  }
  if (!inject && (!ctx || !depth_host || !rgb_host)) return RGBID_E_INVALID;
  if (inject && cfg->world > 1 && cfg->exchange != RGBID_EXCHANGE_TCP) return RGBID_E_INVALID;   // no GPU side: nothing for RCCL to gather from
  if (cfg->world > 1 && (!cfg->master_addr || cfg->master_port <= 0)) return RGBID_E_INVALID;
  if (cfg->world > 1) reset_exchange_seq(cfg->master_port, cfg->rank);   // a job's exchanges count from 0 (counters of a restarted rank / a reused port)
  const int world = cfg->world, rank = cfg->rank, n_chunks = cfg->n_chunks;
  const auto t_setup = Clock::now();
  std::vector<int> first(n_chunks), last(n_chunks);
  int r = rgbid_dist_chunk_ranges(n_frames, n_chunks, first.data(), last.data());
  if (r) return r;
  int start = 0, count = 0;
  rgbid_dist_rank_chunks(n_chunks, world, rank, &start, &count);
  const int lanes = (n_chunks + world - 1) / world;
  int L = 0;
  for (int c = 0; c < n_chunks; ++c) L = std::max(L, last[c] - first[c] + 1);
  // lane -> chunk: the rank's block of chunks; a rank that owns one chunk fewer pads with a lane that re-tracks its last chunk (never read)
  std::vector<int> owned(lanes);
  for (int l = 0; l < lanes; ++l) owned[l] = count ? start + std::min(l, count - 1) : 0;
  const size_t n_local = (size_t)lanes * L;
  std::vector<rgbid_gather_record> all((size_t)world * n_local);
  SeqScratch S;
  rgbid_seq_report rep{};
  rep.lanes = lanes; rep.chunk_len = L; rep.n_chunks = n_chunks; rep.world = world; rep.rccl_ranks = 0;
  double track_ms = 0.0, gather_ms = 0.0;

  const int W = cfg->warmup_frames;
  if (W < 0 || W > 64 || (inject && W != 0)) return RGBID_E_INVALID;   // injected records carry no warm-up
  if (inject) {
    if (cfg->inject_chunk_len != L) return RGBID_E_INVALID;   // the caller's buffer is laid out for another chunk length
    rep.setup_ms = ms_since(t_setup);
    const auto t0 = Clock::now();
    std::vector<rgbid_gather_record> local(n_local);
    for (int l = 0; l < lanes; ++l) memcpy(&local[(size_t)l * L], inject + (size_t)owned[l] * L, sizeof(rgbid_gather_record) * L);
    if (count < lanes) for (int l = count; l < lanes; ++l) for (int j = 0; j < L; ++j) local[(size_t)l * L + j].frame_id = -1;   // padding lanes are not chunks
    track_ms = ms_since(t0);
    const auto t1 = Clock::now();
    r = rgbid_dist_allgather_bytes_tcp(cfg->master_addr, cfg->master_port, world, rank, local.data(), n_local * sizeof(rgbid_gather_record), all.data());
    if (r) return r;
    gather_ms = ms_since(t1);
  } else {
    const rgbid_engine_config& ec = cfg->engine;
    if (ec.rows <= 0 || ec.cols <= 0) return RGBID_E_INVALID;
    void* sv = nullptr;
    if ((r = rgbid_ctx_get_stream(ctx, &sv))) return r;
    hipStream_t es = (hipStream_t)sv;
    int dev = 0;
    if (hipError_t he = hipStreamGetDevice(es, &dev); he != hipSuccess) return (int)he;
    if (hipError_t he = hipSetDevice(dev); he != hipSuccess) return (int)he;
    if (world > 1 && cfg->exchange == RGBID_EXCHANGE_RCCL) {
      rgbid_dist_id id;
      if ((r = rgbid_dist_exchange_id(cfg->master_addr, cfg->master_port, world, rank, &id))) return r;
      if ((r = rgbid_dist_init(&S.comm, ctx, &id, world, rank))) return r;
      rep.rccl_ranks = rgbid_dist_world(S.comm);
    }
    rgbid_engine_config e2 = ec;
    e2.lanes = lanes; e2.record_capacity = L + W; e2.use_graph = 0;   // eager steps read the staged frames in place
    if ((r = rgbid_engine_create(&S.eng, ctx, &e2))) return r;
    size_t eb = 0; rgbid_engine_bytes(S.eng, &eb); rep.engine_bytes = eb;
    const size_t fd = (size_t)ec.rows * ec.cols * 2, fc = (size_t)ec.rows * ec.cols * 3;
    const int steps = L + W;                                   // lock-step steps: W warm-up steps, then the chunk
    const size_t n_staged = (size_t)lanes * steps;
    rep.staged_bytes = (fd + fc) * n_staged;
    hipError_t he = hipMalloc(&S.d_depth, fd * n_staged);
    if (he == hipSuccess) he = hipMalloc(&S.d_rgb, fc * n_staged);
    if (he == hipSuccess) he = hipMalloc(&S.d_local, sizeof(rgbid_gather_record) * n_local);
    if (he == hipSuccess && world > 1 && S.comm) he = hipMalloc(&S.d_all, sizeof(rgbid_gather_record) * n_local * world);
    if (he == hipSuccess) he = hipStreamCreateWithFlags(&S.copy, hipStreamNonBlocking);
    S.ev.resize(steps, nullptr);
    for (int j = 0; j < steps && he == hipSuccess; ++j) he = hipEventCreateWithFlags(&S.ev[j], hipEventDisableTiming);
    if (he != hipSuccess) { (void)hipGetLastError(); return he == hipErrorOutOfMemory ? RGBID_E_NOMEM : (int)he; }
    AsyncGuard async_guard(ctx);   // declared after S: the stream is drained before the engine / staging buffers go
    rgbid_ctx_sync(ctx);
    rep.setup_ms = ms_since(t_setup);
    if (S.comm && (r = rgbid_dist_barrier(S.comm))) return r;   // ranks start their clocks together
    // ---- uploads on the copy stream, one event per step; step j waits for its frames only, so step j + 1's frames travel while step j runs
    const auto t0 = Clock::now();
    // warm-up (cfg->warmup_frames): lane l of chunk c starts at frame first[c] - wc, wc = min(W, first[c]); step j of the lock-step run feeds it frame
    // first[c] - W + j once that frame exists (j >= W - wc) and lets it sit the step out before (rgbid_engine_set_active), so that EVERY lane takes its
    // chunk's first frame at step W and the records of steps W .. W + L - 1 are the chunk's
    std::vector<int> active(lanes, 1), active_prev(lanes, 1);
    for (int j = 0; j < steps; ++j) {
      for (int l = 0; l < lanes; ++l) {
        const int c = owned[l];
        const int wc = std::min(W, first[c]);
        active[l] = j >= W - wc;
        const size_t k = (size_t)std::min(std::max(first[c] - W + j, first[c] - wc), last[c]);   // a shorter chunk repeats its last frame (its records past the chunk are not read)
        he = hipMemcpyAsync((char*)S.d_depth + ((size_t)j * lanes + l) * fd, (const char*)depth_host + k * fd, fd, hipMemcpyHostToDevice, S.copy);
        if (he == hipSuccess) he = hipMemcpyAsync((char*)S.d_rgb + ((size_t)j * lanes + l) * fc, rgb_host + k * fc, fc, hipMemcpyHostToDevice, S.copy);
        if (he != hipSuccess) { (void)hipGetLastError(); return (int)he; }
      }
      he = hipEventRecord(S.ev[j], S.copy);
      if (he != hipSuccess) { (void)hipGetLastError(); return (int)he; }
      if (W && (j == 0 || active != active_prev)) { if ((r = rgbid_engine_set_active(S.eng, active.data()))) return r; active_prev = active; }
      if ((r = rgbid_ctx_wait_event(ctx, S.ev[j]))) return r;
      if ((r = rgbid_engine_step(S.eng, (char*)S.d_depth + (size_t)j * lanes * fd, (char*)S.d_rgb + (size_t)j * lanes * fc))) return r;
    }
    if ((r = rgbid_engine_pack_gather_records(S.eng, W, L, (rgbid_gather_record*)S.d_local))) return r;
    if ((r = rgbid_ctx_sync(ctx))) return r;
    track_ms = ms_since(t0);
    // ---- the exchange: ONE all-gather of the records
    const auto t1 = Clock::now();
    if (world > 1 && S.comm) {
      if ((r = rgbid_dist_gather_records(S.comm, (const rgbid_gather_record*)S.d_local, (int)n_local, (rgbid_gather_record*)S.d_all))) return r;
      he = hipMemcpyAsync(all.data(), S.d_all, sizeof(rgbid_gather_record) * n_local * world, hipMemcpyDeviceToHost, es);
      if (he == hipSuccess) he = hipStreamSynchronize(es);
      if (he != hipSuccess) { (void)hipGetLastError(); return (int)he; }
    } else {
      std::vector<rgbid_gather_record> local(n_local);
      he = hipMemcpyAsync(local.data(), S.d_local, sizeof(rgbid_gather_record) * n_local, hipMemcpyDeviceToHost, es);
      if (he == hipSuccess) he = hipStreamSynchronize(es);
      if (he != hipSuccess) { (void)hipGetLastError(); return (int)he; }
      if (count < lanes) for (int l = count; l < lanes; ++l) for (int j = 0; j < L; ++j) local[(size_t)l * L + j].frame_id = -1;
      if ((r = rgbid_dist_allgather_bytes_tcp(cfg->master_addr, cfg->master_port, world, rank, local.data(), n_local * sizeof(rgbid_gather_record), all.data()))) return r;
    }
    gather_ms = ms_since(t1);
  }
  const auto t2 = Clock::now();
  if (W) { r = rgbid_dist_renumber_warmed_chunks(all.data(), world, lanes, n_chunks, L, first.data(), last.data(), W); if (r) return r; }
  r = rgbid_dist_compose_trajectory(all.data(), world, lanes, n_chunks, L, first.data(), last.data(), R, t, status, cov);
  if (r) return r;
  rep.track_ms = track_ms; rep.gather_ms = gather_ms; rep.compose_ms = ms_since(t2);
  rep.total_ms = rep.track_ms + rep.gather_ms + rep.compose_ms;
  if (report) *report = rep;
  return RGBID_OK;
}

}  // extern "C"
