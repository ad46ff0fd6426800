// RCCL communicator for the multi-GPU database build.  librccl.so is opened
// lazily with dlopen so single-GPU users never load it; one rank per GPU, the ncclUniqueId
// travels between ranks through the host (any side channel the caller has: a file, a socket, MPI, a key-value store).
// A communicator owns a second stream: the runs of the gathered build travel on it while the context's own stream
// fingerprints the next batch.
#include <dlfcn.h>

#include <algorithm>

#include "shz_internal.h"

typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef int ncclResult_t;
// ncclDataType_t values used here (rccl.h): ncclUint8 = 1, ncclUint32 = 3, ncclUint64 = 5
enum { NCCL_U8 = 1, NCCL_U32 = 3, NCCL_U64 = 5 };

struct rccl_api {
  void* lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Broadcast)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Send)(const void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

static rccl_api* rccl() {
  static rccl_api api;
  static bool tried = false;
  if (!tried) {
    tried = true;
    const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
    for (const char* n : names) {
      api.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
      if (api.lib) break;
    }
    if (api.lib) {
      *(void**)&api.GetUniqueId = dlsym(api.lib, "ncclGetUniqueId");
      *(void**)&api.CommInitRank = dlsym(api.lib, "ncclCommInitRank");
      *(void**)&api.CommDestroy = dlsym(api.lib, "ncclCommDestroy");
      *(void**)&api.AllGather = dlsym(api.lib, "ncclAllGather");
      *(void**)&api.Broadcast = dlsym(api.lib, "ncclBroadcast");
      *(void**)&api.Send = dlsym(api.lib, "ncclSend");
      *(void**)&api.Recv = dlsym(api.lib, "ncclRecv");
      *(void**)&api.GroupStart = dlsym(api.lib, "ncclGroupStart");
      *(void**)&api.GroupEnd = dlsym(api.lib, "ncclGroupEnd");
      *(void**)&api.GetErrorString = dlsym(api.lib, "ncclGetErrorString");
      if (!api.GetUniqueId || !api.CommInitRank || !api.CommDestroy || !api.AllGather || !api.Broadcast) {
        dlclose(api.lib);
        api.lib = nullptr;
      }
    }
  }
  return api.lib ? &api : nullptr;
}

// ---- the in-process transport: ranks are THREADS of one process (one context each, on one device or several) -------
// Same rank protocol as over RCCL -- the table code cannot tell the difference -- but an exchange is a rendezvous of the
// threads and device-to-device copies out of the peers' buffers.  What it is for: the N > 1 logic of the sharded build
// (counts and layouts agreed between ranks, runs placed behind each other, the k-way merge over ranks' runs, the all-to-all
// of the key-sharded table) runs on a ONE-GPU box, which the build leases; and a single process that drives several GPUs
// can use it as is (copies between devices go over xGMI peer access).
#include <condition_variable>
#include <map>
#include <mutex>
struct local_group {
  std::mutex mu;
  std::condition_variable cv;
  int nranks = 0, arrived = 0, attached = 0;
  uint64_t generation = 0;
  std::vector<const void*> send;                 // what every rank published for the exchange in flight
  std::vector<std::vector<uint64_t>> displ;      // alltoallv: the publisher's send displacements
  std::vector<std::vector<const void*>> sendv;   // all-gather of lists: the publisher's buffers
  bool broken = false;                           // a rank gave up waiting: everybody fails from then on
};
static std::mutex g_groups_mu;
static std::map<uint64_t, local_group*> g_groups;

// rendezvous of all ranks; false after 120 s without the others (a rank died or never came)
static bool local_barrier(local_group* g) {
  std::unique_lock<std::mutex> lk(g->mu);
  if (g->broken) return false;
  const uint64_t gen = g->generation;
  if (++g->arrived == g->nranks) {
    g->arrived = 0;
    ++g->generation;
    g->cv.notify_all();
    return true;
  }
  if (!g->cv.wait_for(lk, std::chrono::seconds(120), [&] { return g->generation != gen || g->broken; })) g->broken = true;
  if (g->broken) { g->cv.notify_all(); return false; }
  return true;
}

struct shz_comm {
  shz_ctx* ctx;
  ncclComm_t comm;
  int rank, nranks;
  local_group* lg = nullptr;   // non-null: the in-process transport
  uint64_t lg_id = 0;
  hipStream_t xs = nullptr;    // the exchange stream (created on first use)
  void* d_info = nullptr;      // small device block for the rounds' info exchange (not a context workspace slot: the
  void* h_info = nullptr;      //   context's stream may be busy with other work), and its pinned host mirror
};
#define SHZ_COMM_INFO_BYTES (64u << 10)

// one exchange of the in-process transport: every rank publishes its send buffer (data complete: own stream drained), all
// meet, `copy(p, send pointer of rank p, displacements of rank p)` pulls what this rank wants from rank p, all meet again
// (nobody touches its send buffer before every peer has read it)
template <class F>
static int32_t local_exchange(shz_comm* c, hipStream_t s, const void* d_send, const uint64_t* my_displ, F&& copy,
                              const std::vector<const void*>* my_list = nullptr) {
  shz_ctx* ctx = c->ctx;
  local_group* g = c->lg;
  SHZ_HIP(ctx, hipStreamSynchronize(s));
  {
    std::lock_guard<std::mutex> lk(g->mu);
    g->send[c->rank] = d_send;
    if (my_displ) g->displ[c->rank].assign(my_displ, my_displ + c->nranks);
    if (my_list) g->sendv[c->rank] = *my_list;
  }
  if (!local_barrier(g)) SHZ_FAIL(ctx, SHZ_E_RCCL, "in-process exchange: a rank did not arrive");
  for (int p = 0; p < c->nranks; ++p) {
    const void* sp;
    std::vector<uint64_t> dp;
    {
      std::lock_guard<std::mutex> lk(g->mu);
      sp = g->send[p];
      if (my_displ) dp = g->displ[p];
    }
    const hipError_t e = copy(p, sp, dp);   // (a list exchange reads g->sendv[p] itself: nobody writes it between the barriers)
    if (e != hipSuccess) {
      { std::lock_guard<std::mutex> lk(g->mu); g->broken = true; }
      g->cv.notify_all();
      SHZ_FAIL(ctx, SHZ_E_HIP, "in-process exchange: copy from rank %d failed: %s", p, hipGetErrorString(e));
    }
  }
  SHZ_HIP(ctx, hipStreamSynchronize(s));
  if (!local_barrier(g)) SHZ_FAIL(ctx, SHZ_E_RCCL, "in-process exchange: a rank did not arrive");
  return SHZ_OK;
}

extern "C" int32_t shz_comm_create_local(shz_ctx* ctx, uint64_t group_id, int32_t rank, int32_t nranks, shz_comm** out) {
  if (!ctx || !out || nranks < 1 || rank < 0 || rank >= nranks) return SHZ_E_INVALID;
  local_group* g;
  {
    std::lock_guard<std::mutex> lk(g_groups_mu);
    auto it = g_groups.find(group_id);
    if (it == g_groups.end()) {
      g = new local_group();
      g->nranks = nranks;
      g->send.assign(nranks, nullptr);
      g->displ.resize(nranks);
      g->sendv.resize(nranks);
      g_groups[group_id] = g;
    } else {
      g = it->second;
      if (g->nranks != nranks) SHZ_FAIL(ctx, SHZ_E_INVALID, "in-process group %llu has %d ranks, not %d", (unsigned long long)group_id, g->nranks, nranks);
    }
    ++g->attached;
  }
  shz_comm* cm = new shz_comm{ctx, nullptr, rank, nranks};
  cm->lg = g;
  cm->lg_id = group_id;
  *out = cm;
  return SHZ_OK;
}

#define SHZ_NCCL(ctx, call)                                                                          \
  do {                                                                                               \
    ncclResult_t _r = (call);                                                                        \
    if (_r != 0) SHZ_FAIL(ctx, SHZ_E_RCCL, "%s failed: %s", #call,                                    \
                          rccl()->GetErrorString ? rccl()->GetErrorString(_r) : "rccl error");       \
  } while (0)

extern "C" int32_t shz_comm_unique_id(uint8_t id_out[128]) {
  rccl_api* r = rccl();
  if (!r || !id_out) return SHZ_E_RCCL;
  ncclUniqueId id;
  if (r->GetUniqueId(&id) != 0) return SHZ_E_RCCL;
  memcpy(id_out, id.internal, 128);
  return SHZ_OK;
}

extern "C" int32_t shz_comm_create(shz_ctx* ctx, const uint8_t idb[128], int32_t rank, int32_t nranks, shz_comm** out) {
  if (!ctx || !idb || !out || nranks < 1 || rank < 0 || rank >= nranks) return SHZ_E_INVALID;
  rccl_api* r = rccl();
  if (!r) SHZ_FAIL(ctx, SHZ_E_RCCL, "librccl.so could not be loaded: %s", dlerror());
  SHZ_HIP(ctx, hipSetDevice(ctx->device));
  ncclUniqueId id;
  memcpy(id.internal, idb, 128);
  ncclComm_t c;
  SHZ_NCCL(ctx, r->CommInitRank(&c, nranks, id, rank));
  shz_comm* cm = new shz_comm{ctx, c, rank, nranks};
  *out = cm;
  return SHZ_OK;
}

extern "C" int32_t shz_comm_destroy(shz_comm* c) {
  if (!c) return SHZ_E_INVALID;
  (void)hipSetDevice(c->ctx->device);
  (void)hipStreamSynchronize(c->ctx->stream);
  if (c->xs) { (void)hipStreamSynchronize(c->xs); (void)hipStreamDestroy(c->xs); }
  if (c->d_info) (void)hipFree(c->d_info);
  if (c->h_info) (void)hipHostFree(c->h_info);
  if (c->lg) {
    std::lock_guard<std::mutex> lk(g_groups_mu);
    if (--c->lg->attached == 0) {
      g_groups.erase(c->lg_id);
      delete c->lg;
    }
  } else {
    rccl()->CommDestroy(c->comm);
  }
  delete c;
  return SHZ_OK;
}

// internal: used by shz_table_allgather (shz_table.hip)
shz_ctx* shz_comm_ctx(shz_comm* c) { return c->ctx; }
int32_t shz_comm_info(shz_comm* c, int* rank, int* nranks) {
  *rank = c->rank;
  *nranks = c->nranks;
  return SHZ_OK;
}

// the exchange stream and the info blocks, made on first use
int32_t shz_comm_exchange_stream(shz_comm* c, hipStream_t* out) {
  shz_ctx* ctx = c->ctx;
  if (!c->xs) {
    SHZ_HIP(ctx, hipSetDevice(ctx->device));
    SHZ_HIP(ctx, hipStreamCreateWithFlags(&c->xs, hipStreamNonBlocking));
    SHZ_HIP(ctx, hipMalloc(&c->d_info, 2 * SHZ_COMM_INFO_BYTES));
    SHZ_HIP(ctx, hipHostMalloc(&c->h_info, 2 * SHZ_COMM_INFO_BYTES, hipHostMallocDefault));
  }
  *out = c->xs;
  return SHZ_OK;
}

// all-gather of equal-sized byte blocks: recv must hold nranks*bytes
int32_t shz_comm_allgather_bytes_on(shz_comm* c, hipStream_t s, const void* d_send, void* d_recv, uint64_t bytes) {
  shz_ctx* ctx = c->ctx;
  if (c->lg)
    return local_exchange(c, s, d_send, nullptr, [&](int p, const void* sp, const std::vector<uint64_t>&) {
      return bytes ? hipMemcpyAsync((char*)d_recv + (uint64_t)p * bytes, sp, bytes, hipMemcpyDefault, s) : hipSuccess;
    });
  SHZ_NCCL(ctx, rccl()->AllGather(d_send, d_recv, bytes, NCCL_U8, c->comm, s));
  return SHZ_OK;
}
int32_t shz_comm_allgather_bytes(shz_comm* c, const void* d_send, void* d_recv, uint64_t bytes) {
  return shz_comm_allgather_bytes_on(c, c->ctx->stream, d_send, d_recv, bytes);
}

// All-gather of a HOST block of `bytes` (<= SHZ_COMM_INFO_BYTES / nranks) per rank on the exchange stream: h_all receives
// nranks blocks in rank order.  Blocks until the blocks are there -- and, the stream being in order, until every
// transfer queued on it before has finished.
int32_t shz_comm_allgather_host(shz_comm* c, const void* h_mine, void* h_all, uint64_t bytes) {
  shz_ctx* ctx = c->ctx;
  hipStream_t s;
  SHZ_TRY(shz_comm_exchange_stream(c, &s));
  if (bytes * (uint64_t)c->nranks > SHZ_COMM_INFO_BYTES) SHZ_FAIL(ctx, SHZ_E_INVALID, "info block of %llu bytes x %d ranks", (unsigned long long)bytes, c->nranks);
  memcpy(c->h_info, h_mine, bytes);
  char* d_recv = (char*)c->d_info + SHZ_COMM_INFO_BYTES;
  char* h_recv = (char*)c->h_info + SHZ_COMM_INFO_BYTES;
  SHZ_HIP(ctx, hipMemcpyAsync(c->d_info, c->h_info, bytes, hipMemcpyHostToDevice, s));
  SHZ_TRY(shz_comm_allgather_bytes_on(c, s, c->d_info, d_recv, bytes));
  SHZ_HIP(ctx, hipMemcpyAsync(h_recv, d_recv, bytes * c->nranks, hipMemcpyDeviceToHost, s));
  SHZ_HIP(ctx, hipStreamSynchronize(s));
  memcpy(h_all, h_recv, bytes * c->nranks);
  return SHZ_OK;
}

// All-gather of LISTS of buffers on stream s: every rank's buffers send[0..] go to every peer; recv[p] says where the
// buffers of peer p land here (same number and sizes as p's own send list -- every rank knows every list from the
// info exchange before).  A rank's own buffers stay where they are (recv[rank] is ignored).  Returns when the
// transfers are QUEUED (RCCL) -- the caller orders other streams against s with events; the in-process transport
// returns when they are done.  RCCL: the lists are cut into pieces of <= 1 GB; piece i of every rank travels in one
// group -- a send to and a receive from each peer, every pair of GPUs on its own xGMI link.
int32_t shz_comm_allgather_lists_on(shz_comm* c, hipStream_t s, const std::vector<shz_xfer>& send,
                                    const std::vector<std::vector<shz_xfer>>& recv) {
  shz_ctx* ctx = c->ctx;
  if ((int)recv.size() != c->nranks) SHZ_FAIL(ctx, SHZ_E_INVALID, "allgather_lists: %zu receive lists for %d ranks", recv.size(), c->nranks);
  if (c->lg) {
    std::vector<const void*> mine;
    for (const shz_xfer& x : send) mine.push_back(x.p);
    local_group* g = c->lg;
    return local_exchange(c, s, nullptr, nullptr, [&](int p, const void*, const std::vector<uint64_t>&) -> hipError_t {
      if (p == c->rank) return hipSuccess;
      std::vector<const void*> theirs;
      { std::lock_guard<std::mutex> lk(g->mu); theirs = g->sendv[p]; }
      if (theirs.size() != recv[p].size()) return hipErrorInvalidValue;
      for (size_t i = 0; i < theirs.size(); ++i)
        if (recv[p][i].bytes) {
          const hipError_t e = hipMemcpyAsync(recv[p][i].p, theirs[i], recv[p][i].bytes, hipMemcpyDefault, s);
          if (e != hipSuccess) return e;
        }
      return hipSuccess;
    }, &mine);
  }
  if (c->nranks == 1) return SHZ_OK;
  rccl_api* r = rccl();
  if (!r->Send || !r->Recv || !r->GroupStart || !r->GroupEnd) SHZ_FAIL(ctx, SHZ_E_RCCL, "librccl.so lacks ncclSend/ncclRecv");
  const uint64_t PIECE = 1ull << 30;
  auto pieces = [&](const std::vector<shz_xfer>& l) {
    std::vector<shz_xfer> v;
    for (const shz_xfer& x : l)
      for (uint64_t o = 0; o < x.bytes; o += PIECE) v.push_back(shz_xfer{(char*)x.p + o, std::min(PIECE, x.bytes - o)});
    return v;
  };
  const std::vector<shz_xfer> mine = pieces(send);
  std::vector<std::vector<shz_xfer>> theirs(c->nranks);
  size_t rounds = mine.size();
  for (int p = 0; p < c->nranks; ++p)
    if (p != c->rank) { theirs[p] = pieces(recv[p]); rounds = std::max(rounds, theirs[p].size()); }
  for (size_t i = 0; i < rounds; ++i) {
    ncclResult_t bad = (ncclResult_t)0;
    const char* what = "";
    auto step = [&](ncclResult_t rc, const char* w) { if (rc != 0 && bad == 0) { bad = rc; what = w; } };
    step(r->GroupStart(), "ncclGroupStart");
    for (int p = 0; p < c->nranks && bad == 0; ++p) {
      if (p == c->rank) continue;
      if (i < mine.size()) step(r->Send(mine[i].p, mine[i].bytes, NCCL_U8, p, c->comm, s), "ncclSend");
      if (i < theirs[p].size()) step(r->Recv(theirs[p][i].p, theirs[p][i].bytes, NCCL_U8, p, c->comm, s), "ncclRecv");
    }
    step(r->GroupEnd(), "ncclGroupEnd");   // the group is closed whatever happened inside it
    if (bad != 0) SHZ_FAIL(ctx, SHZ_E_RCCL, "%s failed: %s", what, r->GetErrorString ? r->GetErrorString(bad) : "rccl error");
  }
  return SHZ_OK;
}

// Variable-size all-gather: rank r's block lands at d_recv + displ[r] on every rank.  Moved in pieces of <= 1 GB
// (SURVEY 8e), each piece one RCCL group in which all ranks' transfers are in flight together, so that every pair of
// GPUs uses its own xGMI link instead of a ring bound by one link.  Two exchange patterns, selected by SHZ_ALLGATHER:
//   sendrecv (default)  every rank ncclSend-s its piece to each peer and ncclRecv-s each peer's piece: an explicit mesh
//   bcast               one ncclBroadcast per root
// Whatever fails inside a group, the group is closed before the error is returned.
int32_t shz_comm_allgatherv_bytes(shz_comm* c, const void* d_send, void* d_recv, const uint64_t* counts,
                                  const uint64_t* displ) {
  shz_ctx* ctx = c->ctx;
  if (c->lg)
    return local_exchange(c, ctx->stream, d_send, nullptr, [&](int p, const void* sp, const std::vector<uint64_t>&) {
      return counts[p] ? hipMemcpyAsync((char*)d_recv + displ[p], sp, counts[p], hipMemcpyDefault, ctx->stream) : hipSuccess;
    });
  rccl_api* r = rccl();
  static const bool use_bcast = [] { const char* e = getenv("SHZ_ALLGATHER"); return e && !strcmp(e, "bcast"); }();
  const uint64_t PIECE = 1ull << 30;
  uint64_t longest = 0;
  for (int p = 0; p < c->nranks; ++p) longest = std::max(longest, counts[p]);
  if (counts[c->rank])   // the rank's own block: a device copy
    SHZ_HIP(ctx, shz_memcpy(ctx, (char*)d_recv + displ[c->rank], d_send, counts[c->rank], hipMemcpyDeviceToDevice));
  if (c->nranks == 1) return SHZ_OK;
  if (!r->GroupStart || !r->GroupEnd) SHZ_FAIL(ctx, SHZ_E_RCCL, "librccl.so lacks ncclGroupStart/End");
  if (!use_bcast && (!r->Send || !r->Recv)) SHZ_FAIL(ctx, SHZ_E_RCCL, "librccl.so lacks ncclSend/ncclRecv");
  for (uint64_t o = 0; o < longest; o += PIECE) {
    ncclResult_t bad = (ncclResult_t)0;
    const char* what = "";
    auto step = [&](ncclResult_t rc, const char* w) { if (rc != 0 && bad == 0) { bad = rc; what = w; } };
    step(r->GroupStart(), "ncclGroupStart");
    for (int p = 0; p < c->nranks && bad == 0; ++p) {
      const uint64_t mine = counts[c->rank] > o ? std::min(PIECE, counts[c->rank] - o) : 0;
      const uint64_t theirs = counts[p] > o ? std::min(PIECE, counts[p] - o) : 0;
      if (use_bcast) {
        if (!theirs) continue;
        char* dst = (char*)d_recv + displ[p] + o;
        step(r->Broadcast(dst, dst, theirs, NCCL_U8, p, c->comm, ctx->stream), "ncclBroadcast");   // in place: root holds it
      } else if (p != c->rank) {
        if (mine) step(r->Send((const char*)d_send + o, mine, NCCL_U8, p, c->comm, ctx->stream), "ncclSend");
        if (theirs) step(r->Recv((char*)d_recv + displ[p] + o, theirs, NCCL_U8, p, c->comm, ctx->stream), "ncclRecv");
      }
    }
    step(r->GroupEnd(), "ncclGroupEnd");
    if (bad != 0) SHZ_FAIL(ctx, SHZ_E_RCCL, "%s failed: %s", what, r->GetErrorString ? r->GetErrorString(bad) : "rccl error");
  }
  return SHZ_OK;
}

// all-to-all of variable-size byte blocks as one group of point-to-point transfers: the block for rank p
// starts at d_send + sdispl[p] (scount[p] bytes); the block from rank p lands at d_recv + rdispl[p].
// Every pair of GPUs exchanges directly over its own xGMI link; the block a rank keeps for itself is a
// device-to-device copy.
int32_t shz_comm_alltoallv_bytes(shz_comm* c, const void* d_send, const uint64_t* scount, const uint64_t* sdispl,
                                 void* d_recv, const uint64_t* rcount, const uint64_t* rdispl) {
  shz_ctx* ctx = c->ctx;
  if (c->lg)
    return local_exchange(c, ctx->stream, d_send, sdispl, [&](int p, const void* sp, const std::vector<uint64_t>& dp) {
      return rcount[p] ? hipMemcpyAsync((char*)d_recv + rdispl[p], (const char*)sp + dp[c->rank], rcount[p], hipMemcpyDefault, ctx->stream)
                       : hipSuccess;
    });
  rccl_api* r = rccl();
  if (scount[c->rank] != rcount[c->rank]) SHZ_FAIL(ctx, SHZ_E_INVALID, "alltoallv: self block sizes differ");
  if (scount[c->rank])
    SHZ_HIP(ctx, shz_memcpy(ctx, (char*)d_recv + rdispl[c->rank], (const char*)d_send + sdispl[c->rank], scount[c->rank],
                                hipMemcpyDeviceToDevice));
  if (c->nranks == 1) return SHZ_OK;
  if (!r->Send || !r->Recv || !r->GroupStart || !r->GroupEnd) SHZ_FAIL(ctx, SHZ_E_RCCL, "librccl.so lacks ncclSend/ncclRecv");
  ncclResult_t bad = (ncclResult_t)0;
  const char* what = "";
  auto step = [&](ncclResult_t rc, const char* w) { if (rc != 0 && bad == 0) { bad = rc; what = w; } };
  step(r->GroupStart(), "ncclGroupStart");
  for (int p = 0; p < c->nranks && bad == 0; ++p) {
    if (p == c->rank) continue;
    if (scount[p]) step(r->Send((const char*)d_send + sdispl[p], scount[p], NCCL_U8, p, c->comm, ctx->stream), "ncclSend");
    if (rcount[p]) step(r->Recv((char*)d_recv + rdispl[p], rcount[p], NCCL_U8, p, c->comm, ctx->stream), "ncclRecv");
  }
  step(r->GroupEnd(), "ncclGroupEnd");   // the group is closed whatever happened inside it
  if (bad != 0) SHZ_FAIL(ctx, SHZ_E_RCCL, "%s failed: %s", what, r->GetErrorString ? r->GetErrorString(bad) : "rccl error");
  return SHZ_OK;
}

// RCCL sets up the connection between two ranks when they first send to / receive from each other (hundreds of milliseconds
// for a node's 28 pairs); a build that is timed calls this first: one 320-byte block all-gathered and one 256-byte buffer sent
// to and received from every peer, on the exchange stream -- the two collectives the exchange rounds use.
extern "C" int32_t shz_comm_warmup(shz_comm* c) {
  if (!c) return SHZ_E_INVALID;
  shz_ctx* ctx = c->ctx;
  SHZ_HIP(ctx, hipSetDevice(ctx->device));
  hipStream_t s;
  SHZ_TRY(shz_comm_exchange_stream(c, &s));
  uint64_t blk[40] = {0}, all[40 * 64];
  if (c->nranks > 64) SHZ_FAIL(ctx, SHZ_E_INVALID, "warm-up: more than 64 ranks");
  SHZ_TRY(shz_comm_allgather_host(c, blk, all, sizeof(blk)));
  void* d = nullptr;
  SHZ_HIP(ctx, hipMalloc(&d, 256ull * (c->nranks + 1)));
  std::vector<shz_xfer> send{shz_xfer{d, 256}};
  std::vector<std::vector<shz_xfer>> recv(c->nranks);
  for (int p = 0; p < c->nranks; ++p)
    if (p != c->rank) recv[p].push_back(shz_xfer{(char*)d + 256ull * (p + 1), 256});
  const int32_t rc = shz_comm_allgather_lists_on(c, s, send, recv);
  const hipError_t e = hipStreamSynchronize(s);
  (void)hipFree(d);
  SHZ_TRY(rc);
  SHZ_HIP(ctx, e);
  return SHZ_OK;
}

extern "C" int32_t shz_comm_barrier(shz_comm* c) {
  if (!c) return SHZ_E_INVALID;
  shz_ctx* ctx = c->ctx;
  void* p;
  SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_MISC0, 64 + 8ull * c->nranks, &p));
  SHZ_TRY(shz_comm_allgather_bytes(c, p, (char*)p + 64, 8));
  SHZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return SHZ_OK;
}
