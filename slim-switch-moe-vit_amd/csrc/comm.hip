// Expert exchange at the C-ABI (fmoe_cuda.ensure_nccl / expert_exchange / global_scatter / global_gather; SURVEY.md N10-N13,
// section 8b `smoe_ctx_create`, `smoe_a2a_counts`, `smoe_a2a_tokens`): a context = one RCCL communicator built from a
// unique-id blob + ONE dedicated communication stream + a small ring of completion events.  Every exchange runs on the
// context's own stream, fenced against the caller's compute stream with events, so token rows move over xGMI while the
// compute stream keeps running expert GEMMs.  Every exchange gets a TICKET (smoe_a2a_last_ticket); the caller chooses when
// to wait and for which exchange (smoe_a2a_wait_ticket), so two exchanges in flight -- what a micro-batch pipeline creates
// -- do not serialise on one shared event (smoe_a2a_wait = "the latest exchange").
//
// RCCL is resolved at run time (dlopen of the librccl.so already in the process -- torch ships and loads one -- or the
// system's): the library itself has no link-time dependency on it, so single-GPU users never need it.
#include "smoe_common.h"
#include <dlfcn.h>
#include <string.h>
#include <rccl/rccl.h>
#include <mutex>
#include <vector>

namespace {

struct RcclApi {
  void* lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

RcclApi* rccl() {
  static RcclApi api;
  static std::once_flag once;
  std::call_once(once, [] {
    const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
    for (const char* n : names) {
      api.lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD);   // the copy already in the process (torch's), if any
      if (api.lib) break;
    }
    for (int i = 0; !api.lib && i < 3; ++i) api.lib = dlopen(names[i], RTLD_NOW | RTLD_GLOBAL);
    if (!api.lib) return;
#define SMOE_SYM(field, name) api.field = reinterpret_cast<decltype(api.field)>(dlsym(api.lib, name))
    SMOE_SYM(GetUniqueId, "ncclGetUniqueId");
    SMOE_SYM(CommInitRank, "ncclCommInitRank");
    SMOE_SYM(CommDestroy, "ncclCommDestroy");
    SMOE_SYM(Send, "ncclSend");
    SMOE_SYM(Recv, "ncclRecv");
    SMOE_SYM(GroupStart, "ncclGroupStart");
    SMOE_SYM(GroupEnd, "ncclGroupEnd");
    SMOE_SYM(GetErrorString, "ncclGetErrorString");
#undef SMOE_SYM
  });
  if (!api.lib || !api.GetUniqueId || !api.CommInitRank || !api.CommDestroy || !api.Send || !api.Recv || !api.GroupStart || !api.GroupEnd)
    return nullptr;
  return &api;
}

#define SMOE_NCCL(call, what)                                                                              \
  do {                                                                                                     \
    ncclResult_t r__ = (call);                                                                             \
    if (r__ != ncclSuccess) {                                                                              \
      smoe_set_error("%s: RCCL error %d (%s)", what, (int)r__, api->GetErrorString ? api->GetErrorString(r__) : "?"); \
      return 100 + (int)r__;                                                                               \
    }                                                                                                      \
  } while (0)
#define SMOE_HIP(call, what)                                                              \
  do {                                                                                    \
    hipError_t e__ = (call);                                                              \
    if (e__ != hipSuccess) {                                                              \
      smoe_set_error("%s: %s", what, hipGetErrorString(e__));                             \
      return (int)e__;                                                                    \
    }                                                                                     \
  } while (0)

}  // namespace

constexpr int SMOE_DONE_RING = 16;   // exchanges whose completion can be waited for individually

struct smoe_ctx {
  ncclComm_t comm = nullptr;
  int world = 1, rank = 0, device = 0;
  hipStream_t comm_stream = nullptr;
  hipEvent_t ready = nullptr;                 // recorded on the caller's stream: the send buffer is complete
  hipEvent_t done[SMOE_DONE_RING] = {};       // done[t % RING] recorded on the comm stream behind exchange t
  int64_t ticket = 0;                         // exchanges posted so far (ticket t = the t-th exchange, from 1)
};

static void ctx_free(smoe_ctx* c, RcclApi* api) {
  if (!c) return;
  if (c->comm_stream) (void)hipStreamSynchronize(c->comm_stream);
  if (api && c->comm) api->CommDestroy(c->comm);
  if (c->ready) (void)hipEventDestroy(c->ready);
  for (hipEvent_t e : c->done)
    if (e) (void)hipEventDestroy(e);
  if (c->comm_stream) (void)hipStreamDestroy(c->comm_stream);
  delete c;
}

extern "C" int smoe_unique_id_bytes(void) { return NCCL_UNIQUE_ID_BYTES; }

extern "C" int smoe_unique_id(void* out) {
  SMOE_REQUIRE(out, "smoe_unique_id: null pointer");
  RcclApi* api = rccl();
  SMOE_REQUIRE(api, "smoe_unique_id: librccl.so not found in this process");
  ncclUniqueId id;
  SMOE_NCCL(api->GetUniqueId(&id), "ncclGetUniqueId");
  memcpy(out, &id, sizeof(id));
  return 0;
}

extern "C" int smoe_ctx_create(const void* unique_id, int world_size, int rank, smoe_ctx** out) {
  SMOE_REQUIRE(unique_id && out && world_size >= 1 && rank >= 0 && rank < world_size, "smoe_ctx_create: bad arguments");
  RcclApi* api = rccl();
  SMOE_REQUIRE(api, "smoe_ctx_create: librccl.so not found in this process");
  *out = nullptr;
  smoe_ctx* c = new smoe_ctx();
  c->world = world_size;
  c->rank = rank;
  // a failed step frees what the earlier ones made (communicator, stream, events) before the error goes out
  auto build = [&]() -> int {
    SMOE_HIP(hipGetDevice(&c->device), "hipGetDevice");
    ncclUniqueId id;
    memcpy(&id, unique_id, sizeof(id));
    SMOE_NCCL(api->CommInitRank(&c->comm, world_size, id, rank), "ncclCommInitRank");
    SMOE_HIP(hipStreamCreateWithFlags(&c->comm_stream, hipStreamNonBlocking), "hipStreamCreate");
    SMOE_HIP(hipEventCreateWithFlags(&c->ready, hipEventDisableTiming), "hipEventCreate");
    for (hipEvent_t& e : c->done) SMOE_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming), "hipEventCreate");
    return 0;
  };
  if (int rc = build()) {
    ctx_free(c, api);
    return rc;
  }
  *out = c;
  return 0;
}

extern "C" int smoe_ctx_destroy(smoe_ctx* c) {
  ctx_free(c, rccl());
  return 0;
}

extern "C" void* smoe_ctx_comm_stream(smoe_ctx* c) { return c ? (void*)c->comm_stream : nullptr; }
extern "C" int smoe_ctx_world_size(smoe_ctx* c) { return c ? c->world : 0; }
extern "C" int smoe_ctx_rank(smoe_ctx* c) { return c ? c->rank : -1; }

// the compute stream's work so far -> visible to the comm stream
static int fence_in(smoe_ctx* c, hipStream_t s) {
  SMOE_HIP(hipEventRecord(c->ready, s), "hipEventRecord");
  SMOE_HIP(hipStreamWaitEvent(c->comm_stream, c->ready, 0), "hipStreamWaitEvent");
  return 0;
}

// the exchange just posted on the comm stream gets the next ticket and its slot of the completion ring
static int fence_out(smoe_ctx* c) {
  c->ticket += 1;
  SMOE_HIP(hipEventRecord(c->done[c->ticket % SMOE_DONE_RING], c->comm_stream), "hipEventRecord");
  return 0;
}

extern "C" int64_t smoe_a2a_last_ticket(smoe_ctx* c) { return c ? c->ticket : -1; }

// `stream` waits for exchange `ticket` (and, the comm stream being in order, for every earlier one).  A ticket older than
// the ring waits for the exchange that re-used its slot: later on the same stream, so still sufficient.
extern "C" int smoe_a2a_wait_ticket(smoe_ctx* c, int64_t ticket, void* stream) {
  SMOE_REQUIRE(c, "smoe_a2a_wait_ticket: null context");
  SMOE_REQUIRE(ticket >= 0 && ticket <= c->ticket, "smoe_a2a_wait_ticket: no such exchange");
  if (ticket == 0) return 0;   // nothing posted yet
  SMOE_HIP(hipStreamWaitEvent((hipStream_t)stream, c->done[ticket % SMOE_DONE_RING], 0), "hipStreamWaitEvent");
  return 0;
}

extern "C" int smoe_a2a_wait(smoe_ctx* c, void* stream) {
  SMOE_REQUIRE(c, "smoe_a2a_wait: null context");
  return smoe_a2a_wait_ticket(c, c->ticket, stream);
}

// An RCCL call that fails between GroupStart and GroupEnd must not leave the group open: every later RCCL call of the
// process (torch's communicator included) would queue into it.
#define SMOE_NCCL_IN_GROUP(call, what)                                                                     \
  do {                                                                                                     \
    ncclResult_t r__ = (call);                                                                             \
    if (r__ != ncclSuccess) {                                                                              \
      (void)api->GroupEnd();                                                                               \
      smoe_set_error("%s: RCCL error %d (%s)", what, (int)r__, api->GetErrorString ? api->GetErrorString(r__) : "?"); \
      return 100 + (int)r__;                                                                               \
    }                                                                                                      \
  } while (0)

// expert_exchange: send_counts[w * E_local + e] (rows this rank routes to rank w's local expert e) -> recv_counts[w * E_local
// + e] (rows rank w routes to this rank's local expert e).  Both i32 [W * E_local] in device memory.
extern "C" int smoe_a2a_counts(smoe_ctx* c, const int32_t* send_counts, int32_t* recv_counts, int E_local, void* stream,
                               int wait) {
  SMOE_REQUIRE(c && send_counts && recv_counts && E_local >= 1, "smoe_a2a_counts: bad arguments");
  RcclApi* api = rccl();
  SMOE_REQUIRE(api, "smoe_a2a_counts: librccl.so not found");
  const bool in_line = wait == SMOE_A2A_INLINE;   // posted on the caller's stream itself: no event on either side
  hipStream_t cs = in_line ? (hipStream_t)stream : c->comm_stream;
  if (!in_line)
    if (int rc = fence_in(c, (hipStream_t)stream)) return rc;
  SMOE_NCCL(api->GroupStart(), "ncclGroupStart");
  for (int w = 0; w < c->world; ++w) {
    SMOE_NCCL_IN_GROUP(api->Send(send_counts + (size_t)w * E_local, (size_t)E_local, ncclInt32, w, c->comm, cs), "ncclSend");
    SMOE_NCCL_IN_GROUP(api->Recv(recv_counts + (size_t)w * E_local, (size_t)E_local, ncclInt32, w, c->comm, cs), "ncclRecv");
  }
  SMOE_NCCL(api->GroupEnd(), "ncclGroupEnd");
  if (in_line) return 0;
  if (int rc = fence_out(c)) return rc;
  if (wait) return smoe_a2a_wait(c, stream);
  return 0;
}

// global_scatter / global_gather: all-to-all-v of whole rows of d elements.  send holds the rows for rank 0, then rank 1, ...
// (send_rows[w] of them; the expert-sorted send buffer already has that order); recv receives recv_rows[w] rows from rank w,
// rank-major.  send_rows / recv_rows are HOST arrays [W] (they come from the count exchange).  All peers' transfers are
// posted in ONE group, so on the xGMI mesh every link carries its pair's rows concurrently.
extern "C" int smoe_a2a_tokens(smoe_ctx* c, const void* send, const int64_t* send_rows, void* recv, const int64_t* recv_rows,
                               int d, int dtype, void* stream, int wait) {
  SMOE_REQUIRE(c && send_rows && recv_rows && d > 0 && smoe_dtype_ok(dtype), "smoe_a2a_tokens: bad arguments");
  RcclApi* api = rccl();
  SMOE_REQUIRE(api, "smoe_a2a_tokens: librccl.so not found");
  const size_t es = (size_t)smoe_dtype_size(dtype);
  for (int w = 0; w < c->world; ++w) {   // every argument is checked BEFORE the RCCL group opens
    SMOE_REQUIRE(send_rows[w] >= 0 && recv_rows[w] >= 0, "smoe_a2a_tokens: negative row count");
    SMOE_REQUIRE(send || send_rows[w] == 0, "smoe_a2a_tokens: null send buffer");
    SMOE_REQUIRE(recv || recv_rows[w] == 0, "smoe_a2a_tokens: null receive buffer");
  }
  const bool in_line = wait == SMOE_A2A_INLINE;
  hipStream_t cs = in_line ? (hipStream_t)stream : c->comm_stream;
  if (!in_line)
    if (int rc = fence_in(c, (hipStream_t)stream)) return rc;
  SMOE_NCCL(api->GroupStart(), "ncclGroupStart");
  size_t so = 0, ro = 0;
  for (int w = 0; w < c->world; ++w) {
    const size_t sb = (size_t)send_rows[w] * d * es, rb = (size_t)recv_rows[w] * d * es;
    if (sb) SMOE_NCCL_IN_GROUP(api->Send((const char*)send + so, sb, ncclUint8, w, c->comm, cs), "ncclSend");
    if (rb) SMOE_NCCL_IN_GROUP(api->Recv((char*)recv + ro, rb, ncclUint8, w, c->comm, cs), "ncclRecv");
    so += sb;
    ro += rb;
  }
  SMOE_NCCL(api->GroupEnd(), "ncclGroupEnd");
  if (in_line) return 0;
  if (int rc = fence_out(c)) return rc;
  if (wait) return smoe_a2a_wait(c, stream);
  return 0;
}
