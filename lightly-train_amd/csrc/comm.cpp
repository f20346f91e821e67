// The one piece of process-wide state of the library: an RCCL communicator handle, created once per rank, with a side
// stream of its own.  Collectives are enqueued on that stream behind an event of the caller's stream; lt_comm_wait makes a stream wait for
// everything enqueued so far.  RCCL is resolved with dlopen at lt_comm_init (no link-time dependency: a single-GPU user never loads it; a
// process that has torch loaded gets the RCCL torch already mapped, one copy per process).
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstring>
#include <mutex>

#include "lt_common.h"

namespace {
std::mutex mu;
void* lib = nullptr;
ncclComm_t comm = nullptr;
hipStream_t cstream = nullptr;
// one fence event per collective in flight: an event re-recorded while an earlier wait on it has not been consumed by the device yet
// would move that wait to the later record (HIP events are not counting semaphores), so a step that starts its 14 gradient all-reduces
// back to back takes 14 different events; 32 cover two steps' worth.  Re-use is guarded: every slot also carries a completion event recorded
// on the communicator's stream behind its collective, and a slot is only taken again after that event has fired (the host blocks in
// hipEventSynchronize when more than 32 collectives are in flight: ViT-g's 40 blocks x several buckets -- slower, never wrong).
constexpr int N_EV = 32;
hipEvent_t ev_in[N_EV] = {nullptr};
hipEvent_t ev_done[N_EV] = {nullptr};
bool ev_used[N_EV] = {false};
hipEvent_t ev_out = nullptr;
unsigned ev_next = 0;
int world = 0;

ncclResult_t (*p_get_id)(ncclUniqueId*) = nullptr;
ncclResult_t (*p_init)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
ncclResult_t (*p_allreduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
ncclResult_t (*p_destroy)(ncclComm_t) = nullptr;
const char* (*p_errstr)(ncclResult_t) = nullptr;

void release_locked() {   // everything lt_comm_init created, in any partial state
  if (cstream) (void)hipStreamSynchronize(cstream);
  if (comm && p_destroy) p_destroy(comm);
  if (cstream) (void)hipStreamDestroy(cstream);
  for (int i = 0; i < N_EV; ++i) {
    if (ev_in[i]) (void)hipEventDestroy(ev_in[i]);
    if (ev_done[i]) (void)hipEventDestroy(ev_done[i]);
    ev_in[i] = ev_done[i] = nullptr;
    ev_used[i] = false;
  }
  if (ev_out) (void)hipEventDestroy(ev_out);
  comm = nullptr; cstream = nullptr; ev_out = nullptr; world = 0; ev_next = 0;
}

bool resolve() {
  if (lib) return true;
  for (const char* name : {"librccl.so.1", "librccl.so"}) {
    lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
    if (lib) break;
  }
  if (!lib) { lt_set_error("lt_comm: cannot load librccl.so (%s)", dlerror()); return false; }
  p_get_id = (decltype(p_get_id))dlsym(lib, "ncclGetUniqueId");
  p_init = (decltype(p_init))dlsym(lib, "ncclCommInitRank");
  p_allreduce = (decltype(p_allreduce))dlsym(lib, "ncclAllReduce");
  p_destroy = (decltype(p_destroy))dlsym(lib, "ncclCommDestroy");
  p_errstr = (decltype(p_errstr))dlsym(lib, "ncclGetErrorString");
  if (!p_get_id || !p_init || !p_allreduce || !p_destroy) { lt_set_error("lt_comm: librccl.so lacks the expected entry points"); lib = nullptr; return false; }
  return true;
}
int fail(const char* what, ncclResult_t r) {
  lt_set_error("%s: RCCL error %d (%s)", what, (int)r, p_errstr ? p_errstr(r) : "?");
  return LT_ERR_HIP;
}
}  // namespace

extern "C" int lt_comm_unique_id(void* id_out, int bytes) {
  LT_CHECK_ARG(id_out && bytes >= (int)sizeof(ncclUniqueId), "lt_comm_unique_id: the id needs %d bytes", (int)sizeof(ncclUniqueId));
  std::lock_guard<std::mutex> l(mu);
  if (!resolve()) return LT_ERR_HIP;
  ncclUniqueId id;
  const ncclResult_t r = p_get_id(&id);
  if (r != ncclSuccess) return fail("lt_comm_unique_id", r);
  memcpy(id_out, &id, sizeof(id));
  return LT_OK;
}

extern "C" int lt_comm_init(int rank, int world_size, const void* id_in, int bytes) {
  LT_CHECK_ARG(id_in && bytes >= (int)sizeof(ncclUniqueId) && world_size >= 1 && rank >= 0 && rank < world_size, "lt_comm_init: bad arguments");
  std::lock_guard<std::mutex> l(mu);
  if (comm) { lt_set_error("lt_comm_init: this process already holds a communicator (lt_comm_destroy first)"); return LT_ERR_INVALID; }
  if (!resolve()) return LT_ERR_HIP;
  ncclUniqueId id;
  memcpy(&id, id_in, sizeof(id));
  const ncclResult_t r = p_init(&comm, world_size, id, rank);
  if (r != ncclSuccess) { comm = nullptr; return fail("lt_comm_init", r); }
  bool ok = hipStreamCreateWithFlags(&cstream, hipStreamNonBlocking) == hipSuccess && hipEventCreateWithFlags(&ev_out, hipEventDisableTiming) == hipSuccess;
  for (int i = 0; ok && i < N_EV; ++i)
    ok = hipEventCreateWithFlags(&ev_in[i], hipEventDisableTiming) == hipSuccess && hipEventCreateWithFlags(&ev_done[i], hipEventDisableTiming) == hipSuccess;
  if (!ok) {   // leave no half-built handle behind: a retry must not find "already holds a communicator" nor null streams / events
    release_locked();
    lt_set_error("lt_comm_init: stream / event creation failed");
    return LT_ERR_HIP;
  }
  world = world_size;
  return LT_OK;
}

// buf[0:n] <- sum over ranks, in place, on the communicator's stream, after everything `after_stream` holds at this moment
extern "C" int lt_comm_allreduce_f32(float* buf, int64_t n, void* after_stream) {
  LT_CHECK_ARG(buf && n >= 0, "lt_comm_allreduce_f32: bad arguments");
  std::lock_guard<std::mutex> l(mu);
  if (!comm) { lt_set_error("lt_comm_allreduce_f32: no communicator (lt_comm_init)"); return LT_ERR_INVALID; }
  if (n == 0) return LT_OK;
  const unsigned slot = ev_next++ % N_EV;
  if (ev_used[slot] && hipEventSynchronize(ev_done[slot]) != hipSuccess) {   // the slot's previous collective (and with it its fence wait) is over
    lt_set_error("lt_comm_allreduce_f32: waiting for fence slot %u failed", slot);
    return LT_ERR_HIP;
  }
  hipEvent_t ev = ev_in[slot];
  if (hipEventRecord(ev, (hipStream_t)after_stream) != hipSuccess || hipStreamWaitEvent(cstream, ev, 0) != hipSuccess) {
    lt_set_error("lt_comm_allreduce_f32: event fence failed");
    return LT_ERR_HIP;
  }
  const ncclResult_t r = p_allreduce(buf, buf, (size_t)n, ncclFloat, ncclSum, comm, cstream);
  if (r != ncclSuccess) return fail("lt_comm_allreduce_f32", r);
  if (hipEventRecord(ev_done[slot], cstream) != hipSuccess) {
    lt_set_error("lt_comm_allreduce_f32: completion event failed");
    return LT_ERR_HIP;
  }
  ev_used[slot] = true;
  return LT_OK;
}

// `stream` waits for every collective enqueued so far
extern "C" int lt_comm_wait(void* stream) {
  std::lock_guard<std::mutex> l(mu);
  if (!comm) return LT_OK;
  if (hipEventRecord(ev_out, cstream) != hipSuccess || hipStreamWaitEvent((hipStream_t)stream, ev_out, 0) != hipSuccess) {
    lt_set_error("lt_comm_wait: event fence failed");
    return LT_ERR_HIP;
  }
  return LT_OK;
}

extern "C" int lt_comm_size(void) {
  std::lock_guard<std::mutex> l(mu);
  return comm ? world : 0;
}

extern "C" int lt_comm_destroy(void) {
  std::lock_guard<std::mutex> l(mu);
  if (!comm) return LT_OK;
  release_locked();
  return LT_OK;
}
