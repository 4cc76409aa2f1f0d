// Multi-GPU exchange of the cascade: one RCCL communicator per context, used for exactly one thing —
// the all-gather of per-window logits (N x 2 fp32, KB-scale, latency-bound) after each stage, plus a byte all-gather
// for the tiny host-side records of the batch driver (per-patient summaries).  The reference is single-process
// (src/run_batch_simple_2stage.py:258-292 loops over patients in one interpreter), so there is no call it replaces; the
// sharding itself lives in zkast/dist.py.
//
// RCCL is dlopen()ed on first use: libzkast.so has no link-time dependency on it and single-GPU use never loads it.
// It must be the RCCL that belongs to the HIP runtime this process already uses (libzkast.so itself binds to whichever
// libamdhip64 the host loaded, see zkast/lib.py): a PyTorch wheel ships its own libamdhip64.so + librccl.so, and an RCCL
// from another ROCm tree would pull a SECOND HIP / HSA runtime into the process.  Order: $ZKAST_RCCL_LIB when set — an
// explicit override is the ONLY candidate, a file that does not load is an error, never a silent fall-through to another
// RCCL — else the librccl that lies next to the runtime `hipStreamSynchronize` resolved to (dladdr), then the system names.
#include "../../include/zkast.h"
#include "zk_common.h"

#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>

// what comm.hip needs of the context (defined in zkast.hip)
hipStream_t zk_ctx_stream(zk_ctx* c);
int zk_ctx_device(zk_ctx* c);
int zk_ctx_fail(zk_ctx* c, int code, const char* msg);
void** zk_ctx_comm_slot(zk_ctx* c);      // opaque per-context pointer owned by this file
void* zk_ctx_stage_buf(zk_ctx* c, int which, size_t bytes);      // device staging (which = 0 send, 1 recv); NULL on failure
void* zk_ctx_prof_open_allgather(zk_ctx* c);      // zk_prof_* class "allgather": HIP events on the context's stream (NULL: not profiling)
void zk_ctx_prof_close(void* scope);

namespace {

typedef struct { char internal[128]; } rccl_uid_t;      // ncclUniqueId (NCCL_UNIQUE_ID_BYTES = 128)
typedef void* rccl_comm_t;
enum { RCCL_CHAR = 0, RCCL_FLOAT = 7 };                 // ncclInt8 / ncclFloat32

struct RcclApi {
  void* handle = nullptr;
  int (*GetUniqueId)(rccl_uid_t*) = nullptr;
  int (*CommInitRank)(rccl_comm_t*, int, rccl_uid_t, int) = nullptr;
  int (*CommDestroy)(rccl_comm_t) = nullptr;
  int (*AllGather)(const void*, void*, size_t, int, rccl_comm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  std::string err, path;
  bool load() {
    if (handle) return true;
    std::string tried;
    auto attempt = [&](const std::string& name) {
      if (handle || name.empty()) return;
      handle = dlopen(name.c_str(), RTLD_NOW | RTLD_LOCAL);
      if (handle) { path = name; return; }
      const char* why = dlerror();      // read ONCE: dlerror() clears the message it returns
      tried += (tried.empty() ? "" : "; ") + name + ": " + (why ? why : "?");
    };
    if (const char* e = getenv("ZKAST_RCCL_LIB")) {
      if (*e) {
        attempt(e);
        if (!handle) { err = "RCCL not loaded from $ZKAST_RCCL_LIB (" + tried + ")"; return false; }
      }
    }
    Dl_info info;
    memset(&info, 0, sizeof info);
    if (dladdr((const void*)&hipStreamSynchronize, &info) && info.dli_fname) {      // directory of the HIP runtime in use
      std::string dir = info.dli_fname;
      const size_t slash = dir.rfind('/');
      if (slash != std::string::npos) {
        dir.resize(slash + 1);
        attempt(dir + "librccl.so.1");
        attempt(dir + "librccl.so");
      }
    }
    attempt("librccl.so.1");
    attempt("librccl.so");
    if (!handle) { err = "RCCL not found (tried " + tried + ")"; return false; }
#define ZK_SYM(field, sym) field = (decltype(field))dlsym(handle, sym); if (!field) { err = std::string("RCCL lacks ") + sym; handle = nullptr; return false; }
    ZK_SYM(GetUniqueId, "ncclGetUniqueId")
    ZK_SYM(CommInitRank, "ncclCommInitRank")
    ZK_SYM(CommDestroy, "ncclCommDestroy")
    ZK_SYM(AllGather, "ncclAllGather")
    ZK_SYM(GetErrorString, "ncclGetErrorString")
#undef ZK_SYM
    return true;
  }
};
RcclApi g_rccl;

struct Comm {
  rccl_comm_t comm = nullptr;
  int rank = 0, world = 1;
};

int rfail(zk_ctx* c, const char* what, int rc) {
  char buf[256];
  snprintf(buf, sizeof buf, "%s failed: %s", what, g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "?");
  return zk_ctx_fail(c, ZK_E_HIP, buf);
}

bool on_device(const void* p) {
  hipPointerAttribute_t a;
  memset(&a, 0, sizeof a);
  if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; }
  return a.type == hipMemoryTypeDevice || a.type == hipMemoryTypeManaged;
}

// every rank contributes `bytes`; recv holds world*bytes.  Host pointers are staged through the context's buffers.
int allgather_bytes(zk_ctx* c, const void* send, size_t bytes, void* recv, bool timed) {
  Comm* cm = (Comm*)*zk_ctx_comm_slot(c);
  hipStream_t s = zk_ctx_stream(c);
  if (!cm || !cm->comm) {      // no communicator: a world of one
    if (bytes && send != recv) {
      if (hipMemcpyAsync(recv, send, bytes, hipMemcpyDefault, s) != hipSuccess) return zk_ctx_fail(c, ZK_E_HIP, "hipMemcpyAsync failed");
      if (hipStreamSynchronize(s) != hipSuccess) return zk_ctx_fail(c, ZK_E_HIP, "hipStreamSynchronize failed");
    }
    return ZK_OK;
  }
  if (bytes == 0) return ZK_OK;
  const bool sdev = on_device(send), rdev = on_device(recv);
  const void* dsend = send;
  void* drecv = recv;
  if (!sdev) {
    void* b = zk_ctx_stage_buf(c, 0, bytes);
    if (!b) return ZK_E_HIP;
    if (hipMemcpyAsync(b, send, bytes, hipMemcpyHostToDevice, s) != hipSuccess) return zk_ctx_fail(c, ZK_E_HIP, "H2D copy failed");
    dsend = b;
  }
  if (!rdev) {
    drecv = zk_ctx_stage_buf(c, 1, bytes * (size_t)cm->world);
    if (!drecv) return ZK_E_HIP;
  }
  // the collective alone (staging copies are outside the bracket); only the logit gathers count, not the byte gathers the
  // host code uses as barriers — those absorb the ranks' skew by design
  void* scope = timed ? zk_ctx_prof_open_allgather(c) : nullptr;
  const int rc = g_rccl.AllGather(dsend, drecv, bytes, RCCL_CHAR, cm->comm, s);
  zk_ctx_prof_close(scope);
  if (rc) return rfail(c, "ncclAllGather", rc);
  if (!rdev && hipMemcpyAsync(recv, drecv, bytes * (size_t)cm->world, hipMemcpyDeviceToHost, s) != hipSuccess)
    return zk_ctx_fail(c, ZK_E_HIP, "D2H copy failed");
  if (hipStreamSynchronize(s) != hipSuccess) return zk_ctx_fail(c, ZK_E_HIP, "hipStreamSynchronize failed");
  return ZK_OK;
}

}  // namespace

extern "C" {

int zk_comm_unique_id(void* out128) {
  if (!out128) return ZK_E_ARG;
  if (!g_rccl.load()) return zk_ctx_fail(nullptr, ZK_E_STATE, g_rccl.err.c_str());      // zk_last_error(NULL) holds the reason
  rccl_uid_t id;
  if (g_rccl.GetUniqueId(&id)) return ZK_E_HIP;
  memcpy(out128, &id, sizeof id);
  return ZK_OK;
}

int zk_comm_init(zk_ctx* c, int32_t rank, int32_t world, const void* unique_id) {
  if (!c) return ZK_E_ARG;
  if (world < 1 || rank < 0 || rank >= world) return zk_ctx_fail(c, ZK_E_ARG, "zk_comm_init: need 0 <= rank < world");
  if (*zk_ctx_comm_slot(c)) return zk_ctx_fail(c, ZK_E_STATE, "zk_comm_init: the context already has a communicator");
  Comm* cm = new Comm();
  cm->rank = rank; cm->world = world;
  if (world > 1 || unique_id) {      // (a world of one WITH an id still goes through RCCL: the single-GPU test of this path)
    if (!unique_id) { delete cm; return zk_ctx_fail(c, ZK_E_ARG, "zk_comm_init: unique_id is NULL"); }
    if (!g_rccl.load()) { delete cm; return zk_ctx_fail(c, ZK_E_STATE, g_rccl.err.c_str()); }
    if (hipSetDevice(zk_ctx_device(c)) != hipSuccess) { delete cm; return zk_ctx_fail(c, ZK_E_HIP, "hipSetDevice failed"); }
    rccl_uid_t id;
    memcpy(&id, unique_id, sizeof id);
    const int rc = g_rccl.CommInitRank(&cm->comm, world, id, rank);
    if (rc) { delete cm; return rfail(c, "ncclCommInitRank", rc); }
  }
  *zk_ctx_comm_slot(c) = cm;
  return ZK_OK;
}

int zk_comm_destroy(zk_ctx* c) {
  if (!c) return ZK_E_ARG;
  Comm* cm = (Comm*)*zk_ctx_comm_slot(c);
  if (!cm) return ZK_OK;
  if (cm->comm) {
    (void)hipStreamSynchronize(zk_ctx_stream(c));
    (void)g_rccl.CommDestroy(cm->comm);
  }
  delete cm;
  *zk_ctx_comm_slot(c) = nullptr;
  return ZK_OK;
}

int zk_comm_info(zk_ctx* c, int32_t* rank, int32_t* world) {
  if (!c) return ZK_E_ARG;
  Comm* cm = (Comm*)*zk_ctx_comm_slot(c);
  if (rank) *rank = cm ? cm->rank : 0;
  if (world) *world = cm ? cm->world : 1;
  return ZK_OK;
}

int zk_allgather_logits(zk_ctx* c, const float* local, int32_t rows_per_rank, int32_t cols, float* all) {
  if (!c || !local || !all) return ZK_E_ARG;
  if (rows_per_rank < 0 || cols < 1) return zk_ctx_fail(c, ZK_E_ARG, "zk_allgather_logits: bad shape");
  return allgather_bytes(c, local, (size_t)rows_per_rank * cols * sizeof(float), all, true);
}

int zk_comm_allgather_bytes(zk_ctx* c, const void* send, int64_t bytes, void* recv) {
  if (!c || bytes < 0 || (bytes && (!send || !recv))) return ZK_E_ARG;
  return allgather_bytes(c, send, (size_t)bytes, recv, false);
}

}  // extern "C"
