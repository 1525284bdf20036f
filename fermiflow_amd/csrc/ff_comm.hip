// ff_comm.hip -- ff_comm_*: the estimator's collectives for callers that drive the C ABI WITHOUT torch (SURVEY 8(b)'s minimum symbol
// set; 8(e): per sweep one all-reduce of the four estimator sums and one of the 3(He+Hm)-double parameter gradient, both far below
// the size at which xGMI bandwidth matters -- a latency-bound ring / tree exchange each).  The Python package does not use these:
// torch.distributed's "nccl" backend IS RCCL and owns the communicator there (fermiflow_amd/dist.py).
//
// RCCL is bound at run time -- dlopen of the copy the process already holds (torch's) or librccl.so.1 -- so that loading
// libfermiflow_hip.so never pulls a second RCCL into a process, and a single-GPU user never loads one at all.
#include "ff_common.h"
#include <dlfcn.h>
#include <stdlib.h>
#include <string.h>

extern void ff_set_error(const char* msg);

namespace {

typedef struct { char internal[128]; } nccl_uid;      // ncclUniqueId (NCCL_UNIQUE_ID_BYTES = 128, rccl.h)
typedef void* nccl_comm;
enum { NCCL_SUM = 0, NCCL_FLOAT64 = 8 };               // ncclSum, ncclFloat64 (rccl.h)

struct rccl_api {
  void* handle;
  int (*GetUniqueId)(nccl_uid*);
  int (*CommInitRank)(nccl_comm*, int, nccl_uid, int);
  int (*AllReduce)(const void*, void*, size_t, int, int, nccl_comm, hipStream_t);
  int (*CommDestroy)(nccl_comm);
  const char* (*GetErrorString)(int);
};

rccl_api* rccl() {
  static rccl_api api = {};
  static int state = 0;      // 0: not tried, 1: bound, -1: unavailable
  if (state == 0) {
    const char* names[] = {getenv("FF_RCCL_LIB"), "librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1"};
    void* h = nullptr;
    for (const char* nm : names) {      // first a copy that is already loaded (RTLD_NOLOAD), then a fresh one
      if (nm && *nm && (h = dlopen(nm, RTLD_NOW | RTLD_NOLOAD))) break;
    }
    for (int k = 0; !h && k < 4; k++)
      if (names[k] && *names[k]) h = dlopen(names[k], RTLD_NOW | RTLD_GLOBAL);
    if (h) {
      api.handle = h;
      api.GetUniqueId = (int (*)(nccl_uid*))dlsym(h, "ncclGetUniqueId");
      api.CommInitRank = (int (*)(nccl_comm*, int, nccl_uid, int))dlsym(h, "ncclCommInitRank");
      api.AllReduce = (int (*)(const void*, void*, size_t, int, int, nccl_comm, hipStream_t))dlsym(h, "ncclAllReduce");
      api.CommDestroy = (int (*)(nccl_comm))dlsym(h, "ncclCommDestroy");
      api.GetErrorString = (const char* (*)(int))dlsym(h, "ncclGetErrorString");
    }
    state = (h && api.GetUniqueId && api.CommInitRank && api.AllReduce && api.CommDestroy) ? 1 : -1;
  }
  return state == 1 ? &api : nullptr;
}

int fail(rccl_api* r, const char* what, int code) {
  char msg[200];
  snprintf(msg, sizeof msg, "%s: %s", what, (r && r->GetErrorString) ? r->GetErrorString(code) : "RCCL error");
  ff_set_error(msg);
  return FF_ELAUNCH;
}

}  // namespace

struct ff_comm { nccl_comm comm; int world, rank; };

extern "C" {

int ff_comm_unique_id(void* id128) {
  if (!id128) { ff_set_error("ff_comm_unique_id: null pointer"); return FF_EINVAL; }
  rccl_api* r = rccl();
  if (!r) { ff_set_error("ff_comm_unique_id: RCCL (librccl.so) could not be loaded"); return FF_EUNSUPPORTED; }
  nccl_uid id;
  const int st = r->GetUniqueId(&id);
  if (st != 0) return fail(r, "ncclGetUniqueId", st);
  memcpy(id128, id.internal, sizeof id.internal);
  return FF_OK;
}

int ff_comm_init(ff_comm** comm, int world_size, int rank, const void* id128) {
  if (!comm || world_size <= 0 || rank < 0 || rank >= world_size || !id128) { ff_set_error("ff_comm_init: bad argument"); return FF_EINVAL; }
  rccl_api* r = rccl();
  if (!r) { ff_set_error("ff_comm_init: RCCL (librccl.so) could not be loaded"); return FF_EUNSUPPORTED; }
  nccl_uid id;
  memcpy(id.internal, id128, sizeof id.internal);
  nccl_comm c = nullptr;
  const int st = r->CommInitRank(&c, world_size, id, rank);      // on the calling thread's current device (hipSetDevice)
  if (st != 0) return fail(r, "ncclCommInitRank", st);
  ff_comm* out = (ff_comm*)malloc(sizeof(ff_comm));
  if (!out) { r->CommDestroy(c); ff_set_error("ff_comm_init: out of memory"); return FF_ELAUNCH; }
  out->comm = c; out->world = world_size; out->rank = rank;
  *comm = out;
  return FF_OK;
}

int ff_comm_allreduce(ff_comm* comm, void* stream, double* buf, int64_t count) {
  if (!comm || count < 0 || (count > 0 && !buf)) { ff_set_error("ff_comm_allreduce: bad argument"); return FF_EINVAL; }
  if (count == 0) return FF_OK;
  rccl_api* r = rccl();
  const int st = r->AllReduce(buf, buf, (size_t)count, NCCL_FLOAT64, NCCL_SUM, comm->comm, (hipStream_t)stream);
  return st == 0 ? FF_OK : fail(r, "ncclAllReduce", st);
}

int ff_comm_destroy(ff_comm* comm) {
  if (!comm) return FF_OK;
  rccl_api* r = rccl();
  const int st = r ? r->CommDestroy(comm->comm) : 0;
  free(comm);
  return st == 0 ? FF_OK : fail(r, "ncclCommDestroy", st);
}

}  // extern "C"
