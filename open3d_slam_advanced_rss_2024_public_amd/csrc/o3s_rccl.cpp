// o3s_rccl.cpp — libo3dslam_icp_rccl.so: ncclAllReduce-backed exchange for the one-pair-sharded ICP mode
// (include/o3s_rccl.h).  Host code only; RCCL brings its own kernels.
#include "../../include/o3s_rccl.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstring>
#include <string>

static_assert(O3S_RCCL_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "unique id size");

struct o3s_rccl {
  ncclComm_t comm = nullptr;
  int device = 0;
  int rank = 0, world = 1;
  int64_t collectives = 0;
};

namespace {
thread_local std::string g_err;
int fail(const std::string& what) {
  g_err = what;
  return 1;
}
}  // namespace

extern "C" {

const char* o3s_rccl_last_error(void) { return g_err.c_str(); }

int o3s_rccl_unique_id(uint8_t id[O3S_RCCL_ID_BYTES]) {
  if (!id) return fail("NULL id");
  ncclUniqueId u;
  const ncclResult_t r = ncclGetUniqueId(&u);
  if (r != ncclSuccess) return fail(std::string("ncclGetUniqueId: ") + ncclGetErrorString(r));
  std::memcpy(id, u.internal, O3S_RCCL_ID_BYTES);
  return 0;
}

int o3s_rccl_create(const uint8_t id[O3S_RCCL_ID_BYTES], int32_t rank, int32_t world, int device, o3s_rccl** out) {
  if (!id || !out || world < 1 || rank < 0 || rank >= world) return fail("bad argument");
  *out = nullptr;
  if (hipSetDevice(device) != hipSuccess) return fail("hipSetDevice failed");
  ncclUniqueId u;
  std::memcpy(u.internal, id, O3S_RCCL_ID_BYTES);
  o3s_rccl* c = new o3s_rccl();
  c->device = device;
  c->rank = rank;
  c->world = world;
  const ncclResult_t r = ncclCommInitRank(&c->comm, world, u, rank);
  if (r != ncclSuccess) {
    delete c;
    return fail(std::string("ncclCommInitRank: ") + ncclGetErrorString(r));
  }
  *out = c;
  return 0;
}

void o3s_rccl_destroy(o3s_rccl* c) {
  if (!c) return;
  if (c->comm) (void)ncclCommDestroy(c->comm);
  delete c;
}

int o3s_rccl_allreduce(void* user, void* dev_ptr, int64_t /*byte_offset*/, int64_t count, int32_t dtype, void* hip_stream) {
  o3s_rccl* c = static_cast<o3s_rccl*>(user);
  if (!c || !c->comm || !dev_ptr || count <= 0 || (dtype != 0 && dtype != 1)) return fail("bad argument");
  const ncclResult_t r = ncclAllReduce(dev_ptr, dev_ptr, (size_t)count, dtype == 0 ? ncclInt32 : ncclFloat64, ncclSum, c->comm,
                                       reinterpret_cast<hipStream_t>(hip_stream));
  if (r != ncclSuccess) return fail(std::string("ncclAllReduce: ") + ncclGetErrorString(r));
  c->collectives += 1;
  return 0;
}

int64_t o3s_rccl_collectives(const o3s_rccl* c) { return c ? c->collectives : 0; }

}  // extern "C"
