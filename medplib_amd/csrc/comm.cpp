// RCCL helpers of the C ABI (SURVEY §8b Face 2): what `deepspeed.init_distributed` + ZeRO-2's bucketed gradient reduction
// (train_ds_medplib.py:412-419) and DeepSpeed MOELayer's `_AllToAll` (sharded_moe.py, call sites medplib_moe_llama.py:604-614) do
// over NCCL, as plain entry points over RCCL (xGMI inside a node).  One communicator per (process, GPU); the handle is the
// caller's — the library keeps no communicator state of its own.
//
// RCCL is bound at run time (dlopen, first from the images already mapped into the process, so a host that has torch loaded
// shares torch's librccl instead of starting a second copy): libmedplib_hip.so itself has no link-time dependency on RCCL and
// loads on machines without it — the comm entry points then return MP_ERR_ARG with an explanation.
#include "common.h"
#include <dlfcn.h>
#include <string.h>
#include <mutex>
#include <rccl/rccl.h>

namespace {

struct Rccl {
  void* h = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  bool ok = false;
};

// immutable after the once_flag fires: function pointers only
Rccl g_rccl;
std::once_flag g_rccl_once;

void bind_rccl() {
  const char* names[] = {"librccl.so.1", "librccl.so"};
  for (int pass = 0; pass < 2 && !g_rccl.h; ++pass)
    for (const char* n : names) {
      g_rccl.h = dlopen(n, RTLD_NOW | RTLD_LOCAL | (pass == 0 ? RTLD_NOLOAD : 0));
      if (g_rccl.h) break;
    }
  if (!g_rccl.h) return;
#define MP_SYM(F, NAME) g_rccl.F = reinterpret_cast<decltype(g_rccl.F)>(dlsym(g_rccl.h, NAME))
  MP_SYM(GetUniqueId, "ncclGetUniqueId"); MP_SYM(CommInitRank, "ncclCommInitRank"); MP_SYM(CommDestroy, "ncclCommDestroy");
  MP_SYM(AllReduce, "ncclAllReduce"); MP_SYM(Send, "ncclSend"); MP_SYM(Recv, "ncclRecv"); MP_SYM(GroupStart, "ncclGroupStart");
  MP_SYM(GroupEnd, "ncclGroupEnd"); MP_SYM(CommCount, "ncclCommCount"); MP_SYM(CommUserRank, "ncclCommUserRank");
  MP_SYM(GetErrorString, "ncclGetErrorString");
#undef MP_SYM
  g_rccl.ok = g_rccl.GetUniqueId && g_rccl.CommInitRank && g_rccl.CommDestroy && g_rccl.AllReduce && g_rccl.Send && g_rccl.Recv &&
              g_rccl.GroupStart && g_rccl.GroupEnd && g_rccl.CommCount && g_rccl.CommUserRank && g_rccl.GetErrorString;
}

const Rccl* rccl() {
  std::call_once(g_rccl_once, bind_rccl);
  return g_rccl.ok ? &g_rccl : nullptr;
}

#define MP_RCCL(R, CALL, WHAT)                                                              \
  do {                                                                                      \
    ncclResult_t r_ = (CALL);                                                               \
    if (r_ != ncclSuccess) {                                                                \
      mp_set_error("%s: %s", WHAT, (R)->GetErrorString(r_));                                \
      return MP_ERR_LAUNCH;                                                                 \
    }                                                                                       \
  } while (0)

int nccl_type(int dtype_tag, ncclDataType_t* t, size_t* size) {
  if (dtype_tag == MP_BF16) { *t = ncclBfloat16; *size = 2; return MP_OK; }
  if (dtype_tag == MP_F32) { *t = ncclFloat32; *size = 4; return MP_OK; }
  mp_set_error("comm: dtype tag %d (MP_BF16 or MP_F32)", dtype_tag);
  return MP_ERR_DTYPE;
}

}  // namespace

extern "C" int mp_comm_unique_id_bytes() { return (int)sizeof(ncclUniqueId); }

extern "C" int mp_comm_unique_id(void* out, int64_t bytes) {
  const Rccl* R = rccl();
  MP_REQUIRE(R != nullptr, MP_ERR_ARG, "mp_comm_unique_id: librccl.so could not be bound (dlopen)");
  MP_REQUIRE(out != nullptr && bytes >= (int64_t)sizeof(ncclUniqueId), MP_ERR_ARG, "mp_comm_unique_id: need %zu bytes", sizeof(ncclUniqueId));
  MP_RCCL(R, R->GetUniqueId(reinterpret_cast<ncclUniqueId*>(out)), "ncclGetUniqueId");
  return MP_OK;
}

extern "C" int mp_comm_init(int rank, int world, const void* unique_id, void** comm) {
  const Rccl* R = rccl();
  MP_REQUIRE(R != nullptr, MP_ERR_ARG, "mp_comm_init: librccl.so could not be bound (dlopen)");
  MP_REQUIRE(comm != nullptr && unique_id != nullptr && world >= 1 && rank >= 0 && rank < world, MP_ERR_ARG,
             "mp_comm_init: rank %d of %d", rank, world);
  ncclUniqueId id;
  memcpy(&id, unique_id, sizeof(id));
  ncclComm_t c = nullptr;
  MP_RCCL(R, R->CommInitRank(&c, world, id, rank), "ncclCommInitRank");
  *comm = c;
  return MP_OK;
}

extern "C" int mp_comm_destroy(void* comm) {
  const Rccl* R = rccl();
  MP_REQUIRE(R != nullptr && comm != nullptr, MP_ERR_ARG, "mp_comm_destroy: no communicator");
  MP_RCCL(R, R->CommDestroy(reinterpret_cast<ncclComm_t>(comm)), "ncclCommDestroy");
  return MP_OK;
}

extern "C" int mp_comm_count(void* comm, int* world, int* rank) {
  const Rccl* R = rccl();
  MP_REQUIRE(R != nullptr && comm != nullptr, MP_ERR_ARG, "mp_comm_count: no communicator");
  MP_REQUIRE(world != nullptr || rank != nullptr, MP_ERR_ARG, "mp_comm_count: nothing asked for");
  if (world != nullptr) MP_RCCL(R, R->CommCount(reinterpret_cast<ncclComm_t>(comm), world), "ncclCommCount");
  if (rank != nullptr) MP_RCCL(R, R->CommUserRank(reinterpret_cast<ncclComm_t>(comm), rank), "ncclCommUserRank");
  return MP_OK;
}

extern "C" int mp_allreduce_bucket(void* comm, void* buf, int64_t count, int dtype_tag, hipStream_t stream) {
  const Rccl* R = rccl();
  MP_REQUIRE(R != nullptr && comm != nullptr, MP_ERR_ARG, "mp_allreduce_bucket: no communicator");
  MP_REQUIRE(buf != nullptr && count >= 0, MP_ERR_ARG, "mp_allreduce_bucket: buffer / count");
  ncclDataType_t t; size_t sz;
  if (int rc = nccl_type(dtype_tag, &t, &sz)) return rc;
  if (count == 0) return MP_OK;
  MP_RCCL(R, R->AllReduce(buf, buf, (size_t)count, t, ncclSum, reinterpret_cast<ncclComm_t>(comm), stream), "ncclAllReduce");
  return MP_OK;
}

extern "C" int mp_alltoall_tokens(void* comm, const void* send, void* recv, int64_t count_per_peer, int dtype_tag, hipStream_t stream) {
  const Rccl* R = rccl();
  MP_REQUIRE(R != nullptr && comm != nullptr, MP_ERR_ARG, "mp_alltoall_tokens: no communicator");
  MP_REQUIRE(send != nullptr && recv != nullptr && send != recv && count_per_peer >= 0, MP_ERR_ARG, "mp_alltoall_tokens: buffers / count");
  ncclDataType_t t; size_t sz;
  if (int rc = nccl_type(dtype_tag, &t, &sz)) return rc;
  ncclComm_t c = reinterpret_cast<ncclComm_t>(comm);
  int world = 0;
  MP_RCCL(R, R->CommCount(c, &world), "ncclCommCount");
  if (count_per_peer == 0) return MP_OK;
  // chunk p of `send` goes to peer p, chunk p of `recv` comes from peer p: one grouped send/recv pair per peer (point-to-point
  // over xGMI: every pair has its own link, there is no switch to aggregate for)
  MP_RCCL(R, R->GroupStart(), "ncclGroupStart");
  for (int p = 0; p < world; ++p) {
    const char* s = reinterpret_cast<const char*>(send) + (size_t)p * count_per_peer * sz;
    char* d = reinterpret_cast<char*>(recv) + (size_t)p * count_per_peer * sz;
    ncclResult_t r1 = R->Send(s, (size_t)count_per_peer, t, p, c, stream);
    ncclResult_t r2 = r1 == ncclSuccess ? R->Recv(d, (size_t)count_per_peer, t, p, c, stream) : r1;
    if (r2 != ncclSuccess) {
      (void)R->GroupEnd();
      mp_set_error("mp_alltoall_tokens: peer %d: %s", p, R->GetErrorString(r2));
      return MP_ERR_LAUNCH;
    }
  }
  MP_RCCL(R, R->GroupEnd(), "ncclGroupEnd");
  return MP_OK;
}

// Variable all-to-all = grouped point-to-point messages (round 4: the expert exchange with routed rows only).  n_send messages
// (send_peer[i], send_ptr[i] = device address, send_count[i] elements) and n_recv messages likewise, all host arrays; messages between
// one pair of ranks match in array order.  xGMI is point-to-point, so this is what an all-to-all is on this fabric anyway.
extern "C" int mp_alltoallv_tokens(void* comm, int n_send, const int* send_peer, const int64_t* send_ptr, const int64_t* send_count, int n_recv,
                                   const int* recv_peer, const int64_t* recv_ptr, const int64_t* recv_count, int dtype_tag, hipStream_t stream) {
  const Rccl* R = rccl();
  MP_REQUIRE(R != nullptr && comm != nullptr, MP_ERR_ARG, "mp_alltoallv_tokens: no communicator");
  MP_REQUIRE(n_send >= 0 && n_recv >= 0 && (n_send == 0 || (send_peer && send_ptr && send_count)) && (n_recv == 0 || (recv_peer && recv_ptr && recv_count)),
             MP_ERR_ARG, "mp_alltoallv_tokens: message lists");
  ncclDataType_t t; size_t sz;
  if (int rc = nccl_type(dtype_tag, &t, &sz)) return rc;
  ncclComm_t c = reinterpret_cast<ncclComm_t>(comm);
  int world = 0;
  MP_RCCL(R, R->CommCount(c, &world), "ncclCommCount");
  for (int i = 0; i < n_send; ++i) MP_REQUIRE(send_peer[i] >= 0 && send_peer[i] < world && send_count[i] >= 0, MP_ERR_ARG, "mp_alltoallv_tokens: send message %d", i);
  for (int i = 0; i < n_recv; ++i) MP_REQUIRE(recv_peer[i] >= 0 && recv_peer[i] < world && recv_count[i] >= 0, MP_ERR_ARG, "mp_alltoallv_tokens: recv message %d", i);
  if (n_send + n_recv == 0) return MP_OK;
  MP_RCCL(R, R->GroupStart(), "ncclGroupStart");
  ncclResult_t r = ncclSuccess;
  for (int i = 0; i < n_recv && r == ncclSuccess; ++i)
    if (recv_count[i]) r = R->Recv(reinterpret_cast<void*>(recv_ptr[i]), (size_t)recv_count[i], t, recv_peer[i], c, stream);
  for (int i = 0; i < n_send && r == ncclSuccess; ++i)
    if (send_count[i]) r = R->Send(reinterpret_cast<const void*>(send_ptr[i]), (size_t)send_count[i], t, send_peer[i], c, stream);
  if (r != ncclSuccess) {
    (void)R->GroupEnd();
    mp_set_error("mp_alltoallv_tokens: %s", R->GetErrorString(r));
    return MP_ERR_LAUNCH;
  }
  MP_RCCL(R, R->GroupEnd(), "ncclGroupEnd");
  return MP_OK;
}
