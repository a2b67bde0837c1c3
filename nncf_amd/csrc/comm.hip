// RCCL over xGMI for the sharded trainers (no reference counterpart: /root/reference/main.py:34-35 selects
// ONE device).  The library does not link against RCCL: librccl.so.1 is opened at run time, so a process that already
// holds torch's copy shares it (one RCCL per process), and single-GPU users never load it.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstdlib>
#include <cstring>
#include <new>

#include "comm.h"

namespace {

struct Rccl {
  void* lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
Rccl g_rccl;

int load_rccl() {
  if (g_rccl.lib) return NNCF_OK;
  void* h = nullptr;
  for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
    h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
    if (h) break;
  }
  if (!h) return nncf::fail(NNCF_ERR_HIP, "librccl.so.1 not found (%s): the multi-GPU trainers need RCCL", dlerror());
  Rccl r;
  r.lib = h;
#define NNCF_SYM(field, name)                                                          \
  r.field = reinterpret_cast<decltype(r.field)>(dlsym(h, name));                       \
  if (!r.field) return nncf::fail(NNCF_ERR_HIP, "librccl.so.1 lacks the symbol %s", name)
  NNCF_SYM(GetUniqueId, "ncclGetUniqueId");
  NNCF_SYM(CommInitRank, "ncclCommInitRank");
  NNCF_SYM(CommDestroy, "ncclCommDestroy");
  NNCF_SYM(GroupStart, "ncclGroupStart");
  NNCF_SYM(GroupEnd, "ncclGroupEnd");
  NNCF_SYM(Send, "ncclSend");
  NNCF_SYM(Recv, "ncclRecv");
  NNCF_SYM(AllReduce, "ncclAllReduce");
  NNCF_SYM(GetErrorString, "ncclGetErrorString");
#undef NNCF_SYM
  g_rccl = r;
  return NNCF_OK;
}

#define NNCF_RCCL_CHECK(expr)                                                                                   \
  do {                                                                                                          \
    ncclResult_t _r = (expr);                                                                                   \
    if (_r != ncclSuccess) return nncf::fail(NNCF_ERR_HIP, "RCCL: %s failed: %s", #expr, g_rccl.GetErrorString(_r)); \
  } while (0)

}  // namespace

namespace nncf {

int comm_all_to_all(nncf_comm* c, const void* d_send, const int64_t* send_cnt, void* d_recv, const int64_t* recv_cnt, int elem_bytes,
                    hipStream_t st) {
  const int R = c->n_ranks;
  for (int r = 0; r < R; ++r) {
    NNCF_REQUIRE(send_cnt[r] >= 0 && recv_cnt[r] >= 0, "negative split");
    c->wire_bytes += send_cnt[r] * (int64_t)elem_bytes;
  }
  NNCF_REQUIRE(send_cnt[c->rank] == recv_cnt[c->rank], "a rank's own share must have the same size on both sides");
  if (c->fn) return c->fn(c->user, d_send, send_cnt, d_recv, recv_cnt, elem_bytes, R, st) ? fail(NNCF_ERR_HIP, "the transport callback failed") : NNCF_OK;
  NNCF_REQUIRE(c->nccl != nullptr, "communicator without a transport");
  const char* sp = static_cast<const char*>(d_send);
  char* rp = static_cast<char*>(d_recv);
  int64_t so = 0, ro = 0;
  NNCF_RCCL_CHECK(g_rccl.GroupStart());
  // Between GroupStart and GroupEnd nothing may return: an open group is thread-local state of RCCL, and every later
  // collective of this thread (torch's too) would queue behind it without ever launching.  The first failure is kept,
  // the group is always closed, then the failure is reported.
  ncclResult_t bad = ncclSuccess;
  hipError_t hbad = hipSuccess;
  const char* what = "";
  for (int r = 0; r < R && bad == ncclSuccess && hbad == hipSuccess; ++r) {
    const int64_t sb = send_cnt[r] * (int64_t)elem_bytes, rb = recv_cnt[r] * (int64_t)elem_bytes;
    if (r == c->rank && !c->self_via_rccl) {   // the own share never leaves the device
      if (sb > 0 && sp + so != rp + ro) {
        hbad = hipMemcpyAsync(rp + ro, sp + so, (size_t)sb, hipMemcpyDeviceToDevice, st);
        what = "hipMemcpyAsync (own share)";
      }
    } else {
      if (sb > 0) {
        bad = g_rccl.Send(sp + so, (size_t)sb, ncclChar, r, static_cast<ncclComm_t>(c->nccl), st);
        what = "ncclSend";
      }
      if (rb > 0 && bad == ncclSuccess) {
        bad = g_rccl.Recv(rp + ro, (size_t)rb, ncclChar, r, static_cast<ncclComm_t>(c->nccl), st);
        what = "ncclRecv";
      }
    }
    so += sb;
    ro += rb;
  }
  const ncclResult_t end = g_rccl.GroupEnd();
  if (hbad != hipSuccess) return fail(NNCF_ERR_HIP, "%s failed inside an RCCL group: %s", what, hipGetErrorString(hbad));
  if (bad != ncclSuccess) return fail(NNCF_ERR_HIP, "RCCL: %s failed: %s", what, g_rccl.GetErrorString(bad));
  if (end != ncclSuccess) return fail(NNCF_ERR_HIP, "RCCL: ncclGroupEnd failed: %s", g_rccl.GetErrorString(end));
  return NNCF_OK;
}

}  // namespace nncf

using namespace nncf;

extern "C" {

int nncf_comm_unique_id(void* out_id128) {
  NNCF_REQUIRE(out_id128 != nullptr, "out is NULL");
  if (int rc = load_rccl()) return rc;
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
  NNCF_RCCL_CHECK(g_rccl.GetUniqueId(static_cast<ncclUniqueId*>(out_id128)));
  return NNCF_OK;
}

int nncf_comm_create(const void* id128, int32_t n_ranks, int32_t rank, int device, nncf_comm** out) {
  NNCF_REQUIRE(out != nullptr, "out is NULL");
  *out = nullptr;
  NNCF_REQUIRE(id128 != nullptr && n_ranks >= 1 && rank >= 0 && rank < n_ranks, "bad rank %d of %d", rank, n_ranks);
  if (int rc = load_rccl()) return rc;
  DeviceGuard guard(device);
  NNCF_REQUIRE(guard.status == hipSuccess, "device %d cannot be selected", device);
  nncf_comm* c = new (std::nothrow) nncf_comm();
  if (!c) return fail(NNCF_ERR_NOMEM, "out of host memory");
  c->n_ranks = n_ranks; c->rank = rank; c->device = device;
  const char* e = getenv("NNCF_COMM_SELF_VIA_RCCL");
  c->self_via_rccl = e && e[0] && e[0] != '0';
  ncclUniqueId id;
  memcpy(&id, id128, sizeof(id));
  ncclComm_t comm = nullptr;
  const ncclResult_t r = g_rccl.CommInitRank(&comm, n_ranks, id, rank);
  if (r != ncclSuccess) {
    delete c;
    return fail(NNCF_ERR_HIP, "RCCL: ncclCommInitRank(rank %d of %d) failed: %s", rank, n_ranks, g_rccl.GetErrorString(r));
  }
  c->nccl = comm;
  *out = c;
  return NNCF_OK;
}

int nncf_comm_create_callback(nncf_xfer_fn fn, void* user, int32_t n_ranks, int32_t rank, nncf_comm** out) {
  NNCF_REQUIRE(out != nullptr, "out is NULL");
  *out = nullptr;
  NNCF_REQUIRE(fn != nullptr && n_ranks >= 1 && rank >= 0 && rank < n_ranks, "bad argument");
  nncf_comm* c = new (std::nothrow) nncf_comm();
  if (!c) return fail(NNCF_ERR_NOMEM, "out of host memory");
  c->n_ranks = n_ranks; c->rank = rank; c->fn = fn; c->user = user;
  *out = c;
  return NNCF_OK;
}

int nncf_comm_destroy(nncf_comm* c) {
  if (!c) return NNCF_OK;
  if (c->nccl && g_rccl.CommDestroy) {
    DeviceGuard guard(c->device);   // ncclCommDestroy frees device memory of the communicator's own device
    (void)g_rccl.CommDestroy(static_cast<ncclComm_t>(c->nccl));
  }
  delete c;
  return NNCF_OK;
}

int nncf_comm_all_to_all(nncf_comm* c, const void* d_send, const int64_t* send_cnt, void* d_recv, const int64_t* recv_cnt,
                         int32_t elem_bytes, void* stream) {
  NNCF_REQUIRE(c && d_send && d_recv && send_cnt && recv_cnt && elem_bytes > 0, "bad argument");
  return comm_all_to_all(c, d_send, send_cnt, d_recv, recv_cnt, elem_bytes, as_stream(stream));
}

int nncf_comm_all_reduce_f32(nncf_comm* c, float* d_buf, int64_t n, void* stream) {
  NNCF_REQUIRE(c && d_buf && n >= 0, "bad argument");
  c->wire_bytes += n * 4;
  if (c->fn) {   // callback transport: an all-reduce is "everybody sends everything to everybody" only in spirit -- signalled by elem_bytes = 0
    const int64_t cnt[1] = {n};
    return c->fn(c->user, d_buf, cnt, d_buf, cnt, 0, c->n_ranks, as_stream(stream)) ? fail(NNCF_ERR_HIP, "the transport callback failed") : NNCF_OK;
  }
  NNCF_REQUIRE(c->nccl != nullptr, "communicator without a transport");
  NNCF_RCCL_CHECK(g_rccl.AllReduce(d_buf, d_buf, (size_t)n, ncclFloat, ncclSum, static_cast<ncclComm_t>(c->nccl), as_stream(stream)));
  return NNCF_OK;
}

int nncf_comm_wire_bytes(nncf_comm* c, int64_t* out, int32_t reset) {
  NNCF_REQUIRE(c && out, "NULL argument");
  *out = c->wire_bytes;
  if (reset) c->wire_bytes = 0;
  return NNCF_OK;
}

}  // extern "C"
