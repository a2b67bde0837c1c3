// Transport of the multi-GPU trainers' exchanges: RCCL (resolved at run time from librccl.so.1 -- the copy the process
// already holds when torch.distributed is in use, /opt/rocm/lib's otherwise) or a host callback (the CPU-box / two-ranks-
// on-one-GPU tests: gloo through Python).  One process per GPU; every operation is enqueued on the caller's stream.
#pragma once
#include <cstdint>

#include "common.h"

struct nncf_comm {
  int n_ranks = 1, rank = 0, device = 0;
  void* nccl = nullptr;          // ncclComm_t
  nncf_xfer_fn fn = nullptr;     // callback transport (nccl == nullptr)
  void* user = nullptr;
  bool self_via_rccl = false;    // NNCF_COMM_SELF_VIA_RCCL=1: this rank's own share travels through ncclSend / ncclRecv too
  int64_t wire_bytes = 0;        // payload bytes handed to the transport (own share included), for the record
};

namespace nncf {
// variable-split all-to-all: peer r receives send[send_off(r) .. +send_cnt[r]) and fills recv[recv_off(r) .. +recv_cnt[r]),
// offsets = prefix sums of the counts (elements of elem_bytes bytes), on `st`
int comm_all_to_all(nncf_comm* c, const void* d_send, const int64_t* send_cnt, void* d_recv, const int64_t* recv_cnt, int elem_bytes,
                    hipStream_t st);
}  // namespace nncf
