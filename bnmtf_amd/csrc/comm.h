// RCCL exchange for the row/column-sharded sweep (one process per GPU), plus an in-process transport for tests (a
// communicator id starting with "BNMTFLOC": ranks are host threads of one process, see comm.hip).
// librccl is resolved at run time (dlopen) so that the single-GPU path has no
// link-time dependency on it and a process that already loaded RCCL (e.g. through
// torch.distributed in bench.py) shares that copy.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace bnmtf {
struct Comm;
int comm_unique_id(uint8_t out[128]);
int comm_create(Comm** out, const uint8_t id[128], int rank, int world, hipStream_t st);
void comm_destroy(Comm* c);
int comm_info(const Comm* c, int* kind, int* ranks);    // kind: 0 no communicator, 1 RCCL, 2 in-process transport; ranks as the communicator reports them
// gather every rank's freshly drawn block of X ([nglob][KP], rank r owns rows
// [nglob*r/world, nglob*(r+1)/world)) in place (the caller re-lays it out afterwards).
int comm_allgather_factor(Comm* c, float* X, int KP, int nglob, int world, hipStream_t st);
int comm_allreduce_sum(Comm* c, double* buf, int count, hipStream_t st);
int comm_allreduce_max(Comm* c, double* buf, int count, hipStream_t st);
int comm_allreduce_sum_f32(Comm* c, float* buf, int count, hipStream_t st);
}  // namespace bnmtf
