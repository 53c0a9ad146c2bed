// RCCL-backed communicator (see rccl_comm.cpp).
#pragma once
#include "../backend.hpp"
#include "device.hpp"

namespace lga {
// Writes a fresh ncclUniqueId (128 bytes) to `out`.
void rccl_unique_id(void* out, int cap, int* len);
// Collective: every rank calls it with the id created on rank 0.
Comm* make_rccl_comm(int rank, int size, const void* unique_id, int id_len);
// In-process loop-back group for device buffers (single-GPU verification of the distributed learners).
void make_loopback_group(int size, Comm** out);
}  // namespace lga
