// dr_comm.h -- RCCL bound at run time (dlopen), so the library has no link-time RCCL
// dependency and shares the copy a host process (e.g. torch) has already loaded.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <string>

namespace dr {

struct Comm {
    void* comm = nullptr;     // ncclComm_t
    int rank = 0, world = 1;
};

// fills out[128]; returns empty string on success, else the error text
std::string comm_unique_id(void* out128);
std::string comm_init(Comm& c, const void* id128, int rank, int world);
// in-place all-gather of `count` floats per rank inside buf (rank r's chunk at r*count)
std::string comm_allgather_inplace(Comm& c, float* buf, size_t count, hipStream_t st);
// the same for bytes (the ray-count slots of a multi-rank assembly)
std::string comm_allgather_bytes_inplace(Comm& c, unsigned char* buf, size_t bytes_per_rank, hipStream_t st);
void comm_destroy(Comm& c);

}  // namespace dr
