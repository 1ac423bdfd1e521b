// dr_comm.h -- RCCL bound at run time (dlopen, RTLD_LOCAL), so the library has no link-time RCCL
// dependency and shares the copy a host process (e.g. torch) has already loaded.  Types, enum values and
// prototypes come from <rccl/rccl.h>; only the symbol lookup is deferred.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <string>
#include <vector>

namespace dr {

struct Comm {
    void* comm = nullptr;     // ncclComm_t
    int rank = 0, world = 1;  // what the communicator itself reports (ncclCommUserRank / ncclCommCount), read back at init
};

// which RCCL to bind: a path, before the first use (empty / null: the default search); error text if already bound
std::string comm_set_library(const char* path);
// "bound=<file the bound symbols live in>;mapped=<every RCCL file mapped into the process, comma separated>"
std::string comm_library_info();
// fills out[128]; returns empty string on success, else the error text
std::string comm_unique_id(void* out128);
std::string comm_init(Comm& c, const void* id128, int rank, int world);
// one process, several devices: communicators of all of them at once (ncclCommInitAll); devices must be distinct
std::string comm_init_all(std::vector<Comm>& cs, const int* devices, int n);
// in-place all-gather of `count` floats per rank inside buf (rank r's chunk at r*count)
std::string comm_allgather_inplace(Comm& c, float* buf, size_t count, hipStream_t st);
// the same for several communicators of this process in one group call (single-process multi-GPU)
std::string comm_allgather_inplace_group(std::vector<Comm*>& cs, std::vector<float*>& bufs, size_t count,
                                         std::vector<hipStream_t>& sts);
// one int per rank, in place (rank r's value at buf[r]): the ranks' go / no-go before a collective
std::string comm_allgather_i32_inplace(Comm& c, int* buf, hipStream_t st);
// all-to-all of equal blocks: block p of `send` goes to rank p, block p of `recv` comes from rank p (the own block
// is not moved); grouped ncclSend/ncclRecv
std::string comm_alltoall_bytes(Comm& c, const unsigned char* send, unsigned char* recv, size_t block_bytes, hipStream_t st);
std::string comm_group_start();
std::string comm_group_end();
void comm_destroy(Comm& c);

}  // namespace dr
