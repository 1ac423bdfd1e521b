// dr_comm.cpp -- the one data-path collective: all-gather of the residual vector after a
// light pass, through RCCL over xGMI.  The reference is single-GPU (no counterpart).
#include "dr_comm.h"

#include <dlfcn.h>
#include <cstring>
#include <mutex>

namespace dr {
namespace {

struct nccl_id { char internal[128]; };
typedef int (*fn_get_id)(nccl_id*);
typedef int (*fn_init_rank)(void**, int, nccl_id, int);
typedef int (*fn_allgather)(const void*, void*, size_t, int, void*, hipStream_t);
typedef int (*fn_destroy)(void*);
typedef const char* (*fn_errstr)(int);

struct Api {
    void* h = nullptr;
    fn_get_id get_id = nullptr;
    fn_init_rank init_rank = nullptr;
    fn_allgather allgather = nullptr;
    fn_destroy destroy = nullptr;
    fn_errstr errstr = nullptr;
    std::string err;
};

Api& api() {
    static Api a;
    static std::once_flag once;
    std::call_once(once, [] {
        // prefer a copy already mapped into the process (same SONAME as torch's bundled one)
        const char* names[] = { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" };
        for (const char* n : names) {
            a.h = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);
            if (a.h) break;
        }
        if (!a.h)
            for (const char* n : names) {
                a.h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
                if (a.h) break;
            }
        if (!a.h) { a.err = std::string("RCCL not found: ") + dlerror(); return; }
        a.get_id = (fn_get_id)dlsym(a.h, "ncclGetUniqueId");
        a.init_rank = (fn_init_rank)dlsym(a.h, "ncclCommInitRank");
        a.allgather = (fn_allgather)dlsym(a.h, "ncclAllGather");
        a.destroy = (fn_destroy)dlsym(a.h, "ncclCommDestroy");
        a.errstr = (fn_errstr)dlsym(a.h, "ncclGetErrorString");
        if (!a.get_id || !a.init_rank || !a.allgather || !a.destroy) a.err = "RCCL symbols missing";
    });
    return a;
}

std::string nccl_err(const char* what, int rc) {
    Api& a = api();
    return std::string(what) + ": " + (a.errstr ? a.errstr(rc) : "nccl error") + " (" + std::to_string(rc) + ")";
}

}  // namespace

std::string comm_unique_id(void* out128) {
    Api& a = api();
    if (!a.err.empty()) return a.err;
    nccl_id id;
    int rc = a.get_id(&id);
    if (rc) return nccl_err("ncclGetUniqueId", rc);
    memcpy(out128, &id, 128);
    return "";
}

std::string comm_init(Comm& c, const void* id128, int rank, int world) {
    Api& a = api();
    if (!a.err.empty()) return a.err;
    nccl_id id;
    memcpy(&id, id128, 128);
    int rc = a.init_rank(&c.comm, world, id, rank);
    if (rc) return nccl_err("ncclCommInitRank", rc);
    c.rank = rank;
    c.world = world;
    return "";
}

std::string comm_allgather_inplace(Comm& c, float* buf, size_t count, hipStream_t st) {
    Api& a = api();
    if (!c.comm) return "communicator not initialised";
    const int ncclFloat32 = 7;
    int rc = a.allgather(buf + (size_t)c.rank * count, buf, count, ncclFloat32, c.comm, st);
    if (rc) return nccl_err("ncclAllGather", rc);
    return "";
}

std::string comm_allgather_bytes_inplace(Comm& c, unsigned char* buf, size_t bytes_per_rank, hipStream_t st) {
    Api& a = api();
    if (!c.comm) return "communicator not initialised";
    const int ncclInt8 = 0;
    int rc = a.allgather(buf + (size_t)c.rank * bytes_per_rank, buf, bytes_per_rank, ncclInt8, c.comm, st);
    if (rc) return nccl_err("ncclAllGather", rc);
    return "";
}

void comm_destroy(Comm& c) {
    if (c.comm) { api().destroy(c.comm); c.comm = nullptr; }
}

}  // namespace dr
