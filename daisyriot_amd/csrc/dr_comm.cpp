// dr_comm.cpp -- the data-path collectives: all-gather of the residual vector after a light pass and the
// all-to-all of ray-count slots of a multi-rank assembly, through RCCL over xGMI.  The reference is
// single-GPU (no counterpart).
#include "dr_comm.h"

#include <dlfcn.h>
#include <link.h>
#include <rccl/rccl.h>
#include <cstdlib>
#include <cstring>
#include <mutex>

namespace dr {
namespace {

struct Api {
    void* h = nullptr;
    decltype(&ncclGetUniqueId) get_id = nullptr;
    decltype(&ncclCommInitRank) init_rank = nullptr;
    decltype(&ncclCommInitAll) init_all = nullptr;
    decltype(&ncclAllGather) allgather = nullptr;
    decltype(&ncclSend) send = nullptr;
    decltype(&ncclRecv) recv = nullptr;
    decltype(&ncclGroupStart) group_start = nullptr;
    decltype(&ncclGroupEnd) group_end = nullptr;
    decltype(&ncclCommCount) count = nullptr;
    decltype(&ncclCommUserRank) user_rank = nullptr;
    decltype(&ncclCommDestroy) destroy = nullptr;
    decltype(&ncclGetErrorString) errstr = nullptr;
    std::string err;
};

template <class F>
void bind(Api& a, F& f, const char* name) {
    f = (F)dlsym(a.h, name);
    if (!f && a.err.empty()) a.err = std::string("RCCL symbol missing: ") + name;
}

// the library to bind, set before the first use (comm_set_library); empty: DR_RCCL_LIB, else a copy already in the process,
// else the system's
std::string g_forced_lib;
std::mutex g_lib_mu;
bool g_bound = false;

// Everything is opened RTLD_LOCAL.  A host process may map a SECOND RCCL later (importing torch loads the copy bundled with
// it: its NEEDED entry is the file name `librccl.so`, which the loader does not match with an already loaded SONAME
// `librccl.so.1`); with RTLD_GLOBAL the first copy's symbols would interpose the second's and the two runtimes tear each other
// down at exit ("double free or corruption", round 2's gpurun_out/r2_crash.log).  Opened locally the copies stay apart.
Api& api() {
    static Api a;
    static std::once_flag once;
    std::call_once(once, [] {
        { std::lock_guard<std::mutex> lk(g_lib_mu); g_bound = true; }
        // an explicit choice (dr_comm_set_library), or DR_RCCL_LIB (tests: a loop-back stand-in that lets several ranks share
        // one GPU)
        const char* forced = !g_forced_lib.empty() ? g_forced_lib.c_str() : getenv("DR_RCCL_LIB");
        if (forced && *forced) {
            a.h = dlopen(forced, RTLD_NOW | RTLD_LOCAL);
            if (!a.h) { a.err = std::string("RCCL library ") + forced + ": " + dlerror(); return; }
        }
        // prefer a copy already mapped into the process (same SONAME as torch's bundled one)
        const char* names[] = { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" };
        if (!a.h) for (const char* n : names) {
            a.h = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_LOCAL);
            if (a.h) break;
        }
        if (!a.h)
            for (const char* n : names) {
                a.h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
                if (a.h) break;
            }
        if (!a.h) { a.err = std::string("RCCL not found: ") + dlerror(); return; }
        bind(a, a.get_id, "ncclGetUniqueId");
        bind(a, a.init_rank, "ncclCommInitRank");
        bind(a, a.init_all, "ncclCommInitAll");
        bind(a, a.allgather, "ncclAllGather");
        bind(a, a.send, "ncclSend");
        bind(a, a.recv, "ncclRecv");
        bind(a, a.group_start, "ncclGroupStart");
        bind(a, a.group_end, "ncclGroupEnd");
        bind(a, a.count, "ncclCommCount");
        bind(a, a.user_rank, "ncclCommUserRank");
        bind(a, a.destroy, "ncclCommDestroy");
        a.errstr = (decltype(a.errstr))dlsym(a.h, "ncclGetErrorString");
    });
    return a;
}

// every object mapped into the process whose file name says RCCL
int collect_rccl(struct dl_phdr_info* info, size_t, void* out) {
    if (info->dlpi_name && std::strstr(info->dlpi_name, "rccl")) static_cast<std::vector<std::string>*>(out)->push_back(info->dlpi_name);
    return 0;
}

std::string nccl_err(const char* what, ncclResult_t rc) {
    Api& a = api();
    return std::string(what) + ": " + (a.errstr ? a.errstr(rc) : "nccl error") + " (" + std::to_string((int)rc) + ")";
}

// what the communicator itself says it is: a bench line can then prove RCCL saw N ranks
std::string read_back(Comm& c) {
    Api& a = api();
    int n = 0, r = -1;
    ncclResult_t rc = a.count((ncclComm_t)c.comm, &n);
    if (rc != ncclSuccess) return nccl_err("ncclCommCount", rc);
    rc = a.user_rank((ncclComm_t)c.comm, &r);
    if (rc != ncclSuccess) return nccl_err("ncclCommUserRank", rc);
    c.world = n;
    c.rank = r;
    return "";
}

}  // namespace

std::string comm_set_library(const char* path) {
    std::lock_guard<std::mutex> lk(g_lib_mu);
    if (g_bound) return "the RCCL library is already bound (dr_comm_set_library must come before the first dr_comm_* / dr_group_create call)";
    g_forced_lib = path ? path : "";
    return "";
}

std::string comm_library_info() {
    Api& a = api();
    std::string out = "bound=";
    Dl_info di;
    if (a.get_id && dladdr(reinterpret_cast<void*>(a.get_id), &di) && di.dli_fname) out += di.dli_fname;
    else out += a.err.empty() ? "?" : "(none: " + a.err + ")";
    std::vector<std::string> all;
    dl_iterate_phdr(collect_rccl, &all);
    out += ";mapped=";
    for (size_t i = 0; i < all.size(); i++) out += (i ? "," : "") + all[i];
    return out;
}

std::string comm_unique_id(void* out128) {
    Api& a = api();
    if (!a.err.empty()) return a.err;
    static_assert(sizeof(ncclUniqueId) == 128, "the ABI hands the unique id over as 128 bytes");
    ncclUniqueId id;
    ncclResult_t rc = a.get_id(&id);
    if (rc != ncclSuccess) return nccl_err("ncclGetUniqueId", rc);
    memcpy(out128, &id, 128);
    return "";
}

std::string comm_init(Comm& c, const void* id128, int rank, int world) {
    Api& a = api();
    if (!a.err.empty()) return a.err;
    ncclUniqueId id;
    memcpy(&id, id128, 128);
    ncclComm_t comm = nullptr;
    ncclResult_t rc = a.init_rank(&comm, world, id, rank);
    if (rc != ncclSuccess) return nccl_err("ncclCommInitRank", rc);
    c.comm = comm;
    std::string e = read_back(c);
    if (e.empty() && (c.rank != rank || c.world != world))
        e = "RCCL communicator reports rank " + std::to_string(c.rank) + " of " + std::to_string(c.world) + ", asked for " +
            std::to_string(rank) + " of " + std::to_string(world);
    if (!e.empty()) comm_destroy(c);
    return e;
}

std::string comm_init_all(std::vector<Comm>& cs, const int* devices, int n) {
    Api& a = api();
    if (!a.err.empty()) return a.err;
    std::vector<ncclComm_t> comms((size_t)n, nullptr);
    ncclResult_t rc = a.init_all(comms.data(), n, devices);
    if (rc != ncclSuccess) return nccl_err("ncclCommInitAll", rc);
    cs.assign((size_t)n, Comm());
    std::string e;
    for (int r = 0; r < n; r++) {
        cs[r].comm = comms[r];
        if (e.empty()) e = read_back(cs[r]);
        if (e.empty() && (cs[r].rank != r || cs[r].world != n)) e = "ncclCommInitAll: communicator " + std::to_string(r) + " reports another rank or size";
    }
    if (!e.empty()) for (auto& c : cs) comm_destroy(c);
    return e;
}

std::string comm_allgather_inplace(Comm& c, float* buf, size_t count, hipStream_t st) {
    Api& a = api();
    if (!c.comm) return "communicator not initialised";
    ncclResult_t rc = a.allgather(buf + (size_t)c.rank * count, buf, count, ncclFloat32, (ncclComm_t)c.comm, st);
    if (rc != ncclSuccess) return nccl_err("ncclAllGather", rc);
    return "";
}

std::string comm_allgather_i32_inplace(Comm& c, int* buf, hipStream_t st) {
    Api& a = api();
    if (!c.comm) return "communicator not initialised";
    ncclResult_t rc = a.allgather(buf + c.rank, buf, 1, ncclInt32, (ncclComm_t)c.comm, st);
    if (rc != ncclSuccess) return nccl_err("ncclAllGather", rc);
    return "";
}

std::string comm_allgather_inplace_group(std::vector<Comm*>& cs, std::vector<float*>& bufs, size_t count, std::vector<hipStream_t>& sts) {
    Api& a = api();
    ncclResult_t rc = a.group_start();
    if (rc != ncclSuccess) return nccl_err("ncclGroupStart", rc);
    std::string e;
    for (size_t r = 0; r < cs.size() && e.empty(); r++) {
        if (!cs[r]->comm) { e = "communicator not initialised"; break; }
        rc = a.allgather(bufs[r] + (size_t)cs[r]->rank * count, bufs[r], count, ncclFloat32, (ncclComm_t)cs[r]->comm, sts[r]);
        if (rc != ncclSuccess) e = nccl_err("ncclAllGather", rc);
    }
    rc = a.group_end();
    if (rc != ncclSuccess && e.empty()) e = nccl_err("ncclGroupEnd", rc);
    return e;
}

std::string comm_alltoall_bytes(Comm& c, const unsigned char* send, unsigned char* recv, size_t block_bytes, hipStream_t st) {
    Api& a = api();
    if (!c.comm) return "communicator not initialised";
    ncclResult_t rc = a.group_start();
    if (rc != ncclSuccess) return nccl_err("ncclGroupStart", rc);
    std::string e;
    for (int p = 0; p < c.world && e.empty(); p++) {
        if (p == c.rank) continue;
        rc = a.send(send + (size_t)p * block_bytes, block_bytes, ncclUint8, p, (ncclComm_t)c.comm, st);
        if (rc != ncclSuccess) { e = nccl_err("ncclSend", rc); break; }
        rc = a.recv(recv + (size_t)p * block_bytes, block_bytes, ncclUint8, p, (ncclComm_t)c.comm, st);
        if (rc != ncclSuccess) e = nccl_err("ncclRecv", rc);
    }
    rc = a.group_end();
    if (rc != ncclSuccess && e.empty()) e = nccl_err("ncclGroupEnd", rc);
    return e;
}

std::string comm_group_start() {
    Api& a = api();
    if (!a.err.empty()) return a.err;
    ncclResult_t rc = a.group_start();
    return rc == ncclSuccess ? "" : nccl_err("ncclGroupStart", rc);
}
std::string comm_group_end() {
    Api& a = api();
    if (!a.err.empty()) return a.err;
    ncclResult_t rc = a.group_end();
    return rc == ncclSuccess ? "" : nccl_err("ncclGroupEnd", rc);
}

void comm_destroy(Comm& c) {
    if (c.comm) { api().destroy((ncclComm_t)c.comm); c.comm = nullptr; }
}

}  // namespace dr
