// fake_rccl.cpp -- TEST INFRASTRUCTURE: a loop-back stand-in for the handful of RCCL entry points libdaisyriot_hip.so
// binds (dr_comm.cpp), so that the multi-PROCESS code paths -- communicator set-up, the all-gather of the residual after
// every pass, the go / no-go agreement and the all-to-all of ray-count slots in a multi-rank assembly -- can run with
// world = 2, 3 on ONE GPU (real RCCL refuses two ranks on one device).  Messages go through a POSIX shared-memory
// segment named after the unique id; every call is synchronous (the stream is drained first), which preserves the
// stream order the library relies on.  Selected with DR_RCCL_LIB=<this .so> (tests/test_gpu_fake_rccl.py).  Never part
// of the product, never timed.
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <unistd.h>

#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <vector>

extern "C" {

typedef struct { char internal[128]; } ncclUniqueId;
typedef enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclSystemError = 2, ncclInternalError = 3, ncclInvalidArgument = 4 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclUint8 = 1, ncclInt32 = 2, ncclUint32 = 3, ncclInt64 = 4, ncclUint64 = 5, ncclFloat16 = 6, ncclFloat32 = 7, ncclFloat64 = 8 } ncclDataType_t;

struct Shared {
    std::atomic<int> arrived;          // barrier: arrivals of the current generation
    std::atomic<int> generation;
    std::atomic<int> attached;
    char pad[52];
    // then: world * world mailboxes of `slot` bytes (mailbox[src][dst]); all-gather uses mailbox[src][0]
};

struct Comm {
    int rank, world;
    Shared* sh;
    size_t slot, total;
    char name[160];
    int fd;
};
typedef Comm* ncclComm_t;

struct Pending { int kind; const void* s; void* r; size_t bytes; int peer; Comm* c; hipStream_t st; };
static thread_local int g_depth = 0;
static thread_local std::vector<Pending> g_pending;

static size_t dsize(ncclDataType_t t) {
    switch (t) { case ncclInt8: case ncclUint8: return 1; case ncclFloat16: return 2; case ncclInt32: case ncclUint32: case ncclFloat32: return 4; default: return 8; }
}

static void barrier(Comm* c) {
    const int gen = c->sh->generation.load();
    if (c->sh->arrived.fetch_add(1) + 1 == c->world) {
        c->sh->arrived.store(0);
        c->sh->generation.fetch_add(1);
    } else {
        const time_t t0 = time(nullptr);
        while (c->sh->generation.load() == gen) {
            usleep(50);
            if (time(nullptr) - t0 > 120) { fprintf(stderr, "fake_rccl: rank %d waited 120 s at a barrier -- a peer is missing\n", c->rank); abort(); }
        }
    }
}

static char* mailbox(Comm* c, int src, int dst) { return (char*)c->sh + sizeof(Shared) + ((size_t)src * c->world + dst) * c->slot; }

const char* ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "no error" : "fake_rccl error"; }

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
    memset(id, 0, sizeof *id);
    snprintf(id->internal, sizeof id->internal, "/fake_rccl_%d_%ld", (int)getpid(), (long)time(nullptr));
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* out, int world, ncclUniqueId id, int rank) {
    Comm* c = new Comm();
    c->rank = rank; c->world = world;
    const char* e = getenv("FAKE_RCCL_SLOT_MB");
    c->slot = (size_t)(e ? atoi(e) : 8) << 20;
    c->total = sizeof(Shared) + (size_t)world * world * c->slot;
    snprintf(c->name, sizeof c->name, "%s", id.internal);
    c->fd = shm_open(c->name, O_CREAT | O_RDWR, 0600);
    if (c->fd < 0 || ftruncate(c->fd, (off_t)c->total) != 0) { perror("fake_rccl shm"); return ncclSystemError; }
    void* p = mmap(nullptr, c->total, PROT_READ | PROT_WRITE, MAP_SHARED, c->fd, 0);
    if (p == MAP_FAILED) { perror("fake_rccl mmap"); return ncclSystemError; }
    c->sh = (Shared*)p;                      // a fresh segment is zero filled: the atomics start at 0
    c->sh->attached.fetch_add(1);
    const time_t t0 = time(nullptr);
    while (c->sh->attached.load() < world) {
        usleep(100);
        if (time(nullptr) - t0 > 120) { fprintf(stderr, "fake_rccl: rank %d: peers did not attach\n", rank); return ncclSystemError; }
    }
    barrier(c);
    *out = c;
    return ncclSuccess;
}

ncclResult_t ncclCommInitAll(ncclComm_t*, int, const int*) { return ncclInvalidArgument; }   // one process per rank only
ncclResult_t ncclCommCount(const ncclComm_t c, int* n) { *n = c->world; return ncclSuccess; }
ncclResult_t ncclCommUserRank(const ncclComm_t c, int* r) { *r = c->rank; return ncclSuccess; }

ncclResult_t ncclCommDestroy(ncclComm_t c) {
    barrier(c);
    munmap(c->sh, c->total);
    close(c->fd);
    if (c->rank == 0) shm_unlink(c->name);
    delete c;
    return ncclSuccess;
}

ncclResult_t ncclAllGather(const void* send, void* recv, size_t count, ncclDataType_t t, ncclComm_t c, hipStream_t st) {
    const size_t bytes = count * dsize(t);
    if (bytes > c->slot) { fprintf(stderr, "fake_rccl: all-gather chunk of %zu bytes exceeds the %zu-byte mailbox (FAKE_RCCL_SLOT_MB)\n", bytes, c->slot); return ncclInvalidArgument; }
    if (hipStreamSynchronize(st) != hipSuccess) return ncclUnhandledCudaError;
    if (hipMemcpy(mailbox(c, c->rank, 0), send, bytes, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
    barrier(c);
    for (int r = 0; r < c->world; r++)
        if (hipMemcpy((char*)recv + (size_t)r * bytes, mailbox(c, r, 0), bytes, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
    barrier(c);
    return ncclSuccess;
}

static ncclResult_t flush() {
    if (g_pending.empty()) return ncclSuccess;
    Comm* c = g_pending[0].c;
    for (auto& p : g_pending) if (hipStreamSynchronize(p.st) != hipSuccess) return ncclUnhandledCudaError;
    for (auto& p : g_pending)
        if (p.kind == 0) {
            if (p.bytes > c->slot) { fprintf(stderr, "fake_rccl: send of %zu bytes exceeds the %zu-byte mailbox (FAKE_RCCL_SLOT_MB)\n", p.bytes, c->slot); return ncclInvalidArgument; }
            if (hipMemcpy(mailbox(c, c->rank, p.peer), p.s, p.bytes, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
        }
    barrier(c);
    for (auto& p : g_pending)
        if (p.kind == 1 && hipMemcpy(p.r, mailbox(c, p.peer, c->rank), p.bytes, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
    barrier(c);
    g_pending.clear();
    return ncclSuccess;
}

ncclResult_t ncclGroupStart() { g_depth++; return ncclSuccess; }
ncclResult_t ncclGroupEnd() { if (--g_depth == 0) return flush(); return ncclSuccess; }
ncclResult_t ncclSend(const void* s, size_t count, ncclDataType_t t, int peer, ncclComm_t c, hipStream_t st) {
    g_pending.push_back(Pending{ 0, s, nullptr, count * dsize(t), peer, c, st });
    return g_depth == 0 ? flush() : ncclSuccess;
}
ncclResult_t ncclRecv(void* r, size_t count, ncclDataType_t t, int peer, ncclComm_t c, hipStream_t st) {
    g_pending.push_back(Pending{ 1, nullptr, r, count * dsize(t), peer, c, st });
    return g_depth == 0 ? flush() : ncclSuccess;
}

}  // extern "C"
