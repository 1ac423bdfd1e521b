// dr_api.cpp -- implementation of include/daisyriot_hip.h (the drop-in boundary).
#include "../../include/daisyriot_hip.h"
#include "dr_comm.h"
#include "dr_internal.h"

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

using namespace dr;

namespace {
thread_local std::string g_err;

int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}
// the built-in defaults, overridden by the environment -- the ONLY place the library reads it (DR_RCCL_LIB apart: the RCCL
// binding is per process, dr_comm.cpp)
void options_from_env(dr_options* o) {
    std::memset(o, 0, sizeof *o);
    o->size = (int32_t)sizeof *o;
    auto geti = [](const char* name, int dflt) { const char* e = getenv(name); return e && *e ? atoi(e) : dflt; };
    auto is = [](const char* name, const char* v) { const char* e = getenv(name); return e && !strcmp(e, v); };
    o->tree = is("DR_BVH", "lbvh") ? DR_TREE_LBVH : (is("DR_BVH", "sah") ? DR_TREE_SAH : DR_TREE_AUTO);
    o->sah_on_host = geti("DR_SAH_HOST", 0) != 0;
    o->morton_key = geti("DR_BVH_KEY", 0);
    o->sah_bins = geti("DR_SAH_BINS", 32);
    { const char* e = getenv("DR_SAH_DILATE"); o->sah_dilate = e && *e ? (float)atof(e) : 0.5f; }
    o->sah_host_threads = geti("DR_SAH_THREADS", 0);
    o->walk = is("DR_WALK", "threaded") ? DR_WALK_THREADED : (is("DR_WALK", "pairs") ? DR_WALK_PAIRS : (is("DR_WALK", "paths") ? DR_WALK_PATHS : DR_WALK_AUTO));
    o->octant_test = geti("DR_OCTANT", 1) != 0;
    o->vis_exchange = getenv("DR_NO_VIS_EXCHANGE") ? 1 : (getenv("DR_VIS_EXCHANGE_REHEARSE") ? 2 : 0);
    o->tile_stats = getenv("DR_TILE_STATS") ? 1 : 0;
    o->debug_pair_lo = o->debug_pair_hi = -1; o->debug_ray = 0;
    if (const char* dp = getenv("DR_DEBUG_PAIR")) sscanf(dp, "%d,%d,%d", &o->debug_pair_lo, &o->debug_pair_hi, &o->debug_ray);
    o->sweep_ksplit = std::max(geti("DR_SWEEP_KSPLIT", 0), 0);
    o->sweep_taper = geti("DR_SWEEP_TAPER", -1);
    o->sweep_rows_per_wave = geti("DR_SWEEP_RR", 0);
    o->sweep_skew = geti("DR_SWEEP_SKEW", -1);
    o->sweep_mfma = geti("DR_SWEEP_MFMA", 1) != 0;
    o->sweep_fenced = geti("DR_SWEEP_FENCED", 0) != 0;
    o->no_comm = getenv("DR_NO_COMM") ? 1 : 0;
    o->debug_converge = getenv("DR_DEBUG_CONV") ? 1 : 0;
    o->group_exchange = is("DR_GROUP_EXCHANGE", "p2p") ? DR_GROUP_EXCHANGE_P2P : (is("DR_GROUP_EXCHANGE", "rccl") ? DR_GROUP_EXCHANGE_RCCL
                        : (is("DR_GROUP_EXCHANGE", "inpass") ? DR_GROUP_EXCHANGE_INPASS : DR_GROUP_EXCHANGE_AUTO));
    o->fault_assemble_rank = geti("DR_FAULT_ASSEMBLE_RANK", -1);
}

int check_options(const dr_options* o) {
    if (!o) return fail(DR_ERR_INVALID, "options is null");
    if (o->size != (int32_t)sizeof(dr_options)) return fail(DR_ERR_INVALID, "dr_options of %d bytes, the library's is %zu: start from dr_options_defaults", o->size, sizeof(dr_options));
    if (o->tree < DR_TREE_AUTO || o->tree > DR_TREE_SAH) return fail(DR_ERR_INVALID, "options: tree %d", o->tree);
    if (o->walk < DR_WALK_AUTO || o->walk > DR_WALK_PATHS) return fail(DR_ERR_INVALID, "options: walk %d", o->walk);
    if (o->morton_key < 0 || o->morton_key > 2) return fail(DR_ERR_INVALID, "options: morton_key %d", o->morton_key);
    if (o->sah_bins < 2 || o->sah_bins > 128) return fail(DR_ERR_INVALID, "options: sah_bins %d outside 2..128", o->sah_bins);
    if (!(o->sah_dilate >= 0.0f) || !(o->sah_dilate < 1e6f)) return fail(DR_ERR_INVALID, "options: sah_dilate %g", (double)o->sah_dilate);
    if (o->sweep_ksplit < 0 || o->sweep_ksplit > 64) return fail(DR_ERR_INVALID, "options: sweep_ksplit %d outside 0..64", o->sweep_ksplit);
    if (o->sweep_rows_per_wave != 0 && o->sweep_rows_per_wave != 4 && o->sweep_rows_per_wave != 8) return fail(DR_ERR_INVALID, "options: sweep_rows_per_wave %d (0, 4 or 8)", o->sweep_rows_per_wave);
    if (o->group_exchange < DR_GROUP_EXCHANGE_AUTO || o->group_exchange > DR_GROUP_EXCHANGE_INPASS) return fail(DR_ERR_INVALID, "options: group_exchange %d", o->group_exchange);
    if (o->vis_exchange < 0 || o->vis_exchange > 2) return fail(DR_ERR_INVALID, "options: vis_exchange %d", o->vis_exchange);
    return DR_OK;
}

SweepTuning sweep_tuning(const dr_options& o) {
    SweepTuning t;
    t.rows_per_wave = o.sweep_rows_per_wave; t.mfma = o.sweep_mfma; t.ksplit = o.sweep_ksplit; t.skew = o.sweep_skew; t.taper = o.sweep_taper;
    t.fenced = o.sweep_fenced;
    return t;
}
}  // namespace

#define HIPCHK(x)                                                                                   \
    do {                                                                                            \
        hipError_t e_ = (x);                                                                        \
        if (e_ != hipSuccess)                                                                       \
            return fail(e_ == hipErrorOutOfMemory ? DR_ERR_NOMEM : DR_ERR_DEVICE, "%s failed: %s", #x, \
                        hipGetErrorString(e_));                                                     \
    } while (0)
// a start/stop event pair that is destroyed on every exit path
struct EventPair {
    hipEvent_t a = nullptr, b = nullptr;
    hipError_t create() { hipError_t e = hipEventCreate(&a); return e != hipSuccess ? e : hipEventCreate(&b); }
    float ms() const { float t = 0; return hipEventElapsedTime(&t, a, b) == hipSuccess ? t : 0.0f; }
    ~EventPair() { if (a) hipEventDestroy(a); if (b) hipEventDestroy(b); }
};

#define CTX(c)                                                         \
    do {                                                               \
        if (!(c)) return fail(DR_ERR_INVALID, "null context");         \
        HIPCHK(hipSetDevice((c)->device));                             \
    } while (0)

struct dr_context {
    int device = 0;
    hipStream_t own_stream = nullptr, stream = nullptr;
    int rank = 0, world = 1;
    // scene
    int N = 0, V = 0, Nn = 0;
    float *d_vtx = nullptr, *d_nrm = nullptr;
    int *d_tv = nullptr, *d_tn = nullptr;
    PatchRec* d_patch = nullptr;
    TriRec *d_tri = nullptr, *d_tri_sorted = nullptr;
    BvhNode* d_bvh = nullptr;
    BvhNode* d_bvh_lh = nullptr;      // the nodes as lower / upper corners (sign-specialised node test)
    BvhPair *d_pairs = nullptr, *d_pairs_lh = nullptr;   // the tree in sibling-pair form (BvhPair), both box forms
    int tree_depth = 0;               // depth of the written tree (the pair walk's stack holds one item per level)
    BvhNode* d_path_rec = nullptr;    // [N][PATH_RECS] path records of the patches (PathHdr, dr_internal.h)
    PathHdr* d_path_hdr = nullptr;
    int n_nodes = 0;
    float scene_span = 0;     // max |coordinate| + diagonal of the scene
    // shard
    int rpr = 0, row0 = 0, nrows = 0;
    size_t ldF = 0;
    float* d_F = nullptr;
    size_t F_floats = 0;
    bool have_F = false;
    unsigned char* d_vis = nullptr;
    float* d_uv = nullptr;
    int uv_K = 0;                     // samples in d_uv
    // ray-count exchange of a multi-rank assembly (TileParams::vsend / vrecv): per destination / source rank one block of
    // tiles_per_rank^2 slots of 64 x 64 bytes
    unsigned char *d_vsend = nullptr, *d_vrecv = nullptr;
    size_t vx_block = 0;              // bytes of one block
    int* d_agree = nullptr;           // world ints: the ranks' go / no-go before a collective (comm_agree)
    EventPair* asm_ev = nullptr;      // around the tile kernel of an assembly in flight
    unsigned long long h_cnt[16] = { 0 };
    bool split_pending = false;       // dr_formfactors_assemble_split done, ..._finish still to come
    int split_K = 0, split_rule = 0;
    float split_eps = 0;
    unsigned long long* d_counter = nullptr;
    // solver
    int S = 0, n_mat = 0;
    float *d_M = nullptr, *d_E = nullptr, *d_B = nullptr, *d_R[2] = { nullptr, nullptr };
    int* d_mat = nullptr;
    float* d_Gpart = nullptr;
    size_t cstride = 0;               // floats per rank chunk of the residual buffers: S*rpr + RTAIL (the chunk's per-bin sums)
    unsigned* d_tickets = nullptr;    // in-launch reductions of the pass (SweepParams::tickets)
    double* d_blk_sums = nullptr;
    int* d_ctl = nullptr;             // [0] passes done, [1] a pass found the residual converged (SweepParams::ctl), [3] a gate timed out
    // in-pass exchange of a group (dr_options::group_exchange = INPASS): the ranks' published pass numbers as seen by this
    // device, the number of passes launched since the last reset, and the group's contexts (set by the group, else empty)
    unsigned* d_seq = nullptr;
    unsigned pass_seq = 0;
    std::vector<dr_context*> inpass_peers;
    dr_options opt;                   // every choice between equivalent ways (include/daisyriot_hip.h): defaults + environment at creation
    int tree_used = 0, tree_on_host = 0, walk_used = 0;      // what the last set_mesh / assembly really did (dr_get_info)
    int check_every = 8;              // dr_solver_converge looks at d_ctl once per this many queued passes
    bool tails_valid = false;         // the current residual's chunk tails hold its per-bin sums
    // optional zero-block skipping of the light pass (dr_solver_skip_zero_blocks)
    bool skip_zero = false, mask_valid = false;
    unsigned* d_mask = nullptr;
    int mask_words = 0;
    unsigned long long blocks_nonzero = 0, blocks_total = 0;
    float* d_stage = nullptr;     // N x S staging for layout conversion on read-back
    float* d_rgb = nullptr;       // display colours of the local rows (nrows x 3), valid after dr_display_patch_colors
    // buffers of dr_display_vertex_colors, kept between calls (a viewer asks after every pass)
    int *d_voff = nullptr, *d_vadj = nullptr;
    float *d_vout = nullptr, *d_vin = nullptr;
    size_t voff_n = 0, vadj_n = 0, vout_n = 0, vin_n = 0;
    bool have_rgb = false;
    int ksplit = 1;
    int cur = 0;
    bool have_solver = false;
    Comm comm;
    bool manual_exchange = false;
    // measurement
    double last_assemble_ms = 0, last_bvh_ms = 0;
    SahTopology* shared_sah = nullptr;   // set by a group around dr_group_set_mesh: one host build for all its devices
    unsigned long long pairs_traced = 0, stat_visits = 0, stat_leaves = 0;
    bool profile = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_pool;
    size_t ev_used = 0;
    unsigned long long sweep_launches = 0;
    double sweep_ms_total = 0;
};

namespace {

void free_F(dr_context* c) {
    hipFree(c->d_F); c->d_F = nullptr; c->F_floats = 0; c->have_F = false;
    hipFree(c->d_vis); c->d_vis = nullptr;
    hipFree(c->d_vsend); hipFree(c->d_vrecv); c->d_vsend = c->d_vrecv = nullptr; c->vx_block = 0; c->split_pending = false;
    hipFree(c->d_mask); c->d_mask = nullptr; c->mask_valid = false; c->mask_words = 0;
}
void free_solver(dr_context* c) {
    hipFree(c->d_M); hipFree(c->d_E); hipFree(c->d_B); hipFree(c->d_R[0]); hipFree(c->d_R[1]);
    hipFree(c->d_mat); hipFree(c->d_Gpart); c->d_Gpart = nullptr; hipFree(c->d_stage); c->d_stage = nullptr;
    hipFree(c->d_tickets); hipFree(c->d_blk_sums); hipFree(c->d_ctl); hipFree(c->d_seq);
    c->d_tickets = nullptr; c->d_blk_sums = nullptr; c->d_ctl = nullptr; c->d_seq = nullptr; c->pass_seq = 0;
    hipFree(c->d_rgb); c->d_rgb = nullptr; c->have_rgb = false;
    hipFree(c->d_voff); hipFree(c->d_vadj); hipFree(c->d_vout); hipFree(c->d_vin);
    c->d_voff = c->d_vadj = nullptr; c->d_vout = c->d_vin = nullptr; c->voff_n = c->vadj_n = c->vout_n = c->vin_n = 0;
    c->d_M = c->d_E = c->d_B = c->d_R[0] = c->d_R[1] = nullptr; c->d_mat = nullptr;
    c->have_solver = false;
}
void free_scene(dr_context* c) {
    hipFree(c->d_vtx); hipFree(c->d_nrm); hipFree(c->d_tv); hipFree(c->d_tn);
    hipFree(c->d_patch); hipFree(c->d_tri); hipFree(c->d_tri_sorted); hipFree(c->d_bvh);
    hipFree(c->d_path_rec); hipFree(c->d_path_hdr); c->d_path_rec = nullptr; c->d_path_hdr = nullptr;
    hipFree(c->d_bvh_lh); c->d_bvh_lh = nullptr;
    hipFree(c->d_pairs); hipFree(c->d_pairs_lh); c->d_pairs = c->d_pairs_lh = nullptr;
    c->d_vtx = c->d_nrm = nullptr; c->d_tv = c->d_tn = nullptr;
    c->d_patch = nullptr; c->d_tri = nullptr; c->d_tri_sorted = nullptr; c->d_bvh = nullptr; c->N = 0;
}

void shard_rows(int N, int rank, int world, int* row0, int* nrows, int* rpr) {
    int per = (N + world - 1) / world;
    *rpr = ((per + SHARD_ALIGN - 1) / SHARD_ALIGN) * SHARD_ALIGN;
    *row0 = rank * *rpr;
    *nrows = std::max(0, std::min(N - *row0, *rpr));
}

void recompute_shard(dr_context* c) {
    if (c->N <= 0) return;
    shard_rows(c->N, c->rank, c->world, &c->row0, &c->nrows, &c->rpr);
    c->ldF = (size_t)c->world * c->rpr;
}

int ensure_F(dr_context* c) {
    size_t need = (size_t)std::max(c->nrows, 1) * c->ldF;
    if (c->d_F && c->F_floats == need) return DR_OK;
    free_F(c);
    hipError_t e = hipMalloc(&c->d_F, need * sizeof(float));
    if (e != hipSuccess) {
        c->d_F = nullptr;
        return fail(DR_ERR_NOMEM, "F shard of %d x %zu floats (%.2f GB) does not fit: %s", c->nrows, c->ldF,
                    need * 4.0 / 1e9, hipGetErrorString(e));
    }
    c->F_floats = need;
    return DR_OK;
}

// drain the profiled event pairs into the running totals
void drain_events(dr_context* c) {
    for (size_t i = 0; i < c->ev_used; i++) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, c->ev_pool[i].first, c->ev_pool[i].second) == hipSuccess) {
            c->sweep_ms_total += ms;
            c->sweep_launches++;
        }
    }
    c->ev_used = 0;
}

// one bit per 32-row x 256-column block of the F shard (SweepParams::tile_mask); counts them for dr_get_info
int build_tile_mask(dr_context* c) {
    const int row_blocks = (std::max(c->nrows, 1) + 31) / 32;
    const int words = (int)((c->ldF / 256 + 31) / 32);
    if (!c->d_mask || c->mask_words != words) {
        hipFree(c->d_mask); c->d_mask = nullptr;
        HIPCHK(hipMalloc(&c->d_mask, sizeof(unsigned) * (size_t)row_blocks * words));
        c->mask_words = words;
    }
    HIPCHK(hipMemsetAsync(c->d_mask, 0, sizeof(unsigned) * (size_t)row_blocks * words, c->stream));
    HIPCHK(launch_tile_mask(c->stream, c->d_F, c->nrows, c->ldF, c->d_mask, words));
    std::vector<unsigned> h((size_t)row_blocks * words);
    HIPCHK(hipMemcpyAsync(h.data(), c->d_mask, sizeof(unsigned) * h.size(), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    unsigned long long nz = 0;
    for (unsigned w : h) nz += (unsigned long long)__builtin_popcount(w);
    c->blocks_nonzero = nz;
    c->blocks_total = (unsigned long long)((c->nrows + 31) / 32) * (c->ldF / 256);
    c->mask_valid = true;
    return DR_OK;
}

// conv_mode / thr: SweepParams::conv_mode (0 = an unconditional pass)
int sweep_once(dr_context* c, int conv_mode = 0, float thr = 0.0f) {
    HIPCHK(hipSetDevice(c->device));
    if (c->comm.comm && (c->comm.rank != c->rank || c->comm.world != c->world))
        return fail(DR_ERR_STATE, "RCCL communicator is rank %d of %d, the shard is %d of %d", c->comm.rank, c->comm.world, c->rank, c->world);
    SweepParams p;
    p.N = c->N; p.S = c->S; p.rpr = c->rpr; p.world = c->world; p.row0 = c->row0; p.nrows = c->nrows;
    p.ldF = c->ldF; p.F = c->d_F; p.Rin = c->d_R[c->cur]; p.Rout = c->d_R[c->cur ^ 1]; p.rank = c->rank;
    p.cstride = c->cstride;
    p.B = c->d_B; p.M = c->d_M; p.mat = c->d_mat; p.n_mat = c->n_mat;
    p.skew = 0; p.taper = 0; p.ksplit = c->ksplit; p.Gpart = c->d_Gpart;
    p.tickets = c->d_tickets; p.blk_sums = c->d_blk_sums; p.ctl = c->d_ctl; p.conv_mode = conv_mode; p.conv_thr = thr;
    // plain passes (dr_solver_step) do not form the sums; whoever asks for them afterwards gets them from k_chunk_sums
    p.want_sums = conv_mode != 0 ? 1 : 0;
    if (conv_mode != 0 && !c->tails_valid) {
        HIPCHK(launch_chunk_sums(c->stream, c->d_R[c->cur], c->world, c->S, c->rpr, c->cstride));
        c->tails_valid = true;
    }
    if (conv_mode == 0) c->tails_valid = false;
    p.tune = sweep_tuning(c->opt);
    p.n_peers = 0; p.seq = 0;
    for (int q = 0; q < MAX_GROUP; q++) { p.peers[q] = nullptr; p.peer_seq[q] = nullptr; }
    if (!c->inpass_peers.empty()) {
        // in-pass exchange: this pass stores its chunk into every peer's buffer of the same parity and publishes its number;
        // it starts behind a gate that waits until every rank has published the previous pass
        p.n_peers = (int)c->inpass_peers.size();
        for (int q = 0; q < p.n_peers; q++) {
            dr_context* pc = c->inpass_peers[q];
            p.peers[q] = pc == c ? nullptr : pc->d_R[c->cur ^ 1];
            p.peer_seq[q] = pc->d_seq;
        }
        p.seq = c->pass_seq + 1;
        if (c->pass_seq > 0) HIPCHK(launch_wait_peers(c->stream, c->d_seq, c->world, c->pass_seq, c->d_ctl + 3));
        c->pass_seq++;
    }
    p.tile_mask = nullptr; p.mask_words = 0;
    if (c->skip_zero) {
        if (!c->mask_valid) { int rc = build_tile_mask(c); if (rc) return rc; }
        p.tile_mask = c->d_mask; p.mask_words = c->mask_words;
    }
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (c->profile) {
        if (c->ev_used == c->ev_pool.size()) {
            if (c->ev_pool.size() >= 4096) {      // bounded pool: fold what has completed
                HIPCHK(hipStreamSynchronize(c->stream));
                drain_events(c);
            } else {
                hipEvent_t a, b;
                HIPCHK(hipEventCreate(&a));
                HIPCHK(hipEventCreate(&b));
                c->ev_pool.push_back({ a, b });
            }
        }
        e0 = c->ev_pool[c->ev_used].first; e1 = c->ev_pool[c->ev_used].second;
        c->ev_used++;
        HIPCHK(hipEventRecord(e0, c->stream));
    }
    HIPCHK(launch_sweep(c->stream, p));
    if (c->profile) HIPCHK(hipEventRecord(e1, c->stream));
    if (c->comm.comm && !c->manual_exchange) {
        std::string err = comm_allgather_inplace(c->comm, c->d_R[c->cur ^ 1], c->cstride, c->stream);
        if (!err.empty()) return fail(DR_ERR_COMM, "%s", err.c_str());
    }
    c->cur ^= 1;
    return DR_OK;
}

// per-bin sums of the current residual: the chunks' tails (kept by the passes themselves), added up chunk by chunk
int read_sums(dr_context* c, double* sums) {
    if (!c->tails_valid) {
        HIPCHK(launch_chunk_sums(c->stream, c->d_R[c->cur], c->world, c->S, c->rpr, c->cstride));
        c->tails_valid = true;
    }
    std::vector<double> tails((size_t)c->world * MAX_BINS);
    HIPCHK(hipMemcpy2DAsync(tails.data(), sizeof(double) * MAX_BINS, c->d_R[c->cur] + (size_t)c->S * c->rpr, sizeof(float) * c->cstride,
                            sizeof(double) * MAX_BINS, (size_t)c->world, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    for (int s = 0; s < c->S; s++) {
        double t = 0.0;
        for (int r = 0; r < c->world; r++) t += tails[(size_t)r * MAX_BINS + s];
        sums[s] = t;
    }
    return DR_OK;
}

}  // namespace

extern "C" {

const char* dr_last_error(void) { return g_err.c_str(); }

int dr_context_create(int device_id, dr_context** out) {
    if (!out) return fail(DR_ERR_INVALID, "out is null");
    int n = 0;
    HIPCHK(hipGetDeviceCount(&n));
    if (device_id < 0 || device_id >= n) return fail(DR_ERR_INVALID, "device %d out of range (0..%d)", device_id, n - 1);
    HIPCHK(hipSetDevice(device_id));
    dr_context* c = new dr_context();
    c->device = device_id;
    hipError_t e = hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking);
    if (e != hipSuccess) { delete c; return fail(DR_ERR_DEVICE, "hipStreamCreate: %s", hipGetErrorString(e)); }
    c->stream = c->own_stream;
    options_from_env(&c->opt);
    e = hipMalloc(&c->d_counter, 16 * sizeof(unsigned long long));
    if (e != hipSuccess) { hipStreamDestroy(c->own_stream); delete c; return fail(DR_ERR_NOMEM, "hipMalloc: %s", hipGetErrorString(e)); }
    *out = c;
    return DR_OK;
}

int dr_options_defaults(dr_options* out) {
    if (!out) return fail(DR_ERR_INVALID, "out is null");
    options_from_env(out);
    return DR_OK;
}

int dr_set_options(dr_context* c, const dr_options* o) {
    CTX(c);
    int rc = check_options(o);
    if (rc) return rc;
    // a solver that was laid out under other pass options keeps them until the next dr_solver_init: the buffers depend on them
    if (c->have_solver && (o->sweep_ksplit != c->opt.sweep_ksplit || o->sweep_rows_per_wave != c->opt.sweep_rows_per_wave || o->sweep_mfma != c->opt.sweep_mfma))
        return fail(DR_ERR_STATE, "the pass layout options (sweep_ksplit, sweep_rows_per_wave, sweep_mfma) cannot change under an initialised solver: set them before dr_solver_init");
    c->opt = *o;
    return DR_OK;
}

int dr_get_options(dr_context* c, dr_options* out) {
    CTX(c);
    if (!out) return fail(DR_ERR_INVALID, "out is null");
    *out = c->opt;
    return DR_OK;
}

int dr_context_destroy(dr_context* c) {
    if (!c) return DR_OK;
    hipSetDevice(c->device);
    hipStreamSynchronize(c->stream);
    comm_destroy(c->comm);
    free_solver(c); free_F(c); free_scene(c);
    hipFree(c->d_uv); hipFree(c->d_counter); hipFree(c->d_agree);
    delete c->asm_ev;
    for (auto& p : c->ev_pool) { hipEventDestroy(p.first); hipEventDestroy(p.second); }
    if (c->own_stream) hipStreamDestroy(c->own_stream);
    delete c;
    return DR_OK;
}

int dr_set_stream(dr_context* c, void* hip_stream) {
    CTX(c);
    HIPCHK(hipStreamSynchronize(c->stream));
    c->stream = hip_stream ? (hipStream_t)hip_stream : c->own_stream;
    return DR_OK;
}

int dr_set_shard(dr_context* c, int rank, int world) {
    CTX(c);
    if (world < 1 || rank < 0 || rank >= world) return fail(DR_ERR_INVALID, "bad shard %d/%d", rank, world);
    if ((rank != c->rank || world != c->world) && c->comm.comm)
        return fail(DR_ERR_STATE, "the RCCL communicator of this context is rank %d of %d: the shard cannot change under it", c->comm.rank, c->comm.world);
    if (rank != c->rank || world != c->world) { free_F(c); free_solver(c); hipFree(c->d_agree); c->d_agree = nullptr; }
    c->rank = rank; c->world = world;
    recompute_shard(c);
    return DR_OK;
}

int dr_shard_rows(int N, int rank, int world, int* row0, int* nrows, int* rpr) {
    if (N < 1 || world < 1 || rank < 0 || rank >= world || !row0 || !nrows || !rpr)
        return fail(DR_ERR_INVALID, "bad shard query N=%d rank=%d world=%d", N, rank, world);
    shard_rows(N, rank, world, row0, nrows, rpr);
    return DR_OK;
}

int dr_vis_exchange_tracer(int N, int world, int patch_a, int patch_b) {
    if (N < 1 || world < 1 || patch_a < 0 || patch_b < 0 || patch_a >= N || patch_b >= N) return -1;
    int row0, nrows, rpr;
    shard_rows(N, 0, world, &row0, &nrows, &rpr);
    const int ta = patch_a / TILE, tb = patch_b / TILE, T = rpr / TILE;
    if (ta / T == tb / T) return ta / T;           // both in one rank's rows: that rank
    return vx_tracer_rank(ta, tb, T);
}

size_t dr_residual_offset(int i, int s, int S, int rpr) {
    // chunk-major (one chunk per rank), bin-major inside a chunk, each chunk followed by its per-bin sums: see sweep_kernels.hip
    return (size_t)(i / rpr) * dr_residual_chunk_floats(S, rpr) + (size_t)s * rpr + (size_t)(i % rpr);
}

size_t dr_residual_chunk_floats(int S, int rpr) { return (size_t)S * rpr + RTAIL; }

int dr_get_shard(dr_context* c, int* row0, int* nrows, int* rpr) {
    CTX(c);
    if (c->N <= 0) return fail(DR_ERR_STATE, "no mesh");
    if (row0) *row0 = c->row0;
    if (nrows) *nrows = c->nrows;
    if (rpr) *rpr = c->rpr;
    return DR_OK;
}

int dr_scene_set_mesh(dr_context* c, const float* vertices, int V, const float* normals, int Nn,
                      const int32_t* tv, const int32_t* tn, int N) {
    CTX(c);
    if (!vertices || !normals || !tv || !tn) return fail(DR_ERR_INVALID, "null mesh array");
    if (V <= 0 || Nn <= 0 || N <= 0) return fail(DR_ERR_INVALID, "empty mesh (V=%d Nn=%d N=%d)", V, Nn, N);
    if ((size_t)N > (1u << 22)) return fail(DR_ERR_INVALID, "N=%d exceeds %u patches", N, 1u << 22);
    float lo[3] = { INFINITY, INFINITY, INFINITY }, hi[3] = { -INFINITY, -INFINITY, -INFINITY };
    for (size_t k = 0; k < (size_t)3 * N; k++) {
        if (tv[k] < 0 || tv[k] >= V) return fail(DR_ERR_INVALID, "vertex index %d of triangle %zu out of range [0,%d)", tv[k], k / 3, V);
        if (tn[k] < 0 || tn[k] >= Nn) return fail(DR_ERR_INVALID, "normal index %d of triangle %zu out of range [0,%d)", tn[k], k / 3, Nn);
        const float* p = vertices + 3 * (size_t)tv[k];
        for (int a = 0; a < 3; a++) {
            if (!std::isfinite(p[a])) return fail(DR_ERR_INVALID, "non-finite vertex %d", tv[k]);
            lo[a] = std::min(lo[a], p[a]); hi[a] = std::max(hi[a], p[a]);
        }
    }
    // padding of every triangle's gate box: 1e-4 of the largest side of the scene (fp32, as the oracle)
    const float box_pad = 1e-4f * std::max(hi[0] - lo[0], std::max(hi[1] - lo[1], hi[2] - lo[2])) + 1e-30f;
    // growth of every BVH node box that makes the cheap node test conservative (node_hit_mask, geom_kernels.hip)
    float diag2 = 0.0f, maxabs = 0.0f;
    for (int a = 0; a < 3; a++) {
        const float ext = (hi[a] - lo[a]) + 2.0f * box_pad;
        diag2 += ext * ext;
        maxabs = std::max(maxabs, std::max(std::fabs(lo[a]), std::fabs(hi[a])) + box_pad);
    }
    if (!(maxabs < 1e15f)) return fail(DR_ERR_INVALID, "scene coordinates up to %g exceed the supported 1e15", (double)maxabs);
    const float node_pad = 3e-5f * std::sqrt(diag2) + 4e-6f * maxabs;
    HIPCHK(hipStreamSynchronize(c->stream));
    free_solver(c); free_F(c); free_scene(c);      // c->N stays 0 ("no mesh") until everything below has succeeded
    HIPCHK(hipMalloc(&c->d_vtx, sizeof(float) * 3 * (size_t)V));
    HIPCHK(hipMalloc(&c->d_nrm, sizeof(float) * 3 * (size_t)Nn));
    HIPCHK(hipMalloc(&c->d_tv, sizeof(int) * 3 * (size_t)N));
    HIPCHK(hipMalloc(&c->d_tn, sizeof(int) * 3 * (size_t)N));
    HIPCHK(hipMalloc(&c->d_patch, sizeof(PatchRec) * (size_t)N));
    HIPCHK(hipMalloc(&c->d_tri, sizeof(TriRec) * (size_t)N));
    HIPCHK(hipMalloc(&c->d_tri_sorted, sizeof(TriRec) * ((size_t)N + LEAF_MAX)));
    c->n_nodes = 2 * N - 1;
    HIPCHK(hipMalloc(&c->d_bvh, sizeof(BvhNode) * ((size_t)c->n_nodes + 2)));     // + the sentinel + one node the walk's prefetch may touch
    HIPCHK(hipMalloc(&c->d_bvh_lh, sizeof(BvhNode) * ((size_t)c->n_nodes + 2)));
    HIPCHK(hipMalloc(&c->d_pairs, sizeof(BvhPair) * (size_t)std::max(N - 1, 1)));
    HIPCHK(hipMalloc(&c->d_pairs_lh, sizeof(BvhPair) * (size_t)std::max(N - 1, 1)));
    // the per-patch path records (PATH_RECS x 32 B per patch; their byte offsets must fit 32 bits) only for the walk that
    // streams them (dr_options::walk = PATHS: exact, measured no faster -- profiles/r03/assembly_notes.md)
    if (c->opt.walk == DR_WALK_PATHS && (double)N * PATH_RECS * sizeof(BvhNode) < 4.0e9) {
        HIPCHK(hipMalloc(&c->d_path_rec, sizeof(BvhNode) * ((size_t)N * PATH_RECS + 1)));
        HIPCHK(hipMalloc(&c->d_path_hdr, sizeof(PathHdr) * (size_t)N));
    }
    HIPCHK(hipMemcpyAsync(c->d_vtx, vertices, sizeof(float) * 3 * (size_t)V, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(c->d_nrm, normals, sizeof(float) * 3 * (size_t)Nn, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(c->d_tv, tv, sizeof(int) * 3 * (size_t)N, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(c->d_tn, tn, sizeof(int) * 3 * (size_t)N, hipMemcpyHostToDevice, c->stream));
    HIPCHK(launch_patch_records(c->stream, N, c->d_vtx, c->d_nrm, c->d_tv, c->d_tn, box_pad, c->d_patch, c->d_tri));
    EventPair ev;
    HIPCHK(ev.create());
    HIPCHK(hipEventRecord(ev.a, c->stream));
    TreeOptions topt;
    topt.sah = c->opt.tree == DR_TREE_SAH || (c->opt.tree == DR_TREE_AUTO && N >= 6144);
    topt.sah_on_host = c->opt.sah_on_host != 0;
    topt.morton_key = c->opt.morton_key; topt.sah_bins = c->opt.sah_bins; topt.sah_dilate = c->opt.sah_dilate; topt.sah_host_threads = c->opt.sah_host_threads;
    c->tree_used = topt.sah ? DR_TREE_SAH : DR_TREE_LBVH;
    c->tree_on_host = topt.sah && topt.sah_on_host;
    hipError_t be = build_lbvh(c->stream, N, c->d_tri, lo, hi, node_pad, c->d_bvh, c->d_bvh_lh, c->d_tri_sorted, &c->n_nodes, c->d_path_rec, c->d_path_hdr,
                               topt, c->shared_sah, c->d_pairs, c->d_pairs_lh, &c->tree_depth);
    if (be != hipSuccess) return fail(DR_ERR_DEVICE, "LBVH build failed: %s", hipGetErrorString(be));
    HIPCHK(hipEventRecord(ev.b, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    c->last_bvh_ms = ev.ms();
    c->N = N; c->V = V; c->Nn = Nn;
    c->scene_span = maxabs + std::sqrt(diag2);
    recompute_shard(c);
    return DR_OK;
}

// ---- assembly -------------------------------------------------------------------------------------------
// vx_mode 0: self-sufficient (every pair that touches this rank's rows is traced here); 1 / 2: the two launches of an
// assembly with ray-count exchange (TileParams); 2 reuses what 1 set up.  Three steps so that a group of contexts in
// one process can run them on all its devices at once: prepare (allocations, uploads), launch (asynchronous),
// complete (wait, bookkeeping).
static int assemble_prepare(dr_context* c, const float* uv, int K, float eps, int rule, int keep_vis, int trace, int vx_mode) {
    CTX(c);
    if (c->N <= 0) return fail(DR_ERR_STATE, "dr_scene_set_mesh has not been called");
    if (trace && vx_mode != 2 && (!uv || K < 1 || K > 254)) return fail(DR_ERR_INVALID, "need 1 <= K <= 254 samples (K=%d)", K);
    if (rule != DR_RULE_INTEGRAND && rule != DR_RULE_RECIPROCITY) return fail(DR_ERR_INVALID, "unknown rule %d", rule);
    if (c->comm.comm && (c->comm.rank != c->rank || c->comm.world != c->world))
        return fail(DR_ERR_STATE, "RCCL communicator is rank %d of %d, the shard is %d of %d", c->comm.rank, c->comm.world, c->rank, c->world);
    const int tiles_per_rank = c->rpr / TILE;
    if (vx_mode == 2) return DR_OK;
    int rc = ensure_F(c);
    if (rc) return rc;
    c->have_F = false;
    c->mask_valid = false;
    c->split_pending = false;
    if (c->ldF != (size_t)c->N) HIPCHK(hipMemsetAsync(c->d_F, 0, c->F_floats * sizeof(float), c->stream));
    hipFree(c->d_vis); c->d_vis = nullptr;
    if (keep_vis && c->nrows > 0) HIPCHK(hipMalloc(&c->d_vis, (size_t)c->nrows * c->N));
    if (trace) {
        hipFree(c->d_uv); c->d_uv = nullptr; c->uv_K = 0;
        HIPCHK(hipMalloc(&c->d_uv, sizeof(float) * 2 * (size_t)K));
        HIPCHK(hipMemcpyAsync(c->d_uv, uv, sizeof(float) * 2 * (size_t)K, hipMemcpyHostToDevice, c->stream));
        c->uv_K = K;
    }
    HIPCHK(hipMemsetAsync(c->d_counter, 0, 16 * sizeof(unsigned long long), c->stream));
    if (vx_mode == 1) {
        const size_t block = (size_t)tiles_per_rank * tiles_per_rank * (TILE * TILE);
        if (c->vx_block != block || !c->d_vsend || !c->d_vrecv) {
            hipFree(c->d_vsend); hipFree(c->d_vrecv); c->d_vsend = c->d_vrecv = nullptr; c->vx_block = 0;
            hipError_t e = hipMalloc(&c->d_vsend, block * c->world);
            if (e == hipSuccess) e = hipMalloc(&c->d_vrecv, block * c->world);
            if (e != hipSuccess) {
                hipFree(c->d_vsend); c->d_vsend = c->d_vrecv = nullptr;
                return fail(DR_ERR_NOMEM, "ray-count exchange buffers of 2 x %.2f GB: %s", block * c->world / 1e9, hipGetErrorString(e));
            }
            c->vx_block = block;
        }
    }
    c->split_K = K; c->split_rule = rule; c->split_eps = eps;
    return DR_OK;
}

static int assemble_launch(dr_context* c, int K, float eps, int rule, int trace, int vx_mode) {
    CTX(c);
    delete c->asm_ev;
    c->asm_ev = new EventPair();
    HIPCHK(c->asm_ev->create());
    HIPCHK(hipEventRecord(c->asm_ev->a, c->stream));
    if (c->nrows > 0) {
        TileParams p;
        p.N = c->N; p.K = trace ? K : 1; p.rule = rule; p.trace = trace;
        p.nT = (c->N + TILE - 1) / TILE; p.tile0 = c->row0 / TILE; p.nOwnedTiles = (c->nrows + TILE - 1) / TILE;
        p.row0 = c->row0; p.nrows = c->nrows; p.n_nodes = c->n_nodes; p.eps = eps; p.ldF = c->ldF;
        p.F = c->d_F; p.vis = c->d_vis; p.patch = c->d_patch; p.tri = c->d_tri; p.tri_sorted = c->d_tri_sorted; p.bvh = c->d_bvh;
        p.uv = c->d_uv; p.pairs_traced = c->d_counter;
        // dr_options::octant_test = 0: the general node test for every pair (A/B runs)
        p.bvh_lh = c->opt.octant_test ? c->d_bvh_lh : nullptr;
        // the walk: over the sibling-pair records unless the tree is deeper than the walk's stack of PAIR_STACK items (or the
        // options say threaded).  From the root the stack holds at most one item per level; with path records at most
        // (levels of lo's path) + (levels the walk adds below a popped sibling of hi's path, or hi's own levels) + the two leaves
        // <= 2 * depth + 2 -- deeper trees fall back from path records to the walk from the root, and from that to the threaded tree
        const bool pw = c->opt.walk != DR_WALK_THREADED && c->tree_depth <= PAIR_STACK - 2;
        p.pairs = pw ? c->d_pairs : nullptr; p.pairs_lh = (pw && p.bvh_lh) ? c->d_pairs_lh : nullptr;
        // ... its stack filled from the two patches' path records when those were built (dr_options::walk = PATHS at set_mesh)
        { const bool on = pw && c->opt.walk == DR_WALK_PATHS && c->d_path_rec && c->d_path_hdr && 2 * c->tree_depth + 2 <= PAIR_STACK;
          p.path_rec = on ? c->d_path_rec : nullptr; p.path_hdr = on ? c->d_path_hdr : nullptr; }
        c->walk_used = !pw ? DR_WALK_THREADED : (p.path_hdr ? DR_WALK_PATHS : DR_WALK_PAIRS);
        p.vx_mode = vx_mode; p.vx_rank = c->rank; p.vx_tiles_per_rank = c->rpr / TILE; p.vsend = c->d_vsend; p.vrecv = c->d_vrecv;
        p.ts_max = c->scene_span > 1e-19f ? std::min(1e19f / c->scene_span, 1e18f) : 1e18f;
        p.stats = c->opt.tile_stats ? 1 : 0;
        p.dbg_lo = c->opt.debug_pair_lo; p.dbg_hi = c->opt.debug_pair_hi; p.dbg_ray = c->opt.debug_ray;
        if (p.dbg_lo >= 0) p.stats = 1;
        HIPCHK(launch_ff_tiles(c->stream, p));
    }
    HIPCHK(hipEventRecord(c->asm_ev->b, c->stream));
    HIPCHK(hipMemcpyAsync(c->h_cnt, c->d_counter, sizeof c->h_cnt, hipMemcpyDeviceToHost, c->stream));
    return DR_OK;
}

static int assemble_complete(dr_context* c, int vx_mode) {
    CTX(c);
    HIPCHK(hipStreamSynchronize(c->stream));
    const unsigned long long* cnt = c->h_cnt;
    c->last_assemble_ms = (vx_mode == 2 ? c->last_assemble_ms : 0.0) + (c->asm_ev ? c->asm_ev->ms() : 0.0);
    delete c->asm_ev; c->asm_ev = nullptr;
    c->pairs_traced = cnt[0]; c->stat_visits = cnt[1]; c->stat_leaves = cnt[2];
    if (c->opt.debug_pair_lo >= 0)
        fprintf(stderr, "[daisyriot] debug pair: live mask after target test %016llx, final %016llx; blocker of the debug ray: patch %lld "
                        "t %08llx tmax %08llx node %llu\n", cnt[1], cnt[3], (long long)cnt[4], cnt[5], cnt[6], cnt[7]);
    else if (c->opt.tile_stats)
        fprintf(stderr, "[daisyriot] pairs %llu, node visits %llu (%.1f/pair), leaf tests %llu (%.1f/pair); path records streamed %.1f/pair; "
                        "waves with mixed octants %llu; tree depth %d\n", cnt[0], cnt[1], cnt[0] ? (double)cnt[1] / cnt[0] : 0.0, cnt[2],
                cnt[0] ? (double)cnt[2] / cnt[0] : 0.0, cnt[0] ? (double)cnt[12] / cnt[0] : 0.0, cnt[13], c->tree_depth);
    if (vx_mode == 1) {
        c->split_pending = true;
    } else {
        c->split_pending = false;
        c->have_F = true;
        c->mask_valid = false;
        if (vx_mode == 2) {       // 2 N^2/world bytes: not kept
            hipFree(c->d_vsend); hipFree(c->d_vrecv); c->d_vsend = c->d_vrecv = nullptr; c->vx_block = 0;
        }
    }
    return DR_OK;
}

static int assemble_impl(dr_context* c, const float* uv, int K, float eps, int rule, int keep_vis, int trace, int vx_mode = 0) {
    int rc = assemble_prepare(c, uv, K, eps, rule, keep_vis, trace, vx_mode);
    if (rc) return rc;
    rc = assemble_launch(c, K, eps, rule, trace, vx_mode);
    if (rc) return rc;
    return assemble_complete(c, vx_mode);
}

// Device bytes one rank of a world-way assembly WITH ray-count exchange holds: the two slot buffers, its F shard, the
// optional ray counts.  A function of rank-independent quantities only (rows per rank, not this rank's rows), so every
// rank of a homogeneous node takes the same decision -- a rank that decided alone would strand its peers in the collective.
static double vx_bytes_needed(int N, int world, int keep_vis) {
    int row0, nrows, rpr;
    shard_rows(N, 0, world, &row0, &nrows, &rpr);
    const double T = rpr / TILE;
    return 2.0 * world * T * T * (TILE * TILE) + 4.0 * rpr * ((double)world * rpr) + (keep_vis ? (double)rpr * N : 0.0);
}

int dr_vis_exchange_fits(int N, int world, int keep_visibility, size_t device_bytes) {
    if (N < 1 || world < 1) return 0;
    return vx_bytes_needed(N, world, keep_visibility) <= 0.85 * (double)device_bytes ? 1 : 0;
}

// every rank's go (1) / no-go (0) before a collective: all ranks return the minimum
// The buffer exists since dr_comm_init (no allocation here that could fail on one rank only); a local HIP failure on the way
// in does not skip the all-gather -- the peers are in it -- it is reported afterwards, and the verdict is then "no".
static int comm_agree(dr_context* c, int ok, int* all_ok) {
    *all_ok = 0;
    if (!c->d_agree) return fail(DR_ERR_STATE, "no agreement buffer: dr_comm_init has not been called");
    hipError_t he = hipMemcpyAsync(c->d_agree + c->rank, &ok, sizeof(int), hipMemcpyHostToDevice, c->stream);
    if (he == hipSuccess) he = hipStreamSynchronize(c->stream);      // `ok` is a stack variable
    std::string err = comm_allgather_i32_inplace(c->comm, c->d_agree, c->stream);
    if (he != hipSuccess) return fail(DR_ERR_DEVICE, "go / no-go upload failed: %s", hipGetErrorString(he));
    if (!err.empty()) return fail(DR_ERR_COMM, "%s", err.c_str());
    std::vector<int> h((size_t)c->world);
    HIPCHK(hipMemcpyAsync(h.data(), c->d_agree, sizeof(int) * h.size(), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    *all_ok = 1;
    for (int v : h) *all_ok = std::min(*all_ok, v);
    return DR_OK;
}

int dr_formfactors_assemble(dr_context* c, const float* uv, int K, float eps, int rule, int keep_vis) {
    if (!c) return fail(DR_ERR_INVALID, "null context");
    // several ranks joined by RCCL: every pair between two ranks' rows is traced by one of them only and its 64 x 64
    // ray counts travel to the other (one all-to-all of slot blocks); otherwise the rank is self-sufficient
    // (dr_options::vis_exchange = 2: take this path with a single-rank communicator too -- a one-GPU rehearsal of the calls)
    bool exchange = (c->world > 1 || c->opt.vis_exchange == 2) && c->comm.comm && c->opt.vis_exchange != 1;
    if (exchange) {
        // when the slot buffers do not fit beside the F shard every rank traces for itself.  Decided from rank-independent
        // numbers and the card's TOTAL memory, so that all ranks of a homogeneous node decide alike.
        HIPCHK(hipSetDevice(c->device));
        hipDeviceProp_t prop;
        HIPCHK(hipGetDeviceProperties(&prop, c->device));
        if (!dr_vis_exchange_fits(c->N, c->world, keep_vis, prop.totalGlobalMem)) exchange = false;
    }
    if (!exchange) return assemble_impl(c, uv, K, eps, rule, keep_vis, 1);
    // allocations first, then the ranks agree that all of them can go on: an out-of-memory on one rank must not leave the
    // others waiting in the collective
    int rc = assemble_prepare(c, uv, K, eps, rule, keep_vis, 1, 1);
    int all_ok = 0;
    int rc2 = comm_agree(c, rc == DR_OK ? 1 : 0, &all_ok);
    if (rc) return rc;
    if (rc2) return rc2;
    if (!all_ok) return fail(DR_ERR_COMM, "another rank could not set up the assembly; nothing was traced");
    // the first launch, then a second go / no-go: a rank whose kernel failed to launch or faulted must not leave the others
    // waiting in the all-to-all (it still takes part in the agreement -- a failed launch leaves the stream usable)
    rc = (c->opt.fault_assemble_rank == c->rank) ? fail(DR_ERR_DEVICE, "injected failure of the first assembly launch (dr_options::fault_assemble_rank)")
                                              : assemble_launch(c, K, eps, rule, 1, 1);
    if (rc == DR_OK) rc = assemble_complete(c, 1);
    const std::string first_err = rc ? g_err : std::string();
    rc2 = comm_agree(c, rc == DR_OK ? 1 : 0, &all_ok);
    if (rc) return fail(rc, "%s", first_err.c_str());
    if (rc2) return rc2;
    if (!all_ok) { c->split_pending = false; return fail(DR_ERR_COMM, "another rank failed in the first assembly launch; the ray counts were not exchanged"); }
    std::string err = comm_alltoall_bytes(c->comm, c->d_vsend, c->d_vrecv, c->vx_block, c->stream);
    if (!err.empty()) return fail(DR_ERR_COMM, "%s", err.c_str());
    return assemble_impl(c, nullptr, c->split_K, c->split_eps, c->split_rule, 0, 1, 2);
}

/* the same in three steps for a host that moves the slot blocks itself (tests, MPI staging) */
int dr_formfactors_assemble_split(dr_context* c, const float* uv, int K, float eps, int rule, int keep_vis) {
    if (!c) return fail(DR_ERR_INVALID, "null context");
    return assemble_impl(c, uv, K, eps, rule, keep_vis, 1, 1);
}

int dr_vis_exchange_bytes(dr_context* c, size_t* block_bytes) {
    CTX(c);
    if (!block_bytes) return fail(DR_ERR_INVALID, "block_bytes is null");
    if (!c->split_pending) return fail(DR_ERR_STATE, "dr_formfactors_assemble_split has not been called");
    *block_bytes = c->vx_block;
    return DR_OK;
}

int dr_vis_exchange_export(dr_context* c, int dst_rank, void* out, size_t bytes) {
    CTX(c);
    if (!out || dst_rank < 0 || dst_rank >= c->world) return fail(DR_ERR_INVALID, "bad destination rank %d", dst_rank);
    if (!c->split_pending) return fail(DR_ERR_STATE, "dr_formfactors_assemble_split has not been called");
    if (bytes != c->vx_block) return fail(DR_ERR_INVALID, "a slot block is %zu bytes (dr_vis_exchange_bytes), got %zu", c->vx_block, bytes);
    HIPCHK(hipMemcpyAsync(out, c->d_vsend + (size_t)dst_rank * c->vx_block, c->vx_block, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return DR_OK;
}

int dr_vis_exchange_import(dr_context* c, int src_rank, const void* in, size_t bytes) {
    CTX(c);
    if (!in || src_rank < 0 || src_rank >= c->world) return fail(DR_ERR_INVALID, "bad source rank %d", src_rank);
    if (!c->split_pending) return fail(DR_ERR_STATE, "dr_formfactors_assemble_split has not been called");
    if (bytes != c->vx_block) return fail(DR_ERR_INVALID, "a slot block is %zu bytes (dr_vis_exchange_bytes), got %zu", c->vx_block, bytes);
    HIPCHK(hipMemcpyAsync(c->d_vrecv + (size_t)src_rank * c->vx_block, in, c->vx_block, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return DR_OK;
}

int dr_formfactors_assemble_finish(dr_context* c) {
    if (!c) return fail(DR_ERR_INVALID, "null context");
    if (!c->split_pending) return fail(DR_ERR_STATE, "dr_formfactors_assemble_split has not been called");
    return assemble_impl(c, nullptr, c->split_K, c->split_eps, c->split_rule, 0, 1, 2);
}
int dr_formfactors_reserve(dr_context* c) {
    CTX(c);
    if (c->N <= 0) return fail(DR_ERR_STATE, "dr_scene_set_mesh has not been called");
    return ensure_F(c);
}

int dr_formfactors_integrand_only(dr_context* c) {
    return assemble_impl(c, nullptr, 1, 0.0f, DR_RULE_INTEGRAND, 0, 0);
}

static int check_rows(dr_context* c, int row0, int nrows) {
    if (nrows < 0 || row0 < c->row0 || row0 + nrows > c->row0 + c->nrows)
        return fail(DR_ERR_INVALID, "rows [%d,%d) not inside this rank's rows [%d,%d)", row0, row0 + nrows, c->row0, c->row0 + c->nrows);
    return DR_OK;
}

int dr_formfactors_read_rows(dr_context* c, int row0, int nrows, float* out) {
    CTX(c);
    if (!c->have_F) return fail(DR_ERR_STATE, "form factors not assembled");
    if (!out) return fail(DR_ERR_INVALID, "out is null");
    int rc = check_rows(c, row0, nrows);
    if (rc) return rc;
    if (nrows == 0) return DR_OK;
    HIPCHK(hipMemcpy2DAsync(out, sizeof(float) * c->N, c->d_F + (size_t)(row0 - c->row0) * c->ldF, sizeof(float) * c->ldF,
                            sizeof(float) * c->N, nrows, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return DR_OK;
}

int dr_visibility_read_rows(dr_context* c, int row0, int nrows, uint8_t* out) {
    CTX(c);
    if (!c->have_F) return fail(DR_ERR_STATE, "form factors not assembled");
    if (!out) return fail(DR_ERR_INVALID, "out is null");
    int rc = check_rows(c, row0, nrows);
    if (rc) return rc;
    if (nrows == 0) return DR_OK;          // (a rank without rows keeps no counts)
    if (!c->d_vis) return fail(DR_ERR_STATE, "visibility counts were not kept (keep_visibility = 0)");
    HIPCHK(hipMemcpyAsync(out, c->d_vis + (size_t)(row0 - c->row0) * c->N, (size_t)nrows * c->N, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return DR_OK;
}

int dr_formfactors_load_rows(dr_context* c, int row0, int nrows, const float* F) {
    CTX(c);
    if (c->N <= 0) return fail(DR_ERR_STATE, "dr_scene_set_mesh has not been called");
    if (!F) return fail(DR_ERR_INVALID, "F is null");
    int rc = check_rows(c, row0, nrows);
    if (rc) return rc;
    bool fresh = (c->d_F == nullptr);
    rc = ensure_F(c);
    if (rc) return rc;
    if (fresh) HIPCHK(hipMemsetAsync(c->d_F, 0, c->F_floats * sizeof(float), c->stream));
    if (nrows > 0)
        HIPCHK(hipMemcpy2DAsync(c->d_F + (size_t)(row0 - c->row0) * c->ldF, sizeof(float) * c->ldF, F, sizeof(float) * c->N,
                                sizeof(float) * c->N, nrows, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    c->have_F = true;
    c->mask_valid = false;
    return DR_OK;
}

int dr_solver_init(dr_context* c, int S, const float* E, const float* M, int n_mat, const int32_t* mat_of_patch) {
    CTX(c);
    if (c->N <= 0) return fail(DR_ERR_STATE, "dr_scene_set_mesh has not been called");
    if (S < 1 || S > DR_MAX_BINS) return fail(DR_ERR_INVALID, "S=%d outside 1..%d", S, DR_MAX_BINS);
    if (!E || !M || !mat_of_patch || n_mat < 1) return fail(DR_ERR_INVALID, "null solver input");
    for (int i = 0; i < c->N; i++)
        if (mat_of_patch[i] < 0 || mat_of_patch[i] >= n_mat)
            return fail(DR_ERR_INVALID, "material index %d of patch %d out of range [0,%d)", mat_of_patch[i], i, n_mat);
    // dr_options::no_comm: time one rank's shard alone (its residual chunks of other ranks stay at E)
    if (c->world > 1 && !c->comm.comm && !c->manual_exchange && !c->opt.no_comm)
        return fail(DR_ERR_STATE, "world=%d but dr_comm_init has not been called", c->world);
    HIPCHK(hipStreamSynchronize(c->stream));
    free_solver(c);
    c->S = S; c->n_mat = n_mat;
    c->cstride = (size_t)S * c->rpr + RTAIL;
    const size_t full = (size_t)c->world * c->cstride;
    HIPCHK(hipMalloc(&c->d_M, sizeof(float) * (size_t)n_mat * S * S));
    HIPCHK(hipMalloc(&c->d_E, sizeof(float) * full));
    HIPCHK(hipMalloc(&c->d_R[0], sizeof(float) * full));
    HIPCHK(hipMalloc(&c->d_R[1], sizeof(float) * full));
    HIPCHK(hipMalloc(&c->d_B, sizeof(float) * (size_t)S * c->rpr));
    HIPCHK(hipMalloc(&c->d_mat, sizeof(int) * (size_t)c->rpr));
    const SweepTuning tune = sweep_tuning(c->opt);
    c->ksplit = sweep_ksplit(c->nrows, S, (int)c->ldF, tune);
    if (c->ksplit > 1) HIPCHK(hipMalloc(&c->d_Gpart, sizeof(float) * (size_t)c->ksplit * std::max(c->nrows, 1) * S));
    const size_t row_blocks = (size_t)std::max(sweep_row_blocks(c->nrows, S, tune), 1);
    HIPCHK(hipMalloc(&c->d_tickets, sizeof(unsigned) * (1 + row_blocks)));
    HIPCHK(hipMalloc(&c->d_blk_sums, sizeof(double) * row_blocks * S));
    HIPCHK(hipMalloc(&c->d_ctl, sizeof(int) * 4));
    // (written by the other devices of an in-pass-exchanging group: fine-grained where the runtime offers it)
    if (hipExtMallocWithFlags((void**)&c->d_seq, sizeof(unsigned) * MAX_GROUP, hipDeviceMallocFinegrained) != hipSuccess) {
        (void)hipGetLastError();
        HIPCHK(hipMalloc(&c->d_seq, sizeof(unsigned) * MAX_GROUP));
    }
    HIPCHK(hipMemsetAsync(c->d_seq, 0, sizeof(unsigned) * MAX_GROUP, c->stream));
    c->pass_seq = 0;
    HIPCHK(hipMemsetAsync(c->d_tickets, 0, sizeof(unsigned) * (1 + row_blocks), c->stream));
    HIPCHK(hipMemsetAsync(c->d_ctl, 0, sizeof(int) * 4, c->stream));
    HIPCHK(hipMalloc(&c->d_stage, sizeof(float) * std::max((size_t)c->N * S, (size_t)3 * DR_MAX_BINS)));
    float* tmp = c->d_stage;
    HIPCHK(hipMemcpyAsync(tmp, E, sizeof(float) * (size_t)c->N * S, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemsetAsync(c->d_E, 0, sizeof(float) * full, c->stream));
    HIPCHK(hipMemsetAsync(c->d_R[0], 0, sizeof(float) * full, c->stream));
    HIPCHK(hipMemsetAsync(c->d_R[1], 0, sizeof(float) * full, c->stream));
    HIPCHK(launch_scatter_rows(c->stream, tmp, c->N, S, c->rpr, c->cstride, c->d_E));
    HIPCHK(launch_chunk_sums(c->stream, c->d_E, c->world, S, c->rpr, c->cstride));
    HIPCHK(hipMemcpyAsync(c->d_M, M, sizeof(float) * (size_t)n_mat * S * S, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemsetAsync(c->d_mat, 0, sizeof(int) * (size_t)c->rpr, c->stream));
    if (c->nrows > 0)
        HIPCHK(hipMemcpyAsync(c->d_mat, mat_of_patch + c->row0, sizeof(int) * (size_t)c->nrows, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    c->have_solver = true;
    return dr_solver_reset(c);
}

int dr_solver_reset(dr_context* c) {
    CTX(c);
    if (!c->have_solver) return fail(DR_ERR_STATE, "dr_solver_init has not been called");
    const size_t full = (size_t)c->world * c->cstride;
    c->cur = 0;
    HIPCHK(hipMemsetAsync(c->d_seq, 0, sizeof(unsigned) * MAX_GROUP, c->stream));      // (a group resets all its ranks behind a sync)
    c->pass_seq = 0;
    HIPCHK(hipMemcpyAsync(c->d_R[0], c->d_E, sizeof(float) * full, hipMemcpyDeviceToDevice, c->stream));
    c->tails_valid = true;            // d_E carries its sums (dr_solver_init)
    HIPCHK(hipMemcpyAsync(c->d_B, c->d_E + (size_t)c->rank * c->cstride, sizeof(float) * (size_t)c->S * c->rpr,
                          hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return DR_OK;
}

int dr_solver_step(dr_context* c, int n_passes, float* residual_sum_out) {
    CTX(c);
    if (!c->have_solver) return fail(DR_ERR_STATE, "dr_solver_init has not been called");
    if (!c->have_F) return fail(DR_ERR_STATE, "form factors not assembled");
    if (n_passes < 0) return fail(DR_ERR_INVALID, "n_passes < 0");
    for (int k = 0; k < n_passes; k++) {
        int rc = sweep_once(c);
        if (rc) return rc;
    }
    if (residual_sum_out) {
        double sums[DR_MAX_BINS];
        int rc = read_sums(c, sums);
        if (rc) return rc;
        double t = 0;
        for (int s = 0; s < c->S; s++) t += sums[s];
        *residual_sum_out = (float)t;
    }
    return DR_OK;
}

int dr_solver_skip_zero_blocks(dr_context* c, int enable) {
    CTX(c);
    c->skip_zero = enable != 0;
    return DR_OK;
}

int dr_solver_residual_sums(dr_context* c, double* sums) {
    CTX(c);
    if (!c->have_solver) return fail(DR_ERR_STATE, "dr_solver_init has not been called");
    if (!sums) return fail(DR_ERR_INVALID, "sums is null");
    return read_sums(c, sums);
}

int dr_solver_converge(dr_context* c, float threshold, int per_bin, int max_iters, int* iters_out) {
    CTX(c);
    if (!c->have_solver) return fail(DR_ERR_STATE, "dr_solver_init has not been called");
    if (!c->have_F) return fail(DR_ERR_STATE, "form factors not assembled");
    if (max_iters < 0) return fail(DR_ERR_INVALID, "max_iters < 0");
    // The test "has the residual converged?" runs on the device, at the head of every pass, from the per-bin sums the
    // previous pass left in the residual's tails (gathered with it); a pass that finds it converged does nothing.  The
    // host queues passes in batches of check_every and looks at the counters once per batch -- every rank reads the same
    // counters (they are functions of the gathered sums), so all ranks queue the same passes and collectives.
    HIPCHK(hipMemsetAsync(c->d_ctl, 0, sizeof(int) * 4, c->stream));
    const int cur0 = c->cur;
    int queued = 0, ctl[2] = { 0, 0 };
    const int mode = per_bin ? 2 : 1;
    for (;;) {
        const int batch = std::min(c->check_every, max_iters - queued);
        for (int k = 0; k < batch; k++) {
            int rc = sweep_once(c, mode, threshold);
            if (rc) return rc;
        }
        queued += batch;
        if (batch == 0) {
            // cap reached (or max_iters = 0): nothing queued
            break;
        }
        HIPCHK(hipMemcpyAsync(ctl, c->d_ctl, sizeof ctl, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        if (c->opt.debug_converge) fprintf(stderr, "[daisyriot] converge: queued %d, passes done %d, converged flag %d\n", queued, ctl[0], ctl[1]);
        if (ctl[1] != 0 || ctl[0] < queued || queued >= max_iters) break;
    }
    // passes after the converged one did nothing: the current residual is the one the last REAL pass wrote
    c->cur = cur0 ^ (ctl[0] & 1);
    if (iters_out) *iters_out = ctl[0];
    return DR_OK;
}

int dr_solver_set_check_interval(dr_context* c, int passes) {
    CTX(c);
    if (passes < 1) return fail(DR_ERR_INVALID, "check interval %d < 1", passes);
    c->check_every = passes;
    return DR_OK;
}

int dr_solver_read(dr_context* c, float* B, float* R) {
    CTX(c);
    if (!c->have_solver) return fail(DR_ERR_STATE, "dr_solver_init has not been called");
    float* tmp = c->d_stage;
    if (R) {
        HIPCHK(launch_gather_rows(c->stream, c->d_R[c->cur], c->N, c->S, c->rpr, c->cstride, tmp));
        HIPCHK(hipMemcpyAsync(R, tmp, sizeof(float) * (size_t)c->N * c->S, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
    }
    if (B && c->nrows > 0) {
        HIPCHK(launch_gather_rows(c->stream, c->d_B, c->nrows, c->S, c->rpr, (size_t)c->S * c->rpr, tmp));
        HIPCHK(hipMemcpyAsync(B + (size_t)c->row0 * c->S, tmp, sizeof(float) * (size_t)c->nrows * c->S, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
    }
    return DR_OK;
}

int dr_display_patch_colors(dr_context* c, int mode, const float* xyz, float* rgb) {
    CTX(c);
    if (!c->have_solver) return fail(DR_ERR_STATE, "dr_solver_init has not been called");
    if (mode < DR_DISPLAY_BW || mode > DR_DISPLAY_SPECTRAL) return fail(DR_ERR_INVALID, "display mode %d", mode);
    if (mode == DR_DISPLAY_RGB && c->S != 3) return fail(DR_ERR_INVALID, "RGB display needs 3 bins, the solver has %d", c->S);
    if (mode == DR_DISPLAY_BW && c->S != 1) return fail(DR_ERR_INVALID, "BW display needs 1 bin, the solver has %d", c->S);
    if (mode == DR_DISPLAY_SPECTRAL && !xyz) return fail(DR_ERR_INVALID, "xyz_per_bin is null");
    if (!c->d_rgb) HIPCHK(hipMalloc(&c->d_rgb, sizeof(float) * 3 * (size_t)std::max(c->nrows, 1)));
    float* d_xyz = c->d_stage;      // S*3 floats of the (N*S) staging buffer
    if (mode == DR_DISPLAY_SPECTRAL)
        HIPCHK(hipMemcpyAsync(d_xyz, xyz, sizeof(float) * 3 * (size_t)c->S, hipMemcpyHostToDevice, c->stream));
    HIPCHK(launch_patch_colors(c->stream, c->d_B, c->nrows, c->rpr, c->S, mode, d_xyz, c->d_rgb));
    if (rgb && c->nrows > 0)
        HIPCHK(hipMemcpyAsync(rgb, c->d_rgb, sizeof(float) * 3 * (size_t)c->nrows, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    c->have_rgb = true;
    return DR_OK;
}

int dr_display_vertex_colors(dr_context* c, const float* rgb_all, const int32_t* vtx_off, const int32_t* vtx_tri, int V,
                             float* out) {
    CTX(c);
    if (c->N <= 0) return fail(DR_ERR_STATE, "no mesh set");
    if (!vtx_off || !vtx_tri || !out || V < 1) return fail(DR_ERR_INVALID, "null adjacency/output or V < 1");
    if (!rgb_all && (c->world != 1 || !c->have_rgb))
        return fail(DR_ERR_STATE, "rgb_all is null and no device colours of all patches exist (world %d)", c->world);
    if (vtx_off[0] != 0) return fail(DR_ERR_INVALID, "vtx_off[0] must be 0");
    for (int v = 0; v < V; v++)
        if (vtx_off[v + 1] < vtx_off[v]) return fail(DR_ERR_INVALID, "vtx_off decreases at vertex %d", v);
    const int n_adj = vtx_off[V];
    for (int k = 0; k < n_adj; k++)
        if (vtx_tri[k] < 0 || vtx_tri[k] >= c->N) return fail(DR_ERR_INVALID, "vtx_tri[%d] = %d out of range", k, vtx_tri[k]);
    // device buffers are kept in the context and only grow
    auto grow = [](auto*& ptr, size_t& have, size_t want, size_t elem) -> hipError_t {
        if (have >= want && ptr) return hipSuccess;
        hipFree(ptr); ptr = nullptr; have = 0;
        hipError_t er = hipMalloc(&ptr, want * elem);
        if (er == hipSuccess) have = want;
        return er;
    };
    hipError_t e = grow(c->d_voff, c->voff_n, (size_t)V + 1, sizeof(int));
    if (e == hipSuccess) e = grow(c->d_vadj, c->vadj_n, (size_t)std::max(n_adj, 1), sizeof(int));
    if (e == hipSuccess) e = grow(c->d_vout, c->vout_n, (size_t)3 * V, sizeof(float));
    if (e == hipSuccess && rgb_all) e = grow(c->d_vin, c->vin_n, (size_t)3 * c->N, sizeof(float));
    int *d_off = c->d_voff, *d_adj = c->d_vadj;
    float *d_out = c->d_vout, *d_in = c->d_vin;
    if (e == hipSuccess) e = hipMemcpyAsync(d_off, vtx_off, sizeof(int) * ((size_t)V + 1), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess && n_adj > 0) e = hipMemcpyAsync(d_adj, vtx_tri, sizeof(int) * (size_t)n_adj, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess && rgb_all) e = hipMemcpyAsync(d_in, rgb_all, sizeof(float) * 3 * (size_t)c->N, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = launch_vertex_colors(c->stream, rgb_all ? d_in : c->d_rgb, V, d_off, d_adj, d_out);
    if (e == hipSuccess) e = hipMemcpyAsync(out, d_out, sizeof(float) * 3 * (size_t)V, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) return fail(e == hipErrorOutOfMemory ? DR_ERR_NOMEM : DR_ERR_DEVICE, "vertex colours: %s", hipGetErrorString(e));
    return DR_OK;
}

int dr_comm_set_library(const char* path) {
    std::string e = comm_set_library(path);
    if (!e.empty()) return fail(DR_ERR_STATE, "%s", e.c_str());
    return DR_OK;
}

int dr_comm_library_info(char* out, size_t n) {
    if (!out || n == 0) return fail(DR_ERR_INVALID, "out is null");
    const std::string s = comm_library_info();
    snprintf(out, n, "%s", s.c_str());
    return DR_OK;
}

int dr_comm_unique_id(void* out128) {
    if (!out128) return fail(DR_ERR_INVALID, "out128 is null");
    std::string e = comm_unique_id(out128);
    if (!e.empty()) return fail(DR_ERR_COMM, "%s", e.c_str());
    return DR_OK;
}

int dr_comm_init(dr_context* c, const void* id128, int rank, int world) {
    CTX(c);
    if (!id128) return fail(DR_ERR_INVALID, "id128 is null");
    if (rank != c->rank || world != c->world) return fail(DR_ERR_INVALID, "comm %d/%d does not match shard %d/%d", rank, world, c->rank, c->world);
    comm_destroy(c->comm);
    if (!c->d_agree) HIPCHK(hipMalloc(&c->d_agree, sizeof(int) * (size_t)std::max(c->world, 1)));     // comm_agree's buffer
    std::string e = comm_init(c->comm, id128, rank, world);
    if (!e.empty()) return fail(DR_ERR_COMM, "%s", e.c_str());
    return DR_OK;
}

int dr_debug_read_bvh(dr_context* c, void* out, int max_nodes) {
    CTX(c);
    if (c->N <= 0) return fail(DR_ERR_STATE, "no mesh");
    if (!out || max_nodes < c->n_nodes) return fail(DR_ERR_INVALID, "need room for %d nodes", c->n_nodes);
    HIPCHK(hipMemcpyAsync(out, c->d_bvh, sizeof(BvhNode) * (size_t)c->n_nodes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    BvhNode* nd = (BvhNode*)out;            // device form: centre/half-extent, skips as byte offsets; reported: lo/hi, node indices
    for (int i = 0; i < c->n_nodes; i++) {
        nd[i].skip /= (int)sizeof(BvhNode);
        for (int a = 0; a < 3; a++) { const float ce = nd[i].c[a], he = nd[i].h[a]; nd[i].c[a] = ce - he; nd[i].h[a] = ce + he; }
    }
    return DR_OK;
}

int dr_comm_manual(dr_context* c) {
    CTX(c);
    comm_destroy(c->comm);
    c->manual_exchange = true;
    return DR_OK;
}

int dr_exchange_export(dr_context* c, float* out) {
    CTX(c);
    if (!c->have_solver) return fail(DR_ERR_STATE, "dr_solver_init has not been called");
    if (!out) return fail(DR_ERR_INVALID, "chunk_out is null");
    const size_t n = c->cstride;
    HIPCHK(hipMemcpyAsync(out, c->d_R[c->cur] + (size_t)c->rank * n, sizeof(float) * n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return DR_OK;
}

int dr_exchange_import(dr_context* c, int src_rank, const float* in, size_t n_floats) {
    CTX(c);
    if (!c->have_solver) return fail(DR_ERR_STATE, "dr_solver_init has not been called");
    if (!in || src_rank < 0 || src_rank >= c->world) return fail(DR_ERR_INVALID, "bad source rank %d", src_rank);
    const size_t n = c->cstride;
    if (n_floats != n) return fail(DR_ERR_INVALID, "a residual chunk is %zu floats (dr_residual_chunk_floats), got %zu", n, n_floats);
    c->tails_valid = false;           // a host-staged chunk carries whatever tail its exporter had
    HIPCHK(hipMemcpyAsync(c->d_R[c->cur] + (size_t)src_rank * n, in, sizeof(float) * n, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return DR_OK;
}

int dr_debug_sah_topology(int N, const float* boxes, int32_t* order, int32_t* left, int32_t* right, int32_t* first, int32_t* last,
                          int32_t* parent) {
    if (N < 1 || !boxes || !order || !parent || (N > 1 && (!left || !right || !first || !last)))
        return fail(DR_ERR_INVALID, "dr_debug_sah_topology: N >= 1 and every array");
    SahTopology T;
    dr_options o;
    options_from_env(&o);
    TreeOptions topt;
    topt.sah = true; topt.sah_on_host = true; topt.sah_bins = o.sah_bins; topt.sah_dilate = o.sah_dilate; topt.sah_host_threads = o.sah_host_threads;
    sah_topology_from_boxes(N, boxes, T, topt);
    std::memcpy(order, T.order.data(), sizeof(int) * (size_t)N);
    if (N > 1) {
        std::memcpy(left, T.left.data(), sizeof(int) * (size_t)(N - 1)); std::memcpy(right, T.right.data(), sizeof(int) * (size_t)(N - 1));
        std::memcpy(first, T.first.data(), sizeof(int) * (size_t)(N - 1)); std::memcpy(last, T.last.data(), sizeof(int) * (size_t)(N - 1));
    }
    std::memcpy(parent, T.parent.data(), sizeof(int) * (2 * (size_t)N - 1));
    return DR_OK;
}

int dr_debug_sah_topology_device(dr_context* c, int N, const float* boxes, int32_t* order, int32_t* left, int32_t* right, int32_t* first,
                                 int32_t* last, int32_t* parent) {
    CTX(c);
    if (N < 1 || !boxes || !order || !parent || (N > 1 && (!left || !right || !first || !last)))
        return fail(DR_ERR_INVALID, "dr_debug_sah_topology_device: N >= 1 and every array");
    SahTopology T;
    TreeOptions topt;
    topt.sah = true; topt.sah_bins = c->opt.sah_bins; topt.sah_dilate = c->opt.sah_dilate;
    hipError_t e = sah_topology_from_boxes_device(c->stream, N, boxes, T, topt);
    if (e != hipSuccess) return fail(DR_ERR_DEVICE, "device SAH build: %s", hipGetErrorString(e));
    std::memcpy(order, T.order.data(), sizeof(int) * (size_t)N);
    if (N > 1) {
        std::memcpy(left, T.left.data(), sizeof(int) * (size_t)(N - 1)); std::memcpy(right, T.right.data(), sizeof(int) * (size_t)(N - 1));
        std::memcpy(first, T.first.data(), sizeof(int) * (size_t)(N - 1)); std::memcpy(last, T.last.data(), sizeof(int) * (size_t)(N - 1));
    }
    std::memcpy(parent, T.parent.data(), sizeof(int) * (2 * (size_t)N - 1));
    return DR_OK;
}

int dr_debug_read_array(dr_context* c, int which, void* out, size_t bytes) {
    CTX(c);
    const void* src = nullptr;
    size_t have = 0;
    switch (which) {
        case 0: src = c->d_tri; have = sizeof(TriRec) * (size_t)c->N; break;
        case 1: src = c->d_tri_sorted; have = sizeof(TriRec) * ((size_t)c->N + LEAF_MAX); break;
        case 3: src = c->d_uv; have = sizeof(float) * 2 * (size_t)c->uv_K; break;
        case 4: src = c->d_patch; have = sizeof(PatchRec) * (size_t)c->N; break;
        default: return fail(DR_ERR_INVALID, "unknown array %d", which);
    }
    if (!src || !out || bytes > have) return fail(DR_ERR_INVALID, "array %d: %zu bytes asked, %zu available", which, bytes, have);
    HIPCHK(hipMemcpyAsync(out, src, bytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return DR_OK;
}

int dr_get_info(dr_context* c, dr_info* o) {
    CTX(c);
    if (!o) return fail(DR_ERR_INVALID, "out is null");
    if (c->profile) { HIPCHK(hipStreamSynchronize(c->stream)); drain_events(c); }
    memset(o, 0, sizeof *o);
    o->N = c->N; o->S = c->S; o->rank = c->rank; o->world = c->world; o->row0 = c->row0; o->nrows = c->nrows;
    o->rows_per_rank = c->rpr; o->n_bvh_nodes = c->n_nodes; o->ld_F = c->ldF; o->bytes_F = c->F_floats * sizeof(float);
    o->last_assemble_ms = c->last_assemble_ms; o->last_bvh_ms = c->last_bvh_ms; o->pairs_traced = c->pairs_traced;
    o->sweep_launches = c->sweep_launches; o->sweep_ms_total = c->sweep_ms_total;
    o->blocks_nonzero = c->mask_valid ? c->blocks_nonzero : 0; o->blocks_total = c->mask_valid ? c->blocks_total : 0;
    o->tree_used = c->tree_used; o->tree_on_host = c->tree_on_host; o->tree_depth = c->tree_depth; o->walk_used = c->walk_used;
    o->sweep_ksplit = c->have_solver ? c->ksplit : 0;
    return DR_OK;
}

int dr_profile_enable(dr_context* c, int on) {
    CTX(c);
    HIPCHK(hipStreamSynchronize(c->stream));
    drain_events(c);
    c->profile = on != 0;
    return DR_OK;
}

int dr_profile_reset(dr_context* c) {
    CTX(c);
    HIPCHK(hipStreamSynchronize(c->stream));
    drain_events(c);
    c->sweep_launches = 0;
    c->sweep_ms_total = 0;
    return DR_OK;
}

int dr_synchronize(dr_context* c) {
    CTX(c);
    HIPCHK(hipStreamSynchronize(c->stream));
    return DR_OK;
}

int dr_comm_info(dr_context* c, int* rccl_rank, int* rccl_world) {
    CTX(c);
    if (rccl_rank) *rccl_rank = c->comm.comm ? c->comm.rank : -1;
    if (rccl_world) *rccl_world = c->comm.comm ? c->comm.world : 0;
    return DR_OK;
}

/* ---- one process, several GPUs ------------------------------------------------------------------------- */
struct dr_group {
    std::vector<dr_context*> ctx;
    std::vector<int> devices;
    bool rccl = false;                  // a communicator per rank exists (distinct devices): ray counts always travel through it
    bool rccl_pass = false;             // ... and so does the residual after a pass; else peer copies, or
    bool inpass = false;                // the pass stores its chunk into every device's buffer itself (dr_options::group_exchange = INPASS)
    std::vector<hipEvent_t> ev_done;    // per rank: its pass has written its chunk (peer-copy exchange)
};

// how the residual travels after a pass: decided from the options at creation and again by dr_group_set_options
static int group_pick_exchange(dr_group* g, int ex) {
    const int n = (int)g->ctx.size();
    if (ex == DR_GROUP_EXCHANGE_RCCL && !g->rccl && n > 1)
        return fail(DR_ERR_INVALID, "group_exchange = RCCL needs distinct devices (a communicator per rank)");
    if (ex == DR_GROUP_EXCHANGE_INPASS && n > MAX_GROUP) return fail(DR_ERR_INVALID, "in-pass exchange supports up to %d devices", MAX_GROUP);
    if (ex == DR_GROUP_EXCHANGE_INPASS) {
        // ranks that share a device (a rehearsal) wait for each other ON the device: their streams must not share a hardware
        // queue, or a rank's gate kernel would sit in front of the very pass it waits for (it would time out, not hang) --
        // the runtime spreads a device's streams over 4 queues
        int most = 0;
        for (int d : g->devices) { int same = 0; for (int e : g->devices) same += e == d; most = std::max(most, same); }
        if (most > 4) return fail(DR_ERR_INVALID, "in-pass exchange: at most 4 ranks of a group may share one device (%d do)", most);
    }
    g->inpass = n > 1 && ex == DR_GROUP_EXCHANGE_INPASS;
    g->rccl_pass = g->rccl && !g->inpass && ex != DR_GROUP_EXCHANGE_P2P;
    for (dr_context* c : g->ctx) {
        c->manual_exchange = !g->rccl_pass;             // no collective inside sweep_once
        c->inpass_peers.clear();
        if (g->inpass) c->inpass_peers = g->ctx;
    }
    return DR_OK;
}

#define GRP(g) do { if (!(g) || (g)->ctx.empty()) return fail(DR_ERR_INVALID, "null group"); } while (0)

static int group_sync(dr_group* g) {
    for (dr_context* c : g->ctx) { int rc = dr_synchronize(c); if (rc) return rc; }
    if (g->inpass)
        for (dr_context* c : g->ctx) {
            if (!c->d_ctl) continue;
            int err = 0;
            HIPCHK(hipSetDevice(c->device));
            HIPCHK(hipMemcpy(&err, c->d_ctl + 3, sizeof(int), hipMemcpyDeviceToHost));
            if (err) return fail(DR_ERR_COMM, "in-pass exchange: rank %d waited in vain for a peer's pass (gate timed out)", c->rank);
        }
    return DR_OK;
}

int dr_group_destroy(dr_group* g) {
    if (!g) return DR_OK;
    for (dr_context* c : g->ctx) if (c) { hipSetDevice(c->device); hipStreamSynchronize(c->stream); }
    for (size_t r = 0; r < g->ev_done.size(); r++) { hipSetDevice(g->devices[r]); hipEventDestroy(g->ev_done[r]); }
    for (dr_context* c : g->ctx) dr_context_destroy(c);
    delete g;
    return DR_OK;
}

int dr_group_create(const int* device_ids, int n, dr_group** out) {
    if (!out || !device_ids || n < 1) return fail(DR_ERR_INVALID, "need at least one device");
    dr_group* g = new dr_group();
    bool distinct = true;
    for (int r = 0; r < n; r++) {
        for (int q = 0; q < r; q++) distinct = distinct && device_ids[q] != device_ids[r];
        dr_context* c = nullptr;
        int rc = dr_context_create(device_ids[r], &c);
        if (rc) { dr_group_destroy(g); return rc; }
        g->ctx.push_back(c);
        g->devices.push_back(device_ids[r]);
        c->rank = r; c->world = n;
    }
    // Distinct devices: RCCL (ncclCommInitAll), as between processes.  The same device several times (a one-GPU
    // rehearsal of the group) cannot be an RCCL communicator: peer copies then -- also selectable with the options.
    g->rccl = n > 1 && distinct;
    if (g->rccl) {
        std::vector<Comm> cs;
        std::string e = comm_init_all(cs, device_ids, n);
        if (!e.empty()) { dr_group_destroy(g); return fail(DR_ERR_COMM, "%s", e.c_str()); }
        for (int r = 0; r < n; r++) {
            g->ctx[r]->comm = cs[r];
            hipSetDevice(device_ids[r]);
            if (hipMalloc(&g->ctx[r]->d_agree, sizeof(int) * (size_t)n) != hipSuccess) { dr_group_destroy(g); return fail(DR_ERR_NOMEM, "hipMalloc"); }
        }
    }
    for (int r = 0; r < n; r++) {
        hipSetDevice(device_ids[r]);
        hipEvent_t ev;
        hipError_t he = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
        if (he != hipSuccess) { dr_group_destroy(g); return fail(DR_ERR_DEVICE, "hipEventCreate: %s", hipGetErrorString(he)); }
        g->ev_done.push_back(ev);
        for (int q = 0; q < n; q++)        // direct xGMI copies / stores where the devices allow it (an error here only means staged copies)
            if (device_ids[q] != device_ids[r]) { (void)hipDeviceEnablePeerAccess(device_ids[q], 0); (void)hipGetLastError(); }
    }
    { int rc = group_pick_exchange(g, g->ctx[0]->opt.group_exchange); if (rc) { dr_group_destroy(g); return rc; } }
    *out = g;
    return DR_OK;
}

int dr_group_info(dr_group* g, int* n, int* uses_rccl) {
    GRP(g);
    if (n) *n = (int)g->ctx.size();
    if (uses_rccl) *uses_rccl = g->rccl_pass ? 1 : 0;
    return DR_OK;
}

int dr_group_set_options(dr_group* g, const dr_options* o) {
    GRP(g);
    int rc = check_options(o);
    if (rc) return rc;
    rc = group_sync(g);
    if (rc) return rc;
    for (dr_context* c : g->ctx) { rc = dr_set_options(c, o); if (rc) return rc; }
    return group_pick_exchange(g, o->group_exchange);
}

int dr_group_context(dr_group* g, int rank, dr_context** out) {
    GRP(g);
    if (!out || rank < 0 || rank >= (int)g->ctx.size()) return fail(DR_ERR_INVALID, "rank %d outside the group", rank);
    *out = g->ctx[rank];
    return DR_OK;
}

int dr_group_set_mesh(dr_group* g, const float* vertices, int V, const float* normals, int Nn, const int32_t* tv, const int32_t* tn, int N) {
    GRP(g);
    SahTopology topo;                    // the same mesh on every device: the host's tree topology is built once
    int rc = DR_OK;
    for (dr_context* c : g->ctx) {
        c->shared_sah = &topo;
        rc = dr_scene_set_mesh(c, vertices, V, normals, Nn, tv, tn, N);
        c->shared_sah = nullptr;
        if (rc) break;
    }
    return rc;
}

int dr_group_assemble(dr_group* g, const float* uv, int K, float eps, int rule, int keep_vis) {
    GRP(g);
    const int n = (int)g->ctx.size();
    bool exchange = n > 1 && g->ctx[0]->opt.vis_exchange != 1;
    if (exchange) {
        dr_context* c0 = g->ctx[0];
        HIPCHK(hipSetDevice(c0->device));
        hipDeviceProp_t prop;
        HIPCHK(hipGetDeviceProperties(&prop, c0->device));
        // several ranks on ONE card (rehearsal) share its memory
        size_t per_rank = prop.totalGlobalMem;
        if (!g->rccl) { int same = 0; for (int d : g->devices) same += d == g->devices[0]; per_rank /= (size_t)std::max(same, 1); }
        if (!dr_vis_exchange_fits(c0->N, n, keep_vis, per_rank)) exchange = false;
    }
    const int mode1 = exchange ? 1 : 0;
    for (dr_context* c : g->ctx) { int rc = assemble_prepare(c, uv, K, eps, rule, keep_vis, 1, mode1); if (rc) return rc; }
    for (dr_context* c : g->ctx) { int rc = assemble_launch(c, K, eps, rule, 1, mode1); if (rc) return rc; }     // all devices at once
    for (dr_context* c : g->ctx) { int rc = assemble_complete(c, mode1); if (rc) return rc; }
    if (!exchange) return DR_OK;
    if (g->rccl) {
        std::string e = comm_group_start();
        for (int r = 0; r < n && e.empty(); r++) {
            dr_context* c = g->ctx[r];
            HIPCHK(hipSetDevice(c->device));
            e = comm_alltoall_bytes(c->comm, c->d_vsend, c->d_vrecv, c->vx_block, c->stream);
        }
        std::string e2 = comm_group_end();
        if (e.empty()) e = e2;
        if (!e.empty()) return fail(DR_ERR_COMM, "%s", e.c_str());
    } else {
        for (int d = 0; d < n; d++) {
            dr_context* cd = g->ctx[d];
            HIPCHK(hipSetDevice(cd->device));
            for (int r = 0; r < n; r++) {
                if (r == d) continue;
                dr_context* cr = g->ctx[r];
                HIPCHK(hipMemcpyPeerAsync(cd->d_vrecv + (size_t)r * cd->vx_block, cd->device, cr->d_vsend + (size_t)d * cr->vx_block, cr->device,
                                          cd->vx_block, cd->stream));
            }
        }
    }
    for (dr_context* c : g->ctx) { int rc = assemble_prepare(c, nullptr, c->split_K, c->split_eps, c->split_rule, 0, 1, 2); if (rc) return rc; }
    for (dr_context* c : g->ctx) { int rc = assemble_launch(c, c->split_K, c->split_eps, c->split_rule, 1, 2); if (rc) return rc; }
    for (dr_context* c : g->ctx) { int rc = assemble_complete(c, 2); if (rc) return rc; }
    return DR_OK;
}

int dr_group_solver_init(dr_group* g, int S, const float* E, const float* M, int n_mat, const int32_t* mat_of_patch) {
    GRP(g);
    for (dr_context* c : g->ctx) { int rc = dr_solver_init(c, S, E, M, n_mat, mat_of_patch); if (rc) return rc; }
    return DR_OK;
}

int dr_group_solver_reset(dr_group* g) {
    GRP(g);
    for (dr_context* c : g->ctx) { int rc = dr_solver_reset(c); if (rc) return rc; }
    return DR_OK;
}

// one pass on every device, then the exchange of the new residual chunks; nothing waits on the host
static int group_pass(dr_group* g, int conv_mode, float thr) {
    const int n = (int)g->ctx.size();
    if (g->inpass) {
        // every pass stores its chunk into all devices' buffers and waits (on the device) for the others' previous pass:
        // one gate + one pass kernel per rank and pass, nothing else
        for (int r = 0; r < n; r++) { int rc = sweep_once(g->ctx[r], conv_mode, thr); if (rc) return rc; }
        return DR_OK;
    }
    if (g->rccl_pass) {
        std::string e = comm_group_start();
        if (!e.empty()) return fail(DR_ERR_COMM, "%s", e.c_str());
        int rc = DR_OK;
        for (int r = 0; r < n && rc == DR_OK; r++) rc = sweep_once(g->ctx[r], conv_mode, thr);     // kernel + this rank's part of the all-gather
        e = comm_group_end();
        if (rc) return rc;
        if (!e.empty()) return fail(DR_ERR_COMM, "%s", e.c_str());
        return DR_OK;
    }
    for (int r = 0; r < n; r++) {
        dr_context* c = g->ctx[r];
        int rc = sweep_once(c, conv_mode, thr);          // flips c->cur: the new residual is d_R[c->cur]
        if (rc) return rc;
        HIPCHK(hipSetDevice(c->device));
        HIPCHK(hipEventRecord(g->ev_done[r], c->stream));
    }
    if (n == 1) return DR_OK;
    for (int d = 0; d < n; d++) {
        dr_context* cd = g->ctx[d];
        HIPCHK(hipSetDevice(cd->device));
        for (int r = 0; r < n; r++) {
            if (r == d) continue;
            dr_context* cr = g->ctx[r];
            HIPCHK(hipStreamWaitEvent(cd->stream, g->ev_done[r], 0));
            HIPCHK(hipMemcpyPeerAsync(cd->d_R[cd->cur] + (size_t)r * cd->cstride, cd->device, cr->d_R[cr->cur] + (size_t)r * cr->cstride, cr->device,
                                      sizeof(float) * cd->cstride, cd->stream));
        }
    }
    return DR_OK;
}

int dr_group_solver_step(dr_group* g, int n_passes, float* residual_sum_out) {
    GRP(g);
    if (n_passes < 0) return fail(DR_ERR_INVALID, "n_passes < 0");
    for (dr_context* c : g->ctx) {
        if (!c->have_solver) return fail(DR_ERR_STATE, "dr_group_solver_init has not been called");
        if (!c->have_F) return fail(DR_ERR_STATE, "form factors not assembled");
    }
    for (int k = 0; k < n_passes; k++) { int rc = group_pass(g, 0, 0.0f); if (rc) return rc; }
    int rc = group_sync(g);
    if (rc) return rc;
    if (residual_sum_out) {
        double sums[DR_MAX_BINS];
        rc = read_sums(g->ctx[0], sums);
        if (rc) return rc;
        double t = 0;
        for (int s = 0; s < g->ctx[0]->S; s++) t += sums[s];
        *residual_sum_out = (float)t;
    }
    return DR_OK;
}

int dr_group_solver_converge(dr_group* g, float threshold, int per_bin, int max_iters, int* iters_out) {
    GRP(g);
    if (max_iters < 0) return fail(DR_ERR_INVALID, "max_iters < 0");
    for (dr_context* c : g->ctx) {
        if (!c->have_solver) return fail(DR_ERR_STATE, "dr_group_solver_init has not been called");
        if (!c->have_F) return fail(DR_ERR_STATE, "form factors not assembled");
        HIPCHK(hipSetDevice(c->device));
        HIPCHK(hipMemsetAsync(c->d_ctl, 0, sizeof(int) * 4, c->stream));
    }
    dr_context* c0 = g->ctx[0];
    std::vector<int> cur0;
    for (dr_context* c : g->ctx) cur0.push_back(c->cur);
    int queued = 0, ctl[2] = { 0, 0 };
    const int mode = per_bin ? 2 : 1;
    for (;;) {
        const int batch = std::min(c0->check_every, max_iters - queued);
        for (int k = 0; k < batch; k++) { int rc = group_pass(g, mode, threshold); if (rc) return rc; }
        queued += batch;
        if (batch == 0) break;
        int rc = group_sync(g);
        if (rc) return rc;
        HIPCHK(hipSetDevice(c0->device));
        HIPCHK(hipMemcpy(ctl, c0->d_ctl, sizeof ctl, hipMemcpyDeviceToHost));      // the same on every device
        if (ctl[1] != 0 || ctl[0] < queued || queued >= max_iters) break;
    }
    for (size_t r = 0; r < g->ctx.size(); r++) g->ctx[r]->cur = cur0[r] ^ (ctl[0] & 1);
    if (iters_out) *iters_out = ctl[0];
    return DR_OK;
}

int dr_group_solver_read(dr_group* g, float* B, float* R) {
    GRP(g);
    for (size_t r = 0; r < g->ctx.size(); r++) {       // every rank fills its own rows of B; the residual is the same everywhere
        int rc = dr_solver_read(g->ctx[r], B, r == 0 ? R : nullptr);
        if (rc) return rc;
    }
    return DR_OK;
}

int dr_group_synchronize(dr_group* g) {
    GRP(g);
    return group_sync(g);
}

}  // extern "C"
