/*
 * daisyriot_hip.h -- C ABI of libdaisyriot_hip.so: DaisyRiot's radiosity hot
 * path (form-factor assembly + per-bin light-pass iteration) on one MI355X.
 *
 * One context drives one GPU.  Row-sharded multi-GPU runs are either one
 * process per GPU -- one context per rank, all with the same mesh,
 * dr_set_shard + dr_comm_init -- or one process for all GPUs: a dr_group
 * (dr_group_create(device_ids, n, ..)), whose calls run on all its devices at
 * once.  The only data-path collective of a pass is the all-gather of the
 * residual vector (with each rank's convergence sums riding in the same
 * message); the assembly adds one all-to-all of ray counts.
 *
 * Each entry point names the reference interface it replaces ("vs/" =
 * "visual studio/" in asylunatic/DaisyRiot).  No C++ or torch types cross
 * this boundary: plain pointers and sizes.  Host pointers are read/written
 * during the call only and never retained.  All calls are synchronous at the
 * ABI unless stated; a context is not thread-safe.
 *
 * Every function returns DR_OK (0) or a negative dr_status; the message of
 * the last failure on the calling thread is dr_last_error().
 */
#ifndef DAISYRIOT_HIP_H
#define DAISYRIOT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct dr_context dr_context;

typedef enum {
    DR_OK = 0,
    DR_ERR_INVALID = -1,   /* bad argument / index out of range / wrong call order */
    DR_ERR_DEVICE = -2,    /* a HIP call failed (reference: cudaCheckError prints and continues, vs/parallellism.cuh:21-26) */
    DR_ERR_NOMEM = -3,     /* device allocation failed (e.g. N*N*4 does not fit) */
    DR_ERR_COMM = -4,      /* RCCL missing or a collective failed */
    DR_ERR_STATE = -5      /* required earlier step missing (mesh / F / solver) */
} dr_status;

/* which rule produces the reverse entry F[col][row] of a traced pair row<col */
typedef enum {
    /* vs/OptixPrimeFunctionality.cpp:6-34 + 169-218 (ini cuda_on = true): both
     * directions from the integrand, one shared visibility fraction */
    DR_RULE_INTEGRAND = 0,
    /* vs/OptixPrimeFunctionality.cpp:311-366 (cuda_on = false): reverse entry by
     * reciprocity A_row*F/A_col */
    DR_RULE_RECIPROCITY = 1
} dr_rule;

#define DR_MAX_BINS 16            /* S, spectral bins per patch */
#define DR_RAYS_PER_PATCH 50      /* vs/Defines.h:25 */
#define DR_ORIGIN_EPS 0.000001f   /* vs/OptixPrimeFunctionality.cpp:194 */

const char* dr_last_error(void);

/* ---- context ------------------------------------------------------------------ */
/* Replaces the OptiX Prime context + CUDA runtime set-up
 * (vs/OptixPrimeFunctionality.cpp:36-37).  device_id = HIP ordinal. */
int dr_context_create(int device_id, dr_context** out);
int dr_context_destroy(dr_context* ctx);

/* Use the caller's hipStream_t for every launch and copy (NULL = the
 * context's own stream).  Lets a host that already owns streams
 * (torch.cuda.current_stream().cuda_stream) order its work with ours. */
int dr_set_stream(dr_context* ctx, void* hip_stream);

/* Row sharding: this context owns rows [rank*rpr, min(N,(rank+1)*rpr)) of F,
 * rpr = ceil(N/world) rounded up to a multiple of 256.  Call before
 * dr_formfactors_assemble.  Default rank 0 / world 1 (all rows). */
int dr_set_shard(dr_context* ctx, int rank, int world);
int dr_get_shard(dr_context* ctx, int* row0, int* nrows, int* rows_per_rank);
/* The same arithmetic without a context (pure host code, no GPU needed): which rows
 * rank `rank` of `world` owns for N patches, and where element (patch i, bin s) of the
 * residual lives in the gathered device buffer [world][S][rows_per_rank]. */
int dr_shard_rows(int N, int rank, int world, int* row0, int* nrows, int* rows_per_rank);
/* A gathered residual buffer is [world][dr_residual_chunk_floats]: per rank S*rows_per_rank values, bin-major,
 * followed by the chunk's per-bin sums (DR_MAX_BINS doubles = 2*DR_MAX_BINS floats; check_convergence,
 * vs/Lightning.h:255-261) -- the convergence scalars travel in the same all-gather as the residual. */
size_t dr_residual_offset(int i, int s, int S, int rows_per_rank);
size_t dr_residual_chunk_floats(int S, int rows_per_rank);

/* ---- options -------------------------------------------------------------------- */
/* Every choice the library makes between equivalent ways of doing the same thing -- which tree, which walk, how a light
 * pass is cut up, how a group exchanges -- per CONTEXT, settable and readable: none of them changes a result bit (tests
 * assert that), they change time.  dr_options_defaults fills the struct with the built-in defaults overridden by the
 * environment variables named below (read there and nowhere else: dr_context_create / dr_group_create call it once);
 * dr_set_options replaces a context's set (the tree options take effect at the next dr_scene_set_mesh, the pass options
 * at the next dr_solver_init), dr_get_options reads it back, dr_get_info reports what was actually used. */
enum { DR_TREE_AUTO = 0, DR_TREE_LBVH = 1, DR_TREE_SAH = 2 };
enum { DR_WALK_AUTO = 0, DR_WALK_THREADED = 1, DR_WALK_PAIRS = 2, DR_WALK_PATHS = 3 };
enum { DR_GROUP_EXCHANGE_AUTO = 0, DR_GROUP_EXCHANGE_P2P = 1, DR_GROUP_EXCHANGE_RCCL = 2, DR_GROUP_EXCHANGE_INPASS = 3 };
typedef struct {
    int32_t size;               /* sizeof(dr_options) of the caller's header (set by dr_options_defaults; checked by dr_set_options) */
    /* the tree (dr_scene_set_mesh) */
    int32_t tree;               /* DR_BVH=lbvh|sah        AUTO: Morton tree below 6 144 patches, SAH topology from there up */
    int32_t sah_on_host;        /* DR_SAH_HOST=1          1: the SAH topology from the host's reference builder (threads) instead of the device's */
    int32_t morton_key;         /* DR_BVH_KEY=0|1|2       Morton key variant of the LBVH (geom_kernels.hip, k_morton) */
    int32_t sah_bins;           /* DR_SAH_BINS            2..128 bins per axis (32) */
    float   sah_dilate;         /* DR_SAH_DILATE          box growth of the SAH cost in mean patch diagonals (0.5) */
    int32_t sah_host_threads;   /* DR_SAH_THREADS         threads of the host builder, 0 = up to 8 */
    /* the assembly (dr_formfactors_assemble) */
    int32_t walk;               /* DR_WALK=threaded|pairs|paths   AUTO: sibling-pair records, threaded tree when deeper than the walk's stack */
    int32_t octant_test;        /* DR_OCTANT=0            0: the general node test for every pair (default 1: sign-specialised when a wave's rays share an octant) */
    int32_t vis_exchange;       /* DR_NO_VIS_EXCHANGE=1 -> 1 never; DR_VIS_EXCHANGE_REHEARSE=1 -> 2 also with a one-rank communicator; 0 auto */
    int32_t tile_stats;         /* DR_TILE_STATS=1        counted (slow) instantiation of the tile kernel; visits per pair on stderr */
    int32_t debug_pair_lo, debug_pair_hi, debug_ray;   /* DR_DEBUG_PAIR=lo,hi,ray    (-1: none) */
    /* the light pass (dr_solver_init) */
    int32_t sweep_ksplit;       /* DR_SWEEP_KSPLIT        column ranges per row block, 0 = by the shard's size */
    int32_t sweep_taper;        /* DR_SWEEP_TAPER         -1 auto */
    int32_t sweep_rows_per_wave;/* DR_SWEEP_RR            0 auto (8; 4 above 8 bins) */
    int32_t sweep_skew;         /* DR_SWEEP_SKEW          -1 auto */
    int32_t sweep_mfma;         /* DR_SWEEP_MFMA=0        0: the VALU kernel above 8 bins too (default 1) */
    int32_t sweep_fenced;       /* DR_SWEEP_FENCED=1      1: the blocks of a pass hand over through agent-scope release / acquire fences
                                   instead of write-through stores + ticket (the memory-model form; measured 4 % slower) */
    int32_t no_comm;            /* DR_NO_COMM=1           a world > 1 shard timed on its own: no exchange after a pass */
    int32_t debug_converge;     /* DR_DEBUG_CONV=1 */
    /* dr_group */
    int32_t group_exchange;     /* DR_GROUP_EXCHANGE=p2p|rccl|inpass */
    int32_t fault_assemble_rank;/* DR_FAULT_ASSEMBLE_RANK tests: the rank whose first assembly launch "fails" (-1: none) */
} dr_options;
int dr_options_defaults(dr_options* out);
int dr_set_options(dr_context* ctx, const dr_options* opt);
int dr_get_options(dr_context* ctx, dr_options* out);

/* ---- scene --------------------------------------------------------------------- */
/* Exactly the MeshS / SimpleMesh arrays (vs/MeshS.h:14-20, vs/Defines.h:14-23;
 * handed over today at vs/OptixPrimeFunctionality.cpp:13 and :38-44):
 * vertices 3*V, normals 3*Nn, per-triangle vertex and normal indices 3*N,
 * 0-based.  Builds the per-patch records and the BVH (replaces rtpModelUpdate,
 * vs/OptixPrimeFunctionality.cpp:43-47) on the device: a Morton tree, or from
 * 6 144 patches up a binned-SAH topology (dr_options::tree overrides; the
 * results do not depend on the tree). */
int dr_scene_set_mesh(dr_context* ctx, const float* vertices, int V,
                      const float* normals, int Nn,
                      const int32_t* tri_vertex_idx,
                      const int32_t* tri_normal_idx, int N);

/* ---- form factors ---------------------------------------------------------------- */
/* Replaces OptixPrimeFunctionality::cudaCalculateRadiosityMatrix /
 * calculateRadiosityMatrix (vs/OptixPrimeFunctionality.cpp:6, :311) together
 * with parallellism::runCalculateRadiosityMatrix (vs/parallellism.cu:4) and
 * calculateAllVisibility (:169).  uv = K (u,v) samples (the reference's
 * `rands`, :54-63; K = RAYS_PER_PATCH), 1 <= K <= 254.  The dense fp32 matrix
 * stays resident on the device (this rank's rows).  keep_visibility != 0 also
 * keeps the per-pair ray counts (tests; N*nrows bytes). */
int dr_formfactors_assemble(dr_context* ctx, const float* uv, int K,
                            float origin_eps, int rule, int keep_visibility);

/* Allocates this rank's shard of F now (dr_formfactors_assemble / _load_rows do it on first use).  For hosts that want the
 * allocation out of a timed region: a first hipMalloc of 17 GB takes 0.5 - 1 s of driver time on some machines. */
int dr_formfactors_reserve(dr_context* ctx);

/* Multi-rank assembly (no reference counterpart: the reference is single-GPU).  A pair of patches in
 * two ranks' rows is needed by both (F[i][j] and F[j][i] share the ray count); with an RCCL
 * communicator dr_formfactors_assemble traces it on one of the two ranks only and sends the
 * 64 x 64-byte ray-count slots of the tile pairs to the other (one all-to-all: a rank receives
 * (world-1) blocks of (rows_per_rank/64)^2 slots, about N*N/world bytes, not N*N).  Without one
 * (DR_NO_COMM timing runs, a single rank of a shard on its own) every rank traces all pairs that touch
 * its rows.  A host that moves buffers itself does the same in steps: ..._split on every rank, then
 * for every ordered pair of ranks (a, b) the block a exports for b (dr_vis_exchange_bytes bytes)
 * imported into b as coming from a, then ..._finish on every rank. */
/* pure host arithmetic: the rank that traces the pair of patches (a, b) in a `world`-way assembly (-1: bad argument) */
int dr_vis_exchange_tracer(int N, int world, int patch_a, int patch_b);
/* pure host arithmetic: does a world-way assembly take the exchange path on cards of device_bytes (slot buffers +
 * F shard + optional ray counts within 85 % of it)?  Deliberately no rank argument: every rank must decide alike. */
int dr_vis_exchange_fits(int N, int world, int keep_visibility, size_t device_bytes);
int dr_formfactors_assemble_split(dr_context* ctx, const float* uv, int K,
                                  float origin_eps, int rule, int keep_visibility);
int dr_vis_exchange_bytes(dr_context* ctx, size_t* block_bytes);
int dr_vis_exchange_export(dr_context* ctx, int dst_rank, void* block_out, size_t bytes);
int dr_vis_exchange_import(dr_context* ctx, int src_rank, const void* block_in, size_t bytes);
int dr_formfactors_assemble_finish(dr_context* ctx);

/* Unoccluded integrand only (parallellism::calculateRow, vs/parallellism.cu:91-111):
 * F = stored integrand, no rays.  For tests and timing of the integrand alone. */
int dr_formfactors_integrand_only(dr_context* ctx);

/* Copy rows [row0,row0+nrows) (must be owned) to out[nrows*N] row-major.
 * out[r*N+c] = F(row0+r -> c), the RadMat(i,j) of vs/Lightning.h:19. */
int dr_formfactors_read_rows(dr_context* ctx, int row0, int nrows, float* out);
/* ray counts 0..K per entry, 255 = pair not traced; needs keep_visibility */
int dr_visibility_read_rows(dr_context* ctx, int row0, int nrows, uint8_t* out);
/* Upload externally produced rows (the DeserializeMat route, vs/Lightning.h:51-74,
 * and sweep-only tests).  Allocates F on first use. */
int dr_formfactors_load_rows(dr_context* ctx, int row0, int nrows, const float* F);

/* ---- solver ---------------------------------------------------------------------- */
/* Replaces the Lightning constructors' set-up (vs/Lightning.h:114-139, 317-330,
 * 393-404): S bins; E[N*S] patch-major emission (R0 = B0 = E, reset());
 * M[n_mat*S*S] row-major per-material bin-transfer matrices (spectral:
 * Material::M; RGB: diag(Kd); BW: S=1, M=[1]); mat_of_patch[N]. */
int dr_solver_init(dr_context* ctx, int S, const float* E, const float* M,
                   int n_mat, const int32_t* mat_of_patch);
/* n_passes of increment_lightpass (vs/Lightning.h:196-226, 342-349, 419-424):
 * R <- M_mat(i) * (F*R)[i], B += R; all S bins per pass, F streamed once.
 * residual_sum_out (nullable) = sum over patches and bins of R after the last pass. */
int dr_solver_step(dr_context* ctx, int n_passes, float* residual_sum_out);
/* converge_lightning (vs/Lightning.h:145-151: total sum > threshold;
 * per_bin != 0 = the RGB rule :336-340, any bin sum > threshold).  Stops after
 * max_iters passes at the latest (the reference has no cap and BW never
 * terminates in closed scenes).  The test runs on the device at the head of
 * every pass (from the sums the previous pass left in the residual's tails);
 * the host queues passes in batches and looks at the result once per batch
 * (dr_solver_set_check_interval, default 8): passes queued behind the
 * converged one do nothing, iters_out counts the real ones. */
int dr_solver_converge(dr_context* ctx, float threshold, int per_bin,
                       int max_iters, int* iters_out);
int dr_solver_set_check_interval(dr_context* ctx, int passes);
/* reset() (vs/Lightning.h:159-165): R = B = E */
int dr_solver_reset(dr_context* ctx);
/* B (lightningvalues) and R (residualvector), N*S patch-major, either nullable.
 * B holds this rank's rows only; rows of other ranks are left untouched. */
int dr_solver_read(dr_context* ctx, float* B, float* R);
int dr_solver_residual_sums(dr_context* ctx, double* sums /* S */);
/* Optional (default off): the light pass does not read blocks of 32 rows x 256 columns of F that are entirely zero
 * (patches that cannot see each other: same wall, back to back) -- what the reference's sparse RadMat
 * (vs/Lightning.h:19) does element-wise.  Bit-identical results (the skipped products are exact zeros); the block
 * map is built from the resident F on the next pass (one extra read of F). */
int dr_solver_skip_zero_blocks(dr_context* ctx, int enable);

/* ---- display colours (what the viewer shows; computed from the resident B) ---- */
#define DR_DISPLAY_BW       0   /* BWLightning::get_color_of_patch   (vs/Lightning.h:406-408): (B,B,B)        */
#define DR_DISPLAY_RGB      1   /* RGBLightning::get_color_of_patch  (vs/Lightning.h:332-334): (B0,B1,B2)     */
#define DR_DISPLAY_SPECTRAL 2   /* SpectralLightning::update_color_cache (vs/Lightning.h:168-183): XYZ fit ->
                                 * XYZToRGB (vs/color.h:48-52) -> divided by max(r,g,b) when that is > 1       */
/* Colours of this rank's rows into rgb (rows_of_this_rank*3, nullable: the colours also stay on
 * the device for dr_display_vertex_colors).  xyz_per_bin = S*3 floats, the caller's
 * xyz_per_wavelength (vs/Lightning.h:128-131); only read in spectral mode. */
int dr_display_patch_colors(dr_context* ctx, int mode, const float* xyz_per_bin, float* rgb);
/* Drawer::interpolate's corner values (vs/Drawer.cpp:161-186): per vertex the mean colour of the
 * patches around it, summed in the order given, over MeshS::trianglesPerVertex handed over as
 * CSR (vtx_off V+1 offsets into vtx_tri).  rgb_all = colours of all N patches (N*3), or NULL to
 * use what the last dr_display_patch_colors left on the device (single-rank contexts only). */
int dr_display_vertex_colors(dr_context* ctx, const float* rgb_all, const int32_t* vtx_off,
                             const int32_t* vtx_tri, int V, float* out /* V*3 */);

/* ---- multi-GPU exchange (no reference counterpart: the reference is single-GPU) ---- */
/* RCCL is bound at run time, privately (dlopen RTLD_LOCAL): a copy with the SONAME librccl.so.1 that the process has
 * already mapped, else the system's.  dr_comm_set_library names the file to bind instead (before the first dr_comm_* /
 * dr_group_create call; the environment variable DR_RCCL_LIB does the same): a host that will later load another copy --
 * a Python process that imports torch, which brings its own -- passes that copy's path so that one runtime serves both
 * (daisyriot_amd/api.py does).  dr_comm_library_info: "bound=<file>;mapped=<every RCCL file in the process>". */
int dr_comm_set_library(const char* path);
int dr_comm_library_info(char* out, size_t n);
/* 128-byte RCCL unique id made on rank 0 and handed to every rank by the host. */
int dr_comm_unique_id(void* out128);
int dr_comm_init(dr_context* ctx, const void* id128, int rank, int world);
/* what the context's RCCL communicator itself reports (ncclCommUserRank / ncclCommCount); -1 / 0 without one */
int dr_comm_info(dr_context* ctx, int* rccl_rank, int* rccl_world);
/* Host-staged exchange for hosts without RCCL (MPI staging, tests): after dr_comm_manual the
 * passes run without a collective and the host moves the residual chunks itself after EVERY
 * dr_solver_step(ctx, 1, ..): export this rank's new chunk (dr_residual_chunk_floats floats),
 * import every other rank's chunk. */
int dr_comm_manual(dr_context* ctx);
int dr_exchange_export(dr_context* ctx, float* chunk_out);
int dr_exchange_import(dr_context* ctx, int src_rank, const float* chunk_in, size_t n_floats);

/* ---- one process, several GPUs (the reference is one process: main.cpp:55-154, Lightning.h:446-457) ---- */
/* n contexts, rank r on device_ids[r], rows of F sharded over them; every call below runs on all devices at once
 * (asynchronous launches on one stream per device, one wait at the end).  Distinct devices exchange through RCCL
 * (ncclCommInitAll, grouped calls); the same device listed several times -- a rehearsal of the group on one GPU --
 * or dr_options::group_exchange = P2P uses peer copies (hipMemcpyPeerAsync) instead; INPASS: every pass stores its new
 * residual chunk straight into every device's buffer (peer-mapped), no exchange call at all. */
typedef struct dr_group dr_group;
int dr_group_create(const int* device_ids, int n_devices, dr_group** out);
int dr_group_destroy(dr_group* g);
int dr_group_info(dr_group* g, int* n_devices, int* uses_rccl);
/* rank r's context, owned by the group: for reads, dr_get_info, dr_profile_enable, display colours */
int dr_group_context(dr_group* g, int rank, dr_context** out);
int dr_group_set_mesh(dr_group* g, const float* vertices, int V, const float* normals, int Nn,
                      const int32_t* tri_vertex_idx, const int32_t* tri_normal_idx, int N);
int dr_group_assemble(dr_group* g, const float* uv, int K, float origin_eps, int rule, int keep_visibility);
int dr_group_solver_init(dr_group* g, int S, const float* E, const float* M, int n_mat, const int32_t* mat_of_patch);
int dr_group_solver_step(dr_group* g, int n_passes, float* residual_sum_out);
int dr_group_solver_converge(dr_group* g, float threshold, int per_bin, int max_iters, int* iters_out);
int dr_group_solver_reset(dr_group* g);
/* B: all N*S values (every rank writes its rows); R: the gathered residual */
int dr_group_solver_read(dr_group* g, float* B, float* R);
int dr_group_synchronize(dr_group* g);
/* the same options on every context of the group (group_exchange: takes effect at the next dr_group_solver_init) */
int dr_group_set_options(dr_group* g, const dr_options* opt);

/* ---- measurement ----------------------------------------------------------------- */
typedef struct {
    int    N, S, rank, world, row0, nrows, rows_per_rank, n_bvh_nodes;
    size_t ld_F;              /* leading dimension of F in floats */
    size_t bytes_F;           /* resident bytes of the F shard */
    double last_assemble_ms;  /* hipEvent time of the tile kernel(s) of the last assemble (the BVH build is last_bvh_ms) */
    double last_bvh_ms;       /* time of the BVH build in dr_scene_set_mesh between two stream events: the device kernels and,
                                 from 6 144 patches up, the host's SAH topology build in between */
    uint64_t pairs_traced;    /* unordered pairs traced by the last assemble */
    uint64_t sweep_launches;  /* profiled sweep launches since dr_profile_reset */
    double sweep_ms_total;    /* their summed hipEvent durations */
    uint64_t blocks_nonzero;  /* with dr_solver_skip_zero_blocks: 32 x 256 blocks of the F shard that hold a non-zero ... */
    uint64_t blocks_total;    /* ... of this many (0 until the first pass after enabling it) */
    int32_t tree_used;        /* DR_TREE_LBVH / DR_TREE_SAH: what the last dr_scene_set_mesh built */
    int32_t tree_on_host;     /* 1: its topology came from the host builder */
    int32_t tree_depth;       /* depth of the written tree */
    int32_t walk_used;        /* DR_WALK_*: what the last assembly's tile kernel walked */
    int32_t sweep_ksplit;     /* column ranges per row block of the light pass */
    int32_t reserved_;
} dr_info;
int dr_get_info(dr_context* ctx, dr_info* out);
/* record a hipEvent pair around every sweep kernel launch (on its own stream) */
int dr_profile_enable(dr_context* ctx, int on);
int dr_profile_reset(dr_context* ctx);
int dr_synchronize(dr_context* ctx);
/* tests: the threaded LBVH, n_bvh_nodes records of 32 bytes
 * {float lo[3], hi[3]; int32 skip; int32 patch (-1 = internal)} in pre-order */
int dr_debug_read_bvh(dr_context* ctx, void* out, int max_nodes);
/* tests: raw device arrays (0 TriRec[N] original order, 1 TriRec[N+2] Morton order,
 * 3 the uploaded (u,v) samples, 4 PatchRec[N]) */
int dr_debug_read_array(dr_context* ctx, int which, void* out, size_t bytes);
/* tests (host only, no device needed): the SAH tree topology of the host's reference builder (dr_options::sah_on_host)
 * for N boxes {lo[3], hi[3]}.  order[N]: box index at each position of the leaf order; internal nodes 0 .. N-2 (0 = root)
 * with children left[i] / right[i] (>= N-1: leaf at position id - (N-1)) covering positions first[i] .. last[i];
 * parent[2N-1] (-1 for the root). */
int dr_debug_sah_topology(int N, const float* boxes, int32_t* order, int32_t* left, int32_t* right,
                          int32_t* first, int32_t* last, int32_t* parent);
/* tests: the same arrays from the DEVICE builder (what dr_scene_set_mesh runs), on bare boxes, with the context's options */
int dr_debug_sah_topology_device(dr_context* ctx, int N, const float* boxes, int32_t* order, int32_t* left, int32_t* right,
                                 int32_t* first, int32_t* last, int32_t* parent);

#ifdef __cplusplus
}
#endif
#endif
