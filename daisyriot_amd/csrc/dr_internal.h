// dr_internal.h -- shared between the C-ABI implementation and the HIP kernels.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <vector>

namespace dr {

// Per-patch quantities of the form-factor integrand, computed once per patch instead of
// once per pair (vs/triangle_math.cpp:11-74 evaluated per patch).
struct PatchRec {
    float cen[4][3];   // centroids of the four midpoint sub-triangles
    float nrm[3];      // normalised average of the three OBJ vertex normals
    float sa[4];       // sub-triangle areas
    float area;        // patch area
};                      // 20 floats

// a, e1 = b-a, e2 = c-a : uv2xyz (vs/triangle_math.cpp:3-9) and the ray/triangle test;
// lo/hi: bounding box of (a, a+e1, a+e2) padded by 1e-4 of the scene extent -- a triangle can only be
// hit by a segment that passes this box (the gate that makes fp32 "hits" geometric, see oracle.c)
struct TriRec {
    float a[3], e1[3], e2[3];
    int   id;           // original patch index (the leaf arrays are in Morton order)
    float lo[3], hi[3];
};                      // 16 dwords = 64 B



// Threaded BVH node in depth-first pre-order: the first child of an internal node is
// node+1; `skip` is the BYTE offset (pre-order index * 32) of the first node after this node's subtree.
// The box is held as centre and half-extent, grown by the build's node_pad (see node_hit_mask).
// The array ends with one sentinel node (all-space box, tri = BVH_END) that every skip out of the tree
// lands on.
struct BvhNode {
    float c[3], h[3];
    int   skip;
    int   tri;          // leaf: first*8 + (count-1) [+4: both triangles have the same gate box] into the Morton-ordered TriRecs; -1: internal
};                      // 32 B: one s_load_dwordx8
constexpr int BVH_END = 0x7ffffff8;

// The same tree in SIBLING-PAIR form: one 64-byte record per interior node holding the boxes of its TWO CHILDREN (c[0] left,
// c[1] right; box fields as in BvhNode), so that one s_load_dwordx16 buys two node tests and a child that is rejected is never
// fetched.  A child's `skip` field is its ITEM: >= 0 the byte offset of the child's own record (an interior child), < 0 a leaf:
// 0x80000000 | leaf code (leaf code as BvhNode::tri).  Records lie in pre-order of the interior nodes, the root's first.  The
// walk keeps the items still to do on a stack of at most PAIR_STACK entries (the lanes of one VGPR): trees deeper than that
// are walked in the threaded form.
struct BvhPair { BvhNode c[2]; };       // 64 B: one s_load_dwordx16
constexpr int PAIR_STACK = 64;

// Per-patch PATH RECORDS (k_paths, stream_path_records in geom_kernels.hip).  Every ray of a pair (lo, hi) starts on patch lo
// and ends on patch hi, so it is inside every ancestor of the two patches' leaves: testing those nodes (a quarter of a walk
// from the root) tells nothing.  What a walk does need are the SIBLINGS hanging off the two root-to-leaf paths -- together with
// the two leaves they cover the whole tree.  They are the same for every pair the patch takes part in, so the build writes them
// out once per patch: PATH_RECS records in root-to-leaf order, record d-1 = the sibling of the path's node at depth d: its box
// as lower / upper corner (c[] = lo, h[] = hi, as in bvh_lh) and, in `skip`, its ITEM for the pair walk (BvhPair below).
// PathHdr: depth of the patch's leaf (-1: deeper than PATH_RECS, such patches are walked from the root), the leaf's item, the
// path's turns (bit d-1 set: right child at depth d).  The pair's walk streams the records (addresses known up front: no
// pointer chasing) and pushes the items of the siblings its rays touch onto the pair walk's stack.
constexpr int PATH_RECS = 40;
struct PathHdr { int depth; int leaf; unsigned turns_lo, turns_hi; };

constexpr int LEAF_MAX = 2;     // subtrees of up to this many triangles are collapsed into one leaf
                                // (a leaf is fetched whole: LEAF_MAX x 16 SGPRs)

constexpr int MAX_BINS = 16;    // = DR_MAX_BINS of the ABI
// Every rank's chunk of the residual buffers ends with MAX_BINS doubles: the per-bin sums of the chunk's residual
// (check_convergence, vs/Lightning.h:255-261), written by the pass that produced the chunk and carried to the other
// ranks by the same all-gather -- no extra kernel, collective or host round trip to test for convergence.
constexpr int RTAIL = 2 * MAX_BINS;   // floats
constexpr int TILE = 64;        // patch-pair tile edge of the assembly kernel
constexpr int SHARD_ALIGN = 256; // rows per rank are a multiple of this (sweep column tile)

struct TileParams {
    int N, K, rule, trace, nT, tile0, nOwnedTiles, row0, nrows, n_nodes;
    float eps;
    size_t ldF;
    float* F;                 // this rank's rows: F[(row-row0)*ldF + col]
    unsigned char* vis;       // nullable, vis[(row-row0)*N + col]
    const PatchRec* patch;
    const TriRec* tri;          // original order (ray generation)
    const TriRec* tri_sorted;   // Morton order, LEAF_MAX never-hit padding records at the end (leaf tests)
    const BvhNode* bvh;
    const BvhNode* bvh_lh;      // the same nodes with c[] = lower, h[] = upper corner (sign-specialised node test); null: not used
    const BvhPair* pairs;       // sibling-pair form of the tree (BvhPair), centre / half-extent; null: the threaded form is walked
    const BvhPair* pairs_lh;    // ... lower / upper corner
    const BvhNode* path_rec;    // [N][PATH_RECS] (+ 1 record the stream's prefetch may touch) path records (see PathHdr); null: every walk starts at the root
    const PathHdr* path_hdr;    // [N]
    const float* uv;          // K x 2
    unsigned long long* pairs_traced;   // [0] pairs traced, [1] BVH nodes visited, [2] leaves tested (wave level)
    int stats;                          // count [1],[2] too (debug; costs two atomics per pair)
    int dbg_ray;
    int dbg_lo, dbg_hi;                 // STATS build: [3] = final live-ray mask of this pair, [1] = mask after the target test
    // Ray-count exchange between ranks (multi-rank assembly without tracing a pair twice).  A tile pair {o, t} with
    // o in this rank's rows and t in rank B's is traced by exactly one of the two: by the lower rank when o + t is even,
    // by the higher rank when it is odd.  vx_mode 0: no exchange, every rank traces all pairs that touch its rows.
    // vx_mode 1 (first launch): own x own pairs as always; foreign pairs only when they are this rank's, and their
    // 64 x 64 ray counts also go to slot [B][o - own tile0][t - B's tile0] of `vsend` (one contiguous block per
    // destination rank).  Then block B of vsend travels to rank B's vrecv[this rank] (an all-to-all: a rank only
    // receives what it needs, (world-1) * tiles_per_rank^2 slots).  vx_mode 2 (second launch): the foreign pairs the
    // other side traced: counts from vrecv[B][t - B's tile0][o - own tile0], F tile written from them.
    int vx_mode, vx_rank, vx_tiles_per_rank;
    unsigned char* vsend;               // [world][tiles_per_rank][tiles_per_rank][64*64] ray counts, row = tile of the min index
    const unsigned char* vrecv;         // same shape; block B = what rank B traced for this rank
    float ts_max;             // cap of a ray's 1/length scale in the walk's node tests: 1e19 / max|coordinate| (keeps org * iv finite)
};

// Which rank traces the tile pair {a, b} whose tiles lie in two different ranks' rows (tiles_per_rank tiles of 64 rows
// each per rank): the rank of the lower tile when a + b is even, the rank of the higher tile when it is odd -- every
// rank gets half of the pairs it shares with each other rank.  One definition for the kernel and the host.
__host__ __device__ inline int vx_tracer_rank(int a, int b, int tiles_per_rank) {
    const int ra = a / tiles_per_rank, rb = b / tiles_per_rank;
    const int lo = ra < rb ? ra : rb, hi = ra < rb ? rb : ra;
    return (((a + b) & 1) == 0) ? lo : hi;
}

// how a light pass is cut up (dr_options::sweep_*; none of it changes a result bit)
struct SweepTuning {
    int rows_per_wave = 0;   // k_sweep: 8 (4 above 8 bins: the accumulators are RR*S registers); 4 forces 4 below too; 0 auto
    int mfma = 1;            // k_sweep_mfma above 8 bins
    int ksplit = 0;          // column ranges per row block, 0 = by the shard's size (sweep_ksplit)
    int skew = -1;           // per-block start-tile multiplier, -1 = by the residual's size
    int taper = -1;          // shares of the column ranges, -1 = n, n-1, .., 1
    int fenced = 0;          // hand-offs between the blocks of a pass through release / acquire fences instead of write-through stores
};

struct SweepParams {
    int N;            // patches (columns with data)
    int S;
    int rpr;          // rows per rank (column-chunk size of the gathered residual)
    int world;
    int row0, nrows;  // this rank's rows
    size_t ldF;
    const float* F;
    size_t cstride;       // floats per rank chunk of a residual buffer: S*rpr + RTAIL
    const float* Rin;     // [world][cstride]: [S][rpr] residual (bin-major) + the chunk's per-bin sums (MAX_BINS doubles)
    float* Rout;          // same layout; this rank writes chunk `rank` (residual and sums)
    int rank;
    float* B;             // [S][rpr] local
    const float* M;       // [n_mat][S][S]
    const int* mat;       // [rpr] local material index
    int n_mat;
    int skew;         // per-block start-tile multiplier (0 = every block starts at column 0)
    int taper;        // column ranges of decreasing size (the later a block is dispatched, the less it has to do): range y gets a share
                      // proportional to taper + gridDim.y - y; 0 = equal ranges
    int ksplit;       // column ranges (gridDim.y); > 1: partial sums to Gpart, the last range of a row block to finish adds them up
    float* Gpart;     // [ksplit][nrows][S] partial F*R sums when ksplit > 1
    // in-launch reductions: [0] counts the row blocks whose epilogue is done (the last one adds up blk_sums into the
    // chunk's tail), [1 + x] the column ranges of row block x that have written their partial sums
    unsigned* tickets;
    double* blk_sums;     // [row blocks][S]: sums of the new residual per row block
    // device-side convergence test (dr_solver_converge): 0 = none; 1 = go on while the sum over all bins > thr
    // (vs/Lightning.h:145-151); 2 = while any bin's sum > thr (:336-340).  A pass that finds the gathered residual
    // converged does nothing at all (no write, ctl[1] = 1), nor does any pass after it; every real pass counts itself in ctl[0].
    int conv_mode;
    float conv_thr;
    int* ctl;
    int want_sums;        // form the new residual's per-bin sums and leave them in the chunk's tail (always with conv_mode != 0)
    // optional: one bit per block of 32 rows x 256 columns of F, set where the block holds a non-zero; blocks whose
    // bit is clear are not read (null = read everything).  mask_words = 32-bit words per row block.
    const unsigned* tile_mask;
    int mask_words;
    SweepTuning tune;
    // in-pass exchange of a one-process group (dr_options::group_exchange = INPASS): the new residual chunk (and its tail) is
    // also stored into every other device's residual buffer, peer-mapped; peers[r] = device r's Rout (null: none / this rank)
    float* peers[16];
    int n_peers;
    // ... and when the whole pass is done (last row block, after a system-scope fence) its number `seq` goes into slot `rank` of
    // every device's sequence array peer_seq[r] -- the next pass on device r starts behind a one-thread gate kernel that waits
    // until all world slots of ITS array show the previous pass (launch_wait_peers): no collective, no host round trip
    unsigned* peer_seq[16];
    unsigned seq;
};

// launchers implemented in the .hip files
hipError_t launch_patch_records(hipStream_t st, int N, const float* vtx, const float* nrm,
                                const int* tv, const int* tn, float box_pad, PatchRec* patch, TriRec* tri);
// The host's SAH topology of one mesh (sah_hierarchy_host), kept by a group so that its contexts -- which all get the same
// mesh -- build it once instead of once per device.
struct SahTopology {
    int N = -1;
    std::vector<int> order, left, right, first, last, parent;
};
// what dr_scene_set_mesh builds (dr_options, resolved): sah = the binned-SAH topology instead of the Morton tree
struct TreeOptions {
    bool sah = false, sah_on_host = false;
    int morton_key = 0, sah_bins = 32, sah_host_threads = 0;
    float sah_dilate = 0.5f;
};
hipError_t build_lbvh(hipStream_t st, int N, const TriRec* tri, const float scene_lo[3],
                      const float scene_hi[3], float node_pad, BvhNode* nodes /* room for 2N */,
                      BvhNode* nodes_lh /* the same in lower / upper corner form */,
                      TriRec* tri_sorted /* N + LEAF_MAX */, int* n_nodes_out,
                      BvhNode* path_rec /* N * PATH_RECS */, PathHdr* path_hdr /* N */,
                      const TreeOptions& topt, SahTopology* shared = nullptr /* host SAH: N == this mesh's: use it; else fill it */,
                      BvhPair* pairs = nullptr /* room for max(N - 1, 1): the sibling-pair form, centre / half-extent */,
                      BvhPair* pairs_lh = nullptr /* lower / upper corner */, int* depth_out = nullptr /* depth of the written tree */);
void sah_topology_from_boxes(int N, const float* boxes /* N x {lo[3], hi[3]} */, SahTopology& out, const TreeOptions& topt);
hipError_t sah_topology_from_boxes_device(hipStream_t st, int N, const float* boxes, SahTopology& out, const TreeOptions& topt);
hipError_t launch_ff_tiles(hipStream_t st, const TileParams& p);
hipError_t launch_sweep(hipStream_t st, const SweepParams& p);
constexpr int MAX_GROUP = 16;   // devices of one in-pass-exchanging group (SweepParams::peers)
// the gate in front of a pass of an in-pass-exchanging group: returns when seq_local[0 .. world) >= want (or after ~5 s: err[0] = 1)
hipError_t launch_wait_peers(hipStream_t st, const unsigned* seq_local, int world, unsigned want, int* err);
int sweep_ksplit(int nrows, int S, int total_cols, const SweepTuning& t);
hipError_t launch_patch_colors(hipStream_t st, const float* B, int nrows, int rpr, int S, int mode, const float* xyz,
                               float* rgb);
hipError_t launch_vertex_colors(hipStream_t st, const float* rgb, int V, const int* off, const int* adj, float* out);
// builds SweepParams::tile_mask from the resident F shard (reads it once)
hipError_t launch_tile_mask(hipStream_t st, const float* F, int nrows, size_t ldF, unsigned* mask, int mask_words);
// per-bin sums of every chunk of R into the chunk's own tail (after a reset; passes maintain them themselves)
hipError_t launch_chunk_sums(hipStream_t st, float* R, int world, int S, int rpr, size_t cstride);
int sweep_row_blocks(int nrows, int S, const SweepTuning& t);
// layout conversion between the ABI's patch-major N x S and the device's bin-major chunks
hipError_t launch_scatter_rows(hipStream_t st, const float* src_NxS, int N, int S, int rpr, size_t cstride,
                               float* dst_chunks /* [world][cstride] */);
hipError_t launch_gather_rows(hipStream_t st, const float* src_chunks, int N, int S, int rpr, size_t cstride,
                              float* dst_NxS);

}  // namespace dr
