// dr_internal.h -- shared between the C-ABI implementation and the HIP kernels.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

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

constexpr int LEAF_MAX = 2;     // subtrees of up to this many triangles are collapsed into one leaf
                                // (a leaf is fetched whole: LEAF_MAX x 16 SGPRs)

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
    const float* uv;          // K x 2
    unsigned long long* pairs_traced;   // [0] pairs traced, [1] BVH nodes visited, [2] leaves tested (wave level)
    int stats;                          // count [1],[2] too (debug; costs two atomics per pair)
    int dbg_ray;
    int dbg_lo, dbg_hi;                 // STATS build: [3] = final live-ray mask of this pair, [1] = mask after the target test
    // Ray-count exchange between ranks (multi-rank assembly without tracing a pair twice).  A tile pair {o, t} with
    // o in this rank's rows and t in rank B's is traced by exactly one of the two: by the lower rank when o + t is even,
    // by the higher rank when it is odd.  vx_mode 0: no exchange, every rank traces all pairs that touch its rows.
    // vx_mode 1 (first launch): own x own pairs as always; foreign pairs only when they are this rank's, and their
    // 64 x 64 ray counts also go to slot (o, t) of `vex`.  vx_mode 2 (second launch, after the slots of all ranks
    // were gathered): the foreign pairs the other side traced: counts from slot (t, o), F tile written from them.
    int vx_mode, vx_rank, vx_tiles_per_rank;
    unsigned char* vex;                 // [world * tiles_per_rank][nT][64*64] ray counts, row = tile of the min index
};

// Which rank traces the tile pair {a, b} whose tiles lie in two different ranks' rows (tiles_per_rank tiles of 64 rows
// each per rank): the rank of the lower tile when a + b is even, the rank of the higher tile when it is odd -- every
// rank gets half of the pairs it shares with each other rank.  One definition for the kernel and the host.
__host__ __device__ inline int vx_tracer_rank(int a, int b, int tiles_per_rank) {
    const int ra = a / tiles_per_rank, rb = b / tiles_per_rank;
    const int lo = ra < rb ? ra : rb, hi = ra < rb ? rb : ra;
    return (((a + b) & 1) == 0) ? lo : hi;
}

struct SweepParams {
    int N;            // patches (columns with data)
    int S;
    int rpr;          // rows per rank (column-chunk size of the gathered residual)
    int world;
    int row0, nrows;  // this rank's rows
    size_t ldF;
    const float* F;
    const float* Rin;     // [world][S][rpr]
    float* Rout;          // [world][S][rpr]; this rank writes chunk `rank`
    int rank;
    float* B;             // [S][rpr] local
    const float* M;       // [n_mat][S][S]
    const int* mat;       // [rpr] local material index
    int n_mat;
    int skew;         // per-block start-tile multiplier (0 = every block starts at column 0)
    int ksplit;       // column splits (1 = fused epilogue; >1 = partial sums + k_sweep_epilogue)
    float* Gpart;     // [ksplit][nrows][S] partial F*R sums when ksplit > 1
    // optional: one bit per block of 32 rows x 256 columns of F, set where the block holds a non-zero; blocks whose
    // bit is clear are not read (null = read everything).  mask_words = 32-bit words per row block.
    const unsigned* tile_mask;
    int mask_words;
};

// launchers implemented in the .hip files
hipError_t launch_patch_records(hipStream_t st, int N, const float* vtx, const float* nrm,
                                const int* tv, const int* tn, float box_pad, PatchRec* patch, TriRec* tri);
hipError_t build_lbvh(hipStream_t st, int N, const TriRec* tri, const float scene_lo[3],
                      const float scene_hi[3], float node_pad, BvhNode* nodes /* room for 2N */,
                      TriRec* tri_sorted /* N + LEAF_MAX */, int* n_nodes_out);
hipError_t launch_ff_tiles(hipStream_t st, const TileParams& p);
hipError_t launch_sweep(hipStream_t st, const SweepParams& p);
int sweep_ksplit(int nrows, int S, int total_cols);
hipError_t launch_patch_colors(hipStream_t st, const float* B, int nrows, int rpr, int S, int mode, const float* xyz,
                               float* rgb);
hipError_t launch_vertex_colors(hipStream_t st, const float* rgb, int V, const int* off, const int* adj, float* out);
// builds SweepParams::tile_mask from the resident F shard (reads it once)
hipError_t launch_tile_mask(hipStream_t st, const float* F, int nrows, size_t ldF, unsigned* mask, int mask_words);
hipError_t launch_colsums(hipStream_t st, const float* R, int world, int S, int rpr, double* sums);
// layout conversion between the ABI's patch-major N x S and the device's bin-major chunks
hipError_t launch_scatter_rows(hipStream_t st, const float* src_NxS, int N, int S, int rpr, int world,
                               float* dst_chunks /* [world][S][rpr] */);
hipError_t launch_gather_rows(hipStream_t st, const float* src_chunks, int N, int S, int rpr, int world,
                              float* dst_NxS);

}  // namespace dr
