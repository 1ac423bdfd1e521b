// geom_kernels.hip -- per-patch records, the BVH build (Morton tree on the device, or SAH topology from the
// host with bounds and layout on the device), and the fused form-factor tile
// kernel (integrand + visibility + both F tiles written once) for gfx950.
//
// Compiled with -ffp-contract=off: every fp32 operation below is individually rounded, in
// the written order, so that geometry decisions (which pairs are traced, which rays are
// blocked) are bit-identical to the CPU oracle; divisions and square roots are the
// correctly rounded ones (-fhip-fp32-correctly-rounded-divide-sqrt, hipcc's default).
//
// Reference behaviour restated here ("vs/" = "visual studio/"):
//   integrand    vs/triangle_math.cpp:11-74, vs/OptixPrimeFunctionality.cpp:133-161,
//                vs/parallellism.cu:91-151
//   visibility   vs/OptixPrimeFunctionality.cpp:54-63, 169-218, 244-271 (closest hit must be
//                the destination patch; OptiX Prime itself is replaced by the BVH below)
//   assembly     vs/OptixPrimeFunctionality.cpp:6-34 (both directions from the integrand) and
//                :311-366 (reverse entry by reciprocity)
#include "dr_internal.h"

#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <atomic>
#include <thread>
#include <vector>
#include <rocprim/rocprim.hpp>

namespace dr {

// ---------------------------------------------------------------------------------------
// fp32 helpers in glm 0.9.8.4's operation order (func_geometric.inl)
// ---------------------------------------------------------------------------------------
struct f3 { float x, y, z; };
typedef float v8f __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f3 ld3(const float* p) { return f3{ p[0], p[1], p[2] }; }
__device__ __forceinline__ f3 operator+(f3 a, f3 b) { return f3{ a.x + b.x, a.y + b.y, a.z + b.z }; }
__device__ __forceinline__ f3 operator-(f3 a, f3 b) { return f3{ a.x - b.x, a.y - b.y, a.z - b.z }; }
__device__ __forceinline__ f3 operator*(f3 a, float s) { return f3{ a.x * s, a.y * s, a.z * s }; }
__device__ __forceinline__ f3 div3(f3 a, float s) { return f3{ a.x / s, a.y / s, a.z / s }; }
__device__ __forceinline__ float dot3(f3 a, f3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
__device__ __forceinline__ f3 cross3(f3 x, f3 y) {
    return f3{ x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y };
}
__device__ __forceinline__ float len3(f3 a) { return sqrtf(dot3(a, a)); }
__device__ __forceinline__ f3 normalize3(f3 a) { return a * (1.0f / sqrtf(dot3(a, a))); }
// 0.5 (double) * length, truncated to float == 0.5f * length exactly
__device__ __forceinline__ float surface3(f3 a, f3 b, f3 c) { return 0.5f * len3(cross3(b - a, c - a)); }
__device__ __forceinline__ f3 centre3(f3 p0, f3 p1, f3 p2) {
    f3 s = (p0 + p1) + p2;
    return f3{ s.x / 3.0f, s.y / 3.0f, s.z / 3.0f };
}

#define DR_PIF 3.14159265358979323846f

// ---------------------------------------------------------------------------------------
// per-patch records
// ---------------------------------------------------------------------------------------
__global__ void k_patch_records(int N, const float* __restrict__ vtx, const float* __restrict__ nrm,
                                const int* __restrict__ tv, const int* __restrict__ tn, float box_pad,
                                PatchRec* __restrict__ patch, TriRec* __restrict__ tri) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= N) return;
    f3 a = ld3(vtx + 3 * (size_t)tv[3 * (size_t)t + 0]);
    f3 b = ld3(vtx + 3 * (size_t)tv[3 * (size_t)t + 1]);
    f3 c = ld3(vtx + 3 * (size_t)tv[3 * (size_t)t + 2]);
    // midpoint split, vs/triangle_math.cpp:60-74
    f3 iA = div3(b - a, 2.0f) + a;
    f3 iC = div3(c - a, 2.0f) + a;
    f3 iB = div3(b - c, 2.0f) + c;
    f3 q[4][3] = { { a, iC, iA }, { iC, c, iB }, { iA, iB, b }, { iA, iB, iC } };
    PatchRec r;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        f3 ce = centre3(q[k][0], q[k][1], q[k][2]);
        r.cen[k][0] = ce.x; r.cen[k][1] = ce.y; r.cen[k][2] = ce.z;
        r.sa[k] = surface3(q[k][0], q[k][1], q[k][2]);
    }
    // averaged OBJ vertex normal, vs/triangle_math.cpp:23-29
    f3 n0 = ld3(nrm + 3 * (size_t)tn[3 * (size_t)t + 0]);
    f3 n1 = ld3(nrm + 3 * (size_t)tn[3 * (size_t)t + 1]);
    f3 n2 = ld3(nrm + 3 * (size_t)tn[3 * (size_t)t + 2]);
    f3 s = (n0 + n1) + n2;
    f3 n = normalize3(f3{ s.x / 3.0f, s.y / 3.0f, s.z / 3.0f });
    r.nrm[0] = n.x; r.nrm[1] = n.y; r.nrm[2] = n.z;
    r.area = surface3(a, b, c);
    patch[t] = r;
    TriRec T;
    f3 e1 = b - a, e2 = c - a;
    T.a[0] = a.x; T.a[1] = a.y; T.a[2] = a.z;
    T.e1[0] = e1.x; T.e1[1] = e1.y; T.e1[2] = e1.z;
    T.e2[0] = e2.x; T.e2[1] = e2.y; T.e2[2] = e2.z;
    T.id = t;
    const float px[3][3] = { { a.x, a.y, a.z }, { a.x + e1.x, a.y + e1.y, a.z + e1.z }, { a.x + e2.x, a.y + e2.y, a.z + e2.z } };
#pragma unroll
    for (int x = 0; x < 3; x++) {   // the box of the triangle the ray test sees, padded (same floats as the oracle)
        T.lo[x] = fminf(px[0][x], fminf(px[1][x], px[2][x])) - box_pad;
        T.hi[x] = fmaxf(px[0][x], fmaxf(px[1][x], px[2][x])) + box_pad;
    }
    tri[t] = T;
}

hipError_t launch_patch_records(hipStream_t st, int N, const float* vtx, const float* nrm,
                                const int* tv, const int* tn, float box_pad, PatchRec* patch, TriRec* tri) {
    hipLaunchKernelGGL(k_patch_records, dim3((N + 255) / 256), dim3(256), 0, st, N, vtx, nrm, tv, tn, box_pad, patch, tri);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------
// LBVH (Karras 2012): Morton keys -> radix sort -> hierarchy -> bottom-up bounds ->
// pre-order threaded layout.  Replaces rtpModelUpdate (vs/OptixPrimeFunctionality.cpp:43-47).
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long expand21(unsigned int v) {
    unsigned long long x = v & 0x1fffffull;
    x = (x | x << 32) & 0x1f00000000ffffull;
    x = (x | x << 16) & 0x1f0000ff0000ffull;
    x = (x | x << 8) & 0x100f00f00f00f00full;
    x = (x | x << 4) & 0x10c30c30c30c30c3ull;
    x = (x | x << 2) & 0x1249249249249249ull;
    return x;
}

__device__ __forceinline__ void tri_bounds(const TriRec& T, float lo[3], float hi[3]) {
#pragma unroll
    for (int a = 0; a < 3; a++) { lo[a] = T.lo[a]; hi[a] = T.hi[a]; }
}

// key_mode 0: 63-bit Morton code of the gate box's centre (x, y, z interleaved).
// key_mode 1 / 2: three leading bits = the orientation class of the triangle (dominant axis of its geometric normal and the
// normal's sign), then 60 bits of Morton code -- in x, y, z order (1) or with the dominant axis leading (2).  The top of the
// tree then separates surfaces by the way they face before it separates space: parallel sheets make thin nodes, where
// Morton octants of a room's corner make fat ones that every ray through the room's interior enters.  Any key gives a valid
// tree (and bit-identical results: the per-triangle tests define the hits); the key only decides how many nodes a walk visits.
__global__ void k_morton(int N, const TriRec* __restrict__ tri, float3 slo, float3 sinv, int key_mode,
                         unsigned long long* __restrict__ keys, int* __restrict__ vals) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= N) return;
    float lo[3], hi[3];
    const TriRec T = tri[t];
    tri_bounds(T, lo, hi);
    float c[3];
    c[0] = (0.5f * (lo[0] + hi[0]) - slo.x) * sinv.x;
    c[1] = (0.5f * (lo[1] + hi[1]) - slo.y) * sinv.y;
    c[2] = (0.5f * (lo[2] + hi[2]) - slo.z) * sinv.z;
    if (key_mode == 0) {
        unsigned int ix = (unsigned int)fminf(fmaxf(c[0] * 2097152.0f, 0.0f), 2097151.0f);
        unsigned int iy = (unsigned int)fminf(fmaxf(c[1] * 2097152.0f, 0.0f), 2097151.0f);
        unsigned int iz = (unsigned int)fminf(fmaxf(c[2] * 2097152.0f, 0.0f), 2097151.0f);
        keys[t] = (expand21(ix) << 2) | (expand21(iy) << 1) | expand21(iz);
    } else {
        float n[3] = { T.e1[1] * T.e2[2] - T.e1[2] * T.e2[1], T.e1[2] * T.e2[0] - T.e1[0] * T.e2[2],
                       T.e1[0] * T.e2[1] - T.e1[1] * T.e2[0] };
        int dom = 0;
        if (fabsf(n[1]) > fabsf(n[dom])) dom = 1;
        if (fabsf(n[2]) > fabsf(n[dom])) dom = 2;
        const unsigned long long cls = (unsigned long long)(dom * 2 + (n[dom] < 0.0f ? 1 : 0));
        unsigned int q[3];
        for (int a = 0; a < 3; a++) q[a] = (unsigned int)fminf(fmaxf(c[a] * 1048576.0f, 0.0f), 1048575.0f);
        const int a0 = key_mode == 2 ? dom : 0, a1 = (a0 + 1) % 3, a2 = (a0 + 2) % 3;
        keys[t] = (cls << 60) | (expand21(q[a0]) << 2) | (expand21(q[a1]) << 1) | expand21(q[a2]);
    }
    vals[t] = t;
}

// common-prefix length of sorted keys i and j; index bits break ties between equal keys
__device__ __forceinline__ int delta(const unsigned long long* keys, int N, int i, int j) {
    if (j < 0 || j >= N) return -1;
    unsigned long long a = keys[i], b = keys[j];
    if (a == b) return 64 + __clz((unsigned int)i ^ (unsigned int)j);
    return __clzll(a ^ b);
}

// node ids: internal i in [0,N-1), leaf k -> N-1+k
__global__ void k_hierarchy(int N, const unsigned long long* __restrict__ keys, int* __restrict__ left,
                            int* __restrict__ right, int* __restrict__ first, int* __restrict__ last,
                            int* __restrict__ parent) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N - 1) return;
    int d = (delta(keys, N, i, i + 1) - delta(keys, N, i, i - 1)) >= 0 ? 1 : -1;
    int dmin = delta(keys, N, i, i - d);
    int lmax = 2;
    while (delta(keys, N, i, i + lmax * d) > dmin) lmax <<= 1;
    int l = 0;
    for (int t = lmax >> 1; t >= 1; t >>= 1)
        if (delta(keys, N, i, i + (l + t) * d) > dmin) l += t;
    int j = i + l * d;
    int dnode = delta(keys, N, i, j);
    int s = 0;
    int t = l;
    do {
        t = (t + 1) >> 1;
        if (delta(keys, N, i, i + (s + t) * d) > dnode) s += t;
    } while (t > 1);
    int gamma = i + s * d + min(d, 0);
    int lo = min(i, j), hi = max(i, j);
    int lc = (lo == gamma) ? (N - 1 + gamma) : gamma;
    int rc = (hi == gamma + 1) ? (N - 1 + gamma + 1) : (gamma + 1);
    left[i] = lc; right[i] = rc; first[i] = lo; last[i] = hi;
    parent[lc] = i; parent[rc] = i;
    if (i == 0) parent[0] = -1;
}

// Bounds and written sizes of all nodes WITHOUT a bottom-up pass.  Every node covers a stretch [first, last] of the leaf order,
// so its box is a range union over the leaves' boxes and the number of nodes its subtree contributes to the output a range
// count -- both answered from small tables, every node on its own thread, no atomics, no fences.  (Round 2's k_refit climbed
// from every leaf with an arrival counter per node and two agent-scope fences per level: 380 us at 64k patches, a third of
// the build, because a fence writes back and invalidates an XCD's L2.)  Tables: per leaf the unions of its 64-leaf block up to
// it and from it on (pre64 / suf64), the blocks' boxes (blk) and the same one level up for blocks of 64 blocks (4096 leaves).
// Node boxes are plain unions of the triangles' (already padded) gate boxes -- the fp32 slab test is monotone under box
// enlargement, so culling by them is exact -- and min / max are order-free: the same floats as any other order gives.
struct Box6 { float lo[3], hi[3]; };
__device__ __forceinline__ Box6 box_empty() { return Box6{ { INFINITY, INFINITY, INFINITY }, { -INFINITY, -INFINITY, -INFINITY } }; }
__device__ __forceinline__ Box6 box_join(Box6 a, const Box6& b) {
    for (int x = 0; x < 3; x++) { a.lo[x] = fminf(a.lo[x], b.lo[x]); a.hi[x] = fmaxf(a.hi[x], b.hi[x]); }
    return a;
}
__device__ __forceinline__ Box6 box_ld(const float* p) { return Box6{ { p[0], p[1], p[2] }, { p[3], p[4], p[5] } }; }
__device__ __forceinline__ void box_st(float* p, const Box6& b) { for (int x = 0; x < 3; x++) { p[x] = b.lo[x]; p[3 + x] = b.hi[x]; } }

// inclusive unions over the lanes of a wave: up to and including this lane (pre), from this lane on (suf)
__device__ __forceinline__ void wave_box_scans(const Box6& mine, int lane, Box6& pre, Box6& suf) {
    pre = mine; suf = mine;
    for (int o = 1; o < 64; o <<= 1) {
        Box6 u, d;
        for (int x = 0; x < 3; x++) {
            u.lo[x] = __shfl_up(pre.lo[x], o); u.hi[x] = __shfl_up(pre.hi[x], o);
            d.lo[x] = __shfl_down(suf.lo[x], o); d.hi[x] = __shfl_down(suf.hi[x], o);
        }
        if (lane >= o) pre = box_join(pre, u);
        if (lane + o < 64) suf = box_join(suf, d);
    }
}

__global__ __launch_bounds__(256) void k_refit_leaves(int N, const TriRec* __restrict__ tri, const int* __restrict__ sorted_tri, TriRec* __restrict__ tri_sorted,
                                                      int* __restrict__ pos, float* __restrict__ box /* (2N-1) x 6 */, int* __restrict__ esize,
                                                      float* __restrict__ pre64, float* __restrict__ suf64, float* __restrict__ blk) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x, lane = threadIdx.x & 63;
    Box6 mine = box_empty();
    if (k < N) {
        const TriRec T = tri[sorted_tri[k]];
        tri_sorted[k] = T;
        pos[sorted_tri[k]] = k;
        for (int x = 0; x < 3; x++) { mine.lo[x] = T.lo[x]; mine.hi[x] = T.hi[x]; }
        box_st(box + 6 * (size_t)(N - 1 + k), mine);
        esize[N - 1 + k] = 1;
    }
    Box6 pre, suf;
    wave_box_scans(mine, lane, pre, suf);
    if (k < N) { box_st(pre64 + 6 * (size_t)k, pre); box_st(suf64 + 6 * (size_t)k, suf); }
    if (lane == 63 && k - 63 < N) box_st(blk + 6 * (size_t)(k >> 6), pre);       // (lanes past the end carry empty boxes)
}

// the same one level up: entries = the 64-leaf blocks' boxes
__global__ __launch_bounds__(256) void k_refit_blocks(int nblk, const float* __restrict__ blk, float* __restrict__ pre_b, float* __restrict__ suf_b,
                                                      float* __restrict__ sblk) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x, lane = threadIdx.x & 63;
    const Box6 mine = k < nblk ? box_ld(blk + 6 * (size_t)k) : box_empty();
    Box6 pre, suf;
    wave_box_scans(mine, lane, pre, suf);
    if (k < nblk) { box_st(pre_b + 6 * (size_t)k, pre); box_st(suf_b + 6 * (size_t)k, suf); }
    if (lane == 63 && k - 63 < nblk) box_st(sblk + 6 * (size_t)(k >> 6), pre);
}

// w[k] = 1 where a WRITTEN leaf starts: a node of at most LEAF_MAX triangles under a parent of more (its subtree is collapsed)
__global__ void k_refit_marks(int N, const int* __restrict__ first, const int* __restrict__ last, const int* __restrict__ parent, int* __restrict__ w) {
    const int id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= 2 * N - 1) return;
    const bool internal = id < N - 1;
    const int f = internal ? first[id] : id - (N - 1);
    const int cnt = internal ? last[id] - first[id] + 1 : 1;
    if (cnt > LEAF_MAX) return;
    const int p = N > 1 ? parent[id] : -1;
    if (p < 0 || last[p] - first[p] + 1 > LEAF_MAX) w[f] = 1;
}

__global__ void k_refit_nodes(int N, const int* __restrict__ first, const int* __restrict__ last, const int* __restrict__ wsum /* exclusive, N + 1 */,
                              const float* __restrict__ pre64, const float* __restrict__ suf64, const float* __restrict__ blk,
                              const float* __restrict__ pre_b, const float* __restrict__ suf_b, const float* __restrict__ sblk,
                              float* __restrict__ box, int* __restrict__ esize) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N - 1) return;
    const int f = first[i], l = last[i];
    const int bf = f >> 6, bl = l >> 6;
    Box6 r;
    if (bf == bl) {
        r = box_empty();
        for (int k = f; k <= l; k++) r = box_join(r, box_ld(box + 6 * (size_t)(N - 1 + k)));
    } else {
        r = box_join(box_ld(suf64 + 6 * (size_t)f), box_ld(pre64 + 6 * (size_t)l));
        const int m0 = bf + 1, m1 = bl - 1;                      // whole blocks in between
        if (m0 <= m1) {
            const int sf = m0 >> 6, sl = m1 >> 6;
            if (sf == sl) { for (int q = m0; q <= m1; q++) r = box_join(r, box_ld(blk + 6 * (size_t)q)); }
            else {
                r = box_join(r, box_join(box_ld(suf_b + 6 * (size_t)m0), box_ld(pre_b + 6 * (size_t)m1)));
                for (int q = sf + 1; q <= sl - 1; q++) r = box_join(r, box_ld(sblk + 6 * (size_t)q));
            }
        }
    }
    box_st(box + 6 * (size_t)i, r);
    // nodes this subtree contributes to the output: one if it is collapsed into a leaf, else a full binary tree over its written leaves
    esize[i] = (l - f + 1 <= LEAF_MAX) ? 1 : 2 * (wsum[l + 1] - wsum[f]) - 1;
}

// A node is written iff no proper ancestor is collapsed; it is a leaf iff it covers at most
// LEAF_MAX triangles.  Its pre-order index counts, on the way to the root, one per ancestor
// plus the written size of every left sibling subtree.
__global__ void k_emit(int N, const int* __restrict__ left, const int* __restrict__ first,
                       const int* __restrict__ last, const int* __restrict__ parent,
                       const float* __restrict__ box, const int* __restrict__ esize, float node_pad,
                       const TriRec* __restrict__ tri_sorted, BvhNode* __restrict__ nodes, BvhNode* __restrict__ nodes_lh,
                       int* __restrict__ pre /* 2N-1, preset to -1: pre-order index of every node that is written */,
                       BvhPair* __restrict__ pairs, BvhPair* __restrict__ pairs_lh, int* __restrict__ depth_max,
                       int* __restrict__ items /* 2N-1: the pair walk's item of every written node (k_paths) */) {
    int id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= 2 * N - 1) return;
    const bool internal = id < N - 1;
    const int f = internal ? first[id] : id - (N - 1);
    const int cnt = internal ? last[id] - first[id] + 1 : 1;
    int idx = 0;
    // the same walk to the root also gives the node's index among the written INTERIOR nodes in pre-order (a written subtree
    // of es nodes is a full binary tree: (es - 1) / 2 of them are interior), its parent's, and its depth: the pair records
    int iidx = 0, first_di = 0, depth = 0;
    bool is_right = false;
    int cur = id;
    while (cur != 0 && N > 1) {
        int p = parent[cur];
        if (last[p] - first[p] + 1 <= LEAF_MAX) return;      // inside a collapsed subtree
        idx += 1;
        int di = 1;
        if (left[p] != cur) { idx += esize[left[p]]; di += (esize[left[p]] - 1) >> 1; }
        if (cur == id) { first_di = di; is_right = left[p] != cur; }
        iidx += di;
        depth++;
        cur = p;
    }
    BvhNode nd;
    const float* b = box + 6 * (size_t)id;
    for (int a = 0; a < 3; a++) {
        // centre and half-extent of a box that contains [lo - node_pad, hi + node_pad] whatever the roundings
        const float c = 0.5f * b[a] + 0.5f * b[3 + a];
        const float h = fmaxf(b[3 + a] - c, c - b[a]);
        nd.c[a] = c;
        nd.h[a] = h * 1.000001f + node_pad;
    }
    nd.skip = (idx + esize[id]) * (int)sizeof(BvhNode);
    nd.tri = (cnt <= LEAF_MAX) ? f * 8 + (cnt - 1) : -1;
    if (cnt == 2) {
        // the two triangles of a quad have the same gate box: flag it, the walk then tests that box once
        const TriRec &t0 = tri_sorted[f], &t1 = tri_sorted[f + 1];
        bool same = true;
        for (int a = 0; a < 3; a++) same = same && (t0.lo[a] == t1.lo[a]) && (t0.hi[a] == t1.hi[a]);
        if (same) nd.tri |= 4;
    }
    nodes[idx] = nd;
    // the same node as lower / upper corner (c[] = lo, h[] = hi) for the sign-specialised node test (DR_NODE_TEST_S): a box that
    // holds [c - h, c + h] whatever the roundings of the two operations below
    BvhNode lh = nd;
    for (int a = 0; a < 3; a++) {
        const float m = 4e-7f * (fabsf(nd.c[a]) + nd.h[a]);
        lh.c[a] = (nd.c[a] - nd.h[a]) - m;
        lh.h[a] = (nd.c[a] + nd.h[a]) + m;
    }
    nodes_lh[idx] = lh;
    pre[id] = idx;
    // The sibling-pair form of the same tree (BvhPair, dr_internal.h): this node's box and "item" go into its PARENT's record.
    if (pairs != nullptr) {
        const int item = (cnt <= LEAF_MAX) ? (int)(0x80000000u | (unsigned)nd.tri) : iidx * (int)sizeof(BvhPair);
        items[id] = item;
        if (depth > 0) {
            BvhNode a = nd, b = lh;
            a.skip = item; a.tri = 0; b.skip = item; b.tri = 0;
            pairs[iidx - first_di].c[is_right ? 1 : 0] = a;
            pairs_lh[iidx - first_di].c[is_right ? 1 : 0] = b;
        } else if (cnt <= LEAF_MAX) {
            // the root is a leaf (one or two triangles): one pseudo record -- all-space boxes, the leaf and a leaf of the
            // never-hit padding records behind the last triangle
            BvhNode a, b;
            for (int x = 0; x < 3; x++) { a.c[x] = 0.0f; a.h[x] = INFINITY; b.c[x] = -INFINITY; b.h[x] = INFINITY; }
            a.tri = 0; b.tri = 0;
            a.skip = item; b.skip = item;
            pairs[0].c[0] = a; pairs_lh[0].c[0] = b;
            a.skip = b.skip = (int)(0x80000000u | (unsigned)(N * 8));
            pairs[0].c[1] = a; pairs_lh[0].c[1] = b;
        }
        atomicMax(depth_max, depth);
    }
}

// The path records of every patch (PathHdr, dr_internal.h): climb from the patch's leaf to the root, copying the sibling
// of every node on the way -- its box as lower / upper corner (the sign-specialised node test) and its item for the pair walk.
__global__ void k_paths(int N, const int* __restrict__ pos, const int* __restrict__ left, const int* __restrict__ right,
                        const int* __restrict__ parent, const int* __restrict__ pre, const int* __restrict__ items,
                        const BvhNode* __restrict__ nodes, const BvhNode* __restrict__ nodes_lh, BvhNode* __restrict__ rec, PathHdr* __restrict__ hdr) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= N) return;
    int id = N - 1 + pos[p];
    if (N > 1)
        while (pre[id] < 0) id = parent[id];          // inside a collapsed subtree: up to the leaf that was written for it
    PathHdr h;
    h.leaf = (int)(0x80000000u | (unsigned)nodes[N > 1 ? pre[id] : 0].tri);
    h.turns_lo = 0u; h.turns_hi = 0u;
    int D = 0;
    if (N > 1)
        for (int c = id; c != 0; c = parent[c]) D++;
    h.depth = D;
    if (D > PATH_RECS) { h.depth = -1; hdr[p] = h; return; }
    int d = D;
    for (int c = id; d >= 1; d--) {
        const int q = parent[c];
        const bool is_right = right[q] == c;
        const int sib = is_right ? left[q] : right[q];
        if (is_right) { if (d - 1 < 32) h.turns_lo |= 1u << (d - 1); else h.turns_hi |= 1u << (d - 33); }
        BvhNode nd = nodes_lh[pre[sib]];
        nd.skip = items[sib];
        nd.tri = 0;
        rec[(size_t)p * PATH_RECS + (d - 1)] = nd;
        c = q;
    }
    hdr[p] = h;
}

__global__ void k_pad_tris(int N, TriRec* __restrict__ tri_sorted) {
    int k = threadIdx.x;
    if (k < LEAF_MAX) {                 // records a leaf's fixed-width fetch may touch past the end: never hit
        TriRec T;                       // gate = a point at +inf: every slab interval is [inf,inf] or [-inf,-inf]
        for (int a = 0; a < 3; a++) { T.a[a] = 0.0f; T.e1[a] = 0.0f; T.e2[a] = 0.0f; T.lo[a] = INFINITY; T.hi[a] = INFINITY; }
        T.id = -1;
        tri_sorted[N + k] = T;
    }
}

// ---------------------------------------------------------------------------------------
// The SAH topology ON THE DEVICE (dr_options::tree = SAH, the default from 6 144 patches up): the same top-down binned build as
// sah_hierarchy_host below -- 32 bins per axis over the centroids of the patches' gate boxes, cost = (area of the box grown by
// half a mean patch diagonal) x count, cuts that leave less than 1/32 on a side not admitted above 64 patches, the same float
// expressions in the same order, ties to the first (axis, bin) -- level by level: one workgroup per open node and level
// (k_sah_level, grid-stride over the level's node queue), each of which finds its node's centroid bounds, bins its patches in
// LDS (integer atomics on order-preserving keys of the floats), evaluates the 3 x 31 planes, partitions its stretch of the
// leaf order into the other of two order buffers and queues its children; nodes of one or two patches are closed on the spot.
// No admissible plane (every cut lopsided, or all centroids in one bin): the exact median along the longest axis by a radix
// select on the centroids' keys (the host uses nth_element), which keeps the depth logarithmic on any input.  Node ids, ranges
// and arrays are the host builder's (and k_hierarchy's): a node over n leaves owns the interior ids [id, id + n - 1).
// Any tree gives bit-identical results; this one is the host's up to the order inside a leaf pair and tie cases.
// ---------------------------------------------------------------------------------------
struct SahJob { int b, e, id; };
constexpr int SAH_BIG = 4096;         // nodes of this many patches or more are split by 1024-thread workgroups, the others by 256-thread ones
constexpr int SAH_NBMAX = 128;

__device__ __forceinline__ int f2ord(float f) { const int i = __float_as_int(f); return i >= 0 ? i : i ^ 0x7fffffff; }
__device__ __forceinline__ float ord2f(int i) { return __int_as_float(i >= 0 ? i : i ^ 0x7fffffff); }

// centroids of the gate boxes (the host's expression), the identity order, and the mean of the boxes' diagonals: per block a
// fixed subset of the patches summed in a fixed order (then k_sah_diag_mean over the blocks' sums): the same value every run
constexpr int SAH_PREP_BLOCKS = 128;
__global__ __launch_bounds__(256) void k_sah_prepare(int N, const TriRec* __restrict__ tri, float* __restrict__ cen, double* __restrict__ diag_part,
                                                     int* __restrict__ order) {
    __shared__ double sh[256];
    double a = 0.0;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < N; i += SAH_PREP_BLOCKS * 256) {
        const TriRec T = tri[i];
        double q = 0.0;
        for (int x = 0; x < 3; x++) {
            cen[3 * (size_t)i + x] = 0.5f * T.lo[x] + 0.5f * T.hi[x];
            const double d = (double)T.hi[x] - (double)T.lo[x];
            q += d * d;
        }
        a += sqrt(q);
        order[i] = i;
    }
    sh[threadIdx.x] = a;
    __syncthreads();
    for (int w = 128; w >= 1; w >>= 1) {
        if ((int)threadIdx.x < w) sh[threadIdx.x] += sh[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) diag_part[1 + blockIdx.x] = sh[0];
}
__global__ void k_sah_diag_mean(int N, double* __restrict__ diag_part) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double a = 0.0;
    for (int k = 0; k < SAH_PREP_BLOCKS; k++) a += diag_part[1 + k];
    diag_part[0] = a / (double)N;
}

__device__ __forceinline__ float sah_area(const float lo[3], const float hi[3], float grow) {
    const float dx = fmaxf(hi[0] - lo[0], 0.0f) + grow, dy = fmaxf(hi[1] - lo[1], 0.0f) + grow, dz = fmaxf(hi[2] - lo[2], 0.0f) + grow;
    return dx * dy + dy * dz + dz * dx;
}

// exclusive scan of one flag per thread over the block (SAH_NT threads); total in tot
template <int SAH_NT>
__device__ __forceinline__ int block_rank(bool flag, int* sWave /* SAH_NT / 64 + 1 */, int& tot) {
    const unsigned long long m = __ballot(flag);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) sWave[wave] = __popcll(m);
    __syncthreads();
    int base = 0;
    tot = 0;
    for (int w = 0; w < SAH_NT / 64; w++) { const int c = sWave[w]; if (w < wave) base += c; tot += c; }
    __syncthreads();
    return base + __popcll(m & ((1ull << lane) - 1ull));
}

// BIG nodes (SAH_BIG patches or more): their two gathering passes -- centroid bounds, bins -- run on many workgroups, one per
// SAH_CHUNK patches of the node's stretch (a single workgroup binning 64k patches spends its time in same-address LDS atomics:
// 235 us per level measured), each with its own bins in LDS, merged into the node's bins in global memory (integer atomics on
// the order-preserving keys, so the merged bins are the same whatever the order); the node's own workgroup then starts from
// those (k_sah_level<1024, true>).  Block -> (job, chunk): a scan over the level's few big jobs.
constexpr int SAH_CHUNK = 2048;
constexpr int SAH_GB = 3 * SAH_NBMAX * 7;        // ints of one big node's merged bins: per axis and bin: count, lo[3], hi[3]

__device__ __forceinline__ bool sah_find_chunk(const SahJob* __restrict__ q, int njobs, int block, SahJob& J, int& j_out, int& c0, int& c1) {
    int acc = 0;
    for (int j = 0; j < njobs; j++) {
        const SahJob X = q[j];
        const int nch = (X.e - X.b + SAH_CHUNK - 1) / SAH_CHUNK;
        if (block < acc + nch) { J = X; j_out = j; c0 = X.b + (block - acc) * SAH_CHUNK; c1 = min(X.e, c0 + SAH_CHUNK); return true; }
        acc += nch;
    }
    return false;
}

__global__ __launch_bounds__(256) void k_sah_big_bounds(const float* __restrict__ cen, const SahJob* __restrict__ qin, const int* __restrict__ n_in,
                                                        const int* __restrict__ src, int* __restrict__ gcb /* [job][6] ordered keys */) {
    SahJob J; int j, c0, c1;
    if (!sah_find_chunk(qin, *n_in, blockIdx.x, J, j, c0, c1)) return;
    int lo3[3] = { 0x7fffffff, 0x7fffffff, 0x7fffffff }, hi3[3] = { (int)0x80000000, (int)0x80000000, (int)0x80000000 };
    for (int k = c0 + threadIdx.x; k < c1; k += 256) {
        const int t = src[k];
        for (int a = 0; a < 3; a++) { const int o = f2ord(cen[3 * (size_t)t + a]); lo3[a] = min(lo3[a], o); hi3[a] = max(hi3[a], o); }
    }
    for (int a = 0; a < 3; a++) {
        for (int o = 32; o >= 1; o >>= 1) { lo3[a] = min(lo3[a], __shfl_xor(lo3[a], o)); hi3[a] = max(hi3[a], __shfl_xor(hi3[a], o)); }
        if ((threadIdx.x & 63) == 0) { atomicMin(&gcb[6 * j + a], lo3[a]); atomicMax(&gcb[6 * j + 3 + a], hi3[a]); }
    }
}

__global__ __launch_bounds__(256) void k_sah_big_bins(const TriRec* __restrict__ tri, const float* __restrict__ cen, int NB, const SahJob* __restrict__ qin,
                                                      const int* __restrict__ n_in, const int* __restrict__ src, const int* __restrict__ gcb,
                                                      int* __restrict__ gbins /* [job][SAH_GB] */) {
    __shared__ int sB[SAH_GB];
    SahJob J; int j, c0, c1;
    if (!sah_find_chunk(qin, *n_in, blockIdx.x, J, j, c0, c1)) return;
    for (int x = threadIdx.x; x < 3 * NB; x += 256) {
        const int a = x / NB, q = x % NB;
        int* B = sB + (a * SAH_NBMAX + q) * 7;
        B[0] = 0;
        for (int d = 0; d < 3; d++) { B[1 + d] = 0x7fffffff; B[4 + d] = (int)0x80000000; }
    }
    __syncthreads();
    float clo[3], scale[3];
    bool valid[3];
    for (int a = 0; a < 3; a++) {
        clo[a] = ord2f(gcb[6 * j + a]);
        const float ext = ord2f(gcb[6 * j + 3 + a]) - clo[a];
        scale[a] = (float)NB / ext;
        valid[a] = (ext > 0.0f) && isfinite(scale[a]);
    }
    for (int k = c0 + threadIdx.x; k < c1; k += 256) {
        const int t = src[k];
        const TriRec T = tri[t];
        for (int a = 0; a < 3; a++) {
            if (!valid[a]) continue;
            const int q = max(0, min(NB - 1, (int)((cen[3 * (size_t)t + a] - clo[a]) * scale[a])));
            int* B = sB + (a * SAH_NBMAX + q) * 7;
            atomicAdd(&B[0], 1);
            for (int d = 0; d < 3; d++) { atomicMin(&B[1 + d], f2ord(T.lo[d])); atomicMax(&B[4 + d], f2ord(T.hi[d])); }
        }
    }
    __syncthreads();
    int* G = gbins + (size_t)j * SAH_GB;
    for (int x = threadIdx.x; x < 3 * NB; x += 256) {
        const int a = x / NB, q = x % NB;
        const int* B = sB + (a * SAH_NBMAX + q) * 7;
        if (B[0] == 0) continue;
        int* Gq = G + (a * SAH_NBMAX + q) * 7;
        atomicAdd(&Gq[0], B[0]);
        for (int d = 0; d < 3; d++) { atomicMin(&Gq[1 + d], B[1 + d]); atomicMax(&Gq[4 + d], B[4 + d]); }
    }
}

// the merged bounds and bins back to "empty" for the next level's big nodes
__global__ void k_sah_big_reset(int n_jobs_cap, int* __restrict__ gcb, int* __restrict__ gbins) {
    const size_t x = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (x < (size_t)n_jobs_cap * 6) gcb[x] = (x % 6) < 3 ? 0x7fffffff : (int)0x80000000;
    if (x < (size_t)n_jobs_cap * SAH_GB) { const int f = (int)(x % 7); gbins[x] = f == 0 ? 0 : (f < 4 ? 0x7fffffff : (int)0x80000000); }
}

// A big node's patches into the other order buffer, a workgroup per chunk: the node's workgroup has published the split
// (gsel[job]: mode 0 plane / 1 median / 2 halves as they lie, axis, bin or key, patches on the left, how many of the keys equal
// to the median go left, then three cursors); a chunk counts its patches for either side, reserves that many places on each with
// one atomic per side and writes.  (Which of the equal keys go left, and the order inside a side, depend on the order the chunks
// arrive in: any choice is a valid tree, and the results do not depend on the tree.)
__global__ __launch_bounds__(256) void k_sah_big_partition(const float* __restrict__ cen, int NB, const SahJob* __restrict__ qin, const int* __restrict__ n_in,
                                                           const int* __restrict__ src, int* __restrict__ dst, int* __restrict__ order_out,
                                                           const int* __restrict__ gcb, int* __restrict__ gsel) {
    __shared__ int sWave[256 / 64 + 1];
    __shared__ int sBase[3];
    SahJob J; int j, c0, c1;
    if (!sah_find_chunk(qin, *n_in, blockIdx.x, J, j, c0, c1)) return;
    int* G = gsel + 8 * j;
    const int mode = G[0], sax = G[1], sval = G[2], nl = G[3], eq_quota = G[4];
    const int b = J.b, m = b + nl, nr = (J.e - J.b) - nl;
    const int tid = threadIdx.x;
    if (mode == 2) {                         // nothing moves
        for (int k = c0 + tid; k < c1; k += 256) { const int t = src[k]; dst[k] = t; if (((k < m) ? nl : nr) <= 2) order_out[k] = t; }
        return;
    }
    const float clo = ord2f(gcb[6 * j + sax]);
    const float scale = (float)NB / (ord2f(gcb[6 * j + 3 + sax]) - clo);
    auto classify = [&](int t, bool& is_eq) {          // plane: left or right; median: smaller / equal / larger
        is_eq = false;
        if (mode == 0) return max(0, min(NB - 1, (int)((cen[3 * (size_t)t + sax] - clo) * scale))) <= sval;
        const unsigned key = (unsigned)f2ord(cen[3 * (size_t)t + sax]) ^ 0x80000000u;
        is_eq = key == (unsigned)sval;
        return key < (unsigned)sval;
    };
    // the chunk's counts: left for sure, equal to the median's key
    int nL = 0, nE = 0;
    for (int k = c0 + tid; k < c1; k += 256) { bool eq; const bool l = classify(src[k], eq); nL += l ? 1 : 0; nE += eq ? 1 : 0; }
    for (int o = 32; o >= 1; o >>= 1) { nL += __shfl_xor(nL, o); nE += __shfl_xor(nE, o); }
    if (tid == 0) { sBase[0] = 0; sBase[1] = 0; }
    __syncthreads();
    if ((tid & 63) == 0) { atomicAdd(&sBase[0], nL); atomicAdd(&sBase[1], nE); }
    __syncthreads();
    const int totL = sBase[0], totE = sBase[1];
    __syncthreads();
    if (tid == 0) {
        // the equals first: how many of this chunk's go left
        int eq_left = 0;
        if (mode == 1 && totE > 0) { const int at = atomicAdd(&G[7], totE); eq_left = max(0, min(totE, eq_quota - at)); }
        const int left_here = totL + eq_left, right_here = (c1 - c0) - left_here;
        sBase[0] = atomicAdd(&G[5], left_here);
        sBase[1] = atomicAdd(&G[6], right_here);
        sBase[2] = eq_left;
    }
    __syncthreads();
    const int baseL = sBase[0], baseR = sBase[1], eq_left = sBase[2];
    int offL = 0, offR = 0, offE = 0;
    for (int base = c0; base < c1; base += 256) {
        const int k = base + tid;
        const bool in = k < c1;
        int t = 0;
        bool goes_left = false, is_eq = false;
        if (in) { t = src[k]; goes_left = classify(t, is_eq); }
        if (mode == 1) {
            int tE;
            const int re = block_rank<256>(is_eq, sWave, tE);
            if (is_eq) goes_left = (offE + re) < eq_left;
            offE += tE;
        }
        int tL, tR;
        const int rl = block_rank<256>(in && goes_left, sWave, tL);
        const int rr = block_rank<256>(in && !goes_left, sWave, tR);
        if (in) {
            const int at = goes_left ? b + baseL + offL + rl : m + baseR + offR + rr;
            dst[at] = t;
            if ((goes_left ? nl : nr) <= 2) order_out[at] = t;
        }
        offL += tL; offR += tR;
    }
}

// The next level's queues by size class: tiny nodes (3 .. tiny_max patches; one WAVE each, k_sah_level_wave), big ones (SAH_BIG and
// more; chunk-parallel gathering passes), the rest (one 256-thread workgroup each).  tiny_max = 0: no tiny class (more than 32 bins).
struct SahQueues { SahJob* tiny; int* n_tiny; SahJob* small; int* n_small; SahJob* big; int* n_big; int tiny_max; };
__device__ __forceinline__ void sah_push(const SahQueues& Q, int cb, int ce, int cid) {
    const int cn = ce - cb;
    if (cn >= SAH_BIG) { const int at = atomicAdd(Q.n_big, 1); Q.big[at] = SahJob{ cb, ce, cid }; }
    else if (cn > Q.tiny_max) { const int at = atomicAdd(Q.n_small, 1); Q.small[at] = SahJob{ cb, ce, cid }; }
    else { const int at = atomicAdd(Q.n_tiny, 1); Q.tiny[at] = SahJob{ cb, ce, cid }; }
}

// TINY nodes (at most 64 patches, at most 32 bins): one wave per node, four nodes per workgroup, no workgroup barrier anywhere --
// a lane holds one patch in registers, the centroid bounds are a butterfly over the lanes, the bins live in the wave's own piece
// of LDS, the planes' prefix / suffix boxes are scans with lanes = bins, the partition is a ballot.  (Most nodes of a tree are
// tiny: at 64k patches 15 of 16; one 256-thread workgroup per such node spent its time in barriers.)  Same float expressions and
// tie rules as k_sah_level and the host builder; the median fallback ranks the centroids by counting.
constexpr int SAH_WNB = 32;
__global__ __launch_bounds__(256) void k_sah_level_wave(int N, const TriRec* __restrict__ tri, const float* __restrict__ cen, const double* __restrict__ diag_mean,
                                                        float dilate, int NB, const SahJob* __restrict__ qin, const int* __restrict__ n_in, SahQueues Q,
                                                        const int* __restrict__ src, int* __restrict__ dst, int* __restrict__ order_out,
                                                        int* __restrict__ left, int* __restrict__ right, int* __restrict__ first, int* __restrict__ last,
                                                        int* __restrict__ parent) {
    __shared__ int sBins[4][3][SAH_WNB][7];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const float grow = dilate * (float)diag_mean[0];
    const int njobs = *n_in;
    int (*B)[SAH_WNB][7] = sBins[wv];
    for (int j = blockIdx.x * 4 + wv; j < njobs; j += gridDim.x * 4) {
        const SahJob J = qin[j];
        const int b = J.b, e = J.e, n = e - b;
        const bool has = lane < n;
        const int t = has ? src[b + lane] : 0;
        float c[3] = { 0.0f, 0.0f, 0.0f }, blo[3] = { 0.0f, 0.0f, 0.0f }, bhi[3] = { 0.0f, 0.0f, 0.0f };
        if (has)
            for (int a = 0; a < 3; a++) { c[a] = cen[3 * (size_t)t + a]; blo[a] = tri[t].lo[a]; bhi[a] = tri[t].hi[a]; }
        // ---- 1. bounds of the centroids
        float clo[3], chi[3], scale[3];
        bool valid[3];
        for (int a = 0; a < 3; a++) {
            float lo = has ? c[a] : INFINITY, hi = has ? c[a] : -INFINITY;
            for (int o = 32; o >= 1; o >>= 1) { lo = fminf(lo, __shfl_xor(lo, o)); hi = fmaxf(hi, __shfl_xor(hi, o)); }
            clo[a] = lo; chi[a] = hi;
            const float ext = hi - lo;
            scale[a] = (float)NB / ext;
            valid[a] = (ext > 0.0f) && isfinite(scale[a]);
        }
        // ---- 2. bins (this wave's own LDS; a wave's LDS operations complete in order)
        for (int x = lane; x < 3 * NB; x += 64) {
            int* E = B[x / NB][x % NB];
            E[0] = 0;
            for (int d = 0; d < 3; d++) { E[1 + d] = 0x7fffffff; E[4 + d] = (int)0x80000000; }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        int mybin[3] = { 0, 0, 0 };
        if (has)
            for (int a = 0; a < 3; a++) {
                if (!valid[a]) continue;
                const int q = max(0, min(NB - 1, (int)((c[a] - clo[a]) * scale[a])));
                mybin[a] = q;
                atomicAdd(&B[a][q][0], 1);
                for (int d = 0; d < 3; d++) { atomicMin(&B[a][q][1 + d], f2ord(blo[d])); atomicMax(&B[a][q][4 + d], f2ord(bhi[d])); }
            }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // ---- 3. the planes: lanes = bins; inclusive scans from the left and from the right
        float best = INFINITY;
        int best_x = 0x7fffffff, best_c = 0;
        for (int a = 0; a < 3; a++) {
            if (!valid[a]) continue;                    // (wave-uniform)
            int cnt = 0;
            float plo[3] = { INFINITY, INFINITY, INFINITY }, phi[3] = { -INFINITY, -INFINITY, -INFINITY };
            if (lane < NB && B[a][lane][0] > 0) {
                cnt = B[a][lane][0];
                for (int d = 0; d < 3; d++) { plo[d] = ord2f(B[a][lane][1 + d]); phi[d] = ord2f(B[a][lane][4 + d]); }
            }
            int pc = cnt, sc = cnt;
            float slo[3] = { plo[0], plo[1], plo[2] }, shi[3] = { phi[0], phi[1], phi[2] };
            for (int o = 1; o < SAH_WNB; o <<= 1) {
                const int upc = __shfl_up(pc, o), dsc = __shfl_down(sc, o);
                float ulo[3], uhi[3], dlo[3], dhi[3];
                for (int d = 0; d < 3; d++) {
                    ulo[d] = __shfl_up(plo[d], o); uhi[d] = __shfl_up(phi[d], o);
                    dlo[d] = __shfl_down(slo[d], o); dhi[d] = __shfl_down(shi[d], o);
                }
                if (lane >= o) { pc += upc; for (int d = 0; d < 3; d++) { plo[d] = fminf(plo[d], ulo[d]); phi[d] = fmaxf(phi[d], uhi[d]); } }
                if (lane + o < SAH_WNB) { sc += dsc; for (int d = 0; d < 3; d++) { slo[d] = fminf(slo[d], dlo[d]); shi[d] = fmaxf(shi[d], dhi[d]); } }
            }
            // the plane between bins `lane` and `lane + 1`: left = my prefix, right = the next lane's suffix
            const int rc = __shfl_down(sc, 1);
            float rlo[3], rhi[3];
            for (int d = 0; d < 3; d++) { rlo[d] = __shfl_down(slo[d], 1); rhi[d] = __shfl_down(shi[d], 1); }
            if (lane < NB - 1 && pc > 0 && rc > 0) {
                const float cost = sah_area(plo, phi, grow) * (float)pc + sah_area(rlo, rhi, grow) * (float)rc;
                if (cost < best) { best = cost; best_x = a * (NB - 1) + lane; best_c = pc; }
            }
        }
        for (int o = 32; o >= 1; o >>= 1) {
            const float c2 = __shfl_xor(best, o);
            const int x2 = __shfl_xor(best_x, o), n2 = __shfl_xor(best_c, o);
            if (c2 < best || (c2 == best && x2 < best_x)) { best = c2; best_x = x2; best_c = n2; }
        }
        // ---- the split
        bool goes_left;
        int nl;
        if (best_x != 0x7fffffff) {
            const int a = best_x / (NB - 1), q = best_x % (NB - 1);
            goes_left = (a == 0 ? mybin[0] : (a == 1 ? mybin[1] : mybin[2])) <= q;
            nl = best_c;
        } else {
            int axis = -1;
            for (int a = 0; a < 3; a++)
                if (chi[a] - clo[a] > 0.0f && (axis < 0 || chi[a] - clo[a] > chi[axis] - clo[axis])) axis = a;
            nl = n / 2;
            if (axis < 0) goes_left = lane < nl;
            else {
                // the exact median along `axis`: a patch's rank = how many others lie before it (ties by position)
                const float key = axis == 0 ? c[0] : (axis == 1 ? c[1] : c[2]);
                int rank = 0;
                for (int m = 0; m < n; m++) {
                    const float km = __shfl(key, m);
                    rank += (km < key || (km == key && m < lane)) ? 1 : 0;
                }
                goes_left = rank < nl;
            }
        }
        const int nr = n - nl, m = b + nl;
        // ---- partition (into the other order buffer; stretches that are final also into order_out)
        const unsigned long long ml = __ballot(has && goes_left), mr = __ballot(has && !goes_left);
        if (has) {
            const unsigned long long below = (1ull << lane) - 1ull;
            const int at = goes_left ? b + __popcll(ml & below) : m + __popcll(mr & below);
            dst[at] = t;
            if ((goes_left ? nl : nr) <= 2) order_out[at] = t;
        }
        // ---- the node's record and its children
        if (lane == 0) {
            first[J.id] = b; last[J.id] = e - 1;
            const int lc = nl == 1 ? N - 1 + b : J.id + 1;
            const int rc = nr == 1 ? N - 1 + m : J.id + 1 + (nl - 1);
            left[J.id] = lc; right[J.id] = rc; parent[lc] = J.id; parent[rc] = J.id;
            for (int side = 0; side < 2; side++) {
                const int cb = side ? m : b, ce = side ? e : m, cid = side ? rc : lc, cn = ce - cb;
                if (cn == 2) {
                    first[cid] = cb; last[cid] = cb + 1;
                    left[cid] = N - 1 + cb; right[cid] = N - 1 + cb + 1;
                    parent[N - 1 + cb] = cid; parent[N - 1 + cb + 1] = cid;
                } else if (cn >= 3) sah_push(Q, cb, ce, cid);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}


// PRE: the centroid bounds and bins of job j were formed by k_sah_big_bounds / k_sah_big_bins (gcb, gbins)
template <int SAH_NT, bool PRE>
__global__ __launch_bounds__(SAH_NT) void k_sah_level(int N, const TriRec* __restrict__ tri, const float* __restrict__ cen, const double* __restrict__ diag_mean,
                                                      float dilate, int NB, const SahJob* __restrict__ qin, const int* __restrict__ n_in, SahQueues Q,
                                                      const int* __restrict__ src, int* __restrict__ dst, int* __restrict__ order_out,
                                                      int* __restrict__ left, int* __restrict__ right, int* __restrict__ first, int* __restrict__ last,
                                                      int* __restrict__ parent, const int* __restrict__ gcb, const int* __restrict__ gbins, int* __restrict__ gsel) {
    __shared__ int sCnt[3][SAH_NBMAX];
    __shared__ int sLo[3][SAH_NBMAX][3], sHi[3][SAH_NBMAX][3];
    __shared__ int sC[6];                      // centroid bounds as ordered keys: lo[3], hi[3]
    __shared__ float sCost[SAH_NT];
    __shared__ int sIdx[SAH_NT];
    __shared__ int sWave[SAH_NT / 64 + 1];
    __shared__ int sHist[256];
    constexpr int U = 4;                       // patches per thread in flight in the gathering loops
    __shared__ int sSel[4];                    // split: mode (0 plane, 1 median, 2 halves as they lie), axis, bin / key, count on the left
    const int tid = threadIdx.x;
    const float grow = dilate * (float)diag_mean[0];
    const int njobs = *n_in;
    for (int j = blockIdx.x; j < njobs; j += gridDim.x) {
        const SahJob J = qin[j];
        const int b = J.b, e = J.e, n = e - b;
        // ---- 1. bounds of the centroids
        if (PRE) {
            if (tid < 6) sC[tid] = gcb[6 * j + tid];
            for (int x = tid; x < 3 * NB; x += SAH_NT) {
                const int a = x / NB, q = x % NB;
                const int* G = gbins + (size_t)j * SAH_GB + (a * SAH_NBMAX + q) * 7;
                sCnt[a][q] = G[0];
                for (int d = 0; d < 3; d++) { sLo[a][q][d] = G[1 + d]; sHi[a][q][d] = G[4 + d]; }
            }
        } else {
            if (tid < 6) sC[tid] = tid < 3 ? 0x7fffffff : (int)0x80000000;
            for (int x = tid; x < 3 * NB; x += SAH_NT) {
                const int a = x / NB, q = x % NB;
                sCnt[a][q] = 0;
                for (int d = 0; d < 3; d++) { sLo[a][q][d] = 0x7fffffff; sHi[a][q][d] = (int)0x80000000; }
            }
        }
        __syncthreads();
        if (!PRE) {
            int lo3[3] = { 0x7fffffff, 0x7fffffff, 0x7fffffff }, hi3[3] = { (int)0x80000000, (int)0x80000000, (int)0x80000000 };
            for (int k0 = b + tid; k0 < e; k0 += U * SAH_NT) {
                int t[U];
                float c[U][3];
#pragma unroll
                for (int u = 0; u < U; u++) { const int k = k0 + u * SAH_NT; t[u] = k < e ? src[k] : -1; }
#pragma unroll
                for (int u = 0; u < U; u++)
                    for (int a = 0; a < 3; a++) c[u][a] = t[u] >= 0 ? cen[3 * (size_t)t[u] + a] : 0.0f;
#pragma unroll
                for (int u = 0; u < U; u++)
                    if (t[u] >= 0)
                        for (int a = 0; a < 3; a++) { const int o = f2ord(c[u][a]); lo3[a] = min(lo3[a], o); hi3[a] = max(hi3[a], o); }
            }
            for (int a = 0; a < 3; a++) {
                for (int o = 32; o >= 1; o >>= 1) { lo3[a] = min(lo3[a], __shfl_xor(lo3[a], o)); hi3[a] = max(hi3[a], __shfl_xor(hi3[a], o)); }
                if ((tid & 63) == 0) { atomicMin(&sC[a], lo3[a]); atomicMax(&sC[3 + a], hi3[a]); }
            }
        }
        __syncthreads();
        float clo[3], chi[3], scale[3];
        bool valid[3];
        for (int a = 0; a < 3; a++) {
            clo[a] = ord2f(sC[a]); chi[a] = ord2f(sC[3 + a]);
            const float ext = chi[a] - clo[a];
            scale[a] = (float)NB / ext;
            valid[a] = (ext > 0.0f) && isfinite(scale[a]);
        }
        // ---- 2. bins
        for (int k0 = b + tid; !PRE && k0 < e; k0 += U * SAH_NT) {
            int t[U];
            float c[U][3], blo[U][3], bhi[U][3];
#pragma unroll
            for (int u = 0; u < U; u++) { const int k = k0 + u * SAH_NT; t[u] = k < e ? src[k] : -1; }
#pragma unroll
            for (int u = 0; u < U; u++)
                for (int a = 0; a < 3; a++) {
                    c[u][a] = t[u] >= 0 ? cen[3 * (size_t)t[u] + a] : 0.0f;
                    blo[u][a] = t[u] >= 0 ? tri[t[u]].lo[a] : 0.0f; bhi[u][a] = t[u] >= 0 ? tri[t[u]].hi[a] : 0.0f;
                }
#pragma unroll
            for (int u = 0; u < U; u++) {
                if (t[u] < 0) continue;
                for (int a = 0; a < 3; a++) {
                    if (!valid[a]) continue;
                    const int q = max(0, min(NB - 1, (int)((c[u][a] - clo[a]) * scale[a])));
                    atomicAdd(&sCnt[a][q], 1);
                    for (int d = 0; d < 3; d++) { atomicMin(&sLo[a][q][d], f2ord(blo[u][d])); atomicMax(&sHi[a][q][d], f2ord(bhi[u][d])); }
                }
            }
        }
        __syncthreads();
        // ---- 3. the planes: candidate x = a * (NB - 1) + q cuts axis a between bins q and q + 1
        float best = INFINITY;
        int best_x = 0x7fffffff;
        for (int x = tid; x < 3 * (NB - 1); x += SAH_NT) {
            const int a = x / (NB - 1), q = x % (NB - 1);
            if (!valid[a]) continue;
            float lo3[3] = { INFINITY, INFINITY, INFINITY }, hi3[3] = { -INFINITY, -INFINITY, -INFINITY };
            int c = 0;
            for (int p = 0; p <= q; p++) {
                if (sCnt[a][p] == 0) continue;                   // (an empty bin's box is [inf, -inf]: min / max leave the others alone)
                for (int d = 0; d < 3; d++) { lo3[d] = fminf(lo3[d], ord2f(sLo[a][p][d])); hi3[d] = fmaxf(hi3[d], ord2f(sHi[a][p][d])); }
                c += sCnt[a][p];
            }
            float rlo[3] = { INFINITY, INFINITY, INFINITY }, rhi[3] = { -INFINITY, -INFINITY, -INFINITY };
            int rc = 0;
            for (int p = NB - 1; p > q; p--) {
                if (sCnt[a][p] == 0) continue;
                for (int d = 0; d < 3; d++) { rlo[d] = fminf(rlo[d], ord2f(sLo[a][p][d])); rhi[d] = fmaxf(rhi[d], ord2f(sHi[a][p][d])); }
                rc += sCnt[a][p];
            }
            if (c == 0 || rc == 0) continue;
            if (n > 64 && (long long)min(c, rc) * 32 < n) continue;       // (keeps the depth, and the build's cost, logarithmic)
            const float cost = sah_area(lo3, hi3, grow) * (float)c + sah_area(rlo, rhi, grow) * (float)rc;
            if (cost < best) { best = cost; best_x = x; }
        }
        sCost[tid] = best; sIdx[tid] = best_x;
        __syncthreads();
        for (int w = SAH_NT / 2; w >= 1; w >>= 1) {
            if (tid < w) {
                const float c2 = sCost[tid + w]; const int i2 = sIdx[tid + w];
                if (c2 < sCost[tid] || (c2 == sCost[tid] && i2 < sIdx[tid])) { sCost[tid] = c2; sIdx[tid] = i2; }
            }
            __syncthreads();
        }
        // ---- the split: a plane; else the median along the longest axis with extent; else the halves as they lie
        int axis = -1;
        if (sIdx[0] != 0x7fffffff && n > 2) {
            if (tid == 0) {
                const int a = sIdx[0] / (NB - 1), q = sIdx[0] % (NB - 1);
                int c = 0;
                for (int p = 0; p <= q; p++) c += sCnt[a][p];
                sSel[0] = 0; sSel[1] = a; sSel[2] = q; sSel[3] = c;
            }
        } else {
            for (int a = 0; a < 3; a++)
                if (chi[a] - clo[a] > 0.0f && (axis < 0 || chi[a] - clo[a] > chi[axis] - clo[axis])) axis = a;
            if (axis < 0 || n <= 2) { if (tid == 0) { sSel[0] = 2; sSel[1] = 0; sSel[2] = 0; sSel[3] = n / 2; } }
            else {
                // radix select of the (n / 2)-th smallest key (0-based) along `axis`: four rounds of eight bits, most significant first
                unsigned prefix = 0u, pmask = 0u;
                int want = n / 2;                    // rank inside the keys that match the prefix so far
                for (int sh = 24; sh >= 0; sh -= 8) {
                    if (tid < 256) sHist[tid] = 0;
                    __syncthreads();
                    for (int k = b + tid; k < e; k += SAH_NT) {
                        const unsigned key = (unsigned)f2ord(cen[3 * (size_t)src[k] + axis]) ^ 0x80000000u;      // unsigned order
                        if ((key & pmask) == prefix) atomicAdd(&sHist[(key >> sh) & 255u], 1);
                    }
                    __syncthreads();
                    if (tid == 0) {
                        int acc = 0, dsel = 255;
                        for (int d = 0; d < 256; d++) { if (acc + sHist[d] > want) { dsel = d; break; } acc += sHist[d]; }
                        sSel[1] = dsel; sSel[2] = acc;
                    }
                    __syncthreads();
                    prefix |= (unsigned)sSel[1] << sh; pmask |= 255u << sh;
                    want -= sSel[2];
                    __syncthreads();
                }
                // prefix = the median's key; `want` of the keys equal to it go left (with all smaller ones): n / 2 on the left
                if (tid == 0) { sSel[0] = 1; sSel[1] = axis; sSel[2] = (int)prefix; sSel[3] = n / 2; }
                sWave[SAH_NT / 64] = want;          // (every thread writes the same value)
            }
        }
        __syncthreads();
        const int mode = sSel[0], sax = sSel[1], sval = sSel[2], nl = sSel[3], nr = n - nl;
        const int eq_quota = sWave[SAH_NT / 64];
        const int m = b + nl;
        __syncthreads();
        // ---- 4. partition into the other order buffer (stretches that are final -- children of one or two patches -- also into order_out)
        // (a big node only publishes its split: k_sah_big_partition moves its patches, one workgroup per chunk)
        if (PRE && tid == 0) {
            int* G = gsel + 8 * j;
            G[0] = mode; G[1] = sax; G[2] = sval; G[3] = nl; G[4] = eq_quota; G[5] = 0; G[6] = 0; G[7] = 0;
        }
        int offL = 0, offR = 0, offE = 0;
        for (int base = b; !PRE && base < e; base += SAH_NT) {
            const int k = base + tid;
            const bool in = k < e;
            int t = 0;
            bool goes_left = false, is_eq = false;
            if (in) {
                t = src[k];
                if (mode == 0) goes_left = max(0, min(NB - 1, (int)((cen[3 * (size_t)t + sax] - clo[sax]) * scale[sax]))) <= sval;
                else if (mode == 1) {
                    const unsigned key = (unsigned)f2ord(cen[3 * (size_t)t + sax]) ^ 0x80000000u;
                    goes_left = key < (unsigned)sval; is_eq = key == (unsigned)sval;
                } else goes_left = (k - b) < nl;
            }
            if (mode == 1) {
                int totE;
                const int re = block_rank<SAH_NT>(is_eq, sWave, totE);
                if (is_eq) goes_left = (offE + re) < eq_quota;
                offE += totE;
            }
            int totL, totR;
            const int rl = block_rank<SAH_NT>(in && goes_left, sWave, totL);
            const int rr = block_rank<SAH_NT>(in && !goes_left, sWave, totR);
            if (in) {
                const int at = goes_left ? b + offL + rl : m + offR + rr;
                dst[at] = t;
                if ((goes_left ? nl : nr) <= 2) order_out[at] = t;
            }
            offL += totL; offR += totR;
        }
        // ---- 5. the node's record; its children: closed here (one or two patches) or queued
        if (tid == 0) {
            first[J.id] = b; last[J.id] = e - 1;
            const int lc = nl == 1 ? N - 1 + b : J.id + 1;
            const int rc = nr == 1 ? N - 1 + m : J.id + 1 + (nl - 1);
            left[J.id] = lc; right[J.id] = rc; parent[lc] = J.id; parent[rc] = J.id;
            for (int side = 0; side < 2; side++) {
                const int cb = side ? m : b, ce = side ? e : m, cid = side ? rc : lc, cn = ce - cb;
                if (cn == 2) {
                    first[cid] = cb; last[cid] = cb + 1;
                    left[cid] = N - 1 + cb; right[cid] = N - 1 + cb + 1;
                    parent[N - 1 + cb] = cid; parent[N - 1 + cb + 1] = cid;
                } else if (cn >= 3) sah_push(Q, cb, ce, cid);
            }
        }
        __syncthreads();
    }
}

// the first job (the root, when it has three patches or more) into the queue of its size class, and the trees of one and two patches
__global__ void k_sah_seed(int N, SahQueues Q, int* __restrict__ order_out, int* __restrict__ left, int* __restrict__ right, int* __restrict__ first,
                           int* __restrict__ last, int* __restrict__ parent) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    parent[0] = -1;
    if (N >= 3) sah_push(Q, 0, N, 0);
    else {
        for (int k = 0; k < N; k++) order_out[k] = k;
        if (N == 2) { first[0] = 0; last[0] = 1; left[0] = 1; right[0] = 2; parent[1] = 0; parent[2] = 0; }
    }
}

// order_out, left, right, first, last, parent: the arrays k_hierarchy writes for the Morton tree (device pointers).
// Per level two launches: the open nodes of SAH_BIG patches or more on 1024-thread workgroups (few nodes, long stretches),
// the others on 256-thread ones; every level has its own pair of queue counters (zeroed once), so nothing is reset between
// levels, and the host looks once per eight levels whether nodes are still open.
static hipError_t sah_hierarchy_device(hipStream_t st, int N, const TriRec* tri, const TreeOptions& topt, int* order_out, int* left, int* right,
                                       int* first, int* last, int* parent) {
    hipError_t e = hipSuccess;
    float* cen = nullptr; double* dmean = nullptr; int* ord = nullptr; SahJob* q = nullptr; int* cnt = nullptr; int* gb = nullptr;
    char* arena = nullptr;
    constexpr int MAXL = 1024;                                  // levels (the 1/32 rule and the median keep real trees below ~50)
    const size_t qcap = (size_t)N / 3 + 2, qcap_big = (size_t)N / SAH_BIG + 2;
    const int NB = std::min(std::max(topt.sah_bins, 2), SAH_NBMAX);
#define DR_TRY(x) do { e = (x); if (e != hipSuccess) goto done; } while (0)
    {
        size_t off = 0;
        auto take = [&off](size_t bytes) { const size_t at = off; off += (bytes + 255) & ~(size_t)255; return at; };
        const size_t o_cnt = take(sizeof(int) * 3 * (MAXL + 1));
        const size_t o_cen = take(sizeof(float) * 3 * (size_t)N), o_dm = take(sizeof(double) * (1 + SAH_PREP_BLOCKS));
        const size_t o_ord = take(sizeof(int) * 2 * (size_t)N), o_q = take(sizeof(SahJob) * (4 * qcap + 2 * qcap_big));
        const size_t o_gb = take(sizeof(int) * qcap_big * (6 + SAH_GB + 8));
        DR_TRY(hipMalloc(&arena, off));
        cnt = (int*)(arena + o_cnt); cen = (float*)(arena + o_cen); dmean = (double*)(arena + o_dm); ord = (int*)(arena + o_ord);
        q = (SahJob*)(arena + o_q); gb = (int*)(arena + o_gb);
        DR_TRY(hipMemsetAsync(cnt, 0, sizeof(int) * 3 * (MAXL + 1), st));
    }
    {
        int* gcb = gb; int* gbins = gb + qcap_big * 6; int* gsel = gbins + qcap_big * SAH_GB;
        // blocks of the chunk kernels: every big job has at most n / SAH_CHUNK + 1 chunks, all of them together N / SAH_CHUNK + jobs
        const int grid_chunks = (int)((size_t)N / SAH_CHUNK + qcap_big);
        const int grid_reset = (int)((qcap_big * SAH_GB + 255) / 256);
        int* cnt_tiny = cnt; int* cnt_small = cnt + (MAXL + 1); int* cnt_big = cnt + 2 * (MAXL + 1);
        SahJob* q_tiny[2] = { q, q + qcap };
        SahJob* q_small[2] = { q + 2 * qcap, q + 3 * qcap };
        SahJob* q_big[2] = { q + 4 * qcap, q + 4 * qcap + qcap_big };
        const int tiny_max = NB <= SAH_WNB ? 64 : 0;
        auto queues = [&](int buf, int level) { return SahQueues{ q_tiny[buf], cnt_tiny + level, q_small[buf], cnt_small + level, q_big[buf], cnt_big + level, tiny_max }; };
        hipLaunchKernelGGL(k_sah_prepare, dim3(SAH_PREP_BLOCKS), dim3(256), 0, st, N, tri, cen, dmean, ord);
        DR_TRY(hipGetLastError());
        hipLaunchKernelGGL(k_sah_diag_mean, dim3(1), dim3(64), 0, st, N, dmean);
        DR_TRY(hipGetLastError());
        hipLaunchKernelGGL(k_sah_seed, dim3(1), dim3(64), 0, st, N, queues(0, 0), order_out, left, right, first, last, parent);
        DR_TRY(hipGetLastError());
        bool big_open = N >= SAH_BIG, small_open = true;
        int next_look = 2;
        while ((1ll << next_look) < N) next_look++;              // ceil(log2 N): closing nodes of two patches takes log2(N) - 1 levels at least
        int next_look_big = 0;
        while (((long long)SAH_BIG << next_look_big) <= N) next_look_big++;       // levels of even splits until no node has SAH_BIG patches
        const int grid_small = (int)std::min<size_t>(qcap, 4096), grid_big = (int)std::min<size_t>(qcap_big, 256);
        const int grid_tiny = (int)std::min<size_t>((qcap + 3) / 4, 4096);
        for (int level = 0; N >= 3; level++) {
            if (level >= MAXL) { e = hipErrorUnknown; goto done; }      // (cannot happen: every split leaves both sides non-empty)
            const int in = level & 1, out = in ^ 1;
            const SahQueues Qo = queues(out, level + 1);
            const int* src = ord + (size_t)in * N;
            int* dst = ord + (size_t)out * N;
            if (big_open) {
                hipLaunchKernelGGL(k_sah_big_reset, dim3(grid_reset), dim3(256), 0, st, (int)qcap_big, gcb, gbins);
                DR_TRY(hipGetLastError());
                hipLaunchKernelGGL(k_sah_big_bounds, dim3(grid_chunks), dim3(256), 0, st, cen, q_big[in], cnt_big + level, src, gcb);
                DR_TRY(hipGetLastError());
                hipLaunchKernelGGL(k_sah_big_bins, dim3(grid_chunks), dim3(256), 0, st, tri, cen, NB, q_big[in], cnt_big + level, src, gcb, gbins);
                DR_TRY(hipGetLastError());
                hipLaunchKernelGGL((k_sah_level<1024, true>), dim3(grid_big), dim3(1024), 0, st, N, tri, cen, dmean, topt.sah_dilate, NB, q_big[in], cnt_big + level,
                                   Qo, src, dst, order_out, left, right, first, last, parent, gcb, gbins, gsel);
                DR_TRY(hipGetLastError());
                hipLaunchKernelGGL(k_sah_big_partition, dim3(grid_chunks), dim3(256), 0, st, cen, NB, q_big[in], cnt_big + level, src, dst, order_out, gcb, gsel);
                DR_TRY(hipGetLastError());
            }
            if (small_open) {
                hipLaunchKernelGGL((k_sah_level<256, false>), dim3(grid_small), dim3(256), 0, st, N, tri, cen, dmean, topt.sah_dilate, NB, q_small[in], cnt_small + level,
                                   Qo, src, dst, order_out, left, right, first, last, parent, nullptr, nullptr, nullptr);
                DR_TRY(hipGetLastError());
            }
            if (tiny_max > 0) {
                hipLaunchKernelGGL(k_sah_level_wave, dim3(grid_tiny), dim3(256), 0, st, N, tri, cen, dmean, topt.sah_dilate, NB, q_tiny[in], cnt_tiny + level,
                                   Qo, src, dst, order_out, left, right, first, last, parent);
                DR_TRY(hipGetLastError());
            }
            if (big_open && level == next_look_big && level != next_look) {
                // are big nodes still open?  (each level about halves them: first asked when even splits would have ended them;
                // their five launches per level are worth a look)
                int open_big = 0;
                DR_TRY(hipMemcpyAsync(&open_big, cnt_big + level + 1, sizeof(int), hipMemcpyDeviceToHost, st));
                DR_TRY(hipStreamSynchronize(st));
                big_open = open_big != 0;
                next_look_big += 2;
            }
            if (level == next_look) {         // which classes of nodes are still open?  (first when a tree over N leaves can be done)
                next_look += 4;
                int open[3] = { 0, 0, 0 };
                DR_TRY(hipMemcpyAsync(&open[0], cnt_tiny + level + 1, sizeof(int), hipMemcpyDeviceToHost, st));
                DR_TRY(hipMemcpyAsync(&open[1], cnt_small + level + 1, sizeof(int), hipMemcpyDeviceToHost, st));
                DR_TRY(hipMemcpyAsync(&open[2], cnt_big + level + 1, sizeof(int), hipMemcpyDeviceToHost, st));
                DR_TRY(hipStreamSynchronize(st));
                // (children are smaller than their parent: a class that is empty, with none above it open, stays empty)
                big_open = open[2] != 0;
                small_open = big_open || open[1] != 0;
                if (open[0] == 0 && open[1] == 0 && open[2] == 0) break;
            }
        }
    }
    DR_TRY(hipStreamSynchronize(st));
#undef DR_TRY
done:
    (void)hipFree(arena);
    return e;
}

// ---------------------------------------------------------------------------------------
// The REFERENCE builder of the same topology on the host (dr_options::sah_on_host; tests compare the two): a top-down binned surface-area heuristic (one thread: a data-dependent
// recursion over at most a few hundred thousand boxes, tens of milliseconds), in the arrays k_hierarchy would have written
// (internal nodes 0 .. N-2 with 0 the root, leaf k = N-1+k for position k of the leaf order); bounds, collapsing into leaves,
// the pre-order threaded layout and the path records stay on the device (k_refit, k_emit, k_paths).  Any binary tree over
// the triangles gives bit-identical results -- what is hit is decided per triangle -- the tree only decides how many nodes a
// walk visits: the Morton tree's top nodes are octants of the scene (a room's corner: three walls in one fat box that every
// ray through the interior enters), this one's are thin sheets.
// ---------------------------------------------------------------------------------------
static void sah_hierarchy_host(int N, const TriRec* T, std::vector<int>& order, std::vector<int>& left, std::vector<int>& right,
                               std::vector<int>& first, std::vector<int>& last, std::vector<int>& parent, const TreeOptions& topt) {
    constexpr int NBMAX = 128;
    const int NB = std::min(std::max(topt.sah_bins, 2), NBMAX);
    order.resize(N); for (int i = 0; i < N; i++) order[i] = i;
    left.assign(N > 1 ? N - 1 : 1, 0); right = left; first = left; last = left;
    parent.assign(2 * (size_t)N - 1, -1);
    if (N < 2) return;
    std::vector<float> cen(3 * (size_t)N);
    for (int i = 0; i < N; i++) for (int a = 0; a < 3; a++) cen[3 * (size_t)i + a] = 0.5f * T[i].lo[a] + 0.5f * T[i].hi[a];
    // the "rays" are bundles (a pair's 50 rays: about one patch wide at both ends): a bundle meets a box with a probability
    // that goes with the area of the box grown by the bundle's radius -- DR_SAH_DILATE x the mean patch-box diagonal
    double dsum = 0.0;
    for (int i = 0; i < N; i++) { double q = 0; for (int a = 0; a < 3; a++) { const double d = (double)T[i].hi[a] - T[i].lo[a]; q += d * d; } dsum += std::sqrt(q); }
    const float grow = topt.sah_dilate * (float)(dsum / N);      // (0 -> 0.5: 78.7 -> 77.6 node visits per pair)
    auto area = [grow](const float* lo, const float* hi) {
        const float dx = std::max(hi[0] - lo[0], 0.0f) + grow, dy = std::max(hi[1] - lo[1], 0.0f) + grow, dz = std::max(hi[2] - lo[2], 0.0f) + grow;
        return dx * dy + dy * dz + dz * dx;
    };
    // A node over n leaves owns the block of n - 1 interior ids [id, id + n - 1): itself, then its left subtree's block, then its
    // right subtree's -- ids, ranges and the leaf order of a subtree depend on nothing outside it, so subtrees can be built by
    // different threads and the result is the same tree whatever the schedule.
    struct Job { int b, e, id; };
    // splits the node `j` (partitions order[b, e)), writes its record, returns the position of the cut
    auto split_node = [&](const Job& j) -> int {
        const int n = j.e - j.b;
        int m = j.b + n / 2;
        if (n > 2) {
            float clo[3] = { INFINITY, INFINITY, INFINITY }, chi[3] = { -INFINITY, -INFINITY, -INFINITY };
            for (int k = j.b; k < j.e; k++) for (int a = 0; a < 3; a++) {
                const float c = cen[3 * (size_t)order[k] + a]; clo[a] = std::min(clo[a], c); chi[a] = std::max(chi[a], c);
            }
            float best = INFINITY; int best_axis = -1, best_bin = 0;
            for (int a = 0; a < 3; a++) {
                const float ext = chi[a] - clo[a];
                const float scale = (float)NB / ext;
                // a degenerate axis: no extent, or one so small (denormal) that NB / ext overflows -- (cen - clo) * inf would be
                // 0 * inf = NaN for the lowest centroid and its conversion to int undefined
                if (!(ext > 0.0f) || !std::isfinite(scale)) continue;
                int cnt[NBMAX]; float blo[NBMAX][3], bhi[NBMAX][3];
                for (int q = 0; q < NB; q++) { cnt[q] = 0; for (int d = 0; d < 3; d++) { blo[q][d] = INFINITY; bhi[q][d] = -INFINITY; } }
                for (int k = j.b; k < j.e; k++) {
                    const int t = order[k];
                    const int q = std::max(0, std::min(NB - 1, (int)((cen[3 * (size_t)t + a] - clo[a]) * scale)));
                    cnt[q]++;
                    for (int d = 0; d < 3; d++) { blo[q][d] = std::min(blo[q][d], T[t].lo[d]); bhi[q][d] = std::max(bhi[q][d], T[t].hi[d]); }
                }
                // right-to-left suffix boxes, then left-to-right prefix sweep
                float rarea[NBMAX]; int rcnt[NBMAX];
                { float lo3[3] = { INFINITY, INFINITY, INFINITY }, hi3[3] = { -INFINITY, -INFINITY, -INFINITY }; int c = 0;
                  for (int q = NB - 1; q >= 1; q--) {
                      for (int d = 0; d < 3; d++) { lo3[d] = std::min(lo3[d], blo[q][d]); hi3[d] = std::max(hi3[d], bhi[q][d]); }
                      c += cnt[q]; rcnt[q] = c; rarea[q] = c ? area(lo3, hi3) : 0.0f;
                  } }
                float lo3[3] = { INFINITY, INFINITY, INFINITY }, hi3[3] = { -INFINITY, -INFINITY, -INFINITY }; int c = 0;
                for (int q = 0; q < NB - 1; q++) {                 // split between bin q and q + 1
                    for (int d = 0; d < 3; d++) { lo3[d] = std::min(lo3[d], blo[q][d]); hi3[d] = std::max(hi3[d], bhi[q][d]); }
                    c += cnt[q];
                    if (c == 0 || rcnt[q + 1] == 0) continue;
                    if (n > 64 && (long long)std::min(c, rcnt[q + 1]) * 32 < n) continue;   // (keeps the depth, and the build's cost, logarithmic)
                    const float cost = area(lo3, hi3) * (float)c + rarea[q + 1] * (float)rcnt[q + 1];
                    if (cost < best) { best = cost; best_axis = a; best_bin = q; }
                }
            }
            if (best_axis >= 0) {
                const int a = best_axis; const float scale = (float)NB / (chi[a] - clo[a]); const float c0 = clo[a];
                int* mid = std::partition(order.data() + j.b, order.data() + j.e, [&](int t) {
                    return std::max(0, std::min(NB - 1, (int)((cen[3 * (size_t)t + a] - c0) * scale))) <= best_bin; });
                m = (int)(mid - order.data());
                if (m <= j.b || m >= j.e) m = j.b + n / 2;           // (cannot happen: both sides were counted non-empty)
            } else {
                // no admissible plane (all centroids in one bin, or only lopsided cuts): the median along the longest axis
                int a = 0;
                for (int d = 1; d < 3; d++) if (chi[d] - clo[d] > chi[a] - clo[a]) a = d;
                if (chi[a] - clo[a] > 0.0f)
                    std::nth_element(order.begin() + j.b, order.begin() + m, order.begin() + j.e,
                                     [&](int x, int y) { return cen[3 * (size_t)x + a] < cen[3 * (size_t)y + a]; });
            }
        }
        first[j.id] = j.b; last[j.id] = j.e - 1;
        const int nl = m - j.b, nr = j.e - m;
        const int lc = nl == 1 ? N - 1 + j.b : j.id + 1;
        const int rc = nr == 1 ? N - 1 + m : j.id + 1 + (nl - 1);
        left[j.id] = lc; right[j.id] = rc; parent[lc] = j.id; parent[rc] = j.id;
        return m;
    };
    auto children = [&](const Job& j, int m, std::vector<Job>& out) {
        if (m - j.b > 1) out.push_back({ j.b, m, j.id + 1 });
        if (j.e - m > 1) out.push_back({ m, j.e, j.id + 1 + (m - j.b - 1) });
    };
    auto build_subtree = [&](Job root) {
        std::vector<Job> stack; stack.push_back(root);
        while (!stack.empty()) { const Job j = stack.back(); stack.pop_back(); children(j, split_node(j), stack); }
    };
    // the top of the tree on this thread until there are enough subtrees to hand out, then one subtree per task
    int nthreads = (int)std::min(8u, std::max(1u, std::thread::hardware_concurrency()));
    if (topt.sah_host_threads > 0) nthreads = std::min(topt.sah_host_threads, 64);
    if (N < 2048) nthreads = 1;
    std::vector<Job> jobs; jobs.push_back({ 0, N, 0 });
    while (nthreads > 1 && (int)jobs.size() < 8 * nthreads) {
        // split the largest open subtree
        size_t big = 0;
        for (size_t q = 1; q < jobs.size(); q++) if (jobs[q].e - jobs[q].b > jobs[big].e - jobs[big].b) big = q;
        if (jobs[big].e - jobs[big].b < 1024) break;
        const Job j = jobs[big]; jobs.erase(jobs.begin() + (long)big);
        children(j, split_node(j), jobs);
    }
    if (nthreads == 1) { for (const Job& j : jobs) build_subtree(j); }
    else {
        std::sort(jobs.begin(), jobs.end(), [](const Job& x, const Job& y) { return x.e - x.b > y.e - y.b; });     // big ones first
        std::atomic<size_t> next(0);
        std::vector<std::thread> pool;
        for (int t = 0; t < nthreads; t++)
            pool.emplace_back([&]() { for (size_t q; (q = next.fetch_add(1)) < jobs.size();) build_subtree(jobs[q]); });
        for (std::thread& th : pool) th.join();
    }
    parent[0] = -1;
}

// tests: the DEVICE builder on bare boxes (host arrays in and out)
hipError_t sah_topology_from_boxes_device(hipStream_t st, int N, const float* boxes, SahTopology& out, const TreeOptions& topt) {
    std::vector<TriRec> T((size_t)std::max(N, 0));
    for (int i = 0; i < N; i++) {
        std::memset(&T[i], 0, sizeof(TriRec));
        for (int a = 0; a < 3; a++) { T[i].lo[a] = boxes[6 * (size_t)i + a]; T[i].hi[a] = boxes[6 * (size_t)i + 3 + a]; }
    }
    const size_t nn = 2 * (size_t)N - 1;
    TriRec* d_tri = nullptr; int* d_i = nullptr;
    hipError_t e = hipMalloc(&d_tri, sizeof(TriRec) * (size_t)N);
    if (e == hipSuccess) e = hipMalloc(&d_i, sizeof(int) * (5 * (size_t)N + nn));
    if (e == hipSuccess) e = hipMemcpyAsync(d_tri, T.data(), sizeof(TriRec) * (size_t)N, hipMemcpyHostToDevice, st);
    int *order = d_i, *left = d_i + N, *right = d_i + 2 * (size_t)N, *first = d_i + 3 * (size_t)N, *last = d_i + 4 * (size_t)N, *parent = d_i + 5 * (size_t)N;
    if (e == hipSuccess) e = hipMemsetAsync(d_i, 0xff, sizeof(int) * (5 * (size_t)N + nn), st);
    if (e == hipSuccess) e = sah_hierarchy_device(st, N, d_tri, topt, order, left, right, first, last, parent);
    out.order.assign((size_t)N, 0); out.left.assign((size_t)std::max(N - 1, 1), 0); out.right = out.left; out.first = out.left; out.last = out.left;
    out.parent.assign(nn, -1);
    auto get = [&](std::vector<int>& v, const int* src, size_t n) { if (e == hipSuccess && n) e = hipMemcpyAsync(v.data(), src, sizeof(int) * n, hipMemcpyDeviceToHost, st); };
    get(out.order, order, (size_t)N); get(out.left, left, (size_t)std::max(N - 1, 0)); get(out.right, right, (size_t)std::max(N - 1, 0));
    get(out.first, first, (size_t)std::max(N - 1, 0)); get(out.last, last, (size_t)std::max(N - 1, 0)); get(out.parent, parent, nn);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    (void)hipFree(d_tri); (void)hipFree(d_i);
    out.N = N;
    return e;
}

void sah_topology_from_boxes(int N, const float* boxes, SahTopology& out, const TreeOptions& topt) {
    std::vector<TriRec> T((size_t)std::max(N, 0));
    for (int i = 0; i < N; i++) {
        std::memset(&T[i], 0, sizeof(TriRec));
        for (int a = 0; a < 3; a++) { T[i].lo[a] = boxes[6 * (size_t)i + a]; T[i].hi[a] = boxes[6 * (size_t)i + 3 + a]; }
    }
    sah_hierarchy_host(N, T.data(), out.order, out.left, out.right, out.first, out.last, out.parent, topt);
    out.N = N;
}

__global__ void k_sentinel(const int* __restrict__ esize, BvhNode* __restrict__ nodes, BvhNode* __restrict__ nodes_lh) {
    if (threadIdx.x != 0) return;
    const int n = esize[0];
    BvhNode endn;
    for (int a = 0; a < 3; a++) { endn.c[a] = 0.0f; endn.h[a] = INFINITY; }
    endn.skip = (n + 1) * (int)sizeof(BvhNode);
    endn.tri = BVH_END;
    nodes[n] = endn;
    for (int a = 0; a < 3; a++) { endn.c[a] = -INFINITY; endn.h[a] = INFINITY; }          // lower / upper corner form
    nodes_lh[n] = endn;
}

hipError_t build_lbvh(hipStream_t st, int N, const TriRec* tri, const float slo[3], const float shi[3], float node_pad,
                      BvhNode* nodes, BvhNode* nodes_lh, TriRec* tri_sorted, int* n_nodes_out, BvhNode* path_rec, PathHdr* path_hdr,
                      const TreeOptions& topt, SahTopology* shared, BvhPair* pairs, BvhPair* pairs_lh, int* depth_out) {
    hipError_t e;
    unsigned long long *keys = nullptr, *keys2 = nullptr;
    int *vals = nullptr, *vals2 = nullptr, *ibuf = nullptr;
    float* box = nullptr;
    int* wbuf = nullptr;           // the refit's marks of written leaves (N) and their exclusive prefix sums (N + 1)
    void* tmp = nullptr;
    void* tmp2 = nullptr;
    char* arena = nullptr;         // every temporary of the build in ONE allocation (an allocation costs tens of microseconds)
    size_t tmp_bytes = 0, tmp2_bytes = 0;
    const size_t nn = 2 * (size_t)N - 1;
    const size_t nblk = ((size_t)N + 63) / 64, nsb = (nblk + 63) / 64;      // the refit's 64-leaf blocks and blocks of 64 blocks
    float3 lo3 = make_float3(slo[0], slo[1], slo[2]);
    float3 inv3 = make_float3(shi[0] > slo[0] ? 1.0f / (shi[0] - slo[0]) : 0.0f,
                              shi[1] > slo[1] ? 1.0f / (shi[1] - slo[1]) : 0.0f,
                              shi[2] > slo[2] ? 1.0f / (shi[2] - slo[2]) : 0.0f);
#define DR_TRY(x) do { e = (x); if (e != hipSuccess) goto done; } while (0)
    {
        // sizes of the library calls' scratch (host-only queries), then the arena's layout
        if (!topt.sah) DR_TRY(rocprim::radix_sort_pairs(nullptr, tmp_bytes, keys, keys2, vals, vals2, (unsigned int)N, 0, 64, st));
        DR_TRY(rocprim::exclusive_scan(nullptr, tmp2_bytes, wbuf, wbuf, 0, (size_t)N + 1, rocprim::plus<int>(), st));
        size_t off = 0;
        auto take = [&off](size_t bytes) { const size_t at = off; off += (bytes + 255) & ~(size_t)255; return at; };
        const size_t o_keys = take(sizeof(unsigned long long) * N), o_keys2 = take(sizeof(unsigned long long) * N);
        const size_t o_vals = take(sizeof(int) * N), o_vals2 = take(sizeof(int) * N);
        // left,right,first,last: N-1 each (+ one spare block of N); parent, esize, pre, items: 2N-1 each; pos: N
        const size_t o_ibuf = take(sizeof(int) * (6 * (size_t)N + 4 * nn + 1));
        // boxes of all nodes, then the refit's tables: pre64, suf64 (N each), blk, pre_b, suf_b (nblk each), sblk (nsb) -- 6 floats per entry
        const size_t o_box = take(sizeof(float) * 6 * (nn + 2 * (size_t)N + 3 * nblk + nsb));
        const size_t o_wbuf = take(sizeof(int) * (2 * (size_t)N + 2));
        const size_t o_tmp = take(tmp_bytes ? tmp_bytes : 16), o_tmp2 = take(tmp2_bytes ? tmp2_bytes : 16);
        DR_TRY(hipMalloc(&arena, off));
        keys = (unsigned long long*)(arena + o_keys); keys2 = (unsigned long long*)(arena + o_keys2);
        vals = (int*)(arena + o_vals); vals2 = (int*)(arena + o_vals2); ibuf = (int*)(arena + o_ibuf);
        box = (float*)(arena + o_box); wbuf = (int*)(arena + o_wbuf); tmp = arena + o_tmp; tmp2 = arena + o_tmp2;
    }
    {
        int* left = ibuf; int* right = ibuf + N; int* first = ibuf + 2 * (size_t)N;
        int* last = ibuf + 3 * (size_t)N; int* parent = ibuf + 5 * (size_t)N;
        int* esize = parent + nn;
        int* pre = esize + nn;
        int* pos = pre + nn;
        int* depth_max = pos + N;
        int* items = depth_max + 1;
        const int nb = (N + 255) / 256;
        const int key_mode = topt.morton_key;
        // which tree: the SAH topology pays from a few thousand patches up (64k: -7 % assembly time; the reference's own
        // scenes, 6400 and 7712 patches: -5 % and break-even), below that the Morton tree; the caller (dr_options::tree) decides
        const bool sah = topt.sah;
        if (sah && !topt.sah_on_host) {
            e = sah_hierarchy_device(st, N, tri, topt, vals2, left, right, first, last, parent);
            if (e != hipSuccess) goto done;
        } else if (sah) {
            SahTopology local;
            SahTopology& T = shared ? *shared : local;
            if (T.N != N) {
                std::vector<TriRec> h_tri((size_t)N);
                DR_TRY(hipMemcpyAsync(h_tri.data(), tri, sizeof(TriRec) * (size_t)N, hipMemcpyDeviceToHost, st));
                DR_TRY(hipStreamSynchronize(st));
                sah_hierarchy_host(N, h_tri.data(), T.order, T.left, T.right, T.first, T.last, T.parent, topt);
                T.N = N;
            }
            const std::vector<int>&h_order = T.order, &h_left = T.left, &h_right = T.right, &h_first = T.first, &h_last = T.last, &h_parent = T.parent;
            DR_TRY(hipMemcpyAsync(vals2, h_order.data(), sizeof(int) * (size_t)N, hipMemcpyHostToDevice, st));
            if (N > 1) {
                DR_TRY(hipMemcpyAsync(left, h_left.data(), sizeof(int) * (size_t)(N - 1), hipMemcpyHostToDevice, st));
                DR_TRY(hipMemcpyAsync(right, h_right.data(), sizeof(int) * (size_t)(N - 1), hipMemcpyHostToDevice, st));
                DR_TRY(hipMemcpyAsync(first, h_first.data(), sizeof(int) * (size_t)(N - 1), hipMemcpyHostToDevice, st));
                DR_TRY(hipMemcpyAsync(last, h_last.data(), sizeof(int) * (size_t)(N - 1), hipMemcpyHostToDevice, st));
            }
            DR_TRY(hipMemcpyAsync(parent, h_parent.data(), sizeof(int) * nn, hipMemcpyHostToDevice, st));
            DR_TRY(hipStreamSynchronize(st));               // (the host vectors go out of scope)
        } else {
            hipLaunchKernelGGL(k_morton, dim3(nb), dim3(256), 0, st, N, tri, lo3, inv3, key_mode, keys, vals);
            DR_TRY(hipGetLastError());
            DR_TRY(rocprim::radix_sort_pairs(tmp, tmp_bytes, keys, keys2, vals, vals2, (unsigned int)N, 0, 64, st));
        }
        if (N > 1 && !sah) {
            hipLaunchKernelGGL(k_hierarchy, dim3(nb), dim3(256), 0, st, N, keys2, left, right, first, last, parent);
            DR_TRY(hipGetLastError());
        }
        {
            float* pre64 = box + 6 * nn; float* suf64 = pre64 + 6 * (size_t)N; float* blk = suf64 + 6 * (size_t)N;
            float* pre_b = blk + 6 * nblk; float* suf_b = pre_b + 6 * nblk; float* sblk = suf_b + 6 * nblk;
            int* w = wbuf; int* wsum = wbuf + N;
            hipLaunchKernelGGL(k_refit_leaves, dim3(nb), dim3(256), 0, st, N, tri, vals2, tri_sorted, pos, box, esize, pre64, suf64, blk);
            DR_TRY(hipGetLastError());
            hipLaunchKernelGGL(k_refit_blocks, dim3((int)((nblk + 255) / 256)), dim3(256), 0, st, (int)nblk, blk, pre_b, suf_b, sblk);
            DR_TRY(hipGetLastError());
            DR_TRY(hipMemsetAsync(w, 0, sizeof(int) * (size_t)N, st));
            hipLaunchKernelGGL(k_refit_marks, dim3((int)((nn + 255) / 256)), dim3(256), 0, st, N, first, last, parent, w);
            DR_TRY(hipGetLastError());
            DR_TRY(rocprim::exclusive_scan(tmp2, tmp2_bytes, w, wsum, 0, (size_t)N + 1, rocprim::plus<int>(), st));
            if (N > 1) {
                hipLaunchKernelGGL(k_refit_nodes, dim3(nb), dim3(256), 0, st, N, first, last, wsum, pre64, suf64, blk, pre_b, suf_b, sblk, box, esize);
                DR_TRY(hipGetLastError());
            }
        }
        hipLaunchKernelGGL(k_pad_tris, dim3(1), dim3(64), 0, st, N, tri_sorted);
        DR_TRY(hipGetLastError());
        DR_TRY(hipMemsetAsync(pre, 0xff, sizeof(int) * nn, st));
        DR_TRY(hipMemsetAsync(depth_max, 0, sizeof(int), st));
        hipLaunchKernelGGL(k_emit, dim3((int)((nn + 255) / 256)), dim3(256), 0, st, N, left, first, last, parent, box, esize, node_pad, tri_sorted, nodes, nodes_lh, pre,
                           pairs, pairs_lh, depth_max, items);
        DR_TRY(hipGetLastError());
        if (depth_out) DR_TRY(hipMemcpyAsync(depth_out, depth_max, sizeof(int), hipMemcpyDeviceToHost, st));
        if (path_rec && path_hdr) {
            hipLaunchKernelGGL(k_paths, dim3(nb), dim3(256), 0, st, N, pos, left, right, parent, pre, items, nodes, nodes_lh, path_rec, path_hdr);
            DR_TRY(hipGetLastError());
        }
        // nodes written = size of the root's subtree (a lone triangle is its own root leaf); behind them the sentinel the
        // threaded walk ends on -- every skip that leaves the tree lands on it: an all-space box that every live ray hits and
        // whose leaf code says "end" (walk_range) -- in both box forms
        hipLaunchKernelGGL(k_sentinel, dim3(1), dim3(64), 0, st, esize, nodes, nodes_lh);
        DR_TRY(hipGetLastError());
        DR_TRY(hipMemcpyAsync(n_nodes_out, esize, sizeof(int), hipMemcpyDeviceToHost, st));
        DR_TRY(hipStreamSynchronize(st));
    }
#undef DR_TRY
done:
    (void)hipFree(arena);
    return e;
}

// ---------------------------------------------------------------------------------------
// ray / triangle and ray / box
// ---------------------------------------------------------------------------------------
// Two-sided Moller-Trumbore; a hit needs u>=0, v>=0, u+v<=1, t>0 (the reference accepts
// hit.t > 0, vs/OptixPrimeFunctionality.cpp:208).  det == 0 yields inf/NaN, which fail.
__device__ __forceinline__ bool tri_hit(f3 o, f3 d, f3 a, f3 e1, f3 e2, float& t_out) {
    f3 p = cross3(d, e2);
    float det = dot3(e1, p);
    float inv = 1.0f / det;
    f3 tv = o - a;
    float u = dot3(tv, p) * inv;
    f3 q = cross3(tv, e1);
    float v = dot3(d, q) * inv;
    float t = dot3(e2, q) * inv;
    t_out = t;
    return (u >= 0.0f) && (v >= 0.0f) && ((u + v) <= 1.0f) && (t > 0.0f);
}

// Conservative slab test of the segment [0,tmax] against a (padded) box held in SGPRs, as a wave
// mask (v_cmp straight into an SGPR pair; predicate 5 = ordered <=).  t = (plane - org) * inv: the
// subtraction first -- the cheaper plane*inv - org*inv (one fma per plane) cancels catastrophically for
// rays nearly parallel to a slab (|org*inv| ~ 1e7 loses the whole t range) and then culls true
// blockers; seen on the reference's colorballs scene, whose walls are slightly tilted.
// One compare: max(tn,0) <= min(tf*(1+eps), tmax)  <=>  tn <= tf*(1+eps), tf >= 0, tn <= tmax
// (tmax > 0); NaNs from 0*inf are dropped by min/max.
// Bit-identical to the oracle's slab_hit (same operations, same order): the per-triangle gate is
// part of the definition of a hit, and the same test on enclosing boxes can then never cull a
// triangle whose gate passes.
__device__ __forceinline__ unsigned long long box_hit_mask(const float lo[3], const float hi[3], f3 org, f3 inv, float tmax) {
    float t0 = (lo[0] - org.x) * inv.x, t1 = (hi[0] - org.x) * inv.x;
    float tn = fminf(t0, t1), tf = fmaxf(t0, t1);
    t0 = (lo[1] - org.y) * inv.y; t1 = (hi[1] - org.y) * inv.y;
    tn = fmaxf(tn, fminf(t0, t1)); tf = fminf(tf, fmaxf(t0, t1));
    t0 = (lo[2] - org.z) * inv.z; t1 = (hi[2] - org.z) * inv.z;
    tn = fmaxf(fmaxf(tn, fminf(t0, t1)), 0.0f); tf = fminf(tf, fmaxf(t0, t1));
    return __builtin_amdgcn_fcmpf(tn, fminf(tf * 1.00001f, tmax), 5);
}

__device__ __forceinline__ float safe_inv(float d) {
    // a zero component must not turn (plane - origin) * inv into 0*inf = NaN
    return d == 0.0f ? 3.0e38f : 1.0f / d;
}

// Conservative test of the segment [0,tmax] against a BVH NODE (centre c, half-extent h in SGPRs), as a wave mask.
// A node test only has to be conservative -- what is hit is decided per triangle by its gate (box_hit_mask) and
// Moller-Trumbore -- so it need not be the gate's own arithmetic, only never reject a node one of whose
// triangles' gates accepts.  The general form (this function: the counted STATS build; the hand-written loops use the same
// arithmetic in DR_NODE_TEST / DR_NODE_TEST_X, and a sign-specialised 9-instruction variant on lower / upper corners):
//     tc = fma(c, iv, k)            k = -(org*iv), iv = the ray's 1/d clamped to +-1e18, times the ray's scale (per ray, once)
//     tn = fma(h, -|iv|, tc)        tf = fma(h, |iv|, tc)          per axis, no plane selection, no min/max
//     accept  <=>  max(tn_x, tn_y, tn_z, 0) <= min(tf_x, tf_y, tf_z, tmax)
// Its rounding errors (fma form: up to 7 ulp of (|c|+|org|+h)*|iv|) and the slack the gate test grants
// (tf*1.00001, i.e. a miss by up to 1.1e-5 of the ray length) are both bounded by a DISTANCE times |iv|, so
// they are absorbed by growing every node box once, at build time, by
//     node_pad = 3e-5 * scene diagonal + 4e-6 * max|coordinate|
// (needed: 1.2e-5 D + 1e-6 M; DESIGN.md section 4 has the derivation).  The clamp keeps c*iv and org*iv
// finite for axis-parallel rays (iv = 3e38 from safe_inv); with |iv| = 1e18 the padded slab still spans
// [0,tmax] whenever the origin lies within it.
__device__ __forceinline__ unsigned long long node_hit_mask(const float c[3], const float h[3], f3 iv, f3 k, float tmax) {
    const float tcx = __builtin_fmaf(c[0], iv.x, k.x), tcy = __builtin_fmaf(c[1], iv.y, k.y), tcz = __builtin_fmaf(c[2], iv.z, k.z);
    const float ax = fabsf(iv.x), ay = fabsf(iv.y), az = fabsf(iv.z);
    const float tn = fmaxf(fmaxf(fmaxf(__builtin_fmaf(h[0], -ax, tcx), __builtin_fmaf(h[1], -ay, tcy)), __builtin_fmaf(h[2], -az, tcz)), 0.0f);
    const float tf = fminf(fminf(fminf(__builtin_fmaf(h[0], ax, tcx), __builtin_fmaf(h[1], ay, tcy)), __builtin_fmaf(h[2], az, tcz)), tmax);
    return __builtin_amdgcn_fcmpf(tn, tf, 5);
}

// ---------------------------------------------------------------------------------------
// fused tile kernel
// ---------------------------------------------------------------------------------------
constexpr int REC_STRIDE = 21;      // 20 floats + 1: conflict-free LDS rows
constexpr int VIS_STRIDE = TILE + 4;

__device__ __forceinline__ float stored(float f) { return f > 0.0f ? f : 0.0f; }

// Stored integrand of both directions of the pair (i in the I block, j in the J block) from the
// per-patch records (vs/triangle_math.cpp:49-58 summed as vs/OptixPrimeFunctionality.cpp:153-161).
// The 16 point terms of (i,s)->(j,u) and (j,u)->(i,s) are bitwise equal, so they are formed once
// and added in the two orders the reference uses: fa = F(i->j), fb = F(j->i).
__device__ __forceinline__ void integrand_pair(const float* ri, const float* rj, float& fa, float& fb) {
    const f3 ni = f3{ ri[12], ri[13], ri[14] }, nj = f3{ rj[12], rj[13], rj[14] };
    float ff[4][4];
#pragma unroll
    for (int s = 0; s < 4; s++) {
        const f3 ci = f3{ ri[3 * s], ri[3 * s + 1], ri[3 * s + 2] };
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const f3 cj = f3{ rj[3 * u], rj[3 * u + 1], rj[3 * u + 2] };
            f3 d = cj - ci;
            float len = sqrtf(dot3(d, d));
            f3 dn = d * (1.0f / len);
            float c1 = dot3(ni, dn);
            float c2 = -dot3(nj, dn);
            float v = 0.0f;
            if (c1 > 0.0f && c2 > 0.0f) v = ((c1 * c2) / ((len * len) * DR_PIF)) * (ri[15 + s] * rj[15 + u]);
            ff[s][u] = v;
        }
    }
    float accA = 0.0f, accB = 0.0f;
#pragma unroll
    for (int s = 0; s < 4; s++)
#pragma unroll
        for (int u = 0; u < 4; u++) accA = accA + ff[s][u];     // i's sub-triangles outer
#pragma unroll
    for (int u = 0; u < 4; u++)
#pragma unroll
        for (int s = 0; s < 4; s++) accB = accB + ff[s][u];     // j's sub-triangles outer
    fa = stored(accA / ri[19]);
    fb = stored(accB / rj[19]);
}

// The two node register sets of the hand-written walk.  They are named, not allocated (the s_load_dwordx8 needs an
// aligned tuple and the node test its single dwords), and the kernel's SGPR count is at least the highest one named:
// keep them low -- 256-thread blocks are admitted per CU by the SGPR count in steps of 16 (MI355X_MICROARCH.md,
// "Residency": <= 80 -> 8, <= 96 -> 7, above -> 6).  tests/test_abi_cpu.py checks the built code object's count.
#ifndef DR_WALK_SGPR_BASE_HIGH
#define DR_A_ALL "s[56:63]"
#define DR_B_ALL "s[48:55]"
#define DR_A0 "s56"
#define DR_A1 "s57"
#define DR_A2 "s58"
#define DR_A3 "s59"
#define DR_A4 "s60"
#define DR_A5 "s61"
#define DR_A6 "s62"
#define DR_A7 "s63"
#define DR_B0 "s48"
#define DR_B1 "s49"
#define DR_B2 "s50"
#define DR_B3 "s51"
#define DR_B4 "s52"
#define DR_B5 "s53"
#define DR_B6 "s54"
#define DR_B7 "s55"
#define DR_AB_ALL "s[48:63]"
#define DR_WALK_CLOBBERS "s48", "s49", "s50", "s51", "s52", "s53", "s54", "s55", "s56", "s57", "s58", "s59", "s60", "s61", "s62", "s63"
#else
#define DR_A_ALL "s[88:95]"
#define DR_B_ALL "s[80:87]"
#define DR_A0 "s88"
#define DR_A1 "s89"
#define DR_A2 "s90"
#define DR_A3 "s91"
#define DR_A4 "s92"
#define DR_A5 "s93"
#define DR_A6 "s94"
#define DR_A7 "s95"
#define DR_B0 "s80"
#define DR_B1 "s81"
#define DR_B2 "s82"
#define DR_B3 "s83"
#define DR_B4 "s84"
#define DR_B5 "s85"
#define DR_B6 "s86"
#define DR_B7 "s87"
#define DR_AB_ALL "s[80:95]"
#define DR_WALK_CLOBBERS "s80", "s81", "s82", "s83", "s84", "s85", "s86", "s87", "s88", "s89", "s90", "s91", "s92", "s93", "s94", "s95"
#endif

// The triangles of one leaf against the wave's rays: the lanes (of alive_m) for which one of them precedes the destination
// `hi` -- nearer, or at equal t with a lower id.  leaf = first*8 + (count-1) [+4: both triangles share one gate box].
// The leaf's LEAF_MAX records (64 B each: triangle, id, gate box) are fetched together -- no dependent loads inside the
// leaf.  A hit is gate AND Moller-Trumbore (oracle.c); the conjunction is evaluated cheapest-rejection first: the leaf's NODE
// box was touched, so the triangles' gates nearly always pass (10.5 triangle tests start per pair where 11 are possible),
// whereas 46 % of the tests end at u and 80 % of the rest at v (profiles/r02/assembly_notes.md) -- the gate (24 vector
// instructions) is therefore only formed for the one test in nine that survives both exclusion tests.  Same values in the
// same order for every lane that can hit.
__device__ __forceinline__ unsigned long long leaf_blocked_mask(const TriRec* __restrict__ tri_sorted, int leaf, f3 org, f3 dn, f3 inv,
                                                                float tmax, int hi, unsigned long long alive_m) {
    const int first = leaf >> 3, cnt = (leaf & 3) + 1;
    const bool same_gate = (leaf & 4) != 0;          // both triangles share one gate box (the halves of a quad)
    const v4f* tp = reinterpret_cast<const v4f*>(reinterpret_cast<const char*>(tri_sorted) + (unsigned)first * 64u);
    v4f q[4 * LEAF_MAX];
#pragma unroll
    for (int c = 0; c < 4 * LEAF_MAX; c++) q[c] = tp[c];
    unsigned long long blocked_m = 0ull, gm_first = 0ull;
    bool have_first = false;
#pragma unroll
    for (int c = 0; c < LEAF_MAX; c++) {
        const v4f A = q[4 * c], B = q[4 * c + 1], C3 = q[4 * c + 2], D = q[4 * c + 3];
        if (c >= cnt) continue;                    // (slots past the leaf's count belong to the next leaf)
        const int tk = __builtin_amdgcn_readfirstlane(__float_as_int(C3[1]));
        if (tk == hi) continue;                    // the destination: its t is tmax bit for bit, it never precedes itself
        const f3 ta = f3{ A[0], A[1], A[2] };
        const f3 te1 = f3{ A[3], B[0], B[1] };
        const f3 te2 = f3{ B[2], B[3], C3[0] };
        // tri_hit (same operations, same order, same values) with the division moved behind two exact exclusion tests on the
        // NUMERATORS: u = au * fl(1/det) lies outside [0, 1] for sure when |au| > |det| (1 + 2^-21) (then |u| > 1 after both
        // roundings, or is infinite) or when au and det differ in sign and |au| > 2^-100 |det| (then u < 0 strictly: the product
        // cannot underflow to -0, and au = 0 -- whose u = -0 passes u >= 0 -- is never excluded); the same for v, which a hit also needs in [0, 1] (v >= 0, and u + v <= 1 with u >= 0); then the sign of t (below).  Only when
        // some live lane survives all three is the gate formed, and only when some lane passes that too 1/det (11 instructions) and
        // u, v, u + v, t tested as the definition says.
        const f3 pv = cross3(dn, te2);
        const float det = dot3(te1, pv);
        const f3 tv = org - ta;
        const float au = dot3(tv, pv);
        const float ad = fabsf(det), m_hi = ad * 1.0000005f, m_lo = ad * 7.888609e-31f;       // (1 + 2^-21, 2^-100)
        const unsigned long long out_u = __builtin_amdgcn_fcmpf(fabsf(au), m_hi, 2) |
                                         (__builtin_amdgcn_ballot_w64((__float_as_int(au) ^ __float_as_int(det)) < 0) & __builtin_amdgcn_fcmpf(fabsf(au), m_lo, 2));
        unsigned long long cm = alive_m & ~out_u;
        if (cm == 0ull) continue;
        const f3 qv = cross3(tv, te1);
        const float av = dot3(dn, qv);
        const unsigned long long out_v = __builtin_amdgcn_fcmpf(fabsf(av), m_hi, 2) |
                                         (__builtin_amdgcn_ballot_w64((__float_as_int(av) ^ __float_as_int(det)) < 0) & __builtin_amdgcn_fcmpf(fabsf(av), m_lo, 2));
        cm &= ~out_v;
        if (cm == 0ull) continue;
        // t = tn * fl(1/det) > 0 needs tn and det of one sign and tn != 0 (fl(1/det) has det's sign whatever its rounding, a
        // product of opposite signs is negative or -0, and 0 * anything is 0 or NaN): the third exclusion test, on the numerator
        // the definition forms anyway.  It ends nearly every test that comes this far -- most of those are of the ray's own source
        // patch, whose plane the ray leaves at t = -eps
        const float tn = dot3(te2, qv);
        const unsigned long long out_t = __builtin_amdgcn_ballot_w64((__float_as_int(tn) ^ __float_as_int(det)) < 0) | __builtin_amdgcn_fcmpf(tn, 0.0f, 1);
        cm &= ~out_t;
        if (cm == 0ull) continue;
        // the gate (the second triangle of a quad has the first one's box bit for bit: the same mask)
        unsigned long long gm;
        if (c == 1 && same_gate && have_first) gm = gm_first;
        else {
            const float blo[3] = { C3[2], C3[3], D[0] }, bhi[3] = { D[1], D[2], D[3] };
            gm = box_hit_mask(blo, bhi, org, inv, tmax);
            if (c == 0) { gm_first = gm; have_first = true; }
        }
        cm &= gm;
        if (cm == 0ull) continue;
        const float idet = 1.0f / det;
        const float u = au * idet, v = av * idet;
        const unsigned long long vm = cm & __builtin_amdgcn_fcmpf(u, 0.0f, 3) & __builtin_amdgcn_fcmpf(v, 0.0f, 3) & __builtin_amdgcn_fcmpf(u + v, 1.0f, 5);
        if (vm == 0ull) continue;
        const float tt = tn * idet;
        const unsigned long long hm = vm & __builtin_amdgcn_fcmpf(tt, 0.0f, 2);
        // closest hit is not `hi`: something nearer, or an equal-t hit of lower id
        const unsigned long long bm = hm & (__builtin_amdgcn_fcmpf(tt, tmax, 4) | (tk < hi ? __builtin_amdgcn_fcmpf(tt, tmax, 1) : 0ull));
        blocked_m |= bm;
    }
    return blocked_m;
}

// ---- the node tests of the hand-written walks (walk_range, walk_pairs, stream_path_records below) ----
// The node test in its two forms -- general (X: centre / half-extent) and sign-specialised (SX: lower / upper corner, when all live
// rays of the wave point into one octant: near and far plane of every axis known by name, 6 fused multiply-adds instead of 9).
// EXEC holds the live rays while a walk runs, so the compare's
// VCC needs no s_and with the liveness mask and the branch reads VCCZ; and the ray parameter is in units of the ray's own
// length (iv, k scaled per ray by s <= 1/tmax, see k_ff_tiles), so that the VOP3 `clamp` bit does the two end clamps for free:
//     N = max3(clamp(tn_x), tn_y, tn_z)   = max(tn_x, tn_y, tn_z, 0) unless tn_x > 1, where F <= 1 < ... rejects either way
//     F = clamp(min3(tf_x, tf_y, tf_z))   = min(tf_x, tf_y, tf_z, 1) unless that is negative -- then F = 0 <= N
//     accept  <=>  N < F   (strictly)
// Strictness is what makes the second clamp safe (clamping a far value up to 0 and accepting N <= F would admit every box
// in the quadrant behind the origin): N = F happens only for a box that ends at the origin, begins at the destination or
// is grazed along an edge -- a PADDED box, node_pad away from every triangle gate inside it, and the pad exceeds the test's
// rounding errors 2.5 times over, so no triangle in such a box has a gate that accepts.
// 9 vector instructions (12 for the general form), no scalar one.
#define DR_NODE_TEST_X(CX, CY, CZ, HX, HY, HZ)                                \
                "v_fma_f32 %[t0], " CX ", %[ix], %[kx]\n\t"                   \
                "v_fma_f32 %[t1], " CY ", %[iy], %[ky]\n\t"                   \
                "v_fma_f32 %[t2], " CZ ", %[iz], %[kz]\n\t"                   \
                "v_fma_f32 %[t3], " HX ", -|%[ix]|, %[t0] clamp\n\t"          \
                "v_fma_f32 %[t0], " HX ", |%[ix]|, %[t0]\n\t"                 \
                "v_fma_f32 %[t4], " HY ", -|%[iy]|, %[t1]\n\t"                \
                "v_fma_f32 %[t1], " HY ", |%[iy]|, %[t1]\n\t"                 \
                "v_fma_f32 %[t5], " HZ ", -|%[iz]|, %[t2]\n\t"                \
                "v_fma_f32 %[t2], " HZ ", |%[iz]|, %[t2]\n\t"                 \
                "v_max3_f32 %[t3], %[t3], %[t4], %[t5]\n\t"                   \
                "v_min3_f32 %[t0], %[t0], %[t1], %[t2] clamp\n\t"             \
                "v_cmp_lt_f32_e32 vcc, %[t3], %[t0]\n\t"
#define DR_NODE_TEST_SX(NX, NY, NZ, FX, FY, FZ)                               \
                "v_fma_f32 %[t3], " NX ", %[ix], %[kx] clamp\n\t"             \
                "v_fma_f32 %[t4], " NY ", %[iy], %[ky]\n\t"                   \
                "v_fma_f32 %[t5], " NZ ", %[iz], %[kz]\n\t"                   \
                "v_fma_f32 %[t0], " FX ", %[ix], %[kx]\n\t"                   \
                "v_fma_f32 %[t1], " FY ", %[iy], %[ky]\n\t"                   \
                "v_fma_f32 %[t2], " FZ ", %[iz], %[kz]\n\t"                   \
                "v_max3_f32 %[t3], %[t3], %[t4], %[t5]\n\t"                   \
                "v_min3_f32 %[t0], %[t0], %[t1], %[t2] clamp\n\t"             \
                "v_cmp_lt_f32_e32 vcc, %[t3], %[t0]\n\t"
// The sentinel-terminated walk from the root in nine variants inside ONE asm statement (one set of operands, no control
// flow for the compiler to reason about): V = 0..7 the sign-specialised test on bvh_lh for that octant (bit a of V set:
// the rays point towards -axis a, the near corner's coordinate a is the upper one), "8" the general test on bvh.  Local
// labels are V followed by 1..7.  The statement starts with a three-level bit test of %[oct] that jumps to the variant.
#define DR_SEL(S, LO, HI) DR_SEL_##S(LO, HI)
#define DR_SEL_0(LO, HI) LO
#define DR_SEL_1(LO, HI) HI
// Offsets: set A holds the node at %[off], set B the one at %[off] + 32; the offset register moves by 64 after two nodes
// entered in a row (and on every skip), the loads carry the rest as immediates.
#define DR_WALK_VARIANT(V, BVH, TESTA, TESTB)                                                                           \
                V "0:\n\t"                                                                                              \
                "s_load_dwordx8 " DR_A_ALL ", " BVH ", %[off] offset:0x0\n\t"                                           \
                "s_waitcnt lgkmcnt(0)\n"                                                                                \
                V "1:\n\t"                                                                                              \
                "s_load_dwordx8 " DR_B_ALL ", " BVH ", %[off] offset:0x20\n\t"                                          \
                TESTA                                                                                                   \
                "s_cbranch_vccz " V "3f\n\t"                                                                            \
                "s_cmp_lt_i32 " DR_A7 ", 0\n\t"                                                                         \
                "s_cbranch_scc0 " V "5f\n\t"                                                                            \
                "s_waitcnt lgkmcnt(0)\n"                                                                                \
                V "2:\n\t"                                                                                              \
                "s_load_dwordx8 " DR_A_ALL ", " BVH ", %[off] offset:0x40\n\t"                                          \
                TESTB                                                                                                   \
                "s_cbranch_vccz " V "4f\n\t"                                                                            \
                "s_add_u32 %[off], %[off], 64\n\t"                                                                      \
                "s_cmp_lt_i32 " DR_B7 ", 0\n\t"                                                                         \
                "s_cbranch_scc0 " V "6f\n\t"                                                                            \
                "s_waitcnt lgkmcnt(0)\n\t"                                                                              \
                "s_branch " V "1b\n"                                                                                    \
                V "3:\n\t"                                                                                              \
                "s_mov_b32 %[off], " DR_A6 "\n\t"                                                                       \
                "s_load_dwordx8 " DR_A_ALL ", " BVH ", %[off] offset:0x0\n\t"                                           \
                "s_waitcnt lgkmcnt(0)\n\t"                                                                              \
                "s_branch " V "1b\n"                                                                                    \
                V "4:\n\t"                                                                                              \
                "s_sub_u32 %[off], " DR_B6 ", 32\n\t"                                                                   \
                "s_load_dwordx8 " DR_B_ALL ", " BVH ", %[off] offset:0x20\n\t"                                          \
                "s_waitcnt lgkmcnt(0)\n\t"                                                                              \
                "s_branch " V "2b\n"                                                                                    \
                V "5:\n\t"                                                                                              \
                "s_add_u32 %[off], %[off], 32\n\t"                                                                      \
                "s_mov_b32 %[leaf], " DR_A7 "\n\t"                                                                      \
                "s_branch 99f\n"                                                                                        \
                V "6:\n\t"                                                                                              \
                "s_mov_b32 %[leaf], " DR_B7 "\n\t"                                                                      \
                "s_branch 99f\n"
#define DR_WALK_OCTANT(V, SX, SY, SZ)                                                                                   \
        DR_WALK_VARIANT(V, "%[bvhlh]",                                                                                  \
                DR_NODE_TEST_SX(DR_SEL(SX, DR_A0, DR_A3), DR_SEL(SY, DR_A1, DR_A4), DR_SEL(SZ, DR_A2, DR_A5),            \
                               DR_SEL(SX, DR_A3, DR_A0), DR_SEL(SY, DR_A4, DR_A1), DR_SEL(SZ, DR_A5, DR_A2)),           \
                DR_NODE_TEST_SX(DR_SEL(SX, DR_B0, DR_B3), DR_SEL(SY, DR_B1, DR_B4), DR_SEL(SZ, DR_B2, DR_B5),           \
                               DR_SEL(SX, DR_B3, DR_B0), DR_SEL(SY, DR_B4, DR_B1), DR_SEL(SZ, DR_B5, DR_B2)))
#define DR_WALK_ASM_OCTANTS                                                                                             \
            asm volatile(                                                                                               \
                "s_mov_b64 %[sexec], exec\n\t"                                                                          \
                "s_mov_b64 exec, %[alive]\n\t"                                                                          \
                "s_cmp_gt_u32 %[oct], 7\n\t"                                                                            \
                "s_cbranch_scc1 80f\n\t"                                                                                \
                "s_bitcmp1_b32 %[oct], 2\n\t"                                                                           \
                "s_cbranch_scc1 94f\n\t"                                                                                \
                "s_bitcmp1_b32 %[oct], 1\n\t"                                                                           \
                "s_cbranch_scc1 92f\n\t"                                                                                \
                "s_bitcmp1_b32 %[oct], 0\n\t"                                                                           \
                "s_cbranch_scc1 10f\n\t"                                                                                \
                "s_branch 00f\n"                                                                                        \
                "92:\n\t"                                                                                               \
                "s_bitcmp1_b32 %[oct], 0\n\t"                                                                           \
                "s_cbranch_scc1 30f\n\t"                                                                                \
                "s_branch 20f\n"                                                                                        \
                "94:\n\t"                                                                                               \
                "s_bitcmp1_b32 %[oct], 1\n\t"                                                                           \
                "s_cbranch_scc1 96f\n\t"                                                                                \
                "s_bitcmp1_b32 %[oct], 0\n\t"                                                                           \
                "s_cbranch_scc1 50f\n\t"                                                                                \
                "s_branch 40f\n"                                                                                        \
                "96:\n\t"                                                                                               \
                "s_bitcmp1_b32 %[oct], 0\n\t"                                                                           \
                "s_cbranch_scc1 70f\n\t"                                                                                \
                "s_branch 60f\n"                                                                                        \
                DR_WALK_OCTANT("0", 0, 0, 0)                                                                            \
                DR_WALK_OCTANT("1", 1, 0, 0)                                                                            \
                DR_WALK_OCTANT("2", 0, 1, 0)                                                                            \
                DR_WALK_OCTANT("3", 1, 1, 0)                                                                            \
                DR_WALK_OCTANT("4", 0, 0, 1)                                                                            \
                DR_WALK_OCTANT("5", 1, 0, 1)                                                                            \
                DR_WALK_OCTANT("6", 0, 1, 1)                                                                            \
                DR_WALK_OCTANT("7", 1, 1, 1)                                                                            \
                DR_WALK_VARIANT("8", "%[bvh]", DR_NODE_TEST_X(DR_A0, DR_A1, DR_A2, DR_A3, DR_A4, DR_A5),                \
                                DR_NODE_TEST_X(DR_B0, DR_B1, DR_B2, DR_B3, DR_B4, DR_B5))                               \
                "99:\n\t"                                                                                               \
                "s_mov_b64 exec, %[sexec]\n\t"                                                                          \
                "s_waitcnt lgkmcnt(0)"                                                                                  \
                : [off] "+s"(off), [leaf] "=s"(leaf), [sexec] "=&s"(sexec), [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3),   \
                  [t4] "=&v"(t4), [t5] "=&v"(t5)                                                                        \
                : [bvh] "s"(bvh), [bvhlh] "s"(bvh_lh), [oct] "s"(octant), [alive] "s"(alive_m), [kx] "v"(kk.x), [ky] "v"(kk.y),    \
                  [kz] "v"(kk.z), [ix] "v"(iv.x), [iy] "v"(iv.y), [iz] "v"(iv.z)                                        \
                : DR_WALK_CLOBBERS, "vcc", "scc")
// The walk of the THREADED tree from the root for one wave of rays (segments [0,tmax] from org along dn); returns the
// liveness mask with every lane cleared for which something precedes its destination `hi` (wave-uniform: one pair per wave).
//
// The node index is wave-uniform.  The compiler's lowering of this loop spent ~20 scalar instructions per
// node (the CU's single scalar unit serves all four SIMDs) beside the vector ones, so the interior-node
// walk is written out by hand: one s_load_dwordx8 of the node at an SGPR byte offset, the node test with EXEC = the live
// rays (9 vector instructions when the wave's rays share an octant, 12 otherwise), a branch on VCCZ, then
// either the next node in pre-order (the first child: already loading into the other register set) or offset = skip --
// 4 - 5 scalar instructions per node.  The hand-written stretch ends whenever a hit node is a leaf; skips that leave the tree
// land on the sentinel node (all-space box, leaf code BVH_END), which every live lane hits: no end compare at all.
//
// iv / kk: the per-ray constants of the node test (the ray's 1/d clamped to +-1e18 in units of the ray's length, and -(org*iv)).
// octant: 0..7 = all live rays point into that octant (bit a set: towards -axis a): the sign-specialised test on bvh_lh;
// 8: the general test on bvh.
template <bool STATS>
__device__ __forceinline__ unsigned long long walk_range(const BvhNode* __restrict__ bvh, const TriRec* __restrict__ tri_sorted,
                                                         unsigned off, const unsigned end, f3 org, f3 dn, f3 inv, f3 iv, f3 kk,
                                                         float tmax, float tmax_w, int hi, unsigned long long alive_m, int& n_visit, int& n_leaf,
                                                         const BvhNode* __restrict__ bvh_lh = nullptr, int octant = 8) {
    for (;;) {
        int leaf;
        if (STATS) {
            // counted variant of the same walk (debug builds only)
            leaf = BVH_END;
            while (off < end) {
                const v8f raw = *reinterpret_cast<const v8f*>(reinterpret_cast<const char*>(bvh) + off);
                const float nc[3] = { raw[0], raw[1], raw[2] }, nh[3] = { raw[3], raw[4], raw[5] };
                const unsigned nd_skip = (unsigned)__builtin_amdgcn_readfirstlane(__float_as_int(raw[6]));
                const int nd_leaf = __builtin_amdgcn_readfirstlane(__float_as_int(raw[7]));
                n_visit++;
                const unsigned long long hb_m = node_hit_mask(nc, nh, iv, kk, tmax_w) & alive_m;
                if (hb_m == 0ull) { off = nd_skip; continue; }
                off += 32u;
                if (nd_leaf >= 0) { leaf = nd_leaf; break; }
            }
        } else {
            float t0, t1, t2, t3, t4, t5;
            // two copies of the step, on node registers A and B (8 SGPRs each, fixed: DR_WALK_A/B below): while one node
            // is tested the next one in pre-order (the first child, where a hit descends to) is already being fetched
            // into the other set; a miss reloads its own set from the skip offset
            unsigned long long sexec;
            DR_WALK_ASM_OCTANTS;
        }
        if (leaf == BVH_END) break;          // the tree's sentinel was reached
        if (STATS) n_leaf++;
        const unsigned long long blocked_m = leaf_blocked_mask(tri_sorted, leaf, org, dn, inv, tmax, hi, alive_m);
        alive_m &= ~blocked_m;
        if (alive_m == 0ull) break;
    }
    return alive_m;
}

// ---------------------------------------------------------------------------------------
// The walk over the SIBLING-PAIR form of the tree (BvhPair, dr_internal.h) -- the shipped walk.
//
// The threaded walk above fetches every node it tests, one dependent 32-byte scalar load per node, and fetches a rejected
// node only to learn where to go next (profiles/r02/pmc_asm_64k.txt: 148 scalar loads per pair, 44 % of them L2 round trips,
// a wave 46 % of its time in s_waitcnt).  Here a record holds the boxes of BOTH children of an interior node: one
// s_load_dwordx16 (one 64-byte line) lands in the same sixteen SGPRs the threaded walk uses as its two node sets, the two node
// tests run back to back with nothing to wait for in between, a rejected child costs no fetch at all, and the items still to
// do -- the second of two accepted children -- wait on a stack kept in the LANES OF ONE VGPR: v_writelane_b32 / v_readlane_b32
// with the stack pointer in M0 (both ignore EXEC; no memory, no latency).  Half the fetches per pair for the same node tests.
// An item is a record's byte offset (>= 0) or a leaf (< 0: 0x80000000 | leaf code); a leaf item ends the hand-written stretch
// (leaf_blocked_mask runs as compiled code), the walk resumes by popping.  Same node tests as the threaded walk (DR_NODE_TEST_SX
// per octant, DR_NODE_TEST_X for mixed signs), same grown boxes: equally conservative; what is hit is decided per triangle.
// Stack depth: at most one pending item per level of the tree; the host only selects this walk for trees of depth <= PAIR_STACK - 2.
// The item to do next never leaves the record's registers: the loop has two entries, one for an item in the right child's item
// register (A6: also where a pop lands) and one for the left child's (B6), each loading the next record over the very registers
// that held its offset -- no copy into an offset register and no jump back to a common head (the kernel is bound by the
// instructions it issues, scalar ones included: profiles/r03/assembly_notes.md section 8).
// ---------------------------------------------------------------------------------------
#define DR_PWALK_VARIANT(V, BVH, TESTL, TESTR)                                                                          \
                V "0:\n\t"                                                                                              \
                "s_cmp_eq_u32 m0, 0\n\t"                                                                                \
                "s_cbranch_scc1 98f\n\t"                                                                                \
                "s_add_u32 m0, m0, -1\n\t"                                                                              \
                "s_nop 0\n\t"                                                                                           \
                "v_readlane_b32 " DR_A6 ", %[stk], m0\n"                                                                \
                V "1:\n\t"                           /* the item to do is in A6 (popped, or the right child's) */        \
                "s_cmp_lt_i32 " DR_A6 ", 0\n\t"                                                                         \
                "s_cbranch_scc1 " V "5f\n\t"                                                                            \
                "s_load_dwordx16 " DR_AB_ALL ", " BVH ", " DR_A6 " offset:0x0\n\t"                                      \
                "s_waitcnt lgkmcnt(0)\n\t"                                                                              \
                TESTL                                                                                                   \
                "s_cbranch_vccz " V "7f\n\t"                                                                            \
                TESTR                                                                                                   \
                "s_cbranch_vccz " V "2f\n\t"                                                                            \
                "v_writelane_b32 %[stk], " DR_A6 ", m0\n\t"                                                             \
                "s_add_u32 m0, m0, 1\n"                                                                                 \
                V "2:\n\t"                           /* the item to do is in B6 (the left child's) */                    \
                "s_cmp_lt_i32 " DR_B6 ", 0\n\t"                                                                         \
                "s_cbranch_scc1 " V "6f\n\t"                                                                            \
                "s_load_dwordx16 " DR_AB_ALL ", " BVH ", " DR_B6 " offset:0x0\n\t"                                      \
                "s_waitcnt lgkmcnt(0)\n\t"                                                                              \
                TESTL                                                                                                   \
                "s_cbranch_vccz " V "7f\n\t"                                                                            \
                TESTR                                                                                                   \
                "s_cbranch_vccz " V "2b\n\t"                                                                            \
                "v_writelane_b32 %[stk], " DR_A6 ", m0\n\t"                                                             \
                "s_add_u32 m0, m0, 1\n\t"                                                                               \
                "s_branch " V "2b\n"                                                                                    \
                V "7:\n\t"                           /* the left child is missed */                                     \
                TESTR                                                                                                   \
                "s_cbranch_vccz " V "0b\n\t"                                                                            \
                "s_branch " V "1b\n"                                                                                    \
                V "5:\n\t"                                                                                              \
                "s_mov_b32 %[off], " DR_A6 "\n\t"                                                                       \
                "s_branch 99f\n"                                                                                        \
                V "6:\n\t"                                                                                              \
                "s_mov_b32 %[off], " DR_B6 "\n\t"                                                                       \
                "s_branch 99f\n"
// the left child is the record's first half = register set B, the right child set A
#define DR_PWALK_OCTANT(V, SX, SY, SZ)                                                                                  \
        DR_PWALK_VARIANT(V, "%[bvhlh]",                                                                                 \
                DR_NODE_TEST_SX(DR_SEL(SX, DR_B0, DR_B3), DR_SEL(SY, DR_B1, DR_B4), DR_SEL(SZ, DR_B2, DR_B5),           \
                               DR_SEL(SX, DR_B3, DR_B0), DR_SEL(SY, DR_B4, DR_B1), DR_SEL(SZ, DR_B5, DR_B2)),           \
                DR_NODE_TEST_SX(DR_SEL(SX, DR_A0, DR_A3), DR_SEL(SY, DR_A1, DR_A4), DR_SEL(SZ, DR_A2, DR_A5),           \
                               DR_SEL(SX, DR_A3, DR_A0), DR_SEL(SY, DR_A4, DR_A1), DR_SEL(SZ, DR_A5, DR_A2)))
#define DR_PWALK_ASM                                                                                                    \
            asm volatile(                                                                                               \
                "s_mov_b64 %[sexec], exec\n\t"                                                                          \
                "s_mov_b64 exec, %[alive]\n\t"                                                                          \
                "s_mov_b32 m0, %[sp]\n\t"                                                                               \
                "s_cmp_gt_u32 %[oct], 7\n\t"                                                                            \
                "s_cbranch_scc1 80f\n\t"                                                                                \
                "s_bitcmp1_b32 %[oct], 2\n\t"                                                                           \
                "s_cbranch_scc1 94f\n\t"                                                                                \
                "s_bitcmp1_b32 %[oct], 1\n\t"                                                                           \
                "s_cbranch_scc1 92f\n\t"                                                                                \
                "s_bitcmp1_b32 %[oct], 0\n\t"                                                                           \
                "s_cbranch_scc1 10f\n\t"                                                                                \
                "s_branch 00f\n"                                                                                        \
                "92:\n\t"                                                                                               \
                "s_bitcmp1_b32 %[oct], 0\n\t"                                                                           \
                "s_cbranch_scc1 30f\n\t"                                                                                \
                "s_branch 20f\n"                                                                                        \
                "94:\n\t"                                                                                               \
                "s_bitcmp1_b32 %[oct], 1\n\t"                                                                           \
                "s_cbranch_scc1 96f\n\t"                                                                                \
                "s_bitcmp1_b32 %[oct], 0\n\t"                                                                           \
                "s_cbranch_scc1 50f\n\t"                                                                                \
                "s_branch 40f\n"                                                                                        \
                "96:\n\t"                                                                                               \
                "s_bitcmp1_b32 %[oct], 0\n\t"                                                                           \
                "s_cbranch_scc1 70f\n\t"                                                                                \
                "s_branch 60f\n"                                                                                        \
                DR_PWALK_OCTANT("0", 0, 0, 0)                                                                           \
                DR_PWALK_OCTANT("1", 1, 0, 0)                                                                           \
                DR_PWALK_OCTANT("2", 0, 1, 0)                                                                           \
                DR_PWALK_OCTANT("3", 1, 1, 0)                                                                           \
                DR_PWALK_OCTANT("4", 0, 0, 1)                                                                           \
                DR_PWALK_OCTANT("5", 1, 0, 1)                                                                           \
                DR_PWALK_OCTANT("6", 0, 1, 1)                                                                           \
                DR_PWALK_OCTANT("7", 1, 1, 1)                                                                           \
                DR_PWALK_VARIANT("8", "%[bvh]", DR_NODE_TEST_X(DR_B0, DR_B1, DR_B2, DR_B3, DR_B4, DR_B5),               \
                                 DR_NODE_TEST_X(DR_A0, DR_A1, DR_A2, DR_A3, DR_A4, DR_A5))                              \
                "98:\n\t"                                                                                               \
                "s_mov_b32 %[off], 0x7ffffff8\n"                                                                        \
                "99:\n\t"                                                                                               \
                "s_mov_b32 %[sp], m0\n\t"                                                                               \
                "s_mov_b64 exec, %[sexec]"                                                                              \
                : [off] "=&s"(item), [sp] "+s"(sp), [stk] "+v"(stk), [sexec] "=&s"(sexec), [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2),   \
                  [t3] "=&v"(t3), [t4] "=&v"(t4), [t5] "=&v"(t5)                                                        \
                : [bvh] "s"(pairs), [bvhlh] "s"(pairs_lh), [oct] "s"(octant), [alive] "s"(alive_m), [kx] "v"(kk.x), [ky] "v"(kk.y),   \
                  [kz] "v"(kk.z), [ix] "v"(iv.x), [iy] "v"(iv.y), [iz] "v"(iv.z)                                        \
                : DR_WALK_CLOBBERS, "vcc", "scc", "m0")

// stk / sp: the stack on entry (lanes 0 .. sp-1 of stk): {0} = the root's record for a walk of the whole tree
template <bool STATS>
__device__ __forceinline__ unsigned long long walk_pairs(const BvhPair* __restrict__ pairs, const BvhPair* __restrict__ pairs_lh,
                                                         const TriRec* __restrict__ tri_sorted, f3 org, f3 dn, f3 inv, f3 iv, f3 kk,
                                                         float tmax, int hi, unsigned long long alive_m, int octant, int stk, unsigned sp,
                                                         int& n_visit, int& n_leaf) {
    for (;;) {
        int item;
        if (STATS) {
            // counted variant of the same walk (debug builds only)
            item = BVH_END;
            while (sp != 0u) {
                sp--;
                int it = __builtin_amdgcn_readlane(stk, (int)sp);
                while (it >= 0) {
                    const char* rp = reinterpret_cast<const char*>(pairs) + (unsigned)it;
                    const v8f L = *reinterpret_cast<const v8f*>(rp), R = *reinterpret_cast<const v8f*>(rp + 32);
                    const float lc[3] = { L[0], L[1], L[2] }, lh[3] = { L[3], L[4], L[5] }, rc[3] = { R[0], R[1], R[2] }, rh[3] = { R[3], R[4], R[5] };
                    const int li = __builtin_amdgcn_readfirstlane(__float_as_int(L[6])), ri = __builtin_amdgcn_readfirstlane(__float_as_int(R[6]));
                    n_visit += 2;
                    const bool hl = (node_hit_mask(lc, lh, iv, kk, 1.0f) & alive_m) != 0ull, hr = (node_hit_mask(rc, rh, iv, kk, 1.0f) & alive_m) != 0ull;
                    if (hl && hr) { stk = ((int)(threadIdx.x & 63u) == (int)sp) ? ri : stk; sp++; it = li; }
                    else if (hl) it = li;
                    else if (hr) it = ri;
                    else { it = 0; break; }
                }
                if (it < 0) { item = it; break; }
            }
        } else {
            float t0, t1, t2, t3, t4, t5;
            unsigned long long sexec;
            DR_PWALK_ASM;
        }
        if (item == BVH_END) break;
        if (STATS) n_leaf++;
        const unsigned long long blocked_m = leaf_blocked_mask(tri_sorted, item & 0x7fffffff, org, dn, inv, tmax, hi, alive_m);
        alive_m &= ~blocked_m;
        if (alive_m == 0ull) break;
    }
    return alive_m;
}
#undef DR_PWALK_ASM

// ---------------------------------------------------------------------------------------
// PATH RECORDS in front of the pair walk (PathHdr, dr_internal.h).  Every ray of the pair (lo, hi) starts on patch lo and
// ends on patch hi, so it is inside every ancestor of the two patches' leaves: in a walk from the root about 22 of the 78
// node tests per pair are of those nodes and tell nothing.  What a walk needs instead are the SIBLINGS hanging off the two
// root-to-leaf paths -- with the two leaves they cover the whole tree -- and those are the same for every pair a patch takes
// part in: k_paths writes them out once per patch, in root-to-leaf order, box (lower / upper corner) + item.  A pair streams
//   lo's records above the level where the two paths part [0, ell), lo's below it (ell, Dl) -- index ell, the branch
//   that holds hi, is left out -- and hi's below it (ell, Dh)
// with the next record already loading while one is tested (their addresses depend on no test), PUSHES the item of every
// record some live ray touches onto the pair walk's stack, then the two patches' own leaves (inside their boxes by
// construction: no test), and lets the pair walk pop until the stack is empty.  (Round 2's form of this walk entered the tree
// at every touched record and lost more in cold starts than it saved in tests; here the records only fill the stack.)
// Exactness: records + leaves cover every leaf of the tree, each subtree is entered under the same node tests as in a walk
// from the root -- only the ancestors' tests are not made, and if a ray would have failed one of those this walk reaches
// more leaves than the walk from the root, never fewer; what is hit is decided per triangle.
// Only for waves whose rays share an octant (the records are kept in lower / upper corner form only) and patches no deeper
// than PATH_RECS; every other pair walks from the root.
// ---------------------------------------------------------------------------------------
#define DR_PSTREAM_VARIANT(V, TESTA, TESTB)                                                                             \
                V "0:\n\t"                                                                                              \
                "s_load_dwordx8 " DR_A_ALL ", %[base], %[roff] offset:0x0\n\t"                                          \
                "s_waitcnt lgkmcnt(0)\n"                                                                                \
                V "1:\n\t"                                                                                              \
                "s_load_dwordx8 " DR_B_ALL ", %[base], %[roff] offset:0x20\n\t"                                         \
                TESTA                                                                                                   \
                "s_cbranch_vccz " V "3f\n\t"                                                                            \
                "v_writelane_b32 %[stk], " DR_A6 ", m0\n\t"                                                             \
                "s_add_u32 m0, m0, 1\n"                                                                                 \
                V "3:\n\t"                                                                                              \
                "s_add_u32 %[roff], %[roff], 32\n\t"                                                                    \
                "s_cmp_lt_u32 %[roff], %[rend]\n\t"                                                                     \
                "s_cbranch_scc0 99f\n\t"                                                                                \
                "s_waitcnt lgkmcnt(0)\n\t"                                                                              \
                "s_load_dwordx8 " DR_A_ALL ", %[base], %[roff] offset:0x20\n\t"                                         \
                TESTB                                                                                                   \
                "s_cbranch_vccz " V "4f\n\t"                                                                            \
                "v_writelane_b32 %[stk], " DR_B6 ", m0\n\t"                                                             \
                "s_add_u32 m0, m0, 1\n"                                                                                 \
                V "4:\n\t"                                                                                              \
                "s_add_u32 %[roff], %[roff], 32\n\t"                                                                    \
                "s_cmp_lt_u32 %[roff], %[rend]\n\t"                                                                     \
                "s_cbranch_scc0 99f\n\t"                                                                                \
                "s_waitcnt lgkmcnt(0)\n\t"                                                                              \
                "s_branch " V "1b\n"
#define DR_PSTREAM_OCTANT(V, SX, SY, SZ)                                                                                \
        DR_PSTREAM_VARIANT(V,                                                                                           \
                DR_NODE_TEST_SX(DR_SEL(SX, DR_A0, DR_A3), DR_SEL(SY, DR_A1, DR_A4), DR_SEL(SZ, DR_A2, DR_A5),           \
                               DR_SEL(SX, DR_A3, DR_A0), DR_SEL(SY, DR_A4, DR_A1), DR_SEL(SZ, DR_A5, DR_A2)),           \
                DR_NODE_TEST_SX(DR_SEL(SX, DR_B0, DR_B3), DR_SEL(SY, DR_B1, DR_B4), DR_SEL(SZ, DR_B2, DR_B5),           \
                               DR_SEL(SX, DR_B3, DR_B0), DR_SEL(SY, DR_B4, DR_B1), DR_SEL(SZ, DR_B5, DR_B2)))
#define DR_PSTREAM_ASM                                                                                                  \
            asm(                                                                                               \
                "s_mov_b64 %[sexec], exec\n\t"                                                                          \
                "s_mov_b64 exec, %[alive]\n\t"                                                                          \
                "s_mov_b32 m0, %[sp]\n\t"                                                                               \
                "s_bitcmp1_b32 %[oct], 2\n\t"                                                                           \
                "s_cbranch_scc1 94f\n\t"                                                                                \
                "s_bitcmp1_b32 %[oct], 1\n\t"                                                                           \
                "s_cbranch_scc1 92f\n\t"                                                                                \
                "s_bitcmp1_b32 %[oct], 0\n\t"                                                                           \
                "s_cbranch_scc1 10f\n\t"                                                                                \
                "s_branch 00f\n"                                                                                        \
                "92:\n\t"                                                                                               \
                "s_bitcmp1_b32 %[oct], 0\n\t"                                                                           \
                "s_cbranch_scc1 30f\n\t"                                                                                \
                "s_branch 20f\n"                                                                                        \
                "94:\n\t"                                                                                               \
                "s_bitcmp1_b32 %[oct], 1\n\t"                                                                           \
                "s_cbranch_scc1 96f\n\t"                                                                                \
                "s_bitcmp1_b32 %[oct], 0\n\t"                                                                           \
                "s_cbranch_scc1 50f\n\t"                                                                                \
                "s_branch 40f\n"                                                                                        \
                "96:\n\t"                                                                                               \
                "s_bitcmp1_b32 %[oct], 0\n\t"                                                                           \
                "s_cbranch_scc1 70f\n\t"                                                                                \
                "s_branch 60f\n"                                                                                        \
                DR_PSTREAM_OCTANT("0", 0, 0, 0)                                                                         \
                DR_PSTREAM_OCTANT("1", 1, 0, 0)                                                                         \
                DR_PSTREAM_OCTANT("2", 0, 1, 0)                                                                         \
                DR_PSTREAM_OCTANT("3", 1, 1, 0)                                                                         \
                DR_PSTREAM_OCTANT("4", 0, 0, 1)                                                                         \
                DR_PSTREAM_OCTANT("5", 1, 0, 1)                                                                         \
                DR_PSTREAM_OCTANT("6", 0, 1, 1)                                                                         \
                DR_PSTREAM_OCTANT("7", 1, 1, 1)                                                                         \
                "99:\n\t"                                                                                               \
                "s_mov_b32 %[sp], m0\n\t"                                                                               \
                "s_mov_b64 exec, %[sexec]\n\t"                                                                          \
                "s_waitcnt lgkmcnt(0)"                                                                                  \
                : [roff] "+s"(roff), [sp] "+s"(sp), [stk] "+v"(stk), [sexec] "=&s"(sexec), [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2),   \
                  [t3] "=&v"(t3), [t4] "=&v"(t4), [t5] "=&v"(t5)                                                        \
                : [base] "s"(rec), [rend] "s"(rend), [oct] "s"(octant), [alive] "s"(alive_m), [kx] "v"(kk.x), [ky] "v"(kk.y),      \
                  [kz] "v"(kk.z), [ix] "v"(iv.x), [iy] "v"(iv.y), [iz] "v"(iv.z)                                        \
                : DR_WALK_CLOBBERS, "vcc", "scc", "m0")

// one stretch [roff, rend) (byte offsets from rec) of path records: the items of the touched ones onto the stack
template <bool STATS>
__device__ __forceinline__ void stream_path_records(const BvhNode* __restrict__ rec, unsigned roff, unsigned rend, f3 iv, f3 kk,
                                                    unsigned long long alive_m, int octant, int& stk, unsigned& sp, int& n_stream) {
    if (roff >= rend) return;
    if (STATS) {
        for (; roff < rend; roff += 32u) {
            const v8f raw = *reinterpret_cast<const v8f*>(reinterpret_cast<const char*>(rec) + roff);
            // (counted build: the general test on the record's centre / half-extent -- the same box up to rounding outwards)
            const float nc[3] = { 0.5f * raw[0] + 0.5f * raw[3], 0.5f * raw[1] + 0.5f * raw[4], 0.5f * raw[2] + 0.5f * raw[5] };
            const float nh[3] = { (raw[3] - raw[0]) * 0.5000001f + 1e-30f, (raw[4] - raw[1]) * 0.5000001f + 1e-30f, (raw[5] - raw[2]) * 0.5000001f + 1e-30f };
            n_stream++;
            if ((node_hit_mask(nc, nh, iv, kk, 1.0f) & alive_m) != 0ull) {
                const int it = __builtin_amdgcn_readfirstlane(__float_as_int(raw[6]));
                stk = ((int)(threadIdx.x & 63u) == (int)sp) ? it : stk;
                sp++;
            }
        }
    } else {
        float t0, t1, t2, t3, t4, t5;
        unsigned long long sexec;
        DR_PSTREAM_ASM;
    }
}
#undef DR_PSTREAM_ASM

// one item onto the lane stack
__device__ __forceinline__ void lane_push(int& stk, unsigned& sp, int item) {
    asm("s_mov_b32 m0, %[sp]\n\ts_nop 0\n\tv_writelane_b32 %[stk], %[it], m0\n\ts_add_u32 %[sp], %[sp], 1"
                 : [stk] "+v"(stk), [sp] "+s"(sp) : [it] "s"(item) : "m0", "scc");
}


// STATS builds count BVH visits with global atomics inside the pair loop; that store makes
// the compiler give up scalar (SMEM) loads for nodes and triangles, so it is a separate,
// debug-only instantiation.
// amdgpu_num_sgpr(82) -> 80 SGPRs in the code object: the most with which a CU admits 8 blocks of 256 threads (the rest is
// parked in VGPR lanes outside the pair loop); with 64 VGPRs and 19.7 KB of LDS that is 8 waves per SIMD, the hardware's maximum
// WALK: 3 = the walk over the sibling-pair records from the root (walk_pairs); 2 = the same walk with its stack filled from the two
// patches' path records (stream_path_records); 0 = the threaded tree from the root (round 2's; trees deeper than the pair walk's
// stack).  (Round 2 also had tile-pair shaft lists: exact, measured slower, removed; profiles/r02/assembly_notes.md.)
template <int NT, bool STATS, int WALK>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_num_sgpr(82), amdgpu_waves_per_eu(8, 8))) void k_ff_tiles(TileParams P) {
    constexpr bool PATHS = (WALK == 2);
    const int t = blockIdx.x;
    const int o = P.tile0 + blockIdx.y;
    const bool t_owned = (t >= P.tile0) && (t < P.tile0 + P.nOwnedTiles);
    if (t_owned && t < o) return;   // the unordered tile pair {t,o} belongs to block (x=o, y=t-tile0)
    // ray-count exchange (TileParams): who traces a pair with a foreign tile, and what the second launch covers
    bool from_slot = false;
    if (P.vx_mode != 0) {
        if (t_owned) {
            if (P.vx_mode == 2) return;                    // own x own pairs were finished by the first launch
        } else {
            const bool mine = vx_tracer_rank(o, t, P.vx_tiles_per_rank) == P.vx_rank;
            if (mine != (P.vx_mode == 1)) return;
            from_slot = (P.vx_mode == 2);
        }
    }
    const int bi = min(t, o), bj = max(t, o);
    const int I0 = bi * TILE, J0 = bj * TILE;
    const bool diag = (bi == bj);
    const bool ownI = (bi >= P.tile0) && (bi < P.tile0 + P.nOwnedTiles);
    const bool ownJ = (bj >= P.tile0) && (bj < P.tile0 + P.nOwnedTiles);

    __shared__ float sRec[2][TILE][REC_STRIDE];
    __shared__ unsigned char sVis[TILE][VIS_STRIDE];   // ray count per pair, 255 = not traced
    // The queue of traced pairs (up to 4096 entries of 2 bytes): its second half lives in sRec's memory -- the patch records are
    // not read while the pairs are traced (the walk takes the triangles from global memory) and are loaded again for the
    // write-out.  With that the block needs 19.7 instead of 23.3 KB of LDS, and with the kernel's 80 SGPRs a CU holds 8 blocks
    // instead of 7.  sAct: one bit per pair, "traced", between the pass that evaluates the integrand (reads sRec) and the
    // pass that fills the queue (overwrites it).
    __shared__ unsigned short sQueue[TILE * TILE / 2];
    __shared__ unsigned sAct[TILE * TILE / 32];
    __shared__ int sCount;
    unsigned short* const sQueueB = reinterpret_cast<unsigned short*>(&sRec[0][0][0]);
    static_assert(sizeof(float) * 2 * TILE * REC_STRIDE >= sizeof(unsigned short) * TILE * TILE / 2, "second half of the queue does not fit the records");

    const int tid = threadIdx.x;
    const int lane = tid & 63;

    if (tid == 0) sCount = 0;
    for (int x = tid; x < 2 * TILE * 20; x += NT) {
        int side = x / (TILE * 20);
        int r = (x - side * TILE * 20) / 20, c = x % 20;
        int g = (side ? J0 : I0) + r;
        sRec[side][r][c] = (g < P.N) ? reinterpret_cast<const float*>(P.patch + g)[c] : 0.0f;
    }
    __syncthreads();

    if (from_slot) {
        // the other rank traced this pair: its counts, in the same orientation (rows = the tile of the lower index)
        const int T = P.vx_tiles_per_rank, B = t / T;
        const unsigned char* slot = P.vrecv + (((size_t)B * T + (t - B * T)) * T + (o - P.tile0)) * (TILE * TILE);
        for (int p = tid; p < TILE * TILE; p += NT) sVis[p >> 6][p & 63] = slot[p];
    }
    // ---- which pairs are traced: stored integrand lo->hi > 0 (vs/OptixPrimeFunctionality.cpp:190)
    for (int p0 = 0; p0 < (from_slot ? 0 : TILE * TILE); p0 += NT) {
        const int p = p0 + tid;
        const int i = p >> 6, j = p & 63;
        const int gi = I0 + i, gj = J0 + j;
        bool act = false;
        if (p < TILE * TILE) {
            if ((gi < P.N) && (gj < P.N) && (gi < gj)) {
                float fa, fb;
                integrand_pair(sRec[0][i], sRec[1][j], fa, fb);
                act = fa > 0.0f;
            }
            sVis[i][j] = 255;
        }
        const unsigned long long m = __ballot(act);
        if (lane == 0 && p < TILE * TILE) { sAct[p >> 5] = (unsigned)m; sAct[(p >> 5) + 1] = (unsigned)(m >> 32); }    // p is a multiple of 64 here
    }
    __syncthreads();
    // the queue, in the order of p (sRec is dead from here to the write-out)
    for (int p0 = 0; p0 < (from_slot ? 0 : TILE * TILE); p0 += NT) {
        const int p = p0 + tid;
        const bool act = (p < TILE * TILE) && ((sAct[p >> 5] >> (p & 31)) & 1u);
        const unsigned long long m = __ballot(act);
        int base = 0;
        if (lane == 0 && m) base = atomicAdd(&sCount, __popcll(m));
        base = __shfl(base, 0);
        if (act) {
            const int at = base + __popcll(m & ((1ull << lane) - 1ull));
            if (at < TILE * TILE / 2) sQueue[at] = (unsigned short)p;
            else sQueueB[at - TILE * TILE / 2] = (unsigned short)p;
        }
    }
    __syncthreads();
    const int n_act = sCount;
    const unsigned end_all = (unsigned)P.n_nodes * (unsigned)sizeof(BvhNode);
    // every ray starts inside the root's box: an interior root is entered without its test (the walk starts at its first child)
    const unsigned root_off = P.n_nodes >= 3 ? (unsigned)sizeof(BvhNode) : 0u;

    // ---- visibility: one wave per pair, one lane per ray, wave-uniform BVH walk ---------
    if (n_act > 0 && P.trace) {
        // queue entries are dealt round-robin to the block's waves (all indices wave-uniform)
        const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        const int n_act_u = __builtin_amdgcn_readfirstlane(n_act);
        for (int q = wave; q < n_act_u; q += NT / 64) {
            const int p = __builtin_amdgcn_readfirstlane((int)(q < TILE * TILE / 2 ? sQueue[q] : sQueueB[q - TILE * TILE / 2]));
            const int i = p >> 6, j = p & 63;
            const int lo = I0 + i, hi = J0 + j;
            const TriRec Tl = P.tri[lo];
            const TriRec Th = P.tri[hi];
            const f3 la = ld3(Tl.a), le1 = ld3(Tl.e1), le2 = ld3(Tl.e2);
            const f3 ha = ld3(Th.a), he1 = ld3(Th.e1), he2 = ld3(Th.e2);
            int count = 0;
            int n_visit = 0, n_leaf = 0, n_stream = 0;
            for (int k0 = 0; k0 < P.K; k0 += 64) {
                const int k = k0 + lane;
                bool alive = k < P.K;
                f3 org = f3{ 0, 0, 0 }, dn = f3{ 0, 0, 1 };
                float tmax = 0.0f;
                if (alive) {
                    const float u = P.uv[2 * k], v = P.uv[2 * k + 1];
                    // uv2xyz on both patches, vs/triangle_math.cpp:3-9; ray per
                    // vs/OptixPrimeFunctionality.cpp:191-196
                    f3 src = (la + le1 * u) + le2 * v;
                    f3 dst = (ha + he1 * u) + he2 * v;
                    f3 dv = dst - src;
                    dn = dv * (1.0f / sqrtf(dot3(dv, dv)));
                    org = src + dn * P.eps;
                }
                const f3 inv = f3{ safe_inv(dn.x), safe_inv(dn.y), safe_inv(dn.z) };
                // the destination must be hit at all: its gate on [0,inf) and the triangle test
                if (alive) alive = tri_hit(org, dn, ha, he1, he2, tmax);
                alive = alive && ((box_hit_mask(Th.lo, Th.hi, org, inv, INFINITY) >> lane) & 1ull);
                unsigned long long alive_m = __builtin_amdgcn_ballot_w64(alive);
                if (STATS && lo == P.dbg_lo && hi == P.dbg_hi && lane == 0) P.pairs_traced[1] = alive_m;
                if (alive_m != 0ull) {
                    // per-ray constants of the node test
                    // in units of the ray's own length: s just below 1 / tmax (never above: the walk may see the ray a little
                    // longer, not shorter; at most ts_max = 1e19 / max|coordinate| so that org * iv stays finite), the scaled length is then 1
                    const float ts = fminf(__builtin_amdgcn_rcpf(tmax) * 0.99999f, P.ts_max);
                    const f3 iv = f3{ __builtin_amdgcn_fmed3f(inv.x, -1e18f, 1e18f) * ts, __builtin_amdgcn_fmed3f(inv.y, -1e18f, 1e18f) * ts,
                                      __builtin_amdgcn_fmed3f(inv.z, -1e18f, 1e18f) * ts };
                    const f3 kk = f3{ -(org.x * iv.x), -(org.y * iv.y), -(org.z * iv.z) };
                    const float tmax_w = 1.0f;
                    // do all live rays point into one octant?  (they run from one patch to one patch: nearly always)
                    int octant = 8;
                    if ((WALK == 0 ? (const void*)P.bvh_lh : (const void*)P.pairs_lh) != nullptr) {
                        const unsigned long long nx = alive_m & __builtin_amdgcn_fcmpf(dn.x, 0.0f, 4), ny = alive_m & __builtin_amdgcn_fcmpf(dn.y, 0.0f, 4),
                                                 nz = alive_m & __builtin_amdgcn_fcmpf(dn.z, 0.0f, 4);
                        if ((nx == 0ull || nx == alive_m) && (ny == 0ull || ny == alive_m) && (nz == 0ull || nz == alive_m))
                            octant = (nx ? 1 : 0) | (ny ? 2 : 0) | (nz ? 4 : 0);
                    }
                    octant = __builtin_amdgcn_readfirstlane(octant);
                    if (STATS && octant == 8 && lane == 0 && P.dbg_lo < 0) atomicAdd(P.pairs_traced + 13, 1ull);
                    if (WALK == 0) {
                        // the threaded tree, sentinel-terminated: no end-of-range compares in the walk
                        alive_m = walk_range<STATS>(P.bvh, P.tri_sorted, root_off, end_all, org, dn, inv, iv, kk, tmax, tmax_w, hi, alive_m, n_visit, n_leaf,
                                                    P.bvh_lh, octant);
                    } else {
                        int stk = 0;                  // lane 0 = the root's record (offset 0)
                        unsigned sp = 1u;
                        if (PATHS && P.path_hdr != nullptr && octant < 8) {
                            // the two patches' path records (PathHdr): where their root-to-leaf paths part, which records to stream
                            // (looked up here, not ahead of the ray set-up: nothing of it is live across that)
                            const PathHdr hl = P.path_hdr[lo], hh = P.path_hdr[hi];
                            const int Dl = __builtin_amdgcn_readfirstlane(hl.depth), Dh = __builtin_amdgcn_readfirstlane(hh.depth);
                            if (Dl >= 0 && Dh >= 0) {
                                const int m = min(Dl, Dh);
                                const unsigned long long tl = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)hl.turns_hi) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)hl.turns_lo);
                                const unsigned long long th = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)hh.turns_hi) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)hh.turns_lo);
                                const unsigned long long x = (tl ^ th) & (m >= 64 ? ~0ull : ((1ull << m) - 1ull));
                                const unsigned ell = (unsigned)(x ? __builtin_ctzll(x) : m);                 // levels the two paths share
                                const unsigned bl = (unsigned)lo * (unsigned)(PATH_RECS * sizeof(BvhNode)), bh = (unsigned)hi * (unsigned)(PATH_RECS * sizeof(BvhNode));
                                const int leaf_lo = __builtin_amdgcn_readfirstlane(hl.leaf), leaf_hi = __builtin_amdgcn_readfirstlane(hh.leaf);
                                // the stack from the path records: touched siblings, then the two patches' own leaves
                                sp = 0u;
                                stream_path_records<STATS>(P.path_rec, bl, bl + ell * 32u, iv, kk, alive_m, octant, stk, sp, n_stream);
                                stream_path_records<STATS>(P.path_rec, bl + (ell + 1u) * 32u, bl + (unsigned)Dl * 32u, iv, kk, alive_m, octant, stk, sp, n_stream);
                                stream_path_records<STATS>(P.path_rec, bh + (ell + 1u) * 32u, bh + (unsigned)Dh * 32u, iv, kk, alive_m, octant, stk, sp, n_stream);
                                if (STATS) {
                                    stk = ((int)lane == (int)sp) ? leaf_hi : stk; sp++;
                                    if (leaf_lo != leaf_hi) { stk = ((int)lane == (int)sp) ? leaf_lo : stk; sp++; }
                                } else {
                                    lane_push(stk, sp, leaf_hi);
                                    if (leaf_lo != leaf_hi) lane_push(stk, sp, leaf_lo);
                                }
                            }
                        }
                        alive_m = walk_pairs<STATS>(P.pairs, P.pairs_lh, P.tri_sorted, org, dn, inv, iv, kk, tmax, hi, alive_m, octant, stk, sp, n_visit, n_leaf);
                    }
                }
                count += __popcll(alive_m);
                if (STATS && lo == P.dbg_lo && hi == P.dbg_hi && lane == 0) P.pairs_traced[3] = alive_m;
            }
            if (lane == 0) sVis[i][j] = (unsigned char)count;
            if (STATS && lane == 0 && P.dbg_lo < 0) {
                atomicAdd(P.pairs_traced + 1, (unsigned long long)n_visit);
                atomicAdd(P.pairs_traced + 2, (unsigned long long)n_leaf);
                atomicAdd(P.pairs_traced + 12, (unsigned long long)n_stream);
            }
        }
    }
    __syncthreads();
    if (!from_slot) {
        // the patch records again (the queue's second half lay over them)
        for (int x = tid; x < 2 * TILE * 20; x += NT) {
            int side = x / (TILE * 20);
            int r = (x - side * TILE * 20) / 20, c = x % 20;
            int g = (side ? J0 : I0) + r;
            sRec[side][r][c] = (g < P.N) ? reinterpret_cast<const float*>(P.patch + g)[c] : 0.0f;
        }
        __syncthreads();
    }
    if (tid == 0 && n_act > 0 && P.trace && P.pairs_traced) atomicAdd(P.pairs_traced, (unsigned long long)n_act);
    if (P.vx_mode == 1 && !t_owned) {
        // this pair is also the other rank's: hand its ray counts over (slot of own tile o, foreign tile t)
        const int T = P.vx_tiles_per_rank, B = t / T;
        unsigned char* slot = P.vsend + (((size_t)B * T + (o - P.tile0)) * T + (t - B * T)) * (TILE * TILE);
        for (int p = tid; p < TILE * TILE; p += NT) slot[p] = sVis[p >> 6][p & 63];
    }

    // ---- write both F tiles once, coalesced (the integrand is recomputed rather than kept in LDS)
    const float Kf = (float)P.K;
    for (int pass = 0; pass < 2; pass++) {
        // pass 0: rows of the I block (F[I0+r][J0+c]); pass 1: rows of the J block (F[J0+r][I0+c])
        if (pass == 0 ? !ownI : (!ownJ || diag)) continue;
        const int R0 = pass == 0 ? I0 : J0, C0 = pass == 0 ? J0 : I0;
        for (int p = tid; p < TILE * TILE; p += NT) {
            const int r = p >> 6, c = p & 63;
            const int gr = R0 + r, gc = C0 + c;
            if (gr >= P.N || gc >= P.N) continue;
            // (i,j) = tile coordinates of the unordered pair, i in the I block
            int i, j; bool fwd;     // fwd: the entry is F[lo][hi]
            if (diag) { fwd = gr < gc; i = fwd ? r : c; j = fwd ? c : r; }
            else { fwd = (pass == 0); i = fwd ? r : c; j = fwd ? c : r; }
            float val = 0.0f;
            unsigned char vc = 255;
            if (gr != gc) {
                vc = sVis[i][j];
                if (vc != 255 || !P.trace) {
                    float fu_f, fu_r;
                    integrand_pair(sRec[0][i], sRec[1][j], fu_f, fu_r);
                    if (fu_f > 0.0f) {
                        const float V = P.trace ? (float)vc / Kf : 1.0f;
                        if (P.rule == 0) {
                            if (V > 0.0f) val = V * (fwd ? fu_f : fu_r);
                        } else {
                            const float f = fu_f * V;
                            if (f > 0.0f) val = fwd ? f : (sRec[0][i][19] * f) / sRec[1][j][19];
                        }
                    }
                }
                if (!P.trace) vc = 255;
            }
            P.F[(size_t)(gr - P.row0) * P.ldF + gc] = val;
            if (P.vis) P.vis[(size_t)(gr - P.row0) * P.N + gc] = vc;
        }
    }
}

hipError_t launch_ff_tiles(hipStream_t st, const TileParams& p) {
    dim3 grid(p.nT, p.nOwnedTiles);
    const int walk = !p.pairs ? 0 : (p.path_hdr ? 2 : 3);
    if (p.stats & 1) {
        if (walk == 2) hipLaunchKernelGGL((k_ff_tiles<256, true, 2>), grid, dim3(256), 0, st, p);
        else if (walk == 3) hipLaunchKernelGGL((k_ff_tiles<256, true, 3>), grid, dim3(256), 0, st, p);
        else hipLaunchKernelGGL((k_ff_tiles<256, true, 0>), grid, dim3(256), 0, st, p);
        return hipGetLastError();
    }
    if (walk == 2) { hipLaunchKernelGGL((k_ff_tiles<256, false, 2>), grid, dim3(256), 0, st, p); return hipGetLastError(); }
    if (walk == 3) { hipLaunchKernelGGL((k_ff_tiles<256, false, 3>), grid, dim3(256), 0, st, p); return hipGetLastError(); }
    // (512 and 1024 threads per workgroup -- more waves sharing a queue -- measured 5 % and 14 % slower at 64k patches)
    hipLaunchKernelGGL((k_ff_tiles<256, false, 0>), grid, dim3(256), 0, st, p);
    return hipGetLastError();
}

}  // namespace dr
