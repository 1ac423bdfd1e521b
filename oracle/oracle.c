/*
 * oracle.c -- CPU restatement of DaisyRiot's radiosity hot path (see oracle.h).
 *
 * TEST INFRASTRUCTURE ONLY: the checker for tests/, smoke() and the timed
 * cpu_baseline of bench.py.  Never linked into the product library.
 *
 * "vs/" = /root/reference/visual studio/.  Build: see oracle/Makefile
 * (gcc -O2 -ffp-contract=off -fopenmp).
 */
#include "oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ---- fp32 vector helpers, operation order as in the reference's glm 0.9.8.4 --- */
typedef struct { float x, y, z; } v3;

static inline v3 v3_ld(const float* p) { v3 r = { p[0], p[1], p[2] }; return r; }
static inline v3 v3_add(v3 a, v3 b) { v3 r = { a.x + b.x, a.y + b.y, a.z + b.z }; return r; }
static inline v3 v3_sub(v3 a, v3 b) { v3 r = { a.x - b.x, a.y - b.y, a.z - b.z }; return r; }
static inline v3 v3_scale(v3 a, float s) { v3 r = { a.x * s, a.y * s, a.z * s }; return r; }
static inline v3 v3_div(v3 a, float s) { v3 r = { a.x / s, a.y / s, a.z / s }; return r; }
/* glm/detail/func_geometric.inl:54-60 : tmp = x*y; tmp.x + tmp.y + tmp.z */
static inline float v3_dot(v3 a, v3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
/* glm/detail/func_geometric.inl:74-84 */
static inline v3 v3_cross(v3 x, v3 y) {
    v3 r = { x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y };
    return r;
}
/* glm length: sqrt(dot(v,v)) (func_geometric.inl:14-19) */
static inline float v3_length(v3 a) { return sqrtf(v3_dot(a, a)); }
/* glm normalize: v * inversesqrt(dot(v,v)), inversesqrt(x) = 1/sqrt(x)
 * (func_geometric.inl:88-95, func_exponential.inl:129-133) */
static inline v3 v3_normalize(v3 a) { return v3_scale(a, 1.0f / sqrtf(v3_dot(a, a))); }

#define ORC_PIF 3.14159265358979323846f /* M_PIf */

static inline v3 vert(const orc_mesh* m, int tri, int corner) {
    return v3_ld(m->vertices + 3 * (long)m->tri_v[3 * (long)tri + corner]);
}
static inline v3 norm_at(const orc_mesh* m, int tri, int corner) {
    return v3_ld(m->normals + 3 * (long)m->tri_n[3 * (long)tri + corner]);
}

/* triangle_math.cpp:31-35: 0.5*length(cross(ab,ac)); the 0.5 is a double
 * literal, so the product is formed in double and truncated to float. */
static inline float surface3(v3 a, v3 b, v3 c) {
    v3 ab = v3_sub(b, a);
    v3 ac = v3_sub(c, a);
    return (float)(0.5 * (double)v3_length(v3_cross(ab, ac)));
}

float orc_surface(const float a[3], const float b[3], const float c[3]) {
    return surface3(v3_ld(a), v3_ld(b), v3_ld(c));
}

/* triangle_math.cpp:11-14: sum the corners, then each component / 3 */
static inline v3 centre3(v3 p0, v3 p1, v3 p2) {
    v3 s = v3_add(v3_add(p0, p1), p2);
    v3 r = { s.x / 3, s.y / 3, s.z / 3 };
    return r;
}

/* triangle_math.cpp:23-29 */
static inline v3 avg_normal(const orc_mesh* m, int tri) {
    v3 s = v3_add(v3_add(norm_at(m, tri, 0), norm_at(m, tri, 1)), norm_at(m, tri, 2));
    v3 a = { s.x / 3, s.y / 3, s.z / 3 };
    return v3_normalize(a);
}

/* triangle_math.cpp:60-74: midpoint split into four triangles */
static inline void divide4(const orc_mesh* m, int tri, v3 out[4][3]) {
    v3 a = vert(m, tri, 0), b = vert(m, tri, 1), c = vert(m, tri, 2);
    v3 iA = v3_add(v3_div(v3_sub(b, a), 2.0f), a);
    v3 iC = v3_add(v3_div(v3_sub(c, a), 2.0f), a);
    v3 iB = v3_add(v3_div(v3_sub(b, c), 2.0f), c);
    out[0][0] = a;  out[0][1] = iC; out[0][2] = iA;
    out[1][0] = iC; out[1][1] = c;  out[1][2] = iB;
    out[2][0] = iA; out[2][1] = iB; out[2][2] = b;
    out[3][0] = iA; out[3][1] = iB; out[3][2] = iC;
}

/* triangle_math.cpp:49-58.  powf(length,2) is taken as length*length (what
 * gcc folds it to, and the correctly rounded value). */
static inline float point_ff(v3 opos, v3 onrm, v3 dpos, v3 dnrm, float surface) {
    float ff = 0;
    float dot1 = v3_dot(onrm, v3_normalize(v3_sub(dpos, opos)));
    float dot2 = v3_dot(dnrm, v3_normalize(v3_sub(opos, dpos)));
    if (dot1 > 0 && dot2 > 0) {
        float length = v3_length(v3_sub(dpos, opos));
        ff = ((dot1 * dot2) / ((length * length) * ORC_PIF)) * surface;
    }
    return ff;
}

/* OptixPrimeFunctionality.cpp:133-161 */
float orc_p2p_integrand_literal(const orc_mesh* m, int origin, int dest) {
    v3 ot[4][3], dt[4][3], op[4], dp[4];
    divide4(m, origin, ot);
    divide4(m, dest, dt);
    v3 on = avg_normal(m, origin);
    v3 dn = avg_normal(m, dest);
    for (int i = 0; i < 4; i++) {
        op[i] = centre3(ot[i][0], ot[i][1], ot[i][2]);
        dp[i] = centre3(dt[i][0], dt[i][1], dt[i][2]);
    }
    float ff = 0;
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++)
            ff = ff + point_ff(op[i], on, dp[j], dn,
                               surface3(ot[i][0], ot[i][1], ot[i][2]) *
                               surface3(dt[j][0], dt[j][1], dt[j][2]));
    ff = ff / surface3(vert(m, origin, 0), vert(m, origin, 1), vert(m, origin, 2));
    return ff;
}

void orc_patch_records(const orc_mesh* m, float* cen, float* sub_area, float* nrm, float* area) {
    for (int t = 0; t < m->N; t++) {
        v3 s[4][3];
        divide4(m, t, s);
        for (int k = 0; k < 4; k++) {
            v3 c = centre3(s[k][0], s[k][1], s[k][2]);
            cen[(long)t * 12 + k * 3 + 0] = c.x;
            cen[(long)t * 12 + k * 3 + 1] = c.y;
            cen[(long)t * 12 + k * 3 + 2] = c.z;
            sub_area[(long)t * 4 + k] = surface3(s[k][0], s[k][1], s[k][2]);
        }
        v3 n = avg_normal(m, t);
        nrm[(long)t * 3 + 0] = n.x; nrm[(long)t * 3 + 1] = n.y; nrm[(long)t * 3 + 2] = n.z;
        area[t] = surface3(vert(m, t, 0), vert(m, t, 1), vert(m, t, 2));
    }
}

typedef struct { float* cen; float* sa; float* nrm; float* area; } patch_rec;

static patch_rec rec_build(const orc_mesh* m) {
    patch_rec r;
    r.cen = (float*)malloc(sizeof(float) * 12 * (size_t)m->N);
    r.sa = (float*)malloc(sizeof(float) * 4 * (size_t)m->N);
    r.nrm = (float*)malloc(sizeof(float) * 3 * (size_t)m->N);
    r.area = (float*)malloc(sizeof(float) * (size_t)m->N);
    orc_patch_records(m, r.cen, r.sa, r.nrm, r.area);
    return r;
}
static void rec_free(patch_rec* r) { free(r->cen); free(r->sa); free(r->nrm); free(r->area); }

/* the integrand from per-patch records; same operations as the literal form */
static inline float integrand_rec(const patch_rec* r, int i, int j) {
    v3 on = v3_ld(r->nrm + 3 * (long)i), dn = v3_ld(r->nrm + 3 * (long)j);
    float ff = 0;
    for (int s = 0; s < 4; s++) {
        v3 op = v3_ld(r->cen + 12 * (long)i + 3 * s);
        for (int t = 0; t < 4; t++) {
            v3 dp = v3_ld(r->cen + 12 * (long)j + 3 * t);
            ff = ff + point_ff(op, on, dp, dn, r->sa[4 * (long)i + s] * r->sa[4 * (long)j + t]);
        }
    }
    return ff / r->area[i];
}

/* parallellism.cu:98-108: the value stored is F if F > 0 else 0 (NaN -> 0) */
static inline float stored(float f) { return f > 0.0f ? f : 0.0f; }

void orc_integrand_rows(const orc_mesh* m, int row0, int nrows, float* out) {
    patch_rec r = rec_build(m);
#pragma omp parallel for schedule(dynamic, 1)
    for (int i = 0; i < nrows; i++)
        for (int j = 0; j < m->N; j++)
            out[(long)i * m->N + j] = stored(integrand_rec(&r, row0 + i, j));
    rec_free(&r);
}

/* The CUDA twin of the integrand (visual studio/parallellism.cu:197-207), which the reference runs when cuda_on = true:
 * CUDART_PI is a double, so the denominator powf(length,2)*CUDART_PI, the division and the product with `surface` are
 * evaluated in double and rounded to float once, where triangle_math.cpp:49-58 (M_PIf) rounds after every operation.
 * The build follows the CPU file for both rules; these two functions exist to MEASURE that deviation
 * (tests/test_oracle_cpu.py::test_cuda_twin_arithmetic_deviation). */
static inline float point_ff_cuda_twin(v3 opos, v3 onrm, v3 dpos, v3 dnrm, float surface) {
    float ff = 0;
    float dot1 = v3_dot(onrm, v3_normalize(v3_sub(dpos, opos)));
    float dot2 = v3_dot(dnrm, v3_normalize(v3_sub(opos, dpos)));
    if (dot1 > 0 && dot2 > 0) {
        float length = v3_length(v3_sub(dpos, opos));
        ff = (float)((((double)(dot1 * dot2)) / ((double)(length * length) * 3.1415926535897931e+0)) * (double)surface);
    }
    return ff;
}

void orc_integrand_rows_cuda_twin(const orc_mesh* m, int row0, int nrows, float* out) {
    patch_rec r = rec_build(m);
#pragma omp parallel for schedule(dynamic, 1)
    for (int i = 0; i < nrows; i++)
        for (int j = 0; j < m->N; j++) {
            const int a = row0 + i;
            v3 on = v3_ld(r.nrm + 3 * (long)a), dn = v3_ld(r.nrm + 3 * (long)j);
            float ff = 0;
            for (int s = 0; s < 4; s++) {
                v3 op = v3_ld(r.cen + 12 * (long)a + 3 * s);
                for (int t = 0; t < 4; t++) {
                    v3 dp = v3_ld(r.cen + 12 * (long)j + 3 * t);
                    ff = ff + point_ff_cuda_twin(op, on, dp, dn, r.sa[4 * (long)a + s] * r.sa[4 * (long)j + t]);
                }
            }
            out[(long)i * m->N + j] = stored(ff / r.area[a]);
        }
    rec_free(&r);
}

/* triangle_math.cpp:3-9: a + u*(b-a) + v*(c-a) */
static inline v3 uv2xyz(const orc_mesh* m, int tri, float u, float v) {
    v3 a = vert(m, tri, 0), b = vert(m, tri, 1), c = vert(m, tri, 2);
    return v3_add(v3_add(a, v3_scale(v3_sub(b, a), u)), v3_scale(v3_sub(c, a), v));
}
void orc_uv2xyz(const orc_mesh* m, int tri, float u, float v, float out[3]) {
    v3 p = uv2xyz(m, tri, u, v);
    out[0] = p.x; out[1] = p.y; out[2] = p.z;
}

/* ---- visibility: the oracle's definition of the closed-source ray engine ---------------
 * (PARITY UNPINNED here: OptiX Prime is absent; this is a definition, not a restatement.)
 *
 * A triangle k "is hit" by a ray segment [0, tmax] iff
 *   (1) the segment passes the fp32 slab test against k's padded bounding box
 *       (box of a, a+e1, a+e2, padded by pad = 1e-4f*extent + 1e-30f, extent = largest side of
 *       the bounding box of all referenced vertices), and
 *   (2) the two-sided Moller-Trumbore test on (a, e1=b-a, e2=c-a) reports u>=0, v>=0,
 *       u+v<=1 and t>0 (the reference tests hit.t > 0, OptixPrimeFunctionality.cpp:208;
 *       det==0 gives inf/NaN that fail those tests).
 * Gate (1) makes "hit" a geometric notion: without it the fp32 noise of (2) on sliver or
 * grazing triangles reports hits arbitrarily far from the triangle, which no spatial index
 * could reproduce.  The slab test is monotone under box enlargement in fp32, so culling by
 * enclosing boxes (the oracle's BVH below, the product's LBVH) is exact, not approximate.
 *
 * Ray k of pair (lo,hi) is VISIBLE iff hi is hit on [0,inf) at t_hi and no triangle is hit on
 * [0,t_hi] with t < t_hi, or t == t_hi and a lower index -- i.e. the closest hit, ties to the
 * lowest id, is hi (OptixPrimeFunctionality.cpp:205-211).  Every triangle takes part, the source
 * included (OptiX has no tmin here; the ray starts 1e-6 along its direction, :194). */
typedef struct { v3 a, e1, e2; float lo[3], hi[3]; } tri_rec;

static inline int tri_mt(v3 o, v3 d, const tri_rec* T, float* t_out) {
    v3 p = v3_cross(d, T->e2);
    float det = v3_dot(T->e1, p);
    float inv = 1.0f / det;
    v3 tv = v3_sub(o, T->a);
    float u = v3_dot(tv, p) * inv;
    v3 q = v3_cross(tv, T->e1);
    float v = v3_dot(d, q) * inv;
    float t = v3_dot(T->e2, q) * inv;
    if (u >= 0.0f && v >= 0.0f && (u + v) <= 1.0f && t > 0.0f) { *t_out = t; return 1; }
    return 0;
}

/* a zero direction component must not turn (plane - origin) * inv into 0*inf = NaN */
static inline float safe_inv(float d) { return d == 0.0f ? 3.0e38f : 1.0f / d; }

/* fp32 slab test of the segment [0,tmax]; every operation individually rounded, this order:
 * max(tn,0) <= min(tf*1.00001f, tmax).  fminf/fmaxf drop NaNs like the GPU's v_min/v_max. */
static inline int slab_hit(const float lo[3], const float hi[3], v3 o, v3 inv, float tmax) {
    float t0 = (lo[0] - o.x) * inv.x, t1 = (hi[0] - o.x) * inv.x;
    float tn = fminf(t0, t1), tf = fmaxf(t0, t1);
    t0 = (lo[1] - o.y) * inv.y; t1 = (hi[1] - o.y) * inv.y;
    tn = fmaxf(tn, fminf(t0, t1)); tf = fminf(tf, fmaxf(t0, t1));
    t0 = (lo[2] - o.z) * inv.z; t1 = (hi[2] - o.z) * inv.z;
    tn = fmaxf(fmaxf(tn, fminf(t0, t1)), 0.0f); tf = fminf(tf, fmaxf(t0, t1));
    return tn <= fminf(tf * 1.00001f, tmax);
}

static inline int tri_hit(v3 o, v3 d, v3 inv, const tri_rec* T, float tmax, float* t_out) {
    return slab_hit(T->lo, T->hi, o, inv, tmax) && tri_mt(o, d, T, t_out);
}

static float mesh_pad(const orc_mesh* m) {
    float lo[3] = { INFINITY, INFINITY, INFINITY }, hi[3] = { -INFINITY, -INFINITY, -INFINITY };
    for (long k = 0; k < 3L * m->N; k++) {
        const float* p = m->vertices + 3 * (long)m->tri_v[k];
        for (int a = 0; a < 3; a++) { lo[a] = fminf(lo[a], p[a]); hi[a] = fmaxf(hi[a], p[a]); }
    }
    float ext = fmaxf(hi[0] - lo[0], fmaxf(hi[1] - lo[1], hi[2] - lo[2]));
    return 1e-4f * ext + 1e-30f;
}

static tri_rec* tris_build(const orc_mesh* m) {
    tri_rec* T = (tri_rec*)malloc(sizeof(tri_rec) * (size_t)(m->N > 0 ? m->N : 1));
    const float pad = mesh_pad(m);
    for (int k = 0; k < m->N; k++) {
        v3 a = vert(m, k, 0), b = vert(m, k, 1), c = vert(m, k, 2);
        T[k].a = a; T[k].e1 = v3_sub(b, a); T[k].e2 = v3_sub(c, a);
        /* the box of the triangle the ray test sees: a, a+e1, a+e2 */
        float p[3][3] = { { a.x, a.y, a.z },
                          { a.x + T[k].e1.x, a.y + T[k].e1.y, a.z + T[k].e1.z },
                          { a.x + T[k].e2.x, a.y + T[k].e2.y, a.z + T[k].e2.z } };
        for (int x = 0; x < 3; x++) {
            T[k].lo[x] = fminf(p[0][x], fminf(p[1][x], p[2][x])) - pad;
            T[k].hi[x] = fmaxf(p[0][x], fmaxf(p[1][x], p[2][x])) + pad;
        }
    }
    return T;
}

/* closest hit on [0,inf), ties to the lowest id */
static inline int closest_brute(const tri_rec* T, int N, v3 o, v3 d, float* t_out) {
    int best = -1; float bt = 0;
    v3 inv = { safe_inv(d.x), safe_inv(d.y), safe_inv(d.z) };
    for (int k = 0; k < N; k++) {
        float t;
        if (tri_hit(o, d, inv, &T[k], INFINITY, &t) && (best < 0 || t < bt)) { best = k; bt = t; }
    }
    *t_out = best >= 0 ? bt : -1.0f;
    return best;
}

int orc_closest_hit(const orc_mesh* m, const float org[3], const float dir[3], float* t_out) {
    tri_rec* T = tris_build(m);
    int id = closest_brute(T, m->N, v3_ld(org), v3_ld(dir), t_out);
    free(T);
    return id;
}

/* ray k of pair (lo,hi): OptixPrimeFunctionality.cpp:191-196 (optix::normalize
 * = v * (1/sqrtf(dot(v,v)))) */
static inline void make_ray(const orc_mesh* m, int lo, int hi, float u, float v, float eps,
                            v3* org, v3* dir) {
    v3 o = uv2xyz(m, lo, u, v);
    v3 dst = uv2xyz(m, hi, u, v);
    v3 dv = v3_sub(dst, o);
    v3 dn = v3_scale(dv, 1.0f / sqrtf(v3_dot(dv, dv)));
    *org = v3_add(o, v3_scale(dn, eps));
    *dir = dn;
}

static int vis_count_brute(const orc_mesh* m, const tri_rec* T, int lo, int hi,
                           const float* uv, int K, float eps) {
    int cnt = 0;
    for (int k = 0; k < K; k++) {
        v3 o, d; float t_hi;
        make_ray(m, lo, hi, uv[2 * k], uv[2 * k + 1], eps, &o, &d);
        v3 inv = { safe_inv(d.x), safe_inv(d.y), safe_inv(d.z) };
        if (!tri_hit(o, d, inv, &T[hi], INFINITY, &t_hi)) continue;       /* the destination is not hit at all */
        int blocked = 0;
        for (int j = 0; j < m->N && !blocked; j++) {
            float t;
            if (tri_hit(o, d, inv, &T[j], t_hi, &t) && (t < t_hi || (t == t_hi && j < hi))) blocked = 1;
        }
        if (!blocked) cnt++;
    }
    return cnt;
}

int orc_visibility_count(const orc_mesh* m, int lo, int hi, const float* uv, int K, float eps) {
    tri_rec* T = tris_build(m);
    int c = vis_count_brute(m, T, lo, hi, uv, K, eps);
    free(T);
    return c;
}

/* ---- the oracle's own BVH (median split; node boxes = unions of the triangles' padded boxes,
 * so culling is exact) --------------------------------------------------------------------- */
typedef struct { float lo[3], hi[3]; int left, right, first, count; } bnode;
typedef struct { bnode* nodes; int n_nodes; int* order; const tri_rec* T; } cbvh;

static const float* g_sort_key;
static int cmp_key(const void* a, const void* b) {
    float ka = g_sort_key[*(const int*)a], kb = g_sort_key[*(const int*)b];
    return (ka > kb) - (ka < kb);
}

static int bvh_build_rec(cbvh* B, int first, int count, float* cen) {
    int id = B->n_nodes++;
    bnode* nd = &B->nodes[id];
    for (int a = 0; a < 3; a++) { nd->lo[a] = INFINITY; nd->hi[a] = -INFINITY; }
    for (int i = first; i < first + count; i++) {
        const tri_rec* t = &B->T[B->order[i]];
        for (int a = 0; a < 3; a++) {
            nd->lo[a] = fminf(nd->lo[a], t->lo[a]);
            nd->hi[a] = fmaxf(nd->hi[a], t->hi[a]);
        }
    }
    nd->first = first; nd->count = count; nd->left = nd->right = -1;
    if (count <= 4) return id;
    int ax = 0; float ext = -1;
    float clo[3] = { INFINITY, INFINITY, INFINITY }, chi[3] = { -INFINITY, -INFINITY, -INFINITY };
    for (int i = first; i < first + count; i++)
        for (int a = 0; a < 3; a++) {
            float c = cen[3 * (long)B->order[i] + a];
            clo[a] = fminf(clo[a], c); chi[a] = fmaxf(chi[a], c);
        }
    for (int a = 0; a < 3; a++) if (chi[a] - clo[a] > ext) { ext = chi[a] - clo[a]; ax = a; }
    {   /* sort this range by centroid along ax */
        float* key = (float*)malloc(sizeof(float) * (size_t)count);
        int* idx = (int*)malloc(sizeof(int) * (size_t)count);
        int* tmp = (int*)malloc(sizeof(int) * (size_t)count);
        for (int i = 0; i < count; i++) { key[i] = cen[3 * (long)B->order[first + i] + ax]; idx[i] = i; }
        g_sort_key = key;
        qsort(idx, (size_t)count, sizeof(int), cmp_key);
        for (int i = 0; i < count; i++) tmp[i] = B->order[first + idx[i]];
        memcpy(B->order + first, tmp, sizeof(int) * (size_t)count);
        free(idx); free(tmp); free(key);
    }
    int half = count / 2;
    int l = bvh_build_rec(B, first, half, cen);
    int r = bvh_build_rec(B, first + half, count - half, cen);
    B->nodes[id].left = l; B->nodes[id].right = r;
    return id;
}

static cbvh bvh_build(const tri_rec* T, int N) {
    cbvh B;
    B.T = T; B.n_nodes = 0;
    B.nodes = (bnode*)malloc(sizeof(bnode) * (size_t)(2 * N + 1));
    B.order = (int*)malloc(sizeof(int) * (size_t)(N > 0 ? N : 1));
    float* cen = (float*)malloc(sizeof(float) * 3 * (size_t)(N > 0 ? N : 1));
    for (int k = 0; k < N; k++) {
        for (int a = 0; a < 3; a++) cen[3 * (long)k + a] = 0.5f * (T[k].lo[a] + T[k].hi[a]);
        B.order[k] = k;
    }
    if (N > 0) bvh_build_rec(&B, 0, N, cen);
    free(cen);
    return B;
}
static void bvh_free(cbvh* B) { free(B->nodes); free(B->order); }

/* is any triangle hit on [0,t_hi] that precedes `hi`?  (the same predicate as the brute force) */
static int blocked_bvh(const cbvh* B, v3 o, v3 d, v3 inv, float t_hi, int hi) {
    if (B->n_nodes == 0) return 0;
    int stack[128]; int sp = 0; stack[sp++] = 0;
    while (sp) {
        const bnode* nd = &B->nodes[stack[--sp]];
        if (!slab_hit(nd->lo, nd->hi, o, inv, t_hi)) continue;
        if (nd->left < 0) {
            for (int i = nd->first; i < nd->first + nd->count; i++) {
                int k = B->order[i]; float t;
                if (tri_hit(o, d, inv, &B->T[k], t_hi, &t) && (t < t_hi || (t == t_hi && k < hi))) return 1;
            }
        } else { stack[sp++] = nd->left; stack[sp++] = nd->right; }
    }
    return 0;
}

static int vis_count_bvh(const orc_mesh* m, const cbvh* B, int lo, int hi,
                         const float* uv, int K, float eps) {
    int cnt = 0;
    for (int k = 0; k < K; k++) {
        v3 o, d; float t_hi;
        make_ray(m, lo, hi, uv[2 * k], uv[2 * k + 1], eps, &o, &d);
        v3 inv = { safe_inv(d.x), safe_inv(d.y), safe_inv(d.z) };
        if (!tri_hit(o, d, inv, &B->T[hi], INFINITY, &t_hi)) continue;
        if (!blocked_bvh(B, o, d, inv, t_hi, hi)) cnt++;
    }
    return cnt;
}

/* ---- assembly ----------------------------------------------------------------
 * GPU-path rule (OptixPrimeFunctionality.cpp:186-218): for row<col with stored
 * integrand(row,col) > 0 trace K rays row->col, V = count/K; if V > 0 both
 * F[row][col] = V*Fu[row][col] and F[col][row] = V*Fu[col][row] are emitted
 * (double product of two floats, cast to float by SpMat<float> = fp32 product).
 * CPU-path rule (:311-366): F[row][col] = Fu*V, and when that is > 0 the
 * reverse entry is (A_row*F[row][col])/A_col. */
static int assemble_impl(const orc_mesh* m, const float* uv, int K, float eps, int rule,
                         int row0, int nrows, float* F, uint8_t* vis, int threads, int use_bvh, const int32_t* row_list) {
    const int N = m->N;
    patch_rec r = rec_build(m);
    tri_rec* T = tris_build(m);
    cbvh B; if (use_bvh) B = bvh_build(T, N);
    int used = 1;
#ifdef _OPENMP
    if (threads <= 0) threads = omp_get_max_threads();
    used = threads;
#else
    (void)threads;
#endif
#pragma omp parallel for schedule(dynamic, 1) num_threads(used)
    for (int ri = 0; ri < nrows; ri++) {
        int i = row_list ? row_list[ri] : row0 + ri;
        for (int j = 0; j < N; j++) {
            float out = 0.0f; uint8_t vc = 255;
            if (i != j) {
                int lo = i < j ? i : j, hi = i < j ? j : i;
                float fu_lohi = stored(integrand_rec(&r, lo, hi));
                if (rule == ORC_RULE_RECIPROCITY || fu_lohi > 0.0f) {
                    /* the CPU path traces every pair; a pair with Fu == 0 stores nothing
                     * either way, so it is skipped here and reported untraced */
                    if (fu_lohi > 0.0f) {
                        int c = use_bvh ? vis_count_bvh(m, &B, lo, hi, uv, K, eps)
                                        : vis_count_brute(m, T, lo, hi, uv, K, eps);
                        vc = (uint8_t)c;
                        float V = (float)c / (float)K;
                        if (rule == ORC_RULE_INTEGRAND) {
                            if (V > 0.0f) out = V * stored(integrand_rec(&r, i, j));
                        } else {
                            float f_lohi = fu_lohi * V;
                            if (f_lohi > 0.0f)
                                out = (i == lo) ? f_lohi : (r.area[lo] * f_lohi) / r.area[hi];
                        }
                    }
                }
            }
            F[(long)ri * N + j] = out;
            if (vis) vis[(long)ri * N + j] = vc;
        }
    }
    if (use_bvh) bvh_free(&B);
    free(T);
    rec_free(&r);
    return used;
}

int orc_assemble_rows(const orc_mesh* m, const float* uv, int K, float eps, int rule,
                      int row0, int nrows, float* F, uint8_t* vis, int threads) {
    return assemble_impl(m, uv, K, eps, rule, row0, nrows, F, vis, threads, 0, NULL);
}
int orc_assemble_rows_bvh(const orc_mesh* m, const float* uv, int K, float eps, int rule,
                          int row0, int nrows, float* F, uint8_t* vis, int threads) {
    return assemble_impl(m, uv, K, eps, rule, row0, nrows, F, vis, threads, 1, NULL);
}
/* the same for an arbitrary list of rows (records and BVH built once for all of them): F / vis hold nrows rows in list order */
int orc_assemble_row_list_bvh(const orc_mesh* m, const float* uv, int K, float eps, int rule,
                              const int32_t* rows, int nrows, float* F, uint8_t* vis, int threads) {
    for (int k = 0; k < nrows; k++) if (rows[k] < 0 || rows[k] >= m->N) return -1;
    return assemble_impl(m, uv, K, eps, rule, 0, nrows, F, vis, threads, 1, rows);
}

/* ---- solver -------------------------------------------------------------------
 * Lightning.h:196-226 (spectral), :342-349 (RGB = diagonal M), :419-424 (BW =
 * S 1, M = [1]).  Eigen's column-major sparse product adds F(i,j)*x[j] into
 * y[i] for j ascending (Eigen/src/SparseCore/SparseDenseProduct.h:199-207);
 * zeros contribute nothing, so a dense j-ascending fp32 sum is the same value. */
void orc_sweep_rows(int N, int S, const float* F, long ldF, int row0, int nrows,
                    const float* M, const int32_t* mat, const float* Rin,
                    float* Rout, float* B, int threads) {
    int used = 1;
#ifdef _OPENMP
    if (threads <= 0) threads = omp_get_max_threads();
    used = threads;
#else
    (void)threads;
#endif
#pragma omp parallel for schedule(static) num_threads(used)
    for (int r = 0; r < nrows; r++) {
        float G[64];
        for (int s = 0; s < S; s++) G[s] = 0.0f;
        const float* Fr = F + (long)r * ldF;
        for (int j = 0; j < N; j++) {
            float f = Fr[j];
            if (f == 0.0f) continue;
            const float* x = Rin + (long)j * S;
            for (int s = 0; s < S; s++) G[s] = G[s] + f * x[s];
        }
        const float* Mi = M + (long)mat[row0 + r] * S * S;
        for (int so = 0; so < S; so++) {
            float acc = 0.0f;
            for (int s = 0; s < S; s++) acc = acc + Mi[so * S + s] * G[s];
            Rout[(long)r * S + so] = acc;
            B[(long)r * S + so] = B[(long)r * S + so] + acc;
        }
    }
}

void orc_residual_sums(int N, int S, const float* R, double* sums) {
    for (int s = 0; s < S; s++) sums[s] = 0.0;
    for (long i = 0; i < N; i++)
        for (int s = 0; s < S; s++) sums[s] += (double)R[i * S + s];
}

int orc_converge(int N, int S, const float* F, const float* M, const int32_t* mat,
                 float* R, float* B, float threshold, int per_bin, int max_iters, int threads) {
    float* Rn = (float*)malloc(sizeof(float) * (size_t)N * S);
    double sums[64];
    int it = 0;
    for (;;) {
        orc_residual_sums(N, S, R, sums);
        int go = 0;
        if (per_bin) { for (int s = 0; s < S; s++) if (sums[s] > (double)threshold) go = 1; }
        else { double t = 0; for (int s = 0; s < S; s++) t += sums[s]; go = t > (double)threshold; }
        if (!go || it >= max_iters) break;
        orc_sweep_rows(N, S, F, N, 0, N, M, mat, R, Rn, B, threads);
        memcpy(R, Rn, sizeof(float) * (size_t)N * S);
        it++;
    }
    free(Rn);
    return it;
}

/* ---- display colour ---------------------------------------------------------- */
/* daisy_color::cie1931WavelengthToXYZFit, vs/color.h:14-45 (Wyman, Sloan, Shirley 2013): evaluated
 * in double, truncated to float on return */
static double lobe(double w, double mu, double s_lo, double s_hi) {
    double t = (w - mu) * ((w < mu) ? s_lo : s_hi);
    return exp(-0.5 * t * t);
}
void orc_xyz_fit(double w, float out[3]) {
    double x = 0.362 * lobe(w, 442.0, 0.0624, 0.0374) + 1.056 * lobe(w, 599.8, 0.0264, 0.0323)
             - 0.065 * lobe(w, 501.1, 0.0490, 0.0382);
    double y = 0.821 * lobe(w, 568.8, 0.0213, 0.0247) + 0.286 * lobe(w, 530.9, 0.0613, 0.0322);
    double z = 1.217 * lobe(w, 437.0, 0.0845, 0.0278) + 0.681 * lobe(w, 459.0, 0.0385, 0.0725);
    out[0] = (float)x; out[1] = (float)y; out[2] = (float)z;
}

/* mode 0: BWLightning::get_color_of_patch (vs/Lightning.h:406-408); 1: RGBLightning (:332-334);
 * 2: SpectralLightning::update_color_cache (:168-183) with XYZToRGB (vs/color.h:48-52).
 * B: N x S patch-major; rgb: N x 3. */
void orc_patch_colors(int N, int S, const float* B, int mode, const float* xyz, float* rgb) {
    for (long i = 0; i < N; i++) {
        const float* b = B + i * S;
        float r, g, bl;
        if (mode == 0) { r = g = bl = b[0]; }
        else if (mode == 1) { r = b[0]; g = b[1]; bl = b[2]; }
        else {
            float X = 0.0f, Y = 0.0f, Z = 0.0f;
            for (int j = 0; j < S; j++) {
                X = X + xyz[3 * j + 0] * b[j];
                Y = Y + xyz[3 * j + 1] * b[j];
                Z = Z + xyz[3 * j + 2] * b[j];
            }
            r  = 3.240479f * X - 1.537150f * Y - 0.498535f * Z;
            g  = -0.969256f * X + 1.875991f * Y + 0.041556f * Z;
            bl = 0.055648f * X - 0.204043f * Y + 1.057311f * Z;
            float mx = fmaxf(r, fmaxf(g, bl));
            if (mx > 1) { r = r / mx; g = g / mx; bl = bl / mx; }
        }
        rgb[3 * i] = r; rgb[3 * i + 1] = g; rgb[3 * i + 2] = bl;
    }
}

/* corner values of Drawer::interpolate (vs/Drawer.cpp:161-186): mean colour of the patches around each
 * vertex, summed in adjacency order; vertices no triangle uses (never drawn) give 0 */
void orc_vertex_colors(int V, const int32_t* off, const int32_t* adj, const float* rgb, float* out) {
    for (long v = 0; v < V; v++) {
        float a[3] = { 0.0f, 0.0f, 0.0f };
        for (int k = off[v]; k < off[v + 1]; k++)
            for (int c = 0; c < 3; c++) a[c] = a[c] + rgb[3 * (long)adj[k] + c];
        int n = off[v + 1] - off[v];
        for (int c = 0; c < 3; c++) out[3 * v + c] = n > 0 ? a[c] / (float)n : 0.0f;
    }
}

int orc_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* debug aid for tests: per ray of a pair, list triangles whose Moller-Trumbore test alone would block it */
#include <stdio.h>
void orc_debug_pair(const orc_mesh* m, int lo, int hi, const float* uv, int K, float eps) {
    tri_rec* T = tris_build(m);
    for (int k = 0; k < K; k++) {
        v3 o, d; float t_hi = -1;
        make_ray(m, lo, hi, uv[2 * k], uv[2 * k + 1], eps, &o, &d);
        v3 inv = { safe_inv(d.x), safe_inv(d.y), safe_inv(d.z) };
        int gate_hi = slab_hit(T[hi].lo, T[hi].hi, o, inv, INFINITY), mt_hi = tri_mt(o, d, &T[hi], &t_hi);
        if (!(gate_hi && mt_hi)) { printf("ray %d: target gate %d mt %d\n", k, gate_hi, mt_hi); continue; }
        for (int j = 0; j < m->N; j++) {
            float t;
            if (tri_mt(o, d, &T[j], &t) && (t < t_hi || (t == t_hi && j < hi)))
                printf("ray %d: tri %d mt-blocks t=%.9g t_hi=%.9g gate=%d dir=(%.9g %.9g %.9g) org=(%.9g %.9g %.9g) box lo=(%.9g %.9g %.9g) hi=(%.9g %.9g %.9g)\n",
                       k, j, t, t_hi, slab_hit(T[j].lo, T[j].hi, o, inv, t_hi), d.x, d.y, d.z, o.x, o.y, o.z,
                       T[j].lo[0], T[j].lo[1], T[j].lo[2], T[j].hi[0], T[j].hi[1], T[j].hi[2]);
        }
    }
    free(T);
}
