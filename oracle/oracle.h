/*
 * oracle.h -- CPU restatement of DaisyRiot's radiosity hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under daisyriot_amd/ may include, link or
 * call this.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg use it, and there only as the checker / the timed CPU baseline.
 *
 * Pinning status (see DESIGN.md "Oracle"):
 *   - integrand + solver arithmetic: pinned by the reference's single KAT
 *     (vs/unittest1.cpp:15, area == 0.5f) and by oracle/_ref/ref_check, which
 *     evaluates the same expressions through the reference's vendored glm
 *     0.9.8.4 and Eigen 3.2.10 sources (bit-exact for the integrand and the
 *     sparse mat-vec).
 *   - visibility: PARITY UNPINNED at the ray-engine boundary.  The reference's
 *     closest-hit query is closed-source NVIDIA OptiX Prime 4.1.1
 *     (vs/OptixPrimeFunctionality.cpp:66-81), absent from /root/reference, and
 *     no reference test records its results.  The oracle defines it as an
 *     exact brute-force closest hit: a triangle is hit iff the ray segment
 *     passes the fp32 slab test against its padded bounding box AND the
 *     Moller-Trumbore test (fp32, every operation individually rounded);
 *     lowest triangle id wins ties (see oracle.c "visibility").
 *
 * All arithmetic is IEEE-754 binary32 unless a comment says otherwise, in the
 * written order, with contraction off (-ffp-contract=off).
 */
#ifndef DAISYRIOT_ORACLE_H
#define DAISYRIOT_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Geometry handed to the path: exactly the MeshS / SimpleMesh arrays
 * (vs/MeshS.h:14-20, vs/Defines.h:14-23), 0-based indices. */
typedef struct {
    const float*   vertices;   /* 3*V */
    int            V;
    const float*   normals;    /* 3*Nn */
    int            Nn;
    const int32_t* tri_v;      /* 3*N  (TriangleIndex::vertex) */
    const int32_t* tri_n;      /* 3*N  (TriangleIndex::normal) */
    int            N;
} orc_mesh;

/* which rule produces the reverse entry F[col][row] */
enum {
    ORC_RULE_INTEGRAND   = 0,  /* GPU path: OptixPrimeFunctionality.cpp:6-34,169-218 */
    ORC_RULE_RECIPROCITY = 1   /* CPU path: OptixPrimeFunctionality.cpp:311-366      */
};

/* triangle_math.cpp:31-35 (the reference's one known answer is on this) */
float orc_surface(const float a[3], const float b[3], const float c[3]);

/* Step-by-step restatement of OptixPrimeFunctionality.cpp:133-161 (integrand
 * only; visibility excluded), recomputing everything per pair like the
 * reference does. */
float orc_p2p_integrand_literal(const orc_mesh* m, int origin, int dest);

/* Per-patch quantities the integrand uses (same arithmetic, computed once):
 * sub-centroids cen[N][4][3], sub-areas sub_area[N][4], normal nrm[N][3],
 * area[N]. */
void orc_patch_records(const orc_mesh* m, float* cen, float* sub_area,
                       float* nrm, float* area);

/* Stored integrand rows: out[(r-row0)*N + c] = F>0 ? F : 0 (parallellism.cu:98-108). */
void orc_integrand_rows(const orc_mesh* m, int row0, int nrows, float* out);
/* the same with the arithmetic of the reference's CUDA kernel (parallellism.cu:197-207: double pi, double product) */
void orc_integrand_rows_cuda_twin(const orc_mesh* m, int row0, int nrows, float* out);

/* uv2xyz, triangle_math.cpp:3-9 */
void orc_uv2xyz(const orc_mesh* m, int tri, float u, float v, float out[3]);

/* Brute-force closest hit over all N triangles on [0,inf) (box-gated Moller-Trumbore, ties to the
 * lowest id). Returns triangle id or -1; *t_out receives t (or -1). */
int orc_closest_hit(const orc_mesh* m, const float org[3], const float dir[3],
                    float* t_out);

/* Number of the K rays from patch `lo` to patch `hi` whose closest hit is `hi`
 * (OptixPrimeFunctionality.cpp:191-211 / 253-270). */
int orc_visibility_count(const orc_mesh* m, int lo, int hi, const float* uv,
                         int K, float origin_eps);

/* Full rows of F. F[(r-row0)*N + c].  vis (nullable) receives the ray count
 * per entry, 255 where the pair was not traced.  Uses OpenMP over rows when
 * built with it; threads<=0 means all. Returns threads used. */
int orc_assemble_rows(const orc_mesh* m, const float* uv, int K,
                      float origin_eps, int rule, int row0, int nrows,
                      float* F, uint8_t* vis, int threads);

/* Same, visibility through the oracle's own median-split BVH instead of brute
 * force (identical results; tests assert that). For larger scenes and the
 * timed CPU baseline. */
int orc_assemble_rows_bvh(const orc_mesh* m, const float* uv, int K,
                          float origin_eps, int rule, int row0, int nrows,
                          float* F, uint8_t* vis, int threads);
/* ... for an arbitrary list of rows, the BVH built once (bench.py's CPU baseline: rows spread over the matrix) */
int orc_assemble_row_list_bvh(const orc_mesh* m, const float* uv, int K, float eps, int rule,
                              const int32_t* rows, int nrows, float* F, uint8_t* vis, int threads);

/* One light pass for rows [row0,row0+nrows) (Lightning.h:196-226, 342-349,
 * 419-424): G_s[i] = sum_j F[i][j]*R[j][s] (j ascending, fp32);
 * Rout[i][:] = M[mat[i]] * G[i][:];  B[i][:] += Rout[i][:].
 * F: nrows x ldF row-major (row r of F is global row row0+r).
 * Rin: N x S (all patches), Rout/B: nrows x S (local rows). */
void orc_sweep_rows(int N, int S, const float* F, long ldF, int row0,
                    int nrows, const float* M, const int32_t* mat,
                    const float* Rin, float* Rout, float* B, int threads);

/* sum over patches and bins, per bin into sums[S] (double accumulate). */
void orc_residual_sums(int N, int S, const float* R, double* sums);

/* converge loop (Lightning.h:145-151 spectral: total sum > threshold;
 * :336-340 RGB: any bin sum > threshold). F dense N x N. Returns passes. */
int orc_converge(int N, int S, const float* F, const float* M,
                 const int32_t* mat, float* R, float* B, float threshold,
                 int per_bin, int max_iters, int threads);

/* Display colour: vs/color.h:14-52, vs/Lightning.h:168-183/332-334/406-408,
 * vs/Drawer.cpp:161-186.  mode 0 BW, 1 RGB, 2 spectral (xyz = S x 3 fit values). */
void orc_xyz_fit(double wavelength, float out[3]);
void orc_patch_colors(int N, int S, const float* B, int mode, const float* xyz, float* rgb);
void orc_vertex_colors(int V, const int32_t* off, const int32_t* adj, const float* rgb, float* out);

int orc_num_threads(void);

#ifdef __cplusplus
}
#endif
#endif
