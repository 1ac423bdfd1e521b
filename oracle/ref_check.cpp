/*
 * ref_check.cpp -- evaluates the hot path's arithmetic through the REFERENCE'S
 * OWN vendored third-party sources (glm 0.9.8.4 and Eigen 3.2.10 under
 * /root/reference/libraries), compiled where they lie.  Output goes only to
 * oracle/_ref/ (git-ignored).  It exists to pin oracle.c:
 *   - the integrand is evaluated with glm::normalize / dot / cross / length at
 *     the call sites of vs/triangle_math.cpp:11-74 and
 *     vs/OptixPrimeFunctionality.cpp:133-161 -> must equal oracle.c bit for bit;
 *   - the light pass is evaluated with Eigen::SparseMatrix<float> * VectorXf,
 *     cwiseProduct and MatrixXf * VectorXf exactly as vs/Lightning.h:196-226 and
 *     :342-349 write them -> sparse product must equal oracle.c bit for bit.
 *
 *   - OBJ/MTL parsing goes through the vendored tinyobjloader exactly as vs/MeshS.cpp:25-31 calls it;
 *   - config.ini parsing and the RGB -> spectrum table lookup go through the reference's own
 *     "visual studio/INIReader.h" (single header) and "visual studio/rgb2spec.cpp" (needs nothing but libc),
 *     compiled as they lie;
 *   - the display colour is evaluated through the reference's own
 *     "visual studio/color.h" (a first-party header that needs nothing but glm),
 *     included from where it lies, at the call site of vs/Lightning.h:168-183.
 *
 * The reference's other first-party files are NOT built: triangle_math.cpp
 * needs <OptiX_world.h> and Lightning.h needs the OptiX Prime class; the OptiX
 * SDK is absent from this image and is not substituted (DESIGN.md "Oracle").
 *
 * TEST INFRASTRUCTURE ONLY.
 */
#include <cmath>
#include <cstdio>
#include <cstdint>
#include <vector>

#include <glm/glm.hpp>
#include <Eigen/Dense>
#include <Eigen/Sparse>

#include "color.h"      /* $(REF)/visual studio/color.h */
#include "INIReader.h"  /* $(REF)/visual studio/INIReader.h: the reference's single-header ini parser, as it lies */
#include "rgb2spec.h"   /* $(REF)/visual studio/rgb2spec.h; rgb2spec.cpp is compiled beside this file (oracle/Makefile) */
#include <string>
#define TINYOBJLOADER_IMPLEMENTATION
#include <tinyobjloader/tiny_obj_loader.h>   /* $(REF)/libraries/tinyobjloader: the vendored OBJ/MTL parser MeshS.cpp calls */

namespace {

struct Mesh {
    const float* vtx; const float* nrm; const int32_t* tv; const int32_t* tn;
    glm::vec3 v(int tri, int c) const { const float* p = vtx + 3 * (long)tv[3 * (long)tri + c]; return glm::vec3(p[0], p[1], p[2]); }
    glm::vec3 n(int tri, int c) const { const float* p = nrm + 3 * (long)tn[3 * (long)tri + c]; return glm::vec3(p[0], p[1], p[2]); }
};

float area_of(const glm::vec3& a, const glm::vec3& b, const glm::vec3& c) {
    glm::vec3 ab = b - a, ac = c - a;
    return 0.5 * glm::length(glm::cross(ab, ac));
}

struct Quarter { glm::vec3 p[3]; };

void quarters(const Mesh& m, int tri, Quarter q[4]) {
    glm::vec3 a = m.v(tri, 0), b = m.v(tri, 1), c = m.v(tri, 2);
    glm::vec3 mab = ((b - a) / 2.0f) + a;
    glm::vec3 mac = ((c - a) / 2.0f) + a;
    glm::vec3 mcb = ((b - c) / 2.0f) + c;
    q[0] = Quarter{ { a, mac, mab } };
    q[1] = Quarter{ { mac, c, mcb } };
    q[2] = Quarter{ { mab, mcb, b } };
    q[3] = Quarter{ { mab, mcb, mac } };
}

glm::vec3 third_sum(const Quarter& q) {
    glm::vec3 s = (q.p[0] + q.p[1] + q.p[2]);
    return glm::vec3(s.x / 3, s.y / 3, s.z / 3);
}

glm::vec3 patch_normal(const Mesh& m, int tri) {
    glm::vec3 s = (m.n(tri, 0) + m.n(tri, 1) + m.n(tri, 2));
    s = glm::vec3(s.x / 3, s.y / 3, s.z / 3);
    return glm::normalize(s);
}

float point_term(const glm::vec3& op, const glm::vec3& on, const glm::vec3& dp, const glm::vec3& dn, float surf) {
    float out = 0;
    float c1 = glm::dot(on, glm::normalize(dp - op));
    float c2 = glm::dot(dn, glm::normalize(op - dp));
    if (c1 > 0 && c2 > 0) {
        float len = glm::length(dp - op);
        out = ((c1 * c2) / (powf(len, 2) * 3.14159265358979323846f)) * surf;
    }
    return out;
}

}  // namespace

extern "C" {

float ref_surface(const float* a, const float* b, const float* c) {
    return area_of(glm::vec3(a[0], a[1], a[2]), glm::vec3(b[0], b[1], b[2]), glm::vec3(c[0], c[1], c[2]));
}

float ref_p2p_integrand(const float* vtx, const float* nrm, const int32_t* tv, const int32_t* tn, int i, int j) {
    Mesh m{ vtx, nrm, tv, tn };
    Quarter qi[4], qj[4];
    quarters(m, i, qi);
    quarters(m, j, qj);
    glm::vec3 ni = patch_normal(m, i), nj = patch_normal(m, j);
    glm::vec3 ci[4], cj[4];
    for (int k = 0; k < 4; k++) { ci[k] = third_sum(qi[k]); cj[k] = third_sum(qj[k]); }
    float acc = 0;
    for (int s = 0; s < 4; s++)
        for (int t = 0; t < 4; t++)
            acc = acc + point_term(ci[s], ni, cj[t], nj,
                                   area_of(qi[s].p[0], qi[s].p[1], qi[s].p[2]) * area_of(qj[t].p[0], qj[t].p[1], qj[t].p[2]));
    return acc / area_of(m.v(i, 0), m.v(i, 1), m.v(i, 2));
}

void ref_uv2xyz(const float* vtx, const int32_t* tv, int tri, float u, float v, float* out) {
    Mesh m{ vtx, nullptr, tv, nullptr };
    glm::vec3 a = m.v(tri, 0), b = m.v(tri, 1), c = m.v(tri, 2);
    glm::vec3 p = a + u * (b - a) + v * (c - a);
    out[0] = p.x; out[1] = p.y; out[2] = p.z;
}

/* One light pass through Eigen, written as Lightning.h writes it.
 * F dense N x N row-major -> triplets<double> -> SparseMatrix<float> (col-major),
 * R, B: N x S patch-major.  mode 0: spectral (per-patch MatrixXf * VectorXf,
 * Lightning.h:196-226); mode 1: RGB (cwiseProduct with rho = diag(M),
 * Lightning.h:342-349). */
void ref_light_pass(int N, int S, const float* F, const float* M, const int32_t* mat,
                    float* R, float* B, int mode) {
    typedef Eigen::SparseMatrix<float> SpMat;
    typedef Eigen::Triplet<double> Tripl;
    std::vector<Tripl> trip;
    for (int i = 0; i < N; i++)
        for (int j = 0; j < N; j++)
            if (F[(long)i * N + j] != 0.0f) trip.push_back(Tripl(i, j, (double)F[(long)i * N + j]));
    SpMat RadMat(N, N);
    RadMat.setFromTriplets(trip.begin(), trip.end());

    std::vector<Eigen::VectorXf> residual(S), light(S);
    for (int s = 0; s < S; s++) {
        residual[s] = Eigen::VectorXf(N); light[s] = Eigen::VectorXf(N);
        for (int i = 0; i < N; i++) { residual[s][i] = R[(long)i * S + s]; light[s][i] = B[(long)i * S + s]; }
    }
    if (mode == 1) {
        for (int s = 0; s < S; s++) {
            Eigen::VectorXf rho(N);
            for (int i = 0; i < N; i++) rho[i] = M[(long)mat[i] * S * S + s * S + s];
            residual[s] = (RadMat * residual[s]).cwiseProduct(rho);
            light[s] = light[s] + residual[s];
        }
    } else {
        std::vector<Eigen::VectorXf> bounced(S);
        for (int s = 0; s < S; s++) bounced[s] = (RadMat * residual[s]);
        for (int i = 0; i < N; i++) {
            Eigen::MatrixXf Mi(S, S);
            for (int a = 0; a < S; a++) for (int b = 0; b < S; b++) Mi(a, b) = M[(long)mat[i] * S * S + a * S + b];
            Eigen::VectorXf row(S);
            for (int s = 0; s < S; s++) row[s] = bounced[s][i];
            Eigen::VectorXf res = Mi * row;
            for (int s = 0; s < S; s++) residual[s][i] = res[s];
        }
        for (int s = 0; s < S; s++) light[s] = light[s] + residual[s];
    }
    for (int s = 0; s < S; s++)
        for (int i = 0; i < N; i++) { R[(long)i * S + s] = residual[s][i]; B[(long)i * S + s] = light[s][i]; }
}

/* Eigen's VectorXf::sum() as check_convergence uses it (Lightning.h:255-261) */
float ref_sum(int n, const float* x) {
    Eigen::Map<const Eigen::VectorXf> v(x, n);
    return v.sum();
}

/* daisy_color::cie1931WavelengthToXYZFit (vs/color.h:14-45), as Lightning.h:128-131 tabulates it */
void ref_xyz_fit(double wavelength, float* out3) {
    glm::vec3 v = daisy_color::cie1931WavelengthToXYZFit(wavelength);
    out3[0] = v[0]; out3[1] = v[1]; out3[2] = v[2];
}

/* SpectralLightning::update_color_cache for one patch (vs/Lightning.h:168-183); the reference leaves
 * `glm::vec3 xyz;` uninitialised (glm 0.9.8 default constructor), taken as zero here */
void ref_patch_color_spectral(int S, const float* xyz_per_bin, const float* b, float* rgb3) {
    glm::vec3 xyz(0.0f, 0.0f, 0.0f);
    for (int j = 0; j < S; j++)
        xyz += glm::vec3(xyz_per_bin[3 * j], xyz_per_bin[3 * j + 1], xyz_per_bin[3 * j + 2]) * b[j];
    glm::vec3 rgb = { 0.0, 0.0, 0.0 };
    daisy_color::XYZToRGB(xyz, rgb);
    float maxval = std::fmax(rgb[0], std::fmax(rgb[1], rgb[2]));
    if (maxval > 1) rgb = { rgb[0] / maxval, rgb[1] / maxval, rgb[2] / maxval };
    rgb3[0] = rgb[0]; rgb3[1] = rgb[1]; rgb3[2] = rgb[2];
}

/* the reference's own INIReader (main.cpp:63-79 reads config.ini through it) */
void* ref_ini_open(const char* path) { return new INIReader(path); }
void ref_ini_free(void* h) { delete (INIReader*)h; }
int ref_ini_error(void* h) { return ((INIReader*)h)->ParseError(); }
const char* ref_ini_get(void* h, const char* sec, const char* name, const char* def) {
    static std::string tmp;
    tmp = ((INIReader*)h)->Get(sec, name, def);
    return tmp.c_str();
}
long ref_ini_integer(void* h, const char* sec, const char* name, long def) { return ((INIReader*)h)->GetInteger(sec, name, def); }
double ref_ini_real(void* h, const char* sec, const char* name, double def) { return ((INIReader*)h)->GetReal(sec, name, def); }
int ref_ini_boolean(void* h, const char* sec, const char* name, int def) { return ((INIReader*)h)->GetBoolean(sec, name, def != 0) ? 1 : 0; }

/* Material::rgb_to_spectrum (vs/Material.cpp:35-45) through the reference's own rgb2spec_load / rgb2spec_fetch /
 * rgb2spec_eval_precise (vs/rgb2spec.cpp:11-134) on the table at `path`; returns 0 if the table does not load */
int ref_rgb2spec_spectrum(const char* path, const float* rgb_in, const float* wavelengths, int S, float* out) {
    RGB2Spec* model = rgb2spec_load(path);
    if (!model) return 0;
    float coeff[3];
    float rgb[3] = { rgb_in[0], rgb_in[1], rgb_in[2] };
    rgb2spec_fetch(model, rgb, coeff);
    for (int i = 0; i < S; i++) out[i] = rgb2spec_eval_precise(coeff, wavelengths[i]);
    rgb2spec_free(model);
    return 1;
}

/* The F-matrix disk cache (vs/Lightning.h:21-74), restated call for call on the reference's Eigen: SerializeMat dumps
 * rows, cols, nonZeros, outerSize, innerSize, then valuePtr (nnz floats), outerIndexPtr (outerSize ints -- not
 * outerSize + 1) and innerIndexPtr (nnz ints) of the compressed column-major matrix; DeserializeMat reads them back.
 * dense: N x N row-major, dense[i*N + j] = RadMat(i, j). */
int ref_fcache_write(const char* path, int N, const float* dense) {
    typedef Eigen::SparseMatrix<float> SpMat;
    std::vector<Eigen::Triplet<float>> trip;
    for (int i = 0; i < N; i++)
        for (int j = 0; j < N; j++)
            if (dense[(long)i * N + j] != 0.0f) trip.push_back(Eigen::Triplet<float>(i, j, dense[(long)i * N + j]));
    SpMat m(N, N);
    m.setFromTriplets(trip.begin(), trip.end());
    m.makeCompressed();
    FILE* f = std::fopen(path, "wb");
    if (!f) return 0;
    int rows = m.rows(), cols = m.cols(), nnzs = m.nonZeros(), outS = m.outerSize(), innS = m.innerSize();
    std::fwrite(&rows, sizeof(int), 1, f); std::fwrite(&cols, sizeof(int), 1, f); std::fwrite(&nnzs, sizeof(int), 1, f);
    std::fwrite(&outS, sizeof(int), 1, f); std::fwrite(&innS, sizeof(int), 1, f);
    std::fwrite(m.valuePtr(), sizeof(float), m.nonZeros(), f);
    std::fwrite(m.outerIndexPtr(), sizeof(int), m.outerSize(), f);
    std::fwrite(m.innerIndexPtr(), sizeof(int), m.nonZeros(), f);
    std::fclose(f);
    return 1;
}
int ref_fcache_read(const char* path, int N, float* dense) {
    typedef Eigen::SparseMatrix<float> SpMat;
    FILE* f = std::fopen(path, "rb");
    if (!f) return 0;
    int rows, cols, nnz, inSz, outSz;
    bool ok = std::fread(&rows, sizeof(int), 1, f) == 1 && std::fread(&cols, sizeof(int), 1, f) == 1 && std::fread(&nnz, sizeof(int), 1, f) == 1 &&
              std::fread(&inSz, sizeof(int), 1, f) == 1 && std::fread(&outSz, sizeof(int), 1, f) == 1;
    if (!ok || rows != N || cols != N) { std::fclose(f); return 0; }
    SpMat m;
    m.resize(rows, cols);
    m.makeCompressed();
    m.resizeNonZeros(nnz);
    ok = std::fread(m.valuePtr(), sizeof(float), nnz, f) == (size_t)nnz && std::fread(m.outerIndexPtr(), sizeof(int), outSz, f) == (size_t)outSz &&
         std::fread(m.innerIndexPtr(), sizeof(int), nnz, f) == (size_t)nnz;
    std::fclose(f);
    if (!ok) return 0;
    m.finalize();
    for (long k = 0; k < (long)N * N; k++) dense[k] = 0.0f;
    for (int c = 0; c < m.outerSize(); ++c)
        for (SpMat::InnerIterator it(m, c); it; ++it) dense[(long)it.row() * N + it.col()] = it.value();
    return 1;
}

/* tinyobj::LoadObj exactly as MeshS::loadFromFile calls it (vs/MeshS.cpp:25-31: triangulate = false), flattened the
 * way its loops read the result (vs/MeshS.cpp:67-126): corners in shape order, three per triangle */
struct RefObj {
    tinyobj::attrib_t attrib;
    std::vector<tinyobj::shape_t> shapes;
    std::vector<tinyobj::material_t> materials;
    std::string err;
    bool ok;
    std::vector<int> tri_v, tri_n, mat;
};
void* ref_obj_load(const char* obj, const char* mtl_dir) {
    RefObj* r = new RefObj();
    r->ok = tinyobj::LoadObj(&r->attrib, &r->shapes, &r->materials, &r->err, obj, mtl_dir, false);
    for (size_t i = 0; i < r->shapes.size(); i++) {
        tinyobj::shape_t& shape = r->shapes[i];
        for (size_t j = 0; j + 2 < shape.mesh.indices.size(); j += 3) {
            for (int k = 0; k < 3; k++) {
                r->tri_v.push_back(shape.mesh.indices[j + k].vertex_index);
                r->tri_n.push_back(shape.mesh.indices[j + k].normal_index);
            }
            r->mat.push_back(shape.mesh.material_ids[j / 3]);
        }
    }
    return r;
}
void ref_obj_free(void* h) { delete (RefObj*)h; }
void ref_obj_counts(void* h, int* ok, int* V, int* Nn, int* N, int* n_mat) {
    RefObj* r = (RefObj*)h;
    *ok = r->ok ? 1 : 0; *V = (int)(r->attrib.vertices.size() / 3); *Nn = (int)(r->attrib.normals.size() / 3);
    *N = (int)r->mat.size(); *n_mat = (int)r->materials.size();
}
void ref_obj_copy(void* h, float* vertices, float* normals, int* tri_v, int* tri_n, int* mat, float* Kd, float* Ke, float* Ks) {
    RefObj* r = (RefObj*)h;
    for (size_t i = 0; i < r->attrib.vertices.size(); i++) vertices[i] = r->attrib.vertices[i];
    for (size_t i = 0; i < r->attrib.normals.size(); i++) normals[i] = r->attrib.normals[i];
    for (size_t i = 0; i < r->tri_v.size(); i++) { tri_v[i] = r->tri_v[i]; tri_n[i] = r->tri_n[i]; }
    for (size_t i = 0; i < r->mat.size(); i++) mat[i] = r->mat[i];
    for (size_t m = 0; m < r->materials.size(); m++)
        for (int k = 0; k < 3; k++) {
            Kd[3 * m + k] = r->materials[m].diffuse[k]; Ke[3 * m + k] = r->materials[m].emission[k]; Ks[3 * m + k] = r->materials[m].specular[k];
        }
}
const char* ref_obj_material_name(void* h, int m) { return ((RefObj*)h)->materials[(size_t)m].name.c_str(); }

/* one corner of Drawer::interpolate (vs/Drawer.cpp:161-186): sum of the adjacent patches' colours / count */
void ref_vertex_color(int n_adj, const int32_t* adj, const float* rgb_patches, float* out3) {
    glm::vec3 a = { 0.f, 0.f, 0.f };
    for (int k = 0; k < n_adj; k++)
        a += glm::vec3(rgb_patches[3 * (long)adj[k]], rgb_patches[3 * (long)adj[k] + 1], rgb_patches[3 * (long)adj[k] + 2]);
    a = a / glm::vec3((size_t)n_adj);
    out3[0] = a[0]; out3[1] = a[1]; out3[2] = a[2];
}

}  // extern "C"
