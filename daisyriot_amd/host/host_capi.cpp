// host_capi.cpp -- a small C view of the C++ host (MeshS / Material / INIReader) so the loader
// surface can be exercised from pytest without a GPU.  Not part of the hot-path ABI.
#include <cstring>
#include <string>
#include <vector>

#include "ini_reader.h"
#include "lightning.h"
#include "mesh.h"

using namespace daisy;

extern "C" {

struct drh_mesh { MeshS mesh; int S; };

drh_mesh* drh_mesh_load(const char* obj, const char* mtl_dir, const float* wavelengths, int S) {
    std::vector<float> wl(wavelengths, wavelengths + S);
    drh_mesh* h = new drh_mesh{ MeshS(obj, mtl_dir, wl), S };
    return h;
}
void drh_mesh_free(drh_mesh* h) { delete h; }
void drh_mesh_counts(drh_mesh* h, int* V, int* Nn, int* N, int* n_mat) {
    *V = (int)h->mesh.vertices.size(); *Nn = (int)h->mesh.normals.size();
    *N = h->mesh.numtriangles; *n_mat = (int)h->mesh.materials.size();
}
const char* drh_mesh_warnings(drh_mesh* h) { return h->mesh.warnings.c_str(); }
void drh_mesh_copy(drh_mesh* h, float* vertices, float* normals, int* tri_v, int* tri_n, int* mat) {
    std::memcpy(vertices, h->mesh.vertices.data(), sizeof(vec3) * h->mesh.vertices.size());
    std::memcpy(normals, h->mesh.normals.data(), sizeof(vec3) * h->mesh.normals.size());
    for (int t = 0; t < h->mesh.numtriangles; t++)
        for (int k = 0; k < 3; k++) {
            tri_v[3 * t + k] = h->mesh.triangleIndices[(size_t)t].vertex[k];
            tri_n[3 * t + k] = h->mesh.triangleIndices[(size_t)t].normal[k];
        }
    std::memcpy(mat, h->mesh.materialIndexPerTriangle.data(), sizeof(int) * h->mesh.materialIndexPerTriangle.size());
}
// per material: kind (0 plain, 1 UV light, 2 fluorescent), rgbcolor[3], emission[3], spectral_values[S],
// spectral_emission[S], M[S*S]
void drh_mesh_materials(drh_mesh* h, int* kind, float* rgb, float* emission, float* spectral_values,
                        float* spectral_emission, float* M) {
    const int S = h->S;
    for (size_t m = 0; m < h->mesh.materials.size(); m++) {
        const Material& mt = h->mesh.materials[m];
        kind[m] = (int)mt.kind;
        for (int k = 0; k < 3; k++) { rgb[3 * m + k] = mt.rgbcolor[k]; emission[3 * m + k] = mt.emission[k]; }
        std::memcpy(spectral_values + m * S, mt.spectral_values.data(), sizeof(float) * S);
        std::memcpy(spectral_emission + m * S, mt.spectral_emission.data(), sizeof(float) * S);
        std::memcpy(M + m * S * S, mt.M.data(), sizeof(float) * S * S);
    }
}
int drh_vertex_fanout(drh_mesh* h, int vertex) { return (int)h->mesh.trianglesPerVertex[(size_t)vertex].size(); }

struct drh_ini { INIReader r; std::string tmp; };
drh_ini* drh_ini_open(const char* path) { return new drh_ini{ INIReader(path), "" }; }
drh_ini* drh_ini_parse(const char* text) { return new drh_ini{ INIReader::FromString(text), "" }; }
void drh_ini_free(drh_ini* h) { delete h; }
int drh_ini_error(drh_ini* h) { return h->r.ParseError(); }
const char* drh_ini_get(drh_ini* h, const char* sec, const char* name, const char* def) { h->tmp = h->r.Get(sec, name, def); return h->tmp.c_str(); }
long drh_ini_integer(drh_ini* h, const char* sec, const char* name, long def) { return h->r.GetInteger(sec, name, def); }
double drh_ini_real(drh_ini* h, const char* sec, const char* name, double def) { return h->r.GetReal(sec, name, def); }
int drh_ini_boolean(drh_ini* h, const char* sec, const char* name, int def) { return h->r.GetBoolean(sec, name, def != 0) ? 1 : 0; }

// Material::rgb_to_spectrum (Material.cpp:35-45) through the table at `coeff_path`; returns 1 if the table was
// found (0: the smooth stand-in spectrum was used)
int drh_upsample(const char* coeff_path, const float* rgb, const float* wavelengths, int S, float* out) {
    SpectralUpsampler up(coeff_path);
    std::vector<float> wl(wavelengths, wavelengths + S), sp;
    up.spectrum(vec3{ rgb[0], rgb[1], rgb[2] }, wl, sp);
    for (int s = 0; s < S; s++) out[s] = sp[(size_t)s];
    return up.has_table() ? 1 : 0;
}

int drh_fcache_write(const char* path, int N, const float* dense) {
    std::vector<float> d(dense, dense + (size_t)N * N);
    write_fcache(path, N, d);
    return 1;
}
int drh_fcache_read(const char* path, int N, float* dense) {
    std::vector<float> d;
    if (!read_fcache(path, N, d)) return 0;
    for (size_t k = 0; k < d.size(); k++) dense[k] = d[k];
    return 1;
}

void drh_xyz_fit(double wavelength, float* out3) { vec3 v = cie1931_xyz_fit(wavelength); out3[0] = v.x; out3[1] = v.y; out3[2] = v.z; }

void drh_visibility_samples(int K, unsigned seed, float* uv) {
    std::vector<UV> r = make_visibility_samples(K, seed);
    std::memcpy(uv, r.data(), sizeof(UV) * (size_t)K);
}

}  // extern "C"
