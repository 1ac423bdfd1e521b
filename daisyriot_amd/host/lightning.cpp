#include "lightning.h"
#include <initializer_list>
#include <utility>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <random>

namespace daisy {

std::vector<UV> make_visibility_samples(int K, unsigned seed) {
    std::mt19937 gen(seed);
    std::vector<UV> r((size_t)K);
    for (int i = 0; i < K; i++) {
        // the reference draws rand()%RAND_MAX / RAND_MAX with MSVC's 15-bit rand; same lattice here
        float u = (float)(gen() % 32767u) / 32767.0f;
        float v = (float)(gen() % 32767u) / 32767.0f;
        r[(size_t)i] = UV{ u, v * (1 - u) };
    }
    return r;
}

// CIE 1931 colour-matching fit of Wyman, Sloan, Shirley (JCGT 2013), the table SpectralLightning fills per
// wavelength (color.h:14-45, Lightning.h:128-131): evaluated in double, stored as float
vec3 cie1931_xyz_fit(double w) {
    auto g = [](double x, double mu, double s1, double s2) { double t = (x - mu) * (x < mu ? s1 : s2); return std::exp(-0.5 * t * t); };
    double x = 0.362 * g(w, 442.0, 0.0624, 0.0374) + 1.056 * g(w, 599.8, 0.0264, 0.0323) - 0.065 * g(w, 501.1, 0.0490, 0.0382);
    double y = 0.821 * g(w, 568.8, 0.0213, 0.0247) + 0.286 * g(w, 530.9, 0.0613, 0.0322);
    double z = 1.217 * g(w, 437.0, 0.0845, 0.0278) + 0.681 * g(w, 459.0, 0.0385, 0.0725);
    return vec3{ (float)x, (float)y, (float)z };
}

static bool read_fcache_body(std::ifstream& f, int N, std::vector<float>& dense);

// Reader/writer of the reference's F cache (Lightning.h:21-74): ints rows, cols, nnz, outerSize, innerSize,
// then float values[nnz], int outerIndex[outerSize] (not outerSize+1), int innerIndex[nnz] of the
// column-major compressed matrix RadMat(i,j) = F(i->j).
bool read_fcache(const char* path, int N, std::vector<float>& dense) { return read_fcache_status(path, N, dense) == FCACHE_OK; }

FCacheStatus read_fcache_status(const char* path, int N, std::vector<float>& dense) {
    std::ifstream f(path, std::ios::binary);
    if (!f.is_open()) return FCACHE_ABSENT;
    return read_fcache_body(f, N, dense) ? FCACHE_OK : FCACHE_UNREADABLE;
}

static bool read_fcache_body(std::ifstream& f, int N, std::vector<float>& dense) {
    int hdr[5];
    f.read((char*)hdr, sizeof hdr);
    if (!f || hdr[0] != N || hdr[1] != N || hdr[2] < 0 || hdr[3] != N) return false;
    const int nnz = hdr[2];
    std::vector<float> val((size_t)nnz);
    std::vector<int> outer((size_t)N + 1), inner((size_t)nnz);
    f.read((char*)val.data(), sizeof(float) * (size_t)nnz);
    f.read((char*)outer.data(), sizeof(int) * (size_t)N);
    f.read((char*)inner.data(), sizeof(int) * (size_t)nnz);
    if (!f) return false;
    outer[(size_t)N] = nnz;
    dense.assign((size_t)N * N, 0.0f);
    for (int col = 0; col < N; col++)
        for (int k = outer[(size_t)col]; k < outer[(size_t)col + 1]; k++) {
            if (k < 0 || k >= nnz || inner[(size_t)k] < 0 || inner[(size_t)k] >= N) return false;
            dense[(size_t)inner[(size_t)k] * N + col] = val[(size_t)k];
        }
    return true;
}
void write_fcache(const char* path, int N, const std::vector<float>& dense) {
    std::vector<float> val;
    std::vector<int> outer((size_t)N), inner;
    for (int col = 0; col < N; col++) {
        outer[(size_t)col] = (int)val.size();
        for (int row = 0; row < N; row++) {
            float v = dense[(size_t)row * N + col];
            if (v != 0.0f) { val.push_back(v); inner.push_back(row); }
        }
    }
    std::ofstream f(path, std::ios::binary);
    if (!f.is_open()) return;
    int hdr[5] = { N, N, (int)val.size(), N, N };
    f.write((const char*)hdr, sizeof hdr);
    f.write((const char*)val.data(), sizeof(float) * val.size());
    f.write((const char*)outer.data(), sizeof(int) * outer.size());
    f.write((const char*)inner.data(), sizeof(int) * inner.size());
}

namespace {

void chk(int rc, const char* what) {
    if (rc != DR_OK) throw HipError(std::string(what) + ": " + dr_last_error());
}

class LightningHIP : public Lightning {
public:
    LightningHIP(int method, MeshS& mesh, float emission_value, const std::vector<float>& wavelengths, bool cuda_rule,
                 const char* matfile, const LightningOptions& opt)
        : method_(method), max_passes_(opt.max_passes), mesh_(mesh) {
        N_ = mesh.numtriangles;
        if (N_ <= 0) throw HipError("mesh has no triangles");
        if (opt.devices.empty()) throw HipError("no device given");
        const int n_mat = (int)mesh.materials.size();
        std::vector<float> E, M;
        if (method == 2) {                       // SpectralLightning (Lightning.h:114-139, 263-292)
            S_ = (int)wavelengths.size();
            threshold_ = 200.0f; per_bin_ = false;            // Lightning.h:146
            E.assign((size_t)N_ * S_, 0.0f);
            for (int j = 0; j < N_; j++)
                for (int s = 0; s < S_; s++) {
                    float e = mesh.materials[(size_t)mesh.materialIndexPerTriangle[(size_t)j]].spectral_emission[(size_t)s];
                    if (e > 0.0f) E[(size_t)j * S_ + s] = e * emission_value;
                }
            for (const Material& m : mesh.materials) M.insert(M.end(), m.M.begin(), m.M.end());
            for (float w : wavelengths) xyz_.push_back(cie1931_xyz_fit(w));
        } else if (method == 1) {                // RGBLightning (Lightning.h:317-382)
            S_ = 3;
            threshold_ = 0.0001f; per_bin_ = true;            // Lightning.h:337
            E.assign((size_t)N_ * 3, 0.0f);
            for (int j = 0; j < N_; j++)
                for (int s = 0; s < 3; s++) {
                    float e = mesh.materials[(size_t)mesh.materialIndexPerTriangle[(size_t)j]].emission[s];
                    if (e > 0.0f) E[(size_t)j * 3 + s] = e * emission_value;
                }
            for (const Material& m : mesh.materials)
                for (int a = 0; a < 3; a++)
                    for (int b = 0; b < 3; b++) M.push_back(a == b && m.rgbcolor[a] > 0.0f ? m.rgbcolor[a] : 0.0f);
        } else {                                  // BWLightning (Lightning.h:386-443): no reflectance
            S_ = 1;
            threshold_ = 0.0001f; per_bin_ = false;           // Lightning.h:411
            E.assign((size_t)N_, 0.0f);
            for (int j = 0; j < N_; j++) {
                float e = mesh.materials[(size_t)mesh.materialIndexPerTriangle[(size_t)j]].emission[0];
                if (e > 0.0f) E[(size_t)j] = e * emission_value;
            }
            M.assign((size_t)n_mat, 1.0f);
        }
        if (opt.tolerance >= 0.0f) threshold_ = opt.tolerance;
        // one process, all the listed GPUs: rows of F sharded over them (main.cpp:55-154 is one process too)
        chk(dr_group_create(opt.devices.data(), (int)opt.devices.size(), &grp_), "dr_group_create");
        world_ = (int)opt.devices.size();
        if (!opt.exchange.empty() || !opt.tree.empty() || !opt.walk.empty()) {
            dr_options o;
            chk(dr_options_defaults(&o), "dr_options_defaults");
            auto pick = [](const std::string& v, std::initializer_list<std::pair<const char*, int>> names, const char* key) {
                for (const auto& n : names) if (v == n.first) return n.second;
                throw HipError(std::string("config.ini: unknown value '") + v + "' of [acceleration] " + key);
            };
            if (!opt.exchange.empty()) o.group_exchange = pick(opt.exchange, { { "p2p", DR_GROUP_EXCHANGE_P2P }, { "rccl", DR_GROUP_EXCHANGE_RCCL }, { "inpass", DR_GROUP_EXCHANGE_INPASS } }, "exchange");
            if (!opt.tree.empty()) o.tree = pick(opt.tree, { { "lbvh", DR_TREE_LBVH }, { "sah", DR_TREE_SAH } }, "tree");
            if (!opt.walk.empty()) o.walk = pick(opt.walk, { { "threaded", DR_WALK_THREADED }, { "pairs", DR_WALK_PAIRS }, { "paths", DR_WALK_PATHS } }, "walk");
            chk(dr_group_set_options(grp_, &o), "dr_group_set_options");
        }
        for (int r = 0; r < world_; r++) {
            dr_context* c = nullptr;
            chk(dr_group_context(grp_, r, &c), "dr_group_context");
            ctx_.push_back(c);
        }
        std::vector<int32_t> tv((size_t)3 * N_), tn((size_t)3 * N_);
        for (int t = 0; t < N_; t++)
            for (int k = 0; k < 3; k++) {
                tv[(size_t)3 * t + k] = mesh.triangleIndices[(size_t)t].vertex[k];
                tn[(size_t)3 * t + k] = mesh.triangleIndices[(size_t)t].normal[k];
            }
        chk(dr_group_set_mesh(grp_, &mesh.vertices[0].x, (int)mesh.vertices.size(), &mesh.normals[0].x,
                              (int)mesh.normals.size(), tv.data(), tn.data(), N_), "dr_group_set_mesh");
        for (int r = 0; r < world_; r++) {
            int row0 = 0, nrows = 0, rpr = 0;
            chk(dr_get_shard(ctx_[(size_t)r], &row0, &nrows, &rpr), "dr_get_shard");
            row0_.push_back(row0); nrows_.push_back(nrows);
        }
        // initMat / initMatFromFile (Lightning.h:75-96).  Unlike the reference, a cache file that exists but cannot be this
        // scene's (another N, truncated) is neither trusted nor overwritten, and a matrix too large for the cache's int
        // indices is reported instead of silently re-traced on every run.
        std::vector<float> dense;
        const FCacheStatus st = matfile ? read_fcache_status(matfile, N_, dense) : FCACHE_ABSENT;
        if (st == FCACHE_OK) {
            for (int r = 0; r < world_; r++)
                if (nrows_[(size_t)r] > 0)
                    chk(dr_formfactors_load_rows(ctx_[(size_t)r], row0_[(size_t)r], nrows_[(size_t)r], dense.data() + (size_t)row0_[(size_t)r] * N_),
                        "dr_formfactors_load_rows");
            std::printf("Deserialized matrix\n");
            std::printf("(the cache file does not record which rule wrote it: [acceleration] cuda_on = %s now)\n", cuda_rule ? "true" : "false");
        } else {
            if (st == FCACHE_UNREADABLE)
                std::fprintf(stderr, "warning: %s exists but is not a form-factor cache of this scene (%d patches): left untouched, "
                                     "form factors are assembled and NOT cached\n", matfile, N_);
            std::vector<UV> rands = make_visibility_samples(opt.rays_per_patch, opt.seed);
            chk(dr_group_assemble(grp_, &rands[0].u, (int)rands.size(), DR_ORIGIN_EPS,
                                  cuda_rule ? DR_RULE_INTEGRAND : DR_RULE_RECIPROCITY, 0), "dr_group_assemble");
            if (matfile && st == FCACHE_ABSENT) {
                if ((size_t)N_ * N_ <= ((size_t)1 << 28)) {
                    dense.resize((size_t)N_ * N_);
                    for (int r = 0; r < world_; r++)
                        if (nrows_[(size_t)r] > 0)
                            chk(dr_formfactors_read_rows(ctx_[(size_t)r], row0_[(size_t)r], nrows_[(size_t)r], dense.data() + (size_t)row0_[(size_t)r] * N_),
                                "dr_formfactors_read_rows");
                    write_fcache(matfile, N_, dense);
                    std::printf("Loaded & Serialized matrix\n");
                } else {
                    std::printf("form-factor matrix of %d patches not cached: the cache format indexes its entries with 32-bit ints "
                                "(visual studio/Lightning.h:21-50)\n", N_);
                }
            }
        }
        chk(dr_group_solver_init(grp_, S_, E.data(), M.data(), n_mat, mesh.materialIndexPerTriangle.data()), "dr_group_solver_init");
        // RadMat is sparse in the reference; here the all-zero blocks of the dense matrix are not read (same bits out)
        for (dr_context* c : ctx_) chk(dr_solver_skip_zero_blocks(c, 1), "dr_solver_skip_zero_blocks");
        B_.assign((size_t)N_ * S_, 0.0f);
        refresh();
        numpasses_ = 0;
        std::printf("Lightning has been initialized\n");
        converge_lightning();
    }
    ~LightningHIP() override { dr_group_destroy(grp_); }

    // rgb_color_cache of the reference (Lightning.h:168-183, 332-334, 406-408), filled on the device
    vec3 get_color_of_patch(int i) override { return rgb_[(size_t)i]; }
    std::vector<vec3> vertex_colors() override {
        const size_t V = mesh_.vertices.size();
        std::vector<int32_t> off(V + 1, 0), adj;
        for (size_t v = 0; v < V; v++) {
            const std::vector<int>& t = mesh_.trianglesPerVertex[v];
            adj.insert(adj.end(), t.begin(), t.end());
            off[v + 1] = (int32_t)adj.size();
        }
        if (adj.empty()) adj.push_back(0);
        std::vector<vec3> out(V);
        // one rank: the colours are still on the device; several: every rank coloured its own rows, hand all of them over
        chk(dr_display_vertex_colors(ctx_[0], world_ == 1 ? nullptr : &rgb_[0].x, off.data(), adj.data(), (int)V, &out[0].x),
            "dr_display_vertex_colors");
        return out;
    }
    void converge_lightning() override {
        int it = 0;
        chk(dr_group_solver_converge(grp_, threshold_, per_bin_ ? 1 : 0, max_passes_, &it), "dr_group_solver_converge");
        numpasses_ += it;
        refresh();
    }
    void increment_lightpass() override {
        chk(dr_group_solver_step(grp_, 1, nullptr), "dr_group_solver_step");
        numpasses_++;
        refresh();
    }
    void reset() override {
        chk(dr_group_solver_reset(grp_), "dr_group_solver_reset");
        numpasses_ = 0;
        refresh();
    }
    int passes() const override { return numpasses_; }
    float residual_light() override { float r = 0; chk(dr_group_solver_step(grp_, 0, &r), "dr_group_solver_step"); return r; }
    const std::vector<float>& lightningvalues() const override { return B_; }
    int bins() const override { return S_; }
    int ranks() const override { return world_; }
    dr_info info() override {
        dr_info i;
        chk(dr_get_info(ctx_[0], &i), "dr_get_info");
        for (int r = 1; r < world_; r++) {
            dr_info j;
            chk(dr_get_info(ctx_[(size_t)r], &j), "dr_get_info");
            if (j.last_assemble_ms > i.last_assemble_ms) i.last_assemble_ms = j.last_assemble_ms;
            i.pairs_traced += j.pairs_traced;
        }
        return i;
    }

private:
    void refresh() {
        chk(dr_group_solver_read(grp_, B_.data(), nullptr), "dr_group_solver_read");
        rgb_.resize((size_t)N_);
        const int mode = method_ == 2 ? DR_DISPLAY_SPECTRAL : (method_ == 1 ? DR_DISPLAY_RGB : DR_DISPLAY_BW);
        for (int r = 0; r < world_; r++)
            if (nrows_[(size_t)r] > 0)
                chk(dr_display_patch_colors(ctx_[(size_t)r], mode, method_ == 2 ? &xyz_[0].x : nullptr, &rgb_[(size_t)row0_[(size_t)r]].x),
                    "dr_display_patch_colors");
    }
    int method_, max_passes_, N_ = 0, S_ = 0, numpasses_ = 0, world_ = 1;
    float threshold_ = 0;
    bool per_bin_ = false;
    MeshS& mesh_;
    dr_group* grp_ = nullptr;
    std::vector<dr_context*> ctx_;        // owned by grp_
    std::vector<int> row0_, nrows_;
    std::vector<float> B_;
    std::vector<vec3> xyz_, rgb_;
};

}  // namespace

Lightning* Lightning::get_lightning(int method, MeshS& mesh, float& emissionval, std::vector<float> wavelengthsvec,
                                    bool cuda_enabled, const char* matfile, const LightningOptions& options) {
    if (method < 0 || method > 2)       // the reference falls off the end of the function here (Lightning.h:448-456)
        throw HipError("lightning method must be 0 (BW), 1 (RGB) or 2 (Spectral)");
    return new LightningHIP(method, mesh, emissionval, wavelengthsvec, cuda_enabled, matfile, options);
}

Lightning* Lightning::get_lightning(int method, MeshS& mesh, float& emissionval, std::vector<float> wavelengthsvec,
                                    bool cuda_enabled, const char* matfile, int device, int max_passes) {
    LightningOptions o;
    o.devices = { device };
    o.max_passes = max_passes;
    return get_lightning(method, mesh, emissionval, wavelengthsvec, cuda_enabled, matfile, o);
}

}  // namespace daisy
