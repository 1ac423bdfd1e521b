// daisyriot_cli -- headless replacement of the reference's main() (main.cpp:55-154): reads the
// same config.ini keys, loads the same .obj/.mtl scene through MeshS, builds the Lightning the
// ini selects, converges it, and writes the per-patch result instead of opening a window.
//
//   daisyriot_cli [config.ini] [--passes n] [--out file.csv] [--ply file.ply] [--device d | --devices 0,1,..] [--no-matfile]
//
// Keys beyond the reference's (defaults = its compile-time constants): [acceleration] devices = 0,1,.. (GPUs the rows of
// F are sharded over, one process), rays_per_patch (RAYS_PER_PATCH, Defines.h:25), seed (of the visibility samples);
// [lightning] tolerance (convergence threshold; default 200 spectral / 1e-4 RGB, BW), max_passes, bins (number of
// wavelengths from 200 to 600 nm; default 9 = main.cpp:94's {200,250,..,600}).
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>
#include <memory>
#include <sstream>

#include "ini_reader.h"
#include "lightning.h"
#include "mesh.h"

using namespace daisy;

static std::vector<int> parse_devices(const std::string& s) {
    std::vector<int> d;
    std::stringstream ss(s);
    std::string tok;
    while (std::getline(ss, tok, ','))
        if (!tok.empty()) d.push_back(std::atoi(tok.c_str()));
    return d;
}

int main(int argc, char** argv) {
    std::string ini = "config.ini", out, ply, devices_arg;
    int extra_passes = 0;
    bool use_matfile = true;
    for (int a = 1; a < argc; a++) {
        if (!std::strcmp(argv[a], "--passes") && a + 1 < argc) extra_passes = std::atoi(argv[++a]);
        else if (!std::strcmp(argv[a], "--out") && a + 1 < argc) out = argv[++a];
        else if (!std::strcmp(argv[a], "--ply") && a + 1 < argc) ply = argv[++a];
        else if ((!std::strcmp(argv[a], "--device") || !std::strcmp(argv[a], "--devices")) && a + 1 < argc) devices_arg = argv[++a];
        else if (!std::strcmp(argv[a], "--no-matfile")) use_matfile = false;
        else ini = argv[a];
    }
    INIReader reader(ini);
    if (reader.ParseError() != 0) {
        std::cout << "Can't load '" << ini << "'\n";            // main.cpp:64-67
        return 1;
    }
    float emission_value = (float)reader.GetReal("lightning", "emission_value", -1);
    int method = (int)reader.GetInteger("lightning", "method", 0);
    bool cuda_on = reader.GetBoolean("acceleration", "cuda_on", false);
    LightningOptions opt;
    // extra keys (defaults = the reference's constants)
    opt.max_passes = (int)reader.GetInteger("lightning", "max_passes", 100000);     // the reference has no cap
    opt.tolerance = (float)reader.GetReal("lightning", "tolerance", -1.0);         // < 0: 200 (spectral) / 1e-4 (RGB, BW)
    opt.rays_per_patch = (int)reader.GetInteger("acceleration", "rays_per_patch", DR_RAYS_PER_PATCH);
    opt.seed = (unsigned)reader.GetInteger("acceleration", "seed", 20191);
    opt.devices = parse_devices(devices_arg.empty() ? reader.Get("acceleration", "devices", "0") : devices_arg);
    if (opt.devices.empty()) opt.devices.push_back(0);
    opt.exchange = reader.Get("acceleration", "exchange", "");
    opt.tree = reader.Get("acceleration", "tree", "");
    opt.walk = reader.Get("acceleration", "walk", "");
    const int bins = (int)reader.GetInteger("lightning", "bins", 9);
    std::string scene = reader.Get("filepaths", "scene", "UNKNOWN");
    std::string mtl_dir = reader.Get("filepaths", "mtl_dir", "testscenes/");
    // F cache next to the scene: scene path minus its 4-character extension (main.cpp:80-82)
    std::string matfile = scene.size() > 4 ? scene.substr(0, scene.size() - 4) : scene;
    std::cout << "mat file path " << matfile << std::endl;

    // main.cpp:94: {200, 250, .., 600}; `bins` other than 9 spreads that many wavelengths over the same range
    std::vector<float> wavelengths;
    if (bins < 1 || bins > DR_MAX_BINS) { std::cerr << "[lightning] bins must be 1.." << DR_MAX_BINS << "\n"; return 1; }
    for (int b = 0; b < bins; b++) wavelengths.push_back(bins == 1 ? 400.0f : 200.0f + 400.0f * (float)b / (float)(bins - 1));
    try {
        auto t0 = std::chrono::high_resolution_clock::now();
        MeshS mesh(scene.c_str(), mtl_dir.c_str(), wavelengths);
        if (!mesh.warnings.empty()) std::cerr << mesh.warnings;
        if (mesh.numtriangles == 0) { std::cerr << "no triangles loaded from " << scene << "\n"; return 1; }
        std::cout << "Number of triangles: " << mesh.numtriangles << std::endl;
        std::unique_ptr<Lightning> lightning(Lightning::get_lightning(method, mesh, emission_value, wavelengths, cuda_on,
                                                                      use_matfile ? matfile.c_str() : nullptr, opt));
        if (lightning->ranks() > 1) std::cout << "Rows of the form-factor matrix sharded over " << lightning->ranks() << " GPUs" << std::endl;
        for (int k = 0; k < extra_passes; k++) lightning->increment_lightpass();    // key 'L'
        auto t1 = std::chrono::high_resolution_clock::now();
        dr_info info = lightning->info();
        std::cout << "Number of light passes " << lightning->passes() << ". Amount of residual light in scene "
                  << lightning->residual_light() << std::endl;
        std::cout << "Calculation time of form factors + visibility: " << info.last_assemble_ms / 1e3 << " s (BVH "
                  << info.last_bvh_ms << " ms); total " << std::chrono::duration<double>(t1 - t0).count() << " s\n";
        if (!out.empty()) {
            std::ofstream f(out.c_str());
            const int S = lightning->bins();
            const std::vector<float>& B = lightning->lightningvalues();
            f << "patch,r,g,b";
            for (int s = 0; s < S; s++) f << ",B" << s;
            f << "\n";
            f.precision(9);
            for (int i = 0; i < mesh.numtriangles; i++) {
                vec3 c = lightning->get_color_of_patch(i);
                f << i << "," << c.x << "," << c.y << "," << c.z;
                for (int s = 0; s < S; s++) f << "," << B[(size_t)i * S + s];
                f << "\n";
            }
        }
        if (!ply.empty()) {
            // what the reference shows on screen, without a screen: every vertex gets the mean display colour of
            // the patches around it -- the corner values Drawer::interpolate blends (Drawer.cpp:161-186)
            std::ofstream f(ply.c_str());
            f << "ply\nformat ascii 1.0\nelement vertex " << mesh.vertices.size()
              << "\nproperty float x\nproperty float y\nproperty float z\nproperty uchar red\nproperty uchar green\nproperty uchar blue\n"
              << "element face " << mesh.numtriangles << "\nproperty list uchar int vertex_indices\nend_header\n";
            const std::vector<vec3> vc = lightning->vertex_colors();
            for (size_t v = 0; v < mesh.vertices.size(); v++) {
                auto to8 = [](float x) { x = x > 0 ? (x > 1 ? 1 : x) : 0; return (int)(x * 255.0f + 0.5f); };
                f << mesh.vertices[v].x << " " << mesh.vertices[v].y << " " << mesh.vertices[v].z << " "
                  << to8(vc[v].x) << " " << to8(vc[v].y) << " " << to8(vc[v].z) << "\n";
            }
            for (int t = 0; t < mesh.numtriangles; t++)
                f << "3 " << mesh.triangleIndices[(size_t)t].vertex.x << " " << mesh.triangleIndices[(size_t)t].vertex.y << " "
                  << mesh.triangleIndices[(size_t)t].vertex.z << "\n";
        }
    } catch (const std::exception& e) {
        std::cerr << "error: " << e.what() << std::endl;
        return 2;
    }
    return 0;
}
