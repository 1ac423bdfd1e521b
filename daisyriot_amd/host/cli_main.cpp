// daisyriot_cli -- headless replacement of the reference's main() (main.cpp:55-154): reads the
// same config.ini keys, loads the same .obj/.mtl scene through MeshS, builds the Lightning the
// ini selects, converges it, and writes the per-patch result instead of opening a window.
//
//   daisyriot_cli [config.ini] [--passes n] [--out file.csv] [--ply file.ply] [--device d] [--no-matfile]
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>
#include <memory>

#include "ini_reader.h"
#include "lightning.h"
#include "mesh.h"

using namespace daisy;

int main(int argc, char** argv) {
    std::string ini = "config.ini", out, ply;
    int extra_passes = 0, device = 0;
    bool use_matfile = true;
    for (int a = 1; a < argc; a++) {
        if (!std::strcmp(argv[a], "--passes") && a + 1 < argc) extra_passes = std::atoi(argv[++a]);
        else if (!std::strcmp(argv[a], "--out") && a + 1 < argc) out = argv[++a];
        else if (!std::strcmp(argv[a], "--ply") && a + 1 < argc) ply = argv[++a];
        else if (!std::strcmp(argv[a], "--device") && a + 1 < argc) device = std::atoi(argv[++a]);
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
    // extra key (default = effectively the reference's "no cap"): upper bound on converge passes
    int max_passes = (int)reader.GetInteger("lightning", "max_passes", 100000);
    std::string scene = reader.Get("filepaths", "scene", "UNKNOWN");
    std::string mtl_dir = reader.Get("filepaths", "mtl_dir", "testscenes/");
    // F cache next to the scene: scene path minus its 4-character extension (main.cpp:80-82)
    std::string matfile = scene.size() > 4 ? scene.substr(0, scene.size() - 4) : scene;
    std::cout << "mat file path " << matfile << std::endl;

    std::vector<float> wavelengths = { 200.0f, 250.0f, 300.0f, 350.0f, 400.0f, 450.0f, 500.0f, 550.0f, 600.0f };   // main.cpp:94
    try {
        auto t0 = std::chrono::high_resolution_clock::now();
        MeshS mesh(scene.c_str(), mtl_dir.c_str(), wavelengths);
        if (!mesh.warnings.empty()) std::cerr << mesh.warnings;
        if (mesh.numtriangles == 0) { std::cerr << "no triangles loaded from " << scene << "\n"; return 1; }
        std::cout << "Number of triangles: " << mesh.numtriangles << std::endl;
        std::unique_ptr<Lightning> lightning(Lightning::get_lightning(method, mesh, emission_value, wavelengths, cuda_on,
                                                                      use_matfile ? matfile.c_str() : nullptr, device, max_passes));
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
