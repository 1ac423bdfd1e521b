// mesh.h -- the scene loading surface of the reference, kept so .obj/.mtl scenes drop in
// unchanged: MeshS (visual studio/MeshS.h:7-26) and Material (visual studio/Material.h:8-42)
// with the same public field names and meaning.  glm::vec3 becomes daisy::vec3 (three floats,
// same layout), Eigen::MatrixXf becomes a row-major std::vector<float> of S*S.
#pragma once
#include <array>
#include <string>
#include <vector>

namespace daisy {

struct vec3 {
    float x, y, z;
    float& operator[](int i) { return (&x)[i]; }
    const float& operator[](int i) const { return (&x)[i]; }
};
struct ivec3 {
    int x, y, z;
    int& operator[](int i) { return (&x)[i]; }
    const int& operator[](int i) const { return (&x)[i]; }
};

namespace vertex {
struct TriangleIndex {      // visual studio/Vertex.h:11-14
    ivec3 vertex;
    ivec3 normal;
};
}  // namespace vertex

// RGB -> reflectance spectrum.  Reads the Jakob-Hanika coefficient table the reference loads
// from "color_tables/srgb.coeff" ("SPEC", u32 res, float scale[res], float data[3*res^3*3];
// visual studio/rgb2spec.cpp:11-48) when it exists.  The reference tree does not ship that file
// (.MISSING_LARGE_BLOBS), so without it a smooth three-lobe spectrum is used (parity unpinned).
class SpectralUpsampler {
public:
    explicit SpectralUpsampler(const std::string& coeff_path = "color_tables/srgb.coeff");
    bool has_table() const { return res_ > 0; }
    void spectrum(const vec3& rgb, const std::vector<float>& wavelengths, std::vector<float>& out) const;
private:
    void fetch(const float rgb[3], float coeff[3]) const;
    unsigned res_ = 0;
    std::vector<float> scale_, data_;
};

enum class MaterialKind { Plain, UVLight, Fluorescent };

class Material {            // visual studio/Material.h:8-25 (+ the two subclasses folded into `kind`)
public:
    vec3 rgbcolor;
    vec3 emission;
    std::vector<float> spectral_values;
    std::vector<float> spectral_emission;
    std::vector<float> M;   // numwavelengths x numwavelengths, row-major: how a wavelength maps to others
    MaterialKind kind = MaterialKind::Plain;
    vec3 blacklightcolor{ 0, 0, 0 };
    std::vector<float> spectral_from_blacklight;
    std::string name;

    Material(const vec3& rgbcolor, const vec3& emission, const std::vector<float>& wavelengths,
             const SpectralUpsampler& up);
    static Material UVLight(const std::vector<float>& wavelengths, const SpectralUpsampler& up);
    static Material Fluorescent(const vec3& rgbcolor, const vec3& emission, const vec3& blacklight,
                                const std::vector<float>& wavelengths, const SpectralUpsampler& up);
    int numwavelengths = 0;
};

class MeshS {               // visual studio/MeshS.h:7-26
public:
    int numtriangles = 0;
    std::vector<vec3> vertices;
    std::vector<vec3> normals;
    std::vector<vertex::TriangleIndex> triangleIndices;
    std::vector<std::vector<int>> trianglesPerVertex;
    std::vector<Material> materials;
    std::vector<int> materialIndexPerTriangle;
    std::string warnings;   // what the loader had to repair (the reference prints and continues)

    MeshS() = default;
    // same arguments as the reference: obj path, directory of the .mtl, wavelengths of the bins
    MeshS(const char* filepath, const char* mtlpath, const std::vector<float>& wavelengths);
    void loadFromFile(const char* filepath, const char* mtlpath, const std::vector<float>& wavelengths);
};

}  // namespace daisy
