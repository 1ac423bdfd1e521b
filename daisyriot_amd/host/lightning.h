// lightning.h -- the solver interface of the reference (visual studio/Lightning.h:7-15, 446-457)
// on top of libdaisyriot_hip.so: same four virtuals, same factory arguments, same convergence
// thresholds; the radiosity matrix, the passes and the per-patch bin transfer run on the MI355X.
#pragma once
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/daisyriot_hip.h"
#include "mesh.h"

namespace daisy {

struct UV { float u, v; };              // visual studio/Defines.h:3-6

// the K visibility samples every pair reuses (OptixPrimeFunctionality.cpp:54-63: u, v*(1-u)),
// from std::mt19937(seed) instead of srand(time()) so runs are reproducible
std::vector<UV> make_visibility_samples(int K = DR_RAYS_PER_PATCH, unsigned seed = 20191);

// xyz_per_wavelength entry of SpectralLightning (color.h:14-45)
vec3 cie1931_xyz_fit(double wavelength);

// F-matrix disk cache in the reference's format (Lightning.h:21-74): dense N x N row-major <-> file
enum FCacheStatus { FCACHE_OK = 0, FCACHE_ABSENT = 1, FCACHE_UNREADABLE = 2 };   // unreadable: another N, truncated, bad indices
FCacheStatus read_fcache_status(const char* path, int N, std::vector<float>& dense);
bool read_fcache(const char* path, int N, std::vector<float>& dense);
void write_fcache(const char* path, int N, const std::vector<float>& dense);

struct HipError : std::runtime_error { using std::runtime_error::runtime_error; };

// What the reference fixes at compile time (RAYS_PER_PATCH, the srand seed, the thresholds 200 / 1e-4, one GPU), as
// run-time options; the defaults are the reference's constants.  Filled from config.ini by daisyriot_cli:
// [acceleration] devices, rays_per_patch, seed, exchange, tree, walk; [lightning] tolerance, max_passes (bins selects the wavelengths).
struct LightningOptions {
    std::vector<int> devices = { 0 };   // HIP ordinals; rows of F are sharded over them (one process, dr_group)
    int rays_per_patch = DR_RAYS_PER_PATCH;     // visual studio/Defines.h:25
    unsigned seed = 20191;                      // of the K (u,v) samples (the reference: srand(time()))
    float tolerance = -1.0f;                    // < 0: the reference's threshold of the method (200 spectral, 1e-4 RGB / BW)
    int max_passes = 100000;                    // cap of converge_lightning (the reference has none)
    // dr_options by name ("" = the library's default): [acceleration] exchange = p2p | rccl | inpass (how a group moves the
    // residual after a pass), tree = lbvh | sah, walk = threaded | pairs | paths -- none of them changes a result bit
    std::string exchange, tree, walk;
};

class Lightning {
public:
    // method 0 = BW, 1 = RGB, 2 = Spectral ([lightning] method); cuda_enabled selects the assembly rule
    // ([acceleration] cuda_on); matfile: F-matrix cache path or nullptr.
    // max_passes bounds converge_lightning (the reference has no cap; BW never terminates in closed scenes).
    static Lightning* get_lightning(int method, MeshS& mesh, float& emissionval, std::vector<float> wavelengthsvec,
                                    bool cuda_enabled = false, const char* matfile = nullptr, int device = 0,
                                    int max_passes = 100000);
    static Lightning* get_lightning(int method, MeshS& mesh, float& emissionval, std::vector<float> wavelengthsvec,
                                    bool cuda_enabled, const char* matfile, const LightningOptions& options);
    virtual ~Lightning() {}
    virtual vec3 get_color_of_patch(int) = 0;
    virtual void converge_lightning() = 0;
    virtual void increment_lightpass() = 0;
    virtual void reset() = 0;
    // additions for a headless host
    virtual int passes() const = 0;
    virtual float residual_light() = 0;
    virtual const std::vector<float>& lightningvalues() const = 0;   // N x S patch-major
    virtual int bins() const = 0;
    virtual dr_info info() = 0;          // of rank 0 (assembly time = the slowest rank's)
    virtual int ranks() const = 0;       // GPUs the rows are sharded over
    // per-vertex display colour = mean over MeshS::trianglesPerVertex (Drawer.cpp:161-186), on the device
    virtual std::vector<vec3> vertex_colors() = 0;
};

}  // namespace daisy
