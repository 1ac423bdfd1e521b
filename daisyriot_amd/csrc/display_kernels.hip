// display_kernels.hip -- radiance -> display colour, per patch and per vertex.
//
// What the reference computes on the host every time the light changes:
//   SpectralLightning::update_color_cache  (visual studio/Lightning.h:168-183): xyz = sum_s xyz_fit(lambda_s) * B_s[i]
//     (s ascending, fp32), rgb = XYZToRGB(xyz) (color.h:48-52), divided by max(r,g,b) when that exceeds 1;
//   RGBLightning / BWLightning::get_color_of_patch (Lightning.h:332-334, 406-408): B itself;
//   Drawer::interpolate (Drawer.cpp:161-186): every vertex takes the mean colour of the patches around it
//     (summed in trianglesPerVertex order, divided by the count).
// Here B never leaves the device for that: one thread per patch reads its S radiosity values (bin-major rows, so
// a wave reads 64 consecutive floats per bin) and writes 3 floats; one thread per vertex walks its CSR adjacency.
// Built with -ffp-contract=off so every product and sum rounds separately, as the host code the reference
// compiles (x86-64, no FMA) does: the results are bit-identical to a literal host evaluation.
#include <hip/hip_runtime.h>

#include "dr_internal.h"

namespace dr {

// mode 0: BW (b,b,b); 1: RGB (b0,b1,b2); 2: spectral through the XYZ fit
__global__ void k_patch_colors(const float* __restrict__ B, int nrows, int rpr, int S, int mode,
                               const float* __restrict__ xyz, float* __restrict__ rgb) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nrows) return;
    float r, g, b;
    if (mode == 0) {
        r = g = b = B[i];
    } else if (mode == 1) {
        r = B[i]; g = B[(size_t)rpr + i]; b = B[(size_t)2 * rpr + i];
    } else {
        float X = 0.0f, Y = 0.0f, Z = 0.0f;
        for (int s = 0; s < S; s++) {
            const float v = B[(size_t)s * rpr + i];
            X = X + xyz[3 * s + 0] * v;
            Y = Y + xyz[3 * s + 1] * v;
            Z = Z + xyz[3 * s + 2] * v;
        }
        r = 3.240479f * X - 1.537150f * Y - 0.498535f * Z;
        g = -0.969256f * X + 1.875991f * Y + 0.041556f * Z;
        b = 0.055648f * X - 0.204043f * Y + 1.057311f * Z;
        const float mx = fmaxf(r, fmaxf(g, b));
        if (mx > 1.0f) { r = r / mx; g = g / mx; b = b / mx; }
    }
    rgb[(size_t)3 * i + 0] = r;
    rgb[(size_t)3 * i + 1] = g;
    rgb[(size_t)3 * i + 2] = b;
}

__global__ void k_vertex_colors(const float* __restrict__ rgb, int V, const int* __restrict__ off,
                                const int* __restrict__ adj, float* __restrict__ out) {
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= V) return;
    float r = 0.0f, g = 0.0f, b = 0.0f;
    const int k0 = off[v], k1 = off[v + 1];
    for (int k = k0; k < k1; k++) {
        const int t = adj[k];
        r = r + rgb[(size_t)3 * t + 0];
        g = g + rgb[(size_t)3 * t + 1];
        b = b + rgb[(size_t)3 * t + 2];
    }
    const float n = (float)(k1 - k0);     // a vertex no triangle uses is never drawn by the reference: black here
    out[(size_t)3 * v + 0] = k1 > k0 ? r / n : 0.0f;
    out[(size_t)3 * v + 1] = k1 > k0 ? g / n : 0.0f;
    out[(size_t)3 * v + 2] = k1 > k0 ? b / n : 0.0f;
}

hipError_t launch_patch_colors(hipStream_t st, const float* B, int nrows, int rpr, int S, int mode, const float* xyz,
                               float* rgb) {
    if (nrows <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_patch_colors, dim3((nrows + 255) / 256), dim3(256), 0, st, B, nrows, rpr, S, mode, xyz, rgb);
    return hipGetLastError();
}

hipError_t launch_vertex_colors(hipStream_t st, const float* rgb, int V, const int* off, const int* adj, float* out) {
    if (V <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_vertex_colors, dim3((V + 255) / 256), dim3(256), 0, st, rgb, V, off, adj, out);
    return hipGetLastError();
}

}  // namespace dr
