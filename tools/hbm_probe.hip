// hbm_probe.hip -- what a bare streaming read reaches on this MI355X: the practical ceiling that the light
// pass (k_sweep, HBM-bound) is measured against next to the 8 TB/s spec peak.
//   hipcc -O3 --offload-arch=gfx950 -o hbm_probe tools/hbm_probe.hip && ./hbm_probe [GiB]
// Reads a buffer of the F matrix's size (16 GiB by default) once per launch with 16-byte loads, plain and
// non-temporal, several unroll depths and grid sizes; prints GB/s per variant (mean of 20 launches).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float v4f __attribute__((ext_vector_type(4)));

template <int UNROLL, bool NT>
__global__ __launch_bounds__(256) void k_read(const v4f* __restrict__ src, size_t n_vec, float* __restrict__ out) {
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    v4f acc = { 0, 0, 0, 0 };
    for (; i + (UNROLL - 1) * stride < n_vec; i += UNROLL * stride) {
        v4f v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; u++) v[u] = NT ? __builtin_nontemporal_load(src + i + u * stride) : src[i + u * stride];
#pragma unroll
        for (int u = 0; u < UNROLL; u++) acc += v[u];
    }
    for (; i < n_vec; i += stride) acc += src[i];
    float s = acc[0] + acc[1] + acc[2] + acc[3];
    if (s == 12345.678f) out[blockIdx.x] = s;       // keeps the loads alive; practically never true
}

// each block reads one contiguous segment (the way a block of the light pass walks along its rows)
template <int UNROLL, bool NT>
__global__ __launch_bounds__(256) void k_read_seg(const v4f* __restrict__ src, size_t n_vec, float* __restrict__ out) {
    const size_t per_block = n_vec / gridDim.x;
    const v4f* p = src + (size_t)blockIdx.x * per_block;
    v4f acc = { 0, 0, 0, 0 };
    size_t i = threadIdx.x;
    for (; i + (UNROLL - 1) * 256 < per_block; i += UNROLL * 256) {
        v4f v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; u++) v[u] = NT ? __builtin_nontemporal_load(p + i + u * 256) : p[i + u * 256];
#pragma unroll
        for (int u = 0; u < UNROLL; u++) acc += v[u];
    }
    float s = acc[0] + acc[1] + acc[2] + acc[3];
    if (s == 12345.678f) out[blockIdx.x] = s;
}

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int UNROLL, bool NT>
int run(const v4f* buf, size_t n_vec, float* out, int blocks, double bytes) {
    hipEvent_t a, b;
    CHK(hipEventCreate(&a));
    CHK(hipEventCreate(&b));
    for (int w = 0; w < 3; w++) hipLaunchKernelGGL((k_read<UNROLL, NT>), dim3(blocks), dim3(256), 0, 0, buf, n_vec, out);
    CHK(hipDeviceSynchronize());
    double ms_tot = 0;
    const int reps = 20;
    for (int r = 0; r < reps; r++) {
        CHK(hipEventRecord(a, 0));
        hipLaunchKernelGGL((k_read<UNROLL, NT>), dim3(blocks), dim3(256), 0, 0, buf, n_vec, out);
        CHK(hipEventRecord(b, 0));
        CHK(hipEventSynchronize(b));
        float ms = 0;
        CHK(hipEventElapsedTime(&ms, a, b));
        ms_tot += ms;
    }
    std::printf("unroll %d  %-5s blocks %6d : %8.3f ms  %7.1f GB/s\n", UNROLL, NT ? "nt" : "plain", blocks, ms_tot / reps,
                bytes / (ms_tot / reps * 1e-3) / 1e9);
    std::fflush(stdout);
    return 0;
}

// the light pass's own access pattern with nothing else: the buffer as an N x N fp32 matrix, a block of 4 waves takes 32 rows,
// every wave streams its 8 rows tile by tile (1 KiB per row and step, rows 4N bytes apart), next tile in flight
template <bool NT>
__global__ __launch_bounds__(256) void k_read_rows(const v4f* __restrict__ src, int N, float* __restrict__ out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t row0 = ((size_t)blockIdx.x * 4 + wave) * 8;
    const size_t ld = (size_t)N / 4;                     // v4f per row
    const int ntiles = N / 256;
    v4f acc = { 0, 0, 0, 0 }, cur[8], nxt[8];
#pragma unroll
    for (int r = 0; r < 8; r++) { const v4f* p = src + (row0 + r) * ld + lane; cur[r] = NT ? __builtin_nontemporal_load(p) : *p; }
    for (int t = 0; t < ntiles; t++) {
        if (t + 1 < ntiles) {
#pragma unroll
            for (int r = 0; r < 8; r++) { const v4f* p = src + (row0 + r) * ld + (size_t)(t + 1) * 64 + lane; nxt[r] = NT ? __builtin_nontemporal_load(p) : *p; }
        }
#pragma unroll
        for (int r = 0; r < 8; r++) acc += cur[r];
#pragma unroll
        for (int r = 0; r < 8; r++) cur[r] = nxt[r];
    }
    float s = acc[0] + acc[1] + acc[2] + acc[3];
    if (s == 12345.678f) out[blockIdx.x] = s;
}

// the same volume as 32-row x 256-column chunks stored one after the other (a tile-major matrix): block b streams the
// contiguous 32 x N x 4 bytes of its row block, a wave the 8 KiB of its 8 rows per step
template <bool NT>
__global__ __launch_bounds__(256) void k_read_tiled(const v4f* __restrict__ src, int N, float* __restrict__ out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ntiles = N / 256;
    const v4f* base = src + (size_t)blockIdx.x * ntiles * 2048 + (size_t)wave * 512 + lane;     // 2048 v4f per chunk, 512 per wave
    v4f acc = { 0, 0, 0, 0 }, cur[8], nxt[8];
#pragma unroll
    for (int r = 0; r < 8; r++) { const v4f* p = base + r * 64; cur[r] = NT ? __builtin_nontemporal_load(p) : *p; }
    for (int t = 0; t < ntiles; t++) {
        if (t + 1 < ntiles) {
#pragma unroll
            for (int r = 0; r < 8; r++) { const v4f* p = base + (size_t)(t + 1) * 2048 + r * 64; nxt[r] = NT ? __builtin_nontemporal_load(p) : *p; }
        }
#pragma unroll
        for (int r = 0; r < 8; r++) acc += cur[r];
#pragma unroll
        for (int r = 0; r < 8; r++) cur[r] = nxt[r];
    }
    float s = acc[0] + acc[1] + acc[2] + acc[3];
    if (s == 12345.678f) out[blockIdx.x] = s;
}

template <bool NT>
int run_rows(const v4f* buf, int N, float* out) {
    hipEvent_t a, b;
    CHK(hipEventCreate(&a));
    CHK(hipEventCreate(&b));
    const int blocks = N / 32;
    const double bytes = 4.0 * N * (double)N;
    // dynamic LDS caps the resident blocks per CU (160 KiB per CU): 0 = whatever fits, 70 KiB = 2 per CU (the pass's residency)
    for (int lds : { 0, 70 * 1024 }) {
        for (int tiled = 0; tiled < 2; tiled++) {
            auto launch = [&]() {
                if (tiled) hipLaunchKernelGGL((k_read_tiled<NT>), dim3(blocks), dim3(256), lds, 0, buf, N, out);
                else hipLaunchKernelGGL((k_read_rows<NT>), dim3(blocks), dim3(256), lds, 0, buf, N, out);
            };
            for (int w = 0; w < 3; w++) launch();
            CHK(hipDeviceSynchronize());
            double ms_tot = 0;
            const int reps = 20;
            for (int r = 0; r < reps; r++) {
                CHK(hipEventRecord(a, 0));
                launch();
                CHK(hipEventRecord(b, 0));
                CHK(hipEventSynchronize(b));
                float ms = 0;
                CHK(hipEventElapsedTime(&ms, a, b));
                ms_tot += ms;
            }
            std::printf("%s %-5s N %d blocks %6d, %s : %8.3f ms  %7.1f GB/s\n", tiled ? "row blocks stored contiguously" : "rows (the pass's pattern)     ",
                        NT ? "nt" : "plain", N, blocks, lds ? "2 blocks per CU" : "all resident   ", ms_tot / reps, bytes / (ms_tot / reps * 1e-3) / 1e9);
            std::fflush(stdout);
        }
    }
    return 0;
}

template <int UNROLL, bool NT>
int run_seg(const v4f* buf, size_t n_vec, float* out, int blocks, double bytes) {
    hipEvent_t a, b;
    CHK(hipEventCreate(&a));
    CHK(hipEventCreate(&b));
    for (int w = 0; w < 3; w++) hipLaunchKernelGGL((k_read_seg<UNROLL, NT>), dim3(blocks), dim3(256), 0, 0, buf, n_vec, out);
    CHK(hipDeviceSynchronize());
    double ms_tot = 0;
    const int reps = 20;
    for (int r = 0; r < reps; r++) {
        CHK(hipEventRecord(a, 0));
        hipLaunchKernelGGL((k_read_seg<UNROLL, NT>), dim3(blocks), dim3(256), 0, 0, buf, n_vec, out);
        CHK(hipEventRecord(b, 0));
        CHK(hipEventSynchronize(b));
        float ms = 0;
        CHK(hipEventElapsedTime(&ms, a, b));
        ms_tot += ms;
    }
    std::printf("segments unroll %d  %-5s blocks %6d : %8.3f ms  %7.1f GB/s\n", UNROLL, NT ? "nt" : "plain", blocks, ms_tot / reps,
                bytes / (ms_tot / reps * 1e-3) / 1e9);
    std::fflush(stdout);
    return 0;
}

int main(int argc, char** argv) {
    const double gib = argc > 1 ? std::atof(argv[1]) : 16.0;
    const size_t bytes = (size_t)(gib * 1024.0 * 1024.0 * 1024.0) / 4096 * 4096;
    v4f* buf = nullptr;
    float* out = nullptr;
    CHK(hipMalloc(&buf, bytes));
    CHK(hipMalloc(&out, sizeof(float) * 65536));
    CHK(hipMemset(buf, 0, bytes));
    CHK(hipDeviceSynchronize());
    const size_t n_vec = bytes / 16;
    std::printf("streaming read of %.2f GB\n", bytes / 1e9);
    for (int blocks : { 2048, 8192, 32768 }) {
        if (run<4, false>(buf, n_vec, out, blocks, (double)bytes)) return 1;
        if (run<4, true>(buf, n_vec, out, blocks, (double)bytes)) return 1;
        if (run<8, false>(buf, n_vec, out, blocks, (double)bytes)) return 1;
        if (run<8, true>(buf, n_vec, out, blocks, (double)bytes)) return 1;
    }
    for (int blocks : { 131072, 524288 }) {
        if (run<4, true>(buf, n_vec, out, blocks, (double)bytes)) return 1;
        if (run<2, true>(buf, n_vec, out, blocks, (double)bytes)) return 1;
    }
    for (int blocks : { 512, 1024, 2048, 8192, 65536 }) {
        if (run_seg<4, true>(buf, n_vec, out, blocks, (double)bytes)) return 1;
        if (run_seg<8, true>(buf, n_vec, out, blocks, (double)bytes)) return 1;
    }
    {
        int N = 256;
        while ((size_t)(2 * N) * (2 * N) * 4 <= bytes) N *= 2;      // the largest power-of-two matrix that fits the buffer
        if (run_rows<true>(buf, N, out)) return 1;
    }
    (void)hipFree(buf); (void)hipFree(out);
    return 0;
}
