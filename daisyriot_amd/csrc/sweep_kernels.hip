// sweep_kernels.hip -- the light-pass iteration R <- M_mat(i) * (F*R)[i], B += R for all
// S bins in one stream over the dense fp32 F shard (HBM-bound), for gfx950.
//
// Replaces vs/Lightning.h:196-226 (S sparse mat-vecs + N heap-allocating S x S mat-vecs),
// :342-349 (RGB) and :419-424 (BW): F is read ONCE per pass instead of once per bin, the
// residual tile is staged in LDS, and the per-patch S x S transfer is the kernel's epilogue.
//
// Layouts (device): F row-major, leading dimension ldF = world*rpr (zero padded);
// residual R as [world][S][rpr] (rank-major so an in-place all-gather fills it, bin-major
// inside a chunk so a column tile of one bin is contiguous); B as [S][rpr].
#include "dr_internal.h"

namespace dr {

constexpr int TC = 256;   // columns per tile: 64 lanes x float4

template <int S, int RR, int NW>
__global__ __launch_bounds__(NW * 64) void k_sweep(SweepParams P) {
    __shared__ __attribute__((aligned(16))) float sR[2][S][TC];
    __shared__ float sG[NW][RR][S];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int rbase = (blockIdx.x * NW + wave) * RR;       // first local row of this wave
    const int ntiles = (P.world * P.rpr) / TC;
    const int tiles_per_chunk = P.rpr / TC;

    // rows past the shard are clamped for loading and masked at the end
    const float* frow[RR];
#pragma unroll
    for (int r = 0; r < RR; r++) {
        int row = min(rbase + r, P.nrows - 1);
        frow[r] = P.F + (size_t)row * P.ldF + lane * 4;
    }

    float acc[RR][S];
#pragma unroll
    for (int r = 0; r < RR; r++)
#pragma unroll
        for (int s = 0; s < S; s++) acc[r][s] = 0.0f;

    // cooperative staging of one residual tile: S*TC floats = S*64 float4
    constexpr int R4_PER_THREAD = (S * 64 + NW * 64 - 1) / (NW * 64);
    float4 rreg[R4_PER_THREAD];
    auto load_rtile = [&](int t) {
        const int chunk = t / tiles_per_chunk;
        const int l0 = (t - chunk * tiles_per_chunk) * TC;
#pragma unroll
        for (int x = 0; x < R4_PER_THREAD; x++) {
            int q = tid + x * NW * 64;        // float4 index inside the tile: s = q/64, c4 = q%64
            if (q < S * 64) {
                int s = q >> 6, c4 = q & 63;
                rreg[x] = *reinterpret_cast<const float4*>(P.Rin + ((size_t)chunk * S + s) * P.rpr + l0 + c4 * 4);
            }
        }
    };
    auto store_rtile = [&](int buf) {
#pragma unroll
        for (int x = 0; x < R4_PER_THREAD; x++) {
            int q = tid + x * NW * 64;
            if (q < S * 64) {
                int s = q >> 6, c4 = q & 63;
                *reinterpret_cast<float4*>(&sR[buf][s][c4 * 4]) = rreg[x];
            }
        }
    };

    float4 fcur[RR], fnext[RR];
    load_rtile(0);
#pragma unroll
    for (int r = 0; r < RR; r++) fcur[r] = *reinterpret_cast<const float4*>(frow[r]);
    store_rtile(0);
    __syncthreads();

    for (int t = 0; t < ntiles; t++) {
        const bool more = (t + 1) < ntiles;
        if (more) {
#pragma unroll
            for (int r = 0; r < RR; r++) fnext[r] = *reinterpret_cast<const float4*>(frow[r] + (size_t)(t + 1) * TC);
            load_rtile(t + 1);
        }
        const int buf = t & 1;
#pragma unroll
        for (int s = 0; s < S; s++) {
            const float4 x = *reinterpret_cast<const float4*>(&sR[buf][s][lane * 4]);
#pragma unroll
            for (int r = 0; r < RR; r++) {
                acc[r][s] = fmaf(fcur[r].x, x.x, acc[r][s]);
                acc[r][s] = fmaf(fcur[r].y, x.y, acc[r][s]);
                acc[r][s] = fmaf(fcur[r].z, x.z, acc[r][s]);
                acc[r][s] = fmaf(fcur[r].w, x.w, acc[r][s]);
            }
        }
        if (more) store_rtile(buf ^ 1);
        __syncthreads();
#pragma unroll
        for (int r = 0; r < RR; r++) fcur[r] = fnext[r];
    }

    // wave reduction of the RR*S partial sums
#pragma unroll
    for (int r = 0; r < RR; r++)
#pragma unroll
        for (int s = 0; s < S; s++) {
            float v = acc[r][s];
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
            acc[r][s] = v;
        }
    if (lane == 0) {
#pragma unroll
        for (int r = 0; r < RR; r++)
#pragma unroll
            for (int s = 0; s < S; s++) sG[wave][r][s] = acc[r][s];
    }
    __syncthreads();

    // epilogue: per-patch S x S bin transfer, residual out, B += residual
    for (int e = lane; e < RR * S; e += 64) {
        const int so = e / RR, r = e % RR;          // consecutive lanes -> consecutive rows of one bin
        const int row = rbase + r;
        if (row < P.nrows) {
            const float* Mi = P.M + (size_t)P.mat[row] * S * S + so * S;
            float v = 0.0f;
#pragma unroll
            for (int s = 0; s < S; s++) v = fmaf(Mi[s], sG[wave][r][s], v);
            P.Rout[((size_t)P.rank * S + so) * P.rpr + row] = v;
            float* b = P.B + (size_t)so * P.rpr + row;
            *b = *b + v;
        }
    }
}

template <int S>
static hipError_t launch_sweep_s(hipStream_t st, const SweepParams& p) {
    constexpr int RR = (S <= 8) ? 8 : 4;
    constexpr int NW = 4;
    const int rows_per_block = RR * NW;
    dim3 grid((p.nrows + rows_per_block - 1) / rows_per_block);
    hipLaunchKernelGGL((k_sweep<S, RR, NW>), grid, dim3(NW * 64), 0, st, p);
    return hipGetLastError();
}

hipError_t launch_sweep(hipStream_t st, const SweepParams& p) {
    if (p.nrows <= 0) return hipSuccess;
    switch (p.S) {
#define DR_CASE(n) case n: return launch_sweep_s<n>(st, p);
        DR_CASE(1) DR_CASE(2) DR_CASE(3) DR_CASE(4) DR_CASE(5) DR_CASE(6) DR_CASE(7) DR_CASE(8)
        DR_CASE(9) DR_CASE(10) DR_CASE(11) DR_CASE(12) DR_CASE(13) DR_CASE(14) DR_CASE(15) DR_CASE(16)
#undef DR_CASE
    }
    return hipErrorInvalidValue;
}

// per-bin sums of the gathered residual (check_convergence, vs/Lightning.h:255-261), double
__global__ void k_colsums(const float* __restrict__ R, int world, int S, int rpr, double* __restrict__ sums) {
    __shared__ double sh[256];
    const int s = blockIdx.x;
    double a = 0.0;
    for (int c = 0; c < world; c++) {
        const float* x = R + ((size_t)c * S + s) * rpr;
        for (int i = threadIdx.x; i < rpr; i += 256) a += (double)x[i];
    }
    sh[threadIdx.x] = a;
    __syncthreads();
    for (int w = 128; w >= 1; w >>= 1) {
        if ((int)threadIdx.x < w) sh[threadIdx.x] += sh[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) sums[s] = sh[0];
}

hipError_t launch_colsums(hipStream_t st, const float* R, int world, int S, int rpr, double* sums) {
    hipLaunchKernelGGL(k_colsums, dim3(S), dim3(256), 0, st, R, world, S, rpr, sums);
    return hipGetLastError();
}

__global__ void k_scatter_rows(const float* __restrict__ src, int N, int S, int rpr, float* __restrict__ dst) {
    size_t x = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= (size_t)N * S) return;
    int s = (int)(x / N), i = (int)(x % N);
    int c = i / rpr, l = i % rpr;
    dst[((size_t)c * S + s) * rpr + l] = src[(size_t)i * S + s];
}
__global__ void k_gather_rows(const float* __restrict__ src, int N, int S, int rpr, float* __restrict__ dst) {
    size_t x = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= (size_t)N * S) return;
    int s = (int)(x / N), i = (int)(x % N);
    int c = i / rpr, l = i % rpr;
    dst[(size_t)i * S + s] = src[((size_t)c * S + s) * rpr + l];
}

hipError_t launch_scatter_rows(hipStream_t st, const float* src, int N, int S, int rpr, int world, float* dst) {
    (void)world;
    size_t n = (size_t)N * S;
    hipLaunchKernelGGL(k_scatter_rows, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, src, N, S, rpr, dst);
    return hipGetLastError();
}
hipError_t launch_gather_rows(hipStream_t st, const float* src, int N, int S, int rpr, int world, float* dst) {
    (void)world;
    size_t n = (size_t)N * S;
    hipLaunchKernelGGL(k_gather_rows, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, src, N, S, rpr, dst);
    return hipGetLastError();
}

}  // namespace dr
