// sweep_kernels.hip -- the light-pass iteration R <- M_mat(i) * (F*R)[i], B += R for all
// S bins in one stream over the dense fp32 F shard (HBM-bound), for gfx950.
//
// Replaces vs/Lightning.h:196-226 (S sparse mat-vecs + N heap-allocating S x S mat-vecs),
// :342-349 (RGB) and :419-424 (BW): F is read ONCE per pass instead of once per bin, the
// residual tile is staged in LDS, and the per-patch S x S transfer is the kernel's epilogue.
//
// Layouts (device): F row-major, leading dimension ldF = world*rpr (zero padded);
// residual R as [world][cstride], a chunk = [S][rpr] (rank-major so an in-place all-gather fills it, bin-major
// inside a chunk so a column tile of one bin is contiguous) followed by the chunk's per-bin sums (RTAIL floats =
// MAX_BINS doubles); B as [S][rpr].
//
// One launch per pass, whatever the shard: when the columns are cut into ranges (small row shards), the last range
// of a row block to arrive adds the partial sums in range order and runs the epilogue (no second kernel); the last
// row block to finish adds the row blocks' residual sums in a fixed order into the chunk's tail, which the same
// all-gather carries to every rank; and with a convergence rule in SweepParams each pass first looks at the tails
// of the residual it was given and does nothing if that has converged -- dr_solver_converge queues passes without
// a host round trip per pass (check_convergence, vs/Lightning.h:145-151, 255-261, 336-340).
#include "dr_internal.h"

namespace dr {

typedef float v4f __attribute__((ext_vector_type(4)));

// Has the gathered residual Rin converged under the pass's rule?  Every block of every rank reads the same
// world x S doubles and adds them in the same order: the whole job takes the same decision.
__device__ __forceinline__ bool residual_converged(const SweepParams& P) {
    double tot = 0.0;
    bool any = false;
    for (int s = 0; s < P.S; s++) {
        double t = 0.0;
        for (int c = 0; c < P.world; c++)
            t += reinterpret_cast<const double*>(P.Rin + (size_t)c * P.cstride + (size_t)P.S * P.rpr)[s];
        any = any || (t > (double)P.conv_thr);
        tot += t;
    }
    return P.conv_mode == 2 ? !any : !(tot > (double)P.conv_thr);
}

// first tile of column range y of n: equal ranges (taper 0), or shares proportional to taper + n - y (blocks are dispatched in
// the order of y: the last ones to start are the shortest, which evens out when the slots of the chip finish)
__device__ __forceinline__ int range_start(int y, int n, int ntiles, int taper) {
    if (taper == 0) return (int)(((long long)y * ntiles) / n);
    if (taper < 0) taper = 0;          // the steepest: shares n, n-1, .., 1
    // sum_{k<y} (taper + n - k) = y*(taper + n) - y*(y-1)/2
    const long long tot = (long long)n * (taper + n) - (long long)n * (n - 1) / 2;
    const long long acc = (long long)y * (taper + n) - (long long)y * (y - 1) / 2;
    return (int)((acc * ntiles) / tot);
}

// Hand-off between the blocks of one launch without fences (MI355X_MICROARCH.md, "Workgroup dispatch ... inter-workgroup
// visibility", the table of hand-offs with sc1 loads in place of the acquire, first row): every byte that another block
// will read is stored write-through (relaxed agent-scope store = global_store ... sc1) and read with sc1 loads
// (relaxed agent-scope load: served past the CU's L1); each storing wave waits for its stores (vmcnt(0)), then ONE lane
// of the block adds to the counter; the block whose add returns the last ticket reads after a workgroup barrier.  An
// agent-scope release/acquire pair per block (buffer_wbl2 + buffer_inv) cost 4 % of the whole pass at 2048 blocks.
__device__ __forceinline__ void st_agent(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_agent(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_system(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ void st_system(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ float ld_agent(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double ld_agent(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// one arrival at a counter that `expected` blocks arrive at; true for the block that arrives last (it also rearms the
// counter for the next pass)
// HARDWARE ASSUMPTION of the unfenced form: under the HIP / LLVM memory model relaxed stores + a relaxed
// ticket do not order the data before the ticket; on gfx950 they do, because sc1 stores write through to the device's coherence
// point (L2 / memory side), s_waitcnt vmcnt(0) returns only once they are acknowledged there, and sc1 loads are served from
// there.  `fenced` (dr_options::sweep_fenced) is the memory-model form of the same hand-off -- an agent-scope release fence in
// every thread before the ticket, an acquire fence in the last block after it -- kept for cross-checks (tests compare the two).
__device__ __forceinline__ bool arrive_last(unsigned* counter, unsigned expected, int* sFlag, int fenced) {
    if (fenced) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this wave's write-through stores have left
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned t = atomicAdd(counter, 1u);
        const bool last = (t == expected - 1u);
        if (last) atomicExch(counter, 0u);      // through the same path as the adds
        *sFlag = last ? 1 : 0;
    }
    __syncthreads();
    const bool last = *sFlag != 0;
    if (fenced && last) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    return last;
}

// In-pass exchange (SweepParams::peers): called by the LAST block of a pass, all threads, after every block has fenced its
// stores into the peers' buffers at system scope and arrived: the pass number into slot `rank` of every device's sequence array.
__device__ __forceinline__ void publish_pass(const SweepParams& P) {
    __threadfence_system();
    if ((int)threadIdx.x < P.n_peers && P.peer_seq[threadIdx.x])
        __hip_atomic_store(&P.peer_seq[threadIdx.x][P.rank], P.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// The gate in front of the next pass: ONE thread per rank of ONE block (it can never crowd out the kernels it waits for, which
// may share the device in a rehearsal) polls this device's sequence array until every rank has published pass `want`.  A wait
// that does not end within ~5 s (a peer died) sets err[0] and lets the stream go on: the host reports it (dr_group_solver_*).
__global__ void k_wait_peers(const unsigned* __restrict__ seq_local, int world, unsigned want, int* __restrict__ err) {
    const int r = threadIdx.x;
    if (r >= world) return;
    const unsigned long long t0 = wall_clock64();            // 100 MHz
    for (;;) {
        const unsigned have = __hip_atomic_load(&seq_local[r], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
        if ((int)(have - want) >= 0) return;
        if (wall_clock64() - t0 > 500000000ull) { err[0] = 1; return; }
        __builtin_amdgcn_s_sleep(8);
    }
}

hipError_t launch_wait_peers(hipStream_t st, const unsigned* seq_local, int world, unsigned want, int* err) {
    hipLaunchKernelGGL(k_wait_peers, dim3(1), dim3(64), 0, st, seq_local, world, want, err);
    return hipGetLastError();
}

// What follows the stream over F in both pass kernels.  sGf[ROWSB][SPAD] (LDS) holds the block's rows' sums F*R over
// its column range.  SPLIT: those are partial -- written out, and the last range of the row block to arrive adds all
// ranges up in range order (bit-stable).  Then the epilogue: R' = M_mat * G (per-patch S x S bin transfer,
// vs/Lightning.h:205-218), residual out, B += R'; the row block's per-bin sums of R' (rows in order, double); and the
// last row block adds those up (16 interleaved partial sums per bin, then these in order) into the chunk's tail.
template <int ROWSB, int SPAD, int NT, bool SPLIT>
__device__ __forceinline__ void sweep_tail(const SweepParams& P, float* sGf, float* sV /* [S][ROWSB] */, double* sD /* [256] */,
                                           int* sFlag) {
    static_assert(NT >= 256, "the final reduction uses 256 threads");
    const int tid = threadIdx.x;
    const int S = P.S;
    const int row0b = blockIdx.x * ROWSB;
    if (SPLIT) {
        for (int e = tid; e < ROWSB * S; e += NT) {
            const int rl = e / S, s2 = e % S;
            const int row = row0b + rl;
            if (row < P.nrows) st_agent(&P.Gpart[((size_t)blockIdx.y * P.nrows + row) * S + s2], sGf[rl * SPAD + s2]);
        }
        if (!arrive_last(&P.tickets[1 + blockIdx.x], gridDim.y, sFlag, P.tune.fenced)) return;
        for (int e = tid; e < ROWSB * S; e += NT) {
            const int rl = e / S, s2 = e % S;
            const int row = row0b + rl;
            float g = 0.0f;
            if (row < P.nrows)
                for (int k = 0; k < (int)gridDim.y; k++) g += ld_agent(&P.Gpart[((size_t)k * P.nrows + row) * S + s2]);
            sGf[rl * SPAD + s2] = g;
        }
        __syncthreads();
    }
    for (int e = tid; e < ROWSB * S; e += NT) {
        const int so = e / ROWSB, rl = e % ROWSB;          // consecutive threads -> consecutive rows of one bin
        const int row = row0b + rl;
        float v = 0.0f;
        if (row < P.nrows) {
            const float* Mi = P.M + (size_t)P.mat[row] * S * S + so * S;
            for (int s2 = 0; s2 < S; s2++) v = fmaf(Mi[s2], sGf[rl * SPAD + s2], v);
            const size_t at = (size_t)P.rank * P.cstride + (size_t)so * P.rpr + row;
            P.Rout[at] = v;
            for (int q = 0; q < P.n_peers; q++)                  // straight into the other devices' buffers (in-pass exchange),
                if (P.peers[q]) st_system(&P.peers[q][at], v);   // written through at system scope
            float* b = P.B + (size_t)so * P.rpr + row;
            *b = *b + v;
        }
        sV[so * ROWSB + rl] = v;
    }
    if (P.n_peers > 0 && !P.want_sums) {
        // no sums asked for, but the pass still has to say when ALL of it has landed everywhere: the same hand-off, without sums
        // (the peer stores are write-through at system scope and arrive_last waits for this wave's stores: no fence per block)
        if (P.tune.fenced) __threadfence_system();
        if (!arrive_last(&P.tickets[0], gridDim.x, sFlag, P.tune.fenced)) return;
        publish_pass(P);
        return;
    }
    // the per-bin sums of the new residual are only formed for passes that ask for them (dr_solver_converge): the
    // hand-off below holds every block for a few microseconds, about 1 % of a 64k-row pass
    if (!P.want_sums) return;
    __syncthreads();
    if (tid < S) {
        double a = 0.0;
        for (int rl = 0; rl < ROWSB; rl++) a += (double)sV[tid * ROWSB + rl];
        st_agent(&P.blk_sums[(size_t)blockIdx.x * S + tid], a);
    }
    if (P.n_peers > 0 && P.tune.fenced) __threadfence_system();       // (unfenced: write-through stores + arrive_last's wait)
    if (!arrive_last(&P.tickets[0], gridDim.x, sFlag, P.tune.fenced)) return;
    if (tid < 256) {
        const int s2 = tid >> 4, j = tid & 15;
        double a = 0.0;
        if (s2 < S)
            for (int x = j; x < (int)gridDim.x; x += 16) a += ld_agent(&P.blk_sums[(size_t)x * S + s2]);
        sD[tid] = a;
    }
    __syncthreads();
    if (tid < S) {
        double a = 0.0;
        for (int j = 0; j < 16; j++) a += sD[tid * 16 + j];
        const size_t at = (size_t)P.rank * P.cstride + (size_t)S * P.rpr;
        reinterpret_cast<double*>(P.Rout + at)[tid] = a;
        for (int q = 0; q < P.n_peers; q++)
            if (P.peers[q]) st_system(reinterpret_cast<double*>(P.peers[q] + at) + tid, a);
    }
    if (tid == 0) P.ctl[0] = P.ctl[0] + 1;
    if (P.n_peers > 0) { __syncthreads(); publish_pass(P); }
}

// S bins, RR rows per wave, NW waves per block, CPL float4 column groups per lane per row
// and tile (tile = 256*CPL columns), NT = non-temporal F loads (F is streamed once; keep
// the residual, which every block re-reads, resident in L2 instead).
// SPLIT: the columns are cut into gridDim.y ranges (small row shards would otherwise leave most
// CUs with one block or none); each block then writes its partial sums and the last range of the row block
// to arrive adds them in range order and applies the bin transfer (sweep_tail).
// LIST: the instantiation that can run over a list of non-zero tiles (SweepParams::tile_mask); the dense one does
// not carry the list's 8 KiB of LDS.
template <int S, int RR, int NW, int CPL, bool NT, int OCC, bool SPLIT, bool LIST>
__global__ __launch_bounds__(NW * 64, OCC) void k_sweep(SweepParams P) {
    constexpr int TC = 256 * CPL;
    __shared__ __attribute__((aligned(16))) float sR[2][S][TC];
    __shared__ float sG[NW][RR][S];
    __shared__ float sV[S][NW * RR];
    __shared__ double sD[256];
    __shared__ int sFlag;

    // converged: this pass and every pass queued behind it does nothing (latched in ctl[1]: the host keeps flipping the two
    // residual buffers per queued pass, and the older of the two has NOT converged)
    if (P.conv_mode != 0 && (P.ctl[1] != 0 || residual_converged(P))) {
        if (blockIdx.x == 0 && blockIdx.y == 0) {
            if (threadIdx.x == 0) P.ctl[1] = 1;
            if (P.n_peers > 0) publish_pass(P);        // (the gates of the passes queued behind this one still wait for its number)
        }
        return;
    }

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rbase = (blockIdx.x * NW + wave) * RR;       // first local row of this wave (uniform)
    const int ntiles_all = (P.world * P.rpr) / TC;
    // tiles of this block: the column ranges need not be equal, they only have to cover every tile once
    const int tile0 = SPLIT ? range_start(blockIdx.y, gridDim.y, ntiles_all, P.taper) : 0;
    const int ntiles = SPLIT ? range_start(blockIdx.y + 1, gridDim.y, ntiles_all, P.taper) - tile0 : ntiles_all;
    const int tiles_per_chunk = P.rpr / TC;

    // wave-uniform row bases (SGPRs); rows past the shard are clamped for loading, masked at the end
    const float* frow[RR];
#pragma unroll
    for (int r = 0; r < RR; r++) frow[r] = P.F + (size_t)min(rbase + r, P.nrows - 1) * P.ldF;

    float acc[RR][S];
#pragma unroll
    for (int r = 0; r < RR; r++)
#pragma unroll
        for (int s = 0; s < S; s++) acc[r][s] = 0.0f;

    // cooperative staging of one residual tile: S*TC floats = S*64*CPL float4
    constexpr int NV4 = S * 64 * CPL;
    constexpr int R4_PER_THREAD = (NV4 + NW * 64 - 1) / (NW * 64);
    v4f rreg[R4_PER_THREAD];
    auto load_rtile = [&](int tl) {
        const int t = tile0 + tl;
        const int chunk = t / tiles_per_chunk;
        const int l0 = (t - chunk * tiles_per_chunk) * TC;
        const float* base = P.Rin + (size_t)chunk * P.cstride + l0;
#pragma unroll
        for (int x = 0; x < R4_PER_THREAD; x++) {
            const int q = tid + x * NW * 64;        // float4 index inside the tile
            if (NV4 % (NW * 64) == 0 || q < NV4) {
                const int s = q / (64 * CPL), c4 = q % (64 * CPL);
                rreg[x] = *reinterpret_cast<const v4f*>(base + (size_t)s * P.rpr + c4 * 4);
            }
        }
    };
    auto store_rtile = [&](int buf) {
#pragma unroll
        for (int x = 0; x < R4_PER_THREAD; x++) {
            const int q = tid + x * NW * 64;
            if (NV4 % (NW * 64) == 0 || q < NV4) {
                const int s = q / (64 * CPL), c4 = q % (64 * CPL);
                *reinterpret_cast<v4f*>(&sR[buf][s][c4 * 4]) = rreg[x];
            }
        }
    };
    // wave-uniform row base (SGPR pair) + one 32-bit per-lane byte offset shared by all rows: the
    // saddr form of global_load, no 64-bit per-row addresses in VGPRs
    auto load_f = [&](int tl, v4f (&dst)[RR][CPL]) {
        const unsigned voff = (unsigned)(tile0 + tl) * (unsigned)(TC * 4) + (unsigned)lane * 16u;
#pragma unroll
        for (int r = 0; r < RR; r++)
#pragma unroll
            for (int c = 0; c < CPL; c++) {
                const v4f* p = reinterpret_cast<const v4f*>(reinterpret_cast<const char*>(frow[r]) + (voff + (unsigned)(c * 1024)));
                dst[r][c] = NT ? __builtin_nontemporal_load(p) : *p;
            }
    };

    // Optional list of the tiles of this block's range that hold a non-zero in its 32 rows (SweepParams::tile_mask):
    // built in order by one wave; the loop below then runs over the list only -- no loads, no staging, no barrier
    // for a tile that is zero in all of the block's rows.
    constexpr int MAXT = LIST ? 4096 : 1;
    __shared__ unsigned short sTiles[MAXT];
    __shared__ int sNT, sPos;
    const bool use_list = LIST && (RR * NW == 32) && (P.tile_mask != nullptr) && (ntiles <= MAXT);
    // start tile of this block (launch_sweep picks the skew); with a list: the first listed tile at or after it, so
    // that the non-zero tiles are visited in the same order as without the list (bit-identical sums)
    const int start_tile = (int)(((unsigned)blockIdx.x * (unsigned)P.skew) % (unsigned)ntiles);
    int nlist = ntiles;
    if (use_list) {
        if (wave == 0) {
            const unsigned* mrow = P.tile_mask + (size_t)blockIdx.x * P.mask_words;
            int cnt = 0, below = 0;
            for (int base = 0; base < ntiles; base += 64) {
                const int tl = base + lane;
                bool nzb = false;
                if (tl < ntiles) {
#pragma unroll
                    for (int c = 0; c < CPL; c++) {
                        const int bit = (tile0 + tl) * CPL + c;
                        nzb = nzb || ((mrow[bit >> 5] >> (bit & 31)) & 1u);
                    }
                }
                const unsigned long long m = __ballot(nzb);
                if (nzb) sTiles[cnt + __popcll(m & ((1ull << lane) - 1ull))] = (unsigned short)tl;
                cnt += __popcll(m);
                below += __popcll(__ballot(nzb && tl < start_tile));
            }
            if (lane == 0) { sNT = cnt; sPos = (below == cnt) ? 0 : below; }
        }
        __syncthreads();
        nlist = sNT;
    }
    auto tile_at = [&](int pos) { return use_list ? (int)sTiles[pos] : pos; };

    int pos = use_list ? sPos : start_tile;
    v4f fcur[RR][CPL], fnext[RR][CPL];
    if (nlist > 0) {
        load_rtile(tile_at(pos));
        load_f(tile_at(pos), fcur);
        store_rtile(0);
    }
    __syncthreads();

    for (int t = 0; t < nlist; t++) {
        const bool more = (t + 1) < nlist;
        pos = (pos + 1 == nlist) ? 0 : pos + 1;
        if (more) {
            const int tt = tile_at(pos);
            load_f(tt, fnext);
            load_rtile(tt);
        }
        const int buf = t & 1;
#pragma unroll
        for (int c = 0; c < CPL; c++)
#pragma unroll
            for (int s = 0; s < S; s++) {
                const v4f x = *reinterpret_cast<const v4f*>(&sR[buf][s][c * 256 + lane * 4]);
#pragma unroll
                for (int r = 0; r < RR; r++) {
                    acc[r][s] = fmaf(fcur[r][c].x, x.x, acc[r][s]);
                    acc[r][s] = fmaf(fcur[r][c].y, x.y, acc[r][s]);
                    acc[r][s] = fmaf(fcur[r][c].z, x.z, acc[r][s]);
                    acc[r][s] = fmaf(fcur[r][c].w, x.w, acc[r][s]);
                }
            }
        if (more) store_rtile(buf ^ 1);
        __syncthreads();
#pragma unroll
        for (int r = 0; r < RR; r++)
#pragma unroll
            for (int c = 0; c < CPL; c++) fcur[r][c] = fnext[r][c];
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

    sweep_tail<NW * RR, S, NW * 64, SPLIT>(P, &sG[0][0][0], &sV[0][0], sD, &sFlag);
}

// ---------------------------------------------------------------------------------------------
// MFMA variant for 9..16 bins.  With RR*S accumulators per lane the VALU kernel above runs out of
// registers beyond 8 bins (S = 9: 71 %, S = 16: 56 % of the HBM peak).  Here a wave owns 16 rows and
// one v_mfma_f32_16x16x4_f32 accumulator tile (rows x 16 padded bins = 4 registers per lane):
//   D[row][bin] += sum_k A[row][k] * B[k][bin],  A = 16 rows x 4 columns of F, B = 4 columns x 16 bins of R.
// The f32 MFMA runs at the VALU FMA rate -- it is used for its register economy, not for FLOPs.
// F never passes through VGPRs: each wave streams its own 16 x 128 tile into LDS with
// global_load_lds_dwordx4 (1 KiB per wave-instruction, non-temporal), double buffered; the source
// chunk order is XOR-swizzled by the row so that the row-strided ds_read_b128 of the A operand is
// bank-conflict free.  The k order inside a tile is permuted (lane group g, component c <-> column
// 16m + 4g + c) identically for A and B, which a sum over k does not see.
constexpr int MT_TC = 128;      // columns per tile
constexpr int MT_ROWS = 16;     // rows per wave
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;

template <int NW, bool SPLIT, bool LIST>
__global__ __launch_bounds__(NW * 64) void k_sweep_mfma(SweepParams P) {
    __shared__ __attribute__((aligned(16))) float sF[NW][2][MT_ROWS * MT_TC];      // 16 KiB per wave
    __shared__ __attribute__((aligned(16))) float sRt[(MT_TC / 4) * 16 * 4];       // [chunk][bin][4 columns], 8 KiB
    __shared__ float sG[NW][MT_ROWS][16];
    __shared__ float sV[MAX_BINS][NW * MT_ROWS];
    __shared__ double sD[256];
    __shared__ int sFlag;

    // converged: this pass and every pass queued behind it does nothing (latched in ctl[1]: the host keeps flipping the two
    // residual buffers per queued pass, and the older of the two has NOT converged)
    if (P.conv_mode != 0 && (P.ctl[1] != 0 || residual_converged(P))) {
        if (blockIdx.x == 0 && blockIdx.y == 0) {
            if (threadIdx.x == 0) P.ctl[1] = 1;
            if (P.n_peers > 0) publish_pass(P);        // (the gates of the passes queued behind this one still wait for its number)
        }
        return;
    }
    const int S = P.S;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rbase = (blockIdx.x * NW + wave) * MT_ROWS;
    const int ntiles_all = (P.world * P.rpr) / MT_TC;
    const int tile0 = SPLIT ? range_start(blockIdx.y, gridDim.y, ntiles_all, P.taper) : 0;
    const int ntiles = SPLIT ? range_start(blockIdx.y + 1, gridDim.y, ntiles_all, P.taper) - tile0 : ntiles_all;
    const int tiles_per_chunk = P.rpr / MT_TC;

    // bins S..15 of the residual tile are zero and never rewritten
    for (int x = tid; x < (MT_TC / 4) * 16; x += NW * 64)
        if ((x & 15) >= S) *reinterpret_cast<v4f*>(&sRt[x * 4]) = v4f{ 0.0f, 0.0f, 0.0f, 0.0f };

    // per-lane source rows of the eight DMA instructions of a tile (two rows per instruction)
    const int p_in_row = lane & 31;
    auto dma_f = [&](int tl, int b) {
        const size_t col0 = (size_t)(tile0 + tl) * MT_TC;
#pragma unroll
        for (int n = 0; n < 8; n++) {
            const int row_l = 2 * n + (lane >> 5);
            const int chunk = p_in_row ^ row_l;                   // source swizzle (row_l < 16, chunk < 32)
            const int grow = min(rbase + row_l, P.nrows - 1);
            const float* g = P.F + (size_t)grow * P.ldF + col0 + chunk * 4;
            // Issued as inline asm on purpose: with the builtin hipcc treats the DMA as an LDS store that may
            // alias the tile being read and drains it (vmcnt(0)) before the first ds_read of the CURRENT
            // tile, i.e. no overlap.  M0 = LDS byte address of this 1 KiB piece; lane i lands at +16*i.
            const unsigned lds_addr = (unsigned)(size_t)(lds_ptr_t)(&sF[wave][b][n * 256]);
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt"
                         :: "s"(lds_addr), "v"(g) : "memory", "m0");
        }
    };
    constexpr int R4 = ((MT_TC / 4) * 16 + NW * 64 - 1) / (NW * 64);
    v4f rreg[R4];
    auto load_r = [&](int tl) {
        const int t = tile0 + tl;
        const int chunk_rank = t / tiles_per_chunk;
        const int l0 = (t - chunk_rank * tiles_per_chunk) * MT_TC;
        const float* base = P.Rin + (size_t)chunk_rank * P.cstride + l0;
#pragma unroll
        for (int x = 0; x < R4; x++) {
            const int q = tid + x * NW * 64;                      // s = q / 32, 4-column chunk = q % 32
            if (q < (MT_TC / 4) * S) rreg[x] = *reinterpret_cast<const v4f*>(base + (size_t)(q >> 5) * P.rpr + (q & 31) * 4);
        }
    };
    auto store_r = [&]() {
#pragma unroll
        for (int x = 0; x < R4; x++) {
            const int q = tid + x * NW * 64;
            if (q < (MT_TC / 4) * S) *reinterpret_cast<v4f*>(&sRt[((q & 31) * 16 + (q >> 5)) * 4]) = rreg[x];
        }
    };

    // optional list of the non-zero tiles (see k_sweep): a block of this kernel spans NW*16 rows = NW/2 mask rows,
    // a tile half a mask bit
    constexpr int MAXT = LIST ? 4096 : 1;
    __shared__ unsigned short sTiles[MAXT];
    __shared__ int sNT, sPos;
    const bool use_list = LIST && (NW * MT_ROWS) % 32 == 0 && (P.tile_mask != nullptr) && (ntiles <= MAXT);
    const int start_tile = (int)(((unsigned)blockIdx.x * (unsigned)P.skew) % (unsigned)ntiles);
    int nlist = ntiles;
    if (use_list) {
        if (wave == 0) {
            constexpr int MR = NW * MT_ROWS / 32;
            const int n_mask_rows = (P.nrows + 31) / 32;
            int cnt = 0, below = 0;
            for (int base = 0; base < ntiles; base += 64) {
                const int tl = base + lane;
                bool nzb = false;
                if (tl < ntiles) {
                    const int bit = ((tile0 + tl) * MT_TC) / 256;
#pragma unroll
                    for (int m = 0; m < MR; m++) {
                        const int mr = blockIdx.x * MR + m;
                        if (mr < n_mask_rows) nzb = nzb || ((P.tile_mask[(size_t)mr * P.mask_words + (bit >> 5)] >> (bit & 31)) & 1u);
                    }
                }
                const unsigned long long m = __ballot(nzb);
                if (nzb) sTiles[cnt + __popcll(m & ((1ull << lane) - 1ull))] = (unsigned short)tl;
                cnt += __popcll(m);
                below += __popcll(__ballot(nzb && tl < start_tile));
            }
            if (lane == 0) { sNT = cnt; sPos = (below == cnt) ? 0 : below; }
        }
        __syncthreads();
        nlist = sNT;
    }
    auto tile_at = [&](int pos) { return use_list ? (int)sTiles[pos] : pos; };

    int pos = use_list ? sPos : start_tile;
    if (nlist > 0) {
        load_r(tile_at(pos));
        dma_f(tile_at(pos), 0);
        store_r();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the asm DMAs are invisible to the compiler's counters
    __syncthreads();

    typedef float f32x4 __attribute__((ext_vector_type(4)));
    f32x4 acc0 = { 0, 0, 0, 0 }, acc1 = { 0, 0, 0, 0 };
    const int r = lane & 15, g = lane >> 4;
    for (int t = 0; t < nlist; t++) {
        const bool more = (t + 1) < nlist;
        pos = (pos + 1 == nlist) ? 0 : pos + 1;
        if (more) {
            const int tt = tile_at(pos);
            load_r(tt);
            dma_f(tt, (t + 1) & 1);
        }
        const float* f = sF[wave][t & 1];
#pragma unroll
        for (int m = 0; m < MT_TC / 16; m++) {
            const int c = 4 * m + g;                               // 4-column chunk of this lane group
            const v4f fq = *reinterpret_cast<const v4f*>(&f[r * MT_TC + ((c ^ r) << 2)]);
            const v4f rq = *reinterpret_cast<const v4f*>(&sRt[(c * 16 + r) * 4]);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(fq[0], rq[0], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(fq[1], rq[1], acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(fq[2], rq[2], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(fq[3], rq[3], acc1, 0, 0, 0);
        }
        __syncthreads();                 // everyone is done with this residual tile
        // the next tile's DMA had the whole compute phase to land; it must be complete (and the next
        // residual tile stored) before anyone reads either
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (more) store_r();
        __syncthreads();
    }
    // C/D layout of the 16x16 tile: column (bin) = lane & 15, row = (lane >> 4) * 4 + register
#pragma unroll
    for (int k = 0; k < 4; k++) sG[wave][g * 4 + k][r] = acc0[k] + acc1[k];
    __syncthreads();

    // the sweep_tail above reads P.S itself; rows of wave w are block rows w*16 ..
    sweep_tail<NW * MT_ROWS, 16, NW * 64, SPLIT>(P, &sG[0][0][0], &sV[0][0], sD, &sFlag);
}

template <int S, int RR, int NW, int CPL, bool NT, int OCC = 1>
static hipError_t launch_cfg(hipStream_t st, const SweepParams& p) {
    const int rows_per_block = RR * NW;
    dim3 grid((p.nrows + rows_per_block - 1) / rows_per_block, p.ksplit);
    const bool list = p.tile_mask != nullptr;
    if (p.ksplit > 1) {
        if (list) hipLaunchKernelGGL((k_sweep<S, RR, NW, CPL, NT, OCC, true, true>), grid, dim3(NW * 64), 0, st, p);
        else hipLaunchKernelGGL((k_sweep<S, RR, NW, CPL, NT, OCC, true, false>), grid, dim3(NW * 64), 0, st, p);
    } else {
        if (list) hipLaunchKernelGGL((k_sweep<S, RR, NW, CPL, NT, OCC, false, true>), grid, dim3(NW * 64), 0, st, p);
        else hipLaunchKernelGGL((k_sweep<S, RR, NW, CPL, NT, OCC, false, false>), grid, dim3(NW * 64), 0, st, p);
    }
    return hipGetLastError();
}

// How many column ranges to cut the sweep into.  With 512 or more row blocks (2 per CU) the fused
// single pass is fastest; below that, cut columns until there are about 1024 blocks, each block
// keeping at least 8 tiles (the ranges need not be equal).  Measured at N = 65 536 on one
// MI355X (profiles/r01/sweep_shards.md): 8192 rows 0.437 ms unsplit -> 0.342 ms with 4 ranges.
// rows per wave of k_sweep: 8 (4 above 8 bins: the accumulators are RR*S registers); SweepTuning::rows_per_wave = 4 tries 4 below too
static int sweep_rr(int S, const SweepTuning& t) { return (S <= 8 && t.rows_per_wave != 4) ? 8 : 4; }

// rows per workgroup of the pass kernel launch_sweep picks: k_sweep: 4 waves x 8 rows; k_sweep_mfma: 4 x 16
static int sweep_rows_per_block(int S, const SweepTuning& t) {
    if (S > 8 && t.mfma) return 4 * MT_ROWS;
    return 4 * sweep_rr(S, t);
}
int sweep_row_blocks(int nrows, int S, const SweepTuning& t) {
    const int rpb = sweep_rows_per_block(S, t);
    return (nrows + rpb - 1) / rpb;
}

int sweep_ksplit(int nrows, int S, int total_cols, const SweepTuning& t) {
    const int rows_per_block = sweep_rows_per_block(S, t);
    const int row_blocks = (nrows + rows_per_block - 1) / rows_per_block;
    const int ntiles = total_cols / 256;
    const int forced = t.ksplit > 0 ? t.ksplit : -1;
    int want = forced > 0 ? forced : (row_blocks >= 512 ? 1 : (1024 + row_blocks - 1) / (row_blocks > 0 ? row_blocks : 1));
    int ks = 1;
    for (int k = 1; k <= want && k <= 64; k++)
        if (ntiles / k >= (forced > 0 ? 1 : 8)) ks = k;       // every range keeps at least 8 tiles
    return ks;
}

template <int S>
static hipError_t launch_sweep_s(hipStream_t st, const SweepParams& p) {
    // 8 rows per wave (4 above 8 bins: the accumulators are RR*S registers), 4 waves per block,
    // non-temporal F loads: measured best on MI355X (profiles/r01/sweep_variants.md)
    if (S <= 8 && sweep_rr(S, p.tune) == 8) return launch_cfg<S, 8, 4, 1, true>(st, p);
    return launch_cfg<S, 4, 4, 1, true>(st, p);
}

static hipError_t launch_sweep_mfma(hipStream_t st, const SweepParams& p) {
    constexpr int NW = 4;
    dim3 grid((p.nrows + NW * MT_ROWS - 1) / (NW * MT_ROWS), p.ksplit);
    const bool list = p.tile_mask != nullptr;
    if (p.ksplit > 1) {
        if (list) hipLaunchKernelGGL((k_sweep_mfma<NW, true, true>), grid, dim3(NW * 64), 0, st, p);
        else hipLaunchKernelGGL((k_sweep_mfma<NW, true, false>), grid, dim3(NW * 64), 0, st, p);
    } else {
        if (list) hipLaunchKernelGGL((k_sweep_mfma<NW, false, true>), grid, dim3(NW * 64), 0, st, p);
        else hipLaunchKernelGGL((k_sweep_mfma<NW, false, false>), grid, dim3(NW * 64), 0, st, p);
    }
    return hipGetLastError();
}

// A rank WITHOUT rows (shard_rows leaves the last ranks empty when N < world * rows-per-rank, e.g. the reference's 6400-patch
// scene on 8 GPUs) still takes part in a converge run: it must take the same decision from the gathered sums as the ranks that
// sweep, latch it, count the real passes -- dr_solver_converge reads ctl on every rank, and a rank that stopped queueing passes
// would leave its peers alone in the all-gather -- and leave a zero tail in its (all-zero) chunk.
__global__ void k_sweep_norows(SweepParams P) {
    if (P.conv_mode != 0 && (P.ctl[1] != 0 || residual_converged(P))) {
        if (threadIdx.x == 0) P.ctl[1] = 1;
        if (P.n_peers > 0) publish_pass(P);
        return;
    }
    if (P.want_sums) {
        const size_t at = (size_t)P.rank * P.cstride + (size_t)P.S * P.rpr;
        if ((int)threadIdx.x < P.S) {
            reinterpret_cast<double*>(P.Rout + at)[threadIdx.x] = 0.0;
            for (int q = 0; q < P.n_peers; q++)
                if (P.peers[q]) st_system(reinterpret_cast<double*>(P.peers[q] + at) + threadIdx.x, 0.0);
        }
        if (threadIdx.x == 0) P.ctl[0] = P.ctl[0] + 1;
    }
    if (P.n_peers > 0) { __syncthreads(); publish_pass(P); }
}

hipError_t launch_sweep(hipStream_t st, const SweepParams& p_in) {
    if (p_in.nrows <= 0) {
        if (p_in.conv_mode == 0 && !p_in.want_sums && p_in.n_peers == 0) return hipSuccess;
        hipLaunchKernelGGL(k_sweep_norows, dim3(1), dim3(64), 0, st, p_in);
        return hipGetLastError();
    }
    SweepParams p = p_in;
    // Start tile of a block = blockIdx.x * skew (mod tiles).  While the gathered residual fits a slice of L2
    // (4 MiB per XCD) a skew decorrelates the blocks' column positions, worth +0.5 % at N = 65 536; once it does
    // not (N = 131 072, S = 8: 4 MiB), blocks that run through the columns in step keep the few residual tiles
    // they are all reading in L2: 37 -> 0 is +8 % there and +14 % at N = 262 144 (profiles/r01/sweep_shards.md).
    const size_t residual_bytes = sizeof(float) * (size_t)p.world * p.rpr * p.S;
    p.skew = p.tune.skew >= 0 ? p.tune.skew : (residual_bytes <= ((size_t)5 << 19) ? 37 : 0);
    // column ranges of decreasing size (shares n, n-1, .., 1) when the columns are cut at all: the blocks dispatched last are the
    // shortest, which trims the drain of a small shard's pass (8192 rows of the 64k problem, 4 ranges: 0.348 -> 0.333 ms;
    // profiles/r02/sweep_shards.md).  SweepTuning::taper: 0 = equal ranges, t > 0 = shares t + n - y
    p.taper = p.ksplit > 1 ? p.tune.taper : 0;
    const int mfma = p.tune.mfma;
    if (p.S > 8 && mfma) return launch_sweep_mfma(st, p);
    switch (p.S) {
#define DR_CASE(n) case n: return launch_sweep_s<n>(st, p);
        DR_CASE(1) DR_CASE(2) DR_CASE(3) DR_CASE(4) DR_CASE(5) DR_CASE(6) DR_CASE(7) DR_CASE(8)
        DR_CASE(9) DR_CASE(10) DR_CASE(11) DR_CASE(12) DR_CASE(13) DR_CASE(14) DR_CASE(15) DR_CASE(16)
#undef DR_CASE
    }
    return hipErrorInvalidValue;
}

// One bit per block of 32 rows x 256 columns of the F shard: does it hold a non-zero?  A workgroup per row block,
// a wave per 8 rows, the same 1 KiB-per-row reads as the sweep; 32 tiles are collected in a register word before
// the one atomicOr that publishes them.  mask must be zeroed beforehand.
__global__ __launch_bounds__(256) void k_tile_mask(const float* __restrict__ F, int nrows, size_t ldF, unsigned* __restrict__ mask,
                                                   int mask_words) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int rb = blockIdx.x;
    const int ntiles = (int)(ldF / 256);
    unsigned* mrow = mask + (size_t)rb * mask_words;
    for (int w0 = blockIdx.y * 32; w0 < ntiles; w0 += gridDim.y * 32) {
        unsigned word = 0;
        for (int b = 0; b < 32 && w0 + b < ntiles; b++) {
            bool nz = false;
#pragma unroll
            for (int r = 0; r < 8; r++) {
                const int row = rb * 32 + wave * 8 + r;
                if (row < nrows) {
                    const v4f x = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(F + (size_t)row * ldF + (size_t)(w0 + b) * 256 + lane * 4));
                    nz = nz || (x.x != 0.0f) || (x.y != 0.0f) || (x.z != 0.0f) || (x.w != 0.0f);
                }
            }
            if (__ballot(nz) != 0ull) word |= 1u << b;
        }
        if (lane == 0 && word) atomicOr(&mrow[w0 >> 5], word);
    }
}

hipError_t launch_tile_mask(hipStream_t st, const float* F, int nrows, size_t ldF, unsigned* mask, int mask_words) {
    if (nrows <= 0) return hipSuccess;
    const int row_blocks = (nrows + 31) / 32;
    hipError_t e = hipMemsetAsync(mask, 0, sizeof(unsigned) * (size_t)row_blocks * mask_words, st);
    if (e != hipSuccess) return e;
    const int ntiles = (int)(ldF / 256);
    const int gy = std::max(1, std::min((ntiles + 31) / 32, 8));
    hipLaunchKernelGGL(k_tile_mask, dim3(row_blocks, gy), dim3(256), 0, st, F, nrows, ldF, mask, mask_words);
    return hipGetLastError();
}

// per-bin sums of every chunk of a residual buffer into the chunk's own tail (after a reset; the passes keep the
// tails themselves); check_convergence, vs/Lightning.h:255-261, in double
__global__ void k_chunk_sums(float* __restrict__ R, int S, int rpr, size_t cstride) {
    __shared__ double sh[256];
    const int c = blockIdx.x, s = blockIdx.y;
    float* chunk = R + (size_t)c * cstride;
    const float* x = chunk + (size_t)s * rpr;
    double a = 0.0;
    for (int i = threadIdx.x; i < rpr; i += 256) a += (double)x[i];
    sh[threadIdx.x] = a;
    __syncthreads();
    for (int w = 128; w >= 1; w >>= 1) {
        if ((int)threadIdx.x < w) sh[threadIdx.x] += sh[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) reinterpret_cast<double*>(chunk + (size_t)S * rpr)[s] = sh[0];
}

hipError_t launch_chunk_sums(hipStream_t st, float* R, int world, int S, int rpr, size_t cstride) {
    hipLaunchKernelGGL(k_chunk_sums, dim3(world, S), dim3(256), 0, st, R, S, rpr, cstride);
    return hipGetLastError();
}

__global__ void k_scatter_rows(const float* __restrict__ src, int N, int S, int rpr, size_t cstride, float* __restrict__ dst) {
    size_t x = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= (size_t)N * S) return;
    int s = (int)(x / N), i = (int)(x % N);
    int c = i / rpr, l = i % rpr;
    dst[(size_t)c * cstride + (size_t)s * rpr + l] = src[(size_t)i * S + s];
}
__global__ void k_gather_rows(const float* __restrict__ src, int N, int S, int rpr, size_t cstride, float* __restrict__ dst) {
    size_t x = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= (size_t)N * S) return;
    int s = (int)(x / N), i = (int)(x % N);
    int c = i / rpr, l = i % rpr;
    dst[(size_t)i * S + s] = src[(size_t)c * cstride + (size_t)s * rpr + l];
}

hipError_t launch_scatter_rows(hipStream_t st, const float* src, int N, int S, int rpr, size_t cstride, float* dst) {
    size_t n = (size_t)N * S;
    hipLaunchKernelGGL(k_scatter_rows, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, src, N, S, rpr, cstride, dst);
    return hipGetLastError();
}
hipError_t launch_gather_rows(hipStream_t st, const float* src, int N, int S, int rpr, size_t cstride, float* dst) {
    size_t n = (size_t)N * S;
    hipLaunchKernelGGL(k_gather_rows, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, src, N, S, rpr, cstride, dst);
    return hipGetLastError();
}

}  // namespace dr
