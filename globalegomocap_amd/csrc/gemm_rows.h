// fp32 MFMA GEMM for FEW rows (gfx950): the linear layers around the latent code for one sequence,
//
//   forward        pre0[M, T*256] = z[perm[M], 2048] . Wf^T + bf      decoder_input composed with the first decoder conv
//                                                                     (compose_front, gem_api.hip; SeqConvVAE.py:62,67-75,131-135)
//   backward-data  dz[M, 2048]    = dpre0[M, T*256] . Wf              (its adjoint; the VAE is frozen, optimizer.py:261-270)
//
// (or the separate decoder_input products [M, 2048] x [2048, 5120] and back when the composition is off; the shapes quoted
// below are those)
// with M = the windows that are still iterating (<= 240 for BASELINE configs[1]).  A 64x64-tiled GEMM has 4 row tiles here:
// it needs split-K 4..10 to give every CU work (20-40 MB of fp32 slabs per launch plus a reduce pass), pays for 256 rows
// whenever more than 192 windows are active, and hides HBM latency only through 5 co-resident workgroups per CU.
// This kernel turns the shape around:
//
//   * a workgroup owns ONE 64-column tile of the weights and a block of up to RT row tiles of 16 rows, and walks K -- all of
//     it (forward: 3 row blocks x 80 column tiles = 240 workgroups, no split-K, no slabs, no reduce pass) or one of a few K
//     slices (backward: N = 2048 has only 32 column tiles);
//   * one workgroup per CU, one wave per SIMD.  A K-step is 64 floats (256-byte LDS rows); wave w takes the w-th quarter of
//     every K-step and computes the WHOLE tile for it: NRT x 4 independent 16x16 accumulators (v_mfma_f32_16x16x4_f32), fed
//     by NRT + 4 ds_read_b128 per 4 NRT x 4 MFMAs.  With a single wave per SIMD nothing else can fill the matrix pipe while
//     that wave issues other instructions, so there must be few of them: ~0.5 per MFMA here, threaded between the MFMAs
//     (sched_group_barrier); the four partial tiles are summed through LDS once, at the end, in wave order;
//   * latency is hidden by DEPTH instead of occupancy: a ring of S K-steps is kept in flight with global_load_lds_dwordx4
//     (operands never touch a VGPR) and a COUNTED s_waitcnt vmcnt; the fragments of step t+1 are read into registers during
//     the MFMAs of step t; one raw s_barrier per K-step says both "everybody's pieces of the next step have landed" and
//     "everybody is done with the buffer the next DMA overwrites".  Every step issues the same number of DMA instructions
//     (past the end of the slice the last step is fetched again into a buffer nobody reads): exact counts, branch-free body;
//   * rows are cut in tiles of 16, re-dealt over the row blocks on the device every round (row count from n_active):
//     229 active windows cost 15 row tiles, not 256 rows; the loop is compiled once per tile count;
//   * LDS position p of row r holds the row's 16-byte chunk p ^ (r & 15) (the swizzle is applied to the DMA's per-lane source
//     address, its LDS side is lane-linear): the fragment reads of 16 rows x 4 chunks are conflict-free.  The weight fragment is
//     the MFMA's A operand, so a lane ends up with 4 consecutive output columns of one row and stores 16 bytes.
//
// Summation order over K is fixed (K-steps in order inside a wave's quarter, quarters in wave order; the pairing of k values
// inside a quarter is a permutation): results do not depend on how many rows are active and are bitwise reproducible.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

namespace gem {
namespace rows {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int BN = 64;                    // columns per workgroup
constexpr int BK = 64;                    // one K-step = 256 bytes of every row; wave w multiplies bytes 64w .. 64w+63
constexpr int ROW_BYTES = BK * 4;

struct Args {
    const float* A;           // [rows, lda]
    const float* W;           // [N][K], k contiguous
    const float* bias;        // [N] (direct output only) or nullptr
    int lrelu;                // direct output only: LeakyReLU(0.01) after the bias
    float* C;                 // [M, ldc]; split-K: raw fp32 slabs, slab z at C + z * slab_stride
    const int* m_dev;         // device row count (evaluation rounds) or nullptr
    const int* row_map;       // gathered A rows or nullptr
    int lda, ldc, M, N, K;
    // fused compaction (FUSE): the stable partition of the windows -- still iterating first -- that compact_kernel
    // (lbfgs.hip) otherwise computes in a launch of its own.  Every workgroup derives the row count and the row map from the
    // per-window phase array; workgroup 0 also publishes them for the later kernels of the round.
    const int* phase_arr;     // [n_windows] L-BFGS phase of every window
    int n_windows, done_phase, T;
    int* perm_out;            // [n_windows] slot -> window
    int* slot_of_out;         // [n_windows] window -> slot
    int* n_active_out;        // {n_active, n_active * T}
    int* log_slot;            // profiling: n_active of this round
    int n_rb;                 // row blocks of the launch (fixed); the row tiles in use are re-dealt over them
    int n_split, tiles_per_split;
    size_t slab_stride;
};

template <int RT>
struct Geometry {
    static constexpr int A_BYTES = RT * 16 * ROW_BYTES, B_BYTES = BN * ROW_BYTES, STAGE_BYTES = A_BYTES + B_BYTES;
    static constexpr int A_INSTR = A_BYTES / 1024 / 4, B_INSTR = B_BYTES / 1024 / 4;       // 1 KB DMA pieces per wave and K-step
    static constexpr int PIECES = A_INSTR + B_INSTR;
    static_assert(A_BYTES % 4096 == 0, "the A tile must split into whole DMA pieces per wave");
};

#ifndef GEM_ROWS_ABLATE
#define GEM_ROWS_ABLATE 0          // tools/gemm_rows_bench only: 1 no fragment reads, 2 no MFMAs, 3 no DMA
#endif

#ifdef GEM_ROWS_CLOCK
__device__ long long g_rows_clock[6];
#endif

constexpr int FUSE_MAX_WINDOWS = 512;

// The kernel's body as a device function (the early returns leave the body, not the kernel): gemm_rows_kernel below is this and
// nothing else; lbfgs.hip's experimental rows_bwd_lbfgs_kernel runs it in front of a device-wide barrier.
template <int S, int RT, bool FUSE = false>
__device__ __forceinline__ void gemm_rows_body(const Args& a) {
    typedef Geometry<RT> G;
    static_assert(S >= 3 && S * G::STAGE_BYTES <= 160 * 1024, "ring must fit the LDS");
    static_assert(4 * RT * 4096 <= S * G::STAGE_BYTES, "the four partial tiles are reduced through the ring's LDS");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef GEM_ROWS_CLOCK
    const long long w_begin = wall_clock64();
#endif

    // ---- workgroup -> (column tile, K slice, row block).  Consecutive workgroup ids are dealt round-robin over the 8 XCDs;
    // the remap gives every XCD a contiguous range of logical ids, ordered so that the row blocks that share a weight tile
    // are neighbours (one HBM fetch of the tile per XCD).
    const int n_ct = a.N / BN;
    const int total = n_ct * a.n_split * a.n_rb;
    const int id = blockIdx.x;
    if (id >= total) return;
    const int q8 = total >> 3, r8 = total & 7, x8 = id & 7, k8 = id >> 3;
    const int pid = (x8 < r8 ? x8 * (q8 + 1) : r8 * (q8 + 1) + (x8 - r8) * q8) + k8;
    const int rb = pid % a.n_rb;
    const int ks = (pid / a.n_rb) % a.n_split;
    const int ct = pid / (a.n_rb * a.n_split);
    const int n0 = ct * BN;
    const int kTiles = a.K / BK;
    const int kt_begin = ks * a.tiles_per_split, kt_end = min(kTiles, kt_begin + a.tiles_per_split);
    const int nk = kt_end - kt_begin;
    if (nk <= 0) return;

    // ---- DMA sources.  Piece j of this wave covers tile rows 4*(wave*INSTR + j) .. +3; lane l brings the 16-byte chunk
    // (l & 15) ^ (row & 15) of row (l >> 4).  All addresses are a wave-uniform base that advances 256 bytes per K-step plus a
    // fixed 32-bit lane offset.  The weight pieces of the first S-1 steps depend on nothing but the workgroup id: they are on
    // their way to LDS before the row count and the row map (written by the previous kernels of the round) are even read.
    const int lrow = lane >> 4, lchunk = lane & 15;
    unsigned b_off[G::B_INSTR];
#pragma unroll
    for (int j = 0; j < G::B_INSTR; ++j) {
        const int r = (wave * G::B_INSTR + j) * 4 + lrow;
        b_off[j] = (unsigned)r * (unsigned)a.K * 4u + (unsigned)((lchunk ^ (r & 15)) * 16);
    }
    const unsigned char* b_base = reinterpret_cast<const unsigned char*>(a.W + (size_t)n0 * a.K) + (size_t)kt_begin * ROW_BYTES;
#pragma unroll
    for (int i = 0; i < S - 1; ++i) {
        unsigned char* lb = smem + i * G::STAGE_BYTES + G::A_BYTES + wave * (G::B_INSTR * 1024);
        const unsigned char* src = b_base + (size_t)min(i, nk - 1) * ROW_BYTES;
#pragma unroll
        for (int j = 0; j < G::B_INSTR; ++j)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + b_off[j]),
                                             (__attribute__((address_space(3))) void*)(lb + j * 1024), 16, 0, 0);
    }

    int M;
    __shared__ int s_perm[FUSE ? FUSE_MAX_WINDOWS : 1];
    __shared__ int s_wsum[4];
    if (FUSE) {
        // windows that are still iterating, in order, to the front (stable: a window's slot does not depend on the launch)
        int base = 0;
        for (int start = 0; start < a.n_windows; start += 256) {
            const int b = start + tid;
            const bool active = b < a.n_windows && a.phase_arr[b] != a.done_phase;
            const unsigned long long m = __ballot(active);
            const int prefix = __popcll(m & ((1ull << lane) - 1ull));
            if (lane == 0) s_wsum[wave] = __popcll(m);
            __syncthreads();
            int woff = 0, total = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) { if (i < wave) woff += s_wsum[i]; total += s_wsum[i]; }
            const int before = base + woff + prefix;           // active windows in front of b
            if (active) s_perm[before] = b;
            if (blockIdx.x == 0 && b < a.n_windows) {
                // finished windows go behind the active ones, in order: slot = n_active + (finished windows in front of b);
                // n_active is only known after the last chunk, so their slots are written in the second sweep below
                if (active) { a.perm_out[before] = b; a.slot_of_out[b] = before; }
            }
            __syncthreads();
            base += total;
        }
        M = base;
        if (blockIdx.x == 0) {
            int fin = 0;                                        // finished windows in front of the chunk
            for (int start = 0; start < a.n_windows; start += 256) {
                const int b = start + tid;
                const bool done = b < a.n_windows && a.phase_arr[b] == a.done_phase;
                const unsigned long long m = __ballot(done);
                const int prefix = __popcll(m & ((1ull << lane) - 1ull));
                __syncthreads();
                if (lane == 0) s_wsum[wave] = __popcll(m);
                __syncthreads();
                int woff = 0, total = 0;
#pragma unroll
                for (int i = 0; i < 4; ++i) { if (i < wave) woff += s_wsum[i]; total += s_wsum[i]; }
                if (done) { const int sl = M + fin + woff + prefix; a.perm_out[sl] = b; a.slot_of_out[b] = sl; }
                fin += total;
            }
            if (tid == 0) { a.n_active_out[0] = M; a.n_active_out[1] = M * a.T; *a.log_slot = M; }
        }
    } else {
        M = a.m_dev ? *a.m_dev : a.M;
    }
    const int R = (max(M, 0) + 15) >> 4;                       // row tiles in use this round
    const int rpb = (R + a.n_rb - 1) / a.n_rb;                 // ... per row block (<= RT by the launch's choice of n_rb)
    const int t0 = rb * rpb;
    const int nrt = min(min(rpb, R - t0), RT);                 // row tiles of this workgroup
    if (nrt <= 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the LDS must not be handed on with DMA writes still in flight
        return;
    }
    const int m0 = t0 * 16;
    // Rows past this block's tiles are never multiplied; rows past M inside the last tile are multiplied but never stored (a
    // row of the output depends on its own row of A only): both read a valid row.
    unsigned a_off[G::A_INSTR];
#pragma unroll
    for (int j = 0; j < G::A_INSTR; ++j) {
        const int r = (wave * G::A_INSTR + j) * 4 + lrow;
        int src = min(m0 + r, M - 1);
        if (FUSE) src = s_perm[src];
        else if (a.row_map) src = a.row_map[src];
        a_off[j] = (unsigned)src * (unsigned)a.lda * 4u + (unsigned)((lchunk ^ (r & 15)) * 16);
    }
    const unsigned char* a_base = reinterpret_cast<const unsigned char*>(a.A) + (size_t)kt_begin * ROW_BYTES;
#pragma unroll
    for (int i = 0; i < S - 1; ++i) {
        unsigned char* la = smem + i * G::STAGE_BYTES + wave * (G::A_INSTR * 1024);
        const unsigned char* src = a_base + (size_t)min(i, nk - 1) * ROW_BYTES;
#pragma unroll
        for (int j = 0; j < G::A_INSTR; ++j)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + a_off[j]),
                                             (__attribute__((address_space(3))) void*)(la + j * 1024), 16, 0, 0);
    }
    int staged = S - 1;                             // K-steps handed to the DMA so far
    a_base += (size_t)min(S - 1, nk - 1) * ROW_BYTES;
    b_base += (size_t)min(S - 1, nk - 1) * ROW_BYTES;
    auto stage = [&](int buf) {                     // next K-step of this slice -> ring buffer buf
        unsigned char* la = smem + buf * G::STAGE_BYTES + wave * (G::A_INSTR * 1024);
        unsigned char* lb = smem + buf * G::STAGE_BYTES + G::A_BYTES + wave * (G::B_INSTR * 1024);
#pragma unroll
        for (int j = 0; j < G::A_INSTR; ++j)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a_base + a_off[j]),
                                             (__attribute__((address_space(3))) void*)(la + j * 1024), 16, 0, 0);
#pragma unroll
        for (int j = 0; j < G::B_INSTR; ++j)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(b_base + b_off[j]),
                                             (__attribute__((address_space(3))) void*)(lb + j * 1024), 16, 0, 0);
        ++staged;
        const int adv = staged < nk ? ROW_BYTES : 0;       // stay on the last K-step once the slice is exhausted
        a_base += adv;
        b_base += adv;
    };

    // ---- fragments: lane (fr, fh) of wave w reads the 16-byte chunk 4w + fh of row fr of a 16-row block: 4 fp32 = one k
    // value for each of FOUR MFMAs (the pairing of k values inside the quarter is a permutation a sum over k does not see)
    const int fr = lane & 15, fh = lane >> 4;
    const int f_off = fr * ROW_BYTES + (((4 * wave + fh) ^ fr) << 4);        // inside any 16-row block ((16 j + fr) & 15 == fr)
    const int col = n0 + 4 * fh;
    float* Cb = a.C + (a.n_split > 1 ? (size_t)ks * a.slab_stride : 0);

    // the loop is compiled once per row-tile count: straight-line MFMA clusters, no predicates inside
    auto body = [&](auto nrt_c) {
        constexpr int NRT = decltype(nrt_c)::value;
        struct Frags { f32x4 w[4]; f32x4 a[NRT]; };          // one K-step of this wave: 4 weight + NRT activation fragments
        f32x4 acc[4][NRT];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < NRT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        auto fetch = [&](Frags& f, int buf) {
            const unsigned char* base = smem + buf * G::STAGE_BYTES + f_off;
#pragma unroll
            for (int i = 0; i < 4; ++i) f.w[i] = *reinterpret_cast<const f32x4*>(base + G::A_BYTES + i * 16 * ROW_BYTES);
#pragma unroll
            for (int j = 0; j < NRT; ++j) f.a[j] = *reinterpret_cast<const f32x4*>(base + j * 16 * ROW_BYTES);
        };
        auto mfmas = [&](const Frags& f) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < NRT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.w[i][e], f.a[j][e], acc[i][j], 0, 0, 0);
        };
        // ---- main loop.  A wave's DMA pieces retire in issue order: vmcnt((S-3) * PIECES) = "my pieces of step t+1 have
        // landed"; the barrier extends that to every wave's pieces and says that everybody has finished reading step t-1
        // (into registers, a step ago), whose buffer the DMA issued right after the barrier overwrites.
        int cur = 0;        // buffer of step t
        auto step = [&](const Frags& f, Frags& g, auto wait_c) {
            const int nb = cur + 1 == S ? 0 : cur + 1;
            if (GEM_ROWS_ABLATE != 6) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(decltype(wait_c)::value) : "memory");
            if (GEM_ROWS_ABLATE != 6) __builtin_amdgcn_s_barrier();
            if (GEM_ROWS_ABLATE != 3 && GEM_ROWS_ABLATE != 6) stage(cur == 0 ? S - 1 : cur - 1);
            if (GEM_ROWS_ABLATE != 1 && GEM_ROWS_ABLATE != 6) fetch(g, nb);
            if (GEM_ROWS_ABLATE != 2) mfmas(f);
#pragma unroll
            for (int i = 0; i < 16 * NRT; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);       // one MFMA
                __builtin_amdgcn_sched_group_barrier(0x126, 1, 0);       // one of VALU | SALU | VMEM read | DS read
            }
            cur = nb;
        };
        // The prologue issued [weights of steps 0..S-2][activations of steps 0..S-2]: step 0 is complete once at most the
        // activation pieces of the S-2 later steps are outstanding; the first loop steps count the same way.
        static_assert(S == 3 || S == 4, "the peeled first step below is written for rings of 3 or 4");
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((S - 2) * G::A_INSTR) : "memory");
        __builtin_amdgcn_s_barrier();
        const std::integral_constant<int, (S - 3) * G::PIECES> steady;
        Frags f0, f1;
        fetch(f0, 0);
#ifdef GEM_ROWS_CLOCK
        if (GEM_ROWS_ABLATE == 6) f1 = f0;
        const long long c0 = clock64(), w0 = wall_clock64();
#endif
        if constexpr (S == 4) {
            step(f0, f1, std::integral_constant<int, G::A_INSTR>{});         // needs step 1: only step 2's activations may be out
            for (int t = 1; t < nk; t += 2) {
                step(f1, f0, steady);
                if (t + 1 < nk) step(f0, f1, steady);
            }
        } else {
            for (int t = 0; t < nk; t += 2) {
                step(f0, f1, steady);
                if (t + 1 < nk) step(f1, f0, steady);
            }
        }
#ifdef GEM_ROWS_CLOCK
        const long long w1 = wall_clock64();
        if (blockIdx.x == 8 && tid == 0) { g_rows_clock[0] = clock64() - c0; g_rows_clock[1] = w1 - w0; g_rows_clock[2] = w0 - w_begin; }
#endif
        // ---- epilogue: the four partial tiles meet in LDS (the ring is dead: wait for the DMA instructions that re-fetched
        // the last step, then for everybody's last fragment reads); wave w sums column tile w in wave order and stores it.
        // D[n][m]: the lane's registers are columns n0 + 16 i + 4 fh .. +3 of row m0 + 16 j + fr.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < NRT; ++j)
                *reinterpret_cast<f32x4*>(smem + ((wave * 4 + i) * NRT + j) * 1024 + lane * 16) = acc[i][j];
        __syncthreads();
        f32x4 bv = {0.f, 0.f, 0.f, 0.f};
        if (a.n_split == 1 && a.bias) bv = *reinterpret_cast<const f32x4*>(a.bias + col + 16 * wave);
#pragma unroll
        for (int j = 0; j < NRT; ++j) {
            f32x4 v = *reinterpret_cast<const f32x4*>(smem + ((0 * 4 + wave) * NRT + j) * 1024 + lane * 16);
#pragma unroll
            for (int w = 1; w < 4; ++w) v += *reinterpret_cast<const f32x4*>(smem + ((w * 4 + wave) * NRT + j) * 1024 + lane * 16);
            const int row = m0 + j * 16 + fr;
            if (row < M) {
                v += bv;
                if (a.lrelu && a.n_split == 1) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : v[e] * 0.01f;
                }
                *reinterpret_cast<f32x4*>(Cb + (size_t)row * a.ldc + col + 16 * wave) = v;
            }
        }
#ifdef GEM_ROWS_CLOCK
        if (blockIdx.x == 8 && tid == 0) g_rows_clock[3] = wall_clock64() - w1;
#endif
    };
    switch (nrt) {
        case 1: body(std::integral_constant<int, 1>{}); break;
        case 2: body(std::integral_constant<int, (RT >= 2 ? 2 : RT)>{}); break;
        case 3: body(std::integral_constant<int, (RT >= 3 ? 3 : RT)>{}); break;
        case 4: body(std::integral_constant<int, (RT >= 4 ? 4 : RT)>{}); break;
        case 5: body(std::integral_constant<int, (RT >= 5 ? 5 : RT)>{}); break;
        case 6: body(std::integral_constant<int, (RT >= 6 ? 6 : RT)>{}); break;
        case 7: body(std::integral_constant<int, (RT >= 7 ? 7 : RT)>{}); break;
        default: body(std::integral_constant<int, RT>{}); break;
    }
}

template <int S, int RT, bool FUSE = false>
__global__ __launch_bounds__(256) void gemm_rows_kernel(const Args a) { gemm_rows_body<S, RT, FUSE>(a); }

// How a launch is cut: row blocks (fixed for the launch) and K slices.  Cost in units of one K-step of one row tile;
// a slab costs its write plus the consumer's read.
struct Plan { int n_rb, n_split, per; double fill, cost; };
inline Plan plan(int M, int N, int K, int rt_max, int n_cu, bool allow_split, size_t slab_capacity_elems, int ldc) {
    Plan best{0, 0, 0, 0.0, 1e30};
    if (N % BN != 0 || K % BK != 0 || M <= 0) return best;
    const int R = (M + 15) / 16, n_ct = N / BN, nk = K / BK;
    double best_cost = 1e30;
    for (int n_rb = (R + rt_max - 1) / rt_max; n_rb <= R; ++n_rb) {
        const int rpb = (R + n_rb - 1) / n_rb;
        if (rpb < 3 && n_rb > (R + rt_max - 1) / rt_max) break;          // thinner blocks only re-read the weights
        for (int sk = 1; sk <= (allow_split ? 8 : 1); ++sk) {
            if ((long)n_rb * n_ct * sk > n_cu) break;
            if (sk > 1 && (nk / sk < 4 || (size_t)sk * M * ldc > slab_capacity_elems)) break;
            const int per = (nk + sk - 1) / sk;
            const double slab_units = sk > 1 ? sk * ((double)M * N * 8.0 / 3.0e12) / 0.214e-6 : 0.0;
            // a K-step costs its MFMAs (rpb units) plus ~0.7 units of per-step overhead (measured: 40 steps of 4 row tiles
            // take 95.8 k cycles, 20 steps of 8 take 88.8 k: tools/gemm_rows_bench)
            const double cost = ((double)rpb + 0.7) * per + slab_units;
            if (cost < best_cost) { best_cost = cost; best = Plan{n_rb, sk, per, (double)R * n_ct * nk / ((double)n_cu * rpb * per), cost}; }
        }
    }
    return best;
}

}  // namespace rows
}  // namespace gem
