// One-round bf16 GEMM for the composed front layer of the bf16 decoder mode at large batch (gfx950):
//
//   forward   C[M, 2560] = lrelu( trial[perm][M, 2048] . Wf[2560, 2048]^T + bf )     bf16 out   (SeqConvVAE.py:62,67-75: decoder_input o conv 0)
//   backward  dz[M, 2048] = g[M, 2560] . Wf^T[2048, 2560]^T                           fp32 out   (its transpose, backward-data)
//
// Why another kernel: gemm_glds.h (128 x 128 x 64 tiles, four waves, two workgroups per CU) sits at 0.28-0.33 of the bf16 matrix peak
// on these shapes for three reasons that are properties of its tile, not of its inner loop (DESIGN.md 4, round 3 ablation): 64 FLOP
// per byte moved L2 -> LDS puts it on the ~30 B/clk/CU fill rate; 8192 x 2560 is 2.5 rounds of 512 resident workgroups (0.83); every
// tile pays its own pipeline prologue and two-pass epilogue.  This kernel turns all three knobs at once:
//   * tile 256 x 256 (backward, N = 2048: 32 x 8 = 256 tiles) or 256 x 320 (forward, N = 2560: 32 x 8 = 256 tiles): at 8192 rows
//     EXACTLY one tile per CU -- one round, no quantisation, no persistent loop or stream-K fix-up to get there -- and 128 / 142 FLOP
//     per byte of operand traffic (half / 0.45 of the 128 x 128 tile's bytes);
//   * 16 waves (1024 threads) per workgroup = four waves per SIMD, each owning 64 x 64 / 64 x 80 of the tile (16 / 20 independent
//     16x16 accumulators): while one wave waits at the K-step barrier or on its LDS fragments, three others feed the matrix pipe;
//   * one prologue and one epilogue per CU per launch.
// Operands go global -> LDS by global_load_lds_dwordx4 (1 KB pieces: 8 rows x 128 bytes), double-buffered over 64-deep K-steps
// (2 x 64 / 72 KB of LDS); rows are 128 bytes with the 16-byte chunk c of row r at position c ^ ((r >> 1) & 7) (swizzle applied on
// the DMA's per-lane source address), which makes the ds_read_b128 fragment reads conflict-free; the MFMA takes the weight fragment
// as operand A, so a lane ends up with four consecutive output columns of one row; the fp32 tile leaves through LDS in four passes
// of 64 rows as whole rows (16-byte stores).  Fewer than 8192 rows: fewer tiles, same duration -- the caller keeps gemm_glds.h for
// launches that would leave most CUs idle (device-side: both kernels look at the row count and one of them returns at once).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gemm_glds.h"

namespace gem {

namespace big {

using glds::bf16x8;
using glds::f32x4;
using glds::u32x2;
using glds::u32x4;
using glds::pack_bf16;

enum { EPI_BIAS_LRELU = 1, EPI_NONE = 3 };

struct Args {
    const uint16_t* A;        // [rows, lda] bf16
    const uint16_t* W;        // [N][K] bf16, k contiguous
    const float* bias;        // [N] (EPI_BIAS_LRELU)
    void* C;                  // bf16 or fp32 [M, ldc]
    const uint16_t* zero16;   // >= 16 zero bytes: DMA source of rows past M
    const int* m_dev;         // device row count (evaluation rounds) or nullptr
    const int* row_map;       // gathered A rows or nullptr
    int lda, ldc, M, N, K;
    int m_min;                // run only when the row count is >= m_min (the small-tile kernel takes the launches below)
    // K cut (round 5; fp32 output only): slice ks = K-tiles [ks * tiles_per_split, ...) goes to the raw fp32 slab C + ks * slab_stride;
    // the consumer sums the slabs (the tail while staging its input, lbfgs_advance while reading its gradient).  n_split <= 1: none.
    int n_split, tiles_per_split;
    size_t slab_stride;       // elements between slabs
    int m_max;                // > 0: run only when the row count is < m_max
};

// Tile shape = a grid of WM x WN waves, each owning MB x NB blocks of 16 x 16: BM = 16 MB WM = 256 rows, BN = 16 NB WN = 256
// (backward) or 320 (forward) columns.  <4, 4, 4, 4 | 5>: sixteen waves of 64 x 64 | 80 (round 4's first form: every K-step moves
// 256 | 288 KB of fragments LDS -> registers for 2048 | 2560 MFMA cycles per SIMD -- the LDS pipe, 128 bytes per clock, is the
// longer of the two on paper) -- the form the product launches.  Bigger register tiles cut the fragment traffic but measure SLOWER
// (profiles/gemm_big_variants_r04.txt, 8192 rows, forward | backward: 16 waves 0.46 | 0.47 of the bf16 peak; eight waves of
// 64 x 160 | 128: 0.40 | 0.38; eight of 128 x 80 | 64: 0.40 | 0.41; four of 128 x 160 | 128: 0.24 | 0.34): with fewer waves per SIMD
// nothing covers a wave's own LDS latency and barrier waits; the fragment traffic is not what binds this kernel.
template <int WM, int WN, int MB, int NB, int EPI, bool OUT_BF16>
__global__ __launch_bounds__(WM * WN * 64) void gemm_big_kernel(const Args a) {
    constexpr int NW = WM * WN, NT = NW * 64;
    constexpr int BM = 16 * MB * WM, BN = 16 * NB * WN;
    static_assert(BM == 256, "one 256-row panel per workgroup");
    constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, BUF = A_BYTES + B_BYTES;
    constexpr int A_PIECES = A_BYTES / 1024, B_PIECES = B_BYTES / 1024;          // 32; 32 or 40
    constexpr int A_PER = (A_PIECES + NW - 1) / NW, B_PER = (B_PIECES + NW - 1) / NW;      // DMA pieces per wave
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int M = a.m_dev ? *a.m_dev : a.M;
    if (M < a.m_min || M <= 0 || (a.m_max > 0 && M >= a.m_max)) return;
    const int n_mt = (M + BM - 1) / BM, n_nt = a.N / BN;
    const int n_split = (!OUT_BF16 && EPI == EPI_NONE && a.n_split > 1) ? a.n_split : 1;
    const int total = n_mt * n_nt * n_split;
    if ((int)blockIdx.x >= total) return;
    // XCD-aware tile order: the dispatcher deals consecutive workgroup ids round-robin over the 8 XCDs (ids congruent mod 8 share
    // one), so id -> (id % 8) * ceil(total / 8) + id / 8 gives every XCD a contiguous range of logical tiles, row panel major:
    // the column tiles of a row panel run on ONE XCD (its L2 fetches the 1 MB activation panel once, not eight times), and the four
    // row panels an XCD holds walk the weight panels together.  Speed only: any placement computes the same tiles.
    const int q8 = total >> 3, r8 = total & 7, x8 = blockIdx.x & 7, k8 = blockIdx.x >> 3;
    const int pid = (x8 < r8 ? x8 * (q8 + 1) : r8 * (q8 + 1) + (x8 - r8) * q8) + k8;
    // (the K slices of a tile are neighbours: they share the tile's operand panels in one XCD's L2)
    const int ks = pid % n_split, tile = pid / n_split;
    const int mt = tile / n_nt, nt = tile - mt * n_nt;
    const int m0 = mt * BM, n0 = nt * BN;
    const int kAll = a.K / 64;
    const int kt0 = n_split > 1 ? ks * a.tiles_per_split : 0;
    const int kTiles = n_split > 1 ? min(kAll, kt0 + a.tiles_per_split) : kAll;          // end of this slice
    if (kt0 >= kTiles) return;

    // ---- DMA source addressing.  Piece p covers tile rows 8p .. 8p + 7; lane l brings the 16-byte chunk (l & 7) ^ swz(row) of
    // row l >> 3.  Wave w brings pieces w, w + NW, ... of either operand.
    const int lrow = lane >> 3, lchunk = lane & 7;
    const unsigned char* a_src[A_PER];
    const unsigned char* b_src[B_PER];
    const unsigned char* zsrc = reinterpret_cast<const unsigned char*>(a.zero16);
#pragma unroll
    for (int j = 0; j < A_PER; ++j) {
        const int p = wave + NW * j;
        const int r = p * 8 + lrow, row = m0 + r;
        const bool ok = p < A_PIECES && row < M;
        const int src = ok ? (a.row_map ? a.row_map[row] : row) : 0;
        a_src[j] = ok ? reinterpret_cast<const unsigned char*>(a.A) + (size_t)src * a.lda * 2 + ((lchunk ^ ((r >> 1) & 7)) << 4) : nullptr;
    }
#pragma unroll
    for (int j = 0; j < B_PER; ++j) {
        const int p = wave + NW * j;
        const int r = p * 8 + lrow;
        b_src[j] = p < B_PIECES ? reinterpret_cast<const unsigned char*>(a.W) + (size_t)(n0 + r) * a.K * 2 + ((lchunk ^ ((r >> 1) & 7)) << 4) : nullptr;
    }
    auto stage = [&](int buf, int kt) {
        const int kb = kt * 128;
        unsigned char* la = smem + buf * BUF;
        unsigned char* lb = la + A_BYTES;
#pragma unroll
        for (int j = 0; j < A_PER; ++j) {
            if (wave + NW * j < A_PIECES) {
                const unsigned char* p = a_src[j] ? a_src[j] + kb : zsrc;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)p,
                                                 (__attribute__((address_space(3))) void*)(la + (wave + NW * j) * 1024), 16, 0, 0);
            }
        }
#pragma unroll
        for (int j = 0; j < B_PER; ++j) {
            if (wave + NW * j < B_PIECES)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(b_src[j] + kb),
                                                 (__attribute__((address_space(3))) void*)(lb + (wave + NW * j) * 1024), 16, 0, 0);
        }
    };

    // ---- fragments: row R = wave base + 16 blk + (lane & 15), 16-byte chunk 4 s + (lane >> 4) of sub-step s, swizzled.  The
    // wave bases and the block offsets are multiples of 16 rows, so the swizzle term ((R >> 1) & 7) depends on the lane only.
    const int fr = lane & 15, fh = lane >> 4;
    const int sw = (fr >> 1) & 7;
    const int a_base = (wm * MB * 16 + fr) * 128, b_base = A_BYTES + (wn * NB * 16 + fr) * 128;
    const int ch0 = ((0 + fh) ^ sw) << 4, ch1 = ((4 + fh) ^ sw) << 4;

    f32x4 acc[NB][MB];          // [n block][m block]
#pragma unroll
    for (int i = 0; i < NB; ++i)
#pragma unroll
        for (int j = 0; j < MB; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // One K-step: the first fragment reads are issued FIRST, the DMA of the next K-step behind them (its issue -- 2-5 pieces of 100+
    // cycles each -- covers their LDS latency), then the products.
    auto kstep = [&](int buf, int kt_next, bool has_next) {
        const unsigned char* base = smem + buf * BUF;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int ch = s ? ch1 : ch0;
            bf16x8 af[MB];
#pragma unroll
            for (int j = 0; j < MB; ++j) af[j] = *reinterpret_cast<const bf16x8*>(base + a_base + j * 2048 + ch);
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                const bf16x8 wf = *reinterpret_cast<const bf16x8*>(base + b_base + i * 2048 + ch);
                if (s == 0 && i == 0) {           // the first fragments are on their way: the DMA issue covers their latency
                    __builtin_amdgcn_sched_barrier(0);
                    if (has_next) stage(buf ^ 1, kt_next);
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int j = 0; j < MB; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, af[j], acc[i][j], 0, 0, 0);
            }
        }
    };

    // two buffers: the DMA of K-step t + 1 is issued inside step t; the barrier that ends a step (with the vmcnt(0) the compiler
    // puts in front of it) completes step t + 1 for everybody.  The waves of a SIMD hide each other's waits.
    stage(0, kt0);
    __syncthreads();
    int cur = 0;
    for (int kt = kt0; kt < kTiles; ++kt) {
        kstep(cur, kt + 1, kt + 1 < kTiles);
        __syncthreads();
        cur ^= 1;
    }

    // ---- epilogue: four passes of 64 rows: fp32 tile rows -> LDS (row m, 16-byte chunk q at position q ^ (m & 15)) -> whole
    // rows out.  D[n][m]: the lane's accumulator quad is 4 consecutive columns of ONE row.  Pass p = the rows of wave row
    // (64 p) / (16 MB), its m blocks [(64 p) % (16 MB) / 16, + 4).
    constexpr int ROWB = BN * 4;                                  // bytes per staged row
    constexpr int CW = OUT_BF16 ? 8 : 4;                          // columns per store chunk (16 bytes out either way)
    constexpr int CHUNKS = 64 * (BN / CW);                        // per pass
    static_assert(64 * ROWB <= 2 * BUF, "a pass fits the operand buffers");
    unsigned char* Cb = reinterpret_cast<unsigned char*>(a.C) + (n_split > 1 ? (size_t)ks * a.slab_stride * 4 : 0);
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
        if (pass) __syncthreads();                                // the previous pass has been read out
        constexpr int RPW = 16 * MB;                              // rows per wave row
        const int wrow = (pass * 64) / RPW, j0 = ((pass * 64) % RPW) / 16;
        if (wm == wrow) {
#pragma unroll
            for (int i = 0; i < NB; ++i)
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const int m = jj * 16 + fr;
                    const int q = (wn * NB * 16 + i * 16 + 4 * fh) >> 2;
                    *reinterpret_cast<f32x4*>(smem + m * ROWB + ((q ^ (m & 15)) << 4)) = acc[i][j0 + jj];
                }
        }
        __syncthreads();
        for (int idx = tid; idx < CHUNKS; idx += NT) {
            const int r = idx / (BN / CW), c = (idx - r * (BN / CW)) * CW;
            const int row = m0 + pass * 64 + r;
            if (row >= M) continue;
            const int q0 = c >> 2;
            f32x4 v0 = *reinterpret_cast<const f32x4*>(smem + r * ROWB + ((q0 ^ (r & 15)) << 4));
            const size_t off = (size_t)row * a.ldc + n0 + c;
            if (OUT_BF16) {
                f32x4 v1 = *reinterpret_cast<const f32x4*>(smem + r * ROWB + (((q0 + 1) ^ (r & 15)) << 4));
                if (EPI == EPI_BIAS_LRELU) {
                    v0 += *reinterpret_cast<const f32x4*>(a.bias + n0 + c);
                    v1 += *reinterpret_cast<const f32x4*>(a.bias + n0 + c + 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        v0[e] = v0[e] > 0.f ? v0[e] : v0[e] * glds::SLOPE;
                        v1[e] = v1[e] > 0.f ? v1[e] : v1[e] * glds::SLOPE;
                    }
                }
                *reinterpret_cast<u32x4*>(Cb + off * 2) = u32x4{pack_bf16(v0[0], v0[1]), pack_bf16(v0[2], v0[3]), pack_bf16(v1[0], v1[1]), pack_bf16(v1[2], v1[3])};
            } else {
                if (EPI == EPI_BIAS_LRELU) {
                    v0 += *reinterpret_cast<const f32x4*>(a.bias + n0 + c);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v0[e] = v0[e] > 0.f ? v0[e] : v0[e] * glds::SLOPE;
                }
                *reinterpret_cast<f32x4*>(Cb + off * 4) = v0;
            }
        }
    }
}

}  // namespace big
}  // namespace gem
