// EXPERIMENT (round 3, not part of the library): measured with tools/gemm_glds_bench ring -- correct, but NOT faster than
// gemm_glds.h: 0.28 of the bf16 peak at 8192 rows with 4, 5 or 6 stages alike (gemm_glds.h: 0.31), 0.24 at 1536 rows.  The ring
// depth makes no difference, i.e. the products are not bound by the LATENCY of the global -> LDS transfers but by their RATE per
// CU: a 256x128x32 stage is 24 KB per 512 MFMA cycles = 47 B/clk/CU, the LDS-DMA path delivers ~16 B/clk/CU here (64-byte rows:
// half a cache line per request) against the ~29-37 B/clk/CU MI355X_MICROARCH.md measures for it at best.  More FLOP per byte
// needs 256x256 tiles, which this layer's shapes (8196 x 2560: 330 tiles on 256 CUs) quantise badly.  Kept as a record.
//
// bf16 MFMA GEMM with a deep LDS-DMA ring (gfx950): the composed front layer of the "bf16 VAE decoder" mode and its transpose,
//
//   C[M,N] = epi( A[M,K] . W[N][K]^T + bias[N] )        (decoder_input o conv 0: SeqConvVAE.py:62,67-75,131-135; backward-data)
//
// Why another kernel next to gemm_glds.h.  With bf16 operands a 128x128x64 K-step is only 512 MFMA cycles per wave, while a
// global -> LDS transfer takes ~1.3 us (~3000 cycles) to come back once every CU is pulling: the two-buffer loop of gemm_glds.h
// (one K-step in flight per workgroup, two workgroups per CU = 64 KB in flight) spends its time waiting for that round trip --
// measured 1.3 us per K-step whatever the row count, 0.31 of the bf16 peak at 8192 rows, 0.22 at 1536 (tools/gemm_glds_bench).
// What hides a latency is bytes in flight, so this kernel turns almost the whole LDS of a CU into ONE ring:
//
//   * tile 256 x 128, K-steps of 32 (64-byte rows): a stage is 24 KB; SIX stages = 144 KB, one workgroup (8 waves = 2 per SIMD)
//     per CU; up to five stages (120 KB) are in flight while one is being read
//   * operands go global -> LDS by global_load_lds_dwordx4 (3 one-KB pieces per wave and stage), one raw s_barrier per K-step,
//     a COUNTED s_waitcnt vmcnt that leaves the four youngest stages in flight; every step issues the same number of pieces
//     (past the end of K the last step is fetched again into a free slot), so the count is exact
//   * the fragment reads of step k are issued right behind the barrier, the MFMAs of step k-1 run behind them (operands
//     double-buffered in registers): LDS latency hides behind the matrix pipe inside ONE wave, which matters because the eight
//     waves of the only workgroup on the CU march in step
//   * 64-byte rows: the 16-byte chunk c of row r sits at chunk position c ^ F[(r >> 2) & 3], F = {0,3,2,1} -- with it each of
//     ds_read_b128's four 16-lane groups touches 16 distinct 16-byte slots (the swizzle is applied to the DMA's per-lane
//     SOURCE address, the LDS side of a DMA is lane-linear, and to the fragment read address)
//   * wave (wm, wn) of 4 x 2 owns 64 x 64 of the tile = 16 independent 16x16 accumulators (v_mfma_f32_16x16x32_bf16, fp32
//     accumulate); the weight fragment is MFMA operand A, so a lane holds 4 consecutive output columns of one row; the epilogue
//     (bias / LeakyReLU, bf16 or fp32 / split-K slabs) stages the tile through the ring's memory and leaves as whole rows
//   * workgroup -> tile order as in gemm_glds.h (XCD-aware, 8 row tiles x all column tiles share their panels in one L2)
#pragma once
#include "../globalegomocap_amd/csrc/gemm_glds.h"

namespace gem {
namespace ring {

using glds::Args;
using glds::bf16x8;
using glds::f32x4;
using glds::u32x4;
using glds::pack_bf16;

constexpr int BM = 256, BN = 128, BK = 32;
constexpr int ROWB = BK * 2;                       // bytes per operand row and stage
constexpr int A_BYTES = BM * ROWB, B_BYTES = BN * ROWB, STAGE = A_BYTES + B_BYTES;
constexpr int NW = 8, NT = NW * 64;
constexpr int A_PIECES = A_BYTES / 1024 / NW, B_PIECES = B_BYTES / 1024 / NW, PIECES = A_PIECES + B_PIECES;      // 2 + 1 per wave
static_assert(A_PIECES * 1024 * NW == A_BYTES && B_PIECES * 1024 * NW == B_BYTES, "whole 1 KB pieces per wave");

__device__ __forceinline__ int swz4(int row) { return (0x1230 >> (((row >> 2) & 3) * 4)) & 3; }      // F = {0, 3, 2, 1}

// ABL (ablation, timing only -- the results are wrong): 1 = no global -> LDS transfers inside the loop, 2 = no LDS fragment reads
// inside the loop, 3 = no MFMAs (the fragments are consumed by an empty asm), 4 = transfers only (no reads, no MFMAs),
// 5 = MFMAs only (no transfers, no reads, no barrier), 6 = reads + MFMAs without transfers and without the barrier
template <int EPI, bool OUT_BF16, int S = 6, int ABL = 0>
__global__ __launch_bounds__(NT) void gemm_ring_kernel(const Args a) {
    static_assert(S >= 3 && S * STAGE <= 160 * 1024, "ring must fit the LDS of a CU");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int M = a.m_dev ? *a.m_dev : a.M;
    if (M <= 0) return;
    const int n_mt = (M + BM - 1) / BM, n_nt = a.N / BN;
    int mt, nt, ks;
    if (!glds::tile_of_block(blockIdx.x, n_mt, n_nt, a.n_split, mt, nt, ks)) return;
    const int m0 = mt * BM, n0 = nt * BN;
    const int kTiles = a.K / BK;
    const int kt_begin = ks * a.tiles_per_split, kt_end = min(kTiles, kt_begin + a.tiles_per_split);
    const int n_steps = kt_end - kt_begin;
    if (n_steps <= 0) return;

    // ---- DMA source addressing: piece p covers tile rows 16p .. 16p+15; lane l brings LDS chunk (l & 3) of row (l >> 2),
    // i.e. source chunk (l & 3) ^ F(row)
    const unsigned char* a_ptr[A_PIECES];
    int a_step[A_PIECES];
    const unsigned char* zsrc = reinterpret_cast<const unsigned char*>(a.zero16);
#pragma unroll
    for (int j = 0; j < A_PIECES; ++j) {
        const int r = (wave * A_PIECES + j) * 16 + (lane >> 2);
        const int row = m0 + r;
        const bool ok = row < M;
        int src = row;
        if (a.row_map) src = a.row_map[ok ? row : 0];
        a_ptr[j] = ok ? reinterpret_cast<const unsigned char*>(a.A) + (size_t)src * a.lda * 2 + (((lane & 3) ^ swz4(r)) << 4) : zsrc;
        a_step[j] = ok ? ROWB : 0;          // rows past M keep reading the zero line
    }
    const unsigned char* b_ptr[B_PIECES];
#pragma unroll
    for (int j = 0; j < B_PIECES; ++j) {
        const int r = (wave * B_PIECES + j) * 16 + (lane >> 2);
        b_ptr[j] = reinterpret_cast<const unsigned char*>(a.W) + (size_t)(n0 + r) * a.K * 2 + (((lane & 3) ^ swz4(r)) << 4);
    }
    auto stage = [&](int slot, int kt) {
        unsigned char* la = smem + slot * STAGE + wave * (A_PIECES * 1024);
        unsigned char* lb = smem + slot * STAGE + A_BYTES + wave * (B_PIECES * 1024);
#pragma unroll
        for (int j = 0; j < A_PIECES; ++j)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a_ptr[j] + (size_t)kt * a_step[j]),
                                             (__attribute__((address_space(3))) void*)(la + j * 1024), 16, 0, 0);
#pragma unroll
        for (int j = 0; j < B_PIECES; ++j)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(b_ptr[j] + (size_t)kt * ROWB),
                                             (__attribute__((address_space(3))) void*)(lb + j * 1024), 16, 0, 0);
    };

    // ---- fragment read addresses (bytes inside a stage): activation rows wm*64 + 16i + r, weight rows wn*64 + 16i + r; the
    // swizzle term depends on r = lane & 15 only (16i is a multiple of 16)
    const int fr = lane & 15, fq = lane >> 4;
    const int fchunk = (fq ^ swz4(fr)) << 4;
    const int a_off = (wm * 64 + fr) * ROWB + fchunk;
    const int w_off = A_BYTES + (wn * 64 + fr) * ROWB + fchunk;

    f32x4 acc[4][4];      // [n block][m block]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    auto read_frags = [&](int slot, bf16x8 (&af)[4], bf16x8 (&wf)[4]) {
        const unsigned char* base = smem + slot * STAGE;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            wf[q] = *reinterpret_cast<const bf16x8*>(base + w_off + q * 16 * ROWB);
            af[q] = *reinterpret_cast<const bf16x8*>(base + a_off + q * 16 * ROWB);
        }
    };
    auto mfmas = [&](const bf16x8 (&af)[4], const bf16x8 (&wf)[4]) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], af[j], acc[i][j], 0, 0, 0);
    };

    // ---- the ring.  Step k lives in slot k % S.  Prologue: steps 0 .. S-2 requested.  Iteration k: my pieces of step k have
    // landed once at most (S-2) younger steps are outstanding; the barrier extends that to everybody's pieces and says that
    // everybody has read step k-1, whose slot the request for step k+S-1 overwrites next.
#pragma unroll
    for (int s = 0; s < S - 1; ++s) stage(s, kt_begin + min(s, n_steps - 1));
    bf16x8 afX[4], wfX[4], afY[4], wfY[4];
    if (ABL == 2 || ABL == 4 || ABL == 5) {          // (fragments read once, outside the loop)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        read_frags(0, afX, wfX);
        read_frags(0, afY, wfY);
    }
    int slot = 0, slot_in = S - 1;
    auto advance = [&](int k, bf16x8 (&af_new)[4], bf16x8 (&wf_new)[4], const bf16x8 (&af_old)[4], const bf16x8 (&wf_old)[4]) {
        // (lgkmcnt(0): this wave's fragment reads of step k-1 have returned before anybody may overwrite their slot)
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(PIECES * (S - 2)) : "memory");
        if (ABL != 5 && ABL != 6) __builtin_amdgcn_s_barrier();
        if (ABL != 1 && ABL != 5 && ABL != 6) stage(slot_in, kt_begin + min(k + S - 1, n_steps - 1));
        if (ABL != 2 && ABL != 4 && ABL != 5) read_frags(slot, af_new, wf_new);
        __builtin_amdgcn_sched_barrier(0);
        if (ABL == 3) {
#pragma unroll
            for (int q = 0; q < 4; ++q) asm volatile("" ::"v"(af_new[q]), "v"(wf_new[q]));
        } else if (ABL != 4) {
            if (k > 0) mfmas(af_old, wf_old);
        }
        slot = slot + 1 == S ? 0 : slot + 1;
        slot_in = slot_in + 1 == S ? 0 : slot_in + 1;
    };
    int k = 0;
    for (; k + 1 < n_steps; k += 2) {
        advance(k, afX, wfX, afY, wfY);
        advance(k + 1, afY, wfY, afX, wfX);
    }
    if (k < n_steps) {            // odd number of steps: the last one lands in X
        advance(k, afX, wfX, afY, wfY);
        mfmas(afX, wfX);
    } else {
        mfmas(afY, wfY);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();              // every request has landed, every wave has left the loop: the ring becomes the epilogue's staging area

    // ---- epilogue: fp32 tile -> LDS (row m, 16-byte chunk q of its BN columns at position q ^ (m & 31)) -> whole rows out,
    // 128 rows per pass.  D[n][m]: the lane's column is its row m of C, its registers are 4 consecutive n.
    constexpr int ROWC = BN * 4, CH = BN / 4, PR = 128;
    static_assert(PR * ROWC <= S * STAGE, "a pass of the staged fp32 tile must fit the ring");
    const bool split = a.n_split > 1;
    constexpr int TPR = BN / 8, RPP = NT / TPR;          // 16 threads per row (8 columns each), 32 rows per sweep
    const int c8 = (tid % TPR) * 8, r0 = tid / TPR;
    f32x4 bv0 = {0.f, 0.f, 0.f, 0.f}, bv1 = {0.f, 0.f, 0.f, 0.f};
    if ((EPI == glds::EPI_BIAS || EPI == glds::EPI_BIAS_LRELU) && !split && a.bias) {
        bv0 = *reinterpret_cast<const f32x4*>(a.bias + n0 + c8);
        bv1 = *reinterpret_cast<const f32x4*>(a.bias + n0 + c8 + 4);
    }
    unsigned char* Cb = reinterpret_cast<unsigned char*>(a.C);
    if (split) Cb += (size_t)ks * a.slab_stride * 4;
#pragma unroll
    for (int pass = 0; pass < BM / PR; ++pass) {
        if (pass) __syncthreads();
        if (wm * 64 / PR == pass) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int m = (wm * 64) % PR + j * 16 + fr;
                    const int q = (wn * 64 + i * 16 + 4 * fq) >> 2;
                    *reinterpret_cast<f32x4*>(smem + m * ROWC + ((q ^ (m & (CH - 1))) << 4)) = acc[i][j];
                }
        }
        __syncthreads();
#pragma unroll
        for (int p = 0; p < PR / RPP; ++p) {
            const int r = p * RPP + r0, row = m0 + pass * PR + r;
            if (row >= M) continue;
            const int q0 = c8 >> 2;
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(smem + r * ROWC + ((q0 ^ (r & (CH - 1))) << 4));
            const f32x4 v1 = *reinterpret_cast<const f32x4*>(smem + r * ROWC + (((q0 + 1) ^ (r & (CH - 1))) << 4));
            float v[8] = {v0[0] + bv0[0], v0[1] + bv0[1], v0[2] + bv0[2], v0[3] + bv0[3], v1[0] + bv1[0], v1[1] + bv1[1], v1[2] + bv1[2], v1[3] + bv1[3]};
            const size_t off = (size_t)row * a.ldc + n0 + c8;
            if (!split && EPI == glds::EPI_BIAS_LRELU) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = v[e] > 0.f ? v[e] : v[e] * glds::SLOPE;
            }
            if (OUT_BF16 && !split) {
                *reinterpret_cast<u32x4*>(Cb + off * 2) = u32x4{pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3]), pack_bf16(v[4], v[5]), pack_bf16(v[6], v[7])};
            } else {
                *reinterpret_cast<f32x4*>(Cb + off * 4) = f32x4{v[0], v[1], v[2], v[3]};
                *reinterpret_cast<f32x4*>(Cb + off * 4 + 16) = f32x4{v[4], v[5], v[6], v[7]};
            }
        }
    }
}

}  // namespace ring
}  // namespace gem
