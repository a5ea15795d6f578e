// MFMA GEMM / temporal-conv kernel fed by LDS-DMA (gfx950): the engine of the "bf16 VAE decoder" mode (BASELINE
// configs[2..4], bf16 operands) and -- same structure, fp32 operands -- of the fp32 decoder_input products.
// Same contract as gemm_f32_kernel,
//
//   C[M,N] = epi( sum_{tap<TAPS} shift_{tap-1}(A)[M,K] . W[tap][N][K]^T + bias[N] )      (SeqConvVAE.py:36,62-92,131-140)
//
// IN_F32 = false: both operands bf16 in HBM (activations are written as bf16 by the producing kernel's epilogue, weights
// converted once at load time), fp32 accumulation (v_mfma_f32_16x16x32_bf16 / 32x32x16), output bf16 (activations /
// gradients for the next layer) or fp32 (split-K slabs, the latent gradient, the decoded pose).
// IN_F32 = true: both operands fp32 (v_mfma_f32_16x16x4_f32 / 32x32x2: exact fp32 FMA chains), fp32 output.
//
// Tile: BM x BN x (128 bytes of K: 64 bf16 or 32 fp32), one wave per 64x64 of the output (4x4 blocks of 16x16 or 2x2 of
// 32x32: independent accumulators, one wave keeps its SIMD's matrix pipe busy), 4 waves for 128x128.  Both operand
// tiles go global -> LDS with global_load_lds_dwordx4 (no VGPR round trip, no convert in the staging path),
// double-buffered: the DMA of K-step t+1 is issued before the MFMAs of step t.  LDS rows are 128 bytes; the 16-byte
// chunk c of row r is stored at chunk position c ^ ((r >> 1) & 7), which makes the ds_read_b128 fragment reads of
// both MFMA operand layouts conflict-free (the swizzle is applied to the per-lane SOURCE address of the DMA -- its
// LDS side is lane-linear -- and to the fragment read address).  A lane's 16 bytes of a fragment feed several MFMA
// k-steps; the pairing of k values inside a 128-byte row is a permutation that a sum over k does not care about.
// The MFMA takes the WEIGHT fragment as its A operand and the activation fragment as B: D[n][m], so a lane ends up
// with 4 consecutive output columns n of ONE row m per register quad; the epilogue stages the fp32 tile through LDS
// (16-byte chunks, XOR-swizzled) and leaves as whole rows: bias / LeakyReLU / LeakyReLU'-mask are applied there on
// 8 consecutive columns per thread, stores are 16 bytes per lane, 256 contiguous bytes per row.
// Workgroup -> tile mapping is XCD-aware: the dispatcher deals consecutive workgroup ids round-robin over the 8 XCDs,
// so id -> (id % 8) * ceil(n/8) + id / 8 gives every XCD a contiguous range of logical tiles, walked in groups of 8
// row tiles x all column tiles: the ~64 tiles an XCD has in flight share 8 activation panels and 8 weight panels in
// its L2.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gem {

namespace glds {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

enum { EPI_BIAS = 0, EPI_BIAS_LRELU = 1, EPI_MASK = 2, EPI_NONE = 3 };
constexpr float SLOPE = 0.01f;

__device__ __forceinline__ unsigned int pack_bf16(float lo, float hi) {      // round-to-nearest-even (v_cvt_pk_bf16_f32)
    typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
    typedef float f2 __attribute__((ext_vector_type(2)));
    const bf2 r = __builtin_convertvector(f2{lo, hi}, bf2);
    return __builtin_bit_cast(unsigned int, r);
}
__device__ __forceinline__ float bf_lo(unsigned int u) { return __builtin_bit_cast(float, u << 16); }
__device__ __forceinline__ float bf_hi(unsigned int u) { return __builtin_bit_cast(float, u & 0xFFFF0000u); }

struct Args {
    const void* A;            // [rows, lda] bf16 or fp32
    const void* W;            // [TAPS][N][K] bf16 or fp32, k contiguous
    const float* bias;        // [N] or nullptr
    const void* aux;          // EPI_MASK: activation (same type as A) whose sign selects LeakyReLU' (same [M, ldc] layout as the output)
    void* C;                  // bf16 or fp32 [M, ldc]; split-K: fp32 slabs, slab z at C + z * slab_stride
    const void* zero16;       // >= 16 zero bytes in HBM: source of rows that lie outside the window / past M
    const int* m_dev;         // device row count (evaluation rounds) or nullptr
    const int* row_map;       // gathered A rows (TAPS == 1 only) or nullptr
    int lda, ldc, M, N, K, T;
    int n_split, tiles_per_split;     // split-K over the TAPS*K/BK k-tiles (1: none)
    size_t slab_stride;
    int m_max;                // > 0: return at once when the (device) row count is >= m_max (gemm_big.h takes those launches)
};

// logical tile id of this workgroup (XCD-aware, grouped): returns false when the workgroup has nothing to do
__device__ __forceinline__ bool tile_of_block(int id, int n_mt, int n_nt, int n_split, int& mt, int& nt, int& ks) {
    const int per = n_mt * n_nt, total = per * n_split;
    if (id >= total) return false;
    // XCD remap (bijective for any total): ids congruent mod 8 run on one XCD
    const int q = total >> 3, r = total & 7, x = id & 7, k = id >> 3;
    const int pid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + k;
    ks = pid / per;
    const int p = pid - ks * per;
    constexpr int GM = 8;
    const int width = GM * n_nt, g = p / width, first = g * GM;
    const int gsize = min(n_mt - first, GM);
    const int in_g = p - g * width;
    mt = first + in_g % gsize;
    nt = in_g / gsize;
    return true;
}

template <bool IN_F32, int TAPS, int EPI, int BM, int BN, bool OUT_BF16, int MF = 16, int STAGES = 2>
__global__ __launch_bounds__((BM / 64) * (BN / 64) * 64) void gemm_glds_kernel(const Args a) {
    static_assert(STAGES == 2 || STAGES == 3, "double or triple buffering");
    static_assert(!(IN_F32 && OUT_BF16), "fp32 operands write fp32");
    constexpr int ES = IN_F32 ? 4 : 2;           // operand element size
    constexpr int BK = 128 / ES;                 // one K-step = 128 bytes of every row
    constexpr int WAVES_M = BM / 64, WAVES_N = BN / 64, NW = WAVES_M * WAVES_N, NT = NW * 64;
    constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, BUF = A_BYTES + B_BYTES;
    constexpr int A_INSTR = A_BYTES / 1024 / NW, B_INSTR = B_BYTES / 1024 / NW;      // 1 KB DMA pieces per wave
    static_assert(A_BYTES % (1024 * NW) == 0 && B_BYTES % (1024 * NW) == 0, "tile must split into whole DMA pieces per wave");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int M = a.m_dev ? *a.m_dev : a.M;
    if (M <= 0 || (a.m_max > 0 && M >= a.m_max)) return;
    const int n_mt = (M + BM - 1) / BM, n_nt = a.N / BN;
    int mt, nt, ks;
    if (!tile_of_block(blockIdx.x, n_mt, n_nt, a.n_split, mt, nt, ks)) return;
    const int m0 = mt * BM, n0 = nt * BN;
    const int kTiles = a.K / BK, nTiles = TAPS * kTiles;
    const int kt_begin = ks * a.tiles_per_split, kt_end = min(nTiles, kt_begin + a.tiles_per_split);

    // ---- DMA source addressing.  Piece j of this wave covers tile rows 8*(wave*INSTR + j) .. +7; lane l brings the 16-byte
    // chunk (l & 7) ^ swz(row) of row (l >> 3).
    const int lrow = lane >> 3, lchunk = lane & 7;
    const unsigned char* a_src[A_INSTR];      // row base (tap 1), nullptr-free: invalid rows point at the zero line
    int a_t[A_INSTR];                         // frame index of the row inside its window (TAPS == 3), -1: row not valid at all
    int a_sw[A_INSTR];
#pragma unroll
    for (int j = 0; j < A_INSTR; ++j) {
        const int r = (wave * A_INSTR + j) * 8 + lrow;
        const int row = m0 + r;
        const bool ok = row < M;
        int src = row;
        if (TAPS == 1 && a.row_map) src = a.row_map[ok ? row : 0];
        a_src[j] = reinterpret_cast<const unsigned char*>(a.A) + (size_t)src * a.lda * ES;
        a_t[j] = ok ? (TAPS == 3 ? row % a.T : 0) : -1;
        a_sw[j] = (lchunk ^ ((r >> 1) & 7)) * 16;
    }
    const unsigned char* b_src[B_INSTR];
#pragma unroll
    for (int j = 0; j < B_INSTR; ++j) {
        const int r = (wave * B_INSTR + j) * 8 + lrow;
        b_src[j] = reinterpret_cast<const unsigned char*>(a.W) + ((size_t)(n0 + r) * a.K) * ES + (lchunk ^ ((r >> 1) & 7)) * 16;
    }
    const unsigned char* zsrc = reinterpret_cast<const unsigned char*>(a.zero16);
    const size_t tap_stride = (size_t)a.N * a.K * ES;

    auto stage = [&](int buf, int kt) {
        const int tap = (TAPS == 3) ? (kt >= kTiles) + (kt >= 2 * kTiles) : 0;
        const int kb = (kt - tap * kTiles) * 128;                 // byte offset inside the row
        unsigned char* la = smem + buf * BUF + wave * (A_INSTR * 1024);
        unsigned char* lb = smem + buf * BUF + A_BYTES + wave * (B_INSTR * 1024);
#pragma unroll
        for (int j = 0; j < A_INSTR; ++j) {
            bool ok = a_t[j] >= 0;
            if (TAPS == 3) { const int tt = a_t[j] + tap - 1; ok = ok && tt >= 0 && tt < a.T; }
            const unsigned char* p = ok ? a_src[j] + (ptrdiff_t)((TAPS == 3) ? (tap - 1) : 0) * a.lda * ES + kb + a_sw[j] : zsrc;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)p,
                                             (__attribute__((address_space(3))) void*)(la + j * 1024), 16, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < B_INSTR; ++j) {
            const unsigned char* p = b_src[j] + tap * tap_stride + kb;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)p,
                                             (__attribute__((address_space(3))) void*)(lb + j * 1024), 16, 0, 0);
        }
    };

    // ---- main loop.  MF = 32: 2x2 blocks of 32x32 MFMAs per wave; MF = 16: 4x4 blocks of 16x16 (same LDS bytes and MFMA
    // cycles per K-step; for bf16 the chip holds a higher clock on the 16x16 shape).  A lane reads 16 bytes of a row per
    // fragment: 8 bf16 = the k values of ONE bf16 MFMA, or 4 fp32 = one k value each for FOUR fp32 MFMAs.
    constexpr int NB = 64 / MF;                  // blocks per wave and dimension
    constexpr int LG = 64 / MF;                  // lane groups along k (2 for 32x32, 4 for 16x16): group g reads chunk LG*s + g
    constexpr int NSUB = 8 / LG;                 // fragment reads per K-step and block
    typedef float accv __attribute__((ext_vector_type(MF == 32 ? 16 : 4)));
    accv acc[NB][NB];      // [n block][m block]
#pragma unroll
    for (int i = 0; i < NB; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int e = 0; e < (MF == 32 ? 16 : 4); ++e) acc[i][j][e] = 0.f;

    // fragment read addresses: row R = wave base + MF*blk + (lane % MF), 16-byte chunk LG*s + lane / MF of sub-step s
    const int fr = lane % MF, fh = lane / MF;
    int a_off[NB], b_off[NB], a_x[NB], b_x[NB];
#pragma unroll
    for (int q = 0; q < NB; ++q) {
        const int ra = wm * 64 + q * MF + fr, rb = wn * 64 + q * MF + fr;
        a_off[q] = ra * 128; a_x[q] = (ra >> 1) & 7;
        b_off[q] = A_BYTES + rb * 128; b_x[q] = (rb >> 1) & 7;
    }

    auto compute = [&](int buf) {
        const unsigned char* base = smem + buf * BUF;
#pragma unroll
        for (int s = 0; s < NSUB; ++s) {
            if constexpr (IN_F32) {
                f32x4 af[NB], wf[NB];
#pragma unroll
                for (int q = 0; q < NB; ++q) {
                    af[q] = *reinterpret_cast<const f32x4*>(base + a_off[q] + (((LG * s + fh) ^ a_x[q]) << 4));
                    wf[q] = *reinterpret_cast<const f32x4*>(base + b_off[q] + (((LG * s + fh) ^ b_x[q]) << 4));
                }
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int i = 0; i < NB; ++i)
#pragma unroll
                        for (int j = 0; j < NB; ++j) {
                            if constexpr (MF == 32) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[i][e], af[j][e], acc[i][j], 0, 0, 0);
                            else acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[i][e], af[j][e], acc[i][j], 0, 0, 0);
                        }
            } else {
                bf16x8 af[NB], wf[NB];
#pragma unroll
                for (int q = 0; q < NB; ++q) {
                    af[q] = *reinterpret_cast<const bf16x8*>(base + a_off[q] + (((LG * s + fh) ^ a_x[q]) << 4));
                    wf[q] = *reinterpret_cast<const bf16x8*>(base + b_off[q] + (((LG * s + fh) ^ b_x[q]) << 4));
                }
#pragma unroll
                for (int i = 0; i < NB; ++i)
#pragma unroll
                    for (int j = 0; j < NB; ++j) {
                        if constexpr (MF == 32) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[i], af[j], acc[i][j], 0, 0, 0);
                        else acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], af[j], acc[i][j], 0, 0, 0);
                    }
            }
        }
    };

    if constexpr (STAGES == 2) {
        // two buffers: the DMA of tile t+1 is issued before the MFMAs of tile t; the barrier that ends a K-step (with the
        // vmcnt(0) the compiler puts in front of it) completes tile t+1 for everybody
        stage(0, kt_begin);
        __syncthreads();
        int cur = 0;
        for (int kt = kt_begin; kt < kt_end; ++kt) {
            if (kt + 1 < kt_end) stage(cur ^ 1, kt + 1);
            compute(cur);
            __syncthreads();
            cur ^= 1;
        }
    } else {
        // three buffers, tiles two K-steps ahead: a COUNTED s_waitcnt leaves the DMA of tile t+1 in flight across the barrier
        // (one raw s_barrier per K-step, no fence: __syncthreads would drain the DMA queue).  A wave's DMA pieces retire in issue
        // order, so vmcnt(pieces of one tile) = "my part of tile t has landed"; the barrier extends that to every wave's part
        // and also says that everybody has finished reading tile t-1, whose buffer the DMA of tile t+2 overwrites next.
        constexpr int PIECES = A_INSTR + B_INSTR;
        stage(0, kt_begin);
        if (kt_begin + 1 < kt_end) stage(1, kt_begin + 1);
        int cur = 0;
        for (int kt = kt_begin; kt < kt_end; ++kt) {
            if (kt + 1 < kt_end) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (kt + 2 < kt_end) stage(cur >= 1 ? cur - 1 : 2, kt + 2);          // buffer (cur + 2) % 3
            compute(cur);
            cur = cur == 2 ? 0 : cur + 1;
        }
        __syncthreads();          // every wave has left the main loop: the operand buffers become the epilogue's staging area
    }

    // ---- epilogue: fp32 tile -> LDS (row m, 16-byte chunk q of its BN columns at position q ^ (m & (CH-1))) -> whole rows out
    // D[n][m]: the lane's column is its row m of C, its registers are runs of 4 consecutive n.  Tiles taller than 128 rows
    // are staged in passes of 128 rows (the staged fp32 rows have to fit the two operand buffers).
    constexpr int ROWB = BN * 4;                 // bytes per staged row
    constexpr int CH = BN / 4;                   // 16-byte chunks per row (32 for BN = 128, 16 for BN = 64)
    constexpr int PR = BM < 128 ? BM : 128;      // rows per pass
    static_assert(PR * ROWB <= STAGES * BUF, "a pass of the staged fp32 tile must fit the operand buffers");
    const bool split = a.n_split > 1;
    constexpr int TPR = BN / 8;                  // threads per row (8 columns each)
    constexpr int RPP = NT / TPR;                // rows per store sweep
    const int c8 = (tid % TPR) * 8, r0 = tid / TPR;
    f32x4 bv0 = {0.f, 0.f, 0.f, 0.f}, bv1 = {0.f, 0.f, 0.f, 0.f};
    if ((EPI == EPI_BIAS || EPI == EPI_BIAS_LRELU) && !split && a.bias) {
        bv0 = *reinterpret_cast<const f32x4*>(a.bias + n0 + c8);
        bv1 = *reinterpret_cast<const f32x4*>(a.bias + n0 + c8 + 4);
    }
    unsigned char* Cb = reinterpret_cast<unsigned char*>(a.C);
    if (split) Cb += (size_t)ks * a.slab_stride * 4;
#pragma unroll
    for (int pass = 0; pass < BM / PR; ++pass) {
        if (pass) __syncthreads();               // the previous pass has been read out
        if (wm * 64 / PR == pass) {
#pragma unroll
            for (int i = 0; i < NB; ++i)
#pragma unroll
                for (int j = 0; j < NB; ++j) {
                    const int m = (wm * 64) % PR + j * MF + fr;
#pragma unroll
                    for (int g = 0; g < (MF == 32 ? 4 : 1); ++g) {
                        const int q = (wn * 64 + i * MF + (MF == 32 ? 8 * g + 4 * fh : 4 * fh)) >> 2;
                        const f32x4 v = {acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
                        *reinterpret_cast<f32x4*>(smem + m * ROWB + ((q ^ (m & (CH - 1))) << 4)) = v;
                    }
                }
        }
        __syncthreads();
#pragma unroll 4
        for (int p = 0; p < PR / RPP; ++p) {
            const int r = p * RPP + r0, row = m0 + pass * PR + r;
            if (row >= M) continue;
            const int q0 = c8 >> 2;
            f32x4 v0 = *reinterpret_cast<const f32x4*>(smem + r * ROWB + ((q0 ^ (r & (CH - 1))) << 4));
            f32x4 v1 = *reinterpret_cast<const f32x4*>(smem + r * ROWB + (((q0 + 1) ^ (r & (CH - 1))) << 4));
            float v[8] = {v0[0] + bv0[0], v0[1] + bv0[1], v0[2] + bv0[2], v0[3] + bv0[3], v1[0] + bv1[0], v1[1] + bv1[1], v1[2] + bv1[2], v1[3] + bv1[3]};
            const size_t off = (size_t)row * a.ldc + n0 + c8;
            if (!split) {
                if (EPI == EPI_BIAS_LRELU) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = v[e] > 0.f ? v[e] : v[e] * SLOPE;
                }
                if (EPI == EPI_MASK) {
                    if constexpr (IN_F32) {
                        const f32x4 ma = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(a.aux) + off);
                        const f32x4 mb = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(a.aux) + off + 4);
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            v[e] *= ma[e] > 0.f ? 1.f : SLOPE;
                            v[4 + e] *= mb[e] > 0.f ? 1.f : SLOPE;
                        }
                    } else {
                        const u32x4 m4 = *reinterpret_cast<const u32x4*>(reinterpret_cast<const uint16_t*>(a.aux) + off);
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            v[2 * e] *= bf_lo(m4[e]) > 0.f ? 1.f : SLOPE;
                            v[2 * e + 1] *= bf_hi(m4[e]) > 0.f ? 1.f : SLOPE;
                        }
                    }
                }
            }
            if (OUT_BF16 && !split) {
                const u32x4 o = {pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3]), pack_bf16(v[4], v[5]), pack_bf16(v[6], v[7])};
                *reinterpret_cast<u32x4*>(Cb + off * 2) = o;
            } else {
                *reinterpret_cast<f32x4*>(Cb + off * 4) = f32x4{v[0], v[1], v[2], v[3]};
                *reinterpret_cast<f32x4*>(Cb + off * 4 + 16) = f32x4{v[4], v[5], v[6], v[7]};
            }
        }
    }
}

}  // namespace glds
}  // namespace gem
