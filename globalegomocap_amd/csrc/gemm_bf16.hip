// bf16-input MFMA GEMM / temporal-conv kernel with fp32 accumulation (gfx950), same contract as
// gemm_f32_kernel:   C[M,N] = epi( sum_tap shift_tap(A)[M,K] . W[tap][N][K]^T + bias ).
//
// NPROD = 1: operands rounded to bf16 ("bf16 VAE decoder", BASELINE configs[2..3]).
// NPROD = 3: every fp32 operand x is split into hi = bf16(x) and lo = bf16(x - hi) and the product is
//            a_hi*b_hi + a_hi*b_lo + a_lo*b_hi in fp32 accumulators (the dropped lo*lo term is 2^-16 relative):
//            fp32-grade results at 16/3 of the fp32 MFMA rate.
// Activations stay fp32 in HBM; the A tile is split while it is staged into LDS.  Weights are split once at
// load time (Layer::wb_hi / wb_lo).  v_mfma_f32_32x32x16_bf16: lane (r = lane&31, h = lane>>5) holds
// A[row r][k = 8h..8h+7] and B[k = 8h..8h+7][col r] of a 16-deep step, so both LDS images are [rows][BK]
// bf16 with k contiguous and one 16-byte ds_read per operand per step.
#include "gem_internal.h"

namespace gem {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef int i32x8 __attribute__((ext_vector_type(8)));        // per-thread row tables as native vectors (a plain int[] went to scratch)

__device__ __forceinline__ unsigned int f2bf(float x) {           // round-to-nearest-even, finite inputs
    const unsigned int u = __builtin_bit_cast(unsigned int, x);
    return (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
}
__device__ __forceinline__ float bf2f(unsigned int b) { return __builtin_bit_cast(float, b << 16); }

template <int TAPS, int EPI, int NPROD>
__global__ __launch_bounds__(256) void gemm_bf16_kernel(const float* __restrict__ A, int lda, const uint16_t* __restrict__ Whi,
                                                        const uint16_t* __restrict__ Wlo, const float* __restrict__ bias,
                                                        const float* __restrict__ aux, float* __restrict__ C, int ldc, int M, int N,
                                                        int K, int T, int tiles_per_slice, size_t slab_stride,
                                                        const int* __restrict__ m_dev, const int* __restrict__ row_map, int dyn_W) {
    constexpr int BM = 64, BN = 64, BK = 64;
    constexpr int LD = BK + 8;                       // bf16 elements per LDS row (16-byte pad)
    constexpr int IMG = BM * LD;                     // one [64][LD] bf16 image
    constexpr bool SPLIT = NPROD == 3;
    extern __shared__ __attribute__((aligned(16))) unsigned short ldsb[];
    unsigned short* a_hi = ldsb;
    unsigned short* b_hi = ldsb + IMG;
    unsigned short* a_lo = ldsb + 2 * IMG;           // only touched when SPLIT
    unsigned short* b_lo = ldsb + 3 * IMG;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    if (m_dev) M = *m_dev;
    const int kTiles = K / BK;
    bool split_k = gridDim.z > 1;
    int kt_begin = blockIdx.z * tiles_per_slice;
    int kt_end = min(TAPS * kTiles, kt_begin + tiles_per_slice);
    if (dyn_W > 0) {                      // evaluation rounds: slices re-cut for the active rows (dyn_split)
        if (M <= 0) return;
        const int CT = N / BN;
        const DynSplit d = dyn_split(M, BM, CT, dyn_W, TAPS * kTiles, ldc);
        const int id = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
        if (id >= CT * d.RT * d.SK) return;
        const int ks = id / (CT * d.RT);
        n0 = (id % CT) * BN;
        m0 = ((id / CT) % d.RT) * BM;
        kt_begin = ks * d.per;
        kt_end = min(TAPS * kTiles, kt_begin + d.per);
        split_k = true;
        C += (size_t)ks * d.slab;
    } else {
        if (m0 >= M) return;
        if (split_k) C += (size_t)blockIdx.z * slab_stride;
    }

    // A tile: 64 rows x 64 fp32 = 1024 float4 -> 4 per thread (16 threads per row, 16 rows per pass)
    // B tile: 64 rows x 64 bf16 = 512 x 16 B   -> 2 per thread (8 threads per row, 32 rows per pass)
    const int a_c4 = (tid & 15) * 4, a_r = tid >> 4;
    const int b_c8 = (tid & 7) * 8, b_r = tid >> 3;
    i32x8 a_row, a_t, a_src;
    bool a_ok[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r_ = m0 + a_r + 16 * i;
        a_row[i] = r_;
        a_t[i] = (TAPS == 3) ? (r_ % T) : 0;
        int s_ = r_;
        if (TAPS == 1 && row_map) s_ = row_map[r_ < M ? r_ : 0];
        a_src[i] = s_;
    }
    f32x4 ra[4];
    u32x4 rbh[2], rbl[2];

#define GB_LOAD(kt_)                                                                                     \
    {                                                                                                    \
        const int tap_ = (TAPS == 3) ? ((kt_) >= kTiles) + ((kt_) >= 2 * kTiles) : 0;   /* no runtime division */                                               \
        const int k0_ = ((kt_) - tap_ * kTiles) * BK;                                                    \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                  \
            bool ok_ = a_row[i] < M;                                                                     \
            if (TAPS == 3) { const int tt_ = a_t[i] + tap_ - 1; ok_ = ok_ && tt_ >= 0 && tt_ < T; }      \
            const int src_ = ok_ ? a_src[i] + ((TAPS == 3) ? tap_ - 1 : 0) : 0;                          \
            ra[i] = *reinterpret_cast<const f32x4*>(A + (size_t)src_ * lda + k0_ + a_c4);                \
            a_ok[i] = ok_;                                                                               \
        }                                                                                                \
        const size_t wo_ = ((size_t)tap_ * N + n0 + b_r) * K + k0_ + b_c8;                               \
        _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                  \
            rbh[i] = *reinterpret_cast<const u32x4*>(Whi + wo_ + (size_t)32 * i * K);                    \
            if (SPLIT) rbl[i] = *reinterpret_cast<const u32x4*>(Wlo + wo_ + (size_t)32 * i * K);         \
        }                                                                                                \
    }
#define GB_STORE()                                                                                       \
    {                                                                                                    \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                  \
            const f32x4 v_ = a_ok[i] ? ra[i] : f32x4{0.f, 0.f, 0.f, 0.f};                                \
            unsigned int h_[4], l_[4];                                                                   \
            _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                              \
                h_[q] = f2bf(v_[q]);                                                                     \
                l_[q] = SPLIT ? f2bf(v_[q] - bf2f(h_[q])) : 0u;                                          \
            }                                                                                            \
            const int o_ = (a_r + 16 * i) * LD + a_c4;                                                   \
            *reinterpret_cast<u32x2*>(a_hi + o_) = u32x2{h_[0] | (h_[1] << 16), h_[2] | (h_[3] << 16)};  \
            if (SPLIT) *reinterpret_cast<u32x2*>(a_lo + o_) = u32x2{l_[0] | (l_[1] << 16), l_[2] | (l_[3] << 16)}; \
        }                                                                                                \
        _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                  \
            const int o_ = (b_r + 32 * i) * LD + b_c8;                                                   \
            *reinterpret_cast<u32x4*>(b_hi + o_) = rbh[i];                                               \
            if (SPLIT) *reinterpret_cast<u32x4*>(b_lo + o_) = rbl[i];                                    \
        }                                                                                                \
    }

    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    const int fr = lane & 31, fh = lane >> 5;
    const int ao = (wm * 32 + fr) * LD + 8 * fh, bo = (wn * 32 + fr) * LD + 8 * fh;

    GB_LOAD(kt_begin);
    GB_STORE();
    __syncthreads();
    for (int kt = kt_begin; kt < kt_end; ++kt) {
        const bool more = kt + 1 < kt_end;
        if (more) GB_LOAD(kt + 1);
#pragma unroll
        for (int st = 0; st < BK / 16; ++st) {
            const bf16x8 ah = *reinterpret_cast<const bf16x8*>(a_hi + ao + 16 * st);
            const bf16x8 bh = *reinterpret_cast<const bf16x8*>(b_hi + bo + 16 * st);
            if (SPLIT) {
                const bf16x8 al = *reinterpret_cast<const bf16x8*>(a_lo + ao + 16 * st);
                const bf16x8 bl = *reinterpret_cast<const bf16x8*>(b_lo + bo + 16 * st);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);     // small terms first
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
        }
        __syncthreads();                    // every wave has read the tile
        if (more) GB_STORE();
        __syncthreads();
    }
#undef GB_LOAD
#undef GB_STORE

    const int col = n0 + wn * 32 + fr;
    float bv = 0.f;
    if ((EPI == EPI_BIAS || EPI == EPI_BIAS_LRELU) && !split_k) bv = bias[col];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int row = m0 + wm * 32 + (e & 3) + 8 * (e >> 2) + 4 * fh;
        if (row < M) {
            float v = acc[e] + bv;
            if (EPI == EPI_BIAS_LRELU && !split_k) v = v > 0.f ? v : v * LEAKY_SLOPE;
            if (EPI == EPI_MASK && !split_k) v *= (aux[(size_t)row * ldc + col] > 0.f) ? 1.f : LEAKY_SLOPE;
            C[(size_t)row * ldc + col] = v;
        }
    }
}

// Large-batch variant: 128x128 output tile, every wave a 64x64 quarter (2x2 MFMA blocks), no split-K.  At 64x64 the
// kernel above is bound by the CU's 64 B/clk vector-memory path (24 KB of operands per 4 MFMAs per wave); here
// the operand bytes per MFMA are halved.  Used when the launch has enough 128x128 tiles to fill the chip.
template <int TAPS, int EPI, int NPROD>
__global__ __launch_bounds__(256) void gemm_bf16_big_kernel(const float* __restrict__ A, int lda, const uint16_t* __restrict__ Whi,
                                                            const uint16_t* __restrict__ Wlo, const float* __restrict__ bias,
                                                            const float* __restrict__ aux, float* __restrict__ C, int ldc, int M,
                                                            int N, int K, int T, const int* __restrict__ m_dev,
                                                            const int* __restrict__ row_map) {
    constexpr int BM = 128, BN = 128, BK = 64;
    constexpr int LD = BK + 8;
    constexpr int IMG = BM * LD;
    constexpr bool SPLIT = NPROD == 3;
    extern __shared__ __attribute__((aligned(16))) unsigned short ldsb[];
    unsigned short* a_hi = ldsb;
    unsigned short* b_hi = ldsb + IMG;
    unsigned short* a_lo = ldsb + 2 * IMG;           // only touched when SPLIT
    unsigned short* b_lo = ldsb + 3 * IMG;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    if (m_dev) M = *m_dev;
    if (m0 >= M) return;
    const int kTiles = K / BK, nTiles = TAPS * kTiles;

    // A tile: 128 rows x 64 fp32 = 2048 float4 -> 8 per thread; B tile: 128 rows x 64 bf16 = 1024 x 16 B -> 4 per thread
    const int a_c4 = (tid & 15) * 4, a_r = tid >> 4;
    const int b_c8 = (tid & 7) * 8, b_r = tid >> 3;
    i32x8 a_idx, a_t, a_in;          // source row (gathered through row_map for linear layers), its frame index, row < M
    bool a_ok[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int row = m0 + a_r + 16 * i;
        a_in[i] = row < M ? 1 : 0;
        a_t[i] = (TAPS == 3) ? (row % T) : 0;
        int s_ = row;
        if (TAPS == 1 && row_map) s_ = row_map[row < M ? row : 0];
        a_idx[i] = s_;
    }
    f32x4 ra[8];
    u32x4 rbh[4], rbl[4];

#define GBB_LOAD(kt_)                                                                                    \
    {                                                                                                    \
        const int tap_ = (TAPS == 3) ? ((kt_) >= kTiles) + ((kt_) >= 2 * kTiles) : 0;   /* no runtime division */                                               \
        const int k0_ = ((kt_) - tap_ * kTiles) * BK;                                                    \
        _Pragma("unroll") for (int i = 0; i < 8; ++i) {                                                  \
            bool ok_ = a_in[i] != 0;                                                                     \
            if (TAPS == 3) { const int tt_ = a_t[i] + tap_ - 1; ok_ = ok_ && tt_ >= 0 && tt_ < T; }      \
            const int src_ = ok_ ? a_idx[i] + ((TAPS == 3) ? tap_ - 1 : 0) : 0;                          \
            ra[i] = *reinterpret_cast<const f32x4*>(A + (size_t)src_ * lda + k0_ + a_c4);                \
            a_ok[i] = ok_;                                                                               \
        }                                                                                                \
        const size_t wo_ = ((size_t)tap_ * N + n0 + b_r) * K + k0_ + b_c8;                               \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                  \
            rbh[i] = *reinterpret_cast<const u32x4*>(Whi + wo_ + (size_t)32 * i * K);                    \
            if (SPLIT) rbl[i] = *reinterpret_cast<const u32x4*>(Wlo + wo_ + (size_t)32 * i * K);         \
        }                                                                                                \
    }
#define GBB_STORE()                                                                                      \
    {                                                                                                    \
        _Pragma("unroll") for (int i = 0; i < 8; ++i) {                                                  \
            const f32x4 v_ = a_ok[i] ? ra[i] : f32x4{0.f, 0.f, 0.f, 0.f};                                \
            unsigned int h_[4], l_[4];                                                                   \
            _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                              \
                h_[q] = f2bf(v_[q]);                                                                     \
                l_[q] = SPLIT ? f2bf(v_[q] - bf2f(h_[q])) : 0u;                                          \
            }                                                                                            \
            const int o_ = (a_r + 16 * i) * LD + a_c4;                                                   \
            *reinterpret_cast<u32x2*>(a_hi + o_) = u32x2{h_[0] | (h_[1] << 16), h_[2] | (h_[3] << 16)};  \
            if (SPLIT) *reinterpret_cast<u32x2*>(a_lo + o_) = u32x2{l_[0] | (l_[1] << 16), l_[2] | (l_[3] << 16)}; \
        }                                                                                                \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                  \
            const int o_ = (b_r + 32 * i) * LD + b_c8;                                                   \
            *reinterpret_cast<u32x4*>(b_hi + o_) = rbh[i];                                               \
            if (SPLIT) *reinterpret_cast<u32x4*>(b_lo + o_) = rbl[i];                                    \
        }                                                                                                \
    }

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    const int fr = lane & 31, fh = lane >> 5;
    const int ao = (wm * 64 + fr) * LD + 8 * fh, bo = (wn * 64 + fr) * LD + 8 * fh;

    GBB_LOAD(0);
    GBB_STORE();
    __syncthreads();
    for (int kt = 0; kt < nTiles; ++kt) {
        const bool more = kt + 1 < nTiles;
        if (more) GBB_LOAD(kt + 1);
#pragma unroll
        for (int st = 0; st < BK / 16; ++st) {
            bf16x8 ah[2], bh[2], al[2], bl[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                ah[i] = *reinterpret_cast<const bf16x8*>(a_hi + ao + i * 32 * LD + 16 * st);
                bh[i] = *reinterpret_cast<const bf16x8*>(b_hi + bo + i * 32 * LD + 16 * st);
                if (SPLIT) {
                    al[i] = *reinterpret_cast<const bf16x8*>(a_lo + ao + i * 32 * LD + 16 * st);
                    bl[i] = *reinterpret_cast<const bf16x8*>(b_lo + bo + i * 32 * LD + 16 * st);
                }
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    if (SPLIT) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);     // small terms first
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
                    }
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                }
        }
        __syncthreads();                    // every wave has read the tile
        if (more) GBB_STORE();
        __syncthreads();
    }
#undef GBB_LOAD
#undef GBB_STORE

#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = n0 + wn * 64 + j * 32 + fr;
            float bv = 0.f;
            if (EPI == EPI_BIAS || EPI == EPI_BIAS_LRELU) bv = bias[col];
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = m0 + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * fh;
                if (row < M) {
                    float v = acc[i][j][e] + bv;
                    if (EPI == EPI_BIAS_LRELU) v = v > 0.f ? v : v * LEAKY_SLOPE;
                    if (EPI == EPI_MASK) v *= (aux[(size_t)row * ldc + col] > 0.f) ? 1.f : LEAKY_SLOPE;
                    C[(size_t)row * ldc + col] = v;
                }
            }
        }
}

template <int TAPS, int EPI, int NPROD>
static int launch_b(gem_handle* h, const Layer& L, const float* A, int lda, const float* aux, float* C, int ldc, int M, int T,
                    hipStream_t s, const int* row_map) {
    const int* m_dev = h->ws.dyn ? h->ws.n_active + (TAPS == 3 ? 1 : 0) : nullptr;
    constexpr int BK = 64;
    // 128x128 tiles once they fill the chip (developer override GEM_BF16_TILE=1 / 2 forces 64x64 / 128x128)
    static const char* force = dev_env("GEM_BF16_TILE");
    const long big_blocks = (long)((M + 127) / 128) * (L.N / 128);
    if (L.N % 128 == 0 && ((force && force[0] == '2') || (!(force && force[0] == '1') && big_blocks >= 256))) {
        auto kb = gemm_bf16_big_kernel<TAPS, EPI, NPROD>;
        const size_t smem = (size_t)(NPROD == 3 ? 4 : 2) * 128 * (BK + 8) * sizeof(unsigned short);
        static PerDeviceOnce big_once;
        if (big_once.need(h->cfg.device)) {
            GEM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kb), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        }
        note_kernel(h, reinterpret_cast<const void*>(kb));
        hipLaunchKernelGGL(kb, dim3(L.N / 128, (M + 127) / 128, 1), dim3(256), smem, s, A, lda, L.wb_hi, L.wb_lo, L.bias, aux, C, ldc,
                           M, L.N, L.K, T, m_dev, row_map);
        GEM_HIP(hipGetLastError());
        return 0;
    }
    size_t shmem = (size_t)(NPROD == 3 ? 4 : 2) * 64 * (BK + 8) * sizeof(unsigned short);
    auto k = gemm_bf16_kernel<TAPS, EPI, NPROD>;
    static PerDeviceOnce attr_once;
    if (attr_once.need(h->cfg.device)) {
        GEM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    }
    const int n_tiles = TAPS * (L.K / BK);
    dim3 grid(L.N / 64, (M + 63) / 64, 1);
    const size_t slab = (size_t)M * ldc;
    const int sk = pick_splitk(h, (long)grid.x * grid.y, n_tiles, slab);
    const int per = (n_tiles + sk - 1) / sk;
    grid.z = (n_tiles + per - 1) / per;
    const long wgs = (long)grid.x * grid.y * grid.z, per_cu = (wgs + h->n_cu - 1) / h->n_cu;
    if (per_cu <= 8) {
        const size_t want = (((size_t)160 * 1024 - 8192) / per_cu) & ~(size_t)1023;      // see gemm_f32.hip
        if (want > shmem) shmem = want;
    }
    float* out = grid.z == 1 ? C : h->ws.splitk;
    const int dyn_W = (m_dev && grid.z > 1 && (size_t)wgs * 64 * 64 <= h->ws.splitk_elems) ? (int)wgs : 0;
    note_kernel(h, reinterpret_cast<const void*>(k));
    hipLaunchKernelGGL(k, grid, dim3(256), shmem, s, A, lda, L.wb_hi, L.wb_lo, L.bias, aux, out, ldc, M, L.N, L.K, T,
                       grid.z == 1 ? n_tiles : per, grid.z == 1 ? (size_t)0 : slab, m_dev, row_map, dyn_W);
    GEM_HIP(hipGetLastError());
    if (grid.z == 1) return 0;
    if (h->ws.defer_reduce) {                       // the consumer sums the slabs (and applies the epilogue) itself
        SlabSrc& d = h->ws.deferred;
        d.base = h->ws.splitk; d.nslab = (int)grid.z; d.stride = slab;
        d.dyn_W = dyn_W; d.n_tiles = n_tiles; d.ldc = ldc; d.CT = L.N / 64; d.m_dev = m_dev;
        return 0;
    }
    return launch_splitk_reduce(h, EPI, (int)grid.z, slab, L.bias, aux, C, M, L.N, ldc, m_dev, s, dyn_W, n_tiles);
}

template <int TAPS, int EPI>
static int launch_np(gem_handle* h, const Layer& L, int nprod, const float* A, int lda, const float* aux, float* C, int ldc, int M,
                     int T, hipStream_t s, const int* row_map) {
    if (nprod == 3) return launch_b<TAPS, EPI, 3>(h, L, A, lda, aux, C, ldc, M, T, s, row_map);
    return launch_b<TAPS, EPI, 1>(h, L, A, lda, aux, C, ldc, M, T, s, row_map);
}

int launch_gemm_bf16(gem_handle* h, const Layer& L, int epi, int nprod, const float* A, int lda, const float* aux, float* C, int ldc,
                     int M, int T, hipStream_t s, const int* row_map) {
    if (!L.wb_hi || (nprod == 3 && !L.wb_lo)) { set_error("launch_gemm_bf16: layer has no bf16 weights"); return 1; }
    if (L.K % 64 != 0 || L.N % 64 != 0 || lda % 4 != 0) { set_error("launch_gemm_bf16: dimensions must be padded to 64"); return 1; }
    if (M <= 0) return 0;
    if (L.taps == 1) {
        if (epi == EPI_BIAS) return launch_np<1, EPI_BIAS>(h, L, nprod, A, lda, aux, C, ldc, M, T, s, row_map);
        if (epi == EPI_NONE) return launch_np<1, EPI_NONE>(h, L, nprod, A, lda, aux, C, ldc, M, T, s, row_map);
        if (epi == EPI_BIAS_LRELU) return launch_np<1, EPI_BIAS_LRELU>(h, L, nprod, A, lda, aux, C, ldc, M, T, s, row_map);
    } else if (L.taps == 3) {
        if (epi == EPI_BIAS) return launch_np<3, EPI_BIAS>(h, L, nprod, A, lda, aux, C, ldc, M, T, s, row_map);
        if (epi == EPI_BIAS_LRELU) return launch_np<3, EPI_BIAS_LRELU>(h, L, nprod, A, lda, aux, C, ldc, M, T, s, row_map);
        if (epi == EPI_MASK) return launch_np<3, EPI_MASK>(h, L, nprod, A, lda, aux, C, ldc, M, T, s, row_map);
        if (epi == EPI_NONE) return launch_np<3, EPI_NONE>(h, L, nprod, A, lda, aux, C, ldc, M, T, s, row_map);
    }
    set_error("launch_gemm_bf16: unsupported taps / epilogue");
    return 1;
}

}  // namespace gem
