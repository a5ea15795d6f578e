// fp32 MFMA GEMM / temporal-conv kernel for the motion-VAE layers (gfx950).
//
//   C[M,N] = epi( sum_{tap<TAPS} shift_{tap-1}(A)[M,K] . W[tap][N][K]^T + bias[N] )
//
// Rows are (window, frame) pairs, row = b*T + t, so a k=3/s=1/p=1 temporal convolution
// (Conv1d / ConvTranspose1d of networks/models/SeqConvVAE.py:36,70-75,83-92) is three shifted
// GEMMs accumulated into the same tile, with rows whose shifted frame falls outside the window
// contributing zero (the zero padding of the reference).  TAPS = 1 is nn.Linear
// (decoder_input / fc_mu / fc_var, SeqConvVAE.py:44-45,62) and the backward-data products.
//
// Arithmetic: v_mfma_f32_32x32x2_f32 -- exact fp32 FMA chains, the matrix-core rate for fp32
// (157 TFLOP/s peak on MI355X).  Each wave owns RM x RN tiles of 32x32; a workgroup is 2x2 waves.
// K is walked in steps of 32 through a double-buffered LDS image [rows][32+4] (k contiguous, one
// 16-byte pad per row so the ds_read_b128 fragment reads of 32 different rows spread over the banks).
// The MFMA k-pairing is permuted (step s pairs k=s with k=BK/2+s) so each lane's 16 A and 16 B values of
// a K-step are four contiguous 16-byte LDS reads; a sum over k does not care about the pairing.
#include <cstdio>
#include <cstdlib>

#include "gem_internal.h"
#include "gemm_glds.h"
#include "gemm_rows.h"

namespace gem {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));        // per-thread row tables as NATIVE vectors: a plain int[] went to scratch    // native vector: keeps staged tiles in VGPRs (a float4 struct array went to scratch)

constexpr int BK_MIN = 32;      // K padding granularity

#ifdef GEM_TRACE
// tools/gemm_bench.hip only: per-workgroup {start, first MFMA possible, loop end, end} on the 100 MHz wall clock
__device__ long long* g_gemm_trace = nullptr;
#define GEM_TRACE_MARK(slot_)                                                                            \
    if (g_gemm_trace && threadIdx.x == 0)                                                                \
        g_gemm_trace[4 * (blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) + (slot_)] = wall_clock64();
#else
#define GEM_TRACE_MARK(slot_)
#endif

template <int TAPS, int EPI, int RM, int RN, int TAG, int BK>
__global__ __launch_bounds__(256) void gemm_f32_kernel(const float* __restrict__ A, int lda,
                                                       const float* __restrict__ W,
                                                       const float* __restrict__ bias,
                                                       const float* __restrict__ aux, float* __restrict__ C,
                                                       int ldc, int M, int N, int K, int T, int tiles_per_slice,
                                                       size_t slab_stride, const int* __restrict__ m_dev,
                                                       const int* __restrict__ row_map, int dyn_W) {
    constexpr int BM = 64 * RM, BN = 64 * RN;
    constexpr int LDS_LD = BK + 4;
    constexpr int TPR = BK / 4;             // threads per tile row (one float4 each)
    constexpr int RPP = 256 / TPR;          // tile rows per pass of the 256 threads
    constexpr int A_LD4 = BM / RPP;         // float4 loads per thread for the A tile
    constexpr int B_LD4 = BN / RPP;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int BUF = (BM + BN) * LDS_LD;      // floats per LDS buffer: A tile then B tile

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    GEM_TRACE_MARK(0);
    int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    if (m_dev) M = *m_dev;                 // rows in use this round (active windows are compacted to the front)
    const int kTiles = K / BK;
    // split-K: a slice owns k-tiles [kt_begin, kt_end) of the TAPS*K/BK tiles and writes a raw partial slab
    bool split = gridDim.z > 1;
    int kt_begin = blockIdx.z * tiles_per_slice;
    int kt_end = min(TAPS * kTiles, kt_begin + tiles_per_slice);
    if (dyn_W > 0) {
        // evaluation rounds: re-cut the grid for the rows that are still active (see dyn_split)
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
        split = true;
        C += (size_t)ks * d.slab;
    } else {
        if (m0 >= M) return;
        if (split) C += (size_t)blockIdx.z * slab_stride;
    }

    // ---- per-thread global load coordinates (branch-free: out-of-range rows read row 0 and are zeroed)
    const int c4 = (tid % TPR) * 4;
    const int lrow = tid / TPR;
    static_assert(A_LD4 <= 8, "row tables hold up to 8 entries");
    i32x8 a_row, a_t, a_src;
#pragma unroll
    for (int i = 0; i < A_LD4; ++i) {
        const int r_ = m0 + lrow + RPP * i;
        a_row[i] = r_;
        a_t[i] = (TAPS == 3) ? (r_ % T) : 0;
        int s_ = r_;
        if (TAPS == 1 && row_map) s_ = row_map[r_ < M ? r_ : 0];           // gathered A rows (linear only)
        a_src[i] = s_;
    }
    f32x4 ra[A_LD4], rb[B_LD4];
    bool a_ok[A_LD4];

#define GEM_LOAD_TILE(kt_)                                                                                   \
    {                                                                                                        \
        const int tap_ = (TAPS == 3) ? ((kt_) >= kTiles) + ((kt_) >= 2 * kTiles) : 0;   /* no runtime division */                                                   \
        const int k0_ = ((kt_) - tap_ * kTiles) * BK + c4;                                                   \
        _Pragma("unroll") for (int i = 0; i < A_LD4; ++i) {                                                  \
            bool ok_ = a_row[i] < M;                                                                         \
            if (TAPS == 3) {                                                                                 \
                const int tt_ = a_t[i] + tap_ - 1;                                                           \
                ok_ = ok_ && tt_ >= 0 && tt_ < T;                                                            \
            }                                                                                                \
            const int src_ = ok_ ? a_src[i] + ((TAPS == 3) ? tap_ - 1 : 0) : 0;                               \
            ra[i] = *reinterpret_cast<const f32x4*>(A + (size_t)src_ * lda + k0_);                           \
            a_ok[i] = ok_;  /* zeroing happens at the LDS store, after the MFMAs: no wait on the load here */ \
        }                                                                                                    \
        const float* Wt_ = W + (size_t)tap_ * N * K;                                                         \
        _Pragma("unroll") for (int i = 0; i < B_LD4; ++i)                                                    \
            rb[i] = *reinterpret_cast<const f32x4*>(Wt_ + (size_t)(n0 + lrow + RPP * i) * K + k0_);          \
    }
#define GEM_STORE_TILE(buf_)                                                                                 \
    {                                                                                                        \
        _Pragma("unroll") for (int i = 0; i < A_LD4; ++i)                                                    \
            *reinterpret_cast<f32x4*>(lds + (buf_) * BUF + (lrow + RPP * i) * LDS_LD + c4) =                  \
                a_ok[i] ? ra[i] : f32x4{0.f, 0.f, 0.f, 0.f};                                                 \
        _Pragma("unroll") for (int i = 0; i < B_LD4; ++i)                                                    \
            *reinterpret_cast<f32x4*>(lds + (buf_) * BUF + (BM + lrow + RPP * i) * LDS_LD + c4) = rb[i];      \
    }

    f32x16 acc[RM][RN];
#pragma unroll
    for (int i = 0; i < RM; ++i)
#pragma unroll
        for (int j = 0; j < RN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int fr = lane & 31, fh = lane >> 5;
    // 64x64 tiles: ONE LDS buffer (18 KB) so that up to 7 workgroups share a CU and hide each other's
    // barriers; 128x128 tiles: two buffers, one barrier per K-step.
    constexpr bool DBUF = (RM * RN > 1);
    GEM_LOAD_TILE(kt_begin);
    GEM_STORE_TILE(0);
    __syncthreads();
    GEM_TRACE_MARK(1);
    int cur = 0;
#ifndef GEM_ABLATE
#define GEM_ABLATE 0
#endif
    for (int kt = kt_begin; kt < kt_end; ++kt) {
        const bool more = kt + 1 < kt_end;
        if (more && GEM_ABLATE < 1) GEM_LOAD_TILE(kt + 1);
        const float* as = lds + cur * BUF + (wm * 32 * RM + fr) * LDS_LD + fh * (BK / 2);
        const float* bs = lds + cur * BUF + (BM + wn * 32 * RN + fr) * LDS_LD + fh * (BK / 2);
        if (RM * RN == 1 && GEM_ABLATE == 0) {
            // 64x64 tiles: all fragments of the k-tile are requested up front, so that the MFMAs of the first k-steps
            // cover the LDS latency of the later ones (read-then-wait per group exposed ~100 cycles twice per tile)
            f32x4 av[BK / 8], bv[BK / 8];
#pragma unroll
            for (int q = 0; q < BK / 8; ++q) {
                av[q] = *reinterpret_cast<const f32x4*>(as + 4 * q);
                bv[q] = *reinterpret_cast<const f32x4*>(bs + 4 * q);
            }
            __builtin_amdgcn_sched_barrier(0);          // (the scheduler would otherwise sink the later reads again)
#pragma unroll
            for (int q = 0; q < BK / 8; ++q) {
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q].x, bv[q].x, acc[0][0], 0, 0, 0);
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q].y, bv[q].y, acc[0][0], 0, 0, 0);
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q].z, bv[q].z, acc[0][0], 0, 0, 0);
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q].w, bv[q].w, acc[0][0], 0, 0, 0);
            }
        } else {
#pragma unroll
        for (int q = 0; q < BK / 8; ++q) {
            f32x4 av[RM], bv[RN];
#pragma unroll
            for (int i = 0; i < RM; ++i) av[i] = *reinterpret_cast<const f32x4*>(as + i * 32 * LDS_LD + 4 * (GEM_ABLATE >= 3 ? 0 : q));
#pragma unroll
            for (int j = 0; j < RN; ++j) bv[j] = *reinterpret_cast<const f32x4*>(bs + j * 32 * LDS_LD + 4 * (GEM_ABLATE >= 3 ? 0 : q));
            if (GEM_ABLATE >= 3) {      // keep the operands opaque so the loads are not re-issued per q
#pragma unroll
                for (int i = 0; i < RM; ++i) asm volatile("" : "+v"(av[i]));
            }
#pragma unroll
            for (int i = 0; i < RM; ++i)
#pragma unroll
                for (int j = 0; j < RN; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i].x, bv[j].x, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i].y, bv[j].y, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i].z, bv[j].z, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i].w, bv[j].w, acc[i][j], 0, 0, 0);
                }
        }
        }
        if (GEM_ABLATE >= 2) continue;          // ablation build: MFMA + LDS reads only
        if (DBUF) {
            if (more) GEM_STORE_TILE(cur ^ 1);
            __syncthreads();
            cur ^= 1;
        } else {
            __syncthreads();                    // every wave has read the tile
            if (more) GEM_STORE_TILE(0);
            __syncthreads();
        }
    }
#undef GEM_LOAD_TILE
#undef GEM_STORE_TILE
    GEM_TRACE_MARK(2);

    // ---- epilogue: C/D layout of 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
#pragma unroll
    for (int i = 0; i < RM; ++i)
#pragma unroll
        for (int j = 0; j < RN; ++j) {
            const int col = n0 + wn * 32 * RN + j * 32 + fr;
            float bv = 0.f;
            if ((EPI == EPI_BIAS || EPI == EPI_BIAS_LRELU) && !split) bv = bias[col];
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = m0 + wm * 32 * RM + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * fh;
                if (row < M) {
                    float v = acc[i][j][e] + bv;
                    if (EPI == EPI_BIAS_LRELU && !split) v = v > 0.f ? v : v * LEAKY_SLOPE;
                    if (EPI == EPI_MASK && !split) v *= (aux[(size_t)row * ldc + col] > 0.f) ? 1.f : LEAKY_SLOPE;
                    C[(size_t)row * ldc + col] = v;
                }
            }
        }
    GEM_TRACE_MARK(3);
}

// sums the split-K slabs and applies the epilogue: C = epi(sum_z slab[z] + bias)
template <int EPI>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ slabs, int nslab, size_t slab_stride,
                                                            const float* __restrict__ bias, const float* __restrict__ aux,
                                                            float* __restrict__ C, int M, int N, int ldc,
                                                            const int* __restrict__ m_dev, int dyn_W, int n_tiles) {
    if (m_dev) M = *m_dev;
    if (dyn_W > 0) {                     // slices and slab stride of this round, as the GEMM kernel computed them
        const DynSplit d = dyn_split(M, 64, N / 64, dyn_W, n_tiles, ldc);
        nslab = d.SK;
        slab_stride = d.slab;
    }
    const int n4 = N / 4;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)M * n4) return;
    const int row = (int)(i / n4), c = (int)(i - (size_t)row * n4) * 4;
    const size_t off = (size_t)row * ldc + c;
    f32x4 v = *reinterpret_cast<const f32x4*>(slabs + off);
    for (int z = 1; z < nslab; ++z) v += *reinterpret_cast<const f32x4*>(slabs + (size_t)z * slab_stride + off);
    if (EPI == EPI_BIAS || EPI == EPI_BIAS_LRELU) v += *reinterpret_cast<const f32x4*>(bias + c);
    if (EPI == EPI_BIAS_LRELU) {
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = v[q] > 0.f ? v[q] : v[q] * LEAKY_SLOPE;
    }
    if (EPI == EPI_MASK) {
        const f32x4 m = *reinterpret_cast<const f32x4*>(aux + off);
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] *= m[q] > 0.f ? 1.f : LEAKY_SLOPE;
    }
    *reinterpret_cast<f32x4*>(C + off) = v;
}

// How many K slices.  The dispatcher packs a CU with as many workgroups as fit (7 of these), so a grid of
// 1280 small workgroups lands on 183 of the 256 CUs; the launch below therefore also pads the LDS request
// so that exactly ceil(workgroups/256) fit per CU.  Pick the slice count whose workgroup total fills those
// slots best (workgroups / (256 * per_cu)), preferring >= 2 per CU and fewer slices.
constexpr int LDS_PER_CU = 160 * 1024;
int pick_splitk(const gem_handle* h, long blocks, int n_tiles, size_t slab_elems) {
    const int N_CU = h->n_cu;             // compute units of this handle's device
    if (!h->ws.splitk || blocks >= 3 * N_CU) return 1;
    static const int force = dev_env("GEM_FORCE_SK") ? atoi(dev_env("GEM_FORCE_SK")) : 0;      // developer override (sweeps)
    if (force > 0 && n_tiles / force >= 4 && (size_t)force * slab_elems <= h->ws.splitk_elems) return force;
    int best = 1;
    double best_score = -1.0;
    for (int sk = 1; sk <= 8; ++sk) {
        if (sk > 1 && (n_tiles / sk < 6 || (size_t)sk * slab_elems > h->ws.splitk_elems)) break;
        const long wgs = blocks * sk;
        const long per_cu = (wgs + N_CU - 1) / N_CU;
        double score = (double)wgs / (double)(per_cu * N_CU);
        if (per_cu < 2) score *= 0.6;             // one workgroup per CU cannot hide its own barriers
        score -= 0.01 * sk;                        // slabs cost a reduce pass
        if (score > best_score) { best_score = score; best = sk; }
    }
    return best;
}

int launch_splitk_reduce(gem_handle* h, int epi, int nslab, size_t slab, const float* bias, const float* aux, float* C, int M, int N,
                         int ldc, const int* m_dev, hipStream_t s, int dyn_W, int n_tiles) {
    const size_t n4 = (size_t)M * (N / 4);
    const dim3 grid((unsigned)((n4 + 255) / 256));
    switch (epi) {
        case EPI_BIAS: hipLaunchKernelGGL(splitk_reduce_kernel<EPI_BIAS>, grid, dim3(256), 0, s, h->ws.splitk, nslab, slab, bias, aux, C, M, N, ldc, m_dev, dyn_W, n_tiles); break;
        case EPI_BIAS_LRELU: hipLaunchKernelGGL(splitk_reduce_kernel<EPI_BIAS_LRELU>, grid, dim3(256), 0, s, h->ws.splitk, nslab, slab, bias, aux, C, M, N, ldc, m_dev, dyn_W, n_tiles); break;
        case EPI_MASK: hipLaunchKernelGGL(splitk_reduce_kernel<EPI_MASK>, grid, dim3(256), 0, s, h->ws.splitk, nslab, slab, bias, aux, C, M, N, ldc, m_dev, dyn_W, n_tiles); break;
        default: hipLaunchKernelGGL(splitk_reduce_kernel<EPI_NONE>, grid, dim3(256), 0, s, h->ws.splitk, nslab, slab, bias, aux, C, M, N, ldc, m_dev, dyn_W, n_tiles); break;
    }
    GEM_HIP(hipGetLastError());
    return 0;
}

template <int TAPS, int EPI, int RM, int RN, int TAG, int BK>
static int launch_one(gem_handle* h, const Layer& L, const float* A, int lda, const float* aux, float* C, int ldc, int M, int T,
                      hipStream_t s, const int* row_map) {
    // during the evaluation rounds the row count lives on the device: {n_active, n_active*T}
    const int* m_dev = h->ws.dyn ? h->ws.n_active + (TAPS == 3 ? 1 : 0) : nullptr;
    constexpr int BM = 64 * RM, BN = 64 * RN;
    size_t shmem = (RM * RN > 1 ? 2 : 1) * (size_t)(BM + BN) * (BK + 4) * sizeof(float);
    auto k = gemm_f32_kernel<TAPS, EPI, RM, RN, TAG, BK>;
    static PerDeviceOnce attr_once;
    if (attr_once.need(h->cfg.device)) {
        GEM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_PER_CU));
    }
    const int n_tiles = TAPS * (L.K / BK);
    dim3 grid(L.N / BN, (M + BM - 1) / BM, 1);
    const size_t slab = (size_t)M * ldc;
    const int sk = pick_splitk(h, (long)grid.x * grid.y, n_tiles, slab);
    const int per = (n_tiles + sk - 1) / sk;
    grid.z = (n_tiles + per - 1) / per;
    // occupancy cap = even spread: LDS request sized so that only ceil(workgroups/256) fit on a CU
    const long wgs = (long)grid.x * grid.y * grid.z;
    const long per_cu = (wgs + h->n_cu - 1) / h->n_cu;
    if (per_cu <= 8) {
        // (a few KB below the even share: exactly 160 KB / per_cu each did NOT fit per_cu workgroups -- the stragglers
        // ran as a second wave, see tools/gemm_trace.hip)
        const size_t want = (((size_t)LDS_PER_CU - 8192) / per_cu) & ~(size_t)1023;
        if (want > shmem) shmem = want;
    }
    // evaluation rounds on small batches: slices re-cut on the device for the rows that are still active
    const bool dyn = m_dev && RM * RN == 1 && grid.z > 1 && (size_t)wgs * 64 * BN <= h->ws.splitk_elems;
    if (grid.z == 1) {
        note_kernel(h, reinterpret_cast<const void*>(k));
        hipLaunchKernelGGL(k, grid, dim3(256), shmem, s, A, lda, L.w, L.bias, aux, C, ldc, M, L.N, L.K, T, n_tiles, (size_t)0, m_dev,
                           row_map, 0);
        GEM_HIP(hipGetLastError());
        return 0;
    }
    note_kernel(h, reinterpret_cast<const void*>(k));
    hipLaunchKernelGGL(k, grid, dim3(256), shmem, s, A, lda, L.w, L.bias, aux, h->ws.splitk, ldc, M, L.N, L.K, T, per, slab, m_dev,
                       row_map, dyn ? (int)wgs : 0);
    GEM_HIP(hipGetLastError());
    if (h->ws.defer_reduce && BM == 64) {           // the consumer sums the slabs (and applies the epilogue) itself
        SlabSrc& d = h->ws.deferred;
        d.base = h->ws.splitk; d.nslab = (int)grid.z; d.stride = slab;
        d.dyn_W = dyn ? (int)wgs : 0; d.n_tiles = n_tiles; d.ldc = ldc; d.CT = L.N / BN; d.m_dev = m_dev;
        return 0;
    }
    return launch_splitk_reduce(h, EPI, (int)grid.z, slab, L.bias, aux, C, M, L.N, ldc, m_dev, s, dyn ? (int)wgs : 0, n_tiles);
}

// Large batches: the LDS-DMA kernel of gemm_glds.h with fp32 operands (128x128x32 tiles, one wave per 64x64 with 16
// independent 16x16x4 accumulators, operands global -> LDS by DMA).  Measured on the decoder_input shapes at 8192
// windows: 133 TFLOP/s = 0.85 of the fp32 matrix peak (tools/gemm_glds_bench), against 113-127 for the register-staged
// 128x128 kernel above.  No split-K: only used when the launch has >= 1.5 tiles per CU (480 tiles at 1536 windows).
template <int TAPS, int EPI>
static int launch_glds_f32(gem_handle* h, const Layer& L, const float* A, int lda, const float* aux, float* C, int ldc, int M, int T,
                           hipStream_t s, const int* row_map) {
    constexpr int BM = 128, BN = 128;
    auto k = glds::gemm_glds_kernel<true, TAPS, EPI, BM, BN, false, 16>;
    constexpr size_t smem = (size_t)BM * BN * 4;
    static PerDeviceOnce once;
    if (once.need(h->cfg.device))
        GEM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_PER_CU));
    glds::Args a{};
    a.A = A; a.W = L.w; a.bias = L.bias; a.aux = aux; a.C = C; a.zero16 = h->ws.zero16;
    a.m_dev = h->ws.dyn ? h->ws.n_active + (TAPS == 3 ? 1 : 0) : nullptr;
    a.row_map = row_map;
    a.lda = lda; a.ldc = ldc; a.M = M; a.N = L.N; a.K = L.K; a.T = T;
    a.n_split = 1; a.tiles_per_split = TAPS * (L.K / 32); a.slab_stride = 0;
    const int grid = ((M + BM - 1) / BM) * (L.N / BN);
    note_kernel(h, reinterpret_cast<const void*>(k));
    hipLaunchKernelGGL(k, dim3(grid), dim3(256), smem, s, a);
    GEM_HIP(hipGetLastError());
    return 0;
}

// Few rows (one sequence: <= ~256 windows), linear layers: the weight-streaming kernel of gemm_rows.h -- one workgroup per CU,
// all active rows against one 64-column weight tile.  Forward (direct output, bias fused): no split-K, no slabs, no reduce
// pass.  Backward (the consumer sums slabs anyway): as few K slices as fill the chip.  Used when the cut fills >= 3/4 of the
// CUs' matrix time; otherwise (tiny batches, large batches) the tiled kernels below take over.
template <int S, int RT, bool FUSE>
static int launch_rows_as(gem_handle* h, const Layer& L, const float* A, int lda, float* C, int ldc, int M, hipStream_t s,
                          const int* row_map, const rows::Plan& p, bool direct, bool with_bias, bool lrelu) {
    auto k = rows::gemm_rows_kernel<S, RT, FUSE>;
    static PerDeviceOnce once;
    if (once.need(h->cfg.device))          // (the fused-compaction variant also has 2 KB of static LDS: ask for the ring only)
        GEM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    S * rows::Geometry<RT>::STAGE_BYTES));
    Workspace& w = h->ws;
    rows::Args a{};
    a.A = A; a.W = L.w; a.bias = direct && with_bias ? L.bias : nullptr; a.lrelu = lrelu ? 1 : 0; a.C = direct ? C : w.splitk;
    a.m_dev = w.dyn ? w.n_active : nullptr;
    a.row_map = row_map;
    a.lda = lda; a.ldc = ldc; a.M = M; a.N = L.N; a.K = L.K;
    a.n_rb = p.n_rb; a.n_split = p.n_split; a.tiles_per_split = p.per; a.slab_stride = (size_t)M * ldc;
    if (FUSE) {          // this launch also re-packs the windows that are still iterating (see rows::Args)
        a.phase_arr = w.phase; a.n_windows = M; a.done_phase = w.done_phase; a.T = h->T;
        a.perm_out = w.perm; a.slot_of_out = w.slot_of; a.n_active_out = w.n_active; a.log_slot = w.fuse_log;
    }
    const int grid = p.n_rb * (L.N / rows::BN) * p.n_split;
    if (!FUSE && S == 4 && RT == 5 && !direct && w.fuse_lbfgs && w.dyn) {          // experiment: this launch carries lbfgs_advance too
        const int rc = launch_rows_bwd_lbfgs(h, a, grid, (size_t)S * rows::Geometry<RT>::STAGE_BYTES, w.deferred, s);
        if (rc >= 0) return rc;
    }
    note_kernel(h, reinterpret_cast<const void*>(k));
    hipLaunchKernelGGL(k, dim3(grid), dim3(256), (size_t)S * rows::Geometry<RT>::STAGE_BYTES, s, a);
    GEM_HIP(hipGetLastError());
    return 0;
}

// the cut for a linear layer with M rows, or false when this kernel is not the one for the shape
static bool rows_plan(const gem_handle* h, const Layer& L, int lda, int ldc, int M, bool slabs, rows::Plan* out, bool* use8) {
    static const bool off = dev_env("GEM_NO_ROWS") != nullptr;                 // developer overrides (A/B runs)
    static const char* only = dev_env("GEM_ROWS_ONLY");                        // "fwd": direct-output launches only
    if (off || M < 48 || L.N % rows::BN != 0 || L.K % rows::BK != 0 || (size_t)h->ws.Bmax * lda * 4 >= ((size_t)1 << 32) ||
        (size_t)L.N * L.K * 4 >= ((size_t)1 << 32))
        return false;
    if (slabs && only && only[0] == 'f') return false;
    const rows::Plan p5 = rows::plan(M, L.N, L.K, 5, h->n_cu, slabs, h->ws.splitk_elems, ldc);
    const rows::Plan p8 = rows::plan(M, L.N, L.K, 8, h->n_cu, slabs, h->ws.splitk_elems, ldc);
    *use8 = p8.n_rb > 0 && (p5.n_rb == 0 || p8.cost < p5.cost - 1e-9);          // ties: the smaller ring
    *out = *use8 ? p8 : p5;
    static const char* force = dev_env("GEM_ROWS_FORCE");      // developer override: "N:n_rb,n_split" for layers with that N
    if (force) {
        int fn = 0, frb = 0, fsk = 0;
        if (sscanf(force, "%d:%d,%d", &fn, &frb, &fsk) == 3 && fn == L.N && frb > 0 && fsk > 0 && (fsk == 1 || slabs)) {
            const int R = (M + 15) / 16, nk = L.K / rows::BK, rpb = (R + frb - 1) / frb;
            if (rpb <= 8 && (long)frb * (L.N / rows::BN) * fsk <= h->n_cu) {
                *use8 = rpb > 5;
                *out = rows::Plan{frb, fsk, (nk + fsk - 1) / fsk, 1.0, 0.0};
            }
        }
    }
    return out->n_rb != 0 && out->fill >= 0.75;
}

// Will the decoder_input forward product of a B-window round run in the few-rows kernel, so that it can re-pack the active
// windows itself (no compact_kernel launch between the rounds)?
bool rows_can_fuse_compaction(const gem_handle* h, const Layer& L, int lda, int ldc, int B, bool slabs) {
    static const bool off = dev_env("GEM_NO_FUSED_COMPACT") != nullptr;        // developer override (A/B runs)
    rows::Plan p;
    bool use8;
    return !off && h->precision == GEM_PRECISION_F32 && B <= rows::FUSE_MAX_WINDOWS &&
           rows_plan(h, L, lda, ldc, B, slabs && h->ws.splitk, &p, &use8);
}

// returns -1 when the shape is not one for this kernel (the caller falls through to the tiled kernels)
template <int EPI>
static int launch_rows(gem_handle* h, const Layer& L, const float* A, int lda, const float* aux, float* C, int ldc, int M,
                       hipStream_t s, const int* row_map) {
    // direct output unless the consumer takes slabs (decoder_input backward: lbfgs_advance_kernel sums them)
    const bool slabs = h->ws.defer_reduce && h->ws.splitk;
    rows::Plan p;
    bool use8;
    if (!rows_plan(h, L, lda, ldc, M, slabs, &p, &use8)) return -1;
    const bool direct = p.n_split == 1;
    if (!direct) {
        SlabSrc& d = h->ws.deferred;
        d.base = h->ws.splitk; d.nslab = p.n_split; d.stride = (size_t)M * ldc;
        d.dyn_W = 0; d.n_tiles = 0; d.ldc = ldc; d.CT = L.N / 64; d.m_dev = h->ws.dyn ? h->ws.n_active : nullptr;
    }
    (void)aux;
    constexpr bool with_bias = EPI != EPI_NONE, lrelu = EPI == EPI_BIAS_LRELU;
    if (h->ws.fuse_compact) {          // set by the round loop for the first launch of a round only
        h->ws.fuse_compact = false;
        if (!row_map || M > rows::FUSE_MAX_WINDOWS) { set_error("launch_rows: fused compaction on a launch that cannot carry it"); return 1; }
        return use8 ? launch_rows_as<3, 8, true>(h, L, A, lda, C, ldc, M, s, row_map, p, direct, with_bias, lrelu)
                    : launch_rows_as<4, 5, true>(h, L, A, lda, C, ldc, M, s, row_map, p, direct, with_bias, lrelu);
    }
    return use8 ? launch_rows_as<3, 8, false>(h, L, A, lda, C, ldc, M, s, row_map, p, direct, with_bias, lrelu)
                : launch_rows_as<4, 5, false>(h, L, A, lda, C, ldc, M, s, row_map, p, direct, with_bias, lrelu);
}

template <int TAPS, int EPI, int TAG>
static int launch_tile(gem_handle* h, const Layer& L, const float* A, int lda, const float* aux, float* C, int ldc, int M, int T,
                       hipStream_t s, const int* row_map) {
    if (TAPS == 1 && (EPI == EPI_BIAS || EPI == EPI_NONE || EPI == EPI_BIAS_LRELU)) {
        const int rc = launch_rows<EPI>(h, L, A, lda, aux, C, ldc, M, s, row_map);
        if (rc >= 0) return rc;
    }
    static const bool no_glds = dev_env("GEM_NO_GLDS_F32") != nullptr;          // developer override (A/B runs)
    if (!no_glds && L.N % 128 == 0 && L.K % 32 == 0 && (long)((M + 127) / 128) * (L.N / 128) >= 3L * h->n_cu / 2 && h->ws.zero16 &&
        !h->ws.defer_reduce)
        return launch_glds_f32<TAPS, EPI>(h, L, A, lda, aux, C, ldc, M, T, s, row_map);
    // 128x128 tiles only when they still fill the chip (>= 2 workgroups per CU) and divide N
    const long big_blocks = (long)((M + 127) / 128) * (L.N / 128);
    static const char* force = dev_env("GEM_FORCE_TILE");       // developer override: "1" = 64x64, "2" = 128x128, "3" = 64x64 BK64
    if (force && force[0] == '1') return launch_one<TAPS, EPI, 1, 1, TAG, 32>(h, L, A, lda, aux, C, ldc, M, T, s, row_map);
    if (force && force[0] == '2' && L.N % 128 == 0) return launch_one<TAPS, EPI, 2, 2, TAG, 32>(h, L, A, lda, aux, C, ldc, M, T, s, row_map);
    if (force && force[0] == '3' && L.K % 64 == 0) return launch_one<TAPS, EPI, 1, 1, TAG, 64>(h, L, A, lda, aux, C, ldc, M, T, s, row_map);
    if (force && force[0] == '4') return launch_one<TAPS, EPI, 2, 1, TAG, 32>(h, L, A, lda, aux, C, ldc, M, T, s, row_map);
    if (force && force[0] == '5' && L.N % 128 == 0) return launch_one<TAPS, EPI, 1, 2, TAG, 32>(h, L, A, lda, aux, C, ldc, M, T, s, row_map);
    if (L.N % 128 == 0 && big_blocks >= 512) return launch_one<TAPS, EPI, 2, 2, TAG, 32>(h, L, A, lda, aux, C, ldc, M, T, s, row_map);
    return launch_one<TAPS, EPI, 1, 1, TAG, 32>(h, L, A, lda, aux, C, ldc, M, T, s, row_map);
}

int launch_gemm(gem_handle* h, const Layer& L, int epi, const float* A, int lda, const float* aux, float* C, int ldc, int M,
                int T, hipStream_t s, int family, const int* row_map) {
    if (L.K % BK_MIN != 0 || L.N % 64 != 0 || lda % 4 != 0) {
        set_error("launch_gemm: dimensions must be padded (K%32, N%64, lda%4)");
        return 1;
    }
    if (M <= 0) return 0;
    h->ws.deferred = SlabSrc{};
    Profile::Rec rec;
    const bool prof = h->prof.on && family >= 0;
    if (prof) {
        GEM_HIP(hipEventCreate(&rec.a));
        GEM_HIP(hipEventCreate(&rec.b));
        rec.family = family;
        rec.flops = 2.0 * M * (double)L.N * L.K * L.taps;
        if (h->ws.dyn && L.taps == 1) {      // rows = active windows of this round, known only on the device
            rec.log_idx = h->ws.cur_log;
            rec.flops_per_window = 2.0 * (double)L.N * L.K;
        }
        GEM_HIP(hipEventRecord(rec.a, s));
    }
    int rc = 1;
    if (h->precision != GEM_PRECISION_F32 && L.wb_hi && (h->precision == GEM_PRECISION_BF16 || L.wb_lo)) {
        rc = launch_gemm_bf16(h, L, epi, h->precision == GEM_PRECISION_BF16 ? 1 : 3, A, lda, aux, C, ldc, M, T, s, row_map);
    } else if (L.taps == 1) {
        // TAG 1 = the decoder_input products (forward and backward-data): the dominant kernel gets its own symbol
        if (family == 0 && epi == EPI_BIAS) rc = launch_tile<1, EPI_BIAS, 1>(h, L, A, lda, aux, C, ldc, M, T, s, row_map);
        else if (epi == EPI_BIAS) rc = launch_tile<1, EPI_BIAS, 0>(h, L, A, lda, aux, C, ldc, M, T, s, row_map);
        else if (epi == EPI_NONE) rc = launch_tile<1, EPI_NONE, 0>(h, L, A, lda, aux, C, ldc, M, T, s, row_map);
        else if (epi == EPI_BIAS_LRELU) rc = launch_tile<1, EPI_BIAS_LRELU, 1>(h, L, A, lda, aux, C, ldc, M, T, s, row_map);
        else set_error("launch_gemm: unsupported epilogue for a linear layer");
    } else if (L.taps == 3) {
        if (epi == EPI_BIAS) rc = launch_tile<3, EPI_BIAS, 0>(h, L, A, lda, aux, C, ldc, M, T, s, row_map);
        else if (epi == EPI_BIAS_LRELU) rc = launch_tile<3, EPI_BIAS_LRELU, 0>(h, L, A, lda, aux, C, ldc, M, T, s, row_map);
        else if (epi == EPI_MASK) rc = launch_tile<3, EPI_MASK, 0>(h, L, A, lda, aux, C, ldc, M, T, s, row_map);
        else if (epi == EPI_NONE) rc = launch_tile<3, EPI_NONE, 0>(h, L, A, lda, aux, C, ldc, M, T, s, row_map);
    } else {
        set_error("launch_gemm: taps must be 1 or 3");
    }
    if (rc) return rc;
    if (prof) {
        GEM_HIP(hipEventRecord(rec.b, s));
        h->prof.recs.push_back(rec);
    }
    commit_kernel_names(h, prof ? family : -1);
    return 0;
}

}  // namespace gem
