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
// The MFMA k-pairing is permuted (step s pairs k=s with k=16+s) so each lane's 16 A and 16 B values of
// a K-step are four contiguous 16-byte LDS reads; a sum over k does not care about the pairing.
#include "gem_internal.h"

namespace gem {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BK = 32;
constexpr int LDS_LD = BK + 4;

template <int TAPS, int EPI, int RM, int RN, int TAG>
__global__ __launch_bounds__(256) void gemm_f32_kernel(const float* __restrict__ A, int lda,
                                                       const float* __restrict__ W,
                                                       const float* __restrict__ bias,
                                                       const float* __restrict__ aux, float* __restrict__ C,
                                                       int ldc, int M, int N, int K, int T) {
    constexpr int BM = 64 * RM, BN = 64 * RN;
    constexpr int A_LD4 = BM * 8 / 256;   // float4 loads per thread for the A tile
    constexpr int B_LD4 = BN * 8 / 256;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int BUF = (BM + BN) * LDS_LD;      // floats per LDS buffer: A tile then B tile

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int kTiles = K / BK;
    const int nTiles = TAPS * kTiles;

    // ---- per-thread global load coordinates
    int a_row[A_LD4], a_t[A_LD4];
    const int c4 = (tid & 7) * 4;
#pragma unroll
    for (int i = 0; i < A_LD4; ++i) {
        int r = m0 + (tid >> 3) + 32 * i;
        a_row[i] = r;
        a_t[i] = (TAPS == 3) ? (r % T) : 0;
    }
    float4 ra[A_LD4], rb[B_LD4];

    auto load_tile = [&](int kt) {
        const int tap = (TAPS == 3) ? kt / kTiles : 0;
        const int k0 = (kt - tap * kTiles) * BK + c4;
#pragma unroll
        for (int i = 0; i < A_LD4; ++i) {
            const int r = a_row[i];
            bool ok = r < M;
            int src = r;
            if (TAPS == 3) {
                const int tt = a_t[i] + tap - 1;
                ok = ok && tt >= 0 && tt < T;
                src = r + tap - 1;
            }
            ra[i] = ok ? *reinterpret_cast<const float4*>(A + (size_t)src * lda + k0) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        const float* Wt = W + (size_t)tap * N * K;
#pragma unroll
        for (int i = 0; i < B_LD4; ++i) {
            const int n = n0 + (tid >> 3) + 32 * i;
            rb[i] = *reinterpret_cast<const float4*>(Wt + (size_t)n * K + k0);
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int i = 0; i < A_LD4; ++i)
            *reinterpret_cast<float4*>(lds + buf * BUF + ((tid >> 3) + 32 * i) * LDS_LD + c4) = ra[i];
#pragma unroll
        for (int i = 0; i < B_LD4; ++i)
            *reinterpret_cast<float4*>(lds + buf * BUF + (BM + (tid >> 3) + 32 * i) * LDS_LD + c4) = rb[i];
    };

    f32x16 acc[RM][RN];
#pragma unroll
    for (int i = 0; i < RM; ++i)
#pragma unroll
        for (int j = 0; j < RN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int fr = lane & 31, fh = lane >> 5;
    load_tile(0);
    store_tile(0);
    __syncthreads();
    int cur = 0;
    for (int kt = 0; kt < nTiles; ++kt) {
        if (kt + 1 < nTiles) load_tile(kt + 1);
        const float* as = lds + cur * BUF + (wm * 32 * RM + fr) * LDS_LD + fh * 16;
        const float* bs = lds + cur * BUF + (BM + wn * 32 * RN + fr) * LDS_LD + fh * 16;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float4 av[RM], bv[RN];
#pragma unroll
            for (int i = 0; i < RM; ++i) av[i] = *reinterpret_cast<const float4*>(as + i * 32 * LDS_LD + 4 * q);
#pragma unroll
            for (int j = 0; j < RN; ++j) bv[j] = *reinterpret_cast<const float4*>(bs + j * 32 * LDS_LD + 4 * q);
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
        if (kt + 1 < nTiles) store_tile(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }

    // ---- epilogue: C/D layout of 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
#pragma unroll
    for (int i = 0; i < RM; ++i)
#pragma unroll
        for (int j = 0; j < RN; ++j) {
            const int col = n0 + wn * 32 * RN + j * 32 + fr;
            float bv = 0.f;
            if (EPI == EPI_BIAS || EPI == EPI_BIAS_LRELU) bv = bias[col];
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = m0 + wm * 32 * RM + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * fh;
                if (row < M) {
                    float v = acc[i][j][e] + bv;
                    if (EPI == EPI_BIAS_LRELU) v = v > 0.f ? v : v * LEAKY_SLOPE;
                    if (EPI == EPI_MASK) v *= (aux[(size_t)row * ldc + col] > 0.f) ? 1.f : LEAKY_SLOPE;
                    C[(size_t)row * ldc + col] = v;
                }
            }
        }
}

template <int TAPS, int EPI, int RM, int RN, int TAG>
static int launch_one(const Layer& L, const float* A, int lda, const float* aux, float* C, int ldc, int M, int T, hipStream_t s) {
    constexpr int BM = 64 * RM, BN = 64 * RN;
    const size_t shmem = 2 * (size_t)(BM + BN) * LDS_LD * sizeof(float);
    auto k = gemm_f32_kernel<TAPS, EPI, RM, RN, TAG>;
    static bool attr_set = false;
    if (!attr_set) {
        if (shmem > 48 * 1024)
            GEM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
        attr_set = true;
    }
    dim3 grid(L.N / BN, (M + BM - 1) / BM);
    hipLaunchKernelGGL(k, grid, dim3(256), shmem, s, A, lda, L.w, L.bias, aux, C, ldc, M, L.N, L.K, T);
    GEM_HIP(hipGetLastError());
    return 0;
}

template <int TAPS, int EPI, int TAG>
static int launch_tile(const Layer& L, const float* A, int lda, const float* aux, float* C, int ldc, int M, int T, hipStream_t s) {
    // 128x128 tiles only when they still fill the chip (>= 2 workgroups per CU) and divide N
    const long big_blocks = (long)((M + 127) / 128) * (L.N / 128);
    if (L.N % 128 == 0 && big_blocks >= 512) return launch_one<TAPS, EPI, 2, 2, TAG>(L, A, lda, aux, C, ldc, M, T, s);
    return launch_one<TAPS, EPI, 1, 1, TAG>(L, A, lda, aux, C, ldc, M, T, s);
}

int launch_gemm(gem_handle* h, const Layer& L, int epi, const float* A, int lda, const float* aux, float* C, int ldc, int M,
                int T, hipStream_t s, int family) {
    if (L.K % BK != 0 || L.N % 64 != 0 || lda % 4 != 0) {
        set_error("launch_gemm: dimensions must be padded (K%32, N%64, lda%4)");
        return 1;
    }
    if (M <= 0) return 0;
    Profile::Rec rec;
    const bool prof = h->prof.on && family >= 0;
    if (prof) {
        GEM_HIP(hipEventCreate(&rec.a));
        GEM_HIP(hipEventCreate(&rec.b));
        rec.family = family;
        rec.flops = 2.0 * M * (double)L.N * L.K * L.taps;
        GEM_HIP(hipEventRecord(rec.a, s));
    }
    int rc = 1;
    if (L.taps == 1) {
        // TAG 1 = the decoder_input products (forward and backward-data): the dominant kernel gets its own symbol
        if (family == 0 && epi == EPI_BIAS) rc = launch_tile<1, EPI_BIAS, 1>(L, A, lda, aux, C, ldc, M, T, s);
        else if (epi == EPI_BIAS) rc = launch_tile<1, EPI_BIAS, 0>(L, A, lda, aux, C, ldc, M, T, s);
        else if (epi == EPI_NONE) rc = launch_tile<1, EPI_NONE, 0>(L, A, lda, aux, C, ldc, M, T, s);
        else set_error("launch_gemm: unsupported epilogue for a linear layer");
    } else if (L.taps == 3) {
        if (epi == EPI_BIAS) rc = launch_tile<3, EPI_BIAS, 0>(L, A, lda, aux, C, ldc, M, T, s);
        else if (epi == EPI_BIAS_LRELU) rc = launch_tile<3, EPI_BIAS_LRELU, 0>(L, A, lda, aux, C, ldc, M, T, s);
        else if (epi == EPI_MASK) rc = launch_tile<3, EPI_MASK, 0>(L, A, lda, aux, C, ldc, M, T, s);
        else if (epi == EPI_NONE) rc = launch_tile<3, EPI_NONE, 0>(L, A, lda, aux, C, ldc, M, T, s);
    } else {
        set_error("launch_gemm: taps must be 1 or 3");
    }
    if (rc) return rc;
    if (prof) {
        GEM_HIP(hipEventRecord(rec.b, s));
        h->prof.recs.push_back(rec);
    }
    return 0;
}

}  // namespace gem
