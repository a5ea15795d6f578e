// Developer tool: correctness + throughput of the bf16-activation GEMM kernel (csrc/gemm_glds.h) on the decoder's shapes.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/gemm_glds_bench tools/gemm_glds_bench.hip && tools/gemm_glds_bench
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "gemm_ring.h"

using namespace gem::glds;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

static uint16_t f2bf(float x) { uint32_t u; memcpy(&u, &x, 4); return (uint16_t)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16); }
static float bf2f(uint16_t b) { uint32_t u = (uint32_t)b << 16; float f; memcpy(&f, &u, 4); return f; }

// naive reference: one thread per output, fp32 accumulate of the same operands
template <bool F32>
__global__ void ref_kernel(const void* Av, const void* Wv, const float* bias, float* C, int M, int N, int K, int T, int taps, int lrelu) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)M * N) return;
    const int m = (int)(i / N), n = (int)(i % N);
    float acc = 0.f;
    for (int tap = 0; tap < taps; ++tap) {
        int src = m;
        if (taps == 3) { const int tt = m % T + tap - 1; if (tt < 0 || tt >= T) continue; src = m + tap - 1; }
        if (F32) {
            const float* a = (const float*)Av + (size_t)src * K;
            const float* w = (const float*)Wv + ((size_t)tap * N + n) * K;
            for (int k = 0; k < K; ++k) acc += a[k] * w[k];
        } else {
            const uint16_t* a = (const uint16_t*)Av + (size_t)src * K;
            const uint16_t* w = (const uint16_t*)Wv + ((size_t)tap * N + n) * K;
            for (int k = 0; k < K; ++k) acc += __builtin_bit_cast(float, (unsigned)a[k] << 16) * __builtin_bit_cast(float, (unsigned)w[k] << 16);
        }
    }
    acc += bias[n];
    if (lrelu) acc = acc > 0.f ? acc : acc * 0.01f;
    C[i] = acc;
}

template <bool F32, int TAPS, int EPI, int BM, int BN, bool OUT_BF16, int MF = 16, int STAGES = 2>
static double run(const char* name, int M, int N, int K, int T, int n_split, int reps) {
    constexpr int ES = F32 ? 4 : 2;
    std::vector<unsigned char> hA((size_t)M * K * ES), hW((size_t)TAPS * N * K * ES);
    std::vector<float> hb(N);
    srand(1);
    for (size_t i = 0; i < (size_t)M * K; ++i) {
        const float v = (rand() / (float)RAND_MAX) * 2.f - 1.f;
        if (F32) ((float*)hA.data())[i] = v; else ((uint16_t*)hA.data())[i] = f2bf(v);
    }
    for (size_t i = 0; i < (size_t)TAPS * N * K; ++i) {
        const float v = ((rand() / (float)RAND_MAX) * 2.f - 1.f) * 0.05f;
        if (F32) ((float*)hW.data())[i] = v; else ((uint16_t*)hW.data())[i] = f2bf(v);
    }
    for (auto& v : hb) v = (rand() / (float)RAND_MAX) - 0.5f;
    unsigned char *dA, *dW, *dZ; float *db, *dRef; void* dC;
    CK(hipMalloc(&dA, hA.size())); CK(hipMalloc(&dW, hW.size())); CK(hipMalloc(&db, N * 4)); CK(hipMalloc(&dZ, 256));
    const size_t out_elems = (size_t)M * N * (n_split > 1 ? n_split : 1);
    CK(hipMalloc(&dC, out_elems * 4)); CK(hipMalloc(&dRef, (size_t)M * N * 4));
    CK(hipMemcpy(dA, hA.data(), hA.size(), hipMemcpyHostToDevice)); CK(hipMemcpy(dW, hW.data(), hW.size(), hipMemcpyHostToDevice));
    CK(hipMemcpy(db, hb.data(), N * 4, hipMemcpyHostToDevice)); CK(hipMemset(dZ, 0, 256)); CK(hipMemset(dC, 0xFF, out_elems * 4));
    Args a{};
    a.A = dA; a.W = dW; a.bias = db; a.aux = nullptr; a.C = dC; a.zero16 = dZ; a.m_dev = nullptr; a.row_map = nullptr;
    a.lda = K; a.ldc = N; a.M = M; a.N = N; a.K = K; a.T = T;
    const int nTiles = TAPS * (K / (128 / ES));
    a.n_split = n_split; a.tiles_per_split = (nTiles + n_split - 1) / n_split; a.slab_stride = (size_t)M * N;
    auto k = gemm_glds_kernel<F32, TAPS, EPI, BM, BN, OUT_BF16, MF, STAGES>;
    constexpr int BUF = (BM + BN) * 128;
    constexpr int PRB = (BM < 128 ? BM : 128) * BN * 4;
    const size_t smem = (size_t)(STAGES * BUF > PRB ? STAGES * BUF : PRB);
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    const int grid = ((M + BM - 1) / BM) * (N / BN) * n_split;
    const int threads = (BM / 64) * (BN / 64) * 64;
    hipLaunchKernelGGL(k, dim3(grid), dim3(threads), smem, 0, a);
    CK(hipDeviceSynchronize());
    // check
    hipLaunchKernelGGL(ref_kernel<F32>, dim3((unsigned)(((size_t)M * N + 255) / 256)), dim3(256), 0, 0, dA, dW, db, dRef, M, N, K, T, TAPS,
                       (EPI == EPI_BIAS_LRELU && n_split == 1) ? 1 : 0);
    CK(hipDeviceSynchronize());
    std::vector<float> ref((size_t)M * N), got((size_t)M * N, 0.f);
    CK(hipMemcpy(ref.data(), dRef, ref.size() * 4, hipMemcpyDeviceToHost));
    if (n_split > 1) {
        std::vector<float> slabs(out_elems);
        CK(hipMemcpy(slabs.data(), dC, out_elems * 4, hipMemcpyDeviceToHost));
        for (int z = 0; z < n_split; ++z) for (size_t i = 0; i < got.size(); ++i) got[i] += slabs[(size_t)z * M * N + i];
        for (size_t i = 0; i < got.size(); ++i) got[i] += hb[i % N];
    } else if (OUT_BF16) {
        std::vector<uint16_t> o((size_t)M * N);
        CK(hipMemcpy(o.data(), dC, o.size() * 2, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < got.size(); ++i) got[i] = bf2f(o[i]);
    } else {
        CK(hipMemcpy(got.data(), dC, got.size() * 4, hipMemcpyDeviceToHost));
    }
    double maxerr = 0, maxref = 0;
    for (size_t i = 0; i < got.size(); ++i) { maxerr = fmax(maxerr, fabs((double)got[i] - ref[i])); maxref = fmax(maxref, fabs((double)ref[i])); }
    // time
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(k, dim3(grid), dim3(threads), smem, 0, a);
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k, dim3(grid), dim3(threads), smem, 0, a);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / reps, tf = 2.0 * M * N * K * TAPS / (us * 1e-6) / 1e12;
    const double peak = F32 ? 157.3 : 2500.0;
    printf("%-30s %s M %6d N %5d K %5d taps %d split %d  %dx%d mf%d st%d out %s: %8.2f us  %7.1f TFLOP/s (%.3f of %.0f)  max err %.3e (ref max %.2f)\n", name,
           F32 ? "f32 " : "bf16", M, N, K, TAPS, n_split, BM, BN, MF, STAGES, OUT_BF16 ? "bf16" : "f32 ", us, tf, tf / peak, peak, maxerr, maxref);
    CK(hipFree(dA)); CK(hipFree(dW)); CK(hipFree(db)); CK(hipFree(dZ)); CK(hipFree(dC)); CK(hipFree(dRef));
    return tf;
}

template <int EPI, bool OUT_BF16, int S, int ABL = 0>
static double run_ring(const char* name, int M, int N, int K, int n_split, int reps) {
    std::vector<uint16_t> hA((size_t)M * K), hW((size_t)N * K);
    std::vector<float> hb(N);
    srand(1);
    for (auto& v : hA) v = f2bf((rand() / (float)RAND_MAX) * 2.f - 1.f);
    for (auto& v : hW) v = f2bf(((rand() / (float)RAND_MAX) * 2.f - 1.f) * 0.05f);
    for (auto& v : hb) v = (rand() / (float)RAND_MAX) - 0.5f;
    unsigned char *dA, *dW, *dZ; float *db, *dRef; void* dC;
    CK(hipMalloc(&dA, hA.size() * 2)); CK(hipMalloc(&dW, hW.size() * 2)); CK(hipMalloc(&db, N * 4)); CK(hipMalloc(&dZ, 256));
    const size_t out_elems = (size_t)M * N * (n_split > 1 ? n_split : 1);
    CK(hipMalloc(&dC, out_elems * 4)); CK(hipMalloc(&dRef, (size_t)M * N * 4));
    CK(hipMemcpy(dA, hA.data(), hA.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dW, hW.data(), hW.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(db, hb.data(), N * 4, hipMemcpyHostToDevice)); CK(hipMemset(dZ, 0, 256)); CK(hipMemset(dC, 0xFF, out_elems * 4));
    Args a{};
    a.A = dA; a.W = dW; a.bias = db; a.aux = nullptr; a.C = dC; a.zero16 = dZ; a.m_dev = nullptr; a.row_map = nullptr;
    a.lda = K; a.ldc = N; a.M = M; a.N = N; a.K = K; a.T = 10;
    const int nTiles = K / gem::ring::BK;
    a.n_split = n_split; a.tiles_per_split = (nTiles + n_split - 1) / n_split; a.slab_stride = (size_t)M * N;
    auto k = gem::ring::gemm_ring_kernel<EPI, OUT_BF16, S, ABL>;
    const size_t smem = (size_t)S * gem::ring::STAGE;
    const int grid = ((M + gem::ring::BM - 1) / gem::ring::BM) * (N / gem::ring::BN) * n_split, nthreads = gem::ring::NT;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipLaunchKernelGGL(k, dim3(grid), dim3(nthreads), smem, 0, a);
    CK(hipDeviceSynchronize());
    hipLaunchKernelGGL(ref_kernel<false>, dim3((unsigned)(((size_t)M * N + 255) / 256)), dim3(256), 0, 0, dA, dW, db, dRef, M, N, K, 10, 1,
                       (EPI == EPI_BIAS_LRELU && n_split == 1) ? 1 : 0);
    CK(hipDeviceSynchronize());
    std::vector<float> ref((size_t)M * N), got((size_t)M * N, 0.f);
    CK(hipMemcpy(ref.data(), dRef, ref.size() * 4, hipMemcpyDeviceToHost));
    if (n_split > 1) {
        std::vector<float> slabs(out_elems);
        CK(hipMemcpy(slabs.data(), dC, out_elems * 4, hipMemcpyDeviceToHost));
        for (int z = 0; z < n_split; ++z) for (size_t i = 0; i < got.size(); ++i) got[i] += slabs[(size_t)z * M * N + i];
        for (size_t i = 0; i < got.size(); ++i) got[i] += hb[i % N];
    } else if (OUT_BF16) {
        std::vector<uint16_t> o((size_t)M * N);
        CK(hipMemcpy(o.data(), dC, o.size() * 2, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < got.size(); ++i) got[i] = bf2f(o[i]);
    } else {
        CK(hipMemcpy(got.data(), dC, got.size() * 4, hipMemcpyDeviceToHost));
    }
    if (EPI == EPI_NONE && n_split == 1) for (size_t i = 0; i < got.size(); ++i) got[i] += hb[i % N];      // (the reference always adds the bias)
    double maxerr = 0, maxref = 0;
    for (size_t i = 0; i < got.size(); ++i) { maxerr = fmax(maxerr, fabs((double)got[i] - ref[i])); maxref = fmax(maxref, fabs((double)ref[i])); }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(k, dim3(grid), dim3(nthreads), smem, 0, a);
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k, dim3(grid), dim3(nthreads), smem, 0, a);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / reps, tf = 2.0 * M * N * K / (us * 1e-6) / 1e12;
    printf("%-30s ring M %6d N %5d K %5d split %d  256x128x32 S%d out %s: %8.2f us  %7.1f TFLOP/s (%.3f of 2500)  max err %.3e (ref max %.2f)\n", name,
           M, N, K, n_split, S, OUT_BF16 ? "bf16" : "f32 ", us, tf, tf / 2500.0, maxerr, maxref);
    CK(hipFree(dA)); CK(hipFree(dW)); CK(hipFree(db)); CK(hipFree(dZ)); CK(hipFree(dC)); CK(hipFree(dRef));
    return tf;
}

int main(int argc, char** argv) {
    const int reps = argc > 1 ? atoi(argv[1]) : 50;
    const char* which = argc > 2 ? argv[2] : "all";
    const bool all = !strcmp(which, "all");
    if (all || !strcmp(which, "f32")) {
        // fp32 decoder_input products at the headline size (240 windows) and larger
        run<true, 1, EPI_BIAS, 128, 128, false, 16>("dec_in fwd 240w split 8", 240, 5120, 2048, 10, 8, reps);
        run<true, 1, EPI_BIAS, 128, 128, false, 32>("dec_in fwd 240w split 8", 240, 5120, 2048, 10, 8, reps);
        run<true, 1, EPI_BIAS, 128, 128, false, 16>("dec_in fwd 240w split 4", 240, 5120, 2048, 10, 4, reps);
        run<true, 1, EPI_NONE, 128, 128, false, 16>("dec_in bwd 240w split 16", 240, 2048, 5120, 10, 16, reps);
        run<true, 1, EPI_NONE, 128, 128, false, 16>("dec_in bwd 240w split 20", 240, 2048, 5120, 10, 20, reps);
        run<true, 1, EPI_BIAS, 128, 128, false, 16>("dec_in fwd 1536w", 1536, 5120, 2048, 10, 1, reps);
        run<true, 1, EPI_NONE, 128, 128, false, 16>("dec_in bwd 1536w split 3", 1536, 2048, 5120, 10, 3, reps);
        run<true, 1, EPI_BIAS, 128, 128, false, 16>("dec_in fwd 8192w", 8192, 5120, 2048, 10, 1, reps);
        run<true, 1, EPI_BIAS, 128, 128, false, 32>("dec_in fwd 8192w", 8192, 5120, 2048, 10, 1, reps);
        run<true, 1, EPI_NONE, 128, 128, false, 16>("dec_in bwd 8192w", 8192, 2048, 5120, 10, 1, reps);
        run<true, 3, EPI_BIAS_LRELU, 128, 128, false, 16>("conv 512->256 240w split 8", 2400, 256, 512, 10, 8, reps);
        run<true, 3, EPI_NONE, 128, 128, false, 16>("conv 256->512 240w split 4", 2400, 512, 256, 10, 4, reps);
        run<true, 3, EPI_BIAS_LRELU, 128, 128, false, 16>("conv 512->256 8192w", 81920, 256, 512, 10, 1, reps);
    }
    if (!strcmp(which, "pipe")) {
        run<false, 1, EPI_BIAS, 128, 128, true, 16, 2>("dec_in fwd 8192w", 8192, 5120, 2048, 10, 1, reps);
        run<false, 1, EPI_BIAS, 128, 128, true, 16, 3>("dec_in fwd 8192w", 8192, 5120, 2048, 10, 1, reps);
        run<false, 1, EPI_BIAS, 256, 128, true, 16, 3>("dec_in fwd 8192w", 8192, 5120, 2048, 10, 1, reps);
        run<false, 1, EPI_NONE, 128, 128, false, 16, 3>("dec_in bwd 8192w", 8192, 2048, 5120, 10, 1, reps);
        run<false, 1, EPI_NONE, 256, 128, false, 16, 3>("dec_in bwd 8192w", 8192, 2048, 5120, 10, 1, reps);
        run<false, 1, EPI_BIAS, 128, 128, true, 16, 3>("dec_in fwd 1536w", 1536, 5120, 2048, 10, 1, reps);
        run<false, 1, EPI_BIAS, 256, 128, true, 16, 3>("dec_in fwd 1536w", 1536, 5120, 2048, 10, 1, reps);
        run<false, 1, EPI_NONE, 128, 128, false, 16, 3>("dec_in bwd 1536w split 3", 1536, 2048, 5120, 10, 3, reps);
        run<false, 3, EPI_BIAS_LRELU, 128, 128, true, 16, 3>("conv 512->256 8192w", 81920, 256, 512, 10, 1, reps);
        run<false, 3, EPI_BIAS_LRELU, 256, 128, true, 16, 3>("conv 512->256 8192w", 81920, 256, 512, 10, 1, reps);
        run<false, 3, EPI_BIAS_LRELU, 128, 64, true, 16, 3>("conv 64->64 8192w", 81920, 64, 64, 10, 1, reps);
        run<true, 1, EPI_BIAS, 128, 128, false, 16, 3>("dec_in fwd 8192w", 8192, 5120, 2048, 10, 1, reps);
    }
    if (!strcmp(which, "ring")) {
        for (int M : {8192, 8196, 4098, 3072, 2048, 1536, 1024, 768}) {
            const int tiles = ((M + 255) / 256) * 20;
            const int sk = tiles >= 200 ? 1 : (256 / tiles > 4 ? 4 : 256 / tiles);
            if (sk == 1) run_ring<EPI_BIAS_LRELU, true, 6>("front fwd", M, 2560, 2048, 1, reps);
            else run_ring<EPI_BIAS_LRELU, false, 6>("front fwd", M, 2560, 2048, sk, reps);
            const int tb = ((M + 255) / 256) * 16;
            const int skb = tb >= 200 ? 1 : (256 / tb > 4 ? 4 : 256 / tb);
            run_ring<EPI_NONE, false, 6>("front bwd", M, 2048, 2560, skb, reps);
        }
        run_ring<EPI_BIAS_LRELU, true, 4>("front fwd S4", 8192, 2560, 2048, 1, reps);
        run_ring<EPI_BIAS_LRELU, true, 5>("front fwd S5", 8192, 2560, 2048, 1, reps);
        run_ring<EPI_BIAS_LRELU, false, 6>("front fwd 1536 split 1", 1536, 2560, 2048, 1, reps);
        run_ring<EPI_BIAS_LRELU, false, 6>("front fwd 1536 split 4", 1536, 2560, 2048, 4, reps);
    }
    if (!strcmp(which, "ablate")) {      // what bounds the ring kernel: drop one activity at a time (results wrong, timing only)
        run_ring<EPI_BIAS_LRELU, true, 6, 0>("full", 8192, 2560, 2048, 1, reps);
        run_ring<EPI_BIAS_LRELU, true, 6, 1>("no DMA in the loop", 8192, 2560, 2048, 1, reps);
        run_ring<EPI_BIAS_LRELU, true, 6, 2>("no fragment reads", 8192, 2560, 2048, 1, reps);
        run_ring<EPI_BIAS_LRELU, true, 6, 3>("no MFMAs", 8192, 2560, 2048, 1, reps);
        run_ring<EPI_BIAS_LRELU, true, 6, 4>("DMA only", 8192, 2560, 2048, 1, reps);
        run_ring<EPI_BIAS_LRELU, true, 6, 5>("MFMAs only", 8192, 2560, 2048, 1, reps);
        run_ring<EPI_BIAS_LRELU, true, 6, 6>("reads + MFMAs, no barrier", 8192, 2560, 2048, 1, reps);
        run_ring<EPI_BIAS_LRELU, true, 6, 5>("MFMAs only, 7680 rows", 7680, 2560, 2048, 1, reps);
    }
    if (!strcmp(which, "split")) {      // split-K choices of the composed front layer at configs[2] size (1536 windows) and in between
        for (int M : {768, 1024, 1536, 2048, 3072}) {
            run<false, 1, EPI_BIAS_LRELU, 128, 128, true, 16, 2>("front fwd", M, 2560, 2048, 10, 1, reps);
            run<false, 1, EPI_BIAS_LRELU, 128, 128, false, 16, 2>("front fwd", M, 2560, 2048, 10, 2, reps);
            run<false, 1, EPI_BIAS_LRELU, 128, 128, false, 16, 2>("front fwd", M, 2560, 2048, 10, 3, reps);
            run<false, 1, EPI_NONE, 128, 128, false, 16, 2>("front bwd", M, 2048, 2560, 10, 1, reps);
            run<false, 1, EPI_NONE, 128, 128, false, 16, 2>("front bwd", M, 2048, 2560, 10, 2, reps);
            run<false, 1, EPI_NONE, 128, 128, false, 16, 2>("front bwd", M, 2048, 2560, 10, 4, reps);
        }
    }
    if (!strcmp(which, "mid")) {        // the composed front layer below ~2000 rows: tile shape x pipeline depth x K cut (round 4)
        for (int M : {384, 768, 1536}) {
            run<false, 1, EPI_BIAS_LRELU, 128, 128, false, 16, 2>("fwd 128x128 S2", M, 2560, 2048, 10, 2, reps);
            run<false, 1, EPI_BIAS_LRELU, 128, 128, false, 16, 2>("fwd 128x128 S2", M, 2560, 2048, 10, 4, reps);
            run<false, 1, EPI_BIAS_LRELU, 128, 128, false, 16, 3>("fwd 128x128 S3", M, 2560, 2048, 10, 1, reps);
            run<false, 1, EPI_BIAS_LRELU, 128, 128, false, 16, 3>("fwd 128x128 S3", M, 2560, 2048, 10, 2, reps);
            run<false, 1, EPI_BIAS_LRELU, 128, 128, false, 16, 3>("fwd 128x128 S3", M, 2560, 2048, 10, 4, reps);
            run<false, 1, EPI_BIAS_LRELU, 64, 128, false, 16, 2>("fwd 64x128 S2", M, 2560, 2048, 10, 1, reps);
            run<false, 1, EPI_BIAS_LRELU, 64, 128, false, 16, 2>("fwd 64x128 S2", M, 2560, 2048, 10, 2, reps);
            run<false, 1, EPI_BIAS_LRELU, 64, 128, false, 16, 3>("fwd 64x128 S3", M, 2560, 2048, 10, 1, reps);
            run<false, 1, EPI_BIAS_LRELU, 64, 128, false, 16, 3>("fwd 64x128 S3", M, 2560, 2048, 10, 2, reps);
            run<false, 1, EPI_BIAS_LRELU, 64, 128, false, 16, 3>("fwd 64x128 S3", M, 2560, 2048, 10, 4, reps);
            run<false, 1, EPI_BIAS_LRELU, 64, 64, false, 16, 3>("fwd 64x64 S3", M, 2560, 2048, 10, 1, reps);
            run<false, 1, EPI_BIAS_LRELU, 64, 64, false, 16, 3>("fwd 64x64 S3", M, 2560, 2048, 10, 2, reps);
            run<false, 1, EPI_NONE, 128, 128, false, 16, 2>("bwd 128x128 S2", M, 2048, 2560, 10, 2, reps);
            run<false, 1, EPI_NONE, 128, 128, false, 16, 3>("bwd 128x128 S3", M, 2048, 2560, 10, 2, reps);
            run<false, 1, EPI_NONE, 64, 128, false, 16, 3>("bwd 64x128 S3", M, 2048, 2560, 10, 2, reps);
        }
    }
    if (!strcmp(which, "front")) {      // the composed front layer (decoder_input o conv 0) and its transpose, bf16
        run<false, 1, EPI_BIAS_LRELU, 128, 128, true, 16, 2>("front fwd 8192w", 8192, 2560, 2048, 10, 1, reps);
        run<false, 1, EPI_BIAS_LRELU, 128, 128, true, 16, 3>("front fwd 8192w", 8192, 2560, 2048, 10, 1, reps);
        run<false, 1, EPI_BIAS_LRELU, 256, 128, true, 16, 2>("front fwd 8192w", 8192, 2560, 2048, 10, 1, reps);
        run<false, 1, EPI_BIAS_LRELU, 256, 128, true, 16, 3>("front fwd 8192w", 8192, 2560, 2048, 10, 1, reps);
        run<false, 1, EPI_NONE, 128, 128, false, 16, 2>("front bwd 8192w", 8192, 2048, 2560, 10, 1, reps);
        run<false, 1, EPI_NONE, 128, 128, false, 16, 3>("front bwd 8192w", 8192, 2048, 2560, 10, 1, reps);
        run<false, 1, EPI_NONE, 256, 128, false, 16, 3>("front bwd 8192w", 8192, 2048, 2560, 10, 1, reps);
        run<false, 1, EPI_BIAS_LRELU, 128, 128, true, 16, 2>("front fwd 1536w", 1536, 2560, 2048, 10, 1, reps);
        run<false, 1, EPI_BIAS_LRELU, 128, 128, false, 16, 2>("front fwd 1536w split 3", 1536, 2560, 2048, 10, 3, reps);
        run<false, 1, EPI_BIAS_LRELU, 128, 128, false, 16, 3>("front fwd 1536w split 3", 1536, 2560, 2048, 10, 3, reps);
        run<false, 1, EPI_BIAS_LRELU, 256, 128, false, 16, 3>("front fwd 1536w split 2", 1536, 2560, 2048, 10, 2, reps);
        run<false, 1, EPI_BIAS_LRELU, 256, 128, false, 16, 3>("front fwd 1536w split 4", 1536, 2560, 2048, 10, 4, reps);
        run<false, 1, EPI_NONE, 128, 128, false, 16, 2>("front bwd 1536w split 4", 1536, 2048, 2560, 10, 4, reps);
        run<false, 1, EPI_NONE, 256, 128, false, 16, 3>("front bwd 1536w split 2", 1536, 2048, 2560, 10, 2, reps);
        run<false, 1, EPI_NONE, 256, 128, false, 16, 3>("front bwd 1536w split 4", 1536, 2048, 2560, 10, 4, reps);
    }
    if (all || !strcmp(which, "bf16")) {
        run<false, 1, EPI_BIAS, 128, 128, true, 16>("dec_in fwd 8192w", 8192, 5120, 2048, 10, 1, reps);
        run<false, 1, EPI_NONE, 128, 128, false, 16>("dec_in bwd 8192w", 8192, 2048, 5120, 10, 1, reps);
        run<false, 1, EPI_BIAS, 128, 128, true, 16>("dec_in fwd 1536w", 1536, 5120, 2048, 10, 1, reps);
        run<false, 1, EPI_NONE, 128, 128, false, 16>("dec_in bwd 1536w split 3", 1536, 2048, 5120, 10, 3, reps);
        run<false, 1, EPI_BIAS, 128, 128, true, 16>("dec_in fwd 240w split 4", 240, 5120, 2048, 10, 4, reps);
        run<false, 3, EPI_BIAS_LRELU, 128, 128, true, 16>("conv 512->256 8192w", 81920, 256, 512, 10, 1, reps);
        run<false, 3, EPI_NONE, 128, 128, true, 16>("conv 256->512 (adjoint) 8192w", 81920, 512, 256, 10, 1, reps);
        run<false, 3, EPI_BIAS_LRELU, 128, 128, true, 16>("conv 512->256 1536w", 15360, 256, 512, 10, 1, reps);
        run<false, 3, EPI_BIAS_LRELU, 128, 128, true, 16>("conv 256->128 8192w", 81920, 128, 256, 10, 1, reps);
        run<false, 3, EPI_BIAS_LRELU, 128, 64, true, 16>("conv 128->64 8192w", 81920, 64, 128, 10, 1, reps);
        run<false, 3, EPI_BIAS_LRELU, 128, 64, true, 16>("conv 64->64 8192w", 81920, 64, 64, 10, 1, reps);
        run<false, 3, EPI_BIAS, 128, 64, false, 16>("conv 64->45(64) f32 out 8192w", 81920, 64, 64, 10, 1, reps);
    }
    return 0;
}
