// Developer tool: correctness + throughput of the one-round bf16 GEMM (csrc/gemm_big.h) on the composed front layer's shapes.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/gemm_big_bench tools/gemm_big_bench.hip && tools/gemm_big_bench [reps]
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../globalegomocap_amd/csrc/gemm_big.h"

using namespace gem;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
static uint16_t f2bf(float x) { uint32_t u; memcpy(&u, &x, 4); return (uint16_t)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16); }
static float bf2f(uint16_t b) { uint32_t u = (uint32_t)b << 16; float f; memcpy(&f, &u, 4); return f; }

__global__ void ref_kernel(const uint16_t* A, const uint16_t* W, const float* bias, const int* row_map, float* C, int M, int N, int K, int lrelu) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)M * N) return;
    const int m = (int)(i / N), n = (int)(i % N);
    const uint16_t* a = A + (size_t)(row_map ? row_map[m] : m) * K;
    const uint16_t* w = W + (size_t)n * K;
    float acc = 0.f;
    for (int k = 0; k < K; ++k) acc += __builtin_bit_cast(float, (unsigned)a[k] << 16) * __builtin_bit_cast(float, (unsigned)w[k] << 16);
    if (lrelu) { acc += bias[n]; acc = acc > 0.f ? acc : acc * 0.01f; }
    C[i] = acc;
}

template <int WM, int WN, int MB, int NB, int EPI, bool OUT_BF16>
static void run(const char* name, int M, int Mcap, int N, int K, bool gather, int reps) {
    std::vector<uint16_t> hA((size_t)Mcap * K), hW((size_t)N * K);
    std::vector<float> hb(N);
    std::vector<int> hmap(Mcap);
    srand(1);
    for (auto& v : hA) v = f2bf((rand() / (float)RAND_MAX) * 2.f - 1.f);
    for (auto& v : hW) v = f2bf(((rand() / (float)RAND_MAX) * 2.f - 1.f) * 0.05f);
    for (auto& v : hb) v = (rand() / (float)RAND_MAX) - 0.5f;
    for (int i = 0; i < Mcap; ++i) hmap[i] = (int)(((long)i * 7919) % Mcap);
    uint16_t *dA, *dW, *dZ; float *db, *dRef; void* dC; int *dMap, *dM;
    CK(hipMalloc(&dA, hA.size() * 2)); CK(hipMalloc(&dW, hW.size() * 2)); CK(hipMalloc(&db, N * 4)); CK(hipMalloc(&dZ, 256));
    CK(hipMalloc(&dC, (size_t)Mcap * N * 4)); CK(hipMalloc(&dRef, (size_t)M * N * 4)); CK(hipMalloc(&dMap, Mcap * 4)); CK(hipMalloc(&dM, 4));
    CK(hipMemcpy(dA, hA.data(), hA.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dW, hW.data(), hW.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(db, hb.data(), N * 4, hipMemcpyHostToDevice)); CK(hipMemset(dZ, 0, 256)); CK(hipMemset(dC, 0xFF, (size_t)Mcap * N * 4));
    CK(hipMemcpy(dMap, hmap.data(), Mcap * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dM, &M, 4, hipMemcpyHostToDevice));
    big::Args a{};
    a.A = dA; a.W = dW; a.bias = db; a.C = dC; a.zero16 = dZ; a.m_dev = dM; a.row_map = gather ? dMap : nullptr;
    a.lda = K; a.ldc = N; a.M = Mcap; a.N = N; a.K = K; a.m_min = 1;
    auto k = big::gemm_big_kernel<WM, WN, MB, NB, EPI, OUT_BF16>;
    constexpr int BN = 16 * NB * WN, NTH = WM * WN * 64;
    const size_t smem = 2 * (size_t)(256 + BN) * 128;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    const int grid = ((Mcap + 255) / 256) * (N / BN);
    hipLaunchKernelGGL(k, dim3(grid), dim3(NTH), smem, 0, a);
    CK(hipDeviceSynchronize());
    hipLaunchKernelGGL(ref_kernel, dim3((unsigned)(((size_t)M * N + 255) / 256)), dim3(256), 0, 0, dA, dW, db, gather ? dMap : nullptr, dRef, M, N, K,
                       EPI == big::EPI_BIAS_LRELU ? 1 : 0);
    CK(hipDeviceSynchronize());
    std::vector<float> ref((size_t)M * N), got((size_t)M * N);
    CK(hipMemcpy(ref.data(), dRef, ref.size() * 4, hipMemcpyDeviceToHost));
    if (OUT_BF16) {
        std::vector<uint16_t> o((size_t)M * N);
        CK(hipMemcpy(o.data(), dC, o.size() * 2, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < got.size(); ++i) got[i] = bf2f(o[i]);
    } else {
        CK(hipMemcpy(got.data(), dC, got.size() * 4, hipMemcpyDeviceToHost));
    }
    double maxerr = 0, maxref = 0;
    for (size_t i = 0; i < got.size(); ++i) { maxerr = fmax(maxerr, fabs((double)got[i] - ref[i])); maxref = fmax(maxref, fabs((double)ref[i])); }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(k, dim3(grid), dim3(NTH), smem, 0, a);
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k, dim3(grid), dim3(NTH), smem, 0, a);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / reps, tf = 2.0 * M * N * K / (us * 1e-6) / 1e12;
    printf("%-10s %dx%d waves of %dx%d M %5d (cap %5d) N %5d K %5d gather %d out %s: %8.2f us  %7.1f TFLOP/s (%.3f of 2500)  max err %.3e (ref max %.2f)%s\n", name, WM, WN, 16 * MB, 16 * NB, M, Mcap, N, K,
           (int)gather, OUT_BF16 ? "bf16" : "f32 ", us, tf, tf / 2500.0, maxerr, maxref, maxerr > (OUT_BF16 ? 0.02 : 1e-3) * fmax(1.0, maxref) ? "  <-- MISMATCH" : "");
    CK(hipFree(dA)); CK(hipFree(dW)); CK(hipFree(db)); CK(hipFree(dZ)); CK(hipFree(dC)); CK(hipFree(dRef)); CK(hipFree(dMap)); CK(hipFree(dM));
}

// the 128 x 128 kernel the big one replaces at large row counts (timing only), same operands
template <int EPI, bool OUT_BF16>
static void run_old(const char* name, int M, int N, int K, int reps) {
    uint16_t *dA, *dW, *dZ; float* db; void* dC;
    CK(hipMalloc(&dA, (size_t)M * K * 2)); CK(hipMalloc(&dW, (size_t)N * K * 2)); CK(hipMalloc(&db, N * 4)); CK(hipMalloc(&dZ, 256));
    CK(hipMalloc(&dC, (size_t)M * N * 4));
    CK(hipMemset(dA, 0x3c, (size_t)M * K * 2)); CK(hipMemset(dW, 0x3c, (size_t)N * K * 2)); CK(hipMemset(db, 0, N * 4)); CK(hipMemset(dZ, 0, 256));
    glds::Args a{};
    a.A = dA; a.W = dW; a.bias = db; a.aux = nullptr; a.C = dC; a.zero16 = dZ; a.m_dev = nullptr; a.row_map = nullptr;
    a.lda = K; a.ldc = N; a.M = M; a.N = N; a.K = K; a.T = 10; a.n_split = 1; a.tiles_per_split = K / 64; a.slab_stride = (size_t)M * N;
    auto k = glds::gemm_glds_kernel<false, 1, EPI, 128, 128, OUT_BF16, 16>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    const int grid = ((M + 127) / 128) * (N / 128);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(k, dim3(grid), dim3(256), 65536, 0, a);
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k, dim3(grid), dim3(256), 65536, 0, a);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / reps;
    printf("%-10s M %5d N %5d K %5d 128x128 kernel (constant data: clocks higher than on random data): %8.2f us  %.3f of 2500\n", name, M, N, K, us,
           2.0 * M * N * K / (us * 1e-6) / 2.5e15);
    CK(hipFree(dA)); CK(hipFree(dW)); CK(hipFree(db)); CK(hipFree(dZ)); CK(hipFree(dC));
}

int main(int argc, char** argv) {
    const int reps = argc > 1 ? atoi(argv[1]) : 30;
    // sixteen waves of 64 x 64 | 80, eight of 64 x 128 | 160, eight of 128 x 64 | 80, four of 128 x 128 | 160
    run<4, 4, 4, 5, big::EPI_BIAS_LRELU, true>("forward", 8192, 8192, 2560, 2048, true, reps);
    run<4, 4, 4, 4, big::EPI_NONE, false>("backward", 8192, 8192, 2048, 2560, false, reps);
    run<4, 2, 4, 10, big::EPI_BIAS_LRELU, true>("forward", 8192, 8192, 2560, 2048, true, reps);
    run<4, 2, 4, 8, big::EPI_NONE, false>("backward", 8192, 8192, 2048, 2560, false, reps);
    run<2, 4, 8, 5, big::EPI_BIAS_LRELU, true>("forward", 8192, 8192, 2560, 2048, true, reps);
    run<2, 4, 8, 4, big::EPI_NONE, false>("backward", 8192, 8192, 2048, 2560, false, reps);
    run<2, 2, 8, 10, big::EPI_BIAS_LRELU, true>("forward", 8192, 8192, 2560, 2048, true, reps);
    run<2, 2, 8, 8, big::EPI_NONE, false>("backward", 8192, 8192, 2048, 2560, false, reps);
    run<4, 2, 4, 10, big::EPI_BIAS_LRELU, true>("forward", 6001, 8192, 2560, 2048, true, reps);
    run<4, 2, 4, 8, big::EPI_NONE, false>("backward", 6001, 8192, 2048, 2560, false, reps);
    run<4, 2, 4, 8, big::EPI_NONE, false>("backward", 300, 8192, 2048, 2560, false, reps);
    if (argc > 2)
        for (int M : {2048, 3072, 4096, 5120, 6144, 8192}) {
            run_old<glds::EPI_BIAS_LRELU, true>("old fwd", M, 2560, 2048, reps);
            run<4, 4, 4, 5, big::EPI_BIAS_LRELU, true>("big fwd", M, 8192, 2560, 2048, false, reps);
        }
    return 0;
}
