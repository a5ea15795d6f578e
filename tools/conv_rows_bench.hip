// Developer tool: correctness + timing + ablations of the few-rows 3-tap conv of the training step (csrc/conv_rows.h).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/conv_rows_bench tools/conv_rows_bench.hip && tools/conv_rows_bench [reps]
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../globalegomocap_amd/csrc/conv_rows.h"

using namespace gem;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void ref_kernel(const float* A, const float* W, const float* bias, float* C, int rows, int N, int K, int T) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)rows * N) return;
    const int r = (int)(i / N), n = (int)(i % N);
    double acc = bias[n];
    for (int tap = 0; tap < 3; ++tap) {
        const int tt = r % T + tap - 1;
        if (tt < 0 || tt >= T) continue;
        const float* a = A + (size_t)(r + tap - 1) * K;
        const float* w = W + ((size_t)tap * N + n) * K;
        for (int k = 0; k < K; ++k) acc += (double)a[k] * w[k];
    }
    C[i] = (float)acc;
}

template <int KW, int ABLATE>
static float time_kernel(const float* A, const float* W, const float* b, float* C, int rows, int N, int K, int T, int reps) {
    const dim3 grid((rows + 31) / 32, N / 32);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((conv_rows_kernel<KW, ABLATE>), grid, dim3(64 * KW), 0, 0, A, K, W, b, C, N, rows, N, K, T);
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((conv_rows_kernel<KW, ABLATE>), grid, dim3(64 * KW), 0, 0, A, K, W, b, C, N, rows, N, K, T);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1e3f / reps;
}

template <int KW, int D, int ABLATE>
static float time_lds(const float* A, const float* W, const float* b, float* C, int rows, int N, int K, int T, int reps) {
    const dim3 grid((rows + 31) / 32, N / 32);
    auto k = conv_rows_lds_kernel<KW, D, ABLATE>;
    const size_t smem = (size_t)KW * D * 8192;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k, grid, dim3(64 * KW), smem, 0, A, K, W, b, C, N, rows, N, K, T, CrStats{});
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k, grid, dim3(64 * KW), smem, 0, A, K, W, b, C, N, rows, N, K, T, CrStats{});
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1e3f / reps;
}
template <int KW, int D>
static void run_lds(const float* dA, const float* dW, const float* db, float* dC, const std::vector<float>& r, int rows, int N, int K, int T, int reps) {
    if ((size_t)KW * D * 8192 > 160 * 1024) return;
    CK(hipMemset(dC, 0xFF, (size_t)rows * N * 4));
    const float t0 = time_lds<KW, D, 0>(dA, dW, db, dC, rows, N, K, T, reps);
    std::vector<float> c((size_t)rows * N);
    CK(hipMemcpy(c.data(), dC, c.size() * 4, hipMemcpyDeviceToHost));
    double md = 0;
    for (size_t i = 0; i < c.size(); ++i) md = std::max(md, (double)std::fabs(c[i] - r[i]));
    printf("    lds KW %d D %d: max err %.2e | %.2f us | no MFMA %.2f | no loads %.2f\n", KW, D, md, t0, time_lds<KW, D, 1>(dA, dW, db, dC, rows, N, K, T, reps),
           time_lds<KW, D, 2>(dA, dW, db, dC, rows, N, K, T, reps));
}

template <int KW>
static void run(int rows, int N, int K, int T, int reps) {
    if (K % (32 * KW)) return;
    std::vector<float> hA((size_t)rows * K), hW((size_t)3 * N * K), hb(N);
    srand(1);
    for (auto& v : hA) v = (rand() / (float)RAND_MAX) * 2.f - 1.f;
    for (auto& v : hW) v = ((rand() / (float)RAND_MAX) * 2.f - 1.f) * 0.05f;
    for (auto& v : hb) v = (rand() / (float)RAND_MAX) - 0.5f;
    float *dA, *dW, *db, *dC, *dR;
    CK(hipMalloc(&dA, hA.size() * 4)); CK(hipMalloc(&dW, hW.size() * 4)); CK(hipMalloc(&db, N * 4));
    CK(hipMalloc(&dC, (size_t)rows * N * 4)); CK(hipMalloc(&dR, (size_t)rows * N * 4));
    CK(hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dW, hW.data(), hW.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(db, hb.data(), N * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL((conv_rows_kernel<KW, 0>), dim3((rows + 31) / 32, N / 32), dim3(64 * KW), 0, 0, dA, K, dW, db, dC, N, rows, N, K, T);
    hipLaunchKernelGGL(ref_kernel, dim3((unsigned)(((size_t)rows * N + 255) / 256)), dim3(256), 0, 0, dA, dW, db, dR, rows, N, K, T);
    CK(hipDeviceSynchronize());
    std::vector<float> c((size_t)rows * N), r((size_t)rows * N);
    CK(hipMemcpy(c.data(), dC, c.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(r.data(), dR, r.size() * 4, hipMemcpyDeviceToHost));
    double md = 0, mr = 0;
    for (size_t i = 0; i < c.size(); ++i) { md = std::max(md, (double)std::fabs(c[i] - r[i])); mr = std::max(mr, (double)std::fabs(r[i])); }
    const double gf = 2.0 * rows * N * 3.0 * K * 1e-9;
    const float t0 = time_kernel<KW, 0>(dA, dW, db, dC, rows, N, K, T, reps);
    printf("rows %4d N %4d K %4d KW %d: max err %.2e of %.2f | %.2f us (%.1f TFLOP/s) | no MFMA %.2f | no loads %.2f | no A loads %.2f | no B loads %.2f\n", rows, N, K, KW,
           md, mr, t0, gf / t0 * 1e-3 * 1e3, time_kernel<KW, 1>(dA, dW, db, dC, rows, N, K, T, reps), time_kernel<KW, 2>(dA, dW, db, dC, rows, N, K, T, reps),
           time_kernel<KW, 3>(dA, dW, db, dC, rows, N, K, T, reps), time_kernel<KW, 4>(dA, dW, db, dC, rows, N, K, T, reps));
    run_lds<KW, 1>(dA, dW, db, dC, r, rows, N, K, T, reps);
    run_lds<KW, 2>(dA, dW, db, dC, r, rows, N, K, T, reps);
    CK(hipFree(dA)); CK(hipFree(dW)); CK(hipFree(db)); CK(hipFree(dC)); CK(hipFree(dR));
}

int main(int argc, char** argv) {
    const int reps = argc > 1 ? atoi(argv[1]) : 200;
    const int shapes[][2] = {{64, 64}, {128, 64}, {256, 128}, {512, 256}, {256, 512}, {128, 256}, {64, 128}};          // {N, K}
    for (auto& sh : shapes) {
        run<2>(640, sh[0], sh[1], 10, reps);
        run<4>(640, sh[0], sh[1], 10, reps);
        run<8>(640, sh[0], sh[1], 10, reps);
    }
    run<2>(230, 128, 192, 10, reps);          // ragged rows, three chunks per tap and wave
    run<4>(1270, 64, 384, 10, reps);          // ragged rows, three chunks per tap and wave, four waves
    return 0;
}
