// Developer tool: correctness + timing of the few-rows fp32 GEMM (csrc/gemm_rows.h) on the decoder_input shapes.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/gemm_rows_bench tools/gemm_rows_bench.hip && tools/gemm_rows_bench [B]
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../globalegomocap_amd/csrc/gemm_rows.h"

using namespace gem::rows;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void ref_kernel(const float* A, const int* row_map, const float* W, const float* bias, float* C, int M, int N, int K) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)M * N) return;
    const int m = (int)(i / N), n = (int)(i % N);
    const float* a = A + (size_t)(row_map ? row_map[m] : m) * K;
    const float* w = W + (size_t)n * K;
    double acc = 0.0;
    for (int k = 0; k < K; ++k) acc += (double)a[k] * w[k];
    C[i] = (float)(acc + (bias ? bias[n] : 0.f));
}

template <int S, int RT>
static void run(const char* name, int B, int N, int K, bool allow_split, int reps, int force_rb, int force_sk) {
    std::vector<float> hA((size_t)B * K), hW((size_t)N * K), hb(N);
    std::vector<int> hmap(B);
    srand(1);
    for (auto& v : hA) v = (rand() / (float)RAND_MAX) * 2.f - 1.f;
    for (auto& v : hW) v = ((rand() / (float)RAND_MAX) * 2.f - 1.f) * 0.05f;
    for (auto& v : hb) v = (rand() / (float)RAND_MAX) - 0.5f;
    for (int i = 0; i < B; ++i) hmap[i] = (i * 7 + 3) % B;          // a permutation when gcd(7, B) == 1 (any row works)
    float *dA, *dW, *db, *dC, *dRef; int *dmap, *dM; unsigned char* dZ;
    CK(hipMalloc(&dA, hA.size() * 4)); CK(hipMalloc(&dW, hW.size() * 4)); CK(hipMalloc(&db, N * 4)); CK(hipMalloc(&dZ, 256));
    CK(hipMalloc(&dmap, B * 4)); CK(hipMalloc(&dM, 4));
    CK(hipMalloc(&dC, (size_t)8 * B * N * 4)); CK(hipMalloc(&dRef, (size_t)B * N * 4));
    CK(hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dW, hW.data(), hW.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(db, hb.data(), N * 4, hipMemcpyHostToDevice)); CK(hipMemset(dZ, 0, 256));
    CK(hipMemcpy(dmap, hmap.data(), B * 4, hipMemcpyHostToDevice));
    Plan p = plan(B, N, K, RT, 256, allow_split, (size_t)8 * B * N, N);
    if (force_rb > 0) { p.n_rb = force_rb; p.n_split = force_sk; p.per = (K / BK + force_sk - 1) / force_sk; }
    if (p.n_rb == 0) { printf("%s: no plan\n", name); return; }
    Args a{};
    a.A = dA; a.W = dW; a.bias = db; a.C = dC; a.m_dev = dM; a.row_map = dmap;
    a.lda = K; a.ldc = N; a.M = B; a.N = N; a.K = K; a.n_rb = p.n_rb; a.n_split = p.n_split; a.tiles_per_split = p.per;
    a.slab_stride = (size_t)B * N;
    auto k = gemm_rows_kernel<S, RT>;
    const size_t smem = (size_t)S * Geometry<RT>::STAGE_BYTES;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    const int grid = p.n_rb * (N / BN) * p.n_split;
    printf("%s  B=%d N=%d K=%d  S=%d RT=%d: row blocks %d, K slices %d (%d k-steps each), %d workgroups, fill %.3f\n", name, B, N, K, S, RT, p.n_rb,
           p.n_split, p.per, grid, p.fill);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int Ms[] = {B, B - 11 > 0 ? B - 11 : B, (3 * B) / 4, B / 2, B / 4, 17, 1};
    std::vector<float> hC((size_t)B * N), hR((size_t)B * N), hS((size_t)8 * B * N);
    for (int M : Ms) {
        if (M < 1) continue;
        CK(hipMemcpy(dM, &M, 4, hipMemcpyHostToDevice));
        CK(hipMemset(dC, 0xFF, (size_t)8 * B * N * 4));
        hipLaunchKernelGGL(k, dim3(grid), dim3(256), smem, 0, a);
        CK(hipDeviceSynchronize());
        hipLaunchKernelGGL(ref_kernel, dim3((unsigned)(((size_t)M * N + 255) / 256)), dim3(256), 0, 0, dA, dmap, dW, p.n_split == 1 ? db : nullptr,
                           dRef, M, N, K);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(hR.data(), dRef, (size_t)M * N * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(hS.data(), dC, (size_t)p.n_split * B * N * 4, hipMemcpyDeviceToHost));
        double maxerr = 0, maxref = 0;
        for (size_t i = 0; i < (size_t)M * N; ++i) {
            float v = 0.f;
            for (int z = 0; z < p.n_split; ++z) v += hS[(size_t)z * B * N + i];
            maxerr = fmax(maxerr, fabs((double)v - hR[i]));
            maxref = fmax(maxref, fabs((double)hR[i]));
        }
        // rows past M must be untouched (0xFF pattern = NaN)
        bool clean = true;
        if (M < B) { uint32_t u; memcpy(&u, &hS[(size_t)M * N], 4); clean = u == 0xFFFFFFFFu; }
        for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k, dim3(grid), dim3(256), smem, 0, a);
        CK(hipEventRecord(e0));
        for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k, dim3(grid), dim3(256), smem, 0, a);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double us = ms * 1e3 / reps, tf = 2.0 * M * N * K / (us * 1e-6) / 1e12;
#ifdef GEM_ROWS_CLOCK
        { long long hc[6]; CK(hipMemcpyFromSymbol(hc, HIP_SYMBOL(g_rows_clock), 48));
          printf("   workgroup 8: prologue %.2f us, main loop %lld shader cycles in %.2f us = %.0f MHz, epilogue (until the stores are issued) %.2f us\n", hc[2] / 100.0, hc[0], hc[1] / 100.0, hc[0] / (hc[1] / 100.0), hc[3] / 100.0); }
#endif
        printf("   M=%3d: %.2f us  %.1f TFLOP/s (%.2f of 157.3)  max|err| %.2e (max|ref| %.2f)%s\n", M, us, tf, tf / 157.3, maxerr, maxref,
               clean ? "" : "  ROWS PAST M WRITTEN");
    }
    hipFree(dA); hipFree(dW); hipFree(db); hipFree(dC); hipFree(dRef); hipFree(dmap); hipFree(dM); hipFree(dZ);
}

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 240;
    const int reps = 200;
    const int frb = argc > 3 ? atoi(argv[2]) : 0, fsk = argc > 3 ? atoi(argv[3]) : 1;
    run<4, 5>("decoder_input forward ", B, 5120, 2048, false, reps, 0, 1);
    run<3, 5>("decoder_input forward ", B, 5120, 2048, false, reps, 0, 1);
    run<3, 8>("decoder_input backward", B, 2048, 5120, true, reps, frb, fsk);
    run<3, 8>("decoder_input backward", B, 2048, 5120, true, reps, 2, 4);
    run<3, 8>("decoder_input backward", B, 2048, 5120, true, reps, 4, 2);
    run<4, 5>("decoder_input backward", B, 2048, 5120, true, reps, 3, 2);
    return 0;
}
