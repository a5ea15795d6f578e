// Developer tool: register-only fp32 MFMA issue rate (what one wave per SIMD can sustain), 16x16x4 vs 32x32x2.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int NACC>
__global__ __launch_bounds__(256) void k16(float* out, int iters, float a, float b) {
    f32x4 acc[NACC];
    for (int j = 0; j < NACC; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    float av = a + threadIdx.x, bv = b;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[j], 0, 0, 0);
    }
    f32x4 s = acc[0];
    for (int j = 1; j < NACC; ++j) s += acc[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
}
template <int NACC>
__global__ __launch_bounds__(256) void k32(float* out, int iters, float a, float b) {
    f32x16 acc[NACC];
    for (int j = 0; j < NACC; ++j) for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
    float av = a + threadIdx.x, bv = b;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[j], 0, 0, 0);
    }
    float s = 0.f;
    for (int j = 0; j < NACC; ++j) for (int e = 0; e < 16; ++e) s += acc[j][e];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
// same loop with operands that differ per lane and per instruction (random bit patterns: realistic switching activity)
__device__ inline float rnd(unsigned& st) { st = st * 1664525u + 1013904223u; return (float)(int)(st >> 8) * (1.f / 8388608.f) - 1.f; }
template <int NACC>
__global__ __launch_bounds__(256) void k16r(float* out, int iters, float a, float b) {
    f32x4 acc[NACC];
    for (int j = 0; j < NACC; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    unsigned st = threadIdx.x * 7919u + blockIdx.x * 104729u + 1u;
    float av[8], bv[8];
    for (int r = 0; r < 8; ++r) { av[r] = rnd(st) * a; bv[r] = rnd(st) * 0.01f * b; }
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[r], bv[(r + j) & 7], acc[j], 0, 0, 0);
    }
    f32x4 s = acc[0];
    for (int j = 1; j < NACC; ++j) s += acc[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
}
static int g_iters = 2000, g_lds = 0, g_reps = 5;
template <typename K>
static void timeit(const char* name, K k, int grid, int threads, double flop_per_iter_wave, float* out) {
    const int iters = g_iters;
    if (g_lds) CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k, dim3(grid), dim3(threads), g_lds, 0, out, iters, 1.f, 2.f);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int r = 0; r < g_reps; ++r) hipLaunchKernelGGL(k, dim3(grid), dim3(threads), g_lds, 0, out, iters, 1.f, 2.f);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double waves = (double)grid * threads / 64;
    const double tf = flop_per_iter_wave * iters * waves * g_reps / (ms * 1e-3) / 1e12;
    printf("%-44s grid %4d x %3d: %.1f TFLOP/s (%.3f of 157.3)\n", name, grid, threads, tf, tf / 157.3);
}
int main(int argc, char** argv) {
    if (argc > 1) g_iters = atoi(argv[1]);
    if (argc > 2) g_lds = atoi(argv[2]);
    if (argc > 3) g_reps = atoi(argv[3]);
    printf("iters %d, dynamic LDS %d B, %d launches back to back\n", g_iters, g_lds, g_reps);
    float* out; CK(hipMalloc(&out, 4096 * 512 * 4));
    timeit("16x16x4, 5 acc, 1 wave/SIMD", k16<5>, 256, 256, 8.0 * 5 * 2048, out);
    timeit("16x16x4, 5 acc, 2 waves/SIMD", k16<5>, 512, 256, 8.0 * 5 * 2048, out);
    timeit("16x16x4, 16 acc, 1 wave/SIMD", k16<16>, 256, 256, 8.0 * 16 * 2048, out);
    timeit("16x16x4, 2 acc, 1 wave/SIMD", k16<2>, 256, 256, 8.0 * 2 * 2048, out);
    timeit("16x16x4, 1 acc, 1 wave/SIMD", k16<1>, 256, 256, 8.0 * 1 * 2048, out);
    timeit("32x32x2, 4 acc, 1 wave/SIMD", k32<4>, 256, 256, 4.0 * 4 * 4096, out);
    timeit("32x32x2, 4 acc, 2 waves/SIMD", k32<4>, 512, 256, 4.0 * 4 * 4096, out);
    timeit("32x32x2, 1 acc, 1 wave/SIMD", k32<1>, 256, 256, 4.0 * 1 * 4096, out);
    timeit("16x16x4, 5 acc, 1 wave/SIMD, RANDOM operands", k16r<5>, 256, 256, 8.0 * 5 * 2048, out);
    timeit("16x16x4, 16 acc, 1 wave/SIMD, RANDOM operands", k16r<16>, 256, 256, 8.0 * 16 * 2048, out);
    timeit("16x16x4, 5 acc, 2 waves/SIMD, RANDOM operands", k16r<5>, 512, 256, 8.0 * 5 * 2048, out);
    timeit("16x16x4, 5 acc, 240 WGs", k16<5>, 240, 256, 8.0 * 5 * 2048, out);
    return 0;
}
