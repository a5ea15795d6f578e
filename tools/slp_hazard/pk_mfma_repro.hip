// Stand-alone reproducer (developer tool; DESIGN.md section 4, "packed fp32"): on gfx950 (MI355X) a packed-fp32 VALU instruction that
// selects the HIGH half of a source pair for its LOW result (op_sel) returns a wrong low result in lanes 48-63 -- the product term comes
// out as if that operand were 0 -- while another wavefront's MFMAs execute on the same SIMD.  Found as a ~1 % run-to-run difference of
// the bf16 decoder tail once the SLP vectoriser had formed v_pk_fma_f32 ... op_sel:[0,1,0] in its energy terms and two workgroups shared
// a CU (one in its matrix layers, one in its energy terms).
//
// One workgroup = 8 wavefronts: waves 0-3 (one per SIMD) loop on MFMAs, waves 4-7 repeat ONE packed instruction form on per-lane
// inputs and compare with the scalar-FMA result of the same lane.  Two workgroups per CU -> two matrix + two packed waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o pk_mfma_repro pk_mfma_repro.hip && ./pk_mfma_repro [iters] [matrix: 0 none, 1 bf16 16x16x32, 2 bf16 32x32x16, 3 f32 16x16x4]
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// form id -> the instruction under test and its per-half reference (a, b, c: float2 inputs of the lane)
#define FORMS(X)                                                                                                                   \
    X(0, "v_pk_fma_f32 %0, %1, %2, %3", fmaf(a[0], b[0], c[0]), fmaf(a[1], b[1], c[1]))                                             \
    X(1, "v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0]", fmaf(a[0], b[1], c[0]), fmaf(a[1], b[1], c[1]))                               \
    X(2, "v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0]", fmaf(a[1], b[0], c[0]), fmaf(a[1], b[1], c[1]))                               \
    X(3, "v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,1]", fmaf(a[0], b[0], c[1]), fmaf(a[1], b[1], c[1]))                               \
    X(4, "v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]", fmaf(a[0], b[0], c[0]), fmaf(a[1], b[0], c[1]))                            \
    X(5, "v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1]", fmaf(a[0], b[0], c[0]), fmaf(a[0], b[1], c[1]))                            \
    X(6, "v_pk_mul_f32 %0, %1, %2", a[0] * b[0], a[1] * b[1])                                                                        \
    X(7, "v_pk_mul_f32 %0, %1, %2 op_sel:[0,1]", a[0] * b[1], a[1] * b[1])                                                           \
    X(8, "v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]", a[0] * b[0], a[1] * b[0])                                                        \
    X(9, "v_pk_add_f32 %0, %1, %2", a[0] + b[0], a[1] + b[1])                                                                        \
    X(10, "v_pk_add_f32 %0, %1, %2 op_sel:[0,1]", a[0] + b[1], a[1] + b[1])                                                          \
    X(11, "v_pk_mov_b32 %0, %1, %2 op_sel:[1,0]", a[1], b[0])                                                                        \
    X(12, "v_pk_mov_b32 %0, %1, %2", a[0], b[1])
// Round 6: the other VALU forms of the bf16 tail's energy phase / gradient hand-over that share a data path with the packed ones or
// write 16-bit halves (candidates for the one-rounding anomaly of round 5): checked against software references
constexpr int N_PACKED = 13, N_FORMS = 17;
static const char* const EXTRA_NAMES[N_FORMS - N_PACKED] = {
    "v_cvt_pk_bf16_f32 (two floats -> one dword of bf16, RNE)", "v_mov_b64 (64-bit VALU move)",
    "v_mov_b32_dpp row_shr:1 (DPP reduction step)", "v_add_f64 (fp64 sum, against a copy made before the loop)"};

__device__ __forceinline__ unsigned bf16_rne(float f) {          // software round-to-nearest-even, NaN quieted like the hardware
    unsigned u = __builtin_bit_cast(unsigned, f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40u;
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}

template <int FORM>
__device__ __forceinline__ f32x2 apply(f32x2 a, f32x2 b, f32x2 c) {
    f32x2 d;
    if constexpr (FORM == 13) {
        unsigned r;
        asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a[0]), "v"(b[0]));
        return f32x2{__builtin_bit_cast(float, r), a[1]};
    } else if constexpr (FORM == 14) {
        asm volatile("v_mov_b64 %0, %1" : "=v"(d) : "v"(a));
        return d;
    } else if constexpr (FORM == 15) {
        float r = a[0];
        asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(r) : "v"(a[0]));
        return f32x2{r, a[1]};
    } else if constexpr (FORM == 16) {
        double x = __builtin_bit_cast(double, a), y = __builtin_bit_cast(double, b), r;
        asm volatile("v_add_f64 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y));
        return __builtin_bit_cast(f32x2, r);
    }
#define X(id, text, lo, hi)                                                                                 \
    if (FORM == id) asm volatile(text : "=v"(d) : "v"(a), "v"(b), "v"(c));
    FORMS(X)
#undef X
    return d;
}
template <int FORM>
__device__ __forceinline__ f32x2 reference(f32x2 a, f32x2 b, f32x2 c) {
    if constexpr (FORM == 13) return f32x2{__builtin_bit_cast(float, bf16_rne(a[0]) | (bf16_rne(b[0]) << 16)), a[1]};
    if constexpr (FORM == 14) return a;
    if constexpr (FORM == 15) { const float up = __shfl_up(a[0], 1, 64); return f32x2{(threadIdx.x & 15) == 0 ? a[0] : up, a[1]}; }
    if constexpr (FORM == 16) return apply<16>(a, b, c);          // (taken once, before the loop)
#define X(id, text, lo, hi) if (FORM == id) return f32x2{lo, hi};
    FORMS(X)
#undef X
    return f32x2{0.f, 0.f};
}

template <int FORM>
__global__ __launch_bounds__(512, 2) void repro(const float* __restrict__ in, const uint4* __restrict__ frag, unsigned* __restrict__ bad,
                                                unsigned long long* __restrict__ lanemask, int iters, int matrix, float* __restrict__ sink) {
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    if (wave < 4) {
        if (matrix == 0) return;
        const bf16x8 a = __builtin_bit_cast(bf16x8, frag[lane]), b = __builtin_bit_cast(bf16x8, frag[64 + lane]);
        float s = 0.f;
        if (matrix == 1) {
            f32x4 acc[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
            for (int it = 0; it < iters * 4; ++it)
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < 8; ++j) s += acc[j][0] + acc[j][3];
        } else if (matrix == 2) {
            f32x16 acc[4];
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
            for (int it = 0; it < iters * 2; ++it)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < 4; ++j) s += acc[j][0] + acc[j][15];
        } else {
            const float fa = in[lane], fb = in[64 + lane];
            f32x4 acc[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
            for (int it = 0; it < iters * 2; ++it)
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa, fb, acc[j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < 8; ++j) s += acc[j][0] + acc[j][3];
        }
        if (s == 12345.678f) sink[tid] = s;
        return;
    }
    const size_t base = ((size_t)blockIdx.x * 256 + (tid - 256)) * 6;
    const f32x2 a{in[base], in[base + 1]}, b{in[base + 2], in[base + 3]}, c{in[base + 4], in[base + 5]};
    const f32x2 ref = reference<FORM>(a, b, c);          // scalar instructions
    unsigned bad_lo = 0, bad_hi = 0;
    for (int it = 0; it < iters; ++it) {
        const f32x2 d = apply<FORM>(a, b, c);
        bad_lo += __builtin_bit_cast(unsigned, d[0]) != __builtin_bit_cast(unsigned, ref[0]);
        bad_hi += __builtin_bit_cast(unsigned, d[1]) != __builtin_bit_cast(unsigned, ref[1]);
    }
    if (bad_lo | bad_hi) { atomicAdd(bad, bad_lo); atomicAdd(bad + 1, bad_hi); atomicOr(lanemask, 1ull << lane); }
}

typedef void (*kern_t)(const float*, const uint4*, unsigned*, unsigned long long*, int, int, float*);
template <int... I> static kern_t pick(int form, std::integer_sequence<int, I...>) {
    kern_t k = nullptr;
    ((form == I ? (k = repro<I>, 0) : 0), ...);
    return k;
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 20000, blocks = 512;
    const int m_lo = argc > 2 ? atoi(argv[2]) : 0, m_hi = argc > 2 ? atoi(argv[2]) : 3;
    std::vector<float> in((size_t)blocks * 256 * 6);
    srand(1);
    for (auto& v : in) v = (rand() / (float)RAND_MAX) * 4.f - 2.f;
    std::vector<unsigned short> fr(128 * 8);
    for (auto& v : fr) { float f = (rand() / (float)RAND_MAX) * 2.f - 1.f; unsigned u; memcpy(&u, &f, 4); v = (unsigned short)(u >> 16); }
    float *d_in, *d_sink; uint4* d_fr; unsigned* d_bad; unsigned long long* d_mask;
    if (hipMalloc(&d_in, in.size() * 4) != hipSuccess) return 1;
    (void)hipMalloc(&d_fr, fr.size() * 2); (void)hipMalloc(&d_sink, 512 * 4); (void)hipMalloc(&d_bad, 8); (void)hipMalloc(&d_mask, 8);
    (void)hipMemcpy(d_in, in.data(), in.size() * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(d_fr, fr.data(), fr.size() * 2, hipMemcpyHostToDevice);
    const char* names[N_FORMS] = {
#define X(id, text, lo, hi) text,
        FORMS(X)
#undef X
        EXTRA_NAMES[0], EXTRA_NAMES[1], EXTRA_NAMES[2], EXTRA_NAMES[3]};
    const char* mnames[4] = {"no matrix waves", "v_mfma_f32_16x16x32_bf16", "v_mfma_f32_32x32x16_bf16", "v_mfma_f32_16x16x4_f32"};
    int any = 0;
    for (int matrix = m_lo; matrix <= m_hi; ++matrix) {
        printf("-- matrix waves: %s; %lld executions per form\n", mnames[matrix], (long long)blocks * 256 * iters);
        for (int form = 0; form < N_FORMS; ++form) {
            (void)hipMemset(d_bad, 0, 8); (void)hipMemset(d_mask, 0, 8);
            kern_t k = pick(form, std::make_integer_sequence<int, N_FORMS>());
            hipLaunchKernelGGL(k, dim3(blocks), dim3(512), 0, 0, d_in, d_fr, d_bad, d_mask, iters, matrix, d_sink);
            if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
            unsigned bad[2]; unsigned long long mask;
            (void)hipMemcpy(bad, d_bad, 8, hipMemcpyDeviceToHost); (void)hipMemcpy(&mask, d_mask, 8, hipMemcpyDeviceToHost);
            printf("   %-52s wrong low halves %10u, wrong high halves %10u, lanes %016llx\n", names[form], bad[0], bad[1], mask);
            any |= (bad[0] | bad[1]) != 0;
        }
    }
    printf(any ? "PACKED_FP32_HAZARD_SEEN\n" : "PACKED_FP32_CLEAN\n");
    return 0;
}
