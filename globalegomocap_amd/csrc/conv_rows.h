// conv_rows.h -- a 3-tap conv over few rows for the training step (train.hip); also included by tools/conv_rows_bench.hip
#pragma once
#include <hip/hip_runtime.h>

namespace gem {

typedef float cr_f32x4 __attribute__((ext_vector_type(4)));

// BatchNorm statistics formed by the conv's epilogue (training step, conv_rows_lds_kernel; part == nullptr: none): per 32-row tile and
// channel three fp64 sums, part[(row tile * N + n) * 3 + q], added up in tile order by the BatchNorm apply kernels (train.hip).
//   forward flavour (out == nullptr): the tile is the conv output y of a BatchNorm block: sums of y, y^2, (unused)
//   backward flavour: the tile is dOut of a BatchNorm + LeakyReLU block whose output / conv output are out / Y: the epilogue STORES
//   dz = dOut * LeakyReLU'(out) instead of dOut and leaves the sums of dz, dz * xhat, xhat (xhat = (Y - mean) * invstd)
struct CrStats {
    double* part;
    const float* out; const float* Y; const float* mean; const float* invstd;
    float slope;
};
constexpr int CR_STATS_LDS = 2 * 32 * 32 * 4;          // two staging tiles behind the KW reduction tiles

// ---- a 3-tap conv over FEW rows (the reference's batch: 640 rows), forward and backward-data ----------------------------------------
// out[r][n] = bias[n] + sum_tap sum_k A[r + tap - 1][k] W[tap][n][k]  (rows of one window only: frame r % T + tap - 1 in [0, T)).
// The optimiser's 64 x 64-tile kernel needs split-K slabs and a reduce launch to fill the chip at 640 rows (7 + 5 reduce launches per
// step), and its K loop through LDS is a chain of barriers.  Here a workgroup owns a 32 x 32 output tile and its KW waves split K
// among themselves: every wave streams its own k range of both operands straight from L2 into MFMA operand registers, no LDS and no
// barrier in the loop; the KW partial tiles meet in LDS once, are summed in wave order and leave with the bias.  One launch per
// product.  A request is a 32-wide k chunk of one tap: lane (row = lane & 31, half = lane >> 5) takes the 64 contiguous bytes
// [16 half, 16 half + 16) of its row as four dwordx4 -- whole 128-byte lines per row and request (the first cut asked for 32 bytes of a
// line per request and came back to the line three more times, by then evicted from L1) -- and feeds sixteen v_mfma_f32_32x32x2_f32
// with the k pairs (k0 + i, k0 + 16 + i).  Three requests are in flight per wave.
template <int KW, int ABLATE = 0>          // ABLATE (tools/conv_rows_bench only): 1 no MFMA, 2 no loads, 3 no A loads, 4 no B loads
__global__ __launch_bounds__(64 * KW) void conv_rows_kernel(const float* __restrict__ A, int lda, const float* __restrict__ W, const float* __restrict__ bias,
                                                           float* __restrict__ C, int ldc, int rows, int N, int K, int T) {
    typedef float f32x16 __attribute__((ext_vector_type(16)));
    __shared__ __attribute__((aligned(16))) float red[KW][32][32];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, hl = lane >> 5;
    const int m0 = blockIdx.x * 32, n0 = blockIdx.y * 32;
    const int kw = K / KW, cpt = kw / 32, n_it = 3 * cpt;          // this wave's k range per tap, 32-wide chunks per tap, requests
    const int r = m0 + li, rc = min(r, rows - 1), t0 = rc % T;
    const float* a_row = A + (size_t)rc * lda + wave * kw + 16 * hl;
    const float* b_row = W + (size_t)(n0 + li) * K + wave * kw + 16 * hl;
    const size_t b_tap = (size_t)N * K;
    bool ok[3];
#pragma unroll
    for (int tap = 0; tap < 3; ++tap) ok[tap] = r < rows && (unsigned)(t0 + tap - 1) < (unsigned)T;
    cr_f32x4 ra[3][4], rb[3][4];
    int tap_n = 0, c_n = 0;          // the next request (wave-uniform)
    auto issue = [&](cr_f32x4 (&a)[4], cr_f32x4 (&b)[4]) {
        // (a frame outside the window: the row's own address is read instead and the value dropped -- no branch around the load)
        const bool v = tap_n == 0 ? ok[0] : tap_n == 1 ? ok[1] : ok[2];
        const float* pa = a_row + (v ? (ptrdiff_t)(tap_n - 1) * lda : 0) + c_n * 32;
        const float* pb = b_row + tap_n * b_tap + c_n * 32;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            cr_f32x4 x = cr_f32x4{1.f, 1.f, 1.f, 1.f};
            if (ABLATE != 2 && ABLATE != 3) x = *reinterpret_cast<const cr_f32x4*>(pa + 4 * q);
            a[q] = v ? x : cr_f32x4{0.f, 0.f, 0.f, 0.f};
            b[q] = cr_f32x4{1.f, 1.f, 1.f, 1.f};
            if (ABLATE != 2 && ABLATE != 4) b[q] = *reinterpret_cast<const cr_f32x4*>(pb + 4 * q);
        }
        if (++c_n == cpt) { c_n = 0; if (tap_n < 2) ++tap_n; else c_n = cpt - 1; }          // (past the end: the last request again, unused)
    };
#pragma unroll
    for (int d = 0; d < 3; ++d) issue(ra[d], rb[d]);
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    for (int it = 0; it < n_it; it += 3) {          // (n_it = 3 cpt)
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            cr_f32x4 a[4], b[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) { a[q] = ra[d][q]; b[q] = rb[d][q]; }
            issue(ra[d], rb[d]);
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (ABLATE == 1) acc[(4 * q + j) & 15] += a[q][j] * b[q][j];
                    else acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q][j], b[q][j], acc, 0, 0, 0);
                }
        }
    }
    // C/D layout of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
#pragma unroll
    for (int e = 0; e < 16; ++e) red[wave][(e & 3) + 8 * (e >> 2) + 4 * hl][li] = acc[e];
    __syncthreads();
    for (int o = tid; o < 256; o += 64 * KW) {
        const int i = o >> 3, j4 = (o & 7) * 4;
        cr_f32x4 sum = *reinterpret_cast<const cr_f32x4*>(&red[0][i][j4]);
#pragma unroll
        for (int w = 1; w < KW; ++w) sum += *reinterpret_cast<const cr_f32x4*>(&red[w][i][j4]);
        if (bias) sum += *reinterpret_cast<const cr_f32x4*>(bias + n0 + j4);
        if (m0 + i < rows) *reinterpret_cast<cr_f32x4*>(C + (size_t)(m0 + i) * ldc + n0 + j4) = sum;
    }
}


// ---- the same product with the operands staged through wave-private LDS ---------------------------------------------------------
// Measured on the kernel above (tools/conv_rows_bench, 640 rows, N 512, K 256): 21 us with or without its MFMAs, 8 us without its
// loads -- a lane that loads its own MFMA operand row makes every wave-wide load touch 32 different 128-byte lines for 32 bytes
// each, and a CU's L1 hands out a line per ~4 cycles whatever is used of it (4 TB/s chip-wide instead of the ~17 TB/s of whole
// lines).  Here every load instruction covers whole lines (8 lanes x 16 bytes per row, 8 rows), goes straight to LDS
// (global_load_lds_dwordx4: no VGPR round trip) into a ring of D slots that belongs to the wave alone -- no barrier, only the
// wave's own counted s_waitcnt vmcnt -- and the lanes read their operand rows back with ds_read_b128; the 16-byte segments of a row
// are XOR-swizzled with the row number AT THE SOURCE ADDRESS (the DMA's destination is fixed: lane x 16 bytes), so that the eight lanes
// served per LDS cycle hit eight different bank groups.  Dynamic LDS: KW x D x 8 KB.
template <int KW, int D, int ABLATE = 0>
__global__ __launch_bounds__(64 * KW) void conv_rows_lds_kernel(const float* __restrict__ A, int lda, const float* __restrict__ W, const float* __restrict__ bias,
                                                               float* __restrict__ C, int ldc, int rows, int N, int K, int T, const CrStats st) {
    typedef float f32x16 __attribute__((ext_vector_type(16)));
    extern __shared__ __attribute__((aligned(16))) unsigned char cr_smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, hl = lane >> 5;
    const int m0 = blockIdx.x * 32, n0 = blockIdx.y * 32;
    const int kw = K / KW, cpt = kw / 32, n_it = 3 * cpt;          // this wave's k range per tap, 32-wide chunks per tap, requests
    unsigned char* ring = cr_smem + wave * (D * 8192);
    // staging role of this lane: row (lane >> 3) + 8 I of piece I, segment (lane & 7) ^ (lane >> 3) of that row's 128 bytes
    const int srow = lane >> 3, seg = (lane & 7) ^ srow;
    const float* w_src = W + (size_t)(n0 + srow) * K + wave * kw + 4 * seg;
    const size_t b_tap = (size_t)N * K;
    // reading role: operand row li, segments 4 hl + q
    const int r = m0 + li, t0 = min(r, rows - 1) % T;
    bool ok[3];
#pragma unroll
    for (int tap = 0; tap < 3; ++tap) ok[tap] = r < rows && (unsigned)(t0 + tap - 1) < (unsigned)T;
    int rd_off[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) rd_off[q] = li * 128 + (((4 * hl + q) ^ (li & 7)) << 4);
    // backward statistics: this thread's pieces of the block's output / conv output (it reduces row tid >> 3, columns 4 (tid & 7) ..
    // of the tile at the end) are requested now, so that they have arrived when the products are done
    cr_f32x4 e_o = cr_f32x4{0.f, 0.f, 0.f, 0.f}, e_y = e_o, e_mf = e_o, e_is = e_o;
    if (st.part && st.out && tid < 256) {
        const int i = tid >> 3, j4 = (tid & 7) * 4;
        const size_t at = (size_t)min(m0 + i, rows - 1) * ldc + n0 + j4;
        e_o = *reinterpret_cast<const cr_f32x4*>(st.out + at); e_y = *reinterpret_cast<const cr_f32x4*>(st.Y + at);
        e_mf = *reinterpret_cast<const cr_f32x4*>(st.mean + n0 + j4); e_is = *reinterpret_cast<const cr_f32x4*>(st.invstd + n0 + j4);
    }
    int tap_n = 0, c_n = 0;          // the next request (wave-uniform)
    auto issue = [&](int slot) {
        const int koff = c_n * 32;
        unsigned char* la = ring + slot * 8192;
        if (ABLATE != 2) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                // (a frame outside the window / a row beyond the batch: some valid row is staged and dropped by the reader)
                const int rr = min(max(m0 + 8 * i + srow + tap_n - 1, 0), rows - 1);
                const float* p = A + (size_t)rr * lda + wave * kw + 4 * seg + koff;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)p, (__attribute__((address_space(3))) void*)(la + i * 1024), 16, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float* p = w_src + tap_n * b_tap + (size_t)(8 * i) * K + koff;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)p, (__attribute__((address_space(3))) void*)(la + 4096 + i * 1024), 16, 0, 0);
            }
        }
        if (++c_n == cpt) { c_n = 0; if (tap_n < 2) ++tap_n; else c_n = cpt - 1; }          // (past the end: the last request again, unused)
    };
#pragma unroll
    for (int d = 0; d < D; ++d) issue(d);
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    int tap_c = 0, c_c = 0;          // the request being consumed
    for (int it = 0; it < n_it; it += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            if (it + d < n_it) {
                // this wave's DMA pieces retire in issue order: at most (D - 1) x 8 outstanding = slot d has landed
                if (ABLATE != 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((D - 1) * 8) : "memory");
                const unsigned char* la = ring + d * 8192;
                cr_f32x4 a[4], b[4];
                const bool v = tap_c == 0 ? ok[0] : tap_c == 1 ? ok[1] : ok[2];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const cr_f32x4 x = *reinterpret_cast<const cr_f32x4*>(la + rd_off[q]);
                    a[q] = v ? x : cr_f32x4{0.f, 0.f, 0.f, 0.f};
                    b[q] = *reinterpret_cast<const cr_f32x4*>(la + 4096 + rd_off[q]);
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // the slot has been read: its DMA may be overwritten
                issue(d);
                if (++c_c == cpt) { c_c = 0; ++tap_c; }
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if (ABLATE == 1) acc[(4 * q + j) & 15] += a[q][j] * b[q][j];
                        else acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q][j], b[q][j], acc, 0, 0, 0);
                    }
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the unused trailing requests have landed: the ring becomes the reduction buffer
    __syncthreads();
    float (*red)[32][32] = reinterpret_cast<float (*)[32][32]>(cr_smem);
    // C/D layout of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
#pragma unroll
    for (int e = 0; e < 16; ++e) red[wave][(e & 3) + 8 * (e >> 2) + 4 * hl][li] = acc[e];
    __syncthreads();
    for (int o = tid; o < 256; o += 64 * KW) {
        const int i = o >> 3, j4 = (o & 7) * 4;
        cr_f32x4 sum = *reinterpret_cast<const cr_f32x4*>(&red[0][i][j4]);
#pragma unroll
        for (int w = 1; w < KW; ++w) sum += *reinterpret_cast<const cr_f32x4*>(&red[w][i][j4]);
        if (bias) sum += *reinterpret_cast<const cr_f32x4*>(bias + n0 + j4);
        const bool in = m0 + i < rows;
        if (st.part) {
            float (*sv)[32][32] = reinterpret_cast<float (*)[32][32]>(cr_smem + KW * 4096);
            cr_f32x4 v0 = in ? sum : cr_f32x4{0.f, 0.f, 0.f, 0.f}, v1 = cr_f32x4{0.f, 0.f, 0.f, 0.f};
            if (st.out) {          // (KW >= 4: one pass of this loop per thread, o == tid -- the pieces requested at the start)
                cr_f32x4 o4 = e_o, y4 = e_y, mf = e_mf, is = e_is;
                if (KW < 4) {
                    const size_t at = (size_t)(in ? m0 + i : 0) * ldc + n0 + j4;
                    o4 = *reinterpret_cast<const cr_f32x4*>(st.out + at); y4 = *reinterpret_cast<const cr_f32x4*>(st.Y + at);
                    mf = *reinterpret_cast<const cr_f32x4*>(st.mean + n0 + j4); is = *reinterpret_cast<const cr_f32x4*>(st.invstd + n0 + j4);
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    v0[q] = in ? sum[q] * (o4[q] > 0.f ? 1.f : st.slope) : 0.f;
                    v1[q] = in ? (y4[q] - mf[q]) * is[q] : 0.f;
                }
                sum = v0;
            }
            *reinterpret_cast<cr_f32x4*>(&sv[0][i][j4]) = v0;
            *reinterpret_cast<cr_f32x4*>(&sv[1][i][j4]) = v1;
        }
        if (in) *reinterpret_cast<cr_f32x4*>(C + (size_t)(m0 + i) * ldc + n0 + j4) = sum;
    }
    if (st.part) {
        __syncthreads();
        if (tid < 32) {
            const float (*sv)[32][32] = reinterpret_cast<const float (*)[32][32]>(cr_smem + KW * 4096);
            double a = 0.0, b = 0.0, c = 0.0;
            if (st.out) {
#pragma unroll 8
                for (int i = 0; i < 32; ++i) { const float dz = sv[0][i][tid], xh = sv[1][i][tid]; a += dz; b += (double)dz * xh; c += xh; }
            } else {
#pragma unroll 8
                for (int i = 0; i < 32; ++i) { const float y = sv[0][i][tid]; a += (double)y; b += (double)y * y; }
            }
            double* p = st.part + ((size_t)blockIdx.x * N + n0 + tid) * 3;
            p[0] = a; p[1] = b; p[2] = c;
        }
    }
}

}  // namespace gem
