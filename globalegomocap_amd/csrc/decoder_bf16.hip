// "bf16 VAE decoder / fp32 energy" (BASELINE configs[2..4]): one evaluation round with every decoder activation and
// gradient kept in bf16 in HBM, all wide products on the bf16-activation MFMA kernel of gemm_glds.h.
//
//   trial (bf16 copy written by lbfgs_advance) -> decoder_input -> h0 (bf16) -> temporal convs (bf16) -> pose X (fp32)
//   -> energy terms + dE/dX (fp32 arithmetic, energy_device.h) -> adjoint convs (bf16) -> decoder_input^T -> dE/dz (fp32)
//
// Small batches keep the fused fp32 tail kernel for the narrow layers (tail.hip; its input arrives as the wide conv's fp32
// split-K slabs, its output gradient leaves as bf16); large batches run every layer as a batched bf16 GEMM.
// Reference semantics: ConvVAE.decode_to_bodypose (SeqConvVAE.py:131-140) + total_loss.backward() (optimizer.py:226-240,
// 264-268), backward-DATA only (the VAE is frozen).
#include <cstdlib>

#include "gem_internal.h"
#include "gemm_glds.h"
#include "gemm_big.h"

namespace gem {

using glds::Args;
namespace bf16a = glds;

__global__ void f32_to_bf16_kernel(const float* __restrict__ src, uint16_t* __restrict__ dst, size_t n4) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    const bf16a::f32x4 v = reinterpret_cast<const bf16a::f32x4*>(src)[i];
    reinterpret_cast<bf16a::u32x2*>(dst)[i] = bf16a::u32x2{bf16a::pack_bf16(v[0], v[1]), bf16a::pack_bf16(v[2], v[3])};
}
int launch_f32_to_bf16(const float* src, uint16_t* dst, size_t n, hipStream_t s) {      // n % 4 == 0 (padded rows)
    const size_t n4 = n / 4;
    if (!n4) return 0;
    hipLaunchKernelGGL(f32_to_bf16_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, src, dst, n4);
    GEM_HIP(hipGetLastError());
    return 0;
}

// hi = bf16(w), lo = bf16(w - float(hi)): the two images of the split-bf16 ("bf16x3") mode, for weights composed on the device
__global__ void f32_split_bf16_kernel(const float* __restrict__ src, uint16_t* __restrict__ hi, uint16_t* __restrict__ lo, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float v = src[i];
    const unsigned int h2 = bf16a::pack_bf16(v, 0.f) & 0xFFFFu;
    hi[i] = (uint16_t)h2;
    lo[i] = (uint16_t)(bf16a::pack_bf16(v - __builtin_bit_cast(float, h2 << 16), 0.f) & 0xFFFFu);
}
int launch_f32_split_bf16(const float* src, uint16_t* hi, uint16_t* lo, size_t n, hipStream_t s) {
    if (!n) return 0;
    hipLaunchKernelGGL(f32_split_bf16_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, src, hi, lo, n);
    GEM_HIP(hipGetLastError());
    return 0;
}

// sums fp32 split-K slabs, applies bias / LeakyReLU, writes bf16 (the fp32-output reduce is splitk_reduce_kernel)
template <int EPI>
__global__ __launch_bounds__(256) void splitk_reduce_bf16_kernel(const float* __restrict__ slabs, int nslab, size_t slab_stride,
                                                                 const float* __restrict__ bias, uint16_t* __restrict__ C, int M, int N,
                                                                 int ldc, const int* __restrict__ m_dev) {
    if (m_dev) M = *m_dev;
    const int n4 = N / 4;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)M * n4) return;
    const int row = (int)(i / n4), c = (int)(i - (size_t)row * n4) * 4;
    const size_t off = (size_t)row * ldc + c;
    bf16a::f32x4 v = *reinterpret_cast<const bf16a::f32x4*>(slabs + off);
    for (int z = 1; z < nslab; ++z) v += *reinterpret_cast<const bf16a::f32x4*>(slabs + (size_t)z * slab_stride + off);
    if (EPI == EPI_BIAS || EPI == EPI_BIAS_LRELU) v += *reinterpret_cast<const bf16a::f32x4*>(bias + c);
    if (EPI == EPI_BIAS_LRELU) {
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = v[q] > 0.f ? v[q] : v[q] * LEAKY_SLOPE;
    }
    *reinterpret_cast<bf16a::u32x2*>(C + off) = bf16a::u32x2{bf16a::pack_bf16(v[0], v[1]), bf16a::pack_bf16(v[2], v[3])};
}

template <int TAPS, int EPI, int BN, bool OUT_BF16>
static int launch_k(gem_handle* h, const Args& a, int grid, hipStream_t s) {
    constexpr int BM = 128;
    auto k = glds::gemm_glds_kernel<false, TAPS, EPI, BM, BN, OUT_BF16, 16>;
    constexpr int BUF = (BM + BN) * 128;
    constexpr size_t smem = (size_t)(2 * BUF > BM * BN * 4 ? 2 * BUF : BM * BN * 4);
    static PerDeviceOnce once;
    if (once.need(h->cfg.device))
        GEM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    note_kernel(h, reinterpret_cast<const void*>(k));
    hipLaunchKernelGGL(k, dim3(grid), dim3((BM / 64) * (BN / 64) * 64), smem, s, a);
    GEM_HIP(hipGetLastError());
    return 0;
}

template <int TAPS, int EPI>
static int launch_bn(gem_handle* h, const Args& a, int grid, bool bn128, bool out_bf16, hipStream_t s) {
    if (bn128) return out_bf16 ? launch_k<TAPS, EPI, 128, true>(h, a, grid, s) : launch_k<TAPS, EPI, 128, false>(h, a, grid, s);
    return out_bf16 ? launch_k<TAPS, EPI, 64, true>(h, a, grid, s) : launch_k<TAPS, EPI, 64, false>(h, a, grid, s);
}

// The composed front layer at large batch: the one-round kernel of gemm_big.h (256 x 256 / 256 x 320 tiles, 16 waves).  From
// BIG_MIN_ROWS rows on it beats the 128 x 128 kernel (tools/gemm_big_bench: 78 vs ~105 us at 8192 rows, equal near 5500); in
// the evaluation rounds the row count lives on the device, so BOTH kernels are enqueued and each looks at the count itself:
// the big one returns below the threshold, the small one (Args::m_max) at or above it.  Same K order and accumulation as the
// small kernel: the results do not depend on which of the two ran.
constexpr int BIG_MIN_ROWS = 5376;
static bool big_fits(const Layer& L, int epi, bool out_bf16) {
    if (L.taps != 1 || L.K % 64 != 0 || !L.wb_hi) return false;
    if (epi == EPI_BIAS_LRELU && out_bf16) return L.N % 320 == 0;
    if (epi == EPI_NONE && !out_bf16) return L.N % 256 == 0;
    return false;
}
static int launch_big(gem_handle* h, const Layer& L, int epi, const uint16_t* A, int lda, void* C, int ldc, int M, hipStream_t s,
                      const int* row_map, int m_min) {
    Workspace& w = h->ws;
    big::Args a{};
    a.A = A; a.W = L.wb_hi; a.bias = L.bias; a.C = C; a.zero16 = w.zero16;
    a.m_dev = w.dyn ? w.n_active : nullptr; a.row_map = row_map;
    a.lda = lda; a.ldc = ldc; a.M = M; a.N = L.N; a.K = L.K; a.m_min = m_min;
    static PerDeviceOnce once;
    if (once.need(h->cfg.device)) {
        GEM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(big::gemm_big_kernel<4, 4, 4, 5, big::EPI_BIAS_LRELU, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        GEM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(big::gemm_big_kernel<4, 4, 4, 4, big::EPI_NONE, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    }
    const int n_mt = (M + 255) / 256;
    if (epi == EPI_BIAS_LRELU) {
        note_kernel(h, reinterpret_cast<const void*>(big::gemm_big_kernel<4, 4, 4, 5, big::EPI_BIAS_LRELU, true>));
        hipLaunchKernelGGL((big::gemm_big_kernel<4, 4, 4, 5, big::EPI_BIAS_LRELU, true>), dim3(n_mt * (L.N / 320)), dim3(1024), 2 * (256 + 320) * 128, s, a);
    } else {
        note_kernel(h, reinterpret_cast<const void*>(big::gemm_big_kernel<4, 4, 4, 4, big::EPI_NONE, false>));
        hipLaunchKernelGGL((big::gemm_big_kernel<4, 4, 4, 4, big::EPI_NONE, false>), dim3(n_mt * (L.N / 256)), dim3(1024), 2 * (256 + 256) * 128, s, a);
    }
    GEM_HIP(hipGetLastError());
    return 0;
}

// Mid-size batches in the evaluation rounds (round 5): the same 256 x 256 tile with the K range CUT so that the launch is one round of
// ~one workgroup per CU -- 1536 rows: 6 x 10 tiles x 4 slices (forward), 6 x 8 x 5 (backward) = 240 workgroups of eight 64-deep K-steps
// -- writing raw fp32 slabs that the consumer sums anyway (the tail while staging, lbfgs_advance while reading its gradient).  128
// FLOP per byte moved L2 -> LDS instead of the 128 x 128 tile's 64: the fill rate (~30 B/clk/CU) that bounds the small kernel at these
// sizes allows twice the rate.  Only with a device row count and a consumer that takes slabs; the two-lane mode (whose bitwise test
// needs a batch and its halves to be cut alike) keeps the old kernels.
constexpr int BIG_SPLIT_MIN_ROWS = 1024;
static bool launch_big_split(gem_handle* h, const Layer& L, const uint16_t* A, int lda, int ldc, int M, hipStream_t s, const int* row_map,
                             int m_max, int* n_split_out) {
    Workspace& w = h->ws;
    const int tiles = ((M + 255) / 256) * (L.N / 256), k_tiles = L.K / 64;
    int sk = h->n_cu / tiles;
    if (const char* fs = dev_env("GEM_BIG_SPLIT_SK")) sk = atoi(fs);          // developer override (A/B runs)
    if (sk > k_tiles / 4) sk = k_tiles / 4;
    if (sk > 8) sk = 8;
    while (sk > 1 && (size_t)sk * M * ldc > w.splitk_elems) --sk;
    if (sk < 2) return false;
    big::Args a{};
    a.A = A; a.W = L.wb_hi; a.bias = nullptr; a.C = w.splitk; a.zero16 = w.zero16;
    a.m_dev = w.n_active; a.row_map = row_map;
    a.lda = lda; a.ldc = ldc; a.M = M; a.N = L.N; a.K = L.K; a.m_min = 1; a.m_max = m_max;
    a.tiles_per_split = (k_tiles + sk - 1) / sk;
    a.n_split = (k_tiles + a.tiles_per_split - 1) / a.tiles_per_split;
    a.slab_stride = (size_t)M * ldc;
    static PerDeviceOnce once;
    if (once.need(h->cfg.device))
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(big::gemm_big_kernel<4, 4, 4, 4, big::EPI_NONE, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    note_kernel(h, reinterpret_cast<const void*>(big::gemm_big_kernel<4, 4, 4, 4, big::EPI_NONE, false>));
    hipLaunchKernelGGL((big::gemm_big_kernel<4, 4, 4, 4, big::EPI_NONE, false>), dim3(tiles * a.n_split), dim3(1024), 2 * (256 + 256) * 128, s, a);
    if (hipGetLastError() != hipSuccess) return false;
    *n_split_out = a.n_split;
    return true;
}

// C = epi(conv/linear(A)) with bf16 operands.  `out_bf16`: activation / gradient for the next bf16 layer; otherwise fp32.
// allow_split: small row counts cut K over several workgroups (fp32 slabs in ws.splitk); with `defer` the slabs are left to
// the consumer (described in ws.deferred), otherwise a reduce pass applies the epilogue.
static int gemm_bf16a(gem_handle* h, const Layer& L, int epi, const uint16_t* A, int lda, const uint16_t* aux, void* C, bool out_bf16,
                      int ldc, int M, hipStream_t s, int family, const int* row_map, bool allow_split, bool defer) {
    Workspace& w = h->ws;
    if (M <= 0) return 0;
    if (L.K % 64 != 0 || L.N % 64 != 0 || !L.wb_hi) { set_error("gemm_bf16a: layer is not padded to 64 or has no bf16 weights"); return 1; }
    w.deferred = SlabSrc{};
    // developer switch (round 6 A/B): the forward front product as 128 x 64 tiles (1) or 128 x 128 tiles (2) over the WHOLE of K, so
    // that it hands the tail one bf16 activation instead of fp32 split-K slabs
    const char* ffm = (family == 0 && out_bf16 && epi == EPI_BIAS_LRELU && L.taps == 1) ? dev_env("GEM_FRONT_FWD_MODE") : nullptr;
    if (ffm && ffm[0] != '0') allow_split = false;
    const bool bn128 = L.N % 128 == 0 && !(ffm && ffm[0] == '1');
    const int BN = bn128 ? 128 : 64;
    const int tiles = ((M + 127) / 128) * (L.N / BN);
    const int k_tiles = L.taps * (L.K / 64);
    Args a{};
    a.A = A; a.W = L.wb_hi; a.bias = L.bias; a.aux = aux; a.C = C; a.zero16 = w.zero16;
    a.m_dev = w.dyn ? w.n_active + (L.taps == 3 ? 1 : 0) : nullptr;
    a.row_map = row_map;
    a.lda = lda; a.ldc = ldc; a.M = M; a.N = L.N; a.K = L.K; a.T = h->T;
    a.n_split = 1; a.tiles_per_split = k_tiles; a.slab_stride = (size_t)M * ldc;
    // the front products at large batch: gemm_big.h takes the launches with >= BIG_MIN_ROWS rows (family 0 only)
    static const bool no_big = dev_env("GEM_NO_BIG_GEMM") != nullptr;
    static const int big_min = dev_env("GEM_BIG_MIN") ? atoi(dev_env("GEM_BIG_MIN")) : BIG_MIN_ROWS;          // developer override (A/B runs)
    const bool use_big = family == 0 && !no_big && M >= big_min && big_fits(L, epi, out_bf16);
    if (use_big) a.m_max = big_min;
    // mid-size rounds: the K-cut one-round kernel (see launch_big_split); the consumer takes the slabs
    // MEASURED AND NOT ADOPTED (round 5, tools/r05_exp5.sh, windows/s with the 128 x 128 kernel | with this): 1536 windows 182.7 | 175.6 k,
    // 2052: 180.4 | 186.4 k, 3072: 229.5 | 227.4 k, 4092: 235.1 | 232.0 k, 768: 123.6 | 114.4 k -- the four-pass fp32 epilogue of a 256 x 256
    // tile and twice the slab traffic eat what the better FLOP-per-byte ratio gives.  Off unless GEM_DEV=1 GEM_BIG_SPLIT=1.
    static const bool no_big_split = dev_env("GEM_BIG_SPLIT") == nullptr;
    static const int big_split_min = dev_env("GEM_BIG_SPLIT_MIN") ? atoi(dev_env("GEM_BIG_SPLIT_MIN")) : BIG_SPLIT_MIN_ROWS;
    if (family == 0 && !no_big && !no_big_split && !use_big && h->lanes_min == 0 && w.dyn && defer && allow_split && L.taps == 1 && M >= big_split_min &&
        L.N % 256 == 0 && L.K % 64 == 0 && (epi == EPI_NONE || epi == EPI_BIAS_LRELU)) {
        Profile::Rec rec2;
        const bool prof2 = h->prof.on;
        if (prof2) {
            GEM_HIP(hipEventCreate(&rec2.a)); GEM_HIP(hipEventCreate(&rec2.b));
            rec2.family = family; rec2.flops = 2.0 * M * (double)L.N * L.K;
            rec2.log_idx = w.cur_log; rec2.flops_per_window = 2.0 * (double)L.N * L.K;
            GEM_HIP(hipEventRecord(rec2.a, s));
        }
        int ns = 0;
        if (launch_big_split(h, L, A, lda, ldc, M, s, row_map, 0, &ns)) {
            SlabSrc& d = w.deferred;
            d.base = w.splitk; d.nslab = ns; d.stride = (size_t)M * ldc; d.dyn_W = 0; d.m_dev = w.n_active;
            if (prof2) { GEM_HIP(hipEventRecord(rec2.b, s)); h->prof.recs.push_back(rec2); }
            commit_kernel_names(h, prof2 ? family : -1);
            return 0;
        }
    }
    if (allow_split && epi != EPI_MASK) {
        // fill the chip: about two workgroups per CU (the number co-resident with this kernel's 64 KB of LDS; measured at 768 ..
        // 3072 rows, tools/gemm_glds_bench split: the best cut of every shape) while every slice keeps >= 4 k-tiles and the slabs fit
        int sk = (2 * h->n_cu) / tiles;
        if (const char* fs = dev_env("GEM_BF16_SK")) sk = atoi(fs);          // developer override (A/B runs)
        if (sk > k_tiles / 4) sk = k_tiles / 4;
        if (sk > 8) sk = 8;
        while (sk > 1 && (size_t)sk * a.slab_stride > w.splitk_elems) --sk;
        if (sk > 1) {
            a.tiles_per_split = (k_tiles + sk - 1) / sk;
            a.n_split = (k_tiles + a.tiles_per_split - 1) / a.tiles_per_split;
            a.C = w.splitk;
        }
    }
    Profile::Rec rec;
    const bool prof = h->prof.on && family >= 0;
    if (prof) {
        GEM_HIP(hipEventCreate(&rec.a)); GEM_HIP(hipEventCreate(&rec.b));
        rec.family = family;
        rec.flops = 2.0 * M * (double)L.N * L.K * L.taps;
        if (w.dyn && L.taps == 1) { rec.log_idx = w.cur_log; rec.flops_per_window = 2.0 * (double)L.N * L.K; }
        GEM_HIP(hipEventRecord(rec.a, s));
    }
    const int grid = tiles * a.n_split;
    const bool kernel_bf16 = out_bf16 && a.n_split == 1;
    int rc = 1;
    if (use_big) {
        if (a.n_split != 1) { set_error("gemm_bf16a: the one-round kernel's row range must not be cut along K"); return 1; }
        if (launch_big(h, L, epi, A, lda, C, ldc, M, s, row_map, w.dyn ? big_min : 1)) return 1;
    }
    if (use_big && !w.dyn) {
        rc = 0;                    // the row count is known here: the big kernel alone
    } else
    if (L.taps == 1) {
        if (epi == EPI_BIAS) rc = launch_bn<1, EPI_BIAS>(h, a, grid, bn128, kernel_bf16, s);
        else if (epi == EPI_NONE) rc = launch_bn<1, EPI_NONE>(h, a, grid, bn128, kernel_bf16, s);
        else if (epi == EPI_BIAS_LRELU) rc = launch_bn<1, EPI_BIAS_LRELU>(h, a, grid, bn128, kernel_bf16, s);
        else set_error("gemm_bf16a: unsupported epilogue for a linear layer");
    } else if (L.taps == 3) {
        if (epi == EPI_BIAS) rc = launch_bn<3, EPI_BIAS>(h, a, grid, bn128, kernel_bf16, s);
        else if (epi == EPI_BIAS_LRELU) rc = launch_bn<3, EPI_BIAS_LRELU>(h, a, grid, bn128, kernel_bf16, s);
        else if (epi == EPI_MASK) rc = launch_bn<3, EPI_MASK>(h, a, grid, bn128, kernel_bf16, s);
        else if (epi == EPI_NONE) rc = launch_bn<3, EPI_NONE>(h, a, grid, bn128, kernel_bf16, s);
    } else {
        set_error("gemm_bf16a: taps must be 1 or 3");
    }
    if (rc) return rc;
    if (a.n_split > 1) {
        if (defer) {
            SlabSrc& d = w.deferred;
            d.base = w.splitk; d.nslab = a.n_split; d.stride = a.slab_stride; d.dyn_W = 0; d.m_dev = a.m_dev;
        } else if (out_bf16) {
            const size_t n4 = (size_t)M * (L.N / 4);
            const dim3 grid_r((unsigned)((n4 + 255) / 256));
            uint16_t* Cb = static_cast<uint16_t*>(C);
            if (epi == EPI_BIAS) hipLaunchKernelGGL(splitk_reduce_bf16_kernel<EPI_BIAS>, grid_r, dim3(256), 0, s, w.splitk, a.n_split, a.slab_stride, L.bias, Cb, M, L.N, ldc, a.m_dev);
            else if (epi == EPI_BIAS_LRELU) hipLaunchKernelGGL(splitk_reduce_bf16_kernel<EPI_BIAS_LRELU>, grid_r, dim3(256), 0, s, w.splitk, a.n_split, a.slab_stride, L.bias, Cb, M, L.N, ldc, a.m_dev);
            else hipLaunchKernelGGL(splitk_reduce_bf16_kernel<EPI_NONE>, grid_r, dim3(256), 0, s, w.splitk, a.n_split, a.slab_stride, L.bias, Cb, M, L.N, ldc, a.m_dev);
            GEM_HIP(hipGetLastError());
        } else {
            if (launch_splitk_reduce(h, epi, a.n_split, a.slab_stride, L.bias, nullptr, static_cast<float*>(C), M, L.N, ldc, a.m_dev, s)) return 1;
        }
    }
    if (prof) { GEM_HIP(hipEventRecord(rec.b, s)); h->prof.recs.push_back(rec); }
    commit_kernel_names(h, prof ? family : -1);
    return 0;
}

// The rounds of a stage need no compact_kernel launch when every kernel of a round addresses its rows through perm / slot_of and
// nothing depends on the ORDER of the slots: the composed front products (rows gathered through perm) + the fused bf16 tail (windows
// independent of their position in a workgroup: test_bf16_tail_row_tile_variants_compute_the_same) + lbfgs_advance (slot_of).  The
// batched narrow layers (taps = 3) read n_active[1] = rows and are not covered.
bool bf16_rounds_take_slots_atomically(const gem_handle* h, int stage, int B) {
    const StageNet& net = h->net[stage];
    // (no cap on B: one same-address atomic per window and round is spread over the advance kernel's duration -- 8192 windows: 289.0 k
    // against 282.3 k windows/s with compact_kernel's 12 us single-workgroup scan per round; 6144: +0.4 %)
    (void)B;
    if (h->precision != GEM_PRECISION_BF16 || dev_env("GEM_NO_ATOMIC_COMPACT")) return false;
    const char* t16_env = dev_env("GEM_TAIL16");
    if (!net.tb_stream || net.tail_start != 1 || dev_env("GEM_BATCHED_NARROW") || (t16_env && t16_env[0] == '0')) return false;
    return net.front.wb_hi && net.dec.size() > 1 && !dev_env("GEM_NO_FRONT_BF16");
}

// One evaluation in the bf16 decoder mode: decodes ws.trial_b, leaves the pose in ws.dec_act.back() (fp32), the energies in
// ws.f / ws.parts and dE/dz in ws.dz (or in ws.grad_slab for lbfgs_advance, in the rounds).
int evaluate_bf16(gem_handle* h, int stage, int B, const EnergyArgs& ea_in, hipStream_t s, bool forward_only) {
    StageNet& net = h->net[stage];
    Workspace& w = h->ws;
    const int T = h->T, rows = B * T, n_dec = (int)net.dec.size();
    static const bool force_tail = dev_env("GEM_FORCE_TAIL") != nullptr;
    const int tail_g = T <= 16 ? 16 / T : 1;
    const int tail_wgs = (B + tail_g - 1) / tail_g;
    // Three ways to run the narrow layers + energies: the bf16 multi-window tail (tail_bf16.hip: 1 .. 8 windows per workgroup, one
    // launch), the fp32 one-window tail (tail.hip; GEM_TAIL16=0 or networks the bf16 tail does not cover), or batched bf16 GEMMs + the
    // stand-alone energy kernel (networks the tails do not cover).  GEM_TAIL16=1 / 0 forces / forbids the first (GEM_DEV=1).
    const char* t16_env = dev_env("GEM_TAIL16");       // (read per call: the tests flip it inside one process)
    const bool batched_narrow = dev_env("GEM_BATCHED_NARROW") != nullptr;      // neither tail: every layer a batched GEMM (A/B runs, tests)
    // (round 3 sent batches below 256 windows to the fp32 one-window tail; with ONE row tile per workgroup -- one window of ten frames,
    // every window its own CU like the fp32 tail, round 4 -- the bf16 tail wins at every size: 60 / 120 / 240 windows 14.9 / 26.1 /
    // 47.8 k windows/s against 11.1 / 20.6 / 38.7 k)
    const int t16_min = 1;
    const bool use_tail16 = net.tb_stream && net.tail_start >= 1 && !batched_narrow && !(t16_env && t16_env[0] == '0') &&
                            (B >= t16_min || (t16_env && t16_env[0] == '1'));
    const bool use_tail = !use_tail16 && !batched_narrow && net.tail_start >= 1 && (tail_wgs <= 5 * h->n_cu || force_tail);
    const int* perm = w.dyn ? w.perm : nullptr;
    if (w.next_count && !(use_tail16 && net.front.wb_hi && net.tail_start == 1)) {
        set_error("evaluate_bf16: slots are handed out by lbfgs_advance but this evaluation does not run front products + fused tail");
        return 1;
    }
    w.grad_slab = SlabSrc{};
    // decoder_input o conv 0 as ONE product where the weights were composed (compose_front in gem_api.hip), else
    // decoder_input: [B, Dp] x [Dp, T*topp] -> h0 [B*T, topp] bf16 (rows of finished windows are skipped through perm)
    static const bool no_front = dev_env("GEM_NO_FRONT_BF16") != nullptr;          // developer override (A/B runs)
    const bool front = net.front.wb_hi && n_dec > 1 && !no_front && ((!use_tail && !use_tail16) || net.tail_start == 1);
    const uint16_t* in = w.h0_b;
    EnergyArgs ea = ea_in;
    int back_from;                     // first layer of the batched backward chain
    const uint16_t* gin;
    if (!front && gemm_bf16a(h, net.dec_in, EPI_BIAS, w.trial_b, h->Dp, nullptr, w.h0_b, true, net.dec_in.N, B, s, 0, perm, true, false)) return 1;
    if (use_tail16) {
        const int st = net.tail_start;
        SlabSrc in_slab;
        // the layer in front of the tail: in the rounds its fp32 split-K slabs (if it was cut) go straight to the tail, which sums
        // them and applies bias + LeakyReLU while staging; otherwise it writes the bf16 activation itself
        if (front) {
            if (gemm_bf16a(h, net.front, EPI_BIAS_LRELU, w.trial_b, h->Dp, nullptr, w.dec_act_b[0], true, net.front.N, B, s, 0, perm, true, w.dyn))
                return 1;
            in_slab = w.deferred;
        }
        for (int i = 0; i < st && !front; ++i) {
            const bool last_wide = i == st - 1;
            if (gemm_bf16a(h, net.dec[i], EPI_BIAS_LRELU, in, net.dec[i].K, nullptr, w.dec_act_b[i], true, net.dec[i].N, rows, s, -1, nullptr,
                           true, last_wide && w.dyn)) return 1;
            if (last_wide) in_slab = w.deferred;
            in = w.dec_act_b[i];
        }
        TailB16Args ta;
        const size_t tb_lds = plan_tail_bf16(net.dec, st, T, h->J, &ta, tail_bf16_row_tiles(h, B, T));
        if (!tb_lds) { set_error("evaluate_bf16: the bf16 tail's LDS plan failed"); return 1; }
        ta.B = B; ta.forward_only = forward_only ? 1 : 0; ta.dbg_ts = nullptr;
        ta.in_slab = in_slab; ta.in_bias = front ? net.front.bias : net.dec[st - 1].bias;
        ta.in_bias_ld = front ? net.dec[0].N : 0;
        for (int i = 0; i < ta.n; ++i) {
            const Layer& f = net.dec[st + i];
            const Layer& g = net.dec_bwd[st + i];
            ta.fwd[i] = TailB16Layer{f.K, f.N, f.bias};
            ta.bwd[i] = TailB16Layer{g.K, g.N, nullptr};
        }
        ta.a_in_b = w.dec_act_b[st - 1]; ta.g_out_b = w.dec_grad_b[st];
        ta.Xp = (w.dyn && !forward_only) ? nullptr : w.dec_act.back();
        ta.wstream = net.tb_stream; ta.steps_f = net.tb_steps_f; ta.steps_total = net.tb_steps_f + net.tb_steps_b;
        ta.e = ea;
        if (launch_tail_bf16(h, ta, tb_lds, s)) return 1;
        if (forward_only) return 0;
        if (record_mid(h, s)) return 1;
        back_from = st - 1;
        gin = w.dec_grad_b[st];
    } else
    if (use_tail) {
        const int st = net.tail_start;
        SlabSrc in_slab;
        if (front) {                   // feeds the fp32 tail: fp32 slabs in the rounds, a finished fp32 matrix otherwise
            if (gemm_bf16a(h, net.front, EPI_BIAS_LRELU, w.trial_b, h->Dp, nullptr, w.dec_act[0], false, net.front.N, B, s, 0, perm, true, w.dyn))
                return 1;
            in_slab = w.deferred;
        }
        for (int i = 0; i < st && !front; ++i) {
            const bool last_wide = i == st - 1;
            if (last_wide) {           // feeds the fp32 tail: fp32 slabs in the rounds, a finished fp32 matrix otherwise
                if (gemm_bf16a(h, net.dec[i], EPI_BIAS_LRELU, in, net.dec[i].K, nullptr, w.dec_act[i], false, net.dec[i].N, rows, s, -1,
                               nullptr, true, w.dyn)) return 1;
                in_slab = w.deferred;
            } else {
                if (gemm_bf16a(h, net.dec[i], EPI_BIAS_LRELU, in, net.dec[i].K, nullptr, w.dec_act_b[i], true, net.dec[i].N, rows, s, -1,
                               nullptr, true, false)) return 1;
                in = w.dec_act_b[i];
            }
        }
        TailArgs ta;
        plan_tail(net.dec, st, T, h->J, &ta);
        ta.B = B; ta.forward_only = forward_only ? 1 : 0; ta.dbg_ts = nullptr;
        ta.in_slab = in_slab; ta.in_bias = front ? net.front.bias : net.dec[st - 1].bias;
        ta.in_bias_ld = front ? net.dec[0].N : 0;
        for (int i = 0; i < ta.n; ++i) {
            const Layer& f = net.dec[st + i];
            const Layer& g = net.dec_bwd[st + i];
            ta.fwd[i] = TailLayerDev{f.w4, f.bias, f.K, f.N};
            ta.bwd[i] = TailLayerDev{g.w4, nullptr, g.K, g.N};
        }
        ta.a_in = w.dec_act[st - 1]; ta.g_out = nullptr; ta.g_out_b = w.dec_grad_b[st];
        ta.Xp = (w.dyn && !forward_only) ? nullptr : w.dec_act.back();
        ta.e = ea;
        if (launch_tail(h, ta, net.tail_lds, s)) return 1;
        if (forward_only) return 0;
        if (record_mid(h, s)) return 1;
        back_from = st - 1;
        gin = w.dec_grad_b[st];
    } else {
        if (front) {
            if (gemm_bf16a(h, net.front, EPI_BIAS_LRELU, w.trial_b, h->Dp, nullptr, w.dec_act_b[0], true, net.front.N, B, s, 0, perm, true, false))
                return 1;
            in = w.dec_act_b[0];
        }
        for (int i = front ? 1 : 0; i < n_dec; ++i) {
            const bool last = i + 1 == n_dec;
            if (last) {                // the pose itself: fp32
                if (gemm_bf16a(h, net.dec[i], EPI_BIAS, in, net.dec[i].K, nullptr, w.dec_act[i], false, net.dec[i].N, rows, s, -1, nullptr,
                               false, false)) return 1;
            } else {
                if (gemm_bf16a(h, net.dec[i], EPI_BIAS_LRELU, in, net.dec[i].K, nullptr, w.dec_act_b[i], true, net.dec[i].N, rows, s, -1,
                               nullptr, false, false)) return 1;
                in = w.dec_act_b[i];
            }
        }
        if (forward_only) return 0;
        ea.Xp = w.dec_act.back();
        ea.dXp_b = w.dXp_b;
        if (launch_energy(h, ea, B, s)) return 1;
        if (record_mid(h, s)) return 1;
        back_from = n_dec - 1;
        gin = w.dXp_b;
    }
    // backward-data through the batched layers: gradient w.r.t. the input of conv i, masked by LeakyReLU' of that input
    for (int i = back_from; i >= (front ? 1 : 0); --i) {
        const Layer& L = net.dec_bwd[i];
        if (gemm_bf16a(h, L, i > 0 ? EPI_MASK : EPI_NONE, gin, L.K, i > 0 ? w.dec_act_b[i - 1] : nullptr, w.dec_grad_b[i], true, L.N, rows, s,
                       -1, nullptr, i == 0, false)) return 1;
        gin = w.dec_grad_b[i];
    }
    // decoder_input^T: dE/dz fp32; in the rounds lbfgs_advance sums the slabs itself
    const Layer& last_l = front ? net.front_bwd : net.dec_in_bwd;         // front: gin is the gradient w.r.t. conv 0's pre-activation
    if (gemm_bf16a(h, last_l, EPI_NONE, gin, last_l.K, nullptr, w.dz, false, h->Dp, B, s, 0, nullptr, true, w.dyn)) return 1;
    w.grad_slab = w.deferred;
    return 0;
}

}  // namespace gem
