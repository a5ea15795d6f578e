// Fused decoder tail: the narrow temporal convs of the decoder, the energy terms and the matching
// backward-data convs in ONE kernel, activations resident in LDS (gfx950).
//
// After the first (wide) decoder layers the network is narrow (128 -> 64 -> 64 -> 64 -> 45 channels):
// as separate launches these layers, the energy kernel and their adjoints are ~15 dependent kernels of
// a few microseconds each per evaluation, dominated by launch boundaries and split-K reduce passes.
// Here one workgroup (4 waves) owns G = floor(32/T) windows = G*T <= 32 rows (one 32-row MFMA tile):
//
//   a_in rows -> LDS;  for each fused layer:  act[i+1] = lrelu(conv3(act[i]) + b)   (v_mfma_f32_32x32x2_f32)
//   X = act[n] -> energy terms + dE/dX per window (one wave per window, energy_device.h)
//   backward-data through the same layers with the LeakyReLU' masks taken from the LDS activations
//   -> gradient w.r.t. a_in written to HBM for the remaining (wide) backward layers.
//
// A operands come from LDS (rows are (window, frame); the k=3 conv reads rows t-1, t, t+1 of the same
// window, zero outside).  B operands (weights) are read straight from L2 into registers: fp32 MFMA is
// slow enough (64 cycles per instruction) that one coalesced 16-byte load per 4 MFMAs is free; the tail
// weights are stored [tap][K/4][N][4] so that the 32 lanes of a half-wave read 512 contiguous bytes.
// With N = 64 there are only two 32-wide column tiles, so the 4 waves split K two ways and combine
// partial tiles through LDS.
//
// Reference semantics: ConvTranspose1d/Conv1d k=3 s=1 p=1 + BatchNorm(eval) + LeakyReLU of
// networks/models/SeqConvVAE.py:67-92 (folded at load time), total_loss of optimizer.py:226-240.
#include "energy_device.h"

namespace gem {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// B fragments of the first (tile, k-block) a wave will need in layer L: issued before the barrier that ends the
// previous layer, so that their L2 latency overlaps the epilogue / barrier / energy phase.
__device__ __forceinline__ void tail_prefetch_first(const TailLayerDev L, f32x4 (&dst)[4]) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, fr = lane & 31, fh = lane >> 5;
    const int ntiles = L.N / 32;
    const int ksplit = ntiles >= 4 ? 1 : 4 / ntiles;
    const int kpart = ksplit > 1 ? wave / ntiles : 0;
    const int kb = L.K / 32, nblk = 3 * kb;
    const int blk0 = kpart * nblk / ksplit;
    const int tile = ksplit > 1 ? wave % ntiles : wave;
    const int tap = blk0 / kb, k0 = (blk0 - tap * kb) * 32;
    const f32x4* p = reinterpret_cast<const f32x4*>(L.w4) + (size_t)tap * (L.K / 4) * L.N + (size_t)(k0 / 4 + fh) * L.N + tile * 32 + fr;
#pragma unroll
    for (int c = 0; c < 4; ++c) dst[c] = p[(size_t)(2 * c) * L.N];
}

// bpre: in = fragments from tail_prefetch_first(L); out = the same for `next` (if any), issued before the epilogue
template <typename Epi>
__device__ __forceinline__ void tail_gemm(const float* in, int ld_in, const TailLayerDev L, const TailLayerDev next, bool has_next,
                                          int T, int R, float* red, f32x4 (&bpre)[4], Epi epi) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 31, fh = lane >> 5;
    const int ntiles = L.N / 32;
    const int ksplit = ntiles >= 4 ? 1 : 4 / ntiles;          // ntiles is even (N is a multiple of 64)
    const int kpart = ksplit > 1 ? wave / ntiles : 0;
    const int kb = L.K / 32;
    const int nblk = 3 * kb;                                  // (tap, 32-wide k block) pairs
    const int blk0 = kpart * nblk / ksplit, blk1 = (kpart + 1) * nblk / ksplit;
    const int t_row = fr % T;
    const bool row_ok = fr < R;
    const int K4N = (L.K / 4) * L.N;                          // float4 per tap
    const f32x4* W4 = reinterpret_cast<const f32x4*>(L.w4);
    const int tile_step = ksplit > 1 ? ntiles : 4;
    for (int tile = ksplit > 1 ? wave % ntiles : wave; tile < ntiles; tile += tile_step) {
        const int n0 = tile * 32;
        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
        // B fragments of block `blk_` straight from L2: [tap][K/4][N][4], 512 contiguous bytes per half-wave
#define TAIL_LOAD_B(blk_, dst_)                                                                        \
        {                                                                                              \
            const int tap_ = (blk_) / kb, k0_ = ((blk_) - tap_ * kb) * 32;                             \
            const f32x4* p_ = W4 + (size_t)tap_ * K4N + (size_t)(k0_ / 4 + fh) * L.N + n0 + fr;        \
            _Pragma("unroll") for (int c = 0; c < 4; ++c) dst_[c] = p_[(size_t)(2 * c) * L.N];         \
        }
        // A fragments of block `blk_` from LDS (rows t-1 / t / t+1 of the same window, zero outside)
#define TAIL_LOAD_A(blk_, dst_)                                                                        \
        {                                                                                              \
            const int tap_ = (blk_) / kb, k0_ = ((blk_) - tap_ * kb) * 32;                             \
            const int tt_ = t_row + tap_ - 1;                                                          \
            const bool ok_ = row_ok && tt_ >= 0 && tt_ < T;                                            \
            const float* arow_ = in + (ok_ ? fr + tap_ - 1 : 0) * ld_in + k0_ + 4 * fh;                \
            _Pragma("unroll") for (int c = 0; c < 4; ++c) {                                            \
                dst_[c] = *reinterpret_cast<const f32x4*>(arow_ + 8 * c);                              \
                if (!ok_) dst_[c] = f32x4{0.f, 0.f, 0.f, 0.f};                                         \
            }                                                                                          \
        }
#define TAIL_COMPUTE(a_, b_)                                                                           \
        {                                                                                              \
            _Pragma("unroll") for (int c = 0; c < 4; ++c) {                                            \
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a_[c].x, b_[c].x, acc, 0, 0, 0);            \
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a_[c].y, b_[c].y, acc, 0, 0, 0);            \
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a_[c].z, b_[c].z, acc, 0, 0, 0);            \
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a_[c].w, b_[c].w, acc, 0, 0, 0);            \
            }                                                                                          \
        }
        // two register sets for both operands, loads issued one block (16 MFMAs = 1024 cycles) ahead of use
        f32x4 b0[4], b1[4], a0[4], a1[4];
        const bool first_tile = tile == (ksplit > 1 ? wave % ntiles : wave);
        if (first_tile) {
#pragma unroll
            for (int c = 0; c < 4; ++c) b0[c] = bpre[c];
        } else {
            TAIL_LOAD_B(blk0, b0);
        }
        TAIL_LOAD_A(blk0, a0);
        // Branch-free pair loop (a conditional prefetch makes hipcc fall back to vmcnt(0) at the join) with
        // sched_barriers (otherwise both prefetches are hoisted to the loop top and waited for together).
        const int npairs = (blk1 - blk0) / 2;
        for (int p = 0; p < npairs; ++p) {
            const int blk = blk0 + 2 * p;
            TAIL_LOAD_B(blk + 1, b1);
            TAIL_LOAD_A(blk + 1, a1);
            __builtin_amdgcn_sched_barrier(0);
            TAIL_COMPUTE(a0, b0);
            __builtin_amdgcn_sched_barrier(0);
            const int nxt = min(blk + 2, blk1 - 1);            // last pair: harmless re-load of the final block
            TAIL_LOAD_B(nxt, b0);
            TAIL_LOAD_A(nxt, a0);
            __builtin_amdgcn_sched_barrier(0);
            TAIL_COMPUTE(a1, b1);
            __builtin_amdgcn_sched_barrier(0);
        }
        if ((blk1 - blk0) & 1) TAIL_COMPUTE(a0, b0);
        if (has_next && tile + tile_step >= ntiles) tail_prefetch_first(next, bpre);     // last tile of this wave
#undef TAIL_LOAD_A
#undef TAIL_LOAD_B
#undef TAIL_COMPUTE
        if (ksplit > 1) {
            // every wave runs exactly one tile here, so the barriers are uniform
            if (kpart > 0) {
                float* r = red + ((kpart - 1) * ntiles + tile) * 1024 + lane;
#pragma unroll
                for (int e = 0; e < 16; ++e) r[e * 64] = acc[e];
            }
            __syncthreads();
            if (kpart == 0) {
                for (int p = 1; p < ksplit; ++p) {
                    const float* r = red + ((p - 1) * ntiles + tile) * 1024 + lane;
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[e] += r[e * 64];
                }
                epi(acc, n0);
            }
            __syncthreads();
        } else {
            epi(acc, n0);
        }
    }
}

__global__ __launch_bounds__(256, 1) void decoder_tail_kernel(TailArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 31, fh = lane >> 5;
    const int T = a.e.T;
    const int w0 = blockIdx.x * a.G;                         // first slot of this workgroup
    const int B = a.e.n_dev ? *a.e.n_dev : a.B;              // active slots this round
    if (w0 >= B) return;
    const int nwin = min(a.G, B - w0);
    const int R = nwin * T;                                  // valid rows
    const size_t row0 = (size_t)w0 * T;
    float* red = lds + a.off_red;

    // ---- stage the input activation rows
    {
        const int K0 = a.fwd[0].K, q4 = K0 / 4;
        float* dst = lds + a.off_act[0];
        for (int i = tid; i < 32 * q4; i += 256) {
            const int r = i / q4, c = (i - r * q4) * 4;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (r < R) v = *reinterpret_cast<const f32x4*>(a.a_in + (row0 + r) * K0 + c);
            *reinterpret_cast<f32x4*>(dst + r * a.ld_act[0] + c) = v;
        }
    }
    __syncthreads();

    // ---- forward layers
    f32x4 bpre[4];
    tail_prefetch_first(a.fwd[0], bpre);
    for (int i = 0; i < a.n; ++i) {
        const bool last = (i + 1 == a.n);
        float* out = lds + a.off_act[i + 1];
        const int ldo = a.ld_act[i + 1];
        const float* bias = a.fwd[i].bias;
        float* Xp = last ? a.Xp : nullptr;
        // (by value: taking the address of a kernel-argument member would put the whole struct in scratch)
        const TailLayerDev nxt = !last ? a.fwd[i + 1] : a.bwd[a.n - 1];
        tail_gemm(lds + a.off_act[i], a.ld_act[i], a.fwd[i], nxt, !last || !a.forward_only, T, R, red, bpre,
                  [&](const f32x16& acc, int n0) {
            const int col = n0 + fr;
            const float bv = bias[col];
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = (e & 3) + 8 * (e >> 2) + 4 * fh;
                float v = acc[e] + bv;
                if (!last) v = v > 0.f ? v : v * LEAKY_SLOPE;
                out[row * ldo + col] = v;
                if (Xp && row < R) Xp[(row0 + row) * PAD + col] = v;
            }
        });
        __syncthreads();
    }
    if (a.forward_only) return;

    // ---- energy terms + dE/dX: one wave per window
    float* g_cur = lds + a.off_g[0];
    float* g_nxt = lds + a.off_g[1];
    if (wave < nwin) {
        float* scr = lds + a.off_escr + wave * 4 * a.escr;
        energy_window<false>(a.e, a.e.perm ? a.e.perm[w0 + wave] : w0 + wave, lane, lds + a.off_act[a.n] + wave * T * a.ld_act[a.n], a.ld_act[a.n], scr, scr + a.escr,
                             scr + 2 * a.escr, scr + 3 * a.escr, g_cur + wave * T * a.ld_g, a.ld_g, a.fwd[a.n - 1].N);
    }
    __syncthreads();

    // ---- backward-data layers (adjoint convs), LeakyReLU' from the sign of the LDS activations
    for (int i = a.n - 1; i >= 0; --i) {
        const float* act = lds + a.off_act[i];
        const int lda = a.ld_act[i];
        const int ldg = a.ld_g;
        float* gout = a.g_out;
        const int K0 = a.fwd[0].K;
        tail_gemm(g_cur, a.ld_g, a.bwd[i], a.bwd[i > 0 ? i - 1 : 0], i > 0, T, R, red, bpre, [&](const f32x16& acc, int n0) {
            const int col = n0 + fr;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = (e & 3) + 8 * (e >> 2) + 4 * fh;
                const float v = acc[e] * (act[row * lda + col] > 0.f ? 1.f : LEAKY_SLOPE);
                if (i > 0) g_nxt[row * ldg + col] = v;
                else if (row < R) gout[(row0 + row) * K0 + col] = v;
            }
        });
        __syncthreads();
        float* t = g_cur; g_cur = g_nxt; g_nxt = t;
    }
}

// LDS plan for a fused chain starting at decoder conv `start` (input = output of conv start-1).  Returns the
// byte size, or 0 when the chain is not fusable.
size_t plan_tail(const std::vector<Layer>& dec, int start, int T, int J, TailArgs* out) {
    const int n = (int)dec.size() - start;
    if (start < 1 || n < 1 || n > TAIL_MAX_LAYERS || T > 32) return 0;
    TailArgs a{};
    a.n = n;
    a.G = 32 / T;
    int off = 0, maxg = 0;
    for (int i = 0; i <= n; ++i) {
        const int width = i == 0 ? dec[start].K : dec[start + i - 1].N;
        a.off_act[i] = off;
        a.ld_act[i] = width + 4;
        off += 32 * (width + 4);
        if (i >= 1 && width > maxg) maxg = width;
    }
    a.ld_g = maxg + 4;
    a.off_g[0] = off; off += 32 * a.ld_g;
    a.off_g[1] = off; off += 32 * a.ld_g;
    a.off_red = off; off += 3 * 1024;
    a.escr = (T * J * 3 + 3) / 4 * 4;
    a.off_escr = off; off += a.G * 4 * a.escr;
    if (out) *out = a;
    return (size_t)off * sizeof(float);
}

int launch_tail(gem_handle* h, const TailArgs& a, size_t lds_bytes, hipStream_t s) {
    static bool attr_set = false;
    if (!attr_set) {
        GEM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(decoder_tail_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    160 * 1024));
        attr_set = true;
    }
    Profile::Rec rec;
    const bool prof = h->prof.on;
    if (prof) {
        GEM_HIP(hipEventCreate(&rec.a)); GEM_HIP(hipEventCreate(&rec.b));
        rec.family = 1; rec.flops = 0;
        GEM_HIP(hipEventRecord(rec.a, s));
    }
    const int wgs = (a.B + a.G - 1) / a.G;
    hipLaunchKernelGGL(decoder_tail_kernel, dim3(wgs), dim3(256), lds_bytes, s, a);
    GEM_HIP(hipGetLastError());
    if (prof) { GEM_HIP(hipEventRecord(rec.b, s)); h->prof.recs.push_back(rec); }
    return 0;
}

}  // namespace gem
