// Fused decoder tail: the narrow temporal convs of the decoder, the energy terms and the matching
// backward-data convs in ONE kernel, activations resident in LDS (gfx950).
//
// After the first (wide) decoder layers the network is narrow (128 -> 64 -> 64 -> 64 -> 45 channels):
// as separate launches these layers, the energy kernel and their adjoints are ~15 dependent kernels of
// a few microseconds each per evaluation, dominated by launch boundaries and split-K reduce passes.
// Here one workgroup (8 waves) owns G = floor(16/T) windows = G*T <= 16 rows (ONE 16-row MFMA tile; T = 10: one
// window per workgroup, so a 240-window round spreads over 240 CUs instead of piling 3 windows on each of 80):
//
//   a_in rows -> LDS;  for each fused layer:  act[i+1] = lrelu(conv3(act[i]) + b)   (v_mfma_f32_16x16x4_f32)
//   X = act[n] -> energy terms + dE/dX per window (all eight waves when G = 1, else one wave per window; energy_device.h)
//   backward-data through the same layers with the LeakyReLU' masks taken from the LDS activations
//   -> gradient w.r.t. a_in written to HBM for the remaining (wide) backward layers.
//
// A operands come from LDS (rows are (window, frame); the k=3 conv reads rows t-1, t, t+1 of the same
// window; rows outside it read a zero line).  B operands (weights) are read straight from L2 into registers, one
// k-block ahead; the tail weights are stored [tap][K/4][N][4] so that 16 lanes read 256 contiguous bytes.  Every
// workgroup streams the same weights (1.3 MB per launch): with 30 workgroups per XCD the two 256<->128 layers run
// at about half the MFMA rate, bound by L2 -> CU bandwidth; the 64-wide layers are latency chains of ~2 us.
// Output tiles are 16x16 (v_mfma_f32_16x16x4_f32), wave w owns columns 16w..16w+15 (+128 per extra tile) with the
// full K walk: a 128-wide layer is one tile per wave, a 64-wide layer keeps waves 0-3 busy.
//
// Reference semantics: ConvTranspose1d/Conv1d k=3 s=1 p=1 + BatchNorm(eval) + LeakyReLU of
// networks/models/SeqConvVAE.py:67-92 (folded at load time), total_loss of optimizer.py:226-240.
#include <algorithm>

#include "energy_device.h"

namespace gem {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// fp32 value rounded to the nearest bf16 (ties to even), still as fp32 bits (low 16 bits zero)
__device__ __forceinline__ float bf16_round(float x) {
    const __bf16 b = (__bf16)x;
    return (float)b;
}

// 8 waves per workgroup = 2 per SIMD: the per-block latency chain (L2 weight fragment -> LDS fragment -> 16 MFMAs)
// of one wave overlaps the other's, and K is cut twice as fine (each wave walks half as many blocks).  That is the shape for
// batches of at most one workgroup per CU (the 240-window workload): the window's chain is as short as it gets.
// With MORE workgroups than CUs the 8-wave kernel (243 VGPRs, 2 waves per SIMD, 70 KB of LDS) runs them one after the other on a
// CU, each paying its staging / barrier / energy latencies with the MFMA pipes idle.  The 4-wave shape (W = 4: one wave per SIMD,
// each wave owns twice the columns, so one LDS A fragment feeds twice the MFMAs; LDS buffers of the window's T rows instead of 16:
// 48 KB; 167 VGPRs) lets THREE workgroups share a CU: one window's latencies hide behind the others' matrix work.  Same K walk
// per output element, so the layers' results are bitwise those of the 8-wave shape; the energies' fp64 partial sums are
// combined over 4 instead of 8 wavefronts (last-bit differences in f only).
constexpr int TAIL_WAVES = 8;
constexpr int TAIL_THREADS = TAIL_WAVES * 64;
constexpr int TAIL_WAVES_SHARED = 4;       // the three-workgroups-per-CU shape

// Tiling: 16x16 output tiles (v_mfma_f32_16x16x4_f32), tile i of wave w = columns 16w + 128i, FULL K walk per
// tile, so no partial sums have to be combined through LDS.  K blocks of 64: lane (r = lane&15, q = lane>>4) holds, for each of the
// four 16-deep groups g, the float4 A[row r][k0+16g+4q .. +3] (LDS) and B[k0+16g+4q .. +3][col r] (L2, layout
// [tap][K/4][N][4]: 16 lanes read 256 contiguous bytes); MFMA step j of group g uses component j of both, i.e.
// the k-pairing is a permutation inside the group, which a sum over k does not care about.
constexpr int TAIL_ROWS = 16;          // rows of (window, frame) per workgroup = one MFMA tile

// B fragments of the first block of the first tile a wave owns in layer L: issued before the barrier that ends
// the previous layer, so that their L2 latency overlaps the epilogue / barrier / energy phase.
template <int W>
__device__ __forceinline__ void tail_prefetch_first(const TailLayerDev L, f32x4 (&dst)[4]) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = lane & 15, q = lane >> 4;
    const int n0 = wave * 16;
    if (n0 >= L.N) return;                                                                   // this wave idles in layer L
    const f32x4* p = reinterpret_cast<const f32x4*>(L.w4) + (size_t)q * L.N + n0 + c;       // tap 0, k0 = 0
#pragma unroll
    for (int g = 0; g < 4; ++g) dst[g] = p[(size_t)(4 * g) * L.N];
}

// bpre: in = fragments from tail_prefetch_first(L); out = the same for `next` (if any), issued before the epilogue.
// epi(acc, tile_row0, col, bias_value) receives the 4 rows tile_row0 + 4*(lane>>4) + {0..3} of column `col`.
// NT = tiles per wave (N / 128, at least 1): the NT tiles of a wave share the 16 rows, so they walk K together:
// one A fragment feeds NT independent accumulators.
// c0 / use_pre: a layer done as several passes over column ranges (4-wave shape, N = 256: two passes of two tiles per wave, so
// that the operand double buffers stay within the register budget of three waves per SIMD) starts pass p at column c0 = p * NT * TS;
// only the first pass finds its first B fragment in bpre, only the last one (has_next) fetches the next layer's.
template <int W, int NT, typename Epi>
__device__ __forceinline__ void tail_gemm_nt(const float* in, int ld_in, const float* zero_line, const TailLayerDev L,
                                             const TailLayerDev next, bool has_next, int T, int R, f32x4 (&bpre)[4], Epi epi,
                                             int c0 = 0, bool use_pre = true) {
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fq = lane >> 4;
    const int kb = L.K / 64, nblk = 3 * kb;                   // (tap, 64-wide k block) pairs
    const f32x4* W4 = reinterpret_cast<const f32x4*>(L.w4);
    constexpr int TS = 16 * W;                                // column stride between the tiles of a wave
    const int n0 = wave * 16 + c0;                            // tile i of this wave: columns n0 + TS*i
    if (n0 >= L.N) {                                          // 64-wide layer, 8 waves: waves 4-7 only fetch ahead
        if (has_next) tail_prefetch_first<W>(next, bpre);
        return;
    }
    const int row = fr;
    const int t_row = row % T;
    const bool row_ok = row < R;
    float bv[NT];
    f32x4 acc[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        bv[i] = L.bias ? L.bias[n0 + TS * i + fr] : 0.f;      // issued now, consumed in the epilogue
        acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // B: with the [tap][K/4][N][4] layout the (tap, k-block) pairs are consecutive: block blk starts 16*N float4 after
    // block blk-1, whatever the tap.  A: rows outside the window (the conv's zero padding) and rows past R read a
    // zero line of LDS instead -- selecting the ADDRESS keeps the ds_reads in flight behind the MFMAs, a select on
    // the loaded DATA would make the wave wait for them on the spot.
    const f32x4* pB = W4 + (size_t)fq * L.N + n0 + fr;
    const int strideB = 16 * L.N;
    const int a_own = row * ld_in + 4 * fq;                   // this lane's row, tap 1
    const int a_zero = (int)(zero_line - in) + 4 * fq;
    const bool ok_m = row_ok && t_row >= 1, ok_p = row_ok && t_row + 1 < T;
#define TAIL_LOAD_B(blk_, dst_)                                                                        \
    {                                                                                                  \
        const f32x4* p_ = pB + (size_t)(blk_) * strideB;                                               \
        _Pragma("unroll") for (int i = 0; i < NT; ++i)                                                 \
            _Pragma("unroll") for (int g = 0; g < 4; ++g) dst_[i][g] = p_[(size_t)(4 * g) * L.N + TS * i]; \
    }
#define TAIL_LOAD_A(blk_, dst_)                                                                        \
    {                                                                                                  \
        const int tap_ = ((blk_) >= kb) + ((blk_) >= 2 * kb), k0_ = ((blk_) - tap_ * kb) * 64;         \
        const bool ok_ = tap_ == 0 ? ok_m : (tap_ == 1 ? row_ok : ok_p);                               \
        const float* arow_ = in + (ok_ ? a_own + (tap_ - 1) * ld_in + k0_ : a_zero);                   \
        _Pragma("unroll") for (int g = 0; g < 4; ++g) dst_[g] = *reinterpret_cast<const f32x4*>(arow_ + 16 * g); \
    }
#define TAIL_COMPUTE(a_, b_)                                                                           \
    {                                                                                                  \
        _Pragma("unroll") for (int g = 0; g < 4; ++g) {                                                \
            _Pragma("unroll") for (int i = 0; i < NT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_[g].x, b_[i][g].x, acc[i], 0, 0, 0); \
            _Pragma("unroll") for (int i = 0; i < NT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_[g].y, b_[i][g].y, acc[i], 0, 0, 0); \
            _Pragma("unroll") for (int i = 0; i < NT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_[g].z, b_[i][g].z, acc[i], 0, 0, 0); \
            _Pragma("unroll") for (int i = 0; i < NT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_[g].w, b_[i][g].w, acc[i], 0, 0, 0); \
        }                                                                                              \
    }
    // two register sets for both operands, loads issued one block (16*NT MFMAs) ahead of their use
    f32x4 b0[NT][4], b1[NT][4], a0[4], a1[4];
    TAIL_LOAD_B(0, b0);
    if (use_pre) {
#pragma unroll
        for (int g = 0; g < 4; ++g) b0[0][g] = bpre[g];       // tile 0 / block 0 was prefetched by the previous layer
    }
    TAIL_LOAD_A(0, a0);
    // Branch-free pair loop (a conditional prefetch makes hipcc fall back to vmcnt(0) at the join); a sched_barrier between the
    // halves (otherwise both prefetches are hoisted to the loop top and waited for together), and inside a half the loads of the
    // next block are dealt out between the MFMAs of the current one (sched_group_barrier): a wave issues them in the shadow of its
    // own matrix instructions instead of in front of them -- with one wave per SIMD and workgroup (the 4-wave shape) nothing else
    // would feed the pipe meanwhile (1536 windows fp32: 49.4 k against 48.0 k windows/s; no difference for the 8-wave shape).
    const int npairs = nblk / 2;
    // one block = 16 NT MFMAs, 4 NT global loads (B of the next block), 4 LDS reads (A of the next block)
#define TAIL_INTERLEAVE()                                                                                          \
    {                                                                                                              \
        _Pragma("unroll") for (int i_ = 0; i_ < 4 * NT; ++i_) { __builtin_amdgcn_sched_group_barrier(0x008, 2, 0); __builtin_amdgcn_sched_group_barrier(0x020, 1, 0); } \
        _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) { __builtin_amdgcn_sched_group_barrier(0x008, 2 * NT, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); } \
    }
    for (int p = 0; p < npairs; ++p) {
        const int blk = 2 * p;
        TAIL_LOAD_B(blk + 1, b1);
        TAIL_LOAD_A(blk + 1, a1);
        TAIL_COMPUTE(a0, b0);
        TAIL_INTERLEAVE();
        __builtin_amdgcn_sched_barrier(0);
        const int nxt = min(blk + 2, nblk - 1);            // last pair: harmless re-load of the final block
        TAIL_LOAD_B(nxt, b0);
        TAIL_LOAD_A(nxt, a0);
        TAIL_COMPUTE(a1, b1);
        TAIL_INTERLEAVE();
        __builtin_amdgcn_sched_barrier(0);
    }
#undef TAIL_INTERLEAVE
    if (nblk & 1) TAIL_COMPUTE(a0, b0);
    if (has_next) tail_prefetch_first<W>(next, bpre);
#undef TAIL_LOAD_A
#undef TAIL_LOAD_B
#undef TAIL_COMPUTE
#pragma unroll
    for (int i = 0; i < NT; ++i) epi(acc[i], 4 * fq, n0 + TS * i + fr, bv[i]);
}

template <int W, typename Epi>
__device__ __forceinline__ void tail_gemm(const float* in, int ld_in, const float* zero_line, const TailLayerDev L,
                                          const TailLayerDev next, bool has_next, int T, int R, f32x4 (&bpre)[4], Epi epi) {
    // N is 64, 128, 256 or 512 (plan_tail): N / (16 W) tiles per wave, at least one (W = 4: N <= 256, tail_can_share_cu checks)
    if constexpr (W == TAIL_WAVES_SHARED) {
        if (L.N > 32 * W) {
            tail_gemm_nt<W, 2>(in, ld_in, zero_line, L, next, false, T, R, bpre, epi);
            tail_gemm_nt<W, 2>(in, ld_in, zero_line, L, next, has_next, T, R, bpre, epi, 32 * W, false);
        } else if (L.N > 16 * W) tail_gemm_nt<W, 2>(in, ld_in, zero_line, L, next, has_next, T, R, bpre, epi);
        else tail_gemm_nt<W, 1>(in, ld_in, zero_line, L, next, has_next, T, R, bpre, epi);
    } else {
        if (L.N > 32 * W) tail_gemm_nt<W, 4>(in, ld_in, zero_line, L, next, has_next, T, R, bpre, epi);
        else if (L.N > 16 * W) tail_gemm_nt<W, 2>(in, ld_in, zero_line, L, next, has_next, T, R, bpre, epi);
        else tail_gemm_nt<W, 1>(in, ld_in, zero_line, L, next, has_next, T, R, bpre, epi);
    }
}

// NL = number of fused layers (compile time: the two layer loops unroll, so every layer's descriptor sits at a fixed
// kernel-argument offset instead of being fetched by index behind each barrier)
// W = wavefronts per workgroup: 8 (one workgroup per CU) or 4 (two per CU, G == 1 and N <= 256 only); launch_tail chooses
template <int NL, int W>
__global__ __launch_bounds__(64 * W, W == TAIL_WAVES_SHARED ? 3 : 2) void decoder_tail_kernel(TailArgs a) {
    constexpr int THREADS = 64 * W;
    constexpr bool TIGHT = W == TAIL_WAVES_SHARED;           // the LDS carve holds a.rows = G*T rows per buffer instead of 16
    constexpr int STAGE_U = 2048 / THREADS;                  // float4 per thread that cover 16 rows of K0 <= 512
    constexpr int PRE_E = TAIL_THREADS / THREADS;            // parked energy inputs per thread (up to 512 values per window)
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int T = a.e.T;
    int probe = 0;
#define TAIL_PROBE()                                                                                   \
    if (a.dbg_ts && blockIdx.x == 0 && tid == 0) {                                                     \
        a.dbg_ts[2 * probe] = clock64();                                                               \
        a.dbg_ts[2 * probe + 1] = wall_clock64();                                                      \
        ++probe;                                                                                       \
    }
    const int w0 = blockIdx.x * a.G;                         // first slot of this workgroup
    const int B = a.e.n_dev ? *a.e.n_dev : a.B;              // active slots this round
    if (w0 >= B) return;
    const int nwin = min(a.G, B - w0);
    const int R = nwin * T;                                  // valid rows
    const size_t row0 = (size_t)w0 * T;

    // ---- stage the input activation rows: all of a thread's loads in flight before the first LDS store.  In the
    // rounds the producer GEMM leaves its split-K slabs here: they are summed in slab order (bitwise what the reduce
    // kernel would have written), bias and LeakyReLU applied on the way into LDS.
    {
        const int K0 = a.fwd[0].K, q4 = K0 / 4, n4 = (TIGHT ? a.rows : TAIL_ROWS) * q4;      // K0 <= 512: at most 2048 / THREADS float4 per thread
        float* dst = lds + a.off_act[0];
        if (a.in_slab.base) {
            int nslab;
            size_t stride;
            slab_layout(a.in_slab, nslab, stride);
            for (int u = 0; u < STAGE_U; ++u) {
                const int i = tid + u * THREADS, r = i / q4, c = (i - r * q4) * 4;
                if (i >= n4) break;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (r < R) {
                    const float* p = a.in_slab.base + (row0 + r) * K0 + c;
                    for (int z0 = 0; z0 < nslab; z0 += 8) {
                        f32x4 t[8];
#pragma unroll
                        for (int k = 0; k < 8; ++k) t[k] = *reinterpret_cast<const f32x4*>(p + (size_t)min(z0 + k, nslab - 1) * stride);
#pragma unroll
                        for (int k = 0; k < 8; ++k)
                            if (z0 + k < nslab) v += t[k];
                    }
                    v += *reinterpret_cast<const f32x4*>(a.in_bias + (r % T) * a.in_bias_ld + c);
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] = v[q] > 0.f ? v[q] : v[q] * LEAKY_SLOPE;
                }
                *reinterpret_cast<f32x4*>(dst + r * a.ld_act[0] + c) = v;
            }
        } else {
            f32x4 v[STAGE_U];
#pragma unroll
            for (int u = 0; u < STAGE_U; ++u) {
                const int i = tid + u * THREADS, r = i / q4, c = (i - r * q4) * 4;
                v[u] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (i < n4 && r < R) v[u] = *reinterpret_cast<const f32x4*>(a.a_in + (row0 + r) * K0 + c);
            }
#pragma unroll
            for (int u = 0; u < STAGE_U; ++u) {
                const int i = tid + u * THREADS, r = i / q4, c = (i - r * q4) * 4;
                if (i < n4) *reinterpret_cast<f32x4*>(dst + r * a.ld_act[0] + c) = v[u];
            }
        }
        if (tid < 64) lds[a.off_zero + tid] = 0.f;            // the zero line the padded conv rows read
    }
    // What the energy terms need besides the decoded pose does not depend on the layers: requested now (registers), parked in LDS
    // behind the first layer -- instead of two global round trips between the last forward layer and the first adjoint layer.
    float pre_x0[PRE_E], pre_mb = 0.f;
    int pre_par = 0, pre_ch[PRE_E];
#pragma unroll
    for (int k = 0; k < PRE_E; ++k) { pre_x0[k] = 0.f; pre_ch[k] = -1; }
    // (PRE_E elements per thread, 512 per window: windows of more pose values -- T >= 12 with 15 joints -- read their inputs from
    // global memory inside energy_window instead)
    static_assert(MAXJ * MAXJ <= TAIL_THREADS && MAXJ <= 64, "the children table is parked by PRE_E passes of the workgroup");
    const bool pre_on = a.G == 1 && !a.forward_only && T * a.e.J * 3 <= TAIL_THREADS;
    if (pre_on) {
        const int J = a.e.J, n = T * J * 3, bw = a.e.perm ? a.e.perm[w0] : w0;
#pragma unroll
        for (int k = 0; k < PRE_E; ++k) {
            const int e = tid + k * THREADS;
            if (e < n) pre_x0[k] = a.e.X0[(size_t)bw * n + e];
            if (e < J * MAXJ) pre_ch[k] = a.e.children[e];
        }
        if (tid < J) { pre_mb = a.e.mean_bone[(size_t)bw * J + tid]; pre_par = a.e.parents[tid]; }
    }
    TAIL_PROBE();
    __syncthreads();
    TAIL_PROBE();

    // ---- forward layers
    f32x4 bpre[4];
    tail_prefetch_first<W>(a.fwd[0], bpre);
#pragma unroll
    for (int i = 0; i < NL; ++i) {
        const bool last = (i + 1 == NL);
        float* out = lds + a.off_act[i + 1];
        const int ldo = a.ld_act[i + 1];
        float* Xp = last ? a.Xp : nullptr;
        // (by value: taking the address of a kernel-argument member would put the whole struct in scratch)
        const TailLayerDev nxt = !last ? a.fwd[i + 1 < NL ? i + 1 : i] : a.bwd[NL - 1];
        tail_gemm<W>(lds + a.off_act[i], a.ld_act[i], lds + a.off_zero, a.fwd[i], nxt, !last || !a.forward_only, T, R, bpre,
                  [&](const f32x4& acc, int r0, int col, float bv) {
#pragma unroll
                      for (int e = 0; e < 4; ++e) {
                          float v = acc[e] + bv;
                          if (!last) v = v > 0.f ? v : v * LEAKY_SLOPE;
                          if (!TIGHT || r0 + e < a.rows) out[(r0 + e) * ldo + col] = v;
                          if (Xp && r0 + e < R) Xp[(row0 + r0 + e) * PAD + col] = v;
                      }
                  });
        if (i == 0 && pre_on) {
            const int J = a.e.J, n = T * J * 3;
            float* pre = lds + a.off_pre;
            int* prei = reinterpret_cast<int*>(pre + n + J);
#pragma unroll
            for (int k = 0; k < PRE_E; ++k) {
                const int e = tid + k * THREADS;
                if (e < n) pre[e] = pre_x0[k];
                if (e < J * MAXJ) prei[J + e] = pre_ch[k];
            }
            if (tid < J) { pre[n + tid] = pre_mb; prei[tid] = pre_par; }
        }
        __syncthreads();
        TAIL_PROBE();
    }
    if (a.forward_only) return;

    // ---- energy terms + dE/dX: one wave per window
    float* g_cur = lds + a.off_g[0];
    float* g_nxt = lds + a.off_g[1];
    if (a.G == 1) {
        // one window per workgroup: all eight wavefronts share its energy terms
        float* scr = lds + a.off_escr;
        const int J = a.e.J, n = T * J * 3;
        const float* pre = lds + a.off_pre;
        const int* prei = reinterpret_cast<const int*>(pre + n + J);
        if (T == 10 && J == 15)      // the usual window shape: compile-time index arithmetic (same numbers)
            energy_window<true, THREADS, 10, 15>(a.e, a.e.perm ? a.e.perm[w0] : w0, tid, lds + a.off_act[NL], a.ld_act[NL], scr,
                                                      scr + a.escr, scr + 2 * a.escr, scr + 3 * a.escr, g_cur, a.ld_g, a.fwd[NL - 1].N,
                                                      nullptr, pre_on ? pre : nullptr, pre_on ? pre + n : nullptr, pre_on ? prei : nullptr,
                                                      pre_on ? prei + J : nullptr);
        else
            energy_window<true, THREADS>(a.e, a.e.perm ? a.e.perm[w0] : w0, tid, lds + a.off_act[NL], a.ld_act[NL], scr,
                                              scr + a.escr, scr + 2 * a.escr, scr + 3 * a.escr, g_cur, a.ld_g, a.fwd[NL - 1].N,
                                              nullptr, pre_on ? pre : nullptr, pre_on ? pre + n : nullptr, pre_on ? prei : nullptr,
                                              pre_on ? prei + J : nullptr);
    } else if (wave < nwin) {
        float* scr = lds + a.off_escr + wave * 4 * a.escr;
        energy_window<false>(a.e, a.e.perm ? a.e.perm[w0 + wave] : w0 + wave, lane,
                             lds + a.off_act[NL] + wave * T * a.ld_act[NL], a.ld_act[NL], scr, scr + a.escr,
                             scr + 2 * a.escr, scr + 3 * a.escr, g_cur + wave * T * a.ld_g, a.ld_g, a.fwd[NL - 1].N);
    }
    __syncthreads();
    TAIL_PROBE();

    // ---- backward-data layers (adjoint convs), LeakyReLU' from the sign of the LDS activations
#pragma unroll
    for (int i = NL - 1; i >= 0; --i) {
        const float* act = lds + a.off_act[i];
        const int lda = a.ld_act[i];
        const int ldg = a.ld_g;
        float* gout = a.g_out;
        uint16_t* gout_b = a.g_out_b;
        const int K0 = a.fwd[0].K;
        tail_gemm<W>(g_cur, a.ld_g, lds + a.off_zero, a.bwd[i], a.bwd[i > 0 ? i - 1 : 0], i > 0, T, R, bpre,
                  [&](const f32x4& acc, int r0, int col, float) {
#pragma unroll
                      for (int e = 0; e < 4; ++e) {
                          const int row = r0 + e;
                          // (a chain that starts at the first conv reads the linear decoder_input output: no mask)
                          const float v = (i > 0 || a.mask_first) ? acc[e] * (act[row * lda + col] > 0.f ? 1.f : LEAKY_SLOPE) : acc[e];
                          if (i > 0) { if (!TIGHT || row < a.rows) g_nxt[row * ldg + col] = v; }
                          else if (row < R) {
                              if (gout_b) gout_b[(row0 + row) * K0 + col] = (uint16_t)(__builtin_bit_cast(unsigned int, bf16_round(v)) >> 16);
                              else gout[(row0 + row) * K0 + col] = v;
                          }
                      }
                  });
        __syncthreads();
        TAIL_PROBE();
        float* t = g_cur; g_cur = g_nxt; g_nxt = t;
    }
#undef TAIL_PROBE
}

// LDS plan for a fused chain starting at decoder conv `start` (input = output of conv start-1, or of
// decoder_input when start == 0).  Returns the byte size, or 0 when the chain is not fusable.
size_t plan_tail(const std::vector<Layer>& dec, int start, int T, int J, TailArgs* out, bool shared) {
    const int n = (int)dec.size() - start;
    if (start < 0 || n < 1 || n > TAIL_MAX_LAYERS || T > TAIL_ROWS) return 0;
    for (int i = start; i < (int)dec.size(); ++i) {
        const int K = dec[i].K, N = dec[i].N;                 // tail_gemm: 8 waves x 16 columns x {1,2,4} tiles
        auto ok = [](int v) { return v == 64 || v == 128 || v == 256 || v == 512; };
        if (!ok(K) || !ok(N)) return 0;
    }
    TailArgs a{};
    a.n = n;
    a.mask_first = start > 0;
    a.G = TAIL_ROWS / T;
    if (shared && a.G != 1) return 0;
    // shared (4-wave shape, three workgroups per CU): buffers of G*T rows -- rows past them are never read (the A fragments of rows
    // >= R come from the zero line) and the epilogues do not write them
    a.waves = shared ? TAIL_WAVES_SHARED : TAIL_WAVES;
    a.rows = shared ? a.G * T : TAIL_ROWS;
    const int rows = a.rows;
    int off = 0, maxg = 0;
    for (int i = 0; i <= n; ++i) {
        const int width = i == 0 ? dec[start].K : dec[start + i - 1].N;
        a.off_act[i] = off;
        a.ld_act[i] = width + 4;
        off += rows * (width + 4);
        if (i >= 1 && width > maxg) maxg = width;
    }
    a.ld_g = maxg + 4;
    a.off_g[0] = off; off += rows * a.ld_g;
    a.off_g[1] = off; off += rows * a.ld_g;
    a.escr = (T * J * 3 + 3) / 4 * 4;
    a.off_red = off;                       // (unused since the 16x16 tiling needs no k-split scratch)
    a.off_escr = off;
    off += a.G * 4 * a.escr;
    a.off_zero = off;                      // 64 zero floats
    off += 64;
    a.off_pre = off;                       // G == 1: stage-input pose, mean bone lengths, parent / children tables of the window
    if (a.G == 1) off += (T * J * 3 + 2 * J + J * MAXJ + 3) / 4 * 4;
    if (out) *out = a;
    return (size_t)off * sizeof(float);
}

template <int NL, int W>
static void launch_tail_as(gem_handle* h, const TailArgs& a, int wgs, size_t lds_bytes, hipStream_t s) {
    note_kernel(h, reinterpret_cast<const void*>(decoder_tail_kernel<NL, W>));
    hipLaunchKernelGGL((decoder_tail_kernel<NL, W>), dim3(wgs), dim3(64 * W), lds_bytes, s, a);
}

template <int W>
static int launch_tail_w(gem_handle* h, const TailArgs& a, int wgs, size_t lds_bytes, hipStream_t s) {
    switch (a.n) {
        case 1: launch_tail_as<1, W>(h, a, wgs, lds_bytes, s); break;
        case 2: launch_tail_as<2, W>(h, a, wgs, lds_bytes, s); break;
        case 3: launch_tail_as<3, W>(h, a, wgs, lds_bytes, s); break;
        case 4: launch_tail_as<4, W>(h, a, wgs, lds_bytes, s); break;
        case 5: launch_tail_as<5, W>(h, a, wgs, lds_bytes, s); break;
        case 6: launch_tail_as<6, W>(h, a, wgs, lds_bytes, s); break;
        default: set_error("launch_tail: unsupported number of fused layers"); return 1;
    }
    return 0;
}

// More workgroups than CUs: the 4-wave shape, three workgroups per CU (see TAIL_WAVES_SHARED).  It needs one window per workgroup
// (its energy phase is the whole-workgroup one), layers of at most 256 columns (two passes of two 16-column tiles per wave) and
// room for three of its LDS carves per CU.
static bool tail_can_share_cu(const std::vector<Layer>& dec, int start, int T, int J) {
    const size_t bytes = plan_tail(dec, start, T, J, nullptr, true);
    if (!bytes || 3 * bytes > 160 * 1024) return false;
    for (int i = start; i < (int)dec.size(); ++i)
        if (dec[i].N > 64 * TAIL_WAVES_SHARED || dec[i].K > 64 * TAIL_WAVES_SHARED) return false;
    return true;
}

// The carve (and with it the kernel shape, TailArgs::waves) for a launch of `wgs` workgroups; returns the LDS bytes
size_t plan_tail_for(const gem_handle* h, const std::vector<Layer>& dec, int start, int wgs, TailArgs* out) {
    bool shared = wgs > h->n_cu && tail_can_share_cu(dec, start, h->T, h->J);
    if (const char* f = dev_env("GEM_TAIL_WAVES")) shared = atoi(f) == TAIL_WAVES_SHARED && tail_can_share_cu(dec, start, h->T, h->J);
    return plan_tail(dec, start, h->T, h->J, out, shared);
}

// Up to how many workgroups the fused tail beats the batched narrow layers + energy kernel (fp32, windows/s, MI355X, round 5):
//   windows                480     768    1152    1536    2040    2556    3072    4092
//   4-wave, three per CU  42.5 k  47.9 k          48.5 k          52.1 k  54.3 k  52.7 k
//   4-wave, two per CU    42.9 k  45.5 k  44.9 k  48.0 k  51.4 k  51.4 k  53.2 k  52.0 k     (16-row LDS buffers, 70 KB)
//   8-wave, one per CU    40.8 k                  45.1 k                          48.3 k
//   batched layers        30.7 k  38.5 k  41.6 k  45.8 k  50.4 k  49.8 k  55.6 k  54.7 k
// (the fused tail costs a CU about 25 us per window whatever the batch -- 1.3 MB of weights from L2 per workgroup, one 16-row MFMA
// tile with T = 10 rows used; the batched layers amortise their launches and reduce passes)
int tail_cap_workgroups(const gem_handle* h, const std::vector<Layer>& dec, int start) {
    if (const char* f = dev_env("GEM_TAIL_CAP")) return atoi(f) * h->n_cu;
    return (tail_can_share_cu(dec, start, h->T, h->J) ? 10 : 5) * h->n_cu;
}

int launch_tail(gem_handle* h, const TailArgs& a, size_t lds_bytes, hipStream_t s) {
    static PerDeviceOnce attr_once;
    if (attr_once.need(h->cfg.device)) {
        const void* ks[] = {reinterpret_cast<const void*>(decoder_tail_kernel<1, 8>), reinterpret_cast<const void*>(decoder_tail_kernel<2, 8>),
                            reinterpret_cast<const void*>(decoder_tail_kernel<3, 8>), reinterpret_cast<const void*>(decoder_tail_kernel<4, 8>),
                            reinterpret_cast<const void*>(decoder_tail_kernel<5, 8>), reinterpret_cast<const void*>(decoder_tail_kernel<6, 8>),
                            reinterpret_cast<const void*>(decoder_tail_kernel<1, 4>), reinterpret_cast<const void*>(decoder_tail_kernel<2, 4>),
                            reinterpret_cast<const void*>(decoder_tail_kernel<3, 4>), reinterpret_cast<const void*>(decoder_tail_kernel<4, 4>),
                            reinterpret_cast<const void*>(decoder_tail_kernel<5, 4>), reinterpret_cast<const void*>(decoder_tail_kernel<6, 4>)};
        for (const void* k : ks) GEM_HIP(hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    }
    Profile::Rec rec;
    const bool prof = h->prof.on;
    if (prof) {
        GEM_HIP(hipEventCreate(&rec.a)); GEM_HIP(hipEventCreate(&rec.b));
        rec.family = 1;
        // matrix work of the fused layers on the T real rows of a window: 2 * 3 taps * K * N * T per layer, forward + adjoint
        double per_window = 0.0;
        for (int i = 0; i < a.n; ++i) per_window += 2.0 * 3.0 * a.fwd[i].K * a.fwd[i].N * a.e.T;
        if (!a.forward_only) per_window *= 2.0;
        rec.flops = per_window * a.B;
        if (h->ws.dyn) { rec.log_idx = h->ws.cur_log; rec.flops_per_window = per_window; }      // rows = active windows of the round
        GEM_HIP(hipEventRecord(rec.a, s));
    }
    const int wgs = (a.B + a.G - 1) / a.G;
    if (a.waves != TAIL_WAVES && a.waves != TAIL_WAVES_SHARED) { set_error("launch_tail: the arguments do not come from plan_tail"); return 1; }
    if (a.waves == TAIL_WAVES_SHARED ? launch_tail_w<TAIL_WAVES_SHARED>(h, a, wgs, lds_bytes, s) : launch_tail_w<TAIL_WAVES>(h, a, wgs, lds_bytes, s))
        return 1;
    GEM_HIP(hipGetLastError());
    if (prof) { GEM_HIP(hipEventRecord(rec.b, s)); h->prof.recs.push_back(rec); }
    commit_kernel_names(h, prof ? 1 : -1);
    return 0;
}

}  // namespace gem
