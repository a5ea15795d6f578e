// bf16 multi-window fused decoder tail (gfx950): the narrow temporal convs of the decoder, the energy terms and the matching
// backward-data convs of ONE to EIGHT windows in one workgroup -- the "bf16 VAE decoder / fp32 energy" mode of BASELINE configs[2..4].
//
// The fp32 tail (tail.hip) gives one workgroup to one window (10 of 16 tile rows used) and streams 1.3 MB of fp32 weights per
// window from L2; beyond ~1300 windows the narrow layers therefore ran as ~12 batched bf16 GEMM launches of 7-27 us plus the
// stand-alone energy kernel per evaluation round.  Here one workgroup (8 waves) owns NRT = 1 .. 5 row tiles of 16 rows =
// G = min(8, 16 NRT / T) windows (T = 10: 1, 3, 4, 6, 8): the fewest tiles that keep the launch inside one round of two workgroups per
// CU (tail_bf16_row_tiles; round 4: smaller workgroups use more CUs and a second workgroup on a CU fills the first one's gaps):
//
//   input rows (bf16 matrix, or the producer GEMM's fp32 split-K slabs: summed + bias + LeakyReLU on the way in) -> LDS (bf16)
//   for each fused layer:  act[i+1] = lrelu(conv3(act[i]) + b)      v_mfma_f32_16x16x32_bf16, fp32 accumulate, bf16 in LDS
//   X = act[n] (fp32) -> energy terms + dE/dX per window, fp32 (energy_pairs.h): up to three tiles the whole workgroup on the windows'
//       (frame, joint) pairs, beyond that one wavefront per window -- the same bits either way
//   backward-data through the same layers, LeakyReLU' from one sign bit per activation element
//   -> gradient w.r.t. the input's pre-activation, bf16, staged through LDS and written out as whole rows.
//
// Round 4: TWO workgroups per CU.  The kernel is a chain of ~12 short dependent phases (a layer is 6-24 k-steps between two
// barriers; the energy terms are plain VALU code), so one workgroup per CU left the matrix pipe idle most of the time: 33 us per
// workgroup against 5 us of MFMA work.  A second, independent workgroup on the CU fills those gaps -- its MFMA phases run beside
// this one's energy / epilogue / barrier phases -- which needs <= 80 KB of LDS and <= 128 VGPRs per workgroup:
//   * LDS (77 KB at the reference's widths, was 159 KB): two ping-pong buffers P (as wide as the input) and Q (as wide as act[1]).
//     act[j] lives in buffer j & 1 and is overwritten two layers later; what the backward direction needs of it -- the sign of
//     every element -- is kept as one BIT per element (a byte per 8 channels for the staged input, per 4 for the layers' own
//     outputs: the four consecutive channels a lane owns).  The gradient w.r.t. act[j] goes to buffer j & 1 as well; the decoded
//     pose (fp32, dense per window) sits behind act[n-1] in that activation's buffer; the energy terms need no scratch
//     (energy_pairs.h recomputes neighbours instead of storing three arrays per window).
//   * VGPRs: the activation fragments are refreshed in place right behind the MFMA that used them (was a register double buffer),
//     the inputs of the energy terms are requested in front of the LAST forward layer (the cheapest one) instead of at kernel
//     start, the slab-summing input path stages two chunks per thread and trip (was five).
//
// MFMA operands: the WEIGHT fragment is operand A (rows n), the activation fragment operand B (columns m): D[n][m], so a lane
// ends up with 4 consecutive output channels of ONE row = one 8-byte bf16 store into the next layer's LDS image.  Activation
// rows are 2*width + 32 bytes apart: with that stride the 16-byte fragment reads (lane l: row l & 15, 16-byte chunk l >> 4) of
// ds_read_b128's four lane groups hit 16 distinct 16-byte slots each -- conflict-free without a swizzle, and the K walk is an
// immediate offset.  The k=3 conv reads rows t-1, t, t+1 of the same window; rows outside it read a zero line (address select).
// Weights never touch LDS: at load time every layer is cut into the 1 KB fragments (64 lanes x 8 bf16) each wave will need, in
// the order it will need them (build_tail_bf16_stream), so a wave walks ONE contiguous stream -- forward layers, then adjoint
// layers -- with a ring of six fragments in registers, requested six K-steps ahead, across layer boundaries, barriers and the
// energy phase.  Wave w owns output channels 16w..16w+15 (+128 per extra tile) and all five row tiles; in a 64-wide layer waves
// w and w+4 share a channel tile and split the row tiles 3 + 2.
//
// Reference semantics: ConvTranspose1d/Conv1d k=3 s=1 p=1 + BatchNorm(eval) + LeakyReLU of networks/models/SeqConvVAE.py:76-92
// (folded at load time), decode_to_bodypose :131-140, total_loss of optimizer.py:226-240, backward-DATA only (frozen VAE).
#include <algorithm>
#include <cstring>

#include "energy_pairs.h"

namespace gem {

namespace tb {

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef short bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int WAVES = 8, THREADS = WAVES * 64;
constexpr int NRT_MAX = 5;                     // row tiles (of 16 rows) per workgroup: 1 .. 5 (1, 3, 4, 6, 8 windows of 10 frames)
constexpr int ZERO_BYTES = 1024 + 64;          // the zero line covers the K walk of the widest layer (K = 512: 1024 bytes)

__device__ __forceinline__ unsigned int pack2(float lo, float hi) {      // round-to-nearest-even (v_cvt_pk_bf16_f32)
    typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
    typedef float f2 __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(unsigned int, __builtin_convertvector(f2{lo, hi}, bf2));
}
__device__ __forceinline__ float lrelu(float v) { return v > 0.f ? v : v * LEAKY_SLOPE; }
// bit e of the result: the e-th of four packed bf16 values (two dwords) is > 0 (sign bit clear and not zero)
__device__ __forceinline__ unsigned int pos_bits(unsigned int d0, unsigned int d1) {
    const unsigned int a = d0 & 0xFFFFu, b = d0 >> 16, c = d1 & 0xFFFFu, d = d1 >> 16;
    return ((a != 0u && a < 0x8000u) ? 1u : 0u) | ((b != 0u && b < 0x8000u) ? 2u : 0u) | ((c != 0u && c < 0x8000u) ? 4u : 0u) |
           ((d != 0u && d < 0x8000u) ? 8u : 0u);
}

// LDS writes of this wave done, then the workgroup barrier.  (Not __syncthreads: the weight ring's global loads stay in
// flight across the barrier.)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// One fused layer (one column tile of it) for this wave: D[n][m] += W[n][k] . act[m][k] over 3 taps x K.
//   SPLIT 64-wide layer: waves w and w + 4 share column tile w & 3; w < 4 takes the first ceil(NRT / 2) row tiles, w >= 4 the rest
// The K walk is k-block major, taps inner (the weight stream is packed in the same order): a body of 6 k-steps covers two whole
// k-blocks, so taps and ring slots are compile-time.  epi(tile, acc) gets the accumulator of row tile `tile`
// (row 16 * tile + (lane & 15), channels col0 + 4 * (lane >> 4) + 0..3).
// rowbits: 3 bits per row tile of this lane's row (16 * tile + (lane & 15)): bit 0 = the row exists (< R), bit 1 / 2 = its frame
// has a predecessor / successor inside the window (taps 0 / 2; computed once per kernel, no division per layer).
// RING: weight fragments in flight per wave (6 or 3); AFD: 2 = the activation fragments of the next k-step are read into a second
// register set (one workgroup per CU: nothing else covers the LDS latency), 1 = refreshed in place right behind the MFMA that used
// them (two workgroups per CU: 20 registers less).
template <bool SPLIT, int NRT, int RING, int AFD, typename Epi>
__device__ __forceinline__ void gemm_layer(const unsigned char* lds, int in_off, int in_ld, int zero_off, int K, unsigned int rowbits,
                                           bf16x8 (&ring)[RING], const bf16x8* __restrict__ wp, int& consumed, int last_step, Epi epi) {
    constexpr int NS = (NRT + 1) / 2;                 // split layers: waves 0-3 take row tiles [0, NS), waves 4-7 [NS, NRT)
    constexpr int NR = SPLIT ? NS : NRT;
    constexpr int KS = 6;                             // k-steps (32 deep) per unrolled body = two whole k-blocks x 3 taps
    static_assert(KS % RING == 0, "ring slots must be compile-time inside the unrolled body");
    static_assert(AFD == 1 || AFD == 2, "in place or double-buffered");
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r16 = lane & 15, q = lane >> 4;
    const int tile0 = SPLIT ? NS * (wave >> 2) : 0;
    const unsigned int bits = rowbits >> (3 * tile0);
    // fragment read addresses per tap and row tile (bytes); invalid rows (outside the window: the conv's zero padding; past R;
    // the absent sixth tile of the split) read the zero line -- selecting the ADDRESS keeps the reads in flight behind the MFMAs
    int base[3][NR];
#pragma unroll
    for (int i = 0; i < NR; ++i) {
        const int row = 16 * (tile0 + i) + r16;
        const unsigned int b3 = (bits >> (3 * i)) & 7u;
        const bool ok = (b3 & 1u) != 0u;
        const int mid = in_off + row * in_ld + q * 16;
        base[0][i] = (ok && (b3 & 2u)) ? mid - in_ld : zero_off + q * 16;
        base[1][i] = ok ? mid : zero_off + q * 16;
        base[2][i] = (ok && (b3 & 4u)) ? mid + in_ld : zero_off + q * 16;
    }
    f32x4 acc[NR];
#pragma unroll
    for (int i = 0; i < NR; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int nsteps = 3 * K / 32;
    // Program order = issue order (sched_barrier between the groups; left alone, the compiler sinks all ring refills to the end of
    // the body, which halves the distance the stream runs ahead): every MFMA is followed by the read of the NEXT k-step's
    // activation fragment for the same row tile, every k-step by the refill of the ring slot it has just used.  The body's last
    // k-step reads the first fragments of the next body (past the last k-block of the layer that is 16 bytes of row padding /
    // the neighbouring row: read, never used).
    bf16x8 af[AFD][NR];
#pragma unroll
    for (int i = 0; i < NR; ++i) af[0][i] = *reinterpret_cast<const bf16x8*>(lds + base[0][i]);
    for (int s0 = 0; s0 < nsteps; s0 += KS) {
#pragma unroll
        for (int j = 0; j < KS; ++j) {                // k-step j of the body: tap j % 3 of k-block j / 3
            const int jn = j + 1;                     // next k-step (jn == KS: tap 0 of the next body's first k-block)
#pragma unroll
            for (int i = 0; i < NR; ++i) {
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ring[j % RING], af[j % AFD][i], acc[i], 0, 0, 0);
                af[jn % AFD][i] = *reinterpret_cast<const bf16x8*>(lds + base[jn % 3][i] + (jn / 3) * 64);
                __builtin_amdgcn_sched_barrier(0);
            }
            // refill the slot RING steps ahead (past the end of the stream: a harmless re-load of the last fragment)
            ring[j % RING] = wp[(size_t)min(consumed + s0 + j + RING, last_step) * 64 + lane];
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int tap = 0; tap < 3; ++tap)
#pragma unroll
            for (int i = 0; i < NR; ++i) base[tap][i] += (KS / 3) * 64;
    }
    consumed += nsteps;
#pragma unroll
    for (int i = 0; i < NR; ++i)
        if (tile0 + i < NRT) epi(tile0 + i, acc[i]);
}

// N is 64, 128 or 256 (plan_tail_bf16).  A 256-wide layer runs as two passes over K, one per 128 channels (twice the fragment
// reads, half the accumulators: the kernel has to stay under 128 VGPRs); the weight stream is packed pass by pass.
// epi(col0, tile, acc): col0 = first channel of the wave's column tile in this pass.
template <int NRT, int RING, int AFD, typename Epi>
__device__ __forceinline__ void gemm_dispatch(const unsigned char* lds, int in_off, int in_ld, int zero_off, int K, int N, unsigned int rowbits,
                                              bf16x8 (&ring)[RING], const bf16x8* __restrict__ wp, int& consumed, int last_step, Epi epi) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (N == 64) {
        const int col0 = 16 * (wave & 3);
        gemm_layer<true, NRT, RING, AFD>(lds, in_off, in_ld, zero_off, K, rowbits, ring, wp, consumed, last_step, [&](int tile, const f32x4& acc) { epi(col0, tile, acc); });
    } else {
#pragma nounroll
        for (int c = 0; c < N / 128; ++c) {
            const int col0 = 16 * wave + 128 * c;
            gemm_layer<false, NRT, RING, AFD>(lds, in_off, in_ld, zero_off, K, rowbits, ring, wp, consumed, last_step, [&](int tile, const f32x4& acc) { epi(col0, tile, acc); });
        }
    }
}

// input rows -> LDS as bf16 (+ one sign bit per element).  UN chunks (8 values) per thread and trip, all their loads in flight
// together.  SLAB: the producer GEMM left fp32 split-K slabs: summed in slab order, bias + LeakyReLU applied here.
// (Takes what it needs by value: handed the kernel-argument struct by reference, the UN = 5 form made the compiler keep a copy
// of the whole struct in scratch.)
struct StageIn {
    const uint16_t* a_in_b;
    const float* slab_base;
    const float* in_bias;
    size_t slab_stride;
    int nslab, in_bias_ld, K0, off0, ld0, offm, ldm, rows_wg;      // rows_wg: rows of the workgroup image (16 per row tile)
};
// the bf16 path: five chunks per thread and trip (80 rows x 256 channels = one trip), written without arrays
__device__ __forceinline__ void stage_input_bf16(const StageIn si, unsigned char* lds, int tid, int R, size_t row0) {
    const int K0 = si.K0, cpr = K0 / 8, nchunk = si.rows_wg * cpr;
    auto ld = [&](int idx) -> u32x4 {
        const int r = idx / cpr, c8 = (idx - r * cpr) * 8;
        if (idx < nchunk && r < R) return *reinterpret_cast<const u32x4*>(si.a_in_b + (row0 + r) * K0 + c8);
        return u32x4{0u, 0u, 0u, 0u};
    };
    auto st = [&](int idx, const u32x4& o) {
        if (idx >= nchunk) return;
        const int r = idx / cpr, c8 = (idx - r * cpr) * 8;
        *reinterpret_cast<u32x4*>(lds + si.off0 + r * si.ld0 + c8 * 2) = o;
        lds[si.offm + r * si.ldm + (c8 >> 3)] = (unsigned char)(pos_bits(o[0], o[1]) | (pos_bits(o[2], o[3]) << 4));
    };
#pragma nounroll
    for (int u0 = tid; u0 < nchunk; u0 += 5 * THREADS) {
        const u32x4 o0 = ld(u0), o1 = ld(u0 + THREADS), o2 = ld(u0 + 2 * THREADS), o3 = ld(u0 + 3 * THREADS), o4 = ld(u0 + 4 * THREADS);
        st(u0, o0); st(u0 + THREADS, o1); st(u0 + 2 * THREADS, o2); st(u0 + 3 * THREADS, o3); st(u0 + 4 * THREADS, o4);
    }
}
template <bool SLAB, int UN>
__device__ __forceinline__ void stage_input(const StageIn si, unsigned char* lds, int tid, int T, int R, size_t row0) {
    const int K0 = si.K0, cpr = K0 / 8, nchunk = si.rows_wg * cpr;
    for (int u0 = 0; u0 < nchunk; u0 += UN * THREADS) {
        int r[UN], c8[UN];
        bool ok[UN];
        u32x4 o[UN];
#pragma unroll
        for (int k = 0; k < UN; ++k) {
            const int idx = u0 + k * THREADS + tid;
            r[k] = idx / cpr; c8[k] = (idx - r[k] * cpr) * 8;
            ok[k] = idx < nchunk && r[k] < R;
            o[k] = u32x4{0u, 0u, 0u, 0u};
        }
        if (SLAB) {
            // (the first two slabs and the bias are requested together: ONE round trip for the usual cut in two, not three)
            f32x4 v0[UN], v1[UN], s0[UN], s1[UN], b0[UN], b1[UN];
            const size_t second = si.nslab > 1 ? si.slab_stride : 0;
#pragma unroll
            for (int k = 0; k < UN; ++k) {
                const float* p = si.slab_base + (row0 + (ok[k] ? r[k] : 0)) * K0 + (ok[k] ? c8[k] : 0);
                v0[k] = *reinterpret_cast<const f32x4*>(p); v1[k] = *reinterpret_cast<const f32x4*>(p + 4);
                s0[k] = *reinterpret_cast<const f32x4*>(p + second); s1[k] = *reinterpret_cast<const f32x4*>(p + second + 4);
                const float* bp = si.in_bias + (ok[k] ? (r[k] % T) * si.in_bias_ld + c8[k] : 0);
                b0[k] = *reinterpret_cast<const f32x4*>(bp); b1[k] = *reinterpret_cast<const f32x4*>(bp + 4);
            }
            if (si.nslab > 1) {
#pragma unroll
                for (int k = 0; k < UN; ++k) { v0[k] += s0[k]; v1[k] += s1[k]; }
            }
            for (int z = 2; z < si.nslab; ++z) {
                f32x4 t0[UN], t1[UN];
#pragma unroll
                for (int k = 0; k < UN; ++k) {
                    const float* p = si.slab_base + (size_t)z * si.slab_stride + (row0 + (ok[k] ? r[k] : 0)) * K0 + (ok[k] ? c8[k] : 0);
                    t0[k] = *reinterpret_cast<const f32x4*>(p); t1[k] = *reinterpret_cast<const f32x4*>(p + 4);
                }
#pragma unroll
                for (int k = 0; k < UN; ++k) { v0[k] += t0[k]; v1[k] += t1[k]; }
            }
#pragma unroll
            for (int k = 0; k < UN; ++k) {
                v0[k] += b0[k]; v1[k] += b1[k];
                float v[8] = {v0[k][0], v0[k][1], v0[k][2], v0[k][3], v1[k][0], v1[k][1], v1[k][2], v1[k][3]};
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = lrelu(v[e]);
                if (ok[k]) o[k] = u32x4{pack2(v[0], v[1]), pack2(v[2], v[3]), pack2(v[4], v[5]), pack2(v[6], v[7])};
            }
        } else {
#pragma unroll
            for (int k = 0; k < UN; ++k)
                if (ok[k]) o[k] = *reinterpret_cast<const u32x4*>(si.a_in_b + (row0 + r[k]) * K0 + c8[k]);
        }
#pragma unroll
        for (int k = 0; k < UN; ++k) {
            if (u0 + k * THREADS + tid >= nchunk) continue;
            *reinterpret_cast<u32x4*>(lds + si.off0 + r[k] * si.ld0 + c8[k] * 2) = o[k];
            lds[si.offm + r[k] * si.ldm + (c8[k] >> 3)] = (unsigned char)(pos_bits(o[k][0], o[k][1]) | (pos_bits(o[k][2], o[k][3]) << 4));
        }
    }
}

// DENSE: built for TWO workgroups per CU (<= 128 VGPRs: a ring of three weight fragments, activation fragments refreshed in place);
// otherwise for one (launches of at most one workgroup per CU, where nothing but the wave's own run-ahead covers a latency: ring of
// six, double-buffered activation fragments).  Same LDS plan, same arithmetic in the same order: bitwise the same results.
// PROBE: per-phase timestamps of workgroup 0 into a.dbg_ts (tools/tail16_bench); the product launches the probe-free instances.
// NRT: row tiles per workgroup (1 .. 5 = 1, 3, 4, 6, 8 windows of 10 frames): see tail_bf16_row_tiles -- 1536 windows are 192
// workgroups of 8 (a quarter of the chip idle), 256 of 6 (every CU) or 512 of 3 (two per CU: the fastest).
#ifdef GEM_TB_DEBUG_DUMP
__device__ uint16_t* g_tb_dump_gd = nullptr;
#endif
template <bool DENSE, int NRT, bool PROBE>
__global__ __launch_bounds__(THREADS, DENSE ? 4 : 2) void decoder_tail_bf16_kernel(TailB16Args a) {
    constexpr int RING = DENSE ? 3 : 6, AFD = DENSE ? 1 : 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r16 = lane & 15, q = lane >> 4;
    const int T = a.e.T, NL = a.n;
    int probe = 0;
#define TB_PROBE()                                                                                     \
    if (PROBE && a.dbg_ts && blockIdx.x == 0 && tid == 0) {                                            \
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

    // ---- the wave's weight stream: first six fragments requested before anything else
    // (uniform base + lane: the fragment address is an SGPR pair plus a 32-bit lane offset, no 64-bit vector arithmetic per load)
    const bf16x8* wp = reinterpret_cast<const bf16x8*>(a.wstream) + (size_t)wave * a.steps_total * 64;
    // (the forward layers never request past the last forward fragment: the ring is re-primed with the adjoint layers' first
    // fragments behind the energy terms, whose registers it would otherwise occupy -- 24 of 128)
    int last_step = a.steps_f - 1;
    bf16x8 ring[RING];
#pragma unroll
    for (int i = 0; i < RING; ++i) ring[i] = wp[(size_t)min(i, last_step) * 64 + lane];
    int consumed = 0;
    // this lane's rows 16 * tile + (lane & 15): existence and window-edge flags, once for all layers
    unsigned int rowbits = 0;
#pragma unroll
    for (int i = 0; i < NRT; ++i) {
        const int row = 16 * i + r16, t = row % T;
        rowbits |= ((row < R ? 1u : 0u) | (t > 0 ? 2u : 0u) | (t < T - 1 ? 4u : 0u)) << (3 * i);
    }
    const bool fast_e = T == 10 && a.e.J == 15;
    const int bwin = __builtin_amdgcn_readfirstlane((wave < nwin) ? (a.e.perm ? a.e.perm[w0 + wave] : w0 + wave) : 0);
    TB_PROBE();

    // ---- stage the input rows as bf16 (+ one sign bit per element: the LeakyReLU' mask of the last adjoint layer)
    {
        StageIn si;
        si.a_in_b = a.a_in_b; si.slab_base = a.in_slab.base; si.in_bias = a.in_bias; si.in_bias_ld = a.in_bias_ld;
        si.nslab = 0; si.slab_stride = 0;
        si.rows_wg = 16 * NRT; si.K0 = a.fwd[0].K; si.off0 = a.off_act[0]; si.ld0 = a.ld_act[0]; si.offm = a.off_mask[0]; si.ldm = a.ld_mask[0];
        if (a.in_slab.base) {
            slab_layout(a.in_slab, si.nslab, si.slab_stride);
            stage_input<true, 2>(si, lds, tid, T, R, row0);
        } else {
            stage_input_bf16(si, lds, tid, R, row0);
        }
    }
    for (int i = tid; i < ZERO_BYTES / 4; i += THREADS) reinterpret_cast<unsigned int*>(lds + a.off_zero)[i] = 0u;
    if (tid < a.e.J * MAXJ) reinterpret_cast<int*>(lds + a.off_tab)[tid] = a.e.children[tid];
    if (tid < a.e.J) reinterpret_cast<int*>(lds + a.off_tab)[MAXJ * MAXJ + tid] = a.e.parents[tid];
    if (tid < nwin) reinterpret_cast<int*>(lds + a.off_bwin)[tid] = a.e.perm ? a.e.perm[w0 + tid] : w0 + tid;
    if (!a.forward_only && tid < nwin * a.e.J) {
        const int wi = tid / a.e.J, j = tid - wi * a.e.J;
        const int bw = a.e.perm ? a.e.perm[w0 + wi] : w0 + wi;
        reinterpret_cast<float*>(lds + a.off_mb)[wi * MAXJ + j] = a.e.mean_bone[(size_t)bw * a.e.J + j];
    }
    lds_barrier();
    TB_PROBE();

    // ---- forward layers but the last (the layer loops are NOT unrolled: one body's registers at a time)
#pragma nounroll
    for (int i = 0; i + 1 < NL; ++i) {
        const int N = a.fwd[i].N;
        const float* bias = a.fwd[i].bias;
        const int out_off = a.off_act[i + 1], out_ld = a.ld_act[i + 1];
        const int m_off = a.off_mask[i + 1], m_ld = a.ld_mask[i + 1];
        // (the bias of the wave's channels is requested BEFORE the K walk: a load issued in the epilogue would wait for itself and,
        // the vmcnt queue being in order, for the six weight fragments in flight behind it)
        const f32x4 bv0 = *reinterpret_cast<const f32x4*>(bias + (N == 64 ? 16 * (wave & 3) : 16 * wave) + 4 * q);
        const f32x4 bv1 = N > 128 ? *reinterpret_cast<const f32x4*>(bias + 16 * wave + 128 + 4 * q) : bv0;
        gemm_dispatch<NRT, RING, AFD>(lds, a.off_act[i], a.ld_act[i], a.off_zero, a.fwd[i].K, N, rowbits, ring, wp, consumed, last_step,
                      [&](int col0, int tile, const f32x4& acc) {
                          const int n0 = col0 + 4 * q, m = 16 * tile + r16;
                          f32x4 v = acc + (col0 >= 128 ? bv1 : bv0);
#pragma unroll
                          for (int e = 0; e < 4; ++e) v[e] = lrelu(v[e]);
                          const unsigned int d0 = pack2(v[0], v[1]), d1 = pack2(v[2], v[3]);
                          *reinterpret_cast<u32x2*>(lds + out_off + m * out_ld + n0 * 2) = u32x2{d0, d1};
                          lds[m_off + m * m_ld + (n0 >> 2)] = (unsigned char)pos_bits(d0, d1);
                      });
        lds_barrier();
        TB_PROBE();
    }
    // up to three row tiles (the whole workgroup computes the energy terms): the bf16 gradient rows w.r.t. the pose (act[NL]'s slot: the
    // buffer of act[NL - 2], dead since the barrier above) are zeroed here -- their pad columns stay zero, the energy terms write the
    // J*3 value columns behind the next barrier; more tiles: the wavefront that owns a window zeroes its rows itself
    if (NRT <= 3 && !a.forward_only) {
        const int ldg_b = a.ld_act[NL];
        for (int i = tid; i < R * (PAD / 4); i += THREADS) {
            const int r = i / (PAD / 4), c4 = (i - r * (PAD / 4)) * 4;
            *reinterpret_cast<unsigned long long*>(lds + a.off_act[NL] + r * ldg_b + c4 * 2) = 0ull;
        }
    }
    // ---- the last forward layer (64 padded channels: the pose, fp32, no activation)
    {
        const int i = NL - 1;
        const float* bias = a.fwd[i].bias;
        const int JC = a.e.J * 3;
        float* Xd = reinterpret_cast<float*>(lds + a.off_x);
        float* Xp = a.Xp;
        const int col0 = 16 * (wave & 3);
        const f32x4 bv = *reinterpret_cast<const f32x4*>(bias + col0 + 4 * q);
        gemm_layer<true, NRT, RING, AFD>(lds, a.off_act[i], a.ld_act[i], a.off_zero, a.fwd[i].K, rowbits, ring, wp, consumed, last_step,
                         [&](int tile, const f32x4& acc) {
                             const int n0 = col0 + 4 * q, m = 16 * tile + r16;
                             const f32x4 v = acc + bv;
                             if (m < R) {
                                 // the pose itself: fp32, dense per window ([T][J*3]) for the energy terms
                                 const int wdw = m / T, t = m - wdw * T;
#pragma unroll
                                 for (int e = 0; e < 4; ++e)
                                     if (n0 + e < JC) Xd[wdw * a.escr + t * JC + n0 + e] = v[e];
                                 if (Xp) *reinterpret_cast<f32x4*>(Xp + (row0 + m) * PAD + n0) = v;
                             }
                         });
        lds_barrier();
        TB_PROBE();
    }
    if (a.forward_only) return;

    // ---- energy terms + dE/dX: one wavefront per window (fp32); gradient rows leave as bf16 into the buffer of act[NL]
    // Up to three row tiles (one, three or four windows of 10 x 15): the WHOLE workgroup on the windows' pairs (one or two trips of 512
    // threads instead of three trips of one wavefront per window); else one wavefront per window.  Same bits either way.
    {
        const int* ch = reinterpret_cast<const int*>(lds + a.off_tab);
        const int* par = ch + MAXJ * MAXJ;
        const int ldg = a.ld_act[NL] / 2;
        const float* xs0 = reinterpret_cast<const float*>(lds + a.off_x);
        const float* mbl0 = reinterpret_cast<const float*>(lds + a.off_mb);
        uint16_t* gd0 = reinterpret_cast<uint16_t*>(lds + a.off_act[NL]);
        constexpr int G10 = 16 * NRT / 10 < 8 ? 16 * NRT / 10 : 8;          // windows of ten frames per workgroup
        if (NRT <= 3 && fast_e && a.off_epair >= 0) {          // (plan_tail_bf16 reserves off_epair exactly for NRT <= 3)
            energy_pairs_wg<10, 15, (G10 * 150 + THREADS - 1) / THREADS, THREADS>(a.e, reinterpret_cast<const int*>(lds + a.off_bwin), nwin, tid, xs0,
                                                                                a.escr, mbl0, par, ch, gd0, T * ldg, ldg,
                                                                                reinterpret_cast<float*>(lds + a.off_epair), [] { lds_barrier(); });
        } else if (wave < nwin) {
            const int gz = NRT <= 3 ? 0 : PAD;
            if (fast_e) energy_pairs<10, 15, 3>(a.e, bwin, lane, xs0 + wave * a.escr, mbl0 + wave * MAXJ, par, ch, gd0 + wave * T * ldg, ldg, gz,
                                                (PROBE && a.dbg_ts && blockIdx.x == 0 && wave == 0) ? a.dbg_ts + 32 : nullptr);
            else energy_pairs<0, 0, 4>(a.e, bwin, lane, xs0 + wave * a.escr, mbl0 + wave * MAXJ, par, ch, gd0 + wave * T * ldg, ldg, gz);
        }
    }
    last_step = a.steps_total - 1;
#pragma unroll
    for (int i = 0; i < RING; ++i) ring[i] = wp[(size_t)min(consumed + i, last_step) * 64 + lane];
    lds_barrier();
    TB_PROBE();
#ifdef GEM_TB_DEBUG_DUMP          // developer harness only (tools/tail16_bench): the bf16 gradient rows w.r.t. the pose, as the energy terms left them
    if (g_tb_dump_gd)
        for (int i = tid; i < R * PAD; i += THREADS) {
            const int r = i / PAD, c = i - r * PAD;
            g_tb_dump_gd[(row0 + r) * PAD + c] = *reinterpret_cast<const uint16_t*>(lds + a.off_act[NL] + r * a.ld_act[NL] + c * 2);
        }
#endif

    // ---- backward-data layers (adjoint convs): gradient w.r.t. act[j] into the buffer of act[j], masked by LeakyReLU'(act[j])
#pragma nounroll
    for (int j = NL - 1; j >= 0; --j) {
        const int N = a.bwd[j].N;                      // = K of forward layer j = width of act[j]
        const int out_off = a.off_act[j], out_ld = a.ld_act[j];
        const int m_off = a.off_mask[j], m_ld = a.ld_mask[j];
        const bool masked = j > 0 || a.mask_first;
        gemm_dispatch<NRT, RING, AFD>(lds, a.off_act[j + 1], a.ld_act[j + 1], a.off_zero, a.bwd[j].K, N, rowbits, ring, wp, consumed, last_step,
                      [&](int col0, int tile, const f32x4& acc) {
                          const int n0 = col0 + 4 * q, m = 16 * tile + r16;
                          f32x4 v = acc;
                          if (masked) {
                              // bit e: act[j][m][n0 + e] > 0
                              const unsigned int pos = j > 0 ? (unsigned int)lds[m_off + m * m_ld + (n0 >> 2)]
                                                             : (unsigned int)lds[m_off + m * m_ld + (n0 >> 3)] >> (n0 & 7);
#pragma unroll
                              for (int e = 0; e < 4; ++e) v[e] *= ((pos >> e) & 1u) ? 1.f : LEAKY_SLOPE;
                          }
                          *reinterpret_cast<u32x2*>(lds + out_off + m * out_ld + n0 * 2) = u32x2{pack2(v[0], v[1]), pack2(v[2], v[3])};
                      });
        lds_barrier();
        TB_PROBE();
    }
    // ---- the staged gradient rows leave as whole rows (16 bytes per lane, 2 * K0 contiguous bytes per row)
    {
        const int K0 = a.fwd[0].K, cpr = K0 / 8, nchunk = R * cpr;
        for (int idx = tid; idx < nchunk; idx += THREADS) {
            const int r = idx / cpr, c8 = (idx - r * cpr) * 8;
            *reinterpret_cast<u32x4*>(a.g_out_b + (row0 + r) * K0 + c8) = *reinterpret_cast<const u32x4*>(lds + a.off_act[0] + r * a.ld_act[0] + c8 * 2);
        }
    }
    TB_PROBE();
#undef TB_PROBE
}

}  // namespace tb

// LDS plan of the bf16 tail for the chain starting at decoder conv `start`.  Returns the byte size, or 0 when the chain does not
// fit this kernel (layer widths other than 64 / 128 / 256 outputs, more than TB_MAX_LAYERS layers, more than 80 KB: the kernel is
// built for two workgroups per CU).
size_t plan_tail_bf16(const std::vector<Layer>& dec, int start, int T, int J, TailB16Args* out, int nrt) {
    const int n = (int)dec.size() - start;
    if (start < 1 || n < 1 || n > TB_MAX_LAYERS || T < 3 || T > 16 || nrt < 1 || nrt > tb::NRT_MAX || 16 * nrt < T) return 0;
    const int ROWS = 16 * nrt;
    for (int i = start; i < (int)dec.size(); ++i) {
        const int K = dec[i].K, N = dec[i].N;
        auto okK = [](int v) { return v == 64 || v == 128 || v == 256 || v == 512; };
        auto okN = [](int v) { return v == 64 || v == 128 || v == 256; };
        // forward: K_i -> N_i; adjoint: N_i -> K_i (so K_i must be a valid output width too, and N_i a valid K)
        if (!okK(K) || !okN(N) || !okN(K)) return 0;
    }
    if (dec.back().N != PAD || J * 3 > PAD || J > GEM_MAX_JOINTS) return 0;
    TailB16Args a{};
    a.n = n;
    a.mask_first = 1;
    a.nrt = nrt;
    a.G = std::min(8, ROWS / T);
    auto ld = [](int width) { return 2 * width + 32; };
    const int escr = (T * J * 3 + 3) / 4 * 4;
    a.escr = escr;
    // width of act[j] (j = 0 .. n; act[n] is the pose: its slot holds the bf16 gradient rows w.r.t. the pose)
    int width[TB_MAX_LAYERS + 1];
    width[0] = dec[start].K;
    for (int j = 1; j <= n; ++j) width[j] = dec[start + j - 1].N;
    // two ping-pong buffers: act[j] / the gradient w.r.t. act[j] live at the start of buffer j & 1; the decoded pose (fp32)
    // sits behind act[n-1] in that activation's buffer (read by the energy terms while they write the gradient rows into the other)
    int size[2] = {0, 0};
    for (int j = 0; j <= n; ++j) size[j & 1] = std::max(size[j & 1], ROWS * ld(width[j]));
    const int xb = (n - 1) & 1, x_rel = ROWS * ld(width[n - 1]);
    size[xb] = std::max(size[xb], x_rel + a.G * escr * 4);
    const int buf_off[2] = {0, size[0]};
    int off = size[0] + size[1];
    for (int j = 0; j <= n; ++j) { a.off_act[j] = buf_off[j & 1]; a.ld_act[j] = ld(width[j]); }
    a.off_x = buf_off[xb] + x_rel;
    for (int j = 0; j < n; ++j) {          // sign bits of act[j]
        a.off_mask[j] = off;
        a.ld_mask[j] = j == 0 ? width[0] / 8 : width[j] / 4;
        off += ROWS * a.ld_mask[j];
    }
    off = (off + 15) / 16 * 16;
    a.off_zero = off;
    off += tb::ZERO_BYTES;
    a.off_mb = off;                        // mean bone lengths of the workgroup's windows ([G][MAXJ] floats)
    off += a.G * GEM_MAX_JOINTS * 4;
    a.off_tab = off;                       // children lists of the skeleton ([J][MAXJ] ints) + parents ([MAXJ] ints)
    off += (GEM_MAX_JOINTS + 1) * GEM_MAX_JOINTS * 4;
    off = (off + 15) / 16 * 16;
    a.off_bwin = off;                      // global window index of each of the workgroup's windows
    off += 8 * 4;
    // up to three row tiles: the energy terms are computed by the whole workgroup, pairs dealt over all threads (energy_pairs_wg);
    // every pair parks its five terms here for the per-window reduction
    a.off_epair = -1;
    if (nrt <= 3) { a.off_epair = off; off += a.G * T * J * 5 * 4; }
    off = (off + 15) / 16 * 16 + 64;       // (+ slack: nothing reads past its row, this keeps it that way under edits)
    if (off > 80 * 1024) return 0;
    if (out) *out = a;
    return (size_t)off;
}

// Row tiles per workgroup for a launch of B windows (1, 3, 4, 6, 8 windows of 10 frames for 1 .. 5 tiles): the FEWEST that keep the
// launch inside one round of 2 x CUs workgroups (launch_tail_bf16: up to one workgroup per CU the low-latency instance -- ring of six,
// 160+ VGPRs --, beyond that the 128-VGPR instance, two per CU).  A workgroup's time is a chain of ~12 dependent phases that shrinks
// little with its rows; more, smaller workgroups use more CUs, and a second, independent workgroup on a CU fills the first one's gaps:
// 1536 windows as 512 workgroups of three windows take 37 us per launch, as 256 one-per-CU workgroups of six 42 us, as 192 of eight
// 45 us (round 4, `GEM_TAIL16_NRT`; 360 / 600 / 960 / 1200 / 1536 / 2040 / 2580 windows: +5 / +3 / +1 / +4 / +4 / +2 / +5 % windows/s
// over round 4's first rule "fewest of 3 .. 5 tiles with one workgroup per CU, else five").  More than 2 x CUs workgroups even with
// eight windows each: five tiles, several rounds.
int tail_bf16_row_tiles(const gem_handle* h, int B, int T) {
    if (const char* f = dev_env("GEM_TAIL16_NRT")) return atoi(f);          // developer override (A/B runs, tests)
    auto wgs = [&](int nrt) { const int G = std::min(8, 16 * nrt / T); return G >= 1 ? (B + G - 1) / G : 1 << 30; };
    for (int nrt = 1; nrt <= tb::NRT_MAX; ++nrt)
        if (wgs(nrt) <= 2 * h->n_cu) return nrt;
    return tb::NRT_MAX;
}

// Weight fragments in consumption order.  Must mirror gemm_dispatch / gemm_layer: one pass per 128 output channels (column tile c
// of wave w starts at channel 16 * w + 128 * c; 64-wide layers: 16 * (w & 3)), inside a pass k-block major, taps inner; lane l
// holds W[tap][n0 + (l & 15)][32 * kb + 8 * (l >> 4) + 0..7].
int build_tail_bf16_stream(gem_handle* h, StageNet& net) {
    net.tb_stream = nullptr; net.tb_steps_f = net.tb_steps_b = 0; net.tb_lds = 0;
    const int st = net.tail_start;
    if (st < 1) return 0;
    const size_t lds = plan_tail_bf16(net.dec, st, h->T, h->J, nullptr, tb::NRT_MAX);
    if (!lds) return 0;
    const int n = (int)net.dec.size() - st;
    auto steps_of = [](const Layer& L) { const int cpw = L.N / 128 > 0 ? L.N / 128 : 1; return (3 * L.K / 32) * cpw; };
    int sf = 0, sb = 0;
    for (int i = 0; i < n; ++i) { sf += steps_of(net.dec[st + i]); sb += steps_of(net.dec_bwd[st + i]); }
    const int total = sf + sb;
    std::vector<uint16_t> stream((size_t)tb::WAVES * total * 64 * 8, 0);
    auto f2bf = [](float x) { uint32_t u; std::memcpy(&u, &x, 4); return (uint16_t)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16); };
    for (int w = 0; w < tb::WAVES; ++w) {
        size_t step = (size_t)w * total;
        auto emit = [&](const Layer& L, const std::vector<float>& W /* [3][N][K] */) {
            const int cpw = L.N / 128 > 0 ? L.N / 128 : 1;
            for (int c = 0; c < cpw; ++c)
                for (int kb = 0; kb < L.K / 32; ++kb)
                    for (int tap = 0; tap < 3; ++tap) {
                        const int n0 = L.N == 64 ? 16 * (w & 3) : 16 * w + 128 * c;
                        uint16_t* dst = stream.data() + step * 64 * 8;
                        for (int l = 0; l < 64; ++l)
                            for (int e = 0; e < 8; ++e)
                                dst[l * 8 + e] = f2bf(W[((size_t)tap * L.N + n0 + (l & 15)) * L.K + 32 * kb + 8 * (l >> 4) + e]);
                        ++step;
                    }
        };
        for (int i = 0; i < n; ++i) emit(net.dec[st + i], net.host_fwd[st + i]);
        for (int i = n - 1; i >= 0; --i) emit(net.dec_bwd[st + i], net.host_bwd[st + i]);
    }
    void* p = nullptr;
    GEM_HIP(hipMalloc(&p, stream.size() * sizeof(uint16_t)));
    net.allocs.push_back(p);
    GEM_HIP(hipMemcpy(p, stream.data(), stream.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
    net.tb_stream = static_cast<uint16_t*>(p);
    net.tb_steps_f = sf; net.tb_steps_b = sb; net.tb_lds = lds;
    return 0;
}

int launch_tail_bf16(gem_handle* h, const TailB16Args& a, size_t lds_bytes, hipStream_t s) {
    // at most one workgroup per CU: the one-workgroup-per-CU instances (lower latency; 5, 4 or 3 row tiles as planned by the
    // caller: tail_bf16_row_tiles); more: the two-per-CU instance
    const int wgs = (a.B + a.G - 1) / a.G;
    const bool dense = wgs > h->n_cu && !dev_env("GEM_TAIL16_SPARSE");
    typedef void (*kern_t)(TailB16Args);
    kern_t kern = nullptr;
    if (dense) {
        switch (a.nrt) {
            case 5: kern = a.dbg_ts ? tb::decoder_tail_bf16_kernel<true, 5, true> : tb::decoder_tail_bf16_kernel<true, 5, false>; break;
            case 4: kern = tb::decoder_tail_bf16_kernel<true, 4, false>; break;
            case 3: kern = tb::decoder_tail_bf16_kernel<true, 3, false>; break;
            case 2: kern = a.dbg_ts ? tb::decoder_tail_bf16_kernel<true, 2, true> : tb::decoder_tail_bf16_kernel<true, 2, false>; break;
            case 1: kern = tb::decoder_tail_bf16_kernel<true, 1, false>; break;
        }
    } else {
        switch (a.nrt) {
            case 5: kern = a.dbg_ts ? tb::decoder_tail_bf16_kernel<false, 5, true> : tb::decoder_tail_bf16_kernel<false, 5, false>; break;
            case 4: kern = tb::decoder_tail_bf16_kernel<false, 4, false>; break;
            case 3: kern = tb::decoder_tail_bf16_kernel<false, 3, false>; break;
            case 2: kern = tb::decoder_tail_bf16_kernel<false, 2, false>; break;
            case 1: kern = a.dbg_ts ? tb::decoder_tail_bf16_kernel<false, 1, true> : tb::decoder_tail_bf16_kernel<false, 1, false>; break;
        }
    }
    if (!kern) { set_error("launch_tail_bf16: unsupported row-tile count"); return 1; }
    const void* kfn = reinterpret_cast<const void*>(kern);
    static PerDeviceOnce attr_once;
    if (attr_once.need(h->cfg.device)) {
        const kern_t all[] = {tb::decoder_tail_bf16_kernel<true, 5, true>, tb::decoder_tail_bf16_kernel<true, 5, false>, tb::decoder_tail_bf16_kernel<false, 5, true>,
                              tb::decoder_tail_bf16_kernel<false, 5, false>, tb::decoder_tail_bf16_kernel<false, 4, false>, tb::decoder_tail_bf16_kernel<false, 3, false>,
                              tb::decoder_tail_bf16_kernel<false, 2, false>, tb::decoder_tail_bf16_kernel<true, 4, false>, tb::decoder_tail_bf16_kernel<true, 3, false>,
                              tb::decoder_tail_bf16_kernel<true, 2, false>, tb::decoder_tail_bf16_kernel<false, 1, false>, tb::decoder_tail_bf16_kernel<true, 1, false>,
                              tb::decoder_tail_bf16_kernel<true, 2, true>, tb::decoder_tail_bf16_kernel<false, 1, true>};
        for (kern_t k : all) GEM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
    }
    if (a.n < 1 || a.n > TB_MAX_LAYERS || lds_bytes > 80 * 1024) { set_error("launch_tail_bf16: unsupported layer chain"); return 1; }
    Profile::Rec rec;
    const bool prof = h->prof.on;
    if (prof) {
        GEM_HIP(hipEventCreate(&rec.a)); GEM_HIP(hipEventCreate(&rec.b));
        rec.family = 1;
        double per_window = 0.0;          // matrix work on the T real rows of a window, forward + adjoint
        for (int i = 0; i < a.n; ++i) per_window += 2.0 * 3.0 * a.fwd[i].K * a.fwd[i].N * a.e.T;
        if (!a.forward_only) per_window *= 2.0;
        rec.flops = per_window * a.B;
        if (h->ws.dyn) { rec.log_idx = h->ws.cur_log; rec.flops_per_window = per_window; }
        GEM_HIP(hipEventRecord(rec.a, s));
    }
    note_kernel(h, kfn);
    hipLaunchKernelGGL(kern, dim3(wgs), dim3(tb::THREADS), lds_bytes, s, a);
    GEM_HIP(hipGetLastError());
    if (prof) { GEM_HIP(hipEventRecord(rec.b, s)); h->prof.recs.push_back(rec); }
    commit_kernel_names(h, prof ? 1 : -1);
    return 0;
}

}  // namespace gem

#ifdef GEM_TB_DEBUG_DUMP          // developer builds only (tools/r05_nrt_dump.py): where the kernels copy the pose-gradient rows to
extern "C" int gem_debug_set_tb_dump(void* d_rows) {
    return hipMemcpyToSymbol(HIP_SYMBOL(gem::tb::g_tb_dump_gd), &d_rows, sizeof(d_rows)) == hipSuccess ? 0 : 1;
}
#endif
