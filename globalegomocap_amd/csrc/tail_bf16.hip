// bf16 multi-window fused decoder tail (gfx950): the narrow temporal convs of the decoder, the energy terms and the matching
// backward-data convs of EIGHT windows in one workgroup -- the "bf16 VAE decoder / fp32 energy" mode of BASELINE configs[2..4].
//
// The fp32 tail (tail.hip) gives one workgroup to one window (10 of 16 tile rows used) and streams 1.3 MB of fp32 weights per
// window from L2; beyond ~1300 windows the narrow layers therefore ran as ~12 batched bf16 GEMM launches of 7-27 us plus the
// stand-alone energy kernel per evaluation round.  Here one workgroup (8 waves) owns G = min(8, 80 / T) windows = up to 80 rows
// = FIVE full 16-row MFMA tiles:
//
//   input rows (bf16 matrix, or the producer GEMM's fp32 split-K slabs: summed + bias + LeakyReLU on the way in) -> LDS (bf16)
//   for each fused layer:  act[i+1] = lrelu(conv3(act[i]) + b)      v_mfma_f32_16x16x32_bf16, fp32 accumulate, bf16 in LDS
//   X = act[n] (fp32) -> energy terms + dE/dX per window, one wavefront per window, fp32 (energy_device.h)
//   backward-data through the same layers, LeakyReLU' from the sign of the LDS activations
//   -> gradient w.r.t. the input's pre-activation, bf16, staged through LDS and written out as whole rows.
//
// MFMA operands: the WEIGHT fragment is operand A (rows n), the activation fragment operand B (columns m): D[n][m], so a lane
// ends up with 4 consecutive output channels of ONE row = one 8-byte bf16 store into the next layer's LDS image.  Activation
// rows are 2*width + 32 bytes apart: with that stride the 16-byte fragment reads (lane l: row l & 15, 16-byte chunk l >> 4) of
// ds_read_b128's four lane groups hit 16 distinct 16-byte slots each -- conflict-free without a swizzle, and the K walk is an
// immediate offset.  The k=3 conv reads rows t-1, t, t+1 of the same window; rows outside it read a zero line (address select).
// Weights never touch LDS: at load time every layer is cut into the 1 KB fragments (64 lanes x 8 bf16) each wave will need, in
// the order it will need them (build_tail_bf16_stream), so a wave walks ONE contiguous stream -- forward layers, then adjoint
// layers -- with a ring of six fragments in registers, requested six K-steps (~500 MFMA cycles) ahead, across layer boundaries,
// barriers and the energy phase.  Wave w owns output channels 16w..16w+15 (+128 per extra tile) and all five row tiles; in a
// 64-wide layer waves w and w+4 share a channel tile and split the row tiles 3 + 2.
//
// Reference semantics: ConvTranspose1d/Conv1d k=3 s=1 p=1 + BatchNorm(eval) + LeakyReLU of networks/models/SeqConvVAE.py:76-92
// (folded at load time), decode_to_bodypose :131-140, total_loss of optimizer.py:226-240, backward-DATA only (frozen VAE).
#include <algorithm>
#include <cstring>

#include "energy_device.h"

namespace gem {

namespace tb {

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef short bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int WAVES = 8, THREADS = WAVES * 64;
constexpr int NRT = 5, ROWS = NRT * 16;        // row tiles / rows per workgroup
constexpr int RING = 6;                        // weight fragments in flight per wave
constexpr int ZERO_BYTES = 1024 + 64;          // the zero line covers the K walk of the widest layer (K = 512: 1024 bytes)

__device__ __forceinline__ unsigned int pack2(float lo, float hi) {      // round-to-nearest-even (v_cvt_pk_bf16_f32)
    typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
    typedef float f2 __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(unsigned int, __builtin_convertvector(f2{lo, hi}, bf2));
}
__device__ __forceinline__ float lrelu(float v) { return v > 0.f ? v : v * LEAKY_SLOPE; }

// LDS writes of this wave done, then the workgroup barrier.  (Not __syncthreads: the weight ring's global loads stay in
// flight across the barrier.)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// One fused layer for this wave: D[n][m] += W[n][k] . act[m][k] over 3 taps x K.
//   CPW   column tiles (16 channels) per wave: N / 128, at least 1
//   SPLIT 64-wide layer: waves w and w + 4 share column tile w & 3; w < 4 takes row tiles 0-2, w >= 4 row tiles 3-4
// The K walk is k-block major, taps inner (the weight stream is packed in the same order): a body of 6 k-steps covers two whole
// k-blocks, so taps and ring slots are compile-time.  epi(c, tile, acc) gets the accumulator of column tile c
// (channels 16 * ct + 4 * (lane >> 4) + 0..3) and row tile `tile` (row 16 * tile + (lane & 15)).
template <int CPW, bool SPLIT, typename Epi>
__device__ __forceinline__ void gemm_layer(const unsigned char* lds, int in_off, int in_ld, int zero_off, int K, int T, int R,
                                           bf16x8 (&ring)[RING], const bf16x8* __restrict__ wp, int& consumed, int last_step, Epi epi) {
    constexpr int NR = SPLIT ? 3 : NRT;
    static_assert(CPW == 1 || CPW == 2, "one or two column tiles per wave");
    constexpr int KS = 6;                             // k-steps (32 deep) per unrolled body = two whole k-blocks x 3 taps (even: the
    constexpr int U = KS * CPW;                       // register double buffer of the activation fragments keeps its parity)
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r16 = lane & 15, q = lane >> 4;
    const int tile0 = SPLIT ? 3 * (wave >> 2) : 0;
    // fragment read addresses per tap and row tile (bytes); invalid rows (outside the window: the conv's zero padding; past R;
    // the absent sixth tile of the split) read the zero line -- selecting the ADDRESS keeps the reads in flight behind the MFMAs
    int base[3][NR];
#pragma unroll
    for (int i = 0; i < NR; ++i) {
        const int row = 16 * (tile0 + i) + r16;
        const int t = row % T;
        const bool ok = (tile0 + i) < NRT && row < R;
#pragma unroll
        for (int tap = 0; tap < 3; ++tap) {
            const int tt = t + tap - 1;
            base[tap][i] = (ok && tt >= 0 && tt < T) ? in_off + (row + tap - 1) * in_ld + q * 16 : zero_off + q * 16;
        }
    }
    f32x4 acc[CPW][NR];
#pragma unroll
    for (int c = 0; c < CPW; ++c)
#pragma unroll
        for (int i = 0; i < NR; ++i) acc[c][i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int nsteps = (3 * K / 32) * CPW;
    // Program order = issue order (sched_barrier between the groups; left alone, the compiler sinks all six ring refills to the
    // end of the body, which halves the distance the stream runs ahead): every MFMA is followed by the fragment read of the
    // NEXT k-step for the same row tile, every k-step by the refill of the ring slot it has just used.  The activation
    // fragments are double-buffered in registers; the body's last k-step reads the first fragments of the next body (past the
    // last k-block of the layer that is 16 bytes of row padding / the neighbouring row: read, never used).
    bf16x8 af[2][NR];
#pragma unroll
    for (int i = 0; i < NR; ++i) af[0][i] = *reinterpret_cast<const bf16x8*>(lds + base[0][i]);
    for (int s0 = 0; s0 < nsteps; s0 += U) {
#pragma unroll
        for (int j = 0; j < KS; ++j) {                // k-step j of the body: tap j % 3 of k-block j / 3
            const int jn = j + 1;                     // next k-step (jn == KS: tap 0 of the next body's first k-block)
#pragma unroll
            for (int c = 0; c < CPW; ++c) {
                const int u = j * CPW + c;            // step of the body (compile time after unrolling)
#pragma unroll
                for (int i = 0; i < NR; ++i) {
                    acc[c][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ring[u % RING], af[j & 1][i], acc[c][i], 0, 0, 0);
                    if (c == CPW - 1) af[jn & 1][i] = *reinterpret_cast<const bf16x8*>(lds + base[jn % 3][i] + (jn / 3) * 64);
                    __builtin_amdgcn_sched_barrier(0);
                }
                // refill the slot six steps ahead (past the end of the stream: a harmless re-load of the last fragment)
                ring[u % RING] = wp[(size_t)min(consumed + s0 + u + RING, last_step) * 64];
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#pragma unroll
        for (int tap = 0; tap < 3; ++tap)
#pragma unroll
            for (int i = 0; i < NR; ++i) base[tap][i] += (KS / 3) * 64;
    }
    consumed += nsteps;
#pragma unroll
    for (int c = 0; c < CPW; ++c)
#pragma unroll
        for (int i = 0; i < NR; ++i)
            if (tile0 + i < NRT) epi(c, tile0 + i, acc[c][i]);
}

template <typename Epi>
__device__ __forceinline__ void gemm_dispatch(const unsigned char* lds, int in_off, int in_ld, int zero_off, int K, int N, int T, int R,
                                              bf16x8 (&ring)[RING], const bf16x8* __restrict__ wp, int& consumed, int last_step, Epi epi) {
    // N is 64, 128 or 256 (plan_tail_bf16)
    if (N == 64) gemm_layer<1, true>(lds, in_off, in_ld, zero_off, K, T, R, ring, wp, consumed, last_step, epi);
    else if (N == 128) gemm_layer<1, false>(lds, in_off, in_ld, zero_off, K, T, R, ring, wp, consumed, last_step, epi);
    else gemm_layer<2, false>(lds, in_off, in_ld, zero_off, K, T, R, ring, wp, consumed, last_step, epi);
}

// first output channel of column tile c of this wave in a layer of N channels
__device__ __forceinline__ int col0_of(int N, int wave, int c) { return N == 64 ? 16 * (wave & 3) : 16 * wave + 128 * c; }

template <int NL>
__global__ __launch_bounds__(THREADS, 2) void decoder_tail_bf16_kernel(TailB16Args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r16 = lane & 15, q = lane >> 4;
    const int T = a.e.T;
    int probe = 0;
#define TB_PROBE()                                                                                     \
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

    // ---- the wave's weight stream: first six fragments requested before anything else
    const bf16x8* wp = reinterpret_cast<const bf16x8*>(a.wstream) + (size_t)wave * a.steps_total * 64 + lane;
    const int last_step = (a.forward_only ? a.steps_f : a.steps_total) - 1;
    bf16x8 ring[RING];
#pragma unroll
    for (int i = 0; i < RING; ++i) ring[i] = wp[(size_t)min(i, last_step) * 64];
    int consumed = 0;
    // ---- what the energy terms need besides the decoded pose (stage-input pose, bone lengths, cached texel blocks of this
    // wave's window) is requested now and stays in registers across the forward layers: the energy phase then starts without
    // a global round trip.  (The usual window shape, 10 frames x 15 joints, has compile-time index arithmetic as well.)
    const bool fast_e = T == 10 && a.e.J == 15;
    EnergyPre<8, 3> pre;
    const int bwin = (wave < nwin) ? (a.e.perm ? a.e.perm[w0 + wave] : w0 + wave) : 0;
    if (fast_e && !a.forward_only && wave < nwin) energy_prefetch<10, 15>(a.e, bwin, lane, pre);
    TB_PROBE();

    // ---- stage the input rows as bf16 (+ one sign bit per element: the LeakyReLU' mask of the last adjoint layer; the region
    // itself is reused by the energy terms and the output staging).  In the rounds the producer GEMM leaves fp32 split-K slabs:
    // summed in slab order, bias + LeakyReLU applied here.
    {
        const int K0 = a.fwd[0].K, cpr = K0 / 8, nchunk = ROWS * cpr;
        int nslab = 0;
        size_t stride = 0;
        if (a.in_slab.base) slab_layout(a.in_slab, nslab, stride);
        constexpr int UN = 5;                  // chunks (8 values) per thread and trip: all their loads in flight together
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
            if (a.in_slab.base) {
                f32x4 v0[UN], v1[UN], b0[UN], b1[UN];
#pragma unroll
                for (int k = 0; k < UN; ++k) {
                    const float* p = a.in_slab.base + (row0 + (ok[k] ? r[k] : 0)) * K0 + (ok[k] ? c8[k] : 0);
                    const float* bp = a.in_bias + (ok[k] ? (r[k] % T) * a.in_bias_ld + c8[k] : 0);
                    v0[k] = *reinterpret_cast<const f32x4*>(p); v1[k] = *reinterpret_cast<const f32x4*>(p + 4);
                    b0[k] = *reinterpret_cast<const f32x4*>(bp); b1[k] = *reinterpret_cast<const f32x4*>(bp + 4);
                }
                for (int z = 1; z < nslab; ++z) {
                    f32x4 t0[UN], t1[UN];
#pragma unroll
                    for (int k = 0; k < UN; ++k) {
                        const float* p = a.in_slab.base + (size_t)z * stride + (row0 + (ok[k] ? r[k] : 0)) * K0 + (ok[k] ? c8[k] : 0);
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
                    if (ok[k]) o[k] = *reinterpret_cast<const u32x4*>(a.a_in_b + (row0 + r[k]) * K0 + c8[k]);
            }
#pragma unroll
            for (int k = 0; k < UN; ++k) {
                if (u0 + k * THREADS + tid >= nchunk) continue;
                unsigned int bits = 0;
#pragma unroll
                for (int e = 0; e < 4; ++e) {          // bf16 > 0: sign bit clear and not zero
                    const unsigned int lo = o[k][e] & 0xFFFFu, hi = o[k][e] >> 16;
                    bits |= ((lo != 0u && lo < 0x8000u) ? 1u : 0u) << (2 * e);
                    bits |= ((hi != 0u && hi < 0x8000u) ? 1u : 0u) << (2 * e + 1);
                }
                *reinterpret_cast<u32x4*>(lds + a.off_act[0] + r[k] * a.ld_act[0] + c8[k] * 2) = o[k];
                lds[a.off_mask + r[k] * a.ld_mask + (c8[k] >> 3)] = (unsigned char)bits;
            }
        }
        for (int i = tid; i < ZERO_BYTES / 4; i += THREADS) reinterpret_cast<unsigned int*>(lds + a.off_zero)[i] = 0u;
        if (tid < a.e.J * MAXJ) reinterpret_cast<int*>(lds + a.off_tab)[tid] = a.e.children[tid];
    }
    lds_barrier();
    TB_PROBE();

    // ---- forward layers
#pragma unroll
    for (int i = 0; i < NL; ++i) {
        const bool last = (i + 1 == NL);
        const int N = a.fwd[i].N;
        const float* bias = a.fwd[i].bias;
        f32x4 bv[2];
#pragma unroll
        for (int c = 0; c < 2; ++c) bv[c] = (c == 0 || N > 128) ? *reinterpret_cast<const f32x4*>(bias + col0_of(N, wave, c) + 4 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
        const int out_off = a.off_act[i + 1], out_ld = a.ld_act[i + 1];
        const int JC = a.e.J * 3;
        float* Xd = reinterpret_cast<float*>(lds + a.off_x);
        float* Xp = a.Xp;
        gemm_dispatch(lds, a.off_act[i], a.ld_act[i], a.off_zero, a.fwd[i].K, N, T, R, ring, wp, consumed, last_step,
                      [&](int c, int tile, const f32x4& acc) {
                          const int n0 = col0_of(N, wave, c) + 4 * q, m = 16 * tile + r16;
                          f32x4 v = acc + bv[c];
                          if (!last) {
#pragma unroll
                              for (int e = 0; e < 4; ++e) v[e] = lrelu(v[e]);
                              *reinterpret_cast<u32x2*>(lds + out_off + m * out_ld + n0 * 2) = u32x2{pack2(v[0], v[1]), pack2(v[2], v[3])};
                          } else if (m < R) {
                              // the pose itself: fp32, dense per window ([T][J*3]) for the energy terms
                              const int wdw = m / T, t = m - wdw * T;
#pragma unroll
                              for (int e = 0; e < 4; ++e)
                                  if (n0 + e < JC) Xd[(wdw * T + t) * JC + n0 + e] = v[e];
                              if (Xp) *reinterpret_cast<f32x4*>(Xp + (row0 + m) * PAD + n0) = v;
                          }
                      });
        lds_barrier();
        TB_PROBE();
    }
    if (a.forward_only) return;

    // ---- energy terms + dE/dX: one wavefront per window (fp32); gradient rows leave as bf16 into g[NL & 1]
    {
        const int n = T * a.e.J * 3, p = NL & 1;
        if (wave < nwin) {
            float* xs = reinterpret_cast<float*>(lds + a.off_x) + wave * n;
            float* scr = reinterpret_cast<float*>(lds + a.off_escr) + wave * 3 * a.escr;
            const int ldg = a.ld_g[p] / 2;
            uint16_t* gd = reinterpret_cast<uint16_t*>(lds + a.off_g[p]) + wave * T * ldg;
            const int* ch = reinterpret_cast<const int*>(lds + a.off_tab);
            if (fast_e)
                energy_window<false, 64, 10, 15, true, EnergyPre<8, 3>>(a.e, bwin, lane, xs, 45, xs, scr, scr + a.escr, scr + 2 * a.escr, nullptr, ldg,
                                                                         a.fwd[NL - 1].N, gd, nullptr, nullptr, nullptr, ch, &pre);
            else
                energy_window<false, 64, 0, 0, true>(a.e, bwin, lane, xs, a.e.J * 3, xs, scr, scr + a.escr, scr + 2 * a.escr, nullptr, ldg,
                                                     a.fwd[NL - 1].N, gd, nullptr, nullptr, nullptr, ch);
        }
    }
    lds_barrier();
    TB_PROBE();

    // ---- backward-data layers (adjoint convs): gradient w.r.t. act[j] into g[j & 1], masked by LeakyReLU'(act[j])
#pragma unroll
    for (int j = NL - 1; j >= 0; --j) {
        const int N = a.bwd[j].N;                      // = K of forward layer j = width of act[j]
        const int gin = (j + 1) & 1, gout = j & 1;
        const int act_off = a.off_act[j], act_ld = a.ld_act[j];
        const int out_off = j > 0 ? a.off_g[gout] : a.off_act[0], out_ld = j > 0 ? a.ld_g[gout] : a.ld_act[0];
        const bool masked = j > 0 || a.mask_first;
        gemm_dispatch(lds, a.off_g[gin], a.ld_g[gin], a.off_zero, a.bwd[j].K, N, T, R, ring, wp, consumed, last_step,
                      [&](int c, int tile, const f32x4& acc) {
                          const int n0 = col0_of(N, wave, c) + 4 * q, m = 16 * tile + r16;
                          f32x4 v = acc;
                          if (masked) {
                              unsigned int pos;        // bit e: act[m][n0 + e] > 0
                              if (j > 0) {
                                  const u32x2 av = *reinterpret_cast<const u32x2*>(lds + act_off + m * act_ld + n0 * 2);
                                  pos = 0;
#pragma unroll
                                  for (int e = 0; e < 2; ++e) {
                                      const unsigned int lo = av[e] & 0xFFFFu, hi = av[e] >> 16;
                                      pos |= ((lo != 0u && lo < 0x8000u) ? 1u : 0u) << (2 * e);
                                      pos |= ((hi != 0u && hi < 0x8000u) ? 1u : 0u) << (2 * e + 1);
                                  }
                              } else {
                                  pos = (unsigned int)lds[a.off_mask + m * a.ld_mask + (n0 >> 3)] >> (n0 & 7);
                              }
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
// fit this kernel (layer widths other than 64 / 128 / 256 outputs, more than TB_MAX_LAYERS layers, more than 160 KB).
size_t plan_tail_bf16(const std::vector<Layer>& dec, int start, int T, int J, TailB16Args* out) {
    const int n = (int)dec.size() - start;
    if (start < 1 || n < 1 || n > TB_MAX_LAYERS || T < 3 || T > 16) return 0;
    for (int i = start; i < (int)dec.size(); ++i) {
        const int K = dec[i].K, N = dec[i].N;
        auto okK = [](int v) { return v == 64 || v == 128 || v == 256 || v == 512; };
        auto okN = [](int v) { return v == 64 || v == 128 || v == 256; };
        // forward: K_i -> N_i; adjoint: N_i -> K_i (so K_i must be a valid output width too, and N_i a valid K)
        if (!okK(K) || !okN(N) || !okN(K)) return 0;
    }
    if (dec.back().N != PAD || J * 3 > PAD) return 0;
    TailB16Args a{};
    a.n = n;
    a.mask_first = 1;
    a.G = std::min(8, tb::ROWS / T);
    auto ld = [](int width) { return 2 * width + 32; };
    int off = 0;
    const int escr = (T * J * 3 + 3) / 4 * 4;
    a.escr = escr;
    // region 0: the input image; later the energy scratch (3 arrays per window), later the staged output rows
    a.off_act[0] = 0;
    a.ld_act[0] = ld(dec[start].K);
    a.off_escr = 0;
    off = std::max(tb::ROWS * a.ld_act[0], a.G * 3 * escr * 4);
    for (int i = 1; i < n; ++i) {          // act[n] is the pose: kept as fp32 at off_x
        a.off_act[i] = off;
        a.ld_act[i] = ld(dec[start + i - 1].N);
        off += tb::ROWS * a.ld_act[i];
    }
    a.off_act[n] = 0; a.ld_act[n] = 0;
    a.off_x = off;
    off += a.G * escr * 4;
    // gradient w.r.t. act[j] (j = n .. 1) lives in g[j & 1]: each buffer as wide as the widest it ever holds
    int wg[2] = {0, 0};
    for (int j = 1; j <= n; ++j) wg[j & 1] = std::max(wg[j & 1], dec[start + j - 1].N);
    for (int p = 0; p < 2; ++p) {
        a.off_g[p] = off;
        a.ld_g[p] = ld(std::max(wg[p], 64));
        off += tb::ROWS * a.ld_g[p];
    }
    a.off_zero = off;
    off += tb::ZERO_BYTES;
    a.off_mask = off;
    a.ld_mask = dec[start].K / 8;
    off += tb::ROWS * a.ld_mask;
    off = (off + 15) / 16 * 16;
    a.off_tab = off;                       // children lists of the skeleton ([J][MAXJ] ints)
    off += GEM_MAX_JOINTS * GEM_MAX_JOINTS * 4;
    off = (off + 15) / 16 * 16 + 64;       // (+ slack: nothing reads past its row, this keeps it that way under edits)
    if (off > 160 * 1024) return 0;
    if (out) *out = a;
    return (size_t)off;
}

// Weight fragments in consumption order.  Must mirror gemm_layer: k-block major, taps inner, column tiles innermost; wave w's
// column tile c starts at channel col0_of(N, w, c); lane l holds W[tap][n0 + (l & 15)][32 * kb + 8 * (l >> 4) + 0..7].
int build_tail_bf16_stream(gem_handle* h, StageNet& net) {
    net.tb_stream = nullptr; net.tb_steps_f = net.tb_steps_b = 0; net.tb_lds = 0;
    const int st = net.tail_start;
    if (st < 1) return 0;
    const size_t lds = plan_tail_bf16(net.dec, st, h->T, h->J, nullptr);
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
            for (int kb = 0; kb < L.K / 32; ++kb)
                for (int tap = 0; tap < 3; ++tap)
                    for (int c = 0; c < cpw; ++c) {
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
    static PerDeviceOnce attr_once;
    if (attr_once.need(h->cfg.device)) {
        const void* ks[] = {reinterpret_cast<const void*>(tb::decoder_tail_bf16_kernel<1>), reinterpret_cast<const void*>(tb::decoder_tail_bf16_kernel<2>),
                            reinterpret_cast<const void*>(tb::decoder_tail_bf16_kernel<3>), reinterpret_cast<const void*>(tb::decoder_tail_bf16_kernel<4>),
                            reinterpret_cast<const void*>(tb::decoder_tail_bf16_kernel<5>), reinterpret_cast<const void*>(tb::decoder_tail_bf16_kernel<6>)};
        for (const void* k : ks) GEM_HIP(hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    }
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
    const int wgs = (a.B + a.G - 1) / a.G;
    switch (a.n) {
        case 1: note_kernel(h, reinterpret_cast<const void*>(tb::decoder_tail_bf16_kernel<1>)); hipLaunchKernelGGL(tb::decoder_tail_bf16_kernel<1>, dim3(wgs), dim3(tb::THREADS), lds_bytes, s, a); break;
        case 2: note_kernel(h, reinterpret_cast<const void*>(tb::decoder_tail_bf16_kernel<2>)); hipLaunchKernelGGL(tb::decoder_tail_bf16_kernel<2>, dim3(wgs), dim3(tb::THREADS), lds_bytes, s, a); break;
        case 3: note_kernel(h, reinterpret_cast<const void*>(tb::decoder_tail_bf16_kernel<3>)); hipLaunchKernelGGL(tb::decoder_tail_bf16_kernel<3>, dim3(wgs), dim3(tb::THREADS), lds_bytes, s, a); break;
        case 4: note_kernel(h, reinterpret_cast<const void*>(tb::decoder_tail_bf16_kernel<4>)); hipLaunchKernelGGL(tb::decoder_tail_bf16_kernel<4>, dim3(wgs), dim3(tb::THREADS), lds_bytes, s, a); break;
        case 5: note_kernel(h, reinterpret_cast<const void*>(tb::decoder_tail_bf16_kernel<5>)); hipLaunchKernelGGL(tb::decoder_tail_bf16_kernel<5>, dim3(wgs), dim3(tb::THREADS), lds_bytes, s, a); break;
        case 6: note_kernel(h, reinterpret_cast<const void*>(tb::decoder_tail_bf16_kernel<6>)); hipLaunchKernelGGL(tb::decoder_tail_bf16_kernel<6>, dim3(wgs), dim3(tb::THREADS), lds_bytes, s, a); break;
        default: set_error("launch_tail_bf16: unsupported number of fused layers"); return 1;
    }
    GEM_HIP(hipGetLastError());
    if (prof) { GEM_HIP(hipEventRecord(rec.b, s)); h->prof.recs.push_back(rec); }
    commit_kernel_names(h, prof ? 1 : -1);
    return 0;
}

}  // namespace gem
