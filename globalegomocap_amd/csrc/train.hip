// Training step of the motion VAE on the device: the loop body of networks/train.py:65-108 --
// forward in TRAIN mode (BatchNorm batch statistics, running statistics updated), the VAE loss of
// networks/models/SeqConvVAE.py:191-219 (M_N form: mean-squared reconstruction error + kld_weight * KL; kl_weight form: summed
// squared error, opts.recon_sum), backward (data AND
// weight gradients) and one torch.optim.Adam step (L2 weight decay folded into the gradient, bias-corrected moments).
//
// Parameters live in ONE fp32 arena in the padded, packed layouts the GEMM kernels read directly (every conv as its equivalent
// Conv1d taps [3][N][K], k contiguous; fc_mu | fc_var stacked; decoder_input time-major), with same-shaped arenas for gradient
// and the two Adam moments; the reference's checkpoint schema is a permutation of that arena (host side: vae_train.py).  Padded
// entries are zero and stay zero (their gradients are sums over zero activations).
//   conv products, forward and backward-data: conv_rows.h (one launch per product at every batch size: 32 x 32 tiles, the waves of
//     a workgroup split K, operands staged through wave-private LDS by LDS-DMA); the adjoint weight images of the CONV layers (3 %
//     of the parameters) are re-packed from the arena by ONE launch at the start of a step (adjoint_all_kernel)
//   the two linear layers (97 % of the parameters): forward through the optimiser's few-rows kernel (launch_gemm; K slabs summed by
//     the next kernel, which for fc also forms z); backward at batches up to 64 windows in the training-loop mode = ONE pass over the
//     weights (gemm_tn_adam_dx_kernel: weight gradient, Adam step and backward-data product per tile), otherwise backward-data
//     from the weights' own layout (linear_bwd_data) + weight gradient (+ Adam step: gemm_tn_adam_kernel)
//   conv weight gradients: dW[tap][n][k] = sum_r dC[r][n] * A[r + tap - 1][k] (contraction over the ROWS, both operands row-major:
//     staged through LDS, v_mfma_f32_16x16x4_f32), all layers in ONE launch behind the backward chain (gemm_tn3_all_kernel; every
//     BatchNorm layer keeps its dY), row range cut into slabs that ONE launch sums in slab order before Adam (slab_sum_all_kernel)
//   BatchNorm (+ LeakyReLU) forward / backward (incl. the conv bias gradient), the latent / loss gradients, Adam: one kernel each
// Everything is enqueued on the caller's stream; no host synchronisation inside a step.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <memory>

#include <limits>

#include "gem_internal.h"
#include "conv_rows.h"

namespace gem {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct TrainConv {
    int K = 0, N = 0;              // padded input / output channels
    bool bn = false;
    size_t ow = 0, ob = 0, og = 0, obe = 0;     // arena offsets: weight [3][N][K], bias [N], gamma [N], beta [N]
    size_t os = 0;                 // statistics arena: running_mean [N], running_var [N]
    float *Y = nullptr, *out = nullptr, *mean = nullptr, *invstd = nullptr;
    float *adj = nullptr, *slab = nullptr;      // adjoint image [3][K][N]; weight-gradient slabs [nslab][3][N][K]
    float* dY = nullptr;           // BatchNorm layers: gradient w.r.t. the conv's own output [rows][N], kept until the step's ONE
                                   // weight-gradient launch (gemm_tn3_all_kernel) has read it
};
struct TrainLinear {
    int K = 0, N = 0;
    size_t ow = 0, ob = 0;
    float *adj = nullptr, *slab = nullptr;
};
struct AdjDesc { const float* W; float* out; int taps, N, K, tiles; };
struct SumDesc { const float* slab; float* out; unsigned long long n; };
constexpr int TN_ROWS_CONV = 64, TN_ROWS_LINEAR = 256;     // rows per weight-gradient slab (conv: at least; see conv_slab_rows)
constexpr int TN_CONV_SLABS_MAX = 16;
// rows per conv weight-gradient slab for a step of `rows` rows: 64 up to 1024 rows (the reference's batch: ten slabs fill the chip),
// beyond that at most TN_CONV_SLABS_MAX slabs -- at 10 240 rows the sum over 160 slabs of 64 rows read 655 MB per step (round 4)
static inline int conv_slab_rows(int rows) {
    const int per = (rows + TN_CONV_SLABS_MAX - 1) / TN_CONV_SLABS_MAX;
    return std::max(TN_ROWS_CONV, (per + 31) / 32 * 32);
}
constexpr int LOSS_BLOCK = 1024;

}  // namespace gem

struct gem_trainer {
    gem_config cfg;
    gem_handle* h = nullptr;       // workspace / dispatch state of the GEMM launchers
    int T = 0, C = 0, Cp = 0, D = 0, Dp = 0, top = 0, topp = 0, Bmax = 0;
    std::vector<gem::TrainConv> enc, dec;
    gem::TrainLinear fc, dec_in;
    size_t n_params = 0, n_stats = 0;
    float *P = nullptr, *G = nullptr, *M1 = nullptr, *M2 = nullptr, *S = nullptr;
    float *pose_p = nullptr, *mulv = nullptr, *z = nullptr, *h0 = nullptr, *Xp = nullptr;
    float *gA = nullptr, *gB = nullptr, *dmulv = nullptr, *dz = nullptr;
    float *dYT = nullptr, *lin_slab = nullptr; int lin_slab_cap = 8;      // linear_bwd_data: transposed gradient [N][pad64(B)], K-slabs of dX
    float* dx_slab = nullptr; int dx_slab_cap = 64;                       // linear_fused_backward: one [64][K] slab of dX per strip of n tiles
    double* bn_part = nullptr; int bn_nrb_cap = 0;      // large-batch BatchNorm (bnl_*_kernel): [nrb][N][3] partial sums
    double* red = nullptr;         // [8 + partial sums]: [4..6] loss, recon, kld; [8..) per-block partials of the two loss kernels
    gem::AdjDesc* adj_tab = nullptr; int n_adj = 0, adj_tiles = 0;
    gem::SumDesc* sum_tab = nullptr; int n_sum = 0; size_t sum_max = 0;
    void* tn_tab = nullptr; int n_tn = 0, tn_tiles = 0;      // the weight-gradient launch's layer table (TnTable, host copy: passed by value)
    int part_recon = 0, part_latent = 0;       // capacity of the partial-sum regions
    long step = 0;
    bool grads_partial = false;    // the last gem_trainer_step ran with update = 2: the linear layers' ranges of the gradient arena are stale
    std::vector<void*> allocs;
};

namespace gem {

// ---- BatchNorm1d (training mode) + LeakyReLU over [rows, N]: one workgroup per 16 channels, BN_GROUPS row groups -------------
// A thread owns channel c = lane & 15 of the rows g, g + 64, ...; BN_REGS of them live in registers, so that the statistics
// and the normalisation need ONE trip to memory (rows <= 2560); longer inputs: bnl_*_kernel below.
constexpr int BN_GROUPS = 64, BN_THREADS = 16 * BN_GROUPS, BN_REGS = 16;
// column sums of NV per-thread values over the row groups of a workgroup: lanes of a wave that share a channel first (4 row
// groups per wave), then the 16 waves through LDS; every thread returns the totals of its channel
template <int NV>
__device__ __forceinline__ void bn_reduce(double (&v)[NV], double (*sh)[16][17]) {
    const int ch = threadIdx.x & 15, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        v[i] += __shfl_xor(v[i], 16);
        v[i] += __shfl_xor(v[i], 32);
        if ((threadIdx.x & 48) == 0) sh[i][wave][ch] = v[i];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        double s = 0.0;
#pragma unroll
        for (int j = 0; j < 16; ++j) s += sh[i][j][ch];
        v[i] = s;
    }
    __syncthreads();
}
__global__ __launch_bounds__(BN_THREADS) void bn_train_fwd_kernel(const float* __restrict__ Y, int rows, int N, const float* __restrict__ gamma,
                                                                  const float* __restrict__ beta, float* __restrict__ rmean, float* __restrict__ rvar,
                                                                  float* __restrict__ mean_out, float* __restrict__ invstd_out, float* __restrict__ out,
                                                                  float momentum, float eps) {
    __shared__ double sh[2][16][17];
    const int c = blockIdx.x * 16 + (threadIdx.x & 15), g = threadIdx.x >> 4;
    float y[BN_REGS];
    double v[2] = {0.0, 0.0};
#pragma unroll
    for (int j = 0; j < BN_REGS; ++j) { const int r = g + j * BN_GROUPS; y[j] = r < rows ? Y[(size_t)r * N + c] : 0.f; }
#pragma unroll
    for (int j = 0; j < BN_REGS; ++j) { v[0] += (double)y[j]; v[1] += (double)y[j] * y[j]; }
    bn_reduce<2>(v, sh);
    const double mean = v[0] / rows;
    double var = v[1] / rows - mean * mean;                 // biased (what normalises the batch)
    if (var < 0.0) var = 0.0;
    const float invstd = (float)(1.0 / sqrt(var + (double)eps)), mf = (float)mean;
    const float ga = gamma[c], be = beta[c];
#pragma unroll
    for (int j = 0; j < BN_REGS; ++j) {
        const int r = g + j * BN_GROUPS;
        const float o = ga * ((y[j] - mf) * invstd) + be;
        if (r < rows) out[(size_t)r * N + c] = o > 0.f ? o : o * LEAKY_SLOPE;
    }
    if (g == 0) {
        mean_out[c] = mf; invstd_out[c] = invstd;
        const double unbiased = rows > 1 ? var * rows / (rows - 1) : var;
        rmean[c] = (1.f - momentum) * rmean[c] + momentum * mf;
        rvar[c] = (1.f - momentum) * rvar[c] + momentum * (float)unbiased;
    }
}

// dOut (w.r.t. the block's output) -> dY (w.r.t. the conv output), dgamma, dbeta, and the conv's bias gradient sum_r dY (zero up
// to rounding -- the batch mean is subtracted).  LeakyReLU' from the sign of the output.
__global__ __launch_bounds__(BN_THREADS) void bn_train_bwd_kernel(const float* __restrict__ dOut, const float* __restrict__ out,
                                                                  const float* __restrict__ Y, int rows, int N, const float* __restrict__ gamma,
                                                                  const float* __restrict__ mean, const float* __restrict__ invstd,
                                                                  float* __restrict__ dY, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                                  float* __restrict__ dbias) {
    __shared__ double sh[3][16][17];
    const int c = blockIdx.x * 16 + (threadIdx.x & 15), g = threadIdx.x >> 4;
    const float mf = mean[c], is = invstd[c], ga = gamma[c];
    float dzv[BN_REGS], xh[BN_REGS];
    double v[3] = {0.0, 0.0, 0.0};
#pragma unroll
    for (int j = 0; j < BN_REGS; ++j) {
        const int r = g + j * BN_GROUPS;
        const size_t i = (size_t)r * N + c;
        const bool ok = r < rows;
        const float d = ok ? dOut[i] : 0.f, o = ok ? out[i] : 0.f, yy = ok ? Y[i] : mf;
        dzv[j] = d * (o > 0.f ? 1.f : LEAKY_SLOPE);
        xh[j] = (yy - mf) * is;
    }
#pragma unroll
    for (int j = 0; j < BN_REGS; ++j) { v[0] += dzv[j]; v[1] += (double)dzv[j] * xh[j]; v[2] += xh[j]; }
    bn_reduce<3>(v, sh);
    const float mb = (float)(v[0] / rows), mg = (float)(v[1] / rows);
#pragma unroll
    for (int j = 0; j < BN_REGS; ++j) {
        const int r = g + j * BN_GROUPS;
        if (r < rows) dY[(size_t)r * N + c] = ga * is * (dzv[j] - mb - xh[j] * mg);
    }
    // the conv bias gradient sum_r dY = gamma invstd (sum dz - rows mean_dz - mean_dzx sum xhat): zero up to rounding, from the sums
    // (round 4: no second reduction; what torch holds there is rounding noise of the same size)
    if (g == 0) {
        dgamma[c] = (float)v[1]; dbeta[c] = (float)v[0];
        dbias[c] = (float)((double)ga * is * (v[0] - (double)rows * mb - (double)mg * v[2]));
    }
}

// ---- BatchNorm whose statistics come out of the producing conv's epilogue (conv_rows.h: CrStats; rows <= BNF_ROWS_MAX) ------------
// The register-resident kernels above put a whole column block (all rows x 16 channels) through ONE CU: 2560 half-used lines through
// one L1 = ~4.3 us plus the reduction, 6 / 10 us per launch.  With the per-tile sums already there (fp64, one triple per 32-row tile
// and channel, added here in tile order: the same bits in every workgroup) the rest is elementwise: 64 channels x 64 rows per
// workgroup, 10-80 workgroups at the reference's batch.  Same formulas as bn_train_fwd_kernel / bn_train_bwd_kernel.
constexpr int BNF_ROWS_MAX = 2560, BNF_TILES_PER_PART = BNF_ROWS_MAX / 32 / 4;
// totals over the row tiles for the workgroup's 64 channels: four thread groups take a quarter of the tiles each (all of a group's
// values requested at once: as a loop of dependent round trips the same sum cost 10+ us), added in tile order, the quarters in order
template <int NV>
__device__ __forceinline__ void bnf_totals(const double* __restrict__ part, int nrt, int N, int c0, double (*tot)[64]) {
    __shared__ double sh[NV][4][64];
    const int ch = threadIdx.x & 63, pt = threadIdx.x >> 6;
    const int per = (nrt + 3) / 4, t0 = pt * per, t1 = min(nrt, t0 + per);
    double v[NV][BNF_TILES_PER_PART];          // (one round trip for everything this thread adds)
#pragma unroll
    for (int j = 0; j < BNF_TILES_PER_PART; ++j)
#pragma unroll
        for (int q = 0; q < NV; ++q) v[q][j] = t0 + j < t1 ? part[((size_t)(t0 + j) * N + c0 + ch) * 3 + q] : 0.0;
#pragma unroll
    for (int q = 0; q < NV; ++q) {
        double sum = 0.0;
#pragma unroll
        for (int j = 0; j < BNF_TILES_PER_PART; ++j) sum += v[q][j];
        sh[q][pt][ch] = sum;
    }
    __syncthreads();
    if (threadIdx.x < 64 * NV) {
        const int q = threadIdx.x >> 6;
        tot[q][ch] = ((sh[q][0][ch] + sh[q][1][ch]) + sh[q][2][ch]) + sh[q][3][ch];
    }
    __syncthreads();
}
__global__ __launch_bounds__(256) void bnf_fwd_apply_kernel(const float* __restrict__ Y, int rows, int N, const double* __restrict__ part, int nrt,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ rmean,
                                                            float* __restrict__ rvar, float* __restrict__ mean_out, float* __restrict__ invstd_out,
                                                            float* __restrict__ out, float momentum, float eps) {
    __shared__ double tot[2][64];
    __shared__ float mfs[64], iss[64];
    const int c0 = blockIdx.x * 64, tid = threadIdx.x;
    bnf_totals<2>(part, nrt, N, c0, tot);
    if (tid < 64) {
        const double mean = tot[0][tid] / rows;
        double var = tot[1][tid] / rows - mean * mean;                 // biased (what normalises the batch)
        if (var < 0.0) var = 0.0;
        const float invstd = (float)(1.0 / sqrt(var + (double)eps)), mf = (float)mean;
        mfs[tid] = mf; iss[tid] = invstd;
        if (blockIdx.y == 0) {
            const int c = c0 + tid;
            mean_out[c] = mf; invstd_out[c] = invstd;
            const double unbiased = rows > 1 ? var * rows / (rows - 1) : var;
            rmean[c] = (1.f - momentum) * rmean[c] + momentum * mf;
            rvar[c] = (1.f - momentum) * rvar[c] + momentum * (float)unbiased;
        }
    }
    __syncthreads();
    const int cl = (tid & 15) * 4, c = c0 + cl;
    const f32x4 ga = *reinterpret_cast<const f32x4*>(gamma + c), be = *reinterpret_cast<const f32x4*>(beta + c);
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int r = blockIdx.y * 64 + (tid >> 4) + 16 * p;
        if (r >= rows) continue;
        const f32x4 y = *reinterpret_cast<const f32x4*>(Y + (size_t)r * N + c);
        f32x4 o;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float v = ga[q] * ((y[q] - mfs[cl + q]) * iss[cl + q]) + be[q];
            o[q] = v > 0.f ? v : v * LEAKY_SLOPE;
        }
        *reinterpret_cast<f32x4*>(out + (size_t)r * N + c) = o;
    }
}
// dz (the producing conv stored dOut * LeakyReLU'(out)) -> dY, dgamma, dbeta, and the conv's bias gradient from the sums
__global__ __launch_bounds__(256) void bnf_bwd_apply_kernel(const float* __restrict__ dz, const float* __restrict__ Y, int rows, int N,
                                                            const double* __restrict__ part, int nrt, const float* __restrict__ gamma,
                                                            const float* __restrict__ mean, const float* __restrict__ invstd, float* __restrict__ dY,
                                                            float* __restrict__ dgamma, float* __restrict__ dbeta, float* __restrict__ dbias) {
    __shared__ double tot[3][64];
    const int c0 = blockIdx.x * 64, tid = threadIdx.x;
    bnf_totals<3>(part, nrt, N, c0, tot);
    const int cl = (tid & 15) * 4, c = c0 + cl;
    const f32x4 ga = *reinterpret_cast<const f32x4*>(gamma + c), mf = *reinterpret_cast<const f32x4*>(mean + c), is = *reinterpret_cast<const f32x4*>(invstd + c);
    float mb[4], mg[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) { mb[q] = (float)(tot[0][cl + q] / rows); mg[q] = (float)(tot[1][cl + q] / rows); }
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int r = blockIdx.y * 64 + (tid >> 4) + 16 * p;
        if (r >= rows) continue;
        const f32x4 d = *reinterpret_cast<const f32x4*>(dz + (size_t)r * N + c), y = *reinterpret_cast<const f32x4*>(Y + (size_t)r * N + c);
        f32x4 o;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float xh = (y[q] - mf[q]) * is[q];
            o[q] = ga[q] * is[q] * (d[q] - mb[q] - xh * mg[q]);
        }
        *reinterpret_cast<f32x4*>(dY + (size_t)r * N + c) = o;
    }
    if (blockIdx.y == 0 && tid < 64) {
        const int ch = c0 + tid;
        const double v0 = tot[0][tid], v1 = tot[1][tid], v2 = tot[2][tid];
        const float mbs = (float)(v0 / rows), mgs = (float)(v1 / rows);
        dgamma[ch] = (float)v1; dbeta[ch] = (float)v0;
        dbias[ch] = (float)((double)gamma[ch] * invstd[ch] * (v0 - (double)rows * mbs - (double)mgs * v2));
    }
}

// ---- BatchNorm for MORE rows than a workgroup holds in registers (rows > BN_REGS * BN_GROUPS = 1024: batches above 102 windows) ----
// The one-workgroup-per-16-channels kernels above are built for the reference's batch of 64 (640 rows): at 10 240 rows (batch 1024)
// their 4-32 workgroups read the layer two or three times at a few percent of the chip's bandwidth -- 2.1 of that step's 5.2 ms
// (round 4).  Here a workgroup owns 64 channels x BNL_ROWS rows (256-byte row segments per wavefront): a first launch leaves
// per-row-block partial sums (fp64), the second launch's workgroups each add them up for their 64 channels in row-block order (the
// same bits in every workgroup) and normalise / form dY.  No atomics and no device-scope fences inside a kernel (a first version
// that let the last workgroup of a channel group finish the statistics behind `__threadfence()` took 28-32 us per launch against 8
// for the same pass without it: with eight non-coherent L2s a device-scope release is an L2 write-back).  Same formulas as
// bn_train_fwd_kernel / bn_train_bwd_kernel; the conv bias gradient sum_r dY -- zero up to rounding -- is formed from the sums
// (gamma invstd (sum dz - rows mean_dz - mean_dzx sum xhat)) instead of by a third pass.
constexpr int BNL_ROWS = 128;
// sums of NV per-thread values over the 4 row groups of a workgroup: every thread gets the totals of its channel
template <int NV>
__device__ __forceinline__ void bnl_block_sum(double (&v)[NV], double (*sh)[4][64]) {
    const int l = threadIdx.x & 63, rg = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < NV; ++i) sh[i][rg][l] = v[i];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = ((sh[i][0][l] + sh[i][1][l]) + sh[i][2][l]) + sh[i][3][l];
    __syncthreads();
}
// totals over all row blocks of the partials [nrb][N][NV] for this thread's channel: a quarter of the row blocks per row group (all of
// a thread's loads in flight together), the quarters added in order
template <int NV>
__device__ __forceinline__ void bnl_totals(const double* __restrict__ part, int nrb, int N, int c, double (&v)[NV], double (*sh)[4][64]) {
    const int rg = threadIdx.x >> 6;
    const int per = (nrb + 3) / 4, rb0 = rg * per, rb1 = min(nrb, rb0 + per);
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = 0.0;
#pragma unroll 8
    for (int rb = rb0; rb < rb1; ++rb) {
#pragma unroll
        for (int i = 0; i < NV; ++i) v[i] += part[((size_t)rb * N + c) * NV + i];
    }
    bnl_block_sum<NV>(v, sh);
}
__global__ __launch_bounds__(256) void bnl_fwd_stats_kernel(const float* __restrict__ Y, int rows, int N, double* __restrict__ part) {
    __shared__ double sh[2][4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), rg = threadIdx.x >> 6, r0 = blockIdx.y * BNL_ROWS;
    double v[2] = {0.0, 0.0};
#pragma unroll 8
    for (int i = rg; i < BNL_ROWS; i += 4) {
        const int r = r0 + i;
        if (r < rows) { const double q = Y[(size_t)r * N + c]; v[0] += q; v[1] += q * q; }
    }
    bnl_block_sum<2>(v, sh);
    if (rg == 0) { part[((size_t)blockIdx.y * N + c) * 2] = v[0]; part[((size_t)blockIdx.y * N + c) * 2 + 1] = v[1]; }
}
__global__ __launch_bounds__(256) void bnl_fwd_apply_kernel(const float* __restrict__ Y, int rows, int N, const double* __restrict__ part, int nrb,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ rmean,
                                                            float* __restrict__ rvar, float* __restrict__ mean_out, float* __restrict__ invstd_out,
                                                            float* __restrict__ out, float momentum, float eps) {
    __shared__ double sh[2][4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), rg = threadIdx.x >> 6, r0 = blockIdx.y * BNL_ROWS;
    double v[2];
    bnl_totals<2>(part, nrb, N, c, v, sh);
    const double mean = v[0] / rows;
    double var = v[1] / rows - mean * mean;                 // biased (what normalises the batch)
    if (var < 0.0) var = 0.0;
    const float is = (float)(1.0 / sqrt(var + (double)eps)), mf = (float)mean;
    if (blockIdx.y == 0 && rg == 0) {
        mean_out[c] = mf; invstd_out[c] = is;
        const double unbiased = rows > 1 ? var * rows / (rows - 1) : var;
        rmean[c] = (1.f - momentum) * rmean[c] + momentum * mf;
        rvar[c] = (1.f - momentum) * rvar[c] + momentum * (float)unbiased;
    }
    const float ga = gamma[c], be = beta[c];
#pragma unroll 8
    for (int i = rg; i < BNL_ROWS; i += 4) {
        const int r = r0 + i;
        if (r < rows) {
            const float o = ga * ((Y[(size_t)r * N + c] - mf) * is) + be;
            out[(size_t)r * N + c] = o > 0.f ? o : o * LEAKY_SLOPE;
        }
    }
}
// backward, first launch: partial sums of dz, dz xhat and xhat
__global__ __launch_bounds__(256) void bnl_bwd_stats_kernel(const float* __restrict__ dOut, const float* __restrict__ out, const float* __restrict__ Y,
                                                            int rows, int N, const float* __restrict__ mean, const float* __restrict__ invstd,
                                                            double* __restrict__ part) {
    __shared__ double sh[3][4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), rg = threadIdx.x >> 6, r0 = blockIdx.y * BNL_ROWS;
    const float mf = mean[c], is = invstd[c];
    double v[3] = {0.0, 0.0, 0.0};
#pragma unroll 8
    for (int i = rg; i < BNL_ROWS; i += 4) {
        const int r = r0 + i;
        if (r < rows) {
            const size_t k = (size_t)r * N + c;
            const float d = dOut[k] * (out[k] > 0.f ? 1.f : LEAKY_SLOPE), xh = (Y[k] - mf) * is;
            v[0] += d; v[1] += (double)d * xh; v[2] += xh;
        }
    }
    bnl_block_sum<3>(v, sh);
    if (rg == 0) {
#pragma unroll
        for (int i = 0; i < 3; ++i) part[((size_t)blockIdx.y * N + c) * 3 + i] = v[i];
    }
}
// backward, second launch: dgamma, dbeta, the conv bias gradient (row block 0) and dY
__global__ __launch_bounds__(256) void bnl_bwd_apply_kernel(const float* __restrict__ dOut, const float* __restrict__ out, const float* __restrict__ Y,
                                                            int rows, int N, const double* __restrict__ part, int nrb, const float* __restrict__ gamma,
                                                            const float* __restrict__ mean, const float* __restrict__ invstd, float* __restrict__ dY,
                                                            float* __restrict__ dgamma, float* __restrict__ dbeta, float* __restrict__ dbias) {
    __shared__ double sh[3][4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), rg = threadIdx.x >> 6, r0 = blockIdx.y * BNL_ROWS;
    double v[3];
    bnl_totals<3>(part, nrb, N, c, v, sh);
    const float mf = mean[c], is = invstd[c], ga = gamma[c];
    const float mb = (float)(v[0] / rows), mg = (float)(v[1] / rows);
    if (blockIdx.y == 0 && rg == 0) {
        dgamma[c] = (float)v[1]; dbeta[c] = (float)v[0];
        dbias[c] = (float)((double)ga * is * (v[0] - (double)rows * mb - (double)mg * v[2]));
    }
#pragma unroll 8
    for (int i = rg; i < BNL_ROWS; i += 4) {
        const int r = r0 + i;
        if (r < rows) {
            const size_t k = (size_t)r * N + c;
            const float dz0 = dOut[k] * (out[k] > 0.f ? 1.f : LEAKY_SLOPE);
            dY[k] = ga * is * (dz0 - mb - ((Y[k] - mf) * is) * mg);
        }
    }
}

__global__ __launch_bounds__(BN_THREADS) void colsum_kernel(const float* __restrict__ dC, int rows, int N, float* __restrict__ out) {
    __shared__ double sh[1][16][17];
    const int c = blockIdx.x * 16 + (threadIdx.x & 15), g = threadIdx.x >> 4;
    double v[1] = {0.0};
    for (int r = g; r < rows; r += BN_GROUPS) v[0] += dC[(size_t)r * N + c];
    bn_reduce<1>(v, sh);
    if (g == 0) out[c] = (float)v[0];
}

constexpr int CONV_ROWS_MAX = 1 << 30;          // every batch: 1.18 -> 1.11 ms per step at batch 128, 2.17 -> 2.11 at 512, 3.48 -> 3.44 at 1024 (dev switch: GEM_CONV_ROWS_MAX)
// 0 = launched; 1 = error; -1 = not applicable (the caller falls back to launch_gemm)
static int conv_rows(gem_trainer* t, const float* W, const float* bias, const float* A, int lda, float* C, int ldc, int rows, int N, int K, hipStream_t s,
                     const CrStats& st = CrStats{});

// ---- weight gradient: dW[tap][n][k] = sum_r dC[r][n] * A[r + tap - 1][k], slab z = rows [z * rps, (z + 1) * rps) --------------
template <int TAPS>
__device__ __forceinline__ void gemm_tn_tile(const float* __restrict__ dC, int ldc, const float* __restrict__ A, int lda, float* __restrict__ slab, int rows, int N,
                                             int K, int T, int rps, int tile, int tap, int z) {
    __shared__ __attribute__((aligned(16))) float Cs[32][68];
    __shared__ __attribute__((aligned(16))) float As[32][68];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nt = tile / (K / 64), kt = tile - nt * (K / 64);
    const int n0 = nt * 64, k0 = kt * 64;
    const int wm = wave >> 1, wn = wave & 1;
    const int fi = lane & 15, fq = lane >> 4;
    f32x4 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int r_begin = z * rps, r_end = min(rows, r_begin + rps);
    f32x4 vc[2], va[2];
    auto fetch = [&](int r0) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int i = tid + u * 256, rl = i >> 4, c4 = (i & 15) * 4;
            const int row = r0 + rl;
            vc[u] = f32x4{0.f, 0.f, 0.f, 0.f}; va[u] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (row < r_end) {
                vc[u] = *reinterpret_cast<const f32x4*>(dC + (size_t)row * ldc + n0 + c4);
                int src = row;
                bool ok = true;
                if (TAPS == 3) { const int tt = row % T + tap - 1; ok = tt >= 0 && tt < T; src = row + tap - 1; }
                if (ok) va[u] = *reinterpret_cast<const f32x4*>(A + (size_t)src * lda + k0 + c4);
            }
        }
    };
    fetch(r_begin);
    for (int r0 = r_begin; r0 < r_end; r0 += 32) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int i = tid + u * 256, rl = i >> 4, c4 = (i & 15) * 4;
            *reinterpret_cast<f32x4*>(&Cs[rl][c4]) = vc[u];
            *reinterpret_cast<f32x4*>(&As[rl][c4]) = va[u];
        }
        __syncthreads();
        if (r0 + 32 < r_end) fetch(r0 + 32);          // the next 32 rows travel while this block's products run
#pragma unroll
        for (int kk = 0; kk < 32; kk += 4) {
            float a[2], b[2];
#pragma unroll
            for (int q = 0; q < 2; ++q) { a[q] = Cs[kk + fq][wm * 32 + q * 16 + fi]; b[q] = As[kk + fq][wn * 32 + q * 16 + fi]; }
#pragma unroll
            for (int x = 0; x < 2; ++x)
#pragma unroll
                for (int y = 0; y < 2; ++y) acc[x][y] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[x], b[y], acc[x][y], 0, 0, 0);
        }
        __syncthreads();
    }
    float* out = slab + ((size_t)z * TAPS + tap) * N * K;
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int n = n0 + wm * 32 + x * 16 + 4 * fq + e, k = k0 + wn * 32 + y * 16 + fi;
                out[(size_t)n * K + k] = acc[x][y][e];
            }
}
template <int TAPS>
__global__ __launch_bounds__(256) void gemm_tn_kernel(const float* __restrict__ dC, int ldc, const float* __restrict__ A, int lda,
                                                      float* __restrict__ slab, int rows, int N, int K, int T, int rps) {
    gemm_tn_tile<TAPS>(dC, ldc, A, lda, slab, rows, N, K, T, rps, blockIdx.x, blockIdx.y, blockIdx.z);
}
// Two small jobs that nothing but the end of the step waits for ride along with the weight-gradient launch as extra workgroups
// (blockIdx.x past the tiles; tap 0 / slab 0 only): the loss scalars from the per-block partials (block order: out = [loss, recon,
// kld]) and the bias gradient of the last decoder conv (no BatchNorm behind it: the column sums of the loss gradient).
struct StepTail {
    const double* part_recon; const double* part_latent; double* out; double* out2;
    double n_recon, kld_weight;
    int n_recon_parts, n_latent_parts, B;
    const float* cs_src; float* cs_out; int cs_N;          // column sums over `rows` rows of cs_src [rows][cs_N]
};
__device__ __forceinline__ void finish_loss(const StepTail& t) {
    __shared__ double sh[2][256];
    double a = 0.0, b = 0.0;
    if (threadIdx.x < 256) {          // (the first 256 threads of the workgroup; the others only join the barrier)
        for (int i = threadIdx.x; i < t.n_recon_parts; i += 256) a += t.part_recon[i];
        for (int i = threadIdx.x; i < t.n_latent_parts; i += 256) b += t.part_latent[i];
        sh[0][threadIdx.x] = a; sh[1][threadIdx.x] = b;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        a = 0.0; b = 0.0;
        for (int i = 0; i < 256; ++i) { a += sh[0][i]; b += sh[1][i]; }
        const double recon = a / t.n_recon, kld = -0.5 * b / t.B;
        t.out[0] = recon + t.kld_weight * kld; t.out[1] = recon; t.out[2] = kld;
        if (t.out2) { t.out2[0] = t.out[0]; t.out2[1] = recon; t.out2[2] = kld; }          // (the caller's copy: no copy launch behind the step)
    }
}
__device__ __forceinline__ void colsum16(const StepTail& t, int group, int rows) {          // 16 channels, 16 row groups, fp64 sums
    __shared__ double sh[16][17];
    const int c = group * 16 + (threadIdx.x & 15), g = threadIdx.x >> 4;
    double v = 0.0;
    if (g < 16) {
        // eight rows requested at a time (as `v += load` this loop is a chain of rows / 16 dependent round trips: 40 at the
        // reference's batch = 30 us, which WAS the duration of the whole weight-gradient launch)
        for (int r0 = g; r0 < rows; r0 += 128) {
            float x[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) x[j] = t.cs_src[(size_t)min(r0 + 16 * j, rows - 1) * t.cs_N + c];
#pragma unroll
            for (int j = 0; j < 8; ++j) if (r0 + 16 * j < rows) v += x[j];
        }
        sh[g][threadIdx.x & 15] = v;
    }
    __syncthreads();
    if (g == 0) {
        double sum = 0.0;
#pragma unroll
        for (int j = 0; j < 16; ++j) sum += sh[j][threadIdx.x & 15];
        t.cs_out[c] = (float)sum;
    }
}
// every conv layer's weight gradient in ONE launch (they wait for nothing but their layer's dY, and nothing but Adam waits for
// them: eleven launches of ~6 us at the reference's batch otherwise): blockIdx.x walks the layers' tiles (table), y = tap, z = slab
struct TnDesc { const float* dC; const float* A; float* slab; float* g; int N, K, tile0; };
// (the table travels BY VALUE in the kernel arguments: found through global memory, the walk to a workgroup's layer was a chain of up
// to eleven dependent loads, ~10 us in front of the last layers' workgroups)
constexpr int TN_MAX_LAYERS = 16;
struct TnTable { TnDesc d[TN_MAX_LAYERS]; int n; };
__global__ __launch_bounds__(256) void gemm_tn3_all_kernel(const TnTable tabv, int n_tiles, int rows, int T, int rps, int nslab, const StepTail tail) {
    const TnDesc* tab = tabv.d; const int n_layers = tabv.n;
    if ((int)blockIdx.x >= n_tiles) {
        if (blockIdx.y == 0 && blockIdx.z == 0) {
            if ((int)blockIdx.x == n_tiles) finish_loss(tail);
            else colsum16(tail, (int)blockIdx.x - n_tiles - 1, rows);
        }
        return;
    }
    int l = 0;
    while (l + 1 < n_layers && (int)blockIdx.x >= tab[l + 1].tile0) ++l;
    const TnDesc d = tab[l];
    gemm_tn_tile<3>(d.dC, d.N, d.A, d.K, nslab > 1 ? d.slab : d.g, rows, d.N, d.K, T, rps, (int)blockIdx.x - d.tile0, blockIdx.y, blockIdx.z);
}

// ---- a LINEAR layer's weight gradient and its Adam step in one kernel (gem_trainer_step, update = 2) ---------------------------
// The two linear layers hold 97 % of the parameters and their gradient has rank <= batch: dW[n][k] = sum_b dC[b][n] A[b][k] is
// a 64 x 64 x batch product per tile -- nothing next to the 3 x 16 KB of parameter and moment traffic of the tile.  Forming it
// where Adam consumes it removes the gradient's round trip through HBM (2 x 4 bytes of the step's 32 bytes per parameter) and the
// two gemm_tn launches.  Same products in the same order as gemm_tn_kernel<1> with one slab, same arithmetic as adam_kernel: at
// batches up to TN_ROWS_LINEAR the parameters after the step are bitwise those of the two-kernel path.
struct AdamScalars { float lr_bc1, b1, b2, eps, wd, bc2_sqrt; };
__global__ __launch_bounds__(256) void gemm_tn_adam_kernel(const float* __restrict__ dC, int ldc, const float* __restrict__ A, int lda, int rows,
                                                           int N, int K, float* __restrict__ P, float* __restrict__ M1, float* __restrict__ M2,
                                                           AdamScalars ad) {
    __shared__ __attribute__((aligned(16))) float smem[64 * 68];          // operand tiles [32][68] x 2, then the gradient tile [64][68]
    float (*Cs)[68] = reinterpret_cast<float (*)[68]>(smem);
    float (*As)[68] = reinterpret_cast<float (*)[68]>(smem + 32 * 68);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nt = blockIdx.x / (K / 64), kt = blockIdx.x - nt * (K / 64);
    const int n0 = nt * 64, k0 = kt * 64;
    const int wm = wave >> 1, wn = wave & 1;
    const int fi = lane & 15, fq = lane >> 4;
    // the tile's parameters and moments travel while the products run
    f32x4 p4[4], m4[4], v4[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const size_t i = (size_t)(n0 + q * 16 + (tid >> 4)) * K + k0 + (tid & 15) * 4;
        p4[q] = *reinterpret_cast<const f32x4*>(P + i);
        m4[q] = *reinterpret_cast<const f32x4*>(M1 + i);
        v4[q] = *reinterpret_cast<const f32x4*>(M2 + i);
    }
    f32x4 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 vc[2], va[2];
    auto fetch = [&](int r0) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int i = tid + u * 256, rl = i >> 4, c4 = (i & 15) * 4;
            const int row = r0 + rl;
            vc[u] = f32x4{0.f, 0.f, 0.f, 0.f}; va[u] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (row < rows) {
                vc[u] = *reinterpret_cast<const f32x4*>(dC + (size_t)row * ldc + n0 + c4);
                va[u] = *reinterpret_cast<const f32x4*>(A + (size_t)row * lda + k0 + c4);
            }
        }
    };
    fetch(0);
    for (int r0 = 0; r0 < rows; r0 += 32) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int i = tid + u * 256, rl = i >> 4, c4 = (i & 15) * 4;
            *reinterpret_cast<f32x4*>(&Cs[rl][c4]) = vc[u];
            *reinterpret_cast<f32x4*>(&As[rl][c4]) = va[u];
        }
        __syncthreads();
        if (r0 + 32 < rows) fetch(r0 + 32);
#pragma unroll
        for (int kk = 0; kk < 32; kk += 4) {
            float a[2], b[2];
#pragma unroll
            for (int q = 0; q < 2; ++q) { a[q] = Cs[kk + fq][wm * 32 + q * 16 + fi]; b[q] = As[kk + fq][wn * 32 + q * 16 + fi]; }
#pragma unroll
            for (int x = 0; x < 2; ++x)
#pragma unroll
                for (int y = 0; y < 2; ++y) acc[x][y] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[x], b[y], acc[x][y], 0, 0, 0);
        }
        __syncthreads();
    }
    float (*Gt)[68] = reinterpret_cast<float (*)[68]>(smem);
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y)
#pragma unroll
            for (int e = 0; e < 4; ++e) Gt[wm * 32 + x * 16 + 4 * fq + e][wn * 32 + y * 16 + fi] = acc[x][y][e];
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int r = q * 16 + (tid >> 4), c4 = (tid & 15) * 4;
        const f32x4 g4 = *reinterpret_cast<const f32x4*>(&Gt[r][c4]);
        f32x4 po, mo, vo;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float gg = g4[e] + ad.wd * p4[q][e];
            const float mm = ad.b1 * m4[q][e] + (1.f - ad.b1) * gg;
            const float vv = ad.b2 * v4[q][e] + (1.f - ad.b2) * gg * gg;
            mo[e] = mm; vo[e] = vv;
            po[e] = p4[q][e] - ad.lr_bc1 * mm / (sqrtf(vv) / ad.bc2_sqrt + ad.eps);
        }
        const size_t i = (size_t)(n0 + r) * K + k0 + c4;
        *reinterpret_cast<f32x4*>(P + i) = po;
        *reinterpret_cast<f32x4*>(M1 + i) = mo;
        *reinterpret_cast<f32x4*>(M2 + i) = vo;
    }
}
// ---- a linear layer's WHOLE backward at batches of at most 64 windows: weight gradient + Adam step + backward-data product -----
// gemm_tn_adam_kernel reads every weight once (with its moments); the layer's backward-data product dX[b][k] = sum_n dY[b][n] W[n][k]
// read all of them once more (linear_bwd_data: 84 / 42 MB per layer and step) plus a transposed copy of dY.  Here a workgroup owns
// one 64-wide k tile and a STRIP of `tps` 64-wide n tiles; it walks the strip, and while a tile's parameters sit in registers for
// their Adam step they also pass through LDS as the B operand of the tile's share of dX, which accumulates in registers across the
// strip and leaves as one slab per strip (summed in strip order by slab_sum_kernel).  The next tile's parameters and dY tile (and the
// current tile's moments) are requested before the current tile's products, unconditionally (a clamped index on the last tile: loads behind a branch
// would turn every later wait into a full drain).  The weight-gradient products and the Adam arithmetic are gemm_tn_adam_kernel's,
// in the same order: parameters and moments after the step are bitwise the same.  dX is formed from the weights BEFORE their step.
// bias_grad: the column sums of dY (an fp64 sum of at most 64 fp32 values: exact), written by the workgroups of the first k tile.
// Eight waves: four form the weight-gradient tile (gemm_tn_adam_kernel's quadrants), four the backward-data share, side by side on
// the matrix pipes; all 512 threads then take 8 parameters each through the Adam step.
constexpr int DX_THREADS = 512;
__global__ __launch_bounds__(DX_THREADS, 4) void gemm_tn_adam_dx_kernel(const float* __restrict__ dC, const float* __restrict__ A, int rows, int N, int K,
                                                                        int tps, float* __restrict__ P, float* __restrict__ M1, float* __restrict__ M2,
                                                                        AdamScalars ad, float* __restrict__ dx_slab, float* __restrict__ bias_grad) {
    __shared__ __attribute__((aligned(16))) float smem[3 * 64 * 68];
    float (*As)[68] = reinterpret_cast<float (*)[68]>(smem);                   // the layer's input, rows b (zero beyond `rows`), this k tile
    float (*Cs)[68] = reinterpret_cast<float (*)[68]>(smem + 64 * 68);         // dY, rows b, the current n tile
    float (*Ws)[68] = reinterpret_cast<float (*)[68]>(smem + 2 * 64 * 68);     // the current weight tile [n][k]; then the gradient tile
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nkt = K / 64;
    const int strip = blockIdx.x / nkt, kt = blockIdx.x - strip * nkt;
    const int k0 = kt * 64, nt_begin = strip * tps, nt_last = nt_begin + tps - 1;
    const bool dx_wave = wave >= 4;
    const int wm = (wave & 3) >> 1, wn = wave & 1;
    const int fi = lane & 15, fq = lane >> 4;
    const int lr = tid >> 4, lc4 = (tid & 15) * 4;          // this thread's rows lr + 32 q, columns lc4 .. lc4 + 3 of a 64 x 64 tile
    auto load_p = [&](f32x4 (&r)[2], int nt) {
#pragma unroll
        for (int q = 0; q < 2; ++q) r[q] = *reinterpret_cast<const f32x4*>(P + (size_t)(nt * 64 + q * 32 + lr) * K + k0 + lc4);
    };
    auto load_c = [&](f32x4 (&vc)[2], int nt) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int b = min(q * 32 + lr, rows - 1);          // (rows beyond the batch: loaded, zeroed on the way into LDS)
            vc[q] = *reinterpret_cast<const f32x4*>(dC + (size_t)b * N + nt * 64 + lc4);
        }
    };
    f32x4 r0[2], r1[2], c0[2], c1[2];
    load_p(r0, nt_begin);
    load_c(c0, nt_begin);
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int b = q * 32 + lr;
        f32x4 a = f32x4{0.f, 0.f, 0.f, 0.f};
        if (b < rows) a = *reinterpret_cast<const f32x4*>(A + (size_t)b * K + k0 + lc4);
        *reinterpret_cast<f32x4*>(&As[b][lc4]) = a;
    }
    f32x4 accx[2][2];          // (the backward-data waves' accumulators, kept across the strip)
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) accx[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    auto tile = [&](int nt, const f32x4 (&cur)[2], const f32x4 (&ccur)[2], f32x4 (&nxt)[2], f32x4 (&cnxt)[2]) {
        __syncthreads();          // the previous tile's Adam phase has read the gradient tile (Ws); first tile: nothing pending
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int r = q * 32 + lr;
            *reinterpret_cast<f32x4*>(&Cs[r][lc4]) = r < rows ? ccur[q] : f32x4{0.f, 0.f, 0.f, 0.f};
            *reinterpret_cast<f32x4*>(&Ws[r][lc4]) = cur[q];
        }
        __syncthreads();
        // this tile's moments (wanted at the end of the tile), the next tile's parameters and dY
        f32x4 m4[2], v4[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const size_t i = (size_t)(nt * 64 + q * 32 + lr) * K + k0 + lc4;
            m4[q] = *reinterpret_cast<const f32x4*>(M1 + i);
            v4[q] = *reinterpret_cast<const f32x4*>(M2 + i);
        }
        const int nn = min(nt + 1, nt_last);
        load_p(nxt, nn);
        load_c(cnxt, nn);
        f32x4 acc[2][2];
        if (!dx_wave) {
            // weight gradient of the tile: dW[n][k] = sum_b dY[b][n] A[b][k]  (gemm_tn_kernel<1>'s products, b ascending)
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
            for (int kk = 0; kk < 64; kk += 4) {
                float a[2], b[2];
#pragma unroll
                for (int q = 0; q < 2; ++q) { a[q] = Cs[kk + fq][wm * 32 + q * 16 + fi]; b[q] = As[kk + fq][wn * 32 + q * 16 + fi]; }
#pragma unroll
                for (int x = 0; x < 2; ++x)
#pragma unroll
                    for (int y = 0; y < 2; ++y) acc[x][y] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[x], b[y], acc[x][y], 0, 0, 0);
            }
            if (bias_grad && kt == 0 && tid < 64) {
                double sum = 0.0;
#pragma unroll 8
                for (int b = 0; b < 64; ++b) sum += (double)Cs[b][tid];
                bias_grad[nt * 64 + tid] = (float)sum;
            }
        } else {
            // the tile's share of the backward-data product: dX[b][k] += sum_{n in tile} dY[b][n] W[n][k]
#pragma unroll 4
            for (int kk = 0; kk < 64; kk += 4) {
                float a[2], b[2];
#pragma unroll
                for (int q = 0; q < 2; ++q) { a[q] = Cs[wm * 32 + q * 16 + fi][kk + fq]; b[q] = Ws[kk + fq][wn * 32 + q * 16 + fi]; }
#pragma unroll
                for (int x = 0; x < 2; ++x)
#pragma unroll
                    for (int y = 0; y < 2; ++y) accx[x][y] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[x], b[y], accx[x][y], 0, 0, 0);
            }
        }
        __syncthreads();
        if (!dx_wave) {
#pragma unroll
            for (int x = 0; x < 2; ++x)
#pragma unroll
                for (int y = 0; y < 2; ++y)
#pragma unroll
                    for (int e = 0; e < 4; ++e) Ws[wm * 32 + x * 16 + 4 * fq + e][wn * 32 + y * 16 + fi] = acc[x][y][e];
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int r = q * 32 + lr;
            const f32x4 g4 = *reinterpret_cast<const f32x4*>(&Ws[r][lc4]);
            f32x4 po, mo, vo;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float gg = g4[e] + ad.wd * cur[q][e];
                const float mm = ad.b1 * m4[q][e] + (1.f - ad.b1) * gg;
                const float vv = ad.b2 * v4[q][e] + (1.f - ad.b2) * gg * gg;
                mo[e] = mm; vo[e] = vv;
                po[e] = cur[q][e] - ad.lr_bc1 * mm / (sqrtf(vv) / ad.bc2_sqrt + ad.eps);
            }
            const size_t i = (size_t)(nt * 64 + r) * K + k0 + lc4;
            *reinterpret_cast<f32x4*>(P + i) = po;
            *reinterpret_cast<f32x4*>(M1 + i) = mo;
            *reinterpret_cast<f32x4*>(M2 + i) = vo;
        }
    };
    for (int nt = nt_begin; nt < nt_begin + tps; nt += 2) {          // (tps is even: the two register sets swap roles)
        tile(nt, r0, c0, r1, c1);
        tile(nt + 1, r1, c1, r0, c0);
    }
    if (dx_wave) {
        float* out = dx_slab + (size_t)strip * 64 * K;
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
            for (int y = 0; y < 2; ++y)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int b = wm * 32 + x * 16 + 4 * fq + e, k = k0 + wn * 32 + y * 16 + fi;
                    out[(size_t)b * K + k] = accx[x][y][e];
                }
    }
}
// sum over slabs in slab order, eight requests in flight at a time (a plain `s += load` loop waits for every load in turn: ten
// dependent round trips for ten slabs)
__device__ __forceinline__ f32x4 sum_slabs(const float* __restrict__ p, int nslab, size_t stride) {
    f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int z0 = 0; z0 < nslab; z0 += 8) {
        f32x4 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = *reinterpret_cast<const f32x4*>(p + (size_t)min(z0 + j, nslab - 1) * stride);
#pragma unroll
        for (int j = 0; j < 8; ++j) if (z0 + j < nslab) s = (z0 + j == 0) ? v[j] : s + v[j];
    }
    return s;
}
// G = sum over slabs, in slab order; blockIdx.y = layer (table), blockIdx.x = 1024-element chunk
__global__ __launch_bounds__(256) void slab_sum_all_kernel(const SumDesc* __restrict__ tab, int nslab) {
    const SumDesc d = tab[blockIdx.y];
    const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= d.n) return;
    *reinterpret_cast<f32x4*>(d.out + i) = sum_slabs(d.slab + i, nslab, d.n);
}

// The fc layer's dX (= dOut of the encoder's last block, [rows][N] with N = the block's channels) summed from the strips' slabs of
// gemm_tn_adam_dx_kernel in 32 x 32 tiles, with that block's BatchNorm-backward sums formed on the way exactly as a backward-data
// conv's epilogue forms them (conv_rows.h, CrStats backward flavour: dz stored, sums of dz, dz xhat, xhat per tile and channel):
// slab_sum_kernel and the statistics half of bn_train_bwd_kernel in one launch.
__global__ __launch_bounds__(256) void dx_sum_stats_kernel(const float* __restrict__ slab, int nslab, size_t stride, float* __restrict__ dz_out, int rows, int N,
                                                           const CrStats st) {
    __shared__ __attribute__((aligned(16))) float sv[2][32][32];
    const int tid = threadIdx.x, m0 = blockIdx.x * 32, n0 = blockIdx.y * 32;
    const int i = tid >> 3, j4 = (tid & 7) * 4;
    const bool in = m0 + i < rows;
    const size_t at = (size_t)(in ? m0 + i : 0) * N + n0 + j4;
    const f32x4 sum = sum_slabs(slab + at, nslab, stride);
    const f32x4 o4 = *reinterpret_cast<const f32x4*>(st.out + at), y4 = *reinterpret_cast<const f32x4*>(st.Y + at);
    const f32x4 mf = *reinterpret_cast<const f32x4*>(st.mean + n0 + j4), is = *reinterpret_cast<const f32x4*>(st.invstd + n0 + j4);
    f32x4 dz, xh;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        dz[q] = in ? sum[q] * (o4[q] > 0.f ? 1.f : st.slope) : 0.f;
        xh[q] = in ? (y4[q] - mf[q]) * is[q] : 0.f;
    }
    if (in) *reinterpret_cast<f32x4*>(dz_out + at) = dz;
    *reinterpret_cast<f32x4*>(&sv[0][i][j4]) = dz;
    *reinterpret_cast<f32x4*>(&sv[1][i][j4]) = xh;
    __syncthreads();
    if (tid < 32) {
        double a = 0.0, b = 0.0, c = 0.0;
#pragma unroll 8
        for (int r = 0; r < 32; ++r) { const float d = sv[0][r][tid], x = sv[1][r][tid]; a += d; b += (double)d * x; c += x; }
        double* p = st.part + ((size_t)blockIdx.x * N + n0 + tid) * 3;
        p[0] = a; p[1] = b; p[2] = c;
    }
}

// dY [B][N] -> dY^T [N][Bp] (columns >= B zero): the row-major operand the rows-contracting kernel wants for the backward-data
// product of a linear layer, dX[b][k] = sum_n dY[b][n] W[n][k], taken straight from W's own [N][K] layout (linear_bwd_data)
// colsum != nullptr (one row tile only, Bp == 64): also the layer's bias gradient sum_b dY[b][n] (an fp64 sum of at most 64 fp32
// values: exact, so the same number colsum_kernel writes)
__global__ __launch_bounds__(256) void transpose_pad_kernel(const float* __restrict__ src, int B, int N, int Bp, float* __restrict__ dst,
                                                            float* __restrict__ colsum) {
    __shared__ float tile[64][65];
    const int n0 = blockIdx.x * 64, b0 = blockIdx.y * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int b = b0 + ty + 4 * j;
        tile[ty + 4 * j][tx] = b < B ? src[(size_t)b * N + n0 + tx] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 16; ++j) dst[(size_t)(n0 + ty + 4 * j) * Bp + b0 + tx] = tile[tx][ty + 4 * j];
    if (colsum && ty == 0) {
        double sum = 0.0;
#pragma unroll 8
        for (int b = 0; b < 64; ++b) sum += (double)tile[b][tx];
        colsum[n0 + tx] = (float)sum;
    }
}
// out[i] = sum over slabs (slab order), i < n_out <= n_slab_elems
__global__ __launch_bounds__(256) void slab_sum_kernel(const float* __restrict__ slab, int nslab, size_t n_slab_elems, float* __restrict__ out, size_t n_out) {
    const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= n_out) return;
    *reinterpret_cast<f32x4*>(out + i) = sum_slabs(slab + i, nslab, n_slab_elems);
}

// adjoint images for the backward-data products of every CONV layer in one launch (the two linear layers -- 97 % of the parameters
// -- need none: linear_bwd_data): conv [3][N][K] -> [3][K][N] with flipped taps,
// linear [N][K] -> [K][N]; blockIdx.y = layer (table), blockIdx.x = (tap, 64x64 tile), transposed through LDS
__global__ __launch_bounds__(256) void adjoint_all_kernel(const AdjDesc* __restrict__ tab) {
    __shared__ float tile[64][65];
    const AdjDesc d = tab[blockIdx.y];
    if ((int)blockIdx.x >= d.tiles) return;
    const int tk = d.K / 64, per = (d.N / 64) * tk;
    const int tap = blockIdx.x / per, r = blockIdx.x - tap * per;
    const int n0 = (r / tk) * 64, k0 = (r - (r / tk) * tk) * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const float* src = d.W + ((size_t)tap * d.N + n0) * d.K + k0;
#pragma unroll
    for (int j = 0; j < 16; ++j) tile[ty + 4 * j][tx] = src[(size_t)(ty + 4 * j) * d.K + tx];
    __syncthreads();
    float* dst = d.out + ((size_t)(d.taps - 1 - tap) * d.K + k0) * d.N + n0;
#pragma unroll
    for (int j = 0; j < 16; ++j) dst[(size_t)(ty + 4 * j) * d.N + tx] = tile[tx][ty + 4 * j];
}

__device__ __forceinline__ void block_partial(double s, double* out) {
    __shared__ double sh[LOSS_BLOCK / 64];
    s = wave_sum_dpp(s);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) { double t = 0.0; for (int i = 0; i < LOSS_BLOCK / 64; ++i) t += sh[i]; *out = t; }
}
// reconstruction loss and its gradient: X, pose-packed [rows, 64]; columns >= C carry nothing.  part[blockIdx.x] = this block's sum
__global__ __launch_bounds__(LOSS_BLOCK) void recon_loss_kernel(const float* __restrict__ Xp, const float* __restrict__ pose_p, int rows, int C,
                                                                float scale, float* __restrict__ dXp, double* __restrict__ part) {
    const size_t i = (size_t)blockIdx.x * LOSS_BLOCK + threadIdx.x;
    double s = 0.0;
    if (i < (size_t)rows * PAD) {
        float d = 0.f;
        if ((int)(i % PAD) < C) { d = Xp[i] - pose_p[i]; s = (double)d * d; }
        dXp[i] = 2.f * scale * d;
    }
    block_partial(s, part + blockIdx.x);
}

// latent: KL term and the gradient w.r.t. [mu | logvar] from dz (decoder side) + the KL term (SeqConvVAE.py:159-169, 206-213)
// (nslab > 0: dz arrives as the strips' slabs of gemm_tn_adam_dx_kernel, [nslab][64][Dp], and is summed here in slab order -- what
// slab_sum_kernel would do in a launch of its own)
__global__ __launch_bounds__(LOSS_BLOCK) void latent_bwd_kernel(const float* __restrict__ mulv, const float* __restrict__ eps, const float* __restrict__ dz,
                                                                int B, int D, int Dp, float kw_over_B, float* __restrict__ dmulv, double* __restrict__ part,
                                                                int nslab, size_t slab_stride) {
    const size_t i = (size_t)blockIdx.x * LOSS_BLOCK + threadIdx.x;
    double s = 0.0;
    if (i < (size_t)B * Dp) {
        const int b = (int)(i / Dp), d = (int)(i - (size_t)b * Dp);
        float dmu = 0.f, dlv = 0.f;
        if (d < D) {
            const float mu = mulv[(size_t)b * 2 * Dp + d], lv = mulv[(size_t)b * 2 * Dp + Dp + d];
            float g;
            if (nslab > 0) {
                g = 0.f;
                for (int z0 = 0; z0 < nslab; z0 += 8) {          // (eight slab requests in flight; slab order)
                    float v[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = dz[(size_t)min(z0 + j, nslab - 1) * slab_stride + i];
#pragma unroll
                    for (int j = 0; j < 8; ++j) if (z0 + j < nslab) g = (z0 + j == 0) ? v[j] : g + v[j];
                }
            } else g = dz[i];
            const float ev = expf(lv);
            s = (double)(1.f + lv - mu * mu - ev);
            dmu = g + kw_over_B * mu;
            dlv = g * eps[(size_t)b * D + d] * 0.5f * expf(0.5f * lv) + kw_over_B * 0.5f * (ev - 1.f);
        }
        dmulv[(size_t)b * 2 * Dp + d] = dmu;
        dmulv[(size_t)b * 2 * Dp + Dp + d] = dlv;
    }
    block_partial(s, part + blockIdx.x);
}
// torch.optim.Adam (amsgrad off): g += wd * p; m, v moments; p -= lr / bc1 * m / (sqrt(v) / sqrt(bc2) + eps)
// (skip0 / skip1: element ranges, in arena order, that gemm_tn_adam_kernel has already stepped; n counts the elements outside them)
struct SkipRange { size_t begin, len; };
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, size_t n, float lr,
                            float b1, float b2, float eps, float wd, float bc1, float bc2_sqrt, float gscale, SkipRange skip0, SkipRange skip1) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    if (i >= skip0.begin) i += skip0.len;
    if (i >= skip1.begin) i += skip1.len;
    const float gg = g[i] * gscale + wd * p[i];
    const float mm = b1 * m[i] + (1.f - b1) * gg;
    const float vv = b2 * v[i] + (1.f - b2) * gg * gg;
    m[i] = mm; v[i] = vv;
    p[i] -= (lr / bc1) * mm / (sqrtf(vv) / bc2_sqrt + eps);
}

template <typename T>
static int talloc(gem_trainer* t, T** p, size_t n) {
    void* q = nullptr;
    GEM_HIP(hipMalloc(&q, std::max<size_t>(n, 1) * sizeof(T)));
    t->allocs.push_back(q);
    *p = static_cast<T*>(q);
    GEM_HIP(hipMemset(q, 0, std::max<size_t>(n, 1) * sizeof(T)));
    return 0;
}

// weight gradient of a layer (TAPS taps): slabs of `rps` rows into the layer's slab region (summed into G later by
// slab_sum_all_kernel), or -- a single slab -- straight into G at `og`
template <int TAPS>
static int weight_grad(gem_trainer* t, const float* dC, int ldc, const float* A, int lda, int rows, int N, int K, size_t og, float* slab, int rps,
                       hipStream_t s) {
    const int nslab = (rows + rps - 1) / rps;
    if (nslab > 1 && !slab) { set_error("train: weight-gradient slabs missing"); return 1; }
    hipLaunchKernelGGL(gemm_tn_kernel<TAPS>, dim3((N / 64) * (K / 64), TAPS, nslab), dim3(256), 0, s, dC, ldc, A, lda,
                       nslab > 1 ? slab : t->G + og, rows, N, K, t->T, rps);
    GEM_HIP(hipGetLastError());
    return 0;
}

template <int KW, int D>
static int launch_conv_rows(const dim3& grid, const float* W, const float* bias, const float* A, int lda, float* C, int ldc, int rows, int N, int K, int T,
                            hipStream_t s, const CrStats& st) {
    auto k = conv_rows_lds_kernel<KW, D>;
    constexpr size_t smem = std::max((size_t)KW * D * 8192, (size_t)KW * 4096 + CR_STATS_LDS);
    static_assert(smem <= 64 * 1024, "beyond 64 KB of dynamic LDS the kernel needs hipFuncAttributeMaxDynamicSharedMemorySize");
    hipLaunchKernelGGL(k, grid, dim3(64 * KW), smem, s, A, lda, W, bias, C, ldc, rows, N, K, T, st);
    GEM_HIP(hipGetLastError());
    return 0;
}
static int conv_rows(gem_trainer* t, const float* W, const float* bias, const float* A, int lda, float* C, int ldc, int rows, int N, int K, hipStream_t s,
                     const CrStats& st) {
    static const int rows_max = dev_env("GEM_CONV_ROWS_MAX") ? atoi(dev_env("GEM_CONV_ROWS_MAX")) : CONV_ROWS_MAX;
    if (rows > rows_max || K % 64 || N % 32 || dev_env("GEM_TRAIN_NO_CONV_ROWS")) return -1;
    const dim3 grid((rows + 31) / 32, N / 32);
    // waves per workgroup = K cuts of whole 32-wide chunks (K = 64: two waves, 128: four); eight waves when the tiles alone
    // do not fill the chip.  One ring slot per wave (the refill of the slot just read runs under the sixteen MFMAs: a second slot
    // measured the same).  tools/conv_rows_bench: 4.8 us (K <= 128), 7 (N 128, K 256),
    // 11 (N 512, K 256 and N 256, K 512) at 640 rows, against 9 + 4.5 and 13.5 + 4.9 for the tiled kernel and its split-K reduce
    if (K % 256 == 0 && (long)grid.x * grid.y <= t->h->n_cu) return launch_conv_rows<8, 1>(grid, W, bias, A, lda, C, ldc, rows, N, K, t->T, s, st);
    if (K % 128 == 0) return launch_conv_rows<4, 1>(grid, W, bias, A, lda, C, ldc, rows, N, K, t->T, s, st);
    return launch_conv_rows<2, 1>(grid, W, bias, A, lda, C, ldc, rows, N, K, t->T, s, st);
}

// [mu | logvar] = sum of the fc product's K slabs + bias, and z = mu + eps exp(0.5 logvar) (SeqConvVAE.py:159-169) in the same pass:
// splitk_reduce_kernel<EPI_BIAS> followed by reparam_kernel, same sums in the same order, one launch
__global__ __launch_bounds__(256) void fc_reduce_reparam_kernel(const float* __restrict__ slabs, int nslab, size_t slab_stride, const float* __restrict__ bias,
                                                                const float* __restrict__ eps, float* __restrict__ mulv, float* __restrict__ z, int B, int D,
                                                                int Dp) {
    const int n4 = Dp / 4;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= B * n4) return;
    const int b = i / n4, c = (i - b * n4) * 4;
    const size_t off_m = (size_t)b * 2 * Dp + c, off_l = off_m + Dp;
    f32x4 m = *reinterpret_cast<const f32x4*>(slabs + off_m), lv = *reinterpret_cast<const f32x4*>(slabs + off_l);
    for (int zz = 1; zz < nslab; ++zz) {
        m += *reinterpret_cast<const f32x4*>(slabs + (size_t)zz * slab_stride + off_m);
        lv += *reinterpret_cast<const f32x4*>(slabs + (size_t)zz * slab_stride + off_l);
    }
    m += *reinterpret_cast<const f32x4*>(bias + c);
    lv += *reinterpret_cast<const f32x4*>(bias + Dp + c);
    *reinterpret_cast<f32x4*>(mulv + off_m) = m;
    *reinterpret_cast<f32x4*>(mulv + off_l) = lv;
    f32x4 zv;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int k = c + q;
        zv[q] = k < D ? eps[(size_t)b * D + k] * expf(0.5f * lv[q]) + m[q] : 0.f;
    }
    *reinterpret_cast<f32x4*>(z + (size_t)b * Dp + c) = zv;
}

// A linear layer over B rows (64 at the reference's batch): pure weight streaming.  The few-rows kernel (gemm_rows.h) fills the chip
// at so few rows only when it may cut K; letting it "defer" the reduction gives it that freedom, and the slabs are summed (+ bias)
// right behind it.
// reparam_eps != nullptr (the fc layer, C = [mu | logvar]): z = mu + eps exp(0.5 logvar) is formed as well -- by the slab sum itself
// when there are slabs, by reparam_kernel otherwise
static int linear_gemm(gem_trainer* t, const Layer& L, int epi, const float* A, int lda, float* C, int ldc, int M, hipStream_t s,
                       const float* reparam_eps = nullptr) {
    gem_handle* h = t->h;
    h->ws.defer_reduce = true;
    const int rc = launch_gemm(h, L, epi, A, lda, nullptr, C, ldc, M, t->T, s, -1);
    h->ws.defer_reduce = false;
    const SlabSrc d = h->ws.deferred;
    h->ws.deferred = SlabSrc{};
    if (rc) return rc;
    if (d.base && reparam_eps && epi == EPI_BIAS && d.dyn_W == 0 && ldc == 2 * t->Dp && !dev_env("GEM_TRAIN_NO_FC_REPARAM")) {
        hipLaunchKernelGGL(fc_reduce_reparam_kernel, dim3((unsigned)((M * (t->Dp / 4) + 255) / 256)), dim3(256), 0, s, (const float*)d.base, d.nslab, d.stride,
                           L.bias, reparam_eps, C, t->z, M, t->D, t->Dp);
        GEM_HIP(hipGetLastError());
        return 0;
    }
    if (d.base && launch_splitk_reduce(h, epi, d.nslab, d.stride, L.bias, nullptr, C, M, L.N, ldc, nullptr, s, d.dyn_W, d.n_tiles)) return 1;
    if (reparam_eps) return launch_reparam(C, reparam_eps, nullptr, nullptr, nullptr, t->z, M, t->D, t->Dp, s);
    return 0;
}

// Backward-data of a linear layer from the weights' OWN layout: dX[b][k] = sum_n dY[b][n] W[n][k] is a contraction over the rows
// of W -- what gemm_tn_kernel does (rows-contracting, both operands row-major) once dY is transposed (1.3 MB at the reference's
// batch).  The n range is cut into slabs that fill the chip and are summed in slab order.  Replaces the adjoint image of the
// layer (a 2 x 126 MB transpose of both linear layers' weights per step in round 3) and the few-rows product that read it.
// bias_grad != nullptr: the layer's bias gradient (the column sums of dY) is wanted as well: formed by the transpose when the batch is one
// row tile, by colsum_kernel otherwise
static int linear_bwd_data(gem_trainer* t, const float* dY, const float* W, float* dX, int B, int N, int K, hipStream_t s, float* bias_grad) {
    const int Bp = pad64(B);
    const bool fold = bias_grad && Bp == 64;
    if (bias_grad && !fold) {
        hipLaunchKernelGGL(colsum_kernel, dim3(N / 16), dim3(BN_THREADS), 0, s, dY, B, N, bias_grad);
        GEM_HIP(hipGetLastError());
    }
    hipLaunchKernelGGL(transpose_pad_kernel, dim3(N / 64, Bp / 64), dim3(256), 0, s, dY, B, N, Bp, t->dYT, fold ? bias_grad : nullptr);
    GEM_HIP(hipGetLastError());
    const int tiles = (Bp / 64) * (K / 64);
    int nslab = (2 * t->h->n_cu + tiles - 1) / tiles;
    nslab = std::max(1, std::min(nslab, std::min(N / 64, t->lin_slab_cap)));
    int rps = ((N + nslab - 1) / nslab + 31) / 32 * 32;
    nslab = (N + rps - 1) / rps;
    float* out = nslab > 1 ? t->lin_slab : dX;
    if (nslab == 1 && Bp != B) { set_error("train: linear backward-data without slabs needs a batch that is a multiple of 64"); return 1; }
    hipLaunchKernelGGL(gemm_tn_kernel<1>, dim3(tiles, 1, nslab), dim3(256), 0, s, (const float*)t->dYT, Bp, W, K, out, N, Bp, K, t->T, rps);
    GEM_HIP(hipGetLastError());
    if (nslab > 1) {
        const size_t n_out = (size_t)B * K;
        hipLaunchKernelGGL(slab_sum_kernel, dim3((unsigned)((n_out / 4 + 255) / 256)), dim3(256), 0, s, (const float*)t->lin_slab, nslab, (size_t)Bp * K, dX, n_out);
        GEM_HIP(hipGetLastError());
    }
    return 0;
}

// Strip length (n tiles per workgroup) of gemm_tn_adam_dx_kernel; 0 = the layer / batch does not fit the kernel (more than 64
// windows, an odd number of n tiles, more strips than slabs): the separate kernels run.
static int fused_backward_strip(const gem_trainer* t, const TrainLinear& l, int B) {
    if (B > 64 || dev_env("GEM_TRAIN_NO_FUSED_DX")) return 0;
    const int nt = l.N / 64, nkt = l.K / 64;
    if (const char* f = dev_env("GEM_TRAIN_TPS")) { const char* c = strchr(f, ','); const int tps = (&l == &t->dec_in && c) ? atoi(c + 1) : atoi(f); return (tps >= 2 && tps % 2 == 0 && nt % tps == 0 && nt / tps <= t->dx_slab_cap) ? tps : 0; }
    // The kernel is bound by what one CU gets through (two products and the Adam arithmetic per tile), not by HBM: the strip length that
    // leaves every CU the same number of tiles wins -- tiles per CU = ceil(workgroups / CUs) x tps (reference VAE on 256 CUs:
    // fc 80 k tiles x 16 strips of 4 = 1280 workgroups, decoder_input 32 x 8 strips of 10 = 256; measured 0.721 ms per step
    // against 0.731-0.770 for the other divisors); ties go to the longer strip (fewer dX slabs).
    int best = 0; long best_cost = 0;
    for (int tps = 2; tps <= nt; tps += 2) {
        if (nt % tps || nt / tps > t->dx_slab_cap) continue;
        const long wgs = (long)nkt * (nt / tps), cost = (wgs + t->h->n_cu - 1) / t->h->n_cu * tps;
        if (best == 0 || cost <= best_cost) { best = tps; best_cost = cost; }
    }
    return best;
}
static int linear_fused_backward(gem_trainer* t, const TrainLinear& l, int tps, const float* dY, const float* A, float* dX, int B, const AdamScalars& ad,
                                 hipStream_t s) {
    const int nstrip = l.N / 64 / tps;
    hipLaunchKernelGGL(gemm_tn_adam_dx_kernel, dim3((l.K / 64) * nstrip), dim3(DX_THREADS), 0, s, dY, A, B, l.N, l.K, tps, t->P + l.ow, t->M1 + l.ow, t->M2 + l.ow, ad,
                       t->dx_slab, t->G + l.ob);
    GEM_HIP(hipGetLastError());
    if (!dX) return 0;          // (the consumer sums the nstrip slabs itself: latent_bwd_kernel)
    const size_t n_out = (size_t)B * l.K;
    hipLaunchKernelGGL(slab_sum_kernel, dim3((unsigned)((n_out / 4 + 255) / 256)), dim3(256), 0, s, (const float*)t->dx_slab, nstrip, (size_t)64 * l.K, dX, n_out);
    GEM_HIP(hipGetLastError());
    return 0;
}

}  // namespace gem

using namespace gem;

extern "C" {

int gem_trainer_create(const gem_config* cfg, gem_trainer** out) {
    if (!cfg || !out) { set_error("gem_trainer_create: null argument"); return 1; }
    std::unique_ptr<gem_trainer, void (*)(gem_trainer*)> t(new gem_trainer(), gem_trainer_destroy);
    t->cfg = *cfg;
    if (gem_create(cfg, &t->h)) return 1;
    gem_handle* h = t->h;
    t->T = h->T; t->C = h->C; t->Cp = h->Cp; t->D = h->D; t->Dp = h->Dp; t->top = h->top; t->topp = h->topp; t->Bmax = cfg->max_windows;
    const int nh = cfg->n_hidden, T = t->T;
    const size_t rows = (size_t)t->Bmax * T;
    size_t off = 0, soff = 0;
    auto add_conv = [&](std::vector<TrainConv>& v, int ci, int co, bool bn) {
        TrainConv c; c.K = pad64(ci); c.N = pad64(co); c.bn = bn;
        c.ow = off; off += (size_t)3 * c.N * c.K;
        c.ob = off; off += c.N;
        if (bn) { c.og = off; off += c.N; c.obe = off; off += c.N; c.os = soff; soff += 2 * (size_t)c.N; }
        v.push_back(c);
    };
    { int ci = t->C; for (int i = 0; i < nh; ++i) { add_conv(t->enc, ci, cfg->hidden[i], true); ci = cfg->hidden[i]; } }
    t->fc.K = T * t->topp; t->fc.N = 2 * t->Dp; t->fc.ow = off; off += (size_t)t->fc.N * t->fc.K; t->fc.ob = off; off += t->fc.N;
    t->dec_in.K = t->Dp; t->dec_in.N = T * t->topp; t->dec_in.ow = off; off += (size_t)t->dec_in.N * t->dec_in.K; t->dec_in.ob = off; off += t->dec_in.N;
    for (int i = nh - 1; i >= 1; --i) add_conv(t->dec, cfg->hidden[i], cfg->hidden[i - 1], true);
    add_conv(t->dec, cfg->hidden[0], cfg->hidden[0], true);
    add_conv(t->dec, cfg->hidden[0], t->C, false);
    t->n_params = off; t->n_stats = soff;
    gem_trainer* p = t.get();
    if (talloc(p, &p->P, off) || talloc(p, &p->G, off) || talloc(p, &p->M1, off) || talloc(p, &p->M2, off) || talloc(p, &p->S, soff)) return 1;
    size_t max_width = PAD;
    const size_t conv_slabs = std::min<size_t>((rows + TN_ROWS_CONV - 1) / TN_ROWS_CONV, TN_CONV_SLABS_MAX), lin_slabs = ((size_t)p->Bmax + TN_ROWS_LINEAR - 1) / TN_ROWS_LINEAR;
    std::vector<AdjDesc> adj;
    std::vector<SumDesc> sums;          // conv layers first, then (only when they need slabs) the two linear layers
    for (auto* v : {&p->enc, &p->dec})
        for (auto& c : *v) {
            const size_t nw = (size_t)3 * c.N * c.K;
            if (talloc(p, &c.Y, rows * c.N) || talloc(p, &c.out, rows * c.N) || talloc(p, &c.mean, (size_t)c.N) || talloc(p, &c.invstd, (size_t)c.N)) return 1;
            if (c.bn && talloc(p, &c.dY, rows * c.N)) return 1;
            if (&c != &p->enc.front()) {          // (nothing flows back through the first encoder conv)
                if (talloc(p, &c.adj, nw)) return 1;
                adj.push_back(AdjDesc{p->P + c.ow, c.adj, 3, c.N, c.K, 3 * (c.N / 64) * (c.K / 64)});
            }
            if (conv_slabs > 1) {
                if (talloc(p, &c.slab, nw * conv_slabs)) return 1;
                sums.push_back(SumDesc{c.slab, p->G + c.ow, nw});
            }
            max_width = std::max(max_width, (size_t)std::max(c.N, c.K));
        }
    p->n_sum = (int)sums.size();
    for (TrainLinear* l : {&p->fc, &p->dec_in}) {
        const size_t nw = (size_t)l->N * l->K;
        if (lin_slabs > 1) {
            if (talloc(p, &l->slab, nw * lin_slabs)) return 1;
            sums.push_back(SumDesc{l->slab, p->G + l->ow, nw});
        }
    }
    for (const auto& d : adj) p->adj_tiles = std::max(p->adj_tiles, d.tiles);
    for (const auto& d : sums) p->sum_max = std::max(p->sum_max, (size_t)d.n);
    p->n_adj = (int)adj.size();
    if (talloc(p, &p->adj_tab, adj.size()) || talloc(p, &p->sum_tab, sums.size())) return 1;
    GEM_HIP(hipMemcpy(p->adj_tab, adj.data(), adj.size() * sizeof(AdjDesc), hipMemcpyHostToDevice));
    if (!sums.empty()) GEM_HIP(hipMemcpy(p->sum_tab, sums.data(), sums.size() * sizeof(SumDesc), hipMemcpyHostToDevice));
    p->part_recon = (int)((rows * PAD + LOSS_BLOCK - 1) / LOSS_BLOCK);
    p->part_latent = (int)(((size_t)p->Bmax * p->Dp + LOSS_BLOCK - 1) / LOSS_BLOCK);
    p->bn_nrb_cap = std::max((int)((rows + BNL_ROWS - 1) / BNL_ROWS), BNF_ROWS_MAX / 32);          // (bnl_*: 128-row blocks; bnf_*: 32-row conv tiles)
    if (talloc(p, &p->bn_part, (size_t)p->bn_nrb_cap * max_width * 3)) return 1;
    if (talloc(p, &p->pose_p, rows * PAD) || talloc(p, &p->mulv, (size_t)p->Bmax * 2 * p->Dp) || talloc(p, &p->z, (size_t)p->Bmax * p->Dp) ||
        talloc(p, &p->h0, rows * p->topp) || talloc(p, &p->Xp, rows * PAD) || talloc(p, &p->gA, rows * max_width) || talloc(p, &p->gB, rows * max_width) ||
        talloc(p, &p->dmulv, (size_t)p->Bmax * 2 * p->Dp) || talloc(p, &p->dz, (size_t)p->Bmax * p->Dp) ||
        talloc(p, &p->dYT, (size_t)std::max(p->fc.N, p->dec_in.N) * pad64(p->Bmax)) ||
        talloc(p, &p->lin_slab, (size_t)p->lin_slab_cap * pad64(p->Bmax) * std::max(p->fc.K, p->dec_in.K)) ||
        talloc(p, &p->dx_slab, (size_t)p->dx_slab_cap * 64 * std::max(p->fc.K, p->dec_in.K)) ||
        talloc(p, &p->red, (size_t)8 + p->part_recon + p->part_latent))
        return 1;
    { std::vector<TnDesc> tn;
      int tile0 = 0;
      for (auto* v : {&p->enc, &p->dec})
          for (size_t i = 0; i < v->size(); ++i) {
              TrainConv& c = (*v)[i];
              const float* a_in = i > 0 ? (*v)[i - 1].out : (v == &p->enc ? p->pose_p : p->h0);
              tn.push_back(TnDesc{c.bn ? c.dY : p->gA, a_in, c.slab, p->G + c.ow, c.N, c.K, tile0});      // (no BatchNorm: the loss gradient itself, in gA)
              tile0 += (c.N / 64) * (c.K / 64);
          }
      if (tn.size() > (size_t)TN_MAX_LAYERS) { set_error("train: more conv layers than the weight-gradient table holds"); return 1; }
      TnTable* tv = new TnTable();
      for (size_t i = 0; i < tn.size(); ++i) tv->d[i] = tn[i];
      tv->n = (int)tn.size();
      p->tn_tab = tv; p->n_tn = (int)tn.size(); p->tn_tiles = tile0; }
    *out = t.release();
    return 0;
}

void gem_trainer_destroy(gem_trainer* t) {
    if (!t) return;
    if (t->h) { (void)hipSetDevice(t->h->cfg.device); (void)hipDeviceSynchronize(); }
    for (void* p : t->allocs) (void)hipFree(p);
    delete static_cast<gem::TnTable*>(t->tn_tab);
    if (t->h) gem_destroy(t->h);
    delete t;
}

int gem_trainer_sizes(gem_trainer* t, int64_t* n_params, int64_t* n_stats) {
    if (!t) { set_error("gem_trainer_sizes: null trainer"); return 1; }
    if (n_params) *n_params = (int64_t)t->n_params;
    if (n_stats) *n_stats = (int64_t)t->n_stats;
    return 0;
}

/* what = 0 parameters, 1 gradients, 2 running statistics, 3 / 4 first / second Adam moment; host pointers */
int gem_trainer_upload(gem_trainer* t, int what, const float* src, int64_t n) {
    if (!t || !src) { set_error("gem_trainer_upload: null argument"); return 1; }
    float* dst = what == 0 ? t->P : what == 2 ? t->S : what == 3 ? t->M1 : what == 4 ? t->M2 : nullptr;
    const size_t want = what == 2 ? t->n_stats : t->n_params;
    if (!dst || (size_t)n != want) { set_error("gem_trainer_upload: bad selector or size"); return 1; }
    GEM_HIP(hipSetDevice(t->h->cfg.device));
    GEM_HIP(hipDeviceSynchronize());
    GEM_HIP(hipMemcpy(dst, src, want * sizeof(float), hipMemcpyHostToDevice));
    if (what == 0) t->step = 0;
    return 0;
}
int gem_trainer_download(gem_trainer* t, int what, float* dst, int64_t n) {
    if (!t || !dst) { set_error("gem_trainer_download: null argument"); return 1; }
    const float* src = what == 0 ? t->P : what == 1 ? t->G : what == 2 ? t->S : what == 3 ? t->M1 : what == 4 ? t->M2 : nullptr;
    const size_t want = what == 2 ? t->n_stats : t->n_params;
    if (!src || (size_t)n != want) { set_error("gem_trainer_download: bad selector or size"); return 1; }

    GEM_HIP(hipSetDevice(t->h->cfg.device));
    GEM_HIP(hipDeviceSynchronize());
    GEM_HIP(hipMemcpy(dst, src, want * sizeof(float), hipMemcpyDeviceToHost));
    if (what == 1 && t->grads_partial) {
        // the last step ran with update = 2: the linear layers formed their weight gradients inside their Adam step and left nothing
        // in the arena -- those ranges read as NaN ("not available"), never as the stale values of an earlier step
        const float nanv = std::numeric_limits<float>::quiet_NaN();
        std::fill(dst + t->fc.ow, dst + t->fc.ow + (size_t)t->fc.N * t->fc.K, nanv);
        std::fill(dst + t->dec_in.ow, dst + t->dec_in.ow + (size_t)t->dec_in.N * t->dec_in.K, nanv);
    }
    return 0;
}
int gem_trainer_set_step(gem_trainer* t, int64_t step) {
    if (!t || step < 0) { set_error("gem_trainer_set_step: bad argument"); return 1; }
    t->step = (long)step;
    return 0;
}

static int apply_adam(gem_trainer* t, const gem_train_opts* o, double grad_scale, hipStream_t s, bool fused_linears);

int gem_trainer_step(gem_trainer* t, int B, const float* d_pose, const float* d_eps, const gem_train_opts* o, int update, double* d_losses,
                     void* stream) {
    if (!t || !d_pose || !d_eps || !o) { set_error("gem_trainer_step: null argument"); return 1; }
    if (B < 2 || B > t->Bmax) { set_error("gem_trainer_step: need 2 <= B <= max_windows (BatchNorm statistics)"); return 1; }
    if (update < 0 || update > 2) { set_error("gem_trainer_step: update must be 0 (gradients only), 1 (Adam from the arena) or 2 (fused linear layers)"); return 1; }
    t->grads_partial = update == 2;
    gem_handle* h = t->h;
    GEM_HIP(hipSetDevice(h->cfg.device));
    hipStream_t s = (hipStream_t)stream;
    const int T = t->T, rows = B * T;
    h->precision = GEM_PRECISION_F32;
    // update = 2: the linear layers' weight gradients are formed inside their Adam step (gemm_tn_adam_kernel) and NOT left in the
    // gradient arena -- the training loop's mode (networks/train.py:77-83 never looks at p.grad)
    const bool fused = update == 2;
    AdamScalars ad{};
    if (fused) {
        const double st = (double)(t->step + 1);
        const double bc1 = 1.0 - std::pow(o->beta1, st), bc2 = 1.0 - std::pow(o->beta2, st);
        ad = AdamScalars{(float)o->lr / (float)bc1, (float)o->beta1, (float)o->beta2, (float)o->eps, (float)o->weight_decay, (float)std::sqrt(bc2)};
    }
    auto linear_step = [&](const TrainLinear& l, const float* dC, const float* A) -> int {
        hipLaunchKernelGGL(gemm_tn_adam_kernel, dim3((l.N / 64) * (l.K / 64)), dim3(256), 0, s, dC, l.N, A, l.K, B, l.N, l.K, t->P + l.ow, t->M1 + l.ow,
                           t->M2 + l.ow, ad);
        GEM_HIP(hipGetLastError());
        return 0;
    };
    // ---- the adjoint weight images of this step's parameters (one launch for all layers)
    hipLaunchKernelGGL(adjoint_all_kernel, dim3(t->adj_tiles, t->n_adj), dim3(256), 0, s, (const AdjDesc*)t->adj_tab);
    GEM_HIP(hipGetLastError());
    // ---- forward (train mode)
    if (launch_pack_pose(d_pose, t->pose_p, rows, t->C, s)) return 1;
    const bool bn_fused = rows <= BNF_ROWS_MAX && !dev_env("GEM_TRAIN_NO_BN_FUSE");
    auto conv_fwd = [&](TrainConv& c, const float* in) -> int {
        Layer L; L.taps = 3; L.K = c.K; L.N = c.N; L.w = t->P + c.ow; L.bias = t->P + c.ob;
        // (bn_fused: the conv's epilogue leaves the BatchNorm sums per 32-row tile; the apply kernels are elementwise)
        CrStats st{};
        if (c.bn && bn_fused) st.part = t->bn_part;
        const int rc = conv_rows(t, L.w, L.bias, in, c.K, c.bn ? c.Y : c.out, c.N, rows, c.N, c.K, s, st);
        if (rc > 0 || (rc < 0 && launch_gemm(h, L, EPI_BIAS, in, c.K, nullptr, c.bn ? c.Y : c.out, c.N, rows, T, s, -1))) return 1;
        if (c.bn) {
            if (rc == 0 && st.part) {
                hipLaunchKernelGGL(bnf_fwd_apply_kernel, dim3(c.N / 64, (rows + 63) / 64), dim3(256), 0, s, (const float*)c.Y, rows, c.N,
                                   (const double*)t->bn_part, (rows + 31) / 32, (const float*)(t->P + c.og), (const float*)(t->P + c.obe), t->S + c.os,
                                   t->S + c.os + c.N, c.mean, c.invstd, c.out, (float)o->bn_momentum, (float)BN_EPS);
            } else if (rows <= BN_REGS * BN_GROUPS) {
                hipLaunchKernelGGL(bn_train_fwd_kernel, dim3(c.N / 16), dim3(BN_THREADS), 0, s, (const float*)c.Y, rows, c.N, (const float*)(t->P + c.og),
                                   (const float*)(t->P + c.obe), t->S + c.os, t->S + c.os + c.N, c.mean, c.invstd, c.out, (float)o->bn_momentum, (float)BN_EPS);
            } else {
                const int nrb = (rows + BNL_ROWS - 1) / BNL_ROWS;
                hipLaunchKernelGGL(bnl_fwd_stats_kernel, dim3(c.N / 64, nrb), dim3(256), 0, s, (const float*)c.Y, rows, c.N, t->bn_part);
                GEM_HIP(hipGetLastError());
                hipLaunchKernelGGL(bnl_fwd_apply_kernel, dim3(c.N / 64, nrb), dim3(256), 0, s, (const float*)c.Y, rows, c.N, (const double*)t->bn_part, nrb,
                                   (const float*)(t->P + c.og), (const float*)(t->P + c.obe), t->S + c.os, t->S + c.os + c.N, c.mean, c.invstd, c.out,
                                   (float)o->bn_momentum, (float)BN_EPS);
            }
            GEM_HIP(hipGetLastError());
        }
        return 0;
    };
    const float* in = t->pose_p;
    for (auto& c : t->enc) { if (conv_fwd(c, in)) return 1; in = c.out; }
    { Layer L; L.taps = 1; L.K = t->fc.K; L.N = t->fc.N; L.w = t->P + t->fc.ow; L.bias = t->P + t->fc.ob;
      if (linear_gemm(t, L, EPI_BIAS, in, L.K, t->mulv, L.N, B, s, d_eps)) return 1; }
    { Layer L; L.taps = 1; L.K = t->dec_in.K; L.N = t->dec_in.N; L.w = t->P + t->dec_in.ow; L.bias = t->P + t->dec_in.ob;
      if (linear_gemm(t, L, EPI_BIAS, t->z, L.K, t->h0, L.N, B, s)) return 1; }
    in = t->h0;
    for (auto& c : t->dec) { if (conv_fwd(c, in)) return 1; in = c.out; }
    const float* X = t->dec.back().out;
    // ---- loss + its gradient w.r.t. the decoded pose
    const double n_recon = o->recon_sum ? 1.0 : (double)rows * t->C;
    double* part_recon = t->red + 8;
    double* part_latent = part_recon + t->part_recon;
    const int n_pr = (int)(((size_t)rows * PAD + LOSS_BLOCK - 1) / LOSS_BLOCK), n_pl = (int)(((size_t)B * t->Dp + LOSS_BLOCK - 1) / LOSS_BLOCK);
    hipLaunchKernelGGL(recon_loss_kernel, dim3(n_pr), dim3(LOSS_BLOCK), 0, s, X, (const float*)t->pose_p, rows, t->C, (float)(1.0 / n_recon), t->gA, part_recon);
    GEM_HIP(hipGetLastError());
    // ---- backward: decoder
    auto bn_bwd = [&](const float* dOut, const TrainConv& c, float* dY) -> int {
        if (rows <= BN_REGS * BN_GROUPS) {
            hipLaunchKernelGGL(bn_train_bwd_kernel, dim3(c.N / 16), dim3(BN_THREADS), 0, s, dOut, (const float*)c.out, (const float*)c.Y, rows, c.N,
                               (const float*)(t->P + c.og), (const float*)c.mean, (const float*)c.invstd, dY, t->G + c.og, t->G + c.obe, t->G + c.ob);
        } else {
            const int nrb = (rows + BNL_ROWS - 1) / BNL_ROWS;
            hipLaunchKernelGGL(bnl_bwd_stats_kernel, dim3(c.N / 64, nrb), dim3(256), 0, s, dOut, (const float*)c.out, (const float*)c.Y, rows, c.N,
                               (const float*)c.mean, (const float*)c.invstd, t->bn_part);
            GEM_HIP(hipGetLastError());
            hipLaunchKernelGGL(bnl_bwd_apply_kernel, dim3(c.N / 64, nrb), dim3(256), 0, s, dOut, (const float*)c.out, (const float*)c.Y, rows, c.N,
                               (const double*)t->bn_part, nrb, (const float*)(t->P + c.og), (const float*)c.mean, (const float*)c.invstd, dY,
                               t->G + c.og, t->G + c.obe, t->G + c.ob);
        }
        GEM_HIP(hipGetLastError());
        return 0;
    };
    // The chain: dOut (gradient w.r.t. a layer's output) -> BatchNorm backward -> the layer's own dY buffer -> backward-data conv ->
    // `run` = the next layer's dOut.  The layers' dY buffers (and gA, the loss gradient = the last conv's dY) stay untouched until
    // the ONE weight-gradient launch behind the chain.
    // bn_fused: the backward-data conv of a layer forms, in its epilogue, the BatchNorm-backward sums of the layer BELOW (whose dOut it
    // produces) and stores dz = dOut * LeakyReLU'(out) in dOut's place; that layer's BatchNorm backward is then elementwise
    auto bwd_conv = [&](const TrainConv& c, const float* dY, float* dst, const TrainConv* below, bool* have_dz) -> int {
        Layer L; L.taps = 3; L.K = c.N; L.N = c.K; L.w = c.adj; L.bias = nullptr;
        CrStats st{};
        if (below && below->bn && bn_fused) st = CrStats{t->bn_part, below->out, below->Y, below->mean, below->invstd, LEAKY_SLOPE};
        const int rc = conv_rows(t, L.w, nullptr, dY, c.N, dst, c.K, rows, c.K, c.N, s, st);
        if (rc > 0 || (rc < 0 && launch_gemm(h, L, EPI_NONE, dY, c.N, nullptr, dst, c.K, rows, T, s, -1))) return 1;
        *have_dz = rc == 0 && st.part != nullptr;
        return 0;
    };
    auto bn_bwd_any = [&](const float* dOut_or_dz, bool is_dz, const TrainConv& c) -> int {
        if (!is_dz) return bn_bwd(dOut_or_dz, c, c.dY);
        hipLaunchKernelGGL(bnf_bwd_apply_kernel, dim3(c.N / 64, (rows + 63) / 64), dim3(256), 0, s, dOut_or_dz, (const float*)c.Y, rows, c.N,
                           (const double*)t->bn_part, (rows + 31) / 32, (const float*)(t->P + c.og), (const float*)c.mean, (const float*)c.invstd, c.dY,
                           t->G + c.og, t->G + c.obe, t->G + c.ob);
        GEM_HIP(hipGetLastError());
        return 0;
    };
    const float* dOut = t->gA;
    float* run = t->gB;
    bool have_dz = false;
    for (int i = (int)t->dec.size() - 1; i >= 0; --i) {
        TrainConv& c = t->dec[i];
        const float* dY = dOut;
        if (c.bn) {
            if (bn_bwd_any(dOut, have_dz, c)) return 1;
            dY = c.dY;
        }          // (no BatchNorm -- the last conv: its bias gradient, the column sums of dY, rides with the weight-gradient launch)
        if (bwd_conv(c, dY, run, i > 0 ? &t->dec[i - 1] : nullptr, &have_dz)) return 1;
        dOut = run;
    }
    float* g = run;
    // g = dh0 [B, T*topp]: decoder_input
    int dz_slabs = 0;          // > 0: dz is left as that many slabs for latent_bwd_kernel
    bool enc_top_dz = false;
    { const TrainLinear& l = t->dec_in;
      const int tps = fused ? fused_backward_strip(t, l, B) : 0;
      if (tps) { if (linear_fused_backward(t, l, tps, g, t->z, nullptr, B, ad, s)) return 1; dz_slabs = l.N / 64 / tps; }
      else {
          if (!fused && weight_grad<1>(t, g, l.N, t->z, l.K, B, l.N, l.K, l.ow, l.slab, TN_ROWS_LINEAR, s)) return 1;
          if (linear_bwd_data(t, g, t->P + l.ow, t->dz, B, l.N, l.K, s, t->G + l.ob)) return 1;
          if (fused && linear_step(l, g, t->z)) return 1;          // (behind the backward-data product: it reads the weights)
      } }
    hipLaunchKernelGGL(latent_bwd_kernel, dim3(n_pl), dim3(LOSS_BLOCK), 0, s, (const float*)t->mulv, d_eps, dz_slabs ? (const float*)t->dx_slab : (const float*)t->dz, B,
                       t->D, t->Dp, (float)(o->kld_weight / B), t->dmulv, part_latent, dz_slabs, (size_t)64 * t->dec_in.K);
    GEM_HIP(hipGetLastError());
    // fc_mu | fc_var
    { const TrainLinear& l = t->fc;
      const float* flat = t->enc.back().out;
      // (g = gB again: decoder_input's backward above has consumed dh0; gA still holds the last conv's dY)
      const int tps = fused ? fused_backward_strip(t, l, B) : 0;
      const TrainConv& top = t->enc.back();
      if (tps && bn_fused && top.N % 32 == 0 && l.K == T * top.N) {
          // dX = dOut of the encoder's last block: summed from the slabs together with that block's BatchNorm-backward sums
          if (linear_fused_backward(t, l, tps, t->dmulv, flat, nullptr, B, ad, s)) return 1;
          const CrStats st{t->bn_part, top.out, top.Y, top.mean, top.invstd, LEAKY_SLOPE};
          hipLaunchKernelGGL(dx_sum_stats_kernel, dim3((rows + 31) / 32, top.N / 32), dim3(256), 0, s, (const float*)t->dx_slab, l.N / 64 / tps, (size_t)64 * l.K, g,
                             rows, top.N, st);
          GEM_HIP(hipGetLastError());
          enc_top_dz = true;
      } else if (tps) { if (linear_fused_backward(t, l, tps, t->dmulv, flat, g, B, ad, s)) return 1; }
      else {
          if (!fused && weight_grad<1>(t, t->dmulv, l.N, flat, l.K, B, l.N, l.K, l.ow, l.slab, TN_ROWS_LINEAR, s)) return 1;
          if (linear_bwd_data(t, t->dmulv, t->P + l.ow, g, B, l.N, l.K, s, t->G + l.ob)) return 1;
          if (fused && linear_step(l, t->dmulv, flat)) return 1;
      } }
    // encoder
    have_dz = enc_top_dz;          // (the top of the encoder chain comes out of the fc layer's backward: dz with the sums, or plain dOut)
    for (int i = (int)t->enc.size() - 1; i >= 0; --i) {
        TrainConv& c = t->enc[i];
        if (bn_bwd_any(g, have_dz, c)) return 1;
        if (i > 0 && bwd_conv(c, c.dY, g, &t->enc[i - 1], &have_dz)) return 1;
    }
    // every conv layer's weight gradient (slabs of conv_slab_rows(rows) rows, or straight into the gradient arena).  Measured and not
    // kept: the same tiles with the rows split over the eight waves of a 512-thread workgroup instead of over the grid (wave-private
    // LDS-DMA staging, no slabs, no slab sum): 267 workgroups on 256 CUs -- a CU that gets two of them takes twice as long, 31 us
    // against 25 + 10.6 for this launch and its slab sum at batch 64, 3.34 against 3.21 ms per step at batch 1024
    { const TrainConv& last = t->dec.back();
      StepTail tail{part_recon, part_latent, t->red + 4, d_losses, n_recon, o->kld_weight, n_pr, n_pl, B, t->gA, t->G + last.ob, last.N};
      const int rps = conv_slab_rows(rows), nslab = (rows + rps - 1) / rps;
      if (nslab > 1 && t->n_sum != t->n_tn) { set_error("train: weight-gradient slabs missing"); return 1; }
      hipLaunchKernelGGL(gemm_tn3_all_kernel, dim3(t->tn_tiles + 1 + last.N / 16, 3, nslab), dim3(256), 0, s, *static_cast<const TnTable*>(t->tn_tab), t->tn_tiles,
                         rows, T, rps, nslab, tail);
      GEM_HIP(hipGetLastError()); }
    // weight-gradient slabs -> the gradient arena (slab order: deterministic)
    { const int ns_conv = (rows + conv_slab_rows(rows) - 1) / conv_slab_rows(rows), ns_lin = (B + TN_ROWS_LINEAR - 1) / TN_ROWS_LINEAR;
      // (every conv layer's entry was cut into the same slabs of conv_slab_rows(rows) rows: ONE slab count serves the whole table; the two
      // linear layers' entries sit behind the conv entries)
      if (ns_conv > 1 && t->n_sum > 0) {
          hipLaunchKernelGGL(slab_sum_all_kernel, dim3((unsigned)((t->sum_max / 4 + 255) / 256), t->n_sum), dim3(256), 0, s, (const SumDesc*)t->sum_tab, ns_conv);
          GEM_HIP(hipGetLastError());
      }
      if (ns_lin > 1 && !fused)
          hipLaunchKernelGGL(slab_sum_all_kernel, dim3((unsigned)((t->sum_max / 4 + 255) / 256), 2), dim3(256), 0, s, (const SumDesc*)t->sum_tab + t->n_sum, ns_lin);
      GEM_HIP(hipGetLastError()); }
    if (update) return apply_adam(t, o, 1.0, s, fused);
    return 0;
}

// Adam over the gradient arena; fused_linears: the two linear layers' weights have been stepped by gemm_tn_adam_kernel already
// (with THIS step's count: t->step is advanced here)
static int apply_adam(gem_trainer* t, const gem_train_opts* o, double grad_scale, hipStream_t s, bool fused_linears) {
    ++t->step;
    const double bc1 = 1.0 - std::pow(o->beta1, (double)t->step), bc2 = 1.0 - std::pow(o->beta2, (double)t->step);
    SkipRange k0{t->n_params, 0}, k1{t->n_params, 0};
    size_t n = t->n_params;
    if (fused_linears) {
        k0 = SkipRange{t->fc.ow, (size_t)t->fc.N * t->fc.K};
        k1 = SkipRange{t->dec_in.ow, (size_t)t->dec_in.N * t->dec_in.K};          // (arena order: fc before decoder_input)
        n -= k0.len + k1.len;            // (the kernel walks the compacted index space and re-inserts the ranges in this order)
    }
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, t->P, (const float*)t->G, t->M1, t->M2, n,
                       (float)o->lr, (float)o->beta1, (float)o->beta2, (float)o->eps, (float)o->weight_decay, (float)bc1, (float)std::sqrt(bc2),
                       (float)grad_scale, k0, k1);
    GEM_HIP(hipGetLastError());
    return 0;
}

int gem_trainer_apply(gem_trainer* t, const gem_train_opts* o, double grad_scale, void* stream) {
    if (!t || !o) { set_error("gem_trainer_apply: null argument"); return 1; }
    if (t->grads_partial) { set_error("gem_trainer_apply: the last step ran with update = 2, the gradient arena is incomplete"); return 1; }
    GEM_HIP(hipSetDevice(t->h->cfg.device));
    return apply_adam(t, o, grad_scale, (hipStream_t)stream, false);
}

int gem_trainer_arena(gem_trainer* t, int what, void** d_ptr, int64_t* n) {
    if (!t || !d_ptr) { set_error("gem_trainer_arena: null argument"); return 1; }
    float* p = what == 0 ? t->P : what == 1 ? t->G : what == 2 ? t->S : what == 3 ? t->M1 : what == 4 ? t->M2 : nullptr;
    if (!p) { set_error("gem_trainer_arena: bad selector"); return 1; }
    if (what == 1 && t->grads_partial) { set_error("gem_trainer_arena: the last step ran with update = 2, the gradient arena is incomplete"); return 1; }
    *d_ptr = p;
    if (n) *n = (int64_t)(what == 2 ? t->n_stats : t->n_params);
    return 0;
}

}  // extern "C"
