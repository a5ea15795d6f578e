// Training step of the motion VAE on the device: the loop body of networks/train.py:65-108 --
// forward in TRAIN mode (BatchNorm batch statistics, running statistics updated), the VAE loss of
// networks/models/SeqConvVAE.py:191-219 (M_N form: mean-squared reconstruction error + kld_weight * KL), backward (data AND
// weight gradients) and one torch.optim.Adam step (L2 weight decay folded into the gradient, bias-corrected moments).
//
// Parameters live in ONE fp32 arena in the padded, packed layouts the GEMM kernels read directly (every conv as its equivalent
// Conv1d taps [3][N][K], k contiguous; fc_mu | fc_var stacked; decoder_input time-major), with same-shaped arenas for gradient
// and the two Adam moments; the reference's checkpoint schema is a permutation of that arena (host side: vae_train.py).  Padded
// entries are zero and stay zero (their gradients are sums over zero activations).
//   forward / backward-DATA products: the fp32 MFMA kernels of the optimiser (launch_gemm; adjoint weight images are re-packed
//     from the arena every step)
//   weight gradients: gemm_tn_kernel, dW[tap][n][k] = sum_r dC[r][n] * A[r + tap - 1][k] (contraction over the ROWS, both
//     operands row-major: staged through LDS, v_mfma_f32_16x16x4_f32), row range cut into slabs, summed in slab order
//   BatchNorm (+ LeakyReLU) forward / backward, bias gradients, the latent / loss gradients, Adam: one small kernel each
// Everything is enqueued on the caller's stream; no host synchronisation inside a step.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <memory>

#include "gem_internal.h"

namespace gem {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct TrainConv {
    int K = 0, N = 0;              // padded input / output channels
    bool bn = false;
    size_t ow = 0, ob = 0, og = 0, obe = 0;     // arena offsets: weight [3][N][K], bias [N], gamma [N], beta [N]
    size_t os = 0;                 // statistics arena: running_mean [N], running_var [N]
    float *Y = nullptr, *out = nullptr, *mean = nullptr, *invstd = nullptr;
};
struct TrainLinear {
    int K = 0, N = 0;
    size_t ow = 0, ob = 0;
};

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
    float *gA = nullptr, *gB = nullptr, *dmulv = nullptr, *dz = nullptr, *adj = nullptr, *slab = nullptr;
    double* red = nullptr;         // [8]: recon sum, kld sum, ...
    size_t slab_elems = 0, adj_elems = 0;
    long step = 0;
    std::vector<void*> allocs;
};

namespace gem {

// ---- BatchNorm1d (training mode) + LeakyReLU over [rows, N]: one workgroup per 16 channels, 16 row groups -------------------
__global__ __launch_bounds__(256) void bn_train_fwd_kernel(const float* __restrict__ Y, int rows, int N, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, float* __restrict__ rmean, float* __restrict__ rvar,
                                                           float* __restrict__ mean_out, float* __restrict__ invstd_out, float* __restrict__ out,
                                                           float momentum, float eps) {
    __shared__ double sh[2][16][17];
    const int c = blockIdx.x * 16 + (threadIdx.x & 15), g = threadIdx.x >> 4;
    double s = 0.0, q = 0.0;
    for (int r = g; r < rows; r += 16) { const double v = Y[(size_t)r * N + c]; s += v; q += v * v; }
    sh[0][g][threadIdx.x & 15] = s; sh[1][g][threadIdx.x & 15] = q;
    __syncthreads();
    s = 0.0; q = 0.0;
    for (int i = 0; i < 16; ++i) { s += sh[0][i][threadIdx.x & 15]; q += sh[1][i][threadIdx.x & 15]; }
    const double mean = s / rows;
    double var = q / rows - mean * mean;                 // biased (what normalises the batch)
    if (var < 0.0) var = 0.0;
    const float invstd = (float)(1.0 / sqrt(var + (double)eps)), mf = (float)mean;
    const float ga = gamma[c], be = beta[c];
    for (int r = g; r < rows; r += 16) {
        const float v = ga * ((Y[(size_t)r * N + c] - mf) * invstd) + be;
        out[(size_t)r * N + c] = v > 0.f ? v : v * LEAKY_SLOPE;
    }
    if (g == 0) {
        mean_out[c] = mf; invstd_out[c] = invstd;
        const double unbiased = rows > 1 ? var * rows / (rows - 1) : var;
        rmean[c] = (1.f - momentum) * rmean[c] + momentum * mf;
        rvar[c] = (1.f - momentum) * rvar[c] + momentum * (float)unbiased;
    }
}

// dOut (w.r.t. the block's output) -> dY (w.r.t. the conv output), dgamma, dbeta.  LeakyReLU' from the sign of the output.
__global__ __launch_bounds__(256) void bn_train_bwd_kernel(const float* __restrict__ dOut, const float* __restrict__ out, const float* __restrict__ Y,
                                                           int rows, int N, const float* __restrict__ gamma, const float* __restrict__ mean,
                                                           const float* __restrict__ invstd, float* __restrict__ dY, float* __restrict__ dgamma,
                                                           float* __restrict__ dbeta) {
    __shared__ double sh[2][16][17];
    const int c = blockIdx.x * 16 + (threadIdx.x & 15), g = threadIdx.x >> 4;
    const float mf = mean[c], is = invstd[c], ga = gamma[c];
    double sb = 0.0, sg = 0.0;
    for (int r = g; r < rows; r += 16) {
        const size_t i = (size_t)r * N + c;
        const float dzv = dOut[i] * (out[i] > 0.f ? 1.f : LEAKY_SLOPE);
        sb += dzv; sg += (double)dzv * ((Y[i] - mf) * is);
    }
    sh[0][g][threadIdx.x & 15] = sb; sh[1][g][threadIdx.x & 15] = sg;
    __syncthreads();
    sb = 0.0; sg = 0.0;
    for (int i = 0; i < 16; ++i) { sb += sh[0][i][threadIdx.x & 15]; sg += sh[1][i][threadIdx.x & 15]; }
    const float mb = (float)(sb / rows), mg = (float)(sg / rows);
    for (int r = g; r < rows; r += 16) {
        const size_t i = (size_t)r * N + c;
        const float dzv = dOut[i] * (out[i] > 0.f ? 1.f : LEAKY_SLOPE);
        const float xh = (Y[i] - mf) * is;
        dY[i] = ga * is * (dzv - mb - xh * mg);
    }
    if (g == 0) { dgamma[c] = (float)sg; dbeta[c] = (float)sb; }
}

__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ dC, int rows, int N, float* __restrict__ out) {
    __shared__ double sh[16][17];
    const int c = blockIdx.x * 16 + (threadIdx.x & 15), g = threadIdx.x >> 4;
    double s = 0.0;
    for (int r = g; r < rows; r += 16) s += dC[(size_t)r * N + c];
    sh[g][threadIdx.x & 15] = s;
    __syncthreads();
    if (g == 0) {
        s = 0.0;
        for (int i = 0; i < 16; ++i) s += sh[i][threadIdx.x & 15];
        out[c] = (float)s;
    }
}

// ---- weight gradient: dW[tap][n][k] = sum_r dC[r][n] * A[r + tap - 1][k], slab z = rows [z * rps, (z + 1) * rps) --------------
template <int TAPS>
__global__ __launch_bounds__(256) void gemm_tn_kernel(const float* __restrict__ dC, int ldc, const float* __restrict__ A, int lda,
                                                      float* __restrict__ slab, int rows, int N, int K, int T, int rps) {
    __shared__ __attribute__((aligned(16))) float Cs[32][68];
    __shared__ __attribute__((aligned(16))) float As[32][68];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nt = blockIdx.x / (K / 64), kt = blockIdx.x - nt * (K / 64);
    const int tap = blockIdx.y, z = blockIdx.z;
    const int n0 = nt * 64, k0 = kt * 64;
    const int wm = wave >> 1, wn = wave & 1;
    const int fi = lane & 15, fq = lane >> 4;
    f32x4 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int r_begin = z * rps, r_end = min(rows, r_begin + rps);
    for (int r0 = r_begin; r0 < r_end; r0 += 32) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int i = tid + u * 256, rl = i >> 4, c4 = (i & 15) * 4;
            const int row = r0 + rl;
            f32x4 vc = {0.f, 0.f, 0.f, 0.f}, va = {0.f, 0.f, 0.f, 0.f};
            if (row < r_end) {
                vc = *reinterpret_cast<const f32x4*>(dC + (size_t)row * ldc + n0 + c4);
                int src = row;
                bool ok = true;
                if (TAPS == 3) { const int tt = row % T + tap - 1; ok = tt >= 0 && tt < T; src = row + tap - 1; }
                if (ok) va = *reinterpret_cast<const f32x4*>(A + (size_t)src * lda + k0 + c4);
            }
            *reinterpret_cast<f32x4*>(&Cs[rl][c4]) = vc;
            *reinterpret_cast<f32x4*>(&As[rl][c4]) = va;
        }
        __syncthreads();
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
__global__ void slab_sum_kernel(const float* __restrict__ slab, int nslab, size_t n, float* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float s = slab[i];
    for (int zz = 1; zz < nslab; ++zz) s += slab[(size_t)zz * n + i];
    out[i] = s;
}

// adjoint images for the backward-data products: conv [3][N][K] -> [3][K][N] with flipped taps; linear [N][K] -> [K][N]
__global__ void adjoint_kernel(const float* __restrict__ W, float* __restrict__ out, int taps, int N, int K) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)taps * N * K) return;
    const int tap = (int)(i / ((size_t)N * K));
    const size_t r = i - (size_t)tap * N * K;
    const int kin = (int)(r / N), o = (int)(r - (size_t)kin * N);      // out[tap][kin][o]
    out[i] = W[((size_t)(taps - 1 - tap) * N + o) * K + kin];
}

// reconstruction loss and its gradient: X, pose-packed [rows, 64]; columns >= C carry nothing
__global__ __launch_bounds__(1024) void recon_loss_kernel(const float* __restrict__ Xp, const float* __restrict__ pose_p, int rows, int C, float scale,
                                                          float* __restrict__ dXp, double* __restrict__ red) {
    __shared__ double sh[16];
    double s = 0.0;
    for (size_t i = threadIdx.x; i < (size_t)rows * PAD; i += 1024) {
        const int c = (int)(i % PAD);
        float d = 0.f;
        if (c < C) { d = Xp[i] - pose_p[i]; s += (double)d * d; }
        dXp[i] = 2.f * scale * d;
    }
    s = wave_sum_dpp(s);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) { double t = 0.0; for (int i = 0; i < 16; ++i) t += sh[i]; red[0] = t; }
}

// latent: KL term and the gradient w.r.t. [mu | logvar] from dz (decoder side) + the KL term (SeqConvVAE.py:159-169, 206-213)
__global__ __launch_bounds__(1024) void latent_bwd_kernel(const float* __restrict__ mulv, const float* __restrict__ eps, const float* __restrict__ dz,
                                                          int B, int D, int Dp, float kw_over_B, float* __restrict__ dmulv, double* __restrict__ red) {
    __shared__ double sh[16];
    double s = 0.0;
    for (size_t i = threadIdx.x; i < (size_t)B * Dp; i += 1024) {
        const int b = (int)(i / Dp), d = (int)(i - (size_t)b * Dp);
        float dmu = 0.f, dlv = 0.f;
        if (d < D) {
            const float mu = mulv[(size_t)b * 2 * Dp + d], lv = mulv[(size_t)b * 2 * Dp + Dp + d];
            const float ev = expf(lv), g = dz[i];
            s += (double)(1.f + lv - mu * mu - ev);
            dmu = g + kw_over_B * mu;
            dlv = g * eps[(size_t)b * D + d] * 0.5f * expf(0.5f * lv) + kw_over_B * 0.5f * (ev - 1.f);
        }
        dmulv[(size_t)b * 2 * Dp + d] = dmu;
        dmulv[(size_t)b * 2 * Dp + Dp + d] = dlv;
    }
    s = wave_sum_dpp(s);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) { double t = 0.0; for (int i = 0; i < 16; ++i) t += sh[i]; red[1] = -0.5 * t / B; }
}
__global__ void finish_loss_kernel(double* red, double n_recon, double kld_weight, double* out) {
    const double recon = red[0] / n_recon;
    out[0] = recon + kld_weight * red[1]; out[1] = recon; out[2] = red[1];
}

// torch.optim.Adam (amsgrad off): g += wd * p; m, v moments; p -= lr / bc1 * m / (sqrt(v) / sqrt(bc2) + eps)
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, size_t n, float lr,
                            float b1, float b2, float eps, float wd, float bc1, float bc2_sqrt) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float gg = g[i] + wd * p[i];
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

static Layer conv_layer(gem_trainer* t, const TrainConv& c) {
    Layer L; L.taps = 3; L.K = c.K; L.N = c.N; L.w = t->P + c.ow; L.bias = t->P + c.ob; return L;
}

// weight gradient of a layer (TAPS taps) into G at `og`: slabs over the row range, summed in order
template <int TAPS>
static int weight_grad(gem_trainer* t, const float* dC, int ldc, const float* A, int lda, int rows, int N, int K, size_t og, hipStream_t s) {
    const int rps = 256, nslab = (rows + rps - 1) / rps;
    const size_t n = (size_t)TAPS * N * K;
    if (nslab > 1 && n * nslab > t->slab_elems) { set_error("train: weight-gradient scratch too small"); return 1; }
    hipLaunchKernelGGL(gemm_tn_kernel<TAPS>, dim3((N / 64) * (K / 64), TAPS, nslab), dim3(256), 0, s, dC, ldc, A, lda,
                       nslab > 1 ? t->slab : t->G + og, rows, N, K, t->T, rps);
    if (nslab > 1)
        hipLaunchKernelGGL(slab_sum_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (const float*)t->slab, nslab, n, t->G + og);
    GEM_HIP(hipGetLastError());
    return 0;
}
static int adjoint(gem_trainer* t, const float* W, int taps, int N, int K, hipStream_t s) {
    const size_t n = (size_t)taps * N * K;
    if (n > t->adj_elems) { set_error("train: adjoint scratch too small"); return 1; }
    hipLaunchKernelGGL(adjoint_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, W, t->adj, taps, N, K);
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
    size_t max_w = 0, max_width = PAD;
    for (auto* v : {&p->enc, &p->dec})
        for (auto& c : *v) {
            if (talloc(p, &c.Y, rows * c.N) || talloc(p, &c.out, rows * c.N) || talloc(p, &c.mean, (size_t)c.N) || talloc(p, &c.invstd, (size_t)c.N)) return 1;
            max_w = std::max(max_w, (size_t)3 * c.N * c.K);
            max_width = std::max(max_width, (size_t)std::max(c.N, c.K));
        }
    max_w = std::max(max_w, std::max((size_t)p->fc.N * p->fc.K, (size_t)p->dec_in.N * p->dec_in.K));
    p->adj_elems = max_w;
    // weight-gradient slabs of 256 rows (a single slab goes straight to the gradient arena): the conv layers contract over B*T
    // rows of small tensors, the linear layers over B rows of large ones
    size_t conv_w = 0;
    for (auto* v : {&p->enc, &p->dec}) for (auto& c : *v) conv_w = std::max(conv_w, (size_t)3 * c.N * c.K);
    const size_t conv_slabs = (rows + 255) / 256, lin_slabs = ((size_t)p->Bmax + 255) / 256;
    p->slab_elems = std::max(conv_slabs > 1 ? conv_w * conv_slabs : 0,
                             lin_slabs > 1 ? std::max((size_t)p->fc.N * p->fc.K, (size_t)p->dec_in.N * p->dec_in.K) * lin_slabs : 0);
    if (talloc(p, &p->pose_p, rows * PAD) || talloc(p, &p->mulv, (size_t)p->Bmax * 2 * p->Dp) || talloc(p, &p->z, (size_t)p->Bmax * p->Dp) ||
        talloc(p, &p->h0, rows * p->topp) || talloc(p, &p->Xp, rows * PAD) || talloc(p, &p->gA, rows * max_width) || talloc(p, &p->gB, rows * max_width) ||
        talloc(p, &p->dmulv, (size_t)p->Bmax * 2 * p->Dp) || talloc(p, &p->dz, (size_t)p->Bmax * p->Dp) || talloc(p, &p->adj, p->adj_elems) ||
        talloc(p, &p->slab, p->slab_elems) || talloc(p, &p->red, (size_t)8))
        return 1;
    *out = t.release();
    return 0;
}

void gem_trainer_destroy(gem_trainer* t) {
    if (!t) return;
    if (t->h) { (void)hipSetDevice(t->h->cfg.device); (void)hipDeviceSynchronize(); }
    for (void* p : t->allocs) (void)hipFree(p);
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
    return 0;
}
int gem_trainer_set_step(gem_trainer* t, int64_t step) {
    if (!t || step < 0) { set_error("gem_trainer_set_step: bad argument"); return 1; }
    t->step = (long)step;
    return 0;
}

int gem_trainer_step(gem_trainer* t, int B, const float* d_pose, const float* d_eps, const gem_train_opts* o, int update, double* d_losses,
                     void* stream) {
    if (!t || !d_pose || !d_eps || !o) { set_error("gem_trainer_step: null argument"); return 1; }
    if (B < 2 || B > t->Bmax) { set_error("gem_trainer_step: need 2 <= B <= max_windows (BatchNorm statistics)"); return 1; }
    gem_handle* h = t->h;
    GEM_HIP(hipSetDevice(h->cfg.device));
    hipStream_t s = (hipStream_t)stream;
    const int T = t->T, rows = B * T;
    h->precision = GEM_PRECISION_F32;
    // ---- forward (train mode)
    if (launch_pack_pose(d_pose, t->pose_p, rows, t->C, s)) return 1;
    auto conv_fwd = [&](TrainConv& c, const float* in) -> int {
        if (launch_gemm(h, conv_layer(t, c), EPI_BIAS, in, c.K, nullptr, c.bn ? c.Y : c.out, c.N, rows, T, s, -1)) return 1;
        if (c.bn) {
            hipLaunchKernelGGL(bn_train_fwd_kernel, dim3(c.N / 16), dim3(256), 0, s, (const float*)c.Y, rows, c.N, (const float*)(t->P + c.og),
                               (const float*)(t->P + c.obe), t->S + c.os, t->S + c.os + c.N, c.mean, c.invstd, c.out, (float)o->bn_momentum, (float)BN_EPS);
            GEM_HIP(hipGetLastError());
        }
        return 0;
    };
    const float* in = t->pose_p;
    for (auto& c : t->enc) { if (conv_fwd(c, in)) return 1; in = c.out; }
    { Layer L; L.taps = 1; L.K = t->fc.K; L.N = t->fc.N; L.w = t->P + t->fc.ow; L.bias = t->P + t->fc.ob;
      if (launch_gemm(h, L, EPI_BIAS, in, L.K, nullptr, t->mulv, L.N, B, T, s, -1)) return 1; }
    if (launch_reparam(t->mulv, d_eps, nullptr, nullptr, nullptr, t->z, B, t->D, t->Dp, s)) return 1;
    { Layer L; L.taps = 1; L.K = t->dec_in.K; L.N = t->dec_in.N; L.w = t->P + t->dec_in.ow; L.bias = t->P + t->dec_in.ob;
      if (launch_gemm(h, L, EPI_BIAS, t->z, L.K, nullptr, t->h0, L.N, B, T, s, -1)) return 1; }
    in = t->h0;
    for (auto& c : t->dec) { if (conv_fwd(c, in)) return 1; in = c.out; }
    const float* X = t->dec.back().out;
    // ---- loss + its gradient w.r.t. the decoded pose
    const double n_recon = o->recon_sum ? 1.0 : (double)rows * t->C;
    hipLaunchKernelGGL(recon_loss_kernel, dim3(1), dim3(1024), 0, s, X, (const float*)t->pose_p, rows, t->C, (float)(1.0 / n_recon), t->gA, t->red);
    GEM_HIP(hipGetLastError());
    // ---- backward: decoder
    float *g = t->gA, *g2 = t->gB;
    for (int i = (int)t->dec.size() - 1; i >= 0; --i) {
        TrainConv& c = t->dec[i];
        const float* a_in = i > 0 ? t->dec[i - 1].out : t->h0;
        const float* dY = g;
        if (c.bn) {
            hipLaunchKernelGGL(bn_train_bwd_kernel, dim3(c.N / 16), dim3(256), 0, s, (const float*)g, (const float*)c.out, (const float*)c.Y, rows, c.N,
                               (const float*)(t->P + c.og), (const float*)c.mean, (const float*)c.invstd, g2, t->G + c.og, t->G + c.obe);
            GEM_HIP(hipGetLastError());
            dY = g2;
        }
        hipLaunchKernelGGL(colsum_kernel, dim3(c.N / 16), dim3(256), 0, s, dY, rows, c.N, t->G + c.ob);
        if (weight_grad<3>(t, dY, c.N, a_in, c.K, rows, c.N, c.K, c.ow, s)) return 1;
        if (adjoint(t, t->P + c.ow, 3, c.N, c.K, s)) return 1;
        Layer L; L.taps = 3; L.K = c.N; L.N = c.K; L.w = t->adj; L.bias = nullptr;
        float* dA = (dY == g) ? g2 : g;          // the buffer that does not hold dY
        if (launch_gemm(h, L, EPI_NONE, dY, c.N, nullptr, dA, c.K, rows, T, s, -1)) return 1;
        if (dA != g) { float* tmp = g; g = dA; g2 = tmp; }
    }
    // g = dh0 [B, T*topp]: decoder_input
    { const TrainLinear& l = t->dec_in;
      hipLaunchKernelGGL(colsum_kernel, dim3(l.N / 16), dim3(256), 0, s, (const float*)g, B, l.N, t->G + l.ob);
      if (weight_grad<1>(t, g, l.N, t->z, l.K, B, l.N, l.K, l.ow, s)) return 1;
      if (adjoint(t, t->P + l.ow, 1, l.N, l.K, s)) return 1;
      Layer L; L.taps = 1; L.K = l.N; L.N = l.K; L.w = t->adj; L.bias = nullptr;
      if (launch_gemm(h, L, EPI_NONE, g, l.N, nullptr, t->dz, l.K, B, T, s, -1)) return 1; }
    hipLaunchKernelGGL(latent_bwd_kernel, dim3(1), dim3(1024), 0, s, (const float*)t->mulv, d_eps, (const float*)t->dz, B, t->D, t->Dp,
                       (float)(o->kld_weight / B), t->dmulv, t->red);
    hipLaunchKernelGGL(finish_loss_kernel, dim3(1), dim3(1), 0, s, t->red, n_recon, o->kld_weight, t->red + 4);
    GEM_HIP(hipGetLastError());
    // fc_mu | fc_var
    { const TrainLinear& l = t->fc;
      const float* flat = t->enc.back().out;
      hipLaunchKernelGGL(colsum_kernel, dim3(l.N / 16), dim3(256), 0, s, (const float*)t->dmulv, B, l.N, t->G + l.ob);
      if (weight_grad<1>(t, t->dmulv, l.N, flat, l.K, B, l.N, l.K, l.ow, s)) return 1;
      if (adjoint(t, t->P + l.ow, 1, l.N, l.K, s)) return 1;
      Layer L; L.taps = 1; L.K = l.N; L.N = l.K; L.w = t->adj; L.bias = nullptr;
      g = t->gA; g2 = t->gB;
      if (launch_gemm(h, L, EPI_NONE, t->dmulv, l.N, nullptr, g, l.K, B, T, s, -1)) return 1; }
    // encoder
    for (int i = (int)t->enc.size() - 1; i >= 0; --i) {
        TrainConv& c = t->enc[i];
        const float* a_in = i > 0 ? t->enc[i - 1].out : t->pose_p;
        hipLaunchKernelGGL(bn_train_bwd_kernel, dim3(c.N / 16), dim3(256), 0, s, (const float*)g, (const float*)c.out, (const float*)c.Y, rows, c.N,
                           (const float*)(t->P + c.og), (const float*)c.mean, (const float*)c.invstd, g2, t->G + c.og, t->G + c.obe);
        hipLaunchKernelGGL(colsum_kernel, dim3(c.N / 16), dim3(256), 0, s, (const float*)g2, rows, c.N, t->G + c.ob);
        GEM_HIP(hipGetLastError());
        if (weight_grad<3>(t, g2, c.N, a_in, c.K, rows, c.N, c.K, c.ow, s)) return 1;
        if (i > 0) {
            if (adjoint(t, t->P + c.ow, 3, c.N, c.K, s)) return 1;
            Layer L; L.taps = 3; L.K = c.N; L.N = c.K; L.w = t->adj; L.bias = nullptr;
            if (launch_gemm(h, L, EPI_NONE, g2, c.N, nullptr, g, c.K, rows, T, s, -1)) return 1;
        }
    }
    if (d_losses) GEM_HIP(hipMemcpyAsync(d_losses, t->red + 4, 3 * sizeof(double), hipMemcpyDeviceToDevice, s));
    // ---- Adam
    if (update) {
        ++t->step;
        const double bc1 = 1.0 - std::pow(o->beta1, (double)t->step), bc2 = 1.0 - std::pow(o->beta2, (double)t->step);
        hipLaunchKernelGGL(adam_kernel, dim3((unsigned)((t->n_params + 255) / 256)), dim3(256), 0, s, t->P, (const float*)t->G, t->M1, t->M2, t->n_params,
                           (float)o->lr, (float)o->beta1, (float)o->beta2, (float)o->eps, (float)o->weight_decay, (float)bc1, (float)std::sqrt(bc2));
        GEM_HIP(hipGetLastError());
    }
    return 0;
}

}  // extern "C"
