// Sequence post-processing on the device (gfx950): overlap averaging + Gaussian smoothing of the optimised
// windows, and the reference's 18-entry error report, all in float64 like the numpy code they replace.
//
// Reference: optimizer.py:425-450 (merge_batches, gaussian_filter1d(sigma=1)), calculate_errors.py:8-83,114-179
// (global / sequence-aligned / per-frame Procrustes / bone-length-normalised MPJPE, hip-midpoint error),
// utils/rigid_transform_with_scale.py:18-43 (Umeyama with the reflection fix), utils/skeleton.py:124-136
// (skeleton re-growth with fixed bone lengths).
//
// All of it is latency-bound bookkeeping (a 2000-frame sequence is 4 x 720 KB): the point is that the numbers
// come out of the same stream as the optimisation, without the host loops of the reference (one LAPACK SVD per
// frame and alignment: ~1 s of numpy per 2000 frames).  Reductions are fixed-order trees: results are
// bitwise reproducible.
#include "gem_internal.h"

namespace gem {

// ---------------------------------------------------------------------------------------------------
// 3x3 SVD by one-sided Jacobi (Hestenes): A = U diag(S) V^T, columns of U/V orthonormal, S >= 0 unsorted.
// Accurate to eps * cond(A) (no A^T A squaring).  Row-major 3x3 arrays.
__device__ inline void svd3(const double* A, double* U, double* S, double* V) {
    double a[9];
    for (int i = 0; i < 9; ++i) { a[i] = A[i]; V[i] = (i % 4 == 0) ? 1.0 : 0.0; }
    for (int sweep = 0; sweep < 30; ++sweep) {
        double off = 0.0;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                const double alpha = a[p] * a[p] + a[3 + p] * a[3 + p] + a[6 + p] * a[6 + p];
                const double beta = a[q] * a[q] + a[3 + q] * a[3 + q] + a[6 + q] * a[6 + q];
                const double gamma = a[p] * a[q] + a[3 + p] * a[3 + q] + a[6 + p] * a[6 + q];
                if (gamma == 0.0 || fabs(gamma) <= 1e-300) continue;
                const double rel = fabs(gamma) / sqrt(alpha * beta);
                off = rel > off ? rel : off;
                if (!(rel > 1e-17)) continue;
                const double zeta = (beta - alpha) / (2.0 * gamma);
                const double t = (zeta >= 0.0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                const double c = 1.0 / sqrt(1.0 + t * t), s = c * t;
                for (int r = 0; r < 3; ++r) {
                    const double ap = a[3 * r + p], aq = a[3 * r + q];
                    a[3 * r + p] = c * ap - s * aq;
                    a[3 * r + q] = s * ap + c * aq;
                    const double vp = V[3 * r + p], vq = V[3 * r + q];
                    V[3 * r + p] = c * vp - s * vq;
                    V[3 * r + q] = s * vp + c * vq;
                }
            }
        if (off < 1e-16) break;
    }
    double smax = 0.0;
    for (int k = 0; k < 3; ++k) {
        S[k] = sqrt(a[k] * a[k] + a[3 + k] * a[3 + k] + a[6 + k] * a[6 + k]);
        smax = S[k] > smax ? S[k] : smax;
    }
    int bad = -1;
    for (int k = 0; k < 3; ++k) {
        if (S[k] > 1e-14 * smax && S[k] > 0.0) {
            for (int r = 0; r < 3; ++r) U[3 * r + k] = a[3 * r + k] / S[k];
        } else {
            bad = k;
        }
    }
    if (bad >= 0) {          // rank-deficient (coplanar points): complete U with the cross product of the other two
        const int i = (bad + 1) % 3, j = (bad + 2) % 3;
        U[bad] = U[3 + i] * U[6 + j] - U[6 + i] * U[3 + j];
        U[3 + bad] = U[6 + i] * U[j] - U[i] * U[6 + j];
        U[6 + bad] = U[i] * U[3 + j] - U[3 + i] * U[j];
        S[bad] = 0.0;
    }
}

__device__ inline double det3(const double* m) {
    return m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) + m[2] * (m[3] * m[7] - m[4] * m[6]);
}

struct Sim3 {
    double cR[9];      // c * R, row-major: aligned = p @ cR + t  (row vector convention of the reference)
    double t[3];
};

// rigid_transform_3D from moments: cov = (P-mp)^T (Q-mq) / n, var = sum_d var(P_d).
__device__ inline void umeyama_moments(const double* mp, const double* mq, const double* cov, double var, Sim3* out) {
    double U[9], S[3], V[9];
    svd3(cov, U, S, V);
    // numpy: cov = Vn diag(S) Wn, R = Vn @ Wn with (S[-1], Vn[:, -1]) negated when det(Vn) det(Wn) < 0; here
    // Vn = U, Wn = V^T, and "last" = the smallest singular value.
    int kmin = 0;
    for (int k = 1; k < 3; ++k)
        if (S[k] < S[kmin]) kmin = k;
    double d[3] = {1.0, 1.0, 1.0};
    if (det3(U) * det3(V) < 0.0) d[kmin] = -1.0;
    const double c = (d[0] * S[0] + d[1] * S[1] + d[2] * S[2]) / var;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double r = 0.0;
            for (int k = 0; k < 3; ++k) r += U[3 * i + k] * d[k] * V[3 * j + k];
            out->cR[3 * i + j] = c * r;
        }
    for (int j = 0; j < 3; ++j) out->t[j] = mq[j] - (mp[0] * out->cR[j] + mp[1] * out->cR[3 + j] + mp[2] * out->cR[6 + j]);
}

// ---------------------------------------------------------------------------------------------------
// Per-frame metrics: six threads per frame (one per source and variant), each with its three J x 3 working sets
// in LDS columns ([item][thread], conflict-free).  Output row layout (11 + J doubles per frame, stored [col][frame]):
//   0-2   sum_j |src - gt|            src = est, mid, opt
//   3-4   |hip midpoint src - gt|      src = est, opt
//   5-7   sum_j |procrustes(src) - gt|
//   8-10  sum_j |procrustes(resized src) - resized^k gt|, k = 1, 2, 3   (the reference re-normalises gt per call)
//   11..  per-joint error of the last one
constexpr int ERR_FT = 64;                 // frames (threads) per workgroup

struct ErrArgs {
    const double* src[3];
    const double* gt;
    double* frame_out;      // [11 + MAXJ_ERR][F]
    double* out;            // [17 + J]
    int F, J;
    int parents[MAXJ_ERR];
    double bone_mm[MAXJ_ERR];
    // blockIdx.y = the sequence of a batch of equally long ones laid end to end: strides of src / gt, frame_out and out between them
    size_t seq_stride, frame_stride, out_stride;
};

__device__ inline void errors_select_sequence(ErrArgs& a, int c) {
    for (int s = 0; s < 3; ++s) a.src[s] += c * a.seq_stride;
    a.gt += c * a.seq_stride;
    a.frame_out += c * a.frame_stride;
    a.out += c * a.out_stride;
}

#define LD3(buf, j, d) buf[((j) * 3 + (d)) * ERR_FT + tx]

__device__ inline void frame_umeyama(const double* P, const double* Q, int J, int tx, Sim3* sim) {
    double mp[3] = {0, 0, 0}, mq[3] = {0, 0, 0};
    for (int j = 0; j < J; ++j)
        for (int d = 0; d < 3; ++d) { mp[d] += LD3(P, j, d); mq[d] += LD3(Q, j, d); }
    for (int d = 0; d < 3; ++d) { mp[d] /= J; mq[d] /= J; }
    double cov[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, var = 0.0;
    for (int j = 0; j < J; ++j) {
        double p[3], q[3];
        for (int d = 0; d < 3; ++d) { p[d] = LD3(P, j, d) - mp[d]; q[d] = LD3(Q, j, d) - mq[d]; }
        for (int a = 0; a < 3; ++a) {
            var += p[a] * p[a];
            for (int b = 0; b < 3; ++b) cov[3 * a + b] += p[a] * q[b];
        }
    }
    for (int i = 0; i < 9; ++i) cov[i] /= J;
    umeyama_moments(mp, mq, cov, var / J, sim);
}

__device__ inline void apply_sim(const Sim3& s, const double* p, double* o) {
    for (int j = 0; j < 3; ++j) o[j] = p[0] * s.cR[j] + p[1] * s.cR[3 + j] + p[2] * s.cR[6 + j] + s.t[j];
}

// skeleton.py:124-136: bone vectors from the ORIGINAL joints, rescaled to bone_mm/1000, then re-grown in index order.
__device__ inline void resize_skeleton(double* X, double* W, const ErrArgs& a, int tx) {
    const int J = a.J;
    for (int j = 0; j < J; ++j) {
        double v[3], n2 = 0.0;
        for (int d = 0; d < 3; ++d) { v[d] = LD3(X, j, d) - LD3(X, a.parents[j], d); n2 += v[d] * v[d]; }
        const double sc = j == 0 ? 0.0 : a.bone_mm[j] / sqrt(n2);
        for (int d = 0; d < 3; ++d) LD3(W, j, d) = v[d] * sc / 1000.0;
    }
    for (int j = 0; j < J; ++j)
        for (int d = 0; d < 3; ++d) LD3(X, j, d) = LD3(X, a.parents[j], d) + LD3(W, j, d);
}

__global__ __launch_bounds__(ERR_FT) void errors_frame_kernel(ErrArgs a) {
    extern __shared__ double lds_d[];
    errors_select_sequence(a, blockIdx.y);
    // six independent tasks per frame: (source s = est / mid / opt) x (plain + Procrustes | bone-length normalised)
    const int tx = threadIdx.x, J = a.J;
    const long task = (long)blockIdx.x * ERR_FT + tx;
    const int f = (int)(task / 6), k = (int)(task % 6), s = k % 3;
    double* G = lds_d;
    double* C = G + J * 3 * ERR_FT;
    double* W = C + J * 3 * ERR_FT;
    if (f >= a.F) return;                    // (no barriers in this kernel: every thread works on its own columns)
    const size_t base = (size_t)f * J * 3;
    auto put = [&](int col, double v) { a.frame_out[(size_t)col * a.F + f] = v; };
    for (int i = 0; i < J * 3; ++i) {
        G[i * ERR_FT + tx] = a.gt[base + i];
        C[i * ERR_FT + tx] = a.src[s][base + i];
    }
    if (k < 3) {
        double e = 0.0;
        for (int j = 0; j < J; ++j) {
            double n2 = 0.0;
            for (int d = 0; d < 3; ++d) { const double x = LD3(C, j, d) - LD3(G, j, d); n2 += x * x; }
            e += sqrt(n2);
        }
        put(s, e);
        if (s != 1) {                        // hip midpoint = (joint 7 + joint 11) / 2 (calculate_errors.py:33-47)
            double n2 = 0.0;
            for (int d = 0; d < 3; ++d) {
                const double x = (LD3(C, 7, d) + LD3(C, 11, d)) / 2 - (LD3(G, 7, d) + LD3(G, 11, d)) / 2;
                n2 += x * x;
            }
            put(s == 0 ? 3 : 4, sqrt(n2));
        }
    } else {
        // the reference re-normalises the (already normalised) gt on every call: s + 1 re-growths for source s
        for (int r = 0; r <= s; ++r) resize_skeleton(G, W, a, tx);
        resize_skeleton(C, W, a, tx);
    }
    Sim3 sim;
    frame_umeyama(C, G, J, tx, &sim);
    double e = 0.0;
    for (int j = 0; j < J; ++j) {
        double p[3] = {LD3(C, j, 0), LD3(C, j, 1), LD3(C, j, 2)}, o[3], n2 = 0.0;
        apply_sim(sim, p, o);
        for (int d = 0; d < 3; ++d) { const double x = o[d] - LD3(G, j, d); n2 += x * x; }
        const double en = sqrt(n2);
        e += en;
        if (k == 5) put(11 + j, en);
    }
    put(k < 3 ? 5 + s : 8 + s, e);
}

// ---------------------------------------------------------------------------------------------------
// Sequence-level part: three workgroups, one per source sequence (est, mid, opt).  Each does its similarity
// alignment over all F*J points (means, centred moments, SVD by thread 0, aligned errors) with ONE barrier per
// group of sums, and a third of the column sums of the per-frame table.
constexpr int ERR_ST = 1024;
constexpr int ERR_NW = ERR_ST / 64;

// K sums at once: wavefront DPP reductions, [K][16] partials in LDS, every thread adds the 16 partials of each value
// in the same order.  Alternating LDS halves: one barrier per call is enough (see lbfgs.hip's BlockRed).
template <int K>
__device__ inline void block_sums(double (&v)[K], double* lds, int& parity) {
    double* r = lds + parity * (16 * ERR_NW);
    parity ^= 1;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const double w = wave_sum_dpp(v[k]);
        if ((threadIdx.x & 63) == 0) r[k * ERR_NW + (threadIdx.x >> 6)] = w;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < K; ++k) {
        double t = 0.0;
        for (int i = 0; i < ERR_NW; ++i) t += r[k * ERR_NW + i];
        v[k] = t;
    }
}

__global__ __launch_bounds__(ERR_ST) void errors_sequence_kernel(ErrArgs a) {
    __shared__ double red[2 * 16 * ERR_NW];
    __shared__ Sim3 sim_s;
    int parity = 0;
    errors_select_sequence(a, blockIdx.y);
    const int tid = threadIdx.x, J = a.J, F = a.F, s = blockIdx.x;
    const size_t N = (size_t)F * J;
    double* out = a.out;
    {
        const double* P = a.src[s];
        const double* Q = a.gt;
        double m[6] = {0, 0, 0, 0, 0, 0};
        for (size_t i = tid; i < N; i += ERR_ST)
            for (int d = 0; d < 3; ++d) { m[d] += P[i * 3 + d]; m[3 + d] += Q[i * 3 + d]; }
        block_sums<6>(m, red, parity);
        for (int k = 0; k < 6; ++k) m[k] /= (double)N;
        double c[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        for (size_t i = tid; i < N; i += ERR_ST) {
            double p[3], q[3];
            for (int d = 0; d < 3; ++d) { p[d] = P[i * 3 + d] - m[d]; q[d] = Q[i * 3 + d] - m[3 + d]; }
            for (int x = 0; x < 3; ++x) {
                c[9] += p[x] * p[x];
                for (int y = 0; y < 3; ++y) c[3 * x + y] += p[x] * q[y];
            }
        }
        block_sums<10>(c, red, parity);
        for (int k = 0; k < 10; ++k) c[k] /= (double)N;
        if (tid == 0) umeyama_moments(m, m + 3, c, c[9], &sim_s);
        __syncthreads();
        const Sim3 sim = sim_s;
        double e[2] = {0.0, 0.0};
        for (size_t i = tid; i < N; i += ERR_ST) {
            double p[3] = {P[i * 3], P[i * 3 + 1], P[i * 3 + 2]}, o[3], n2 = 0.0;
            apply_sim(sim, p, o);
            for (int d = 0; d < 3; ++d) { const double x = o[d] - Q[i * 3 + d]; n2 += x * x; }
            e[0] += sqrt(n2);
        }
        for (int f = tid; f < F; f += ERR_ST) {
            const double* p7 = P + ((size_t)f * J + 7) * 3;
            const double* p11 = P + ((size_t)f * J + 11) * 3;
            const double* q7 = Q + ((size_t)f * J + 7) * 3;
            const double* q11 = Q + ((size_t)f * J + 11) * 3;
            double a7[3], a11[3], n2 = 0.0;
            apply_sim(sim, p7, a7);
            apply_sim(sim, p11, a11);
            for (int d = 0; d < 3; ++d) { const double x = (a7[d] + a11[d]) / 2 - (q7[d] + q11[d]) / 2; n2 += x * x; }
            e[1] += sqrt(n2);
        }
        block_sums<2>(e, red, parity);
        if (tid == 0) { out[8 + s] = e[0] / (double)N; out[5 + s] = e[1] / (double)F; }
    }
    for (int col = s; col < 11 + J; col += 3) {
        double v[1] = {0.0};
        for (int f = tid; f < F; f += ERR_ST) v[0] += a.frame_out[(size_t)col * F + f];
        block_sums<1>(v, red, parity);
        if (tid == 0) {
            if (col < 3) out[col] = v[0] / (double)N;                       // *_global_mpjpe
            else if (col < 5) out[col] = v[0] / (double)F;                   // *_camera_pos_error
            else if (col < 8) out[11 + (col - 5)] = v[0] / (double)N;        // per-frame Procrustes
            else if (col < 11) out[14 + (col - 8)] = v[0] / (double)N;       // bone-length normalised
            else out[17 + (col - 11)] = v[0] / (double)F;                    // joints_error
        }
    }
}

size_t errors_frame_lds_bytes(int J) { return (size_t)3 * J * 3 * ERR_FT * sizeof(double); }

int launch_errors(gem_handle* h, const double* est, const double* mid, const double* opt, const double* gt, int F,
                  const double* bone_mm, double* frame_out, double* out, hipStream_t s, int n_seq) {
    ErrArgs a;
    a.src[0] = est; a.src[1] = mid; a.src[2] = opt; a.gt = gt;
    a.frame_out = frame_out; a.out = out; a.F = F; a.J = h->J;
    a.seq_stride = (size_t)F * h->J * 3; a.frame_stride = (size_t)(11 + MAXJ_ERR) * F; a.out_stride = 17 + h->J;
    for (int j = 0; j < MAXJ_ERR; ++j) {
        a.parents[j] = j < h->J ? h->cfg.parents[j] : 0;
        a.bone_mm[j] = j < h->J ? bone_mm[j] : 0.0;
    }
    const size_t lds = errors_frame_lds_bytes(h->J);
    static PerDeviceOnce attr_once;
    if (attr_once.need(h->cfg.device)) {
        GEM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(errors_frame_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)errors_frame_lds_bytes(MAXJ_ERR)));
    }
    hipLaunchKernelGGL(errors_frame_kernel, dim3((unsigned)(((long)F * 6 + ERR_FT - 1) / ERR_FT), (unsigned)n_seq), dim3(ERR_FT), lds, s, a);
    GEM_HIP(hipGetLastError());
    hipLaunchKernelGGL(errors_sequence_kernel, dim3(3, (unsigned)n_seq), dim3(ERR_ST), 0, s, a);
    GEM_HIP(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------------------------------
// merge_batches (optimizer.py:425-437) per chunk + gaussian_filter1d(sigma=1, mode='reflect', truncate=4)
// along the frames of the chunk (optimizer.py:448-450).  windows [n_chunks*wpc, T, JC] f64 ->
// merged [n_chunks*fpc, JC] f64 with fpc = wpc*(T-overlap) + overlap.
__global__ void merge_windows_kernel(const double* __restrict__ win, double* __restrict__ out, int n_chunks, int wpc, int T, int JC,
                                     int overlap) {
    const int stride = T - overlap, fpc = wpc * stride + overlap;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)n_chunks * fpc * JC) return;
    const int c = (int)(i % JC);
    const size_t fr = i / JC;
    const int chunk = (int)(fr / fpc), f = (int)(fr % fpc);
    int w = f / stride, t = f - w * stride;
    if (w >= wpc) { w = wpc - 1; t = f - w * stride; }                 // the last `overlap` frames of the chunk
    const double* base = win + (size_t)chunk * wpc * T * JC;
    double v = base[((size_t)w * T + t) * JC + c];
    if (t < overlap && w > 0) v = (base[((size_t)(w - 1) * T + t + stride) * JC + c] + v) / 2;
    out[i] = v;
}

__global__ void gauss_smooth_kernel(const double* __restrict__ in, double* __restrict__ out, int n_chunks, int fpc, int JC) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)n_chunks * fpc * JC) return;
    const int c = (int)(i % JC);
    const size_t fr = i / JC;
    const int chunk = (int)(fr / fpc), f = (int)(fr % fpc);
    // scipy: radius = int(truncate * sigma + 0.5) = 4, weights exp(-x^2 / 2) normalised, correlate1d, mode='reflect'
    double w[5], wsum = 0.0;
    for (int k = 0; k <= 4; ++k) w[k] = exp(-0.5 * k * k);
    wsum = w[0] + 2 * (w[1] + w[2] + w[3] + w[4]);
    const double* base = in + (size_t)chunk * fpc * JC;
    double acc = 0.0;
    for (int k = -4; k <= 4; ++k) {
        int g = f + k;
        // 'reflect' = (d c b a | a b c d | d c b a), period 2n
        const int period = 2 * fpc;
        g = ((g % period) + period) % period;
        if (g >= fpc) g = period - 1 - g;
        acc += w[k < 0 ? -k : k] / wsum * base[(size_t)g * JC + c];
    }
    out[i] = acc;
}

int launch_merge(const double* win, double* tmp, double* out, int n_chunks, int wpc, int T, int JC, int overlap, int smooth,
                 hipStream_t s) {
    const int fpc = wpc * (T - overlap) + overlap;
    const size_t n = (size_t)n_chunks * fpc * JC;
    if (n == 0) return 0;
    const dim3 grid((unsigned)((n + 255) / 256));
    hipLaunchKernelGGL(merge_windows_kernel, grid, dim3(256), 0, s, win, smooth ? tmp : out, n_chunks, wpc, T, JC, overlap);
    GEM_HIP(hipGetLastError());
    if (smooth) {
        hipLaunchKernelGGL(gauss_smooth_kernel, grid, dim3(256), 0, s, (const double*)tmp, out, n_chunks, fpc, JC);
        GEM_HIP(hipGetLastError());
    }
    return 0;
}

// ---------------------------------------------------------------------------------------------------
// Input lifting: network outputs -> estimated_local_skeleton.
//
// Reference: Skeleton.set_skeleton_from_file / set_skeleton / get_max_preds (utils/skeleton.py:74-90,32-45,176-204)
// and FishEyeCameraCalibrated.camera2world (utils/fisheye/FishEyeCalibrated.py:18-33).  The reference blows the
// 64x64 heat-map up to 1024x1024 with cv2.INTER_NEAREST, pads 128 zero columns left and right, and takes the
// row-major argmax of the 1280-wide image.  Every source texel becomes a constant 16x16 block, so the first
// maximum of the big image is the top-left pixel of the block of the first row-major maximum of the 64x64 map:
// (x, y) = (16 sx + 128, 16 sy); a non-positive maximum gives (0, 0) (the `maxvals > 0` mask, and the zero
// padding wins the argmax at index 0 anyway).  The 1.3 M-pixel intermediate is never built.
//
// HBM-bound: one pass over the heat-maps in their pickle layout [F,H,W,J] (245 KB per frame for 0.4 KB out).
// One 256-thread workgroup per frame; thread (g = tid>>4, j = tid&15) scans pixels g, g+16, ... of joint j: a
// wavefront's load covers 4 pixels x 15 joints = 240 contiguous bytes.
struct LiftArgs {
    const float* heat;
    const double* depth;
    double* out64;
    float* out32;
    int F, H, W, J, up, pad_x, pad_y, n_poly;
    double poly[GEM_MAX_POLY];     // polynomialC2W, ascending powers
    double cx, cy;
};

__device__ inline bool lift_better(float v, int i, float b, int bi) {
    // numpy argmax: NaN is maximal, first occurrence wins
    const bool vn = v != v, bn = b != b;
    if (vn || bn) return vn && (!bn || i < bi);
    return v > b || (v == b && i < bi);
}

__global__ __launch_bounds__(256) void lift_skeleton_kernel(LiftArgs a) {
    __shared__ float s_val[16][16];
    __shared__ int s_idx[16][16];
    const int f = blockIdx.x, tid = threadIdx.x, g = tid >> 4, j = tid & 15;
    const int P = a.H * a.W, J = a.J;
    float best = -INFINITY;
    int bi = 0x7fffffff;
    if (j < J) {
        const float* base = a.heat + (size_t)f * P * J + j;
#pragma unroll 8
        for (int p = g; p < P; p += 16) {
            const float v = base[(size_t)p * J];
            if (lift_better(v, p, best, bi)) { best = v; bi = p; }
        }
    }
    s_val[g][j] = best;
    s_idx[g][j] = bi;
    __syncthreads();
    if (tid < J) {
        float b = s_val[0][tid];
        int i = s_idx[0][tid];
        for (int k = 1; k < 16; ++k)
            if (lift_better(s_val[k][tid], s_idx[k][tid], b, i)) { b = s_val[k][tid]; i = s_idx[k][tid]; }
        double X = 0.0, Y = 0.0;
        if (b > 0.f) {                                   // NaN > 0 is false: masked like the reference
            X = (double)((i % a.W) * a.up + a.pad_x);
            Y = (double)((i / a.W) * a.up + a.pad_y);
        }
        const double x = X - a.cx, y = Y - a.cy;
        const double r = sqrt(x * x + y * y);
        double z = a.poly[a.n_poly - 1];                 // np.polyval(p[::-1], r): Horner from the highest power
        for (int k = a.n_poly - 2; k >= 0; --k) z = z * r + a.poly[k];
        const double v[3] = {x, y, -z};
        const double n = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
        const double d = a.depth[(size_t)f * J + tid];
        for (int c = 0; c < 3; ++c) {
            const double o = v[c] / n * d;
            if (a.out64) a.out64[((size_t)f * J + tid) * 3 + c] = o;
            if (a.out32) a.out32[((size_t)f * J + tid) * 3 + c] = (float)o;
        }
    }
}

// Streaming variant (used when a frame is a whole number of float4): A = the largest thread count <= 256 with
// 4A a multiple of J, so that a thread's four lanes of every 16-byte load always hold the same four joints
// ((4t + c) mod J, the stride 4A being a multiple of J) and the running maxima stay in registers.  A wavefront's
// load is 1 KB contiguous.  The 4A candidates are then combined per joint by 16 lanes each.
typedef float lift_f4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void lift_skeleton_stream_kernel(LiftArgs a, int A) {
    __shared__ float c_val[16 * 80];
    __shared__ int c_idx[16 * 80];
    const int f = blockIdx.x, tid = threadIdx.x, J = a.J;
    const int n4 = a.H * a.W * J / 4, n_slots = 4 * A / J;
    const lift_f4* base = reinterpret_cast<const lift_f4*>(a.heat + (size_t)f * a.H * a.W * J);
    if (tid < A) {
        float best[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        int bi[4] = {0x7fffffff, 0x7fffffff, 0x7fffffff, 0x7fffffff};
#pragma unroll 8
        for (int q = tid; q < n4; q += A) {
            const lift_f4 v = __builtin_nontemporal_load(base + q);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                // ascending scan: a strict improvement (or the first NaN) takes over; the pixel index (a division by
                // the runtime J) is only worked out then
                if (v[c] > best[c] || (v[c] != v[c] && best[c] == best[c])) { best[c] = v[c]; bi[c] = (4 * q + c) / J; }
            }
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int e = 4 * tid + c, j = e % J, slot = e / J;
            c_val[j * 80 + slot] = best[c];
            c_idx[j * 80 + slot] = bi[c];
        }
    }
    __syncthreads();
    const int j = tid >> 4, part = tid & 15;
    float b = -INFINITY;
    int i = 0x7fffffff;
    if (j < J)
        for (int sl = part; sl < n_slots; sl += 16)
            if (lift_better(c_val[j * 80 + sl], c_idx[j * 80 + sl], b, i)) { b = c_val[j * 80 + sl]; i = c_idx[j * 80 + sl]; }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) {
        const float ob = __shfl_xor(b, o, 16);
        const int oi = __shfl_xor(i, o, 16);
        if (lift_better(ob, oi, b, i)) { b = ob; i = oi; }
    }
    if (j < J && part == 0) {
        double X = 0.0, Y = 0.0;
        if (b > 0.f) {
            X = (double)((i % a.W) * a.up + a.pad_x);
            Y = (double)((i / a.W) * a.up + a.pad_y);
        }
        const double x = X - a.cx, y = Y - a.cy;
        const double r = sqrt(x * x + y * y);
        double z = a.poly[a.n_poly - 1];
        for (int k = a.n_poly - 2; k >= 0; --k) z = z * r + a.poly[k];
        const double v[3] = {x, y, -z};
        const double n = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
        const double d = a.depth[(size_t)f * J + j];
        for (int c = 0; c < 3; ++c) {
            const double o = v[c] / n * d;
            if (a.out64) a.out64[((size_t)f * J + j) * 3 + c] = o;
            if (a.out32) a.out32[((size_t)f * J + j) * 3 + c] = (float)o;
        }
    }
}

int launch_lift(gem_handle* h, const float* heat, const double* depth, int F, const double* poly, int n_poly, int up, int pad_x,
                int pad_y, double* out64, float* out32, hipStream_t s) {
    LiftArgs a;
    a.heat = heat; a.depth = depth; a.out64 = out64; a.out32 = out32;
    a.F = F; a.H = h->cfg.heat_h; a.W = h->cfg.heat_w; a.J = h->J; a.up = up; a.pad_x = pad_x; a.pad_y = pad_y; a.n_poly = n_poly;
    for (int i = 0; i < GEM_MAX_POLY; ++i) a.poly[i] = i < n_poly ? poly[i] : 0.0;
    a.cx = h->cfg.cx; a.cy = h->cfg.cy;
    Profile::Rec rec;
    const bool prof = h->prof.on;
    if (prof) {
        GEM_HIP(hipEventCreate(&rec.a)); GEM_HIP(hipEventCreate(&rec.b));
        rec.family = 3; rec.flops = (double)F * a.H * a.W * a.J * sizeof(float);      // algorithmic bytes of this launch
        GEM_HIP(hipEventRecord(rec.a, s));
    }
    const int Jr = a.J / (a.J % 4 == 0 ? 4 : a.J % 2 == 0 ? 2 : 1);      // J / gcd(J, 4)
    const int A = 256 / Jr * Jr;
    static const bool simple = dev_env("GEM_LIFT_SIMPLE") != nullptr;
    if (!simple && (a.H * a.W * a.J) % 4 == 0 && 4 * A / a.J <= 80 && (reinterpret_cast<uintptr_t>(heat) & 15) == 0)
        hipLaunchKernelGGL(lift_skeleton_stream_kernel, dim3(F), dim3(256), 0, s, a, A);
    else
        hipLaunchKernelGGL(lift_skeleton_kernel, dim3(F), dim3(256), 0, s, a);
    GEM_HIP(hipGetLastError());
    if (prof) { GEM_HIP(hipEventRecord(rec.b, s)); h->prof.recs.push_back(rec); }
    return 0;
}

}  // namespace gem
