// Per-window energy terms and their analytic gradient w.r.t. the decoded pose, plus the small
// data-movement kernels around the decoder (gfx950).
//
// energy_kernel: one 64-lane wavefront per window.  The decoded pose X[T,J,3] (450 floats) is staged
// in LDS; every lane owns up to ceil(T*J/64) joints.  Terms (reference file:line):
//   E_3d     sum (X - X_init)^2                                   optimizer.py:210-213
//   E_smooth sum_t |X[t-1] - 2 X[t] + X[t+1]|^2                   optimizer.py:202-208
//   E_bone   sum_{t,j} (|X[t,j]-X[t,parent j]| - mean_len[j])^2   optimizer.py:172-177
//   E_vae    sum X^2 (on the pose, weight 0 by default)           optimizer.py:238
//   E_reproj -sum bilinear(H[t,j], pi(X[t,j]))                    optimizer.py:139-149 with
//            pi = fisheye polynomial projection                   utils/fisheye/FishEyeCalibrated.py:96-129
// Terms are evaluated in fp32 like the reference; the five sums are accumulated in fp64 with a
// wavefront butterfly reduction (the reference sums in fp32 and casts the total to a python float).
// Heat-maps are read in place from the pickle layout [frame][y][x][joint]: 4 texels per joint per
// evaluation, no transposed copy (the reference makes one per window, optimizer.py:251-252).
#include "energy_device.h"

namespace gem {

__global__ void pack_pose_kernel(const float* __restrict__ src, float* __restrict__ dst, int rows, int C) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * PAD) return;
    const int r = i / PAD, c = i % PAD;
    dst[i] = c < C ? src[(size_t)r * C + c] : 0.f;
}
__global__ void unpack_pose_kernel(const float* __restrict__ src, float* __restrict__ dst, int rows, int C) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * C) return;
    const int r = i / C, c = i % C;
    dst[i] = src[(size_t)r * PAD + c];
}
int launch_pack_pose(const float* src, float* dst, int rows, int C, hipStream_t s) {
    const int n = rows * PAD;
    hipLaunchKernelGGL(pack_pose_kernel, dim3((n + 255) / 256), dim3(256), 0, s, src, dst, rows, C);
    GEM_HIP(hipGetLastError());
    return 0;
}
int launch_unpack_pose(const float* src, float* dst, int rows, int C, hipStream_t s) {
    const int n = rows * C;
    hipLaunchKernelGGL(unpack_pose_kernel, dim3((n + 255) / 256), dim3(256), 0, s, src, dst, rows, C);
    GEM_HIP(hipGetLastError());
    return 0;
}

// z = mu + eps * exp(0.5 logvar)   (SeqConvVAE.py:159-169); mulv is [B, 2*Dp] = [mu | logvar]
__global__ void reparam_kernel(const float* __restrict__ mulv, const float* __restrict__ eps, float* mu, float* logvar,
                               float* z, float* z2, int B, int D, int Dp) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * Dp) return;
    const int b = i / Dp, k = i % Dp;
    float m = 0.f, lv = 0.f, zz = 0.f;
    if (k < D) {
        m = mulv[(size_t)b * 2 * Dp + k];
        lv = mulv[(size_t)b * 2 * Dp + Dp + k];
        zz = eps ? eps[(size_t)b * D + k] * expf(0.5f * lv) + m : m;
        if (mu) mu[(size_t)b * D + k] = m;
        if (logvar) logvar[(size_t)b * D + k] = lv;
        if (z) z[(size_t)b * D + k] = zz;
    }
    if (z2) z2[i] = zz;     // padded copy [B,Dp] for the optimiser
}
int launch_reparam(const float* mulv, const float* eps, float* mu, float* logvar, float* z, float* z2, int B, int D, int Dp,
                   hipStream_t s) {
    const int n = B * Dp;
    hipLaunchKernelGGL(reparam_kernel, dim3((n + 255) / 256), dim3(256), 0, s, mulv, eps, mu, logvar, z, z2, B, D, Dp);
    GEM_HIP(hipGetLastError());
    return 0;
}
__global__ void pad_latent_kernel(const float* __restrict__ z, float* __restrict__ zp, int B, int D, int Dp) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * Dp) return;
    const int b = i / Dp, k = i % Dp;
    zp[i] = k < D ? z[(size_t)b * D + k] : 0.f;
}
__global__ void unpad_latent_kernel(const float* __restrict__ zp, float* __restrict__ z, int B, int D, int Dp) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * D) return;
    const int b = i / D, k = i % D;
    z[i] = zp[(size_t)b * Dp + k];
}
int launch_pad_latent(const float* z, float* zp, int B, int D, int Dp, hipStream_t s) {
    const int n = B * Dp;
    hipLaunchKernelGGL(pad_latent_kernel, dim3((n + 255) / 256), dim3(256), 0, s, z, zp, B, D, Dp);
    GEM_HIP(hipGetLastError());
    return 0;
}
int launch_unpad_latent(const float* zp, float* z, int B, int D, int Dp, hipStream_t s) {
    const int n = B * D;
    hipLaunchKernelGGL(unpad_latent_kernel, dim3((n + 255) / 256), dim3(256), 0, s, zp, z, B, D, Dp);
    GEM_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void energy_kernel(EnergyArgs a) {
    __shared__ float xs[ENERGY_SCRATCH];      // decoded pose
    __shared__ float gs[ENERGY_SCRATCH];      // gradient accumulator
    __shared__ float bs[ENERGY_SCRATCH];      // coef * bone vector of joint j (for the parent gather)
    __shared__ float as[ENERGY_SCRATCH];      // accelerations
    const int slot = blockIdx.x;
    if (a.n_dev && slot >= *a.n_dev) return;
    const int b = a.perm ? a.perm[slot] : slot;
    // (the usual window shape gets compile-time index arithmetic: same numbers, fewer instructions)
    if (a.T == 10 && a.J == 15)
        energy_window<true, 64, 10, 15>(a, b, threadIdx.x, a.Xp + (size_t)slot * 10 * PAD, PAD, xs, gs, bs, as, a.dXp + (size_t)slot * 10 * PAD,
                                        PAD, PAD, a.dXp_b ? a.dXp_b + (size_t)slot * 10 * PAD : nullptr);
    else
        energy_window<true>(a, b, threadIdx.x, a.Xp + (size_t)slot * a.T * PAD, PAD, xs, gs, bs, as, a.dXp + (size_t)slot * a.T * PAD,
                            PAD, PAD, a.dXp_b ? a.dXp_b + (size_t)slot * a.T * PAD : nullptr);
}

int launch_energy(gem_handle* h, const EnergyArgs& a, int B, hipStream_t s) {
    if (a.T > MAXT || a.J > MAXJ) { set_error("energy kernel: T <= 16 and J <= 16 supported"); return 1; }
    Profile::Rec rec;
    const bool prof = h->prof.on;
    if (prof) {
        GEM_HIP(hipEventCreate(&rec.a)); GEM_HIP(hipEventCreate(&rec.b));
        rec.family = 1; rec.flops = 0;
        GEM_HIP(hipEventRecord(rec.a, s));
    }
    note_kernel(h, reinterpret_cast<const void*>(energy_kernel));
    hipLaunchKernelGGL(energy_kernel, dim3(B), dim3(64), 0, s, a);
    GEM_HIP(hipGetLastError());
    if (prof) { GEM_HIP(hipEventRecord(rec.b, s)); h->prof.recs.push_back(rec); }
    commit_kernel_names(h, prof ? rec.family : -1);
    return 0;
}

// ------------------------------------------------------------------------------------------------
// mean_bone_length of a chunk: mean over frames of |X_j - X_parent(j)|   (optimizer.py:42-43,89-94)
__global__ __launch_bounds__(256) void mean_bone_kernel(const float* __restrict__ pose, int n_frames, int J,
                                                        const int* __restrict__ parents, float* __restrict__ out) {
    __shared__ float red[4][MAXJ];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int j = 0; j < J; ++j) {
        float acc = 0.f;
        const int par = parents[j];
        for (int f = tid; f < n_frames; f += 256) {
            const float* a = pose + ((size_t)f * J + j) * 3;
            const float* p = pose + ((size_t)f * J + par) * 3;
            const float dx = a[0] - p[0], dy = a[1] - p[1], dz = a[2] - p[2];
            acc += sqrtf(dx * dx + dy * dy + dz * dz);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
        if (lane == 0) red[wave][j] = acc;
    }
    __syncthreads();
    if (tid < J) out[tid] = (red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid]) / (float)n_frames;
}
int launch_mean_bone(gem_handle* h, const float* pose, int n_frames, float* out, hipStream_t s) {
    hipLaunchKernelGGL(mean_bone_kernel, dim3(1), dim3(256), 0, s, pose, n_frames, h->J, h->d_parents, out);
    GEM_HIP(hipGetLastError());
    return 0;
}

// window gather: frames stored once -> [B,T,J*3]
__global__ void gather_windows_kernel(const float* __restrict__ frames, const int32_t* __restrict__ frame0,
                                      float* __restrict__ out, int B, int T, int JC) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * T * JC) return;
    const int b = i / (T * JC), r = i - b * T * JC;
    out[i] = frames[(size_t)frame0[b] * JC + r];
}
int launch_gather_windows(const float* frames, const int32_t* frame0, float* out, int B, int T, int JC, hipStream_t s) {
    const int n = B * T * JC;
    hipLaunchKernelGGL(gather_windows_kernel, dim3((n + 255) / 256), dim3(256), 0, s, frames, frame0, out, B, T, JC);
    GEM_HIP(hipGetLastError());
    return 0;
}

// ---- rigid transforms between the stages, float64 like the numpy reference -----------------------
__device__ void inv4_rigid_general(const double* m, double* o) {
    // general 4x4 inverse of [R t; 0 1]-like matrices via the 3x3 adjugate (np.linalg.inv, utils.py:104);
    // the last row is assumed (0,0,0,1) as produced by slam_reader.py:110-121.
    const double a = m[0], b = m[1], c = m[2], d = m[4], e = m[5], f = m[6], g = m[8], hh = m[9], i = m[10];
    const double A = e * i - f * hh, Bc = -(d * i - f * g), Cc = d * hh - e * g;
    const double det = a * A + b * Bc + c * Cc;
    const double id = 1.0 / det;
    double r[9] = {A * id, -(b * i - c * hh) * id, (b * f - c * e) * id,
                   Bc * id, (a * i - c * g) * id, -(a * f - c * d) * id,
                   Cc * id, -(a * hh - b * g) * id, (a * e - b * d) * id};
    const double tx = m[3], ty = m[7], tz = m[11];
    for (int q = 0; q < 3; ++q) {
        o[q * 4 + 0] = r[q * 3 + 0]; o[q * 4 + 1] = r[q * 3 + 1]; o[q * 4 + 2] = r[q * 3 + 2];
        o[q * 4 + 3] = -(r[q * 3 + 0] * tx + r[q * 3 + 1] * ty + r[q * 3 + 2] * tz);
    }
    o[12] = 0; o[13] = 0; o[14] = 0; o[15] = 1;
}

// X_rel[t] = C0^-1 C_t X_loc[t]   (utils/utils.py:99-112), result cast to fp32 for the global stage
__global__ __launch_bounds__(64) void relative_global_kernel(const float* __restrict__ local, const double* __restrict__ cams,
                                                             const int32_t* __restrict__ frame0, float* __restrict__ rel,
                                                             int T, int J) {
    __shared__ double M[MAXT][12];
    const int b = blockIdx.x, lane = threadIdx.x;
    const double* c0 = cams + (size_t)frame0[b] * 16;
    if (lane < T) {
        double inv0[16];
        inv4_rigid_general(c0, inv0);
        const double* ct = c0 + (size_t)lane * 16;
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 4; ++c) {
                double acc = 0;
                for (int k = 0; k < 4; ++k) acc += inv0[r * 4 + k] * ct[k * 4 + c];
                M[lane][r * 4 + c] = acc;
            }
    }
    __syncthreads();
    for (int p = lane; p < T * J; p += 64) {
        const int t = p / J;
        const float* x = local + ((size_t)b * T * J + p) * 3;
        const double X = x[0], Y = x[1], Z = x[2];
        float* o = rel + ((size_t)b * T * J + p) * 3;
        for (int r = 0; r < 3; ++r)
            o[r] = (float)(M[t][r * 4 + 0] * X + M[t][r * 4 + 1] * Y + M[t][r * 4 + 2] * Z + M[t][r * 4 + 3]);
    }
}
// X_glob[t] = C0 X_rel[t]   (optimizer.py:302-308), float64 out
__global__ __launch_bounds__(64) void to_global_kernel(const float* __restrict__ rel, const double* __restrict__ cams,
                                                       const int32_t* __restrict__ frame0, double* __restrict__ out, int T, int J) {
    const int b = blockIdx.x, lane = threadIdx.x;
    const double* c0 = cams + (size_t)frame0[b] * 16;
    for (int p = lane; p < T * J; p += 64) {
        const float* x = rel + ((size_t)b * T * J + p) * 3;
        const double X = x[0], Y = x[1], Z = x[2];
        double* o = out + ((size_t)b * T * J + p) * 3;
        for (int r = 0; r < 3; ++r) o[r] = c0[r * 4 + 0] * X + c0[r * 4 + 1] * Y + c0[r * 4 + 2] * Z + c0[r * 4 + 3];
    }
}
int launch_relative_global(const float* local, const double* cams, const int32_t* frame0, float* rel, int B, int T, int J,
                           hipStream_t s) {
    hipLaunchKernelGGL(relative_global_kernel, dim3(B), dim3(64), 0, s, local, cams, frame0, rel, T, J);
    GEM_HIP(hipGetLastError());
    return 0;
}
int launch_to_global(const float* rel, const double* cams, const int32_t* frame0, double* out, int B, int T, int J,
                     hipStream_t s) {
    hipLaunchKernelGGL(to_global_kernel, dim3(B), dim3(64), 0, s, rel, cams, frame0, out, T, J);
    GEM_HIP(hipGetLastError());
    return 0;
}

}  // namespace gem
