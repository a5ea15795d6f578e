// Energy terms + analytic gradient of ONE window, executed by ONE wavefront (64 lanes) or by a whole workgroup.
// Shared by the stand-alone energy kernel (energy.hip) and the fused decoder-tail kernels (tail.hip, tail_bf16.hip).
// Reference: optimizer.py:139-149,172-177,202-213,226-240; utils/fisheye/FishEyeCalibrated.py:96-129.
#pragma once
#include "gem_internal.h"

namespace gem {

__device__ __forceinline__ double wave_sum(double v) { return wave_sum_dpp(v); }
typedef float f32x4_t __attribute__((ext_vector_type(4)));

constexpr int MAXT = 16;               // frames per window supported by the LDS carve
constexpr int MAXJ = GEM_MAX_JOINTS;
constexpr int ENERGY_SCRATCH = MAXT * MAXJ * 3;     // floats per scratch array

// BLOCK_SYNC: the cooperating threads are the whole workgroup (plain __syncthreads); otherwise several wavefronts
// of a workgroup each run their own window, and only the wavefront's own LDS traffic has to be ordered.
template <bool BLOCK_SYNC>
__device__ __forceinline__ void energy_sync() {
    if (BLOCK_SYNC) {
        __syncthreads();
    } else {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
}

// Per-lane inputs of a window's energy terms that do not depend on the decoded pose, fetched AHEAD (the bf16 tail requests
// them at kernel start and keeps them in registers across its forward layers): the stage-input pose values the lane will
// meet in its strided passes, and -- per (frame, joint) pair of the lane -- mean bone length, parent index and the cached
// texel block of the reprojection term.  NE / NP = passes over the T*J*3 values / the T*J pairs with 64 lanes.
template <int NE, int NP>
struct EnergyPre {
    float x0[NE];
    float mb[NP];
    int par[NP];
    int key[NP];
    f32x4_t val[NP];
};
struct NoEnergyPre {};

// strided pass over [0, n): e = lane, lane + NT, ...; f(e, it) with the pass index `it`.  CN > 0: n is the compile-time CN,
// the pass is unrolled (so `it` is a constant inside f: register arrays indexed by it stay in registers).
template <int NT, int CN, typename F>
__device__ __forceinline__ void lane_pass(int lane, int n, F f) {
    if constexpr (CN > 0) {
#pragma unroll
        for (int it = 0; it < (CN + NT - 1) / NT; ++it) {
            const int e = lane + it * NT;
            if (e < CN) f(e, it);
        }
    } else {
        int it = 0;
        for (int e = lane; e < n; e += NT, ++it) f(e, it);
    }
}

// the prefetch itself (one wavefront per window): issued early by the caller, consumed by energy_window<..., Pre>
template <int CT, int CJ>
__device__ __forceinline__ void energy_prefetch(const EnergyArgs& a, int b, int lane, EnergyPre<(CT * CJ * 3 + 63) / 64, (CT * CJ + 63) / 64>& pre) {
    constexpr int n = CT * CJ * 3, TJ = CT * CJ;
#pragma unroll
    for (int it = 0; it < (n + 63) / 64; ++it) {
        const int e = lane + it * 64;
        pre.x0[it] = e < n ? a.X0[(size_t)b * n + e] : 0.f;
    }
#pragma unroll
    for (int it = 0; it < (TJ + 63) / 64; ++it) {
        const int p = lane + it * 64;
        pre.mb[it] = 0.f; pre.par[it] = 0; pre.key[it] = -1; pre.val[it] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        if (p < TJ) {
            const int j = p % CJ;
            pre.mb[it] = a.mean_bone[(size_t)b * CJ + j];
            pre.par[it] = a.parents[j];
            if (a.tex_key && a.wr != 0.f) {
                const size_t ci = (size_t)b * TJ + p;
                pre.key[it] = a.tex_key[ci];
                pre.val[it] = *reinterpret_cast<const f32x4_t*>(a.tex_val + ci * 4);
            }
        }
    }
}

// xsrc: decoded pose rows [T][ldx] (global or LDS); xs/gs/bs/as: LDS scratch of ENERGY_SCRATCH floats each (xs may BE xsrc
// when that is already the dense [T][J*3] image: XS_IS_SRC); gdst: gradient rows [T][ldg], columns [J*3, gcols) are zero-filled.
// NT threads work on the window (`lane` = 0..NT-1): 64 = one wavefront; more = the whole workgroup (BLOCK_SYNC), so
// that the T*J*3 = 450 values are one pass and every global-memory latency (x0, mean bone, heat-map texels) is paid
// once instead of once per 64-lane pass.
// CT / CJ: compile-time frames / joints (0: run-time a.T / a.J).  With them the index arithmetic (e / JC, p / J) folds into
// multiplies and the passes unroll; the arithmetic on the DATA is the same instruction for instruction, so results are
// bitwise those of the run-time version.  Pre: EnergyPre of this lane (NT == 64, CT and CJ given) or NoEnergyPre.
template <bool BLOCK_SYNC, int NT = 64, int CT = 0, int CJ = 0, bool XS_IS_SRC = false, typename Pre = NoEnergyPre>
__device__ __forceinline__ void energy_window(const EnergyArgs& a, int b, int lane, const float* xsrc, int ldx, float* xs,
                                              float* gs, float* bs, float* as, float* gdst, int ldg, int gcols,
                                              uint16_t* gdst_b = nullptr, const float* x0_w = nullptr, const float* mb_w = nullptr,
                                              const int* par_w = nullptr, const int* ch_w = nullptr, const Pre* pre = nullptr) {
    static_assert(NT == 64 || BLOCK_SYNC, "more than one wavefront per window needs workgroup barriers");
    static_assert((CT > 0) == (CJ > 0), "give both compile-time dimensions or none");
    constexpr bool HAS_PRE = !__is_same(Pre, NoEnergyPre);
    static_assert(!HAS_PRE || (NT == 64 && CT > 0), "prefetched inputs are per lane of ONE wavefront with compile-time dimensions");
    const int T = CT ? CT : a.T, J = CJ ? CJ : a.J, JC = J * 3, n = T * JC;
    constexpr int CN = CT * CJ * 3, CP = CT * CJ;
    // x0_w / mb_w / par_w / ch_w: this window's stage-input pose, mean bone lengths and the skeleton tables when the caller has
    // already brought them on chip (the fused tail loads them into LDS while its first layers run); else from global memory
    const float* x0 = x0_w ? x0_w : a.X0 + (size_t)b * n;
    const float* mbone = mb_w ? mb_w : a.mean_bone + (size_t)b * J;
    const int* parents = par_w ? par_w : a.parents;
    const int* children = ch_w ? ch_w : a.children;

    double e3d = 0, esm = 0, ebone = 0, evae = 0, erep = 0;
    lane_pass<NT, CN>(lane, n, [&](int e, int it) {
        const int t = e / JC, c = e - t * JC;
        const float x = xsrc[t * ldx + c];
        if (!XS_IS_SRC) xs[e] = x;
        float x0v;
        if constexpr (HAS_PRE) x0v = pre->x0[it]; else x0v = x0[e];
        const float d = x - x0v;
        e3d += (double)(d * d);
        evae += (double)(x * x);
        gs[e] = 2.f * a.w3d * d + 2.f * a.wv * x;
    });
    energy_sync<BLOCK_SYNC>();
    // smoothness: acceleration a_t (t = 1..T-2) then gather
    lane_pass<NT, CN>(lane, n, [&](int e, int) {
        const int t = e / JC;
        float acc = 0.f;
        if (t >= 1 && t <= T - 2) {
            acc = xs[e - JC] - 2.f * xs[e] + xs[e + JC];
            esm += (double)(acc * acc);
        }
        as[e] = acc;
    });
    // bone length: per (t, joint)
    lane_pass<NT, CP>(lane, T * J, [&](int p, int it) {
        const int t = p / J, j = p - t * J;
        int par;
        float mbv;
        if constexpr (HAS_PRE) { par = pre->par[it]; mbv = pre->mb[it]; } else { par = parents[j]; mbv = mbone[j]; }
        const float* xj = xs + (t * J + j) * 3;
        const float* xp = xs + (t * J + par) * 3;
        const float bx = xj[0] - xp[0], by = xj[1] - xp[1], bz = xj[2] - xp[2];
        const float len = sqrtf(bx * bx + by * by + bz * bz);
        const float diff = len - mbv;
        ebone += (double)(diff * diff);
        const float coef = len > 0.f ? 2.f * a.wb * diff / len : 0.f;     // d|v|/dv := 0 at v = 0 (torch)
        float* o = bs + (t * J + j) * 3;
        o[0] = coef * bx; o[1] = coef * by; o[2] = coef * bz;
    });
    energy_sync<BLOCK_SYNC>();
    lane_pass<NT, CN>(lane, n, [&](int e, int) {
        const int t = e / JC;
        float g = gs[e];
        const float w2 = 2.f * a.ws;
        if (t >= 1 && t <= T - 2) g -= 2.f * w2 * as[e];
        if (t >= 2) g += w2 * as[e - JC];
        if (t <= T - 3) g += w2 * as[e + JC];
        gs[e] = g;
    });
    energy_sync<BLOCK_SYNC>();
    lane_pass<NT, CP>(lane, T * J, [&](int p, int it) {
        const int t = p / J, j = p - t * J;
        float gx = bs[p * 3 + 0], gy = bs[p * 3 + 1], gz = bs[p * 3 + 2];
        const int* ch = children + j * MAXJ;
        for (int q = 0; q < MAXJ && ch[q] >= 0; ++q) {
            const float* o = bs + (t * J + ch[q]) * 3;
            gx -= o[0]; gy -= o[1]; gz -= o[2];
        }
        // reprojection (only joints of frames whose heat-map exists)
        if (a.wr != 0.f) {
            const float x = xs[p * 3 + 0], y = xs[p * 3 + 1], z = xs[p * 3 + 2];
            const float zz = -z;
            const float nn = sqrtf(x * x + y * y);
            // nn == 0 (joint on the optical axis) is rejected by the reference: Exception("norm is zero!"),
            // FishEyeCalibrated.py:124-127.  Here it poisons the window's energy with NaN explicitly (the masked texels
            // alone would leave f finite); lbfgs_advance latches a NaN closure value into the window's status, which the
            // host wrapper turns into the same exception.
            if (nn == 0.f) erep = __builtin_nan("");
            const float inv = 1.f / nn;
            const float theta = atanf(zz / nn);
            float rho = a.poly[0], drho = 0.f, ti = 1.f;
            for (int i = 1; i < a.n_poly; ++i) {
                drho += (float)i * a.poly[i] * ti;
                ti *= theta;
                rho += ti * a.poly[i];
            }
            const float ux = x * inv, uy = y * inv;
            const float u = ux * rho + a.cx, v = uy * rho + a.cy;
            // optimizer.py:143-147 + grid_sample(align_corners=True) un-normalisation
            const float gxn = ((u - 128.f) - 512.f) / 512.f, gyn = (v - 512.f) / 512.f;
            const float ix = ((gxn + 1.f) / 2.f) * (float)(a.W - 1);
            const float iy = ((gyn + 1.f) / 2.f) * (float)(a.H - 1);
            const float fx0 = floorf(ix), fy0 = floorf(iy);
            const float fx = ix - fx0, fy = iy - fy0;
            // a projection that is not finite (or far outside) samples nothing, like zeros padding.  Branch-free: the
            // four texels are fetched together from clamped (always valid) addresses and masked afterwards.
            const bool in = fx0 >= -1.f && fx0 < (float)a.W && fy0 >= -1.f && fy0 < (float)a.H;
            const int x0i = in ? (int)fx0 : 0, y0i = in ? (int)fy0 : 0;
            const bool xl = in && x0i >= 0, xr = in && x0i + 1 < a.W, yt = y0i >= 0, yb = y0i + 1 < a.H;
            const int xa = x0i < 0 ? 0 : x0i, xb = x0i + 1 < a.W ? x0i + 1 : a.W - 1;
            const int ya = y0i < 0 ? 0 : y0i, yc = y0i + 1 < a.H ? y0i + 1 : a.H - 1;
            // The 2x2 texel block under a joint rarely changes from one evaluation to the next (the joint moves by a fraction
            // of a texel): the four raw texels of the last evaluation are kept per (window, frame, joint) and re-read as ONE
            // coalesced 20-byte record instead of up to four scattered cache lines of the [frame][y][x][joint] heat-maps
            // (57 KB of lines per window and evaluation otherwise: the stand-alone energy kernel is bound by them at large
            // batch).  Same values, bit for bit; the cache is emptied at the start of every stage.
            const float* hm = a.heat + ((size_t)(a.frame0[b] + t) * a.H * a.W) * J + j;
            float nw, ne, sw, se;
            const int key = in ? (ya * a.W + xa) | ((yc * a.W + xb) << 16) : -1;      // the block's clamped corner texels (H*W <= 65536/2)
            const size_t ci = (size_t)b * (T * J) + p;
            bool hit = false;
            if (a.tex_key && in) {
                if constexpr (HAS_PRE) {
                    hit = pre->key[it] == key;
                    if (hit) { nw = pre->val[it][0]; ne = pre->val[it][1]; sw = pre->val[it][2]; se = pre->val[it][3]; }
                } else {
                    hit = a.tex_key[ci] == key;
                    if (hit) {
                        const f32x4_t c = *reinterpret_cast<const f32x4_t*>(a.tex_val + ci * 4);
                        nw = c[0]; ne = c[1]; sw = c[2]; se = c[3];
                    }
                }
            }
            if (!hit) {
                nw = hm[((size_t)ya * a.W + xa) * J];
                ne = hm[((size_t)ya * a.W + xb) * J];
                sw = hm[((size_t)yc * a.W + xa) * J];
                se = hm[((size_t)yc * a.W + xb) * J];
                if (a.tex_key && in) {
                    a.tex_key[ci] = key;
                    *reinterpret_cast<f32x4_t*>(a.tex_val + ci * 4) = f32x4_t{nw, ne, sw, se};
                }
            }
            nw = (yt && xl) ? nw : 0.f;
            ne = (yt && xr) ? ne : 0.f;
            sw = (yb && xl) ? sw : 0.f;
            se = (yb && xr) ? se : 0.f;
            const float gxw = 1.f - fx, gyw = 1.f - fy;
            const float val = nw * gxw * gyw + ne * fx * gyw + sw * gxw * fy + se * fx * fy;
            erep -= (double)val;
            const float dix = (ne - nw) * gyw + (se - sw) * fy;
            const float diy = (sw - nw) * gxw + (se - ne) * fx;
            const float gu = -a.wr * dix * ((float)(a.W - 1) / 1024.f);
            const float gv = -a.wr * diy * ((float)(a.H - 1) / 1024.f);
            const float r2 = nn * nn + zz * zz;
            const float dth_dn = -zz / r2, dth_dz = -nn / r2;
            const float i3 = inv * inv * inv;
            const float dudx = rho * (inv - x * x * i3) + ux * drho * dth_dn * ux;
            const float dudy = rho * (-x * y * i3) + ux * drho * dth_dn * uy;
            const float dudz = ux * drho * dth_dz;
            const float dvdx = rho * (-x * y * i3) + uy * drho * dth_dn * ux;
            const float dvdy = rho * (inv - y * y * i3) + uy * drho * dth_dn * uy;
            const float dvdz = uy * drho * dth_dz;
            gx += gu * dudx + gv * dvdx;
            gy += gu * dudy + gv * dvdy;
            gz += gu * dudz + gv * dvdz;
        }
        gs[p * 3 + 0] += gx; gs[p * 3 + 1] += gy; gs[p * 3 + 2] += gz;
    });
    energy_sync<BLOCK_SYNC>();
    // gradient rows, zero-padded
    if (CT > 0 && gcols == 64) {
        lane_pass<NT, CT * 64>(lane, T * 64, [&](int i, int) {
            const int t = i >> 6, c = i & 63;
            const float v = c < JC ? gs[t * JC + c] : 0.f;
            if (gdst_b) gdst_b[t * ldg + c] = (uint16_t)(__builtin_bit_cast(unsigned int, (float)(__bf16)v) >> 16);
            else gdst[t * ldg + c] = v;
        });
    } else {
        for (int i = lane; i < T * gcols; i += NT) {
            const int t = i / gcols, c = i - t * gcols;
            const float v = c < JC ? gs[t * JC + c] : 0.f;
            if (gdst_b) gdst_b[t * ldg + c] = (uint16_t)(__builtin_bit_cast(unsigned int, (float)(__bf16)v) >> 16);
            else gdst[t * ldg + c] = v;
        }
    }
    e3d = wave_sum(e3d); esm = wave_sum(esm); ebone = wave_sum(ebone); evae = wave_sum(evae); erep = wave_sum(erep);
    if (NT > 64) {                   // combine the wavefronts' partial sums in wave order (as[] is free by now)
        double* red = reinterpret_cast<double*>(as);
        const int wv = lane >> 6;
        if ((lane & 63) == 0) { red[wv * 5 + 0] = e3d; red[wv * 5 + 1] = esm; red[wv * 5 + 2] = ebone; red[wv * 5 + 3] = evae; red[wv * 5 + 4] = erep; }
        __syncthreads();
        if (lane == 0) {
            e3d = esm = ebone = evae = erep = 0.0;
            for (int w = 0; w < NT / 64; ++w) {
                e3d += red[w * 5 + 0]; esm += red[w * 5 + 1]; ebone += red[w * 5 + 2]; evae += red[w * 5 + 3]; erep += red[w * 5 + 4];
            }
        }
    }
    if (lane == 0) {
        if (a.parts) {
            double* p = a.parts + (size_t)b * 5;
            p[0] = e3d; p[1] = esm; p[2] = ebone; p[3] = evae; p[4] = erep;
        }
        a.f[b] = a.dw3d * e3d + a.dws * esm + a.dwb * ebone + a.dwv * evae + a.dwr * erep;
    }
}

}  // namespace gem
