// Energy terms + analytic gradient of ONE window by ONE wavefront in a single pass over its (frame, joint) PAIRS -- the form
// the bf16 multi-window tail (tail_bf16.hip) uses from round 4 on.  energy_window (energy_device.h) walks the T*J*3 values five
// times with three LDS scratch arrays per window and a wave-level hand-off between the passes; here a lane owns a pair, keeps its
// three coordinates in registers and RECOMPUTES what the other form stores -- the accelerations of the neighbouring frames (five
// frames of the pair's joint are read) and the bone terms of the joint's children -- so there is no scratch, no hand-off, and
// the 150 pairs of a 10 x 15 window are three trips of 64 lanes instead of forty.  Same formulas as energy_window; what differs
// is rounding at the fp32 level: reciprocals and square roots are the 1-ulp hardware forms, and a lane sums its (at most nine)
// terms of each energy in fp32 before the fp64 wave reduction -- three orders of magnitude below the bf16 noise of the decoded
// pose this header is used with.
// Reference: optimizer.py:139-149,172-177,202-213,226-240; utils/fisheye/FishEyeCalibrated.py:96-129.
#pragma once
#include "energy_device.h"

namespace gem {

#ifdef GEM_TB_DEBUG_DUMP
__device__ float* g_ep_dump = nullptr;       // developer harness only: [B][T*J][8] intermediates of the x component
#endif

// 1-ulp reciprocal / square root (v_rcp_f32, v_sqrt_f32).  The bf16 decoder mode that uses this header already carries 2^-9 of
// relative noise on every decoded coordinate; the correctly rounded sequences (about ten instructions each) buy nothing here.
__device__ __forceinline__ float rcp1(float v) { return __builtin_amdgcn_rcpf(v); }
__device__ __forceinline__ float sqrt1(float v) { return __builtin_amdgcn_sqrtf(v); }

// ---- one (frame, joint) pair: its gradient values and its contributions to the five energies -------------------------------------
// xs: the decoded pose of the pair's window, dense [T][J*3] fp32 in LDS; mbl: the window's mean bone lengths [J] in LDS; par_l / ch_l:
// the skeleton tables in LDS ([J] parents, [J][MAXJ] child lists, -1 terminated, 16-byte aligned rows); x0v / pkey / pval: the pair's
// stage-input pose and texel-cache record, requested by the caller before anything else and consumed as late as the arithmetic
// allows (the 3-D term closes the pair).  es = {3-D, smoothness, bone, vae, reprojection} of THIS pair (0 when !valid: a lane past
// the last pair works on a clamped pair and its results are dropped).
// Shape: every LDS read whose address does not depend on another read is issued up front (the pair's five frames, the parent index,
// the first three children, its bone length: ONE round trip), then the reads behind the skeleton tables (parent and child
// coordinates, their bone lengths: a second), then straight-line arithmetic: conditions on the frame index are selects of read
// offsets / of 0.f terms, an absent child is the joint itself (a zero bone: contributes -0), so the only branches left are the rare
// fourth-and-later children and the texel fetch of the reprojection term.
template <int CT, int CJ>
__device__ __forceinline__ void pair_terms(const EnergyArgs& a, int b, int p, bool valid, int frame0, const float* __restrict__ xs,
                                           const float* __restrict__ mbl, const int* __restrict__ par_l, const int* __restrict__ ch_l,
                                           const float (&x0v)[3], int pkey, const f32x4_t& pval, bool use_cache, float (&gout)[3],
                                           float (&es)[5]) {
    typedef int i32x4_t __attribute__((ext_vector_type(4)));
    const int T = CT ? CT : a.T, J = CJ ? CJ : a.J, JC = J * 3, TJ = T * J;
    const float w2 = 2.f * a.ws;
    const int t = p / J, j = p - t * J;
    const int e0 = p * 3;
    // ---- first round trip
    const bool c0 = t >= 1 && t <= T - 2, cm = t >= 2, cp = t <= T - 3;
    const int om1 = t >= 1 ? JC : 0, om2 = cm ? 2 * JC : 0, op1 = t <= T - 2 ? JC : 0, op2 = cp ? 2 * JC : 0;
    float x[3], xm1[3], xm2[3], xp1[3], xp2[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const int e = e0 + c;
        x[c] = xs[e]; xm1[c] = xs[e - om1]; xm2[c] = xs[e - om2]; xp1[c] = xs[e + op1]; xp2[c] = xs[e + op2];
    }
    const int par = par_l[j];
    const i32x4_t ch4 = *reinterpret_cast<const i32x4_t*>(ch_l + j * MAXJ);
    const float mbj = mbl[j];
    // ---- second round trip: behind the skeleton tables
    float xpar[3], xc[3][3], mbc[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) xpar[c] = xs[(t * J + par) * 3 + c];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const int cj = ch4[q] >= 0 ? ch4[q] : j;         // no such child: the joint itself = a zero bone
#pragma unroll
        for (int c = 0; c < 3; ++c) xc[q][c] = xs[(t * J + cj) * 3 + c];
        mbc[q] = mbl[cj];
    }
    // ---- smoothness term (a_t = x_(t-1) - 2 x_t + x_(t+1) for t = 1 .. T-2; dE/dx_t = 2 ws (a_(t-1) - 2 a_t + a_(t+1)))
    float gs[3], esm = 0.f, evae = 0.f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float acc = c0 ? xm1[c] - 2.f * x[c] + xp1[c] : 0.f;
        esm += acc * acc; evae += x[c] * x[c];
        gs[c] = -2.f * w2 * acc;
        gs[c] += w2 * (cm ? xm2[c] - 2.f * xm1[c] + x[c] : 0.f);
        gs[c] += w2 * (cp ? x[c] - 2.f * xp1[c] + xp2[c] : 0.f);
    }
    // ---- bone length: own bone (towards the parent), minus the bones of the children (recomputed from their coordinates)
    float gx, gy, gz, ebone;
    {
        const float bx = x[0] - xpar[0], by = x[1] - xpar[1], bz = x[2] - xpar[2];
        const float len = sqrt1(bx * bx + by * by + bz * bz);
        const float diff = len - mbj;
        ebone = diff * diff;
        const float coef = len > 0.f ? 2.f * a.wb * diff * rcp1(len) : 0.f;     // d|v|/dv := 0 at v = 0 (torch)
        gx = coef * bx; gy = coef * by; gz = coef * bz;
    }
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const float bx = xc[q][0] - x[0], by = xc[q][1] - x[1], bz = xc[q][2] - x[2];
        const float len = sqrt1(bx * bx + by * by + bz * bz);
        const float diff = len - mbc[q];
        const float coef = len > 0.f ? 2.f * a.wb * diff * rcp1(len) : 0.f;
        gx -= coef * bx; gy -= coef * by; gz -= coef * bz;
    }
    if (ch4[3] >= 0) {                                   // more than three children (not the reference's skeleton)
        const int* ch = ch_l + j * MAXJ;
#pragma nounroll
        for (int q = 3; q < MAXJ && ch[q] >= 0; ++q) {
            const int cj = ch[q];
            const float* xq = xs + (t * J + cj) * 3;
            const float bx = xq[0] - x[0], by = xq[1] - x[1], bz = xq[2] - x[2];
            const float len = sqrt1(bx * bx + by * by + bz * bz);
            const float diff = len - mbl[cj];
            const float coef = len > 0.f ? 2.f * a.wb * diff * rcp1(len) : 0.f;
            gx -= coef * bx; gy -= coef * by; gz -= coef * bz;
        }
    }
    // ---- reprojection (see energy_window for the derivation and the texel-block cache)
    float erep = 0.f;
    if (a.wr != 0.f) {
        const float xx = x[0], y = x[1], z = x[2];
        const float zz = -z;
        const float nn = sqrt1(xx * xx + y * y);
        const float inv = rcp1(nn);
        const float theta = atanf(zz * inv);
        float rho = a.poly[0], drho = 0.f, ti = 1.f;
#pragma unroll
        for (int i = 1; i < GEM_MAX_POLY; ++i) {          // (compile-time coefficient slots: scalar loads issued together)
            if (i < a.n_poly) {
                drho += (float)i * a.poly[i] * ti;
                ti *= theta;
                rho += ti * a.poly[i];
            }
        }
        const float ux = xx * inv, uy = y * inv;
        const float u = ux * rho + a.cx, v = uy * rho + a.cy;
        const float gxn = ((u - 128.f) - 512.f) / 512.f, gyn = (v - 512.f) / 512.f;
        const float ix = ((gxn + 1.f) / 2.f) * (float)(a.W - 1);
        const float iy = ((gyn + 1.f) / 2.f) * (float)(a.H - 1);
        const float fx0 = floorf(ix), fy0 = floorf(iy);
        const float fx = ix - fx0, fy = iy - fy0;
        const bool in = fx0 >= -1.f && fx0 < (float)a.W && fy0 >= -1.f && fy0 < (float)a.H;
        const int x0i = in ? (int)fx0 : 0, y0i = in ? (int)fy0 : 0;
        const bool xl = in && x0i >= 0, xr = in && x0i + 1 < a.W, yt = y0i >= 0, yb = y0i + 1 < a.H;
        const int xa = x0i < 0 ? 0 : x0i, xb = x0i + 1 < a.W ? x0i + 1 : a.W - 1;
        const int ya = y0i < 0 ? 0 : y0i, yc = y0i + 1 < a.H ? y0i + 1 : a.H - 1;
        const float* hm = a.heat + ((size_t)(frame0 + t) * a.H * a.W) * J + j;
        float nw, ne, sw, se;
        const int key = in ? (ya * a.W + xa) | ((yc * a.W + xb) << 16) : -1;
        const size_t ci = (size_t)b * TJ + p;
        const bool hit = use_cache && in && pkey == key;
        if (hit) {
            nw = pval[0]; ne = pval[1]; sw = pval[2]; se = pval[3];
        } else {
            nw = hm[((size_t)ya * a.W + xa) * J];
            ne = hm[((size_t)ya * a.W + xb) * J];
            sw = hm[((size_t)yc * a.W + xa) * J];
            se = hm[((size_t)yc * a.W + xb) * J];
            if (use_cache && in && valid) {
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
        erep = nn == 0.f ? __builtin_nanf("") : -val;       // the reference raises "norm is zero!" (FishEyeCalibrated.py:124-127)
        const float dix = (ne - nw) * gyw + (se - sw) * fy;
        const float diy = (sw - nw) * gxw + (se - ne) * fx;
        const float gu = -a.wr * dix * ((float)(a.W - 1) / 1024.f);
        const float gv = -a.wr * diy * ((float)(a.H - 1) / 1024.f);
        const float ir2 = rcp1(nn * nn + zz * zz);
        const float dth_dn = -zz * ir2, dth_dz = -nn * ir2;
        const float i3 = inv * inv * inv;
        const float dudx = rho * (inv - xx * xx * i3) + ux * drho * dth_dn * ux;
        const float dudy = rho * (-xx * y * i3) + ux * drho * dth_dn * uy;
        const float dudz = ux * drho * dth_dz;
        const float dvdx = rho * (-xx * y * i3) + uy * drho * dth_dn * ux;
        const float dvdy = rho * (inv - y * y * i3) + uy * drho * dth_dn * uy;
        const float dvdz = uy * drho * dth_dz;
#ifdef GEM_TB_DEBUG_DUMP
        if (g_ep_dump && valid) {
            float* dd = g_ep_dump + ((size_t)b * TJ + p) * 8;
            dd[0] = inv - xx * xx * i3; dd[1] = ux * drho * dth_dn * ux; dd[2] = inv; dd[3] = dudx; dd[4] = xx * xx; dd[5] = rho; dd[6] = i3; dd[7] = ux;
        }
#endif
        gx += gu * dudx + gv * dvdx;
        gy += gu * dudy + gv * dvdy;
        gz += gu * dudz + gv * dvdz;
    }
    // ---- 3-D term last: its input comes from global memory
    const float gb[3] = {gx, gy, gz};
    float e3d = 0.f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float d = x[c] - x0v[c];
        e3d += d * d;
        gout[c] = ((2.f * a.w3d * d + 2.f * a.wv * x[c]) + gs[c]) + gb[c];
    }
    es[0] = valid ? e3d : 0.f; es[1] = valid ? esm : 0.f; es[2] = valid ? ebone : 0.f; es[3] = valid ? evae : 0.f; es[4] = valid ? erep : 0.f;
}

// the wave reduction of a window's five sums (a lane holds, in fp32, the sum of its pairs' terms in pair order) and its outputs
__device__ __forceinline__ void energy_finish(const EnergyArgs& a, int b, int lane, const float (&s)[5]) {
    const double e3d = wave_sum((double)s[0]), esm = wave_sum((double)s[1]), ebone = wave_sum((double)s[2]), evae = wave_sum((double)s[3]),
                 erep = wave_sum((double)s[4]);
    if (lane == 0) {
        if (a.parts) {
            double* pp = a.parts + (size_t)b * 5;
            pp[0] = e3d; pp[1] = esm; pp[2] = ebone; pp[3] = evae; pp[4] = erep;
        }
        a.f[b] = a.dw3d * e3d + a.dws * esm + a.dwb * ebone + a.dwv * evae + a.dwr * erep;
    }
}

// ---- one WAVEFRONT per window: lane l owns the pairs l, l + 64, ... (NP = ceil(T*J / 64) trips, compile time) -----------------------
// gd: bf16 gradient rows [T][ldg] of the window in LDS; gcols > 0: zeroed here first (the pad columns [J*3, gcols) stay zero; LDS
// operations of one wave complete in order), gcols = 0: the caller has zeroed them.
template <int CT, int CJ, int NP>
__device__ __forceinline__ void energy_pairs(const EnergyArgs& a, int b, int lane, const float* __restrict__ xs,
                                             const float* __restrict__ mbl, const int* __restrict__ par_l, const int* __restrict__ ch_l,
                                             uint16_t* __restrict__ gd, int ldg, int gcols, long long* dbg = nullptr) {
    // (the lane index is re-derived here -- two v_mbcnt -- rather than kept alive, or spilled, across the caller's matrix layers)
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane));
    const int T = CT ? CT : a.T, J = CJ ? CJ : a.J, TJ = T * J;
    int dbg_n = 0;
#define EP_PROBE() if (dbg && lane == 0) dbg[dbg_n++] = clock64();
    EP_PROBE();
    // the texel-block cache records of the lane's pairs: requested first, needed only behind the projection arithmetic
    const bool use_cache = a.tex_key && a.wr != 0.f;
    int pkey[NP];
    f32x4_t pval[NP];
    float x0v[NP][3];
#pragma unroll
    for (int it = 0; it < NP; ++it) {
        pkey[it] = -1;
        pval[it] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        const int p = lane + it * 64;
        const float* x0 = a.X0 + ((size_t)b * TJ + (p < TJ ? p : TJ - 1)) * 3;
        x0v[it][0] = x0[0]; x0v[it][1] = x0[1]; x0v[it][2] = x0[2];
        if (use_cache && p < TJ) {
            pkey[it] = a.tex_key[(size_t)b * TJ + p];
            pval[it] = *reinterpret_cast<const f32x4_t*>(a.tex_val + ((size_t)b * TJ + p) * 4);
        }
    }
    const int frame0 = a.wr != 0.f ? a.frame0[b] : 0;
    if (gcols > 0) {
        const int per_row = gcols / 4;                       // 8-byte pieces per row
        for (int i = lane; i < T * per_row; i += 64) {
            const int t = i / per_row, c4 = (i - t * per_row) * 4;
            *reinterpret_cast<unsigned long long*>(gd + t * ldg + c4) = 0ull;
        }
    }
    float s[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    float gout[NP][3];
    EP_PROBE();
#pragma unroll
    for (int it = 0; it < NP; ++it) {
        const bool valid = lane + it * 64 < TJ;
        const int p = valid ? lane + it * 64 : TJ - 1;
        float es[5];
        pair_terms<CT, CJ>(a, b, p, valid, frame0, xs, mbl, par_l, ch_l, x0v[it], pkey[it], pval[it], use_cache, gout[it], es);
#pragma unroll
        for (int k = 0; k < 5; ++k) s[k] += es[k];
        EP_PROBE();
    }
#pragma unroll
    for (int it = 0; it < NP; ++it) {
        const int p = lane + it * 64;
        if (p < TJ) {
            const int t = p / J, j = p - t * J;
            uint16_t* o = gd + t * ldg + j * 3;
#pragma unroll
            for (int c = 0; c < 3; ++c) o[c] = (uint16_t)(__builtin_bit_cast(unsigned int, (float)(__bf16)gout[it][c]) >> 16);
        }
    }
    EP_PROBE();
    energy_finish(a, b, lane, s);
    EP_PROBE();
#undef EP_PROBE
}

// ---- the WHOLE workgroup on the pairs of its windows: thread tid owns the pairs tid, tid + NTHR, ... of the workgroup's G x T*J pairs
// (NPW = ceil(G * T*J / NTHR) trips, compile time; pair gp belongs to window gp / (T*J)).  Three windows of 10 x 15 are 450 pairs =
// ONE trip of 512 threads where a wavefront per window takes three trips and leaves five of eight wavefronts idle.  A window's sums
// must not depend on how its pairs were dealt: every pair parks its five terms in LDS (`epair`: [G * T*J][5] fp32) and, behind a
// barrier, wavefront w reduces window w exactly the way energy_pairs does -- lane l adds its pairs l, l + 64, ... in that order, then
// the wave reduction -- so both forms return the same bits.
// bwin_l: [G] global window index of each of the workgroup's windows (LDS); xs0 / mbl0 / gd0: window 0's pose / bone lengths /
// gradient rows, window strides x_stride (floats), MAXJ (floats), gd_stride (bf16 elements).
template <int CT, int CJ, int NPW, int NTHR, typename Barrier>
__device__ __forceinline__ void energy_pairs_wg(const EnergyArgs& a, const int* __restrict__ bwin_l, int nwin, int tid,
                                                const float* __restrict__ xs0, int x_stride, const float* __restrict__ mbl0,
                                                const int* __restrict__ par_l, const int* __restrict__ ch_l, uint16_t* __restrict__ gd0,
                                                int gd_stride, int ldg, float* __restrict__ epair, Barrier barrier) {
    static_assert(CT > 0 && CJ > 0, "compile-time window shape");
    constexpr int TJ = CT * CJ, NP = (TJ + 63) / 64;
    const bool use_cache = a.tex_key && a.wr != 0.f;
    const int npairs = nwin * TJ;
    int wi[NPW], pp[NPW], bb[NPW], fr0[NPW], pkey[NPW];
    f32x4_t pval[NPW];
    float x0v[NPW][3];
#pragma unroll
    for (int it = 0; it < NPW; ++it) {
        const int gp = min(tid + it * NTHR, npairs - 1);
        wi[it] = gp / TJ; pp[it] = gp - wi[it] * TJ;
        bb[it] = bwin_l[wi[it]];
        const size_t ci = (size_t)bb[it] * TJ + pp[it];
        const float* x0 = a.X0 + ci * 3;
        x0v[it][0] = x0[0]; x0v[it][1] = x0[1]; x0v[it][2] = x0[2];
        pkey[it] = -1;
        pval[it] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        if (use_cache) {
            pkey[it] = a.tex_key[ci];
            pval[it] = *reinterpret_cast<const f32x4_t*>(a.tex_val + ci * 4);
        }
        fr0[it] = a.wr != 0.f ? a.frame0[bb[it]] : 0;
    }
#pragma unroll
    for (int it = 0; it < NPW; ++it) {
        const bool valid = tid + it * NTHR < npairs;
        float gout[3], es[5];
        pair_terms<CT, CJ>(a, bb[it], pp[it], valid, fr0[it], xs0 + wi[it] * x_stride, mbl0 + wi[it] * MAXJ, par_l, ch_l, x0v[it], pkey[it],
                           pval[it], use_cache, gout, es);
        if (valid) {
            const int t = pp[it] / CJ, j = pp[it] - t * CJ;
            uint16_t* o = gd0 + wi[it] * gd_stride + t * ldg + j * 3;
#pragma unroll
            for (int c = 0; c < 3; ++c) o[c] = (uint16_t)(__builtin_bit_cast(unsigned int, (float)(__bf16)gout[c]) >> 16);
            float* e = epair + (size_t)(wi[it] * TJ + pp[it]) * 5;
#pragma unroll
            for (int k = 0; k < 5; ++k) e[k] = es[k];
        }
    }
    barrier();
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    if (wave < nwin) {
        float s[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int it = 0; it < NP; ++it) {
            const int p = lane + it * 64;
            if (p < TJ) {
                const float* e = epair + (size_t)(wave * TJ + p) * 5;
#pragma unroll
                for (int k = 0; k < 5; ++k) s[k] += e[k];
            }
        }
        energy_finish(a, bwin_l[wave], lane, s);
    }
}

}  // namespace gem
