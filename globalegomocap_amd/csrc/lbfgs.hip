// Batched L-BFGS with strong-Wolfe line search: one resumable state machine per window (gfx950).
//
// The reference runs `torch.optim.LBFGS(lr=2, max_iter=25, tolerance_change=1e-6,
// line_search_fn='strong_wolfe').step(closure)` once per window-stage (optimizer.py:261-270).  Its
// control flow is data dependent (bracketing / zoom phases, early exits), so here every window carries
// its own state and all windows advance in lock-step "evaluation rounds": a round evaluates
// (loss, dloss/dz) at every window's trial point with the batched decoder + energy kernels, then this
// kernel consumes the evaluation and either finishes the window or emits its next trial point.
// At most max_eval+1 rounds are needed (lbfgs.py: the line search gets max_ls = max_eval - evals and
// uses at most max_ls+1 evaluations).
//
// Semantics restated from torch/optim/lbfgs.py (torch 2.10): `_cubic_interpolate`, `_strong_wolfe`
// (c1=1e-4, c2=0.9, inner tolerance_change=1e-9), and `LBFGS.step` (first step min(1,1/|g|_1)*lr, memory
// update only if y.s > 1e-10, H_diag = y.s/y.y, two-loop recursion, exits on max_iter, max_eval,
// max|g| <= tol_grad, max|t d| <= tol_change, |dloss| < tol_change, g.d > -tol_change).
// Vectors are fp32 (as the reference's tensors); scalars of the line search are fp64.
//
// One 256-thread workgroup per window; the D latent values are strided over the threads (coalesced),
// dot products / max-norms are wavefront butterflies + a 4-entry LDS combine.
#include "gem_internal.h"
#include "gemm_rows.h"

namespace gem {

typedef float f32x4 __attribute__((ext_vector_type(4)));

enum { PH_INIT = 0, PH_BRACKET = 1, PH_ZOOM = 2, PH_DONE = 3 };
static_assert(PH_DONE == 3, "Workspace::done_phase (gem_internal.h) names this value for the fused compaction of gemm_rows.h");

struct AdvArgs {
    LbfgsState* state;
    const double* f;
    double* trace;            // [B] closure values consumed in this round (nullptr: not recorded)
    const float* gnew;
    float *x, *d, *g, *gp, *bg0, *bg1, *trial, *S, *Y;
    uint16_t* trial_b;        // bf16 copy of the trial point for the bf16 decoder mode (nullptr: none)
    int* phase_arr;           // [B] copy of state.phase: what compact_kernel scans (4 bytes per window instead of the 1.2 KB record)
    const int* slot_of;       // window -> slot of its gradient row (nullptr: identity)
    int *next_perm, *next_slot_of, *next_count;      // != nullptr: a window that keeps iterating takes its slot of the NEXT round here (one atomic add)
    SlabSrc gslab;            // base != nullptr: the gradient rows still lie in split-K slabs (summed here, in slab order)
    unsigned long long* clk;  // developer aid (GEM_LBFGS_CLK, -DGEM_LB_PROBE builds): [32] accumulated 100 MHz ticks per phase, [31] = windows counted
    int Dp, hist_cap;
    gem_lbfgs_opts o;
};

__device__ __forceinline__ double cubic_interpolate(double x1, double f1, double g1, double x2, double f2, double g2,
                                                    bool has_bounds, double lo, double hi) {
    if (!has_bounds) {
        lo = x1 <= x2 ? x1 : x2;
        hi = x1 <= x2 ? x2 : x1;
    }
    const double d1 = g1 + g2 - 3.0 * (f1 - f2) / (x1 - x2);
    const double d2s = d1 * d1 - g1 * g2;
    if (d2s >= 0.0) {
        const double d2 = sqrt(d2s);
        double m;
        if (x1 <= x2) m = x2 - (x2 - x1) * ((g2 + d2 - d1) / (g2 - g1 + 2.0 * d2));
        else m = x1 - (x1 - x2) * ((g1 + d2 - d1) / (g1 - g2 + 2.0 * d2));
        // python's min(max(m, lo), hi): a NaN m propagates as `lo` would not be chosen; mirror max/min order
        double r = (m > lo) ? m : lo;
        if (!(m == m)) r = m;          // max(nan, lo) -> nan in python (first argument kept)
        double q = (r < hi) ? r : hi;
        if (!(r == r)) q = r;
        return q;
    }
    return (lo + hi) / 2.0;
}

// Block-wide reductions with ONE barrier each: partials go to alternating 4-entry LDS slots, so the next
// reduction cannot overwrite values a slow wave has not read yet (a slot is reused two barriers later).
template <int NT>
struct BlockRed {
    double* red;      // [2][4][3]
    int parity;
    __device__ __forceinline__ double sum(double v) {
        double w[1] = {v};
        sumN<1>(w);
        return w[0];
    }
    // N sums with ONE barrier (independent DPP chains overlap)
    template <int N>
    __device__ __forceinline__ void sumN(double (&v)[N]) {
#pragma unroll
        for (int n = 0; n < N; ++n) v[n] = wave_sum_dpp(v[n]);
        if (NT == 64) return;
        double* r = red + 12 * parity;
        parity ^= 1;
        if ((threadIdx.x & 63) == 0) {
#pragma unroll
            for (int n = 0; n < N; ++n) r[(threadIdx.x >> 6) * 3 + n] = v[n];
        }
        __syncthreads();
#pragma unroll
        for (int n = 0; n < N; ++n) v[n] = r[n] + r[3 + n] + r[6 + n] + r[9 + n];
    }
    __device__ __forceinline__ double max(double v) {
        v = wave_max_dpp(v);
        if (NT == 64) return v;
        double* r = red + 12 * parity;
        parity ^= 1;
        if ((threadIdx.x & 63) == 0) r[(threadIdx.x >> 6) * 3] = v;
        __syncthreads();
        return nan_max(nan_max(nan_max(r[0], r[3]), r[6]), r[9]);
    }
};

// FULL: Dp == EPT * NT exactly (D = 2048 with 8 x 256): no bounds predicate around the strip loads / stores
// HB: the (s, y) ring is stored as bf16 (bf16 decoder mode: the ring is the kernel's HBM traffic at large batch, 4 k reads of
// D values per iteration with k pairs; the curvature pairs tolerate 8 bits, the iterate, gradients and all scalars stay fp32)
template <int EPT, int NT, bool FULL, bool HB>
__device__ __forceinline__ void lbfgs_advance_body(const AdvArgs& a) {
    __shared__ double red[24];
    __shared__ double ro_s[MAX_HIST];
    __shared__ double cadj_s[MAX_HIST];
    __shared__ double al_s[MAX_HIST];
    const int b = blockIdx.x, tid = threadIdx.x;
    LbfgsState* sp = a.state + b;
    int phase = sp->phase;
    // (what every live window needs to ADDRESS its round's inputs is requested with the phase: the kernel's prologue is two dependent
    // round trips -- these scalars, then state + ring scalars + gradient + d + x together -- instead of four)
    const int gslot = a.slot_of ? a.slot_of[b] : b;
    const double f_new = a.f[b];
    if (phase == PH_DONE) return;
    BlockRed<NT> R{red, 0};
    // developer aid: a library built with -DGEM_LB_PROBE (tools/lbfgs_phase_run.sh) and run with GEM_LBFGS_CLK=1 prints the mean
    // time of every phase of the windows that compute a new direction from the ring; the product build has no probes
#ifdef GEM_LB_PROBE
    unsigned long long ts[10];
#pragma unroll
    for (int i = 0; i < 10; ++i) ts[i] = 0ull;
#define LB_PROBE(n) do { if (a.clk) ts[n] = wall_clock64(); } while (0)
#else
#define LB_PROBE(n) do { } while (0)
#endif
    LB_PROBE(0);
    const int hmask = a.hist_cap - 1;        // the ring capacity is a power of two (checked on the host): no runtime modulo
    const gem_lbfgs_opts& o = a.o;
    const int Dp = a.Dp;
    const size_t off = (size_t)b * Dp;

    int n_iter = sp->n_iter, evals = sp->evals, ls_iter = sp->ls_iter, ls_evals = sp->ls_evals, max_ls = sp->max_ls;
    int first_bracket = sp->first_bracket, ls_done = sp->ls_done, insuf = sp->insuf, low = sp->low, high = sp->high;
    int hist_count = sp->hist_count, hist_start = sp->hist_start;
    double loss = sp->loss, prev_loss = sp->prev_loss, t = sp->t, gtd = sp->gtd, d_norm = sp->d_norm, H_diag = sp->H_diag;
    double t_prev = sp->t_prev, f_prev = sp->f_prev, gtd_prev = sp->gtd_prev;
    double br_t[2] = {sp->br_t[0], sp->br_t[1]}, br_f[2] = {sp->br_f[0], sp->br_f[1]};
    double br_gtd[2] = {sp->br_gtd[0], sp->br_gtd[1]};
    static_assert(MAX_HIST <= NT, "one ring scalar per thread");

    if (a.trace && tid == 0) a.trace[b] = f_new;
    if (f_new != f_new && tid == 0) sp->nan_seen = 1;
    const float* gsrc = a.gnew + (size_t)gslot * Dp;
    float gn[EPT], xv[EPT], dv[EPT], gcur[EPT], yv[EPT], sv[EPT];
    bool have_x = false, have_d = false;
#pragma unroll
    for (int i = 0; i < EPT; ++i) xv[i] = dv[i] = gcur[i] = yv[i] = sv[i] = 0.f;
    // element i of a thread's strip: 16-byte groups when the strip allows it (Dp is a multiple of 64)
    auto load = [&](const float* p, float (&v)[EPT]) {
        if constexpr (EPT % 4 == 0) {
#pragma unroll
            for (int i = 0; i < EPT / 4; ++i) {
                const int e = (tid + NT * i) * 4;
                f32x4 w = {0.f, 0.f, 0.f, 0.f};
                if (FULL || e < Dp) w = *reinterpret_cast<const f32x4*>(p + off + e);
                v[4 * i] = w[0]; v[4 * i + 1] = w[1]; v[4 * i + 2] = w[2]; v[4 * i + 3] = w[3];
            }
        } else {
#pragma unroll
            for (int i = 0; i < EPT; ++i) { const int e = tid + NT * i; v[i] = (FULL || e < Dp) ? p[off + e] : 0.f; }
        }
    };
    auto store = [&](float* p, const float (&v)[EPT]) {
        if constexpr (EPT % 4 == 0) {
#pragma unroll
            for (int i = 0; i < EPT / 4; ++i) {
                const int e = (tid + NT * i) * 4;
                const f32x4 w = {v[4 * i], v[4 * i + 1], v[4 * i + 2], v[4 * i + 3]};
                if (FULL || e < Dp) *reinterpret_cast<f32x4*>(p + off + e) = w;
            }
        } else {
#pragma unroll
            for (int i = 0; i < EPT; ++i) { const int e = tid + NT * i; if (FULL || e < Dp) p[off + e] = v[i]; }
        }
    };
    // ring accessors: vector `slot` of this window in the ring `base` (a.S or a.Y); with HB the ring holds 2-byte values
    auto hload = [&](const float* base, int slot, float (&v)[EPT]) {
        const size_t eo = ((size_t)b * a.hist_cap + slot) * Dp;
        if constexpr (!HB) { load(base + eo - off, v); }
        else {
            const uint16_t* q = reinterpret_cast<const uint16_t*>(base) + eo;
            if constexpr (EPT % 4 == 0) {
#pragma unroll
                for (int i = 0; i < EPT / 4; ++i) {
                    const int e = (tid + NT * i) * 4;
                    typedef unsigned int u2 __attribute__((ext_vector_type(2)));
                    u2 w = {0u, 0u};
                    if (FULL || e < Dp) w = *reinterpret_cast<const u2*>(q + e);
                    v[4 * i] = __builtin_bit_cast(float, w[0] << 16); v[4 * i + 1] = __builtin_bit_cast(float, w[0] & 0xFFFF0000u);
                    v[4 * i + 2] = __builtin_bit_cast(float, w[1] << 16); v[4 * i + 3] = __builtin_bit_cast(float, w[1] & 0xFFFF0000u);
                }
            } else {
#pragma unroll
                for (int i = 0; i < EPT; ++i) { const int e = tid + NT * i; v[i] = (FULL || e < Dp) ? __builtin_bit_cast(float, (unsigned)q[e] << 16) : 0.f; }
            }
        }
    };
    auto hstore = [&](float* base, int slot, const float (&v)[EPT]) {
        const size_t eo = ((size_t)b * a.hist_cap + slot) * Dp;
        if constexpr (!HB) { store(base + eo - off, v); }
        else {
            uint16_t* q = reinterpret_cast<uint16_t*>(base) + eo;
            if constexpr (EPT % 4 == 0) {
#pragma unroll
                for (int i = 0; i < EPT / 4; ++i) {
                    const int e = (tid + NT * i) * 4;
                    typedef __bf16 bf4 __attribute__((ext_vector_type(4)));
                    const bf4 w = __builtin_convertvector(f32x4{v[4 * i], v[4 * i + 1], v[4 * i + 2], v[4 * i + 3]}, bf4);
                    if (FULL || e < Dp) *reinterpret_cast<bf4*>(q + e) = w;
                }
            } else {
#pragma unroll
                for (int i = 0; i < EPT; ++i) { const int e = tid + NT * i; if (FULL || e < Dp) q[e] = (uint16_t)(__builtin_bit_cast(unsigned int, (float)(__bf16)v[i]) >> 16); }
            }
        }
    };
    auto store_trial = [&](const float (&v)[EPT]) {       // the next point to evaluate: fp32, or bf16 in the bf16 decoder mode (whose
        if (!a.trial_b) store(a.trial, v);                // products read only the bf16 copy: 8 KB per window and round not written)
        if (a.trial_b) {
            uint16_t* p = a.trial_b + off;
            if constexpr (EPT % 4 == 0) {
#pragma unroll
                for (int i = 0; i < EPT / 4; ++i) {
                    const int e = (tid + NT * i) * 4;
                    typedef __bf16 bf4 __attribute__((ext_vector_type(4)));
                    const bf4 w = __builtin_convertvector(f32x4{v[4 * i], v[4 * i + 1], v[4 * i + 2], v[4 * i + 3]}, bf4);
                    if (FULL || e < Dp) *reinterpret_cast<bf4*>(p + e) = w;
                }
            } else {
#pragma unroll
                for (int i = 0; i < EPT; ++i) {
                    const int e = tid + NT * i;
                    if (FULL || e < Dp) p[e] = (uint16_t)(__builtin_bit_cast(unsigned int, (float)(__bf16)v[i]) >> 16);
                }
            }
        }
    };
    auto copy = [&](float* dst, const float* src) {
        float tmp[EPT];
        load(src, tmp);
        store(dst, tmp);
    };
    auto dot = [&](const float (&u)[EPT], const float (&v)[EPT]) {
        float p = 0.f;
#pragma unroll
        for (int i = 0; i < EPT; ++i) p += u[i] * v[i];
        return R.sum((double)p);
    };
    auto maxabs = [&](const float (&u)[EPT], float scale) {
        float p = 0.f;
#pragma unroll
        for (int i = 0; i < EPT; ++i) { const float w = fabsf(u[i] * scale); p = (w > p || w != w) ? w : p; }
        return R.max((double)p);
    };
    LB_PROBE(1);          // 1: state read
    // d and x are needed on every path but the very first evaluation, the ring's scalars by the two-loop recursion: requested
    // behind the gradient's first loads, in the same round trip
    double ro_v = 0.0, cadj_v = 0.0;
    auto request_rest = [&]() {
        if (tid < a.hist_cap) { ro_v = sp->ro[tid]; cadj_v = sp->cadj[tid]; }
        if (phase != PH_INIT) { load(a.d, dv); load(a.x, xv); have_d = have_x = true; }
    };
    if (a.gslab.base) {
        int nslab;
        size_t stride;
        slab_layout(a.gslab, nslab, stride);
        const float* gs = a.gslab.base + (size_t)gslot * Dp;
#pragma unroll
        for (int i = 0; i < EPT; ++i) gn[i] = 0.f;
        for (int z0 = 0; z0 < nslab; z0 += 8) {                 // eight slabs in flight per trip (the usual cut is 8)
            float t[8][EPT];
#pragma unroll
            for (int k = 0; k < 8; ++k) load(gs + (size_t)min(z0 + k, nslab - 1) * stride - off, t[k]);
            if (z0 == 0) request_rest();
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (z0 + k < nslab) {
#pragma unroll
                    for (int i = 0; i < EPT; ++i) gn[i] += t[k][i];
                }
        }
        if (nslab <= 0) request_rest();
    } else {
        load(gsrc - off, gn);
        request_rest();
    }
    if (tid < a.hist_cap) { ro_s[tid] = ro_v; cadj_s[tid] = cadj_v; }
    __syncthreads();
    float* BG[2] = {a.bg0, a.bg1};

    bool do_zoom_head = false, do_ls_end = false, do_start_iter = false, finished = false, emit = false;
    bool have_ys = false;

    LB_PROBE(2);          // 2: gradient (+ d, x) loads issued
    if (phase == PH_INIT) {
        loss = f_new;
        evals = 1;
        n_iter = 0;
        store(a.g, gn);
#pragma unroll
        for (int i = 0; i < EPT; ++i) gcur[i] = gn[i];
        if (maxabs(gn, 1.f) <= o.tol_grad) finished = true;
        else do_start_iter = true;
    } else {
        const double gtd_new = dot(gn, dv);
        if (phase == PH_BRACKET) {
            if (!first_bracket) ls_iter++;
            first_bracket = 0;
            bool enter_zoom = false;
            if (ls_iter < max_ls) {
                const bool c1 = f_new > (loss + o.c1 * t * gtd) || (ls_iter > 1 && f_new >= f_prev);
                const bool c2 = fabs(gtd_new) <= -o.c2 * gtd;
                const bool c3 = gtd_new >= 0.0;
                if (c1 || (!c2 && c3)) {
                    br_t[0] = t_prev; br_t[1] = t;
                    br_f[0] = f_prev; br_f[1] = f_new;
                    br_gtd[0] = gtd_prev; br_gtd[1] = gtd_new;
                    copy(a.bg0, a.gp);
                    store(a.bg1, gn);
                    ls_done = 0;
                    enter_zoom = true;
                } else if (c2) {
                    br_t[0] = br_t[1] = t;
                    br_f[0] = br_f[1] = f_new;
                    br_gtd[0] = br_gtd[1] = gtd_new;
                    store(a.bg0, gn);
                    ls_done = 1;
                    enter_zoom = true;
                } else {
                    const double lo_b = t + 0.01 * (t - t_prev), hi_b = t * 10.0;
                    const double t_next = cubic_interpolate(t_prev, f_prev, gtd_prev, t, f_new, gtd_new, true, lo_b, hi_b);
                    t_prev = t; f_prev = f_new; gtd_prev = gtd_new;
                    store(a.gp, gn);
                    t = t_next;
                    emit = true;
                }
            } else {   // ran out of line-search iterations while bracketing: bracket = [0, t]
                br_t[0] = 0.0; br_t[1] = t;
                br_f[0] = loss; br_f[1] = f_new;
                br_gtd[0] = gtd; br_gtd[1] = gtd_new;
                copy(a.bg0, a.g);
                store(a.bg1, gn);
                ls_done = 0;
                enter_zoom = true;
            }
            if (enter_zoom) {
                insuf = 0;
                if (br_f[0] <= br_f[1]) { low = 0; high = 1; } else { low = 1; high = 0; }
                do_zoom_head = true;
            }
        } else {   // PH_ZOOM
            ls_iter++;
            if (f_new > (loss + o.c1 * t * gtd) || f_new >= br_f[low]) {
                br_t[high] = t; br_f[high] = f_new; br_gtd[high] = gtd_new;
                store(BG[high], gn);
                if (br_f[0] <= br_f[1]) { low = 0; high = 1; } else { low = 1; high = 0; }
            } else {
                if (fabs(gtd_new) <= -o.c2 * gtd) {
                    ls_done = 1;
                } else if (gtd_new * (br_t[high] - br_t[low]) >= 0.0) {
                    br_t[high] = br_t[low]; br_f[high] = br_f[low]; br_gtd[high] = br_gtd[low];
                    copy(BG[high], BG[low]);
                }
                br_t[low] = t; br_f[low] = f_new; br_gtd[low] = gtd_new;
                store(BG[low], gn);
            }
            do_zoom_head = true;
        }
    }

    LB_PROBE(3);          // 3: g.d reduced, bracket / zoom update
    if (do_zoom_head) {
        if (ls_done || ls_iter >= max_ls) {
            do_ls_end = true;
        } else if (fabs(br_t[1] - br_t[0]) * d_norm < o.ls_tol_change) {
            do_ls_end = true;
        } else {
            double tn = cubic_interpolate(br_t[0], br_f[0], br_gtd[0], br_t[1], br_f[1], br_gtd[1], false, 0.0, 0.0);
            const double hi = br_t[0] > br_t[1] ? br_t[0] : br_t[1];
            const double lo = br_t[0] > br_t[1] ? br_t[1] : br_t[0];
            const double eps = 0.1 * (hi - lo);
            const double m1 = hi - tn, m2 = tn - lo;
            if ((m1 < m2 ? m1 : m2) < eps) {
                if (insuf || tn >= hi || tn <= lo) {
                    tn = (fabs(tn - hi) < fabs(tn - lo)) ? hi - eps : lo + eps;
                    insuf = 0;
                } else {
                    insuf = 1;
                }
            } else {
                insuf = 0;
            }
            t = tn;
            phase = PH_ZOOM;
            emit = true;
        }
    }

    LB_PROBE(4);          // 4: zoom head
    if (do_ls_end) {
        __syncthreads();                      // bracket gradients written above are read back below
        t = br_t[low];
        loss = br_f[low];
        float bl[EPT], gold[EPT];
        load(BG[low], bl);
        load(a.g, gold);
        const float tf = (float)t;
#pragma unroll
        for (int i = 0; i < EPT; ++i) {
            xv[i] = xv[i] + tf * dv[i];
            yv[i] = bl[i] - gold[i];
            sv[i] = dv[i] * tf;
            gcur[i] = bl[i];
        }
        store(a.x, xv);
        store(a.g, bl);
        evals += ls_evals;
        const double gmax = maxabs(bl, 1.f);
        const double dmax = maxabs(sv, 1.f);
        if (n_iter == o.max_iter || evals >= o.max_eval || gmax <= o.tol_grad || dmax <= o.tol_change ||
            fabs(loss - prev_loss) < o.tol_change) {
            finished = true;
        } else {
            do_start_iter = true;
            have_ys = true;
        }
    }

    LB_PROBE(5);          // 5: end of the line search (x, g stored, two max-norms)
    bool full_path = false;
    if (do_start_iter) {
        n_iter++;
        if (n_iter == 1) {
#pragma unroll
            for (int i = 0; i < EPT; ++i) dv[i] = -gcur[i];
            H_diag = 1.0;
        } else {
            (void)have_ys;
            auto ring_slot = [&](int k) { return (hist_start + k) & hmask; };      // k-th oldest stored pair -> ring slot
            // y.s, y.y and (for the pairwise two-loop below) s_prev.y of the newest stored pair, in ONE reduction
            float sprev[EPT];
            const bool has_prev = hist_count >= 1;
            if (has_prev) hload(a.S, ring_slot(hist_count - 1), sprev);
            double r3[3] = {0.0, 0.0, 0.0};
            {
                float p0 = 0.f, p1 = 0.f, p2 = 0.f;
#pragma unroll
                for (int i = 0; i < EPT; ++i) {
                    p0 += yv[i] * sv[i];
                    p1 += yv[i] * yv[i];
                    if (has_prev) p2 += sprev[i] * yv[i];
                }
                r3[0] = (double)p0; r3[1] = (double)p1; r3[2] = (double)p2;
            }
            R.template sumN<3>(r3);
            full_path = true;
            LB_PROBE(6);          // 6: y.s, y.y, s_prev.y
            const double ys = r3[0];
            const int limit = o.history < a.hist_cap ? o.history : a.hist_cap;
            if (ys > 1e-10) {
                const int prev_slot = (hist_start + hist_count - 1) & hmask;      // (the newest pair keeps its slot when the oldest goes)
                if (hist_count == limit) {          // shift history by one (limited memory)
                    hist_start = (hist_start + 1) & hmask;
                    hist_count--;
                }
                const int slot = (hist_start + hist_count) & hmask;
                hstore(a.Y, slot, yv);
                hstore(a.S, slot, sv);
                const bool link = has_prev && hist_count >= 1;                     // a previous pair survives: record s_prev . y_new
                hist_count++;
                H_diag = ys / r3[1];
                if (tid == 0) {
                    ro_s[slot] = 1.0 / ys; sp->ro[slot] = 1.0 / ys;
                    if (link) { cadj_s[prev_slot] = r3[2]; sp->cadj[prev_slot] = r3[2]; }
                }
                __syncthreads();
            }
            float q[EPT];
#pragma unroll
            for (int i = 0; i < EPT; ++i) q[i] = -gcur[i];
            // Two-loop recursion, TWO pairs per reduction: with c = s_B . y_A known (A the next newer pair of B),
            //   alpha_A = ro_A (s_A . q),  alpha_B = ro_B (s_B . (q - alpha_A y_A)) = ro_B (s_B . q - alpha_A c)
            // needs both dot products against the SAME q, so they share one barrier (and their DPP chains overlap);
            // the second loop likewise (y_B . (d + cf_A s_A) = y_B . d + cf_A c).  Same arithmetic as torch's loop up to
            // rounding.  Two pairs are in flight while two are being used (sets X and Z).
            const int hc = hist_count, np2 = (hc + 1) / 2;
            float sXa[EPT], yXa[EPT], sXb[EPT], yXb[EPT], sZa[EPT], yZa[EPT], sZb[EPT], yZb[EPT];
            auto clampk = [&](int k) { return k < 0 ? 0 : (k >= hc ? hc - 1 : k); };
            auto ld2 = [&](int kA, int kB, float (&sA)[EPT], float (&yA)[EPT], float (&sB)[EPT], float (&yB)[EPT]) {
                hload(a.S, ring_slot(clampk(kA)), sA); hload(a.Y, ring_slot(clampk(kA)), yA);
                hload(a.S, ring_slot(clampk(kB)), sB); hload(a.Y, ring_slot(clampk(kB)), yB);
            };
            auto dstep1 = [&](int kA, float (&sA)[EPT], float (&yA)[EPT], float (&sB)[EPT], float (&yB)[EPT]) {
                const int kB = kA - 1;
                const bool vB = kB >= 0;
                const int slA = (hist_start + kA) & hmask, slB = (hist_start + clampk(kB)) & hmask;
                double r2[2];
                {
                    float pa = 0.f, pb = 0.f;
#pragma unroll
                    for (int i = 0; i < EPT; ++i) { pa += sA[i] * q[i]; pb += sB[i] * q[i]; }
                    r2[0] = (double)pa; r2[1] = (double)pb;
                }
                R.template sumN<2>(r2);
                const double alA = r2[0] * ro_s[slA];
                const double alB = vB ? (r2[1] - (double)(float)alA * cadj_s[slB]) * ro_s[slB] : 0.0;
                if (tid == 0) { al_s[kA] = alA; if (vB) al_s[kB] = alB; }
                const float fa = (float)alA, fb = (float)alB;
#pragma unroll
                for (int i = 0; i < EPT; ++i) q[i] -= fa * yA[i] + fb * yB[i];
            };
            auto dstep2 = [&](int kA, float (&sA)[EPT], float (&yA)[EPT], float (&sB)[EPT], float (&yB)[EPT]) {
                const int kB = kA + 1;
                const bool vB = kB < hc;
                const int slA = (hist_start + kA) & hmask, slB = (hist_start + clampk(kB)) & hmask;
                double r2[2];
                {
                    float pa = 0.f, pb = 0.f;
#pragma unroll
                    for (int i = 0; i < EPT; ++i) { pa += yA[i] * q[i]; pb += yB[i] * q[i]; }
                    r2[0] = (double)pa; r2[1] = (double)pb;
                }
                R.template sumN<2>(r2);
                const float cfA = (float)(al_s[kA] - r2[0] * ro_s[slA]);
                const float cfB = vB ? (float)(al_s[clampk(kB)] - (r2[1] + (double)cfA * cadj_s[slA]) * ro_s[slB]) : 0.f;
#pragma unroll
                for (int i = 0; i < EPT; ++i) q[i] += cfA * sA[i] + cfB * sB[i];
            };
            if (hc > 0) ld2(hc - 1, hc - 2, sXa, yXa, sXb, yXb);
            for (int p2 = 0; p2 < np2; p2 += 2) {
                const int kA = hc - 1 - 2 * p2;
                ld2(kA - 2, kA - 3, sZa, yZa, sZb, yZb);
                dstep1(kA, sXa, yXa, sXb, yXb);
                ld2(kA - 4, kA - 5, sXa, yXa, sXb, yXb);
                if (p2 + 1 < np2) dstep1(kA - 2, sZa, yZa, sZb, yZb);
            }
            __syncthreads();
            const float hd = (float)H_diag;
#pragma unroll
            for (int i = 0; i < EPT; ++i) q[i] *= hd;
            if (hc > 0) ld2(0, 1, sXa, yXa, sXb, yXb);
            for (int p2 = 0; p2 < np2; p2 += 2) {
                const int kA = 2 * p2;
                ld2(kA + 2, kA + 3, sZa, yZa, sZb, yZb);
                dstep2(kA, sXa, yXa, sXb, yXb);
                ld2(kA + 4, kA + 5, sXa, yXa, sXb, yXb);
                if (p2 + 1 < np2) dstep2(kA + 2, sZa, yZa, sZb, yZb);
            }
#pragma unroll
            for (int i = 0; i < EPT; ++i) dv[i] = q[i];
        }
        LB_PROBE(7);          // 7: the two loops
        have_d = true;
        store(a.d, dv);
        prev_loss = loss;
        if (n_iter == 1) {
            float p = 0.f;
#pragma unroll
            for (int i = 0; i < EPT; ++i) p += fabsf(gcur[i]);
            const double l1 = R.sum((double)p);
            const double inv = 1.0 / l1;
            t = (inv < 1.0 ? inv : 1.0) * o.lr;
        } else {
            t = o.lr;
        }
        gtd = dot(gcur, dv);
        if (gtd > -o.tol_change) {
            finished = true;
        } else {
            d_norm = maxabs(dv, 1.f);
            max_ls = o.max_eval - evals;
            ls_iter = 0;
            ls_evals = 0;
            t_prev = 0.0; f_prev = loss; gtd_prev = gtd;
            store(a.gp, gcur);
            phase = PH_BRACKET;
            first_bracket = 1;
            emit = true;
        }
    }

    LB_PROBE(8);          // 8: g.d, max|d|
    if (finished) {
        phase = PH_DONE;
        if (!have_x) load(a.x, xv);
        store_trial(xv);
    } else if (emit) {
        if (!have_x) load(a.x, xv);
        if (!have_d) load(a.d, dv);
        const float tf = (float)t;
        float tr[EPT];
#pragma unroll
        for (int i = 0; i < EPT; ++i) tr[i] = xv[i] + tf * dv[i];
        store_trial(tr);
        ls_evals++;
    }
    if (tid == 0) {
        if (a.next_count && phase != PH_DONE) {
            // the next round's compaction, done here: slots in arrival order (no kernel of a round depends on the order)
            const int slot = atomicAdd(a.next_count, 1);
            a.next_perm[slot] = b;
            a.next_slot_of[b] = slot;
        }
        if (a.phase_arr) a.phase_arr[b] = phase;
        sp->phase = phase; sp->n_iter = n_iter; sp->evals = evals; sp->ls_iter = ls_iter; sp->ls_evals = ls_evals;
        sp->max_ls = max_ls; sp->first_bracket = first_bracket; sp->ls_done = ls_done; sp->insuf = insuf;
        sp->low = low; sp->high = high; sp->hist_count = hist_count; sp->hist_start = hist_start;
        sp->loss = loss; sp->prev_loss = prev_loss; sp->t = t; sp->gtd = gtd; sp->d_norm = d_norm; sp->H_diag = H_diag;
        sp->t_prev = t_prev; sp->f_prev = f_prev; sp->gtd_prev = gtd_prev;
        sp->br_t[0] = br_t[0]; sp->br_t[1] = br_t[1]; sp->br_f[0] = br_f[0]; sp->br_f[1] = br_f[1];
        sp->br_gtd[0] = br_gtd[0]; sp->br_gtd[1] = br_gtd[1];
    }
    LB_PROBE(9);          // 9: trial point and state stored
#ifdef GEM_LB_PROBE
    if (a.clk && tid == 0 && full_path) {
        unsigned long long prev = ts[0];
#pragma unroll
        for (int i = 1; i < 10; ++i)
            if (ts[i]) { atomicAdd(a.clk + i, ts[i] - prev); prev = ts[i]; }
        atomicAdd(a.clk + 30, (unsigned long long)hist_count);
        atomicAdd(a.clk + 31, 1ull);
    }
#endif
    (void)full_path;
#undef LB_PROBE
}

// (Round 4 measured the bf16-ring variant forced to six workgroups per CU -- 78 VGPRs, 24 spilled; all 1536 windows of configs[2]
// resident at once instead of 1024: 168.7 k vs 169.5 k windows/s at 1536 windows, 293.6 k vs 295.0 k at 8192.  Residency is not what
// bounds this kernel; not kept.  Keeping the bf16 history vectors packed in registers until use: 101 -> 97 VGPRs, same four
// waves per SIMD, no gain; not kept either.)
template <int EPT, bool FULL, bool HB = false>
__global__ __launch_bounds__(256) void lbfgs_advance_kernel(AdvArgs a) { lbfgs_advance_body<EPT, 256, FULL, HB>(a); }

__global__ void lbfgs_init_kernel(LbfgsState* st, int* __restrict__ phase_arr, const float* __restrict__ trial, float* __restrict__ x, int B,
                                  int Dp) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < (size_t)B * Dp) x[i] = trial[i];
    if (i < (size_t)B) {
        phase_arr[i] = PH_INIT;
        LbfgsState* s = st + i;
        s->phase = PH_INIT; s->n_iter = 0; s->evals = 0; s->ls_iter = 0; s->ls_evals = 0; s->max_ls = 0;
        s->first_bracket = 0; s->ls_done = 0; s->insuf = 0; s->low = 0; s->high = 1; s->hist_count = 0; s->hist_start = 0;
        s->nan_seen = 0;
        s->loss = 0; s->prev_loss = 0; s->t = 0; s->gtd = 0; s->d_norm = 0; s->H_diag = 1;
        s->t_prev = 0; s->f_prev = 0; s->gtd_prev = 0;
        s->br_t[0] = s->br_t[1] = 0; s->br_f[0] = s->br_f[1] = 0; s->br_gtd[0] = s->br_gtd[1] = 0;
    }
}

__global__ void lbfgs_stats_kernel(const LbfgsState* st, gem_window_stats* out, int B) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B) return;
    out[i].n_iter = st[i].n_iter;
    out[i].func_evals = st[i].evals;
    out[i].final_loss = (float)st[i].loss;
    out[i].status = (st[i].phase == PH_DONE ? 1 : 0) | (st[i].nan_seen ? 2 : 0);
}

static AdvArgs make_args(gem_handle* h, const gem_lbfgs_opts& o) {
    Workspace& w = h->ws;
    AdvArgs a;
    a.state = w.state; a.f = w.f; a.gnew = w.dz;
    a.trace = (w.round >= 0 && w.round < TRACE_ROUNDS) ? w.trace + (size_t)w.round * w.Bmax : nullptr;
    a.x = w.x; a.d = w.d; a.g = w.g; a.gp = w.gp; a.bg0 = w.bg0; a.bg1 = w.bg1; a.trial = w.trial; a.S = w.S; a.Y = w.Y;
    a.trial_b = h->precision == GEM_PRECISION_BF16 ? w.trial_b : nullptr;
    a.phase_arr = w.phase;
    a.slot_of = w.dyn ? w.slot_of : nullptr;
    a.next_perm = w.dyn ? w.next_perm : nullptr; a.next_slot_of = w.dyn ? w.next_slot_of : nullptr; a.next_count = w.dyn ? w.next_count : nullptr;
    a.gslab = w.dyn ? w.grad_slab : SlabSrc{};
    a.clk = w.lbfgs_clk;
    a.Dp = h->Dp; a.hist_cap = w.hist_cap; a.o = o;        // hist_cap: power of two (gem_create)
    return a;
}

// ---- EXPERIMENT: one launch for the backward front product and the L-BFGS advance, a device-wide barrier in between -----------------
// Every workgroup runs its tile of dE/dz = dpre0 . Wf (split-K slabs, gemm_rows.h), publishes them (agent-scope release), arrives at
// the barrier (one atomic add on a monotonic counter), waits until all workgroups of the launch have arrived (bounded spin: a grid
// that is not co-resident must not hang the device), acquires, and advances window blockIdx.x.  The grid is one workgroup per CU
// (the product's LDS ring allows no second one), so co-residency holds on an otherwise idle chip.
template <int S, int RT>
__global__ __launch_bounds__(256) void rows_bwd_lbfgs_kernel(const rows::Args ra, AdvArgs la, int B, unsigned* bar, unsigned target) {
    rows::gemm_rows_body<S, RT, false>(ra);
    // (MI355X_MICROARCH.md "Correctness boundaries": every storing wave drains its stores, the workgroup meets, ONE lane releases at
    // agent scope -- the per-XCD L2s are not coherent with each other: that is an L2 write-back --, arrives, polls, acquires -- an
    // invalidate of this CU's L1 and its XCD's L2 --, the workgroup meets again, plain loads follow)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        while ((int)(__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0 && ++spins < (1u << 22))
            __builtin_amdgcn_s_sleep(1);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    if ((int)blockIdx.x < B) lbfgs_advance_body<8, 256, true, false>(la);
}

int launch_rows_bwd_lbfgs(gem_handle* h, const rows::Args& ra, int rows_grid, size_t rows_smem, const SlabSrc& gslab, hipStream_t s) {
    Workspace& w = h->ws;
    if (!w.fuse_lbfgs || !w.grid_bar || h->Dp != 2048 || h->precision != GEM_PRECISION_F32 || ra.M > h->n_cu || rows_grid > h->n_cu) return -1;
    const int B = ra.M;
    w.grad_slab = gslab;
    AdvArgs la = make_args(h, *w.fuse_lbfgs);
    auto k = rows_bwd_lbfgs_kernel<4, 5>;
    static PerDeviceOnce once;
    if (once.need(h->cfg.device))
        GEM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)rows_smem));
    const int grid = rows_grid > B ? rows_grid : B;
    w.grid_bar_target += (unsigned)grid;
    note_kernel(h, reinterpret_cast<const void*>(k));
    hipLaunchKernelGGL(k, dim3(grid), dim3(256), rows_smem, s, ra, la, B, w.grid_bar, w.grid_bar_target);
    GEM_HIP(hipGetLastError());
    w.lbfgs_fused_done = true;
    return 0;
}

int launch_lbfgs_init(gem_handle* h, int B, const gem_lbfgs_opts& o, hipStream_t s) {
    (void)o;
    const size_t n = (size_t)B * h->Dp;
    hipLaunchKernelGGL(lbfgs_init_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, h->ws.state, h->ws.phase, h->ws.trial, h->ws.x,
                       B, h->Dp);
    GEM_HIP(hipGetLastError());
    return 0;
}

int launch_lbfgs_advance(gem_handle* h, int B, const gem_lbfgs_opts& o, hipStream_t s) {
    AdvArgs a = make_args(h, o);
    Profile::Rec rec;
    const bool prof = h->prof.on;
    if (prof) {
        GEM_HIP(hipEventCreate(&rec.a)); GEM_HIP(hipEventCreate(&rec.b));
        rec.family = 2; rec.flops = 0;
        GEM_HIP(hipEventRecord(rec.a, s));
    }
    const int ept = (h->Dp + 255) / 256;
    typedef void (*kern_t)(AdvArgs);
    kern_t kern = nullptr;
    if (ept <= 1) kern = lbfgs_advance_kernel<1, false>;
    else if (ept <= 2) kern = lbfgs_advance_kernel<2, false>;
    else if (ept <= 4) kern = lbfgs_advance_kernel<4, false>;
    else if (h->Dp == 2048 && h->precision == GEM_PRECISION_BF16) kern = lbfgs_advance_kernel<8, true, true>;
    else if (h->Dp == 2048) kern = lbfgs_advance_kernel<8, true>;       // the reference's latent size
    else if (ept <= 8) kern = lbfgs_advance_kernel<8, false>;
    else if (ept <= 16) kern = lbfgs_advance_kernel<16, false>;
    else { set_error("latent_dim > 4096 is not supported by the L-BFGS kernel"); return 1; }
    note_kernel(h, reinterpret_cast<const void*>(kern));
    hipLaunchKernelGGL(kern, dim3(B), dim3(256), 0, s, a);
    GEM_HIP(hipGetLastError());
    if (prof) { GEM_HIP(hipEventRecord(rec.b, s)); h->prof.recs.push_back(rec); }
    commit_kernel_names(h, prof ? 2 : -1);
    return 0;
}

// Stable partition of the windows: those still iterating first (slots [0, n)), finished ones behind.
// One 1024-thread block; B is at most a few thousand.
// zero_after: that many n_log entries BEHIND log_slot are zeroed here (the per-round counters of a stage whose rounds hand out their
// slots atomically, gem_api.hip stage_begin) -- by this kernel rather than by a hipMemsetAsync node, so that inside a captured graph
// the zeroing is ordered like every other kernel of the call
__global__ __launch_bounds__(1024) void compact_kernel(const int* __restrict__ phase_arr, int B, int T, int* __restrict__ perm,
                                                       int* __restrict__ slot_of, int* __restrict__ n_active, int force_all,
                                                       int* __restrict__ log_slot, int zero_after) {
    // One pass, two barriers (round 4; before: two passes over chunks of 1024 windows with three barriers each -- 14 us at 8192
    // windows): thread t owns the E consecutive windows [t E, (t + 1) E), E = ceil(B / 1024) <= 64.  ONE block scan of the
    // per-thread active counts places both groups: an active window goes to slot (#active before it), a finished one to
    // n_active + (#finished before it) = n_active + (its index - #active before it).  Same stable order as before.
    __shared__ int wsum[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int E = (B + 1023) >> 10, b0 = tid * E;
    unsigned long long mask = 0ull;
    for (int e = 0; e < E; ++e) {
        const int b = b0 + e;
        if (b < B && (force_all || phase_arr[b] != PH_DONE)) mask |= 1ull << e;
    }
    const int cnt = __popcll(mask);
    int incl = cnt;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int v = __shfl_up(incl, d);
        if (lane >= d) incl += v;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int woff = 0, total = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) { if (i < wave) woff += wsum[i]; total += wsum[i]; }
    int pa = woff + incl - cnt;               // active windows before this thread's range
    int pi = total + (b0 - pa);               // slot of the first finished window of the range
    for (int e = 0; e < E; ++e) {
        const int b = b0 + e;
        if (b >= B) break;
        if ((mask >> e) & 1ull) { perm[pa] = b; slot_of[b] = pa++; }
        else { perm[pi] = b; slot_of[b] = pi++; }
    }
    if (tid == 0) { n_active[0] = total; n_active[1] = total * T; *log_slot = total; }
    if (tid < zero_after) log_slot[1 + tid] = 0;
}

int launch_compact(gem_handle* h, int B, int force_all, hipStream_t s, int zero_after) {
    Workspace& w = h->ws;
    if (B > 65536) { set_error("compact: more than 65536 windows per call"); return 1; }      // (64 windows per thread of the scan)
    if (zero_after < 0 || zero_after > 1024 || (w.log_pos % N_LOG) + zero_after >= N_LOG) { set_error("compact: bad counter range"); return 1; }
    hipLaunchKernelGGL(compact_kernel, dim3(1), dim3(1024), 0, s, w.phase, B, h->T, w.perm, w.slot_of, w.n_active, force_all,
                       w.n_log + (w.log_pos % N_LOG), zero_after);
    GEM_HIP(hipGetLastError());
    w.cur_log = w.log_pos++;
    return 0;
}

int launch_lbfgs_stats(gem_handle* h, int B, gem_window_stats* out, hipStream_t s) {
    hipLaunchKernelGGL(lbfgs_stats_kernel, dim3((B + 255) / 256), dim3(256), 0, s, h->ws.state, out, B);
    GEM_HIP(hipGetLastError());
    return 0;
}

}  // namespace gem
