// C-ABI entry points of libgem_hip.so (see include/gem_hip.h) and the host-side orchestration of the
// evaluation rounds.  Host code only: weight folding / packing, workspace management and kernel
// sequencing; all arithmetic of the path runs in the HIP kernels of gemm_f32.hip, energy.hip, lbfgs.hip.
#include <cxxabi.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>

#include "gem_internal.h"

namespace gem {

static thread_local std::string g_error;
void set_error(const std::string& msg) { g_error = msg; }
bool hip_ok(hipError_t e, const char* what) {
    if (e == hipSuccess) return true;
    g_error = std::string(what) + ": " + hipGetErrorString(e);
    return false;
}

const char* dev_env(const char* name) {
    const char* on = getenv("GEM_DEV");
    if (!on || on[0] != '1') return nullptr;
    return getenv(name);
}

void note_kernel(gem_handle* h, const void* host_fn) {
    if (!h->prof.on) return;
    const char* m = hipKernelNameRefByPtr(host_fn, nullptr);
    if (!m) return;
    int st = 0;
    char* d = abi::__cxa_demangle(m, nullptr, nullptr, &st);
    std::string n = (st == 0 && d) ? d : m;
    free(d);
    if (n.rfind("void ", 0) == 0) n = n.substr(5);
    // drop the parameter list: everything from the '(' that closes the template-argument list
    int depth = 0;
    for (size_t i = 0; i < n.size(); ++i) {
        if (n[i] == '<') ++depth;
        else if (n[i] == '>') --depth;
        else if (n[i] == '(' && depth == 0) { n.resize(i); break; }
    }
    h->prof.pending.insert(n);
}

// Every captured call bakes weight and workspace pointers into its kernel arguments: whatever re-allocates them drops the cache.
static void drop_graphs(gem_handle* h) {
    for (auto& g : h->graphs) { if (g.exec) (void)hipGraphExecDestroy(g.exec); if (g.graph) (void)hipGraphDestroy(g.graph); }
    h->graphs.clear();
}

__global__ void fill_u32_kernel(uint32_t* __restrict__ p, uint32_t v, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}
static int launch_fill_u32(uint32_t* p, uint32_t v, size_t n, hipStream_t s) {
    if (!n) return 0;
    hipLaunchKernelGGL(fill_u32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, p, v, n);
    GEM_HIP(hipGetLastError());
    return 0;
}

template <typename T>
static int dev_alloc(std::vector<void*>& owner, T** p, size_t n) {
    void* q = nullptr;
    GEM_HIP(hipMalloc(&q, (n ? n : 1) * sizeof(T)));
    owner.push_back(q);                 // owned from here on, whatever happens next
    *p = static_cast<T*>(q);
    GEM_HIP(hipMemset(q, 0, (n ? n : 1) * sizeof(T)));
    return 0;
}
static void free_all(std::vector<void*>& owner) {
    for (void* p : owner) (void)hipFree(p);
    owner.clear();
}

template <typename T>
static int upload(std::vector<void*>& owner, T** p, const std::vector<T>& v) {
    if (dev_alloc(owner, p, v.size())) return 1;
    GEM_HIP(hipMemcpy(*p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    return 0;
}

static uint16_t host_f2bf(float x) {
    uint32_t u;
    std::memcpy(&u, &x, 4);
    return (uint16_t)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
}
static float host_bf2f(uint16_t b) {
    const uint32_t u = (uint32_t)b << 16;
    float f;
    std::memcpy(&f, &u, 4);
    return f;
}
// bf16 hi / lo images of a packed fp32 weight array (same [taps][N][K] layout)
static int upload_bf16(std::vector<void*>& owner, Layer* L, const std::vector<float>& w) {
    std::vector<uint16_t> hi(w.size()), lo(w.size());
    for (size_t i = 0; i < w.size(); ++i) {
        hi[i] = host_f2bf(w[i]);
        lo[i] = host_f2bf(w[i] - host_bf2f(hi[i]));
    }
    return upload(owner, &L->wb_hi, hi) || upload(owner, &L->wb_lo, lo);
}

// taps[k][ci][co] in double, BatchNorm folded
struct FoldedConv {
    int ci, co;
    std::vector<double> taps, bias;
};

static FoldedConv fold_conv(const float* w, const float* b, const float* bn /* 4 blobs or null */, const float* const* bnp,
                            int ci, int co, bool transposed) {
    (void)bn;
    FoldedConv f;
    f.ci = ci; f.co = co;
    f.taps.assign((size_t)3 * ci * co, 0.0);
    f.bias.assign(co, 0.0);
    for (int k = 0; k < 3; ++k)
        for (int i = 0; i < ci; ++i)
            for (int o = 0; o < co; ++o) {
                // Conv1d weight [co][ci][3]: out[t] = sum_k in[t+k-1] w[:, :, k]
                // ConvTranspose1d (s=1,p=1) weight [ci][co][3]: out[t] = sum_k in[t+1-k] w[:, :, k]  -> tap k' = 2-k
                const double v = transposed ? (double)w[((size_t)i * co + o) * 3 + (2 - k)] : (double)w[((size_t)o * ci + i) * 3 + k];
                f.taps[((size_t)k * ci + i) * co + o] = v;
            }
    for (int o = 0; o < co; ++o) f.bias[o] = b[o];
    if (bnp) {
        const float *gamma = bnp[0], *beta = bnp[1], *mean = bnp[2], *var = bnp[3];
        for (int o = 0; o < co; ++o) {
            const double s = (double)gamma[o] / std::sqrt((double)var[o] + BN_EPS);
            for (int k = 0; k < 3; ++k)
                for (int i = 0; i < ci; ++i) f.taps[((size_t)k * ci + i) * co + o] *= s;
            f.bias[o] = (f.bias[o] - (double)mean[o]) * s + (double)beta[o];
        }
    }
    return f;
}

static int make_conv_layers(StageNet& net, const FoldedConv& f, Layer* fwd, Layer* bwd, std::vector<float>* keep_fwd = nullptr,
                            std::vector<float>* keep_bwd = nullptr) {
    const int Kp = pad64(f.ci), Np = pad64(f.co);
    std::vector<float> wf((size_t)3 * Np * Kp, 0.f), bf(Np, 0.f);
    for (int k = 0; k < 3; ++k)
        for (int i = 0; i < f.ci; ++i)
            for (int o = 0; o < f.co; ++o) wf[((size_t)k * Np + o) * Kp + i] = (float)f.taps[((size_t)k * f.ci + i) * f.co + o];
    for (int o = 0; o < f.co; ++o) bf[o] = (float)f.bias[o];
    fwd->taps = 3; fwd->K = Kp; fwd->N = Np;
    if (upload(net.allocs, &fwd->w, wf) || upload(net.allocs, &fwd->bias, bf) || upload_bf16(net.allocs, fwd, wf)) return 1;
    auto to_w4 = [](const std::vector<float>& w, int N, int K) {      // [tap][N][K] -> [tap][K/4][N][4]
        std::vector<float> o(w.size());
        for (int t = 0; t < 3; ++t)
            for (int n = 0; n < N; ++n)
                for (int k = 0; k < K; ++k) o[(((size_t)t * (K / 4) + k / 4) * N + n) * 4 + (k & 3)] = w[((size_t)t * N + n) * K + k];
        return o;
    };
    if (bwd && upload(net.allocs, &fwd->w4, to_w4(wf, Np, Kp))) return 1;
    if (bwd) {
        // adjoint: dIn[r] = sum_tap' dOut[r + tap' - 1] . taps[2-tap']^T   ->  W[tap'][n=ci][k=co]
        std::vector<float> wb((size_t)3 * Kp * Np, 0.f);
        for (int k = 0; k < 3; ++k)
            for (int i = 0; i < f.ci; ++i)
                for (int o = 0; o < f.co; ++o) wb[((size_t)k * Kp + i) * Np + o] = (float)f.taps[((size_t)(2 - k) * f.ci + i) * f.co + o];
        bwd->taps = 3; bwd->K = Np; bwd->N = Kp;
        if (upload(net.allocs, &bwd->w, wb) || upload(net.allocs, &bwd->w4, to_w4(wb, Kp, Np)) || upload_bf16(net.allocs, bwd, wb))
            return 1;
        bwd->bias = nullptr;
        if (keep_bwd) *keep_bwd = std::move(wb);
    }
    if (keep_fwd) *keep_fwd = std::move(wf);
    return 0;
}

// ---- decoder_input o conv 0 as ONE linear layer -----------------------------------------------------------------------------
// h0 = Wd z + bd (decoder_input, rows (t', ci)) feeds ConvTranspose1d 0 + BatchNorm with NO activation in between
// (SeqConvVAE.py:62, 67-75, 131-135), so
//     pre0[(t, co)] = sum_tap sum_ci taps[tap][ci][co] h0[(t + tap - 1, ci)] + bc[co]      (frames outside the window: zero)
//                   = (Wf z + bf)[(t, co)],   Wf[(t, co)][k] = sum_tap sum_ci taps[tap][ci][co] Wd[(t + tap - 1, ci)][k].
// Wf is [T*C1p, Dp]: 2 x 2048 x 2560 FLOP per window instead of 2 x 2048 x 5120 + 2 x 3 x 512 x 256 x 10 (a third of the
// matrix work of the two layers, half their weight bytes), one launch instead of two (three in the backward direction, where
// its transpose replaces the conv adjoint, the split-K reduce behind it and the decoder_input backward product).  Built once
// per gem_load_vae in fp64 from the fp64 folded conv taps and the fp32 decoder_input weights, rounded to fp32 once.
__global__ __launch_bounds__(256) void compose_front_kernel(const double* __restrict__ taps /* [3][ci][co] */, const float* __restrict__ Wd /* [T*Cip][Dp] */,
                                                            int T, int Ci, int Cip, int Co, int Cop, int Dp, float* __restrict__ Wf /* [T*Cop][Dp] */,
                                                            float* __restrict__ WfT /* [Dp][T*Cop] */) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    const int n = blockIdx.y, t = n / Cop, co = n - t * Cop;
    if (k >= Dp) return;
    double acc = 0.0;
    if (co < Co) {
        for (int tap = 0; tap < 3; ++tap) {
            const int ts = t + tap - 1;
            if (ts < 0 || ts >= T) continue;
            const double* tp = taps + (size_t)tap * Ci * Co + co;
            const float* wd = Wd + (size_t)ts * Cip * Dp + k;
            for (int ci = 0; ci < Ci; ++ci) acc += tp[(size_t)ci * Co] * (double)wd[(size_t)ci * Dp];
        }
    }
    Wf[(size_t)n * Dp + k] = (float)acc;
    WfT[(size_t)k * ((size_t)T * Cop) + n] = (float)acc;
}

static int compose_front(gem_handle* h, StageNet& net, const FoldedConv& f, const float* dec_in_bias_host /* time-major, padded */) {
    const int T = h->T, Dp = h->Dp, Cip = h->topp, Cop = pad64(f.co), Nf = T * Cop;
    double* d_taps = nullptr;
    std::vector<void*> tmp;
    if (upload(tmp, &d_taps, f.taps)) { free_all(tmp); return 1; }
    float *Wf = nullptr, *WfT = nullptr;
    if (dev_alloc(net.allocs, &Wf, (size_t)Nf * Dp) || dev_alloc(net.allocs, &WfT, (size_t)Dp * Nf)) { free_all(tmp); return 1; }
    hipLaunchKernelGGL(compose_front_kernel, dim3((Dp + 255) / 256, Nf), dim3(256), 0, 0, d_taps, net.dec_in.w, T, f.ci, Cip, f.co, Cop, Dp, Wf, WfT);
    if (!hip_ok(hipGetLastError(), "compose_front_kernel") || !hip_ok(hipDeviceSynchronize(), "compose_front_kernel")) { free_all(tmp); return 1; }
    free_all(tmp);
    std::vector<float> bf((size_t)Nf, 0.f), zb((size_t)Dp, 0.f);
    for (int t = 0; t < T; ++t)
        for (int co = 0; co < f.co; ++co) {
            double acc = f.bias[co];
            for (int tap = 0; tap < 3; ++tap) {
                const int ts = t + tap - 1;
                if (ts < 0 || ts >= T) continue;
                for (int ci = 0; ci < f.ci; ++ci) acc += f.taps[((size_t)tap * f.ci + ci) * f.co + co] * (double)dec_in_bias_host[(size_t)ts * Cip + ci];
            }
            bf[(size_t)t * Cop + co] = (float)acc;
        }
    net.front.taps = 1; net.front.K = Dp; net.front.N = Nf; net.front.w = Wf;
    net.front_bwd.taps = 1; net.front_bwd.K = Nf; net.front_bwd.N = Dp; net.front_bwd.w = WfT;
    // bf16 images for the bf16 decoder mode (rounded once from the fp64-composed weights)
    if (dev_alloc(net.allocs, &net.front.wb_hi, (size_t)Nf * Dp) || dev_alloc(net.allocs, &net.front_bwd.wb_hi, (size_t)Dp * Nf) ||
        dev_alloc(net.allocs, &net.front.wb_lo, (size_t)Nf * Dp) || dev_alloc(net.allocs, &net.front_bwd.wb_lo, (size_t)Dp * Nf) ||
        launch_f32_split_bf16(Wf, net.front.wb_hi, net.front.wb_lo, (size_t)Nf * Dp, nullptr) ||
        launch_f32_split_bf16(WfT, net.front_bwd.wb_hi, net.front_bwd.wb_lo, (size_t)Dp * Nf, nullptr))
        return 1;
    GEM_HIP(hipDeviceSynchronize());
    return upload(net.allocs, &net.front.bias, bf) || upload(net.allocs, &net.front_bwd.bias, zb);
}

}  // namespace gem

using namespace gem;

extern "C" {

const char* gem_last_error(void) { return g_error.c_str(); }
int gem_version(void) { return 1; }

int gem_create(const gem_config* cfg, gem_handle** out) {
    if (!cfg || !out) { set_error("gem_create: null argument"); return 1; }
    if (cfg->seq_len < 3 || cfg->seq_len > 16 || cfg->n_joints < 1 || cfg->n_joints > GEM_MAX_JOINTS) {
        set_error("gem_create: seq_len must be 3..16 and n_joints 1..16"); return 1;
    }
    if (cfg->n_joints * 3 > PAD) { set_error("gem_create: n_joints*3 must be <= 64"); return 1; }
    if (cfg->n_hidden < 1 || cfg->n_hidden > GEM_MAX_HIDDEN || cfg->latent_dim < 1 || cfg->latent_dim > 4096) {
        set_error("gem_create: n_hidden must be 1..8 and latent_dim 1..4096"); return 1;
    }
    if (cfg->n_poly < 1 || cfg->n_poly > GEM_MAX_POLY || cfg->max_windows < 1) { set_error("gem_create: bad n_poly / max_windows"); return 1; }
    GEM_HIP(hipSetDevice(cfg->device));
    std::unique_ptr<gem_handle, void (*)(gem_handle*)> h(new gem_handle(), gem_destroy);     // frees the device memory on any early return
    h->cfg = *cfg;
    h->T = cfg->seq_len; h->J = cfg->n_joints; h->C = cfg->n_joints * 3; h->Cp = pad64(h->C);
    h->D = cfg->latent_dim; h->Dp = pad64(cfg->latent_dim);
    h->top = cfg->hidden[cfg->n_hidden - 1]; h->topp = pad64(h->top);
    GEM_HIP(hipDeviceGetAttribute(&h->n_cu, hipDeviceAttributeMultiprocessorCount, cfg->device));
    if (const char* lm = dev_env("GEM_LANES_MIN")) h->lanes_min = atoi(lm);          // developer override of gem_set_lanes' default (A/B runs)
    Workspace& w = h->ws;
    const int B = cfg->max_windows, T = h->T;
    const size_t rows = (size_t)B * T;
    w.Bmax = B;
    if (dev_alloc(w.allocs, &w.pose_p, rows * PAD)) return 1;
    w.enc_act.resize(cfg->n_hidden);
    for (int i = 0; i < cfg->n_hidden; ++i)
        if (dev_alloc(w.allocs, &w.enc_act[i], rows * pad64(cfg->hidden[i]))) return 1;
    if (dev_alloc(w.allocs, &w.mulv, (size_t)B * 2 * h->Dp)) return 1;
    if (dev_alloc(w.allocs, &w.h0, rows * h->topp)) return 1;
    // decoder conv widths: reversed hidden, then hidden[0] again, then C
    std::vector<int> outs;
    for (int i = cfg->n_hidden - 2; i >= 0; --i) outs.push_back(cfg->hidden[i]);
    outs.push_back(cfg->hidden[0]);
    outs.push_back(h->C);
    w.dec_act.resize(outs.size());
    w.dec_grad.resize(outs.size());
    int cin = h->top;
    for (size_t i = 0; i < outs.size(); ++i) {
        if (dev_alloc(w.allocs, &w.dec_act[i], rows * pad64(outs[i]))) return 1;
        if (dev_alloc(w.allocs, &w.dec_grad[i], rows * pad64(cin))) return 1;
        cin = outs[i];
    }
    if (dev_alloc(w.allocs, &w.dXp, rows * PAD)) return 1;
    // bf16 twins (gemm_bf16a.h / decoder_bf16.hip)
    w.dec_act_b.assign(outs.size(), nullptr);
    w.dec_grad_b.assign(outs.size(), nullptr);
    {
        int ci = h->top;
        for (size_t i = 0; i < outs.size(); ++i) {
            if (i + 1 < outs.size() && dev_alloc(w.allocs, &w.dec_act_b[i], rows * pad64(outs[i]))) return 1;
            if (dev_alloc(w.allocs, &w.dec_grad_b[i], rows * pad64(ci))) return 1;
            ci = outs[i];
        }
    }
    if (dev_alloc(w.allocs, &w.trial_b, (size_t)B * h->Dp) || dev_alloc(w.allocs, &w.h0_b, rows * h->topp) ||
        dev_alloc(w.allocs, &w.dXp_b, rows * PAD) || dev_alloc(w.allocs, &w.zero16, (size_t)128)) return 1;
    if (dev_alloc(w.allocs, &w.dz, (size_t)B * h->Dp)) return 1;
    float** vecs[] = {&w.x, &w.d, &w.g, &w.gp, &w.bg0, &w.bg1, &w.trial};
    for (float** v : vecs)
        if (dev_alloc(w.allocs, v, (size_t)B * h->Dp)) return 1;
    w.hist_cap = 32;     // >= max_iter - 1 pairs for the reference's max_iter = 25 (checked per call); a power of two:
                         // lbfgs.hip wraps ring indices with a mask
    static_assert(MAX_HIST >= 32, "ring capacity");
    if (dev_env("GEM_LBFGS_CLK")) {
        if (dev_alloc(w.allocs, &w.lbfgs_clk, (size_t)32)) return 1;
        GEM_HIP(hipMemset(w.lbfgs_clk, 0, 32 * sizeof(unsigned long long)));
    }
    if (dev_alloc(w.allocs, &w.S, (size_t)B * w.hist_cap * h->Dp)) return 1;
    if (dev_alloc(w.allocs, &w.Y, (size_t)B * w.hist_cap * h->Dp)) return 1;
    if (dev_alloc(w.allocs, &w.state, (size_t)B) || dev_alloc(w.allocs, &w.phase, (size_t)B)) return 1;
    if (dev_alloc(w.allocs, &w.f, (size_t)B)) return 1;
    if (dev_alloc(w.allocs, &w.parts, (size_t)B * 5)) return 1;
    if (dev_alloc(w.allocs, &w.trace, (size_t)TRACE_ROUNDS * B)) return 1;
    if ((size_t)cfg->heat_h * cfg->heat_w <= 32768) {        // (the texel-block key packs two texel indices into 32 bits)
        if (dev_alloc(w.allocs, &w.tex_key, rows * h->J) || dev_alloc(w.allocs, &w.tex_val, rows * h->J * 4)) return 1;
    }
    if (dev_alloc(w.allocs, &w.pose_a, rows * h->C)) return 1;
    if (dev_alloc(w.allocs, &w.pose_b, rows * h->C)) return 1;
    if (dev_alloc(w.allocs, &w.n_log, (size_t)N_LOG)) return 1;
    if (dev_alloc(w.allocs, &w.perm2, (size_t)B) || dev_alloc(w.allocs, &w.slot_of2, (size_t)B)) return 1;
    if (dev_alloc(w.allocs, &w.grid_bar, 1)) return 1;
    GEM_HIP(hipMemset(w.grid_bar, 0, sizeof(unsigned)));
    if (dev_alloc(w.allocs, &w.perm, (size_t)B) || dev_alloc(w.allocs, &w.slot_of, (size_t)B) || dev_alloc(w.allocs, &w.n_active, 2))
        return 1;
    w.perm_home = w.perm; w.slot_of_home = w.slot_of; w.n_active_home = w.n_active;
    // split-K slabs: only launches with few output tiles cut K; 64 MB, more when mid-size batches need it for the
    // decoder_input backward product (rows x Dp x up to 4 slices)
    w.splitk_elems = std::max((size_t)16 << 20, (size_t)std::min(B, 4096) * h->Dp * 4);
    if (dev_alloc(w.allocs, &w.splitk, w.splitk_elems)) return 1;
    std::vector<int> parents(cfg->parents, cfg->parents + cfg->n_joints);
    std::vector<int> children((size_t)GEM_MAX_JOINTS * GEM_MAX_JOINTS, -1);
    for (int j = 0; j < cfg->n_joints; ++j) {
        if (parents[j] < 0 || parents[j] >= cfg->n_joints) { set_error("gem_create: bad parent index"); return 1; }
        int n = 0;
        for (int c = 0; c < cfg->n_joints; ++c)
            if (c != j && parents[c] == j) children[(size_t)j * GEM_MAX_JOINTS + n++] = c;
    }
    if (upload(w.allocs, &h->d_parents, parents) || upload(w.allocs, &h->d_children, children)) return 1;
    *out = h.release();
    return 0;
}

void gem_destroy(gem_handle* h) {
    if (!h) return;
    (void)hipSetDevice(h->cfg.device);
    (void)hipDeviceSynchronize();
    for (auto& r : h->prof.recs) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
    if (h->ws.lbfgs_clk) {
        unsigned long long c[32];
        if (hipMemcpy(c, h->ws.lbfgs_clk, sizeof(c), hipMemcpyDeviceToHost) == hipSuccess && c[31]) {
            fprintf(stderr, "[GEM_LBFGS_CLK] %llu window-rounds with a new direction from the ring, mean pairs %.1f; us per phase:", c[31], (double)c[30] / c[31]);
            double tot = 0;
            for (int i = 1; i < 10; ++i) { fprintf(stderr, " %d:%.2f", i, c[i] / 100.0 / c[31]); tot += c[i] / 100.0 / c[31]; }
            fprintf(stderr, " total %.2f\n", tot);
        }
    }
    drop_graphs(h);
    if (h->lane2) { gem_destroy(h->lane2); h->lane2 = nullptr; }          // (its nets own nothing: the weights are freed below)
    if (h->lane_stream) (void)hipStreamDestroy(h->lane_stream);
    if (h->lane_stream_a) (void)hipStreamDestroy(h->lane_stream_a);
    if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
    if (h->ev_join) (void)hipEventDestroy(h->ev_join);
    if (h->ev_join_a) (void)hipEventDestroy(h->ev_join_a);
    for (hipEvent_t e : h->ev_pool) (void)hipEventDestroy(e);
    free_all(h->net[0].allocs);
    free_all(h->net[1].allocs);
    free_all(h->ws.allocs);
    if (h->post_work) (void)hipFree(h->post_work);
    delete h;
}

int gem_load_vae(gem_handle* h, int stage, int n_blobs, const float* const* blobs, const int64_t* n_elem) {
    if (!h || stage < 0 || stage > 1 || !blobs || !n_elem) { set_error("gem_load_vae: bad argument"); return 1; }
    GEM_HIP(hipSetDevice(h->cfg.device));
    const gem_config& c = h->cfg;
    const int nh = c.n_hidden, T = h->T, D = h->D, Dp = h->Dp, C = h->C;
    const int flat = h->top * T;
    // expected blob list (globalegomocap_amd.vae.VAEShape.schema order)
    std::vector<int64_t> expect;
    auto conv_bn = [&](int ci, int co, bool bn) {
        expect.push_back((int64_t)ci * co * 3); expect.push_back(co);
        if (bn) for (int q = 0; q < 4; ++q) expect.push_back(co);
    };
    { int ci = C; for (int i = 0; i < nh; ++i) { conv_bn(ci, c.hidden[i], true); ci = c.hidden[i]; } }
    for (int q = 0; q < 2; ++q) { expect.push_back((int64_t)D * flat); expect.push_back(D); }
    expect.push_back((int64_t)flat * D); expect.push_back(flat);
    for (int i = nh - 1; i >= 1; --i) conv_bn(c.hidden[i], c.hidden[i - 1], true);
    conv_bn(c.hidden[0], c.hidden[0], true);
    conv_bn(c.hidden[0], C, false);
    if ((int)expect.size() != n_blobs) { set_error("gem_load_vae: expected " + std::to_string(expect.size()) + " blobs, got " + std::to_string(n_blobs)); return 1; }
    for (int i = 0; i < n_blobs; ++i)
        if (expect[i] != n_elem[i] || !blobs[i]) { set_error("gem_load_vae: size mismatch for blob " + std::to_string(i)); return 1; }

    StageNet& net = h->net[stage];
    if (net.loaded) GEM_HIP(hipDeviceSynchronize());      // reloading: launches that still read the old weights must be done
    drop_graphs(h);                                       // captured calls hold pointers to the weights freed below
    ++h->cfg_gen;                                         // a second lane mirrors the new StageNet on its next call
    free_all(net.allocs);
    net = StageNet();
    int bi = 0;
    // ---- encoder convs
    { int ci = C;
      for (int i = 0; i < nh; ++i) {
          FoldedConv f = fold_conv(blobs[bi], blobs[bi + 1], nullptr, blobs + bi + 2, ci, c.hidden[i], false);
          bi += 6;
          Layer L;
          if (make_conv_layers(net, f, &L, nullptr)) return 1;
          net.enc.push_back(L);
          ci = c.hidden[i];
      } }
    // ---- fc_mu | fc_var  ->  N = 2*Dp, K = T*topp, k = t*topp + c  <-  reference index c*T + t
    { const int Kp = T * h->topp;
      std::vector<float> wv((size_t)2 * Dp * Kp, 0.f), bv((size_t)2 * Dp, 0.f);
      for (int q = 0; q < 2; ++q) {
          const float* W = blobs[bi + 2 * q]; const float* b = blobs[bi + 2 * q + 1];
          for (int n = 0; n < D; ++n) {
              for (int cc = 0; cc < h->top; ++cc)
                  for (int t = 0; t < T; ++t) wv[((size_t)q * Dp + n) * Kp + (size_t)t * h->topp + cc] = W[(size_t)n * flat + (size_t)cc * T + t];
              bv[(size_t)q * Dp + n] = b[n];
          }
      }
      bi += 4;
      net.fc.taps = 1; net.fc.K = Kp; net.fc.N = 2 * Dp;
      if (upload(net.allocs, &net.fc.w, wv) || upload(net.allocs, &net.fc.bias, bv) || upload_bf16(net.allocs, &net.fc, wv)) return 1; }
    // ---- decoder_input: forward N = T*topp (n = t*topp + c), K = Dp; backward-data is the transpose
    std::vector<float> dec_in_bias_tm;      // time-major, padded (for compose_front)
    { const int Np = T * h->topp;
      const float* W = blobs[bi]; const float* b = blobs[bi + 1];
      bi += 2;
      std::vector<float> wf((size_t)Np * Dp, 0.f), bf(Np, 0.f), wb((size_t)Dp * Np, 0.f), zb(Dp, 0.f);
      for (int cc = 0; cc < h->top; ++cc)
          for (int t = 0; t < T; ++t) {
              const size_t n = (size_t)t * h->topp + cc, src = (size_t)cc * T + t;
              bf[n] = b[src];
              for (int k = 0; k < D; ++k) {
                  const float v = W[src * D + k];
                  wf[n * Dp + k] = v;
                  wb[(size_t)k * Np + n] = v;
              }
          }
      net.dec_in.taps = 1; net.dec_in.K = Dp; net.dec_in.N = Np;
      net.dec_in_bwd.taps = 1; net.dec_in_bwd.K = Np; net.dec_in_bwd.N = Dp;
      if (upload(net.allocs, &net.dec_in.w, wf) || upload(net.allocs, &net.dec_in.bias, bf) || upload_bf16(net.allocs, &net.dec_in, wf))
          return 1;
      if (upload(net.allocs, &net.dec_in_bwd.w, wb) || upload(net.allocs, &net.dec_in_bwd.bias, zb) ||
          upload_bf16(net.allocs, &net.dec_in_bwd, wb)) return 1;
      dec_in_bias_tm = bf; }
    // ---- decoder convs
    FoldedConv first_conv;
    auto add_dec = [&](int ci, int co, bool transposed, bool bn) -> int {
        FoldedConv f = fold_conv(blobs[bi], blobs[bi + 1], nullptr, bn ? blobs + bi + 2 : nullptr, ci, co, transposed);
        bi += bn ? 6 : 2;
        if (net.dec.empty()) first_conv = f;
        Layer Lf, Lb;
        net.host_fwd.emplace_back();
        net.host_bwd.emplace_back();
        if (make_conv_layers(net, f, &Lf, &Lb, &net.host_fwd.back(), &net.host_bwd.back())) return 1;
        net.dec.push_back(Lf);
        net.dec_bwd.push_back(Lb);
        return 0;
    };
    for (int i = nh - 1; i >= 1; --i)
        if (add_dec(c.hidden[i], c.hidden[i - 1], true, true)) return 1;
    if (add_dec(c.hidden[0], c.hidden[0], true, true)) return 1;
    if (add_dec(c.hidden[0], C, false, false)) return 1;
    // fuse as many trailing decoder convs as fit the LDS of one CU (tail.hip); GEM_NO_TAIL=1 disables it
    net.tail_start = -1;
    // The chain may start at conv 0 (GEM_TAIL_START=0), but its 512x256 weights (3 MB per workgroup and round from
    // L2) cost more than the batched GEMM they replace: 13.4 k vs 14.3 k windows/s at 240 windows.
    const int first = dev_env("GEM_TAIL_START") ? atoi(dev_env("GEM_TAIL_START")) : 1;
    if (!dev_env("GEM_NO_TAIL"))
        for (int st = first; st < (int)net.dec.size(); ++st) {
            const size_t bytes = plan_tail(net.dec, st, T, h->J, nullptr);
            if (bytes && bytes <= 160 * 1024) { net.tail_start = st; net.tail_lds = bytes; break; }
        }
    // decoder_input o conv 0 as one layer, when the tail takes over right behind conv 0 (GEM_NO_FRONT=1 keeps the two layers)
    if (net.tail_start == 1 && !dev_env("GEM_NO_FRONT") && compose_front(h, net, first_conv, dec_in_bias_tm.data())) return 1;
    // the same tail layers as per-wave bf16 fragment streams for the multi-window bf16 tail (tail_bf16.hip)
    if (build_tail_bf16_stream(h, net)) return 1;
    net.host_fwd.clear(); net.host_fwd.shrink_to_fit();
    net.host_bwd.clear(); net.host_bwd.shrink_to_fit();
    net.loaded = true;
    return 0;
}

}  // extern "C"

// ------------------------------------------------------------------------------------------------
namespace gem {

static int check_call(gem_handle* h, int stage, int B, const char* who) {
    if (!h) { set_error(std::string(who) + ": null handle"); return 1; }
    if (stage < 0 || stage > 1 || !h->net[stage].loaded) { set_error(std::string(who) + ": VAE weights of this stage are not loaded"); return 1; }
    if (B < 0 || B > h->ws.Bmax) { set_error(std::string(who) + ": B exceeds max_windows"); return 1; }
    GEM_HIP(hipSetDevice(h->cfg.device));
    return 0;
}

static int encoder_forward(gem_handle* h, int stage, int B, const float* d_pose, hipStream_t s) {
    StageNet& net = h->net[stage];
    Workspace& w = h->ws;
    const int rows = B * h->T;
    if (launch_pack_pose(d_pose, w.pose_p, rows, h->C, s)) return 1;
    const float* in = w.pose_p;
    int lda = PAD;
    for (size_t i = 0; i < net.enc.size(); ++i) {
        if (launch_gemm(h, net.enc[i], EPI_BIAS_LRELU, in, lda, nullptr, w.enc_act[i], net.enc[i].N, rows, h->T, s, -1)) return 1;
        in = w.enc_act[i];
        lda = net.enc[i].N;
    }
    return launch_gemm(h, net.fc, EPI_BIAS, in, net.fc.K, nullptr, w.mulv, net.fc.N, B, h->T, s, -1);
}

static int decoder_forward(gem_handle* h, int stage, int B, const float* zp, hipStream_t s) {
    StageNet& net = h->net[stage];
    Workspace& w = h->ws;
    const int rows = B * h->T;
    const bool front = net.front.w && net.dec.size() > 1 && h->precision != GEM_PRECISION_BF16;
    const float* in = w.h0;
    if (front) {          // decoder_input o conv 0 as one product (compose_front)
        if (launch_gemm(h, net.front, EPI_BIAS_LRELU, zp, h->Dp, nullptr, w.dec_act[0], net.front.N, B, h->T, s, 0, w.dyn ? w.perm : nullptr)) return 1;
        in = w.dec_act[0];
    } else if (launch_gemm(h, net.dec_in, EPI_BIAS, zp, h->Dp, nullptr, w.h0, net.dec_in.N, B, h->T, s, 0, w.dyn ? w.perm : nullptr)) {
        return 1;
    }
    for (size_t i = front ? 1 : 0; i < net.dec.size(); ++i) {
        const int epi = (i + 1 < net.dec.size()) ? EPI_BIAS_LRELU : EPI_BIAS;
        if (launch_gemm(h, net.dec[i], epi, in, net.dec[i].K, nullptr, w.dec_act[i], net.dec[i].N, rows, h->T, s, -1)) return 1;
        in = w.dec_act[i];
    }
    return 0;
}

// backward-data from decoder conv `from` down to the latent; gin = gradient w.r.t. the output of conv `from`
static int decoder_backward(gem_handle* h, int stage, int B, hipStream_t s, int from, const float* gin) {
    StageNet& net = h->net[stage];
    Workspace& w = h->ws;
    const int rows = B * h->T;
    const bool front = net.front.w && net.dec.size() > 1 && h->precision != GEM_PRECISION_BF16 && from >= 1;
    for (int i = from; i >= (front ? 1 : 0); --i) {
        const Layer& L = net.dec_bwd[i];
        const float* aux = i > 0 ? w.dec_act[i - 1] : nullptr;      // LeakyReLU' from the sign of the stored activation
        if (launch_gemm(h, L, i > 0 ? EPI_MASK : EPI_NONE, gin, L.K, aux, w.dec_grad[i], L.N, rows, h->T, s, -1)) return 1;
        gin = w.dec_grad[i];
    }
    // in the rounds lbfgs_advance sums the slabs of this product itself (its bias is zero)
    w.defer_reduce = w.dyn;
    const Layer& last = front ? net.front_bwd : net.dec_in_bwd;          // front: gin is the gradient w.r.t. conv 0's pre-activation
    const int rc = launch_gemm(h, last, EPI_BIAS, gin, last.K, nullptr, w.dz, h->Dp, B, h->T, s, 0);
    w.grad_slab = w.defer_reduce ? w.deferred : SlabSrc{};
    w.defer_reduce = false;
    return rc;
}

static EnergyArgs energy_args(gem_handle* h, const float* X0, const float* heat, const int32_t* frame0, const float* mean_bone,
                              const gem_energy_weights& wt) {
    Workspace& w = h->ws;
    EnergyArgs a;
    a.Xp = w.dec_act.back(); a.X0 = X0; a.heat = heat; a.frame0 = frame0; a.mean_bone = mean_bone;
    a.dXp = w.dXp; a.dXp_b = nullptr; a.f = w.f; a.parts = w.parts;
    a.tex_key = w.tex_on ? w.tex_key : nullptr; a.tex_val = w.tex_on ? w.tex_val : nullptr;
    a.w3d = (float)wt.w3d; a.ws = (float)wt.smooth; a.wb = (float)wt.bone; a.wv = (float)wt.vae; a.wr = (float)wt.reproj;
    a.dw3d = wt.w3d; a.dws = wt.smooth; a.dwb = wt.bone; a.dwv = wt.vae; a.dwr = wt.reproj;
    a.T = h->T; a.J = h->J; a.H = h->cfg.heat_h; a.W = h->cfg.heat_w; a.n_poly = h->cfg.n_poly;
    for (int i = 0; i < GEM_MAX_POLY; ++i) a.poly[i] = i < h->cfg.n_poly ? (float)h->cfg.poly[i] : 0.f;
    a.cx = (float)h->cfg.cx; a.cy = (float)h->cfg.cy;
    a.parents = h->d_parents; a.children = h->d_children;
    a.n_dev = w.dyn ? w.n_active : nullptr;
    a.perm = w.dyn ? w.perm : nullptr;
    return a;
}

// forward_only: decode to w.dec_act.back() only (the final pose of a stage), same kernels
static int evaluate(gem_handle* h, int stage, int B, const float* zp, const EnergyArgs& ea, hipStream_t s, bool forward_only = false) {
    StageNet& net = h->net[stage];
    Workspace& w = h->ws;
    // The fused tail trades throughput for latency (~45 us per workgroup whatever the batch, one or two workgroups per CU at a
    // time): measured against the batched GEMMs for the narrow layers it wins up to ten workgroups per CU in its two-per-CU
    // shape, five otherwise (tail_cap_workgroups, tail.hip, has the table).  GEM_FORCE_TAIL=1 keeps it on for any batch.
    if (h->precision == GEM_PRECISION_BF16) return evaluate_bf16(h, stage, B, ea, s, forward_only);      // zp == ws.trial, mirrored in ws.trial_b
    static const bool force_tail = dev_env("GEM_FORCE_TAIL") != nullptr;
    const int tail_g = h->T <= 16 ? 16 / h->T : 1;
    const int tail_wgs = (B + tail_g - 1) / tail_g;
    const int tail_cap = net.tail_start >= 0 ? tail_cap_workgroups(h, net.dec, net.tail_start) : 0;
    if (net.tail_start < 0 || (tail_wgs > tail_cap && !force_tail)) {
        if (w.next_count) { set_error("evaluate: the batched layers need compact_kernel's slot order (stage_begin chose otherwise)"); return 1; }
        if (decoder_forward(h, stage, B, zp, s)) return 1;
        if (forward_only) return 0;
        if (launch_energy(h, ea, B, s) || record_mid(h, s)) return 1;
        return decoder_backward(h, stage, B, s, (int)net.dec.size() - 1, w.dXp);
    }
    // wide layers as batched GEMMs, the narrow tail + energy + its adjoints in one kernel
    const int st = net.tail_start, rows = B * h->T;
    const bool front = net.front.w && st == 1 && h->precision != GEM_PRECISION_BF16;      // (the bf16 decoder mode has its own evaluate)
    if (w.next_count && !front) { set_error("evaluate: slots handed out by lbfgs_advance need the composed front layer"); return 1; }
    const float* in = w.h0;
    SlabSrc in_slab;
    if (front) {
        // decoder_input and conv 0 as ONE product (compose_front); in the rounds its split-K slabs (if any) go to the tail
        w.defer_reduce = w.dyn;
        const int rc = launch_gemm(h, net.front, EPI_BIAS_LRELU, zp, h->Dp, nullptr, w.dec_act[0], net.front.N, B, h->T, s, 0,
                                   w.dyn ? w.perm : nullptr);
        if (w.defer_reduce) in_slab = w.deferred;
        w.defer_reduce = false;
        if (rc) return 1;
        in = w.dec_act[0];
    } else {
        if (launch_gemm(h, net.dec_in, EPI_BIAS, zp, h->Dp, nullptr, w.h0, net.dec_in.N, B, h->T, s, 0, w.dyn ? w.perm : nullptr)) return 1;
        for (int i = 0; i < st; ++i) {
            // in the rounds the last wide conv leaves its split-K slabs to the tail kernel (sum + bias + LeakyReLU while staging)
            w.defer_reduce = w.dyn && i == st - 1;
            const int rc = launch_gemm(h, net.dec[i], EPI_BIAS_LRELU, in, net.dec[i].K, nullptr, w.dec_act[i], net.dec[i].N, rows, h->T, s, -1);
            if (w.defer_reduce) in_slab = w.deferred;
            w.defer_reduce = false;
            if (rc) return 1;
            in = w.dec_act[i];
        }
    }
    TailArgs ta;
    const size_t tail_lds = plan_tail_for(h, net.dec, st, tail_wgs, &ta);
    ta.B = B; ta.forward_only = forward_only ? 1 : 0; ta.dbg_ts = nullptr;
    ta.in_slab = in_slab; ta.in_bias = front ? net.front.bias : (st > 0 ? net.dec[st - 1].bias : nullptr);
    ta.in_bias_ld = front ? net.dec[0].N : 0;
    for (int i = 0; i < ta.n; ++i) {
        const Layer& f = net.dec[st + i];
        const Layer& g = net.dec_bwd[st + i];
        ta.fwd[i] = TailLayerDev{f.w4, f.bias, f.K, f.N};
        ta.bwd[i] = TailLayerDev{g.w4, nullptr, g.K, g.N};
    }
    ta.a_in = st > 0 ? w.dec_act[st - 1] : w.h0; ta.g_out = w.dec_grad[st]; ta.g_out_b = nullptr; ta.Xp = (w.dyn && !forward_only) ? nullptr : w.dec_act.back();     // the pose is only read back outside the rounds
    ta.e = ea;
    if (launch_tail(h, ta, tail_lds, s)) return 1;
    if (forward_only) return 0;
    if (record_mid(h, s)) return 1;
    if (front) {
        // dE/dz = Wf^T . (gradient w.r.t. the pre-activation of conv 0): replaces the conv adjoint, its reduce pass and the
        // decoder_input backward product; in the rounds lbfgs_advance sums the slabs of this product itself (its bias is zero)
        w.defer_reduce = w.dyn;
        w.fuse_lbfgs = w.fuse_lbfgs_req;          // (experiment: THIS launch may carry lbfgs_advance behind a device-wide barrier)
        const int rc = launch_gemm(h, net.front_bwd, EPI_BIAS, w.dec_grad[st], net.front_bwd.K, nullptr, w.dz, h->Dp, B, h->T, s, 0);
        w.fuse_lbfgs = nullptr;
        w.grad_slab = w.defer_reduce ? w.deferred : SlabSrc{};
        w.defer_reduce = false;
        return rc;
    }
    return decoder_backward(h, stage, B, s, st - 1, w.dec_grad[st]);
}

// One stage of B windows as three host steps, so that two half-batches ("lanes", below) can be driven round by round from one
// loop: begin (encode, initial state), round r (one evaluation + one L-BFGS advance for every window still iterating), finish
// (decode the result).  optimize_stage_impl runs them back to back.
struct StageRun {
    gem_handle* h = nullptr;
    int stage = 0, B = 0;
    const float* pose_in = nullptr; const float* heat = nullptr; const int32_t* frame0 = nullptr; const float* mean_bone = nullptr;
    const float* eps = nullptr;
    gem_energy_weights wt{}; gem_lbfgs_opts opt{};
    float* pose_out = nullptr; gem_window_stats* stats = nullptr;
    hipStream_t s = nullptr;
    EnergyArgs ea{};
    bool fuse = false;
    bool atomic_slots = false;          // lbfgs_advance hands out the next round's slots itself (Workspace::next_*): no compact launch
    long log0 = 0;                      // ... n_log entry of round 0's count (round k: log0 + k)
    int rounds = 0;
};

// the workspace's compaction pointers back on their allocations (a stage with atomic slots moves them round by round)
static void compaction_home(Workspace& w) {
    w.perm = w.perm_home; w.slot_of = w.slot_of_home; w.n_active = w.n_active_home;
    w.next_perm = w.next_slot_of = w.next_count = nullptr;
}

static int stage_begin(StageRun& r) {
    gem_handle* h = r.h;
    Workspace& w = h->ws;
    const int B = r.B, stage = r.stage;
    hipStream_t s = r.s;
    if (r.wt.reproj != 0.0 && (!r.heat || !r.frame0)) { set_error("optimize: reproj weight != 0 needs heat-maps and frame indices"); return 1; }
    if (r.opt.max_iter < 1 || r.opt.max_eval < 1 || r.opt.max_iter - 1 > w.hist_cap || r.opt.max_iter > MAX_HIST) {
        set_error("optimize: max_iter must be 1.." + std::to_string(w.hist_cap + 1)); return 1;
    }
    // (the per-round counters of a stage are zeroed by one 1024-thread workgroup, and the trace keeps TRACE_ROUNDS rounds)
    if (r.opt.max_eval > 1021) { set_error("optimize: max_eval must be at most 1021 (torch's default for max_iter = 25 is 31)"); return 1; }
    r.rounds = r.opt.max_eval + 1;          // upper bound on evaluations per window (see lbfgs.hip)
    if (B == 0) return 0;
    if (encoder_forward(h, stage, B, r.pose_in, s)) return 1;
    if (launch_reparam(w.mulv, r.eps, nullptr, nullptr, nullptr, w.trial, B, h->D, h->Dp, s)) return 1;
    if (h->precision == GEM_PRECISION_BF16 && launch_f32_to_bf16(w.trial, w.trial_b, (size_t)B * h->Dp, s)) return 1;
    if (launch_lbfgs_init(h, B, r.opt, s)) return 1;
    // Rounds run on the windows that are still iterating: after every advance they are re-packed to the front
    // (perm / n_active on the device) and the kernels of the next round read their row count from there.
    static const bool no_compact = dev_env("GEM_NO_COMPACT") != nullptr;
    compaction_home(w);
    r.atomic_slots = false;
    // The active windows are re-packed between the rounds: inside the decoder_input forward launch of the next round (one sequence in
    // fp32: gemm_rows.h, FUSE), by lbfgs_advance handing out the next round's slots itself (atomic_slots: every path whose kernels
    // address rows through perm / slot_of only and do not depend on the slot ORDER -- the bf16 fused path, and the fp32 composed front
    // layer + fused tail beyond one sequence), else by compact_kernel.
    StageNet& net_ = h->net[stage];
    const bool front_ = net_.front.w && net_.tail_start == 1 && h->precision != GEM_PRECISION_BF16;
    const Layer& first_ = front_ ? net_.front : net_.dec_in;
    const int tail_g_ = h->T <= 16 ? 16 / h->T : 1;
    const bool tail_path_ = net_.tail_start >= 0 && (B + tail_g_ - 1) / tail_g_ <= tail_cap_workgroups(h, net_.dec, net_.tail_start);
    const bool fuse_ = !no_compact && tail_path_ && rows_can_fuse_compaction(h, first_, h->Dp, first_.N, B, /*slabs=*/front_);
    if (!no_compact) {
        r.atomic_slots = bf16_rounds_take_slots_atomically(h, stage, B) ||
                         (h->precision == GEM_PRECISION_F32 && front_ && tail_path_ && !fuse_ && !dev_env("GEM_NO_ATOMIC_COMPACT"));
        if (r.atomic_slots) {
            // rounds + 2 consecutive n_log entries: round 0's count (written by the compaction below), then one zeroed counter per round
            if ((w.log_pos % N_LOG) + r.rounds + 2 > N_LOG) w.log_pos += N_LOG - (w.log_pos % N_LOG);
            r.log0 = w.log_pos;
        }
        // identity: every window takes part in round 0 (logs B at n_log[log0]); with atomic slots the kernel also zeroes the rounds' counters
        if (launch_compact(h, B, 1, s, r.atomic_slots ? r.rounds + 1 : 0)) return 1;
        if (r.atomic_slots) w.log_pos = r.log0 + r.rounds + 2;
        w.dyn = true;
    }
    // texel-block cache of the reprojection term: valid for this stage's heat-maps / windows only
    static const bool no_tex = dev_env("GEM_NO_TEXCACHE") != nullptr;
    w.tex_on = !no_tex && h->tex_cache && w.tex_key && r.wt.reproj != 0.0;
    // (a fill KERNEL, not hipMemsetAsync: inside a captured graph a memset node was seen to run out of order with the kernels around it
    // once two graphs replayed side by side on two streams -- round 5, ROCm 7.2; a late invalidation here would hand the stage texels
    // of the previous contents of the heat-maps)
    if (w.tex_on && launch_fill_u32(reinterpret_cast<uint32_t*>(w.tex_key), 0xFFFFFFFFu, (size_t)B * h->T * h->J, s)) return 1;
    r.ea = energy_args(h, r.pose_in, r.heat, r.frame0, r.mean_bone, r.wt);
    w.tex_on = false;
    // closure values of this stage, one row per round (0xFF bytes = NaN: "window took no evaluation in this round")
    if (launch_fill_u32(reinterpret_cast<uint32_t*>(w.trace), 0xFFFFFFFFu, (size_t)TRACE_ROUNDS * w.Bmax * 2, s)) return 1;
    r.fuse = w.dyn && fuse_;
    return 0;
}

static int stage_round(StageRun& r, int k) {
    gem_handle* h = r.h;
    Workspace& w = h->ws;
    if (r.B == 0) return 0;
    int rc = 0;
    w.round = k;
    // (dyn / tex state of THIS lane's workspace: another lane may have run in between)
    if (w.dyn && r.atomic_slots) {
        // this round's set and the set lbfgs_advance fills for the next one
        int* cnt = w.n_log + (r.log0 + k) % N_LOG;
        w.perm = (k & 1) ? w.perm2 : w.perm_home;       w.next_perm = (k & 1) ? w.perm_home : w.perm2;
        w.slot_of = (k & 1) ? w.slot_of2 : w.slot_of_home; w.next_slot_of = (k & 1) ? w.slot_of_home : w.slot_of2;
        w.n_active = cnt; w.next_count = cnt + 1;
        w.cur_log = r.log0 + k;
        r.ea.n_dev = w.n_active; r.ea.perm = w.perm;
    } else
    if (k > 0 && w.dyn) {
        if (r.fuse) {
            w.fuse_compact = true;
            w.fuse_log = w.n_log + (w.log_pos % N_LOG);
            w.cur_log = w.log_pos++;
        } else {
            rc = launch_compact(h, r.B, 0, r.s);
        }
    }
    // (experiment, GEM_DEV=1 GEM_FUSE_BWD_LBFGS=1: the backward front product carries the advance behind a device-wide barrier)
    const bool fuse_exp = dev_env("GEM_FUSE_BWD_LBFGS") != nullptr;          // (read per round: a test flips it inside one process)
    w.lbfgs_fused_done = false;
    w.fuse_lbfgs_req = (fuse_exp && w.dyn && r.fuse && !h->graphs_on && !h->prof.on && h->precision == GEM_PRECISION_F32) ? &r.opt : nullptr;
    rc = rc || evaluate(h, r.stage, r.B, w.trial, r.ea, r.s);
    w.fuse_lbfgs_req = nullptr;
    if (!rc && !w.lbfgs_fused_done) rc = launch_lbfgs_advance(h, r.B, r.opt, r.s);
    w.lbfgs_fused_done = false;
    if (w.fuse_compact) { set_error("optimize: the fused compaction was not picked up"); rc = 1; w.fuse_compact = false; }
    if (w.mid_event) {          // (no evaluation path picked the half-round marker up: record it now rather than never)
        GEM_HIP(hipEventRecord(w.mid_event, r.s));
        w.mid_event = nullptr;
    }
    w.round = -1;
    return rc;
}

static int stage_finish(StageRun& r) {
    gem_handle* h = r.h;
    Workspace& w = h->ws;
    w.round = -1;
    w.dyn = false;
    compaction_home(w);
    if (r.B == 0) return 0;
    // every window is finished now: trial == x*; decode it with the same kernels as the rounds (all windows again)
    if (evaluate(h, r.stage, r.B, w.trial, energy_args(h, r.pose_in, r.heat, r.frame0, r.mean_bone, r.wt), r.s, true)) return 1;
    if (launch_unpack_pose(w.dec_act.back(), r.pose_out, r.B * h->T, h->C, r.s)) return 1;
    if (r.stats && launch_lbfgs_stats(h, r.B, r.stats, r.s)) return 1;
    return 0;
}

static int optimize_stage_impl(gem_handle* h, int stage, int B, const float* d_pose_in, const float* d_heat,
                               const int32_t* d_frame0, const float* d_mean_bone, const float* d_eps,
                               const gem_energy_weights& wt, const gem_lbfgs_opts& opt, float* d_pose_out,
                               gem_window_stats* d_stats, hipStream_t s) {
    StageRun r;
    r.h = h; r.stage = stage; r.B = B; r.pose_in = d_pose_in; r.heat = d_heat; r.frame0 = d_frame0; r.mean_bone = d_mean_bone; r.eps = d_eps;
    r.wt = wt; r.opt = opt; r.pose_out = d_pose_out; r.stats = d_stats; r.s = s;
    if (stage_begin(r)) { h->ws.dyn = false; compaction_home(h->ws); return 1; }
    int rc = 0;
    for (int k = 0; k < r.rounds && !rc; ++k) rc = stage_round(r, k);
    if (rc) { h->ws.round = -1; h->ws.dyn = false; compaction_home(h->ws); return 1; }
    return stage_finish(r);
}

// ---- both stages of the window loop for one lane (optimizer.py:370-423), as host steps around the stage rounds ---------------
struct WindowsRun {
    gem_handle* h = nullptr;
    int B = 0;
    const float* local_pose = nullptr; const double* cams = nullptr; const float* heat = nullptr; const int32_t* frame0 = nullptr;
    const float* mean_bone = nullptr; const float* eps_local = nullptr; const float* eps_global = nullptr;
    gem_energy_weights w_local{}, w_global{}; gem_lbfgs_opts opt{};
    float* mid_local = nullptr; double* global = nullptr; gem_window_stats* stats_local = nullptr; gem_window_stats* stats_global = nullptr;
    hipStream_t s = nullptr;
    StageRun st;
    float* mid = nullptr;
};

static int windows_begin_local(WindowsRun& r) {
    gem_handle* h = r.h;
    Workspace& w = h->ws;
    if (launch_gather_windows(r.local_pose, r.frame0, w.pose_a, r.B, h->T, h->C, r.s)) return 1;
    r.mid = r.mid_local ? r.mid_local : w.pose_b;
    StageRun& s = r.st;
    s = StageRun{};
    s.h = h; s.stage = GEM_STAGE_LOCAL; s.B = r.B; s.pose_in = w.pose_a; s.heat = r.heat; s.frame0 = r.frame0; s.mean_bone = r.mean_bone;
    s.eps = r.eps_local; s.wt = r.w_local; s.opt = r.opt; s.pose_out = r.mid; s.stats = r.stats_local; s.s = r.s;
    return stage_begin(s);
}
static int windows_begin_global(WindowsRun& r) {       // local stage -> fp64 relative-global transform -> global stage set up
    gem_handle* h = r.h;
    Workspace& w = h->ws;
    if (stage_finish(r.st)) return 1;
    if (launch_relative_global(r.mid, r.cams, r.frame0, w.pose_a, r.B, h->T, h->J, r.s)) return 1;
    StageRun& s = r.st;
    s = StageRun{};
    s.h = h; s.stage = GEM_STAGE_GLOBAL; s.B = r.B; s.pose_in = w.pose_a; s.heat = r.heat; s.frame0 = r.frame0; s.mean_bone = r.mean_bone;
    s.eps = r.eps_global; s.wt = r.w_global; s.opt = r.opt; s.pose_out = w.pose_b;      // (the stage-A result kept there, if any, is dead after the transform)
    s.stats = r.stats_global; s.s = r.s;
    return stage_begin(s);
}
static int windows_end(WindowsRun& r) {
    gem_handle* h = r.h;
    if (stage_finish(r.st)) return 1;
    return launch_to_global(h->ws.pose_b, r.cams, r.frame0, r.global, r.B, h->T, h->J, r.s);
}
static void windows_abort(WindowsRun& r) { r.h->ws.round = -1; r.h->ws.dyn = false; r.h->ws.mid_event = nullptr; compaction_home(r.h->ws); }

static int windows_single(WindowsRun& r) {
    int rc = windows_begin_local(r);
    for (int k = 0; k < r.st.rounds && !rc; ++k) rc = stage_round(r.st, k);
    rc = rc || windows_begin_global(r);
    for (int k = 0; k < r.st.rounds && !rc; ++k) rc = stage_round(r.st, k);
    rc = rc || windows_end(r);
    if (rc) windows_abort(r);
    return rc;
}

// ---- two lanes --------------------------------------------------------------------------------------------------------------
// Windows are independent (optimizer.py:370), so a large batch can run as two half-batches on two streams, shifted by HALF an
// evaluation round: while lane A runs its backward product and its HBM-bound L-BFGS advance, lane B runs its forward product and
// its fused tail, and vice versa -- the memory-bound kernel of one lane shares the machine with the matrix-bound kernels of the
// other instead of owning it alone.  The shift is enforced, not hoped for: every round of a lane waits for the event the other
// lane records behind its tail kernel (B's round r for A's tail of round r, A's round r+1 for B's tail of round r), so the two
// tails never run together and the lanes cannot drift back into step.  Inside a captured graph the events become edges between
// the two chains.  Each lane has its own workspace (a second handle that shares the weights); a window's result does not depend
// on which other windows share its batch as long as no product is cut along K (lanes are used from 4352 windows on, where
// neither the full batch nor its halves are): bitwise the single-lane result (tests/test_hip_full_size.py).
static int ensure_lane(gem_handle* h, int B_lane) {
    if (h->lane2 && h->lane2->ws.Bmax >= B_lane) return 0;
    if (h->lane2) { GEM_HIP(hipDeviceSynchronize()); gem_destroy(h->lane2); h->lane2 = nullptr; }
    h->lane_gen = 0;                            // a fresh lane has mirrored nothing yet: the next sync_lane copies nets, precision, texel cache
    gem_config cfg = h->cfg;
    cfg.max_windows = std::max(B_lane, (h->ws.Bmax + 1) / 2 + 8);
    gem_handle* l = nullptr;
    if (gem_create(&cfg, &l)) return 1;
    h->lane2 = l;
    if (!h->lane_stream) GEM_HIP(hipStreamCreateWithFlags(&h->lane_stream, hipStreamNonBlocking));
    if (!h->lane_stream_a) GEM_HIP(hipStreamCreateWithFlags(&h->lane_stream_a, hipStreamNonBlocking));
    if (!h->ev_fork) {
        GEM_HIP(hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming));
        GEM_HIP(hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming));
        GEM_HIP(hipEventCreateWithFlags(&h->ev_join_a, hipEventDisableTiming));
    }
    return 0;
}
static void sync_lane(gem_handle* h) {          // the lane evaluates the same networks with the same settings (weights are shared, not copied)
    gem_handle* l = h->lane2;
    l->prof.on = false;
    if (h->lane_gen == h->cfg_gen) return;      // nothing loaded or switched since the last two-lane call (incl. every graph replay)
    h->lane_gen = h->cfg_gen;
    for (int st = 0; st < 2; ++st) {
        l->net[st] = h->net[st];
        l->net[st].allocs.clear();              // owned by h
    }
    l->precision = h->precision; l->tex_cache = h->tex_cache; l->prof.on = false;
}
static hipEvent_t lane_event(gem_handle* h, size_t i) {
    while (h->ev_pool.size() <= i) {
        hipEvent_t e = nullptr;
        if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return nullptr;
        h->ev_pool.push_back(e);
    }
    return h->ev_pool[i];
}

static int windows_dual(WindowsRun& a, WindowsRun& b, gem_handle* h, hipStream_t caller) {
    hipStream_t sa = a.s, sb = b.s;
    GEM_HIP(hipEventRecord(h->ev_fork, caller));
    if (sa != caller) GEM_HIP(hipStreamWaitEvent(sa, h->ev_fork, 0));
    GEM_HIP(hipStreamWaitEvent(sb, h->ev_fork, 0));
    int rc = windows_begin_local(a) || windows_begin_local(b);
    size_t ev = 0;
    hipEvent_t last_b = nullptr;
    // (GEM_DEV=1 GEM_LANES_FREE=1, A/B runs: the two lanes run free on their streams, no half-round lock)
    const bool free_run = dev_env("GEM_LANES_FREE") != nullptr;
    auto rounds = [&]() {
        for (int k = 0; k < a.st.rounds && !rc; ++k) {
            if (free_run) {
                rc = rc || stage_round(a.st, k) || stage_round(b.st, k);
                continue;
            }
            hipEvent_t ea = lane_event(h, ev++), eb = lane_event(h, ev++);
            if (!ea || !eb) { set_error("optimize: hipEventCreate failed"); rc = 1; break; }
            if (last_b) rc = rc || !hip_ok(hipStreamWaitEvent(sa, last_b, 0), "hipStreamWaitEvent");
            a.h->ws.mid_event = ea;
            rc = rc || stage_round(a.st, k);
            rc = rc || !hip_ok(hipStreamWaitEvent(sb, ea, 0), "hipStreamWaitEvent");
            b.h->ws.mid_event = eb;
            rc = rc || stage_round(b.st, k);
            last_b = eb;
        }
    };
    rounds();
    rc = rc || windows_begin_global(a) || windows_begin_global(b);
    rounds();
    rc = rc || windows_end(a) || windows_end(b);
    if (rc) { windows_abort(a); windows_abort(b); }
    // join: the caller's stream continues when both lanes are done (also on an error: the capture, if any, must see the join)
    if (hipEventRecord(h->ev_join, sb) != hipSuccess || hipStreamWaitEvent(caller, h->ev_join, 0) != hipSuccess) rc = 1;
    if (sa != caller && (hipEventRecord(h->ev_join_a, sa) != hipSuccess || hipStreamWaitEvent(caller, h->ev_join_a, 0) != hipSuccess)) rc = 1;
    return rc;
}

// ---- hipGraph replay of a whole call ------------------------------------------------------------------------------------
// An optimisation call is a fixed sequence of ~700 launches whose grids and arguments do not depend on the data (row counts
// live on the device, finished windows are skipped inside the kernels), i.e. it is capture-safe as it stands.  With graphs
// enabled, the first call with a given signature runs eagerly (it also performs the one-time hipFuncSetAttribute settings),
// the second one is captured into a hipGraph and instantiated, every later one is a single hipGraphLaunch: the host cost of a
// call drops from ~3 ms of launches to one launch (BASELINE configs[4]; several sequences in flight from one host thread).
static bool same_key(const GraphKey& a, const GraphKey& b) {
    if (a.kind != b.kind || a.stage != b.stage || a.B != b.B || a.precision != b.precision || a.stream != b.stream || a.tex_cache != b.tex_cache ||
        a.lanes != b.lanes)
        return false;
    for (int i = 0; i < 12; ++i)
        if (a.ptr[i] != b.ptr[i]) return false;
    return std::memcmp(a.w, b.w, sizeof(a.w)) == 0 && std::memcmp(&a.opt, &b.opt, sizeof(a.opt)) == 0;
}

template <typename Body>
static int run_graphed(gem_handle* h, const GraphKey& key, hipStream_t s, Body body) {
    // the legacy default stream cannot be captured; event-based profiling records events between launches
    if (!h->graphs_on || s == nullptr || h->prof.on) return body();
    GraphEntry* e = nullptr;
    for (auto& g : h->graphs)
        if (same_key(g.key, key)) { e = &g; break; }
    ++h->graph_tick;
    if (!e) {                                   // first sighting: eager run (warm-up), remember the signature
        if (h->graphs.size() >= 16) {           // bounded cache: drop the least recently used entry
            size_t lru = 0;
            for (size_t i = 1; i < h->graphs.size(); ++i)
                if (h->graphs[i].last_use < h->graphs[lru].last_use) lru = i;
            if (h->graphs[lru].exec) (void)hipGraphExecDestroy(h->graphs[lru].exec);
            if (h->graphs[lru].graph) (void)hipGraphDestroy(h->graphs[lru].graph);
            h->graphs.erase(h->graphs.begin() + lru);
        }
        GraphEntry n;
        n.key = key; n.last_use = h->graph_tick;
        h->graphs.push_back(n);
        return body();
    }
    e->last_use = h->graph_tick;
    if (!e->exec) {
        GEM_HIP(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        const int rc = body();
        hipGraph_t g = nullptr;
        const hipError_t ec = hipStreamEndCapture(s, &g);
        if (rc) { if (g) (void)hipGraphDestroy(g); return rc; }
        if (!hip_ok(ec, "hipStreamEndCapture")) return 1;
        hipGraphExec_t x = nullptr;
        if (!hip_ok(hipGraphInstantiate(&x, g, nullptr, nullptr, 0), "hipGraphInstantiate")) { (void)hipGraphDestroy(g); return 1; }
        e->graph = g; e->exec = x;
        ++h->graph_captures;
    }
    GEM_HIP(hipGraphLaunch(e->exec, s));
    ++h->graph_replays;
    return 0;
}

}  // namespace gem

extern "C" {

int gem_mean_bone_length(gem_handle* h, const float* d_pose, int n_frames, float* d_out, void* stream) {
    if (!h || !d_pose || !d_out || n_frames < 1) { set_error("gem_mean_bone_length: bad argument"); return 1; }
    GEM_HIP(hipSetDevice(h->cfg.device));
    return launch_mean_bone(h, d_pose, n_frames, d_out, (hipStream_t)stream);
}

int gem_encode(gem_handle* h, int stage, int B, const float* d_pose, const float* d_eps, float* d_mu, float* d_logvar,
               float* d_z, void* stream) {
    if (check_call(h, stage, B, "gem_encode")) return 1;
    hipStream_t s = (hipStream_t)stream;
    if (B == 0) return 0;
    if (encoder_forward(h, stage, B, d_pose, s)) return 1;
    return launch_reparam(h->ws.mulv, d_eps, d_mu, d_logvar, d_z, nullptr, B, h->D, h->Dp, s);
}

int gem_decode(gem_handle* h, int stage, int B, const float* d_z, float* d_pose, void* stream) {
    if (check_call(h, stage, B, "gem_decode")) return 1;
    hipStream_t s = (hipStream_t)stream;
    if (B == 0) return 0;
    if (launch_pad_latent(d_z, h->ws.trial, B, h->D, h->Dp, s)) return 1;
    if (decoder_forward(h, stage, B, h->ws.trial, s)) return 1;
    return launch_unpack_pose(h->ws.dec_act.back(), d_pose, B * h->T, h->C, s);
}

int gem_energy_grad(gem_handle* h, int stage, int B, const float* d_z, const float* d_pose_init, const float* d_heat,
                    const int32_t* d_frame0, const float* d_mean_bone, const gem_energy_weights* wt, double* d_energy,
                    double* d_parts, float* d_dz, float* d_pose, void* stream) {
    if (check_call(h, stage, B, "gem_energy_grad")) return 1;
    if (B == 0) return 0;
    if (!wt || !d_z || !d_pose_init || !d_mean_bone) { set_error("gem_energy_grad: null argument"); return 1; }
    if (wt->reproj != 0.0 && (!d_heat || !d_frame0)) { set_error("gem_energy_grad: reproj weight != 0 needs heat-maps"); return 1; }
    hipStream_t s = (hipStream_t)stream;
    if (B == 0) return 0;
    Workspace& w = h->ws;
    if (launch_pad_latent(d_z, w.trial, B, h->D, h->Dp, s)) return 1;
    if (h->precision == GEM_PRECISION_BF16 && launch_f32_to_bf16(w.trial, w.trial_b, (size_t)B * h->Dp, s)) return 1;
    const EnergyArgs ea = energy_args(h, d_pose_init, d_heat, d_frame0, d_mean_bone, *wt);
    if (evaluate(h, stage, B, w.trial, ea, s)) return 1;
    if (d_energy) GEM_HIP(hipMemcpyAsync(d_energy, w.f, (size_t)B * sizeof(double), hipMemcpyDeviceToDevice, s));
    if (d_parts) GEM_HIP(hipMemcpyAsync(d_parts, w.parts, (size_t)B * 5 * sizeof(double), hipMemcpyDeviceToDevice, s));
    if (d_dz && launch_unpad_latent(w.dz, d_dz, B, h->D, h->Dp, s)) return 1;
    if (d_pose && launch_unpack_pose(w.dec_act.back(), d_pose, B * h->T, h->C, s)) return 1;
    return 0;
}

int gem_optimize_stage(gem_handle* h, int stage, int B, const float* d_pose_in, const float* d_heat, const int32_t* d_frame0,
                       const float* d_mean_bone, const float* d_eps, const gem_energy_weights* wt, const gem_lbfgs_opts* opt,
                       float* d_pose_out, gem_window_stats* d_stats, void* stream) {
    if (check_call(h, stage, B, "gem_optimize_stage")) return 1;
    if (B == 0) return 0;
    if (!d_pose_in || !d_mean_bone || !wt || !opt || !d_pose_out) { set_error("gem_optimize_stage: null argument"); return 1; }
    GraphKey key;
    key.kind = 1; key.stage = stage; key.B = B; key.precision = h->precision; key.stream = stream; key.tex_cache = h->tex_cache;
    h->last_split = 0;
    const void* ptrs[] = {d_pose_in, d_heat, d_frame0, d_mean_bone, d_eps, d_pose_out, d_stats};
    for (int i = 0; i < 7; ++i) key.ptr[i] = ptrs[i];
    key.w[0] = *wt; key.opt = *opt;
    return run_graphed(h, key, (hipStream_t)stream, [&]() {
        return optimize_stage_impl(h, stage, B, d_pose_in, d_heat, d_frame0, d_mean_bone, d_eps, *wt, *opt, d_pose_out, d_stats,
                                   (hipStream_t)stream);
    });
}

int gem_optimize_windows(gem_handle* h, int B, const float* d_local_pose, const double* d_cams, const float* d_heat,
                         const int32_t* d_frame0, const float* d_mean_bone, const float* d_eps_local, const float* d_eps_global,
                         const gem_energy_weights* w_local, const gem_energy_weights* w_global, const gem_lbfgs_opts* opt,
                         float* d_mid_local, double* d_global, gem_window_stats* d_stats, void* stream) {
    if (check_call(h, 0, B, "gem_optimize_windows") || check_call(h, 1, B, "gem_optimize_windows")) return 1;
    if (B == 0) return 0;
    if (!d_local_pose || !d_cams || !d_frame0 || !d_mean_bone || !w_local || !w_global || !opt || !d_global) {
        set_error("gem_optimize_windows: null argument"); return 1;
    }
    hipStream_t s = (hipStream_t)stream;
    if (B == 0) return 0;
    GraphKey key;
    key.kind = 2; key.B = B; key.precision = h->precision; key.stream = stream; key.tex_cache = h->tex_cache;
    const void* ptrs[] = {d_local_pose, d_cams, d_heat, d_frame0, d_mean_bone, d_eps_local, d_eps_global, d_mid_local, d_global, d_stats};
    for (int i = 0; i < 10; ++i) key.ptr[i] = ptrs[i];
    key.w[0] = *w_local; key.w[1] = *w_global; key.opt = *opt;
    // two lanes (windows_dual above) from lanes_min windows on; not while event profiling is on (the per-kernel timings would be
    // those of kernels sharing the machine)
    // (bf16 decoder mode only: that is where the products' K cuts were checked to be the same for a batch and its halves)
    const bool dual = h->lanes_min > 0 && B >= h->lanes_min && !h->prof.on && h->precision == GEM_PRECISION_BF16;
    const int BA = dual ? ((B + 1) / 2 + 7) / 8 * 8 : B;        // whole tail workgroups (8 windows) in the first lane
    if (dual) {
        if (ensure_lane(h, B - BA > BA ? B - BA : BA)) return 1;
        sync_lane(h);
    }
    key.lanes = dual ? 2 : 1;
    h->last_split = dual ? BA : 0;
    return run_graphed(h, key, s, [&]() -> int {
        WindowsRun a;
        a.h = h; a.B = BA; a.local_pose = d_local_pose; a.cams = d_cams; a.heat = d_heat; a.frame0 = d_frame0; a.mean_bone = d_mean_bone;
        a.eps_local = d_eps_local; a.eps_global = d_eps_global; a.w_local = *w_local; a.w_global = *w_global; a.opt = *opt;
        a.mid_local = d_mid_local; a.global = d_global; a.stats_local = d_stats; a.stats_global = d_stats ? d_stats + B : nullptr; a.s = s;
        if (!dual) return windows_single(a);
        // lane A: the caller's stream, unless that is the legacy default stream -- its implicit synchronisation with every
        // blocking stream of the process would sit between the two lanes' kernels (measured: 39.6 instead of 32.0 ms per
        // 8196-window step as soon as other streams exist); then a private non-blocking stream, forked and joined by events
        a.s = s ? s : h->lane_stream_a;
        WindowsRun b = a;
        b.h = h->lane2; b.B = B - BA; b.frame0 = d_frame0 + BA; b.mean_bone = d_mean_bone + (size_t)BA * h->J;
        b.eps_local = d_eps_local ? d_eps_local + (size_t)BA * h->D : nullptr; b.eps_global = d_eps_global ? d_eps_global + (size_t)BA * h->D : nullptr;
        b.mid_local = d_mid_local ? d_mid_local + (size_t)BA * h->T * h->C : nullptr; b.global = d_global + (size_t)BA * h->T * h->C;
        b.stats_local = d_stats ? d_stats + BA : nullptr; b.stats_global = d_stats ? d_stats + B + BA : nullptr; b.s = h->lane_stream;
        return windows_dual(a, b, h, s);
    });
}

int gem_read_trace(gem_handle* h, int B, int n_rounds, double* d_out, void* stream) {
    if (!h || !d_out || B < 0 || B > h->ws.Bmax || n_rounds < 0 || n_rounds > TRACE_ROUNDS) {
        set_error("gem_read_trace: need 0 <= B <= max_windows and 0 <= n_rounds <= 64"); return 1;
    }
    GEM_HIP(hipSetDevice(h->cfg.device));
    if (B == 0 || n_rounds == 0) return 0;
    // (after a two-lane gem_optimize_windows call the windows [last_split, B) live in the second lane's workspace)
    const int BA = (h->last_split > 0 && h->lane2 && h->last_split < B) ? h->last_split : B;
    GEM_HIP(hipMemcpy2DAsync(d_out, (size_t)B * sizeof(double), h->ws.trace, (size_t)h->ws.Bmax * sizeof(double),
                             (size_t)BA * sizeof(double), (size_t)n_rounds, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    if (BA < B)
        GEM_HIP(hipMemcpy2DAsync(d_out + BA, (size_t)B * sizeof(double), h->lane2->ws.trace, (size_t)h->lane2->ws.Bmax * sizeof(double),
                                 (size_t)(B - BA) * sizeof(double), (size_t)n_rounds, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return 0;
}

int gem_set_texel_cache(gem_handle* h, int on) {
    if (!h) { set_error("gem_set_texel_cache: null handle"); return 1; }
    h->tex_cache = on != 0;
    ++h->cfg_gen;
    return 0;
}

int gem_graph_enable(gem_handle* h, int on) {
    if (!h) { set_error("gem_graph_enable: null handle"); return 1; }
    if (!on && !h->graphs.empty()) {
        // switching replay off drops the captured calls: they hold the ADDRESSES of the caller's tensors, and a caller about to free
        // those tensors must be able to make sure no graph is ever replayed on whatever is allocated there next (bench.py's shards)
        GEM_HIP(hipSetDevice(h->cfg.device));
        GEM_HIP(hipDeviceSynchronize());
        drop_graphs(h);
    }
    h->graphs_on = on != 0;
    return 0;
}

#ifdef GEM_DEBUG_EXPORTS          // developer builds only (tools/r05_nrt_dump.py): internal buffers by number, no copy
int gem_debug_buffer(gem_handle* h, int which, void** d_ptr) {
    if (!h || !d_ptr) return 1;
    Workspace& w = h->ws;
    const int st = h->net[0].tail_start;
    *d_ptr = which == 0 ? (void*)w.dec_grad_b[st] : which == 1 ? (void*)w.dec_act_b[st - 1] : which == 2 ? (void*)w.dz : nullptr;
    return *d_ptr ? 0 : 1;
}
#endif

int gem_graph_stats(gem_handle* h, int64_t* n_captures, int64_t* n_replays) {
    if (!h) { set_error("gem_graph_stats: null handle"); return 1; }
    if (n_captures) *n_captures = h->graph_captures;
    if (n_replays) *n_replays = h->graph_replays;
    return 0;
}

int gem_set_lanes(gem_handle* h, int min_windows) {
    if (!h || min_windows < 0) { set_error("gem_set_lanes: min_windows must be >= 0 (0 = one lane always)"); return 1; }
    h->lanes_min = min_windows;
    return 0;
}

int gem_set_precision(gem_handle* h, int mode) {
    if (!h || mode < 0 || mode > 2) { set_error("gem_set_precision: mode must be 0 (f32), 1 (bf16x3) or 2 (bf16)"); return 1; }
    h->precision = mode;
    ++h->cfg_gen;
    return 0;
}

static int post_scratch(gem_handle* h, size_t elems) {
    if (elems <= h->post_work_elems) return 0;
    GEM_HIP(hipDeviceSynchronize());                 // a previous call may still be reading the old buffer
    if (h->post_work) GEM_HIP(hipFree(h->post_work));
    h->post_work = nullptr; h->post_work_elems = 0;
    GEM_HIP(hipMalloc((void**)&h->post_work, elems * sizeof(double)));
    h->post_work_elems = elems;
    return 0;
}

int gem_merge_windows(gem_handle* h, const double* d_windows, int n_chunks, int windows_per_chunk, int overlap, int smooth,
                      double* d_out, void* stream) {
    if (!h) { set_error("gem_merge_windows: null handle"); return 1; }
    if (n_chunks < 0 || windows_per_chunk < 1 || overlap < 0 || 2 * overlap > h->T) {
        set_error("gem_merge_windows: need windows_per_chunk >= 1 and 0 <= 2*overlap <= seq_len"); return 1;
    }
    if (n_chunks == 0) return 0;
    if (!d_windows || !d_out) { set_error("gem_merge_windows: null argument"); return 1; }
    GEM_HIP(hipSetDevice(h->cfg.device));
    const int fpc = windows_per_chunk * (h->T - overlap) + overlap;
    const size_t n = (size_t)n_chunks * fpc * h->C;
    if (smooth && post_scratch(h, n)) return 1;
    return launch_merge(d_windows, h->post_work, d_out, n_chunks, windows_per_chunk, h->T, h->C, overlap, smooth, (hipStream_t)stream);
}

int gem_calculate_errors(gem_handle* h, const double* d_est, const double* d_mid, const double* d_opt, const double* d_gt,
                         int n_frames, const double* h_bone_mm, double* d_out, void* stream) {
    if (!h) { set_error("gem_calculate_errors: null handle"); return 1; }
    if (h->J < 12) { set_error("gem_calculate_errors: the hip-midpoint error needs joints 7 and 11 (n_joints >= 12)"); return 1; }
    if (n_frames < 1) { set_error("gem_calculate_errors: n_frames must be >= 1"); return 1; }
    if (!d_est || !d_mid || !d_opt || !d_gt || !h_bone_mm || !d_out) { set_error("gem_calculate_errors: null argument"); return 1; }
    GEM_HIP(hipSetDevice(h->cfg.device));
    if (post_scratch(h, (size_t)(11 + MAXJ_ERR) * n_frames)) return 1;
    return launch_errors(h, d_est, d_mid, d_opt, d_gt, n_frames, h_bone_mm, h->post_work, d_out, (hipStream_t)stream, 1);
}

int gem_calculate_errors_chunks(gem_handle* h, const double* d_est, const double* d_mid, const double* d_opt, const double* d_gt,
                                int n_chunks, int frames_per_chunk, const double* h_bone_mm, double* d_out, void* stream) {
    if (!h) { set_error("gem_calculate_errors_chunks: null handle"); return 1; }
    if (h->J < 12) { set_error("gem_calculate_errors_chunks: the hip-midpoint error needs joints 7 and 11 (n_joints >= 12)"); return 1; }
    if (n_chunks < 0 || frames_per_chunk < 1) { set_error("gem_calculate_errors_chunks: need n_chunks >= 0 and frames_per_chunk >= 1"); return 1; }
    if (n_chunks == 0) return 0;
    if (!d_est || !d_mid || !d_opt || !d_gt || !h_bone_mm || !d_out) { set_error("gem_calculate_errors_chunks: null argument"); return 1; }
    GEM_HIP(hipSetDevice(h->cfg.device));
    const size_t per = (size_t)(11 + MAXJ_ERR) * frames_per_chunk;
    if (post_scratch(h, per * n_chunks)) return 1;           // (every sequence its own scratch: blockIdx.y picks the sequence)
    if (n_chunks > 65535) { set_error("gem_calculate_errors_chunks: at most 65535 sequences per call"); return 1; }
    return launch_errors(h, d_est, d_mid, d_opt, d_gt, frames_per_chunk, h_bone_mm, h->post_work, d_out, (hipStream_t)stream, n_chunks);
}

int gem_lift_skeleton(gem_handle* h, const float* d_heat, const double* d_depth, int n_frames, const double* h_poly_c2w,
                      int n_poly_c2w, int upscale, int pad_x, int pad_y, double* d_out64, float* d_out32, void* stream) {
    if (!h) { set_error("gem_lift_skeleton: null handle"); return 1; }
    if (n_frames < 0 || n_poly_c2w < 1 || n_poly_c2w > GEM_MAX_POLY || upscale < 1 || pad_x < 0 || pad_y < 0) {
        set_error("gem_lift_skeleton: need n_frames >= 0, 1 <= n_poly_c2w <= 16, upscale >= 1, pads >= 0"); return 1;
    }
    if (n_frames == 0) return 0;
    if (!d_heat || !d_depth || !h_poly_c2w || (!d_out64 && !d_out32)) { set_error("gem_lift_skeleton: null argument"); return 1; }
    GEM_HIP(hipSetDevice(h->cfg.device));
    return launch_lift(h, d_heat, d_depth, n_frames, h_poly_c2w, n_poly_c2w, upscale, pad_x, pad_y, d_out64, d_out32,
                       (hipStream_t)stream);
}

int gem_profile_enable(gem_handle* h, int on) {
    if (!h) { set_error("gem_profile_enable: null handle"); return 1; }
    h->prof.on = on != 0;
    return 0;
}

int gem_profile_kernels(gem_handle* h, int family, char* buf, int buf_len) {
    if (!h || family < 0 || family > 3 || !buf || buf_len < 1) { set_error("gem_profile_kernels: bad argument"); return 1; }
    std::string out;
    for (const auto& n : h->prof.names[family]) {
        if (!out.empty()) out += "; ";
        out += n;
    }
    h->prof.names[family].clear();
    snprintf(buf, (size_t)buf_len, "%s", out.c_str());
    return 0;
}

int gem_profile_read(gem_handle* h, int family, double* total_ms, int64_t* n_launches, double* flops) {
    if (!h || family < 0 || family > 3) { set_error("gem_profile_read: bad argument"); return 1; }
    GEM_HIP(hipSetDevice(h->cfg.device));
    Profile& p = h->prof;
    // fold finished event pairs into the totals (caller has synchronised the stream)
    std::vector<int> nlog;
    for (auto& r : p.recs) {
        float ms = 0.f;
        GEM_HIP(hipEventSynchronize(r.b));
        GEM_HIP(hipEventElapsedTime(&ms, r.a, r.b));
        if (r.log_idx >= 0) {          // compacted round: FLOPs of the rows that were actually active
            if (nlog.empty()) {
                nlog.resize(N_LOG);
                GEM_HIP(hipMemcpy(nlog.data(), h->ws.n_log, (size_t)N_LOG * sizeof(int), hipMemcpyDeviceToHost));
            }
            if (h->ws.log_pos - r.log_idx <= N_LOG) r.flops = r.flops_per_window * nlog[r.log_idx % N_LOG];
        }
        static const char* dump = dev_env("GEM_PROFILE_DUMP");          // developer aid: one line per timed launch
        if (dump) {
            if (FILE* f = fopen(dump, "a")) {
                fprintf(f, "%d %d %.3f\n", r.family, r.log_idx >= 0 && !nlog.empty() ? nlog[r.log_idx % N_LOG] : -1, ms * 1e3);
                fclose(f);
            }
        }
        p.total_ms[r.family] += ms;
        p.n[r.family] += 1;
        p.flops[r.family] += r.flops;
        (void)hipEventDestroy(r.a);
        (void)hipEventDestroy(r.b);
    }
    p.recs.clear();
    if (total_ms) *total_ms = p.total_ms[family];
    if (n_launches) *n_launches = p.n[family];
    if (flops) *flops = p.flops[family];
    p.total_ms[family] = 0; p.n[family] = 0; p.flops[family] = 0;
    return 0;
}

}  // extern "C"
