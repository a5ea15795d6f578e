// Internal declarations shared by the HIP translation units of libgem_hip.so (gfx950 only).
#pragma once
// gfx950: a packed fp32 VALU instruction whose op_sel takes the high half of its second source for the low result (the SLP vectoriser
// forms them) returns a wrong low result in lanes 48-63 while another wavefront's bf16 MFMA executes on the same SIMD (DESIGN.md
// section 4; tools/slp_hazard/pk_mfma_repro.hip reproduces it in isolation).  Every translation unit of this library is therefore
// compiled with packed fp32 arithmetic switched off; a build that forgets the flag must not compile.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(GEM_NO_PACKED_FP32)
#error "compile with: -Xclang -target-feature -Xclang -packed-fp32-ops -DGEM_NO_PACKED_FP32 (__graft_entry__.NO_PACKED_FP32)"
#endif

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>
#include <set>
#include <string>
#include <vector>

#include "../../include/gem_hip.h"

namespace gem {

constexpr int PAD = 64;                    // every feature dimension is zero-padded to a multiple of 64
constexpr float LEAKY_SLOPE = 0.01f;       // torch.nn.LeakyReLU default (SeqConvVAE.py:38,77,90)
constexpr double BN_EPS = 1e-5;            // torch.nn.BatchNorm1d default
constexpr int MAX_HIST = 128;              // capacity of the per-window (s,y) ring
constexpr int MAXJ_ERR = GEM_MAX_JOINTS;   // joints handled by the error-report kernels (errors.hip)
constexpr int N_LOG = 1 << 16;             // ring of active-window counts kept for the profiling hook
constexpr int TRACE_ROUNDS = 64;           // evaluation rounds whose closure values are kept per stage (gem_read_trace)

inline int pad64(int x) { return (x + PAD - 1) / PAD * PAD; }

void set_error(const std::string& msg);
// Developer switches (A/B runs, sweeps, forcing a kernel path in the tests) are read from the environment ONLY when GEM_DEV=1 is
// set as well: a stray variable in a production environment cannot change which kernels evaluate the network.
const char* dev_env(const char* name);
bool hip_ok(hipError_t e, const char* what);
#define GEM_HIP(call) do { if (!gem::hip_ok((call), #call)) return 1; } while (0)

// epilogues of the GEMM kernel
enum Epi { EPI_BIAS = 0, EPI_BIAS_LRELU = 1, EPI_MASK = 2, EPI_NONE = 3 };

// One dense layer as the MFMA kernel sees it:  C[M,N] = epi( sum_tap shift_tap(A)[M,K] * W[tap][N][K]^T + bias ).
struct Layer {
    int taps = 1;          // 1 = linear, 3 = temporal conv (k=3, s=1, p=1)
    int K = 0, N = 0;      // padded
    float* w = nullptr;    // device, [taps][N][K], k contiguous
    float* bias = nullptr; // device, [N] (forward layers only)
    float* w4 = nullptr;   // device, [taps][K/4][N][4]: layout of the fused decoder-tail kernel (conv layers only)
    uint16_t* wb_hi = nullptr;   // device, bf16 [taps][N][K]: bf16(w)                       (precision modes 1, 2)
    uint16_t* wb_lo = nullptr;   // device, bf16 [taps][N][K]: bf16(w - float(bf16(w)))      (precision mode 1)
};

struct StageNet {
    bool loaded = false;
    std::vector<Layer> enc;        // conv blocks of the encoder (all + LeakyReLU)
    Layer fc;                      // [mu | logvar] = flat @ W^T + b, N = 2*Dp
    Layer dec_in;                  // decoder_input, N = T*Cp (time-major: n = t*Cp + c)
    std::vector<Layer> dec;        // decoder convs, last one without activation
    Layer dec_in_bwd;              // backward-data twins (no bias)
    std::vector<Layer> dec_bwd;    // dec_bwd[i] is the adjoint of dec[i]
    // decoder_input followed by the first decoder conv (no activation in between: SeqConvVAE.py:62,67-75,131-135) composed into
    // ONE linear layer z -> pre-activation of conv 0, N = T * pad64(C1) (n = t * C1p + c), K = Dp, and its transpose
    // (see compose_front in gem_api.hip).  Empty (w == nullptr) when the fused tail does not start at conv 1.
    Layer front, front_bwd;
    int tail_start = -1;           // decoder convs [tail_start, end) run in the fused tail kernel (-1: none)
    size_t tail_lds = 0;
    // bf16 multi-window tail (tail_bf16.hip): the same layers' weights as bf16 MFMA fragments in the order each of the 8 waves
    // consumes them, forward layers then adjoint layers: [8][tb_steps_f + tb_steps_b][64 lanes][8 bf16]; nullptr: not available
    uint16_t* tb_stream = nullptr;
    int tb_steps_f = 0, tb_steps_b = 0;
    size_t tb_lds = 0;
    std::vector<std::vector<float>> host_fwd, host_bwd;   // [3][N][K] padded fp32 weights of the decoder convs / their adjoints (load time only)
    std::vector<void*> allocs;
};

// per-window scalar state of the L-BFGS / strong-Wolfe machine (doubles: torch keeps these as python
// floats or 0-dim tensors; see lbfgs.hip)
struct alignas(16) LbfgsState {
    int phase, n_iter, evals, ls_iter, ls_evals, max_ls, first_bracket, ls_done, insuf, low, high;
    int hist_count, hist_start, nan_seen;      // nan_seen: a closure value was NaN (degenerate projection), latched
    double loss, prev_loss, t, gtd, d_norm, H_diag;
    double t_prev, f_prev, gtd_prev;
    double br_t[2], br_f[2], br_gtd[2];
    double ro[MAX_HIST];
    double cadj[MAX_HIST];     // s_k . y_(k+1) of ring slot k (the next newer pair): lets the two-loop recursion take two pairs per reduction
};

// Where a split-K GEMM left its partial slabs for a consumer that sums them itself (the fused tail stages the wide
// conv's output this way, lbfgs_advance its gradient rows): saves the reduce launch between producer and consumer.
struct SlabSrc {
    const float* base = nullptr;   // nullptr: nothing deferred, the consumer reads the finished matrix
    int nslab = 0;                 // static cut: slab z starts at base + z * stride
    size_t stride = 0;
    int dyn_W = 0, n_tiles = 0, ldc = 0, CT = 0;      // device-adaptive cut (dyn_W > 0): see dyn_split
    const int* m_dev = nullptr;
};

struct Workspace {
    int Bmax = 0;
    // activations, all [B*T, Cpad] float
    float* pose_p = nullptr;            // encoder input [B*T, 64]
    std::vector<float*> enc_act;        // outputs of the encoder convs
    float* mulv = nullptr;              // [B, 2*Dp]
    float* h0 = nullptr;                // [B, T*Cp_top]
    std::vector<float*> dec_act;        // outputs of decoder convs (last = Xp [B*T, 64])
    std::vector<float*> dec_grad;       // gradient w.r.t. the input of decoder conv i
    float* dXp = nullptr;               // [B*T, 64]
    float* dz = nullptr;                // [B, Dp]
    // bf16 twins of the decoder activations / gradients: the "bf16 VAE decoder" mode keeps them in bf16 in HBM
    // (gemm_glds.h); the decoded pose, the latent gradient and the L-BFGS state stay fp32
    uint16_t* trial_b = nullptr;        // [B, Dp] bf16 copy of `trial` (written by lbfgs_advance / f32_to_bf16)
    uint16_t* h0_b = nullptr;           // [B*T, topp]
    std::vector<uint16_t*> dec_act_b;   // outputs of decoder convs but the last
    std::vector<uint16_t*> dec_grad_b;  // gradient w.r.t. the input of decoder conv i
    uint16_t* dXp_b = nullptr;          // [B*T, 64]
    uint16_t* zero16 = nullptr;         // a line of zeros: DMA source of padded / absent rows
    // L-BFGS vectors, each [B, Dp]
    float *x = nullptr, *d = nullptr, *g = nullptr, *gp = nullptr, *bg0 = nullptr, *bg1 = nullptr, *trial = nullptr;
    float *S = nullptr, *Y = nullptr;   // [B, hist_cap, Dp]
    int hist_cap = 0;
    unsigned long long* lbfgs_clk = nullptr;      // developer aid (GEM_LBFGS_CLK): per-phase clock sums of lbfgs_advance (-DGEM_LB_PROBE builds)
    LbfgsState* state = nullptr;        // [B]
    int* phase = nullptr;               // [B] copy of state[b].phase for the compaction scan
    double* f = nullptr;                // [B] energy of the trial point
    double* parts = nullptr;            // [B,5]
    int* tex_key = nullptr;             // [B, T*J] / float [B, T*J, 4]: heat-map texels of the last evaluation (energy_device.h)
    float* tex_val = nullptr;
    bool tex_on = false;                // energy_args() hands the cache to the kernels (inside a stage only)
    double* trace = nullptr;            // [TRACE_ROUNDS][Bmax] closure value each window consumed in round r of the last stage (NaN: none)
    int round = -1;                     // evaluation round being enqueued (-1: outside the rounds)
    hipEvent_t mid_event = nullptr;     // two lanes: recorded behind the tail / energy kernel of the round being enqueued (then reset)
    // pipeline scratch
    float* pose_a = nullptr;            // [B,T,J,3] gathered local poses / stage outputs
    float* pose_b = nullptr;
    float* splitk = nullptr;            // partial slabs of the split-K GEMM launches
    bool defer_reduce = false;          // next launch_gemm: leave the slabs to the consumer, describe them in `deferred`
    SlabSrc deferred;
    SlabSrc grad_slab;                  // dE/dz of the current round as left by the decoder_input backward product
    size_t splitk_elems = 0;
    // active-window compaction (lbfgs.hip compact_kernel): windows still iterating occupy slots [0, n_active)
    int* perm = nullptr;                // [B] slot -> window
    int* slot_of = nullptr;             // [B] window -> slot
    int* n_active = nullptr;            // [2] = {n_active, n_active*T}
    // Slots handed out by lbfgs_advance itself (bf16 decoder mode with the fused tail): a window
    // that keeps iterating takes the next free slot of the coming round with one atomic add, so the rounds need no compact_kernel
    // launch.  Two (perm, slot_of) buffer pairs alternate by round; a round's count lives in its n_log entry (zeroed at stage begin).
    // `perm`, `slot_of`, `n_active` above always point at the CURRENT round's set; *_home are the allocations they return to.
    int *perm2 = nullptr, *slot_of2 = nullptr, *perm_home = nullptr, *slot_of_home = nullptr, *n_active_home = nullptr;
    int *next_perm = nullptr, *next_slot_of = nullptr, *next_count = nullptr;      // what lbfgs_advance of this round fills (nullptr: off)
    // EXPERIMENT (GEM_DEV=1 GEM_FUSE_BWD_LBFGS=1, fp32, <= 256 windows, eager launches only): the backward front product and
    // lbfgs_advance as ONE kernel with a device-wide barrier between them -- what a grid barrier costs against a launch boundary
    // in this pipeline.  fuse_lbfgs: the round's options while such a launch is wanted; lbfgs_fused_done:
    // the product's launch carried the advance; grid_bar: monotonic arrival counter, grid_bar_target: its value after this launch.
    const gem_lbfgs_opts* fuse_lbfgs_req = nullptr;      // set by the round loop, handed to the backward launch only (evaluate)
    const gem_lbfgs_opts* fuse_lbfgs = nullptr;
    bool lbfgs_fused_done = false;
    unsigned* grid_bar = nullptr;
    unsigned grid_bar_target = 0;
    bool dyn = false;                   // rounds in flight: GEMM / energy launches read their row count from n_active
    int* n_log = nullptr;               // [N_LOG] n_active after every compaction (profiling: true row counts)
    long log_pos = 0, cur_log = -1;
    bool fuse_compact = false;          // next decoder_input forward launch re-packs the active windows itself (gemm_rows.h)
    int* fuse_log = nullptr;            // ... and logs n_active here
    int done_phase = 3;                 // lbfgs.hip PH_DONE
    std::vector<void*> allocs;
};

// "done once" flag per device ordinal: hipFuncSetAttribute is a per-device setting and a process may own handles on
// several devices (one thread per handle; a handle itself is not thread-safe)
struct PerDeviceOnce {
    std::atomic<bool> done[64] = {};
    // exchange: exactly one of several threads (one per handle) sees "need" for a device; the setting itself is idempotent
    bool need(int dev) { if (dev < 0 || dev >= 64) return true; return !done[dev].exchange(true); }
};

struct Profile {
    bool on = false;
    struct Rec { hipEvent_t a, b; int family; double flops; long log_idx = -1; double flops_per_window = 0; };
    std::vector<Rec> recs;
    double total_ms[4] = {0, 0, 0, 0};
    int64_t n[4] = {0, 0, 0, 0};
    double flops[4] = {0, 0, 0, 0};      // family 3 (input lifting) counts algorithmic BYTES here
    // names (as rocprofv3 prints them) of the kernels launched for each timed family since the last gem_profile_kernels call:
    // bench.py labels its roofline objects with what actually ran, not with what it expects to run
    std::set<std::string> names[4];
    std::set<std::string> pending;       // kernels launched since the current timed launcher began
};

// Wavefront reductions on the DPP cross-lane path (no LDS traffic): quad swaps, half-row and row mirrors
// leave every lane of a 16-lane row with the row total; the four row totals are read back with
// v_readlane and combined in a fixed order.  Doubles travel as two 32-bit halves.
template <int CTRL>
__device__ __forceinline__ double dpp_move(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    // (mov_dpp = update_dpp with an undefined "old" operand: every lane has a valid source for these permutes, and the
    // compiler does not have to copy the source into the destination first)
    lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xf, 0xf, true);
    hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double read_lane(double v, int lane) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane),
                            __builtin_amdgcn_readlane(__double2loint(v), lane));
}
__device__ __forceinline__ double nan_max(double a, double b) { return (b > a || b != b) ? b : a; }   // NaN wins, like torch's max

__device__ __forceinline__ double wave_sum_dpp(double v) {
    v += dpp_move<0xB1>(v);      // quad_perm [1,0,3,2]
    v += dpp_move<0x4E>(v);      // quad_perm [2,3,0,1]
    v += dpp_move<0x141>(v);     // row_half_mirror
    v += dpp_move<0x140>(v);     // row_mirror
    return ((read_lane(v, 0) + read_lane(v, 16)) + read_lane(v, 32)) + read_lane(v, 48);
}
__device__ __forceinline__ double wave_max_dpp(double v) {
    v = nan_max(v, dpp_move<0xB1>(v));
    v = nan_max(v, dpp_move<0x4E>(v));
    v = nan_max(v, dpp_move<0x141>(v));
    v = nan_max(v, dpp_move<0x140>(v));
    return nan_max(nan_max(nan_max(read_lane(v, 0), read_lane(v, 16)), read_lane(v, 32)), read_lane(v, 48));
}

}  // namespace gem

namespace gem {
// hipGraph cache of whole optimisation calls (gem_graph_enable): a call is identified by everything that is baked into its
// kernel arguments -- entry point, batch size, precision, every caller pointer, the energy weights and optimiser options
struct GraphKey {
    int kind = 0, stage = 0, B = 0, precision = 0;
    bool tex_cache = true;
    int lanes = 1;
    const void* ptr[12] = {};
    gem_energy_weights w[2] = {};
    gem_lbfgs_opts opt = {};
    void* stream = nullptr;
};
struct GraphEntry {
    GraphKey key;
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;      // nullptr: seen once (ran eagerly), captured on the next identical call
    uint64_t last_use = 0;
};
}  // namespace gem

struct gem_handle {
    gem_config cfg;
    int T, J, C, Cp, D, Dp, top, topp;
    gem::StageNet net[2];
    gem::Workspace ws;
    gem::Profile prof;
    int precision = 0;             // GEM_PRECISION_*
    int n_cu = 256;                // compute units of the device (hipDeviceAttributeMultiprocessorCount)
    double* post_work = nullptr;   // scratch of the post-processing calls (errors.hip), grown on demand
    size_t post_work_elems = 0;
    bool tex_cache = true;         // gem_set_texel_cache
    bool graphs_on = false;        // gem_graph_enable
    std::vector<gem::GraphEntry> graphs;
    uint64_t graph_tick = 0;
    int64_t graph_replays = 0, graph_captures = 0;
    int* d_parents = nullptr;
    int* d_children = nullptr;     // [J][J] child lists, -1 terminated
    // two lanes (gem_api.hip windows_dual): a second handle with its own workspace that shares this handle's weights
    gem_handle* lane2 = nullptr;
    hipStream_t lane_stream = nullptr, lane_stream_a = nullptr;      // both lanes run on streams of their own (non-blocking: the
                                                                     // caller's may be the legacy default stream, whose implicit
                                                                     // synchronisation with every blocking stream of the process
                                                                     // would sit between the lanes' kernels)
    hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_join_a = nullptr;
    std::vector<hipEvent_t> ev_pool;
    int lanes_min = 0;             // gem_set_lanes: batches of at least this many windows run as two lanes (0: never = the default
                                   // since round 4: with two tail workgroups per CU one lane is the faster arrangement)
    int last_split = 0;            // windows in the first lane of the last gem_optimize_windows call (0: one lane)
    uint64_t cfg_gen = 1;          // bumped by gem_load_vae / gem_set_precision / gem_set_texel_cache: what a second lane mirrors
    uint64_t lane_gen = 0;         // ... and the generation the second lane was last synchronised with
};

namespace gem {

// two lanes: the half-round marker of the round being enqueued (Workspace::mid_event), recorded behind the tail / energy kernel
inline int record_mid(gem_handle* h, hipStream_t s) {
    if (!h->ws.mid_event) return 0;
    const hipError_t e = hipEventRecord(h->ws.mid_event, s);
    h->ws.mid_event = nullptr;
    return gem::hip_ok(e, "hipEventRecord(mid)") ? 0 : 1;
}

// profiling hook: remember the (demangled) name of a kernel about to be launched; no-op unless event profiling is on
void note_kernel(gem_handle* h, const void* host_fn);
inline void commit_kernel_names(gem_handle* h, int family) {
    if (family >= 0 && family < 4) h->prof.names[family].insert(h->prof.pending.begin(), h->prof.pending.end());
    h->prof.pending.clear();
}

// ---- kernel launchers (each enqueues on `s`, returns 0/1) -------------------------------------------
int launch_gemm(gem_handle* h, const Layer& L, int epi, const float* A, int lda, const float* aux, float* Cout, int ldc,
                int M, int T, hipStream_t s, int family, const int* row_map = nullptr);

int pick_splitk(const gem_handle* h, long blocks, int n_tiles, size_t slab_elems);

// Device-adaptive split-K of the evaluation rounds: the launch has W workgroups; with M rows still active
// there are RT = ceil(M/BM) row tiles, and the K walk of every (row tile, column tile) is cut into as many
// slices as the W workgroups allow (at least 4 k-tiles per slice).  Used identically by the GEMM kernels and
// by splitk_reduce_kernel.  All M == M_host rounds reproduce the static configuration.
struct DynSplit { int RT, SK, per; size_t slab; };
__host__ __device__ inline DynSplit dyn_split(int M, int BM, int CT, int W, int n_tiles, int ldc) {
    DynSplit d;
    d.RT = (M + BM - 1) / BM;
    if (d.RT < 1) d.RT = 1;
    int cap = W / (d.RT * CT);
    if (cap < 1) cap = 1;
    int mx = n_tiles / 4;
    if (mx < 1) mx = 1;
    const int sk = cap < mx ? cap : mx;
    d.per = (n_tiles + sk - 1) / sk;
    d.SK = (n_tiles + d.per - 1) / d.per;
    d.slab = (size_t)d.RT * BM * ldc;
    return d;
}
__device__ inline void slab_layout(const SlabSrc& s, int& nslab, size_t& stride) {
    nslab = s.nslab;
    stride = s.stride;
    if (s.dyn_W > 0) {
        const DynSplit d = dyn_split(*s.m_dev, 64, s.CT, s.dyn_W, s.n_tiles, s.ldc);
        nslab = d.SK;
        stride = d.slab;
    }
}
bool rows_can_fuse_compaction(const gem_handle* h, const Layer& L, int lda, int ldc, int B, bool slabs);
bool bf16_rounds_take_slots_atomically(const gem_handle* h, int stage, int B);      // decoder_bf16.hip: fused bf16 tail right behind the composed front layer
int launch_splitk_reduce(gem_handle* h, int epi, int nslab, size_t slab, const float* bias, const float* aux, float* C, int M, int N,
                         int ldc, const int* m_dev, hipStream_t s, int dyn_W = 0, int n_tiles = 0);
// bf16-input MFMA variant of launch_gemm (gemm_bf16.hip): nprod = 1 (plain bf16) or 3 (hi/lo split, fp32-grade)
int launch_gemm_bf16(gem_handle* h, const Layer& L, int epi, int nprod, const float* A, int lda, const float* aux, float* Cout,
                     int ldc, int M, int T, hipStream_t s, const int* row_map);

// bf16-activation decoder path (decoder_bf16.hip): one evaluation = decode + energies + backward-data, like evaluate()
struct EnergyArgs;
int evaluate_bf16(gem_handle* h, int stage, int B, const EnergyArgs& ea, hipStream_t s, bool forward_only);
int launch_f32_to_bf16(const float* src, uint16_t* dst, size_t n, hipStream_t s);
int launch_f32_split_bf16(const float* src, uint16_t* hi, uint16_t* lo, size_t n, hipStream_t s);

int launch_pack_pose(const float* src, float* dst, int rows, int C, hipStream_t s);      // [rows,C] -> [rows,64]
int launch_unpack_pose(const float* src, float* dst, int rows, int C, hipStream_t s);    // [rows,64] -> [rows,C]
int launch_reparam(const float* mulv, const float* eps, float* mu, float* logvar, float* z, float* z2, int B, int D, int Dp,
                   hipStream_t s);
int launch_pad_latent(const float* z, float* zp, int B, int D, int Dp, hipStream_t s);
int launch_unpad_latent(const float* zp, float* z, int B, int D, int Dp, hipStream_t s);

struct EnergyArgs {
    const float* Xp;          // [B*T, 64] decoded pose (padded rows)
    const float* X0;          // [B,T,J,3] stage input pose
    const float* heat;        // [F,H,W,J] or nullptr
    const int32_t* frame0;    // [B]
    const float* mean_bone;   // [B,J]
    float* dXp;               // [B*T, 64]
    uint16_t* dXp_b;          // bf16 gradient rows instead of dXp (stand-alone energy kernel, bf16 decoder mode) or nullptr
    double* f;                // [B]
    double* parts;            // [B,5]
    float w3d, ws, wb, wv, wr;
    double dw3d, dws, dwb, dwv, dwr;
    int T, J, H, W, n_poly;
    float poly[GEM_MAX_POLY];
    float cx, cy;
    const int* parents;
    const int* children;
    int* tex_key;             // [B, T*J] texel-block cache of the reprojection term: key of the cached 2x2 block (-1: empty) ...
    float* tex_val;           // [B, T*J, 4] ... and its four texels (nw, ne, sw, se); nullptr: no cache
    const int* n_dev;         // device count of active slots (nullptr: all B)
    const int* perm;          // slot -> window (nullptr: identity); X / dX rows are slot-ordered, the rest window-ordered
};
int launch_energy(gem_handle* h, const EnergyArgs& a, int B, hipStream_t s);

// fused decoder tail (tail.hip)
constexpr int TAIL_MAX_LAYERS = 6;
struct TailLayerDev { const float* w4; const float* bias; int K, N; };
struct TailArgs {
    int n, B, G, forward_only, escr, mask_first;
    int waves, rows;         // kernel shape (8 wavefronts, one workgroup per CU | 4, three per CU) and rows per LDS buffer (16 | G*T): plan_tail
    SlabSrc in_slab;         // a_in still lies in split-K slabs (+ in_bias, LeakyReLU to apply) when in_slab.base != nullptr
    const float* in_bias;    // bias of row r, column c: in_bias[(r % T) * in_bias_ld + c]
    int in_bias_ld;          // 0: one bias per channel (a conv produced a_in); K0: per (frame, channel) (the composed front layer)
    long long* dbg_ts;       // developer probe (tools/tail_bench): [32] {shader clock, 100 MHz wall clock} pairs of workgroup 0, or nullptr
    TailLayerDev fwd[TAIL_MAX_LAYERS], bwd[TAIL_MAX_LAYERS];
    const float* a_in;       // [B*T, K0] input activation of the first fused layer
    float* g_out;            // [B*T, K0] gradient w.r.t. its pre-activation
    uint16_t* g_out_b;       // the same as bf16 (bf16 decoder mode: consumed by the bf16 backward GEMMs) or nullptr
    float* Xp;               // [B*T, 64] decoded pose
    int off_act[TAIL_MAX_LAYERS + 1], ld_act[TAIL_MAX_LAYERS + 1], off_g[2], ld_g, off_red, off_escr, off_zero, off_pre;   // LDS plan (floats); off_pre: per-window inputs of the energy terms (G == 1)
    EnergyArgs e;
};
size_t plan_tail(const std::vector<Layer>& dec, int start, int T, int J, TailArgs* out, bool shared = false);
size_t plan_tail_for(const gem_handle* h, const std::vector<Layer>& dec, int start, int wgs, TailArgs* out);   // shape chosen for `wgs` workgroups

// bf16 multi-window fused tail (tail_bf16.hip): nrt = 1 .. 5 row tiles of 16 rows = G = min(8, 16 nrt / T) windows per workgroup,
// up to two workgroups per CU (<= 80 KB of LDS, <= 128 VGPRs)
constexpr int TB_MAX_LAYERS = 6;
struct TailB16Layer { int K, N; const float* bias; };
struct TailB16Args {
    int n, B, G, nrt, forward_only, mask_first;      // nrt: 16-row tiles per workgroup (1 .. 5); G = min(8, 16 nrt / T) windows
    SlabSrc in_slab;           // the input still lies in fp32 split-K slabs (+ in_bias, LeakyReLU to apply) when in_slab.base != nullptr
    const float* in_bias;      // bias of row r, column c: in_bias[(r % T) * in_bias_ld + c]
    int in_bias_ld;
    const uint16_t* a_in_b;    // [B*T, K0] bf16 input activation (post-LeakyReLU) when there are no slabs
    uint16_t* g_out_b;         // [B*T, K0] bf16 gradient w.r.t. the input's pre-activation
    float* Xp;                 // [B*T, 64] decoded pose (fp32) or nullptr
    long long* dbg_ts;         // developer probe (tools/tail16_bench): {shader clock, 100 MHz wall clock} pairs of workgroup 0, or nullptr
    const uint16_t* wstream;   // StageNet::tb_stream
    int steps_f, steps_total;  // steps (1 KB fragments) per wave: forward part / forward + adjoint
    TailB16Layer fwd[TB_MAX_LAYERS], bwd[TB_MAX_LAYERS];
    // LDS plan, byte offsets / row strides in bytes (row stride = 2 * width + 32: conflict-free ds_read_b128 fragment reads).
    // Two ping-pong buffers: act[j] and, in the backward direction, the gradient w.r.t. act[j] both live in buffer j & 1 (the
    // activations are dead by then: LeakyReLU' comes from one sign BIT per element, kept in mask[j]); the decoded pose (fp32, dense
    // per window) sits above the activations of act[n-1]'s buffer.
    int off_act[TB_MAX_LAYERS + 1], ld_act[TB_MAX_LAYERS + 1];      // act[0] = input; [n]: the gradient rows w.r.t. the pose (bf16)
    int off_mask[TB_MAX_LAYERS], ld_mask[TB_MAX_LAYERS];            // mask[0]: one byte per 8 channels; mask[j > 0]: one byte per 4
    int off_x, escr, off_mb, off_zero, off_tab;
    int off_bwin, off_epair;   // [G] global window indices; (up to three row tiles, 10 x 15 windows) [G * T*J][5] fp32 energy terms per pair, else -1
    EnergyArgs e;
};
size_t plan_tail_bf16(const std::vector<Layer>& dec, int start, int T, int J, TailB16Args* out, int nrt = 5);
int tail_bf16_row_tiles(const gem_handle* h, int B, int T);
int build_tail_bf16_stream(gem_handle* h, StageNet& net);
int launch_tail_bf16(gem_handle* h, const TailB16Args& a, size_t lds_bytes, hipStream_t s);
int launch_tail(gem_handle* h, const TailArgs& a, size_t lds_bytes, hipStream_t s);
int tail_cap_workgroups(const gem_handle* h, const std::vector<Layer>& dec, int start);   // fused tail up to this many workgroups
int launch_mean_bone(gem_handle* h, const float* pose, int n_frames, float* out, hipStream_t s);
int launch_gather_windows(const float* frames, const int32_t* frame0, float* out, int B, int T, int JC, hipStream_t s);
int launch_relative_global(const float* local, const double* cams, const int32_t* frame0, float* rel, int B, int T, int J,
                           hipStream_t s);
int launch_to_global(const float* rel, const double* cams, const int32_t* frame0, double* out, int B, int T, int J,
                     hipStream_t s);

// sequence post-processing (errors.hip)
int launch_errors(gem_handle* h, const double* est, const double* mid, const double* opt, const double* gt, int F,
                  const double* bone_mm, double* frame_out, double* out, hipStream_t s, int n_seq);
int launch_merge(const double* win, double* tmp, double* out, int n_chunks, int wpc, int T, int JC, int overlap, int smooth,
                 hipStream_t s);
size_t errors_frame_lds_bytes(int J);
int launch_lift(gem_handle* h, const float* heat, const double* depth, int F, const double* poly, int n_poly, int up, int pad_x,
                int pad_y, double* out64, float* out32, hipStream_t s);

int launch_lbfgs_init(gem_handle* h, int B, const gem_lbfgs_opts& o, hipStream_t s);
int launch_lbfgs_advance(gem_handle* h, int B, const gem_lbfgs_opts& o, hipStream_t s);
int launch_lbfgs_stats(gem_handle* h, int B, gem_window_stats* out, hipStream_t s);
int launch_compact(gem_handle* h, int B, int force_all, hipStream_t s, int zero_after = 0);
namespace rows { struct Args; }
// lbfgs.hip (experiment): gemm_rows_body<4, 5> + device-wide barrier + lbfgs_advance in one launch; -1: shape not covered
int launch_rows_bwd_lbfgs(gem_handle* h, const rows::Args& ra, int rows_grid, size_t rows_smem, const SlabSrc& gslab, hipStream_t s);


}  // namespace gem
