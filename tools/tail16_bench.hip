// Stand-alone timing harness for csrc/tail_bf16.hip (developer tool): per-phase timestamps of workgroup 0 and launch times.
//   hipcc -w --offload-arch=gfx950 -O3 -std=c++17 -Xclang -target-feature -Xclang -packed-fp32-ops -DGEM_NO_PACKED_FP32 -o tools/tail16_bench tools/tail16_bench.hip && ./tools/tail16_bench 1536 [nslab]
//   TAIL_DETERMINISM=1 [TAIL_HEAT=1] [TAIL_SCRAMBLE=1] ./tools/tail16_bench 8192: repeated launches must write bitwise identical rows
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../globalegomocap_amd/csrc/tail_bf16.hip"
namespace gem {
void set_error(const std::string& m) { fprintf(stderr, "error: %s\n", m.c_str()); }
const char* dev_env(const char*) { return nullptr; }
void note_kernel(gem_handle*, const void*) {}
bool hip_ok(hipError_t e, const char* what) { if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", what, hipGetErrorString(e)); return false; } return true; }
}
using namespace gem;
// fills every CU's LDS with launch-dependent garbage: a read of LDS the tail kernel has not written shows up as non-determinism
__global__ void lds_scramble_kernel(unsigned seed, unsigned* sink) {
    extern __shared__ unsigned sl[];
    unsigned v = seed * 2654435761u + blockIdx.x * 40503u + threadIdx.x;
    for (int i = threadIdx.x; i < 160 * 1024 / 4; i += blockDim.x) { v = v * 1664525u + 1013904223u; sl[i] = v; }
    __syncthreads();
    if (sl[(seed + threadIdx.x) % (160 * 1024 / 4)] == 0x12345678u) sink[0] = 1;
}
static std::vector<float> host_rand(size_t n, unsigned seed, float scale) {
    std::vector<float> h(n); srand(seed);
    for (size_t i = 0; i < n; ++i) h[i] = scale * ((rand() / (float)RAND_MAX) * 2.f - 1.f);
    return h;
}
static float* dev_rand(size_t n, unsigned seed, float scale) {
    std::vector<float> h = host_rand(n, seed, scale);
    float* d; hipMalloc(&d, n * 4); hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice); return d;
}
int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 1536, nslab = argc > 2 ? atoi(argv[2]) : 0, iters = 50, T = 10, J = 15;
    const int dims[6] = {256, 128, 64, 64, 64, 64};
    gem_handle h; h.prof.on = false; h.T = T; h.J = J; h.cfg.device = 0;
    StageNet net;
    net.dec.resize(6); net.dec_bwd.resize(6); net.host_fwd.resize(6); net.host_bwd.resize(6);
    net.dec[0].K = 512; net.dec[0].N = 256;
    for (int i = 1; i < 6; ++i) {
        net.dec[i].K = dims[i - 1]; net.dec[i].N = dims[i]; net.dec[i].taps = 3;
        net.dec_bwd[i].K = dims[i]; net.dec_bwd[i].N = dims[i - 1]; net.dec_bwd[i].taps = 3;
        net.host_fwd[i] = host_rand((size_t)3 * dims[i - 1] * dims[i], 10 + i, 0.05f);
        net.host_bwd[i] = host_rand((size_t)3 * dims[i - 1] * dims[i], 30 + i, 0.05f);
        net.dec[i].bias = dev_rand(dims[i], 20 + i, 0.05f);
    }
    if (getenv("TAIL_HEAT")) {
        std::vector<float> pb(64, 0.f);
        for (int j = 0; j < 15; ++j) { pb[3 * j] = 0.05f * (j - 7); pb[3 * j + 1] = 0.04f * (j % 5 - 2); pb[3 * j + 2] = 0.4f + 0.05f * j; }
        hipMemcpy((void*)net.dec[5].bias, pb.data(), 64 * 4, hipMemcpyHostToDevice);
    }
    net.tail_start = 1;
    if (build_tail_bf16_stream(&h, net) || !net.tb_stream) { fprintf(stderr, "no stream\n"); return 1; }
    TailB16Args a;
    gem_handle hh; hh.n_cu = 256;
    const int nrt = getenv("TAIL_NRT") ? atoi(getenv("TAIL_NRT")) : tail_bf16_row_tiles(&hh, B, T);
    const size_t lds = plan_tail_bf16(net.dec, 1, T, J, &a, nrt);
    printf("LDS %zu bytes, n=%d, row tiles %d, G=%d, steps %d + %d\n", lds, a.n, a.nrt, a.G, net.tb_steps_f, net.tb_steps_b);
    for (int i = 0; i < a.n; ++i) {
        a.fwd[i] = TailB16Layer{net.dec[1 + i].K, net.dec[1 + i].N, net.dec[1 + i].bias};
        a.bwd[i] = TailB16Layer{net.dec_bwd[1 + i].K, net.dec_bwd[1 + i].N, nullptr};
    }
    a.B = B; a.dbg_ts = nullptr; a.forward_only = 0;
    a.wstream = net.tb_stream; a.steps_f = net.tb_steps_f; a.steps_total = net.tb_steps_f + net.tb_steps_b;
    const size_t rows = (size_t)B * T;
    hipMalloc((void**)&a.a_in_b, rows * 256 * 2); hipMemset((void*)a.a_in_b, 0x3c, rows * 256 * 2);      // bf16 0x3c3c ~ 0.0115
    hipMalloc((void**)&a.g_out_b, rows * 256 * 2);
    a.Xp = nullptr;
    a.in_slab = SlabSrc{};
    a.in_bias = dev_rand((size_t)T * 256, 5, 0.05f); a.in_bias_ld = 256;
    if (nslab > 0) { a.in_slab.base = dev_rand(rows * 256 * nslab, 6, 0.5f); a.in_slab.nslab = nslab; a.in_slab.stride = rows * 256; }
    EnergyArgs& e = a.e;
    e = EnergyArgs{};
    e.X0 = dev_rand((size_t)B * T * 45, 2, 1.f);
    const int F = 8 * B + 10;
    hipMalloc((void**)&e.heat, (size_t)F * 64 * 64 * 15 * 4); hipMemset((void*)e.heat, 0, (size_t)F * 64 * 64 * 15 * 4);
    if (getenv("TAIL_HEAT")) {       // non-zero texels (bytes 0x3e.. = 0.18 .. 0.25 as fp32): the reprojection gradient is live
        std::vector<unsigned char> pat(1 << 20);
        srand(5); for (auto& v : pat) v = (unsigned char)(rand() & 0xFF);
        for (size_t i = 3; i < pat.size(); i += 4) pat[i] = 0x3e;
        for (size_t off = 0; off < (size_t)F * 64 * 64 * 15 * 4; off += pat.size())
            hipMemcpy((char*)e.heat + off, pat.data(), std::min(pat.size(), (size_t)F * 64 * 64 * 15 * 4 - off), hipMemcpyHostToDevice);
    }
    std::vector<int> f0(B); for (int b = 0; b < B; ++b) f0[b] = 8 * b;
    hipMalloc((void**)&e.frame0, B * 4); hipMemcpy((void*)e.frame0, f0.data(), B * 4, hipMemcpyHostToDevice);
    e.mean_bone = dev_rand((size_t)B * 15, 3, 0.3f);
    hipMalloc(&e.f, B * 8); hipMalloc(&e.parts, B * 40);
    e.w3d = e.ws = e.wb = 0.01f; e.wv = 0; e.wr = getenv("TAIL_NO_REPROJ") ? 0.f : 0.01f; e.dw3d = e.dws = e.dwb = 0.01; e.dwr = 0.01;
    e.T = T; e.J = J; e.H = 64; e.W = 64; e.n_poly = 11;
    const float poly[11] = {478.6f, 350.4f, 79.f, 62.3f, 32.6f, 15.7f, 7.77f, 2.19f, -0.108f, -0.19f, -0.0278f};
    for (int i = 0; i < 11; ++i) e.poly[i] = poly[i];
    e.cx = 659.7f; e.cy = 530.f;
    const int par[15] = {0, 0, 1, 2, 0, 4, 5, 1, 7, 8, 9, 4, 11, 12, 13};
    std::vector<int> ch(256, -1);
    for (int j = 0; j < 15; ++j) { int n = 0; for (int c = 0; c < 15; ++c) if (c != j && par[c] == j) ch[j * 16 + n++] = c; }
    int *dp, *dc; hipMalloc(&dp, 60); hipMemcpy(dp, par, 60, hipMemcpyHostToDevice);
    hipMalloc(&dc, 1024); hipMemcpy(dc, ch.data(), 1024, hipMemcpyHostToDevice);
    e.parents = dp; e.children = dc; e.n_dev = nullptr; e.perm = nullptr;
    if (getenv("TAIL_TEXCACHE")) { hipMalloc(&e.tex_key, rows * J * 4); hipMemset(e.tex_key, 0xFF, rows * J * 4); hipMalloc(&e.tex_val, rows * J * 16); }
    hipStream_t s; hipStreamCreate(&s);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    {   // phase timestamps of workgroup 0
        long long* d_ts; hipMalloc(&d_ts, 64 * 8); hipMemset(d_ts, 0, 64 * 8);
        a.dbg_ts = d_ts;
        for (int i = 0; i < 20; ++i) launch_tail_bf16(&h, a, lds, s);
        hipStreamSynchronize(s);
        long long ts[64]; hipMemcpy(ts, d_ts, sizeof(ts), hipMemcpyDeviceToHost);
        a.dbg_ts = nullptr;
        const char* names[14] = {"ring issued", "staged", "fwd 256->128", "fwd 128->64", "fwd 64->64", "fwd 64->64", "fwd 64->45",
                                 "energy", "bwd 45->64", "bwd 64->64", "bwd 64->64", "bwd 64->128", "bwd 128->256", "rows out"};
        for (int i = 1; i < 14; ++i)
            printf("  %-14s %7.2f us  (%6lld shader clocks, %.2f GHz)\n", names[i], (ts[2 * i + 1] - ts[2 * i - 1]) * 0.01,
                   ts[2 * i] - ts[2 * i - 2], (ts[2 * i] - ts[2 * i - 2]) / ((ts[2 * i + 1] - ts[2 * i - 1]) * 10.0 + 1e-9));
        printf("  total inside the kernel %.2f us\n", (ts[27] - ts[1]) * 0.01);
        printf("  energy sub-steps (shader clocks since its start):");
        for (int i = 1; i < 20 && ts[32 + i]; ++i) printf(" %lld", ts[32 + i] - ts[32]);
        printf("\n");
    }
    if (getenv("TAIL_DETERMINISM")) {   // repeated launches must write bitwise identical gradient rows
        std::vector<uint16_t> hin(rows * 256);
        srand(77);
        for (auto& v : hin) { float f = 0.5f * ((rand() / (float)RAND_MAX) * 2.f - 1.f); uint32_t u; memcpy(&u, &f, 4); v = (uint16_t)(u >> 16); }
        hipMemcpy((void*)a.a_in_b, hin.data(), hin.size() * 2, hipMemcpyHostToDevice);
        std::vector<uint16_t> ref(rows * 256), cur(rows * 256);
        int bad_launches = 0;
#ifdef GEM_TB_DEBUG_DUMP
        uint16_t* d_gd; hipMalloc(&d_gd, rows * 64 * 2); hipMemset(d_gd, 0, rows * 64 * 2);
        hipMemcpyToSymbol(HIP_SYMBOL(gem::tb::g_tb_dump_gd), &d_gd, sizeof(d_gd));
        float* d_xp; hipMalloc(&d_xp, rows * 64 * 4); hipMemset(d_xp, 0, rows * 64 * 4);
        a.Xp = d_xp;
        float* d_ep; hipMalloc(&d_ep, (size_t)B * 150 * 8 * 4); hipMemset(d_ep, 0, (size_t)B * 150 * 8 * 4);
        hipMemcpyToSymbol(HIP_SYMBOL(gem::g_ep_dump), &d_ep, sizeof(d_ep));
        std::vector<float> ep_ref((size_t)B * 150 * 8), ep_cur((size_t)B * 150 * 8);
        std::vector<uint16_t> gd_ref(rows * 64), gd_cur(rows * 64);
        std::vector<float> xp_ref(rows * 64), xp_cur(rows * 64);
#endif
        for (int rep = 0; rep < 12; ++rep) {
            hipMemsetAsync(a.g_out_b, 0xFF, rows * 256 * 2, s);
            if (getenv("TAIL_SCRAMBLE")) {
                static unsigned* sink = nullptr;
                if (!sink) { hipMalloc(&sink, 4); hipFuncSetAttribute(reinterpret_cast<const void*>(lds_scramble_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); }
                hipLaunchKernelGGL(lds_scramble_kernel, dim3(1024), dim3(256), 160 * 1024, s, (unsigned)rep + 1u, sink);
            }
            launch_tail_bf16(&h, a, lds, s);
            hipStreamSynchronize(s);
            hipMemcpy(rep ? cur.data() : ref.data(), a.g_out_b, rows * 256 * 2, hipMemcpyDeviceToHost);
#ifdef GEM_TB_DEBUG_DUMP
            hipMemcpy(rep ? gd_cur.data() : gd_ref.data(), d_gd, rows * 64 * 2, hipMemcpyDeviceToHost);
            hipMemcpy(rep ? xp_cur.data() : xp_ref.data(), d_xp, rows * 64 * 4, hipMemcpyDeviceToHost);
            hipMemcpy(rep ? ep_cur.data() : ep_ref.data(), d_ep, ep_cur.size() * 4, hipMemcpyDeviceToHost);
            if (rep) {
                size_t cnt[8] = {0}; int shown2 = 0;
                for (size_t i = 0; i < ep_cur.size(); ++i)
                    if (memcmp(&ep_cur[i], &ep_ref[i], 4)) {
                        ++cnt[i % 8];
                        if (shown2 < 4 && i % 8 == 3) {
                            const size_t b8 = i - 3;
                            printf("      fields ref:"); for (int k = 0; k < 8; ++k) printf(" %.9g", ep_ref[b8 + k]);
                            printf("\n      fields cur:"); for (int k = 0; k < 8; ++k) printf(" %.9g", ep_cur[b8 + k]);
                            printf("\n      rho*a+c = %.9g\n", (double)ep_ref[b8 + 5] * ep_ref[b8 + 0] + ep_ref[b8 + 1]);
                        }
                        if (shown2 < 4) { ++shown2; printf("    ep differs: window %zu pair %zu (lane %zu) field %zu: %g vs %g\n", i / 8 / 150, (i / 8) % 150, ((i / 8) % 150) & 63, i % 8, ep_ref[i], ep_cur[i]); }
                    }
                printf("  rep %d intermediates differing: a %zu c %zu inv %zu dudx %zu xx2 %zu rho %zu i3 %zu ux %zu\n", rep, cnt[0], cnt[1], cnt[2], cnt[3], cnt[4], cnt[5], cnt[6], cnt[7]);
            }
            if (rep) {
                size_t nx = 0, ng = 0; int lanehist[64] = {0}, triphist[3] = {0}, comphist[3] = {0}, shown = 0;
                for (size_t i = 0; i < xp_cur.size(); ++i) nx += memcmp(&xp_cur[i], &xp_ref[i], 4) != 0;
                for (size_t i = 0; i < gd_cur.size(); ++i)
                    if (gd_cur[i] != gd_ref[i]) {
                        ++ng;
                        const size_t row = i / 64; const int c = (int)(i % 64), t = (int)(row % T), j = c / 3, p = t * J + j;
                        if (c < 45) { ++lanehist[p & 63]; ++triphist[p >> 6]; ++comphist[c % 3]; }
                        if (shown < 6) { ++shown; printf("    gd differs: window %zu (wg %zu, win-in-wg %zu) t %d col %d (joint %d comp %d, pair %d = trip %d lane %d): %04x vs %04x\n",
                                                         row / T, row / T / 8, (row / T) % 8, t, c, j, c % 3, p, p >> 6, p & 63, gd_ref[i], gd_cur[i]); }
                    }
                printf("  rep %d: pose values differing %zu, pose-gradient values differing %zu; by trip %d %d %d; by comp %d %d %d; by lane:", rep, nx, ng,
                       triphist[0], triphist[1], triphist[2], comphist[0], comphist[1], comphist[2]);
                for (int l = 0; l < 64; ++l) printf(" %d", lanehist[l]);
                printf("\n");
            }
#endif
            if (!rep) continue;
            size_t nbad = 0; long first = -1;
            int colhist[16] = {0}, tilehist[5] = {0};
            for (size_t i = 0; i < cur.size(); ++i)
                if (cur[i] != ref[i]) {
                    if (first < 0) first = (long)i;
                    ++nbad; ++colhist[(i % 256) / 16]; ++tilehist[((i / 256) % 80) / 16];
                }
            if (nbad) {
                ++bad_launches;
                printf("  rep %d: %zu differing values, first at row %ld (workgroup %ld, row-in-wg %ld) col %ld; by 16-col group:", rep, nbad, first / 256,
                       first / 256 / 80, (first / 256) % 80, first % 256);
                for (int c = 0; c < 16; ++c) printf(" %d", colhist[c]);
                printf("; by row tile:");
                for (int c = 0; c < 5; ++c) printf(" %d", tilehist[c]);
                printf("\n");
            }
        }
        printf("determinism B=%d: %d of 11 launches differ from the first\n", B, bad_launches);
        if (getenv("TAIL_NRT_B")) {          // the same batch through ANOTHER row-tile instantiation: every window's rows must come out the same
            const int nrt_b = atoi(getenv("TAIL_NRT_B"));
            const size_t lds_b = plan_tail_bf16(net.dec, 1, T, J, &a, nrt_b);
            hipMemsetAsync(a.g_out_b, 0xFF, rows * 256 * 2, s);
            launch_tail_bf16(&h, a, lds_b, s);
            hipStreamSynchronize(s);
            hipMemcpy(cur.data(), a.g_out_b, rows * 256 * 2, hipMemcpyDeviceToHost);
            size_t nbad = 0, nwin = 0; long lastw = -1;
            for (size_t i = 0; i < cur.size(); ++i)
                if (cur[i] != ref[i]) {
                    ++nbad;
                    const long w = (long)(i / 256 / T);
                    if (w != lastw) { ++nwin; lastw = w; if (nwin <= 6) printf("   window %ld row %zu col %zu: %04x vs %04x\n", w, (i / 256) % T, i % 256, ref[i], cur[i]); }
                }
            printf("row tiles %d vs %d: %zu differing values in %zu of %d windows\n", nrt, nrt_b, nbad, nwin, B);
        }
        return 0;
    }
    for (int fo = 1; fo >= 0; --fo) {
        a.forward_only = fo;
        for (int i = 0; i < 3; ++i) launch_tail_bf16(&h, a, lds, s);
        hipEventRecord(e0, s);
        for (int i = 0; i < iters; ++i) launch_tail_bf16(&h, a, lds, s);
        hipEventRecord(e1, s); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("B=%d forward_only=%d: %.1f us\n", B, fo, ms * 1e3 / iters);
    }
    return 0;
}
