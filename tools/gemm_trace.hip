// Developer tool: lifetime of every workgroup of the decoder_input / conv GEMM launches at 240 windows.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DGEM_TRACE -o tools/gemm_trace tools/gemm_trace.hip && ./tools/gemm_trace
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../globalegomocap_amd/csrc/gemm_f32.hip"
namespace gem {
void set_error(const std::string& m) { fprintf(stderr, "error: %s\n", m.c_str()); }
bool hip_ok(hipError_t e, const char* what) { if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", what, hipGetErrorString(e)); return false; } return true; }
int launch_gemm_bf16(gem_handle*, const Layer&, int, int, const float*, int, const float*, float*, int, int, int, hipStream_t, const int*) { return 1; }
}
using namespace gem;
static float* dev_rand(size_t n, unsigned seed, float scale) {
    std::vector<float> h(n); srand(seed);
    for (size_t i = 0; i < n; ++i) h[i] = scale * ((rand() / (float)RAND_MAX) * 2.f - 1.f);
    float* d; hipMalloc(&d, n * 4); hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice); return d;
}
int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 240, T = 10;
    struct Case { const char* name; int M, N, K, taps, epi, family; };
    Case cases[] = {{"dec_in fwd", B, 5120, 2048, 1, EPI_BIAS, 0}, {"dec_in bwd", B, 2048, 5120, 1, EPI_BIAS, 0},
                    {"conv1 fwd", B * T, 256, 512, 3, EPI_BIAS_LRELU, -1}, {"conv1 bwd", B * T, 512, 256, 3, EPI_NONE, -1}};
    gem_handle h; h.prof.on = false; h.cfg.device = 0;
    h.ws.splitk_elems = (size_t)8 << 20; hipMalloc(&h.ws.splitk, h.ws.splitk_elems * 4);
    hipStream_t s; hipStreamCreate(&s);
    const int MAXWG = 4096;
    long long* d_tr; hipMalloc(&d_tr, (size_t)MAXWG * 4 * 8);
    hipMemcpyToSymbol(HIP_SYMBOL(g_gemm_trace), &d_tr, sizeof(d_tr));
    for (auto& c : cases) {
        Layer L; L.taps = c.taps; L.K = c.K; L.N = c.N;
        L.w = dev_rand((size_t)c.taps * c.N * c.K, 1, 0.05f); L.bias = dev_rand(c.N, 2, 0.1f);
        float* A = dev_rand((size_t)c.M * c.K, 3, 1.f);
        float* C; hipMalloc(&C, (size_t)c.M * c.N * 4);
        for (int i = 0; i < 300; ++i) launch_gemm(&h, L, c.epi, A, c.K, nullptr, C, c.N, c.M, T, s, c.family);     // warm clocks
        hipStreamSynchronize(s);
        hipMemset(d_tr, 0, (size_t)MAXWG * 4 * 8);
        launch_gemm(&h, L, c.epi, A, c.K, nullptr, C, c.N, c.M, T, s, c.family);
        hipStreamSynchronize(s);
        std::vector<long long> tr((size_t)MAXWG * 4);
        hipMemcpy(tr.data(), d_tr, tr.size() * 8, hipMemcpyDeviceToHost);
        long long t0 = -1, t1 = 0; int n = 0;
        std::vector<double> pro, loop, epi, life, start;
        for (int w = 0; w < MAXWG; ++w) {
            const long long* q = &tr[4 * w];
            if (!q[0] || !q[3]) continue;
            if (t0 < 0 || q[0] < t0) t0 = q[0];
            t1 = std::max(t1, q[3]); ++n;
        }
        for (int w = 0; w < MAXWG; ++w) {
            const long long* q = &tr[4 * w];
            if (!q[0] || !q[3]) continue;
            pro.push_back((q[1] - q[0]) * 0.01); loop.push_back((q[2] - q[1]) * 0.01); epi.push_back((q[3] - q[2]) * 0.01);
            life.push_back((q[3] - q[0]) * 0.01); start.push_back((q[0] - t0) * 0.01);
        }
        auto med = [](std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
        auto mx = [](const std::vector<double>& v) { return *std::max_element(v.begin(), v.end()); };
        printf("%-11s %4d workgroups, span %.1f us | start: median %.1f max %.1f | fill %.1f (max %.1f) | k-loop %.1f (max %.1f) | store %.1f (max %.1f) | lifetime %.1f (max %.1f)\n",
               c.name, n, (t1 - t0) * 0.01, med(start), mx(start), med(pro), mx(pro), med(loop), mx(loop), med(epi), mx(epi), med(life), mx(life));
    }
    return 0;
}
