// Probe: where does a k-step of the halo-B conv study kernel (hbconv_study.hip) spend its time?  Builds the kernel with cycle stamps
// (HB_TRACE), runs the 256-channel os16 layer of a 1080p window (68 x 120, 3 frames) on synthetic data and prints the stamps of
// workgroup 8 / wave 0: per k-step the cycles in  wait(vmcnt) | barrier | DMA issue | fragment reads + MFMAs.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DTCVOM_F16 -DHB_TRACE -munsafe-fp-atomics -I../../tcvom_amd/csrc -o hb_probe hb_probe.hip && ./hb_probe
#include <cstdarg>
#include <cstdio>
#include <vector>
#include "hbconv_study.hip"

thread_local char g_tcvom_err[512] = "";
int tcvom_fail(int code, const char* fmt, ...) {
    va_list ap; va_start(ap, fmt); vsnprintf(g_tcvom_err, sizeof(g_tcvom_err), fmt, ap); va_end(ap);
    fprintf(stderr, "error: %s\n", g_tcvom_err);
    return code;
}

int main() {
    const int H = 68, W = 120, C = 256, K = 256, F = 3;
    tcvom_conv_desc d = {};
    d.N = 1; d.H = H; d.W = W; d.C = C; d.OH = H; d.OW = W; d.K = K; d.PH = H; d.PW = W;
    d.in_step = d.out_step = 1; d.ntaps = 9; d.wt = 9; d.ldo = K; d.act = 1; d.batch = F;
    for (int t = 0; t < 9; ++t) { d.tap_dh[t] = t / 3 - 1; d.tap_dw[t] = t % 3 - 1; d.tap_w[t] = t; }
    d.in_bstride = (long long)H * W * C; d.w_bstride = (long long)K * 9 * C; d.out_bstride = (long long)H * W * K;
    const int groups = hbconv_stats_groups(&d, 1);
    d.stats_bstride = groups;
    printf("stats groups per frame %d\n", groups);
    h16raw *in, *w, *out, *zp; float* stats;
    hipMalloc(&in, sizeof(h16raw) * F * H * W * C); hipMalloc(&w, sizeof(h16raw) * F * K * 9 * C); hipMalloc(&out, sizeof(h16raw) * F * H * W * K);
    hipMalloc(&zp, 256); hipMemset(zp, 0, 256);
    hipMalloc(&stats, sizeof(float) * F * groups * 2 * K);
    hipMemset(in, 0x11, sizeof(h16raw) * F * H * W * C); hipMemset(w, 0x12, sizeof(h16raw) * F * K * 9 * C);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0, 0);
        for (int i = 0; i < 20; ++i)
            if (hbconv_try_launch(in, w, out, nullptr, nullptr, nullptr, stats, &d, 1, zp, nullptr) != 1) { printf("not launched\n"); return 1; }
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("launch: %.1f us\n", ms * 1000.f / 20);
    }
    std::vector<unsigned long long> tr(4096);
    hipMemcpyFromSymbol(tr.data(), HIP_SYMBOL(hb_trace_buf), sizeof(unsigned long long) * 4096);
    const int nstep = 36;
    double tot[4] = {0, 0, 0, 0};
    for (int s = 0; s < nstep; ++s) {
        const unsigned long long* p = tr.data() + s * 5;
        printf("step %2d: wait %5llu  barrier %5llu  issue %5llu  mfma %5llu   (step %llu)\n", s, p[1] - p[0], p[2] - p[1], p[3] - p[2], p[4] - p[3],
               s + 1 < nstep ? tr[(s + 1) * 5] - p[0] : p[4] - p[0]);
        for (int i = 0; i < 4; ++i) tot[i] += (double)(p[i + 1] - p[i]);
    }
    printf("mean per step: wait %.0f  barrier %.0f  issue %.0f  mfma %.0f cycles; first to last stamp %llu cycles\n", tot[0] / nstep, tot[1] / nstep,
           tot[2] / nstep, tot[3] / nstep, tr[(nstep - 1) * 5 + 4] - tr[0]);
    return 0;
}
