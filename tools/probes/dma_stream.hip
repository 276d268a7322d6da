// Probe: what HBM -> LDS rate do persistent workgroups reach with global_load_lds (16 B / lane) double buffering, as a function of
// the tile shape (rows x contiguous bytes), the number of co-resident workgroups per CU and a per-tile compute delay?
// Models the halo conv / halo wgrad tile loop (wait vmcnt(0) -> barrier -> issue next tile -> work on this one).
//   hipcc --offload-arch=gfx950 -O3 -o dma_stream dma_stream.hip && ./dma_stream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

struct Args {
    const char* base;
    long long stride;        // bytes between tile rows (= image row pitch)
    int rows, seg_kb;        // tile = rows x seg_kb KiB
    int tiles_x, ntiles, tiles_per_wg;
    int sleep;               // s_sleep units per tile (64 clocks each)
    int slots;
};

__global__ __launch_bounds__(256) void dma_stream_kernel(const Args a, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ninst = a.rows * a.seg_kb, tile_bytes = ninst * 1024;
    const int t0 = blockIdx.x * a.tiles_per_wg, t1 = min(a.ntiles, t0 + a.tiles_per_wg);
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
    auto issue = [&](int tile, int slot) {
        const int tx = tile % a.tiles_x, ty = tile / a.tiles_x;
        const char* org = a.base + (long long)ty * a.rows * a.stride + (long long)tx * a.seg_kb * 1024;
        for (int ii = wave; ii < ninst; ii += 4) {
            const int r = ii / a.seg_kb, s = ii - r * a.seg_kb;
            __builtin_amdgcn_global_load_lds((gptr_t)(org + r * a.stride + s * 1024 + lane * 16), (lptr_t)(lds + slot * tile_bytes + ii * 1024), 16, 0, 0);
        }
    };
    float accv = 0.f;
    if (t0 < t1) issue(t0, 0);
    if (a.slots == 3 && t0 + 1 < t1) issue(t0 + 1, 1);
    int slot = 0;
    for (int t = t0; t < t1; ++t) {
        if (a.slots == 3 && t + 1 < t1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");   // 3 slots only with 16-instruction tiles: 4 per wave
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const int ahead = a.slots - 1;
        if (t + ahead < t1) issue(t + ahead, (slot + ahead) % a.slots);
        accv += *(const float*)(lds + slot * tile_bytes + threadIdx.x * 4);
        for (int i = 0; i < a.sleep; ++i) __builtin_amdgcn_s_sleep(1);
        slot = (slot + 1) % a.slots;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (accv == 123.456f) sink[0] = accv;
}

int main() {
    const long long pitch = 1920ll * 64, H = 1088 * 3;            // three 1080p frames of 32 bf16 channels
    const long long bytes = pitch * H;
    char* buf; float* sink;
    hipMalloc(&buf, bytes + (1 << 20)); hipMemset(buf, 1, bytes); hipMalloc(&sink, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    struct Cfg { int rows, seg_kb, wg_per_cu, sleep, slots; };
    std::vector<Cfg> cfgs;
    for (int sl : {0, 8, 16})
        for (int wpc : {1, 2, 4}) {
            cfgs.push_back({8, 2, wpc, sl, 2});      // the halo tile: 8 rows x 2 KiB
            cfgs.push_back({16, 2, wpc, sl, 2});
            cfgs.push_back({4, 4, wpc, sl, 2});
            cfgs.push_back({2, 8, wpc, sl, 2});
            cfgs.push_back({1, 16, wpc, sl, 2});     // fully contiguous 16 KiB
            cfgs.push_back({1, 30, wpc, sl, 2});     // a quarter image row
            cfgs.push_back({8, 2, wpc, sl, 3});
        }
    printf("%5s %6s %6s %5s %5s %8s %8s\n", "rows", "segKB", "wg/cu", "sleep", "slots", "us", "TB/s");
    for (const Cfg& c : cfgs) {
        Args a;
        a.base = buf; a.stride = pitch; a.rows = c.rows; a.seg_kb = c.seg_kb; a.sleep = c.sleep; a.slots = c.slots;
        a.tiles_x = (int)(pitch / (c.seg_kb * 1024));
        a.ntiles = a.tiles_x * (int)(H / c.rows);
        const int lds_bytes = c.rows * c.seg_kb * 1024 * c.slots;
        int wgs = 256 * c.wg_per_cu;
        // keep the requested residency: pad the dynamic LDS so that exactly wg_per_cu fit (160 KiB per CU)
        int pad = 160 * 1024 / c.wg_per_cu - 1024;
        if (pad < lds_bytes) { printf("%5d %6d %6d %5d %5d   (does not fit)\n", c.rows, c.seg_kb, c.wg_per_cu, c.sleep, c.slots); continue; }
        a.tiles_per_wg = (a.ntiles + wgs - 1) / wgs;
        wgs = (a.ntiles + a.tiles_per_wg - 1) / a.tiles_per_wg;
        hipFuncSetAttribute((const void*)dma_stream_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, pad);
        for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(dma_stream_kernel, dim3(wgs), dim3(256), pad, 0, a, sink);
        hipEventRecord(e0);
        const int it = 10;
        for (int i = 0; i < it; ++i) hipLaunchKernelGGL(dma_stream_kernel, dim3(wgs), dim3(256), pad, 0, a, sink);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (hipGetLastError() != hipSuccess) { printf("launch error\n"); return 1; }
        const double us = ms * 1000 / it;
        printf("%5d %6d %6d %5d %5d %8.1f %8.2f\n", c.rows, c.seg_kb, c.wg_per_cu, c.sleep, c.slots, us,
               (double)a.ntiles * c.rows * c.seg_kb * 1024 / us * 1e-6);
    }
    return 0;
}
