// Probe: issue rate of v_mfma_f32_32x32x16_bf16 with independent accumulators, in s_memtime ticks and in wall-clock time,
// with 1 or 2 waves per SIMD (512-thread blocks, one block per CU) and the whole chip busy.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_rate.hip -o /tmp/mfma_rate && /tmp/mfma_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

template <int NACC>
__global__ __launch_bounds__(512) void mfma_loop(float* out, unsigned long long* ticks, int iters) {
    f32x16_t acc[NACC];
    for (int a = 0; a < NACC; ++a)
        for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    bf16x8_t x, y;
    for (int k = 0; k < 8; ++k) { x[k] = (__bf16)(threadIdx.x * 0.001f + k); y[k] = (__bf16)(k * 0.5f); }
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int a = 0; a < NACC; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, acc[a], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int a = 0; a < NACC; ++a) s += acc[a][0];
    if (threadIdx.x == 0) { out[blockIdx.x] = s; ticks[blockIdx.x] = t1 - t0; }
}

template <int NACC>
static void run(int threads, int blocks) {
    float* out; unsigned long long* ticks;
    hipMalloc(&out, blocks * 4); hipMalloc(&ticks, blocks * 8);
    const int iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(mfma_loop<NACC>, dim3(blocks), dim3(threads), 0, 0, out, ticks, 100);
    hipEventRecord(e0);
    hipLaunchKernelGGL(mfma_loop<NACC>, dim3(blocks), dim3(threads), 0, 0, out, ticks, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[4]; hipMemcpy(h, ticks, 32, hipMemcpyDeviceToHost);
    const double n = (double)iters * NACC;                       // MFMAs per wave
    const double waves_per_simd = threads / 256.0;
    const double tflops = n * 32768.0 * (threads / 64.0) * blocks / (ms * 1e-3) / 1e12;
    printf("acc=%d threads=%d blocks=%d: %.1f ticks/MFMA per wave, %.2f ns/MFMA per SIMD, %.0f TFLOP/s, tick rate %.2f GHz\n", NACC, threads, blocks,
           h[0] / n, ms * 1e6 / (n * waves_per_simd), tflops, h[0] / (ms * 1e6));
    hipFree(out); hipFree(ticks);
}
int main() {
    run<8>(256, 256);
    run<8>(512, 256);
    run<4>(256, 256);
    run<2>(256, 256);
    run<8>(256, 32);      // a lightly loaded chip: 1/8 of the CUs
    return 0;
}
