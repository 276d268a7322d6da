// Probe of ds_read_b64_tr_b16 semantics on gfx950: prints, for two addressing patterns, which LDS element index
// every (lane, element) of the result came from.   hipcc --offload-arch=gfx950 tr_probe.hip -o tr_probe && ./tr_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_t;
__global__ void probe(unsigned short* out, int mode) {
    __shared__ unsigned short lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) {
        // value i encoded exactly in bf16 is impossible for i > 256: store raw bit patterns instead
        lds[i] = (unsigned short)i;
    }
    __syncthreads();
    const int lane = threadIdx.x;
    int off;                                    // element offset of this lane's 4-element (8-byte) source
    if (mode == 0) off = lane * 4;              // lane-linear
    else if (mode == 1) off = (lane & 15) * 64 + (lane >> 4) * 4;     // 16 rows of 64 elements, 4 column groups
    else off = (lane & 15) * 16 + (lane >> 4) * 256;                  // 16 rows of 16 elements per 16-lane group
    bf16x4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4_t*)(lds + off));
    unsigned short r[4];
    __builtin_memcpy(r, &v, 8);
    for (int e = 0; e < 4; ++e) out[lane * 4 + e] = r[e];
}
int main() {
    unsigned short* d; hipMalloc(&d, 512);
    unsigned short h[256];
    for (int mode = 0; mode < 3; ++mode) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, mode);
        hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
        printf("mode %d\n", mode);
        for (int l = 0; l < 64; ++l) printf("lane %2d: %4d %4d %4d %4d\n", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3]);
    }
    return 0;
}
