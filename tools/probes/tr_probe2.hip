// ds_read_b64_tr_b16 with the addressing of a [row][col] tile whose 16-lane group reads a 4 (rows) x 16 (cols) block: lane L
// addresses the 4 contiguous elements (row L >> 2, cols 4 (L & 3) ..).  Prints which (row, col) every (lane, element) received.
//   hipcc --offload-arch=gfx950 tr_probe2.hip -o tr_probe2 && ./tr_probe2
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
__global__ void probe(unsigned short* out) {
    __shared__ unsigned short lds[64 * 64];
    for (int i = threadIdx.x; i < 64 * 64; i += 64) lds[i] = (unsigned short)i;      // element (row, col) of a 64-column tile = row * 64 + col
    __syncthreads();
    const int lane = threadIdx.x, L = lane & 15, g = lane >> 4;
    const int row = g * 8 + (L >> 2), col = 4 * (L & 3);
    const unsigned a = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)(lds + row * 64 + col);
    u32x2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a));
    unsigned short r[4];
    __builtin_memcpy(r, &v, 8);
    for (int e = 0; e < 4; ++e) out[lane * 4 + e] = r[e];
}
int main() {
    unsigned short* d; hipMalloc(&d, 512);
    unsigned short h[256];
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; ++l) {
        printf("lane %2d:", l);
        for (int e = 0; e < 4; ++e) printf("  (r%2d,c%2d)", h[l * 4 + e] / 64, h[l * 4 + e] % 64);
        printf("\n");
    }
    return 0;
}
