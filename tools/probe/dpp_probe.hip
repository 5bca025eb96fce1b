// Probe: direction and boundary behaviour of the DPP wave shifts on gfx950.
#include <hip/hip_runtime.h>
#include <stdio.h>
__device__ inline int shl1(int v, int old) { return __builtin_amdgcn_update_dpp(old, v, 0x130, 0xf, 0xf, false); }
__device__ inline int shr1(int v, int old) { return __builtin_amdgcn_update_dpp(old, v, 0x138, 0xf, 0xf, false); }
__global__ void k(int* out) {
    const int l = threadIdx.x;
    out[l] = shl1(l, -7);
    out[64 + l] = shr1(l, -7);
    int r = -9;
    if (l == 5 || l == 4) r = shl1(l, -7);   // partial exec: lane 4 reads lane 5 (active), lane 5 reads lane 6 (inactive)
    out[128 + l] = r;
}
int main() {
    int* d; int h[192];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("shl1: "); for (int i = 0; i < 64; ++i) printf("%d ", h[i]); printf("\n");
    printf("shr1: "); for (int i = 0; i < 64; ++i) printf("%d ", h[64 + i]); printf("\n");
    printf("part: "); for (int i = 0; i < 8; ++i) printf("%d ", h[128 + i]); printf("\n");
    return 0;
}
