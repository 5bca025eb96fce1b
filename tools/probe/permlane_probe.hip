#include <hip/hip_runtime.h>
__global__ void k(unsigned* o) {
    unsigned v = threadIdx.x;
    auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);
    auto q = __builtin_amdgcn_permlane16_swap(v, v, false, false);
    o[threadIdx.x] = r[0]; o[64 + threadIdx.x] = r[1]; o[128 + threadIdx.x] = q[0]; o[192 + threadIdx.x] = q[1];
}
int main() {
    unsigned* d; hipMalloc(&d, 256 * 4); k<<<1, 64>>>(d); unsigned h[256]; hipMemcpy(h, d, 1024, hipMemcpyDeviceToHost);
    for (int j = 0; j < 4; ++j) { for (int i = 0; i < 64; ++i) printf("%u ", h[64 * j + i]); printf("\n"); }
    return 0;
}
