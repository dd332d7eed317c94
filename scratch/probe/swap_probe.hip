#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(unsigned* o) {
    unsigned u = threadIdx.x;
    auto r = __builtin_amdgcn_permlane16_swap(u, u + 100, false, false);
    auto s = __builtin_amdgcn_permlane32_swap(u, u + 100, false, false);
    o[threadIdx.x] = r[0]; o[64 + threadIdx.x] = r[1]; o[128 + threadIdx.x] = s[0]; o[192 + threadIdx.x] = s[1];
}
int main() {
    unsigned* d; hipMalloc(&d, 256 * 4); k<<<1, 64>>>(d); unsigned h[256]; hipMemcpy(h, d, 1024, hipMemcpyDeviceToHost);
    const char* names[4] = {"p16 r0", "p16 r1", "p32 r0", "p32 r1"};
    for (int a = 0; a < 4; ++a) { printf("%s:", names[a]); for (int i = 0; i < 64; i += 8) printf(" [%d]=%u", i, h[a * 64 + i]); printf("\n"); }
    return 0;
}
