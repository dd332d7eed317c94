#include <hip/hip_runtime.h>
#include <stdio.h>
#include "../../efficient-visual-document-retrieval_amd/csrc/maxsim_device.h"
__global__ void k(float* o) {
    const float v = (float)(threadIdx.x + 1);
    o[threadIdx.x] = evdr::row16_sum(v);
    o[64 + threadIdx.x] = evdr::row32_sum(v);
    o[128 + threadIdx.x] = evdr::xgroup_max(v);
    o[192 + threadIdx.x] = evdr::xhalf_max(v);
}
int main() {
    float* d; hipMalloc(&d, 256 * 4); k<<<1, 64>>>(d); float h[256]; hipMemcpy(h, d, 1024, hipMemcpyDeviceToHost);
    const char* names[4] = {"row16_sum", "row32_sum", "xgroup_max", "xhalf_max"};
    for (int a = 0; a < 4; ++a) { printf("%s:", names[a]); for (int i = 0; i < 64; i += 7) printf(" [%d]=%g", i, h[a * 64 + i]); printf("\n"); }
    printf("expected row16 sums: 136 392 648 904; row32 sums: 528 1552; xgroup_max(l)=l%%16+49; xhalf_max(l)=l%%32+33\n");
    return 0;
}
