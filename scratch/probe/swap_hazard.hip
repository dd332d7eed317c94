// What hipcc's gfx950 hazard recognizer inserts around the BUILTIN permlane swaps: a VALU write directly in front of the swap and a
// VALU / DPP read directly behind it.  scratch/swap_hazard_probe.py compiles this to a listing and prints the swap neighbourhoods:
// the pads there are the hazard table's entries for these instructions, which the inline-asm helpers (csrc/maxsim_device.h:
// swap16 / swap32) have to reproduce by hand.
#include <hip/hip_runtime.h>
__global__ void k(float* o, const float* x) {
    float v = x[threadIdx.x];
    float a = v * 2.f;           // VALU write right before the swap reads it
    float b = v + 1.f;
    auto r = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b), false, false);
    float m = __builtin_bit_cast(float, r[0]) * 3.f + __builtin_bit_cast(float, r[1]);    // VALU read right after
    auto r2 = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, m), __builtin_bit_cast(unsigned, m * 5.f), false, false);
    float m2 = __builtin_bit_cast(float, r2[0]) - __builtin_bit_cast(float, r2[1]);
    float d = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, m2), 0xB1, 0xF, 0xF, false));   // DPP read of a swap-derived value
    o[threadIdx.x] = d + m2;
}
