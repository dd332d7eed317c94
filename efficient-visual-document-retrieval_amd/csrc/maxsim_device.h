// Device-side helpers shared by the MaxSim forward kernels (maxsim_fwd.hip, maxsim_fwd16.hip).
#pragma once
#include "evdr_common.h"

namespace evdr {

constexpr int kWaves = 8;                                         // waves per forward workgroup (2 per SIMD)
constexpr int kTileBytes = EVDR_TILE_PATCHES * EVDR_D * 2;        // 8 KiB: one bf16 plane of one 32-patch tile

__device__ __forceinline__ float neg_inf() { return -__builtin_inff(); }

// ---- sentinel build (-DEVDR_SENTINEL, libevdr_sentinel.so: tests/test_gpu_sentinel.py; never the product library) ----
// Every LDS-DMA piece first POISONS the 1 KiB it is about to fill (0x7B7B in every 16-bit element: 1.3e36 as bf16, 61 280
// as fp16) and retires that write before the DMA is issued.  A ds_read that gets at a piece before its data has landed
// (RAW), or a wave still reading a ring slot's previous tenant when the next refill starts (WAR), then sees the poison in
// all 128 dims of the affected rows -- an astronomically wrong score on EVERY affected token -- instead of the old bytes
// of a neighbouring tile, which change a score only when they happen to hold a token's best patch (~0.4 % per piece).
#ifdef EVDR_SENTINEL
__device__ __forceinline__ void sentinel_fill(uint32_t lds_base) {
    // two VGPRs (address, pattern): the kernels that sit at 256 registers have no room for a 128-bit store operand
    const uint32_t a = lds_base + (threadIdx.x & 63u) * 16u;
    const uint32_t v = 0x7B7B7B7Bu;
    asm volatile("ds_write2_b32 %0, %1, %1 offset1:1\n\tds_write2_b32 %0, %1, %1 offset0:2 offset1:3\n\ts_waitcnt lgkmcnt(0)"
                 ::"v"(a), "v"(v) : "memory");
}
#else
__device__ __forceinline__ void sentinel_fill(uint32_t) {}
#endif

// LDS-DMA: 64 lanes x 16 B from per-lane global addresses to LDS [lds_base, lds_base + 1 KiB), lane-linear.
// Issued as inline asm on purpose: hipcc does not see it, so (a) it cannot put a vmcnt(0) in front of the ds_reads of
// the ring (it treats a builtin LDS-DMA as an LDS store that may alias them) and (b) the ring's completion is counted
// by hand (ring_barrier<N>() below).  M0 is written and restored inside the statement (cdna_hip_programming.md §5.7); the
// s_nop between the M0 write and the load is the SALU-writes-M0 -> LDS-DMA wait state.
__device__ __forceinline__ void lds_dma_16B(const void* gsrc, uint32_t lds_base) {
    sentinel_fill(lds_base);
    uint32_t keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, off\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(gsrc), "s"(lds_base)
        : "memory");
}

// The same with a wave-uniform 64-bit base in SGPRs and a 32-bit (zero-extended) per-lane byte offset: one VGPR per
// piece instead of a 64-bit per-lane pointer (the staged kernel issues pieces in the middle of its MFMA block, where
// VGPRs are scarce).  s_nop 4 covers a base that was just produced by a VALU->SGPR move (§5.7 item 2).
__device__ __forceinline__ void lds_dma_16B_sbase(const void* sbase, uint32_t voff, uint32_t lds_base) {
    sentinel_fill(lds_base);
    uint32_t keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %3\n\t"
        "s_nop 4\n\t"
        "global_load_lds_dwordx4 %1, %2\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(voff), "s"(sbase), "s"(lds_base)
        : "memory");
}

// The same with the non-temporal cache policy: for corpus bytes that exactly ONE workgroup reads exactly once (a launch with a
// single query group: 1-32 queries): they need no place in L2 / Infinity Cache, and the stream lands sooner
// (MI355X_MICROARCH.md "nt-weights": issued -> landed -18 %).  Never when several query groups re-read a page chunk from L2.
__device__ __forceinline__ void lds_dma_16B_sbase_nt(const void* sbase, uint32_t voff, uint32_t lds_base) {
    sentinel_fill(lds_base);
    uint32_t keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %3\n\t"
        "s_nop 4\n\t"
        "global_load_lds_dwordx4 %1, %2 nt\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(voff), "s"(sbase), "s"(lds_base)
        : "memory");
}

// counted wait on the vector-memory queue (loads, stores and LDS-DMA count together, in issue order)
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// ---- stage hand-over of an LDS-DMA ring: ONE statement, ordered by construction ------------------------------------
// RAW (a stage's data before any wave reads it): every wave retires its own pieces of the stage (all but its N youngest
// vector-memory operations) and only then arrives at the barrier; a reader has passed the barrier, hence every wave's wait.
// WAR (a slot's last reads before its refill): the refill of a slot is issued behind a LATER ring_barrier than the
// stage that read it; lgkmcnt(0) in front of the barrier retires every ds_read the wave has issued so far -- also those
// whose consumers the scheduler may have placed after the barrier -- so no read of the old tenant is in flight when any wave
// passes.  Waits and s_barrier sit in one volatile asm with a memory clobber: the compiler can neither put an LDS access
// between them nor move one across them (the bare __builtin_amdgcn_s_barrier() is IntrNoMem to LLVM: no fence at all), and no
// schedule change or recompile can take the order apart (cdna_hip_programming.md, "Read a staged buffer one phase AFTER the
// wait that retires it"; scratch/audit_ring_isa.py checks every instance's listing for exactly this form).
template <int N>
__device__ __forceinline__ void ring_barrier() {
#if defined(EVDR_RING_FAULT) && EVDR_RING_FAULT == 1
    // positive control of the sentinel instrument (scratch/sentinel_control.py; never a shipped build, and
    // scratch/audit_ring_isa.py rejects it): the hand-over WITHOUT its vmcnt wait, i.e. a deliberate read-before-landed race
    // (EVDR_RING_FAULT == 2 is the WAR control: the flat kernel issues its refill IN FRONT of the hand-over, maxsim_fwd16.hip)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#elif defined(EVDR_RING_NOLGKM)
    // A/B only (scratch/r04_ab_hardening.sh): what the lgkmcnt(0) of the hand-over costs; rejected by scratch/audit_ring_isa.py
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(N) : "memory");
#else
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(N) : "memory");
#endif
}

// ---- cross-lane reductions for the page epilogue, at VALU speed (no LDS round trips) ------------------------------
// gfx950's v_permlane16_swap / v_permlane32_swap exchange 16-lane rows / 32-lane halves between TWO registers
// (measured lane maps, a = vdst, b = src:  16: a' = [a.row0, b.row0, a.row2, b.row2], b' = [a.row1, b.row1, a.row3, b.row3];
//  32: a' = [a.lo, b.lo], b' = [a.hi, b.hi]).  Fed two copies of one value, a' and b' together give every lane its own and
// its partner row's / half's value.  Inline asm with two "+v" operands: the copies are then necessarily in different
// registers (a register swapped with itself is a no-op) and both results are unambiguous -- the builtin form returned
// the first result twice when both inputs carried the same value (hipcc 7.2).  The two v_nop are the VALU-write ->
// permlane-read wait states, which hipcc does not insert inside an asm string (cdna_hip_programming.md T21).
__device__ __forceinline__ void swap16(float& a, float& b) {
    asm volatile("v_nop\n\tv_nop\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ void swap32(float& a, float& b) {
    asm volatile("v_nop\n\tv_nop\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ float xhalf_max(float v) {           // max over {l, l^32}, in every lane
    float a = v, b = v;
    swap32(a, b);
    return __builtin_fmaxf(a, b);
}
__device__ __forceinline__ float xgroup_max(float v) {          // max over the four lane groups {l, l^16, l^32, l^48}
    float a = v, b = v;
    swap16(a, b);
    return xhalf_max(__builtin_fmaxf(a, b));
}
__device__ __forceinline__ float row16_sum(float v) {           // sum over the 16 lanes of a DPP row, in every lane
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, false));   // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, false));   // quad_perm [2,3,0,1]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, false));  // row_half_mirror
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, false));  // row_mirror
    return v;
}
__device__ __forceinline__ float row32_sum(float v) {           // sum over the 32 lanes of a half, in every lane
    float a = row16_sum(v), b = a;
    swap16(a, b);
    return a + b;
}

// Block -> (query group, page chunk).  Blocks b and b+8 share an XCD under round-robin dispatch, so consecutive b>>3
// walk the query groups of ONE page chunk: its bytes are served from that XCD's L2 (speed only, never correctness).
struct BlockWork {
    int qg, pg0, npages;
    bool valid;
};
__device__ __forceinline__ BlockWork block_work(const EvdrFwdParams& p) {
    const int b = blockIdx.x;
    const int xi = b >> 3;
    BlockWork w;
    w.qg = xi % p.n_qgroups;
    const int chunk = (xi / p.n_qgroups) * 8 + (b & 7);
    w.valid = chunk < p.n_chunks;
    w.pg0 = chunk * p.pages_per_block;
    w.npages = min(p.pages_per_block, p.np - w.pg0);
    return w;
}

}  // namespace evdr

// Host side: queries per workgroup -> launch geometry shared by every forward kernel.
static inline int64_t evdr_set_geometry(EvdrFwdParams& p, int queries_per_wg, int wgs_per_cu = 1) {
    p.ntiles = (p.lp + EVDR_TILE_PATCHES - 1) / EVDR_TILE_PATCHES;
    p.n_qgroups = (p.nq + queries_per_wg - 1) / queries_per_wg;
    p.pages_per_block = evdr_pages_per_block(p.np, p.n_qgroups, p.ntiles, 256 * wgs_per_cu);
    if (const int v = evdr_pages_per_block_override(); v > 0)       // evdr_debug_set_pages_per_block (A/B experiments)
        p.pages_per_block = v < p.np ? v : p.np;
    p.n_chunks = (p.np + p.pages_per_block - 1) / p.pages_per_block;
    return (int64_t)((p.n_chunks + 7) / 8) * 8 * p.n_qgroups;     // grid size (whole groups of 8 for the XCD map)
}
