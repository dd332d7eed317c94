// Score-row cache of a frozen page tensor (include/evdr.h, "score-row cache"): device-side lookup / plan / exchange.
//
// Reference behaviour being served: mainv2_iter_distill_infonce.py:282-283 scores every batch against the frozen teacher pages in
// every epoch; a query's teacher scores are constant.  The cache is keyed on the query row's BITS (+ mask row +
// the plane shift of the batch it arrived in), verified by a full-row comparison, and lives entirely on the device: which queries
// of a batch are missing is decided, compacted and handed to the forward kernel without the host ever looking.
// HBM-bound byte work: per query one read of its row (16 KiB at 32 x 128 fp32), on a hash match one read of the stored row.
#include "evdr_common.h"

namespace {

constexpr int QC_THREADS = 256;
constexpr int QC_MAXPROBE = 32;

// splitmix64 finaliser: the per-word mixer of the row hash (position enters through the upper half of the input)
__device__ __forceinline__ uint64_t mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__device__ __forceinline__ uint32_t first_slot(uint64_t h, int64_t n_slots) {
    return (uint32_t)((h ^ (h >> 32)) & (uint64_t)(n_slots - 1));
}

__device__ __forceinline__ int shift_of(const uint32_t* q_amax) { return q_amax ? evdr_h2_shift(*q_amax) : 0; }

// block-wide sum of one 64-bit value (wrap-around add: order-free), result in every thread
__device__ __forceinline__ uint64_t block_sum64(uint64_t v, uint64_t* sh) {
    for (int o = 32; o > 0; o >>= 1) {
        const uint32_t lo = __shfl_xor((uint32_t)v, o), hi = __shfl_xor((uint32_t)(v >> 32), o);
        v += ((uint64_t)hi << 32) | lo;
    }
    __syncthreads();                                     // sh may still be read from an earlier use
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    uint64_t t = 0;
    for (int w = 0; w < QC_THREADS / 64; ++w) t += sh[w];
    return t;
}

// Misses in ascending query order -> qsel / *qsel_count; while entries are left every miss gets one (hit[q] = -2 - entry) and is
// entered into the table (atomicCAS on empty slots: the inserts of one launch race only with each other).  Run by ONE workgroup.
__device__ void qcache_plan(const EvdrQCache& c, const uint64_t* __restrict__ hashes, int k, int nq, int32_t* __restrict__ hit,
                            int32_t* __restrict__ qsel, int32_t* __restrict__ qsel_count, int* sh_wave, int* sh_base) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int NW = QC_THREADS / 64;
    const int first_entry = *c.n_entries;
    if (tid == 0) *sh_base = 0;
    __syncthreads();
    for (int q0 = 0; q0 < nq; q0 += QC_THREADS) {
        const int q = q0 + tid;
        const bool miss = q < nq && __atomic_load_n(&hit[q], __ATOMIC_RELAXED) < 0;
        const unsigned long long b = __ballot(miss);
        const int rank_in_wave = __popcll(b & ((1ull << lane) - 1ull));
        if (lane == 0) sh_wave[wave] = __popcll(b);
        __syncthreads();
        int before = *sh_base, total = 0;
        for (int w = 0; w < NW; ++w) {
            if (w < wave) before += sh_wave[w];
            total += sh_wave[w];
        }
        if (miss) {
            const int rank = before + rank_in_wave;       // ascending in q: chunks, waves and lanes are walked in order
            qsel[rank] = q;
            int code = -1;
            const int64_t entry = (int64_t)first_entry + rank;
            if (entry < c.capacity) {
                const uint64_t h = __atomic_load_n(&hashes[q], __ATOMIC_RELAXED);
                uint32_t s = first_slot(h, c.n_slots);
                for (int probe = 0; probe < QC_MAXPROBE; ++probe, s = (s + 1) & (uint32_t)(c.n_slots - 1)) {
                    if (atomicCAS(&c.slots[s], 0, (int)entry + 1) == 0) {
                        c.ent_hash[entry] = h;
                        c.ent_k[entry] = k;
                        code = -2 - (int)entry;
                        break;
                    }
                }
                // no free slot within the probe window (a crowded neighbourhood): the entry index stays unused and the row is not stored
            }
            hit[q] = code;
        }
        __syncthreads();
        if (tid == 0) *sh_base += total;
        __syncthreads();
    }
    if (tid == 0) {
        *qsel_count = *sh_base;
        const int64_t n = (int64_t)first_entry + *sh_base;
        *c.n_entries = (int)(n < c.capacity ? n : c.capacity);
    }
}

// One workgroup per query: hashes[q] = hash of (row words, mask bytes, k, lq); hit[q] = the entry whose stored row, mask and k are
// IDENTICAL, else -1 (reads the table only).  The LAST workgroup to finish (ticket word, self-resetting) then plans the launch:
// qcache_plan above -- one kernel, no second launch between "which rows are known" and "which rows must be scored".
__global__ void __launch_bounds__(QC_THREADS) qcache_lookup_plan_kernel(EvdrQCache c, const uint8_t* __restrict__ Q,
                                                                       const uint8_t* __restrict__ qmask,
                                                                       const uint32_t* __restrict__ q_amax, int nq,
                                                                       uint64_t* __restrict__ hashes, int32_t* __restrict__ hit,
                                                                       int32_t* __restrict__ qsel, int32_t* __restrict__ qsel_count,
                                                                       uint32_t* __restrict__ ticket) {
    __shared__ uint64_t sh[QC_THREADS / 64];
    __shared__ int sh_diff, sh_last, sh_wave[QC_THREADS / 64], sh_base;
    constexpr int KEEP = 16;                             // row words a thread keeps in registers (16 KiB rows: all of them)
    const int q = blockIdx.x;
    const int tid = threadIdx.x;
    const int64_t words = c.row_bytes / 4;
    const uint32_t* row = reinterpret_cast<const uint32_t*>(Q + (int64_t)q * c.row_bytes);
    const int k = shift_of(q_amax);
    uint32_t mine[KEEP];
    uint64_t acc = 0;
#pragma unroll
    for (int j = 0; j < KEEP; ++j) {
        const int64_t i = tid + (int64_t)j * QC_THREADS;
        mine[j] = i < words ? row[i] : 0u;
    }
#pragma unroll
    for (int j = 0; j < KEEP; ++j) {
        const int64_t i = tid + (int64_t)j * QC_THREADS;
        if (i < words) acc += mix64(((uint64_t)(uint32_t)i << 32) | mine[j]);
    }
    for (int64_t i = tid + (int64_t)KEEP * QC_THREADS; i < words; i += QC_THREADS) acc += mix64(((uint64_t)(uint32_t)i << 32) | row[i]);
    for (int64_t i = tid; i < c.lq; i += QC_THREADS) {
        const uint32_t m = (qmask == nullptr || qmask[(int64_t)q * c.lq + i] != 0) ? 1u : 0u;
        acc += mix64(((uint64_t)(0x80000000u | (uint32_t)i) << 32) | m);
    }
    uint64_t h = block_sum64(acc, sh);
    h = mix64(h ^ ((uint64_t)(uint32_t)k << 40) ^ (uint64_t)c.lq) & c.hash_mask;
    int found = -1;
    uint32_t s = first_slot(h, c.n_slots);
    for (int probe = 0; probe < QC_MAXPROBE; ++probe, s = (s + 1) & (uint32_t)(c.n_slots - 1)) {
        const int e = c.slots[s];                        // uniform: every thread reads the same word
        if (e == 0) break;
        if (c.ent_hash[e - 1] != h || c.ent_k[e - 1] != k) continue;
        // candidate: the whole row and the whole mask row must be the same bits
        if (tid == 0) sh_diff = 0;
        __syncthreads();
        const uint32_t* srow = reinterpret_cast<const uint32_t*>(c.ent_q + (int64_t)(e - 1) * c.row_bytes);
        int diff = 0;
#pragma unroll
        for (int j = 0; j < KEEP; ++j) {
            const int64_t i = tid + (int64_t)j * QC_THREADS;
            if (i < words) diff |= (srow[i] != mine[j]);
        }
        for (int64_t i = tid + (int64_t)KEEP * QC_THREADS; i < words; i += QC_THREADS) diff |= (srow[i] != row[i]);
        for (int64_t i = tid; i < c.lq; i += QC_THREADS) {
            const uint8_t m = (qmask == nullptr || qmask[(int64_t)q * c.lq + i] != 0) ? 1 : 0;
            diff |= (c.ent_mask[(int64_t)(e - 1) * c.lq + i] != m);
        }
        if (diff) sh_diff = 1;                           // benign race: every writer stores 1
        __syncthreads();
        const int d = sh_diff;
        __syncthreads();
        if (!d) { found = e - 1; break; }
    }
    if (tid == 0) {
        __atomic_store_n(&hashes[q], h, __ATOMIC_RELAXED);
        __atomic_store_n(&hit[q], found, __ATOMIC_RELAXED);
        __threadfence();                                 // this workgroup's results before its ticket
        const uint32_t t = atomicAdd(ticket, 1u);
        sh_last = (t == (uint32_t)nq - 1u);
    }
    __syncthreads();
    if (!sh_last) return;
    __threadfence();                                     // the other workgroups' results after the last ticket
    if (tid == 0) *ticket = 0u;                          // ready for the next call on this stream
    qcache_plan(c, hashes, k, nq, hit, qsel, qsel_count, sh_wave, &sh_base);
}

// One workgroup per query, after the forward: hit -> out row from the cache; stored miss -> score row, query row and mask row into it.
__global__ void __launch_bounds__(QC_THREADS) qcache_exchange_kernel(EvdrQCache c, const uint8_t* __restrict__ Q,
                                                                    const uint8_t* __restrict__ qmask, const int32_t* __restrict__ hit,
                                                                    float* __restrict__ out, int64_t out_stride) {
    const int q = blockIdx.x, tid = threadIdx.x;
    const int code = hit[q];
    if (code >= 0) {
        const float* src = c.ent_scores + (int64_t)code * c.np;
        for (int64_t i = tid; i < c.np; i += QC_THREADS) out[(int64_t)q * out_stride + i] = src[i];
    } else if (code <= -2) {
        const int64_t e = -2 - code;
        float* dst = c.ent_scores + e * c.np;
        for (int64_t i = tid; i < c.np; i += QC_THREADS) dst[i] = out[(int64_t)q * out_stride + i];
        const uint32_t* row = reinterpret_cast<const uint32_t*>(Q + (int64_t)q * c.row_bytes);
        uint32_t* srow = reinterpret_cast<uint32_t*>(c.ent_q + e * c.row_bytes);
        for (int64_t i = tid; i < c.row_bytes / 4; i += QC_THREADS) srow[i] = row[i];
        for (int64_t i = tid; i < c.lq; i += QC_THREADS)
            c.ent_mask[e * c.lq + i] = (qmask == nullptr || qmask[(int64_t)q * c.lq + i] != 0) ? 1 : 0;
    }
}

}  // namespace

hipError_t evdr_launch_qcache_lookup_plan(const EvdrQCache& c, const void* Q, const uint8_t* qmask, const uint32_t* q_amax, int64_t nq,
                                          uint64_t* hashes, int32_t* hit, int32_t* qsel, int32_t* qsel_count, uint32_t* ticket,
                                          hipStream_t stream) {
    hipLaunchKernelGGL(qcache_lookup_plan_kernel, dim3((unsigned)nq), dim3(QC_THREADS), 0, stream, c, (const uint8_t*)Q, qmask, q_amax,
                       (int)nq, hashes, hit, qsel, qsel_count, ticket);
    return hipGetLastError();
}

hipError_t evdr_launch_qcache_exchange(const EvdrQCache& c, const void* Q, const uint8_t* qmask, int64_t nq, const int32_t* hit,
                                       float* out, int64_t out_stride, hipStream_t stream) {
    hipLaunchKernelGGL(qcache_exchange_kernel, dim3((unsigned)nq), dim3(QC_THREADS), 0, stream, c, (const uint8_t*)Q, qmask, hit, out, out_stride);
    return hipGetLastError();
}
