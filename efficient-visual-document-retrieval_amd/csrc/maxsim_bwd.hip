// Backward of MaxSim w.r.t. the page embeddings, and the InfoNCE-distillation loss with its gradient.
//
// Reference behaviour being replaced: autograd of evaluator/retrieval.py:195-210 materialises a dense
// one-hot (Q, C, Lq, Lp) gradient and runs a dense einsum over it (mainv2_iter_distill_infonce.py:290).
// Here the forward kernel recorded argmax[q,p,n]; the backward is a gather of <= B*Lq query-token rows
// per page, summed in LDS (ds_add_f32) and written once:  HBM/LDS-bound, no MFMA, no global atomics.
//   bytes per page: read 2*B*Lq (argmax) + <= B*Lq*512 (Q rows, L2-resident) ; write Lp*512 (dP).
#include "evdr_common.h"

namespace {

constexpr int BW_THREADS = 256;
constexpr int BW_ROWS = 128;          // patch rows of dP accumulated per workgroup: 64 KiB of LDS

__global__ void __launch_bounds__(BW_THREADS) maxsim_bwd_kernel(const float* __restrict__ g,
                                                               const float* __restrict__ Q,
                                                               const uint8_t* __restrict__ qmask,
                                                               const uint8_t* __restrict__ pmask,
                                                               const uint16_t* __restrict__ argmax,
                                                               float* __restrict__ dP, int nq, int lq, int np, int lp) {
    __shared__ __attribute__((aligned(16))) float acc[BW_ROWS * EVDR_D];
    __shared__ int sh_has;
    const int page = blockIdx.x;
    const int r0 = blockIdx.y * BW_ROWS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    if (tid == 0) sh_has = (pmask == nullptr) ? 1 : 0;
    for (int i = tid; i < BW_ROWS * EVDR_D / 4; i += BW_THREADS) reinterpret_cast<f32x4*>(acc)[i] = f32x4{0, 0, 0, 0};
    __syncthreads();
    if (pmask != nullptr) {                               // has(p) = any(pmask[p])  (retrieval.py:192)
        int any = 0;
        for (int i = tid; i < lp; i += BW_THREADS) any |= pmask[(int64_t)page * lp + i];
        if (any) sh_has = 1;                              // benign race: every writer stores 1
    }
    __syncthreads();
    if (sh_has) {
        const int npairs = nq * lq;
        for (int i = wave; i < npairs; i += BW_THREADS / 64) {
            const int q = i / lq, n = i - q * lq;
            const int a = (int)argmax[((int64_t)q * np + page) * lq + n] - r0;
            if (a < 0 || a >= BW_ROWS) continue;          // wave-uniform
            if (qmask != nullptr && qmask[i] == 0) continue;
            const float w = g[(int64_t)q * np + page];
            if (w == 0.f) continue;
            const float2 qv = *reinterpret_cast<const float2*>(Q + (int64_t)i * EVDR_D + lane * 2);
            atomicAdd(&acc[a * EVDR_D + lane * 2], w * qv.x);
            atomicAdd(&acc[a * EVDR_D + lane * 2 + 1], w * qv.y);
        }
    }
    __syncthreads();
    const int rows = min(BW_ROWS, lp - r0);
    f32x4* out = reinterpret_cast<f32x4*>(dP + ((int64_t)page * lp + r0) * EVDR_D);
    for (int i = tid; i < rows * EVDR_D / 4; i += BW_THREADS) out[i] = reinterpret_cast<const f32x4*>(acc)[i];
}

// ---- infonce_distillation_loss (criterion.py:56-68) and d loss / d score_s, one workgroup per query row
__device__ __forceinline__ float wave_max(float v) {
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

__global__ void __launch_bounds__(256) infonce_row_kernel(const float* __restrict__ ss, const float* __restrict__ st,
                                                         int64_t n, float temp, float inv_tb,
                                                         float* __restrict__ row_loss, float* __restrict__ dscore) {
    __shared__ float red[4];
    __shared__ int redi[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* s = ss + (int64_t)blockIdx.x * n;
    const float* t = st + (int64_t)blockIdx.x * n;
    // teacher argmax (first maximal index) and student max
    float tbest = -__builtin_inff(), smax = -__builtin_inff();
    int tidx = 0x7FFFFFFF;
    for (int64_t i = tid; i < n; i += 256) {
        const float tv = t[i];
        if (tv > tbest) { tbest = tv; tidx = (int)i; }
        smax = fmaxf(smax, s[i] / temp);
    }
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(tbest, o);
        const int oi = __shfl_xor(tidx, o);
        if (ov > tbest || (ov == tbest && oi < tidx)) { tbest = ov; tidx = oi; }
    }
    smax = wave_max(smax);
    if (lane == 0) { red[wave] = tbest; redi[wave] = tidx; }
    __syncthreads();
    tbest = red[0]; tidx = redi[0];
    for (int w = 1; w < 4; ++w)
        if (red[w] > tbest || (red[w] == tbest && redi[w] < tidx)) { tbest = red[w]; tidx = redi[w]; }
    __syncthreads();
    if (lane == 0) red[wave] = smax;
    __syncthreads();
    smax = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float sum = 0.f;
    for (int64_t i = tid; i < n; i += 256) sum += expf(s[i] / temp - smax);
    sum = wave_sum(sum);
    if (lane == 0) red[wave] = sum;
    __syncthreads();
    sum = red[0] + red[1] + red[2] + red[3];
    const float lse = smax + logf(sum);
    if (tid == 0) row_loss[blockIdx.x] = lse - s[tidx] / temp;
    if (dscore != nullptr) {
        float* d = dscore + (int64_t)blockIdx.x * n;
        const float inv_sum = 1.f / sum;
        for (int64_t i = tid; i < n; i += 256) {
            const float pr = expf(s[i] / temp - smax) * inv_sum;
            d[i] = (pr - ((int)i == tidx ? 1.f : 0.f)) * inv_tb;
        }
    }
}

__global__ void __launch_bounds__(256) mean_kernel(const float* __restrict__ x, int64_t n, float* __restrict__ out) {
    __shared__ float red[4];
    float s = 0.f;
    for (int64_t i = threadIdx.x; i < n; i += 256) s += x[i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = (red[0] + red[1] + red[2] + red[3]) / (float)n;
}

}  // namespace

hipError_t evdr_launch_maxsim_bwd(const float* g, const float* Q, const uint8_t* qmask, const uint8_t* pmask,
                                  const uint16_t* argmax, float* dP, int64_t nq, int64_t lq, int64_t np, int64_t lp,
                                  hipStream_t stream) {
    if (np == 0 || lp == 0) return hipSuccess;
    dim3 grid((unsigned)np, (unsigned)((lp + BW_ROWS - 1) / BW_ROWS));
    hipLaunchKernelGGL(maxsim_bwd_kernel, grid, dim3(BW_THREADS), 0, stream, g, Q, qmask, pmask, argmax, dP, (int)nq,
                       (int)lq, (int)np, (int)lp);
    return hipGetLastError();
}

hipError_t evdr_launch_infonce(const float* ss, const float* st, int64_t b, int64_t n, float temperature, float* loss,
                               float* dscore, float* row_loss, hipStream_t stream) {
    if (b == 0) return hipSuccess;
    hipLaunchKernelGGL(infonce_row_kernel, dim3((unsigned)b), dim3(256), 0, stream, ss, st, n, temperature,
                       1.f / (temperature * (float)b), row_loss, dscore);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(mean_kernel, dim3(1), dim3(256), 0, stream, row_loss, b, loss);
    return hipGetLastError();
}
