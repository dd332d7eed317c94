/* CPU oracle in plain C for the MaxSim forward.  TEST INFRASTRUCTURE ONLY (same rule as maxsim_oracle.py: tests,
 * smoke() and the cpu_baseline leg may use it, the product path never does).
 *
 * A second, independent restatement of evaluator/retrieval.py:166-213 (score_multi_vector_masked) next to the torch
 * one: scalar loops, double accumulation, no library.  Pinned the same way -- tests/test_oracle_golden.py checks it
 * against the fixtures produced by the reference's own function (tests/golden/a1_*.npz).
 *
 *   sim[q,p,n,m] = Q[q,n,:] . P[p,m,:]                               (:190, einsum)
 *   sim          = pmask[p,m] ? sim : -1e4                            (:185,198, masked_fill)
 *   best, arg    = max over m (first maximal index)                   (:201)
 *   out[q,p]     = sum_n best * any(pmask[p,:]) * qmask[q,n]          (:192,204-209)
 *
 * Build: gcc -O2 -shared -fPIC -o oracle/_build/libmaxsim_oracle.so oracle/maxsim_oracle.c   (done by __graft_entry__.build())
 */
#include <stddef.h>

void evdr_oracle_maxsim(const float* Q, const float* P, const unsigned char* qmask, const unsigned char* pmask,
                        long nq, long lq, long np, long lp, long d, double* out, int* argmax_or_null) {
    for (long q = 0; q < nq; ++q)
        for (long p = 0; p < np; ++p) {
            int has = 0;
            for (long m = 0; m < lp; ++m) has |= pmask[p * lp + m] != 0;
            double total = 0.0;
            for (long n = 0; n < lq; ++n) {
                double best = 0.0;
                long arg = -1;
                for (long m = 0; m < lp; ++m) {
                    double sim = -1e4;
                    if (pmask[p * lp + m]) {
                        const float* a = Q + (q * lq + n) * d;
                        const float* b = P + (p * lp + m) * d;
                        sim = 0.0;
                        for (long i = 0; i < d; ++i) sim += (double)a[i] * (double)b[i];
                    }
                    if (arg < 0 || sim > best) {
                        best = sim;
                        arg = m;
                    }
                }
                if (argmax_or_null) argmax_or_null[(q * np + p) * lq + n] = (int)arg;
                total += best * (has ? 1.0 : 0.0) * (qmask[q * lq + n] ? 1.0 : 0.0);
            }
            out[q * np + p] = total;
        }
}
