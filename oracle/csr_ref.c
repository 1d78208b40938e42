/* CPU ORACLE (test infrastructure, NOT product code) -- plain C restatement of
 * the hot path in its batched-CSR form, double-precision accumulation.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * the library built from this file (oracle/Makefile -> oracle/_build/).  The
 * product library (ed-gated-gcn_amd/csrc) neither links nor calls it.
 *
 * Parity is PINNED: tests/test_oracle_golden.py checks these functions against
 * tests/golden/ *.npz, which hold outputs of the reference's own classes
 * (oracle/make_golden.py).
 *
 * What each function follows (paths relative to the reference checkout):
 *   oracle_csr_from_dense  models/gcn.py:33,35  adj.float(); the non-zero
 *                          pattern of adj[b] becomes row lists with GLOBAL node
 *                          ids b*T+j, so a batch is one block-diagonal CSR
 *   oracle_gcn_layer_csr   models/gcn.py:34-45  hidden = text.W ;
 *                          out = (adj.hidden)/(rowsum(adj)+1) (+ bias)
 *   oracle_gate_pool       models/bert_amir5.py:627-640  y*gate broadcast over
 *                          the T rows of a graph, max over those rows
 *   oracle_gate_overlap    models/bert_amir5.py:638  xy = mean_b sum_h x1*y1
 */
#include <stdint.h>
#include <stdlib.h>
#include <float.h>

/* adj: dense [B,T,T] float32 contiguous.  rowptr[N+1], colidx/vals sized B*T*T
 * by the caller.  Returns nnz. */
int64_t oracle_csr_from_dense(const float *adj, int B, int T,
                              int32_t *rowptr, int32_t *colidx, float *vals)
{
    int64_t e = 0;
    for (int b = 0; b < B; ++b)
        for (int i = 0; i < T; ++i) {
            rowptr[(int64_t)b * T + i] = (int32_t)e;
            const float *row = adj + ((int64_t)b * T + i) * T;
            for (int j = 0; j < T; ++j)
                if (row[j] != 0.0f) {
                    colidx[e] = b * T + j;
                    vals[e] = row[j];
                    ++e;
                }
        }
    rowptr[(int64_t)B * T] = (int32_t)e;
    return e;
}

/* X [N,K], W [K,F] (in x out, gcn.py:18), bias [F] or NULL, vals NULL => all 1.
 * out [N,F].  hidden is materialised as the reference does (gcn.py:34). */
int oracle_gcn_layer_csr(const float *X, const float *W, const float *bias,
                         const int32_t *rowptr, const int32_t *colidx, const float *vals,
                         int64_t N, int K, int F, float *out)
{
    double *hidden = (double *)malloc(sizeof(double) * (size_t)N * F);
    if (!hidden) return 1;
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < N; ++i) {
        double *h = hidden + i * F;
        for (int f = 0; f < F; ++f) h[f] = 0.0;
        for (int k = 0; k < K; ++k) {
            const double x = X[i * K + k];
            const float *w = W + (int64_t)k * F;
            for (int f = 0; f < F; ++f) h[f] += x * (double)w[f];
        }
    }
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < N; ++i) {
        double denom = 1.0;                                  /* gcn.py:35 "+ 1" */
        for (int32_t e = rowptr[i]; e < rowptr[i + 1]; ++e)
            denom += vals ? (double)vals[e] : 1.0;
        for (int f = 0; f < F; ++f) {
            double acc = 0.0;
            for (int32_t e = rowptr[i]; e < rowptr[i + 1]; ++e)
                acc += (vals ? (double)vals[e] : 1.0) * hidden[(int64_t)colidx[e] * F + f];
            acc /= denom;                                    /* gcn.py:41 */
            if (bias) acc += (double)bias[f];                /* gcn.py:43 */
            out[i * F + f] = (float)acc;
        }
    }
    free(hidden);
    return 0;
}

/* y [B*T,F]; gate [B,F] or NULL (=1).  gated [B*T,F] or NULL; pooled [B,F]. */
void oracle_gate_pool(const float *y, const float *gate, int B, int T, int F,
                      float *gated, float *pooled)
{
    for (int b = 0; b < B; ++b)
        for (int f = 0; f < F; ++f) {
            float m = -FLT_MAX;
            const float g = gate ? gate[(int64_t)b * F + f] : 1.0f;
            for (int t = 0; t < T; ++t) {
                const int64_t at = ((int64_t)b * T + t) * F + f;
                const float v = y[at] * g;                   /* bert_amir5.py:627,631,639 */
                if (gated) gated[at] = v;
                if (v > m) m = v;                            /* bert_amir5.py:635,636,640 */
            }
            pooled[(int64_t)b * F + f] = m;
        }
}

double oracle_gate_overlap(const float *x1, const float *y1, int B, int F)
{
    double tot = 0.0;
    for (int b = 0; b < B; ++b) {
        double s = 0.0;
        for (int f = 0; f < F; ++f) s += (double)x1[(int64_t)b * F + f] * (double)y1[(int64_t)b * F + f];
        tot += s;
    }
    return tot / (double)B;
}
