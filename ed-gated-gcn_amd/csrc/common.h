// Shared host-side helpers for libggcn_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "../../include/ggcn.h"

namespace ggcn {

struct DropSpec;
constexpr int kWave = 64;  // CDNA wavefront

// Thread-local message behind ggcn_last_error().
char *error_buffer();
int fail(int code, const char *fmt, ...);

inline hipStream_t as_stream(ggcn_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

// Every launch goes through this: a refused launch becomes GGCN_ELAUNCH, never a silent no-op.
inline int check_launch(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(GGCN_ELAUNCH, "%s: %s", what, hipGetErrorString(e));
    return GGCN_OK;
}

inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// ---- per-kernel launchers (one per .hip file) --------------------------------
int csr_from_dense(const void *adj, int adj_dtype, int B, int T, int64_t sb, int64_t sr, int64_t sc,
                   int32_t *rowptr, int32_t *colidx, float *vals, int64_t capacity, uint32_t *rowmask,
                   int32_t *flags, void *workspace, hipStream_t st);
size_t csr_workspace_bytes(int64_t n_rows);
int csr_transpose(const int32_t *rowptr, const int32_t *colidx, const float *vals, int B, int T, int32_t *rowptr_t,
                  int32_t *colidx_t, float *vals_t, void *workspace, hipStream_t st);
int rowmask_from_dense(const void *adj, int adj_dtype, int B, int T, int64_t sb, int64_t sr, int64_t sc,
                       uint32_t *rowmask, int32_t *flags, hipStream_t st);

int linear_fp32(const float *X, int64_t ldx, const float *W, int64_t ldw, float *Y, int64_t ldy,
                int64_t M, int K, int F, hipStream_t st);

size_t weight_pack_bytes(int K, int F, int precision);
int weight_pack(const float *W, int64_t ldw, int K, int F, int precision, bool transposed, void *wpack, hipStream_t st);
int linear_packed(const float *X, int64_t ldx, const void *wpack, float *Y, int64_t ldy,
                  int64_t M, int K, int F, int precision, hipStream_t st);

bool linear_scaled_takes(const float *X, int64_t ldx, const float *Y, int64_t ldy, int K, int F);
int linear_scaled(const float *X, int64_t ldx, const void *wpack, float *Y, int64_t ldy, int64_t M, int K, int F, const float *amax,
                  hipStream_t st);
int linear_packed_h(const void *X, int64_t ldx, const void *wpack, void *Y, int64_t ldy, int64_t M, int K,
                    int F, int precision, hipStream_t st);
int aggregate_t(const float *G, int64_t ldg, const int32_t *rowptr_t, const int32_t *colidx_t,
                const float *vals_t, const float *src_scale, int B, int T, int F, float *out, int64_t ldo,
                hipStream_t st);
int inv_denominators(const int32_t *rowptr, const float *vals, int64_t n, float *inv, hipStream_t st);
int aggregate_h(const void *Hd, int64_t ldh, const int32_t *rowptr, const int32_t *colidx, const float *vals,
                const float *bias, int B, int T, int F, const float *store_gate, const float *pool_gate_a,
                const float *pool_gate_b, void *out, int64_t ldo, float *pool_a, float *pool_b, hipStream_t st);

int layer_fused_h(const void *X, int64_t ldx, const void *wpack, const int32_t *rowptr, const int32_t *colidx,
                  const float *vals, const float *bias, int B, int T, int K, int F, const float *store_gate,
                  const float *pool_gate_a, const float *pool_gate_b, void *out, int64_t ldo, float *pool_a,
                  float *pool_b, hipStream_t st);

int aggregate(const float *Hd, int64_t ldh, const int32_t *rowptr, const int32_t *colidx,
              const float *vals, const float *bias, int B, int T, int F, const float *store_gate,
              const float *pool_gate_a, const float *pool_gate_b, float *out, int64_t ldo,
              float *pool_a, float *pool_b, hipStream_t st);

int subword_pool(const float *A, int64_t sa_b, int64_t sa_r, int64_t sa_c, const float *X, int64_t x_batch,
                 int64_t ldx, float *Y, int64_t y_batch, int64_t ldy, int B, int R, int C, int D, hipStream_t st);
int csr_rowmask(const int32_t *rowptr, const int32_t *colidx, int B, int T, uint32_t *rowmask, hipStream_t st);
int graph_operands(const uint32_t *rowmask, int B, int T, void *ops, hipStream_t st);
int graph_edge_lists(const uint32_t *rowmask, int B, int T, void *lists, hipStream_t st);   // fused_wide8.hip
int graph_operands2(const uint32_t *rowmask, int B, int T, int plane, void *ops2, hipStream_t st);
int graph_operands_weighted(const int32_t *rowptr, const int32_t *colidx, const float *vals, int B, int T, int plane, void *ops,
                            int *flag, hipStream_t st);
int layer_fused_weighted(const float *X, int64_t ldx, const void *wpack, const void *graph_opsw, const float *bias, const float *zero_mid,
                         int B, int T, int K, int F, const float *store_gate, const float *pool_gate_a, const float *pool_gate_b,
                         float *out, int64_t ldo, float *pool_a, float *pool_b, float *overlap_partial, const float *overlap_in,
                         float *overlap_out, int precision, hipStream_t st);
int layer_fused(const float *X, int64_t ldx, const void *wpack, const uint32_t *rowmask, const void *graph_ops, const float *bias,
                int B, int T, int K, int F, const float *store_gate, const float *pool_gate_a,
                const float *pool_gate_b, float *out, int64_t ldo, float *pool_a, float *pool_b,
                float *overlap_partial, const float *overlap_in, float *overlap_out, int precision, hipStream_t st,
                const struct DropSpec *drop = nullptr, const float *bias_pre = nullptr);
int dropout_mask(int64_t rows, int F, float p, uint64_t seed, int sel, float *out, hipStream_t st);

int block_fused(const float *X, int64_t ldx, const void *wpack1, const void *wpack12, const void *graph_ops, const void *graph_ops2,
                const float *bias1, const float *bias_mid, const float *bias2, int B, int T, int K, int F,
                const float *gate1, const float *gate2, float *gcn1, int64_t ld1, float *x_out, int64_t ld2,
                float *x1, float *y1, float *pool_out, float *overlap_partial, int precision, hipStream_t st,
                unsigned long long *stamps = nullptr);
int mfma_calibrate(int n_wg, int stages, unsigned long long *stamps, float *sink, hipStream_t st);   // calib.hip (diagnostics)
int overlap_reduce(const float *partials, int B, int F, float *xy, hipStream_t st);

int gate_pool_backward(const float *out, int64_t ldo, const float *store_gate, const float *gate_a,
                       const float *gate_b, const float *d_out, int64_t ldd, const float *d_pa,
                       const float *d_pb, int B, int T, int F, float *dY, int64_t ldy, float *d_sg,
                       float *d_ga, float *d_gb, float *d_bsum, hipStream_t st, const struct DropSpec *drop = nullptr);
int gate_pool_backward_agg(const float *out, int64_t ldo, const float *store_gate, const float *gate_a,
                           const float *gate_b, const float *d_out, int64_t ldd, const float *d_pa,
                           const float *d_pb, const uint32_t *rowmask, int B, int T, int F, float *dH, int64_t ldh,
                           float *d_sg, float *d_ga, float *d_gb, float *d_bsum, hipStream_t st, const struct DropSpec *drop = nullptr,
                           float *dh_amax = nullptr);
int rowmask_transpose(const uint32_t *rowmask, int B, int T, uint32_t *rowmask_t, hipStream_t st);   // gate_pool_backward_mma.hip
int gate_pool_backward_mma(const float *out, int64_t ldo, const float *store_gate, const float *gate_a, const float *gate_b,
                           const float *d_out, int64_t ldd, const float *d_pa, const float *d_pb, const void *graph_ops,
                           const void *graph_ops_t, int B, int T, int F, float *dH, int64_t ldh, float *d_sg, float *d_ga,
                           float *d_gb, float *d_bsum, float *dh_amax, hipStream_t st);
size_t colsum_workspace_bytes(int F);
int colsum(const float *X, int64_t ld, int64_t M, int F, float *out, void *workspace, hipStream_t st);

size_t dweight_workspace_bytes(int64_t N, int K, int F);
size_t dweight_bx3_workspace_bytes(int64_t N, int K, int F);
bool dweight_tn_takes(const float *X, int64_t ldx, const float *G, int64_t ldg, int64_t N, int K, int F);   // dweight_tn.hip
size_t dweight_tn_workspace_bytes(int64_t N, int K, int F);
int dweight_tn(const float *X, int64_t ldx, const float *G, int64_t ldg, int64_t N, int K, int F, float *dW, int64_t lddw,
               void *workspace, hipStream_t st);
int dweight_bx3(const float *X, int64_t ldx, const float *G, int64_t ldg, int64_t N, int K, int F, float *dW,
                int64_t lddw, void *workspace, hipStream_t st);
int weight_pack_rows(const float *W, int64_t ldw, int64_t K_valid, int F, int k_steps_total, void *pack, hipStream_t st);
int dweight(const float *X, int64_t ldx, const float *G, int64_t ldg, int64_t N, int K, int F, float *dW,
            int64_t lddw, void *workspace, hipStream_t st);

int absmax(const void *X, int is_half, int64_t ld, int64_t M, int K, float *out, hipStream_t st);
int poison_lds(uint32_t pattern, hipStream_t st);                       // range_check.hip (test hook)
int range_flag_linear(unsigned int *dst, int clear, hipStream_t st);   // linear_split.hip
int range_flag_fused(unsigned int *dst, int clear, hipStream_t st);    // fused_layer.hip
int range_flag_wide(unsigned int *dst, int clear, hipStream_t st);     // fused_wide.hip
int range_flag_wide8(unsigned int *dst, int clear, hipStream_t st);    // fused_wide8.hip
int range_flag_block8(unsigned int *dst, int clear, hipStream_t st);   // fused_block8.hip
int lab_block_fused8(const float *X, int64_t ldx, const void *wpack1, const void *wpack12, const void *graph_ops, const void *graph_ops2,
                     const float *bias1, const float *bias_mid, const float *bias2, int B, int T, int K, int F, const float *gate1,
                     const float *gate2, float *x_out, int64_t ld2, float *x1, float *y1, float *pool_out, float *overlap_partial,
                     hipStream_t st, unsigned long long *stamps = nullptr, int flags = 0);
constexpr int kBlock8RowMajor = 1, kBlock8NoDma = 2, kBlock8SameSlots = 4, kBlock8Product = 8;
bool block8_shape(int B, int T, int K, int F);
bool block8_takes(const float *X, int64_t ldx, int B, int T, int K, int F, const float *gate1, const float *gate2, const float *bias1,
                  const float *bias_mid, const float *bias2, const void *graph_ops, const void *graph_ops2, const float *x_out, int64_t ld2);
int range_flag_fused6(unsigned int *dst, int clear, hipStream_t st);   // fused6.hip (GGCN_WITH_F16MX6)

int transpose_f32(const float *W, int rows, int cols, int64_t ldw, float *Wt, hipStream_t st);
int gate_mlp(const float *aspect, int64_t lda, int B, int H, const float *w1t_a, const float *b1_a, const float *w2t_a,
             const float *b2_a, float *gate_a, const float *w1t_b, const float *b1_b, const float *w2t_b,
             const float *b2_b, float *gate_b, hipStream_t st);
int scores_head(const float *X, int64_t ldx, const float *aspect, int64_t lda, const float *logits, int64_t ldl,
                const float *fcw, int64_t ldw, const float *fcb, const float *dist, int64_t ldd, int B, int T, int H,
                int C, float *scores, int64_t lds_, float *kl_part, hipStream_t st);

int dense_head(const float *pooled, int64_t ldp, const float *Wt, int64_t ldw, const float *bias, int B, int H, int C,
               float *logits, int64_t ldl, const float *partials, int F_block, float *xy, hipStream_t st, unsigned *signal = nullptr);

size_t overlap_workspace_bytes(int B);
int gate_overlap(const float *x1, const float *y1, int B, int F, float *xy, void *workspace,
                 hipStream_t st);

}  // namespace ggcn
