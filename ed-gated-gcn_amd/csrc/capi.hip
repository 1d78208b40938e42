// extern "C" surface of libggcn_hip.so -- see include/ggcn.h for the contract.
// Argument checking lives here and in the per-kernel launchers; nothing in this
// library allocates, frees, copies to the host or synchronises.
#include "common.h"
#include <cstdlib>
#include "dropout_hash.h"

#include <cstring>

namespace ggcn {

char *error_buffer()
{
    static thread_local char buf[512] = "";
    return buf;
}

int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(error_buffer(), 512, fmt, ap);
    va_end(ap);
    return code;
}

}  // namespace ggcn

using namespace ggcn;

extern "C" {

int ggcn_abi_version(void) { return GGCN_ABI_VERSION; }

int ggcn_has_f16mx6(void)
{
#ifdef GGCN_WITH_F16MX6
    return 1;
#else
    return 0;
#endif
}

const char *ggcn_last_error(void) { return error_buffer(); }

size_t ggcn_csr_workspace_bytes(int64_t n_rows) { return csr_workspace_bytes(n_rows); }

int ggcn_csr_from_dense(const void *adj, int adj_dtype, int B, int T, int64_t stride_b,
                        int64_t stride_r, int64_t stride_c, int32_t *rowptr, int32_t *colidx,
                        float *vals, int64_t capacity, uint32_t *rowmask, int32_t *flags,
                        void *workspace, ggcn_stream_t stream)
{
    return csr_from_dense(adj, adj_dtype, B, T, stride_b, stride_r, stride_c, rowptr, colidx, vals,
                          capacity, rowmask, flags, workspace, as_stream(stream));
}

int ggcn_csr_transpose(const int32_t *rowptr, const int32_t *colidx, const float *vals, int B, int T,
                       int32_t *rowptr_t, int32_t *colidx_t, float *vals_t, void *workspace, ggcn_stream_t stream)
{
    return csr_transpose(rowptr, colidx, vals, B, T, rowptr_t, colidx_t, vals_t, workspace, as_stream(stream));
}

int ggcn_rowmask_from_dense(const void *adj, int adj_dtype, int B, int T, int64_t stride_b, int64_t stride_r,
                            int64_t stride_c, uint32_t *rowmask, int32_t *flags, ggcn_stream_t stream)
{
    return rowmask_from_dense(adj, adj_dtype, B, T, stride_b, stride_r, stride_c, rowmask, flags, as_stream(stream));
}

int ggcn_csr_rowmask(const int32_t *rowptr, const int32_t *colidx, int B, int T, uint32_t *rowmask,
                     ggcn_stream_t stream)
{
    return csr_rowmask(rowptr, colidx, B, T, rowmask, as_stream(stream));
}

size_t ggcn_graph_operands_bytes(int B) { return B > 0 ? (size_t)B * GGCN_GRAPH_OPS_BYTES : 0; }
size_t ggcn_graph_operands2_bytes(int B) { return B > 0 ? (size_t)B * GGCN_GRAPH_OPS2_BYTES : 0; }
int ggcn_graph_operands2(const uint32_t *rowmask, int B, int T, int plane, void *graph_ops2, ggcn_stream_t stream)
{
    return graph_operands2(rowmask, B, T, plane, graph_ops2, as_stream(stream));
}

size_t ggcn_graph_edge_lists_bytes(int B) { return B > 0 ? (size_t)B * GGCN_EDGE_LISTS_BYTES : 0; }

int ggcn_graph_edge_lists(const uint32_t *rowmask, int B, int T, void *lists, ggcn_stream_t stream)
{
    return graph_edge_lists(rowmask, B, T, lists, as_stream(stream));
}

int ggcn_graph_operands(const uint32_t *rowmask, int B, int T, void *graph_ops, ggcn_stream_t stream)
{
    return graph_operands(rowmask, B, T, graph_ops, as_stream(stream));
}

int ggcn_layer_fused(const float *X, int64_t ldx, const void *wpack, const uint32_t *rowmask, const void *graph_ops,
                     const float *bias, int B, int T, int K, int F, const float *store_gate,
                     const float *pool_gate_a, const float *pool_gate_b, float *out, int64_t ldo,
                     float *pool_a, float *pool_b, float *overlap_partial, const float *overlap_in,
                     float *overlap_out, int precision, ggcn_stream_t stream)
{
    return layer_fused(X, ldx, wpack, rowmask, graph_ops, bias, B, T, K, F, store_gate, pool_gate_a, pool_gate_b, out,
                       ldo, pool_a, pool_b, overlap_partial, overlap_in, overlap_out, precision, as_stream(stream));
}

int ggcn_layer_fused_prebias(const float *X, int64_t ldx, const void *wpack, const uint32_t *rowmask, const float *bias,
                             const float *bias_pre, int B, int T, int K, int F, const float *store_gate,
                             const float *pool_gate_a, const float *pool_gate_b, float *out, int64_t ldo,
                             float *pool_a, float *pool_b, int precision, ggcn_stream_t stream)
{
    if (!bias_pre) return fail(GGCN_EINVAL, "ggcn_layer_fused_prebias: null bias_pre (ggcn_layer_fused is the plain form)");
    return layer_fused(X, ldx, wpack, rowmask, nullptr, bias, B, T, K, F, store_gate, pool_gate_a, pool_gate_b, out,
                       ldo, pool_a, pool_b, nullptr, nullptr, nullptr, precision, as_stream(stream), nullptr, bias_pre);
}

int ggcn_graph_operands_weighted(const int32_t *rowptr, const int32_t *colidx, const float *vals, int B, int T, int plane,
                                 void *graph_opsw, int32_t *flag, ggcn_stream_t stream)
{
    return graph_operands_weighted(rowptr, colidx, vals, B, T, plane, graph_opsw, flag, as_stream(stream));
}

int ggcn_layer_fused_weighted(const float *X, int64_t ldx, const void *wpack, const void *graph_opsw, const float *bias,
                              const float *zero_mid, int B, int T, int K, int F, const float *store_gate,
                              const float *pool_gate_a, const float *pool_gate_b, float *out, int64_t ldo, float *pool_a,
                              float *pool_b, float *overlap_partial, const float *overlap_in, float *overlap_out, int precision,
                              ggcn_stream_t stream)
{
    return layer_fused_weighted(X, ldx, wpack, graph_opsw, bias, zero_mid, B, T, K, F, store_gate, pool_gate_a, pool_gate_b, out, ldo,
                                pool_a, pool_b, overlap_partial, overlap_in, overlap_out, precision, as_stream(stream));
}

int ggcn_block_fused(const float *X, int64_t ldx, const void *wpack1, const void *wpack12, const void *graph_ops,
                     const void *graph_ops2, const float *bias1, const float *bias_mid, const float *bias2, int B, int T, int K, int F,
                     const float *gate1, const float *gate2, float *gcn1, int64_t ld1, float *x_out, int64_t ld2,
                     float *x1, float *y1, float *pool_out, float *overlap_partial, int precision,
                     ggcn_stream_t stream)
{
    return block_fused(X, ldx, wpack1, wpack12, graph_ops, graph_ops2, bias1, bias_mid, bias2, B, T, K, F, gate1, gate2, gcn1, ld1,
                       x_out, ld2, x1, y1, pool_out, overlap_partial, precision, as_stream(stream));
}

int ggcn_overlap_reduce(const float *partials, int B, int F, float *xy, ggcn_stream_t stream)
{
    return overlap_reduce(partials, B, F, xy, as_stream(stream));
}

size_t ggcn_weight_pack_bytes(int K, int F, int precision) { return weight_pack_bytes(K, F, precision); }

int ggcn_weight_pack(const float *W, int64_t ldw, int K, int F, int precision, int transposed, void *wpack,
                     ggcn_stream_t stream)
{
    return weight_pack(W, ldw, K, F, precision, transposed != 0, wpack, as_stream(stream));
}

int ggcn_aggregate_t(const float *G, int64_t ldg, const int32_t *rowptr_t, const int32_t *colidx_t,
                     const float *vals_t, const float *src_scale, int B, int T, int F, float *out, int64_t ldo,
                     ggcn_stream_t stream)
{
    return aggregate_t(G, ldg, rowptr_t, colidx_t, vals_t, src_scale, B, T, F, out, ldo, as_stream(stream));
}

int ggcn_inv_denominators(const int32_t *rowptr, const float *vals, int64_t n_rows, float *inv,
                          ggcn_stream_t stream)
{
    return inv_denominators(rowptr, vals, n_rows, inv, as_stream(stream));
}

int ggcn_linear(const float *X, int64_t ldx, const float *W, int64_t ldw, const void *wpack, float *Y,
                int64_t ldy, int64_t M, int K, int F, int precision, ggcn_stream_t stream)
{
    if (!X || !Y) return fail(GGCN_EINVAL, "ggcn_linear: null pointer");
    if (M <= 0 || K <= 0 || F <= 0)
        return fail(GGCN_EINVAL, "ggcn_linear: M=%lld K=%d F=%d must be positive", (long long)M, K, F);
    if (ldx < K || ldy < F) return fail(GGCN_EINVAL, "ggcn_linear: leading dimension too small");
    switch (precision) {
        case GGCN_PREC_BF16X3:
        case GGCN_PREC_F16MX8:
            return linear_packed(X, ldx, wpack, Y, ldy, M, K, F, precision, as_stream(stream));
        case GGCN_PREC_FP32:
            if (!W) return fail(GGCN_EINVAL, "ggcn_linear(fp32): W is NULL");
            if (ldw < F) return fail(GGCN_EINVAL, "ggcn_linear(fp32): ldw < F");
            return linear_fp32(X, ldx, W, ldw, Y, ldy, M, K, F, as_stream(stream));
        default:
            return fail(GGCN_EINVAL, "ggcn_linear: unknown precision %d", precision);
    }
}

int ggcn_aggregate(const float *Hd, int64_t ldh, const int32_t *rowptr, const int32_t *colidx,
                   const float *vals, const float *bias, int B, int T, int F, const float *store_gate,
                   const float *pool_gate_a, const float *pool_gate_b, float *out, int64_t ldo,
                   float *pool_a, float *pool_b, ggcn_stream_t stream)
{
    return aggregate(Hd, ldh, rowptr, colidx, vals, bias, B, T, F, store_gate, pool_gate_a, pool_gate_b,
                     out, ldo, pool_a, pool_b, as_stream(stream));
}

int ggcn_linear_h(const void *X, int64_t ldx, const void *wpack, void *Y, int64_t ldy, int64_t M, int K, int F,
                  int precision, ggcn_stream_t stream)
{
    if (precision != GGCN_PREC_BF16X3 && precision != GGCN_PREC_F16MX8 && precision != GGCN_PREC_F16)
        return fail(GGCN_EUNSUPPORTED, "ggcn_linear_h: precision %d (use bf16x3, f16mx8 or f16)", precision);
    if (!X || !Y) return fail(GGCN_EINVAL, "ggcn_linear_h: null pointer");
    if (M <= 0 || K <= 0 || F <= 0)
        return fail(GGCN_EINVAL, "ggcn_linear_h: M=%lld K=%d F=%d must be positive", (long long)M, K, F);
    if (ldx < K || ldy < F) return fail(GGCN_EINVAL, "ggcn_linear_h: leading dimension too small");
    return linear_packed_h(X, ldx, wpack, Y, ldy, M, K, F, precision, as_stream(stream));
}

int ggcn_aggregate_h(const void *Hd, int64_t ldh, const int32_t *rowptr, const int32_t *colidx,
                     const float *vals, const float *bias, int B, int T, int F, const float *store_gate,
                     const float *pool_gate_a, const float *pool_gate_b, void *out, int64_t ldo,
                     float *pool_a, float *pool_b, ggcn_stream_t stream)
{
    return aggregate_h(Hd, ldh, rowptr, colidx, vals, bias, B, T, F, store_gate, pool_gate_a, pool_gate_b, out,
                       ldo, pool_a, pool_b, as_stream(stream));
}

int ggcn_layer_fused_h(const void *X, int64_t ldx, const void *wpack, const int32_t *rowptr, const int32_t *colidx,
                       const float *vals, const float *bias, int B, int T, int K, int F, const float *store_gate,
                       const float *pool_gate_a, const float *pool_gate_b, void *out, int64_t ldo, float *pool_a,
                       float *pool_b, ggcn_stream_t stream)
{
    return layer_fused_h(X, ldx, wpack, rowptr, colidx, vals, bias, B, T, K, F, store_gate, pool_gate_a, pool_gate_b,
                         out, ldo, pool_a, pool_b, as_stream(stream));
}

int ggcn_gate_pool_backward(const float *out, int64_t ldo, const float *store_gate, const float *gate_a,
                            const float *gate_b, const float *d_out, int64_t ldd, const float *d_pa,
                            const float *d_pb, int B, int T, int F, float *dY, int64_t ldy, float *d_sg,
                            float *d_ga, float *d_gb, float *d_bsum, ggcn_stream_t stream)
{
    return gate_pool_backward(out, ldo, store_gate, gate_a, gate_b, d_out, ldd, d_pa, d_pb, B, T, F, dY, ldy, d_sg,
                              d_ga, d_gb, d_bsum, as_stream(stream));
}

int ggcn_gate_pool_backward_drop(const float *out, int64_t ldo, const float *store_gate, const float *gate_a,
                                 const float *gate_b, const float *d_out, int64_t ldd, const float *d_pa,
                                 const float *d_pb, int B, int T, int F, float *dY, int64_t ldy, float *d_sg,
                                 float *d_ga, float *d_gb, float *d_bsum, float p, uint64_t seed, int sel_store,
                                 int sel_a, int sel_b, ggcn_stream_t stream)
{
    if (!(p >= 0.0f && p < 1.0f) || sel_store < 0 || sel_store > 2 || sel_a < 0 || sel_a > 2 || sel_b < 0 || sel_b > 2)
        return fail(GGCN_EINVAL, "ggcn_gate_pool_backward_drop: p=%g streams %d %d %d", (double)p, sel_store, sel_a, sel_b);
    const DropSpec d = make_drop_spec(p, seed, sel_store, sel_a, sel_b);
    return gate_pool_backward(out, ldo, store_gate, gate_a, gate_b, d_out, ldd, d_pa, d_pb, B, T, F, dY, ldy, d_sg,
                              d_ga, d_gb, d_bsum, as_stream(stream), &d);
}

int ggcn_gate_pool_backward_agg(const float *out, int64_t ldo, const float *store_gate, const float *gate_a,
                                const float *gate_b, const float *d_out, int64_t ldd, const float *d_pa,
                                const float *d_pb, const uint32_t *rowmask, int B, int T, int F, float *dH, int64_t ldh,
                                float *d_sg, float *d_ga, float *d_gb, float *d_bsum, float p, uint64_t seed, int sel_store,
                                int sel_a, int sel_b, float *dh_amax, ggcn_stream_t stream)
{
    if (!(p >= 0.0f && p < 1.0f) || sel_store < 0 || sel_store > 2 || sel_a < 0 || sel_a > 2 || sel_b < 0 || sel_b > 2)
        return fail(GGCN_EINVAL, "ggcn_gate_pool_backward_agg: p=%g streams %d %d %d", (double)p, sel_store, sel_a, sel_b);
    const DropSpec d = make_drop_spec(p, seed, sel_store, sel_a, sel_b);
    return gate_pool_backward_agg(out, ldo, store_gate, gate_a, gate_b, d_out, ldd, d_pa, d_pb, rowmask, B, T, F, dH, ldh, d_sg,
                                  d_ga, d_gb, d_bsum, as_stream(stream), p > 0.0f ? &d : nullptr, dh_amax);
}

int ggcn_block_fused_form(int B, int T, int K, int F)
{
    const char *form = getenv("GGCN_BLOCK_FORM");
    return (block8_shape(B, T, K, F) && !(form && form[0] == '4')) ? 8 : 4;
}

int ggcn_lab_block_fused8(const float *X, int64_t ldx, const void *wpack1, const void *wpack12, const void *graph_ops,
                          const void *graph_ops2, const float *bias1, const float *bias_mid, const float *bias2, int B, int T, int K, int F,
                          const float *gate1, const float *gate2, float *x_out, int64_t ld2, float *x1, float *y1, float *pool_out,
                          float *overlap_partial, uint64_t *stamps, ggcn_stream_t stream)
{
    const int flags = (getenv("GGCN_LAB_BLOCK8_ROWMAJOR") ? kBlock8RowMajor : 0) | (getenv("GGCN_LAB_BLOCK8_NODMA") ? kBlock8NoDma : 0) |
                      (getenv("GGCN_LAB_BLOCK8_SAMESLOTS") ? kBlock8SameSlots : 0);   // (the experiment's switches, read per call)
    return lab_block_fused8(X, ldx, wpack1, wpack12, graph_ops, graph_ops2, bias1, bias_mid, bias2, B, T, K, F, gate1, gate2, x_out, ld2,
                            x1, y1, pool_out, overlap_partial, as_stream(stream), reinterpret_cast<unsigned long long *>(stamps), flags);
}

int ggcn_rowmask_transpose(const uint32_t *rowmask, int B, int T, uint32_t *rowmask_t, ggcn_stream_t stream)
{
    return rowmask_transpose(rowmask, B, T, rowmask_t, as_stream(stream));
}

int ggcn_gate_pool_backward_mma(const float *out, int64_t ldo, const float *store_gate, const float *gate_a, const float *gate_b,
                                const float *d_out, int64_t ldd, const float *d_pa, const float *d_pb, const void *graph_ops,
                                const void *graph_ops_t, int B, int T, int F, float *dH, int64_t ldh, float *d_sg, float *d_ga,
                                float *d_gb, float *d_bsum, float *dh_amax, ggcn_stream_t stream)
{
    return gate_pool_backward_mma(out, ldo, store_gate, gate_a, gate_b, d_out, ldd, d_pa, d_pb, graph_ops, graph_ops_t, B, T, F, dH, ldh,
                                  d_sg, d_ga, d_gb, d_bsum, dh_amax, as_stream(stream));
}

int ggcn_linear_scaled(const float *X, int64_t ldx, const void *wpack, float *Y, int64_t ldy, int64_t M, int K, int F,
                       const float *amax, ggcn_stream_t stream)
{
    return linear_scaled(X, ldx, wpack, Y, ldy, M, K, F, amax, as_stream(stream));
}

int ggcn_layer_fused_drop(const float *X, int64_t ldx, const void *wpack, const uint32_t *rowmask, const void *graph_ops,
                          const float *bias, int B, int T, int K, int F, const float *store_gate,
                          const float *pool_gate_a, const float *pool_gate_b, float *out, int64_t ldo,
                          float *pool_a, float *pool_b, int precision, float p, uint64_t seed, int sel_store,
                          int sel_a, int sel_b, ggcn_stream_t stream)
{
    if (!(p >= 0.0f && p < 1.0f) || sel_store < 0 || sel_store > 2 || sel_a < 0 || sel_a > 2 || sel_b < 0 || sel_b > 2)
        return fail(GGCN_EINVAL, "ggcn_layer_fused_drop: p=%g streams %d %d %d", (double)p, sel_store, sel_a, sel_b);
    const DropSpec d = make_drop_spec(p, seed, sel_store, sel_a, sel_b);
    return layer_fused(X, ldx, wpack, rowmask, graph_ops, bias, B, T, K, F, store_gate, pool_gate_a, pool_gate_b, out,
                       ldo, pool_a, pool_b, nullptr, nullptr, nullptr, precision, as_stream(stream), &d);
}

int ggcn_dropout_mask(int64_t rows, int F, float p, uint64_t seed, int stream_id, float *mask, ggcn_stream_t stream)
{
    return dropout_mask(rows, F, p, seed, stream_id, mask, as_stream(stream));
}

size_t ggcn_colsum_workspace_bytes(int F) { return colsum_workspace_bytes(F); }

int ggcn_colsum(const float *X, int64_t ld, int64_t M, int F, float *out, void *workspace, ggcn_stream_t stream)
{
    return colsum(X, ld, M, F, out, workspace, as_stream(stream));
}

size_t ggcn_dweight_workspace_bytes(int64_t n_rows, int K, int F, int precision)
{
    return precision == GGCN_PREC_FP32 ? dweight_workspace_bytes(n_rows, K, F) : dweight_bx3_workspace_bytes(n_rows, K, F);
}

int ggcn_dweight(const float *X, int64_t ldx, const float *dH, int64_t ldg, int64_t n_rows, int K, int F,
                 float *dW, int64_t lddw, int precision, void *workspace, ggcn_stream_t stream)
{
    if (precision == GGCN_PREC_FP32)
        return dweight(X, ldx, dH, ldg, n_rows, K, F, dW, lddw, workspace, as_stream(stream));
    if (precision != GGCN_PREC_BF16X3)
        return fail(GGCN_EUNSUPPORTED, "ggcn_dweight: precision %d (gradients need fp32 range: use bf16x3 or fp32)", precision);
    return dweight_bx3(X, ldx, dH, ldg, n_rows, K, F, dW, lddw, workspace, as_stream(stream));
}

int ggcn_subword_pool(const float *A, int64_t sa_b, int64_t sa_r, int64_t sa_c, const float *X, int64_t x_batch,
                      int64_t ldx, float *Y, int64_t y_batch, int64_t ldy, int B, int R, int C, int D,
                      ggcn_stream_t stream)
{
    return subword_pool(A, sa_b, sa_r, sa_c, X, x_batch, ldx, Y, y_batch, ldy, B, R, C, D, as_stream(stream));
}

int ggcn_absmax(const void *X, int is_half, int64_t ld, int64_t M, int K, float *out, ggcn_stream_t stream)
{
    return absmax(X, is_half, ld, M, K, out, as_stream(stream));
}

int ggcn_debug_poison_lds(uint32_t pattern, ggcn_stream_t stream) { return poison_lds(pattern, as_stream(stream)); }

int ggcn_debug_mfma_calibrate(int n_wg, int stages, uint64_t *stamps, float *sink, ggcn_stream_t stream)
{
    return mfma_calibrate(n_wg, stages, reinterpret_cast<unsigned long long *>(stamps), sink, as_stream(stream));
}

int ggcn_debug_block_fused_stamped(const float *X, int64_t ldx, const void *wpack1, const void *wpack12, const void *graph_ops,
                                   const void *graph_ops2, const float *bias1, const float *bias_mid, const float *bias2, int B,
                                   int T, int K, int F, const float *gate1, const float *gate2, float *gcn1, int64_t ld1,
                                   float *x_out, int64_t ld2, float *x1, float *y1, float *pool_out, float *overlap_partial,
                                   int precision, uint64_t *stamps, ggcn_stream_t stream)
{
    if (!stamps) return fail(GGCN_EINVAL, "ggcn_debug_block_fused_stamped: null stamp buffer");
    return block_fused(X, ldx, wpack1, wpack12, graph_ops, graph_ops2, bias1, bias_mid, bias2, B, T, K, F, gate1, gate2, gcn1, ld1,
                       x_out, ld2, x1, y1, pool_out, overlap_partial, precision, as_stream(stream),
                       reinterpret_cast<unsigned long long *>(stamps));
}

int ggcn_range_flag(uint32_t *flag, int clear, ggcn_stream_t stream)
{
    if (!flag) return fail(GGCN_EINVAL, "ggcn_range_flag: null flag pointer");
    // one copy of the flag per translation unit that holds an f16mx8 main loop
    int rc = range_flag_linear(flag, clear, as_stream(stream));
    rc = rc ? rc : range_flag_fused(flag, clear, as_stream(stream));
    rc = rc ? rc : range_flag_wide(flag, clear, as_stream(stream));
    rc = rc ? rc : range_flag_wide8(flag, clear, as_stream(stream));
    rc = rc ? rc : range_flag_block8(flag, clear, as_stream(stream));
#ifdef GGCN_WITH_F16MX6
    rc = rc ? rc : range_flag_fused6(flag, clear, as_stream(stream));
#endif
    return rc;
}

int ggcn_transpose(const float *W, int rows, int cols, int64_t ldw, float *Wt, ggcn_stream_t stream)
{
    return transpose_f32(W, rows, cols, ldw, Wt, as_stream(stream));
}

int ggcn_gate_mlp(const float *aspect, int64_t lda, int B, int H, const float *w1t_a, const float *b1_a,
                  const float *w2t_a, const float *b2_a, float *gate_a, const float *w1t_b, const float *b1_b,
                  const float *w2t_b, const float *b2_b, float *gate_b, ggcn_stream_t stream)
{
    return gate_mlp(aspect, lda, B, H, w1t_a, b1_a, w2t_a, b2_a, gate_a, w1t_b, b1_b, w2t_b, b2_b, gate_b, as_stream(stream));
}

int ggcn_scores_head(const float *X, int64_t ldx, const float *aspect, int64_t lda, const float *logits, int64_t ldl,
                     const float *fc_weight, int64_t ldw, const float *fc_bias, const float *dist, int64_t ldd, int B,
                     int T, int H, int C, float *scores, int64_t ld_scores, float *kl_part, ggcn_stream_t stream)
{
    return scores_head(X, ldx, aspect, lda, logits, ldl, fc_weight, ldw, fc_bias, dist, ldd, B, T, H, C, scores, ld_scores,
                       kl_part, as_stream(stream));
}

int ggcn_dense_head(const float *pooled, int64_t ldp, const float *Wt, int64_t ldw, const float *bias, int B, int H, int C,
                    float *logits, int64_t ldl, const float *overlap_partials, int F_block, float *xy, ggcn_stream_t stream)
{
    return dense_head(pooled, ldp, Wt, ldw, bias, B, H, C, logits, ldl, overlap_partials, F_block, xy, as_stream(stream));
}

int ggcn_dense_head_signal(const float *pooled, int64_t ldp, const float *Wt, int64_t ldw, const float *bias, int B, int H, int C,
                           float *logits, int64_t ldl, const float *overlap_partials, int F_block, float *xy, uint32_t *signal,
                           ggcn_stream_t stream)
{
    if (!signal) return fail(GGCN_EINVAL, "ggcn_dense_head_signal: null signal words (ggcn_dense_head is the plain form)");
    return dense_head(pooled, ldp, Wt, ldw, bias, B, H, C, logits, ldl, overlap_partials, F_block, xy, as_stream(stream), signal);
}

size_t ggcn_overlap_workspace_bytes(int B) { return overlap_workspace_bytes(B); }

int ggcn_gate_overlap(const float *x1, const float *y1, int B, int F, float *xy, void *workspace,
                      ggcn_stream_t stream)
{
    return gate_overlap(x1, y1, B, F, xy, workspace, as_stream(stream));
}

}  // extern "C"
