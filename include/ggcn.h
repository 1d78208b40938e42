/* ggcn.h -- C ABI of libggcn_hip.so: the gated graph-convolution hot path of
 * laiviet/ed-gated-gcn as hand-written HIP kernels for MI355X (gfx950).
 *
 * The reference has no native code and no FFI: its boundary for this path is
 * the Python class GraphConvolution (models/gcn.py:9-45) and the ~20 lines of
 * BertAmir55.forward that wrap it (models/bert_amir5.py:621-640).  These entry
 * points are what a binding for that path calls; each one names the reference
 * lines it replaces.  INTEGRATION.md shows the reference-side binding (a ctypes
 * stub behind the unchanged nn.Module signature).
 *
 * Conventions
 *  - every pointer is a DEVICE pointer (hipMalloc'd or a torch CUDA/HIP tensor's
 *    data_ptr) unless named host_*; nothing is copied, owned or freed here;
 *  - every call only ENQUEUES work on `stream` (a hipStream_t passed as void*;
 *    NULL = the default stream); no call synchronises, allocates or frees, so
 *    every call may be captured in a hipGraph;
 *  - return 0 on success, a GGCN_E* code otherwise; ggcn_last_error() gives the
 *    message of the calling thread's last failure (never NULL);
 *  - feature matrices are row-major fp32 with an explicit leading dimension in
 *    ELEMENTS; a batch of B graphs of T nodes is the N = B*T rows b*T .. b*T+T-1;
 *  - batched CSR: rowptr int32[N+1], colidx int32[nnz] holding GLOBAL node ids
 *    (block-diagonal: every id of row i lies in i's own graph), vals fp32[nnz]
 *    or NULL for a binary adjacency (all ones);
 *  - row masks (graphs of at most GGCN_MASK_MAX_T = 256 nodes): rowmask uint32[N][W], W = ceil(T/32) words
 *    per node, bit j%32 of word j/32 of node b*T+i set iff adj[b,i,j] != 0 -- the 0/1 adjacency at one bit
 *    per entry (T <= 32: one word per node), consumed by ggcn_layer_fused / ggcn_block_fused.
 */
#ifndef GGCN_H
#define GGCN_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GGCN_ABI_VERSION 13
#define GGCN_MASK_MAX_T 256   /* largest graph the row-mask (one-launch) path takes */

typedef void *ggcn_stream_t;

enum ggcn_status {
    GGCN_OK = 0,
    GGCN_EINVAL = 1,       /* bad argument (null pointer, negative size, misaligned, ...) */
    GGCN_ELAUNCH = 2,      /* the HIP runtime refused the launch */
    GGCN_EUNSUPPORTED = 3  /* valid request this build has no kernel for */
};

/* dtype of a dense adjacency handed to ggcn_csr_from_dense (models/gcn.py:33
 * `adj.float()` accepts any real dtype). */
enum ggcn_adj_dtype {
    GGCN_ADJ_F32 = 0,
    GGCN_ADJ_U8 = 1,   /* torch.uint8 and torch.bool */
    GGCN_ADJ_I32 = 2,
    GGCN_ADJ_I64 = 3,
    GGCN_ADJ_F64 = 4,
    GGCN_ADJ_F16 = 5
};

/* bits of the `flags` word written by ggcn_csr_from_dense */
enum ggcn_csr_flags {
    GGCN_FLAG_WEIGHTED = 1 /* some non-zero entry differs from 1: the adjacency carries edge weights */
};

/* arithmetic of the dense linear (models/gcn.py:34 `torch.matmul(text, weight)`) */
enum ggcn_precision {
    GGCN_PREC_BF16X3 = 0,  /* fp32 operands split into bf16 hi+lo, 3 bf16 MFMAs per
                              product, fp32 accumulate: |err| ~ 2^-16 |x||w| per product */
    GGCN_PREC_FP32 = 1,    /* v_mfma_f32_32x32x2_f32: a k-ordered fp32 FMA chain, exact fp32 */
    GGCN_PREC_F16MX8 = 2,  /* operands split into fp16 hi + residual; hi.hi on the fp16 MFMA, both cross
                              terms in ONE block-scaled fp8 MFMA (v_mfma_scale_f32_32x32x64_f8f6f4):
                              2/3 of bf16x3's matrix-core time, |err| ~ 2^-15 |x||w| per product.
                              Needs |x|, |w| < 65504 (fp16 range; larger values saturate to inf) */
    GGCN_PREC_F16 = 3,     /* HALF-PRECISION FEATURES ONLY (ggcn_linear_h): plain fp16 MFMA on the fp16 image of W
                              (the fp16 fragments of the GGCN_PREC_F16MX8 pack), fp32 accumulate; |err| ~ 2^-12 |x||w|
                              per product -- below the rounding of the fp16 output it feeds */
    GGCN_PREC_F16MX6 = 4   /* the F16MX8 scheme with the correction products in fp6 (e2m3): gfx950 runs the block-scaled
                              MFMA twice as fast when both operands are 6-bit, so a product costs 3/4 of F16MX8's
                              matrix-core time.  e2m3 spans 6 binades, so both operands carry TRUE per-(row or column,
                              32 k) power-of-two scales (the activations' are computed in the kernel); |err| as F16MX8
                              (+10 %).  Same range rule.  Taken by ggcn_layer_fused / ggcn_block_fused for graphs of
                              <= 32 nodes with K % 32 == 0 and 16-byte aligned rows; its own ggcn_weight_pack image */
};

int ggcn_abi_version(void);
/* 1 when the library holds the experimental GGCN_PREC_F16MX6 form of the one-launch layer (csrc: make F16MX6=1); the
 * product build does not (measured slower than GGCN_PREC_F16MX8) and refuses that precision with GGCN_EUNSUPPORTED. */
int ggcn_has_f16mx6(void);
const char *ggcn_last_error(void);

/* ---- batched CSR from the reference's dense adjacency ---------------------
 * Replaces models/gcn.py:33 (`adj.float()`) and the structure half of :35/:41.
 * adj is [B,T,T] of `adj_dtype` with strides in ELEMENTS (the reference passes
 * a non-contiguous slice, models/bert_amir5.py:589).  Every non-zero adj[b,i,j]
 * becomes colidx entry b*T+j of row b*T+i (ascending j) with its value in vals
 * (vals may be NULL when the caller knows adj is 0/1).  `capacity` is the
 * number of entries colidx/vals can hold (B*T*T always suffices); entries
 * beyond it are dropped (rowptr still holds the true counts).
 * rowmask (NULL or uint32[B*T*ceil(T/32)], needs T <= GGCN_MASK_MAX_T) receives the row masks; flags (NULL or
 * one device int32) receives ggcn_csr_flags.
 * `workspace` needs ggcn_csr_workspace_bytes(B*T) bytes. */
size_t ggcn_csr_workspace_bytes(int64_t n_rows);
int ggcn_csr_from_dense(const void *adj, int adj_dtype, int B, int T,
                        int64_t stride_b, int64_t stride_r, int64_t stride_c,
                        int32_t *rowptr, int32_t *colidx, float *vals, int64_t capacity,
                        uint32_t *rowmask, int32_t *flags,
                        void *workspace, ggcn_stream_t stream);

/* Row masks straight from the dense adjacency (T <= GGCN_MASK_MAX_T): one pass, no CSR arrays.  This is all
 * ggcn_layer_fused needs, so the drop-in forward(text, adj) builds nothing else. */
int ggcn_rowmask_from_dense(const void *adj, int adj_dtype, int B, int T,
                            int64_t stride_b, int64_t stride_r, int64_t stride_c,
                            uint32_t *rowmask, int32_t *flags, ggcn_stream_t stream);

/* Transposed batched CSR (rows = source nodes): the backward pass applies A^T (train.py:120).  rowptr_t
 * int32[N+1], colidx_t int32[nnz], vals_t fp32[nnz] or NULL together with vals; rows of the result are sorted
 * by column; workspace: 4*B*T bytes.  One workgroup per graph (the adjacency is block-diagonal). */
int ggcn_csr_transpose(const int32_t *rowptr, const int32_t *colidx, const float *vals, int B, int T,
                       int32_t *rowptr_t, int32_t *colidx_t, float *vals_t, void *workspace,
                       ggcn_stream_t stream);

/* Per-graph operand blocks of the one-launch layer / block for graphs of <= 32 nodes, from the row masks
 * (one word per node): GGCN_GRAPH_OPS_BYTES per graph = the 0/1 adjacency laid out as the A operand of the
 * aggregation MFMA (2 x 1 KiB) + 1/(rowsum(adj)+1) of its 32 rows in accumulator order (models/gcn.py:35).
 * Built once per adjacency (it replaces, per column tile and wavefront, the expansion of the masks, 32 IEEE
 * divisions and 16 cross-lane moves per graph).  graph_ops: ggcn_graph_operands_bytes(B) bytes, 16-byte aligned. */
#define GGCN_GRAPH_OPS_BYTES 2176
/* The block's second layer applies the normalised adjacency twice (bert_amir5.py:626,639 without a non-linearity in between):
 * ggcn_graph_operands2 folds the two into ONE operand per graph, M2 = (D.A)^2 with D = diag(1 / (rowsum(A) + 1)) (gcn.py:35),
 * as the A operand of the aggregation MFMA in the launch's plane type (`plane`: 0 = bf16 pairs for GGCN_PREC_BF16X3, 1 = fp16
 * pairs for GGCN_PREC_F16MX8): hi and lo parts of M2 * 2^10 (the scale keeps the lo part out of fp16's subnormals; the
 * epilogue multiplies by 2^-10), two k-steps each, plus rowsum(D.A) per row (the factor of the `mid` bias):
 * GGCN_GRAPH_OPS2_BYTES per graph.  ggcn_block_fused reads it for its W12 column tiles. */
#define GGCN_GRAPH_OPS2_BYTES 4224
/* Graphs of 129..256 nodes (the eight-wavefront one-launch layer): the neighbour sums walk per-row EDGE LISTS.  ggcn_graph_edge_lists
 * makes them once per adjacency tensor from the row masks -- per graph GGCN_EDGE_LISTS_BYTES = the kernel's LDS image: per row up to
 * 16 source rows as byte offsets into its fp32 tile, the degree, 1/(rowsum+1) (models/gcn.py:35), a zero row -- and
 * ggcn_layer_fused takes the blocks in its graph_ops argument (NULL: every (graph, 256 columns) workgroup builds its lists itself,
 * as before; rows with more than 16 neighbours walk their mask words either way, so the row masks stay required). */
#define GGCN_EDGE_LISTS_BYTES 11264
size_t ggcn_graph_edge_lists_bytes(int B);
int ggcn_graph_edge_lists(const uint32_t *rowmask, int B, int T, void *lists, ggcn_stream_t stream);
size_t ggcn_graph_operands_bytes(int B);
int ggcn_graph_operands(const uint32_t *rowmask, int B, int T, void *graph_ops, ggcn_stream_t stream);
size_t ggcn_graph_operands2_bytes(int B);
int ggcn_graph_operands2(const uint32_t *rowmask, int B, int T, int plane, void *graph_ops2, ggcn_stream_t stream);

/* Row masks from an existing batched CSR (T <= GGCN_MASK_MAX_T). */
int ggcn_csr_rowmask(const int32_t *rowptr, const int32_t *colidx, int B, int T,
                     uint32_t *rowmask, ggcn_stream_t stream);

/* ---- dense linear ----------------------------------------------------------
 * Replaces models/gcn.py:34: Y[M,F] = X[M,K] . W[K,F]  (W is in x out, gcn.py:18).
 * GGCN_PREC_BF16X3 and GGCN_PREC_F16MX8 need `wpack`, the split image of W in MFMA
 * fragment order, made once per weight update by ggcn_weight_pack for THAT precision
 * (ggcn_weight_pack_bytes(K,F,precision) bytes; the two images differ); GGCN_PREC_FP32
 * reads W itself and ignores wpack.  transposed != 0 packs W^T instead: W is then
 * [F,K] row-major with ldw >= K, the image is that of the [K,F] matrix W^T (the dX
 * linear of the backward pass, train.py:120, uses it with the forward's weight). */
size_t ggcn_weight_pack_bytes(int K, int F, int precision);
int ggcn_weight_pack(const float *W, int64_t ldw, int K, int F, int precision, int transposed,
                     void *wpack, ggcn_stream_t stream);
int ggcn_linear(const float *X, int64_t ldx, const float *W, int64_t ldw, const void *wpack,
                float *Y, int64_t ldy, int64_t M, int K, int F, int precision,
                ggcn_stream_t stream);

/* ---- gated aggregation: one wavefront per destination node ----------------
 * Replaces models/gcn.py:35-45 and models/bert_amir5.py:627-640 in one pass:
 *   y[i,:]      = ( sum_e vals[e] * Hd[colidx[e],:] ) / ( sum_e vals[e] + 1 ) + bias
 *   out[i,:]    = y[i,:] * store_gate[g(i),:]           (store_gate NULL => 1)
 *   pool_a[g,:] = max over the T rows of graph g of y * pool_gate_a[g,:]
 *   pool_b[g,:] = likewise with pool_gate_b
 * Hd [N,F] is the linear's output; gates and pools are [B,F] contiguous.
 * out may be NULL (pooled outputs only); pool_x NULL disables that pool
 * (pool_gate_x NULL with pool_x set => gate of ones); bias may be NULL.
 * N = B*T.  Layer 1 of the block (bert_amir5.py:626-636): store_gate NULL,
 * pools (gate1,x1) and (gate2,y1).  Layer 2 (:639-640): store_gate = gate2,
 * pool (gate2,out). */
int ggcn_aggregate(const float *Hd, int64_t ldh,
                   const int32_t *rowptr, const int32_t *colidx, const float *vals,
                   const float *bias, int B, int T, int F,
                   const float *store_gate, const float *pool_gate_a, const float *pool_gate_b,
                   float *out, int64_t ldo, float *pool_a, float *pool_b,
                   ggcn_stream_t stream);

/* ---- backward pass of the layer (train.py:115-121 trains through gc1/gc2) ----------------
 * For Y = D.A.(X.W) + b with D = diag(1/(rowsum(A)+1)) and an upstream gradient dY [N,F]:
 *   dH = A^T.(D.dY)   ggcn_aggregate_t on the TRANSPOSED adjacency (CSR of adj^T: rows = source
 *                     nodes), src_scale = the 1/(rowsum+1) of the ORIGINAL rows from
 *                     ggcn_inv_denominators;  out[j] = sum_e vals_t[e]*src_scale[colidx_t[e]]*G[colidx_t[e]]
 *   dX = dH.W^T       ggcn_linear with the image made by ggcn_weight_pack(transposed = 1)
 *                     (packs the transpose of the stored matrix: the packed operand has
 *                     K = F_layer rows and F = K_layer columns)
 *   dW = X^T.dH       ggcn_dweight, split over the node rows, deterministic (fixed-order slab sum):
 *                     GGCN_PREC_FP32  exact fp32 MFMA on the rows as they lie (16-byte row loads when
 *                     aligned, element loads otherwise);
 *                     GGCN_PREC_BF16X3  X is transposed and dH packed once, then the forward's
 *                     bf16x3 main loop runs split-K (about 3x faster at config 2; needs
 *                     ~2 x 4*N*max(K,F) bytes of workspace).  f16mx8 is refused: gradients need
 *                     the fp32 exponent range.  (workspace: ggcn_dweight_workspace_bytes(N, K, F,
 *                     precision) bytes, 16-byte aligned)
 *   db = sum_rows dY  ggcn_gate_pool_backward's d_bsum (per graph) + ggcn_colsum over the graphs. */
/* Backward of the gate / max-pool epilogue (models/bert_amir5.py:627-640): from the stored
 * layer output `out` (= y*store_gate), the gates and the upstream gradients of out and of the
 * two pooled outputs, produce dY (gradient of the ungated layer output y) and the gate gradients:
 *   dY[t] = d_out[t]*sg + [t=argmax_a] d_pa*ga + [t=argmax_b] d_pb*gb
 *   d_sg = sum_t d_out*y,  d_ga = d_pa*y[argmax_a],  d_gb = d_pb*y[argmax_b]
 * Any of store_gate, gate_a/d_pa, gate_b/d_pb, d_out, d_sg, d_ga, d_gb may be NULL. */
int ggcn_gate_pool_backward(const float *out, int64_t ldo,
                            const float *store_gate, const float *gate_a, const float *gate_b,
                            const float *d_out, int64_t ldd, const float *d_pa, const float *d_pb,
                            int B, int T, int F, float *dY, int64_t ldy,
                            float *d_sg, float *d_ga, float *d_gb, float *d_bsum, ggcn_stream_t stream);
/* The same with the gates' dropout keep factors of ggcn_layer_fused_drop (same p, seed and streams as the forward launch):
 * sg, ga, gb become sg*k_store[t], ga*k_a[t], gb*k_b[t].  A token whose store factor is 0 (out[t] = 0) contributes y = 0:
 * exact when the pool gates in use share the store gate's stream or there is no store gate (the block's two layers). */
int ggcn_gate_pool_backward_drop(const float *out, int64_t ldo, const float *store_gate, const float *gate_a,
                                 const float *gate_b, const float *d_out, int64_t ldd, const float *d_pa,
                                 const float *d_pb, int B, int T, int F, float *dY, int64_t ldy, float *d_sg,
                                 float *d_ga, float *d_gb, float *d_bsum, float p, uint64_t seed, int stream_store,
                                 int stream_a, int stream_b, ggcn_stream_t stream);
/* ggcn_gate_pool_backward[_drop] followed by ggcn_aggregate_t in ONE launch, for graphs of T <= 32 nodes with a 0/1 adjacency
 * given as row masks (ggcn_rowmask_from_dense / _from_csr: one word per node): writes dH = A^T . D . dY (the gradient of
 * hidden = text . W, models/gcn.py:34,41 under train.py:120) straight away -- dY is consumed by nothing else and never
 * reaches memory.  p = 0: no gate dropout.  Needs F % 4 == 0, leading dimensions % 4 == 0, 16-byte aligned pointers
 * (GGCN_EUNSUPPORTED otherwise: take the two calls).  Same sums in the same order as the two calls. */
int ggcn_gate_pool_backward_agg(const float *out, int64_t ldo, const float *store_gate, const float *gate_a,
                                const float *gate_b, const float *d_out, int64_t ldd, const float *d_pa,
                                const float *d_pb, const uint32_t *rowmask, int B, int T, int F, float *dH, int64_t ldh,
                                float *d_sg, float *d_ga, float *d_gb, float *d_bsum, float p, uint64_t seed,
                                int stream_store, int stream_a, int stream_b, float *dh_amax, ggcn_stream_t stream);
/* dh_amax (NULL, or one device float the caller zeroed): receives max |dH| of the launch (an atomic maximum of bit patterns:
 * +inf when a gradient is not finite).  ggcn_linear_scaled reads it:
 *   Y[M,F] = X[M,K] . W   on the two-unit f16mx8 product (GGCN_PREC_F16MX8 image of W) for rows of ANY magnitude: every
 *   workgroup derives the same power of two s from *amax (|x| * s < 256), multiplies x by s before the split and the result by
 *   1 / s -- both exact -- so gradients far below fp16's range keep the product's ~2^-16 accuracy relative to the largest
 *   entry (values below amax * 2^-32 flush: they cannot matter to a sum the largest entries dominate).  This is the backward's
 *   dX = dH . W^T (train.py:120 through models/gcn.py:34) at 325-335 us instead of the three-product form's 420 at config 2.
 *   Needs K % 32 == 0, F % 4 == 0 and 16-byte aligned rows (GGCN_EUNSUPPORTED otherwise: ggcn_linear with GGCN_PREC_BF16X3).
 *   A non-finite amax leaves the data unscaled (the result is non-finite either way). */
int ggcn_linear_scaled(const float *X, int64_t ldx, const void *wpack, float *Y, int64_t ldy, int64_t M, int K, int F,
                       const float *amax, ggcn_stream_t stream);
/* The same two steps on the matrix cores (no gate dropout): dH_g = A_g^T . (D.dY_g) as one MFMA chain per (graph, 32 columns)
 * -- D.dY split into three bf16 planes (2^-25 of its scale), A^T an exact 0/1 operand -- instead of 1024 scalar bit tests per
 * thread.  graph_ops: the graph's ggcn_graph_operands blocks (their 1/(rowsum+1) table); graph_ops_t: ggcn_graph_operands blocks of
 * the TRANSPOSED row masks (ggcn_rowmask_transpose: bit t of word s = bit s of word t; once per adjacency tensor).  Same results as
 * ggcn_gate_pool_backward_agg up to the order of additions (gate gradients and bias sums: two partial sums per column instead of one
 * sequential sum; arg-max ties go to the smaller row in both).  Needs F % 4 == 0, ldh % 4 == 0, 16-byte aligned dH. */
int ggcn_rowmask_transpose(const uint32_t *rowmask, int B, int T, uint32_t *rowmask_t, ggcn_stream_t stream);
int ggcn_gate_pool_backward_mma(const float *out, int64_t ldo, const float *store_gate, const float *gate_a,
                                const float *gate_b, const float *d_out, int64_t ldd, const float *d_pa, const float *d_pb,
                                const void *graph_ops, const void *graph_ops_t, int B, int T, int F, float *dH, int64_t ldh,
                                float *d_sg, float *d_ga, float *d_gb, float *d_bsum, float *dh_amax, ggcn_stream_t stream);
/* d_bsum (NULL or [B,F]) receives sum_t dY per graph; db = sum_rows dY is then ggcn_colsum over its B rows.
 * ggcn_colsum: out[f] = sum_r X[r,f] for X [M, ld], deterministic (fixed-order slab sums);
 * workspace: ggcn_colsum_workspace_bytes(F) bytes. */
size_t ggcn_colsum_workspace_bytes(int F);
int ggcn_colsum(const float *X, int64_t ld, int64_t M, int F, float *out, void *workspace, ggcn_stream_t stream);
size_t ggcn_dweight_workspace_bytes(int64_t n_rows, int K, int F, int precision);
int ggcn_dweight(const float *X, int64_t ldx, const float *dH, int64_t ldg, int64_t n_rows, int K, int F,
                 float *dW, int64_t lddw, int precision, void *workspace, ggcn_stream_t stream);
int ggcn_inv_denominators(const int32_t *rowptr, const float *vals, int64_t n_rows, float *inv,
                          ggcn_stream_t stream);
int ggcn_aggregate_t(const float *G, int64_t ldg,
                     const int32_t *rowptr_t, const int32_t *colidx_t, const float *vals_t,
                     const float *src_scale, int B, int T, int F,
                     float *out, int64_t ldo, ggcn_stream_t stream);

/* ---- fp16 features (BASELINE configs[3]: 512-token graphs, hidden 1024) ------------------
 * Same operations with X / hidden / out stored as IEEE half and fp32 accumulation; weights
 * (wpack), bias, gates and pooled outputs stay fp32.  The reference cannot run half inputs
 * (models/gcn.py:33-34 raise a dtype mismatch); parity is against the fp32 reference
 * on fp16-rounded inputs.  An fp16 value splits exactly into two bf16 terms, so the linear uses
 * the same three-product scheme as GGCN_PREC_BF16X3. */
int ggcn_linear_h(const void *X, int64_t ldx, const void *wpack, void *Y, int64_t ldy,
                  int64_t M, int K, int F, int precision, ggcn_stream_t stream);
int ggcn_aggregate_h(const void *Hd, int64_t ldh,
                     const int32_t *rowptr, const int32_t *colidx, const float *vals,
                     const float *bias, int B, int T, int F,
                     const float *store_gate, const float *pool_gate_a, const float *pool_gate_b,
                     void *out, int64_t ldo, float *pool_a, float *pool_b,
                     ggcn_stream_t stream);

/* The two calls above as ONE launch for long graphs (T <= GGCN_LONG_MAX_T = 512) with fp16 features: models/gcn.py:34-45
 * (+ the gates and pools of models/bert_amir5.py:627-640) without `hidden` leaving the CU.  A workgroup owns a
 * graph and 128 output columns: plain fp16 MFMA (the arithmetic of GGCN_PREC_F16; wpack from
 * ggcn_weight_pack(..., GGCN_PREC_F16)), `hidden` rounded to fp16 into LDS exactly as ggcn_linear_h stores it, then the
 * CSR neighbour sums out of LDS.  Same arguments and results as ggcn_linear_h(GGCN_PREC_F16) + ggcn_aggregate_h
 * (rounding of the fp32 sums differs by summation order only).  Needs K % 64 == 0, F % 8 == 0, 16-byte aligned
 * X / out / bias / gates with ldx, ldo multiples of 8; anything else returns GGCN_EUNSUPPORTED (use the two calls). */
#define GGCN_LONG_MAX_T 512
int ggcn_layer_fused_h(const void *X, int64_t ldx, const void *wpack,
                       const int32_t *rowptr, const int32_t *colidx, const float *vals,
                       const float *bias, int B, int T, int K, int F,
                       const float *store_gate, const float *pool_gate_a, const float *pool_gate_b,
                       void *out, int64_t ldo, float *pool_a, float *pool_b,
                       ggcn_stream_t stream);

/* ---- one whole gated layer in one launch (graphs of <= GGCN_MASK_MAX_T = 256 nodes, binary adjacency) ----
 * Replaces models/gcn.py:34-45 + models/bert_amir5.py:627-640 without materialising
 * `hidden`: the linear's accumulator tile (one graph x 32 features) is multiplied by
 * the graph's 0/1 adjacency with a second MFMA, then divided, biased, gated, pooled and
 * stored.  Same outputs and argument meaning as ggcn_linear(GGCN_PREC_BF16X3) followed by
 * ggcn_aggregate; X is [B*T, K], wpack from ggcn_weight_pack(K, F, precision); T <= 32 reads graph_ops
 * (ggcn_graph_operands; rowmask may be NULL), T > 32 reads rowmask uint32[B*T][ceil(T/32)] (graph_ops may be
 * NULL); precision is GGCN_PREC_BF16X3 or GGCN_PREC_F16MX8.  T <= 32: one 32x32
 * accumulator tile is one graph.  32 < T <= 128: a graph takes a 64- or 128-row slot of a wavefront's tile and
 * its adjacency is applied as ceil(T/32)^2 blocks of 32x32 bits (LitBank: ORI_ML = 100, constant.py:227).
 * 128 < T <= 256 (ACE cased: ORI_ML = 231, constant.py:267): eight wavefronts per graph; the accumulators of a
 * 32-column tile go to an fp32 tile in LDS and the neighbour sums run over per-row edge lists made from the
 * row masks (exact fp32 sums).
 * The gate-diversity regulariser (models/bert_amir5.py:638) can ride along instead of taking
 * ggcn_gate_overlap's two launches: overlap_partial (NULL or float[B * ceil(F/64)]) receives, per graph
 * and 64-column group, sum_f pool_a[g,f]*pool_b[g,f] of THIS launch (layer 1: x1.y1); overlap_in /
 * overlap_out (both or neither) make this launch first reduce the partials an EARLIER launch on the
 * same stream wrote (same B, F) to *overlap_out = mean_b sum_f, in a fixed order (deterministic). */
int ggcn_layer_fused(const float *X, int64_t ldx, const void *wpack,
                     const uint32_t *rowmask, const void *graph_ops,
                     const float *bias, int B, int T, int K, int F,
                     const float *store_gate, const float *pool_gate_a, const float *pool_gate_b,
                     float *out, int64_t ldo, float *pool_a, float *pool_b,
                     float *overlap_partial, const float *overlap_in, float *overlap_out,
                     int precision, ggcn_stream_t stream);

/* The one-launch layer for graphs of 33..256 nodes with a bias added BEFORE the aggregation:
 *   y = D.A.(X.W + 1.bias_pre^T) + bias  =  D.A.X.W + rowsum(D.A).bias_pre + bias
 * -- the second half of the folded two-layer block when its input rows are already aggregated once: with Z = D.A.X
 * (ggcn_aggregate on the features), gc2(gc1(X)) = D.A.(Z.W12 + 1.bias_mid^T) + b2 (models/bert_amir5.py:626,639;
 * W12 = W1.W2, bias_mid = W2^T.b1), so an evaluation that needs only `out` (train.py:227) never multiplies by W1
 * (gated_block.py: the eval form for LitBank / ACE-cased lengths, constant.py:227,267).  Other arguments as in
 * ggcn_layer_fused (row masks required: T > 32; no regulariser partials). */
int ggcn_layer_fused_prebias(const float *X, int64_t ldx, const void *wpack, const uint32_t *rowmask,
                             const float *bias, const float *bias_pre, int B, int T, int K, int F,
                             const float *store_gate, const float *pool_gate_a, const float *pool_gate_b,
                             float *out, int64_t ldo, float *pool_a, float *pool_b, int precision, ggcn_stream_t stream);

/* The one-launch layer for a REAL-valued adjacency (models/gcn.py:33-41 take any `adj`: denom = rowsum(adj) + 1,
 * adj.hidden / denom; the reference's own graphs are 0/1, graph.py:66-74), graphs of <= 32 nodes:
 * ggcn_graph_operands_weighted turns the batched CSR with its weights (vals; NULL = all ones) into one operand block per
 * graph, M = D.A_w * 2^10 as hi / lo parts in the launch's plane type (`plane` as in ggcn_graph_operands2, same size:
 * ggcn_graph_operands2_bytes(B)); *flag (optional, device memory, zeroed by the caller) gets bit 0 when an entry does not
 * fit the plane type (|w / (rowsum + 1)| >= ~58 with fp16 planes, or a non-finite value: mixed-sign weights can do that) --
 * such an adjacency stays with ggcn_linear + ggcn_aggregate.  ggcn_layer_fused_weighted is ggcn_layer_fused on those blocks:
 * one split of `hidden` and 6 MFMAs per tile (hi.hi, hi.lo, lo.hi; the lo.lo term is 2^-22 of the result) where the 0/1
 * form needs 4, sums within 2^-21 of ggcn_aggregate's fp32 sums.  zero_mid: [F] zeros, 16-byte aligned (the kernel's
 * `mid` row, unused here).  Inference only (no gate dropout); other arguments as in ggcn_layer_fused. */
int ggcn_graph_operands_weighted(const int32_t *rowptr, const int32_t *colidx, const float *vals, int B, int T, int plane,
                                 void *graph_opsw, int32_t *flag, ggcn_stream_t stream);
int ggcn_layer_fused_weighted(const float *X, int64_t ldx, const void *wpack, const void *graph_opsw, const float *bias,
                              const float *zero_mid, int B, int T, int K, int F, const float *store_gate,
                              const float *pool_gate_a, const float *pool_gate_b, float *out, int64_t ldo, float *pool_a,
                              float *pool_b, float *overlap_partial, const float *overlap_in, float *overlap_out, int precision,
                              ggcn_stream_t stream);

/* ---- the whole gated block in one launch (graphs of <= 32 nodes, binary adjacency, inference) ----
 * Replaces models/bert_amir5.py:626-640 -- gc1, both gates, both max-pools, gc2, its gate and pool -- with
 * ONE launch that reads X once and never writes gcn1 unless asked to.  The reference feeds gc2 with the
 * UNGATED gcn1 and applies no non-linearity between the layers (models/gcn.py:19 declares a Tanh that
 * forward never calls; models/bert_amir5.py:626,639), so with D = diag(1/(rowsum(A)+1)):
 *     gcn1 = D.A.X.W1 + b1
 *     gcn2 = D.A.gcn1.W2 + b2 = D.A.( D.A.(X.W12) + bias_mid ) + b2,   W12 = W1.W2,  bias_mid = W2^T.b1
 * wpack1 / wpack12: ggcn_weight_pack images of W1 [K,F] and W12 [K,F] for `precision`; the caller makes
 * W12 and bias_mid once per weight update (ggcn_linear with GGCN_PREC_FP32: an exact fp32 product;
 * bias_mid is a vector of F zeros when gc1 has no bias).  bias1 / bias2 may be NULL.
 * Outputs: x1 = max_t gcn1*gate1, y1 = max_t gcn1*gate2 ([B,F], required), x_out = gate2*gcn2 [N, ld2]
 * (or NULL), pool_out = max_t x ([B,F] or NULL), gcn1 [N, ld1] only when non-NULL (nothing downstream
 * of the block reads it), overlap_partial (NULL or float[B*ceil(F/64)]) as in ggcn_layer_fused -- finish
 * it with ggcn_overlap_reduce.  Same flop count as two ggcn_layer_fused launches; X is read once and the
 * 4.N.F-byte write + read of gcn1 disappears.  Training keeps the two-launch path (autograd needs gcn1).
 * graph_ops: ggcn_graph_operands blocks (layer 1's aggregation); graph_ops2: ggcn_graph_operands2 blocks in the plane type
 * of `precision` (layer 2: both aggregations as one application of (D.A)^2, the `mid` bias times rowsum(D.A)). */
int ggcn_block_fused(const float *X, int64_t ldx, const void *wpack1, const void *wpack12,
                     const void *graph_ops, const void *graph_ops2, const float *bias1, const float *bias_mid, const float *bias2,
                     int B, int T, int K, int F, const float *gate1, const float *gate2,
                     float *gcn1, int64_t ld1, float *x_out, int64_t ld2,
                     float *x1, float *y1, float *pool_out, float *overlap_partial,
                     int precision, ggcn_stream_t stream);
/* ---- training-mode dropout of the gates inside the one-launch layer (graphs of <= 256 nodes) ------------------
 * models/bert_amir5.py:621-625 repeats each gate to [B,T,H] and THEN applies F.dropout: one Bernoulli draw per (token,
 * feature) and gate.  Here the keep factors k[t,f] in {0, 1/(1-p)} come from a counter-based hash of (seed, element)
 * (csrc/dropout_hash.h) evaluated in the layer's epilogue -- nothing of size [B,T,H] is materialised -- and again in
 * the backward pass.  Two independent streams exist per seed (1, 2); a gate slot names the stream that drops it (0 =
 * not dropped).  The block uses stream 1 for gate1 and stream 2 for gate2 in BOTH layers (the reference drops gate2
 * once and uses it at :631 and :639):  layer 1: (store 0, pool a 1, pool b 2);  layer 2: (store 2, pool a 2, pool b 0).
 *   out = y * store_gate * k_store,   pool_x = max_t (y * pool_gate_x * k_x)
 * precision GGCN_PREC_BF16X3 or GGCN_PREC_F16MX8; B*T*F < 2^32; rowmask / graph_ops as in ggcn_layer_fused (T <= 32 reads
 * graph_ops, larger graphs the row masks).  ggcn_dropout_mask writes k (rows x F floats) of one stream: what a test or a
 * host-side oracle multiplies the repeated gate by. */
int ggcn_layer_fused_drop(const float *X, int64_t ldx, const void *wpack, const uint32_t *rowmask, const void *graph_ops,
                          const float *bias, int B, int T, int K, int F,
                          const float *store_gate, const float *pool_gate_a, const float *pool_gate_b,
                          float *out, int64_t ldo, float *pool_a, float *pool_b, int precision,
                          float p, uint64_t seed, int stream_store, int stream_a, int stream_b, ggcn_stream_t stream);
int ggcn_dropout_mask(int64_t rows, int F, float p, uint64_t seed, int stream_id, float *mask, ggcn_stream_t stream);

/* models/bert_amir5.py:638 from the partials a ggcn_block_fused / ggcn_layer_fused launch left:
 * *xy = mean_b sum_f x1*y1, fixed summation order (deterministic), one small launch. */
int ggcn_overlap_reduce(const float *partials, int B, int F, float *xy, ggcn_stream_t stream);

/* ---- dense head on the block's pooled output (+ the regulariser's final sum) in one launch -----------
 * Replaces the share of models/bert_amir5.py:643 `logits = self.dense(cat[.., out])` that reads the block's output -- and,
 * when asked, models/bert_amir5.py:638's final sum, which otherwise is ggcn_overlap_reduce's own launch:
 *   logits[b,c] = (bias ? bias[c] : 0) + sum_k pooled[b,k] * Wt[k,c]      pooled [B, ldp], Wt [H, ldw] (nn.Linear's weight
 *   slice, transposed: ggcn_transpose), C <= 64, logits [B, ldl]; plain fp32 FMA chains in a fixed order per row, so a row's
 *   logits do not depend on the batch (or shard) it sits in -- what every rank all-gathers per step when the batch is sharded;
 *   overlap_partials / xy (both or neither): the float[B * ceil(F_block/64)] partials of a ggcn_block_fused /
 *   ggcn_layer_fused launch with B graphs and F_block columns -> *xy = mean_b sum (fixed order; equal to
 *   ggcn_overlap_reduce's result up to the order of additions). */
int ggcn_dense_head(const float *pooled, int64_t ldp, const float *Wt, int64_t ldw, const float *bias, int B, int H, int C,
                    float *logits, int64_t ldl, const float *overlap_partials, int F_block, float *xy, ggcn_stream_t stream);

/* The same launch, counting itself done in memory: signal = two zero-initialised uint32 words in device memory; signal[1]
 * becomes n once the n-th launch on these words has written (and fenced, system scope) all of its logits and xy -- another stream
 * gated by hipStreamWaitValue32(signal + 1, >= n) may then read them, with no event record on the launching stream (the sharded
 * step hands its logits to the collective's stream this way: ed-gated-gcn_amd/shard.py, bench.py --gather-mode flag).
 * signal[0] is the launch's own arrival counter (zero between launches).  One launch at a time per pair of words. */
int ggcn_dense_head_signal(const float *pooled, int64_t ldp, const float *Wt, int64_t ldw, const float *bias, int B, int H, int C,
                           float *logits, int64_t ldl, const float *overlap_partials, int F_block, float *xy, uint32_t *signal,
                           ggcn_stream_t stream);

/* ---- gate-diversity regulariser --------------------------------------------
 * Replaces models/bert_amir5.py:638: *xy = mean_b sum_f x1[b,f]*y1[b,f].
 * x1,y1 [B,F] contiguous, xy one device float.  Deterministic (fixed-order
 * tree reduction).  workspace: ggcn_overlap_workspace_bytes(B) bytes. */
size_t ggcn_overlap_workspace_bytes(int B);
int ggcn_gate_overlap(const float *x1, const float *y1, int B, int F, float *xy,
                      void *workspace, ggcn_stream_t stream);

/* ---- the gate MLPs in one launch ------------------------------------------
 * Replaces models/bert_amir5.py:562-571,621-622: gate_k = Sigmoid(Linear2(Sigmoid(Linear1(Sigmoid(aspect))))) for
 * both gates at once; aspect [B,H] (lda in elements) -> gate_a, gate_b [B,H] contiguous (the reference's
 * .repeat(1,T).view() to [B,T,H] is never made: the layer kernels take [B,H] gates).  w*t are the nn.Linear
 * weights TRANSPOSED to [in,out] contiguous (ggcn_transpose, once per weight update); biases [H] or NULL.
 * The second gate (w1t_b .. gate_b) may be all NULL.  Plain fp32 FMA chains. */
int ggcn_transpose(const float *W, int rows, int cols, int64_t ldw, float *Wt, ggcn_stream_t stream);
int ggcn_gate_mlp(const float *aspect, int64_t lda, int B, int H,
                  const float *w1t_a, const float *b1_a, const float *w2t_a, const float *b2_a, float *gate_a,
                  const float *w1t_b, const float *b1_b, const float *w2t_b, const float *b2_b, float *gate_b,
                  ggcn_stream_t stream);

/* ---- scores / kl head in one launch ----------------------------------------
 * Replaces models/bert_amir5.py:645-648:
 *     output_w = fc(cat[x, aspect repeated over t])       [B,T,C], fc.weight [C, 2H], fc.bias [C]
 *     scores   = sum_c logits[b,c] * output_w[b,t,c]      [B,T]
 *     kl       = mean_b sum_t softmax_t(scores) * softmax_t(dist)
 * without the [B,T,2H] concat or the [B,T,C] product: scores[b,t] = (Wx^T.logits_b).x_t + logits_b.(Wa.a_b + b).
 * X [B*T, ldx] is the block's gated layer-2 output; dist [B, ldd] float (the reference's dist_to_target.float());
 * kl_part [B] receives sum_t ... per sentence -- ggcn_overlap_reduce(kl_part, B, 1, kl) finishes the mean.
 * dist and kl_part may both be NULL (scores only). */
int ggcn_scores_head(const float *X, int64_t ldx, const float *aspect, int64_t lda,
                     const float *logits, int64_t ldl, const float *fc_weight, int64_t ldw, const float *fc_bias,
                     const float *dist, int64_t ldd, int B, int T, int H, int C,
                     float *scores, int64_t ld_scores, float *kl_part, ggcn_stream_t stream);

/* ---- range check for GGCN_PREC_F16MX8 (on demand, not on the forward path) ----------------
 * out[0] = max |x| over the finite entries of X [M,K] (fp32, or IEEE half when is_half != 0; ld in
 * elements), out[1] = 1.0f when some entry is NaN or infinite.  f16mx8 needs |x|, |w| < 65504 and keeps
 * its full accuracy for |x| <= 448; the caller reads the two floats back when it wants the verdict. */
int ggcn_absmax(const void *X, int is_half, int64_t ld, int64_t M, int K, float *out, ggcn_stream_t stream);

/* ---- sticky range flag of GGCN_PREC_F16MX8 (what makes it safe as a default) ----------------------
 * models/gcn.py:34 multiplies in fp32 and has no range limit; f16mx8 meets the 1e-4 parity gate only inside a window, and
 * says so when the data leaves it.  Every f16mx8 main loop (ggcn_linear, ggcn_layer_fused, ggcn_block_fused) keeps the
 * running maximum of the |x| it splits (one v_max3 per two values: < 1 % of the block) and ORs these bits into a sticky
 * per-device flag:
 *   GGCN_RANGE_OVERFLOW  a value reached fp16's largest finite value 65504 (inf included; a NaN input shows as NaN in the
 *                        output instead); ggcn_weight_pack sets it for such a weight.  Results are saturated.
 *   GGCN_RANGE_WINDOW    a value left |x| <= 448, the range in which the fp8 correction terms are unsaturated: from there
 *                        on the product has plain fp16 accuracy (2^-12 relative), outside the parity gate.
 *   GGCN_RANGE_HIDDEN    the one-launch layer / block of graphs of <= 32 nodes (fp16 aggregation planes) could not rule
 *                        out |hidden| >= 65504: max|x| * max_f sum_k |w[k,f]| (+ max |mid bias|) reached it (a sufficient
 *                        bound from the weight image's trailer, not a detection: NaN is possible).
 * Any bit means "re-run with GGCN_PREC_BF16X3" (full fp32 range).  This call ORs the flag into *flag (device memory,
 * 4 bytes, zeroed by the caller) in stream order and, with clear != 0, resets it; the caller copies the word to the host
 * whenever it likes.  (The experimental f16mx6 form sets OVERFLOW and HIDDEN; its block scales have no window.) */
enum ggcn_range_bits { GGCN_RANGE_OVERFLOW = 1, GGCN_RANGE_WINDOW = 2, GGCN_RANGE_HIDDEN = 4 };
int ggcn_range_flag(uint32_t *flag, int clear, ggcn_stream_t stream);

/* ---- test hook: known garbage in every CU's LDS ---------------------------------------------------
 * Fills the whole LDS (160 KiB) of every CU with the 32-bit `pattern`, in stream order: the next kernel on the stream
 * starts on LDS whose stale contents are known.  The hardware never clears LDS between kernels, so a kernel that reads
 * a location it did not write itself returns whatever an EARLIER kernel left there -- a result that depends on the
 * process's history (models/gcn.py:41 and bert_amir5.py:635-640 have no such state).  The parity tests run every
 * LDS-resident path behind several patterns (+inf, -inf, NaN / id 0xFFFF) and demand bit-identical outputs. */
int ggcn_debug_poison_lds(uint32_t pattern, ggcn_stream_t stream);

/* ---- box calibration (diagnostics for bench.py's `box` block; never on the product path) ---------
 * The block kernel runs at the board's power cap, so its time follows the clock a given device holds under an MFMA-dense
 * load (devices differ by > 10 %).  Two probes make a figure taken on one box readable on another:
 * ggcn_debug_mfma_calibrate launches n_wg workgroups (256 threads, two per CU) that issue, per wavefront and stage, the
 * matrix-pipe work of the f16mx8 main loop -- 16 v_mfma_f32_32x32x16_f16 + 8 v_mfma_scale_f32_32x32x64_f8f6f4 on random
 * register operands -- and nothing else (no memory or LDS traffic, no barrier).  n_wg = 6144, stages = 24 is exactly the
 * main-loop MFMA work of ggcn_block_fused at BASELINE config 2.  stamps (NULL or uint64[2 * n_wg]) receives per workgroup
 * {d(s_memtime) in shader cycles, d(s_memrealtime) in 10 ns ticks} around its loop: clock under load = cycles / ticks x 100 MHz.
 * ggcn_debug_block_fused_stamped is ggcn_block_fused through a diagnostic instantiation of the same kernel that takes the
 * same two stamps around ITS main loop (f16mx8, T = 32, B % 4 == 0, K % 32 == 0, 16-byte aligned rows only; stamps
 * uint64[2 * workgroups], workgroups = ceil(B/16) * ceil(F/256) * 8): same results, a few percent slower; no product launch
 * executes a stamp. */
int ggcn_debug_mfma_calibrate(int n_wg, int stages, uint64_t *stamps, float *sink, ggcn_stream_t stream);
/* Which kernel ggcn_block_fused takes for the whole block (all outputs, GGCN_PREC_F16MX8, 16-byte aligned operands) at this shape:
 * 8 = one eight-wavefront workgroup per (row block, 256-column slice) that stages the row block's X planes once for its W1 and its
 * W12 tiles (fused_block8.hip) -- batches that make >= 6 rounds of one workgroup per CU, or >= 3 whole rounds; 2-4 % less time in
 * steady state at the board's power cap, the same bits -- else 4 = two four-wavefront workgroups per CU (fused_layer.hip).
 * GGCN_BLOCK_FORM=4 in the environment (read per call) keeps the four-wavefront kernel. */
int ggcn_block_fused_form(int B, int T, int K, int F);

/* EXPERIMENT, called by nothing in the product (tools/block8_timing.py, one parity test): ggcn_block_fused's tiles in workgroups of
 * EIGHT wavefronts that share a row block's X planes between the W1 and the W12 column tiles of a 256-column slice (f16mx8, T <= 32,
 * K % 32 == 0, 16-byte rows, no gcn1).  Bit-identical results; measured slower. */
int ggcn_lab_block_fused8(const float *X, int64_t ldx, const void *wpack1, const void *wpack12, const void *graph_ops,
                          const void *graph_ops2, const float *bias1, const float *bias_mid, const float *bias2,
                          int B, int T, int K, int F, const float *gate1, const float *gate2, float *x_out, int64_t ld2,
                          float *x1, float *y1, float *pool_out, float *overlap_partial, uint64_t *stamps, ggcn_stream_t stream);
/* (stamps: NULL, or uint64[workgroups * 2 groups * 4]: s_memrealtime at workgroup start / main loop start / main loop end / end) */
int ggcn_debug_block_fused_stamped(const float *X, int64_t ldx, const void *wpack1, const void *wpack12,
                                   const void *graph_ops, const void *graph_ops2, const float *bias1, const float *bias_mid,
                                   const float *bias2, int B, int T, int K, int F, const float *gate1, const float *gate2,
                                   float *gcn1, int64_t ld1, float *x_out, int64_t ld2, float *x1, float *y1, float *pool_out,
                                   float *overlap_partial, int precision, uint64_t *stamps, ggcn_stream_t stream);

/* ---- sub-word -> word pooling (the step before the path) ---------------
 * Replaces models/bert_amir5.py:600 `x = torch.bmm(transform, x)`:
 *   Y[b,r,:] = sum_c A[b,r,c] * X[b,c,:]
 * A [B,R,C] float32 with ELEMENT strides (the reference passes the non-contiguous slice
 * inputs['transform'][:, :T, :L]; data_utils.py:749-766 puts 1/l on the l sub-word positions of
 * word r), C <= 2048.  X [B,C,D] and Y [B,R,D] float32: row c of batch b starts at
 * X + b*x_batch + c*ldx (elements), likewise Y.  Only the non-zeros of A are multiplied, in
 * ascending c.  The backward dX = A^T . dY is the same call with sa_r / sa_c and R / C swapped. */
int ggcn_subword_pool(const float *A, int64_t sa_b, int64_t sa_r, int64_t sa_c,
                      const float *X, int64_t x_batch, int64_t ldx,
                      float *Y, int64_t y_batch, int64_t ldy,
                      int B, int R, int C, int D, ggcn_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* GGCN_H */
