// The two small pieces of BertAmir55.forward that sit directly on either side of the gated block
// (SURVEY 8f ranks 1 and 4), each as ONE launch:
//
//   gate MLPs    models/bert_amir5.py:562-571,621-622   gate_k = Sigmoid(Linear(Sigmoid(Linear(Sigmoid(aspect)))))
//                for k = 1, 2: [B,H] -> two [B,H] gates (the reference then repeats them to [B,T,H]; here they
//                stay [B,H] and go straight into ggcn_block_fused / ggcn_layer_fused)
//   scores / kl  models/bert_amir5.py:645-648           output_w = fc(cat[x, aspect]); scores = sum_c logits*output_w;
//                kl = mean_b sum_t softmax_t(scores) * softmax_t(dist)
//
// scores needs no [B,T,C] intermediate: fc(cat[x_t, a]) = Wx.x_t + Wa.a + b, so
//     scores[b,t] = logits_b . (Wx.x_t + Wa.a_b + b) = (Wx^T.logits_b) . x_t + logits_b . (Wa.a_b + b)
// -- one H-vector and one scalar per sentence, then one dot product per token: x is read once (HBM-bound).
// Both kernels use plain fp32 FMA chains (exact fp32, no matrix cores: 0.1-0.3 GFLOP at the reference's
// batch of 256 sentences).
#include "common.h"

namespace ggcn {
namespace {

__device__ __forceinline__ float sigmoidf(float v) { return 1.0f / (1.0f + expf(-v)); }

// Wt[k][j] = W[j][k] for an nn.Linear weight W [out, in] (contiguous): once per weight update
__global__ __launch_bounds__(256) void transpose_kernel(const float *__restrict__ W, int rows, int cols, int64_t ldw,
                                                        float *__restrict__ Wt)
{
    __shared__ float tile[32][33];
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    for (int i = ty; i < 32; i += 8)
        if (r0 + i < rows && c0 + tx < cols) tile[i][tx] = W[(int64_t)(r0 + i) * ldw + c0 + tx];
    __syncthreads();
    for (int i = ty; i < 32; i += 8)
        if (c0 + i < cols && r0 + tx < rows) Wt[(int64_t)(c0 + i) * rows + r0 + tx] = tile[tx][i];
}

struct GateParams {
    const float *w1t, *b1, *w2t, *b2;   // transposed weights [in=H][out=H], biases [H]
    float *out;                         // [B,H]
};

constexpr int kGateRows = 8;   // sentences per workgroup

// grid (ceil(B / 8), 2 gates); dynamic LDS: 2 * 8 * H floats
__global__ __launch_bounds__(256) void gate_mlp_kernel(const float *__restrict__ aspect, int64_t lda, int B, int H,
                                                       GateParams ga, GateParams gb)
{
    extern __shared__ float lds_f[];
    float *s = lds_f, *h = lds_f + kGateRows * H;
    const GateParams gp = blockIdx.y == 0 ? ga : gb;
    const int r0 = blockIdx.x * kGateRows;
    const int nr = B - r0 < kGateRows ? B - r0 : kGateRows;
    for (int i = threadIdx.x; i < kGateRows * H; i += 256) {
        const int r = i / H, k = i - r * H;
        s[i] = r < nr ? sigmoidf(aspect[(int64_t)(r0 + r) * lda + k]) : 0.0f;       // the Sequential's leading Sigmoid
    }
    __syncthreads();
    for (int layer = 0; layer < 2; ++layer) {
        const float *wt = layer == 0 ? gp.w1t : gp.w2t, *bias = layer == 0 ? gp.b1 : gp.b2;
        const float *in = layer == 0 ? s : h;
        for (int j = threadIdx.x; j < H; j += 256) {
            float acc[kGateRows];
#pragma unroll
            for (int r = 0; r < kGateRows; ++r) acc[r] = 0.0f;
            for (int k = 0; k < H; ++k) {
                const float w = wt[(int64_t)k * H + j];        // lanes = consecutive j: one coalesced row piece per k
#pragma unroll
                for (int r = 0; r < kGateRows; ++r) acc[r] = fmaf(in[r * H + k], w, acc[r]);
            }
            const float bj = bias ? bias[j] : 0.0f;
            if (layer == 0) {
#pragma unroll
                for (int r = 0; r < kGateRows; ++r) h[r * H + j] = sigmoidf(acc[r] + bj);
            } else {
#pragma unroll
                for (int r = 0; r < kGateRows; ++r)
                    if (r < nr) gp.out[(int64_t)(r0 + r) * H + j] = sigmoidf(acc[r] + bj);
            }
        }
        __syncthreads();
    }
}

// one workgroup per sentence; dynamic LDS: (H + C + T + 8) floats
__global__ __launch_bounds__(256) void scores_head_kernel(const float *__restrict__ X, int64_t ldx,
                                                          const float *__restrict__ aspect, int64_t lda,
                                                          const float *__restrict__ logits, int64_t ldl,
                                                          const float *__restrict__ fcw, int64_t ldw,
                                                          const float *__restrict__ fcb, const float *__restrict__ dist,
                                                          int64_t ldd, int T, int H, int C, float *__restrict__ scores,
                                                          int64_t lds_, float *__restrict__ kl_part)
{
    extern __shared__ float lds_f[];
    float *v = lds_f, *u = v + H, *sc = u + C, *red = sc + T;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float *lg = logits + (int64_t)b * ldl, *a = aspect + (int64_t)b * lda;
    // v[h] = sum_c logits[c] * Wx[c][h]  (Wx = the first H columns of fc.weight [C, 2H])
    for (int hh = tid; hh < H; hh += 256) {
        float acc = 0.0f;
        for (int c = 0; c < C; ++c) acc = fmaf(lg[c], fcw[(int64_t)c * ldw + hh], acc);
        v[hh] = acc;
    }
    // u[c] = Wa[c,:] . a + b[c]   (wavefront per class, lanes over h)
    for (int c = wave; c < C; c += 4) {
        float acc = 0.0f;
        for (int hh = lane; hh < H; hh += 64) acc = fmaf(fcw[(int64_t)c * ldw + H + hh], a[hh], acc);
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d);
        if (lane == 0) u[c] = acc + (fcb ? fcb[c] : 0.0f);
    }
    __syncthreads();
    float c0 = 0.0f;
    for (int c = 0; c < C; ++c) c0 = fmaf(lg[c], u[c], c0);       // the same fixed order in every thread
    // scores[t] = v . x_t + c0   (wavefront per token, lanes over h: coalesced row reads)
    for (int t = wave; t < T; t += 4) {
        const float *xr = X + ((int64_t)b * T + t) * ldx;
        float acc = 0.0f;
        for (int hh = lane; hh < H; hh += 64) acc = fmaf(xr[hh], v[hh], acc);
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d);
        if (lane == 0) {
            sc[t] = acc + c0;
            scores[(int64_t)b * lds_ + t] = acc + c0;
        }
    }
    __syncthreads();
    if (!kl_part) return;
    // kl_b = sum_t softmax_t(scores) * softmax_t(dist): one wavefront, fixed order
    if (wave == 0) {
        const float *dr = dist + (int64_t)b * ldd;
        float ms = -INFINITY, md = -INFINITY;
        for (int t = lane; t < T; t += 64) { ms = fmaxf(ms, sc[t]); md = fmaxf(md, dr[t]); }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) { ms = fmaxf(ms, __shfl_xor(ms, d)); md = fmaxf(md, __shfl_xor(md, d)); }
        float zs = 0.0f, zd = 0.0f, cross = 0.0f;
        for (int t = lane; t < T; t += 64) {
            const float es = expf(sc[t] - ms), ed = expf(dr[t] - md);
            zs += es; zd += ed; cross = fmaf(es, ed, cross);
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) { zs += __shfl_xor(zs, d); zd += __shfl_xor(zd, d); cross += __shfl_xor(cross, d); }
        if (lane == 0) kl_part[b] = cross / (zs * zd);
    }
    (void)red;
}


// ---- dense head on the pooled output (bert_amir5.py:643: the share of `dense` that reads the block's `out`) ------------
// logits[b,c] = bias[c] + sum_k pooled[b,k] * Wt[k,c], C <= 64.  One workgroup of 16 wavefronts per 8 sentences: their pooled
// rows go to LDS, lane = class, wavefront w takes k-slice w (Wt rows are read coalesced and all at once), the slices
// meet in LDS in a fixed order: a row's logits do not depend on which batch (or shard) the row sits in.  The LAST workgroup
// of the launch, when the caller passes them, adds the regulariser's partials of a ggcn_block_fused launch (bert_amir5.py:638)
// in reduce_partials' order (fused_common.h) -- the one-workgroup ggcn_overlap_reduce launch rides along: at a 512-graph shard
// the two tail launches were 13 us of a 95 us step, this one is ~4.
// ggcn_dense_head_signal: the launch counts itself done in memory.  signal[0] collects the workgroups of this launch (every
// workgroup's results are fenced before it arrives; the last arriver clears it for the next launch), signal[1] counts finished
// launches -- a stream gated on it by hipStreamWaitValue32(>= n) may read the logits and xy of the n-th launch without an event
// record on the launching stream (the sharded step: shard.PooledGather, `flag` hand-off).
__device__ __forceinline__ void head_signal(unsigned *__restrict__ signal, int tid)
{
    if (!signal) return;   // kernel-uniform
    __syncthreads();       // every store of this workgroup has been issued
    if (tid == 0) {
        __threadfence_system();
        if (atomicAdd(signal, 1u) == gridDim.x - 1) {
            signal[0] = 0u;
            __threadfence_system();
            atomicAdd_system(signal + 1, 1u);
        }
    }
}

constexpr int kHeadRows = 8, kHeadThreads = 1024, kHeadWaves = kHeadThreads / 64;
static_assert(kHeadWaves == 2 * kHeadRows, "two wavefronts stage a pooled row");
__global__ __launch_bounds__(kHeadThreads) void dense_head_kernel(const float *__restrict__ pooled, int64_t ldp, const float *__restrict__ Wt,
                                                                  int64_t ldw, const float *__restrict__ bias, int B, int H, int C,
                                                                  float *__restrict__ logits, int64_t ldl, const float *__restrict__ part,
                                                                  int n_part, float *__restrict__ xy, int head_blocks, unsigned *__restrict__ signal)
{
    extern __shared__ __attribute__((aligned(16))) float head_lds[];   // [kHeadRows][H] pooled rows, then [kHeadWaves][kHeadRows][64] partial sums
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if ((int)blockIdx.x >= head_blocks) {   // the regulariser's final sum (one workgroup; fixed order: deterministic)
        float sdot = 0.0f;
        for (int idx = tid; idx < n_part; idx += kHeadThreads) sdot += part[idx];
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) sdot += __shfl_xor(sdot, d);
        if (lane == 0) head_lds[wave] = sdot;
        __syncthreads();
        if (tid == 0) {
            float t = 0.0f;
#pragma unroll
            for (int w = 0; w < kHeadWaves; ++w) t += head_lds[w];
            *xy = t / (float)B;
        }
        head_signal(signal, tid);
        return;
    }
    const int b0 = blockIdx.x * kHeadRows;
    float *rows = head_lds, *red = head_lds + kHeadRows * H;
    {   // the workgroup's pooled rows, read once and coalesced: wavefront (r, half) brings half of row r (no division by H)
        const int r = wave >> 1, hb = (H + 1) >> 1, kb = (wave & 1) * hb, ke = kb + hb < H ? kb + hb : H;
        const bool live = b0 + r < B;
        const float *src = pooled + (int64_t)(live ? b0 + r : 0) * ldp;
        if ((hb & 3) == 0 && (ldp & 3) == 0 && (reinterpret_cast<uintptr_t>(pooled) & 15u) == 0) {
            for (int k = kb + 4 * lane; k < ke; k += 256) {
                const float4 v = *reinterpret_cast<const float4 *>(src + k);
                *reinterpret_cast<float4 *>(rows + r * H + k) = live ? v : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            }
        } else {
            for (int k = kb + lane; k < ke; k += 64) rows[r * H + k] = live ? src[k] : 0.0f;
        }
    }
    __syncthreads();
    // lane = class, wavefront w = k-slice w of 16: every Wt row of the slice is one coalesced load, all of them independent
    const int kq = (H + kHeadWaves - 1) / kHeadWaves, k0 = wave * kq, k1 = k0 + kq < H ? k0 + kq : H;
    const int c = lane < C ? lane : 0;
    float acc[kHeadRows] = {};
    if ((H & 3) == 0 && (kq & 3) == 0) {   // four k per step: one 16-byte LDS broadcast read per row instead of four 4-byte ones
#pragma unroll 4
        for (int k = k0; k < k1; k += 4) {
            float w[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) w[q] = Wt[(int64_t)(k + q) * ldw + c];
#pragma unroll
            for (int r = 0; r < kHeadRows; ++r) {
                const float4 xv = *reinterpret_cast<const float4 *>(rows + r * H + k);
                acc[r] = fmaf(xv.x, w[0], acc[r]); acc[r] = fmaf(xv.y, w[1], acc[r]);     // (k ascending: the order of the scalar loop)
                acc[r] = fmaf(xv.z, w[2], acc[r]); acc[r] = fmaf(xv.w, w[3], acc[r]);
            }
        }
    } else {
#pragma unroll 8
        for (int k = k0; k < k1; ++k) {
            const float w = Wt[(int64_t)k * ldw + c];
#pragma unroll
            for (int r = 0; r < kHeadRows; ++r) acc[r] = fmaf(rows[r * H + k], w, acc[r]);   // (an LDS broadcast read)
        }
    }
#pragma unroll
    for (int r = 0; r < kHeadRows; ++r) red[(wave * kHeadRows + r) * 64 + lane] = acc[r];
    __syncthreads();
    if (wave < kHeadRows && lane < C && b0 + wave < B) {   // wavefront r finishes row r: the 16 slices in ascending order
        float t = 0.0f;
#pragma unroll
        for (int w = 0; w < kHeadWaves; ++w) t += red[(w * kHeadRows + wave) * 64 + lane];
        logits[(int64_t)(b0 + wave) * ldl + lane] = t + (bias ? bias[lane] : 0.0f);
    }
    head_signal(signal, tid);
}

}  // namespace

int transpose_f32(const float *W, int rows, int cols, int64_t ldw, float *Wt, hipStream_t st)
{
    if (!W || !Wt) return fail(GGCN_EINVAL, "ggcn_transpose: null pointer");
    if (rows <= 0 || cols <= 0 || ldw < cols) return fail(GGCN_EINVAL, "ggcn_transpose: rows=%d cols=%d ldw=%lld", rows, cols, (long long)ldw);
    hipLaunchKernelGGL(transpose_kernel, dim3((unsigned)((cols + 31) / 32), (unsigned)((rows + 31) / 32)), dim3(256), 0, st,
                       W, rows, cols, ldw, Wt);
    return check_launch("ggcn_transpose");
}

int gate_mlp(const float *aspect, int64_t lda, int B, int H, const float *w1t_a, const float *b1_a, const float *w2t_a,
             const float *b2_a, float *gate_a, const float *w1t_b, const float *b1_b, const float *w2t_b,
             const float *b2_b, float *gate_b, hipStream_t st)
{
    if (!aspect || !w1t_a || !w2t_a || !gate_a) return fail(GGCN_EINVAL, "ggcn_gate_mlp: null pointer");
    if ((gate_b != nullptr) != (w1t_b != nullptr && w2t_b != nullptr))
        return fail(GGCN_EINVAL, "ggcn_gate_mlp: the second gate needs both of its weights and its output (or none)");
    if (B <= 0 || H <= 0 || lda < H) return fail(GGCN_EINVAL, "ggcn_gate_mlp: B=%d H=%d lda=%lld", B, H, (long long)lda);
    const size_t lds = (size_t)2 * kGateRows * H * sizeof(float);
    if (lds > 64 * 1024) return fail(GGCN_EUNSUPPORTED, "ggcn_gate_mlp: H=%d needs more than 64 KiB of LDS", H);
    GateParams ga{w1t_a, b1_a, w2t_a, b2_a, gate_a}, gb{w1t_b, b1_b, w2t_b, b2_b, gate_b};
    hipLaunchKernelGGL(gate_mlp_kernel, dim3((unsigned)((B + kGateRows - 1) / kGateRows), gate_b ? 2u : 1u), dim3(256), lds, st,
                       aspect, lda, B, H, ga, gb);
    return check_launch("ggcn_gate_mlp");
}

int scores_head(const float *X, int64_t ldx, const float *aspect, int64_t lda, const float *logits, int64_t ldl,
                const float *fcw, int64_t ldw, const float *fcb, const float *dist, int64_t ldd, int B, int T, int H,
                int C, float *scores, int64_t lds_, float *kl_part, hipStream_t st)
{
    if (!X || !aspect || !logits || !fcw || !scores) return fail(GGCN_EINVAL, "ggcn_scores_head: null pointer");
    if ((kl_part != nullptr) != (dist != nullptr)) return fail(GGCN_EINVAL, "ggcn_scores_head: dist and kl_part go together");
    if (B <= 0 || T <= 0 || H <= 0 || C <= 0) return fail(GGCN_EINVAL, "ggcn_scores_head: B=%d T=%d H=%d C=%d", B, T, H, C);
    if (ldx < H || lda < H || ldl < C || ldw < 2 * H || lds_ < T || (dist && ldd < T))
        return fail(GGCN_EINVAL, "ggcn_scores_head: leading dimension too small");
    const size_t lds = (size_t)(H + C + T + 8) * sizeof(float);
    if (lds > 64 * 1024) return fail(GGCN_EUNSUPPORTED, "ggcn_scores_head: H + C + T = %d needs more than 64 KiB of LDS", H + C + T);
    hipLaunchKernelGGL(scores_head_kernel, dim3((unsigned)B), dim3(256), lds, st, X, ldx, aspect, lda, logits, ldl, fcw, ldw,
                       fcb, dist, ldd, T, H, C, scores, lds_, kl_part);
    return check_launch("ggcn_scores_head");
}

int dense_head(const float *pooled, int64_t ldp, const float *Wt, int64_t ldw, const float *bias, int B, int H, int C,
               float *logits, int64_t ldl, const float *partials, int F_block, float *xy, hipStream_t st, unsigned *signal)
{
    if (!pooled || !Wt || !logits) return fail(GGCN_EINVAL, "ggcn_dense_head: null pointer");
    if ((partials != nullptr) != (xy != nullptr)) return fail(GGCN_EINVAL, "ggcn_dense_head: overlap_partials and xy go together");
    if (B <= 0 || H <= 0 || C <= 0 || (partials && F_block <= 0)) return fail(GGCN_EINVAL, "ggcn_dense_head: B=%d H=%d C=%d F_block=%d", B, H, C, F_block);
    if (C > 64) return fail(GGCN_EUNSUPPORTED, "ggcn_dense_head: C=%d classes (one lane per class: at most 64)", C);
    if (ldp < H || ldw < C || ldl < C) return fail(GGCN_EINVAL, "ggcn_dense_head: leading dimension too small");
    const int head_blocks = (B + kHeadRows - 1) / kHeadRows;
    const size_t lds = ((size_t)kHeadRows * H + (size_t)kHeadWaves * kHeadRows * 64) * sizeof(float);
    if (lds > 64 * 1024) return fail(GGCN_EUNSUPPORTED, "ggcn_dense_head: H=%d needs more than 64 KiB of LDS", H);
    hipLaunchKernelGGL(dense_head_kernel, dim3((unsigned)(head_blocks + (partials ? 1 : 0))), dim3(kHeadThreads), lds, st, pooled, ldp, Wt, ldw, bias,
                       B, H, C, logits, ldl, partials, partials ? B * ((F_block + 63) / 64) : 0, xy, head_blocks, signal);
    return check_launch("ggcn_dense_head");
}

}  // namespace ggcn
