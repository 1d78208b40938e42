// Dense adjacency [B,T,T] -> batched (block-diagonal) CSR, entirely on the device.
//
// Replaces models/gcn.py:33 (`adj = adj.float()`) and supplies the structure
// that :35 (`sum(adj, dim=2) + 1`) and :41 (`matmul(adj, hidden)`) consume.
// The reference ships the adjacency dense and padded with identity rows
// (graph.py:66-74, data_utils.py:376,394); padding rows therefore
// keep exactly their self loop here -- they are NOT skipped.
//
// Three launches, no host round trip:
//   1. count   one wavefront per row: 64 columns per step, __ballot + popcount
//   2. scan    exclusive prefix sum of the counts, in place in rowptr
//              (1024 rows per workgroup, workgroup totals scanned by one workgroup)
//   3. fill    same walk as 1; lane position = rowptr[row] + popcount(ballot below lane)
// HBM traffic = one read of adj per walk (2 walks) + O(N + nnz) writes.
#include "common.h"

#include <hip/hip_fp16.h>

namespace ggcn {
namespace {

constexpr int kScanBlock = 256;
constexpr int kScanItems = 4;
constexpr int kScanTile = kScanBlock * kScanItems;  // rows per scan workgroup

template <typename A>
__device__ __forceinline__ float load_adj(const A *p) { return static_cast<float>(*p); }
template <>
__device__ __forceinline__ float load_adj<__half>(const __half *p) { return __half2float(*p); }

template <typename A, bool FILL>
__global__ __launch_bounds__(256) void csr_walk_rows(
    const A *__restrict__ adj, int64_t n_rows, int T, int64_t sb, int64_t sr, int64_t sc,
    int32_t *__restrict__ rowptr, int32_t *__restrict__ colidx, float *__restrict__ vals,
    int64_t capacity, uint32_t *__restrict__ rowmask, int32_t *__restrict__ flags)
{
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n_rows) return;  // whole wave leaves together: row is wave-uniform
    const int64_t b = row / T;
    const int i = (int)(row - b * T);
    const A *r = adj + b * sb + (int64_t)i * sr;
    const int64_t base = FILL ? (int64_t)rowptr[row] : 0;
    int cnt = 0;
    bool weighted = false;
    for (int j0 = 0; j0 < T; j0 += kWave) {
        const int j = j0 + lane;
        const float v = (j < T) ? load_adj<A>(r + (int64_t)j * sc) : 0.0f;
        const bool nz = (v != 0.0f);
        const unsigned long long m = __ballot(nz);
        if (FILL) {
            weighted |= nz && (v != 1.0f);
            // the ballot IS 64 bits of the row's 0/1 adjacency (bit j = edge i<-j): ceil(T/32) words per row
            if (rowmask && lane == 0) {
                const int W = (T + 31) >> 5;
                rowmask[row * W + (j0 >> 5)] = (uint32_t)m;
                if (j0 + 32 < T) rowmask[row * W + (j0 >> 5) + 1] = (uint32_t)(m >> 32);
            }
        }
        if (FILL && nz) {
            const int64_t pos = base + cnt + __popcll(m & ((1ull << lane) - 1ull));
            if (pos < capacity) {
                colidx[pos] = (int32_t)(b * T + j);
                if (vals) vals[pos] = v;
            }
        }
        cnt += __popcll(m);
    }
    if (!FILL && lane == 0) rowptr[row] = cnt;
    if (FILL && flags && __any(weighted) && lane == 0) atomicOr(flags, GGCN_FLAG_WEIGHTED);
}

// Exclusive scan of one value per thread across a 256-thread workgroup; returns the
// exclusive prefix and (through *total) the workgroup sum.
__device__ __forceinline__ int block_exclusive_scan(int v, int *total, int *lds /*>=4 ints*/)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int inc = v;
#pragma unroll
    for (int d = 1; d < kWave; d <<= 1) {
        const int o = __shfl_up(inc, d);
        if (lane >= d) inc += o;
    }
    if (lane == kWave - 1) lds[wave] = inc;
    __syncthreads();
    int wave_off = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < kScanBlock / kWave; ++w) {
        const int s = lds[w];
        if (w < wave) wave_off += s;
        tot += s;
    }
    __syncthreads();
    *total = tot;
    return wave_off + inc - v;
}

__global__ __launch_bounds__(kScanBlock) void scan_tile_totals(const int32_t *__restrict__ counts,
                                                               int64_t n, int32_t *__restrict__ tile_sum)
{
    __shared__ int lds[4];
    const int64_t base = (int64_t)blockIdx.x * kScanTile + (int64_t)threadIdx.x * kScanItems;
    int v = 0;
#pragma unroll
    for (int k = 0; k < kScanItems; ++k)
        if (base + k < n) v += counts[base + k];
    int total;
    (void)block_exclusive_scan(v, &total, lds);
    if (threadIdx.x == 0) tile_sum[blockIdx.x] = total;
}

// One workgroup: exclusive scan of tile_sum[0..n_tiles) in place, 256 per step with a carry.
__global__ __launch_bounds__(kScanBlock) void scan_tile_offsets(int32_t *__restrict__ tile_sum, int n_tiles)
{
    __shared__ int lds[4];
    int carry = 0;
    for (int t0 = 0; t0 < n_tiles; t0 += kScanBlock) {
        const int t = t0 + threadIdx.x;
        const int v = (t < n_tiles) ? tile_sum[t] : 0;
        int total;
        const int ex = block_exclusive_scan(v, &total, lds);
        if (t < n_tiles) tile_sum[t] = carry + ex;
        carry += total;
    }
}

__global__ __launch_bounds__(kScanBlock) void scan_finish(int32_t *__restrict__ rowptr, int64_t n,
                                                          const int32_t *__restrict__ tile_off)
{
    __shared__ int lds[4];
    const int64_t base = (int64_t)blockIdx.x * kScanTile + (int64_t)threadIdx.x * kScanItems;
    int c[kScanItems];
    int v = 0;
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {
        c[k] = (base + k < n) ? rowptr[base + k] : 0;
        v += c[k];
    }
    int total;
    int run = block_exclusive_scan(v, &total, lds) + tile_off[blockIdx.x];
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {
        if (base + k < n) rowptr[base + k] = run;
        run += c[k];
        if (base + k == n - 1) rowptr[n] = run;  // the grand total closes the array
    }
}

// Row masks only (T <= 32): one pass over adj, two rows per wavefront (lanes 0-31 / 32-63), the
// ballot halves ARE the masks.  This is all the fused layer kernel needs from a dense adjacency;
// the CSR arrays are built only when something asks for them.
template <typename A>
__global__ __launch_bounds__(256) void rowmask_rows(const A *__restrict__ adj, int64_t n_rows, int T,
                                                    int64_t sb, int64_t sr, int64_t sc,
                                                    uint32_t *__restrict__ rowmask, int32_t *__restrict__ flags)
{
    const int lane = threadIdx.x & 63;
    const int64_t row = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + (lane >> 5);
    const int j = lane & 31;
    float v = 0.0f;
    if (row < n_rows && j < T) {
        const int64_t b = row / T;
        const int i = (int)(row - b * T);
        v = load_adj<A>(adj + b * sb + (int64_t)i * sr + (int64_t)j * sc);
    }
    const unsigned long long m = __ballot(v != 0.0f);
    if (j == 0 && row < n_rows) rowmask[row] = (uint32_t)(m >> (lane & 32));
    if (flags && __any(v != 0.0f && v != 1.0f) && lane == 0) atomicOr(flags, GGCN_FLAG_WEIGHTED);
}

// Row masks for 32 < T <= 128: one wavefront per row, 64 columns per ballot, ceil(T/32) words per row.
template <typename A>
__global__ __launch_bounds__(256) void rowmask_rows_wide(const A *__restrict__ adj, int64_t n_rows, int T,
                                                         int64_t sb, int64_t sr, int64_t sc,
                                                         uint32_t *__restrict__ rowmask, int32_t *__restrict__ flags)
{
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n_rows) return;
    const int64_t b = row / T;
    const int i = (int)(row - b * T);
    const A *r = adj + b * sb + (int64_t)i * sr;
    const int W = (T + 31) >> 5;
    bool weighted = false;
    for (int j0 = 0; j0 < T; j0 += kWave) {
        const int j = j0 + lane;
        const float v = (j < T) ? load_adj<A>(r + (int64_t)j * sc) : 0.0f;
        weighted |= (v != 0.0f) && (v != 1.0f);
        const unsigned long long m = __ballot(v != 0.0f);
        if (lane == 0) {
            rowmask[row * W + (j0 >> 5)] = (uint32_t)m;
            if (j0 + 32 < T) rowmask[row * W + (j0 >> 5) + 1] = (uint32_t)(m >> 32);
        }
    }
    if (flags && __any(weighted) && lane == 0) atomicOr(flags, GGCN_FLAG_WEIGHTED);
}

template <typename A>
int run_mask(const void *adj, int B, int T, int64_t sb, int64_t sr, int64_t sc, uint32_t *rowmask,
             int32_t *flags, hipStream_t st)
{
    const int64_t n = (int64_t)B * T;
    if (flags) (void)hipMemsetAsync(flags, 0, sizeof(int32_t), st);
    if (T <= 32)
        hipLaunchKernelGGL((rowmask_rows<A>), dim3((unsigned)((n + 7) / 8)), dim3(256), 0, st,
                           static_cast<const A *>(adj), n, T, sb, sr, sc, rowmask, flags);
    else
        hipLaunchKernelGGL((rowmask_rows_wide<A>), dim3((unsigned)((n + 3) / 4)), dim3(256), 0, st,
                           static_cast<const A *>(adj), n, T, sb, sr, sc, rowmask, flags);
    return check_launch("ggcn_rowmask_from_dense");
}

template <typename A>
int run(const void *adj, int B, int T, int64_t sb, int64_t sr, int64_t sc, int32_t *rowptr,
        int32_t *colidx, float *vals, int64_t capacity, uint32_t *rowmask, int32_t *flags,
        void *workspace, hipStream_t st)
{
    const int64_t n = (int64_t)B * T;
    const A *a = static_cast<const A *>(adj);
    const unsigned row_blocks = (unsigned)((n + 3) / 4);
    const unsigned tiles = (unsigned)((n + kScanTile - 1) / kScanTile);
    int32_t *tile_sum = static_cast<int32_t *>(workspace);

    hipLaunchKernelGGL((csr_walk_rows<A, false>), dim3(row_blocks), dim3(256), 0, st, a, n, T, sb, sr,
                       sc, rowptr, nullptr, nullptr, (int64_t)0, nullptr, nullptr);
    if (flags) (void)hipMemsetAsync(flags, 0, sizeof(int32_t), st);
    hipLaunchKernelGGL(scan_tile_totals, dim3(tiles), dim3(kScanBlock), 0, st, rowptr, n, tile_sum);
    hipLaunchKernelGGL(scan_tile_offsets, dim3(1), dim3(kScanBlock), 0, st, tile_sum, (int)tiles);
    hipLaunchKernelGGL(scan_finish, dim3(tiles), dim3(kScanBlock), 0, st, rowptr, n, tile_sum);
    hipLaunchKernelGGL((csr_walk_rows<A, true>), dim3(row_blocks), dim3(256), 0, st, a, n, T, sb, sr,
                       sc, rowptr, colidx, vals, capacity, rowmask, flags);
    return check_launch("ggcn_csr_from_dense");
}

// Transposed batched CSR (the backward pass applies A^T, train.py:120).  The adjacency is block-diagonal with
// blocks of T rows, so graph g's entries occupy the SAME range [rowptr[gT], rowptr[(g+1)T]) in both CSRs: one
// workgroup per graph, no global scan.  Thread j (a column of A = a row of A^T) counts its entries, a serial
// scan over the T counts gives the row pointers, then it walks the rows i in ascending order and copies its
// entries: the transposed rows come out sorted by column, deterministically.  O(T * nnz_g) per graph.
__global__ __launch_bounds__(256) void csr_transpose_kernel(const int32_t *__restrict__ rowptr,
                                                            const int32_t *__restrict__ colidx,
                                                            const float *__restrict__ vals, int T,
                                                            int32_t *__restrict__ rowptr_t, int32_t *__restrict__ colidx_t,
                                                            float *__restrict__ vals_t, int32_t *__restrict__ scratch)
{
    const int g = blockIdx.x;
    const int64_t base = (int64_t)g * T;
    const int e0 = rowptr[base], e1 = rowptr[base + T];
    int32_t *cnt = scratch + base;          // T counters of this graph (global scratch: T is not bounded by LDS)
    for (int j = threadIdx.x; j < T; j += 256) {
        int c = 0;
        for (int e = e0; e < e1; ++e) c += (colidx[e] - base == j);
        cnt[j] = c;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int run = e0;
        for (int j = 0; j < T; ++j) {
            const int c = cnt[j];
            rowptr_t[base + j] = run;
            cnt[j] = run;
            run += c;
        }
        if (g == (int)gridDim.x - 1) rowptr_t[base + T] = e1;
    }
    __syncthreads();
    for (int j = threadIdx.x; j < T; j += 256) {
        int pos = cnt[j];
        for (int i = 0; i < T; ++i)
            for (int e = rowptr[base + i]; e < rowptr[base + i + 1]; ++e)
                if (colidx[e] - base == j) {
                    colidx_t[pos] = (int32_t)(base + i);
                    if (vals_t) vals_t[pos] = vals ? vals[e] : 1.0f;
                    ++pos;
                }
    }
}

}  // namespace

int csr_transpose(const int32_t *rowptr, const int32_t *colidx, const float *vals, int B, int T, int32_t *rowptr_t,
                  int32_t *colidx_t, float *vals_t, void *workspace, hipStream_t st)
{
    if (!rowptr || !colidx || !rowptr_t || !colidx_t || !workspace) return fail(GGCN_EINVAL, "ggcn_csr_transpose: null pointer");
    if (B <= 0 || T <= 0) return fail(GGCN_EINVAL, "ggcn_csr_transpose: B=%d T=%d must be positive", B, T);
    if ((vals != nullptr) != (vals_t != nullptr))
        return fail(GGCN_EINVAL, "ggcn_csr_transpose: vals and vals_t go together (both NULL for a 0/1 adjacency)");
    hipLaunchKernelGGL(csr_transpose_kernel, dim3((unsigned)B), dim3(256), 0, st, rowptr, colidx, vals, T, rowptr_t,
                       colidx_t, vals_t, static_cast<int32_t *>(workspace));
    return check_launch("ggcn_csr_transpose");
}

size_t csr_workspace_bytes(int64_t n_rows)
{
    if (n_rows < 0) return 0;
    const int64_t tiles = (n_rows + kScanTile - 1) / kScanTile;
    return (size_t)(tiles > 0 ? tiles : 1) * sizeof(int32_t);
}

int rowmask_from_dense(const void *adj, int adj_dtype, int B, int T, int64_t sb, int64_t sr, int64_t sc,
                       uint32_t *rowmask, int32_t *flags, hipStream_t st)
{
    if (!adj || !rowmask) return fail(GGCN_EINVAL, "ggcn_rowmask_from_dense: null pointer");
    if (B <= 0 || T <= 0) return fail(GGCN_EINVAL, "ggcn_rowmask_from_dense: B=%d T=%d must be positive", B, T);
    if (T > GGCN_MASK_MAX_T)
        return fail(GGCN_EUNSUPPORTED, "ggcn_rowmask_from_dense: row masks need T <= %d, got T=%d", GGCN_MASK_MAX_T, T);
    switch (adj_dtype) {
        case GGCN_ADJ_F32: return run_mask<float>(adj, B, T, sb, sr, sc, rowmask, flags, st);
        case GGCN_ADJ_U8: return run_mask<uint8_t>(adj, B, T, sb, sr, sc, rowmask, flags, st);
        case GGCN_ADJ_I32: return run_mask<int32_t>(adj, B, T, sb, sr, sc, rowmask, flags, st);
        case GGCN_ADJ_I64: return run_mask<int64_t>(adj, B, T, sb, sr, sc, rowmask, flags, st);
        case GGCN_ADJ_F64: return run_mask<double>(adj, B, T, sb, sr, sc, rowmask, flags, st);
        case GGCN_ADJ_F16: return run_mask<__half>(adj, B, T, sb, sr, sc, rowmask, flags, st);
        default: return fail(GGCN_EINVAL, "ggcn_rowmask_from_dense: unknown adj_dtype %d", adj_dtype);
    }
}

int csr_from_dense(const void *adj, int adj_dtype, int B, int T, int64_t sb, int64_t sr, int64_t sc,
                   int32_t *rowptr, int32_t *colidx, float *vals, int64_t capacity, uint32_t *rowmask,
                   int32_t *flags, void *workspace, hipStream_t st)
{
    if (rowmask && T > GGCN_MASK_MAX_T)
        return fail(GGCN_EUNSUPPORTED, "ggcn_csr_from_dense: row masks need T <= %d, got T=%d", GGCN_MASK_MAX_T, T);
    if (!adj || !rowptr || !colidx || !workspace)
        return fail(GGCN_EINVAL, "ggcn_csr_from_dense: null pointer");
    if (B <= 0 || T <= 0) return fail(GGCN_EINVAL, "ggcn_csr_from_dense: B=%d T=%d must be positive", B, T);
    if (capacity < 0) return fail(GGCN_EINVAL, "ggcn_csr_from_dense: negative capacity");
    if ((int64_t)B * T * T >= (int64_t)INT32_MAX)
        return fail(GGCN_EUNSUPPORTED, "ggcn_csr_from_dense: B*T*T=%lld does not fit int32 row pointers",
                    (long long)B * T * T);
    switch (adj_dtype) {
        case GGCN_ADJ_F32: return run<float>(adj, B, T, sb, sr, sc, rowptr, colidx, vals, capacity, rowmask, flags, workspace, st);
        case GGCN_ADJ_U8: return run<uint8_t>(adj, B, T, sb, sr, sc, rowptr, colidx, vals, capacity, rowmask, flags, workspace, st);
        case GGCN_ADJ_I32: return run<int32_t>(adj, B, T, sb, sr, sc, rowptr, colidx, vals, capacity, rowmask, flags, workspace, st);
        case GGCN_ADJ_I64: return run<int64_t>(adj, B, T, sb, sr, sc, rowptr, colidx, vals, capacity, rowmask, flags, workspace, st);
        case GGCN_ADJ_F64: return run<double>(adj, B, T, sb, sr, sc, rowptr, colidx, vals, capacity, rowmask, flags, workspace, st);
        case GGCN_ADJ_F16: return run<__half>(adj, B, T, sb, sr, sc, rowptr, colidx, vals, capacity, rowmask, flags, workspace, st);
        default: return fail(GGCN_EINVAL, "ggcn_csr_from_dense: unknown adj_dtype %d", adj_dtype);
    }
}

}  // namespace ggcn
