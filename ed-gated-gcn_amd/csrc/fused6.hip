// The one-launch layer / block of graphs of <= 32 nodes with the f16mx6 main loop (f16mx6_core.h: fp6 correction products,
// 96 instead of 128 matrix-pipe cycles per 32^3 block).  Measured 4 % SLOWER than f16mx8: an experiment, built
// only with `make F16MX6=1` (-DGGCN_WITH_F16MX6); the product library refuses precision GGCN_PREC_F16MX6 without it.
#include "fused_common.h"
#include "f16mx6_core.h"

namespace ggcn {
namespace {

static_assert(mx6::kRaw >= 32768 && EpiLds<mx6::kRaw>::kEnd <= mx6::kLdsBytes6, "f16mx6: the operands go where the RAW stages were");

// ---- the same layer / block with the f16mx6 main loop (f16mx6_core.h): 96 instead of 128 matrix-pipe cycles per 32^3 block ----
// Fast path only: K % 32 == 0, 16-byte aligned rows of X (LDS-DMA moves 16 bytes per lane).  LDS: PLANE + RAW =
// 64 KiB, two workgroups per CU; the epilogue's operands are fetched into the RAW stages once the main loop has released them.
template <bool FULLT, bool VST>
__global__ __launch_bounds__(kThreads, kWavesPerSimd) void layer_fused6_kernel(const FusedArgs a)
{
    __shared__ __attribute__((aligned(16))) char lds[mx6::kLdsBytes6];
    const int B = a.B, T = a.T, K = a.K, F = a.F;
    if (a.ov_in && blockIdx.x == 0) reduce_partials(a.ov_in, B * ((F + 63) / 64), B, a.ov_out, reinterpret_cast<float *>(lds));
    int g_tile, n_wgi;
    bool second = false;
    if (a.n_parts == 1) {
        if (!tile_of_block(blockIdx.x, a.g_tiles, a.n_wg, g_tile, n_wgi)) return;
    } else {   // four XCDs take the W1 tiles, four the W12 tiles (layer_fused_kernel)
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        second = xcd >= 4;
        g_tile = (slot / a.n_wg) * 4 + (xcd & 3);
        n_wgi = slot % a.n_wg;
        if (g_tile >= a.g_tiles) return;
    }
    const LayerPart &lp = a.part[second ? 1 : 0];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g0 = g_tile * 4;
    const int n_tiles_total = (F + NT - 1) / NT;
    const int nt0 = n_wgi * (BN / NT) + wave * RN;

    // DMA sources: piece j of this wavefront = tile rows 32 wave + 8 j + (lane >> 3), chunk (lane & 7) ^ swizzle(row)
    const float *xtile = a.X + (int64_t)g0 * T * a.ldx;   // workgroup-uniform; 128 rows x ldx floats stay below 4 GiB (launcher)
    uint32_t aoff[2];   // pieces 2, 3 = pieces 0, 1 sixteen rows further (same swizzle): a uniform stride in the FULLT build
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int row = 32 * wave + 8 * j + (lane >> 3);
        const int g = g0 + (row >> 5), r = row & 31;
        const bool ok = (g < B) && (FULLT || r < T);
        const int node = ok ? (row >> 5) * T + r : 0;       // inside the tile; clamped: the split zeroes such a block
        aoff[j] = (uint32_t)(((int64_t)node * a.ldx + 4 * ((lane & 7) ^ ((row >> 1) & 7))) * 4);
    }
    const int urow = 32 * wave + (lane >> 1);   // the row whose half block this lane splits
    const bool uvalid = (g0 + (urow >> 5) < B) && (FULLT || (urow & 31) < T);

    f32x16 acc[4][RN];
    const int64_t rows_left = (int64_t)B * T - (int64_t)g0 * T;
    const uint32_t xtile_bytes = (uint32_t)(((rows_left < 128 ? rows_left : 128) - 1) * a.ldx * 4 + (int64_t)K * 4);
    float amax;
    mx6::mainloop<!FULLT>(xtile, xtile_bytes, aoff, (uint32_t)(16 * a.ldx * 4), uvalid, lp.wpack, K, a.k_steps / 2, nt0, n_tiles_total, lds, acc, &amax);
    if constexpr (((GGCN_LAB_OFF) & 128) != 0) {   // ladder: no epilogue
        float sacc = 0.0f;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < RN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) sacc += acc[i][j][r];
        if (sacc == 123.456f && lp.pool_a) lp.pool_a[tid] = sacc;
        return;
    }
    // everything the epilogue derives from the thread id is derived AFTER the loop (the asm makes the id opaque): hipcc
    // otherwise computes those per-lane offsets and pointers up front and keeps ~25 registers alive across a loop that
    // has none to spare
    int tid_e = threadIdx.x;
    asm volatile("" : "+v"(tid_e));
    stage_epilogue_operands<mx6::kRaw>(a, lp, g0, n_wgi, lds, tid_e);   // the RAW stages are free: the loop ended on a barrier
    __syncthreads();
    fused_range_verdict<mx6::kRaw>(amax, lp.wpack, (int64_t)n_tiles_total * (a.k_steps / 2) * mx6::STAGE_PACK_BYTES, lds, false);
    if (lp.mid) {
        if (lp.out) epilogue<2, FULLT, VST, true, true, mx6::kRaw>(a, lp, acc, g0, nt0, n_tiles_total, lds, tid_e);
        else epilogue<2, FULLT, VST, true, false, mx6::kRaw>(a, lp, acc, g0, nt0, n_tiles_total, lds, tid_e);
    } else {
        if (lp.out) epilogue<2, FULLT, VST, false, true, mx6::kRaw>(a, lp, acc, g0, nt0, n_tiles_total, lds, tid_e);
        else epilogue<2, FULLT, VST, false, false, mx6::kRaw>(a, lp, acc, g0, nt0, n_tiles_total, lds, tid_e);
    }
}


}  // namespace

int launch_fused6(const char *who, const FusedArgs &a, bool fullt, bool vst, int64_t grid, hipStream_t st)
{
#define GGCN_LAUNCH6(FT, VS) hipLaunchKernelGGL((layer_fused6_kernel<FT, VS>), dim3((unsigned)grid), dim3(kThreads), 0, st, a)
    if (fullt && vst) GGCN_LAUNCH6(true, true);
    else if (fullt) GGCN_LAUNCH6(true, false);
    else if (vst) GGCN_LAUNCH6(false, true);
    else GGCN_LAUNCH6(false, false);
#undef GGCN_LAUNCH6
    return check_launch(who);
}

GGCN_RANGE_FLAG_TU(range_flag_fused6)

}  // namespace ggcn
