// Box calibration (diagnostics, never on the product path): what a given MI355X makes of a fixed amount of matrix-pipe
// work.  The block kernel (fused_layer.hip) is pinned at the board's power cap, so its time follows the clock each device
// holds under an MFMA-dense load (MI355X_MICROARCH.md, "DVFS give-back" (5): 12 % between devices); bench.py times this
// loop right after its timed region and prints the result in the `box` block of its JSON line, so that a figure taken
// on one box can be read against a figure taken on another.
//
//   * the matrix-pipe work of the f16mx8 main loop and nothing else: per wavefront and "stage" 16 x
//     v_mfma_f32_32x32x16_f16 + 8 x v_mfma_scale_f32_32x32x64_f8f6f4 (fp8) on random register operands -- 128 rows x 64
//     columns x 32 k of the product X.W (f16mx8_core.h) -- with NO memory traffic, LDS traffic or barrier inside the loop;
//   * the block kernel's occupancy: 256 threads, two workgroups per CU (an LDS allocation enforces it), 128 accumulators;
//   * thread 0 of every workgroup stamps s_memtime (shader cycles) and s_memrealtime (100 MHz) around its loop into a buffer
//     nothing else reads: clock held under this load = median over workgroups of d(cycles) / d(ticks) x 100 MHz
//     (MI355X_MICROARCH.md, "DVFS give-back" (6)).
// With n_wg = 6144 and stages = 24 the launch issues exactly the main-loop MFMAs of ggcn_block_fused at BASELINE config 2
// (4096 graphs x 32 nodes, K = F = 768, two parts): its duration is that kernel's matrix-pipe floor on this box.
#include "common.h"

namespace ggcn {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t mix32(uint32_t x)   // lowbias32
{
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
// two fp16 values in [1, 2) with random mantissas and signs; four fp8 (e4m3) bytes with exponents 4..11 (never NaN)
__device__ __forceinline__ uint32_t rand_h2(uint32_t s) { return (mix32(s) & 0x83FF83FFu) | 0x3C003C00u; }
__device__ __forceinline__ uint32_t rand_q4(uint32_t s) { return (mix32(s) & 0xBFBFBFBFu) | 0x20202020u; }

constexpr int kCalibLds = 68 * 1024;   // two workgroups per CU (160 KiB), as the block kernel runs

__global__ __launch_bounds__(256, 2) void mfma_calib_kernel(int stages, unsigned long long *__restrict__ stamps,
                                                            float *__restrict__ sink)
{
    __shared__ char pad[kCalibLds];
    const uint32_t seed = (blockIdx.x * 256u + threadIdx.x) * 64u;
    if (stages < 0) pad[threadIdx.x] = (char)seed;   // never taken: the allocation must not be dropped
    f16x8 ah[4][2], b0[2], b1[2];
    i32x8 aq[4], bm[2];
    uint32_t n = seed;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            i32x4 t;
#pragma unroll
            for (int d = 0; d < 4; ++d) t[d] = (int)rand_h2(n++);
            ah[i][s] = __builtin_bit_cast(f16x8, t);
        }
#pragma unroll
        for (int d = 0; d < 8; ++d) aq[i][d] = (int)rand_q4(n++);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        i32x4 t, u;
#pragma unroll
        for (int d = 0; d < 4; ++d) { t[d] = (int)rand_h2(n++); u[d] = (int)rand_h2(n++); }
        b0[j] = __builtin_bit_cast(f16x8, t);
        b1[j] = __builtin_bit_cast(f16x8, u);
#pragma unroll
        for (int d = 0; d < 8; ++d) bm[j][d] = (int)rand_q4(n++);
    }
    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    __syncthreads();
    const unsigned long long c0 = __builtin_readcyclecounter();          // s_memtime: shader cycles
    const unsigned long long w0 = __builtin_amdgcn_s_memrealtime();      // constant 100 MHz
    for (int st = 0; st < stages; ++st) {
        // the slot order of mx8::mainloop: per 32-row block 4 fp16 MFMAs, then per block 2 MX MFMAs
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i][0], b0[0], acc[i][0], 0, 0, 0);
            acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i][0], b0[1], acc[i][1], 0, 0, 0);
            acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i][1], b1[0], acc[i][0], 0, 0, 0);
            acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i][1], b1[1], acc[i][1], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            acc[i][0] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(aq[i], bm[0], acc[i][0], 0, 0, 0, 127, 0, 127);
            acc[i][1] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(aq[i], bm[1], acc[i][1], 0, 0, 0, 127, 0, 127);
        }
        // (the operands stay; consecutive MFMAs already alternate between different A and B registers)
    }
    // the stamps must not be taken before the loop's last MFMA has retired: read one accumulator element first
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    asm volatile("" :: "v"(s));
    __syncthreads();
    const unsigned long long c1 = __builtin_readcyclecounter();
    const unsigned long long w1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0 && stamps) {
        stamps[2 * (size_t)blockIdx.x] = c1 - c0;
        stamps[2 * (size_t)blockIdx.x + 1] = w1 - w0;
    }
    if (s == 123.456f && sink) sink[threadIdx.x] = s;   // keeps the products alive
}

}  // namespace

int mfma_calibrate(int n_wg, int stages, unsigned long long *stamps, float *sink, hipStream_t st)
{
    if (n_wg <= 0 || stages <= 0 || stages > (1 << 20))
        return fail(GGCN_EINVAL, "ggcn_debug_mfma_calibrate: n_wg=%d stages=%d must be positive", n_wg, stages);
    hipLaunchKernelGGL(mfma_calib_kernel, dim3((unsigned)n_wg), dim3(256), 0, st, stages, stamps, sink);
    return check_launch("ggcn_debug_mfma_calibrate");
}

}  // namespace ggcn
