// Lab-only instrumentation for the matrix kernels (tools/lab.py builds with -DGGCN_LAB_TRACE).
// In the product build every hook below is empty: nothing here changes a product kernel.
#pragma once

// occupancy experiments: extra static LDS per workgroup (0 in the product build)
#ifndef GGCN_LAB_LDS_PAD
#define GGCN_LAB_LDS_PAD 0
#endif

// timing-only probes of the correction MFMA's operand format (cbsz = blgp: 0 fp8 e4m3, 2 fp6 e2m3, 3 bf6 e3m2, 4 fp4);
// anything but 0 reads the fp8 operands as another format: wrong results, right instruction timing
#ifndef GGCN_LAB_MXFMT
#define GGCN_LAB_MXFMT 0
#endif

// timing-only elimination ladder of the f16mx8 main loop (bit set = that step is switched off; wrong results):
// 1 X global loads, 2 split VALU, 4 LDS plane writes, 8 LDS fragment reads, 16 W loads, 32 wh8 converts, 64 barrier,
// 128 epilogue (fused_layer.hip)
#ifndef GGCN_LAB_NT_STORE
#define GGCN_LAB_NT_STORE 0     // 1: the one-launch kernels store the [N,F] output with the non-temporal hint
#endif
#ifndef GGCN_LAB_WIDE8_DENSE
#define GGCN_LAB_WIDE8_DENSE 0   // 1: layer_fused_wide8_kernel aggregates with dense 32 x 32 adjacency blocks on the MFMAs (its first form)
#endif
#ifndef GGCN_LAB_EPI
#define GGCN_LAB_EPI 0           // timing-only switches of the 32-node epilogue: 1 = no [N,F] stores, 2 = no aggregation MFMAs (first application)
#endif
#ifndef GGCN_LAB_WIDE_SB8
#define GGCN_LAB_WIDE_SB8 0   // 1: graphs of 129..256 nodes through the older lone-wavefront form (layer_fused_wide_kernel<.., 8>)
#endif
#ifndef GGCN_LAB_OFF
#define GGCN_LAB_OFF 0
#endif
#ifndef GGCN_LAB_NO_DMA_STAGE
#define GGCN_LAB_NO_DMA_STAGE 0   // 1: the 32-node kernel stages its epilogue operands through registers as before round 5 (A/B switch)
#endif
// timing-only: staging passes (X loads, split, plane writes) a thread performs per stage (4 = all; 2 prices a workgroup of
// eight wavefronts that shares one set of X planes between two column halves; wrong results below 4)
#ifndef GGCN_LAB_XPASSES
#define GGCN_LAB_XPASSES 4
#endif

// f16mx8: 1 = the second column tile's fp16 -> fp8 converts behind the first tile's first MX MFMA
#ifndef GGCN_LAB_WH8
#define GGCN_LAB_WH8 0
#endif

#ifdef GGCN_LAB_TRACE
// timeline probe: per workgroup {block, HW_ID, XCC_ID, t_start, t_loop_begin, t_loop_end, t_end} in 10 ns ticks
__device__ unsigned long long ggcn_trace_buf[8192 * 8];
#define GGCN_TRACE(slot)                                                                                        \
    do {                                                                                                        \
        if (threadIdx.x == 0 && blockIdx.x < 8192) ggcn_trace_buf[blockIdx.x * 8 + (slot)] = wall_clock64();   \
    } while (0)
#define GGCN_TRACE_IDS()                                                                                         \
    do {                                                                                                         \
        if (threadIdx.x == 0 && blockIdx.x < 8192) {                                                             \
            ggcn_trace_buf[blockIdx.x * 8 + 0] = blockIdx.x;                                                     \
            ggcn_trace_buf[blockIdx.x * 8 + 1] = __builtin_amdgcn_s_getreg(4 | (31 << 11));  /* HW_REG_HW_ID */  \
            ggcn_trace_buf[blockIdx.x * 8 + 2] = __builtin_amdgcn_s_getreg(20 | (31 << 11)); /* HW_REG_XCC_ID */ \
        }                                                                                                        \
    } while (0)
#define GGCN_TRACE_READER                                                                                \
    extern "C" int ggcn_lab_trace_read(void *dst, size_t bytes)                                          \
    {                                                                                                    \
        return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(ggcn_trace_buf), bytes, 0, hipMemcpyDeviceToHost); \
    }
#else
#define GGCN_TRACE(slot) do { } while (0)
#define GGCN_TRACE_IDS() do { } while (0)
#define GGCN_TRACE_READER
#endif

// the same for layer_fused_long_kernel (fused_long.hip; tools/long_trace.py, build flag -DGGCN_LAB_TRACE_LONG)
#ifdef GGCN_LAB_TRACE_LONG
__device__ unsigned long long ggcn_trace_long[4096 * 8];
#define GGCN_LT(slot) do { if (threadIdx.x == 0 && blockIdx.x < 4096) ggcn_trace_long[blockIdx.x * 8 + (slot)] = wall_clock64(); } while (0)
#define GGCN_LT_WAVE7(slot) do { if (threadIdx.x == 448 && blockIdx.x < 4096) ggcn_trace_long[blockIdx.x * 8 + (slot)] = wall_clock64(); } while (0)
#define GGCN_LT_READER                                                                                    \
    extern "C" int ggcn_lab_trace_read_long(void *dst, size_t bytes)                                      \
    {                                                                                                     \
        return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(ggcn_trace_long), bytes, 0, hipMemcpyDeviceToHost); \
    }
#else
#define GGCN_LT(slot) do { } while (0)
#define GGCN_LT_WAVE7(slot) do { } while (0)
#define GGCN_LT_READER
#endif
