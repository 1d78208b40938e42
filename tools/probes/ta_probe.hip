// Load-shape probe (development tool): how fast does a CU take in a 128-row x 64-k fp32 stage of X when
//   (A) 16 lanes share a row's 256 B per wave-instruction (coalesced: the shape of the current staging), or
//   (C) every lane owns one (row, 32-k block) and loads its 128 contiguous bytes with 8 x dwordx4
//       (64 different 128-B lines per wave-instruction) -- the shape a per-(row, 32 k) scaled split wants.
// Same bytes, same grid as the block kernel: 1024 row blocks x 6 column-tile workgroups re-reading them.
//   hipcc --offload-arch=gfx950 -O3 -o ta_probe ta_probe.hip && ./ta_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

constexpr int K = 768, ROWS = 128, BK = 64;

template <int SHAPE>
__global__ __launch_bounds__(256, 2) void stream(const float *__restrict__ X, float *__restrict__ sink, int n_tiles)
{
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int tile = (slot / 6) * 8 + xcd;          // 6 column-tile workgroups of a row block side by side on one XCD
    if (tile >= n_tiles) return;
    const int tid = threadIdx.x;
    const float *base = X + (size_t)tile * ROWS * K;
    float acc = 0.f;
    if (SHAPE == 0) {
        // per pass: 16 lanes x 16 B = one row's 64 k; 16 rows per pass, 8 passes
        const float *p = base + (size_t)(tid >> 4) * K + (tid & 15) * 4;
        for (int k0 = 0; k0 < K; k0 += BK) {
            float4 v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = *reinterpret_cast<const float4 *>(p + (size_t)(16 * i) * K + k0);
#pragma unroll
            for (int i = 0; i < 8; ++i) acc += (v[i].x + v[i].y) + (v[i].z + v[i].w);
        }
    } else {
        // lane = (row = tid / 2, block = tid % 2): 128 contiguous bytes
        const float *p = base + (size_t)(tid >> 1) * K + (tid & 1) * 32;
        for (int k0 = 0; k0 < K; k0 += BK) {
            float4 v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = *reinterpret_cast<const float4 *>(p + k0 + 4 * i);
#pragma unroll
            for (int i = 0; i < 8; ++i) acc += (v[i].x + v[i].y) + (v[i].z + v[i].w);
        }
    }
    if (acc == 123.456f) sink[blockIdx.x * 256 + tid] = acc;
}

int main()
{
    const int n_tiles = 1024;
    const size_t n = (size_t)n_tiles * ROWS * K;
    float *X, *sink;
    hipMalloc(&X, n * 4);
    hipMalloc(&sink, 1 << 24);
    hipMemset(X, 0, n * 4);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    const int grid = n_tiles / 8 * 6 * 8;
    for (int shape = 0; shape < 2; ++shape) {
        for (int rep = 0; rep < 3; ++rep) {
            for (int w = 0; w < 20; ++w) {
                if (shape == 0) hipLaunchKernelGGL(stream<0>, dim3(grid), dim3(256), 0, 0, X, sink, n_tiles);
                else hipLaunchKernelGGL(stream<1>, dim3(grid), dim3(256), 0, 0, X, sink, n_tiles);
            }
            hipEventRecord(a);
            for (int w = 0; w < 20; ++w) {
                if (shape == 0) hipLaunchKernelGGL(stream<0>, dim3(grid), dim3(256), 0, 0, X, sink, n_tiles);
                else hipLaunchKernelGGL(stream<1>, dim3(grid), dim3(256), 0, 0, X, sink, n_tiles);
            }
            hipEventRecord(b);
            hipEventSynchronize(b);
            float ms;
            hipEventElapsedTime(&ms, a, b);
            const double us = ms * 1e3 / 20;
            printf("shape %c: %8.1f us per pass over 6 x %.0f MB  (%.2f TB/s into the CUs)\n", shape ? 'C' : 'A', us, n * 4 / 1e6,
                   6.0 * n * 4 / us / 1e6);
        }
    }
    return 0;
}
