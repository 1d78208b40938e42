#include <hip/hip_runtime.h>
typedef short s4 __attribute__((ext_vector_type(4)));
// tile [32 rows k][128 cols] of 16-bit, image (b): off = 256*row + 16*(ch ^ (((row&3)<<2) | ((row>>2)&3)))
__device__ __forceinline__ int off_b(int row, int ch) { return 256 * row + 16 * (ch ^ (((row & 3) << 2) | ((row >> 2) & 3))); }
__global__ void k(const unsigned short* in /*[32][128]*/, unsigned short* out /*[64 lanes][8]*/, int s, int c0blk){
  __shared__ __attribute__((aligned(16))) char lds[32*256];
  for (int i = threadIdx.x; i < 32*128; i += 64) { int r = i / 128, c = i % 128; *(unsigned short*)(lds + off_b(r, c >> 3) + (c & 7) * 2) = in[i]; }
  __syncthreads();
  const int lane = threadIdx.x, g = lane >> 4, li = lane & 15, q = li >> 2, p = li & 3;
  const int h = g >> 1;
  const int c0 = c0blk * 32 + 16 * (g & 1);
  unsigned short res[8];
  for (int half = 0; half < 2; ++half) {
    const int r0 = 16 * s + 8 * h + 4 * half;
    const int addr = off_b(r0 + q, (c0 >> 3) + (p >> 1)) + 8 * (p & 1);
    s4 v;
    const unsigned a32 = (unsigned)(size_t)(lds + addr);   // LDS address
    asm volatile("ds_read_b64_tr_b16 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a32));
    for (int e = 0; e < 4; ++e) res[4*half + e] = (unsigned short)v[e];
  }
  for (int e = 0; e < 8; ++e) out[lane*8+e] = res[e];
}
extern "C" int run_tr(const unsigned short* in, unsigned short* out, int s, int c0blk){ hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, in, out, s, c0blk); return (int)hipDeviceSynchronize(); }
