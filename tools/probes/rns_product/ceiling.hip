// Micro-benchmark: what an fp64-accurate complex product emulated on the int8 matrix pipe could reach on MI355X
// (residue arithmetic: 17 moduli p = 1 mod 4, a complex number mod p split into its two images x + iy, x - iy
// with i^2 = -1 mod p, so one complex product = 2 real int8 products per modulus = 34 planes; the int32 sums
// of every plane are folded into the fraction of the Chinese remainder reconstruction in fp64 VALU).
// Register-only: operands never move, so this is the ceiling of the arithmetic alone, per wave tile of 32 x 32
// complex outputs and K = 320:  per plane 5 k-steps x (2 x 2) v_mfma_i32_16x16x64_i8, then the fold of the
// lane's 16 sums.  Compare with the fp64 3M product of the library (85 TFLOP/s in the same 8-flop-per-complex-
// multiply-add units, K loop alone).
// Build: hipcc --offload-arch=gfx950 -O3 tools/ozaki_ceiling.hip -o tools/ozaki_ceiling
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));

struct Plane {
  float p, invp;
  double xh, xl, yh, yl;
};
constexpr int NPLANES = 34;
__constant__ Plane planes[NPLANES];

template <bool MFMA, bool FOLD, int FOLD_KIND>
__global__ __launch_bounds__(256, 2) void tile_loop(double* out, int iters, int seed) {
  double S[16][4];
#pragma unroll
  for (int e = 0; e < 16; ++e) S[e][0] = S[e][1] = S[e][2] = S[e][3] = 0.0;
  v4i a0 = {seed + (int)threadIdx.x, seed * 3, seed * 5 + 1, seed ^ 0x55aa55aa}, a1 = a0 * 3, b0 = a0 * 7, b1 = a0 * 11;
  double sum = 0.0;
#pragma unroll 1
  for (int it = 0; it < iters; ++it) {
#pragma unroll 1
    for (int pl = 0; pl < NPLANES; ++pl) {
      v4i c[4] = {{pl, 0, 0, 0}, {0, pl, 0, 0}, {0, 0, pl, 0}, {0, 0, 0, pl}};
      if (MFMA) {
#pragma unroll
        for (int k = 0; k < 5; ++k) {
          c[0] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a0, b0, c[0], 0, 0, 0);
          c[1] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a0, b1, c[1], 0, 0, 0);
          c[2] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a1, b0, c[2], 0, 0, 0);
          c[3] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a1, b1, c[3], 0, 0, 0);
        }
      }
      if (FOLD) {
        const Plane P = planes[pl];
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int t = c[e >> 2][e & 3];
          if (FOLD_KIND == 0) {  // reduce mod p in fp32, then hi / lo sums for both parts
            const float tf = (float)t;
            const float q = __builtin_rintf(tf * P.invp);
            const float r = __builtin_fmaf(-P.p, q, tf);
            const double rd = (double)r;
            S[e][0] = __builtin_fma(rd, P.xh, S[e][0]);
            S[e][1] = __builtin_fma(rd, P.xl, S[e][1]);
            S[e][2] = __builtin_fma(rd, P.yh, S[e][2]);
            S[e][3] = __builtin_fma(rd, P.yl, S[e][3]);
          } else {  // no reduction: the raw sum times three-piece weights would need 6 sums; priced here as 4 + cvt
            const double rd = (double)t;
            S[e][0] = __builtin_fma(rd, P.xh, S[e][0]);
            S[e][1] = __builtin_fma(rd, P.xl, S[e][1]);
            S[e][2] = __builtin_fma(rd, P.yh, S[e][2]);
            S[e][3] = __builtin_fma(rd, P.yl, S[e][3]);
          }
        }
      } else {
        sum += (double)(c[0][0] + c[1][1] + c[2][2] + c[3][3]);
      }
    }
    // end of the tile: fraction of the reconstruction, scaled
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const double fx = (S[e][0] - __builtin_rint(S[e][0])) + S[e][1], fy = (S[e][2] - __builtin_rint(S[e][2])) + S[e][3];
      sum += fx * 1.5 + fy;
      S[e][0] = S[e][1] = S[e][2] = S[e][3] = 0.0;
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = sum;
}

template <bool MFMA, bool FOLD, int KIND>
void run(const char* what, int blocks_per_cu, int iters) {
  const int blocks = 256 * blocks_per_cu;
  double* d;
  hipMalloc(&d, sizeof(double) * blocks * 256);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  tile_loop<MFMA, FOLD, KIND><<<blocks, 256>>>(d, 20, 1);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  tile_loop<MFMA, FOLD, KIND><<<blocks, 256>>>(d, iters, 1);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double tiles = (double)blocks * 4 * iters;
  const double equiv = tiles * 8.0 * 285.0 * 1024.0;               // fp64 flops the tile stands for (8 per complex multiply-add)
  const double i8ops = tiles * NPLANES * 20.0 * 2.0 * 16 * 16 * 64;  // int8 operations issued
  printf("%-34s waves/SIMD=%d  %8.2f ms  %7.1f TFLOP/s fp64-equivalent  %6.2f POP/s int8  %7.1f SIMD-cycles@2.4GHz per output\n", what, blocks_per_cu, ms,
         equiv / ms * 1e-9, MFMA ? i8ops / ms * 1e-12 : 0.0, ms * 1e-3 * 2.4e9 / (iters * 1024.0 * blocks_per_cu));
  hipFree(d);
}

int main() {
  Plane h[NPLANES];
  const int mods[17] = {241, 233, 229, 197, 193, 181, 173, 169, 157, 149, 137, 125, 113, 109, 101, 97, 89};
  for (int i = 0; i < NPLANES; ++i) {
    const double p = mods[i / 2];
    h[i] = {(float)p, (float)(1.0 / p), (double)(float)(37.0 / p), 37.0 / p - (double)(float)(37.0 / p), (double)(float)(91.0 / p), 91.0 / p - (double)(float)(91.0 / p)};
  }
  hipMemcpyToSymbol(HIP_SYMBOL(planes), h, sizeof(h));
  for (int w = 1; w <= 2; ++w) {
    run<true, false, 0>("int8 products only", w, 400);
    run<false, true, 0>("fold only (reduce mod p first)", w, 400);
    run<true, true, 0>("products + fold (reduce mod p)", w, 400);
    run<true, true, 1>("products + fold (4 fma, no reduce)", w, 400);
  }
  return 0;
}
