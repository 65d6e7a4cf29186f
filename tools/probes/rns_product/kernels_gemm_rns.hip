// The dense complex product of the synthesis (scri/waveform_grid.py:475-484) on the INT8 matrix pipe, to fp64 accuracy.
//
// MI355X multiplies fp64 matrices at 78.6 TFLOP/s (60 sustained) and int8 matrices at 5 POP/s (3.4-4.2 sustained): a
// factor of ~60.  The product C = A . B (A: time steps x modes, B: modes x directions, complex128) is therefore computed in
// exact integer arithmetic, in residues:
//   * every row of A and every column of B is scaled by a power of two and rounded to integers X with |X| <= 2^53 (so an
//     entry keeps the 53 bits fp64 gives it relative to the largest entry of its row / column: the normwise accuracy of an
//     fp64 product, whose own error bound is eps sum_k |a_k| |b_k|), the L1 norm of a scaled row of A <= 2^62;
//   * 16 pairwise coprime moduli m <= 241, all with -1 a quadratic residue (primes = 1 mod 4, 13^2, 5^3), product
//     P = 2^116.59 > 2 |C'| for every integer result C';
//   * modulo such an m the Gaussian integers split: with iota^2 = -1 (mod m), z = x + i y -> (x + iota y, x - iota y) is a
//     ring homomorphism into Z_m x Z_m, so ONE COMPLEX product is TWO REAL products of int8 residues per modulus -- 32
//     planes of v_mfma_i32_16x16x64_i8 in all (the fp64 form spends 3 real fp64 products, each 60x more expensive);
//   * the int32 sums t+, t- of a modulus give the residues of the result, x = (t+ + t-) / 2, y = (t+ - t-) / (2 iota), and the
//     Chinese remainder theorem the result itself: X_C / P = sum_l x_l c_l / m_l (mod 1).  The kernel folds each modulus
//     into that fraction as it completes:  r = (t+ +- t-) mod m (|r| <= m, exact in fp64), S_hi += r theta_hi (theta_hi has 39
//     bits: products and sums exact), S_lo += r theta_lo; at the end frac = (S_hi - rint(S_hi)) + S_lo to ~2^-80 and
//     X_C = P frac in double-double, rounded once to fp64.  Nothing else is rounded: the integer product is exact.
// tools/probes/rns_product/prototype.py is the same arithmetic in numpy, checked against Python integers; on matrices whose
// rows span 1, 1e6 and 1e12 in magnitude its error is below that of numpy's fp64 product.
//
// Memory layout of the residue planes ("fragment order": what one v_mfma_i32_16x16x64_i8 operand register holds is
// 16 contiguous bytes, what a wave loads for it 1 KB): plane p = 2 l + image; within a plane
//     [block of 16 rows][k step of 64][lane = 16 (k / 16 % 4) + row % 16][16 bytes: k % 16]
// for A (rows = time steps) and for B transposed (rows = directions).  A workgroup tile of 64 rows x all k of one plane is
// one contiguous piece; the operand reads from LDS are lane-contiguous ds_read_b128 (conflict free by construction).
#include <cstdint>
#include "kernels.h"

namespace bms {

typedef int v4i32 __attribute__((ext_vector_type(4)));

struct RnsMod {
  double m, inv, iota, xh, xl, yh, yl;
};
constexpr int RNS_NMOD = 16;
// m, 1/m, iota (the smaller square root of -1), and theta = (c / 2 mod m) / m, eta = (c / (2 iota) mod m) / m with
// c = (P / m)^-1 mod m, each split into 39 bits + remainder (generated with exact integer arithmetic:
// tools/probes/rns_product/prototype.py::constants / split_weight)
__device__ __constant__ const RnsMod rns_mod[RNS_NMOD] = {
    {241.0, 0x1.0fef010fef011p-8, 64.0, 0x1.42ebd142e8000p-1, 0x1.e8a175e8a175fp-40, 0x1.450baf4508000p-1, 0x1.d7a285d7a285dp-40},
    {233.0, 0x1.19453808ca29cp-8, 89.0, 0x1.2f3ea06978000p-2, 0x1.f5034bcfa81a6p-42, 0x1.499d1daa4c000p-1, 0x1.d1daa4ce8ed52p-42},
    {229.0, 0x1.1e2ef3b3fb874p-8, 107.0, 0x1.4f5f0596e8000p-1, 0x1.6141f4d22a7b0p-40, 0x1.d348a9ebe0000p-1, 0x1.65bab0a0fa691p-42},
    {197.0, 0x1.4cab88725af6ep-8, 14.0, 0x1.7640f980a0000p-3, 0x1.95710e4b5edcfp-41, 0x1.c4392d7b70000p-2, 0x1.d3d137e0cfeb3p-41},
    {193.0, 0x1.5390948f40febp-8, 81.0, 0x1.43a5cd9888000p-2, 0x1.f2bc5a3267761p-42, 0x1.9889f2bc58000p-2, 0x1.1933bb06a1d2ep-41},
    {181.0, 0x1.6a13cd1537290p-8, 19.0, 0x1.3454dca410000p-1, 0x1.f1db39fd2bd86p-42, 0x1.1db39fd2bc000p-1, 0x1.865d591adf784p-41},
    {173.0, 0x1.7ad2208e0ecc3p-8, 80.0, 0x1.80bd691040000p-2, 0x1.c1d986a8b1928p-40, 0x1.e2679574e4000p-1, 0x1.6c05eb4882384p-40},
    {169.0, 0x1.83c977ab2beddp-8, 70.0, 0x1.cc7f3e1b44000p-1, 0x1.535048b5c6702p-44, 0x1.535048b5c0000p-5, 0x1.9c060f25deacbp-43},
    {157.0, 0x1.a16d3f97a4b02p-8, 28.0, 0x1.f637708270000p-1, 0x1.11efb1bb84139p-40, 0x1.11efb1bb84000p-1, 0x1.3911efb1bb841p-45},
    {149.0, 0x1.b7d6c3dda338bp-8, 44.0, 0x1.eed19c5950000p-2, 0x1.e7f24149e112ep-40, 0x1.79fc905278000p-1, 0x1.12e63a6a86037p-43},
    {137.0, 0x1.de5d6e3f8868ap-8, 37.0, 0x1.de5d6e3f80000p-6, 0x1.0d148e03bcbaep-43, 0x1.d6e3f88688000p-1, 0x1.2380ef2eb71fcp-40},
    {125.0, 0x1.0624dd2f1a9fcp-7, 57.0, 0x1.b22d0e5600000p-2, 0x1.0624dd2f1a9fcp-40, 0x1.a9fbe76c88000p-1, 0x1.a1cac083126e9p-40},
    {113.0, 0x1.21fb78121fb78p-7, 15.0, 0x1.6a7a5616a0000p-3, 0x1.e9585a9e9585bp-41, 0x1.616a7a5610000p-2, 0x1.a9e9585a9e958p-40},
    {109.0, 0x1.2c9fb4d812ca0p-7, 33.0, 0x1.d5b98a9190000p-3, 0x1.ab7315233ab73p-40, 0x1.b98a919d58000p-2, 0x1.cc548ceadcc55p-41},
    {101.0, 0x1.446f86562d9fbp-7, 10.0, 0x1.d260511be0000p-2, 0x1.958b67ebb907ap-42, 0x1.c83cd4e930000p-2, 0x1.446f86562d9fbp-45},
    {97.0, 0x1.51d07eae2f815p-7, 22.0, 0x1.8699127964000p-1, 0x1.76c34c893cb37p-40, 0x1.bb61a64490000p-3, 0x1.cb376c34c893dp-40},
};
constexpr double RNS_P_HI = 0x1.8167ea70a7151p+116, RNS_P_LO = 0x1.f1f9402555558p+61;  // P = 125071372061214021585550301528240125

long long rns_blocks(long long n) { return (n + 127) / 128 * 8; }  // 16-row blocks, padded to whole 128-row (column) tiles
int rns_ksteps(int K) { return (K + 63) / 64; }
size_t rns_plane_bytes(long long n, int K) { return (size_t)rns_blocks(n) * rns_ksteps(K) * 1024; }
size_t rns_planes_bytes(long long n, int K) { return 2 * RNS_NMOD * rns_plane_bytes(n, K); }

// ------------------------------------------------------------------------------------------------ scales
// One wave per row: s = min(53 - exponent of the largest part, 62 - exponent of the L1 norm of the parts), clamped; a row that
// holds a NaN or an infinity gets the inverse scale NaN, which makes every result of that row NaN (as the product would).
__global__ __launch_bounds__(256) void rns_scale_kernel(const double* __restrict__ X, long long s_n, long long s_k, long long n, int K, double* __restrict__ scale,
                                                        double* __restrict__ inv_scale) {
  const long long i = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (i >= n) return;
  double mx = 0.0, l1 = 0.0;
  bool bad = false;
  for (int k = lane; k < K; k += 64) {
    const double2 v = *reinterpret_cast<const double2*>(X + i * s_n + k * s_k);
    const double ar = __builtin_fabs(v.x), ai = __builtin_fabs(v.y);
    bad = bad || !(ar < __builtin_inf()) || !(ai < __builtin_inf());
    mx = __builtin_fmax(mx, __builtin_fmax(ar, ai));
    l1 += ar + ai;
  }
#pragma unroll
  for (int o = 32; o; o >>= 1) {
    mx = __builtin_fmax(mx, __shfl_xor(mx, o));
    l1 += __shfl_xor(l1, o);
    bad = bad || __shfl_xor((int)bad, o);
  }
  if (lane == 0) {
    int e_max = 0, e_l1 = 0;
    (void)frexp(mx, &e_max);
    (void)frexp(l1, &e_l1);
    int s = min(53 - e_max, 62 - e_l1);
    s = max(-1000, min(1000, s));
    if (mx == 0.0) s = 0;
    scale[i] = bad ? 0.0 : ldexp(1.0, s);
    inv_scale[i] = bad ? __builtin_nan("") : ldexp(1.0, -s);
  }
}

// ------------------------------------------------------------------------------------------------ residue planes
// wave = (block of 16 rows, k step): lane (row = lane & 15, kb = lane >> 4) owns the 16 entries k = 64 ks + 16 kb + 0..15 of
// its row and writes their 16 bytes into each of the 32 planes -- one contiguous KB per wave and plane.
__device__ __forceinline__ unsigned rns_pack4(float a, float b, float c, float d) {
  const unsigned ia = (unsigned)(int)a, ib = (unsigned)(int)b, ic = (unsigned)(int)c, id = (unsigned)(int)d;
  const unsigned ab = __builtin_amdgcn_perm(ib, ia, 0x0c0c0400u), cd = __builtin_amdgcn_perm(id, ic, 0x0c0c0400u);
  return __builtin_amdgcn_perm(cd, ab, 0x05040100u);
}
__global__ __launch_bounds__(256) void rns_residue_kernel(const double* __restrict__ X, long long s_n, long long s_k, long long n, int K,
                                                          const double* __restrict__ scale, int8_t* __restrict__ planes, long long plane_stride, int nks,
                                                          long long n_waves) {
  const long long w = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (w >= n_waves) return;
  const int lane = threadIdx.x & 63;
  const long long blk = w / nks;
  const int ks = (int)(w - blk * nks);
  const long long row = blk * 16 + (lane & 15);
  const int k0 = ks * 64 + (lane >> 4) * 16;
  double xr[16], xi[16];
  const double sc = row < n ? scale[row] : 0.0;
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    double2 v = {0.0, 0.0};
    if (row < n && k0 + e < K) v = *reinterpret_cast<const double2*>(X + row * s_n + (long long)(k0 + e) * s_k);
    xr[e] = __builtin_rint(v.x * sc);
    xi[e] = __builtin_rint(v.y * sc);
  }
  int8_t* dst = planes + (blk * nks + ks) * 1024 + lane * 16;
#pragma unroll 1
  for (int l = 0; l < RNS_NMOD; ++l) {
    const double m = rns_mod[l].m, inv = rns_mod[l].inv;
    const float mf = (float)m, invf = (float)inv, iota = (float)rns_mod[l].iota;
    float u[16], v[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      // remainder of each part: the quotient's rounding is off by one only when its fraction is within 2^-6 of a half, the
      // remainder then within m / 2 + 4; the images are reduced once more (small numbers: exact)
      const float rx = (float)__builtin_fma(-m, __builtin_rint(xr[e] * inv), xr[e]);
      const float ry = (float)__builtin_fma(-m, __builtin_rint(xi[e] * inv), xi[e]);
      const float uu = __builtin_fmaf(iota, ry, rx), vv = __builtin_fmaf(-iota, ry, rx);
      u[e] = __builtin_fmaf(-mf, __builtin_rintf(uu * invf), uu);
      v[e] = __builtin_fmaf(-mf, __builtin_rintf(vv * invf), vv);
    }
    uint4 pu, pv;
    pu.x = rns_pack4(u[0], u[1], u[2], u[3]), pu.y = rns_pack4(u[4], u[5], u[6], u[7]);
    pu.z = rns_pack4(u[8], u[9], u[10], u[11]), pu.w = rns_pack4(u[12], u[13], u[14], u[15]);
    pv.x = rns_pack4(v[0], v[1], v[2], v[3]), pv.y = rns_pack4(v[4], v[5], v[6], v[7]);
    pv.z = rns_pack4(v[8], v[9], v[10], v[11]), pv.w = rns_pack4(v[12], v[13], v[14], v[15]);
    *reinterpret_cast<uint4*>(dst + (2 * l) * plane_stride) = pu;
    *reinterpret_cast<uint4*>(dst + (2 * l + 1) * plane_stride) = pv;
  }
}

// X: n "rows" of K complex entries, entry (i, k) at X + i s_n + k s_k (doubles): A as it is (s_n = lda, s_k = 2), B transposed
// (s_n = 2, s_k = ldb).  planes: rns_planes_bytes(n, K); scale / inv_scale: n doubles each.
hipError_t launch_rns_residues(hipStream_t stream, const double* X, long long s_n, long long s_k, long long n, int K, int8_t* planes, double* scale,
                               double* inv_scale) {
  if (n <= 0 || K <= 0) return hipSuccess;
  hipLaunchKernelGGL(rns_scale_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, stream, X, s_n, s_k, n, K, scale, inv_scale);
  const int nks = rns_ksteps(K);
  const long long n_waves = rns_blocks(n) * nks;
  hipLaunchKernelGGL(rns_residue_kernel, dim3((unsigned)((n_waves + 3) / 4)), dim3(256), 0, stream, X, s_n, s_k, n, K, scale, planes,
                     (long long)rns_plane_bytes(n, K), nks, n_waves);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ the product
// The product is bound by operand bytes before it is bound by the matrix pipe (32 bytes per complex entry, and a CU draws
// ~70 GB/s from its L2, ~33 GB/s from the Infinity Cache), so the tile is as large as the registers allow: the running sums
// of the reconstruction cost 8 registers per result, and a wave holds 32 x 64 results (256 registers) next to its
// accumulators.  Workgroup tile 64 rows x 128 columns, 4 waves (2 x 2) of 32 x 64 = 2 x 4 MFMA blocks, one workgroup per CU.
// A stage = the operands of one (modulus, image, chunk of KC k steps): 4 KC KB of A + 8 KC KB of B, brought global -> LDS by
// LDS-DMA (global_load_lds_dwordx4: the planes are stored in the order the lanes want them, so a wave's KB is lane-linear
// on both sides) through a ring of NBUF buffers: the requests of stage s + NBUF - 1 are issued at the start of stage s, a
// counted s_waitcnt vmcnt leaves them in flight across the raw barrier that publishes stage s + 1.  Every stage issues the
// same number of requests (a short last chunk re-requests its first piece) so that the count is a constant.  After the
// second image of a modulus every lane folds its 32 results into the four running sums of each.
struct RnsFold {
  double xh, xl, yh, yl;
};
typedef __attribute__((address_space(3))) void* rns_lds_ptr;
typedef const __attribute__((address_space(1))) void* rns_glb_ptr;

template <int KC, int NBUF>
__global__ __launch_bounds__(256, 1) void zgemm_rns_kernel(const int8_t* __restrict__ Ap, long long a_stride, const int8_t* __restrict__ Bp, long long b_stride,
                                                           int nks, const double* __restrict__ a_inv, const double* __restrict__ b_inv,
                                                           double* __restrict__ C, long long ldc, long long M, int N, int nbm, int nbn, int st_rows_log2,
                                                           const double* __restrict__ col_off, const double* __restrict__ col_scale, int knock) {
  extern __shared__ __attribute__((aligned(16))) v4i32 rns_lds[];  // [NBUF][A: 4 blocks x KC | B: 8 blocks x KC][64 lanes]
  const int b = blockIdx.x;
  const int xcd = b & 7;
  const int q = b >> 3;
  const int st_cols_log2 = 6 - st_rows_log2;
  const int nsn = (nbn + (1 << st_cols_log2) - 1) >> st_cols_log2;
  const int S = (q >> 6) * 8 + xcd;
  const int r = q & 63;
  const int bm = ((S / nsn) << st_rows_log2) + (r >> st_cols_log2);
  const int bn = ((S % nsn) << st_cols_log2) + (r & ((1 << st_cols_log2) - 1));
  if (bm >= nbm || bn >= nbn) return;

  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int wm = wave >> 1, wn = wave & 1;
  const int n_chunks = (nks + KC - 1) / KC;
  const int n_stages = 2 * RNS_NMOD * n_chunks;
  // staging: wave w copies block w of the A tile and blocks 2 w, 2 w + 1 of the B tile, one KB per block and k step
  const v4i32* a_src = reinterpret_cast<const v4i32*>(Ap + ((long long)(4 * bm + wave) * nks) * 1024) + lane;
  const v4i32* b_src = reinterpret_cast<const v4i32*>(Bp + ((long long)(8 * bn + 2 * wave) * nks) * 1024) + lane;
  constexpr int BUF = 12 * KC * 64;  // v4i32 per buffer
  constexpr int AHEAD = NBUF - 1;

  auto issue_stage = [&](int s) {
    const int plane = s / n_chunks, ch = s - plane * n_chunks;
    const int ks0 = ch * KC;
    const v4i32* ap = a_src + (plane * a_stride) / 16 + ks0 * 64;
    const v4i32* bp = b_src + (plane * b_stride) / 16 + ks0 * 64;
    v4i32* as_w = rns_lds + (s % NBUF) * BUF + (wave * KC) * 64;  // (wave-uniform: the lanes land at + 16 bytes each)
    v4i32* bs_w = rns_lds + (s % NBUF) * BUF + (4 * KC + 2 * wave * KC) * 64;
#pragma unroll
    for (int j = 0; j < KC; ++j) {
      const int jj = ks0 + j < nks ? j : 0;
      __builtin_amdgcn_global_load_lds((rns_glb_ptr)(ap + jj * 64), (rns_lds_ptr)(as_w + j * 64), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((rns_glb_ptr)(bp + jj * 64), (rns_lds_ptr)(bs_w + j * 64), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((rns_glb_ptr)(bp + (nks + jj) * 64), (rns_lds_ptr)(bs_w + (KC + j) * 64), 16, 0, 0);
    }
  };

  RnsFold F[32];
#pragma unroll
  for (int e = 0; e < 32; ++e) F[e].xh = F[e].xl = F[e].yh = F[e].yl = 0.0;
  v4i32 acc[2][8];
#pragma unroll
  for (int im = 0; im < 2; ++im)
#pragma unroll
    for (int t = 0; t < 8; ++t) acc[im][t] = v4i32{0, 0, 0, 0};

#pragma unroll
  for (int p = 0; p < AHEAD; ++p)
    if (p < n_stages) issue_stage(p);
  if (AHEAD > 1 && n_stages > 1)
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * KC * (AHEAD - 1)) : "memory");
  else
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  const bool wave_has_columns = bn * 128 + wn * 64 < N;
  int s = 0;
#pragma unroll 1
  for (int l = 0; l < RNS_NMOD; ++l) {
#pragma unroll
    for (int im = 0; im < 2; ++im) {
#pragma unroll 1
      for (int ch = 0; ch < n_chunks; ++ch, ++s) {
        // (the buffer of stage s + AHEAD was last read in stage s - 1, behind that stage's barrier)
        if (s + AHEAD < n_stages && !(knock & 4)) issue_stage(s + AHEAD);
        const v4i32* as = rns_lds + (s % NBUF) * BUF + (2 * wm * KC) * 64 + lane;
        const v4i32* bs = rns_lds + (s % NBUF) * BUF + (4 * KC + 4 * wn * KC) * 64 + lane;
        const int kcount = min(KC, nks - ch * KC);
        if (wave_has_columns && !(knock & 2)) {
#pragma unroll
          for (int j = 0; j < KC; ++j)
            if (j < kcount) {
              const v4i32 a0 = as[j * 64], a1 = as[(KC + j) * 64];
#pragma unroll
              for (int t = 0; t < 4; ++t) {
                const v4i32 bt = bs[(t * KC + j) * 64];
                acc[im][t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a0, bt, acc[im][t], 0, 0, 0);
                acc[im][4 + t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a1, bt, acc[im][4 + t], 0, 0, 0);
              }
            }
        }
        if (im == 1 && ch == n_chunks - 1 && !(knock & 1)) {
          // fold this modulus: r = (t+ +- t-) mod m, exact in fp64 whatever K; sums of r theta_hi are exact (39-bit weights)
          const double m = rns_mod[l].m, inv = rns_mod[l].inv;
          const double xh = rns_mod[l].xh, xl = rns_mod[l].xl, yh = rns_mod[l].yh, yl = rns_mod[l].yl;
#pragma unroll
          for (int e = 0; e < 32; ++e) {
            const int tp = acc[0][e >> 2][e & 3], tm = acc[1][e >> 2][e & 3];
            const double ts = (double)(tp + tm), td = (double)(tp - tm);
            const double rs = __builtin_fma(-m, __builtin_rint(ts * inv), ts), rd = __builtin_fma(-m, __builtin_rint(td * inv), td);
            F[e].xh = __builtin_fma(rs, xh, F[e].xh);
            F[e].xl = __builtin_fma(rs, xl, F[e].xl);
            F[e].yh = __builtin_fma(rd, yh, F[e].yh);
            F[e].yl = __builtin_fma(rd, yl, F[e].yl);
          }
#pragma unroll
          for (int t = 0; t < 8; ++t) acc[0][t] = acc[1][t] = v4i32{0, 0, 0, 0};
        }
        // stage s + 1 complete (later stages may stay in flight), every wave done reading stage s
        if (AHEAD > 1 && s + AHEAD < n_stages && !(knock & 4))
          asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * KC * (AHEAD - 1)) : "memory");
        else
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
      }
    }
  }

  // reconstruction: X_C = P frac, frac = (S_hi - rint(S_hi)) + S_lo brought into [-1/2, 1/2], in double-double; then the scales
  // of the row and of the column, and the affine map of the plain product's epilogue
  const int fi = lane & 15, fq = lane >> 4;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int col = bn * 128 + wn * 64 + j * 16 + fi;
    if (col >= N) continue;
    const double binv = b_inv[col];
    double off_r = 0.0, off_i = 0.0, sc_r = 1.0, sc_i = 1.0;
    if (col_off) off_r = col_off[2 * col], off_i = col_off[2 * col + 1];
    if (col_scale) sc_r = col_scale[2 * col], sc_i = col_scale[2 * col + 1];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const long long row = (long long)bm * 64 + wm * 32 + i * 16 + 4 * fq + rr;
        if (row >= M) continue;
        const RnsFold& G = F[(4 * i + j) * 4 + rr];
        double out[2];
#pragma unroll
        for (int part = 0; part < 2; ++part) {
          const double sh = part ? G.yh : G.xh, sl = part ? G.yl : G.xl;
          double fh = sh - __builtin_rint(sh);
          fh -= __builtin_rint(fh + sl);
          const double h = RNS_P_HI * fh;
          const double err = __builtin_fma(RNS_P_HI, fh, -h);
          out[part] = h + (err + __builtin_fma(RNS_P_HI, sl, RNS_P_LO * fh));
        }
        const double ainv = a_inv[row];
        double2 v;
        v.x = ((out[0] * binv) * ainv - off_r) * sc_r;
        v.y = ((out[1] * binv) * ainv - off_i) * sc_i;
        *reinterpret_cast<double2*>(C + row * ldc + 2LL * col) = v;
      }
    }
  }
}

// C[M x N] = (A . B - col_off) * col_scale from the residue planes of A (M rows) and of B transposed (N rows), K entries each
int rns_supertile_rows_log2 = -1, rns_knock = 0, rns_nbuf = 2;  // (measurement hooks of tools/rns_product_check)
template <int KC, int NBUF>
static hipError_t launch_zgemm_rns_kc(hipStream_t stream, const int8_t* Ap, const double* a_inv, const int8_t* Bp, const double* b_inv, double* C, long long ldc,
                                      long long M, int N, int K, const double* col_off, const double* col_scale) {
  const int nbm = (int)((M + 63) / 64), nbn = (N + 127) / 128;
  const int st_rows_log2 = rns_supertile_rows_log2 >= 0 ? rns_supertile_rows_log2 : 6;
  const int sr = 1 << st_rows_log2, sc = 64 >> st_rows_log2;
  const long long n_super = (long long)((nbm + sr - 1) / sr) * ((nbn + sc - 1) / sc);
  const long long grid = ((n_super + 7) / 8) * 8 * 64;
  const size_t lds = (size_t)NBUF * 12 * KC * 1024;
  hipError_t e = allow_dynamic_lds((const void*)zgemm_rns_kernel<KC, NBUF>);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL((zgemm_rns_kernel<KC, NBUF>), dim3((unsigned)grid), dim3(256), lds, stream, Ap, (long long)rns_plane_bytes(M, K), Bp,
                     (long long)rns_plane_bytes(N, K), rns_ksteps(K), a_inv, b_inv, C, ldc, M, N, nbm, nbn, st_rows_log2, col_off, col_scale, rns_knock);
  return hipGetLastError();
}
hipError_t launch_zgemm_rns(hipStream_t stream, const int8_t* Ap, const double* a_inv, const int8_t* Bp, const double* b_inv, double* C, long long ldc,
                            long long M, int N, int K, const double* col_off, const double* col_scale) {
  if (M <= 0 || N <= 0) return hipSuccess;
  const int nks = rns_ksteps(K);
  if (rns_nbuf == 3) {  // three buffers of <= 4 k steps (144 KB)
    const int per = (nks + 3) / 4, kc = (nks + per - 1) / per;
    switch (kc) {
      case 1: return launch_zgemm_rns_kc<1, 3>(stream, Ap, a_inv, Bp, b_inv, C, ldc, M, N, K, col_off, col_scale);
      case 2: return launch_zgemm_rns_kc<2, 3>(stream, Ap, a_inv, Bp, b_inv, C, ldc, M, N, K, col_off, col_scale);
      case 3: return launch_zgemm_rns_kc<3, 3>(stream, Ap, a_inv, Bp, b_inv, C, ldc, M, N, K, col_off, col_scale);
      default: return launch_zgemm_rns_kc<4, 3>(stream, Ap, a_inv, Bp, b_inv, C, ldc, M, N, K, col_off, col_scale);
    }
  }
  const int per = (nks + 4) / 5, kc = (nks + per - 1) / per;  // two buffers of <= 5 k steps (120 KB)
  switch (kc) {
    case 1: return launch_zgemm_rns_kc<1, 2>(stream, Ap, a_inv, Bp, b_inv, C, ldc, M, N, K, col_off, col_scale);
    case 2: return launch_zgemm_rns_kc<2, 2>(stream, Ap, a_inv, Bp, b_inv, C, ldc, M, N, K, col_off, col_scale);
    case 3: return launch_zgemm_rns_kc<3, 2>(stream, Ap, a_inv, Bp, b_inv, C, ldc, M, N, K, col_off, col_scale);
    case 4: return launch_zgemm_rns_kc<4, 2>(stream, Ap, a_inv, Bp, b_inv, C, ldc, M, N, K, col_off, col_scale);
    default: return launch_zgemm_rns_kc<5, 2>(stream, Ap, a_inv, Bp, b_inv, C, ldc, M, N, K, col_off, col_scale);
  }
}

}  // namespace bms
