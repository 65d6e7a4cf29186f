// Dense fp64 contraction on the CDNA4 matrix cores: the mode x harmonic synthesis
// (np.tensordot at scri/waveform_grid.py:475-484; sf.Modes.evaluate at
// scri/asymptotic_bondi_data/transformations.py:324-334) and the dense quadrature analysis
// (spinsfast.map2salm at scri/waveform_grid.py:303-307) both run through this kernel.
//
// The complex product over interleaved (re, im) data is a *real* GEMM against the 2K x 2N matrices built by
// kernels_swsh.hip, so one kernel, no complex arithmetic, no wasted MFMA work:
//     C[M x N] = (A[M x K] . B[K x N] - col_off[N]) * col_scale[N]          (all fp64, row-major)
//
// v_mfma_f64_16x16x4_f64 (64 cycles / 2048 flop per SIMD => 78.6 TFLOP/s chip peak): per lane one f64 of
// A[i = lane&15][k = lane>>4], one of B[k = lane>>4][j = lane&15], 4 results D[row = (lane>>4) + 4 r][col = lane&15].
// Workgroup tile 128 x 128 x 16, 4 wavefronts (2 x 2), 64 x 64 per wavefront = 16 independent accumulator tiles;
// operands are staged global -> registers -> LDS with a two-deep LDS ring (one barrier per K-step); the LDS
// pitches (A rows 18 doubles, B rows 144 doubles) make every ds_read_b64 fragment read conflict free.
// Blocks are dealt in 8 x 8 super-tiles per XCD so that A slabs and B panels are re-used out of that XCD's L2.
#include <cstdlib>
#include "kernels.h"

namespace bms {

typedef double v4d __attribute__((ext_vector_type(4)));

constexpr int G_BM = 128, G_BN = 128, G_BK = 16;
constexpr int G_LDA = G_BK + 2;    // LDS pitch of an A row (doubles)
constexpr int G_LDB = G_BN + 16;   // LDS pitch of a B row (doubles)
constexpr int G_ASZ = G_BM * G_LDA;
constexpr int G_BSZ = G_BK * G_LDB;

__global__ __launch_bounds__(256, 2) void dgemm_mfma_kernel(const double* __restrict__ A, long long lda,
                                                            const double* __restrict__ B, long long ldb,
                                                            double* __restrict__ C, long long ldc, long long M, int N,
                                                            int K, int nbm, int nbn, int st_rows_log2,
                                                            const double* __restrict__ col_off,
                                                            const double* __restrict__ col_scale) {
  __shared__ __attribute__((aligned(16))) double lds[2 * G_ASZ + 2 * G_BSZ];
  double* As = lds;
  double* Bs = lds + 2 * G_ASZ;

  // XCD-aware block -> tile map.  Blocks b and b+8 share an XCD (round-robin dispatch), and an XCD holds 64 resident
  // workgroups (2 per CU): give each XCD whole super-tiles of 2^r row slabs x 2^(6-r) column panels.  The 64 workgroups of
  // a super-tile walk K together, so every 16-deep slice of its A slabs and B panels is fetched into that XCD's L2 once and
  // re-used from there.  Measured on cfg3 (r = 0..6): 6.13, 6.06, 8.73, 6.11, 6.25, 5.76, 5.85 ms -> r = 5 (32 x 2).
  const int b = blockIdx.x;
  const int xcd = b & 7;
  const int q = b >> 3;
  const int st_cols_log2 = 6 - st_rows_log2;
  const int nsn = (nbn + (1 << st_cols_log2) - 1) >> st_cols_log2;  // super-tile columns
  const int S = (q >> 6) * 8 + xcd;
  const int r = q & 63;
  const int bm = ((S / nsn) << st_rows_log2) + (r >> st_cols_log2);
  const int bn = ((S % nsn) << st_cols_log2) + (r & ((1 << st_cols_log2) - 1));
  if (bm >= nbm || bn >= nbn) return;
  const long long m0 = (long long)bm * G_BM;
  const int n0 = bn * G_BN;

  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int wm = wave >> 1, wn = wave & 1;
  const int fi = lane & 15, fk = lane >> 4;

  v4d acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = v4d{0.0, 0.0, 0.0, 0.0};

  double2 ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3;
  const int nk = (K + G_BK - 1) / G_BK;
  // staging maps: A tile 128 rows x 8 double2 (thread -> row = idx>>3, chunk = idx&7);
  //               B tile 16 rows x 64 double2 (thread -> row = idx>>6, chunk = idx&63)
  const int a_row = tid >> 3, a_ch = tid & 7;   // + 32 rows per pass
  const int b_row = tid >> 6, b_ch = tid & 63;  // + 4 rows per pass
  const double* a_ptr = A + (m0 + a_row) * lda + 2 * a_ch;
  const double* b_ptr = B + (long long)b_row * ldb + n0 + 2 * b_ch;
  const bool a_ok0 = (m0 + a_row) < M, a_ok1 = (m0 + a_row + 32) < M, a_ok2 = (m0 + a_row + 64) < M,
             a_ok3 = (m0 + a_row + 96) < M;
  const double2 zero2 = {0.0, 0.0};

#define G_LOAD_GLOBAL(kt)                                                                              \
  {                                                                                                    \
    const int k0 = (kt)*G_BK;                                                                          \
    const bool kok = (k0 + 2 * a_ch) < K;                                                              \
    ra0 = (a_ok0 && kok) ? *reinterpret_cast<const double2*>(a_ptr + k0) : zero2;                      \
    ra1 = (a_ok1 && kok) ? *reinterpret_cast<const double2*>(a_ptr + 32 * lda + k0) : zero2;           \
    ra2 = (a_ok2 && kok) ? *reinterpret_cast<const double2*>(a_ptr + 64 * lda + k0) : zero2;           \
    ra3 = (a_ok3 && kok) ? *reinterpret_cast<const double2*>(a_ptr + 96 * lda + k0) : zero2;           \
    const double* bp = b_ptr + (long long)k0 * ldb;                                                    \
    rb0 = *reinterpret_cast<const double2*>(bp);                                                       \
    rb1 = *reinterpret_cast<const double2*>(bp + 4 * ldb);                                             \
    rb2 = *reinterpret_cast<const double2*>(bp + 8 * ldb);                                             \
    rb3 = *reinterpret_cast<const double2*>(bp + 12 * ldb);                                            \
  }
#define G_STORE_LDS(buf)                                                                               \
  {                                                                                                    \
    double* as_w = As + (buf)*G_ASZ + a_row * G_LDA + 2 * a_ch;                                        \
    *reinterpret_cast<double2*>(as_w) = ra0;                                                           \
    *reinterpret_cast<double2*>(as_w + 32 * G_LDA) = ra1;                                              \
    *reinterpret_cast<double2*>(as_w + 64 * G_LDA) = ra2;                                              \
    *reinterpret_cast<double2*>(as_w + 96 * G_LDA) = ra3;                                              \
    double* bs_w = Bs + (buf)*G_BSZ + b_row * G_LDB + 2 * b_ch;                                        \
    *reinterpret_cast<double2*>(bs_w) = rb0;                                                           \
    *reinterpret_cast<double2*>(bs_w + 4 * G_LDB) = rb1;                                               \
    *reinterpret_cast<double2*>(bs_w + 8 * G_LDB) = rb2;                                               \
    *reinterpret_cast<double2*>(bs_w + 12 * G_LDB) = rb3;                                              \
  }

  G_LOAD_GLOBAL(0);
  G_STORE_LDS(0);
  __syncthreads();

  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) G_LOAD_GLOBAL(kt + 1);
    const double* as = As + buf * G_ASZ + (wm * 64 + fi) * G_LDA + fk;
    const double* bs = Bs + buf * G_BSZ + fk * G_LDB + wn * 64 + fi;
#pragma unroll
    for (int kk = 0; kk < G_BK / 4; ++kk) {
      double a[4], bb[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = as[i * 16 * G_LDA + kk * 4];
#pragma unroll
      for (int j = 0; j < 4; ++j) bb[j] = bs[kk * 4 * G_LDB + j * 16];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], bb[j], acc[i][j], 0, 0, 0);
    }
    if (kt + 1 < nk) G_STORE_LDS(buf ^ 1);
    __syncthreads();
  }

  // epilogue: affine per-column map (type-specific inhomogeneous term and conformal factor,
  // scri/waveform_grid.py:485-503,559) fused into the store
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int col = n0 + wn * 64 + j * 16 + fi;
    if (col >= N) continue;
    const double off = col_off ? col_off[col] : 0.0;
    const double sc = col_scale ? col_scale[col] : 1.0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const long long row = m0 + wm * 64 + i * 16 + fk + 4 * r;
        if (row < M) C[row * ldc + col] = (acc[i][j][r] - off) * sc;
      }
    }
  }
}

#undef G_LOAD_GLOBAL
#undef G_STORE_LDS

// ------------------------------------------------------------------------------------------------ complex, 3 products
// The synthesis contraction is complex x complex.  Written as the real GEMM above it costs 4 real MFMA products per
// complex multiply-add; the matrix pipe is the bound (60 TFLOP/s sustained), so the complex kernel below spends 3:
//     P1 = Ar.Br,  P2 = Ai.Bi,  P3 = (Ar + Ai).(Br + Bi)   =>   Re = P1 - P2,  Im = P3 - P1 - P2
// (the "3M" scheme; the error bound stays normwise O(K eps |A||B|), the imaginary part just loses the componentwise
// bound, which the parity tolerances never relied on).  The operand sums cost one v_add_f64 per fragment in registers.
//     C[M x N] = (A[M x K] . B[K x N] - col_off) * col_scale      (complex128 interleaved; off/scale per real column)
// Workgroup tile 64 rows x 64 complex columns x 8 complex k, 4 wavefronts (2 x 2), 32 x 32 per wavefront =
// 3 x 4 accumulator tiles (96 VGPRs), three workgroups per CU.  Fragments are (re, im) pairs read with ds_read_b128:
// A rows at a pitch of 10 complex and B rows at 64 complex make every 16-lane group of a read hit 16 distinct slots.
constexpr int Z_BM = 64, Z_BN = 64, Z_KC = 8;
constexpr int Z_PA = 10;  // LDS pitch of an A row (complex)
constexpr int Z_PB = 64;  // LDS pitch of a B row (complex)
constexpr int Z_ASZ = Z_BM * Z_PA;
constexpr int Z_BSZ = Z_KC * Z_PB;

// FOUR = true: the plain 4-product form (Re = Ar.Br - Ai.Bi, Im = Ar.Bi + Ai.Br), componentwise-accurate imaginary parts
// at 4/3 of the matrix work and 2 workgroups per CU.  Kept for A/B measurements only (SCRI_AMD_ZGEMM_4M=1): on the
// reference's exhaustive analytic sweeps both forms sit at the same 1.3-1.7e-14 of its 5e-14 tolerance
// (tools/tolerance_probe.py, profiles/r02_a_tolerance_probe.json) -- the digits were in the harmonics, not here.
template <bool FOUR>
__global__ __launch_bounds__(256, FOUR ? 2 : 3) void zgemm3m_mfma_kernel(const double* __restrict__ A, long long lda,
                                                              const double* __restrict__ B, long long ldb,
                                                              double* __restrict__ C, long long ldc, long long M, int N,
                                                              int K, int nbm, int nbn, int st_rows_log2,
                                                              const double* __restrict__ col_off,
                                                              const double* __restrict__ col_scale) {
  __shared__ __attribute__((aligned(16))) double2 lds[2 * Z_ASZ + 2 * Z_BSZ];
  double2* As = lds;
  double2* Bs = lds + 2 * Z_ASZ;

  // same XCD-aware super-tile map as dgemm_mfma_kernel
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
  const long long m0 = (long long)bm * Z_BM;
  const int n0 = bn * Z_BN;

  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int wm = wave >> 1, wn = wave & 1;
  const int fi = lane & 15, fk = lane >> 4;

  v4d p1[2][2], p2[2][2], p3[2][2], p4[FOUR ? 2 : 1][FOUR ? 2 : 1];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) p1[i][j] = p2[i][j] = p3[i][j] = v4d{0.0, 0.0, 0.0, 0.0};
  if constexpr (FOUR) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) p4[i][j] = v4d{0.0, 0.0, 0.0, 0.0};
  }

  // staging maps: A tile 64 rows x 8 complex (thread -> row = tid>>3 (+32), k = tid&7);
  //               B tile 8 rows x 64 complex (thread -> row = tid>>6 (+4), column = tid&63)
  const int a_row = tid >> 3, a_k = tid & 7;
  const int b_row = tid >> 6, b_col = tid & 63;
  const double* a_ptr = A + (m0 + a_row) * lda + 2 * a_k;
  const double* b_ptr = B + (long long)b_row * ldb + 2 * (n0 + b_col);
  const bool a_ok0 = (m0 + a_row) < M, a_ok1 = (m0 + a_row + 32) < M;
  const double2 zero2 = {0.0, 0.0};
  double2 ra0, ra1, rb0, rb1;
  const int nk = (K + Z_KC - 1) / Z_KC;

#define Z_LOAD_GLOBAL(kt)                                                                      \
  {                                                                                            \
    const int k0 = (kt)*Z_KC;                                                                  \
    const bool kok = (k0 + a_k) < K;                                                           \
    ra0 = (a_ok0 && kok) ? *reinterpret_cast<const double2*>(a_ptr + 2 * k0) : zero2;          \
    ra1 = (a_ok1 && kok) ? *reinterpret_cast<const double2*>(a_ptr + 32 * lda + 2 * k0) : zero2; \
    const double* bp = b_ptr + (long long)k0 * ldb;                                            \
    rb0 = *reinterpret_cast<const double2*>(bp);                                               \
    rb1 = *reinterpret_cast<const double2*>(bp + 4 * ldb);                                     \
  }
#define Z_STORE_LDS(buf)                                        \
  {                                                             \
    double2* as_w = As + (buf)*Z_ASZ + a_row * Z_PA + a_k;      \
    as_w[0] = ra0;                                              \
    as_w[32 * Z_PA] = ra1;                                      \
    double2* bs_w = Bs + (buf)*Z_BSZ + b_row * Z_PB + b_col;    \
    bs_w[0] = rb0;                                              \
    bs_w[4 * Z_PB] = rb1;                                       \
  }

  Z_LOAD_GLOBAL(0);
  Z_STORE_LDS(0);
  __syncthreads();

  // In the last column panel the right-hand waves (columns 32..63 of the tile) may lie entirely beyond N: they keep
  // staging and synchronising but leave the matrix pipe to the other workgroups of the CU (cfg3: 1297 = 20 x 64 + 17).
  const bool wave_has_columns = n0 + wn * 32 < N;
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) Z_LOAD_GLOBAL(kt + 1);
    const double2* as = As + buf * Z_ASZ + (wm * 32 + fi) * Z_PA + fk;
    const double2* bs = Bs + buf * Z_BSZ + fk * Z_PB + wn * 32 + fi;
#pragma unroll
    for (int kk = 0; wave_has_columns && kk < Z_KC / 4; ++kk) {
      const double2 a0 = as[kk * 4], a1 = as[16 * Z_PA + kk * 4];
      const double2 b0 = bs[kk * 4 * Z_PB], b1 = bs[kk * 4 * Z_PB + 16];
      p1[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0.x, b0.x, p1[0][0], 0, 0, 0);
      p1[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0.x, b1.x, p1[0][1], 0, 0, 0);
      p1[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1.x, b0.x, p1[1][0], 0, 0, 0);
      p1[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1.x, b1.x, p1[1][1], 0, 0, 0);
      p2[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0.y, b0.y, p2[0][0], 0, 0, 0);
      p2[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0.y, b1.y, p2[0][1], 0, 0, 0);
      p2[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1.y, b0.y, p2[1][0], 0, 0, 0);
      p2[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1.y, b1.y, p2[1][1], 0, 0, 0);
      if constexpr (FOUR) {
        p3[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0.x, b0.y, p3[0][0], 0, 0, 0);
        p3[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0.x, b1.y, p3[0][1], 0, 0, 0);
        p3[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1.x, b0.y, p3[1][0], 0, 0, 0);
        p3[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1.x, b1.y, p3[1][1], 0, 0, 0);
        p4[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0.y, b0.x, p4[0][0], 0, 0, 0);
        p4[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0.y, b1.x, p4[0][1], 0, 0, 0);
        p4[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1.y, b0.x, p4[1][0], 0, 0, 0);
        p4[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1.y, b1.x, p4[1][1], 0, 0, 0);
      } else {
        const double sa0 = a0.x + a0.y, sa1 = a1.x + a1.y, sb0 = b0.x + b0.y, sb1 = b1.x + b1.y;
        p3[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(sa0, sb0, p3[0][0], 0, 0, 0);
        p3[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(sa0, sb1, p3[0][1], 0, 0, 0);
        p3[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(sa1, sb0, p3[1][0], 0, 0, 0);
        p3[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(sa1, sb1, p3[1][1], 0, 0, 0);
      }
    }
    if (kt + 1 < nk) Z_STORE_LDS(buf ^ 1);
    __syncthreads();
  }

  // epilogue: recombine, then the affine per-column map (type-specific inhomogeneous term and conformal factor,
  // scri/waveform_grid.py:485-503,559) fused into the 16-byte store
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int col = n0 + wn * 32 + j * 16 + fi;
    if (col >= N) continue;
    const double off_r = col_off ? col_off[2 * col] : 0.0, off_i = col_off ? col_off[2 * col + 1] : 0.0;
    const double sc_r = col_scale ? col_scale[2 * col] : 1.0, sc_i = col_scale ? col_scale[2 * col + 1] : 1.0;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const long long row = m0 + wm * 32 + i * 16 + fk + 4 * rr;
        if (row < M) {
          const double re = p1[i][j][rr] - p2[i][j][rr];
          double im;
          if constexpr (FOUR) im = p3[i][j][rr] + p4[i][j][rr];
          else im = (p3[i][j][rr] - p1[i][j][rr]) - p2[i][j][rr];
          double2 v;
          v.x = (re - off_r) * sc_r;
          v.y = (im - off_i) * sc_i;
          *reinterpret_cast<double2*>(C + row * ldc + 2LL * col) = v;
        }
      }
    }
  }
}

#undef Z_LOAD_GLOBAL
#undef Z_STORE_LDS

hipError_t launch_zgemm3m(hipStream_t stream, const double* A, long long lda, const double* B, long long ldb, double* C,
                          long long ldc, long long M, int N, int K, const double* col_off, const double* col_scale) {
  if (M <= 0 || N <= 0) return hipSuccess;
  const int nbm = (int)((M + Z_BM - 1) / Z_BM);
  const int nbn = (N + Z_BN - 1) / Z_BN;
  // Super-tile of 64 workgroup tiles per XCD turn: 64 row tiles x 1 column tile for tall launches (every workgroup of the
  // XCD shares one 64-column slice of B, which then lives in L2/L1 while A streams: 4.07 against 4.18 ms at cfg3, 1563 row
  // tiles), 32 x 2 otherwise (cfg5's chunks are ~130 row tiles: 64-row super-tiles would leave every third one nearly
  // empty; 94.6 against 96.6 ms).  SCRI_AMD_ZGEMM_ST_ROWS_LOG2 overrides.
  static const int st_env = BMS_PROBE_ENV("SCRI_AMD_ZGEMM_ST_ROWS_LOG2") ? atoi(BMS_PROBE_ENV("SCRI_AMD_ZGEMM_ST_ROWS_LOG2")) : -1;
  const int st_rows_log2 = (st_env >= 0 && st_env <= 6) ? st_env : (nbm >= 512 ? 6 : 5);
  const int sr = 1 << st_rows_log2, sc = 64 >> st_rows_log2;
  const long long n_super = (long long)((nbm + sr - 1) / sr) * ((nbn + sc - 1) / sc);
  const long long grid = ((n_super + 7) / 8) * 8 * 64;
  static const bool four = BMS_PROBE_ENV("SCRI_AMD_ZGEMM_4M") && atoi(BMS_PROBE_ENV("SCRI_AMD_ZGEMM_4M")) != 0;
  if (four)
    hipLaunchKernelGGL(zgemm3m_mfma_kernel<true>, dim3((unsigned)grid), dim3(256), 0, stream, A, lda, B, ldb, C, ldc, M, N,
                       K, nbm, nbn, st_rows_log2, col_off, col_scale);
  else
    hipLaunchKernelGGL(zgemm3m_mfma_kernel<false>, dim3((unsigned)grid), dim3(256), 0, stream, A, lda, B, ldb, C, ldc, M, N,
                       K, nbm, nbn, st_rows_log2, col_off, col_scale);
  return hipGetLastError();
}

hipError_t launch_dgemm(hipStream_t stream, const double* A, long long lda, const double* B, long long ldb, double* C,
                        long long ldc, long long M, int N, int K, const double* col_off, const double* col_scale) {
  if (M <= 0 || N <= 0) return hipSuccess;
  const int nbm = (int)((M + G_BM - 1) / G_BM);
  const int nbn = (N + G_BN - 1) / G_BN;
  static const int st_rows_log2 = BMS_PROBE_ENV("SCRI_AMD_GEMM_ST_ROWS_LOG2") ? atoi(BMS_PROBE_ENV("SCRI_AMD_GEMM_ST_ROWS_LOG2")) : 5;
  const int sr = 1 << st_rows_log2, sc = 64 >> st_rows_log2;
  const long long n_super = (long long)((nbm + sr - 1) / sr) * ((nbn + sc - 1) / sc);
  const long long grid = ((n_super + 7) / 8) * 8 * 64;
  hipLaunchKernelGGL(dgemm_mfma_kernel, dim3((unsigned)grid), dim3(256), 0, stream, A, lda, B, ldb, C, ldc, M, N, K, nbm,
                     nbn, st_rows_log2, col_off, col_scale);
  return hipGetLastError();
}

}  // namespace bms
