// Separable SWSH analysis on the equiangular grid (spinsfast.map2salm, scri/waveform_grid.py:303-307,
// scri/asymptotic_bondi_data/transformations.py:419-429).
//
// The Huffenberger-Wandelt analysis collapses exactly (oracle/spinsfast_ref.py, tests/test_oracle_wigner.py) to
//     a_lm = sum_j [q_j / n_phi] sLambda_lm(theta_j) F_m(theta_j),      F_m(theta_j) = sum_k f_jk exp(-i m phi_k)
// with real theta-quadrature weights q_j.  Step 1 (phi-DFT) is a thin real GEMM [rows n_theta x 2 n_phi] . [2 n_phi x
// 2 (2L+1)] against the matrix built by dft_matrix_kernel (run through dgemm_mfma_kernel); step 2 is
// theta_quadrature_kernel below.  Compared with the dense quadrature GEMM this is ~10x less arithmetic and sums
// n_phi and n_theta terms per output instead of 2 n_pix (shorter fp64 chains => smaller rounding error).
#include "wigner.h"
#include "kernels.h"

namespace bms {

// B[2k][2mi] = cos(m phi_k), B[2k][2mi+1] = -sin(m phi_k), B[2k+1][2mi] = sin(m phi_k), B[2k+1][2mi+1] = cos(m phi_k);
// mi = m + L, phi_k = 2 pi k / n_phi; the angle is reduced in integers before the sincos.
__global__ __launch_bounds__(256) void dft_matrix_kernel(int n_phi, int L, double* __restrict__ B, long long ldb) {
  const int id = blockIdx.x * blockDim.x + threadIdx.x;
  const int nm = 2 * L + 1;
  if (id >= n_phi * nm) return;
  const int k = id / nm, mi = id % nm, m = mi - L;
  long long r = ((long long)m * k) % n_phi;
  if (r < 0) r += n_phi;
  double s, c;
  sincospi(2.0 * (double)r / (double)n_phi, &s, &c);
  double* r0 = B + (2LL * k) * ldb + 2 * mi;
  double* r1 = r0 + ldb;
  r0[0] = c;
  r0[1] = -s;
  r1[0] = s;
  r1[1] = c;
}

hipError_t launch_dft_matrix(hipStream_t stream, int n_phi, int L, double* B, long long ldb) {
  const int n = n_phi * (2 * L + 1);
  hipLaunchKernelGGL(dft_matrix_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, n_phi, L, B, ldb);
  return hipGetLastError();
}

// T[o][j] = w_theta[j] * Re sYlm_o(theta_j, phi = 0);  Y: c16[n_theta][n_out] from swsh_kernel<0> at the rotors R(theta_j, 0)
__global__ __launch_bounds__(256) void theta_table_kernel(const double* __restrict__ Y, const double* __restrict__ w_theta,
                                                          int n_theta, int n_out, double* __restrict__ T) {
  const int id = blockIdx.x * blockDim.x + threadIdx.x;
  if (id >= n_theta * n_out) return;
  const int o = id / n_theta, j = id % n_theta;
  T[(long long)o * n_theta + j] = w_theta[j] * Y[((long long)j * n_out + o) * 2];
}

hipError_t launch_theta_table(hipStream_t stream, const double* Y, const double* w_theta, int n_theta, int n_out, double* T) {
  const int n = n_theta * n_out;
  hipLaunchKernelGGL(theta_table_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, Y, w_theta, n_theta, n_out, T);
  return hipGetLastError();
}

// a[t][o] = sum_j T[o][j] F[t][j][m(o)].  Thread <-> output mode o with its T row in registers; the F tile of one time
// step (n_theta x (2L+1) complex, 19.5 kB at cfg3) is staged through LDS; outputs are written 16 B per lane, coalesced.
template <int NT, int MAXT>
__global__ __launch_bounds__(MAXT) void theta_quadrature_kernel(const double* __restrict__ F, long long n_rows, int n_theta,
                                                                int nm /* 2L+1 */, int n_out,
                                                                const int* __restrict__ m_index /* [n_out] */,
                                                                const double* __restrict__ T, double* __restrict__ out,
                                                                long long ldo) {
  extern __shared__ double Fs[];  // [n_theta][2 nm]
  const int o = blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = o < n_out;
  double tj[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) tj[j] = (live && j < n_theta) ? T[(long long)o * n_theta + j] : 0.0;
  const int mi = live ? m_index[o] : 0;
  const int tile = n_theta * 2 * nm;  // doubles
  for (long long t = blockIdx.y; t < n_rows; t += gridDim.y) {
    const double* src = F + t * tile;
    __syncthreads();  // previous tile fully consumed
    for (int e = threadIdx.x * 2; e < tile; e += blockDim.x * 2)
      *reinterpret_cast<double2*>(Fs + e) = *reinterpret_cast<const double2*>(src + e);
    __syncthreads();
    double ar = 0.0, ai = 0.0;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      if (j < n_theta) {
        const double2 f = *reinterpret_cast<const double2*>(Fs + (j * nm + mi) * 2);
        ar = fma(tj[j], f.x, ar);
        ai = fma(tj[j], f.y, ai);
      }
    }
    if (live) *reinterpret_cast<double2*>(out + t * ldo + 2LL * o) = double2{ar, ai};
  }
}

hipError_t launch_theta_quadrature(hipStream_t stream, const double* F, long long n_rows, int n_theta, int nm, int n_out,
                                   const int* m_index, const double* T, double* out, long long ldo) {
  if (n_rows <= 0 || n_out <= 0) return hipSuccess;
  const size_t lds = sizeof(double) * (size_t)n_theta * 2 * nm;
  const long long by = n_rows < 2048 ? n_rows : 2048;
  // the T row lives in registers: NT doubles per thread, so larger grids get smaller workgroups
#define LAUNCH_TQ(NT, MAXT)                                                                                              \
  {                                                                                                                      \
    int threads = ((n_out + 63) / 64) * 64;                                                                              \
    if (threads > MAXT) threads = MAXT;                                                                                  \
    const int bx = (n_out + threads - 1) / threads;                                                                      \
    hipError_t e = hipFuncSetAttribute((const void*)theta_quadrature_kernel<NT, MAXT>,                                   \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                            \
    if (e != hipSuccess) return e;                                                                                       \
    hipLaunchKernelGGL((theta_quadrature_kernel<NT, MAXT>), dim3(bx, (unsigned)by), dim3(threads), lds, stream, F, n_rows,  \
                       n_theta, nm, n_out, m_index, T, out, ldo);                                                        \
  }
  if (n_theta <= 24)
    LAUNCH_TQ(24, 1024)
  else if (n_theta <= 40)
    LAUNCH_TQ(40, 1024)
  else if (n_theta <= 72)
    LAUNCH_TQ(72, 512)
  else if (n_theta <= 104)
    LAUNCH_TQ(104, 256)
  else
    return hipErrorInvalidValue;
#undef LAUNCH_TQ
  return hipGetLastError();
}



// =====================================================================================================================
// Fused analysis: one workgroup per time step, the grid row never leaves the chip between the two steps.
//
//   step 1 (MFMA):  the phi-DFT with the cos/sin split.  The grid is de-interleaved into real rows (2j = Re ring j,
//                   2j+1 = Im ring j; A operand [2 n_theta x n_phi]) and multiplied by the real matrix
//                   D[k][m] = cos(m phi_k) (m = 0..L), D[k][L+m] = sin(m phi_k) (m = 1..L)  ([n_phi x (2L+1)]):
//                   C_m[r] = sum_k G[r][k] cos(m phi_k),  S_m[r] = sum_k G[r][k] sin(m phi_k).
//                   Both signs of m come from one product:  F_{+-m}(j) = (C[2j] +- S[2j+1]) + i (C[2j+1] -+ S[2j]).
//                   This is 2.7x less MFMA work than the interleaved complex form (no 66 -> 128 column padding, real
//                   twiddles) and sums n_phi terms per output.
//   step 2 (VALU):  a_o = sum_j T[o][j] F_{m(o)}(j) with the T row of thread o in registers (as theta_quadrature_kernel).
// HBM traffic = the algorithmic minimum: read the grid row once (16 n_pix B), write the modes once (16 n_out B).
// =====================================================================================================================
typedef double v4d_t __attribute__((ext_vector_type(4)));

// D[k][c]: c in [0, L] -> cos(c phi_k); c in [L+1, 2L] -> sin((c-L) phi_k); zero padded to [kpad][pd]
__global__ __launch_bounds__(256) void dft_cs_matrix_kernel(int n_phi, int L, double* __restrict__ D, int pd) {
  const int id = blockIdx.x * blockDim.x + threadIdx.x;
  const int nc = 2 * L + 1;
  if (id >= n_phi * nc) return;
  const int k = id / nc, c = id % nc;
  const int m = c <= L ? c : c - L;
  const long long r = ((long long)m * k) % n_phi;
  double s, co;
  sincospi(2.0 * (double)r / (double)n_phi, &s, &co);
  D[(long long)k * pd + c] = c <= L ? co : s;
}

struct FusedGeom {
  int n_theta, n_phi, L, n_out;
  int mt, ks;  // 16-row tiles along the 2 n_theta rows; k-steps of 4 along n_phi
  int pd;      // LDS pitch (doubles) of the DFT matrix rows
};

// NT: max n_theta (T row length held in registers); PA: LDS pitch of the A rows (2 x odd, >= 4 ks);
// NTN: 16-column tiles of the 2L+1 cos|sin columns; EPT: grid pixels per thread.
template <int NT, int PA, int NTN, int EPT>
__global__ __launch_bounds__(512) void analysis_fused_kernel(const double* __restrict__ G, long long ldg, long long n_rows,
                                                             FusedGeom g, const int* __restrict__ m_index,
                                                             const double* __restrict__ T, const double* __restrict__ Dg,
                                                             double* __restrict__ out, long long ldo) {
  constexpr int PC = 16 * NTN + 2;  // LDS pitch of the C|S rows: compile-time so that step-2 reads use immediate offsets
  extern __shared__ double lds[];
  double* Gs = lds;                    // [16 mt][PA]
  double* Cs = Gs + 16 * g.mt * PA;    // [16 mt][PC]
  double* Ds = Cs + 16 * g.mt * PC;    // [4 ks][pd]
  const int tid = threadIdx.x, nthreads = blockDim.x;
  const int wave = tid >> 6, lane = tid & 63, nwaves = nthreads >> 6;
  const int n_pix = g.n_theta * g.n_phi;

  // one-time set-up: zero the A operand (its padding stays zero), copy the DFT matrix, T row to registers
  for (int e = tid; e < 16 * g.mt * PA; e += nthreads) Gs[e] = 0.0;
  for (int e = tid; e < 4 * g.ks * g.pd; e += nthreads) Ds[e] = Dg[e];
  const int o = tid;
  const bool live = o < g.n_out;
  double tj[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) tj[j] = (live && j < g.n_theta) ? T[(long long)o * g.n_theta + j] : 0.0;
  const int m = live ? m_index[o] - g.L : 0;
  const int am = m < 0 ? -m : m;
  const double s_on = m == 0 ? 0.0 : (m < 0 ? -1.0 : 1.0);
  const double* rc = Cs + am;          // C_|m| column
  const double* rs = Cs + g.L + am;    // S_|m| column (multiplied by 0 for m = 0)

  // LDS slots of this thread's pixels in the de-interleaved A operand, and the prefetch registers
  int slot[EPT];
  double2 pre[EPT];
#pragma unroll
  for (int q = 0; q < EPT; ++q) {
    const int e = tid + q * nthreads;
    const int j = e / g.n_phi, k = e - j * g.n_phi;
    slot[q] = e < n_pix ? (2 * j) * PA + k : -1;
  }
  long long t = blockIdx.x;
  if (t < n_rows) {
#pragma unroll
    for (int q = 0; q < EPT; ++q)
      if (slot[q] >= 0) pre[q] = *reinterpret_cast<const double2*>(G + t * ldg + 2LL * (tid + q * nthreads));
  }
  __syncthreads();
  for (; t < n_rows; t += gridDim.x) {
    // ---- de-interleave the row into the A operand (row 2j = Re, row 2j+1 = Im of ring j)
#pragma unroll
    for (int q = 0; q < EPT; ++q)
      if (slot[q] >= 0) {
        Gs[slot[q]] = pre[q].x;
        Gs[slot[q] + PA] = pre[q].y;
      }
    __syncthreads();  // A operand complete; every thread has finished step 2 of the previous row
    const long long tn = t + gridDim.x;
    if (tn < n_rows) {
#pragma unroll
      for (int q = 0; q < EPT; ++q)
        if (slot[q] >= 0) pre[q] = *reinterpret_cast<const double2*>(G + tn * ldg + 2LL * (tid + q * nthreads));
    }
    // ---- step 1: one 16-row tile x all NTN column tiles per wave trip: one A fragment feeds NTN independent MFMA chains
    for (int tm = wave; tm < g.mt; tm += nwaves) {
      const double* ap = Gs + (tm * 16 + (lane & 15)) * PA + (lane >> 4);
      const double* bp = Ds + (lane >> 4) * g.pd + (lane & 15);
      v4d_t acc[NTN];
#pragma unroll
      for (int n = 0; n < NTN; ++n) acc[n] = v4d_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll 2
      for (int s = 0; s < g.ks; ++s) {
        const double a = ap[4 * s];
#pragma unroll
        for (int n = 0; n < NTN; ++n) acc[n] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bp[4 * s * g.pd + 16 * n], acc[n], 0, 0, 0);
      }
      double* cp = Cs + (tm * 16 + (lane >> 4)) * PC + (lane & 15);
#pragma unroll
      for (int n = 0; n < NTN; ++n)
#pragma unroll
        for (int r = 0; r < 4; ++r) cp[4 * r * PC + 16 * n] = acc[n][r];
    }
    __syncthreads();  // C|S complete
    // ---- step 2: theta quadrature; F_{+-m}(j) = (C[2j] +- S[2j+1]) + i (C[2j+1] -+ S[2j])
    if (live) {
      double p1 = 0.0, p2 = 0.0, p3 = 0.0, p4 = 0.0;
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        if (j < g.n_theta) {
          p1 = fma(tj[j], rc[2 * j * PC], p1);       // T C[Re]
          p3 = fma(tj[j], rc[(2 * j + 1) * PC], p3);  // T C[Im]
          p4 = fma(tj[j], rs[2 * j * PC], p4);       // T S[Re]
          p2 = fma(tj[j], rs[(2 * j + 1) * PC], p2);  // T S[Im]
        }
      }
      *reinterpret_cast<double2*>(out + t * ldo + 2LL * o) = double2{p1 + s_on * p2, p3 - s_on * p4};
    }
  }
}

static void fused_geometry(int n_theta, int n_phi, int L, int n_out, FusedGeom& g, int& pa, int& ntn, size_t& lds_bytes) {
  g.n_theta = n_theta;
  g.n_phi = n_phi;
  g.L = L;
  g.n_out = n_out;
  g.mt = (2 * n_theta + 15) / 16;
  g.ks = (n_phi + 3) / 4;
  ntn = (2 * L + 1 + 15) / 16;
  pa = n_phi <= 24 ? 26 : 42;  // 2 x odd: conflict-free ds_read_b64 of the A fragment
  int pd = 16 * ntn;
  while ((pd & 31) != 16) ++pd;  // = 16 mod 32: the two 16-lane halves of a B fragment read hit different banks
  g.pd = pd;
  const int pc = 16 * ntn + 2;
  lds_bytes = sizeof(double) * ((size_t)16 * g.mt * pa + (size_t)16 * g.mt * pc + (size_t)4 * g.ks * g.pd);
}

int fused_analysis_supported(int n_theta, int n_phi, int L, int n_out) {
  if (n_theta > 40 || n_phi > 40 || n_out > 512 || 2 * L + 1 > 64) return 0;
  if ((long long)n_theta * n_phi > 4LL * 512) return 0;
  return 1;
}

void fused_pitches(int n_theta, int n_phi, int L, int* ks, int* pd) {
  FusedGeom g;
  int pa, ntn;
  size_t lds;
  fused_geometry(n_theta, n_phi, L, 1, g, pa, ntn, lds);
  *ks = g.ks;
  *pd = g.pd;
}

hipError_t launch_dft_cs_matrix(hipStream_t stream, int n_phi, int L, double* D, int pd) {
  const int n = n_phi * (2 * L + 1);
  hipLaunchKernelGGL(dft_cs_matrix_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, n_phi, L, D, pd);
  return hipGetLastError();
}

template <int NT, int PA, int NTN>
static hipError_t launch_fused_t(hipStream_t stream, dim3 grid, dim3 block, size_t lds, const double* G, long long ldg,
                                 long long n_rows, const FusedGeom& g, const int* m_index, const double* T, const double* D,
                                 double* out, long long ldo) {
  hipError_t e = hipFuncSetAttribute((const void*)analysis_fused_kernel<NT, PA, NTN, 4>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL((analysis_fused_kernel<NT, PA, NTN, 4>), grid, block, lds, stream, G, ldg, n_rows, g, m_index, T, D, out,
                     ldo);
  return hipGetLastError();
}

hipError_t launch_analysis_fused(hipStream_t stream, const double* G, long long ldg, long long n_rows, int n_theta, int n_phi,
                                 int L, int n_out, const int* m_index, const double* T, const double* D, double* out,
                                 long long ldo) {
  if (n_rows <= 0) return hipSuccess;
  FusedGeom g;
  int pa, ntn;
  size_t lds;
  fused_geometry(n_theta, n_phi, L, n_out, g, pa, ntn, lds);
  // enough threads for one output mode each and for at most 4 grid pixels each
  int threads = ((n_out + 63) / 64) * 64;
  const int t_pix = (((n_theta * n_phi + 3) / 4 + 63) / 64) * 64;
  if (t_pix > threads) threads = t_pix;
  const long long blocks = n_rows < 512 ? n_rows : 512;
  const dim3 grid((unsigned)blocks), block(threads);
#define FUSED_CASE(NT, PA)                                                                                     \
  switch (ntn) {                                                                                               \
    case 1: return launch_fused_t<NT, PA, 1>(stream, grid, block, lds, G, ldg, n_rows, g, m_index, T, D, out, ldo); \
    case 2: return launch_fused_t<NT, PA, 2>(stream, grid, block, lds, G, ldg, n_rows, g, m_index, T, D, out, ldo); \
    case 3: return launch_fused_t<NT, PA, 3>(stream, grid, block, lds, G, ldg, n_rows, g, m_index, T, D, out, ldo); \
    case 4: return launch_fused_t<NT, PA, 4>(stream, grid, block, lds, G, ldg, n_rows, g, m_index, T, D, out, ldo); \
    default: return hipErrorInvalidValue;                                                                      \
  }
  if (n_theta <= 24 && n_phi <= 24) {
    FUSED_CASE(24, 26)
  } else {
    FUSED_CASE(40, 42)
  }
#undef FUSED_CASE
}

}  // namespace bms
