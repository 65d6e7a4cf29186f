// Mode-space operators of spherical_functions.Modes / scri.ModesTimeSeries on series resident in HBM
// (scri/modes_time_series.py:7-139 and the sf.Modes algebra it inherits: eth, ethbar, bar, real, imag, +, -, scalar
// products, truncate_ell; used by scri/asymptotic_bondi_data/bms_charges.py:14-286 and map_to_superrest_frame.py).
// All of them are the same map along the mode axis,
//     out[t][j] = r_t ( ca_j op_a(A[t][ia_j]) + cb_j op_b(B[t][ib_j]) ),      op = identity or complex conjugation,
// with per-column tables (ia, ca, ib, cb) the host derives from (l, m, s): eth = diagonal factor, bar = the permutation
// (l, m) -> (l, -m) with a sign and a conjugation, real = (a + bar a) / 2, a sum of series with different l ranges = two
// index maps with -1 for the modes one side lacks.  One pass over the data, lanes along the mode axis (16-byte accesses;
// the gathers stay inside the row a wave is reading anyway), r_t an optional per-row real factor (the `t *` of the boost
// charge, bms_charges.py:163-182).  HBM-bound: 16 (1 or 2) n_cols bytes read + 16 n_cols written per row.
#include "wigner.h"
#include "kernels.h"

namespace bms {

__global__ __launch_bounds__(256) void mode_map_kernel(double* __restrict__ out, long long ld_out, long long n_rows, int n_cols,
                                                       ModeMapSide A, ModeMapSide B, const double* __restrict__ row_scale) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n_cols) return;
  const int ia = A.idx[j];
  const cplx ca = {A.coef[2 * j], A.coef[2 * j + 1]};
  const bool has_b = B.data != nullptr;
  const int ib = has_b ? B.idx[j] : -1;
  const cplx cb = has_b ? cplx{B.coef[2 * j], B.coef[2 * j + 1]} : cplx{0.0, 0.0};
  for (long long t = blockIdx.y; t < n_rows; t += gridDim.y) {
    cplx v = {0.0, 0.0};
    if (ia >= 0) {
      const double2 a = *reinterpret_cast<const double2*>(A.data + (t * A.ld + ia) * 2);
      v = cmul(ca, cplx{a.x, A.conj ? -a.y : a.y});
    }
    if (ib >= 0) {
      const double2 b = *reinterpret_cast<const double2*>(B.data + (t * B.ld + ib) * 2);
      const cplx w = cmul(cb, cplx{b.x, B.conj ? -b.y : b.y});
      v.re += w.re;
      v.im += w.im;
    }
    if (row_scale) {
      const double r = row_scale[t];
      v.re *= r;
      v.im *= r;
    }
    *reinterpret_cast<double2*>(out + (t * ld_out + j) * 2) = double2{v.re, v.im};
  }
}

// WaveformBase.norm (scri/waveform_base.py:19-35,535-551): s_t = sum_j (re^2 + im^2) accumulated in column order, one term at a time
// and without fused multiply-adds -- the reference's loop, so that the sums (and the parity-violation measures built on them)
// come out to the bit.  One wave per 64 rows: 64 x 16 tiles travel through LDS with 256-byte row segments, every lane then adds the
// 16 terms of its own row in order.
__global__ __launch_bounds__(64) void row_norm_kernel(const double* __restrict__ data, long long ld, long long n_rows, int n_cols,
                                                      int take_sqrt, double* __restrict__ out) {
#pragma clang fp contract(off)
  __shared__ double2 tile[64][17];
  const int lane = threadIdx.x, sub = lane >> 4, col = lane & 15;
  for (long long r0 = 64LL * blockIdx.x; r0 < n_rows; r0 += 64LL * gridDim.x) {
    double s = 0.0;
    for (int c0 = 0; c0 < n_cols; c0 += 16) {
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        const long long r = r0 + 4 * k + sub;
        tile[4 * k + sub][col] = (r < n_rows && c0 + col < n_cols) ? *reinterpret_cast<const double2*>(data + (r * ld + c0 + col) * 2)
                                                                   : double2{0.0, 0.0};
      }
      __syncthreads();
      const int nc = n_cols - c0 < 16 ? n_cols - c0 : 16;
      for (int j = 0; j < nc; ++j) {
        const double2 v = tile[lane][j];
        const double a = v.x * v.x, b = v.y * v.y;
        s += a + b;
      }
      __syncthreads();
    }
    if (r0 + lane < n_rows) out[r0 + lane] = take_sqrt ? sqrt(s) : s;
  }
}

hipError_t launch_row_norm(hipStream_t stream, const double* data, long long ld, long long n_rows, int n_cols, int take_sqrt, double* out) {
  if (n_rows <= 0) return hipSuccess;
  const long long blocks = (n_rows + 63) / 64;
  hipLaunchKernelGGL(row_norm_kernel, dim3((unsigned)(blocks < 65536 ? blocks : 65536)), dim3(64), 0, stream, data, ld, n_rows, n_cols, take_sqrt, out);
  return hipGetLastError();
}

hipError_t launch_mode_map(hipStream_t stream, double* out, long long ld_out, long long n_rows, int n_cols, const ModeMapSide& A,
                           const ModeMapSide& B, const double* row_scale) {
  if (n_rows <= 0 || n_cols <= 0) return hipSuccess;
  dim3 grid((n_cols + 255) / 256, (unsigned)(n_rows < 8192 ? n_rows : 8192));
  hipLaunchKernelGGL(mode_map_kernel, grid, dim3(256), 0, stream, out, ld_out, n_rows, n_cols, A, B, row_scale);
  return hipGetLastError();
}

// ---- several series in one row (trailing data dimensions, scri/waveform_grid.py:299-308): the reference stores the trailing index
// FASTEST (element (t, mode, j) at (t n_modes + mode) F + j); the engine wants every series as a block of unit-stride columns.  Both
// conversions are plain permutations at HBM rate, so the host never makes a strided copy.
// in: c16[n_rows][ld_in] with (mode, j) at column mode * F + j   ->   out: c16[n_rows][F * n_modes] with series j in columns [j n_modes, ...)
__global__ __launch_bounds__(256) void series_to_blocks_kernel(const double2* __restrict__ in, long long ld_in, double2* __restrict__ out,
                                                               long long n_rows, int n_modes, int F) {
  const long long per_row = (long long)n_modes * F, total = n_rows * per_row;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += gridDim.x * 256LL) {
    const long long t = i / per_row;
    const int c = (int)(i - t * per_row), j = c / n_modes, m = c - j * n_modes;
    out[i] = in[t * ld_in + (long long)m * F + j];
  }
}
// in: c16[F][block_rows][n_cols] (series-major results, the first n_rows rows of each block valid)   ->   out: c16[n_rows][n_cols * F]
// with (col, j) at column col * F + j
__global__ __launch_bounds__(256) void blocks_to_series_kernel(const double2* __restrict__ in, long long block_rows, double2* __restrict__ out,
                                                               long long n_rows, int n_cols, int F) {
  const long long per_row = (long long)n_cols * F, total = n_rows * per_row;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += gridDim.x * 256LL) {
    const long long t = i / per_row;
    const int c = (int)(i - t * per_row), col = c / F, j = c - col * F;
    out[i] = in[((long long)j * block_rows + t) * n_cols + col];
  }
}

hipError_t launch_series_to_blocks(hipStream_t stream, const double* in, long long ld_in, double* out, long long n_rows, int n_modes, int F) {
  const long long total = n_rows * n_modes * F;
  if (total <= 0) return hipSuccess;
  const long long blocks = std::min<long long>((total + 255) / 256, 16384);
  hipLaunchKernelGGL(series_to_blocks_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, (const double2*)in, ld_in, (double2*)out, n_rows, n_modes, F);
  return hipGetLastError();
}
hipError_t launch_blocks_to_series(hipStream_t stream, const double* in, long long block_rows, double* out, long long n_rows, int n_cols, int F) {
  const long long total = n_rows * n_cols * F;
  if (total <= 0) return hipSuccess;
  const long long blocks = std::min<long long>((total + 255) / 256, 16384);
  hipLaunchKernelGGL(blocks_to_series_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, (const double2*)in, block_rows, (double2*)out, n_rows, n_cols, F);
  return hipGetLastError();
}

}  // namespace bms
