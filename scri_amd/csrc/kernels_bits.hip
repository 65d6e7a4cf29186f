// Bit transforms of the storage formats (scri/utilities.py:194-406): XOR differencing of a time
// series, the "multi-shuffle" bit transposition and the Fletcher-32 checksum.  Integer / byte work, HBM-bound, bit-exact.
#include <cstdint>
#include "kernels.h"

namespace bms {

// ------------------------------------------------------------------------------------------------ xor differencing
// forward (utilities.py:195-217): out[i] = in[i-1] ^ in[i] for i >= 1, out[0] = in[0]; rows of n_cols 64-bit words
__global__ __launch_bounds__(256) void xor_forward_kernel(const uint64_t* __restrict__ in, uint64_t* __restrict__ out,
                                                          long long n_rows, long long n_cols) {
  const long long total = n_rows * n_cols;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x)
    out[e] = e >= n_cols ? in[e] ^ in[e - n_cols] : in[e];
}

// reverse (utilities.py:220-232): running XOR down the rows.  Three phases over tiles of rows: tile totals, exclusive scan
// of the totals, running XOR inside each tile started from its carry.  Lanes across columns (coalesced 8-byte words).
__global__ __launch_bounds__(256) void xor_reverse_kernel(const uint64_t* __restrict__ in, uint64_t* __restrict__ out,
                                                          uint64_t* __restrict__ carry, long long n_rows, long long n_cols,
                                                          int tile, int phase) {
  const long long col = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (col >= n_cols) return;
  const long long n_tiles = (n_rows + tile - 1) / tile;
  if (phase == 2) {
    if (blockIdx.y != 0) return;
    uint64_t run = 0;
    for (long long t = 0; t < n_tiles; ++t) {
      const uint64_t v = carry[t * n_cols + col];
      carry[t * n_cols + col] = run;
      run ^= v;
    }
    return;
  }
  const long long r0 = (long long)blockIdx.y * tile;
  long long r1 = r0 + tile;
  if (r1 > n_rows) r1 = n_rows;
  uint64_t run = phase == 3 ? carry[blockIdx.y * n_cols + col] : 0;
  for (long long r = r0; r < r1; ++r) {
    run ^= in[r * n_cols + col];
    if (phase == 3) out[r * n_cols + col] = run;
  }
  if (phase == 1) carry[blockIdx.y * n_cols + col] = run;
}

hipError_t launch_xor_timeseries(hipStream_t stream, const void* in, void* out, void* carry, long long n_rows, long long n_cols,
                                 int reverse) {
  if (n_rows <= 0 || n_cols <= 0) return hipSuccess;
  if (!reverse) {
    const long long blocks = (n_rows * n_cols + 255) / 256;
    hipLaunchKernelGGL(xor_forward_kernel, dim3((unsigned)(blocks < 65536 ? blocks : 65536)), dim3(256), 0, stream,
                       (const uint64_t*)in, (uint64_t*)out, n_rows, n_cols);
    return hipGetLastError();
  }
  const int tile = 256;
  const long long n_tiles = (n_rows + tile - 1) / tile;
  const dim3 grid((unsigned)((n_cols + 255) / 256), (unsigned)n_tiles), one((unsigned)((n_cols + 255) / 256), 1);
  for (int phase = 1; phase <= 3; ++phase)
    hipLaunchKernelGGL(xor_reverse_kernel, phase == 2 ? one : grid, dim3(256), 0, stream, (const uint64_t*)in, (uint64_t*)out,
                       (uint64_t*)carry, n_rows, n_cols, tile, phase);
  return hipGetLastError();
}
long long xor_carry_words(long long n_rows, long long n_cols) { return ((n_rows + 255) / 256) * n_cols; }

// ------------------------------------------------------------------------------------------------ multi-shuffle
// The n elements of W bits are cut into pieces (widths listed from the most significant end); the output is the bit
// stream "piece k of element 0, of element 1, ..., of element n-1" for k from the LEAST significant piece upwards
// (utilities.py:271-406).  forward: one thread per output word gathers the W bits of its slot; reverse: one thread per
// element gathers its pieces back.
struct ShufflePieces {
  int n;             // number of pieces
  int width[64];     // piece widths, least significant piece first
  int shift[64];     // bit position of the piece inside an element
  long long off[65];  // first bit of the piece's section in the stream
};

template <typename T>
__global__ __launch_bounds__(256) void multishuffle_kernel(const T* __restrict__ a, T* __restrict__ b, long long n, ShufflePieces S) {
  constexpr int W = 8 * sizeof(T);
  for (long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (long long)gridDim.x * blockDim.x) {
    long long P = j * W;
    int i = 0;
    while (i + 1 < S.n && S.off[i + 1] <= P) ++i;
    uint64_t val = 0;
    int filled = 0;
    while (filled < W) {
      const int w = S.width[i];
      const long long L = P - S.off[i];
      const long long e = L / w;
      const int r = (int)(L - e * w);
      int k = w - r;
      if (k > W - filled) k = W - filled;
      const uint64_t bits = ((uint64_t)a[e] >> (S.shift[i] + r)) & (k == 64 ? ~0ull : ((1ull << k) - 1));
      val |= bits << filled;
      filled += k;
      P += k;
      if (P == S.off[i + 1]) ++i;
    }
    b[j] = (T)val;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void multiunshuffle_kernel(const T* __restrict__ b, T* __restrict__ a, long long n, ShufflePieces S) {
  constexpr int W = 8 * sizeof(T);
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long long)gridDim.x * blockDim.x) {
    uint64_t val = 0;
    for (int i = 0; i < S.n; ++i) {
      const int w = S.width[i];
      const long long P = S.off[i] + e * w;
      const long long word = P / W;
      const int be = (int)(P - word * W);
      const uint64_t mask = w == 64 ? ~0ull : ((1ull << w) - 1);
      uint64_t piece = ((uint64_t)b[word] >> be) & mask;
      if (be + w > W) piece |= ((uint64_t)b[word + 1] << (W - be)) & mask;
      val |= piece << S.shift[i];
    }
    a[e] = (T)val;
  }
}

hipError_t launch_multishuffle(hipStream_t stream, const void* in, void* out, long long n, const int* widths, int n_widths,
                               int bit_width, int forward) {
  if (n <= 0) return hipSuccess;
  if (n_widths < 1 || n_widths > 64) return hipErrorInvalidValue;
  ShufflePieces S;
  S.n = n_widths;
  int shift = 0;
  long long off = 0;
  for (int i = 0; i < n_widths; ++i) {
    const int w = widths[n_widths - 1 - i];  // least significant piece first
    if (w < 1) return hipErrorInvalidValue;
    S.width[i] = w;
    S.shift[i] = shift;
    S.off[i] = off;
    shift += w;
    off += (long long)w * n;
  }
  S.off[n_widths] = off;
  if (shift != bit_width) return hipErrorInvalidValue;
  const long long blocks = (n + 255) / 256;
  const dim3 grid((unsigned)(blocks < 65536 ? blocks : 65536)), block(256);
#define MS_GO(T)                                                                                        \
  if (forward)                                                                                          \
    hipLaunchKernelGGL(multishuffle_kernel<T>, grid, block, 0, stream, (const T*)in, (T*)out, n, S);     \
  else                                                                                                  \
    hipLaunchKernelGGL(multiunshuffle_kernel<T>, grid, block, 0, stream, (const T*)in, (T*)out, n, S);
  switch (bit_width) {
    case 8: MS_GO(uint8_t) break;
    case 16: MS_GO(uint16_t) break;
    case 32: MS_GO(uint32_t) break;
    case 64: MS_GO(uint64_t) break;
    default: return hipErrorInvalidValue;
  }
#undef MS_GO
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ Fletcher-32
// c0 = sum d_j mod 65535, c1 = sum (N - j) d_j mod 65535 over the N 16-bit words (utilities.py:235-268 reduces in blocks
// of 360, which leaves the same residues); partial sums in 64-bit accumulators, two atomics per workgroup.
__global__ __launch_bounds__(256) void fletcher32_kernel(const uint16_t* __restrict__ d, long long n, unsigned long long* __restrict__ acc) {
  __shared__ unsigned long long s0[256], s1[256];
  unsigned long long c0 = 0, c1 = 0;
  for (long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (long long)gridDim.x * blockDim.x) {
    const unsigned long long v = d[j];
    c0 += v;
    c1 += ((unsigned long long)((n - j) % 65535)) * v;
    if (c1 >= (1ull << 62)) c1 %= 65535;
  }
  s0[threadIdx.x] = c0 % 65535;
  s1[threadIdx.x] = c1 % 65535;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if ((int)threadIdx.x < st) {
      s0[threadIdx.x] += s0[threadIdx.x + st];
      s1[threadIdx.x] += s1[threadIdx.x + st];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    atomicAdd(&acc[0], s0[0]);
    atomicAdd(&acc[1], s1[0]);
  }
}

hipError_t launch_fletcher32(hipStream_t stream, const void* data, long long n_words, unsigned long long* acc /* [2], zeroed */) {
  if (n_words <= 0) return hipSuccess;
  const long long blocks = (n_words + 256 * 8 - 1) / (256 * 8);
  hipLaunchKernelGGL(fletcher32_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, stream,
                     (const uint16_t*)data, n_words, acc);
  return hipGetLastError();
}

}  // namespace bms
