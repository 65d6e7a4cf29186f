// Separable synthesis for transformations WITHOUT a boost (scri/waveform_grid.py:130-174 with beta = 0): the output grid
// is then the equiangular grid seen through the constant frame rotation, so after rotating the modes once (Wigner D of
// the frame rotor, kernels_rotate*.hip) the map modes -> grid factors into a theta stage and a phi stage per time step,
//     F_m(theta_j) = sum_l  sLambda_lm(theta_j) a_lm ,        g(theta_j, phi_k) = sum_m F_m(theta_j) e^{i m phi_k} ,
// ~0.4 MFLOP per step at l_max = 16 instead of the 3.1 MFLOP of the dense product with the 1369 x 285 matrix of sYlm
// values (which a boost makes necessary: its pixels are on no common rings).  The kernel is the mirror image of
// analysis_split_kernel (kernels_analysis.hip), one workgroup per CU, two time rows per trip:
//
//   theta waves: thread (list, two rings j, j+1).  The m values are dealt into lists of equal total length (host, longest
//                first); a thread keeps sLambda_lm(theta_j) of its list's modes in registers and walks the list, reading
//                a_lm of both rows from LDS (staged by these same waves one trip ahead) and flushing F_m(theta_j) to
//                LDS at the end of every m.
//   phi waves:   wave w owns 16 of the pair's 2 n_theta rings as the A rows of its MFMA tiles.  With
//                P_m = F_m + F_-m, Q_m = i (F_m - F_-m):   g_k = F_0 + sum_m P_m cos(m phi_k) + Q_m sin(m phi_k),
//                g_{n-k} = F_0 + sum_m P_m cos(m phi_k) - Q_m sin(m phi_k)   (k <= n/2): four real products over
//                m = 1..16 (K) by k = 1..16 (one 16-column tile) plus a four-block 4x4x4 product for k = 0, 17, 18, 19,
//                twiddles in registers; the epilogue adds F_0, subtracts the
//                per-pixel offset times the row's eliminated-constant value (the inhomogeneous term of h and sigma,
//                engine_modes.hip) and stores both halves of the ring, 16 bytes per lane, 256-byte runs.
//   F is double buffered in LDS: one barrier per trip.
// HBM traffic = the algorithmic minimum: 16 (n_modes + 1) bytes read, 16 n_pix written per step.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <vector>

#include "kernels.h"

namespace bms {

typedef double v4d_t __attribute__((ext_vector_type(4)));

constexpr int SYN_MAX_WAVES = 10;
constexpr int SYN_XCOLS = 3;  // ring columns beyond k = 16: the side product has four, one of them is k = 0 (n_phi <= 39)
constexpr int SYN_META_FLUSH = 1 << 16, SYN_META_VALID = 1 << 17;

// SCALED: every result is multiplied by scale[pixel] (2 doubles per pixel, equal: the conformal factor's power behind a boost along
// the grid's axis, engine_tables.hip::separable_rotor_grid), kept in LDS next to the offsets
template <int NT, int LEN, bool SCALED>
__global__ __launch_bounds__(640, 1) void synthesis_split_kernel(const double* __restrict__ A, long long lda, long long n_rows,
                                                                 SynGeom g, const double* __restrict__ Tsyn,
                                                                 const int* __restrict__ meta, const double* __restrict__ off,
                                                                 const double* __restrict__ scale, double* __restrict__ Y,
                                                                 long long ldy) {
  // pitch (complex) of an F_m row, a multiple of 16: the phi waves' ds_read_b128 of (ring fi, m fk) then hit 16 distinct 16-byte
  // slots in each of the instruction's four lane groups {0-3, 12-15, 20-27}, ... (with the odd pitch NT + 1 every group was 2-way);
  // the last slot of the m = 0 row carries the row's constant
  constexpr int PJ = NT + 8;
  extern __shared__ double lds[];
  const int fsz = (2 * g.L + 1) * PJ;
  const int na = g.n_modes + 1;
  double2* Fs = reinterpret_cast<double2*>(lds);  // [2 buffers][2 rows][2L+1][PJ]
  double2* abuf = Fs + 4 * fsz;                   // [2 buffers][2 rows][n_modes + 1]
  double2* offl = abuf + 4 * na;                  // [n_pix]
  double* scl = reinterpret_cast<double*>(offl + g.n_theta * g.n_phi);  // [n_pix] (SCALED)
  int* metal = reinterpret_cast<int*>(scl + (SCALED ? g.n_theta * g.n_phi : 0));  // [n_lists][LEN]: byte offset of a_lm | m slot << 16 | flush << 24
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int n_waves = g.nth + g.nph;
  // the phi waves carry the MFMA work: waves 0..3 first (one per SIMD), a fifth and sixth on the SIMDs that hold two waves
  const auto phi_of = [&](int w) {
    const int extra = n_waves >= 8 ? 6 : 4;
    const int f = w < 4 ? w : (w >= extra && w < extra + 2 ? 4 + w - extra : -1);
    return f < g.nph ? f : -1;
  };
  const int phi = phi_of(wave);
  int th = wave;  // rank among the theta waves
  for (int w = 0; w < wave; ++w)
    if (phi_of(w) >= 0) --th;
  // (no zeroing of F: every slot that is read is written in every trip)
  for (int e = tid; e < g.n_theta * g.n_phi; e += blockDim.x)
    offl[e] = off ? *reinterpret_cast<const double2*>(off + 2LL * e) : double2{0.0, 0.0};
  if (SCALED)
    for (int e = tid; e < g.n_theta * g.n_phi; e += blockDim.x) scl[e] = scale[2LL * e];
  for (int e = tid; e < g.n_lists * LEN; e += blockDim.x) {
    const int mt = meta[e];
    metal[e] = ((mt & 1023) * 16) | (((mt >> 10) & 63) << 16) | ((mt & SYN_META_FLUSH) ? 1 << 24 : 0);
  }
  const long long n_pairs = (n_rows + 1) / 2;
  const long long stride = gridDim.x;
  const long long p0 = blockIdx.x;
  // (rows t0, t0 + 1 with t0 = min(2 pp, n_rows - 2): the last pair of an odd series overlaps the one before it)
  auto first_row = [&](long long pp) {
    if (pp >= n_pairs) pp = n_pairs - 1;
    long long t = 2 * pp;
    return t > n_rows - 2 ? n_rows - 2 : t;
  };
  // (without `off` a row carries no constant column: nothing is read there -- the element after the last row's modes may lie past
  // the caller's buffer -- and the constant counts as zero)
  for (int e = tid; e < 2 * na; e += blockDim.x)
    abuf[e] = (off || e % na != g.n_modes) ? *reinterpret_cast<const double2*>(A + (first_row(p0) + e / na) * lda + 2LL * (e % na))
                                           : double2{0.0, 0.0};
  __syncthreads();

  if (phi < 0) {
    // ------------------------------------------------------------------------------------------------ theta waves
    // thread (list, pair of rings): every a_lm read from LDS serves two rings x two rows (with one ring per thread the theta
    // waves were bound by their LDS reads: 80 16-byte reads per thread and trip at l_max = 16)
    const int tt = 64 * th + lane;
    const int npair = (g.n_theta + 1) / 2;
    const bool active = tt < g.n_lists * npair;
    // rings j and j + npair: neighbouring lanes write neighbouring 16-byte slots of an F row (2 j, 2 j + 1 made every store 2-way)
    const int li = active ? tt / npair : 0, j = active ? tt - li * npair : 0;
    const bool second = active && j + npair < g.n_theta;
    // sLambda_lm(theta_j), sLambda_lm(theta_j+1) of this thread's list in registers; the list itself (LDS byte offset of a_lm |
    // m slot << 16 | flush << 24 per entry) stays in LDS and is read two pairs of entries ahead of its use
    double treg[LEN], tre2[LEN];
#pragma unroll
    for (int e = 0; e < LEN; ++e) {
      const int mt = meta[li * LEN + e];
      const bool ok = active && (mt & SYN_META_VALID);
      treg[e] = ok ? Tsyn[(long long)(mt & 1023) * g.n_theta + j] : 0.0;
      tre2[e] = (ok && second) ? Tsyn[(long long)(mt & 1023) * g.n_theta + j + npair] : 0.0;
    }
    const int2* ml = reinterpret_cast<const int2*>(metal) + li * (LEN / 2);
    // this thread's share of a pair's mode rows: elements tt and tt + 64 nth of the 2 (n_modes + 1)
    const int e1 = tt, e2 = tt + 64 * g.nth;
    const bool h1 = e1 < 2 * na && (off || e1 % na != g.n_modes), h2 = e2 < 2 * na && (off || e2 % na != g.n_modes);
    const long long o1 = h1 ? (long long)(e1 / na) * lda + 2LL * (e1 % na) : 0, o2 = h2 ? (long long)(e2 / na) * lda + 2LL * (e2 % na) : 0;
    auto request = [&](long long pp, double2& x1, double2& x2) {
      const double* row = A + first_row(pp) * lda;
      x1 = *reinterpret_cast<const double2*>(row + o1);
      x2 = *reinterpret_cast<const double2*>(row + o2);
    };
    double2 x1, x2;
    int buf = 0;
    for (long long p = p0; p < n_pairs; p += stride, buf ^= 1) {
      request(p + stride, x1, x2);
      double2* Fb = Fs + buf * 2 * fsz;
      const char* a0 = reinterpret_cast<const char*>(abuf + buf * 2 * na);
      const char* a1 = a0 + 16 * na;
      double r0 = 0.0, i0 = 0.0, r1 = 0.0, i1 = 0.0, s0 = 0.0, k0 = 0.0, s1 = 0.0, k1 = 0.0;  // ring j (r, i), ring j + 1 (s, k); rows 0, 1
      // entries two at a time, the reads of the next two issued before the multiply-adds of these two (the empty asm keeps
      // that order: left alone the compiler issues every read first)
      static_assert(LEN % 2 == 0, "entries are walked in pairs");
      double2 u[2][4];
      int2 mq[3];
      auto read2 = [&](const int2& mt, double2(&d)[4]) {
        d[0] = *reinterpret_cast<const double2*>(a0 + (mt.x & 0xffff));
        d[1] = *reinterpret_cast<const double2*>(a1 + (mt.x & 0xffff));
        d[2] = *reinterpret_cast<const double2*>(a0 + (mt.y & 0xffff));
        d[3] = *reinterpret_cast<const double2*>(a1 + (mt.y & 0xffff));
      };
      auto entry = [&](int e, int mt, const double2& x, const double2& y) {
        r0 = fma(treg[e], x.x, r0);
        i0 = fma(treg[e], x.y, i0);
        r1 = fma(treg[e], y.x, r1);
        i1 = fma(treg[e], y.y, i1);
        s0 = fma(tre2[e], x.x, s0);
        k0 = fma(tre2[e], x.y, k0);
        s1 = fma(tre2[e], y.x, s1);
        k1 = fma(tre2[e], y.y, k1);
        if (mt & (1 << 24)) {
          const int at = ((mt >> 16) & 63) * PJ + j;
          if (active) {
            Fb[at] = double2{r0, i0};
            Fb[fsz + at] = double2{r1, i1};
          }
          if (second) {
            Fb[at + npair] = double2{s0, k0};
            Fb[fsz + at + npair] = double2{s1, k1};
          }
          r0 = i0 = r1 = i1 = s0 = k0 = s1 = k1 = 0.0;
        }
      };
      mq[0] = ml[0];
      if (LEN > 2) mq[1] = ml[1];
      read2(mq[0], u[0]);
#pragma unroll
      for (int pi = 0; pi < LEN / 2; ++pi) {
        const int b = pi & 1;
        if (pi + 2 < LEN / 2) mq[(pi + 2) % 3] = ml[pi + 2];
        if (pi + 1 < LEN / 2) read2(mq[(pi + 1) % 3], u[b ^ 1]);
        entry(2 * pi, mq[pi % 3].x, u[b][0], u[b][1]);
        entry(2 * pi + 1, mq[pi % 3].y, u[b][2], u[b][3]);
        asm volatile("" : "+v"(r0), "+v"(i0), "+v"(r1), "+v"(i1), "+v"(s0), "+v"(k0), "+v"(s1), "+v"(k1)::"memory");
      }
      const double2* a0c = reinterpret_cast<const double2*>(a0);
      const double2* a1c = reinterpret_cast<const double2*>(a1);
      if (tt < 2) Fb[tt * fsz + g.L * PJ + PJ - 1] = off ? (tt ? a1c : a0c)[g.n_modes] : double2{0.0, 0.0};  // the row's eliminated-constant value
      double2* an = abuf + (buf ^ 1) * 2 * na;
      if (h1) an[e1] = x1;
      if (h2) an[e2] = x2;
      __syncthreads();  // F of this pair and the modes of the next complete; the phi waves are done with the other F
    }
  } else {
    // -------------------------------------------------------------------------------------------------- phi waves
    const int fi = lane & 15, fk = lane >> 4;
    // The ring has n_phi / 2 + 1 independent columns k = 0 .. nk - 1.  k = 1..16 are the 16 columns of ONE 16x16x4 tile; k = 0
    // (cos = 1, sin = 0) and the up to three columns beyond 16 go through the four-block 4x4x4 MFMA, whose A fragment is the
    // very same register (lane = row-in-block + 4 block + 16 k, i.e. ring slot + 16 k) and whose 4 columns per block are all
    // that is needed: a quarter of the cycles of a second 16-column tile that would carry 3 useful columns.
    double cs[4], sn[4], cx[4], sx[4];
    const int jx = lane & 3;                 // column of the side product: k = 0, 17, 18, 19
    const int kx = jx == 0 ? 0 : 16 + jx;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int m = 4 * s + fk + 1;
      double sv, cv;
      sincospi(2.0 * (double)(((long long)m * (fi + 1)) % g.n_phi) / (double)g.n_phi, &sv, &cv);
      const bool ok = fi + 1 < g.nk && m <= g.L;
      cs[s] = ok ? cv : 0.0;
      sn[s] = ok ? sv : 0.0;
      sincospi(2.0 * (double)(((long long)m * kx) % g.n_phi) / (double)g.n_phi, &sv, &cv);
      const bool okx = kx < g.nk && m <= g.L;
      cx[s] = okx ? cv : 0.0;
      sx[s] = okx ? sv : 0.0;
    }
    // A row of this lane: ring slot 16 phi + fi of the pair's 2 n_theta
    const int q = 16 * phi + fi;
    const bool okr = q < 2 * g.n_theta;
    const int ra = okr ? q / g.n_theta : 0;
    const int fa = ra * fsz + (okr ? q - ra * g.n_theta : 0);
    // results of the tile: D rows fk + 4 v = ring slots 16 phi + fk + 4 v, column k = fi + 1
    int fv[4], pv[4];
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const int qv = 16 * phi + fk + 4 * v;
      const bool ok = qv < 2 * g.n_theta;
      const int rv = ok ? qv / g.n_theta : 0, jv = ok ? qv - rv * g.n_theta : 0;
      fv[v] = ok ? rv * fsz + g.L * PJ + jv : -1;  // F_0 of the ring
      pv[v] = (rv << 24) | (jv * g.n_phi);         // row of the pair | first pixel of the ring
    }
    // result of the side product: ring slot 16 phi + 4 block + row = 16 phi + 4 ((lane >> 2) & 3) + (lane >> 4), column jx
    const int qs = 16 * phi + 4 * ((lane >> 2) & 3) + (lane >> 4);
    const bool has_x = qs < 2 * g.n_theta && kx < g.nk;
    const int rs = has_x ? qs / g.n_theta : 0, js = has_x ? qs - rs * g.n_theta : 0;
    const bool has_off = off != nullptr;
    int buf = 0;
    for (long long p = p0; p < n_pairs; p += stride, buf ^= 1) {
      __syncthreads();
      const double2* Fb = Fs + buf * 2 * fsz;
      double* yrow = Y + first_row(p) * ldy;
      int opaque = 0;  // (an add of zero the compiler cannot see through: the output addresses of a lane are recomputed per
      asm volatile("" : "+v"(opaque));  // trip instead of being carried, and spilled, across the loop)
      v4d_t ure{0.0, 0.0, 0.0, 0.0}, uim = ure, vre = ure, vim = ure;
      double xur = 0.0, xui = 0.0, xvr = 0.0, xvi = 0.0;
      const int ksm = (g.L + 3) / 4;  // k steps that carry an m <= L at all (two of the four at l_max = 8)
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        if (s >= ksm) break;
        const int m = 4 * s + fk + 1;
        const int mm = m <= g.L ? m : 0;  // (rows beyond L: their twiddles are zero)
        const double2 fp = Fb[fa + (g.L + mm) * PJ], fm = Fb[fa + (g.L - mm) * PJ];
        const double px = fp.x + fm.x, py = fp.y + fm.y, qx = fm.y - fp.y, qy = fp.x - fm.x;
        ure = __builtin_amdgcn_mfma_f64_16x16x4f64(px, cs[s], ure, 0, 0, 0);
        uim = __builtin_amdgcn_mfma_f64_16x16x4f64(py, cs[s], uim, 0, 0, 0);
        vre = __builtin_amdgcn_mfma_f64_16x16x4f64(qx, sn[s], vre, 0, 0, 0);
        vim = __builtin_amdgcn_mfma_f64_16x16x4f64(qy, sn[s], vim, 0, 0, 0);
        xur = __builtin_amdgcn_mfma_f64_4x4x4f64(px, cx[s], xur, 0, 0, 0);
        xui = __builtin_amdgcn_mfma_f64_4x4x4f64(py, cx[s], xui, 0, 0, 0);
        xvr = __builtin_amdgcn_mfma_f64_4x4x4f64(qx, sx[s], xvr, 0, 0, 0);
        xvi = __builtin_amdgcn_mfma_f64_4x4x4f64(qy, sx[s], xvi, 0, 0, 0);
      }
      if (has_x) {
        const double2 f0 = Fb[rs * fsz + g.L * PJ + js];
        const double c = Fb[rs * fsz + g.L * PJ + PJ - 1].x;  // (the eliminated constant series is real)
        double* yr = yrow + rs * ldy;
        const int pix0 = js * g.n_phi + opaque;
        const double ux = f0.x + xur, uy = f0.y + xui;
        {
          const double2 o = has_off ? offl[pix0 + kx] : double2{0.0, 0.0};
          const double w = SCALED ? scl[pix0 + kx] : 1.0;
          *reinterpret_cast<double2*>(yr + 2LL * (pix0 + kx)) = double2{(ux + xvr - o.x * c) * w, (uy + xvi - o.y * c) * w};
        }
        if (kx >= 1 && 2 * kx != g.n_phi) {
          const int k2 = g.n_phi - kx;
          const double2 o = has_off ? offl[pix0 + k2] : double2{0.0, 0.0};
          const double w = SCALED ? scl[pix0 + k2] : 1.0;
          *reinterpret_cast<double2*>(yr + 2LL * (pix0 + k2)) = double2{(ux - xvr - o.x * c) * w, (uy - xvi - o.y * c) * w};
        }
      }
      const int k = fi + 1;
      if (k < g.nk) {
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          if (fv[v] < 0) continue;
          const double2 f0 = Fb[fv[v]];
          const int rv = pv[v] >> 24, pix0 = (pv[v] & 0xffffff) + opaque;
          const double c = Fb[rv * fsz + g.L * PJ + PJ - 1].x;
          double* yr = yrow + rv * ldy;
          const double ux = f0.x + ure[v], uy = f0.y + uim[v];
          {
            const double2 o = has_off ? offl[pix0 + k] : double2{0.0, 0.0};
            const double w = SCALED ? scl[pix0 + k] : 1.0;
            *reinterpret_cast<double2*>(yr + 2LL * (pix0 + k)) = double2{(ux + vre[v] - o.x * c) * w, (uy + vim[v] - o.y * c) * w};
          }
          if (2 * k != g.n_phi) {
            const int k2 = g.n_phi - k;
            const double2 o = has_off ? offl[pix0 + k2] : double2{0.0, 0.0};
            const double w = SCALED ? scl[pix0 + k2] : 1.0;
            *reinterpret_cast<double2*>(yr + 2LL * (pix0 + k2)) = double2{(ux - vre[v] - o.x * c) * w, (uy - vim[v] - o.y * c) * w};
          }
        }
      }
    }
  }
}

size_t synthesis_split_scale_bytes(int n_theta, int n_phi) { return sizeof(double) * (((size_t)n_theta * n_phi + 1) & ~(size_t)1); }

// The m values of the input modes dealt into lists of nearly equal total length (longest first into the shortest list);
// meta[list][e] = mode index | m slot << 10 | flush << 16 | valid << 17.
int synthesis_split_plan(int n_theta, int n_phi, int ell_min, int ell_max, SynGeom& g, std::vector<int>& meta, size_t& lds_bytes,
                         int& nt, int& len) {
  if (n_theta < 3 || n_theta > 40 || n_phi < 2 || n_phi / 2 + 1 > 17 + SYN_XCOLS || ell_max < 1 || ell_max > 16 || ell_min < 0 || ell_min > ell_max) return 0;
  const int n_modes = (ell_max + 1) * (ell_max + 1) - ell_min * ell_min;
  if (n_modes > 1023) return 0;
  g.n_theta = n_theta, g.n_phi = n_phi, g.L = ell_max, g.n_modes = n_modes, g.nk = n_phi / 2 + 1;
  g.nph = (2 * n_theta + 15) / 16;
  if (g.nph > 6) return 0;
  struct Item {
    int m, w;
  };
  std::vector<Item> items;
  for (int m = -ell_max; m <= ell_max; ++m) items.push_back({m, ell_max - std::max(ell_min, std::abs(m)) + 1});
  std::stable_sort(items.begin(), items.end(), [](const Item& a, const Item& b) { return a.w > b.w; });
  // lists no longer than 8 .. 20 entries (the kernel is built for these lengths): the shortest that fits 10 waves
  const int npair = (n_theta + 1) / 2;
  for (int target : {8, 12, 16, 20})  // (two table rows per thread: beyond 20 entries they no longer fit the registers)
    for (g.nth = 1; g.nth + g.nph <= SYN_MAX_WAVES; ++g.nth) {
      g.n_lists = std::min<int>(64 * g.nth / npair, (int)items.size());
      if (g.n_lists < 1 || n_modes + 1 > 64 * g.nth) continue;
      std::vector<std::vector<int>> lists(g.n_lists);
      std::vector<int> load(g.n_lists, 0);
      for (const Item& it : items) {
        const int l = (int)(std::min_element(load.begin(), load.end()) - load.begin());
        lists[l].push_back(it.m);
        load[l] += it.w;
      }
      if (*std::max_element(load.begin(), load.end()) > target) continue;
      len = g.len = target;
      meta.assign((size_t)g.n_lists * len, 0);
      for (int l = 0; l < g.n_lists; ++l) {
        int e = 0;
        for (int m : lists[l])
          for (int ell = std::max(ell_min, std::abs(m)); ell <= ell_max; ++ell) {
            const int o = ell * ell + ell + m - ell_min * ell_min;
            meta[(size_t)l * len + e++] = o | ((m + ell_max) << 10) | (ell == ell_max ? SYN_META_FLUSH : 0) | SYN_META_VALID;
          }
      }
      nt = n_theta <= 24 ? 24 : 40;
      lds_bytes = sizeof(double2) * ((size_t)4 * (2 * ell_max + 1) * (nt + 8) + 4 * (size_t)(n_modes + 1) + (size_t)n_theta * n_phi) +
                  sizeof(int) * meta.size();
      return lds_bytes <= 160 * 1024 ? 1 : 0;
    }
  return 0;
}

hipError_t launch_synthesis_split(hipStream_t stream, const double* A, long long lda, long long n_rows, const SynGeom& g, int nt,
                                  const double* Tsyn, const int* meta, const double* off, double* Y, long long ldy, size_t lds_bytes,
                                  int n_cu, const double* scale) {
  if (n_rows < 2) return hipErrorInvalidValue;
  if (scale) lds_bytes += synthesis_split_scale_bytes(g.n_theta, g.n_phi);
  if (lds_bytes > 160 * 1024) return hipErrorInvalidValue;
  const long long n_pairs = (n_rows + 1) / 2;
  // one workgroup per CU at l_max = 16 (10 waves, 124 KB of LDS); small shapes need fewer waves and less LDS and get two or three
  long long per_cu = std::min<long long>(3, std::min<long long>(SYN_MAX_WAVES / (g.nth + g.nph), (160 * 1024) / (long long)lds_bytes));
  if (per_cu < 1) per_cu = 1;
  const long long max_blocks = per_cu * n_cu;
  const dim3 grid((unsigned)(n_pairs < max_blocks ? n_pairs : max_blocks)), block(64 * (g.nth + g.nph));
#define SYN_GO_S(NT, LEN, SC)                                                                                                       \
  {                                                                                                                                  \
    hipError_t e = allow_dynamic_lds((const void*)synthesis_split_kernel<NT, LEN, SC>);                                                                              \
    if (e != hipSuccess) return e;                                                                                                   \
    hipLaunchKernelGGL((synthesis_split_kernel<NT, LEN, SC>), grid, block, lds_bytes, stream, A, lda, n_rows, g, Tsyn, meta, off, scale, \
                       Y, ldy);                                                                                                      \
    return hipGetLastError();                                                                                                        \
  }
#define SYN_GO(NT, LEN)            \
  {                                \
    if (scale) SYN_GO_S(NT, LEN, true) \
    SYN_GO_S(NT, LEN, false)       \
  }
#define SYN_LEN(NT)                  \
  {                                  \
    if (g.len == 8) SYN_GO(NT, 8)    \
    if (g.len == 12) SYN_GO(NT, 12)  \
    if (g.len == 16) SYN_GO(NT, 16)  \
    SYN_GO(NT, 20)                   \
  }
  if (nt == 24) SYN_LEN(24)
  SYN_LEN(40)
#undef SYN_LEN
#undef SYN_GO
#undef SYN_GO_S
}

}  // namespace bms
