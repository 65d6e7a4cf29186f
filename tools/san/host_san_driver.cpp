// Host-side sanitizer run (no GPU sanitizer exists on this pool: ASan / UBSan cover what runs on the host).
// `make -C scri_amd/csrc SAN=1` compiles this file -- which INCLUDES the host side of the engine (engine_*.hip, split by entry family
// behind engine.h), so that every planning helper is instrumented -- host-only with -fsanitize=address,undefined and runs it: shard plans, output windows, knot ranges, column
// parts, chunk walks, rotor / harmonic / conformal tables and the frame integrator over the five BASELINE shapes, 1..8 shards,
// 1..8 column parts, series of 2..9 samples and odd grids.  Nothing here touches a device.
#include "../../scri_amd/csrc/engine_context.hip"
#include "../../scri_amd/csrc/engine_tables.hip"
#include "../../scri_amd/csrc/engine_rotate.hip"
#include "../../scri_amd/csrc/engine_modes.hip"
#include "../../scri_amd/csrc/engine_abd.hip"
#include "../../scri_amd/csrc/engine_blocks.hip"

#include <cstdio>
#include <random>

namespace {

struct Shape {
  const char* name;
  int ell_max, n_theta, n_phi, lst;
  long long n;
  double dt, boost_scale;
  bool abd;
};

int g_checks = 0;
#define REQUIRE(cond)                                                          \
  do {                                                                         \
    ++g_checks;                                                                \
    if (!(cond)) {                                                             \
      std::fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond);   \
      std::exit(2);                                                            \
    }                                                                          \
  } while (0)

void fill_transformation(bms_transformation& tr, std::vector<cplx>& st, const Shape& s) {
  st.assign((size_t)(s.lst + 1) * (s.lst + 1), cplx{0, 0});
  // a real supertranslation: alpha_{l,-m} = (-1)^m conj(alpha_{l,m})
  std::mt19937_64 rng(7);
  std::normal_distribution<double> g(0.0, 1e-3);
  for (int l = 0; l <= s.lst; ++l) {
    st[LM_index(l, 0, 0)] = {g(rng), 0.0};
    for (int m = 1; m <= l; ++m) {
      const cplx a = {g(rng), g(rng)};
      const double sg = (m & 1) ? -1.0 : 1.0;
      st[LM_index(l, m, 0)] = a;
      st[LM_index(l, -m, 0)] = {sg * a.re, -sg * a.im};
    }
  }
  tr = bms_transformation{};
  tr.supertranslation = st.data();
  tr.ell_max_supertranslation = s.lst;
  const double q[4] = {1, 2, 3, 4};
  const double nq = std::sqrt(30.0);
  for (int i = 0; i < 4; ++i) tr.frame_rotation[i] = q[i] / nq;
  tr.boost_velocity[0] = 1e-4 * s.boost_scale, tr.boost_velocity[1] = 2e-4 * s.boost_scale, tr.boost_velocity[2] = 3e-4 * s.boost_scale;
  tr.n_theta = s.n_theta, tr.n_phi = s.n_phi;
  tr.ell_max_out = s.ell_max;
}

void run_shape(const Shape& s) {
  std::vector<cplx> st;
  bms_transformation tr;
  fill_transformation(tr, st, s);
  std::vector<double> t((size_t)s.n);
  for (long long i = 0; i < s.n; ++i) t[(size_t)i] = s.dt * (double)i;
  bms_ctx dummy;  // (never given a device: the helpers use it for error texts only)
  bool regular = false;
  REQUIRE(validate_common(&dummy, s.n, t.data(), &tr, 0, -1, &regular, s.abd ? 2 : 4) == BMS_OK);
  REQUIRE(regular);
  REQUIRE(spline_tile_for(t.data(), s.n) == SPLINE_TILE);
  PixelTables T;
  build_pixel_tables(&tr, T);
  REQUIRE(T.n_pix == s.n_theta * s.n_phi);
  int64_t i_lo, i_hi;
  if (s.abd)
    output_window_abd(T, t.data(), s.n, i_lo, i_hi);
  else
    output_window(T, t.data(), s.n, i_lo, i_hi);
  REQUIRE(0 <= i_lo && i_lo <= i_hi && i_hi <= s.n);
  for (int shards = 1; shards <= 8; ++shards) {
    int64_t covered = i_lo;
    for (int r = 0; r < shards; ++r) {
      const int64_t o0 = i_lo + (i_hi - i_lo) * r / shards, o1 = i_lo + (i_hi - i_lo) * (r + 1) / shards;
      int64_t need[2], win[2];
      REQUIRE(bms_shard_plan(nullptr, t.data(), s.n, &tr, o0, o1, need, win) == BMS_OK);
      REQUIRE(win[0] == i_lo || s.abd);  // (the planner computes the WaveformModes window; the ABD one differs by rounding of 1/gamma only)
      if (o1 > o0) {
        REQUIRE(0 <= need[0] && need[0] <= o0 && o1 <= need[1] && need[1] <= s.n);
        int64_t ja, jb;
        needed_knots(T, t.data(), s.n, o0, o1, ja, jb);
        REQUIRE(need[0] <= ja && jb < need[1]);
        // the window of time samples a shard uploads, the skew ranges of its column blocks and the search bound of the evaluation
        bms_shard sh = {need[0], need[1] - need[0], o0, o1, 0, 0};
        int64_t lo, hi;
        time_window(s.n, &sh, lo, hi);
        REQUIRE(lo <= need[0] && need[1] <= hi);
        const int n_cols = T.n_pix;
        for (int parts = 1; parts <= 8; ++parts)
          for (int part = 0; part < parts; ++part) {
            sh.col_part = part, sh.col_parts = parts;
            int cA, cB;
            REQUIRE(column_range(&dummy, &sh, n_cols, cA, cB) == BMS_OK);
            REQUIRE(0 <= cA && cA <= cB && cB <= n_cols);
            if (cB > cA) {
              const BsplineSpread sp = skew_spread(T, cA, cB, t.data());
              REQUIRE(sp.skew_rate_range >= 0 && sp.skew_offset_range >= 0);
              REQUIRE(eval_search_halfwidth(T, cA, cB, t.data(), need[0], need[1]) >= 0);
            }
          }
      }
      covered = o1;
    }
    REQUIRE(covered == i_hi);
  }
  // rotor grid, harmonics, conformal factors of the (boosted, rotated) grid through the ctx = NULL building blocks
  std::vector<double> rot((size_t)4 * T.n_pix);
  REQUIRE(bms_rotor_grid(nullptr, tr.frame_rotation, tr.boost_velocity, s.n_theta, s.n_phi, rot.data()) == BMS_OK);
  const int lm = std::min(s.ell_max, 12), nm = LM_total_size(0, lm);
  std::vector<cplx> Y((size_t)T.n_pix * nm);
  for (int spin = -2; spin <= 2; spin += 2) REQUIRE(bms_swsh_grid(nullptr, rot.data(), T.n_pix, spin, 0, lm, Y.data()) == BMS_OK);
  std::vector<double> k(T.n_pix), ik(T.n_pix), ik3(T.n_pix);
  std::vector<cplx> ek(T.n_pix);
  REQUIRE(bms_conformal_factors(nullptr, tr.boost_velocity, rot.data(), T.n_pix, k.data(), ek.data(), ik.data(), ik3.data()) == BMS_OK);
  std::vector<double> th((size_t)s.n_theta);
  (void)bms_ring_colatitudes(tr.frame_rotation, tr.boost_velocity, s.n_theta, s.n_phi, th.data());
  const double vz[3] = {0, 0, 0.1}, id[4] = {1, 0, 0, 0};
  REQUIRE(bms_ring_colatitudes(id, vz, s.n_theta, s.n_phi, th.data()) == 1);  // a boost along the grid's axis keeps the rings
  std::vector<double> q;
  theta_quadrature_weights(s.n_theta, q);
  REQUIRE((int)q.size() == s.n_theta);
  std::printf("%-22s n=%lld grid %dx%d window [%lld, %lld)\n", s.name, s.n, s.n_theta, s.n_phi, (long long)i_lo, (long long)i_hi);
}

void short_series_and_odd_grids() {
  for (int n = 2; n <= 9; ++n)
    for (int grid : {3, 4, 5, 8, 9, 11}) {
      Shape s = {"short", 2, grid, grid + (grid & 1 ? 0 : 1), 1, n, 0.37, 10.0, n < 4};
      std::vector<cplx> st;
      bms_transformation tr;
      fill_transformation(tr, st, s);
      tr.ell_max_out = 1;
      std::vector<double> t((size_t)n);
      for (int i = 0; i < n; ++i) t[(size_t)i] = 0.37 * i + 0.01 * i * i;
      bms_ctx dummy;
      REQUIRE(validate_common(&dummy, n, t.data(), &tr, 0, -1, nullptr, 2) == BMS_OK);
      PixelTables T;
      build_pixel_tables(&tr, T);
      int64_t a, b;
      output_window(T, t.data(), n, a, b);
      output_window_abd(T, t.data(), n, a, b);
      REQUIRE(0 <= a && a <= b && b <= n);
      if (b > a) {
        int64_t ja, jb;
        needed_knots(T, t.data(), n, a, b, ja, jb);
        REQUIRE(0 <= ja && ja <= jb && jb <= n - 1);
      }
      if (n >= 4) {
        int64_t need[2], win[2];
        REQUIRE(bms_shard_plan(nullptr, t.data(), n, &tr, 0, n, need, win) == BMS_OK);
      }
    }
  // rejected inputs go through the error path (formatted messages)
  bms_ctx dummy;
  std::vector<cplx> st;
  bms_transformation tr;
  Shape s = {"bad", 2, 5, 5, 1, 4, 0.1, 1.0, false};
  fill_transformation(tr, st, s);
  double t_bad[4] = {0.0, 0.1, 0.1, 0.3};
  REQUIRE(validate_common(&dummy, 4, t_bad, &tr) != BMS_OK);
  REQUIRE(std::strlen(bms_last_error(&dummy)) > 0);
  tr.boost_velocity[0] = 2.0;
  double t_ok[4] = {0.0, 0.1, 0.2, 0.3};
  REQUIRE(validate_common(&dummy, 4, t_ok, &tr) != BMS_OK);
}

void frame_integration() {
  const int n = 400;
  std::vector<double> t(n), om(3 * n), R(4 * n);
  for (int i = 0; i < n; ++i) {
    t[i] = 0.05 * i + 1e-4 * i * i;
    om[3 * i] = 0.3 * std::sin(0.1 * t[i]), om[3 * i + 1] = 0.2, om[3 * i + 2] = 1.0 + 0.01 * t[i];
  }
  const double R0[4] = {1, 0, 0, 0};
  REQUIRE(bms_integrate_angular_velocity(nullptr, t.data(), n, om.data(), R0, 1e-12, R.data()) == BMS_OK);
  for (int i = 0; i < n; ++i) {
    const double nn = R[4 * i] * R[4 * i] + R[4 * i + 1] * R[4 * i + 1] + R[4 * i + 2] * R[4 * i + 2] + R[4 * i + 3] * R[4 * i + 3];
    REQUIRE(std::fabs(nn - 1) < 1e-12);
  }
  for (int ell : {0, 1, 2, 7, 16, 24}) {
    std::vector<double> D;
    delta_matrix<long double>(ell, D);
    REQUIRE((int)D.size() == (2 * ell + 1) * (2 * ell + 1));
  }
}


// The slab the named work-space buffers are carved from (bms_ctx_reserve): random grow / release sequences against a brute-force
// picture of the address range -- regions never overlap, freed neighbours coalesce, what is free plus what is held is the slab.
void slab_allocator() {
  std::mt19937_64 rng(11);
  for (int round = 0; round < 40; ++round) {
    Slab sl;
    sl.cap = 1 << 20;
    sl.free[0] = sl.cap;
    struct Held {
      size_t off, len;
    };
    std::vector<Held> held;
    for (int step = 0; step < 400; ++step) {
      const bool take = held.empty() || (rng() % 3) != 0;
      if (take) {
        const size_t want = ((size_t)(rng() % 60000) + 1 + 255) & ~(size_t)255;
        size_t off = 0;
        if (sl.take(want, &off)) {
          REQUIRE(off + want <= sl.cap && off % 256 == 0);
          for (const Held& h : held) REQUIRE(off + want <= h.off || h.off + h.len <= off);
          held.push_back({off, want});
        } else {
          for (const auto& f : sl.free) REQUIRE(f.second < want);  // refused only if no hole is large enough
        }
      } else {
        const size_t i = rng() % held.size();
        sl.give(held[i].off, held[i].len);
        held.erase(held.begin() + i);
      }
      size_t free_total = 0, prev_end = (size_t)-1;
      for (const auto& f : sl.free) {
        REQUIRE(f.second > 0);
        REQUIRE(prev_end == (size_t)-1 || prev_end < f.first);  // sorted, and coalesced: no two holes touch
        prev_end = f.first + f.second;
        free_total += f.second;
      }
      size_t held_total = 0;
      for (const Held& h : held) held_total += h.len;
      REQUIRE(free_total + held_total == sl.cap);
    }
    for (const Held& h : held) sl.give(h.off, h.len);
    REQUIRE(sl.free.size() == 1 && sl.free.begin()->first == 0 && sl.free.begin()->second == sl.cap);
  }
}

}  // namespace

int main() {
  // BASELINE.json configs 1..5 (cfg4's 1e6 steps and cfg5's 2e5 as they are: the planners are O(n log n) at worst)
  const Shape shapes[] = {
      {"cfg1 (l<=4, 2000)", 4, 11, 11, 1, 2000, 0.055, 1.0, false},
      {"cfg2 (l<=8, 1e5)", 8, 21, 21, 2, 100000, 0.1, 0.0, false},
      {"cfg3 (l<=16, 1e5)", 16, 37, 37, 2, 100000, 0.1, 1.0, false},
      {"cfg3 beta=1e-2", 16, 37, 37, 2, 100000, 0.1, 26.7, false},
      {"cfg3 beta=0.1", 16, 37, 37, 2, 100000, 0.1, 267.0, false},
      {"cfg4 (l<=16, 1e6)", 16, 37, 37, 2, 1000000, 0.1, 1.0, false},
      {"cfg5 (ABD l<=24, 2e5)", 24, 99, 99, 2, 200000, 0.1, 1.0, true},
  };
  for (const Shape& s : shapes) run_shape(s);
  short_series_and_odd_grids();
  frame_integration();
  slab_allocator();
  std::printf("host sanitizer run: %d checks, clean\n", g_checks);
  return 0;
}
