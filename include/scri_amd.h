/* scri_amd -- C ABI of the MI355X-native BMS-transformation engine.
 *
 * Drop-in boundary for the BMS hot path of moble/scri (reference scri 2024.0.13).  The reference has no FFI:
 * its seam is Python attribute grafting (scri/__init__.py:125-150, scri/waveform_modes.py:705-719,
 * scri/asymptotic_bondi_data/__init__.py:235-263).  Each entry point below replaces the arithmetic of the
 * cited reference function; argument parsing / validation / bookkeeping (kwargs precedence, history, frame
 * update) stays in the Python shim `scri_amd` which mirrors the reference's behaviour and binds these symbols
 * with ctypes (INTEGRATION.md shows the stub a scri maintainer would add).
 *
 * Conventions
 *   - complex data are interleaved (re, im) float64 = numpy complex128; "c16[N][ld]" means N rows with a
 *     row stride of `ld` complex elements (C-contiguous when ld == row length);
 *   - `mem` selects where the bulk buffers live: BMS_HOST (numpy memory; the library copies to the GPU and
 *     back) or BMS_DEVICE (HIP device pointers, e.g. torch.Tensor.data_ptr()); time arrays and all small
 *     parameter arrays are always host memory;
 *   - the caller owns every buffer; the library keeps no pointer after return.  Device work space is cached
 *     behind the opaque context and released by bms_ctx_destroy;
 *   - every function returns 0 on success or a negative bms_status; bms_last_error(ctx) gives the text;
 *   - a context is bound to one GPU and one HIP stream; calls on one context are serialised by the caller,
 *     different contexts are independent.  There is NO CPU fallback: without a gfx950 device
 *     bms_ctx_create fails.
 */
#ifndef SCRI_AMD_H
#define SCRI_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct bms_ctx bms_ctx;

enum bms_status {
  BMS_OK = 0,
  BMS_ERR_INVALID = -1,     /* bad argument (maps to ValueError in the shim) */
  BMS_ERR_HIP = -2,         /* HIP runtime failure */
  BMS_ERR_NOMEM = -3,       /* device (or host) allocation failed */
  BMS_ERR_UNSUPPORTED = -4, /* valid request outside the implemented range */
  BMS_ERR_NODEVICE = -5,    /* no usable GPU */
  BMS_ERR_INTERNAL = -6     /* a C++ exception inside the library, caught at the boundary (none ever crosses it) */
};

enum bms_mem { BMS_HOST = 0, BMS_DEVICE = 1 };

/* type-specific inhomogeneous term of the WaveformModes path, scri/waveform_grid.py:485-557 */
enum bms_type_term {
  BMS_TERM_NONE = 0,  /* psi4, hdot, news (and unknown types, which the reference treats as psi4) */
  BMS_TERM_H = 1,     /* h:     f -= ethbar^2 alpha (NP normalisation), waveform_grid.py:486-494 */
  BMS_TERM_SIGMA = 2, /* sigma: f -= eth^2 alpha / 2 ... GHP,            waveform_grid.py:495-503 */
  BMS_TERM_PSI = 3    /* psi0..psi3: sum over higher Weyl scalars,       waveform_grid.py:504-550 */
};

int bms_version(void);

/* ---- context ------------------------------------------------------------------------------------------- */
int bms_ctx_create(int device, bms_ctx** ctx);
void bms_ctx_destroy(bms_ctx* ctx);
const char* bms_last_error(const bms_ctx* ctx); /* ctx may be NULL: error of the last failed bms_ctx_create */
/* run on a caller-provided hipStream_t (NULL: the context's own stream) */
int bms_ctx_set_stream(bms_ctx* ctx, void* hip_stream);
/* run on the device's default (null) stream -- handle 0, which bms_ctx_set_stream takes as "own stream" -- so that work a
 * caller queued there (torch's default stream: allocations, copies, memsets) is ordered before the engine's kernels */
int bms_ctx_use_default_stream(bms_ctx* ctx);
/* Route options of ONE context: choices between routes that give the same results to rounding (dense product vs separable
 * synthesis, evaluating product vs product + back substitution, fused vs two-pass boost-free route, ...; the list is
 * scri_amd/csrc/env.h, names with or without the SCRI_AMD_ prefix, e.g. "NO_GEMM_EVAL").  Flags take 0 / 1; AXIS_BOOST_MIN_WORK a
 * count of multiply-adds (< 0: built-in threshold, 0: always) and GEMM_EVAL_STEP 0 / 61 / 64.  A context takes its DEFAULTS from the
 * SCRI_AMD_<NAME> environment variables once, inside bms_ctx_create; no call reads the environment afterwards, so two contexts of
 * one process may run different routes concurrently ("re-entrant per ctx").  BMS_ERR_INVALID for an unknown name. */
int bms_ctx_set_option(bms_ctx* ctx, const char* name, int64_t value);
int bms_ctx_get_option(bms_ctx* ctx, const char* name, int64_t* value);
/* cap on the grid work space in bytes (time axis is processed in chunks that fit); 0 = default: min(96 GB, a third of the device
 * memory that is free at the time).  With the default a call that runs out of device memory halves the cap and tries again. */
int bms_ctx_set_workspace_limit(bms_ctx* ctx, uint64_t bytes);
/* For short-lived callers: ONE device allocation of `bytes` (0: one and a half times the work-space cap) from which the context carves
 * its work-space buffers afterwards, so that the first full-size call of the process allocates nothing (device allocations cost
 * 70 - 120 ms per GB on this platform: the first device-resident map_to_superrest_frame of a process took 2.45 - 3.3 s, the second
 * 0.92 s).  May be called again to add room.  No reference counterpart. */
int bms_ctx_reserve(bms_ctx* ctx, uint64_t bytes);
/* Diagnostics of the evaluating product (the dense route's synthesis + spline evaluation): out[0] = tiles launched since the last
 * reset, out[1] = tiles whose samples left the window of output times staged in LDS (non-uniform time axes, strong boosts: same
 * results through global memory, slower), out[2] = per-column marches continued from global memory.  No reference counterpart. */
int bms_ctx_get_eval_stats(bms_ctx* ctx, int64_t* out /* [3] */, int reset);
/* block until all work queued by this context has finished */
int bms_ctx_synchronize(bms_ctx* ctx);

/* Optional per-kernel timing: when enabled every kernel launch is bracketed by two HIP events on the
 * context's stream; bms_ctx_get_timing synchronises and returns the accumulated milliseconds and launch
 * counts per kernel class since the last reset. */
enum bms_kernel_tag {
  BMS_TAG_ROTATE = 0,          /* rotate_modes_resident_kernel (l <= 27), rotate_modes_mfma_kernel (l <= 33), rotate_modes_kernel */
  BMS_TAG_SETUP = 1,           /* pixel_sort_kernel, pixel_tables_kernel, swsh_kernel (synthesis matrix), bspline table kernels */
  BMS_TAG_GEMM_SYNTHESIS = 2,  /* modes -> grid: zgemm3m_mfma_kernel (with a boost), synthesis_split_kernel or
                                  theta_synthesis_mfma_kernel + phi_synthesis_folded_kernel (without one) */
  BMS_TAG_SPLINE_FORWARD = 3,  /* bspline_forward_modes_kernel, abd_mix_forward_kernel (slope form: spline_forward_kernel) */
  BMS_TAG_SPLINE_BACKWARD = 4, /* bspline_backward_eval_kernel (slope form: spline_backward_eval_kernel, spline_slopes_kernel) */
  BMS_TAG_GEMM_ANALYSIS = 5,   /* dgemm_mfma_kernel / zgemm3m_mfma_kernel, grid -> modes: phi-DFT of the unfused analysis, dense
                                  quadrature beyond n_theta = 104, a column part's analysis */
  BMS_TAG_POINTWISE = 6,       /* psi mixing / affine / Horner kernels, grid products, spline prefix sums and evaluation */
  BMS_TAG_THETA_QUADRATURE = 7, /* theta_quadrature_kernel (second step of the unfused separable analysis) */
  BMS_TAG_ANALYSIS_FUSED = 8,   /* analysis_split_kernel / analysis_fused_kernel (phi-DFT on MFMA + theta quadrature, one kernel) */
  BMS_TAG_ANALYSIS_LARGE = 9,   /* phi_dft_folded_kernel + theta_quadrature_mfma_kernel (grids with n_theta > 40) */
  BMS_TAG_COUNT = 10
};
int bms_ctx_enable_timing(bms_ctx* ctx, int on);
int bms_ctx_get_timing(bms_ctx* ctx, double ms[BMS_TAG_COUNT], int64_t calls[BMS_TAG_COUNT], int reset);

/* ---- rotation of modes ----------------------------------------------------------------------------------
 * replaces _rotate_decomposition_basis_by_constant (scri/rotations.py:346-367) and
 * _rotate_decomposition_basis_by_series (scri/rotations.py:370-392), including the Wigner-D evaluation
 * (sf._Wigner_D_matrices, rotations.py:327,381).  In place on data c16[n_times][ld], modes (l,m),
 * l = ell_min..ell_max at columns l(l+1) - ell_min^2 + m.
 */
int bms_rotate_const(bms_ctx* ctx, void* data, int mem, int64_t n_times, int64_t ld, int ell_min, int ell_max,
                     const double quaternion[4] /* w,x,y,z */);
int bms_rotate_series(bms_ctx* ctx, void* data, int mem, int64_t n_times, int64_t ld, int ell_min, int ell_max,
                      const void* spinors /* c16[n_times][2] = (w + i z, y + i x), same memory space as data */);
/* _rotate_decomposition_basis_by_constant at its own seam (scri/rotations.py:346-367): the caller hands the packed Wigner
 * matrices D (host c16, block l row-major (m', m) at sf._linear_matrix_offset(l, ell_min); rotations.py:327 fills it with
 * sf._Wigner_D_matrices), not the rotor.  data[t, l, m] <- sum_m' data[t, l, m'] D^l[m', m], in place. */
int bms_rotate_const_D(bms_ctx* ctx, void* data, int mem, int64_t n_times, int64_t ld, int ell_min, int ell_max,
                       const void* D_host);
/* packed Wigner-D matrices of one rotor, layout of sf.Wigner_D_matrices (tests/test_rotations.py:163-165):
 * host c16[(4 L^3 - L)/3 ...], block l at sf._linear_matrix_offset(l, ell_min), row-major (m', m) */
int bms_wigner_D(bms_ctx* ctx, const double quaternion[4], int ell_min, int ell_max, void* D_host);

/* ---- BMS transformation of WaveformModes ---------------------------------------------------------------
 * replaces WaveformGrid.from_modes + WaveformGrid.to_modes (scri/waveform_grid.py:331-613, 274-329), i.e.
 * WaveformModes.transform (scri/waveform_modes.py:705-719), after the kwargs have been parsed
 * (process_transformation_kwargs, scri/waveform_grid.py:20-127).
 */
typedef struct {
  int64_t n_times;
  const double* t;      /* host f8[n_times], strictly increasing */
  const void* data;     /* c16[n_times][ld] */
  int64_t ld;           /* row stride (complex elements) */
  int mem;              /* memory space of data, aux_data and of data_out */
  int ell_min, ell_max; /* modes present in data */
  int spin_weight;      /* SpinWeights[dataType], scri/__init__.py:81 */
  int conformal_weight; /* waveform_base.py:445-446 */
  int type_term;        /* enum bms_type_term */
  /* BMS_TERM_PSI only: the n_aux higher-index Weyl scalars psi_{n+1}..psi_4 (psi*_modes kwargs) */
  int n_aux;
  const void* aux_data[4]; /* c16[n_times][aux_ld[i]] */
  int64_t aux_ld[4];
  int aux_ell_min[4], aux_ell_max[4], aux_spin[4];
  double aux_coeff[4]; /* scipy.special.comb(5 - n, 5 - n_i), waveform_grid.py:547 */
  int aux_power[4];    /* n_i - n */
} bms_wm_input;

typedef struct {
  const void* supertranslation; /* host c16[(ell_max_supertranslation+1)^2], l from 0 */
  int ell_max_supertranslation; /* >= 1 (the reference pads to 4 modes) */
  double frame_rotation[4];     /* unit quaternion w,x,y,z */
  double boost_velocity[3];
  int n_theta, n_phi;           /* equiangular grid, theta includes both poles */
  int ell_max_out;              /* largest l of the output modes (smallest is |spin_weight|) */
} bms_transformation;

/* t_out: host f8[n_times]; data_out: c16[n_times][(ell_max_out+1)^2 - s^2] (caller allocates n_times rows);
 * the first *n_times_out rows are valid on return. */
int bms_transform_modes(bms_ctx* ctx, const bms_wm_input* in, const bms_transformation* tr, double* t_out,
                        void* data_out, int64_t* n_times_out);

/* Time-axis sharding (one process per GPU).  A rank holds rows [data_row0, data_row0 + data_rows) of the global
 * data (in->data / in->aux_data point at row data_row0; in->t and in->n_times stay GLOBAL) and produces the
 * output samples whose global input index lies in [out_i0, out_i1) (intersected with the valid window of
 * scri/waveform_grid.py:564-568).  bms_shard_plan returns the rows a rank must hold for that
 * (spline halo + boost/supertranslation time skew): need_rows = [first, one-past-last];
 * window = the global valid output index range [i_lo, i_hi). */
typedef struct {
  int64_t data_row0, data_rows;
  int64_t out_i0, out_i1;
  /* Pixel-column partition (SURVEY 8(e) plan B, for boosts whose time skew makes the row halo comparable to a time
   * shard): with col_parts > 1 the call synthesises, splines and analyses only part col_part of the grid columns and
   * data_out receives that part's CONTRIBUTION to every output sample (the analysis is linear in the grid columns):
   * the sum over the col_parts parts is the result, i.e. one reduce-scatter over ranks.  0 or 1: all columns. */
  int32_t col_part, col_parts;
} bms_shard;
int bms_shard_plan(bms_ctx* ctx, const double* t, int64_t n_times, const bms_transformation* tr, int64_t out_i0,
                   int64_t out_i1, int64_t need_rows[2], int64_t window[2]);
/* bms_transform_modes for HOST arrays as a three-stage pipeline over `pieces` time shards of the output range: upload of
 * shard k + 1, kernels of shard k and download of shard k - 1 run side by side on three streams (a long series in host memory
 * waits for PCIe, not for the kernels).  in->mem must be BMS_HOST, no auxiliary fields.  data_out: host
 * c16[window rows][n_out], t_out f8[window rows] (size them with bms_output_window); full rate needs page-locked arrays
 * (bms_host_alloc for the result, bms_host_register for a caller's input).  Same results as the sharded path. */
int bms_transform_modes_pipelined(bms_ctx* ctx, const bms_wm_input* in, const bms_transformation* tr, int pieces, double* t_out,
                                  void* data_out, int64_t* n_times_out);

/* The same pipeline for AsymptoticBondiData.transform (scri/asymptotic_bondi_data/transformations.py:205-423): host arrays in and
 * out, raw = c16[6][n_times][(ell_max+1)^2], raw_out = c16[6][n_times_out][(ell_max_out+1)^2] (best page-locked: bms_host_alloc /
 * bms_host_register); the rows of the six fields of one time shard travel up while the previous shard is transformed and the one
 * before it travels down.  u_out must hold n_times doubles.  Results equal bms_transform_abd's to rounding. */
int bms_transform_abd_pipelined(bms_ctx* ctx, const double* u, const void* raw, int64_t n_times, int ell_max,
                                const bms_transformation* tr, int pieces, double* u_out, void* raw_out, int64_t* n_times_out);

/* One process, several GPUs, host arrays in and out (the callers the reference has: scri/waveform_modes.py:705-719 and
 * scri/asymptotic_bondi_data/transformations.py:391-412 take numpy arrays in one process).  The output window is cut into `pieces`
 * exactly as the two calls above cut it; these transform pieces [piece0, piece1) only and put them at their place in t_out / data_out,
 * which are the arrays of the WHOLE window (page-locked memory from bms_host_alloc / bms_host_register is visible to every device);
 * *n_times_out is the whole window's row count.  The caller deals the pieces over one context per device and calls from one host
 * thread per context (the library is re-entrant per context): every device receives its own rows + halo at upload time, so nothing
 * travels GPU to GPU, and the result equals the one-context call with the same `pieces` bit for bit. */
int bms_transform_modes_pipelined_part(bms_ctx* ctx, const bms_wm_input* in, const bms_transformation* tr, int pieces, int piece0,
                                       int piece1, double* t_out, void* data_out, int64_t* n_times_out);
int bms_transform_abd_pipelined_part(bms_ctx* ctx, const double* u, const void* raw, int64_t n_times, int ell_max,
                                     const bms_transformation* tr, int pieces, int piece0, int piece1, double* u_out, void* raw_out,
                                     int64_t* n_times_out);
/* The dealing itself, for bindings that do not want to manage threads: ONE call takes the n_ctx contexts (one per device; created
 * by the caller with bms_ctx_create, each listed once), deals the `pieces` time shards over them in contiguous runs and runs
 * bms_transform_*_pipelined_part on one host thread per context.  Host arrays in and out as above.  On failure the status of the first
 * failing context is returned and bms_last_error(ctxs[0]) names it ("context k (device d): ..."). */
int bms_transform_modes_multi(bms_ctx* const* ctxs, int n_ctx, const bms_wm_input* in, const bms_transformation* tr, int pieces,
                              double* t_out, void* data_out, int64_t* n_times_out);
int bms_transform_abd_multi(bms_ctx* const* ctxs, int n_ctx, const double* u, const void* raw, int64_t n_times, int ell_max,
                            const bms_transformation* tr, int pieces, double* u_out, void* raw_out, int64_t* n_times_out);
/* Several series under one transformation: the extra trailing data dimensions of the reference's waveform objects
 * (scri/waveform_grid.py:299-308 `final_dim`, :574-594): every trailing index is an independent series on the same time axis.
 * Arrays are in the REFERENCE'S layout -- the trailing index fastest, as numpy stores data[N, n_modes, F]: in->data is
 * c16[n_times][in->ld] with element (mode, j) of a row at column mode * n_series + j (in->ld >= n_modes n_series); psi companions
 * likewise.  Exactly one of data_out -- c16[n_times][n_out n_series] -- and grid_out -- c16[n_times][n_theta n_phi n_series],
 * WaveformGrid.from_modes -- is non-NULL, same convention, in the memory space in->mem; the first *n_times_out rows are written.
 * The block crosses PCIe once as it is (the permutation to one block of columns per series and back runs on the device); time axis,
 * spline tables, per-direction tables and window are set up once and shared by the series. */
int bms_transform_modes_series(bms_ctx* ctx, const bms_wm_input* in, int n_series, const bms_transformation* tr, double* t_out,
                               void* data_out, void* grid_out, int64_t* n_times_out);
/* WaveformGrid.from_modes on its own (scri/waveform_grid.py:331-613): the first half of bms_transform_modes -- the field on the
 * boost-distorted grid at the new time slices, grid_out c16[n_times][n_theta * n_phi] (only the first *n_times_out rows are
 * written; grid order, theta-major), in the memory space in->mem.  bms_map2salm of it is WaveformGrid.to_modes (:274-329). */
int bms_modes_to_grid(bms_ctx* ctx, const bms_wm_input* in, const bms_transformation* tr, double* t_out, void* grid_out,
                      int64_t* n_times_out);
/* The valid output window [i_lo, i_hi) of a transformation (scri/waveform_grid.py:564-568; abd != 0: the
 * AsymptoticBondiData flavour, transformations.py:391-396), from the same per-direction tables the transformations
 * compute on the GPU: output sample r of bms_transform_modes / bms_transform_abd has input index i_lo + r, so a caller
 * can size its (device) output buffers exactly: i_hi - i_lo rows. */
int bms_output_window(bms_ctx* ctx, const double* t, int64_t n_times, const bms_transformation* tr, int abd,
                      int64_t window[2]);
/* as bms_transform_modes; t_out/data_out receive only the rank's samples, *first_index_out their first global
 * input index (output row r corresponds to input index *first_index_out + r).  shard == NULL: whole series. */
int bms_transform_modes_shard(bms_ctx* ctx, const bms_wm_input* in, const bms_transformation* tr,
                              const bms_shard* shard, double* t_out, void* data_out, int64_t* n_times_out,
                              int64_t* first_index_out);

/* ---- page-locked host memory -----------------------------------------------------------------------------
 * Results that go back to host arrays (mem = BMS_HOST) cross PCIe at full rate only into page-locked memory; a caller
 * that lets the library allocate its result arrays gets that without a staging copy (scri_amd/_lib.py: pinned_empty).
 * NULL on failure. */
void* bms_host_alloc(uint64_t bytes);
void bms_host_free(void* p);
/* The other direction: page-lock a caller's INPUT array in place (and release it), so that repeated transformations of
 * the same host array upload at PCIe rate instead of through the runtime's staging copy.  Registration costs about one
 * upload; scri_amd/engine.py registers an array the second time it sees it and unregisters when it is freed. */
int bms_host_register(void* p, uint64_t bytes);
int bms_host_unregister(void* p);

/* ---- BMS transformation of AsymptoticBondiData ---------------------------------------------------------
 * replaces AsymptoticBondiData.transform (scri/asymptotic_bondi_data/transformations.py:199-431) after
 * _process_transformation_kwargs (:8-97).  raw: c16[6][n_times][(ell_max+1)^2] = psi0..psi4, sigma
 * (scri/asymptotic_bondi_data/__init__.py:36-75).  tr->n_theta = tr->n_phi = 2 working_ell_max + 1;
 * tr->ell_max_out = output_ell_max.  raw_out: c16[6][n_times][(ell_max_out+1)^2] (field stride n_times rows).
 */
int bms_transform_abd(bms_ctx* ctx, const double* u, const void* raw, int mem, int64_t n_times, int ell_max,
                      const bms_transformation* tr, double* u_out, void* raw_out, int64_t* n_times_out);

/* Time-axis shard of the same transformation (cfg5 of BASELINE.json: one shard per GPU).  u and n_times stay GLOBAL;
 * raw holds rows [shard->data_row0, +data_rows) of every field, c16[6][data_rows][(ell_max+1)^2]; the call produces the
 * output samples whose global input index lies in [out_i0, out_i1) (intersected with the valid window,
 * transformations.py:391-396): u_out[*n_times_out], raw_out c16[6][out_i1 - out_i0][(ell_max_out+1)^2] (only the first
 * *n_times_out rows of each field are written), *first_index_out = global input index of output row 0.
 * bms_shard_plan gives the rows a shard must hold.  shard == NULL: whole series (== bms_transform_abd). */
int bms_transform_abd_shard(bms_ctx* ctx, const double* u, const void* raw, int mem, int64_t n_times, int ell_max,
                            const bms_transformation* tr, const bms_shard* shard, double* u_out, void* raw_out,
                            int64_t* n_times_out, int64_t* first_index_out);

/* ---- building blocks (exported for the parity tests and for the "next" rows of the scope table) --------- */
/* boosted_grid / R_j_k (transformations.py:100-148, waveform_grid.py:130-174): host f8[n_theta][n_phi][4] */
int bms_rotor_grid(bms_ctx* ctx, const double frame_rotation[4], const double boost_velocity[3], int n_theta,
                   int n_phi, double* rotors_host);
/* Does that rotor grid keep its rings?  Returns 1 and the ring colatitudes Theta_j (thetas_out[n_theta]) when every rotor is
 * frame_rotation * R(Theta_j, phi'_k) -- no boost (Theta_j = theta'_j), or a boost along the polar axis of the rotated grid, whose
 * aberration (waveform_grid.py:141-161) only moves whole rings -- and 0 otherwise; < 0 on bad arguments.  Host only (no context):
 * the test the transformations use to pick the separable synthesis, made on the rotors themselves. */
int bms_ring_colatitudes(const double frame_rotation[4], const double boost_velocity[3], int n_theta, int n_phi,
                         double* thetas_out);
/* conformal_factors (transformations.py:151-196) on the given rotors (host f8[n][4]): k = 1 / (gamma (1 - v.r)) with r the
 * direction the rotor takes z to, eth k / k (spin weight 1, c16[n]), 1/k and 1/k^3.  Host evaluation with the code the
 * transformations run per direction on the GPU (pixel_math.h); ctx may be NULL. */
int bms_conformal_factors(bms_ctx* ctx, const double boost_velocity[3], const double* rotors_host, int64_t n_rotors,
                          double* k, void* ethk_over_k, double* one_over_k, double* one_over_k_cubed);
/* sf.SWSH_grid(R, s, ell_max)[..., ell_min^2:] : host c16[n_rotors][(ell_max+1)^2 - ell_min^2].  ctx = NULL: host
 * evaluation of the header the kernel compiles (wigner.h; double-double recurrence, correctly rounded values). */
int bms_swsh_grid(bms_ctx* ctx, const double* rotors_host /* f8[n][4] */, int64_t n_rotors, int spin, int ell_min,
                  int ell_max, void* Y_host);
/* spinsfast.map2salm(grid[n_maps][n_theta][n_phi], s, ell_max)[..., ell_min^2:] (waveform_grid.py:303-307) */
int bms_map2salm(bms_ctx* ctx, const void* grid, int mem, int64_t n_maps, int n_theta, int n_phi, int spin,
                 int ell_min, int ell_max, void* modes_out);
/* sf.Modes.evaluate(R): out[t][p] = sum_k modes[t][k] sY_k(R_p) at arbitrary rotors -- the dense contraction of
 * scri/asymptotic_bondi_data/transformations.py:312-334, waveform_grid.py:475-484, bms_transformations.py:179 on its own.
 * modes: c16[n_rows][ld] with l = ell_min..ell_max; rotors: host f8[n_rotors][4]; out: c16[n_rows][n_rotors] in `mem`. */
int bms_evaluate_modes(bms_ctx* ctx, const void* modes, int mem, int64_t n_rows, int64_t ld, int spin, int ell_min,
                       int ell_max, const double* rotors_host, int64_t n_rotors, void* out);
/* spinsfast.salm2map(modes[n_maps][(ell_max+1)^2], s, ell_max, n_theta, n_phi) -> c16[n_maps][n_theta][n_phi] */
int bms_salm2map(bms_ctx* ctx, const void* modes, int mem, int64_t n_maps, int spin, int ell_max, int n_theta, int n_phi,
                 void* grid_out);
/* not-a-knot cubic spline through (x[n], y c16[n][n_cols]) evaluated at x_new[n_new] (all x host; y/out in
 * `mem`): scipy CubicSpline(x, y)(x_new) of waveform_base.py:964 / modes_time_series.py:90 */
int bms_cubic_spline(bms_ctx* ctx, const double* x, int64_t n, const void* y, int64_t ld, int64_t n_cols, int mem,
                     const double* x_new, int64_t n_new, void* out);

/* ---- "next" rows of the scope table: time-series calculus and grid products ------------- */
/* scipy CubicSpline(x, y).derivative(order) (order 1..3), the spline itself (0) or .antiderivative(-order) (order -1 .. -16;
 * zero at x[0]) evaluated at x_new (any order of samples): ModesTimeSeries.interpolate(new_time, derivative_order) and
 * .dot / .ddot / .int / .iint (scri/modes_time_series.py:72-126).  y: c16[n][ld], out: c16[n_new][n_cols], both in `mem`. */
int bms_spline_derivative(bms_ctx* ctx, const double* x, int64_t n, const void* y, int64_t ld, int64_t n_cols, int mem,
                          const double* x_new, int64_t n_new, int order, void* out);
/* The mode-space operators of spherical_functions.Modes / scri.ModesTimeSeries (scri/modes_time_series.py:7-139 and the
 * algebra it inherits: eth, ethbar, bar, real, imag, +, -, scalar factors, truncate_ell) as ONE map along the mode axis,
 *     out[t][j] = r_t ( ca_j op_a(a[t][ia_j]) + cb_j op_b(b[t][ib_j]) ),   op = identity or complex conjugation,
 * with per-column tables the caller derives from (l, m, s) (scri_amd/device_series.py): idx_* int32[n_cols] (-1: zero),
 * coef_* c16[n_cols], host arrays.  a c16[n_rows][ld_a], b (may be NULL) and out c16[n_rows][ld_out] live in `mem`;
 * row_scale f8[n_rows] (may be NULL) too.  This is what keeps the charge and frame-fixing loops of
 * scri/asymptotic_bondi_data/bms_charges.py:14-286 resident in HBM between transformations. */
int bms_mode_map(bms_ctx* ctx, void* out, int64_t ld_out, int64_t n_rows, int n_cols, const void* a, int64_t ld_a,
                 const int32_t* idx_a, const void* coef_a, int conj_a, const void* b, int64_t ld_b, const int32_t* idx_b,
                 const void* coef_b, int conj_b, const double* row_scale, int mem);
/* WaveformBase.norm (scri/waveform_base.py:19-35,535-551): out[t] = sum_j |data[t][j]|^2, or its square root with take_sqrt
 * (complex_array_norm / complex_array_abs), the terms re^2 + im^2 added one at a time in column order, no fused multiply-adds --
 * the reference's loop, so the sums agree with it to the bit (and with them the parity-violation measures,
 * scri/waveform_modes.py:769-778 ...).  data c16[n_rows][ld] and out f8[n_rows] live in `mem`. */
int bms_row_norm(bms_ctx* ctx, const void* data, int64_t ld, int64_t n_rows, int n_cols, int mem, int take_sqrt, double* out);
/* ModesTimeSeries.grid_multiply (scri/modes_time_series.py:142-202): modes a (spin_a, l = 0..ell_max_a,
 * c16[n_times][(ell_max_a+1)^2]) and b likewise are evaluated on the (2 working_ell_max + 1)^2 equiangular grid
 * (spinsfast.salm2map), multiplied there, and the product is analysed (map2salm, spin spin_a + spin_b) into
 * out c16[n_times][(output_ell_max+1)^2].  Spin weights of the factors and of the product up to +-4 (the boost flux of
 * scri/flux.py multiplies ethbar h, s = -3, with its conjugate). */
int bms_grid_multiply(bms_ctx* ctx, const void* a, int spin_a, int ell_max_a, const void* b, int spin_b, int ell_max_b, int mem,
                      int64_t n_times, int working_ell_max, int output_ell_max, void* out);

/* ---- what feeds the rotation path -------------------------------------------------------- */
/* <Ldt> (f8[n][3]), <LL> (f8[n][3][3]) and the angular velocity omega = -<LL>^-1 <Ldt> (f8[n][3]) of a waveform from its
 * modes data c16[n][ld] (l = ell_min..ell_max) and their cubic-spline time derivative: LdtVector / LLMatrix /
 * angular_velocity of scri/mode_calculations.py:46-57, 298-313, 403-432.  Outputs are host arrays; any may be NULL. */
int bms_angular_velocity(bms_ctx* ctx, const double* t, int64_t n_times, const void* data, int64_t ld, int ell_min, int ell_max,
                         int mem, double* ldt_out, double* ll_out, double* omega_out);

/* Frame R[n][4] with R[0] = R0 and dR/dt = (1/2) Omega R, Omega(t) = not-a-knot cubic spline through omega[n][3]:
 * quaternion.integrate_angular_velocity as used by corotating_frame (scri/mode_calculations.py:435-491).  Host routine
 * (sequential in time); ctx may be NULL.  tolerance: absolute tolerance of the integration (<= 0: 1e-12). */
int bms_integrate_angular_velocity(bms_ctx* ctx, const double* t, int64_t n_times, const double* omega, const double R0[4],
                                   double tolerance, double* R_out);

/* ---- bit transforms of the storage formats (scri/utilities.py:194-406), bit-exact ------------ */
/* xor_timeseries (reverse = 0) / xor_timeseries_reverse (reverse = 1), in place: data viewed as uint64[n_rows][words_per_row],
 * time along the rows; row 0 is unchanged, row i becomes row[i-1] ^ row[i] (forward) or the running XOR (reverse). */
int bms_xor_timeseries(bms_ctx* ctx, void* data, int mem, int64_t n_rows, int64_t words_per_row, int reverse);
/* multishuffle(shuffle_widths, forward)(a): n elements of sum(widths) in {8,16,32,64} bits; widths from the most significant
 * piece down.  in and out must not overlap. */
int bms_multishuffle(bms_ctx* ctx, const void* in, void* out, int mem, int64_t n, const int* widths, int n_widths, int forward);
/* fletcher32(data): 16-bit words, modulus 65535; returns c1 << 16 | c0 */
int bms_fletcher32(bms_ctx* ctx, const void* data, int mem, int64_t n_bytes, uint32_t* checksum);

#ifdef __cplusplus
}
#endif
#endif /* SCRI_AMD_H */
