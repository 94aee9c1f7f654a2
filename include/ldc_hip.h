/*
 * ldc_hip.h -- C ABI of libldc_hip.so: the MI355X (gfx950) hot path of the Chebyshev
 * P_N-P_{N-2} artificial-compressibility lid-driven-cavity solver.
 *
 * The reference has no FFI: its plugin boundary is the Python class
 * `solvers.spectral.sg.SGSolver` (reference src/solvers/spectral/sg.py:29) driven by
 * `LidDrivenCavitySolver.solve` (src/solvers/base.py:202).  This header is the boundary a
 * maintainer binds underneath that class (ctypes stub: INTEGRATION.md).  Each entry point
 * names the reference lines whose work it replaces.
 *
 * Conventions
 *  - every pointer in `ldc_problem` is a DEVICE pointer into caller-owned memory
 *    (torch.float64 / torch.int32 tensors on the host side); the library allocates no
 *    device memory; its only process-wide state is a mutex and one private stream per device for graph captures
 *    and creation-time copies (rare operations, serialised; the launches take no lock).  A solver handle owns only its captured
 *    hipGraph executables.  It belongs to the HIP device that is current when it is created
 *    (dynamic-LDS attributes of its kernels are set there); using it with another device
 *    current returns LDC_E_STATE.
 *  - grids: M nodes per axis; independent x / y grids (reference sg.py:103-119: nx != ny; no shipped
 *    configuration uses it) through ldc_problem::Mx / My below (launch path, one-XCD and chip-wide kernel).  Lx != Ly is supported.
 *  - all 2-D arrays are row-major LD x LD doubles, zero padded, element [ix][iy]
 *    (reference sg.py:108, indexing="ij"); LD is a multiple of 16 and >= 16*T + 16.
 *  - "transposed copy" XT means XT[iy][ix] = X[ix][iy]; the kernels keep both so that
 *    every product on the path is of the NT form C[i][j] = sum_k X[i][k] Y[j][k].
 *  - "packed twin" XK: the same LD x LD values as X, stored as (LD/16)^2 blocks of 16 x 16 doubles,
 *    block (R, G) (rows 16R.., columns 16G..) at XK + (R*(LD/16) + G)*256, and inside a block in the
 *    order one f64 MFMA operand load consumes them: element (r, 4c+s) at (16c + r)*4 + s.  A wave's
 *    operand fragment is then 2 KB contiguous (lane l reads 32 bytes at 32 l); read from the row-major
 *    array the same fragment is 16 rows x 64 bytes per instruction and is delivered 3.6x slower
 *    (profiles/r01_aql_probe.log).  Every array that feeds the MFMAs of the iteration loop has a
 *    twin; the kernels keep both forms in step, the host packs what it uploads (ldc_pack).
 *  - functions return 0 on success, a negative LDC_E_* code for argument errors or a
 *    positive hipError_t.  Nothing throws; nothing synchronises unless stated.
 *  - `stream` is a hipStream_t passed as void*.
 */
#ifndef LDC_HIP_H
#define LDC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LDC_ABI_VERSION 7

#define LDC_E_ARG      (-1)  /* null pointer / inconsistent geometry */
#define LDC_E_STATE    (-2)  /* handle not valid for the call */
#define LDC_E_NODEVICE (-3)  /* no HIP device / wrong architecture */
#define LDC_E_SYNC     (-4)  /* a persistent kernel (modes 3, 5) gave up a bounded wait for another work-group (ldc_solver_status) */

/* slots of one history record written per iteration (ldc_problem.rec) */
enum {
  LDC_REC_REL = 0,  /* max(|du|/(|u_prev|+1e-12), same for v)       base.py:250-258 */
  LDC_REC_RU  = 1,  /* ||R_u||_2 of the stage-4 residual, all nodes  sg.py:463-473  */
  LDC_REC_RV  = 2,
  LDC_REC_RP  = 3,
  LDC_REC_E   = 4,  /* kinetic energy                                sg.py:495-508  */
  LDC_REC_Z   = 5,  /* enstrophy                                     sg.py:524-533  */
  LDC_REC_P   = 6,  /* palinstrophy                                  sg.py:535-550  */
  LDC_REC_DT  = 7,  /* pseudo time step used by this iteration       sg.py:387-408  */
  LDC_REC_LEN = 8
};

/* ctrl[] (int32) slots */
enum {
  LDC_CTRL_DONE      = 0, /* latch: 1 converged, 2 non-finite, 3 set by the HOST (a batched trial reached its own
                             iteration cap); any non-zero value turns later launches into no-ops */
  LDC_CTRL_ITER      = 1, /* iterations finalized (records written)                     */
  LDC_CTRL_STEP      = 2, /* stage-4 state updates completed                            */
  LDC_CTRL_FLUSHED   = 3, /* records whose Z/P slots have been folded                   */
  LDC_CTRL_PDONE     = 4, /* state index whose Z/P partial sums are complete            */
  LDC_CTRL_DROWS     = 5, /* rows of those partial-sum slabs                            */
  LDC_CTRL_LIVE      = 6, /* 1 when the last stage-4 launch updated the state (0: it was latched, the post launch
                             behind it has nothing to transform)                        */
  LDC_CTRL_LEN       = 8
};

/* scal[] (double) slots */
enum {
  LDC_SCAL_DT   = 0,  /* step for the NEXT iteration (device-resident, sg.py:427)       */
  LDC_SCAL_UMAX = 1,
  LDC_SCAL_VMAX = 2,
  LDC_SCAL_LEN  = 8
};

#define LDC_NPART 12   /* doubles per work-group in `partials` */

/* sync[] (uint32) slots of the persistent kernels */
enum {
  LDC_SYNC_GIVEUP = 96,  /* set to 1 by a work-group whose bounded wait for another one ran out     */
  LDC_SYNC_XLAUNCH = 128, /* small-N trial kernel (mode 3): the launch words of a single-trial launch (tickets per XCD,
                             the trial each XCD slot took) */
  LDC_SYNC_XFLAGS = 2048, /* its hand-over flags: 32 work-groups, each flag on a 128-byte line of its own */
  LDC_SYNC_XRING  = 8192, /* its scratch for the boundary ring of grad p: 25 tiles x 64 doubles */
  LDC_SYNC_WFLAGS = 16384, /* chip-wide trial kernel (mode 5): hand-over flags, 256 work-groups, each on a 128-byte line of its own */
  LDC_SYNC_WRING  = 24576, /* its scratch for the boundary ring of grad p: 256 tiles x 64 doubles, then the four ring vectors */
  LDC_SYNC_LEN    = 98304
};

typedef struct ldc_problem {
  /* geometry */
  int32_t M;      /* nodes per axis, N+1                                               */
  int32_t LD;     /* leading dimension of every padded array                           */
  int32_t T;      /* 16x16 tiles per axis done on MFMA: ceil((M-1)/16)                 */
  int32_t tail;   /* 1 when 16*T == M-1: index M-1 is handled by rank-1/edge paths (needs
                     T <= 16); T = ceil(M/16), tail = 0 is always valid                 */
  /* physics / control (reference conf/solver/spectral/sg.yaml, conf/config.yaml)      */
  double nu;          /* 1/Re                                                          */
  double beta2;       /* beta_squared                                                  */
  double cfl;         /* CFL                                                           */
  double hx_min, hy_min; /* min node spacing, sg.py:118-119                            */
  double lid_speed;   /* lid_velocity (lower bound of u_max, sg.py:396)                */
  double tol;         /* convergence tolerance on LDC_REC_REL                          */
  int32_t warmup;     /* iterations without convergence test (10, base.py:264,283)     */
  int32_t nan_guard;  /* latch on a non-finite change norm (quirk Q6)                  */
  int32_t stage_pressure; /* 0: SG (quirk Q1, grad p^n in all stages); 1: FSG smoother (multigrid/fsg.py:880) */
  int32_t rec_cap;    /* capacity of `rec` in records                                  */
  /* operators, read-only: row-major LD x LD, zero padded                               */
  const double *Dx, *D2x, *Dy, *D2y;   /* sg.py:188-193                                */
  const double *IxF, *GxF;  /* Interp_x embedded in full indexing; Dx @ IxF  (sg.py:209, 270-275) */
  const double *IyF, *GyF;  /* Interp_y embedded;                  Dy @ IyF  (sg.py:210, 270-276) */
                            /* Rows 1 .. M-2 of IxF / IyF must be UNIT rows (the interpolant through the inner nodes,    */
                            /* evaluated at an inner node, is that node's value: the reference's matrices are, to        */
                            /* rounding; the host sets them exactly): mode 3 uses d/dx p = GxF p off the ring columns.   */
  const double *wx, *wy;    /* quadrature weights, length LD       (sg.py:489-490)     */
  const double *ulid;       /* lid profile u_lid(x_i), length LD   (corner.py:80-112)  */
  const double *DxL, *D2xL, *DyL, *D2yL; /* column M-1 of Dx, D2x, Dy, D2y as contiguous length-LD vectors */
  /* state phi^n and its transposed copies                                              */
  double *U, *UT, *V, *VT, *P;
  /* RK stage buffers (sg.py:438-442), ping-pong A/B.  Of the velocity buffers the kernels maintain only the
     packed twins (UAK ...); the row-major forms below are read at index M-1 only (boundary values, written by
     the host) and are otherwise stale.  PA / PB are maintained in both forms.            */
  double *UA, *UAT, *VA, *VAT, *PA;
  double *UB, *UBT, *VB, *VBT, *PB;
  /* pressure path: T1T = (P IyF^T)^T, T2T = (P GyF^T)^T, grad p on the full grid       */
  double *T1T, *T2T, *PX, *PY;
  /* diagnostics: vorticity and its transpose                                           */
  double *W, *WT;
  /* packed twins (see Conventions) of the operators, the state, the stage buffers, the pressure
     transforms and the vorticity                                                        */
  const double *DxK, *D2xK, *DyK, *D2yK, *IxFK, *GxFK, *IyFK, *GyFK;
  double *UK, *UTK, *VK, *VTK, *PK;
  double *UAK, *UATK, *VAK, *VATK, *PAK;
  double *UBK, *UBTK, *VBK, *VBTK, *PBK;
  double *T1TK, *T2TK, *WK, *WTK;
  /* reductions / control                                                               */
  double  *partials;  /* 5 slabs of partials_stride doubles: stage 4 | Z parity 0,1 | P parity 0,1 */
  int64_t  partials_stride; /* doubles between slabs                                    */
  double  *scal;      /* LDC_SCAL_LEN doubles                                           */
  int32_t *ctrl;      /* LDC_CTRL_LEN int32                                             */
  double  *rec;       /* rec_cap * LDC_REC_LEN doubles, ring indexed by iteration       */
  uint32_t *sync;     /* LDC_SYNC_LEN uint32 (256-byte aligned) for the persistent trial kernel, or NULL:
                         the launch-per-stage path is then the only one                  */
  /* independent x / y grids (reference sg.py:103-119: nx != ny).  0 = M.  M is then the LARGER of the two node  */
  /* counts -- the tiling, LD and every padded array are built for it, tail = 0 --, the operators and vectors of  */
  /* the shorter axis are zero padded like everything else; the launch-per-stage path, the one-XCD kernel (mode 3) */
  /* and the chip-wide kernel (mode 5, index M-1 inside the tiles) run such grids; the trial-per-CU kernel (mode  */
  /* 4) answers LDC_E_ARG.                                                                                        */
  int32_t Mx, My;     /* nodes along x (first index) / y (second index)                                           */
} ldc_problem;

typedef struct ldc_solver ldc_solver;   /* opaque */

/* library / device -------------------------------------------------------------------- */
int         ldc_version(void);
const char *ldc_error_string(int code);
/* fills arch[] with the gcnArchName of the current device; 0 if it is gfx950            */
int         ldc_device_check(char *arch, int arch_len);

/* solver handle ------------------------------------------------------------------------ */
/* validates the description and copies it; no device work                               */
int ldc_solver_create(const ldc_problem *desc, ldc_solver **out);
int ldc_solver_destroy(ldc_solver *s);

/* one RK stage (k = 0..3) of SGSolver.step: residual + update + BCs fused               */
/* replaces sg.py:434-447 (_compute_residuals :278-346, axpy :438-446, BCs :348-385)     */
/* k | 16 launches the variant that also carries the fused diagnostics (stages 0 and 1 only)   */
int ldc_stage(ldc_solver *s, int k, void *stream);
/* T1T/T2T from the current pressure (first half of _interpolate_pressure_gradient,     */
/* sg.py:270); `which` 0: P, 1: PA, 2: PB                                                */
int ldc_pressure_transform(ldc_solver *s, int which, void *stream);
/* vorticity, enstrophy and palinstrophy partial sums of the current state               */
/* replaces sg.py:510-550 (called per iteration at base.py:274-276)                      */
int ldc_diagnostics(ldc_solver *s, void *stream);
/* E, Z, P of the current state into out3[0..2] (device pointer); one-off form of the        */
/* per-iteration diagnostics, sg.py:495-550 (used where the reference calls them once,       */
/* e.g. fsg.py:120-123)                                                                      */
int ldc_global_quantities(ldc_solver *s, double *out3, void *stream);
/* reduces the partial sums: change norms, residual norms, E/Z/P, next dt, latch, record */
/* replaces sg.py:387-408, base.py:250-258, :283-286, sg.py:463-473                      */
int ldc_finalize(ldc_solver *s, int with_diagnostics, void *stream);
/* computes dt for the first iteration from the current state (sg.py:387-408)            */
int ldc_prime(ldc_solver *s, void *stream);

/* n_iters iterations of base.py:243-313 enqueued on `stream` (hipGraph replay where     */
/* possible); with_diagnostics=0 gives the step()-only loop.  Does not synchronise.      */
int ldc_solver_enqueue(ldc_solver *s, int n_iters, int with_diagnostics, void *stream);
/* iterations captured per graph (default 64); must be set before the first enqueue      */
int ldc_solver_set_graph_iters(ldc_solver *s, int iters_per_graph);
/* How ldc_solver_enqueue runs the loop.  mode 0: one launch per RK stage (hipGraph replay).                       */
/* (modes 1 and 2 -- the round-2 persistent trial kernel -- were removed with ABI 7: LDC_E_ARG.)                       */
/* -1 (default): mode 3 where it applies and ceil(M/16)^2 <= LDC_XCD_AUTO_TILES, else mode 5 where it applies, else mode 0.  */
/* mode 3: the small-N trial kernel (csrc/ldc_xcd_kernel.inc) -- ALL n_iters iterations in one launch, the trial's     */
/* ceil(M/16)^2 work-groups on ONE XCD elected at run time, one contraction family per wave over the full contraction   */
/* index, the operator fragments resident in registers, state exchanged through that XCD's L2, the pressure path and    */
/* the fold of the partial sums on waves of their own.  Needs desc->sync and ceil(M/16)^2 <= LDC_XCD_TILES (M <= 80),    */
/* else LDC_E_ARG.  Reads phi^n from the row-major arrays and leaves row-major and packed forms behind, so it can be     */
/* mixed with the launch path; trajectories agree with the launch path to rounding, not bit for bit (one accumulation   */
/* chain per contraction instead of four K-quarters).  In mode -1 it is chosen when ceil(M/16)^2 <= LDC_XCD_AUTO_TILES.  */
/* A batch (ldc_batch_enqueue) whose trials all resolve to mode 3 runs every trial on an XCD of its own in one launch.   */
#define LDC_XCD_TILES 25
#define LDC_XCD_AUTO_TILES 25      /* measured faster than the launch path at every size it applies to (profiles/r03_xcd_ab.log) */
/* mode 4: the trial-per-CU kernel (csrc/ldc_cu_kernel.inc) -- ONE work-group advances the trial: the stage state and the  */
/* operators live in the CU's LDS, one wave per 16 x 16 tile runs all contractions of its tile, no hand-over between        */
/* work-groups at all.  Needs M <= LDC_CU_MAX_M (the LDS), else LDC_E_ARG.  For one trial it is slower than mode 3 (one CU   */
/* instead of up to 25); its place is the BATCH: ldc_batch_enqueue advances all trials of a batch at once with it, one CU   */
/* each (256 at a time), when every trial asked for mode 4, or in mode -1 from LDC_CU_AUTO_TRIALS(_T3, _M33) trials on.  Same    */
/* entry and exit state as mode 3 (row-major phi^n in, row-major and packed forms out); trajectories agree with the other   */
/* paths to rounding.                                                                                                       */
#define LDC_CU_MAX_M 44
/* mode 5: the chip-wide trial kernel (csrc/ldc_wide_kernel.inc) -- ALL n_iters iterations in one launch, the trial's T x T      */
/* work-groups ONE PER CU ON ALL XCDs (T = ceil(M/16), 6 <= T <= 16; for M = 16 T + 1 and the SG loop the tail layout: index M-1   */
/* outside the tiles, N = 96, 112 ... 256 on T x T work-groups), the four operator panels of a tile resident in LDS, one           */
/* contraction family per wave over the full contraction index, state exchanged through the fabric as write-through tiles +       */
/* one flag per work-group handed to the 2T-1 work-groups of its tile row and column, the fold of the partial sums beside the     */
/* contractions of stage 1.  Needs desc->sync and T*T <= CUs, all of them free (one work-group per CU must be resident at once:   */
/* LDC_E_SYNC after a bounded wait otherwise); the smoother at M = 16 T + 1 = 257 has no such form (LDC_E_ARG).  Entry and exit    */
/* state as mode 3; trajectories agree with the other paths to rounding.  Chosen in mode -1 wherever it applies and mode 3 does   */
/* not (N = 81 ... 256).                                                                                                          */
/* (measured, profiles/r03_cu_ab.log: the trial-per-CU kernel scales with the number of trials up to 256, the small-N kernel    */
/*  saturates where its XCDs are full -- 64 trials at ceil(M/16) <= 2 (5.7 M trial-iterations/s), 8 ... 24 above (1.9 M); the    */
/*  thresholds are where the two lines cross: profiles/r03_cu_ab_thresholds.log)                                                                                     */
#define LDC_CU_AUTO_TRIALS 80        /* ceil(M/16) <= 2                                                  */
#define LDC_CU_AUTO_TRIALS_T3 80     /* ceil(M/16) == 3                                                  */
#define LDC_CU_AUTO_TRIALS_M33 32    /* M == 33 (N = 32): four tile waves + two edge + two helper waves, not nine tile waves */
int ldc_solver_set_persistent(ldc_solver *s, int mode);
/* the mode ldc_solver_enqueue will really use for more than one iteration (0, 1, 2, 3 or 4): what set_persistent asked   */
/* for, resolved against what the handle's size and device allow.  A host that drives several streams uses it to keep  */
/* launches that need co-resident work-groups (modes 3, 5) from overlapping each other.                                */
int ldc_solver_mode(ldc_solver *s);
/* 0, or LDC_E_SYNC when a persistent launch of this handle gave up a bounded wait (a work-group was not         */
/* resident): the state is then undefined.  Reads one word of desc->sync on the host, through the library's      */
/* private stream: the CALLER has waited for the stream its launches ran on (no device-wide synchronise here: a  */
/* sweep drives the library from several host threads, and HIP refuses that call while another thread captures).  */
int ldc_solver_status(ldc_solver *s);
/* compute units and XCDs of the current device as the library counts them (what modes 3 / 5 size their launches by) */
int ldc_device_info(int *n_cus, int *n_xcds);
/* how many times this process set its kernels' attributes (dynamic LDS above 64 KiB): once per device that has had a   */
/* handle, never again at later creates -- a host with several threads relies on that                    */
int ldc_attribute_rounds(void);

/* batched trials (the sweep axis of the reference on ONE GPU): n_trials solver handles of      */
/* identical geometry (M, LD, mode) advanced by the same launches, blockIdx.y = trial; each     */
/* keeps its own state, dt, latch and history.  `workspace` = caller-owned DEVICE memory,        */
/* 256-byte aligned, at least ldc_batch_workspace_bytes(n_trials) bytes, alive as long as the    */
/* batch; it receives the per-trial kernel argument blocks by synchronous copies on a stream of  */
/* the library's own, which first waits (an event) for everything `stream` -- the stream the     */
/* caller last used on that memory and will launch the batch on -- holds at the time of the call */
/* (since ABI 7; until then it was the caller's duty to have nothing pending there).             */
typedef struct ldc_batch ldc_batch;
size_t ldc_batch_workspace_bytes(int n_trials);
int ldc_batch_create(ldc_solver *const *solvers, int n_trials, void *workspace, size_t workspace_bytes, void *stream,
                     ldc_batch **out);
int ldc_batch_destroy(ldc_batch *b);
/* n_iters iterations of base.py:243-313 for every trial that is not latched yet                 */
int ldc_batch_enqueue(ldc_batch *b, int n_iters, int with_diagnostics, void *stream);
/* how ldc_batch_enqueue will advance more than one iteration: 0 shared launches per stage (blockIdx.y = trial), 3 the        */
/* small-N kernel (every trial on an XCD of its own), 4 the trial-per-CU kernel (one work-group per trial)                    */
int ldc_batch_mode(ldc_batch *b);

/* packed twin of a row-major LD x LD array (both device pointers, src != dst); used by the host   */
/* after it uploads or edits operators / state                                                     */
int ldc_pack(const double *src, double *dst, int LD, void *stream);

/* debugging aid for parity tests: one residual evaluation of state `which`              */
/* (0: U/V, 1: UA/VA, 2: UB/VB) with every intermediate written to LD x LD arrays:       */
/* out[0..9] = du_dx, du_dy, dv_dx, dv_dy, lap_u, lap_v, dp_dx, dp_dy, R_u, R_v; out[10] = R_p */
int ldc_residual_debug(ldc_solver *s, int which, double *const out[11], void *stream);

/* generic fp64 MFMA product used by the stream-function solve -------------------------- */
/* C[i][j] = sum_{k<K16} A[i][k] * B[j][k]  (i,j < R16; all LD-strided, zero padded);     */
/* scale_mode 0: none; 1: C /= (lam_r[i] + lam_c[j]); transpose_out: store C^T            */
int ldc_gemm_nt(const double *A, const double *B, double *C, int R16, int K16, int LD,
                int transpose_out, int scale_mode, const double *lam_r, const double *lam_c,
                void *stream);
/* psi from A Psi + Psi B^T = F by fast diagonalisation (replaces the sparse LU of        */
/* sg.py:556-619): F, work, out are LD x LD padded interior-indexed arrays               */
int ldc_poisson_fastdiag(const double *Qx, const double *Qxinv, const double *Qy,
                         const double *Qyinv, const double *lamx, const double *lamy,
                         const double *F, double *work0, double *work1, double *Psi,
                         int Mi, int LD, void *stream);
/* argmin psi, argmax |omega| and the three corner maxima (sg.py:621-709).               */
/* out_val[5], out_idx[5] (flat index ix*LD+iy): 0 primary, 1 |omega| max, 2 BR, 3 BL, 4 TL */
int ldc_vortex_extrema(const double *Psi, const double *W, const double *x, const double *y,
                       int M, int LD, double *out_val, int32_t *out_idx, void *stream);
/* the same on an Mx x My node grid (nx != ny)                                             */
int ldc_vortex_extrema_xy(const double *Psi, const double *W, const double *x, const double *y,
                          int Mx, int My, int LD, double *out_val, int32_t *out_idx, void *stream);

/* Timing experiments.  The switches exist only in the INSTRUMENTED build of the library (csrc/ldc_kernels.hip       */
/* compiled with -DLDC_TIMING -> lib/libldc_hip_timing.so, loaded by tools/kbench.py, kstamps.py, pstamps.py,          */
/* ab_masks.py); in the product library the kernels carry no switch and both calls return LDC_E_STATE.                 */
/* ldc_timing_build() says which build is loaded (1 instrumented, 0 product).                                          */
/* Instrumented build: results are WRONG while bits 0-5 are set: bit 0 skips the MFMAs, bit 1 the operand loads of    */
/* the stage kernel; bits 7 / 8 (128 / 256) force plain / write-through state stores (results stay right; takes       */
/* effect for launches and graphs built afterwards); bits 4 / 9-12 (16 / 512, 1024, 2048, 4096) switch parts of the   */
/* index-(M-1) jobs off: all / K-loop part / epilogue part / LDS-direct rows / per-group dot products (the recorded    */
/* |R|, Z, P then miss those nodes)                                                                                    */
int ldc_debug_ablate(ldc_solver *s, int mask);
/* Instrumented build: with mask bit 64 set every wave of the stage kernel writes seven cycle stamps
 * (s_memtime) to buf[((block * 8 + wave) * 8 + point)]; buf holds T*T*64 doubles.  NULL switches off. */
int ldc_debug_stamps(ldc_solver *s, double *buf);
int ldc_timing_build(void);

/* hardware self-test: D = A(16x4) * B(4x16) with the f64 MFMA; used by tests to pin the  */
/* operand / result lane maps                                                             */
int ldc_mfma_selftest(const double *A, const double *B, double *D, void *stream);
/* fp64 MFMA issue-rate micro-benchmark: runs `iters` x 8 independent MFMAs per wave on    */
/* every SIMD; returns nothing (time it with events); flops = grid*4*iters*8*2048          */
int ldc_mfma_peak(double *sink, int iters, int grid, void *stream);

/* worker streams for sweeps (solvers.spectral.batched.run_concurrently): a non-blocking HIP stream of the given   */
/* priority on the current device.  HIP has three priority levels here (ldc_stream_priority_range: least 1,        */
/* greatest -1), torch offers two of them; streams of DIFFERENT priority never share a hardware queue, which is     */
/* what lets their launches overlap reliably.  The caller owns the stream (wrap it, e.g. torch.cuda.ExternalStream) */
/* and destroys it when no work is pending on it.                                                                   */
int ldc_stream_priority_range(int *least, int *greatest);
int ldc_stream_create(int priority, void **stream);
int ldc_stream_destroy(void *stream);

#ifdef __cplusplus
}
#endif
#endif /* LDC_HIP_H */
