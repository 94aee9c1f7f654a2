// ldc_kernels.hip -- gfx950 (MI355X) kernels + C ABI of libldc_hip.so.
//
// Hot path of the Chebyshev P_N-P_{N-2} artificial-compressibility lid-driven-cavity
// solver (reference: src/solvers/spectral/sg.py, src/solvers/base.py:202-330), written
// for CDNA4: fp64 MFMA (v_mfma_f64_16x16x4_f64), 64-wide waves, one work-group per 16x16
// output tile -- 8 waves in the RK stage kernel (2 per SIMD, split by derivative direction and by
// quarter of the contraction index), 4 in the post / diagnostics kernels (one per SIMD).
//
// Layout recap (include/ldc_hip.h): every array is LD x LD doubles, row-major, zero
// padded, element [ix][iy]; each field has a transposed copy.  With that, EVERY product
// on the path is of the "NT" form   C[i][j] = sum_k X[i][k] * Y[j][k].  The MFMA operands of
// the iteration loop are read from PACKED TWINS (16 x 16 blocks in operand order, ldpk below):
// a wave's fragment is then 2 KB contiguous; fetched from the row-major form (16 rows x 64 bytes
// per instruction) the same bytes arrive 3.6x slower (profiles/r01_aql_probe.log).
//
//   d/dx  (Dx @ U)[i][j]   = sum_k Dx[i][k] * UT[j][k]
//   d/dy  (U @ Dy^T)[i][j] = sum_k U [i][k] * Dy[j][k]
//
// Forms of the iteration loop (include/ldc_hip.h, ldc_solver_set_persistent):
//   launch per RK stage  stage_kernel x 4 + post_kernel (finalize block first, pressure transforms), hipGraph replays: every size
//   small-N kernel       ldc_xcd_kernel.inc: all iterations of a chunk in one launch, the trial's tiles on ONE XCD (N <= 79)
//   trial-per-CU kernel  ldc_cu_kernel.inc: one work-group = one trial, batches of many small trials (M <= 44)
//   chip-wide kernel     ldc_wide_kernel.inc: one launch, T x T work-groups one per CU on all XCDs (N = 81 ... 256)
//
// MFMA lane maps (v_mfma_f64_16x16x4_f64; pinned by tests/test_gpu_parity.py::test_mfma_lane_maps):
//   A: lane l holds A[row l&15][k l>>4]      B: lane l holds B[k l>>4][col l&15]
//   D: lane l, reg r holds D[row (l>>4)+4r][col l&15]
// A lane loads 4 consecutive k (k0+4q .. k0+4q+3, q = l>>4) and feeds element s to the
// s-th of four MFMAs, i.e. k-step s contracts k = k0 + 4q + s: any bijection of k works
// as long as A and B use the same one.

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <mutex>
#include <new>
#include <type_traits>
#include <vector>

#include "ldc_hip.h"

namespace {

typedef double v4d __attribute__((ext_vector_type(4)));

#define MFMA_F64(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64((a), (b), (c), 0, 0, 0)

// Timing switches and cycle stamps (ldc_debug_ablate / ldc_debug_stamps) exist only in the INSTRUMENTED build of this
// file (-DLDC_TIMING -> lib/libldc_hip_timing.so, what tools/kbench.py, kstamps.py, pstamps.py and ab_masks.py load).  In the
// product library the argument blocks have no `ablate` field, LDC_ABL folds to 0 at compile time, the stamp macros are
// empty and the two debug entry points return LDC_E_STATE.
#ifdef LDC_TIMING
#define LDC_ABL(x, bits) ((x).ablate & (bits))
#else
#define LDC_ABL(x, bits) 0
#endif

constexpr int kWaves = 4;          // waves per work-group == K-split factor
constexpr int kThreads = 256;
// launch bounds of the 256-thread kernels that issue MFMAs: told that a SIMD carries one wave, hipcc puts MFMA accumulators into
// AGPRs and copies them to VGPRs and back around the loop (tools/probes/mfma_rate_probe.hip: 140 instead of 64 cycles per MFMA)
constexpr int kMfmaBounds = 512;

// ---------------------------------------------------------------------------------------
// small device helpers
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ v4d ldfrag(const double* X, int ld, int r0, int k0, int lane) {
  return *reinterpret_cast<const v4d*>(X + (size_t)(r0 + (lane & 15)) * ld + k0 + 4 * (lane >> 4));
}

// The same fragment from a packed twin: block (R, G) = rows 16R.., k = 16G.. is 256 doubles in the order the
// lanes consume them, lane l's four k-steps at 4 l (include/ldc_hip.h, Conventions).  NB = LD / 16.
// (explicitly GLOBAL address space, here and in gl / st_out: the batched kernels take their pointers from an
//  argument block in device memory, where the compiler cannot infer it and falls back to flat_load / flat_store,
//  which also count against lgkmcnt and made it wait with vmcnt(0) lgkmcnt(0) before every group's MFMAs)
#define LDC_GLOBAL __attribute__((address_space(1)))
__device__ __forceinline__ v4d ldpk(const double* XK, int NB, int R, int G, int lane) {
  return *(const LDC_GLOBAL v4d*)(XK + (((size_t)(R * NB + G) << 6) + lane) * 4);
}
// element i of a device array
template <typename T>
__device__ __forceinline__ T gl(const T* p, size_t i) {
  return ((const LDC_GLOBAL T*)p)[i];
}

// ---- coherent forms (COH): what the persistent kernels (small-N, chip-wide) use for every byte another work-group may have
// written in the same launch.  Producer side: write-through (sc1) stores, every storing wave's s_waitcnt vmcnt(0),
// the work-group barrier, ONE lane's agent-scope counter add.  Consumer side: ONE lane polls the counter with
// sc1 loads, the work-group barrier, then EVERY load of handed-off bytes is an sc1 load to registers (or an sc1
// LDS-direct load): sc1 loads bypass the CU's L1, which no other CU's store ever refreshes
// (MI355X_MICROARCH.md, "Workgroup dispatch, XCD placement & inter-workgroup visibility", valid forms, row 1).
// 16-byte loads go through the buffer builtin (aux 16 = sc1): hipcc counts them in vmcnt like any load.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t raw_rsrc(const double* base) {
  return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, -1, 0x00020000);
}
typedef unsigned int v4u __attribute__((ext_vector_type(4)));
typedef double v2d_t __attribute__((ext_vector_type(2)));
template <bool COH>
__device__ __forceinline__ v4d ldpk_t(const double* XK, int NB, int R, int G, int lane) {
  if constexpr (!COH) {
    return ldpk(XK, NB, R, G, lane);
  } else {
    const __amdgpu_buffer_rsrc_t r = raw_rsrc(XK);
    const int off = ((((R * NB + G) << 6) + lane) * 4) * (int)sizeof(double);
    const v2d_t lo = __builtin_bit_cast(v2d_t, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 16));
    const v2d_t hi = __builtin_bit_cast(v2d_t, __builtin_amdgcn_raw_buffer_load_b128(r, off + 16, 0, 16));
    return (v4d){lo[0], lo[1], hi[0], hi[1]};
  }
}
template <bool COH, typename T>
__device__ __forceinline__ T gl_t(const T* p, size_t i) {
  if constexpr (!COH) return gl(p, i);
  else return __hip_atomic_load((const LDC_GLOBAL T*)p + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Store of state that only LATER launches read.  wt != 0: agent-scope write-through (global_store ... sc1): the
// bytes leave the XCD's L2 while the other tiles still compute instead of at the end-of-kernel write-back
// (N=256: 4-7 MB dirty per stage launch, -3.3 us per iteration; neutral at N <= 64; non-temporal stores bought
// nothing).  The kernels must hold their argument block BY VALUE for this: through a reference into device memory
// (batched launches) every such store made the compiler re-load the fields it needed next.
// 16-byte form (p 16-byte aligned).  There is no 16-byte atomic store to lower from; an asm store is safe here (no
// result register, nothing in the kernel reads these bytes back, s_endpgm completes outstanding stores).
__device__ __forceinline__ void st_out2(double* p, double v0, double v1, int wt) {
  const v2d_t v = {v0, v1};
  // (the trailing s_nop 1: hipcc pads no hazard inside an asm string, and the instruction after a store of more than
  //  64 bits must not overwrite its data registers for two wait states)
  if (wt) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" :: "v"((LDC_GLOBAL v2d_t*)p), "v"(v) : "memory");
  else *(LDC_GLOBAL v2d_t*)p = v;
}
__device__ __forceinline__ void st_out(double* p, double v, int wt) {
  if (wt) __hip_atomic_store((LDC_GLOBAL double*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else *(LDC_GLOBAL double*)p = v;
}

// wave-uniform value that no thread of THIS launch writes: scalar load (waits on lgkmcnt, not on the
// vector-memory queue behind which the operand prefetch sits)
template <typename T>
__device__ __forceinline__ T sload(const T* p) {
  return *(const __attribute__((address_space(4))) T*)p;
}

// Wave reductions on DPP lane moves (a __shfl_xor of a double is two ds_bpermute round trips per step:
// six steps cost ~1 us in the stage-4 epilogue).  Every lane of a row of 16 ends with the row total, the
// four row totals meet through readlane.  Fixed order => deterministic.
template <int CTRL>
__device__ __forceinline__ double dpp_move(double x) {
  const long long b = __builtin_bit_cast(long long, x);
  const int lo = __builtin_amdgcn_update_dpp(0, (int)b, CTRL, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, 0xf, 0xf, true);
  return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ double lane_value(double x, int l) {
  const long long b = __builtin_bit_cast(long long, x);
  const int lo = __builtin_amdgcn_readlane((int)b, l), hi = __builtin_amdgcn_readlane((int)(b >> 32), l);
  return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ double wave_sum(double x) {
  x += dpp_move<0xB1>(x);    // quad_perm [1,0,3,2]
  x += dpp_move<0x4E>(x);    // quad_perm [2,3,0,1]
  x += dpp_move<0x141>(x);   // row_half_mirror
  x += dpp_move<0x140>(x);   // row_mirror
  return (lane_value(x, 0) + lane_value(x, 16)) + (lane_value(x, 32) + lane_value(x, 48));
}
__device__ __forceinline__ double wave_max(double x) {
  x = fmax(x, dpp_move<0xB1>(x));
  x = fmax(x, dpp_move<0x4E>(x));
  x = fmax(x, dpp_move<0x141>(x));
  x = fmax(x, dpp_move<0x140>(x));
  return fmax(fmax(lane_value(x, 0), lane_value(x, 16)), fmax(lane_value(x, 32), lane_value(x, 48)));
}
// ---- node arithmetic, shared by every kernel that evaluates a stage ---------------------------------------------
// Explicit FMAs with contraction switched off: which multiply of `a*b + c*d` hipcc fuses depends on the code around
// the expression, and the launch-per-stage kernels and the tile-resident kernel must round identically (their
// trajectories are compared bit for bit).  Every multiply-add of the epilogues goes through these.
#pragma clang fp contract(off)
__device__ __forceinline__ double nm_madd(double a, double b, double c) { return fma(a, b, c); }       // a b + c
__device__ __forceinline__ double nm_sq2(double a, double b) { return fma(b, b, a * a); }              // a^2 + b^2
// rank-1 completion for k = M-1 of the six contractions at a node (tail layout)
__device__ __forceinline__ void nm_tail(double& ux, double& vx, double& uy, double& vy, double& lu, double& lv,
                                        double dxl, double d2xl, double dyl, double d2yl, double ue, double ve,
                                        double un_, double vn_) {
  ux = fma(dxl, ue, ux); vx = fma(dxl, ve, vx);
  uy = fma(un_, dyl, uy); vy = fma(vn_, dyl, vy);
  lu += fma(d2xl, ue, un_ * d2yl);
  lv += fma(d2xl, ve, vn_ * d2yl);
}
// -(u dq/dx + v dq/dy) - dp/dq + nu lap q      (sg.py:331-338)
__device__ __forceinline__ double nm_momentum(double uin, double vin, double qx, double qy, double gp, double nu,
                                              double lap) {
  return fma(nu, lap, -fma(vin, qy, uin * qx) - gp);
}
__device__ __forceinline__ double nm_continuity(double beta2, double ux, double vy) { return -beta2 * (ux + vy); }
#pragma clang fp contract(fast)

// Two dot products that share their first row, a . b1 and a . b2, with every operand of the first 512 elements
// requested BEFORE the first multiply: the plain loop waits for its loads once per trip (five trips at N = 256, each a
// cold round trip: the rows of index M-1 of the pressure transforms took ~5 us this way and were the critical path of
// the post launch).  Same multiply-adds per lane in the same order as dot_rows.
template <bool COH>
__device__ __forceinline__ void dot_rows2(const double* a, const double* b1, const double* b2, int n, int lane,
                                          double& s1, double& s2) {
  constexpr int U = 8;
  double av[U], x1[U], x2[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int k = lane + 64 * u;
    const bool in = k < n;
    av[u] = in ? gl_t<COH>(a, (size_t)k) : 0.0;
    x1[u] = in ? b1[k] : 0.0;
    x2[u] = in ? b2[k] : 0.0;
  }
  double t1 = 0.0, t2 = 0.0;
#pragma unroll
  for (int u = 0; u < U; ++u) {
    if (lane + 64 * u < n) { t1 = nm_madd(av[u], x1[u], t1); t2 = nm_madd(av[u], x2[u], t2); }
  }
  for (int k = lane + 64 * U; k < n; k += 64) {
    const double x = gl_t<COH>(a, (size_t)k);
    t1 = nm_madd(x, b1[k], t1); t2 = nm_madd(x, b2[k], t2);
  }
  s1 = wave_sum(t1); s2 = wave_sum(t2);
}

// wave-cooperative dot product of two contiguous rows (fixed order => deterministic)
__device__ __forceinline__ double dot_rows(const double* a, const double* b, int n, int lane) {
  double s = 0.0;
  for (int k = lane; k < n; k += 64) s = nm_madd(a[k], b[k], s);
  return wave_sum(s);
}

// Sum the four waves' partial accumulators through LDS.  After the call thread
// (wv, lane) owns tile element  row = (lane>>4) + 4*wv,  col = lane&15.
template <int NA>
__device__ __forceinline__ void kreduce(const v4d (&acc)[NA], double* red, int lane, int wv,
                                        double (&out)[NA]) {
#pragma unroll
  for (int a = 0; a < NA; ++a)
#pragma unroll
    for (int r = 0; r < 4; ++r) red[((wv * NA + a) * 4 + r) * 64 + lane] = acc[a][r];
  __syncthreads();
#pragma unroll
  for (int a = 0; a < NA; ++a) {
    double s = red[((0 * NA + a) * 4 + wv) * 64 + lane];
    s += red[((1 * NA + a) * 4 + wv) * 64 + lane];
    s += red[((2 * NA + a) * 4 + wv) * 64 + lane];
    s += red[((3 * NA + a) * 4 + wv) * 64 + lane];
    out[a] = s;
  }
}

// block-wide reduction of NV sums and NM maxima into partials[]; all threads call.
template <int NV, int NM>
__device__ __forceinline__ void block_reduce_store(double (&sums)[NV], double (&maxs)[NM > 0 ? NM : 1],
                                                   double* sm, double* dst, int lane, int wv) {
#pragma unroll
  for (int v = 0; v < NV; ++v) sums[v] = wave_sum(sums[v]);
#pragma unroll
  for (int v = 0; v < NM; ++v) maxs[v] = wave_max(maxs[v]);
  __syncthreads();   // sm may alias the kreduce buffer
  if (lane == 0) {
#pragma unroll
    for (int v = 0; v < NV; ++v) sm[wv * (NV + NM) + v] = sums[v];
#pragma unroll
    for (int v = 0; v < NM; ++v) sm[wv * (NV + NM) + NV + v] = maxs[v];
  }
  __syncthreads();
  const int t = wv * 64 + lane;
  if (t < NV) {
    dst[t] = ((sm[t] + sm[(NV + NM) + t]) + sm[2 * (NV + NM) + t]) + sm[3 * (NV + NM) + t];
  } else if (t < NV + NM) {
    dst[t] = fmax(fmax(sm[t], sm[(NV + NM) + t]), fmax(sm[2 * (NV + NM) + t], sm[3 * (NV + NM) + t]));
  }
}

// blockIdx -> tile.  Blocks are dealt round-robin over the 8 XCDs (b and b+8 share an L2), so give each XCD a
// compact (T/4 x T/2 when T%8==0) patch of tiles: the operand panels it re-reads then stay in its own L2.
// Inside a patch the order is column-major with the columns rotated by the row, so that the diagonal tiles (I, I)
// and the tiles (I, I+1 mod T) come FIRST: they carry the index-(M-1) jobs (stage kernel) and run 1-2 us longer,
// and a launch dispatches its 256 work-groups over ~1.7 us in block order -- started first, their extra time
// hides under the ramp instead of ending the launch late.  Pure speed; any bijection is correct.
// Branch-free on purpose: a branch here splits the kernel's entry block, and the scalar loads of the
// kernel arguments used after it are then issued one cold miss after the other instead of together.
__device__ __forceinline__ void tile_of_block(int b, int T, int& I, int& J) {
  const bool patched = (T & 7) == 0;
  const int xcd = patched ? (b & 7) : 0, loc = patched ? (b >> 3) : b;
  const int pr = patched ? (T >> 2) : T, pc = patched ? (T >> 1) : T;   // patch rows / columns (4 x 2 patches)
  const int pI = xcd >> 1, pJ = xcd & 1;
  const int jj = loc / pr, i = loc - jj * pr;
  I = pI * pr + i;
  J = pJ * pc + (I + jj) % pc;
}

// ---------------------------------------------------------------------------------------
// finalize arguments (the fold of the per-work-group partial sums: change norms, |R|, E, next dt, latch, record)
// ---------------------------------------------------------------------------------------
struct FinalArgs {
  int nblk4, nblkZ, nblkP;   // rows in each slab
  int with_diag, warmup, nan_guard, rec_cap;
  int do_critical;           // 0: flush only
  double cfl, beta2, nu, hx, hy, lid, tol;
  const double *part4, *partZ0, *partP0;   // parity slabs: partZ0 + parity * stride
  long long stride;
  double* scal;
  int* ctrl;
  double* rec;
};

// the two halves of the denominator of the CFL step (sg.py:387-408), separately: fin_decide_wave evaluates them on two lanes
// side by side (each is a square root and two divisions in a row)
__device__ __forceinline__ double dt_lambda_x(double umax, const FinalArgs& a) {
  const double um = fmax(umax, a.lid);
  return (um + sqrt(nm_madd(um, um, a.beta2))) / a.hx + a.nu / (a.hx * a.hx);
}
__device__ __forceinline__ double dt_lambda_y(double vmax, const FinalArgs& a) {
  const double vm = fmax(vmax, 1e-10);
  return (vm + sqrt(nm_madd(vm, vm, a.beta2))) / a.hy + a.nu / (a.hy * a.hy);
}
__device__ __forceinline__ double next_dt(double umax, double vmax, const FinalArgs& a) {
  return a.cfl / (dt_lambda_x(umax, a) + dt_lambda_y(vmax, a));
}

// Control block of one trial inside the persistent trial kernel (LDS; every work-group keeps its own, identical copy:
// they all fold the same partial sums in the same order).  Mirrors ctrl[] / scal[] of the launch-per-stage path.
struct TrialState {
  int done, iter, step, flushed, pdone, drows;
  int abort;                 // a bounded spin of a grid barrier gave up (LDC_E_SYNC): every thread leaves
  unsigned arrive;           // chip-wide kernel: waves whose stores are drained, counted over the hand-overs of the launch
  double dt, umax, vmax;
  double dtp;                // chip-wide kernel: the step of the iteration being closed (dt already holds the next)
};

// ---------------------------------------------------------------------------------------
// RK stage kernel
// ---------------------------------------------------------------------------------------
struct StageArgs {
  int M, LD, T, tail;
  int Mx, My;                             // nodes along x / y (nx != ny: the tiling is built for M = max, the arrays are zero padded)
  double nu, beta2, alpha;
  const double *Dx, *D2x, *Dy, *D2y, *IxF, *GxF;
  const double *Uin, *UinT, *Vin, *VinT;  // stage input state
  const double *U0, *V0, *P0;             // step-start state (aliases the outputs when LAST)
  const double *U0T, *V0T;                // its transposed copies (read by the column nodes of index M-1)
  const double *T1T, *T2T;                // pressure transforms (GP only)
  double *PX, *PY;                        // grad p: written when GP, read otherwise
  double *Uout, *UoutT, *Vout, *VoutT, *Pout;
  const double *ulid, *wx, *wy;
  const double *DxL, *D2xL, *DyL, *D2yL;   // last columns (index M-1) of the four operators
  double *W, *WT;                         // vorticity (+ transposed copy): written by DIAG 1, read by DIAG 2
  double *partZ0, *partP0;                // parity slabs of the enstrophy / palinstrophy partial sums
  long long stride;
  const double* scal;
  int* ctrl;
  double* partials;
#ifdef LDC_TIMING
  int ablate;                             // timing experiments only: 1 skip MFMAs, 2 skip operand loads,
                                          // 4 no stage-4 reduction, 8 no transposed stores, 16 no M-1 nodes, 32 no p store, 64 cycle stamps
#endif
  double* dump[11];
  // packed twins (operand order, see ldpk): what the K loop reads / what the epilogue keeps in step
  int NB;
  const double *DxK, *D2xK, *DyK, *D2yK, *IxFK, *GxFK;
  const double *UinK, *UinTK, *VinK, *VinTK, *T1TK, *T2TK;
  double *UoutK, *UoutTK, *VoutK, *VoutTK, *PoutK, *WK, *WTK;
  int wt;   // write-through stores (st_out)
  int rm_out;   // also store the row-major forms of the velocity outputs (stage 4 only: nothing reads those of the
                // stage buffers -- tile nodes take their stage input from the packed twin, index M-1 is never rewritten)
};

// slots of the stage-4 partial sums
enum { PS_DU2 = 0, PS_DV2, PS_U02, PS_V02, PS_RU2, PS_RV2, PS_RP2, PS_E, PS_NSUM, PS_UMAX = PS_NSUM, PS_VMAX, PS_N };
static_assert(PS_N <= LDC_NPART, "partials row too small");
static_assert(PS_NSUM == 8, "stage-4 reduction assigns one wave per sum");

// What ONE wave makes of the totals tot[q] (q < PS_N + 2 values): their
// square roots and the next dt by SEPARATE lanes in parallel (eight fp64 square roots, the divisions and next_dt in a
// row on one lane were most of the finalize block's tail), everything else uniformly on every lane through
// readlane.  `writer`: lane 0 also stores the record, the control words and the scalars.  Results are uniform over
// the wave.  The same function serves the finalize block of the post launch, the finalize kernel and the persistent
// kernel: one arithmetic, one rounding.
struct FinOut {
  int latch;                 // 0, or the latch this iteration sets
  int iter, flushed;         // control words after this call
  double dt, umax, vmax;     // scalars after this call (unchanged when there was nothing to close)
};
__device__ __forceinline__ FinOut fin_decide_wave(const FinalArgs& a, const double* sm, const int lane, const bool crit,
                                                  const bool flush, const int iter, const int flushed,
                                                  const double dt_cur, const bool writer) {
  auto total = [&](int q) { return sm[q]; };
  double xr = 0.0, xs = 0.0;
  if (lane < PS_N + 2) {
    xr = total(lane);
    xs = (lane < PS_RP2 + 1) ? sqrt(xr) : 0.0;     // DU2, DV2, U02, V02, RU2, RV2, RP2
  } else if (lane == 16 && crit) {
    xr = dt_lambda_x(total(PS_UMAX), a);
  } else if (lane == 17 && crit) {
    xr = dt_lambda_y(total(PS_VMAX), a);
  }
  double r[PS_N + 2], sq[PS_RP2 + 1];
#pragma unroll
  for (int q = 0; q < PS_N + 2; ++q) r[q] = lane_value(xr, q);
#pragma unroll
  for (int q = 0; q < PS_RP2 + 1; ++q) sq[q] = lane_value(xs, q);
  // the three divisions that remain -- the two relative changes and cfl / (lambda_x + lambda_y) -- by three lanes at once
  // (one after the other on every lane they were ~900 cycles of the fold's tail); the same operations, the same rounding
  const double lxy = lane_value(xr, 16) + lane_value(xr, 17);
  const double num = (lane == 0) ? sq[PS_DU2] : (lane == 1) ? sq[PS_DV2] : a.cfl;
  const double den = (lane == 0) ? (sq[PS_U02] + 1e-12) : (lane == 1) ? (sq[PS_V02] + 1e-12) : lxy;
  const double quo = num / den;
  const double dt_new = lane_value(quo, 2);
  FinOut o;
  o.latch = 0; o.iter = iter; o.flushed = flushed; o.dt = dt_cur; o.umax = 0.0; o.vmax = 0.0;
  const bool w0 = writer && lane == 0;
  if (flush) {
    if (w0) {
      double* rec = a.rec + (size_t)((iter - 1) % a.rec_cap) * LDC_REC_LEN;
      rec[LDC_REC_Z] = 0.5 * r[PS_N];
      rec[LDC_REC_P] = 0.5 * r[PS_N + 1];
    }
    o.flushed = iter;
  }
  if (crit) {
    const double relu = lane_value(quo, 0);
    const double relv = lane_value(quo, 1);
    // Python's max(a, b) returns a unless b > a: a NaN in relv is dropped, one in relu sticks
    const double rel = (relv > relu) ? relv : relu;
    if (w0) {
      double* rec = a.rec + (size_t)(iter % a.rec_cap) * LDC_REC_LEN;   // `iter` = 0-based index of this iteration
      rec[LDC_REC_REL] = rel;
      rec[LDC_REC_RU] = sq[PS_RU2];
      rec[LDC_REC_RV] = sq[PS_RV2];
      rec[LDC_REC_RP] = sq[PS_RP2];
      rec[LDC_REC_E] = 0.5 * r[PS_E];
      rec[LDC_REC_Z] = 0.0;
      rec[LDC_REC_P] = 0.0;
      rec[LDC_REC_DT] = dt_cur;
    }
    o.latch = (iter >= a.warmup && rel < a.tol) ? 1
              : (a.nan_guard && !(fabs(rel) <= 1.79769313486231570815e308)) ? 2 : 0;
    o.umax = r[PS_UMAX]; o.vmax = r[PS_VMAX]; o.dt = dt_new;
    o.iter = iter + 1;
    if (!a.with_diag) o.flushed = iter + 1;      // nothing to fold for this record
  }
  return o;
}

// the control words and scalars a closed iteration leaves behind (launch path: ctrl / scal in memory)
__device__ __forceinline__ void fin_publish(const FinalArgs& a, const FinOut& o, const bool crit, const bool flush) {
  if (flush || crit) a.ctrl[LDC_CTRL_FLUSHED] = o.flushed;
  if (crit) {
    a.scal[LDC_SCAL_UMAX] = o.umax;
    a.scal[LDC_SCAL_VMAX] = o.vmax;
    a.scal[LDC_SCAL_DT] = o.dt;
    a.ctrl[LDC_CTRL_ITER] = o.iter;
    if (o.latch) a.ctrl[LDC_CTRL_DONE] = o.latch;
  }
}


#ifndef LDC_WT_MIN_TILES
#define LDC_WT_MIN_TILES 1       // see st_out: write-through measured neutral-or-better at every size (tools/ab_*.py)
#endif
constexpr int kStageWaves = 8;                 // 2 waves per SIMD: needed to saturate the f64 MFMA pipe
constexpr int kStageThreads = 64 * kStageWaves;
// timing experiments (ldc_debug_stamps): cycle stamps per wave at fixed points of the stage kernel
// (point 0 = kernel entry is taken unconditionally into t_entry and stored with point 1: a conditional store
//  in the prologue would split the entry block, see tile_of_block)
#ifdef LDC_TIMING
#define LDC_STAMP(k) do { if ((a.ablate & 64) && lane == 0) { \
  double* st_ = a.dump[0] + ((size_t)bx * kStageWaves + wv) * 8; \
  st_[k] = (double)__builtin_amdgcn_s_memtime(); if ((k) == 1) st_[0] = (double)t_entry; } } while (0)
#else
#define LDC_STAMP(k) do { } while (0)
#endif
constexpr int kEdgeRowDoubles = 6 * 4 * 16;    // per wave: 6 rows of index M-1 x 4 groups x 16 (tail needs T <= 16)
constexpr size_t kLdsLimit = 160 * 1024;

// acc += x . y as four chained FMAs: the index-(M-1) jobs run on the same FP64 units as the MFMAs next to them
// (fp64 vector rate == fp64 matrix rate on this chip), so their cost is their operation count -- 4 per dot product
// instead of the 6 of (x0 y0 + x1 y1) + (x2 y2 + x3 y3) followed by an add
__device__ __forceinline__ void fma4(double& acc, const v4d& x, const v4d& y) {
  acc = fma(x[0], y[0], acc);
  acc = fma(x[1], y[1], acc);
  acc = fma(x[2], y[2], acc);
  acc = fma(x[3], y[3], acc);
}
// sum over the four k-quads of a wave: lanes l, l^16, l^32, l^48 hold the same (row|col) index
// gfx950 lane-swap instructions instead of shuffles: a __shfl_xor of a double is two ds_bpermute round trips, and the
// 15 sums a job tile needs after its K loop cost it ~1.2 us; v_permlane16_swap / v_permlane32_swap exchange the odd
// 16- / 32-lane rows of one register with the even rows of another on the VALU.  Same association as before,
// (x0 + x1) + (x2 + x3) over the four rows, so the results are bit-identical.
template <bool ROWS32>
__device__ __forceinline__ double swap_add(double x) {
  const long long b = __builtin_bit_cast(long long, x);
  const unsigned lo = (unsigned)b, hi = (unsigned)(b >> 32);
  const auto rl = ROWS32 ? __builtin_amdgcn_permlane32_swap(lo, lo, false, false)
                         : __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
  const auto rh = ROWS32 ? __builtin_amdgcn_permlane32_swap(hi, hi, false, false)
                         : __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
  const double even = __builtin_bit_cast(double, ((long long)rh[0] << 32) | (long long)rl[0]);
  const double odd = __builtin_bit_cast(double, ((long long)rh[1] << 32) | (long long)rl[1]);
  return even + odd;     // every lane: its row pair's even-row value + odd-row value
}
__device__ __forceinline__ double quad_sum(double x) {
  return swap_add<true>(swap_add<false>(x));
}

// One wave's operands.  Both roles contract the full 2x2 combination {A0,A1} x {B0,B1}:
//   role 0 (x-derivatives): A0 = Dx, A1 = D2x (rows I)   B0 = UT, B1 = VT (rows J)   A2 = GxF, B2 = T1T
//        c00 = du/dx   c01 = dv/dx   c10 = d2u/dx2   c11 = d2v/dx2   c4 = dp/dx
//   role 1 (y-derivatives): A0 = U,  A1 = V   (rows I)   B0 = Dy, B1 = D2y (rows J)  A2 = IxF, B2 = T2T
//        c00 = du/dy   c10 = dv/dy   c01 = d2u/dy2   c11 = d2v/dy2   c4 = dp/dy
struct RoleOps {
  const double *A0, *A1, *B0, *B1, *A2, *B2;   // packed twins: the fragments of the K loop
  const double* row[6];                        // the row-major forms, slots A0 A1 A2 B0 B1 B2 (rows of index M-1)
#ifdef LDC_TIMING
  int ablate;
#endif
  int role; // 0: x-derivative chains, 1: y-derivative chains
  int x4;   // fifth contraction: 0: A2.B2 (grad p)   1: A0.B2 (d omega/dx = Dx . WT)   2: A2.B0 (d omega/dy = W . Dy)
};

struct RoleFrags {     // the four operands of the 2x2 combination, double-buffered by the caller
  v4d a0, a1, b0, b1;
};
struct ExtraFrags {    // operands of the fifth contraction: single-buffered (loaded at the top of a group,
  v4d a2, b2;          // consumed by that group's LAST four MFMAs, ~1600 cycles later)
};

// (timing switch `ablate & 2`: every lane reads element 0 instead -- same instructions, no operand traffic;
//  a branch around the loads would make the compiler wait for them at the join)
// I, J: block row of the A / B operands; G: 16-k group
template <bool COH>
__device__ __forceinline__ void load_role(RoleFrags& f, const RoleOps& o, int NB, int I, int J, int G, int lane) {
  const int keep = LDC_ABL(o, 2) ? 0 : 1;
  I *= keep; J *= keep; G *= keep; lane *= keep;
  f.a0 = ldpk_t<COH>(o.A0, NB, I, G, lane); f.a1 = ldpk_t<COH>(o.A1, NB, I, G, lane);
  f.b0 = ldpk_t<COH>(o.B0, NB, J, G, lane); f.b1 = ldpk_t<COH>(o.B1, NB, J, G, lane);
}

template <bool COH>
__device__ __forceinline__ void load_extra(ExtraFrags& x, const RoleOps& o, int NB, int I, int J, int G, int lane) {
  const int keep = LDC_ABL(o, 2) ? 0 : 1;
  I *= keep; J *= keep; G *= keep; lane *= keep;
  // both, always: a load that only one role issues makes the number of loads in flight path-dependent, and the
  // compiler then waits with vmcnt(0) before every group's MFMAs (prefetch included).  The operand a role does
  // not use (x4 = 1: a2, x4 = 2: b2) points at a panel it reads anyway (A0 / B0): a cache hit.  (Timing-neutral.)
  x.a2 = ldpk_t<COH>(o.A2, NB, I, G, lane);
  x.b2 = ldpk_t<COH>(o.B2, NB, J, G, lane);
}

// LDS-direct 16-byte load: lane l's 16 bytes land at lds_dst + 16*l bytes (lds_dst wave-uniform);
// no registers, counted by vmcnt like any load (semantics pinned by tools/probes/dma_probe.hip)
typedef double v2d __attribute__((ext_vector_type(2)));
template <bool COH>
__device__ __forceinline__ void dma16(const double* src, double* lds_dst) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds_dst, 16, 0, COH ? 16 : 0);   // aux 16 = sc1
}

template <bool GP, int NA>
__device__ __forceinline__ void mfma_role(const RoleFrags& f, const ExtraFrags& x, v4d (&acc)[NA], int ablate, int x4) {
#ifdef LDC_TIMING
  if (ablate & 1) {   // keep the operands live without issuing MFMAs
    acc[0][0] += f.a0[0] + f.b0[1] + f.a1[2] + f.b1[3];
    if (GP) acc[4][0] += x.a2[0] + x.b2[0];
    return;
  }
#else
  (void)ablate;
#endif
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    acc[0] = MFMA_F64(f.a0[s], f.b0[s], acc[0]);
    acc[1] = MFMA_F64(f.a0[s], f.b1[s], acc[1]);
    acc[2] = MFMA_F64(f.a1[s], f.b0[s], acc[2]);
    acc[3] = MFMA_F64(f.a1[s], f.b1[s], acc[3]);
  }
  if (GP) {
    // The fifth contraction goes LAST, and the scheduler must leave it there: its operands were requested at the top
    // of this group, and hipcc otherwise interleaves these MFMAs with the sixteen above (one after every four: the
    // accumulators are independent), so that the wave waits for the extra fragments ~300 cycles after asking for them.
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const double p = (x4 == 1) ? f.a0[s] : x.a2[s];
      const double q = (x4 == 2) ? f.b0[s] : x.b2[s];
      acc[4] = MFMA_F64(p, q, acc[4]);
    }
  }
}

// Index M-1 lies outside the MFMA tiles when 16*T == M-1 (`tail`).  The residual at those
// nodes feeds only ||R_u||, ||R_v|| (quirk Q4: boundary nodes count) but must be exact.
// A tile that holds the right fragments gets it cheaply: the row (column) vector of index M-1
// (staged in LDS) dotted with fragments that are already in registers, on the VALU next to the MFMAs.
//   er[ai][bj] : row node  (M-1, c0+idx) = A_i[M-1, k] . B_j[c0+idx, k]
//   ec[ai][bj] : col node  (r0+idx, M-1) = A_i[r0+idx, k] . B_j[M-1, k]
//   ek[ai][bj] : corner    (M-1, M-1)
struct EdgeAcc {
  double er[5], ec[5], ek[5];
};

// VEL: 1 = the four velocity contractions (R_u, R_v at the nodes: LAST, DUMP)
//      2 = ONE contraction in slot 0, the one the vorticity needs from this role (DIAG 1):
//          role 0: A0.B1 = Dx . VT (dv/dx)      role 1: A0.B0 = U . Dy (du/dy)
// GP : the fifth contraction (grad p, or grad omega by o.x4)
// SA2 / SB2: the LDS slots that hold the rows of index M-1 of the fifth contraction's operands (the launch kernels
// stage them in slots 2 and 5; the tile-resident kernel keeps the constant rows in place and uses separate slots
// for the rows that change)
template <int VEL, bool GP, int SA2 = 2, int SB2 = 5>
__device__ __forceinline__ void edge_group(EdgeAcc& e, const RoleFrags& f, const ExtraFrags& x, const RoleOps& o,
                                           const double* erow, int n, int lane, bool rowE, bool colE, bool cornE) {
  // erow: this WAVE's pieces of the six rows of index M-1 (slots A0 A1 A2 B0 B1 B2) x its (at most four)
  // groups x 16 doubles, put into LDS by LDS-direct loads at kernel entry: no registers, no barrier, and
  // the reads here keep the vector-memory queue (the fragment prefetch) out of the way.
  const int off = n * 16 + 4 * (lane >> 4);
  auto row_of = [&](int slot) { return *reinterpret_cast<const v4d*>(erow + slot * 64 + off); };
  v4d ar0, ar1, xr, bc0, bc1, yc;
  const bool need_a0_row = (VEL != 0) || (GP && o.x4 == 1);
  const bool need_b0_col = (VEL == 1) || (VEL == 2 && o.role == 1) || (GP && o.x4 == 2);
  const bool need_b1_col = (VEL == 1) || (VEL == 2 && o.role == 0);
  if (rowE || cornE) {
    if (need_a0_row) ar0 = row_of(0);
    if (VEL == 1) ar1 = row_of(1);
    if (GP) xr = (o.x4 == 1) ? ar0 : row_of(SA2);
  }
  if (colE || cornE) {
    if (need_b0_col) bc0 = row_of(3);
    if (need_b1_col) bc1 = row_of(4);
    if (GP) yc = (o.x4 == 2) ? bc0 : row_of(SB2);
  }
  if (rowE) {
    if (VEL == 1) {
      fma4(e.er[0], ar0, f.b0); fma4(e.er[1], ar0, f.b1);
      fma4(e.er[2], ar1, f.b0); fma4(e.er[3], ar1, f.b1);
    }
    if (VEL == 2) fma4(e.er[0], ar0, o.role == 0 ? f.b1 : f.b0);
    if (GP) fma4(e.er[4], xr, (o.x4 == 2) ? f.b0 : x.b2);
  }
  if (colE) {
    if (VEL == 1) {
      fma4(e.ec[0], f.a0, bc0); fma4(e.ec[1], f.a0, bc1);
      fma4(e.ec[2], f.a1, bc0); fma4(e.ec[3], f.a1, bc1);
    }
    if (VEL == 2) fma4(e.ec[0], f.a0, o.role == 0 ? bc1 : bc0);
    if (GP) fma4(e.ec[4], (o.x4 == 1) ? f.a0 : x.a2, yc);
  }
  if (cornE) {
    if (VEL == 1) {
      fma4(e.ek[0], ar0, bc0); fma4(e.ek[1], ar0, bc1);
      fma4(e.ek[2], ar1, bc0); fma4(e.ek[3], ar1, bc1);
    }
    if (VEL == 2) fma4(e.ek[0], ar0, o.role == 0 ? bc1 : bc0);
    if (GP) fma4(e.ek[4], xr, yc);
  }
}

// block-wide reduction over the 8 stage waves (fixed order => deterministic)
template <int NV, int NM>
__device__ __forceinline__ void stage_block_reduce(double (&sums)[NV], double (&maxs)[NM], double* sm, double* dst,
                                                   int lane, int wv) {
#pragma unroll
  for (int v = 0; v < NV; ++v) sums[v] = wave_sum(sums[v]);
#pragma unroll
  for (int v = 0; v < NM; ++v) maxs[v] = wave_max(maxs[v]);
  if (lane == 0) {
#pragma unroll
    for (int v = 0; v < NV; ++v) sm[wv * (NV + NM) + v] = sums[v];
#pragma unroll
    for (int v = 0; v < NM; ++v) sm[wv * (NV + NM) + NV + v] = maxs[v];
  }
  __syncthreads();
  const int t = wv * 64 + lane;
  if (t < NV + NM) {
    double r = sm[t];
#pragma unroll
    for (int w = 1; w < kStageWaves; ++w) {
      const double x = sm[w * (NV + NM) + t];
      r = (t < NV) ? (r + x) : fmax(r, x);
    }
    dst[t] = r;
  }
}

// dynamic LDS carve (doubles): [0, RED) per-wave accumulators; then edge partials, two
// transposition tiles and the block-reduction scratch
template <bool GP, bool EDGES>
struct StageLds {
  static constexpr int NA = GP ? 5 : 4;
  static constexpr int RED = kStageWaves * NA * 4 * 64;
  static constexpr int EDGE = RED;                         // [wave][15][16]
  static constexpr int TILE = EDGE + kStageWaves * 15 * 16;  // 4 x 16 x 17 (u, v, omega, p)
  static constexpr int SCR = TILE + 4 * 16 * 17;           // kStageWaves * PS_N
  static constexpr int EROW = SCR + kStageWaves * PS_N;    // [wave][6 rows of index M-1][4 groups][16]
  static constexpr int TOTAL = EROW + (EDGES ? kStageWaves * kEdgeRowDoubles : 0);
  static constexpr size_t BYTES = sizeof(double) * TOTAL;
};

// GP   : also contract the pressure transforms (px, py) and store them
// LAST : stage 4 -- pressure update, in-place state, reductions (incl. the index-M-1 nodes)
// DUMP : parity-test mode, writes every intermediate, touches no state
// BATCH: blockIdx.y selects one of several independent trials; their argument blocks live in
//        device memory (a_arr), the single-trial path keeps them in the kernarg segment (a_val)
// DIAG : 0 none; 1 (stage 1 of iteration n+1) vorticity of phi^(n+1) = dv/dx - du/dy, which this stage
//        forms anyway, its enstrophy partial sums and the omega / omega^T arrays; 2 (stage 2) the two
//        contractions Dx.omega, omega.Dy^T and the palinstrophy partial sums.  Both belong to the record
//        of iteration n and are folded by the finalize block of the next post launch.
// (the body is a function of its own so that `a` is a by-value copy the compiler can keep in scalar registers; the
//  tile-resident form of the same stage for the persistent trial kernel is tile_stage below)
// RECT: nx != ny -- the node classes (wall, lid, interior) come from (a.Mx, a.My) instead of M.  A template parameter, not
// a run-time one: with the two extra scalars in every instantiation the N=256 iteration was 0.35 us longer on the same box
// (50.1 -> 50.45 us, profiles/r03_ab_stage_rect.log) -- the square kernels are the code they were.
template <bool GPV, bool LAST, bool DUMP, bool BATCH, int DIAG, bool RECT = false>
__device__ __forceinline__ void stage_body(const StageArgs a, const int bx, const int nblk, double* lds) {
  constexpr bool GP = GPV || (DIAG == 2);      // "has a fifth contraction" (grad p or grad omega)
  static_assert(!(GPV && DIAG == 2), "stage 2 of SG carries no pressure contraction");
  constexpr int VEL = (LAST || DUMP) ? 1 : (DIAG == 1 ? 2 : 0);   // what the nodes of index M-1 need here
  constexpr bool EDGES = GP || (VEL != 0);
  using L = StageLds<GP, EDGES>;
  constexpr int NA = L::NA;
  double* red = lds;
#ifdef LDC_TIMING
  const unsigned long long t_entry = __builtin_amdgcn_s_memtime();
#endif

  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int role = wv >> 2, kq = wv & 3;
  const int M = a.M, LD = a.LD, T = a.T, m1 = M - 1, NB = a.NB;

  int I, J;
  tile_of_block(bx, T, I, J);
  const int r0 = 16 * I, c0 = 16 * J;
  // The nodes of index M-1 are spread over 2T+1 tiles, one job each (the VALU dot products of a job cost a
  // tile ~0.7 us; all three jobs in the corner tile made it the slowest of the launch by 2-3 us):
  //   row nodes (M-1, 16J + idx) need the B fragments of column block J  -> the diagonal tile (J, J)
  //   column nodes (16I + idx, M-1) need the A fragments of row block I  -> tile (I, I+1 mod T)
  //   the corner node needs no fragments                                 -> tile (0, 2) (T < 3: tile (0, 0))
  const bool etile = EDGES && a.tail && !LDC_ABL(a, 16);
  const bool rowE = etile && (I == J);
  const bool colE = etile && (J == (I + 1) % T);
  const bool cornE = etile && (T >= 3 ? (I == 0 && J == 2) : (I == 0 && J == 0));
  const bool anyE = (rowE || colE || cornE) && !LDC_ABL(a, 512);     // block-uniform (512: timing, no K-loop part)

  RoleOps o;
#ifdef LDC_TIMING
  o.ablate = a.ablate;
#endif
  o.role = role;
  o.x4 = (DIAG == 2) ? (role == 0 ? 1 : 2) : 0;
  if (role == 0) {
    o.A0 = a.DxK; o.A1 = a.D2xK; o.B0 = a.UinTK; o.B1 = a.VinTK; o.A2 = (DIAG == 2) ? a.DxK : a.GxFK; o.B2 = (DIAG == 2) ? a.WTK : a.T1TK;
    o.row[0] = a.Dx; o.row[1] = a.D2x; o.row[2] = a.GxF; o.row[3] = a.UinT; o.row[4] = a.VinT; o.row[5] = (DIAG == 2) ? a.WT : a.T1T;
  } else {
    o.A0 = a.UinK; o.A1 = a.VinK; o.B0 = a.DyK; o.B1 = a.D2yK; o.A2 = (DIAG == 2) ? a.WK : a.IxFK; o.B2 = (DIAG == 2) ? a.DyK : a.T2TK;
    o.row[0] = a.Uin; o.row[1] = a.Vin; o.row[2] = (DIAG == 2) ? a.W : a.IxF; o.row[3] = a.Dy; o.row[4] = a.D2y; o.row[5] = a.T2T;
  }

  // this wave's K groups: kq, kq+4, ... (ng of them); group n is 16-k block gk(n)
  const int ng = (T - kq + 3) / 4;
  auto gk = [&](int n) { return kq + 4 * n; };
  double* erow = lds + L::EROW + wv * kEdgeRowDoubles;

  // ---- tiles with an index-(M-1) job: this wave's pieces of the six rows, LDS-direct, issued FIRST:
  // they are then older than every fragment load, so waiting for a fragment never leaves one of them pending (issued
  // between the first fragments and the pointwise loads they made the compiler wait with vmcnt(0) before the MFMAs of
  // every group, prefetch included: the K loop lost its overlap)
  if (anyE && !LDC_ABL(a, 2048)) {      // (2048: timing, no LDS-direct rows)
    const int h = lane & 1, q = (lane >> 1) & 3, gi = (lane >> 3) & 3, sl = lane >> 5;
    const size_t off = (size_t)m1 * LD + 16 * (kq + 4 * gi) + 4 * q + 2 * h;
    const bool live = (kq + 4 * gi) < T;
    const bool needA = rowE || cornE, needB = colE || cornE;
    const double* s01 = sl ? o.row[1] : o.row[0];
    const double* s23 = sl ? o.row[3] : o.row[2];
    const double* s45 = sl ? o.row[5] : o.row[4];
    if (live && needA && s01 != nullptr) dma16<false>(s01 + off, erow);
    if (live && (sl ? needB : needA) && s23 != nullptr) dma16<false>(s23 + off, erow + 128);
    if (live && needB && s45 != nullptr) dma16<false>(s45 + off, erow + 256);
  }
  // ---- first fragments in flight before anything else ------------------------------------
  RoleFrags fa, fb;
  ExtraFrags fx;
  load_role<false>(fa, o, NB, I, J, ng > 0 ? gk(0) : 0, lane);

  // The latch, the step counter and dt are only READ here (scalar loads, nobody waits for them yet); the
  // latch is acted upon after the K loop.  An early `return` at this point made the compiler sink the
  // fragment loads below it: kernel arguments -> latch -> operand pointers -> fragments became four cold
  // misses in a row (~1.4 us before the first MFMA).  A latched launch now runs its K loop for nothing,
  // which only happens in the last partial batch of a solve.
  const int latched = DUMP ? 0 : sload(a.ctrl + LDC_CTRL_DONE);
  const int step0 = sload(a.ctrl + LDC_CTRL_STEP);
  const double adt = a.alpha * sload(a.scal + LDC_SCAL_DT);

  // ---- pointwise operands of the epilogue, issued now so that they land under the MFMAs ----------
  // threads 0..255 own node (i, j) of the tile; in a tile with an index-(M-1) job threads 256.. own
  // that job's nodes (kind 0: (M-1, c0+idx), 1: (r0+idx, M-1), 2: the corner)
  const int ti = 4 * (wv & 3) + (lane >> 4), tj = lane & 15;
  const bool owner = tid < 256;
  const int ekind = (tid - 256) >> 4, eidx = (tid - 256) & 15;
  const bool edge_thr = !owner && !LDC_ABL(a, 1024) &&     // (1024: timing, no epilogue part)
                        ((ekind == 0 && rowE) || (ekind == 1 && colE) || (ekind == 2 && eidx == 0 && cornE));
  const int i = owner ? (r0 + ti) : ((ekind == 1) ? (r0 + eidx) : m1);
  const int j = owner ? (c0 + tj) : ((ekind == 0) ? (c0 + eidx) : m1);
  const size_t ij = (size_t)i * LD + j;
  double uin = 0, vin = 0, u0 = 0, v0 = 0, p0 = 0, px = 0, py = 0;
  double dxl = 0, d2xl = 0, dyl = 0, d2yl = 0, ue = 0, ve = 0, un_ = 0, vn_ = 0, lidv = 0, wxi = 0, wyj = 0, we = 0, wn = 0;
  // column nodes (r0+idx, M-1) read the transposed copies: contiguous instead of one cache line per lane
  // (16 lanes x 10 operands x a cold line each held the whole wave back by ~2 us); their grad p sits in
  // the padding row M of PX / PY, put there by stage 1
  const bool colnode = !owner && ekind == 1;
  const size_t ijT = (size_t)j * LD + i;
  if (owner || edge_thr) {
    if (owner) {     // stage input at a tile node: from the packed twin (block (I, J), element (ti, tj))
      const size_t kin = ((size_t)(I * NB + J) << 8) + (size_t)((((tj >> 2) << 4) + ti) << 2) + (tj & 3);
      uin = gl(a.UinK, kin);
      vin = gl(a.VinK, kin);
    } else {         // node of index M-1: row-major forms (never rewritten by the tiles)
      uin = gl(colnode ? a.UinT : a.Uin, colnode ? ijT : ij);
      vin = gl(colnode ? a.VinT : a.Vin, colnode ? ijT : ij);
    }
    if (!DUMP) {
      u0 = gl(colnode ? a.U0T : a.U0, colnode ? ijT : ij);
      v0 = gl(colnode ? a.V0T : a.V0, colnode ? ijT : ij);
    }
    if (owner && !DUMP && a.Pout != nullptr) p0 = gl(a.P0, ij);
    if (!GPV && (owner || VEL == 1)) {
      const size_t ip = colnode ? (size_t)M * LD + i : ij;
      px = gl(a.PX, ip); py = gl(a.PY, ip);
    }
    if (a.tail) {
      dxl = gl(a.DxL, i); d2xl = gl(a.D2xL, i); dyl = gl(a.DyL, j); d2yl = gl(a.D2yL, j);
      ue = gl(a.Uin, (size_t)m1 * LD + j); ve = gl(a.Vin, (size_t)m1 * LD + j);    // east-wall row
      un_ = gl(a.UinT, (size_t)m1 * LD + i); vn_ = gl(a.VinT, (size_t)m1 * LD + i);  // lid column, read along the transposed copy
      if (DIAG == 2) { we = gl(a.W, (size_t)m1 * LD + j); wn = gl(a.WT, (size_t)m1 * LD + i); }
    }
    lidv = gl(a.ulid, i);
    // (only loaded here: their product is formed in the epilogue -- arithmetic on a loaded value at this point
    //  makes the wave wait for every load issued so far, first fragments included: +1.2 us before the K loop)
    if (LAST || DIAG != 0) { wxi = gl(a.wx, i); wyj = gl(a.wy, j); }
  }

  LDC_STAMP(1);
  // Everything issued so far (LDS-direct rows, first fragments, pointwise operands) is drained HERE, once: with
  // LDS-direct loads possibly outstanding the compiler otherwise protects every later LDS read and, after the loop's
  // joins, every group's MFMAs with vmcnt(0) -- which also waits for the prefetch just issued (all variants with
  // index-(M-1) handling had that; with the drain the loop waits with vmcnt(8..14) as intended).  Same-box A/B at
  // N=256: no drain 55.1, drain here 53.9, drain before the pointwise loads are issued 54.3 us per iteration.
  if (EDGES || BATCH) __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0), lgkmcnt / expcnt untouched
  // Batched launches: the work-groups of a trial whose latch has fired leave HERE, before the K loop, so that a
  // batch gets cheaper as its trials converge (trials of one batch can differ by 25 % in iteration count).  The
  // wait above keeps the prologue's loads in front of this branch (a return any earlier gets them sunk below it).
  // (a latched stage 4 tells the post launch behind it that there is nothing new to transform: ctrl[LIVE])
  if (LAST && latched != 0 && bx == 0 && tid == 0) a.ctrl[LDC_CTRL_LIVE] = 0;
  if (BATCH && latched != 0) return;
  // (measured flat, round 2: s_setprio 1 for waves 4-7 -- the second-dispatched half, the arbitration loser of a SIMD's
  //  two waves per MI355X_MICROARCH.md -- 8.29 vs 8.35 us per plain stage launch, profiles/r02_kbench_setprio.log)
  // ---- contraction over k: this wave's quarter, loads one group ahead (A/B ping-pong) --------
  v4d acc[NA];
#pragma unroll
  for (int q = 0; q < NA; ++q) acc[q] = (v4d){0.0, 0.0, 0.0, 0.0};
  EdgeAcc ea;
#pragma unroll
  for (int q = 0; q < 5; ++q) { ea.er[q] = 0.0; ea.ec[q] = 0.0; ea.ek[q] = 0.0; }
  // Loads one group ahead (A/B ping-pong); with the packed twins one group ahead is enough (the probe delivers
  // the whole 278 KB of a tile in ~1.9 us whatever the depth, profiles/r01_aql_probe.log).
  for (int n = 0; n < ng; n += 2) {
    if (GP) load_extra<false>(fx, o, NB, I, J, gk(n), lane);          // first: loads return in issue order
    load_role<false>(fb, o, NB, I, J, gk(n + 1 < ng ? n + 1 : n), lane);   // clamped: harmless reload
    // the prefetch stays IN FRONT of this group's MFMAs: without the branches of the timing switches around them
    // (product build) hipcc sinks the eight loads behind the sixteen MFMAs and the next group waits with vmcnt(0)
    __builtin_amdgcn_sched_barrier(0);
    mfma_role<GP, NA>(fa, fx, acc, LDC_ABL(a, ~0), o.x4);
    if (anyE && !LDC_ABL(a, 4096)) edge_group<VEL, GP>(ea, fa, fx, o, erow, n, lane, rowE, colE, cornE);
    if (n + 1 < ng) {
      if (GP) load_extra<false>(fx, o, NB, I, J, gk(n + 1), lane);
      load_role<false>(fa, o, NB, I, J, gk(n + 2 < ng ? n + 2 : n + 1), lane);
      __builtin_amdgcn_sched_barrier(0);
      mfma_role<GP, NA>(fb, fx, acc, LDC_ABL(a, ~0), o.x4);
      if (anyE && !LDC_ABL(a, 4096)) edge_group<VEL, GP>(ea, fb, fx, o, erow, n + 1, lane, rowE, colE, cornE);
    }
  }

  LDC_STAMP(2);
  if (latched != 0) return;     // block-uniform; nothing has been stored yet
  // ---- all partial results to LDS ---------------------------------------------------------------
#pragma unroll
  for (int q = 0; q < NA; ++q)
#pragma unroll
    for (int r = 0; r < 4; ++r) red[((wv * NA + q) * 4 + r) * 64 + lane] = acc[q][r];
  if (anyE) {
    double* er = lds + L::EDGE + wv * 15 * 16;
#pragma unroll
    for (int q = 0; q < 5; ++q) {
      if ((q == 0 && VEL == 0) || (q >= 1 && q < 4 && VEL != 1) || (q == 4 && !GP)) continue;
      // only the kinds this tile carries (block-uniform; a reader only looks at its own kind's slots)
      if (rowE) { const double x = quad_sum(ea.er[q]); if (lane < 16) er[q * 16 + lane] = x; }
      if (colE) { const double y = quad_sum(ea.ec[q]); if (lane < 16) er[(5 + q) * 16 + lane] = y; }
      if (cornE) { const double z = quad_sum(ea.ek[q]); if (lane < 16) er[(10 + q) * 16 + lane] = z; }
    }
  }
  __syncthreads();

  LDC_STAMP(3);
  // sum of one accumulator over the four K-quarters of a role, at this thread's node
  auto rsum = [&](int rl, int q) {
    const int w = wv & 3;
    double x = red[(((rl * 4 + 0) * NA + q) * 4 + w) * 64 + lane];
    x += red[(((rl * 4 + 1) * NA + q) * 4 + w) * 64 + lane];
    x += red[(((rl * 4 + 2) * NA + q) * 4 + w) * 64 + lane];
    x += red[(((rl * 4 + 3) * NA + q) * 4 + w) * 64 + lane];
    return x;
  };
  // same for an edge value: kind 0 row node, 1 column node, 2 corner; idx = position in the tile
  auto esum = [&](int rl, int kind, int q, int idx) {
    const double* base = lds + L::EDGE + (kind * 5 + q) * 16 + idx;
    double x = base[(rl * 4 + 0) * 15 * 16];
    x += base[(rl * 4 + 1) * 15 * 16];
    x += base[(rl * 4 + 2) * 15 * 16];
    x += base[(rl * 4 + 3) * 15 * 16];
    return x;
  };

  double sums[PS_NSUM] = {0, 0, 0, 0, 0, 0, 0, 0};
  double maxs[2] = {0, 0};
  double un = 0.0, vn = 0.0;
  double dsum = 0.0;                 // this thread's share of the enstrophy (DIAG 1) / palinstrophy (DIAG 2) sum
  double* tu = lds + L::TILE;
  double* tv = tu + 16 * 17;
  double* tw = tv + 16 * 17;
  double* tp = tw + 16 * 17;

  const double wq = wxi * wyj;
  if (owner || edge_thr) {
    // ---- one code path for tile nodes and for the nodes of index M-1 ------------------------------
    // role 0: c0 du/dx, c1 dv/dx, c2 d2u/dx2, c3 d2v/dx2, c4 dp/dx | d(omega)/dx
    // role 1: c0 du/dy, c2 dv/dy, c1 d2u/dy2, c3 d2v/dy2, c4 dp/dy | d(omega)/dy
    auto C = [&](int rl, int q) { return owner ? rsum(rl, q) : esum(rl, ekind, q, eidx); };
    // (Mx = My = M except for nx != ny, which runs in the non-tail layout: beyond Mx / My everything is zero padding)
    const int Mx = RECT ? a.Mx : M, My = RECT ? a.My : M;
    const bool valid = (i < Mx) && (j < My);
    const bool interior = (i >= 1) && (i <= Mx - 2) && (j >= 1) && (j <= My - 2);
    const bool full = owner || VEL == 1;      // all four velocity contractions are available here
    double ux = 0, vx = 0, uy = 0, vy = 0, lu = 0, lv = 0;
    if (full) {
      ux = C(0, 0); vx = C(0, 1); lu = C(0, 2) + C(1, 1); lv = C(0, 3) + C(1, 3);
      uy = C(1, 0); vy = C(1, 2);
    } else if (VEL == 2) {                     // edge node in stage 1: slot 0 holds what omega needs
      vx = C(0, 0); uy = C(1, 0);
    }
    if (a.tail) {
      // k = M-1 lies outside the MFMA range: exact rank-1 completion of every contraction
      nm_tail(ux, vx, uy, vy, lu, lv, dxl, d2xl, dyl, d2yl, ue, ve, un_, vn_);
    }
    if (GPV) {
      px = valid ? C(0, 4) : 0.0;   // row M-1 of T1T/T2T meets a zero column of GxF/IxF: no completion term
      py = valid ? C(1, 4) : 0.0;
      if (!DUMP) {
        st_out(a.PX + ij, px, a.wt); st_out(a.PY + ij, py, a.wt);
        if (colnode) { st_out(a.PX + (size_t)M * LD + i, px, a.wt); st_out(a.PY + (size_t)M * LD + i, py, a.wt); }
      }
    }
    if (DIAG == 1) {
      const double w = valid ? (vx - uy) : 0.0;          // sg.py:510-522 on phi^(n+1) (= this stage's input)
      // a tile node's omega goes to the packed twins only (below): nobody reads the row-major body of W / WT before
      // the stand-alone omega pass that closes an enqueue rewrites it; the nodes of index M-1 are read row-major
      // by stage 2 (rank-1 completion and job rows)
      if (owner) tw[ti * 17 + tj] = w;
      else { st_out(a.W + ij, w, a.wt); st_out(a.WT + (size_t)j * LD + i, w, a.wt); }
      dsum = valid ? wq * w * w : 0.0;
    }
    if (DIAG == 2) {
      double gx = C(0, 4), gy = C(1, 4);                 // sg.py:546-547
      if (a.tail) { gx = nm_madd(dxl, we, gx); gy = nm_madd(wn, dyl, gy); }
      dsum = valid ? wq * nm_sq2(gx, gy) : 0.0;
    }
    const double Ru = nm_momentum(uin, vin, ux, uy, px, a.nu, lu);
    const double Rv = nm_momentum(uin, vin, vx, vy, py, a.nu, lv);
    const double Rp = nm_continuity(a.beta2, ux, vy);
    if (DUMP) {
      if (valid) {
        a.dump[0][ij] = ux; a.dump[1][ij] = uy; a.dump[2][ij] = vx; a.dump[3][ij] = vy;
        a.dump[4][ij] = lu; a.dump[5][ij] = lv; a.dump[6][ij] = px; a.dump[7][ij] = py;
        a.dump[8][ij] = Ru; a.dump[9][ij] = Rv;
        if (interior) a.dump[10][ij] = Rp;
      }
    } else if (owner) {
      un = nm_madd(adt, Ru, u0); vn = nm_madd(adt, Rv, v0);
      // walls first, lid last (sg.py:348-385): the lid row wins the two top corners
      if (!valid) { un = 0.0; vn = 0.0; }
      else if (j == My - 1) { un = lidv; vn = 0.0; }
      else if (i == 0 || i == Mx - 1 || j == 0) { un = 0.0; vn = 0.0; }
      if (a.rm_out) {
        st_out(a.Uout + ij, un, a.wt);
        st_out(a.Vout + ij, vn, a.wt);
      }
      if (a.Pout != nullptr && !LDC_ABL(a, 32)) {
        const double pn = interior ? nm_madd(adt, Rp, p0) : 0.0;
        st_out(a.Pout + ij, pn, a.wt);
        tp[ti * 17 + tj] = pn;
      }
      tu[ti * 17 + tj] = un;
      tv[ti * 17 + tj] = vn;
      if (LAST) {
        const double du = un - u0, dv = vn - v0;
        sums[PS_DU2] = valid ? du * du : 0.0;
        sums[PS_DV2] = valid ? dv * dv : 0.0;
        sums[PS_U02] = valid ? u0 * u0 : 0.0;
        sums[PS_V02] = valid ? v0 * v0 : 0.0;
        sums[PS_RU2] = valid ? Ru * Ru : 0.0;
        sums[PS_RV2] = valid ? Rv * Rv : 0.0;
        sums[PS_RP2] = interior ? Rp * Rp : 0.0;
        sums[PS_E] = valid ? wq * nm_sq2(un, vn) : 0.0;
        maxs[0] = fabs(un);
        maxs[1] = fabs(vn);
      }
    } else if (LAST) {
      // node of index M-1: its value is a boundary condition and never changes
      sums[PS_U02] = u0 * u0; sums[PS_V02] = v0 * v0;
      sums[PS_RU2] = Ru * Ru; sums[PS_RV2] = Rv * Rv;
      sums[PS_E] = wq * nm_sq2(u0, v0);
      maxs[0] = fabs(u0); maxs[1] = fabs(v0);
    }
  }
  if (DUMP) return;

  LDC_STAMP(4);
  __syncthreads();
  LDC_STAMP(5);
  if (owner && !LDC_ABL(a, 8)) {
    const int tr = tid >> 4, tc = tid & 15;   // write UT[c0+tr][r0+tc] = tile[tc][tr]
    const size_t ot = (size_t)(c0 + tr) * LD + r0 + tc;
    if (a.rm_out) {
      st_out(a.UoutT + ot, tu[tc * 17 + tr], a.wt);
      st_out(a.VoutT + ot, tv[tc * 17 + tr], a.wt);
    }
    // packed twins of this tile: block (I, J) of the array, block (J, I) of its transposed copy, 16 bytes per
    // store: thread h = tid & 127 stores doubles 2h, 2h+1 of a 2-KB block = elements (row pr, columns pc, pc+1);
    // threads 0..127 take the u (and omega) arrays, threads 128..255 the v (and p) arrays
    const int hh = tid & 127, pl = hh >> 1, pr = pl & 15, pc = 4 * (pl >> 4) + 2 * (hh & 1);
    const size_t kb = ((size_t)(I * NB + J) << 8) + 2 * hh, kbT = ((size_t)(J * NB + I) << 8) + 2 * hh;
    const int e = pr * 17 + pc, eT = pc * 17 + pr;
    if (tid < 128) {
      st_out2(a.UoutK + kb, tu[e], tu[e + 1], a.wt); st_out2(a.UoutTK + kbT, tu[eT], tu[eT + 17], a.wt);
      if (DIAG == 1) { st_out2(a.WK + kb, tw[e], tw[e + 1], a.wt); st_out2(a.WTK + kbT, tw[eT], tw[eT + 17], a.wt); }
    } else {
      st_out2(a.VoutK + kb, tv[e], tv[e + 1], a.wt); st_out2(a.VoutTK + kbT, tv[eT], tv[eT + 17], a.wt);
      if (a.Pout != nullptr && !LDC_ABL(a, 32)) st_out2(a.PoutK + kb, tp[e], tp[e + 1], a.wt);
    }
  }
  if (DIAG != 0) {
    // one partial sum per work-group into the parity slab of the state this stage started from
    red[tid] = dsum;
    __syncthreads();
    if (wv == 0) {
      double x = 0.0;
#pragma unroll
      for (int m = 0; m < kStageWaves; ++m) x += red[lane + 64 * m];
      x = wave_sum(x);
      double* slab = (DIAG == 1 ? a.partZ0 : a.partP0) + (size_t)((step0 > 0 ? step0 - 1 : 0) & 1) * a.stride;
      if (lane == 0) slab[(size_t)bx * LDC_NPART] = x;
    }
    if (DIAG == 2 && bx == 0 && tid == 0) {      // Z and P partials of state `step0` are complete
      a.ctrl[LDC_CTRL_PDONE] = step0;
      a.ctrl[LDC_CTRL_DROWS] = nblk;
    }
  }
  if (LAST && !LDC_ABL(a, 4)) {
    // block reduction through LDS (the accumulator region is free again): one wave per value,
    // fixed order; ten shuffle trees per wave would cost far more than this transpose
#pragma unroll
    for (int q = 0; q < PS_NSUM; ++q) red[q * kStageThreads + tid] = sums[q];
    red[PS_UMAX * kStageThreads + tid] = maxs[0];
    red[PS_VMAX * kStageThreads + tid] = maxs[1];
    __syncthreads();
    double* dst = a.partials + (size_t)bx * LDC_NPART;
    {
      double x = 0.0;
#pragma unroll
      for (int m = 0; m < kStageWaves; ++m) x += red[wv * kStageThreads + lane + 64 * m];
      x = wave_sum(x);
      if (lane == 0) dst[wv] = x;       // kStageWaves == PS_NSUM
    }
    if (wv < 2) {
      double x = 0.0;
#pragma unroll
      for (int m = 0; m < kStageWaves; ++m) x = fmax(x, red[(PS_NSUM + wv) * kStageThreads + lane + 64 * m]);
      x = wave_max(x);
      if (lane == 0) dst[PS_NSUM + wv] = x;
    }
    if (bx == 0 && tid == 0) {
      a.ctrl[LDC_CTRL_STEP] = step0 + 1;   // one more state update done
      a.ctrl[LDC_CTRL_LIVE] = 1;           // and a new pressure for the post launch to transform
    }
  }
  LDC_STAMP(6);
}

template <bool GPV, bool LAST, bool DUMP, bool BATCH, int DIAG, bool RECT = false>
__global__ __launch_bounds__(kStageThreads, 2) void stage_kernel(const StageArgs a_val, const StageArgs* a_arr) {
  // by value: with a reference into device memory (BATCH) every write-through store made the compiler re-load the
  // fields it needs next (+3 us per launch)
  const StageArgs a = BATCH ? a_arr[blockIdx.y] : a_val;
  extern __shared__ __attribute__((aligned(16))) double lds[];
  stage_body<GPV, LAST, DUMP, BATCH, DIAG, RECT>(a, (int)blockIdx.x, (int)gridDim.x, lds);
}

// ---------------------------------------------------------------------------------------
// finalize: ONE work-group folds the per-block partials in a fixed order.  It rides along
// as the last block of the post launch (no launch of its own inside the iteration loop).
//   ctrl[ITER]    iterations finalized          ctrl[STEP]    stage-4 updates completed
//   ctrl[FLUSHED] records whose Z/P are filled  ctrl[DONE]    latch
// Z and P of iteration n are produced AFTER its finalize (omega tiles of the same launch,
// then the palinstrophy kernel), so they are folded one launch later ("lagged flush") from
// parity slabs; an idempotent flush launch at the end of every enqueue closes the last record.
// ---------------------------------------------------------------------------------------
// Threads 0 .. kThreads-1 of one block do the work, sm holds kThreads * (PS_N + 2) doubles.
// PERSIST = false: the finalize block of a post launch / the finalize kernel (256 threads, control words in ctrl/scal).
// PERSIST = true : a phase of the persistent trial kernel.  ALL threads of the (larger) work-group call it (the
//   barriers inside are the work-group's), the control words live in *S, the partial sums other work-groups wrote
//   in this launch are read with coherent loads, and only the work-group with `writer` set stores the record.
template <bool PERSIST>
__device__ __forceinline__ void fin_work(const FinalArgs& a, double* sm, int t, TrialState* S, bool writer) {
  // Launch path: nobody writes the control words while this block runs (it is the only writer, at its end), so every
  // wave reads them itself by scalar loads -- the same values for all, hence a block-uniform decision below.
  const int done = PERSIST ? S->done : sload(a.ctrl + LDC_CTRL_DONE), iter = PERSIST ? S->iter : sload(a.ctrl + LDC_CTRL_ITER);
  const int step = PERSIST ? S->step : sload(a.ctrl + LDC_CTRL_STEP);
  const int flushed = PERSIST ? S->flushed : sload(a.ctrl + LDC_CTRL_FLUSHED);
  const int pdone = PERSIST ? S->pdone : sload(a.ctrl + LDC_CTRL_PDONE), drows = PERSIST ? S->drows : sload(a.ctrl + LDC_CTRL_DROWS);
  // record iter-1 lacks Z, P, and the partial sums of state `iter` (its end state) are complete
  const bool flush = a.with_diag && (flushed < iter) && (pdone >= iter);
  const bool crit = a.do_critical && !done && (step > iter);       // iteration `iter` awaits its record
  // Launch path: this thread's first partial-sum row is requested BEFORE the control words above are looked at (the
  // compiler keeps the scalar loads and these in flight together): one cold round trip instead of two.  The rows
  // are harmless to read whatever the decision turns out to be.
  double pre[PS_N];
#pragma unroll
  for (int q = 0; q < PS_N; ++q) pre[q] = 0.0;
  const bool pre_row = !PERSIST && a.do_critical && t < kThreads && t < a.nblk4;
  if (pre_row) {
    const double* p = a.part4 + (size_t)t * LDC_NPART;
#pragma unroll
    for (int q = 0; q < PS_N; ++q) pre[q] = gl(p, q);
  }
  if (!flush && !crit) return;
  // The fold, in ONE fixed order whoever runs it: thread t sums rows t, t + 256, ...; the 256 x 12 values cross LDS
  // ([value][thread]); wave w takes values 3w .. 3w+2, lane l adds the four entries l, l+64, l+128, l+192 and ONE DPP
  // tree per value finishes it.  (Twelve trees in each of the four waves, the form before, were 1 200 cycles of the
  // finalize block's critical path; this is three per wave.)
  const bool active = t < kThreads;
  double v[PS_N + 2];
#pragma unroll
  for (int q = 0; q < PS_N + 2; ++q) v[q] = 0.0;
  if (crit && active) {
    int r = t;
    if (pre_row) {              // row t is already here
#pragma unroll
      for (int q = 0; q < PS_NSUM; ++q) v[q] += pre[q];
      v[PS_UMAX] = fmax(v[PS_UMAX], pre[PS_UMAX]);
      v[PS_VMAX] = fmax(v[PS_VMAX], pre[PS_VMAX]);
      r += kThreads;
    }
    for (; r < a.nblk4; r += kThreads) {
      const double* p = a.part4 + (size_t)r * LDC_NPART;
#pragma unroll
      for (int q = 0; q < PS_NSUM; ++q) v[q] += gl_t<PERSIST>(p, q);
      v[PS_UMAX] = fmax(v[PS_UMAX], gl_t<PERSIST>(p, PS_UMAX));
      v[PS_VMAX] = fmax(v[PS_VMAX], gl_t<PERSIST>(p, PS_VMAX));
    }
  }
  if (flush && active) {
    const int par = (iter - 1) & 1;
    const double* pz = a.partZ0 + (size_t)par * a.stride;
    const double* pp = a.partP0 + (size_t)par * a.stride;
    for (int r = t; r < drows; r += kThreads) v[PS_N] += gl_t<PERSIST>(pz, (size_t)r * LDC_NPART);
    for (int r = t; r < drows; r += kThreads) v[PS_N + 1] += gl_t<PERSIST>(pp, (size_t)r * LDC_NPART);
  }
  static_assert((PS_N + 2) % kWaves == 0, "the fold deals its values evenly over the four waves");
  __shared__ double fin_tot[PS_N + 2 + 2];
  if (active) {
#pragma unroll
    for (int q = 0; q < PS_N + 2; ++q) sm[q * kThreads + t] = v[q];
  }
  __syncthreads();
  if (active) {
    const int w = t >> 6, l = t & 63;
#pragma unroll
    for (int k = 0; k < (PS_N + 2) / 4; ++k) {
      const int q = w * ((PS_N + 2) / 4) + k;
      const bool is_max = (q == PS_UMAX || q == PS_VMAX);
      const double* b = sm + q * kThreads + l;
      double x;
      if (is_max) x = fmax(fmax(fmax(b[0], b[64]), b[128]), b[192]);
      else x = ((b[0] + b[64]) + b[128]) + b[192];
      x = is_max ? wave_max(x) : wave_sum(x);
      if (l == 0) fin_tot[q] = x;
    }
  }
  __syncthreads();
  if (t < 64) {
    const FinOut o = fin_decide_wave(a, fin_tot, t, crit, flush, iter, flushed, PERSIST ? S->dt : sload(a.scal + LDC_SCAL_DT),
                                     writer);
    if (t == 0) {
      if (PERSIST) {
        S->flushed = o.flushed;
        if (crit) {
          S->umax = o.umax; S->vmax = o.vmax; S->dt = o.dt; S->iter = o.iter;
          if (o.latch) S->done = o.latch;
        }
      } else {
        fin_publish(a, o, crit, flush);
      }
    }
  }
  if (PERSIST) __syncthreads();      // *S is complete for every thread of this work-group
}

template <bool BATCH>
__global__ __launch_bounds__(kThreads) void finalize_kernel(const FinalArgs a_val, const FinalArgs* a_arr) {
  __shared__ double sm[kThreads * (PS_N + 2)];
  fin_work<false>(BATCH ? a_arr[blockIdx.y] : a_val, sm, threadIdx.x, nullptr, true);
}

// ---------------------------------------------------------------------------------------
// "post" kernel: pressure transform T1T/T2T (+ vorticity and enstrophy partials)
// ---------------------------------------------------------------------------------------
struct PostArgs {
  int M, LD, T, tail, do_omega;
  const double *Dx, *Dy, *IyF, *GyF;
  const double *U, *V, *VT, *P;
  double *T1T, *T2T, *W, *WT;
  const double *wx, *wy;
  const int* ctrl;
  double* partZ0;  // parity slabs, one double per block (stride LDC_NPART)
  long long stride;
  int ungated;     // stand-alone calls: always run
  int fin_block;   // index of the finalize block in this launch, or -1
  int NB, wt;
  const double *PK, *IyFK, *GyFK;   // packed twins read by the T tiles
  double *T1TK, *T2TK;              // packed twins they keep in step
  const double *DxK, *DyK, *UK, *VTK;   // packed twins read by the omega tiles (stand-alone form)
  double *WK, *WTK;                     // packed twins of omega they keep in step (the palinstrophy kernel reads them)
  FinalArgs fin;
};

template <bool BATCH>
__global__ __launch_bounds__(kMfmaBounds) void post_kernel(const PostArgs a_val, const PostArgs* a_arr) {
  const PostArgs a = BATCH ? a_arr[blockIdx.y] : a_val;   // by value, see stage_kernel
  __shared__ __attribute__((aligned(16))) double red[kThreads * (PS_N + 2)];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int M = a.M, LD = a.LD, T = a.T, nt = T * T, m1 = M - 1;
  int b = (int)blockIdx.x;
  // The finalize block closes the iteration: change norms, |R|, E, next dt, latch, record.  It is the longest block of
  // the launch (a cold round trip for the partial sums, twelve wave reductions, the square roots): block 0, first out.
  if (a.fin_block >= 0) {
    if (b == 0) { fin_work<false>(a.fin, red, tid, nullptr, true); return; }
    b -= 1;
  }
  // T tiles: the operand fragments of this wave's first four k-groups go out BEFORE the gate below (harmless
  // reads); behind it every thread used to sit through a cold miss for two control words and a barrier before its
  // first load, and then through one exposed load latency per group (-0.6 us per iteration at N=64, neutral at 256).
  constexpr int kPre = 4;
  const bool t_tile = b < nt;
  int I = 0, J = 0;
  v4d fP[kPre], fI[kPre], fG[kPre];
  if (t_tile) {
    tile_of_block(b, T, I, J);
#pragma unroll
    for (int u = 0; u < kPre; ++u) {
      const int g = wv + u * kWaves < T ? wv + u * kWaves : 0;     // clamped: harmless reload, not used
      fP[u] = ldpk(a.PK, a.NB, I, g, lane);
      fI[u] = ldpk(a.IyFK, a.NB, J, g, lane); fG[u] = ldpk(a.GyFK, a.NB, J, g, lane);
    }
  }
  // A post launch behind a latched (no-op) stage 4 has nothing new to transform: its tiles would reproduce what
  // is there bit for bit, so they leave.  The word they look at, ctrl[LIVE], is written by that stage-4 launch --
  // complete before this one starts -- and by nobody in THIS launch: the finalize block next door flips DONE,
  // ITER and FLUSHED while the tiles run, and a gate on those could see them change under it.
  // Every wave reads the two words itself, by scalar loads issued with the prefetch above: the same values for all
  // (a block-uniform decision, there are barriers below) and no cold miss, LDS hop and barrier in front of the MFMAs.
  const int step = sload(a.ctrl + LDC_CTRL_STEP);
  if (!a.ungated && sload(a.ctrl + LDC_CTRL_LIVE) == 0) return;
  double* partZ = a.partZ0 + (size_t)((step > 0 ? step - 1 : 0) & 1) * a.stride;

  if (t_tile) {
    // ---- T1T[j][i] = sum_k P[i][k] IyF[j][k],  T2T[j][i] = sum_k P[i][k] GyF[j][k] -------
    const int r0 = 16 * I, c0 = 16 * J;
    v4d acc[2] = {(v4d){0, 0, 0, 0}, (v4d){0, 0, 0, 0}};
#pragma unroll
    for (int u = 0; u < kPre; ++u) {
      if (wv + u * kWaves < T) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          acc[0] = MFMA_F64(fP[u][s], fI[u][s], acc[0]);
          acc[1] = MFMA_F64(fP[u][s], fG[u][s], acc[1]);
        }
      }
    }
    for (int g = wv + kPre * kWaves; g < T; g += kWaves) {     // T > 16 only
      const v4d gP = ldpk(a.PK, a.NB, I, g, lane);
      const v4d gI = ldpk(a.IyFK, a.NB, J, g, lane), gG = ldpk(a.GyFK, a.NB, J, g, lane);
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        acc[0] = MFMA_F64(gP[s], gI[s], acc[0]);
        acc[1] = MFMA_F64(gP[s], gG[s], acc[1]);
      }
    }
    double s[2];
    kreduce<2>(acc, red, lane, wv, s);
    const int ti = 4 * wv + (lane >> 4), tj = lane & 15;
    __syncthreads();
    double* t1 = red;
    double* t2 = red + 16 * 17;
    t1[ti * 17 + tj] = s[0];
    t2[ti * 17 + tj] = s[1];
    __syncthreads();
    const int tr = tid >> 4, tc = tid & 15;
    const size_t o = (size_t)(c0 + tr) * LD + r0 + tc;
    const bool ok = (c0 + tr < M) && (r0 + tc < M);
    st_out(a.T1T + o, ok ? t1[tc * 17 + tr] : 0.0, a.wt);
    st_out(a.T2T + o, ok ? t2[tc * 17 + tr] : 0.0, a.wt);
    {   // packed twins: block (J, I) of T1T / T2T, 16 bytes per store: threads 0..127 T1T, 128..255 T2T; thread h of
        // a half stores doubles 2h, 2h+1 of the 2-KB block = elements (row pr, columns pc, pc+1), like the stage kernel
      const int hh = tid & 127, pl = hh >> 1, pr = pl & 15, pc = 4 * (pl >> 4) + 2 * (hh & 1);
      const size_t kbT = ((size_t)(J * a.NB + I) << 8) + 2 * hh;
      const bool ok0 = (c0 + pr < M) && (r0 + pc < M), ok1 = (c0 + pr < M) && (r0 + pc + 1 < M);
      const double* tt = tid < 128 ? t1 : t2;
      st_out2((tid < 128 ? a.T1TK : a.T2TK) + kbT, ok0 ? tt[pc * 17 + pr] : 0.0, ok1 ? tt[(pc + 1) * 17 + pr] : 0.0, a.wt);
    }
    return;
  }
  b -= nt;
  const int nPedge = a.tail ? (M + kWaves - 1) / kWaves : 0;
  if (b < nPedge) {
    // ---- tail: row M-1 of T1T/T2T = P @ IyF[M-1,:], one wave per k -----------------------
    const int k = b * kWaves + wv;
    if (k < M) {
      const double* pk = a.P + (size_t)k * LD;
      double t1, t2;
      dot_rows2<false>(pk, a.IyF + (size_t)m1 * LD, a.GyF + (size_t)m1 * LD, M, lane, t1, t2);
      if (lane == 0) { st_out(a.T1T + (size_t)m1 * LD + k, t1, a.wt); st_out(a.T2T + (size_t)m1 * LD + k, t2, a.wt); }
    }
    return;
  }
  b -= nPedge;
  if (!a.do_omega) return;
  if (b < nt) {
    // ---- omega = Dx @ V - U @ Dy^T, enstrophy partial ---------------------------------------
    int I, J;
    tile_of_block(b, T, I, J);
    const int r0 = 16 * I, c0 = 16 * J;
    v4d acc[2] = {(v4d){0, 0, 0, 0}, (v4d){0, 0, 0, 0}};
    for (int g = wv; g < T; g += kWaves) {      // operands from the packed twins, like every MFMA of the loop
      const v4d fDx = ldpk(a.DxK, a.NB, I, g, lane), fVT = ldpk(a.VTK, a.NB, J, g, lane);
      const v4d fU = ldpk(a.UK, a.NB, I, g, lane), fDy = ldpk(a.DyK, a.NB, J, g, lane);
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        acc[0] = MFMA_F64(fDx[s], fVT[s], acc[0]);
        acc[1] = MFMA_F64(fU[s], fDy[s], acc[1]);
      }
    }
    double s[2];
    kreduce<2>(acc, red, lane, wv, s);
    const int ti = 4 * wv + (lane >> 4), tj = lane & 15;
    const int i = r0 + ti, j = c0 + tj;
    const bool valid = (i < M) && (j < M);
    double vx = s[0], uy = s[1];
    if (a.tail) {
      vx += a.Dx[(size_t)i * LD + m1] * a.V[(size_t)m1 * LD + j];
      uy += a.U[(size_t)i * LD + m1] * a.Dy[(size_t)j * LD + m1];
    }
    const double w = valid ? (vx - uy) : 0.0;
    a.W[(size_t)i * LD + j] = w;
    __syncthreads();
    double* tw = red;
    tw[ti * 17 + tj] = w;
    __syncthreads();
    const int tr = tid >> 4, tc = tid & 15;
    a.WT[(size_t)(c0 + tr) * LD + r0 + tc] = tw[tc * 17 + tr];
    {   // packed twins of this tile: block (I, J) of omega, block (J, I) of its transposed copy (thread t = double t)
      const int pl = tid >> 2, pr = pl & 15, pc = 4 * (pl >> 4) + (tid & 3);
      a.WK[((size_t)(I * a.NB + J) << 8) + tid] = tw[pr * 17 + pc];
      a.WTK[((size_t)(J * a.NB + I) << 8) + tid] = tw[pc * 17 + pr];
    }
    double sums[1] = {valid ? a.wx[i] * a.wy[j] * w * w : 0.0};
    double dummy[1] = {0.0};
    block_reduce_store<1, 0>(sums, dummy, red + 16 * 17, partZ + (size_t)b * LDC_NPART, lane, wv);
    return;
  }
  b -= nt;
  {
    // ---- tail: omega on the row/column of index M-1 ------------------------------------------
    const int e = b * kWaves + wv;
    double sums[1] = {0.0};
    double dummy[1] = {0.0};
    if (e < 2 * M - 1) {
      const int i = (e < M) ? m1 : (e - M);
      const int j = (e < M) ? e : m1;
      const double vx = dot_rows(a.Dx + (size_t)i * LD, a.VT + (size_t)j * LD, M, lane);
      const double uy = dot_rows(a.U + (size_t)i * LD, a.Dy + (size_t)j * LD, M, lane);
      const double w = vx - uy;
      if (lane == 0) {
        a.W[(size_t)i * LD + j] = w;
        a.WT[(size_t)j * LD + i] = w;
        sums[0] = a.wx[i] * a.wy[j] * w * w;
      }
    }
    block_reduce_store<1, 0>(sums, dummy, red, partZ + (size_t)(nt + b) * LDC_NPART, lane, wv);
  }
}

// spins of the persistent kernels give up after this long, so that a grid always drains
constexpr unsigned long long kSpinLimitTicks = 200000000ull;    // 2 s of the 100 MHz s_memrealtime counter
typedef const __attribute__((address_space(4))) char* kernarg_ptr;

// ---------------------------------------------------------------------------------------
// small-N trial kernel: a trial's T x T work-groups on ONE XCD, contraction families per wave, resident operators
// ---------------------------------------------------------------------------------------
#include "ldc_xcd_kernel.inc"

// ---------------------------------------------------------------------------------------
// trial-per-CU kernel: one work-group = one trial, the stage state in LDS, a launch carries a batch
// ---------------------------------------------------------------------------------------
#include "ldc_cu_kernel.inc"

// ---------------------------------------------------------------------------------------
// chip-wide trial kernel: a trial's T x T work-groups one per CU on all XCDs, operator panels resident in LDS,
// write-through tiles + per-work-group flags handed to the row / column mates
// ---------------------------------------------------------------------------------------
#include "ldc_wide_kernel.inc"

// ---------------------------------------------------------------------------------------
// palinstrophy kernel:  P = 1/2 sum W ((Dx w)^2 + (w Dy^T)^2)
// ---------------------------------------------------------------------------------------
struct PalinArgs {
  int M, LD, T, tail;
  const double *Dx, *Dy, *W, *WT, *wx, *wy;
  int* ctrl;
  double* partP0;
  long long stride;
  int ungated;
  int NB;
  const double *DxK, *DyK, *WK, *WTK;      // packed twins: the operands of the tiles
};

template <bool BATCH>
__global__ __launch_bounds__(kMfmaBounds) void palin_kernel(const PalinArgs a_val, const PalinArgs* a_arr) {
  const PalinArgs& a = BATCH ? a_arr[blockIdx.y] : a_val;
  __shared__ __attribute__((aligned(16))) double red[kWaves * 2 * 4 * 64];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int M = a.M, LD = a.LD, T = a.T, nt = T * T, m1 = M - 1;
  const int b = (int)blockIdx.x;
  __shared__ int gate[2];            // block-uniform decision, see post_kernel
  if (tid == 0) {
    const int st0 = a.ctrl[LDC_CTRL_STEP];
    gate[0] = st0;
    gate[1] = (!a.ungated && a.ctrl[LDC_CTRL_DONE] != 0 && a.ctrl[LDC_CTRL_FLUSHED] >= st0) ? 1 : 0;
  }
  __syncthreads();
  const int step = gate[0];
  if (gate[1]) return;
  double* partP = a.partP0 + (size_t)((step > 0 ? step - 1 : 0) & 1) * a.stride;
  double sums[1] = {0.0};
  double dummy[1] = {0.0};
  if (b < nt) {
    int I, J;
    tile_of_block(b, T, I, J);
    const int r0 = 16 * I, c0 = 16 * J;
    v4d acc[2] = {(v4d){0, 0, 0, 0}, (v4d){0, 0, 0, 0}};
    for (int g = wv; g < T; g += kWaves) {
      const v4d fDx = ldpk(a.DxK, a.NB, I, g, lane), fWT = ldpk(a.WTK, a.NB, J, g, lane);
      const v4d fW = ldpk(a.WK, a.NB, I, g, lane), fDy = ldpk(a.DyK, a.NB, J, g, lane);
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        acc[0] = MFMA_F64(fDx[s], fWT[s], acc[0]);
        acc[1] = MFMA_F64(fW[s], fDy[s], acc[1]);
      }
    }
    double s[2];
    kreduce<2>(acc, red, lane, wv, s);
    const int i = r0 + 4 * wv + (lane >> 4), j = c0 + (lane & 15);
    const bool valid = (i < M) && (j < M);
    double gx = s[0], gy = s[1];
    if (a.tail) {
      gx += a.Dx[(size_t)i * LD + m1] * a.W[(size_t)m1 * LD + j];
      gy += a.W[(size_t)i * LD + m1] * a.Dy[(size_t)j * LD + m1];
    }
    sums[0] = valid ? a.wx[i] * a.wy[j] * (gx * gx + gy * gy) : 0.0;
  } else {
    const int e = (b - nt) * kWaves + wv;
    if (e < 2 * M - 1) {
      const int i = (e < M) ? m1 : (e - M);
      const int j = (e < M) ? e : m1;
      const double gx = dot_rows(a.Dx + (size_t)i * LD, a.WT + (size_t)j * LD, M, lane);
      const double gy = dot_rows(a.W + (size_t)i * LD, a.Dy + (size_t)j * LD, M, lane);
      if (lane == 0) sums[0] = a.wx[i] * a.wy[j] * (gx * gx + gy * gy);
    }
  }
  block_reduce_store<1, 0>(sums, dummy, red, partP + (size_t)b * LDC_NPART, lane, wv);
  if (b == 0 && tid == 0) {          // Z (post, omega blocks) and P partials of state `step` are complete
    a.ctrl[LDC_CTRL_PDONE] = step;
    a.ctrl[LDC_CTRL_DROWS] = (int)gridDim.x;
  }
}

// ---------------------------------------------------------------------------------------
// finalize: one work-group folds the per-block partials in a fixed order
// ---------------------------------------------------------------------------------------
// E, Z, P of the current state in one go: folds the Z/P partial slabs (written by an ungated
// post + palin pair just before, parity slab selected by ctrl[STEP] exactly as they do) and
// integrates the kinetic energy directly.  One work-group, fixed order.
__global__ __launch_bounds__(kThreads) void quantities_parity_kernel(const double* U, const double* V, const double* wx,
                                                                    const double* wy, int M, int LD, const double* partZ0,
                                                                    const double* partP0, long long stride, int nblk,
                                                                    const int* ctrl, double* out) {
  __shared__ double sm[3 * kThreads];
  const int step = ctrl[LDC_CTRL_STEP];
  const size_t off = (size_t)((step > 0 ? step - 1 : 0) & 1) * (size_t)stride;
  const double *partZ = partZ0 + off, *partP = partP0 + off;
  const int t = threadIdx.x;
  double e = 0.0, z = 0.0, p = 0.0;
  for (int q = t; q < M * M; q += kThreads) {
    const int i = q / M, j = q % M;
    const double u = U[(size_t)i * LD + j], v = V[(size_t)i * LD + j];
    e += wx[i] * wy[j] * (u * u + v * v);
  }
  for (int r = t; r < nblk; r += kThreads) { z += partZ[(size_t)r * LDC_NPART]; p += partP[(size_t)r * LDC_NPART]; }
  sm[t] = e; sm[kThreads + t] = z; sm[2 * kThreads + t] = p;
  __syncthreads();
  for (int h = kThreads / 2; h > 0; h >>= 1) {
    if (t < h) {
      sm[t] += sm[t + h]; sm[kThreads + t] += sm[kThreads + t + h]; sm[2 * kThreads + t] += sm[2 * kThreads + t + h];
    }
    __syncthreads();
  }
  if (t == 0) { out[0] = 0.5 * sm[0]; out[1] = 0.5 * sm[kThreads]; out[2] = 0.5 * sm[2 * kThreads]; }
}

// dt of the very first iteration: max |u|, max |v| over the whole padded arrays
__global__ __launch_bounds__(kThreads) void prime_kernel(const double* U, const double* V, int n, FinalArgs a) {
  __shared__ double sm[2 * kWaves];
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  double mu = 0.0, mv = 0.0;
  for (int q = t; q < n; q += kThreads) { mu = fmax(mu, fabs(U[q])); mv = fmax(mv, fabs(V[q])); }
  mu = wave_max(mu); mv = wave_max(mv);
  if (lane == 0) { sm[wv] = mu; sm[kWaves + wv] = mv; }
  __syncthreads();
  if (t == 0) {
    mu = fmax(fmax(sm[0], sm[1]), fmax(sm[2], sm[3]));
    mv = fmax(fmax(sm[4], sm[5]), fmax(sm[6], sm[7]));
    a.scal[LDC_SCAL_UMAX] = mu;
    a.scal[LDC_SCAL_VMAX] = mv;
    a.scal[LDC_SCAL_DT] = next_dt(mu, mv, a);
  }
}

// ---------------------------------------------------------------------------------------
// generic NT product (stream-function solve)
// ---------------------------------------------------------------------------------------
// packed twin of a row-major LD x LD array: one work-group per 16 x 16 block, thread t stores double t
__global__ __launch_bounds__(kThreads) void pack_kernel(const double* src, double* dst, int LD) {
  const int NB = LD >> 4, R = (int)blockIdx.x / NB, G = (int)blockIdx.x % NB, t = threadIdx.x;
  const int l = t >> 2, r = l & 15, c = 4 * (l >> 4) + (t & 3);
  dst[((size_t)blockIdx.x << 8) + t] = src[(size_t)(16 * R + r) * LD + 16 * G + c];
}

struct GemmArgs {
  const double *A, *B;
  double* C;
  int R, K, LD, transpose_out, scale_mode;
  const double *lam_r, *lam_c;
};

__global__ __launch_bounds__(kMfmaBounds) void gemm_nt_kernel(const GemmArgs a) {
  __shared__ __attribute__((aligned(16))) double red[kWaves * 1 * 4 * 64 + 16 * 17];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int I = blockIdx.y, J = blockIdx.x, r0 = 16 * I, c0 = 16 * J;
  v4d acc[1] = {(v4d){0, 0, 0, 0}};
  for (int g = wv; g < a.K; g += kWaves) {
    const v4d fA = ldfrag(a.A, a.LD, r0, 16 * g, lane), fB = ldfrag(a.B, a.LD, c0, 16 * g, lane);
#pragma unroll
    for (int s = 0; s < 4; ++s) acc[0] = MFMA_F64(fA[s], fB[s], acc[0]);
  }
  double s[1];
  kreduce<1>(acc, red, lane, wv, s);
  const int ti = 4 * wv + (lane >> 4), tj = lane & 15;
  double c = s[0];
  if (a.scale_mode == 1) c /= (a.lam_r[r0 + ti] + a.lam_c[c0 + tj]);
  if (!a.transpose_out) {
    a.C[(size_t)(r0 + ti) * a.LD + c0 + tj] = c;
  } else {
    double* tt = red + kWaves * 4 * 64;
    tt[ti * 17 + tj] = c;
    __syncthreads();
    const int tr = tid >> 4, tc = tid & 15;
    a.C[(size_t)(c0 + tr) * a.LD + r0 + tc] = tt[tc * 17 + tr];
  }
}

// ---------------------------------------------------------------------------------------
// vortex extrema (sg.py:621-709): first index in C order wins ties, like numpy.argmin/argmax
// ---------------------------------------------------------------------------------------
struct Best { double v; int idx; };
__device__ __forceinline__ void take_min(Best& b, double v, int idx) {
  if (v < b.v || (v == b.v && idx < b.idx)) { b.v = v; b.idx = idx; }
}
__device__ __forceinline__ void take_max(Best& b, double v, int idx) {
  if (v > b.v || (v == b.v && idx < b.idx)) { b.v = v; b.idx = idx; }
}

__global__ __launch_bounds__(1024) void extrema_kernel(const double* Psi, const double* W, const double* x,
                                                      const double* y, int M, int My, int LD, double* out_val,
                                                      int32_t* out_idx) {
  __shared__ double sv[5][1024];
  __shared__ int si[5][1024];
  const int t = threadIdx.x;
  const double inf = __builtin_huge_val();
  Best b[5] = {{inf, 0x7fffffff}, {-inf, 0x7fffffff}, {-inf, 0x7fffffff}, {-inf, 0x7fffffff}, {-inf, 0x7fffffff}};
  for (int q = t; q < M * My; q += 1024) {       // (M x My nodes: x index i < M, y index j < My)
    const int i = q / My, j = q % My;
    const int idx = i * LD + j;
    const double p = Psi[idx], w = W[idx], xi = x[i], yj = y[j];
    take_min(b[0], p, q);
    take_max(b[1], fabs(w), q);
    take_max(b[2], (xi > 0.5 && yj < 0.5) ? p : -inf, q);
    take_max(b[3], (xi < 0.5 && yj < 0.5) ? p : -inf, q);
    take_max(b[4], (xi < 0.5 && yj > 0.5) ? p : -inf, q);
  }
  for (int k = 0; k < 5; ++k) { sv[k][t] = b[k].v; si[k][t] = b[k].idx; }
  __syncthreads();
  for (int h = 512; h > 0; h >>= 1) {
    if (t < h) {
      for (int k = 0; k < 5; ++k) {
        Best m = {sv[k][t], si[k][t]};
        if (k == 0) take_min(m, sv[k][t + h], si[k][t + h]); else take_max(m, sv[k][t + h], si[k][t + h]);
        sv[k][t] = m.v; si[k][t] = m.idx;
      }
    }
    __syncthreads();
  }
  if (t < 5) {
    const int q = si[t][0];
    const int i = q / My, j = q % My;
    out_idx[t] = i * LD + j;
    out_val[t] = (t == 1) ? W[i * LD + j] : sv[t][0];
  }
}

// ---------------------------------------------------------------------------------------
// hardware self-tests
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void mfma_selftest_kernel(const double* A, const double* B, double* D) {
  const int l = threadIdx.x;
  const double a = A[(l & 15) * 4 + (l >> 4)];   // A is 16 x 4 row-major
  const double b = B[(l >> 4) * 16 + (l & 15)];  // B is 4 x 16 row-major
  v4d c = {0, 0, 0, 0};
  c = MFMA_F64(a, b, c);
#pragma unroll
  for (int r = 0; r < 4; ++r) D[((l >> 4) + 4 * r) * 16 + (l & 15)] = c[r];
}

// (launch bounds of 512: with fewer threads hipcc keeps the accumulators in AGPRs and copies all 64 registers to VGPRs and back
//  around every trip of the loop -- 140 cycles per MFMA and wave instead of 64, which rounds 1 to 3 reported as the chip's
//  sustained rate, 49 TFLOP/s; tools/probes/mfma_rate_probe.hip, profiles/r04_mfma_rate_probe.log)
__global__ __launch_bounds__(512) void mfma_peak_kernel(double* sink, int iters) {
  const int l = threadIdx.x;
  double a = 1.0 + 1e-9 * l, b = 1.0 - 1e-9 * l;
  v4d c[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) c[q] = (v4d){0.0, 0.0, 0.0, (double)q};
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int q = 0; q < 8; ++q) c[q] = MFMA_F64(a, b, c[q]);
  }
  double s = 0.0;
#pragma unroll
  for (int q = 0; q < 8; ++q) s += c[q][0] + c[q][1] + c[q][2] + c[q][3];
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (s == 123.456) sink[blockIdx.x * kThreads + l + 2] = s;
  if (blockIdx.x == 0 && l == 0) {   // shader cycles and 100 MHz ticks spent in the loop
    sink[0] = (double)(t1 - t0);
    sink[1] = (double)(r1 - r0);
  }
}

}  // namespace

// =======================================================================================
// host side
// =======================================================================================
struct ldc_solver {
  ldc_problem p;
  int device;        // HIP device that was current at ldc_solver_create: every launch of this handle belongs there
  int nt;            // T*T
  int n_edge_blocks; // blocks of 4 edge nodes (tail case), else 0
  int n_pedge_blocks;
  int iters_per_graph;
  int ablate;
  double* stamps;            // ldc_debug_stamps
  hipGraphExec_t graph[2];   // [with_diagnostics]
  hipGraphExec_t chunk_graph[2];   // a whole enqueue of chunk_iters[] iterations on one of the trial kernels (modes 3, 5) with its closing launches
  int chunk_iters[2];
  hipStream_t capture_stream;
  int persist_mode;          // -1 auto, 0 launch per stage, 3 small-N kernel, 4 trial-per-CU kernel, 5 chip-wide kernel
  int n_cus;                 // compute units of the handle's device
  int n_xcds;                // its XCDs (gfx950: 32 active CUs each; a CPX partition is one)
};


// B independent trials of identical geometry advanced by every launch (blockIdx.y = trial).
struct ldc_batch {
  int B;
  std::vector<ldc_solver*> s;
  StageArgs* d_stage[4];     // device argument arrays, one entry per trial
  PostArgs* d_post[2];       // [with_diagnostics], inside the loop (with finalize block)
  PostArgs* d_postT[2];      // T-only launches on PA / PB (smoother mode)
  PalinArgs* d_palin;        // stand-alone (closing) form
  PostArgs* d_post_close;    // stand-alone post with omega blocks
  FinalArgs* d_flush;
  PostArgs* d_postP;         // stand-alone transforms of P (ungated): what follows a launch of the small-N trial kernel
  XArgs* d_xargs[2];         // [with_diagnostics] argument blocks of the small-N trial kernel
  unsigned* d_xsync;         // XG_LEN launch words, then XS_LEN counter words per trial (zeroed before every launch)
  CArgs* d_cargs[2];         // [with_diagnostics] argument blocks of the trial-per-CU kernel
  WArgs* d_wargs[2];         // [with_diagnostics] argument blocks of the chip-wide kernel
  int post_grid[2], postT_grid, post_close_grid, postP_grid;
  int iters_per_graph;
  hipGraphExec_t graph[2];
  hipStream_t capture_stream;
};

namespace {

#define HIP_TRY(expr)                         \
  do {                                        \
    hipError_t _e = (expr);                   \
    if (_e != hipSuccess) return (int)_e;     \
  } while (0)

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

// write-through stores (st_out); ldc_debug_ablate bits 128 / 256 force plain / write-through for A/B timing
// (tiles = work-groups of 16 x 16 nodes in the launch, all trials of a batch together)
int write_through_policy(const ldc_solver* s, int tiles) {
#ifdef LDC_TIMING
  if (s->ablate & 128) return 0;
  if (s->ablate & 256) return 1;
#else
  (void)s;
#endif
  return tiles >= LDC_WT_MIN_TILES ? 1 : 0;
}

StageArgs make_stage_args(const ldc_solver* s, int k) {
  const ldc_problem& p = s->p;
  StageArgs a;
  memset(&a, 0, sizeof(a));
  a.M = p.M; a.LD = p.LD; a.T = p.T; a.tail = p.tail;
  a.Mx = p.Mx > 0 ? p.Mx : p.M; a.My = p.My > 0 ? p.My : p.M;
  a.nu = p.nu; a.beta2 = p.beta2;
  static const double alphas[4] = {0.25, 1.0 / 3.0, 0.5, 1.0};   // sg.py:430
  a.alpha = alphas[k];
  a.Dx = p.Dx; a.D2x = p.D2x; a.Dy = p.Dy; a.D2y = p.D2y; a.IxF = p.IxF; a.GxF = p.GxF;
  a.U0 = p.U; a.V0 = p.V; a.P0 = p.P;
  a.U0T = p.UT; a.V0T = p.VT;
  a.T1T = p.T1T; a.T2T = p.T2T; a.PX = p.PX; a.PY = p.PY;
  a.ulid = p.ulid; a.wx = p.wx; a.wy = p.wy; a.scal = p.scal; a.ctrl = p.ctrl;
  a.DxL = p.DxL; a.D2xL = p.D2xL; a.DyL = p.DyL; a.D2yL = p.D2yL;
  a.partials = p.partials;
  a.W = p.W; a.WT = p.WT;
  a.partZ0 = p.partials + p.partials_stride; a.partP0 = p.partials + 3 * p.partials_stride;
  a.stride = p.partials_stride;
  a.wt = write_through_policy(s, s->nt);
  a.rm_out = (k == 3) ? 1 : 0;
#ifdef LDC_TIMING
  a.ablate = s->stamps ? s->ablate : (s->ablate & ~64);
  if (s->ablate & 16384) a.rm_out = 0;     // (16384: timing, stage 4 without its row-major velocity stores)
  a.dump[0] = s->stamps;
#endif
  // ping-pong: 0: S0 -> A, 1: A -> B, 2: B -> A, 3: A -> S0 (in place)
  const double *in[4][4] = {{p.U, p.UT, p.V, p.VT}, {p.UA, p.UAT, p.VA, p.VAT},
                            {p.UB, p.UBT, p.VB, p.VBT}, {p.UA, p.UAT, p.VA, p.VAT}};
  double* out[4][4] = {{p.UA, p.UAT, p.VA, p.VAT}, {p.UB, p.UBT, p.VB, p.VBT},
                       {p.UA, p.UAT, p.VA, p.VAT}, {p.U, p.UT, p.V, p.VT}};
  a.Uin = in[k][0]; a.UinT = in[k][1]; a.Vin = in[k][2]; a.VinT = in[k][3];
  a.Uout = out[k][0]; a.UoutT = out[k][1]; a.Vout = out[k][2]; a.VoutT = out[k][3];
  // the same ping-pong for the packed twins
  a.NB = p.LD / 16;
  a.DxK = p.DxK; a.D2xK = p.D2xK; a.DyK = p.DyK; a.D2yK = p.D2yK; a.IxFK = p.IxFK; a.GxFK = p.GxFK;
  a.T1TK = p.T1TK; a.T2TK = p.T2TK; a.WK = p.WK; a.WTK = p.WTK;
  const double *ink[4][4] = {{p.UK, p.UTK, p.VK, p.VTK}, {p.UAK, p.UATK, p.VAK, p.VATK},
                             {p.UBK, p.UBTK, p.VBK, p.VBTK}, {p.UAK, p.UATK, p.VAK, p.VATK}};
  double* outk[4][4] = {{p.UAK, p.UATK, p.VAK, p.VATK}, {p.UBK, p.UBTK, p.VBK, p.VBTK},
                        {p.UAK, p.UATK, p.VAK, p.VATK}, {p.UK, p.UTK, p.VK, p.VTK}};
  a.UinK = ink[k][0]; a.UinTK = ink[k][1]; a.VinK = ink[k][2]; a.VinTK = ink[k][3];
  a.UoutK = outk[k][0]; a.UoutTK = outk[k][1]; a.VoutK = outk[k][2]; a.VoutTK = outk[k][3];
  if (p.stage_pressure) {
    double* pout[4] = {p.PA, p.PB, p.PA, p.P};   // FSG smoother: p_stage is consumed by the next stage
    double* poutk[4] = {p.PAK, p.PBK, p.PAK, p.PK};
    a.Pout = pout[k]; a.PoutK = poutk[k];
  } else {
    a.Pout = (k == 3) ? p.P : nullptr;           // SG never consumes the stage pressures (quirk Q1)
    a.PoutK = (k == 3) ? p.PK : nullptr;
  }
  return a;
}

// dynamic LDS above 64 KiB has to be enabled per kernel and device (hipFuncSetAttribute): done once per device by the
// first ldc_solver_create / ldc_batch_create there (ensure_kernel_attributes), so a launch itself touches nothing but its arguments.
template <bool GP, bool LAST, bool DUMP, bool BATCH, int DIAG>
int enable_stage_lds() {
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(stage_kernel<GP, LAST, DUMP, BATCH, DIAG, false>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsLimit);
  if (e != hipSuccess) return (int)e;
  e = hipFuncSetAttribute(reinterpret_cast<const void*>(stage_kernel<GP, LAST, DUMP, BATCH, DIAG, true>),
                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsLimit);
  return (int)e;
}
template <bool BATCH>
int enable_stage_lds_all() {
  int e;
  if ((e = enable_stage_lds<true, false, false, BATCH, 0>()) != 0) return e;
  if ((e = enable_stage_lds<true, true, false, BATCH, 0>()) != 0) return e;
  if ((e = enable_stage_lds<true, false, false, BATCH, 1>()) != 0) return e;
  if ((e = enable_stage_lds<false, false, false, BATCH, 2>()) != 0) return e;
  if ((e = enable_stage_lds<false, false, false, BATCH, 0>()) != 0) return e;
  if ((e = enable_stage_lds<false, true, false, BATCH, 0>()) != 0) return e;
  if (!BATCH && (e = enable_stage_lds<true, false, true, false, 0>()) != 0) return e;
  return 0;
}

template <bool GP, bool LAST, bool DUMP, bool BATCH, int DIAG>
int launch_stage_kernel(const StageArgs& a, const StageArgs* arr, int nt, int nbatch, hipStream_t st, bool rect) {
  constexpr size_t lds_bytes = StageLds<GP || DIAG == 2, GP || DIAG != 0 || LAST || DUMP>::BYTES;
  static_assert(lds_bytes <= kLdsLimit, "stage kernel LDS");
  // (rect: nx != ny; a batch has one geometry, so the caller's flag holds for every trial of it)
  if (rect) hipLaunchKernelGGL((stage_kernel<GP, LAST, DUMP, BATCH, DIAG, true>), dim3(nt, nbatch), dim3(kStageThreads), lds_bytes, st, a, arr);
  else hipLaunchKernelGGL((stage_kernel<GP, LAST, DUMP, BATCH, DIAG, false>), dim3(nt, nbatch), dim3(kStageThreads), lds_bytes, st, a, arr);
  return (int)hipGetLastError();
}

// which instantiation runs RK stage k (diag: fuse the omega / palinstrophy work into stages 1 and 2)
template <bool BATCH>
int launch_stage_any(const StageArgs& a, const StageArgs* arr, int k, bool stage_pressure, bool diag, int nt,
                     int nbatch, hipStream_t st, bool rect) {
  if (stage_pressure) {   // FSG smoother: every stage differentiates its own input pressure; no diagnostics
    if (k < 3) return launch_stage_kernel<true, false, false, BATCH, 0>(a, arr, nt, nbatch, st, rect);
    return launch_stage_kernel<true, true, false, BATCH, 0>(a, arr, nt, nbatch, st, rect);
  }
  if (k == 0) return diag ? launch_stage_kernel<true, false, false, BATCH, 1>(a, arr, nt, nbatch, st, rect)
                          : launch_stage_kernel<true, false, false, BATCH, 0>(a, arr, nt, nbatch, st, rect);
  if (k == 1) return diag ? launch_stage_kernel<false, false, false, BATCH, 2>(a, arr, nt, nbatch, st, rect)
                          : launch_stage_kernel<false, false, false, BATCH, 0>(a, arr, nt, nbatch, st, rect);
  if (k == 2) return launch_stage_kernel<false, false, false, BATCH, 0>(a, arr, nt, nbatch, st, rect);
  return launch_stage_kernel<false, true, false, BATCH, 0>(a, arr, nt, nbatch, st, rect);
}

int launch_stage(ldc_solver* s, int k, int diag, hipStream_t st) {
  const StageArgs a = make_stage_args(s, k);
  return launch_stage_any<false>(a, nullptr, k, s->p.stage_pressure != 0, diag != 0, s->nt, 1, st, s->p.Mx != s->p.My);
}

FinalArgs make_final_args(const ldc_solver* s, int with_diag, int do_critical) {
  const ldc_problem& p = s->p;
  FinalArgs a;
  memset(&a, 0, sizeof(a));
  a.nblk4 = s->nt; a.nblkZ = s->nt + s->n_edge_blocks; a.nblkP = a.nblkZ;
  a.with_diag = with_diag; a.warmup = p.warmup; a.nan_guard = p.nan_guard; a.rec_cap = p.rec_cap;
  a.do_critical = do_critical;
  a.cfl = p.cfl; a.beta2 = p.beta2; a.nu = p.nu; a.hx = p.hx_min; a.hy = p.hy_min;
  a.lid = p.lid_speed; a.tol = p.tol;
  a.part4 = p.partials; a.partZ0 = p.partials + p.partials_stride; a.partP0 = p.partials + 3 * p.partials_stride;
  a.stride = p.partials_stride;
  a.scal = p.scal; a.ctrl = p.ctrl; a.rec = p.rec;
  return a;
}

// T1T/T2T (+ omega tiles) (+ the finalize block); `loop` = inside the iteration loop
PostArgs make_post_args(const ldc_solver* s, const double* P, int do_omega, int loop, int with_diag, int* grid_out) {
  const ldc_problem& p = s->p;
  PostArgs a;
  memset(&a, 0, sizeof(a));
  a.M = p.M; a.LD = p.LD; a.T = p.T; a.tail = p.tail; a.do_omega = do_omega;
  a.Dx = p.Dx; a.Dy = p.Dy; a.IyF = p.IyF; a.GyF = p.GyF;
  a.U = p.U; a.V = p.V; a.VT = p.VT; a.P = P;
  a.NB = p.LD / 16;
  a.wt = write_through_policy(s, s->nt);
  a.PK = (P == p.PA) ? p.PAK : (P == p.PB) ? p.PBK : p.PK;
  a.IyFK = p.IyFK; a.GyFK = p.GyFK; a.T1TK = p.T1TK; a.T2TK = p.T2TK;
  a.DxK = p.DxK; a.DyK = p.DyK; a.UK = p.UK; a.VTK = p.VTK; a.WK = p.WK; a.WTK = p.WTK;
  a.T1T = p.T1T; a.T2T = p.T2T; a.W = p.W; a.WT = p.WT; a.wx = p.wx; a.wy = p.wy;
  a.ctrl = p.ctrl; a.partZ0 = p.partials + p.partials_stride; a.stride = p.partials_stride;
  a.ungated = loop ? 0 : 1;
  int grid = s->nt + s->n_pedge_blocks;
  if (do_omega) grid += s->nt + s->n_edge_blocks;
  a.fin_block = -1;
  a.fin = make_final_args(s, with_diag, 1);       // (the persistent kernel takes its finalize arguments from here too)
  if (loop) {
    a.fin_block = 0;            // one more block, in FRONT: the finalize block is the longest of the launch
    ++grid;
  }
  *grid_out = grid;
  return a;
}

int launch_post(const ldc_solver* s, const double* P, int do_omega, int loop, int with_diag, hipStream_t st) {
  int grid = 0;
  const PostArgs a = make_post_args(s, P, do_omega, loop, with_diag, &grid);
  hipLaunchKernelGGL(post_kernel<false>, dim3(grid), dim3(kThreads), 0, st, a, (const PostArgs*)nullptr);
  return (int)hipGetLastError();
}

PalinArgs make_palin_args(const ldc_solver* s, int loop) {
  const ldc_problem& p = s->p;
  PalinArgs a;
  memset(&a, 0, sizeof(a));
  a.M = p.M; a.LD = p.LD; a.T = p.T; a.tail = p.tail;
  a.Dx = p.Dx; a.Dy = p.Dy; a.W = p.W; a.WT = p.WT; a.wx = p.wx; a.wy = p.wy;
  a.ctrl = p.ctrl; a.partP0 = p.partials + 3 * p.partials_stride; a.stride = p.partials_stride;
  a.ungated = loop ? 0 : 1;
  a.NB = p.LD / 16;
  a.DxK = p.DxK; a.DyK = p.DyK; a.WK = p.WK; a.WTK = p.WTK;
  return a;
}

int launch_palin(const ldc_solver* s, int loop, hipStream_t st) {
  const PalinArgs a = make_palin_args(s, loop);
  hipLaunchKernelGGL(palin_kernel<false>, dim3(s->nt + s->n_edge_blocks), dim3(kThreads), 0, st, a,
                     (const PalinArgs*)nullptr);
  return (int)hipGetLastError();
}

int launch_finalize(const ldc_solver* s, int with_diag, int do_critical, hipStream_t st) {
  const FinalArgs a = make_final_args(s, with_diag, do_critical);
  hipLaunchKernelGGL(finalize_kernel<false>, dim3(1), dim3(kThreads), 0, st, a, (const FinalArgs*)nullptr);
  return (int)hipGetLastError();
}

// one iteration of base.py:243-313 as launches on `st`; the record's Z/P are folded by the
// next iteration's post launch or by the flush at the end of ldc_solver_enqueue
int launch_iteration(ldc_solver* s, int with_diag, hipStream_t st) {
  int e;
  for (int k = 0; k < 4; ++k) {
    if ((e = launch_stage(s, k, with_diag, st)) != 0) return e;
    if (s->p.stage_pressure && k < 3) {
      // transforms of the stage pressure just produced (PA, PB, PA) for the next stage
      const double* pk = (k == 1) ? s->p.PB : s->p.PA;
      if ((e = launch_post(s, pk, 0, 0, 0, st)) != 0) return e;
    }
  }
  // omega / Z / P of the new state ride along in stages 1 and 2 of the NEXT iteration
  return launch_post(s, s->p.P, 0, 1, with_diag, st);
}

// close the last record of an enqueue: nothing follows that would carry its omega / Z / P, so compute them
// stand-alone (ungated, idempotent), then fold
int launch_closing_diagnostics(ldc_solver* s, hipStream_t st) {
  int e;
  if ((e = launch_post(s, s->p.P, 1, 0, 0, st)) != 0) return e;
  if ((e = launch_palin(s, 0, st)) != 0) return e;
  return launch_finalize(s, 1, 0, st);
}

bool xcd_available(const ldc_solver* s);
bool cu_available(const ldc_solver* s);
bool wide_available(const ldc_solver* s);
int xcd_tiles(const ldc_solver* s);
// 0: launch per stage   3: small-N trial kernel (one XCD)   4: trial-per-CU kernel   5: chip-wide trial kernel
// (modes 1 / 2, the round-2 persistent trial kernel, are gone: it lost to mode 0 at every size and to modes 3 / 5 where a
//  persistent kernel pays -- DESIGN.md 3 keeps its measurements)
int persistent_mode(const ldc_solver* s) {
  if (s->persist_mode == 0) return 0;
  if (s->persist_mode == 5) return wide_available(s) ? 5 : 0;
  if (s->persist_mode == 4) return cu_available(s) ? 4 : 0;
  if (s->persist_mode == 3) return xcd_available(s) ? 3 : 0;
  if (s->persist_mode == -1 && xcd_available(s) && xcd_tiles(s) * xcd_tiles(s) <= LDC_XCD_AUTO_TILES) return 3;
  if (s->persist_mode == -1 && wide_available(s)) return 5;      // measured faster than the launch path at every size it covers (profiles/r04_wide_ab_*.log)
  return 0;
}

// ---- small-N trial kernel (mode 3) -----------------------------------------------------------------------------
static_assert(XLds::BYTES_NST <= kLdsLimit - 1024, "small-N trial kernel LDS (plus its static words)");
template <int T>
int enable_xcd_lds_t() {
  const void* k[3] = {reinterpret_cast<const void*>(xcd_kernel<T, false, false>), reinterpret_cast<const void*>(xcd_kernel<T, false, true>),
                      reinterpret_cast<const void*>(xcd_kernel<T, true, false>)};
  for (const void* f : k) {
    const hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kXLdsBytes<T>);
    if (e != hipSuccess) return (int)e;
  }
  return 0;
}
int enable_xcd_lds() {
  int e;
  if ((e = enable_xcd_lds_t<1>()) != 0 || (e = enable_xcd_lds_t<2>()) != 0 || (e = enable_xcd_lds_t<3>()) != 0 ||
      (e = enable_xcd_lds_t<4>()) != 0 || (e = enable_xcd_lds_t<5>()) != 0) return e;
  return 0;
}
int xcd_tiles(const ldc_solver* s) { return (s->p.M + 15) / 16; }
// every tile's work-group on one XCD, one per CU; the packed arrays hold T x T blocks; a partial-sum row per tile
bool xcd_available(const ldc_solver* s) {
  const int T = xcd_tiles(s);
  // (nx != ny runs here too -- arrays and tiling of M = max(Mx, My), node classes and ring lines from (Mx, My) -- but not in a
  //  batch: ldc_batch_create refuses unlike grids)
  return s->p.sync != nullptr && T <= kXT && T * T <= s->n_cus / s->n_xcds && s->p.LD / 16 >= T &&
         s->p.partials_stride >= (int64_t)T * T * LDC_NPART;
}
bool use_xcd(const ldc_solver* s) { return persistent_mode(s) == 3; }

XArgs make_xargs(const ldc_solver* s, int with_diag, unsigned* sync) {
  const ldc_problem& p = s->p;
  XArgs a;
  memset(&a, 0, sizeof(a));
  a.M = p.M; a.LD = p.LD; a.NB = p.LD / 16; a.T = xcd_tiles(s);
  a.Mx = p.Mx; a.My = p.My;
  a.with_diag = with_diag;
  a.nu = p.nu; a.beta2 = p.beta2;
  a.DxK = p.DxK; a.D2xK = p.D2xK; a.DyK = p.DyK; a.D2yK = p.D2yK;
  a.IxFK = p.IxFK; a.GxFK = p.GxFK; a.IyFK = p.IyFK; a.GyFK = p.GyFK;
  a.ulid = p.ulid; a.wx = p.wx; a.wy = p.wy;
  a.U = p.U; a.UT = p.UT; a.V = p.V; a.VT = p.VT; a.P = p.P;
  a.UK[0] = p.UK; a.UK[1] = p.UAK; a.UK[2] = p.UBK;
  a.UTK[0] = p.UTK; a.UTK[1] = p.UATK; a.UTK[2] = p.UBTK;
  a.VK[0] = p.VK; a.VK[1] = p.VAK; a.VK[2] = p.VBK;
  a.VTK[0] = p.VTK; a.VTK[1] = p.VATK; a.VTK[2] = p.VBTK;
  a.PK[0] = p.PK; a.PK[1] = p.PAK; a.PK[2] = p.PBK;
  a.PTK[0] = p.T1TK; a.PTK[1] = p.T2TK; a.PTK[2] = p.WTK;     // borrowed (SG uses [0] only, the smoother carries no omega)
  a.WK = p.WK; a.WTK = p.WTK;
  a.GxF = p.GxF; a.GyF = p.GyF; a.IxF = p.IxF; a.IyF = p.IyF;
  a.ring = reinterpret_cast<double*>(p.sync + LDC_SYNC_XRING);
  a.part4 = p.partials; a.partZ0 = p.partials + p.partials_stride; a.partP0 = p.partials + 3 * p.partials_stride;
  a.stride = p.partials_stride;
  a.fin = make_final_args(s, with_diag, 1);
  a.sync = sync;
  a.giveup = p.sync + LDC_SYNC_GIVEUP;
  a.stamps = s->stamps;
  return a;
}

template <typename K>
int xcd_launch_kernel(K kern, const XLaunch& xl, int nwg, int n_xcds, hipStream_t st, size_t lds_bytes) {
  // work-groups are dealt round-robin over the XCDs: (slots x tiles + 8) per XCD put at least slots x tiles of them on
  // every XCD; the surplus leaves at once
  const int per_xcd = ((xl.B + n_xcds - 1) / n_xcds) * nwg + 8;
  hipLaunchKernelGGL(kern, dim3(n_xcds * per_xcd), dim3(kStageThreads), lds_bytes, st, xl);
  return (int)hipGetLastError();
}
template <int T>
int xcd_launch_t(const XLaunch& xl, bool sp, bool diag, int n_xcds, hipStream_t st) {
  if (sp) return xcd_launch_kernel(xcd_kernel<T, true, false>, xl, T * T, n_xcds, st, kXLdsBytes<T>);
  if (diag) return xcd_launch_kernel(xcd_kernel<T, false, true>, xl, T * T, n_xcds, st, kXLdsBytes<T>);
  return xcd_launch_kernel(xcd_kernel<T, false, false>, xl, T * T, n_xcds, st, kXLdsBytes<T>);
}
int xcd_launch_any(const XLaunch& xl, bool sp, bool diag, int T, int n_xcds, hipStream_t st) {
  switch (T) {
    case 1: return xcd_launch_t<1>(xl, sp, diag, n_xcds, st);
    case 2: return xcd_launch_t<2>(xl, sp, diag, n_xcds, st);
    case 3: return xcd_launch_t<3>(xl, sp, diag, n_xcds, st);
    case 4: return xcd_launch_t<4>(xl, sp, diag, n_xcds, st);
    case 5: return xcd_launch_t<5>(xl, sp, diag, n_xcds, st);
    default: return LDC_E_ARG;
  }
}

// one trial: the launch words in [LDC_SYNC_XLAUNCH, +XG_LEN) and the flags in [LDC_SYNC_XFLAGS, +XS_LEN) of its own sync array
int launch_xcd(ldc_solver* s, int n_iters, int with_diag, hipStream_t st) {
  const int T = xcd_tiles(s);
  XLaunch xl;
  memset(&xl, 0, sizeof(xl));
  xl.B = 1; xl.n_iters = n_iters; xl.slots_per_xcd = 1;
  xl.gsync = s->p.sync + LDC_SYNC_XLAUNCH;
  xl.trials = nullptr;
  xl.one = make_xargs(s, with_diag, s->p.sync + LDC_SYNC_XFLAGS);
  static_assert(LDC_SYNC_XLAUNCH + XG_LEN <= LDC_SYNC_XFLAGS && LDC_SYNC_XFLAGS + XS_LEN <= LDC_SYNC_XRING &&
                LDC_SYNC_XRING + 2 * kXT * kXT * 64 <= LDC_SYNC_LEN, "sync array layout");
  HIP_TRY(hipMemsetAsync(s->p.sync + LDC_SYNC_XLAUNCH, 0, sizeof(uint32_t) * (LDC_SYNC_XFLAGS + XS_LEN - LDC_SYNC_XLAUNCH), st));
  return xcd_launch_any(xl, s->p.stage_pressure != 0, with_diag != 0, T, s->n_xcds, st);
}

// ---- chip-wide trial kernel (mode 5) ---------------------------------------------------------------------------
// Tail layout (index M-1 outside the tiles: T x T instead of (T+1) x (T+1) work-groups; M = 16 T + 1, SG's loop only -- the
// smoother's stages carry their own pressure: its line nodes are not built) where the tiles do not fit the chip: N = 256.  Where both
// fit (N = 96 ... 240) index M-1 inside the tiles is the faster form with diagnostics (N=128 24.3 against 26.4 us per iteration,
// N=240 32.5 against 35.1; step-only they are equal: profiles/r04_wide_ab_layouts.log) -- the boundary-line jobs cost more than
// 2 T + 1 more work-groups.  LDC_WIDE_LAYOUT=tail | tiles picks one where both are possible (tests run both forms of one size).
bool wide_tail(const ldc_solver* s) {
  if ((s->p.M - 1) % 16 != 0 || s->p.stage_pressure != 0 || s->p.Mx != s->p.My) return false;
  const int Tt = (s->p.M + 15) / 16;
  const bool tiles_fit = Tt <= kWT && Tt * Tt <= s->n_cus;
  const char* e = getenv("LDC_WIDE_LAYOUT");
  if (e != nullptr && strcmp(e, "tiles") == 0 && tiles_fit) return false;
  if (e != nullptr && strcmp(e, "tail") == 0) return true;
  return !tiles_fit;
}
int wide_tiles(const ldc_solver* s) { return wide_tail(s) ? (s->p.M - 1) / 16 : (s->p.M + 15) / 16; }
// one work-group per CU, all of them resident at once; the packed arrays hold T x T blocks; a partial-sum row per tile; the
// flags and the ring scratch of the trial in its sync array
bool wide_available(const ldc_solver* s) {
  const int T = wide_tiles(s);
  // (nx != ny: like the one-XCD kernel, in the layout with index M-1 inside the tiles)
  return s->p.sync != nullptr && T >= kWTmin && T <= kWT && T * T <= s->n_cus && s->p.LD / 16 >= T &&
         s->p.partials_stride >= (int64_t)PS_N * ((T * T + 3) & ~3) && s->p.partials_stride % 4 == 0 && wlds_bytes(T) + 256 <= kLdsLimit;
}
bool use_wide(const ldc_solver* s) { return persistent_mode(s) == 5; }
int enable_wide_lds() {
  const void* k[5] = {reinterpret_cast<const void*>(wide_kernel<false, false, false>), reinterpret_cast<const void*>(wide_kernel<false, true, false>),
                      reinterpret_cast<const void*>(wide_kernel<true, false, false>),
                      reinterpret_cast<const void*>(wide_kernel<false, false, true>), reinterpret_cast<const void*>(wide_kernel<false, true, true>)};
  for (const void* f : k) {
    const hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)wlds_bytes(kWT));
    if (e != hipSuccess) return (int)e;
  }
  return 0;
}

WArgs make_wargs(const ldc_solver* s, int with_diag) {
  const ldc_problem& p = s->p;
  WArgs a;
  memset(&a, 0, sizeof(a));
  a.M = p.M; a.LD = p.LD; a.NB = p.LD / 16; a.T = wide_tiles(s);
  a.Mx = p.Mx; a.My = p.My;
  a.tail = wide_tail(s) ? 1 : 0;
  a.with_diag = with_diag;
  a.nu = p.nu; a.beta2 = p.beta2;
  a.DxK = p.DxK; a.D2xK = p.D2xK; a.DyK = p.DyK; a.D2yK = p.D2yK; a.GxFK = p.GxFK; a.GyFK = p.GyFK;
  a.ulid = p.ulid; a.wx = p.wx; a.wy = p.wy;
  a.U = p.U; a.UT = p.UT; a.V = p.V; a.VT = p.VT; a.P = p.P;
  a.UK[0] = p.UK; a.UK[1] = p.UAK; a.UK[2] = p.UBK;
  a.UTK[0] = p.UTK; a.UTK[1] = p.UATK; a.UTK[2] = p.UBTK;
  a.VK[0] = p.VK; a.VK[1] = p.VAK; a.VK[2] = p.VBK;
  a.VTK[0] = p.VTK; a.VTK[1] = p.VATK; a.VTK[2] = p.VBTK;
  a.PK[0] = p.PK; a.PK[1] = p.PAK; a.PK[2] = p.PBK;
  a.PTK[0] = p.T1TK; a.PTK[1] = p.T2TK; a.PTK[2] = p.WTK;     // borrowed (SG uses [0] only, the smoother carries no omega)
  a.WK = p.WK; a.WTK = p.WTK;
  a.IxF = p.IxF; a.IyF = p.IyF;
  a.Dx = p.Dx; a.D2x = p.D2x; a.Dy = p.Dy; a.D2y = p.D2y; a.GxF = p.GxF; a.GyF = p.GyF;
  a.DxL = p.DxL; a.D2xL = p.D2xL; a.DyL = p.DyL; a.D2yL = p.D2yL;
  a.ring = reinterpret_cast<double*>(p.sync + LDC_SYNC_WRING);
  a.rvec = a.ring + 64 * kWT * kWT;
  a.wedge = a.rvec + 4 * 16 * kWT;
  a.jobs = a.wedge + 2 * 16 * kWT;
  a.part4 = p.partials; a.partZ0 = p.partials + p.partials_stride; a.partP0 = p.partials + 3 * p.partials_stride;
  a.stride = p.partials_stride;
  a.fin = make_final_args(s, with_diag, 1);
  a.flags = p.sync + LDC_SYNC_WFLAGS;
  a.giveup = p.sync + LDC_SYNC_GIVEUP;
  a.stamps = s->stamps;
  return a;
}
static_assert(LDC_SYNC_WFLAGS + 32 * kWT * kWT <= LDC_SYNC_WRING && LDC_SYNC_WRING + 2 * (64 * kWT * kWT + 6 * 16 * kWT + (2 * kWT + 1) * kWJobDoubles) <= LDC_SYNC_LEN,
              "sync array layout (chip-wide kernel)");

int wide_launch_any(const WLaunch& wl, int T, bool sp, bool diag, bool tail, hipStream_t st) {
  const dim3 grid(wl.B * T * T), block(kStageThreads);
  const size_t bytes = wlds_bytes(T);
  if (sp) hipLaunchKernelGGL((wide_kernel<true, false, false>), grid, block, bytes, st, wl);
  else if (tail && diag) hipLaunchKernelGGL((wide_kernel<false, true, true>), grid, block, bytes, st, wl);
  else if (tail) hipLaunchKernelGGL((wide_kernel<false, false, true>), grid, block, bytes, st, wl);
  else if (diag) hipLaunchKernelGGL((wide_kernel<false, true, false>), grid, block, bytes, st, wl);
  else hipLaunchKernelGGL((wide_kernel<false, false, false>), grid, block, bytes, st, wl);
  return (int)hipGetLastError();
}
int launch_wide(ldc_solver* s, int n_iters, int with_diag, hipStream_t st) {
  const int T = wide_tiles(s);
  WLaunch wl;
  memset(&wl, 0, sizeof(wl));
  wl.B = 1; wl.n_iters = n_iters; wl.trials = nullptr;
  wl.one = make_wargs(s, with_diag);
  HIP_TRY(hipMemsetAsync(s->p.sync + LDC_SYNC_WFLAGS, 0, sizeof(uint32_t) * 32 * T * T, st));
  return wide_launch_any(wl, T, s->p.stage_pressure != 0, with_diag != 0, wide_tail(s), st);
}

// ---- trial-per-CU kernel (mode 4) ------------------------------------------------------------------------------
constexpr size_t kCuLdsMax = kLdsLimit - 128;        // (the kernel also has 48 bytes of static LDS: its control block)
template <int T, bool EDGE>
int enable_cu_lds_t() {
  const void* k[3] = {reinterpret_cast<const void*>(cu_kernel<T, EDGE, false, false>),
                      reinterpret_cast<const void*>(cu_kernel<T, EDGE, false, true>),
                      reinterpret_cast<const void*>(cu_kernel<T, EDGE, true, false>)};
  for (const void* f : k) {
    const hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kCuLdsMax);
    if (e != hipSuccess) return (int)e;
  }
  return 0;
}
int enable_cu_lds() {
  int e;
  if ((e = enable_cu_lds_t<1, false>()) != 0 || (e = enable_cu_lds_t<2, false>()) != 0 || (e = enable_cu_lds_t<3, false>()) != 0 ||
      (e = enable_cu_lds_t<1, true>()) != 0 || (e = enable_cu_lds_t<2, true>()) != 0) return e;
  return 0;
}
// the stage state of the trial, in both orientations, and the operators fit one CU's LDS
bool cu_available(const ldc_solver* s) {
  const int M = s->p.M;
  return s->p.Mx == s->p.My && M >= 3 && M <= kCMaxM && cu_tiles(M) <= kCT && s->p.LD / 16 >= (M + 15) / 16 && cu_lds_bytes(M) <= kCuLdsMax &&
         s->p.partials_stride >= (int64_t)LDC_NPART;
}
bool use_cu(const ldc_solver* s) { return persistent_mode(s) == 4; }

CArgs make_cargs(const ldc_solver* s, int with_diag) {
  const ldc_problem& p = s->p;
  CArgs a;
  memset(&a, 0, sizeof(a));
  a.M = p.M; a.LD = p.LD; a.NB = p.LD / 16;
  a.with_diag = with_diag;
  a.nu = p.nu; a.beta2 = p.beta2;
  a.Dx = p.Dx; a.D2x = p.D2x; a.Dy = p.Dy; a.D2y = p.D2y;
  a.GxF = p.GxF; a.GyF = p.GyF; a.IxF = p.IxF; a.IyF = p.IyF;
  a.ulid = p.ulid; a.wx = p.wx; a.wy = p.wy;
  a.U = p.U; a.UT = p.UT; a.V = p.V; a.VT = p.VT; a.P = p.P;
  a.UK = p.UK; a.UTK = p.UTK; a.VK = p.VK; a.VTK = p.VTK; a.PK = p.PK;
  a.partZ0 = p.partials + p.partials_stride; a.partP0 = p.partials + 3 * p.partials_stride;
  a.stride = p.partials_stride;
  a.fin = make_final_args(s, with_diag, 1);
  a.stamps = s->stamps;
  return a;
}

template <int T, bool EDGE>
int cu_launch_t(const CLaunch& cl, bool sp, bool diag, size_t lds_bytes, hipStream_t st) {
  const dim3 grid(cl.B), block(64 * (T * T + ((EDGE || T < 3) ? kCHelp : 0)));
  if (sp) hipLaunchKernelGGL((cu_kernel<T, EDGE, true, false>), grid, block, lds_bytes, st, cl);
  else if (diag) hipLaunchKernelGGL((cu_kernel<T, EDGE, false, true>), grid, block, lds_bytes, st, cl);
  else hipLaunchKernelGGL((cu_kernel<T, EDGE, false, false>), grid, block, lds_bytes, st, cl);
  return (int)hipGetLastError();
}
int cu_launch_any(const CLaunch& cl, const ldc_solver* s0, int with_diag, hipStream_t st) {
  const int M = s0->p.M, T = cu_tiles(M);
  const bool sp = s0->p.stage_pressure != 0, diag = with_diag != 0;
  const size_t bytes = cu_lds_bytes(M);
  if (cu_edge(M)) {
    switch (T) {
      case 1: return cu_launch_t<1, true>(cl, sp, diag, bytes, st);
      case 2: return cu_launch_t<2, true>(cl, sp, diag, bytes, st);
      default: return LDC_E_ARG;
    }
  }
  switch (T) {
    case 1: return cu_launch_t<1, false>(cl, sp, diag, bytes, st);
    case 2: return cu_launch_t<2, false>(cl, sp, diag, bytes, st);
    case 3: return cu_launch_t<3, false>(cl, sp, diag, bytes, st);
    default: return LDC_E_ARG;
  }
}
int launch_cu(ldc_solver* s, int n_iters, int with_diag, hipStream_t st) {
  CLaunch cl;
  memset(&cl, 0, sizeof(cl));
  cl.B = 1; cl.n_iters = n_iters; cl.trials = nullptr;
  cl.one = make_cargs(s, with_diag);
  return cu_launch_any(cl, s, with_diag, st);
}

// Captures record kernel launches on a private non-blocking stream and nothing else, so no call anywhere needs to be
// prohibited while one runs: relaxed mode.  (In the stricter modes HIP refuses, for instance, a synchronous copy in
// ANOTHER host thread while this one captures -- sweeps advance two batches from two threads, solve_concurrently.)
constexpr hipStreamCaptureMode kCaptureMode = hipStreamCaptureModeRelaxed;

// The rare, heavy runtime operations of the library -- graph capture and instantiation, destruction of graph
// executables, the creation-time argument copy of a batch -- run one at a time per process, on ONE private non-blocking
// stream per device that is created on first use and never destroyed.  Sweeps drive the library from two host threads
// (solve_concurrently).  What aborted in round 2 was not this code path by itself: replayed pattern by pattern
// (tools/probes/stream_race_probe.py, profiles/r03_stream_race_probe.log) per-handle stream creation / capture /
// instantiation / destruction in one thread runs clean beside another thread's launches, graph replays and STREAM-level
// waits; the one failing pattern is a DEVICE-wide synchronise in the other thread while a capture is open here --
// hipDeviceSynchronize returns hipErrorStreamCaptureUnsupported, the open capture is invalidated, and torch ends the process
// when it meets the stale error in a non-throwing path.  The Python solvers therefore wait on their own stream only
// (SGSolver._sync); serialising the rare setup operations here keeps captures from ever overlapping each other and costs
// nothing.  The hot calls (graph and kernel launches on the caller's stream) take no lock.  This mutex and these streams are
// the only process-wide state of the library.
std::mutex g_setup_mutex;
hipStream_t g_setup_stream[64] = {};
hipError_t setup_stream(hipStream_t* out) {      // call with g_setup_mutex held
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  if (dev < 0 || dev >= 64) return hipErrorInvalidDevice;
  if (g_setup_stream[dev] == nullptr) {
    e = hipStreamCreateWithFlags(&g_setup_stream[dev], hipStreamNonBlocking);
    if (e != hipSuccess) return e;
  }
  *out = g_setup_stream[dev];
  return hipSuccess;
}

// A small synchronous copy that stays off the legacy stream (which would wait for, and order itself against, every
// blocking stream of the process): the private stream, waited for.
hipError_t copy_now(void* dst, const void* src, size_t bytes, hipMemcpyKind kind) {
  std::lock_guard<std::mutex> lock(g_setup_mutex);
  hipStream_t st = nullptr;
  hipError_t e = setup_stream(&st);
  if (e != hipSuccess) return e;
  e = hipMemcpyAsync(dst, src, bytes, kind, st);
  const hipError_t w = hipStreamSynchronize(st);
  return e != hipSuccess ? e : w;
}

// Dynamic LDS above 64 KiB has to be enabled per kernel and device (hipFuncSetAttribute).  ONCE per device and process, under
// the setup mutex, before the first handle of that device exists -- not at every create: a sweep creates handles in one host
// thread while another launches the same kernels (the small-N trial kernel asks for 70 KB, the stage kernels for up to 107 KB).
// Round 3's full GPU suite ended with "Fatal Python error: Aborted" twice at the two-thread test
// tests/test_fsg.py::test_gpu_config5_shape_batched_fsg_vs_oracle (profiles/r03_abort_{o,r}_gpu_tests.log: the Python frames
// of both threads, no runtime or HSA message) while every create still rewrote the attributes; the cause is INFERRED from that
// and from the aborts stopping with this change, not read off a fault record.  ldc_attribute_rounds() counts the rounds so a
// test can hold the library to "once per device" (tests/test_gpu_batched.py).
int g_attr_rounds = 0;
bool g_attrs_done[64] = {};
int ensure_kernel_attributes() {
  std::lock_guard<std::mutex> lock(g_setup_mutex);
  int dev = 0;
  HIP_TRY(hipGetDevice(&dev));
  if (dev < 0 || dev >= 64) return (int)hipErrorInvalidDevice;
  if (g_attrs_done[dev]) return 0;
  int e;
  if ((e = enable_stage_lds_all<false>()) != 0 || (e = enable_stage_lds_all<true>()) != 0 || (e = enable_xcd_lds()) != 0 ||
      (e = enable_cu_lds()) != 0 || (e = enable_wide_lds()) != 0)
    return e;
  g_attrs_done[dev] = true;
  ++g_attr_rounds;
  return 0;
}

#ifdef LDC_TIMING
// Timing probe (instrumented build, LDC_FORK_PROBE=1; RESULTS ARE WRONG): the iteration captured as a FORKED graph -- after
// stage 3 the post launch goes to a side branch beside stage 4 instead of behind it.  The branch reads what stage 4 is
// still writing (the real thing would need a div(u3) -> p kernel in front of the transforms), so only the TIMING means
// anything: it answers whether a second launch finds room beside the 256 x 512-thread work-groups of stage 4 at all.
int launch_iteration_forked(ldc_solver* s, int with_diag, hipStream_t st, hipStream_t side, hipEvent_t fork, hipEvent_t join) {
  int e;
  for (int k = 0; k < 3; ++k) if ((e = launch_stage(s, k, with_diag, st)) != 0) return e;
  HIP_TRY(hipEventRecord(fork, st));
  HIP_TRY(hipStreamWaitEvent(side, fork, 0));
  if ((e = launch_stage(s, 3, with_diag, st)) != 0) return e;
  if ((e = launch_post(s, s->p.P, 0, 1, with_diag, side)) != 0) return e;
  HIP_TRY(hipEventRecord(join, side));
  HIP_TRY(hipStreamWaitEvent(st, join, 0));
  return 0;
}
#endif

int build_graph(ldc_solver* s, int with_diag) {
  std::lock_guard<std::mutex> lock(g_setup_mutex);
  HIP_TRY(setup_stream(&s->capture_stream));
  hipGraph_t g = nullptr;
#ifdef LDC_TIMING
  static hipStream_t side = nullptr;
  static hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  const char* fp = getenv("LDC_FORK_PROBE");
  const bool forked = fp != nullptr && fp[0] == '1' && s->p.stage_pressure == 0;
  if (forked && side == nullptr) {
    HIP_TRY(hipStreamCreateWithFlags(&side, hipStreamNonBlocking));
    HIP_TRY(hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&ev_join, hipEventDisableTiming));
  }
#endif
  HIP_TRY(hipStreamBeginCapture(s->capture_stream, kCaptureMode));
  int e = 0;
#ifdef LDC_TIMING
  if (forked) {
    for (int it = 0; it < s->iters_per_graph && e == 0; ++it)
      e = launch_iteration_forked(s, with_diag, s->capture_stream, side, ev_fork, ev_join);
  } else
#endif
  for (int it = 0; it < s->iters_per_graph && e == 0; ++it) e = launch_iteration(s, with_diag, s->capture_stream);
  hipError_t ce = hipStreamEndCapture(s->capture_stream, &g);
  if (e != 0) { if (g) (void)hipGraphDestroy(g); return e; }
  if (ce != hipSuccess) return (int)ce;
  hipError_t ie = hipGraphInstantiate(&s->graph[with_diag], g, nullptr, nullptr, 0);
  (void)hipGraphDestroy(g);
  return (int)ie;
}

int enqueue_wide_chunk(ldc_solver* s, int n_iters, int with_diag, hipStream_t st) {
  int e = launch_wide(s, n_iters, with_diag, st);
  if (e) return e;
  // (the closing diagnostics' own post launch forms the transforms too: no second one in front of it)
  return with_diag ? launch_closing_diagnostics(s, st) : launch_post(s, s->p.P, 0, 0, 0, st);
}
int build_chunk_graph(ldc_solver* s, int n_iters, int with_diag) {
  std::lock_guard<std::mutex> lock(g_setup_mutex);
  if (s->chunk_graph[with_diag]) { (void)hipGraphExecDestroy(s->chunk_graph[with_diag]); s->chunk_graph[with_diag] = nullptr; s->chunk_iters[with_diag] = 0; }
  HIP_TRY(setup_stream(&s->capture_stream));
  hipGraph_t g = nullptr;
  HIP_TRY(hipStreamBeginCapture(s->capture_stream, kCaptureMode));
  const int e = enqueue_wide_chunk(s, n_iters, with_diag, s->capture_stream);
  const hipError_t ce = hipStreamEndCapture(s->capture_stream, &g);
  if (e != 0) { if (g) (void)hipGraphDestroy(g); return e; }
  if (ce != hipSuccess) return (int)ce;
  const hipError_t ie = hipGraphInstantiate(&s->chunk_graph[with_diag], g, nullptr, nullptr, 0);
  (void)hipGraphDestroy(g);
  if (ie == hipSuccess) s->chunk_iters[with_diag] = n_iters;
  return (int)ie;
}

size_t batch_bytes(int B) {
  auto up = [](size_t x) { return (x + 255) & ~size_t(255); };
  return 4 * up(sizeof(StageArgs) * B) + 6 * up(sizeof(PostArgs) * B) + up(sizeof(PalinArgs) * B) +
         up(sizeof(FinalArgs) * B) + 2 * up(sizeof(XArgs) * B) + up(sizeof(uint32_t) * (XG_LEN + (size_t)XS_LEN * B)) +
         2 * up(sizeof(CArgs) * B);
}

int batch_launch_stage(ldc_batch* b, int k, int diag, hipStream_t st) {
  const ldc_solver* s0 = b->s[0];
  const StageArgs dummy = {};
  return launch_stage_any<true>(dummy, b->d_stage[k], k, s0->p.stage_pressure != 0, diag != 0, s0->nt, b->B, st,
                                s0->p.Mx != s0->p.My);
}

int batch_launch_iteration(ldc_batch* b, int with_diag, hipStream_t st) {
  const ldc_solver* s0 = b->s[0];
  const PostArgs pdummy = {};
  int e;
  for (int k = 0; k < 4; ++k) {
    if ((e = batch_launch_stage(b, k, with_diag, st)) != 0) return e;
    if (s0->p.stage_pressure && k < 3) {
      hipLaunchKernelGGL(post_kernel<true>, dim3(b->postT_grid, b->B), dim3(kThreads), 0, st, pdummy,
                         (const PostArgs*)b->d_postT[k == 1 ? 1 : 0]);
      if ((e = (int)hipGetLastError()) != 0) return e;
    }
  }
  hipLaunchKernelGGL(post_kernel<true>, dim3(b->post_grid[with_diag], b->B), dim3(kThreads), 0, st, pdummy,
                     (const PostArgs*)b->d_post[with_diag]);
  return (int)hipGetLastError();
}

int batch_closing_diagnostics(ldc_batch* b, hipStream_t st) {
  const ldc_solver* s0 = b->s[0];
  const PostArgs pdummy = {};
  const PalinArgs qdummy = {};
  const FinalArgs fdummy = {};
  int e;
  hipLaunchKernelGGL(post_kernel<true>, dim3(b->post_close_grid, b->B), dim3(kThreads), 0, st, pdummy,
                     (const PostArgs*)b->d_post_close);
  if ((e = (int)hipGetLastError()) != 0) return e;
  hipLaunchKernelGGL(palin_kernel<true>, dim3(s0->nt + s0->n_edge_blocks, b->B), dim3(kThreads), 0, st, qdummy,
                     (const PalinArgs*)b->d_palin);
  if ((e = (int)hipGetLastError()) != 0) return e;
  hipLaunchKernelGGL(finalize_kernel<true>, dim3(1, b->B), dim3(kThreads), 0, st, fdummy, (const FinalArgs*)b->d_flush);
  return (int)hipGetLastError();
}

int batch_build_graph(ldc_batch* b, int with_diag) {
  std::lock_guard<std::mutex> lock(g_setup_mutex);
  HIP_TRY(setup_stream(&b->capture_stream));
  hipGraph_t g = nullptr;
  HIP_TRY(hipStreamBeginCapture(b->capture_stream, kCaptureMode));
  int e = 0;
  for (int it = 0; it < b->iters_per_graph && e == 0; ++it) e = batch_launch_iteration(b, with_diag, b->capture_stream);
  hipError_t ce = hipStreamEndCapture(b->capture_stream, &g);
  if (e != 0) { if (g) (void)hipGraphDestroy(g); return e; }
  if (ce != hipSuccess) return (int)ce;
  hipError_t ie = hipGraphInstantiate(&b->graph[with_diag], g, nullptr, nullptr, 0);
  (void)hipGraphDestroy(g);
  return (int)ie;
}

bool bad_ptr(const void* p) { return p == nullptr; }

// a handle is used on the device it was created on (its kernels' attributes and its graphs live there)
int on_own_device(const ldc_solver* s) {
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess || dev != s->device) return LDC_E_STATE;
  return 0;
}

}  // namespace

extern "C" {

int ldc_version(void) { return LDC_ABI_VERSION; }

const char* ldc_error_string(int code) {
  switch (code) {
    case 0: return "ok";
    case LDC_E_ARG: return "ldc: invalid argument (null pointer or inconsistent geometry)";
    case LDC_E_STATE: return "ldc: invalid solver handle/state";
    case LDC_E_NODEVICE: return "ldc: no gfx950 HIP device";
    case LDC_E_SYNC: return "ldc: persistent trial kernel gave up a barrier wait (a work-group was not resident)";
    default: return code > 0 ? hipGetErrorString((hipError_t)code) : "ldc: unknown error";
  }
}

int ldc_device_check(char* arch, int arch_len) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n == 0) return LDC_E_NODEVICE;
  int dev = 0;
  HIP_TRY(hipGetDevice(&dev));
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, dev));
  if (arch && arch_len > 0) { strncpy(arch, prop.gcnArchName, arch_len - 1); arch[arch_len - 1] = 0; }
  return strncmp(prop.gcnArchName, "gfx950", 6) == 0 ? 0 : LDC_E_NODEVICE;
}

int ldc_solver_create(const ldc_problem* d, ldc_solver** out) {
  if (!d || !out) return LDC_E_ARG;
  if (d->M < 4 || d->T < 1) return LDC_E_ARG;
  if (d->LD % 16 != 0 || d->LD < 16 * d->T + 16 || d->LD < d->M) return LDC_E_ARG;
  // either T = ceil((M-1)/16) with tail = (16 T == M-1), or every index inside the tiles (T = ceil(M/16), tail = 0)
  const bool lay_a = d->T == (d->M - 1 + 15) / 16 && d->tail == ((16 * d->T == d->M - 1) ? 1 : 0);
  const bool lay_b = d->T == (d->M + 15) / 16 && d->tail == 0;
  if (!lay_a && !lay_b) return LDC_E_ARG;
  if (d->rec_cap < 1 || (d->stage_pressure != 0 && d->stage_pressure != 1)) return LDC_E_ARG;
  if (d->tail && d->T > 16) return LDC_E_ARG;   // the index-(M-1) jobs stage four groups per wave
  {
    // nx != ny: M is the larger of the two node counts (the tiling), everything beyond Mx / My is zero padding; index
    // M-1 must lie inside the tiles (no tail layout: its rank-1 paths assume the last index of BOTH axes)
    const int Mx = d->Mx > 0 ? d->Mx : d->M, My = d->My > 0 ? d->My : d->M;
    if (Mx < 4 || My < 4 || (Mx > My ? Mx : My) != d->M) return LDC_E_ARG;
    if (Mx != My && d->tail) return LDC_E_ARG;
  }
  if (d->stage_pressure && (!d->PA || !d->PB || !d->PAK || !d->PBK)) return LDC_E_ARG;
  const void* req[] = {d->Dx, d->D2x, d->Dy, d->D2y, d->IxF, d->GxF, d->IyF, d->GyF, d->wx, d->wy, d->ulid,
                       d->DxL, d->D2xL, d->DyL, d->D2yL,
                       d->U, d->UT, d->V, d->VT, d->P, d->UA, d->UAT, d->VA, d->VAT, d->UB, d->UBT, d->VB,
                       d->VBT, d->T1T, d->T2T, d->PX, d->PY, d->W, d->WT, d->partials, d->scal, d->ctrl, d->rec,
                       d->DxK, d->D2xK, d->DyK, d->D2yK, d->IxFK, d->GxFK, d->IyFK, d->GyFK,
                       d->UK, d->UTK, d->VK, d->VTK, d->PK, d->UAK, d->UATK, d->VAK, d->VATK,
                       d->UBK, d->UBTK, d->VBK, d->VBTK, d->T1TK, d->T2TK, d->WK, d->WTK};
  for (const void* q : req) if (bad_ptr(q)) return LDC_E_ARG;
  int dev = 0;
  HIP_TRY(hipGetDevice(&dev));
  { const int e = ensure_kernel_attributes(); if (e) return e; }
  int n_cus = 0;
  HIP_TRY(hipDeviceGetAttribute(&n_cus, hipDeviceAttributeMultiprocessorCount, dev));
  if (d->sync != nullptr) {
    if ((reinterpret_cast<uintptr_t>(d->sync) & 255) != 0) return LDC_E_ARG;
  }
  ldc_solver* s = new (std::nothrow) ldc_solver;
  if (!s) return LDC_E_STATE;
  s->p = *d;
  if (s->p.Mx <= 0) s->p.Mx = d->M;
  if (s->p.My <= 0) s->p.My = d->M;
  s->device = dev;
  s->n_cus = n_cus;
  s->n_xcds = n_cus >= 64 ? n_cus / 32 : 1;
  s->persist_mode = -1;
  s->nt = d->T * d->T;
  s->n_edge_blocks = d->tail ? (2 * d->M - 1 + kWaves - 1) / kWaves : 0;
  s->n_pedge_blocks = d->tail ? (d->M + kWaves - 1) / kWaves : 0;
  if (d->partials_stride < (int64_t)(s->nt + s->n_edge_blocks) * LDC_NPART) { delete s; return LDC_E_ARG; }
  s->iters_per_graph = 64;
  s->ablate = 0;
  s->stamps = nullptr;
  s->graph[0] = s->graph[1] = nullptr;
  s->chunk_graph[0] = s->chunk_graph[1] = nullptr;
  s->chunk_iters[0] = s->chunk_iters[1] = 0;
  s->capture_stream = nullptr;
  *out = s;
  return 0;
}

int ldc_solver_destroy(ldc_solver* s) {
  if (!s) return LDC_E_STATE;
  {
    std::lock_guard<std::mutex> lock(g_setup_mutex);
    for (int q = 0; q < 2; ++q) if (s->graph[q]) (void)hipGraphExecDestroy(s->graph[q]);
    for (int q = 0; q < 2; ++q) if (s->chunk_graph[q]) (void)hipGraphExecDestroy(s->chunk_graph[q]);
  }
  delete s;
  return 0;
}

int ldc_solver_set_graph_iters(ldc_solver* s, int n) {
  if (!s || n < 1 || n > 4096) return LDC_E_ARG;
  if (s->graph[0] || s->graph[1]) return LDC_E_STATE;
  s->iters_per_graph = n;
  return 0;
}

int ldc_solver_set_persistent(ldc_solver* s, int mode) {
  if (!s) return LDC_E_STATE;
  if (mode < -1 || mode > 5 || mode == 1 || mode == 2) return LDC_E_ARG;          // (1, 2: the round-2 persistent kernel, removed)
  {
    std::lock_guard<std::mutex> lock(g_setup_mutex);          // (a chunk graph holds the kernel of the mode it was captured under)
    for (int q = 0; q < 2; ++q) if (s->chunk_graph[q]) { (void)hipGraphExecDestroy(s->chunk_graph[q]); s->chunk_graph[q] = nullptr; s->chunk_iters[q] = 0; }
  }
  s->persist_mode = mode;
  if (mode == 3 && !xcd_available(s)) { s->persist_mode = -1; return LDC_E_ARG; }
  if (mode == 4 && !cu_available(s)) { s->persist_mode = -1; return LDC_E_ARG; }
  if (mode == 5 && !wide_available(s)) { s->persist_mode = -1; return LDC_E_ARG; }
  return 0;
}

int ldc_solver_mode(ldc_solver* s) {
  if (!s) return LDC_E_STATE;
  return persistent_mode(s);
}

int ldc_solver_status(ldc_solver* s) {
  if (!s) return LDC_E_STATE;
  if (s->p.sync == nullptr) return 0;
  // (no device-wide synchronise: with two host threads at work that would also wait for the other thread's stream, and HIP
  //  refuses it outright -- and invalidates the capture -- while the other thread captures a graph: DESIGN.md 3.  The caller
  //  has waited for the stream its launches ran on; the copy goes through the library's private stream.)
  uint32_t flag = 0;
  HIP_TRY(copy_now(&flag, s->p.sync + LDC_SYNC_GIVEUP, sizeof(flag), hipMemcpyDeviceToHost));
  return flag ? LDC_E_SYNC : 0;
}

int ldc_attribute_rounds(void) {
  std::lock_guard<std::mutex> lock(g_setup_mutex);
  return g_attr_rounds;
}

int ldc_device_info(int* n_cus, int* n_xcds) {
  if (!n_cus || !n_xcds) return LDC_E_ARG;
  int dev = 0, cus = 0;
  HIP_TRY(hipGetDevice(&dev));
  HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
  *n_cus = cus;
  *n_xcds = cus >= 64 ? cus / 32 : 1;
  return 0;
}

// timing experiments: only the instrumented build (-DLDC_TIMING) has the switches; the product library refuses
int ldc_debug_ablate(ldc_solver* s, int mask) {
#ifdef LDC_TIMING
  if (!s) return LDC_E_STATE;
  s->ablate = mask;
  return 0;
#else
  (void)s; (void)mask;
  return LDC_E_STATE;
#endif
}

int ldc_debug_stamps(ldc_solver* s, double* buf) {
#ifdef LDC_TIMING
  if (!s) return LDC_E_STATE;
  s->stamps = buf;
  return 0;
#else
  (void)s; (void)buf;
  return LDC_E_STATE;
#endif
}

int ldc_timing_build(void) {
#ifdef LDC_TIMING
  return 1;
#else
  return 0;
#endif
}

int ldc_stage(ldc_solver* s, int k, void* stream) {
  if (!s) return LDC_E_STATE;
  { const int e = on_own_device(s); if (e) return e; }
  const int diag = (k & 16) ? 1 : 0;     // bit 4: the variant that also carries the fused diagnostics
  k &= 15;
  if (k < 0 || k > 3) return LDC_E_ARG;
  return launch_stage(s, k, diag, as_stream(stream));
}

int ldc_pressure_transform(ldc_solver* s, int which, void* stream) {
  if (!s) return LDC_E_STATE;
  const double* P = which == 0 ? s->p.P : which == 1 ? s->p.PA : which == 2 ? s->p.PB : nullptr;
  if (!P) return LDC_E_ARG;
  return launch_post(s, P, 0, 0, 0, as_stream(stream));
}

int ldc_diagnostics(ldc_solver* s, void* stream) {
  if (!s) return LDC_E_STATE;
  int e = launch_post(s, s->p.P, 1, 0, 0, as_stream(stream));
  if (e) return e;
  return launch_palin(s, 0, as_stream(stream));
}

int ldc_global_quantities(ldc_solver* s, double* out3, void* stream) {
  if (!s || !out3) return LDC_E_ARG;
  hipStream_t st = as_stream(stream);
  int e = launch_post(s, s->p.P, 1, 0, 0, st);
  if (e) return e;
  if ((e = launch_palin(s, 0, st)) != 0) return e;
  const ldc_problem& p = s->p;
  hipLaunchKernelGGL(quantities_parity_kernel, dim3(1), dim3(kThreads), 0, st, p.U, p.V, p.wx, p.wy, p.M, p.LD,
                     p.partials + p.partials_stride, p.partials + 3 * p.partials_stride, (long long)p.partials_stride,
                     s->nt + s->n_edge_blocks, p.ctrl, out3);
  return (int)hipGetLastError();
}

int ldc_finalize(ldc_solver* s, int with_diag, void* stream) {
  if (!s) return LDC_E_STATE;
  // stand-alone form: close the iteration, then fold its Z/P right away
  int e = launch_finalize(s, with_diag ? 1 : 0, 1, as_stream(stream));
  if (e || !with_diag) return e;
  return launch_finalize(s, 1, 0, as_stream(stream));
}

int ldc_prime(ldc_solver* s, void* stream) {
  if (!s) return LDC_E_STATE;
  int e = launch_post(s, s->p.P, 0, 0, 0, as_stream(stream));
  if (e) return e;
  const FinalArgs a = make_final_args(s, 0, 0);
  hipLaunchKernelGGL(prime_kernel, dim3(1), dim3(kThreads), 0, as_stream(stream), s->p.U, s->p.V,
                     s->p.LD * s->p.LD, a);
  return (int)hipGetLastError();
}

int ldc_solver_enqueue(ldc_solver* s, int n_iters, int with_diag, void* stream) {
  if (!s) return LDC_E_STATE;
  if (n_iters < 0) return LDC_E_ARG;
  { const int e = on_own_device(s); if (e) return e; }
  with_diag = with_diag ? 1 : 0;
  hipStream_t st = as_stream(stream);
  // (a single iteration always goes launch by launch: nothing to gain, and it is the form the host uses for the first
  //  iteration after an upload, when phi^n may still carry other values on the row / column of index M-1 than the
  //  stage buffers do -- the tile-resident kernel stages those boundary values ONCE, from phi^n)
  if (n_iters > 1 && use_xcd(s)) {
    // the small-N trial kernel, then the transforms of the final pressure in the launch path's form (row M-1 of the
    // row-major T1T / T2T in the tail layout): whatever runs next finds the state it expects
    int e = launch_xcd(s, n_iters, with_diag, st);
    if (e) return e;
    return with_diag ? launch_closing_diagnostics(s, st) : launch_post(s, s->p.P, 0, 0, 0, st);      // (either forms the transforms)
  }
  if (n_iters > 1 && use_wide(s)) {
    // the chip-wide trial kernel, then the transforms of the final pressure in the launch path's form and the closing record: five
    // nodes (the flags' memset, the kernel, up to four closing launches) as ONE graph per chunk length -- launched one by one
    // they cost 76 us of gaps per enqueue, as much as two iterations (the driver's bench line times chunks of 20)
    if (s->chunk_graph[with_diag] == nullptr || s->chunk_iters[with_diag] != n_iters) {
      const int e = build_chunk_graph(s, n_iters, with_diag);
      if (e) return e;
    }
    HIP_TRY(hipGraphLaunch(s->chunk_graph[with_diag], st));
    return 0;
  }
  if (n_iters > 1 && use_cu(s)) {
    int e = launch_cu(s, n_iters, with_diag, st);
    if (e) return e;
    return with_diag ? launch_closing_diagnostics(s, st) : launch_post(s, s->p.P, 0, 0, 0, st);
  }
  int left = n_iters;
  if (left >= s->iters_per_graph) {
    if (!s->graph[with_diag]) { int e = build_graph(s, with_diag); if (e) return e; }
    while (left >= s->iters_per_graph) {
      HIP_TRY(hipGraphLaunch(s->graph[with_diag], st));
      left -= s->iters_per_graph;
    }
  }
  for (; left > 0; --left) { int e = launch_iteration(s, with_diag, st); if (e) return e; }
  return (with_diag && n_iters > 0) ? launch_closing_diagnostics(s, st) : 0;
}

size_t ldc_batch_workspace_bytes(int n_trials) { return n_trials > 0 ? batch_bytes(n_trials) : 0; }

// The library's private stream behind everything `stream` holds at this moment: an event recorded there, waited for here.
hipError_t order_setup_behind(hipStream_t stream) {
  std::lock_guard<std::mutex> lock(g_setup_mutex);
  hipStream_t st = nullptr;
  hipError_t e = setup_stream(&st);
  if (e != hipSuccess) return e;
  hipEvent_t ev = nullptr;
  if ((e = hipEventCreateWithFlags(&ev, hipEventDisableTiming)) != hipSuccess) return e;
  e = hipEventRecord(ev, stream);
  if (e == hipSuccess) e = hipStreamWaitEvent(st, ev, 0);
  const hipError_t d = hipEventDestroy(ev);      // (released by the runtime once the wait has been served)
  return e != hipSuccess ? e : d;
}

int ldc_batch_create(ldc_solver* const* solvers, int n_trials, void* workspace, size_t workspace_bytes, void* stream,
                     ldc_batch** out) {
  if (!solvers || !out || !workspace || n_trials < 1 || n_trials > 4096) return LDC_E_ARG;
  if (workspace_bytes < batch_bytes(n_trials) || (reinterpret_cast<uintptr_t>(workspace) & 255) != 0) return LDC_E_ARG;
  const ldc_solver* s0 = solvers[0];
  if (!s0) return LDC_E_STATE;
  for (int q = 0; q < n_trials; ++q) {
    const ldc_solver* t = solvers[q];
    if (!t) return LDC_E_STATE;
    // one geometry, one tiling and one device for all: every launch takes its grid from solver 0
    if (t->p.M != s0->p.M || t->p.LD != s0->p.LD || t->p.stage_pressure != s0->p.stage_pressure) return LDC_E_ARG;
    if (t->p.Mx != s0->p.Mx || t->p.My != s0->p.My) return LDC_E_ARG;
    if (t->p.T != s0->p.T || t->p.tail != s0->p.tail || t->nt != s0->nt || t->n_edge_blocks != s0->n_edge_blocks ||
        t->n_pedge_blocks != s0->n_pedge_blocks || t->device != s0->device)
      return LDC_E_ARG;
  }
  { int e = on_own_device(s0); if (e) return e; if ((e = ensure_kernel_attributes()) != 0) return e; }
  // The argument blocks go into the caller's workspace by copies on the library's own stream: whatever the caller's stream still
  // holds for that memory (a fill that zeroes it, queued behind another worker's long launch, once wiped the blocks: round 3)
  // comes first -- by an event, not by a rule for the caller.
  HIP_TRY(order_setup_behind(static_cast<hipStream_t>(stream)));
  ldc_batch* b = new (std::nothrow) ldc_batch;
  if (!b) return LDC_E_STATE;
  b->B = n_trials;
  b->s.assign(solvers, solvers + n_trials);
  b->iters_per_graph = s0->iters_per_graph;
  b->graph[0] = b->graph[1] = nullptr;
  b->capture_stream = nullptr;
  auto up = [](size_t x) { return (x + 255) & ~size_t(255); };
  char* w = static_cast<char*>(workspace);
  auto carve = [&](size_t bytes) { char* r = w; w += up(bytes); return r; };
  hipError_t he = hipSuccess;
  auto put = [&](void* dst, const void* src, size_t bytes) {
    if (he == hipSuccess) he = copy_now(dst, src, bytes, hipMemcpyHostToDevice);
  };
  for (int k = 0; k < 4; ++k) {
    std::vector<StageArgs> h(n_trials);
    for (int q = 0; q < n_trials; ++q) {
      h[q] = make_stage_args(solvers[q], k);
      h[q].wt = write_through_policy(solvers[q], solvers[q]->nt * n_trials);
    }
    b->d_stage[k] = reinterpret_cast<StageArgs*>(carve(sizeof(StageArgs) * n_trials));
    put(b->d_stage[k], h.data(), sizeof(StageArgs) * n_trials);
  }
  for (int wd = 0; wd < 2; ++wd) {
    std::vector<PostArgs> h(n_trials);
    for (int q = 0; q < n_trials; ++q) {
      h[q] = make_post_args(solvers[q], solvers[q]->p.P, 0, 1, wd, &b->post_grid[wd]);
      h[q].wt = write_through_policy(solvers[q], solvers[q]->nt * n_trials);
    }
    b->d_post[wd] = reinterpret_cast<PostArgs*>(carve(sizeof(PostArgs) * n_trials));
    put(b->d_post[wd], h.data(), sizeof(PostArgs) * n_trials);
  }
  for (int ab = 0; ab < 2; ++ab) {
    std::vector<PostArgs> h(n_trials);
    for (int q = 0; q < n_trials; ++q) {
      h[q] = make_post_args(solvers[q], ab == 0 ? solvers[q]->p.PA : solvers[q]->p.PB, 0, 0, 0, &b->postT_grid);
      h[q].wt = write_through_policy(solvers[q], solvers[q]->nt * n_trials);
    }
    b->d_postT[ab] = reinterpret_cast<PostArgs*>(carve(sizeof(PostArgs) * n_trials));
    put(b->d_postT[ab], h.data(), sizeof(PostArgs) * n_trials);
  }
  {
    std::vector<PostArgs> hc(n_trials);
    for (int q = 0; q < n_trials; ++q) hc[q] = make_post_args(solvers[q], solvers[q]->p.P, 1, 0, 0, &b->post_close_grid);
    b->d_post_close = reinterpret_cast<PostArgs*>(carve(sizeof(PostArgs) * n_trials));
    put(b->d_post_close, hc.data(), sizeof(PostArgs) * n_trials);
    std::vector<PalinArgs> h(n_trials);
    for (int q = 0; q < n_trials; ++q) h[q] = make_palin_args(solvers[q], 0);
    b->d_palin = reinterpret_cast<PalinArgs*>(carve(sizeof(PalinArgs) * n_trials));
    put(b->d_palin, h.data(), sizeof(PalinArgs) * n_trials);
    std::vector<FinalArgs> f(n_trials);
    for (int q = 0; q < n_trials; ++q) f[q] = make_final_args(solvers[q], 1, 0);
    b->d_flush = reinterpret_cast<FinalArgs*>(carve(sizeof(FinalArgs) * n_trials));
    put(b->d_flush, f.data(), sizeof(FinalArgs) * n_trials);
  }
  {
    std::vector<PostArgs> hp(n_trials);
    for (int q = 0; q < n_trials; ++q) {
      hp[q] = make_post_args(solvers[q], solvers[q]->p.P, 0, 0, 0, &b->postP_grid);
      hp[q].wt = write_through_policy(solvers[q], solvers[q]->nt * n_trials);
    }
    b->d_postP = reinterpret_cast<PostArgs*>(carve(sizeof(PostArgs) * n_trials));
    put(b->d_postP, hp.data(), sizeof(PostArgs) * n_trials);
    // the counter words first (their addresses go into the argument blocks), then the blocks themselves
    char* xa0 = carve(sizeof(XArgs) * n_trials);
    char* xa1 = carve(sizeof(XArgs) * n_trials);
    b->d_xsync = reinterpret_cast<unsigned*>(carve(sizeof(uint32_t) * (XG_LEN + (size_t)XS_LEN * n_trials)));
    for (int wd = 0; wd < 2; ++wd) {
      std::vector<XArgs> hx(n_trials);
      for (int q = 0; q < n_trials; ++q) hx[q] = make_xargs(solvers[q], wd, b->d_xsync + XG_LEN + (size_t)XS_LEN * q);
      b->d_xargs[wd] = reinterpret_cast<XArgs*>(wd == 0 ? xa0 : xa1);
      put(b->d_xargs[wd], hx.data(), sizeof(XArgs) * n_trials);
    }
    for (int wd = 0; wd < 2; ++wd) {
      std::vector<CArgs> hc(n_trials);
      for (int q = 0; q < n_trials; ++q) hc[q] = make_cargs(solvers[q], wd);
      b->d_cargs[wd] = reinterpret_cast<CArgs*>(carve(sizeof(CArgs) * n_trials));
      put(b->d_cargs[wd], hc.data(), sizeof(CArgs) * n_trials);
    }
  }
  if (he != hipSuccess) { delete b; return (int)he; }
  *out = b;
  return 0;
}

int ldc_batch_destroy(ldc_batch* b) {
  if (!b) return LDC_E_STATE;
  {
    std::lock_guard<std::mutex> lock(g_setup_mutex);
    for (int q = 0; q < 2; ++q) if (b->graph[q]) (void)hipGraphExecDestroy(b->graph[q]);
  }
  delete b;
  return 0;
}

// Which kernel advances a batch of small trials.  The small-N kernel gives every trial an XCD: 8 (at T'^2 <= 16: up to
// 8 x floor(32 / T'^2)) trials advance at once, the rest wait; the trial-per-CU kernel advances all of them at once, each
// several times slower.  Asked for explicitly (mode 4 on every trial) it is taken whenever it applies; in auto mode from
// LDC_CU_AUTO_TRIALS (ceil(M/16) == 3: LDC_CU_AUTO_TRIALS_T3; M == 33: LDC_CU_AUTO_TRIALS_M33) trials on.
bool batch_uses_cu(const ldc_batch* b) {
  bool all_avail = true, all_asked = true, all_auto = true;
  for (const ldc_solver* t : b->s) {
    all_avail = all_avail && cu_available(t);
    all_asked = all_asked && t->persist_mode == 4;
    all_auto = all_auto && t->persist_mode == -1;
  }
  if (!all_avail) return false;
  if (all_asked) return true;
  const int M = b->s[0]->p.M;
  const int need = (M == 33) ? LDC_CU_AUTO_TRIALS_M33 : ((M + 15) / 16 >= 3 ? LDC_CU_AUTO_TRIALS_T3 : LDC_CU_AUTO_TRIALS);
  return all_auto && b->B >= need;
}

int ldc_batch_mode(ldc_batch* b) {
  if (!b) return LDC_E_STATE;
  if (batch_uses_cu(b)) return 4;
  bool all_xcd = true;
  for (const ldc_solver* t : b->s) all_xcd = all_xcd && use_xcd(t);
  return all_xcd ? 3 : 0;
}

int ldc_batch_enqueue(ldc_batch* b, int n_iters, int with_diag, void* stream) {
  if (!b) return LDC_E_STATE;
  if (n_iters < 0) return LDC_E_ARG;
  { const int e = on_own_device(b->s[0]); if (e) return e; }
  with_diag = with_diag ? 1 : 0;
  hipStream_t st = as_stream(stream);
  if (n_iters > 1 && batch_uses_cu(b)) {
    // trial-per-CU kernel: one work-group per trial, the whole batch in one launch, then the transforms of the final
    // pressures in the launch path's form
    const ldc_solver* s0 = b->s[0];
    CLaunch cl;
    memset(&cl, 0, sizeof(cl));
    cl.B = b->B; cl.n_iters = n_iters; cl.trials = b->d_cargs[with_diag];
    { const int e = cu_launch_any(cl, s0, with_diag, st); if (e) return e; }
    const PostArgs pdummy = {};
    hipLaunchKernelGGL(post_kernel<true>, dim3(b->postP_grid, b->B), dim3(kThreads), 0, st, pdummy,
                       (const PostArgs*)b->d_postP);
    { const int e = (int)hipGetLastError(); if (e) return e; }
    return with_diag ? batch_closing_diagnostics(b, st) : 0;
  }
  {
    // small-N trial kernel: every trial of the batch on an XCD of its own (as many trials per launch as the XCDs hold,
    // the rest in further launches), then the transforms of the final pressures in the launch path's form
    bool all_xcd = n_iters > 1;
    for (const ldc_solver* t : b->s) all_xcd = all_xcd && use_xcd(t);
    if (all_xcd) {
      const ldc_solver* s0 = b->s[0];
      const int T = xcd_tiles(s0), nwg = T * T;
      const int slots = (s0->n_cus / s0->n_xcds) / nwg, per_launch = slots * s0->n_xcds;
      HIP_TRY(hipMemsetAsync(b->d_xsync + XG_LEN, 0, sizeof(uint32_t) * (size_t)XS_LEN * b->B, st));
      for (int lo = 0; lo < b->B; lo += per_launch) {
        XLaunch xl;
        memset(&xl, 0, sizeof(xl));
        xl.B = (b->B - lo < per_launch) ? (b->B - lo) : per_launch;
        // slots in use on every XCD in THIS launch: exactly what the trials need, so that the over-subscription (8 more
        // work-groups per XCD than needed) can never open a slot that would not fill up
        xl.n_iters = n_iters; xl.slots_per_xcd = (xl.B + s0->n_xcds - 1) / s0->n_xcds;
        xl.gsync = b->d_xsync;
        xl.trials = b->d_xargs[with_diag] + lo;
        HIP_TRY(hipMemsetAsync(b->d_xsync, 0, sizeof(uint32_t) * XG_LEN, st));
        const int e = xcd_launch_any(xl, s0->p.stage_pressure != 0, with_diag != 0, T, s0->n_xcds, st);
        if (e) return e;
      }
      const PostArgs pdummy = {};
      hipLaunchKernelGGL(post_kernel<true>, dim3(b->postP_grid, b->B), dim3(kThreads), 0, st, pdummy,
                         (const PostArgs*)b->d_postP);
      { const int e = (int)hipGetLastError(); if (e) return e; }
      return with_diag ? batch_closing_diagnostics(b, st) : 0;
    }
  }
  int left = n_iters;
  if (left >= b->iters_per_graph) {
    if (!b->graph[with_diag]) { int e = batch_build_graph(b, with_diag); if (e) return e; }
    while (left >= b->iters_per_graph) {
      HIP_TRY(hipGraphLaunch(b->graph[with_diag], st));
      left -= b->iters_per_graph;
    }
  }
  for (; left > 0; --left) { int e = batch_launch_iteration(b, with_diag, st); if (e) return e; }
  return (with_diag && n_iters > 0) ? batch_closing_diagnostics(b, st) : 0;
}

int ldc_pack(const double* src, double* dst, int LD, void* stream) {
  if (!src || !dst || src == dst || LD < 16 || LD % 16 != 0) return LDC_E_ARG;
  hipLaunchKernelGGL(pack_kernel, dim3((LD / 16) * (LD / 16)), dim3(kThreads), 0, as_stream(stream), src, dst, LD);
  return (int)hipGetLastError();
}

int ldc_residual_debug(ldc_solver* s, int which, double* const out[11], void* stream) {
  if (!s || !out) return LDC_E_ARG;
  { const int e = on_own_device(s); if (e) return e; }
  for (int q = 0; q < 11; ++q) if (!out[q]) return LDC_E_ARG;
  if (which < 0 || which > 2) return LDC_E_ARG;
  hipStream_t st = as_stream(stream);
  int e = launch_post(s, s->p.P, 0, 0, 0, st);   // SG differentiates p^n whatever the stage (Q1)
  if (e) return e;
  StageArgs a = make_stage_args(s, which == 0 ? 0 : which == 1 ? 1 : 2);
  for (int q = 0; q < 11; ++q) a.dump[q] = out[q];
  return launch_stage_kernel<true, false, true, false, 0>(a, nullptr, s->nt, 1, st, s->p.Mx != s->p.My);
}

int ldc_gemm_nt(const double* A, const double* B, double* C, int R16, int K16, int LD, int transpose_out,
                int scale_mode, const double* lam_r, const double* lam_c, void* stream) {
  if (!A || !B || !C || R16 < 1 || K16 < 1 || LD % 16 != 0 || LD < 16 * R16 || LD < 16 * K16) return LDC_E_ARG;
  if (scale_mode != 0 && (scale_mode != 1 || !lam_r || !lam_c)) return LDC_E_ARG;
  GemmArgs a = {A, B, C, R16, K16, LD, transpose_out, scale_mode, lam_r, lam_c};
  hipLaunchKernelGGL(gemm_nt_kernel, dim3(R16, R16), dim3(kThreads), 0, as_stream(stream), a);
  return (int)hipGetLastError();
}

int ldc_poisson_fastdiag(const double* Qx, const double* Qxinv, const double* Qy, const double* Qyinv,
                         const double* lamx, const double* lamy, const double* F, double* w0, double* w1,
                         double* Psi, int Mi, int LD, void* stream) {
  if (!Qx || !Qxinv || !Qy || !Qyinv || !lamx || !lamy || !F || !w0 || !w1 || !Psi || Mi < 1) return LDC_E_ARG;
  const int R = (Mi + 15) / 16;
  int e;
  // X = F Qyinv^T, stored transposed:           w0[j][i] = sum_k F[i][k] Qyinv[j][k]
  if ((e = ldc_gemm_nt(F, Qyinv, w0, R, R, LD, 1, 0, nullptr, nullptr, stream)) != 0) return e;
  // Phat = Qxinv X / (lamx_i + lamy_j):         w1[i][j] = sum_k Qxinv[i][k] w0[j][k]
  if ((e = ldc_gemm_nt(Qxinv, w0, w1, R, R, LD, 0, 1, lamx, lamy, stream)) != 0) return e;
  // Y = Phat Qy^T, stored transposed:           w0[j][i] = sum_k w1[i][k] Qy[j][k]
  if ((e = ldc_gemm_nt(w1, Qy, w0, R, R, LD, 1, 0, nullptr, nullptr, stream)) != 0) return e;
  // Psi = Qx Y:                                 Psi[i][j] = sum_k Qx[i][k] w0[j][k]
  return ldc_gemm_nt(Qx, w0, Psi, R, R, LD, 0, 0, nullptr, nullptr, stream);
}

int ldc_vortex_extrema_xy(const double* Psi, const double* W, const double* x, const double* y, int Mx, int My, int LD,
                          double* out_val, int32_t* out_idx, void* stream) {
  if (!Psi || !W || !x || !y || !out_val || !out_idx || Mx < 2 || My < 2 || LD < Mx || LD < My) return LDC_E_ARG;
  hipLaunchKernelGGL(extrema_kernel, dim3(1), dim3(1024), 0, as_stream(stream), Psi, W, x, y, Mx, My, LD, out_val, out_idx);
  return (int)hipGetLastError();
}
int ldc_vortex_extrema(const double* Psi, const double* W, const double* x, const double* y, int M, int LD,
                       double* out_val, int32_t* out_idx, void* stream) {
  return ldc_vortex_extrema_xy(Psi, W, x, y, M, M, LD, out_val, out_idx, stream);
}

int ldc_mfma_selftest(const double* A, const double* B, double* D, void* stream) {
  if (!A || !B || !D) return LDC_E_ARG;
  hipLaunchKernelGGL(mfma_selftest_kernel, dim3(1), dim3(64), 0, as_stream(stream), A, B, D);
  return (int)hipGetLastError();
}

int ldc_mfma_peak(double* sink, int iters, int grid, void* stream) {
  if (!sink || iters < 1 || grid < 1) return LDC_E_ARG;
  hipLaunchKernelGGL(mfma_peak_kernel, dim3(grid), dim3(kThreads), 0, as_stream(stream), sink, iters);
  return (int)hipGetLastError();
}

int ldc_stream_priority_range(int* least, int* greatest) {
  if (!least || !greatest) return LDC_E_ARG;
  HIP_TRY(hipDeviceGetStreamPriorityRange(least, greatest));
  return 0;
}

int ldc_stream_create(int priority, void** stream) {
  if (!stream) return LDC_E_ARG;
  hipStream_t st = nullptr;
  {
    std::lock_guard<std::mutex> lock(g_setup_mutex);
    HIP_TRY(hipStreamCreateWithPriority(&st, hipStreamNonBlocking, priority));
  }
  *stream = st;
  return 0;
}

int ldc_stream_destroy(void* stream) {
  if (!stream) return LDC_E_ARG;
  std::lock_guard<std::mutex> lock(g_setup_mutex);
  HIP_TRY(hipStreamDestroy(as_stream(stream)));
  return 0;
}

}  // extern "C"
